"""GPU: the range guard of the split-precision form (GPNERF_FLAG_SPLIT_GUARD) and the edge cases of the early-termination
segmented form (one launch per 16-sample segment over the rays still alive).

The split form writes every fp32 MFMA operand as an f16 hi + lo pair, which only carries the value below the f16 range
(65504).  The guard flags the 32-ray tiles in which an operand reaches that range and renders them again in the fp32 form:
the result must not depend on the range of the data."""
import importlib
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def fm():
    return importlib.import_module("gp-nerf_amd.frame")


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def build_frame(fm, sc):
    return fm.Frame(to_dev(sc["src_imgs"][0]), to_dev(sc["featmaps"]), [to_dev(v) for v in sc["volumes"]], to_dev(sc["src_Ks"][0]),
                    to_dev(sc["src_poses"][0]), sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0],
                    fm.pack_head(sc["head"], torch.device("cuda:0")))


def rays_of(sc):
    return to_dev(np.concatenate([sc["ray_o"][0], sc["ray_d"][0], sc["near"][0][:, None], sc["far"][0][:, None]], 1).astype(np.float32))


WANT = ("weights", "z_vals", "rgb_in", "ray_mask", "guard_tiles")
KEYS = ("rgb_map", "depth_map", "acc_map", "weights", "rgb_in_map")


def overflow_scene(syn, size, seed, what):
    """A scene whose MFMA operands leave the f16 range in part of the image: huge source-view features (raw inputs and the
    cross-view variance overflow) or a huge volume level (the sigma-feature layer's inputs and its activations overflow)."""
    sc = syn.make_scene(H=size, W=size, seed=seed, fill="full", pose="identity")
    sc = dict(sc)
    if what == "featmaps":
        f = sc["featmaps"].copy()
        f[:, :, : f.shape[2] // 3] *= 3.0e5
        sc["featmaps"] = f
    else:
        v = [a.copy() for a in sc["volumes"]]
        v[1][..., : v[1].shape[-2] // 3, : v[1].shape[-1] // 3] *= 2.0e5
        sc["volumes"] = v
    return sc


@pytest.mark.parametrize("what", ["featmaps", "volume"])
@pytest.mark.parametrize("size,S,early", [(64, 32, False), (200, 48, False), (512, 64, False), (512, 128, True)])
def test_guarded_split_form_does_not_depend_on_the_range_of_the_data(what, size, S, early, fm, syn):
    sc = overflow_scene(syn, size, 5, what)
    fr, rays = build_frame(fm, sc), rays_of(sc)
    kw = dict(early_term=early, term_eps=1e-5)
    # the fp32 form as the fix-up launch runs it (fold=False: every level through the sigma feature layer per sample; on values of
    # 1e5 the folded coarse levels' different summation order alone moves a weight by 1e-3)
    ref = fm.render_fused(fr, rays, S, want=WANT, fold=False, **kw)
    bad = fm.render_fused(fr, rays, S, want=WANT, split_f16=True, guard=False, **kw)
    got = fm.render_fused(fr, rays, S, want=WANT, split_f16=True, **kw)                    # guard on by default
    n_tiles = (rays.shape[0] + 31) // 32
    flagged = int(got["guard_tiles"])
    assert 0 < flagged <= n_tiles and (flagged < n_tiles or size < 200), (flagged, n_tiles)
    assert int(ref["guard_tiles"]) == 0 and int(bad["guard_tiles"]) == 0
    # without the guard the split form is simply wrong on this data ...
    assert float((bad["rgb_map"] - ref["rgb_map"]).abs().max()) > 1e-2
    # ... with it, every map is the fp32 form's to the usual bound
    for k in KEYS:
        assert float((got[k] - ref[k]).abs().max()) < TOL, k
    # (ray_mask counts the evaluated samples with two valid views: under early termination it depends on where a ray stops)
    assert torch.equal(got["z_vals"], ref["z_vals"]) and (early or torch.equal(got["ray_mask"], ref["ray_mask"]))
    if early:
        return      # the fix-up launch terminates per tile, the chained launch per ray: equal to the bound above, not to the bit
    # and the flagged tiles ARE the fp32 form's, bit for bit: the fix-up launch runs the same code on the same 32 rays, one
    # wavefront per whole ray (load_balance=False keeps the reference launch from splitting a small frame's samples over
    # several wavefronts, which re-associates the composite; fold=False: the fix-up launch runs the sigma feature layer per
    # sample, as the split form does, not on the folded coarse levels)
    whole = fm.render_fused(fr, rays, S, load_balance=False, fold=False, **kw)
    diff = (got["rgb_map"] != whole["rgb_map"]).any(1)
    pad = torch.zeros(n_tiles * 32, dtype=torch.bool, device=diff.device)
    pad[: diff.numel()] = diff
    assert int((~pad.view(n_tiles, 32).any(1)).sum()) >= flagged


def test_guard_is_silent_and_free_of_side_effects_on_ordinary_data(fm, syn):
    sc = syn.make_scene(H=96, W=96, seed=2, fill="full", pose="random")
    fr, rays = build_frame(fm, sc), rays_of(sc)
    a = fm.render_fused(fr, rays, 40, want=WANT, split_f16=True, guard=False)
    b = fm.render_fused(fr, rays, 40, want=WANT, split_f16=True, guard=True)
    assert int(b["guard_tiles"]) == 0
    for k in KEYS:
        assert float((a[k] - b[k]).abs().max()) < 1e-6, k


def test_guard_needs_the_workspace(fm, syn):
    sc = syn.make_scene(H=16, W=16, seed=1, fill="full", pose="identity")
    fr, rays = build_frame(fm, sc), rays_of(sc)
    L = importlib.import_module("gp-nerf_amd._lib")
    with pytest.raises(L.GpnerfError):
        fm.render_fused(fr, rays, 8, split_f16=True, guard=True, load_balance=False)
    fm.render_fused(fr, rays, 8, split_f16=True, load_balance=False)          # default: no workspace, no guard


@pytest.mark.parametrize("split_f16", [False, True])
@pytest.mark.parametrize("n_rays,S", [(65536 + 17, 80), (70001, 33), (131072 + 31, 64), (65536, 4000)])
def test_early_termination_in_segments_on_ragged_sizes(n_rays, S, split_f16, fm, syn):
    """Frames of at least one full round of wavefronts walk their samples in segments, one launch per segment over the rays
    still alive, 32 to a wavefront (per-ray termination).  Ragged ray counts, sample counts that are not a multiple of the
    segment length and more segments than the default length allows: the result is the same bits in any ray order, and within
    what termination may drop of launches too small for the chained form (which stop a 32-ray tile as a whole)."""
    size = 64 if S > 1000 else 96
    sc = syn.make_scene(H=size, W=size, seed=4, fill="full", pose="identity", sigma_bias=1.0)
    fr = build_frame(fm, sc)
    base = rays_of(sc)
    rays = base[torch.arange(n_rays, device=base.device) % base.shape[0]].contiguous()
    want = ("weights", "z_vals", "rgb_in", "ray_mask", "samples_done")
    kw = dict(early_term=True, term_eps=1e-5, want=want, split_f16=split_f16, guard=False)
    chained = fm.render_fused(fr, rays, S, **kw)
    keys = ("rgb_map", "depth_map", "acc_map", "disp_map", "weights", "z_vals", "rgb_in_map", "ray_mask", "samples_done")
    perm = torch.randperm(n_rays, generator=torch.Generator().manual_seed(n_rays)).to(rays.device)
    shuffled = fm.render_fused(fr, rays[perm].contiguous(), S, **kw)
    for k in keys:
        assert torch.equal(shuffled[k], chained[k][perm]), k
    # the rays repeat with period size*size: copies of a ray have the same bits wherever they sit in the launch
    period = base.shape[0]
    assert torch.equal(chained["rgb_map"][:n_rays - period], chained["rgb_map"][period:])
    parts = [fm.render_fused(fr, rays[a:a + 32768], S, **kw) for a in range(0, n_rays, 32768)]
    for k in ("rgb_map", "depth_map", "acc_map", "weights", "rgb_in_map"):
        assert float((chained[k] - torch.cat([p[k] for p in parts])).abs().max()) < 5e-5, k
    assert torch.equal(chained["z_vals"], torch.cat([p["z_vals"] for p in parts]))
    done, done_tiles = chained["samples_done"], torch.cat([p["samples_done"] for p in parts])
    assert int(done.min()) >= 1 and int(done.max()) <= S and bool((done <= done_tiles).all())
    assert float(done.float().mean()) < 0.9 * float(done_tiles.float().mean())


@pytest.mark.parametrize("sigma_bias", [-20.0, 12.0])
def test_early_termination_in_segments_when_nothing_or_everything_terminates(sigma_bias, fm, syn):
    """The two ends: a density that never makes a ray opaque (every ray is on every segment's list, and the result is the
    unterminated launch's, bit for bit), and one that makes every ray opaque within the first segment (every later launch finds
    its list empty)."""
    S = 96
    sc = syn.make_scene(H=96, W=96, seed=6, fill="full", pose="identity", sigma_bias=sigma_bias)
    fr = build_frame(fm, sc)
    base = rays_of(sc)
    n_rays = 65536 + 96
    rays = base[torch.arange(n_rays, device=base.device) % base.shape[0]].contiguous()
    want = ("weights", "z_vals", "rgb_in", "samples_done")
    chained = fm.render_fused(fr, rays, S, early_term=True, term_eps=1e-5, want=want)
    plain = fm.render_fused(fr, rays, S, early_term=False, want=want)
    done = chained["samples_done"]
    if sigma_bias < 0.0:
        assert int(done.min()) == S
        for k in ("rgb_map", "depth_map", "acc_map", "disp_map", "weights", "z_vals", "rgb_in_map"):
            assert torch.equal(chained[k].view(torch.int32), plain[k].view(torch.int32)), k      # bit patterns: disp is NaN where acc = 0
    else:
        assert int(done.max()) <= 16
        for k in ("rgb_map", "depth_map", "acc_map"):
            assert float((chained[k] - plain[k]).abs().max()) < 1e-4, k


@pytest.mark.parametrize("split_f16", [False, True])
@pytest.mark.parametrize("n_rays,S", [(65536 + 7, 24), (65536 + 255 * 32 + 1, 64), (65536 + 1000 * 32, 33), (2 * 65536 + 40 * 32 + 31, 16)])   # the first and the last take the remainder launch
def test_a_frames_last_partial_round_runs_several_samples_per_step(n_rays, S, split_f16, fm, syn):
    """A frame of whole rounds of wavefronts plus a few tiles: the few go to the segmented form's kernel as one segment of all S
    samples, 2 / 4 / 8 samples of a ray side by side.  Same bits as the launch without a workspace (one wavefront per tile, every
    sample in turn), for every output."""
    sc = syn.make_scene(H=64, W=64, seed=9, fill="full", pose="random")
    fr = build_frame(fm, sc)
    base = rays_of(sc)
    rays = base[torch.arange(n_rays, device=base.device) % base.shape[0]].contiguous()
    want = ("weights", "z_vals", "rgb_in", "ray_mask", "raw", "samples_done")
    a = fm.render_fused(fr, rays, S, want=want, split_f16=split_f16, guard=False)
    b = fm.render_fused(fr, rays, S, want=want, split_f16=split_f16, guard=False, load_balance=False)
    for k in a:
        x, y = a[k], b[k]
        if x.dtype == torch.float32:
            x, y = x.view(torch.int32), y.view(torch.int32)
        assert torch.equal(x, y), k
    g = fm.render_fused(fr, rays, S, want=want + ("guard_tiles",), split_f16=split_f16, guard=True) if split_f16 else None
    if g is not None:
        assert int(g["guard_tiles"]) == 0 and torch.equal(g["rgb_map"], a["rgb_map"])


def test_every_launch_form_runs_on_the_callers_stream(fm, syn):
    """The library enqueues on the stream it is handed (torch's current stream) and nowhere else: the same calls inside
    `torch.cuda.stream(side)` give the same bits, and the side stream alone orders them (the default stream is kept busy)."""
    sc = syn.make_scene(H=96, W=96, seed=11, fill="full", pose="identity", sigma_bias=1.0)
    fr = build_frame(fm, sc)
    base = rays_of(sc)
    rays = base[torch.arange(70000, device=base.device) % base.shape[0]].contiguous()
    calls = [dict(), dict(split_f16=True), dict(early_term=True, term_eps=1e-5), dict(early_term=True, term_eps=1e-5, split_f16=True)]
    ref = [fm.render_fused(fr, rays, 48, **kw) for kw in calls]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    busy = torch.empty((64, 1024, 1024), device=rays.device)
    with torch.cuda.stream(side):
        got = []
        for kw in calls:
            busy.normal_()                               # work on the DEFAULT stream would not order anything here
            got.append(fm.render_fused(fr, rays, 48, **kw))
    side.synchronize()
    for a, b in zip(ref, got):
        for k in ("rgb_map", "depth_map", "acc_map", "weights", "z_vals"):
            assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("kw", [dict(), dict(early_term=True, term_eps=1e-5), dict(split_f16=True), dict(early_term=True, term_eps=1e-5, split_f16=True)],
                         ids=["plain", "early_term", "split_guard", "early_term_split_guard"])
def test_a_calls_launch_sequence_captures_into_a_hip_graph(kw, fm, syn):
    """gpnerf_render_fused only enqueues (memsets + kernels; every length the later launches need -- ray lists of the segmented
    form, the guard's flag count -- stays on the device), so the whole call captures into a HIP graph and replays with the same
    bits, however often and whatever else ran in between: a caller with a launch-bound loop around it can take the host out of
    that loop.  (The counters a call starts from are zeroed by a kernel of the library's own: a captured hipMemsetAsync node was
    not reliably ordered before the kernel after it -- replays after the first found the queues exhausted.)  Two whole rounds
    of wavefronts, and a frame that ends with a remainder launch."""
    sc = syn.make_scene(H=96, W=96, seed=11, fill="full", pose="identity", sigma_bias=1.0)
    fr = build_frame(fm, sc)
    base = rays_of(sc)
    for n_rays in (131072, 70000):
        rays = base[torch.arange(n_rays, device=base.device) % base.shape[0]].contiguous()
        ref = fm.render_fused(fr, rays, 48, **kw)
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            fm.render_fused(fr, rays, 48, **kw)              # warm-up on the capture stream (allocator pool, kernel attributes)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = fm.render_fused(fr, rays, 48, **kw)
        first = None
        for _ in range(3):
            for v in out.values():
                v.zero_()
            g.replay()
            torch.cuda.synchronize()
            for k in ("rgb_map", "depth_map", "acc_map", "weights", "z_vals"):
                assert torch.equal(out[k], ref[k]), (n_rays, k)
            if first is None:
                first = {k: v.clone() for k, v in out.items()}
            for k in first:
                assert torch.equal(out[k], first[k]), (n_rays, k)          # every replay gives the same bits


def test_hardware_does_not_interlock_an_asm_consumer_of_an_mfma_result(tmp_path):
    """tools/micro/mfma_asm_hazard.hip on the device: an inline-asm VALU instruction issued right behind an MFMA reads the
    destination register BEFORE the MFMA has written it (no hardware interlock, no compiler padding for asm statements) -- the
    mechanism behind the run-to-run differences round 2 saw with a branch inside the split form's layer chain.  Behind an
    explicit `s_nop 11`, and for a compiler-visible consumer, the result is right.  The product's kernels are kept free of
    such pairs by tests/test_abi.py::test_no_consumer_sits_inside_an_mfma_shadow_in_the_shipped_code."""
    import shutil
    import subprocess
    if not shutil.which("hipcc"):
        pytest.skip("hipcc not available on this box")
    exe = tmp_path / "mfma_asm_hazard"
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-o", str(exe), os.path.join(ROOT, "tools", "micro", "mfma_asm_hazard.hip")],
                          stderr=subprocess.DEVNULL)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120).stdout
    print(out.strip())
    m = re.search(r"opaque consumer: (\d+) of (\d+) results differ.*behind s_nop 11: (\d+) differ", out)
    assert m, out
    assert int(m.group(3)) == 0, "with the wait states in place the asm consumer must see the MFMA's result"
    assert int(m.group(1)) > int(m.group(2)) // 2, "expected the unpadded asm consumer to read stale registers on gfx950"
