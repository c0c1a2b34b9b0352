"""GPU: BASELINE.json configs[2] (512x512x128 + early termination) and configs[3] (1024x1024x64, rays sharded over 8 ranks)
at full size: an oracle-checked ray sample (>= 768 rays), size-independent invariants over every ray, and shard invariance."""
import importlib

import numpy as np
import pytest
import torch

from golden_cases import assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-4          # north_star: max-abs on rgb / depth against the reference CPU path


@pytest.fixture(scope="module")
def fm():
    return importlib.import_module("gp-nerf_amd.frame")


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def build_frame(fm, sc):
    return fm.Frame(to_dev(sc["src_imgs"][0]), to_dev(sc["featmaps"]), [to_dev(v) for v in sc["volumes"]], to_dev(sc["src_Ks"][0]),
                    to_dev(sc["src_poses"][0]), sc["Rh"][0], sc["Th"][0], sc["bounds"][0, 0], sc["voxel_size"], sc["out_sh"][0],
                    fm.pack_head(sc["head"], torch.device("cuda:0")))


def cpu(d):
    return {k: v.cpu().numpy() for k, v in d.items()}


def check_invariants(got, S):
    w, zv = got["weights"], got["z_vals"]
    assert (w >= 0).all() and np.abs(w.sum(1) - got["acc_map"]).max() < 2e-5
    assert got["acc_map"].max() <= 1 + 1e-5
    assert (np.diff(zv, axis=1) >= 0).all(), "z_vals must be sorted front to back"
    assert np.abs((w * zv).sum(1) - got["depth_map"]).max() < 1e-4
    assert np.isfinite(got["rgb_map"]).all() and got["rgb_map"].min() >= 0 and got["rgb_map"].max() <= 1 + 1e-5


# ---- configs[2]: 512x512 frame, 128 samples per ray, early termination ------------------------------------------------
@pytest.mark.parametrize("split_f16", [False, True])
def test_config3_512x512x128_early_termination_against_the_oracle(split_f16, fm, oracle, syn):
    """The early-terminated render is compared with the ORACLE's full (never terminated) render, so the bound covers the
    kernel's own error plus what termination drops: rgb / acc lose at most T_stop <= term_eps, depth at most
    term_eps * far (far ~ 3.9 here) -- term_eps = 1e-5 keeps the sum inside 1e-4."""
    S, eps = 128, 1e-5
    sc = syn.make_scene(H=512, W=512, seed=0, fill="full", pose="identity", sigma_bias=1.0)
    fr = build_frame(fm, sc)
    rays_h = oracle.rays_of(sc)
    assert rays_h.shape[0] == 512 * 512 and float(rays_h[:, 7].max()) * eps < 5e-5
    rays = to_dev(rays_h)
    want = ("weights", "z_vals", "rgb_in", "ray_mask", "samples_done")
    cut = cpu(fm.render_fused(fr, rays, S, early_term=True, term_eps=eps, want=want, split_f16=split_f16))
    # (1) oracle on 1024 rays spread over the frame, without termination
    idx = np.linspace(0, rays_h.shape[0] - 1, 1024).astype(np.int64)
    ref = oracle.render(sc, S, rays=rays_h[idx])
    for k in ("rgb_map", "depth_map", "acc_map"):
        assert_close(cut[k][idx], ref[k], TOL, f"config 3 {k} vs oracle")
    assert_close(cut["rgb_in_map"][idx], ref["rgb_in_map"], TOL, "config 3 rgb_in_map vs oracle")
    # (2) termination really fires, and only ever drops samples whose weight the oracle puts below the threshold
    done = cut["samples_done"]
    assert done.min() >= 1 and done.max() <= S
    frac = done.astype(np.float64).mean() / S
    assert frac < 0.7, f"early termination evaluated {frac:.2f} of the samples: the scene is not opaque enough"
    for j, i in enumerate(idx):
        assert ref["weights"][j, done[i]:].sum() <= eps * 1.01
    assert np.abs(cut["weights"][idx] - ref["weights"]).max() < 2e-5
    # (2b) the oracle with the same per-ray rule (a ray stops at the first sample at whose start its T is below the threshold):
    # same stopping sample -- except where T sits within rounding of the threshold -- and the same maps, ray_mask included
    ref_et = oracle.render(sc, S, rays=rays_h[idx], term_eps=eps)
    same = done[idx] == ref_et["samples_done"]
    assert same.mean() > 0.98, same.mean()
    for k in ("rgb_map", "depth_map", "acc_map", "rgb_in_map", "weights"):
        assert_close(cut[k][idx], ref_et[k], 2e-5, f"config 3 {k} vs the oracle's per-ray termination")
    assert np.array_equal(cut["ray_mask"][idx][same].astype(bool), ref_et["ray_mask"][same].astype(bool))
    # (3) invariants over all 262 144 rays; every ray stops on its own: after the first sample that takes ITS transmittance
    # below the threshold (read off the unterminated launch's weights)
    check_invariants(cut, S)
    full_w = fm.render_fused(fr, rays, S, want=("weights",), split_f16=split_f16)["weights"].double()
    T = torch.cat([torch.ones((full_w.shape[0], 1), dtype=torch.float64, device=full_w.device), 1.0 - torch.cumsum(full_w, 1)], 1)   # T[:, k] = after k samples
    d = torch.from_numpy(done.astype(np.int64)).to(T.device)
    noise = 5e-7                                    # 1 - sum of fp32 weights against the kernel's running product
    t_stop = T.gather(1, d[:, None])[:, 0]          # transmittance after the samples the ray evaluated
    t_prev = T.gather(1, (d - 1)[:, None])[:, 0]    # ... and one sample earlier: still at or above the threshold
    assert bool(((t_stop < eps + noise) | (d == S)).all()) and bool((t_prev >= eps - noise).all())
    assert frac < 0.3, f"per-ray termination evaluated {frac:.2f} of the samples"
    # (4) a ray's result is a function of that ray alone, so any order of the rays gives the same bits per ray (the launch
    # packs whichever rays are still alive 32 to a wavefront, segment after segment) ...
    g = torch.Generator(device="cpu").manual_seed(1)
    perm = torch.randperm(rays.shape[0], generator=g).to(rays.device)
    full = fm.render_fused(fr, rays, S, early_term=True, term_eps=eps, split_f16=split_f16, want=("weights", "samples_done"))
    shuf = fm.render_fused(fr, rays[perm].contiguous(), S, early_term=True, term_eps=eps, split_f16=split_f16, want=("weights", "samples_done"))
    for k in ("rgb_map", "depth_map", "acc_map", "weights", "samples_done"):
        assert torch.equal(shuf[k], full[k][perm]), k
    # ... and a shard of the rays too small for the chained form (tile-granular termination: a tile goes on until its last ray
    # is opaque) stays within what termination is allowed to drop
    for a, b in ((0, 4096), (32 * 4001, 32 * 4001 + 555), (262144 - 64, 262144)):
        part = fm.render_fused(fr, rays[a:b], S, early_term=True, term_eps=eps, split_f16=split_f16, want=("weights",))
        for k in ("rgb_map", "depth_map", "acc_map", "weights"):
            assert float((part[k] - full[k][a:b]).abs().max()) <= 5e-5, (k, a, b)


def test_config3_without_termination_matches_the_oracle_at_128_samples(fm, oracle, syn):
    sc = syn.make_scene(H=512, W=512, seed=0, fill="full", pose="identity", sigma_bias=1.0)
    fr = build_frame(fm, sc)
    rays_h = oracle.rays_of(sc)
    idx = np.linspace(0, rays_h.shape[0] - 1, 768).astype(np.int64)
    got = cpu(fm.render_fused(fr, to_dev(rays_h[idx]), 128))
    ref = oracle.render(sc, 128, rays=rays_h[idx])
    for k in ("rgb_map", "depth_map", "acc_map", "weights", "rgb_in_map"):
        assert_close(got[k], ref[k], 2e-5, k)


# ---- configs[3]: 1024x1024 frame, 64 samples per ray, rays sharded over 8 ranks -----------------------------------------
def test_config4_1024x1024x64_and_its_8_way_sharding(fm, oracle, syn):
    par = importlib.import_module("gp-nerf_amd.parallel")
    S = 64
    sc = syn.make_scene(H=1024, W=1024, seed=0, fill="full", pose="identity")
    fr = build_frame(fm, sc)
    rays_h = oracle.rays_of(sc)
    n = rays_h.shape[0]
    assert n == 1024 * 1024
    rays = to_dev(rays_h)
    out = fm.render_fused(fr, rays, S)
    got = cpu(out)
    # (1) oracle on 768 rays spread over the frame
    idx = np.linspace(0, n - 1, 768).astype(np.int64)
    ref = oracle.render(sc, S, rays=rays_h[idx])
    for k in ("rgb_map", "depth_map", "acc_map", "weights", "rgb_in_map"):
        assert_close(got[k][idx], ref[k], TOL, f"config 4 {k} vs oracle")
    assert np.array_equal(got["ray_mask"][idx].astype(bool), ref["ray_mask"].astype(bool))
    # (2) invariants over all 1 048 576 rays
    check_invariants(got, S)
    # (3) the 8-rank partition of parallel.py renders the same frame: every rank's share through the same entry point,
    # re-assembled as the all-gather would, bit-exact against the single-launch frame
    plan = par.ShardPlan(n, 8, torch.device("cuda:0"))
    packed = torch.empty((8, plan.share, 4), device="cuda:0")
    for r in range(8):
        mine = plan.take(rays, r)
        assert mine.shape[0] == plan.share
        o = fm.render_fused(fr, mine, S, want=())
        packed[r] = par.pack_pixels(o)
    full = plan.unpermute(packed.view(-1, 4))
    assert full.shape == (n, 4)
    assert torch.equal(full[:, :3], out["rgb_map"]) and torch.equal(full[:, 3], out["depth_map"])
    # (4) the split-precision form at this size
    fast = cpu(fm.render_fused(fr, rays, S, split_f16=True, want=()))
    assert np.abs(fast["rgb_map"] - got["rgb_map"]).max() < 2e-5 and np.abs(fast["depth_map"] - got["depth_map"]).max() < 1e-4


# ---- the reference's own output at BASELINE.json's full sizes ------------------------------------------------------------
def _full_names():
    from golden_cases import full_size_case_names
    return full_size_case_names()


@pytest.mark.parametrize("name", _full_names())
def test_full_size_frames_match_the_reference_itself(name, fm):
    """configs[1] (512x512x64), configs[2] (512x512x128) and configs[3] (1024x1024x64) on the scenes bench.py renders, against what
    the REFERENCE's Renderer.render produced for every 64th / 256th ray (tests/golden/make_golden.py FULL_CASES; 262 144 - 1 048 576
    rays through libs/renders/BaseRender.py on the CPU): the three kernel forms (reference order = the default, folded, split f16), and for configs[2] the early-terminated render too
    (its bound: the kernel's error + what termination drops, term_eps = 1e-5)."""
    from golden_cases import load, scene_of
    z, meta = load(name)
    sc = scene_of(meta)
    S, st = meta["n_samples"], int(z["ray_stride"])
    fr = build_frame(fm, sc)
    rays = to_dev(np.concatenate([sc["ray_o"][0], sc["ray_d"][0], sc["near"][0][:, None], sc["far"][0][:, None]], 1).astype(np.float32))
    assert rays.shape[0] == meta["n_rays"]
    worst = {}
    for label, kw in (("fp32", {}), ("fp32-folded", {"fold": True}), ("split", {"split_f16": True})) + ((("fp32+early-term", {"early_term": True, "term_eps": 1e-5}),) if S == 128 else ()):
        got = cpu(fm.render_fused(fr, rays, S, want=("rgb_in",), **kw))
        for k in ("rgb_map", "depth_map", "acc_map") + (("rgb_in_map",) if "early" not in label else ()):
            worst[(label, k)] = assert_close(got[k][::st], z[k], TOL, f"{name} {label} {k}")
    print(name, {f"{a}:{k}": float(f"{v:.2e}") for (a, k), v in worst.items()})
