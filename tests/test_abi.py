"""CPU: the C-ABI library loads and exports every symbol include/gpnerf_hip.h declares; struct layouts agree."""
import ctypes as C
import importlib
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gpnerf_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpnerf_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(pkg):
    names = declared_symbols()
    assert len(names) >= 15
    lib = pkg._lib.lib()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gpnerf_hip.h but not exported"
        assert n in pkg._lib.SYMBOLS, f"{n} has no ctypes signature in _lib.SYMBOLS"
    assert sorted(pkg._lib.SYMBOLS) == names


def test_nothing_is_exported_that_the_header_lacks():
    """The converse: the product library's dynamic symbol table holds NO gpnerf_* entry beyond include/gpnerf_hip.h -- no
    gpnerf_debug_* reader of a diagnostic build, no experiment entry point.  The lab's libraries are other files,
    built by gp-nerf_amd/csrc/diag/Makefile."""
    import shutil
    nm = shutil.which("nm") or shutil.which("llvm-nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.check_output([nm, "-D", "--defined-only", os.path.join(ROOT, "gp-nerf_amd", "csrc", "libgpnerf_hip.so")], text=True)
    exported = sorted({l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-1].startswith("gpnerf_")})
    assert exported == declared_symbols(), sorted(set(exported) ^ set(declared_symbols()))


def test_the_product_sources_carry_no_lab_paths():
    """No experiment switch in the product's translation units: no GPNERF_X_* path, no stamps / wavetimes code, no getenv -- the
    kernels' few hook points resolve to the EMPTY definitions of csrc/nodiag/gpnerf_diag.h, and the product Makefile refuses any
    -DGPNERF_* switch (__graft_entry__.build() checks both before it compiles)."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    assert g.lab_paths_in_product() == []
    csrc = os.path.join(ROOT, "gp-nerf_amd", "csrc")
    hooks = open(os.path.join(csrc, "nodiag", "gpnerf_diag.h")).read()
    code = re.sub(r"//[^\n]*", "", hooks)
    assert "getenv" not in code and "atomicAdd" not in code and "s_memtime" not in code
    r = subprocess.run(["make", "-n", "-C", csrc, "HIPFLAGS=-O3 -DGPNERF_X_ANYTHING"], capture_output=True, text=True)
    assert r.returncode != 0 and "takes no -DGPNERF_" in (r.stderr + r.stdout)


def test_struct_layouts_match_the_header(pkg):
    """Compile the header with gcc and compare sizeof/offsetof with the ctypes mirrors."""
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "gpnerf_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(GpnerfFrame), offsetof(GpnerfFrame, vol_dhw), offsetof(GpnerfFrame, featmaps),
         offsetof(GpnerfFrame, proj), offsetof(GpnerfFrame, out_sh), offsetof(GpnerfFrame, head_blob), offsetof(GpnerfFrame, imgs));
  printf("%zu %zu\n", offsetof(GpnerfFrame, occ), offsetof(GpnerfFrame, head_blob_split));
  printf("%zu %zu\n", sizeof(GpnerfHeadParams), sizeof(GpnerfOutputs));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        a, occ_off, b = subprocess.check_output([exe]).decode().strip().split("\n")
    L = pkg._lib
    F = L.GpnerfFrame
    assert [int(x) for x in a.split()] == [C.sizeof(F), F.vol_dhw.offset, F.featmaps.offset, F.proj.offset, F.out_sh.offset,
                                           F.head_blob.offset, F.imgs.offset]
    assert [int(x) for x in occ_off.split()] == [F.occ.offset, F.head_blob_split.offset]
    assert [int(x) for x in b.split()] == [C.sizeof(L.GpnerfHeadParams), C.sizeof(L.GpnerfOutputs)]


def test_host_only_entry_points_run_without_a_gpu(pkg):
    lib = pkg._lib.lib()
    assert lib.gpnerf_rays_per_tile() == 32
    n = lib.gpnerf_head_blob_floats()
    assert 0 < n * 4 <= 160 * 1024, "the head image must fit one CU's LDS"
    assert b"gfx950" in lib.gpnerf_build_info()
    assert lib.gpnerf_strerror(0) == b"ok" and lib.gpnerf_strerror(-1) == b"invalid argument"
    assert lib.gpnerf_pack_head(None, None) == -1
    t = (C.c_int32 * 48)()
    assert lib.gpnerf_head_layout(t) == 0
    assert t[0] == 64 and t[1] == 2 and t[2] == 0


def test_the_workspace_covers_the_colour_list_at_its_worst(pkg):
    """gpnerf_render_workspace_bytes (host arithmetic only): a launch that lists the samples whose colour branch has to run needs an
    entry and a result per SAMPLE at worst (every weight non-zero: 16 + 16 bytes), a padded unit per visit of a work unit on top and
    a flag per unit (include/gpnerf_hip.h `workspace`); launches beyond 2^26 samples or of less than a round of wavefronts do not
    list.  It grows with the launch and is positive for every launch that can use a tile queue."""
    lib = pkg._lib.lib()
    ws = lambda n, s: int(lib.gpnerf_render_workspace_bytes(n, s))
    n, s = 512 * 512, 64
    assert ws(n, s) >= 32 * n * s + 16 * 8 * n, "room for the list at full occupancy plus eight padded units per 32-ray tile"
    assert ws(n, s) < 40 * n * s, "... and not much more than that"
    assert ws(1024 * 1024, 128) < 16 * 1024 * 1024 * 128, "2^27 samples: no list (the wavefronts keep their passes)"
    assert ws(2048, 64) < 32 * 2048 * 64, "a launch of one round of workgroups at most never lists"
    assert 0 < ws(4096, 64) <= ws(65536, 64) <= ws(n, s)
    assert ws(0, 64) == 0


def test_missing_library_fails_loudly(pkg, monkeypatch):
    L = pkg._lib
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libgpnerf_hip.so")
    import pytest
    with pytest.raises(L.GpnerfError, match="no fallback"):
        L.lib()


def test_cpu_tensors_are_refused(pkg):
    import pytest
    import torch
    fm = importlib.import_module("gp-nerf_amd.frame")
    with pytest.raises(pkg.GpnerfError, match="no CPU fallback"):
        fm.render_fused(None, torch.zeros(4, 8), 8)


def _frame(L, dhw=(8, 8, 8), img=(16, 16), feat=(4, 4)):
    """A GpnerfFrame whose pointers are non-null dummies: argument checks run on the host before any launch."""
    f = L.GpnerfFrame()
    for l in range(L.LEVELS):
        f.vol[l] = 0x1000
        for a in range(3):
            f.vol_dhw[l][a] = dhw[a]
    f.featmaps, f.feat_h, f.feat_w = 0x1000, feat[0], feat[1]
    f.imgs, f.img_h, f.img_w = 0x1000, img[0], img[1]
    f.head_blob = 0x1000
    return f


def test_render_fused_rejects_bad_arguments_on_the_host(pkg):
    """gpnerf_render_fused returns GPNERF_E_ARG (never launches, never throws) for frames its 32-bit / 24-bit tap
    addressing cannot cover and for missing tensors; an empty ray list is a no-op."""
    L = pkg._lib
    lib = L.lib()
    out = L.GpnerfOutputs()
    out.rgb = out.depth = out.acc = out.disp = 0x1000

    def call(frame, n_rays=64, n_samples=8, rays=0x1000, o=out):
        return lib.gpnerf_render_fused(C.byref(frame), rays, n_rays, n_samples, 0, 1e-4, None, C.byref(o), None, 0, None)

    assert call(_frame(L), n_rays=0) == 0                                   # nothing to do
    assert call(_frame(L), n_samples=0) == -1
    assert call(_frame(L), rays=None) == -1
    assert call(_frame(L), o=L.GpnerfOutputs()) == -1                       # rgb / depth / acc / disp are required
    assert call(_frame(L, dhw=(4096, 4096, 8))) == -1                       # D*H >= 2^24
    assert call(_frame(L, dhw=(8, 8, 1 << 17))) == -1                       # one x-row >= 2^24 bytes
    assert call(_frame(L, dhw=(1024, 1024, 1024))) == -1                    # level >= 4 GiB
    assert call(_frame(L, img=(8, 1 << 20))) == -1                          # image row >= 2^24 bytes
    assert call(_frame(L, feat=(1 << 12, 1 << 13))) == -1                   # feature maps >= 4 GiB per view
    f = _frame(L)
    f.vol[2] = None
    assert call(f) == -1
    f = _frame(L)
    f.head_blob = None
    assert call(f) == -1
    assert lib.gpnerf_strerror(-1) == b"invalid argument"


def test_fold_volumes_rejects_bad_arguments_on_the_host(pkg):
    """gpnerf_fold_volumes checks its arguments before it touches the device: the frame, its head image, the coarse levels
    (GPNERF_FOLD_FIRST_LEVEL..) and their output pointers; the finer levels' entries are not looked at."""
    L = pkg._lib
    lib = L.lib()
    outs = (C.c_void_p * L.LEVELS)(None, None, 0x1000, 0x1000)
    assert lib.gpnerf_fold_volumes(None, outs, None) == -1
    assert lib.gpnerf_fold_volumes(C.byref(_frame(L)), None, None) == -1
    f = _frame(L)
    f.head_blob = None
    assert lib.gpnerf_fold_volumes(C.byref(f), outs, None) == -1
    f = _frame(L)
    f.vol[L.FOLD_FIRST_LEVEL] = None
    assert lib.gpnerf_fold_volumes(C.byref(f), outs, None) == -1
    assert lib.gpnerf_fold_volumes(C.byref(_frame(L)), (C.c_void_p * L.LEVELS)(0x1000, 0x1000, 0x1000, None), None) == -1
    assert L.FOLD_FIRST_LEVEL == 2 and "GPNERF_FOLD_FIRST_LEVEL 2" in open(os.path.join(ROOT, "include", "gpnerf_hip.h")).read()


def _isa_tool():
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_mfma_hazards", os.path.join(ROOT, "tools", "isa_mfma_hazards.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_no_consumer_sits_inside_an_mfma_shadow_in_the_shipped_code():
    """gfx950 does not interlock a VALU / memory instruction that touches an MFMA's destination VGPRs before the MFMA has
    written them; LLVM pads such consumers with s_nop -- except `asm()` statements, which it cannot see into (the kernels
    convert operands with inline v_fma_mixlo/hi_f16).  tools/micro/mfma_asm_hazard.hip shows the failure on hardware (65 535 of
    65 536 results wrong); tools/isa_mfma_hazards.py walks the built library's ISA and must find no such pair."""
    import shutil
    if not (shutil.which("llvm-objdump") or os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump")):
        pytest.skip("llvm-objdump not available")
    tool = _isa_tool()
    res = tool.scan(os.path.join(ROOT, "gp-nerf_amd", "csrc", "libgpnerf_hip.so"))
    assert len(res) >= 30 and sum(n for _, n, _, _ in res) > 10000, "the disassembly did not find the MFMA kernels"
    bad = [(name, b[:2]) for name, _, _, b in res if b]
    assert not bad, bad


def test_the_hazard_checker_flags_an_opaque_consumer(tmp_path):
    """The checker itself: on the micro kernel the asm consumer right behind the MFMA is reported, the compiler-visible consumer
    and the asm consumer behind `s_nop 11` are not."""
    import shutil
    import subprocess
    if not shutil.which("hipcc"):
        pytest.skip("hipcc not available")
    out = tmp_path / "hz.s"
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", str(out),
                           os.path.join(ROOT, "tools", "micro", "mfma_asm_hazard.hip")], stderr=subprocess.DEVNULL)
    res = {name: bad for name, _, _, bad in _isa_tool().scan(str(out))}
    by_mode = {m: next(b for n, b in res.items() if f"ILi{m}E" in n) for m in (0, 1, 2)}
    assert not by_mode[0] and not by_mode[2]
    assert len(by_mode[1]) == 1 and by_mode[1][0][5] is True and "v_add_f32" in by_mode[1][0][1]


def test_the_hazard_checker_flags_an_opaque_producer_in_front_of_its_mfma(tmp_path):
    """The converse pair (round 6, the root-cause class of round 5's wrong split-precision deferred colour passes): a VGPR written by an
    inline-asm VALU instruction and read as an MFMA source operand before the wait states gfx950 needs -- it does not interlock the
    pair (tools/micro/asm_producer_hazards.hip on the GPU: 129 032 of 131 072 lanes read the register's previous contents with no
    wait state, none with one; the fp32 MFMA behind a plain v_add_f32 needs two).  On the micro kernels' listing the checker must
    flag exactly the variants below the requirement: f16 MFMA consumers need 1 wait state, fp32 MFMA consumers are held to 2."""
    import shutil
    if not shutil.which("hipcc"):
        pytest.skip("hipcc not available")
    out = tmp_path / "aph.s"
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", str(out),
                           os.path.join(ROOT, "tools", "micro", "asm_producer_hazards.hip")], stderr=subprocess.DEVNULL)
    res = {name: [b for b in bad if b[5] == "producer"] for name, _, _, bad in _isa_tool().scan(str(out))}
    flagged = lambda case, wait: bool(next(b for n, b in res.items() if f"ILi{case}ELi{wait}E" in n))
    assert flagged(1, 0) and not flagged(1, 1) and not flagged(1, 2)                 # v_fma_mixhi_f16 -> v_mfma_f32_32x32x16_f16
    assert flagged(5, 0) and not flagged(5, 1)                                       # (plain v_mov_b32, written inside an asm block)
    assert flagged(3, 0) and flagged(3, 1) and not flagged(3, 2) and not flagged(3, 4)   # v_permlane32_swap_b32 -> v_mfma_f32_32x32x2_f32
    assert flagged(4, 1) and not flagged(4, 2)                                       # v_add_f32 -> the fp32 MFMA: two wait states (measured)


def _ticket_tool():
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_ticket_release", os.path.join(ROOT, "tools", "isa_ticket_release.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_every_waves_tile_sums_are_acknowledged_before_the_norm_ticket():
    """The fused InstanceNorm table (gpnerf_conv.hip finalize_if_last): the last workgroup of an (image, channel group) reads
    every other workgroup's write-through tile sums, so each wave must `s_waitcnt vmcnt(0)` before the barrier in front of the
    ticket atomic -- a workgroup-scope release fence does not emit that wait on gfx9 (round 3's build had it BEHIND the ticket:
    a timing-dependent stale read no golden vector can catch).  tools/isa_ticket_release.py walks the ISA the compiler emitted."""
    import shutil
    if not (shutil.which("llvm-objdump") or os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump")):
        pytest.skip("llvm-objdump not available")
    res = _ticket_tool().scan(os.path.join(ROOT, "gp-nerf_amd", "csrc", "libgpnerf_hip.so"))
    assert len(res) >= 15, "the disassembly did not find the convolution kernels with a norm ticket"
    bad = [(name, b) for name, _, b in res if b]
    assert not bad, bad


def test_the_ticket_checker_flags_an_unwaited_store():
    tool = _ticket_tool()
    good = ["global_store_dword v[20:21], v22, off sc1", "s_waitcnt vmcnt(0)", "s_waitcnt lgkmcnt(0)", "s_barrier",
            "global_atomic_add v20, v20, v21, s[12:13] sc0"]
    late = ["global_store_dword v[20:21], v22, off sc1", "s_waitcnt lgkmcnt(0)", "s_barrier",
            "global_atomic_add v20, v20, v21, s[12:13] sc0", "s_waitcnt vmcnt(0)"]
    none = ["global_store_dword v[20:21], v22, off sc1", "s_waitcnt vmcnt(0)", "global_atomic_add v20, v20, v21, s[12:13] sc0"]
    rows = lambda ls: list(enumerate(ls, 1))
    assert tool.check(rows(good)) == (1, [])
    assert len(tool.check(rows(late))[1]) == 1 and "not waited for" in tool.check(rows(late))[1][0][2]
    assert len(tool.check(rows(none))[1]) == 1 and "no s_barrier" in tool.check(rows(none))[1][0][2]
