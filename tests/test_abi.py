"""CPU: the C-ABI library loads and exports every symbol include/gpnerf_hip.h declares; struct layouts agree."""
import ctypes as C
import importlib
import os
import re
import subprocess
import sys
import tempfile

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
