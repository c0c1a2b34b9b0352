"""Per-frame producers of the dense feature pyramid -- NOT on the per-ray path.

The reference builds the 4 dense levels with the external spconv v1.2.1 CUDA library
(libs/nerfheads/networks/SparseConvNet.py:22-124), which is neither in the reference tree nor installable here, so its
arithmetic cannot be pinned ("parity unpinned" at this boundary).  This module owns the reference's
parameters under the reference's names (so reference checkpoints load with strict=True) and computes on the GPU only:
the vertex-code attention is one HIP launch (gpnerf_vertex_attention), the sparse convolutions are gpnerf_volume.hip.
There is no CPU / PyTorch fallback; the torch-CPU restatements the tests check against live in oracle/producers_ref.py.

`MultiHeadAttention` mirrors libs/nerfheads/networks/MultiHeadAttention.py:42-98 (parameter names
w_qs, w_ks, w_vs, fc, layer_norm).
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L


class MultiHeadAttention(nn.Module):
    """Query = SMPL vertex code (length 1), keys/values = that vertex's features in the V source views."""

    def __init__(self, n_head, d_model, d_k, d_v, dropout=0.1, kv_dim=None, sum=True):
        super().__init__()
        self.n_head, self.d_k, self.d_v, self.sum_flag = n_head, d_k, d_v, sum
        kv_dim = d_model if kv_dim is None else kv_dim
        self.w_qs = nn.Linear(d_model, n_head * d_k, bias=False)
        self.w_ks = nn.Linear(kv_dim, n_head * d_k, bias=False)
        self.w_vs = nn.Linear(kv_dim, n_head * d_v, bias=False)
        self.fc = nn.Linear(n_head * d_v, d_model, bias=False)
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)  # owned for checkpoint parity; used only when sum=True

    def forward(self, q, k, v, mask=None):
        """MultiHeadAttention.forward as the renderers call it (trainhead.py:51, demo_render.py:143-146): q [N,1,d_model] vertex
        codes, k = v [N,V,kv_dim] the vertices' per-view features, no mask, sum=False -> (out [N,1,d_model], None).
        One HIP launch; the attention weights the reference returns second are not materialised (no caller reads them)."""
        if mask is not None or q.dim() != 3 or q.shape[1] != 1 or k is not v and not torch.equal(k, v):
            raise L.GpnerfError("the HIP attention serves the volume builder's call: q [N,1,d], k = v [N,V,kv], no mask")
        return self.fuse_vertices(q[:, 0], k).unsqueeze(1), None

    def fuse_vertices(self, code, feat):
        """The call the volume builder makes (trainhead.py:48-52): code [N,d_model], feat [N,V,kv_dim] -> [N,d_model],
        as ONE HIP launch (gpnerf_vertex_attention) instead of ~12 library calls.  GPU, inference, sum=False only."""
        if self.sum_flag or self.training or not code.is_cuda:
            raise L.GpnerfError("fuse_vertices is the inference form of the sum=False attention on the GPU")
        code, feat = code.contiguous().float(), feat.contiguous().float()
        n, d = code.shape
        out = torch.empty((n, d), device=code.device, dtype=torch.float32)
        w = [m.weight.detach().contiguous().float() for m in (self.w_qs, self.w_ks, self.w_vs, self.fc)]
        L.check(L.lib().gpnerf_vertex_attention(code.data_ptr(), feat.data_ptr(), w[0].data_ptr(), w[1].data_ptr(), w[2].data_ptr(),
                                                w[3].data_ptr(), n, d, feat.shape[2], self.n_head, feat.shape[1], out.data_ptr(),
                                                torch.cuda.current_stream(code.device).cuda_stream), "gpnerf_vertex_attention")
        return out


# ------------------------------------------------------------------------------------------------
# sparse 3-D convolution by rulebook
# ------------------------------------------------------------------------------------------------
class SparseConvTensor:
    """What the reference's renderers hand to the sparse net: `spconv.SparseConvTensor(features, indices, spatial_shape,
    batch_size)` (trainhead.py:54, demo_render.py:154).  The net only reads these four attributes, so a real spconv tensor
    works too.  `dense_levels` (optional): the 4 dense levels already built -- short-cuts the convolutions, like
    sp_input['volumes'] does for NeRFHead.forward."""

    def __init__(self, features, indices, spatial_shape, batch_size, dense_levels=None):
        self.features, self.indices, self.spatial_shape, self.batch_size = features, indices, spatial_shape, batch_size
        self.dense_levels = dense_levels


class _SparseConv3d(nn.Module):
    """Weight [k,k,k,Cin,Cout] as spconv v1.x stores it; out[o] = sum_k W[k] in[o*stride - pad + k]."""

    def __init__(self, cin, cout, ksize, stride=1, padding=0, subm=False):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.padding, self.subm = cin, cout, ksize, stride, padding, subm
        self.weight = nn.Parameter(torch.empty(ksize, ksize, ksize, cin, cout))
        nn.init.kaiming_uniform_(self.weight.view(-1, cout), a=5 ** 0.5)

    def forward(self, x):
        raise L.GpnerfError("sparse convolutions run inside SparseConvNet.dense_levels_hip (gpnerf_volume.hip); there is no torch path")


class _SparseSequential(nn.Sequential):
    """conv -> BatchNorm1d -> ReLU chains (spconv.SparseSequential): a parameter container here, see SparseConvNet."""

    def forward(self, x):
        raise L.GpnerfError("sparse convolutions run inside SparseConvNet.dense_levels_hip (gpnerf_volume.hip); there is no torch path")


def _bn(c):
    return nn.BatchNorm1d(c, eps=1e-3, momentum=0.01)


def double_conv(cin, cout):      # SparseConvNet.py:33-49: SubM 3^3 x2
    return _SparseSequential(_SparseConv3d(cin, cout, 3, subm=True), _bn(cout), nn.ReLU(),
                             _SparseConv3d(cout, cout, 3, subm=True), _bn(cout), nn.ReLU())


def stride_conv(cin, cout):      # SparseConvNet.py:78-87: k3 s2 p1
    return _SparseSequential(_SparseConv3d(cin, cout, 3, stride=2, padding=1), _bn(cout), nn.ReLU())


class _NotCopied:
    """holder of a per-module cache that copy.deepcopy / pickle leave behind (ctypes arrays of pointers cannot be copied, and a copy
    of the module has to build its own anyway)"""

    def __init__(self, v):
        self.v = v

    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (type(None), ())


class SparseConvNet(nn.Module):
    """Same module tree / state_dict keys as SparseConvNet.py:90-103 (`net.{0..8}`)."""

    def __init__(self, n_layers=4, in_dim=16, out_dim=(32, 32, 32, 32)):
        super().__init__()
        self.n_layers = n_layers
        assert len(out_dim) == n_layers
        net = []
        for i in range(n_layers):
            cin = in_dim if i == 0 else out_dim[i - 1]
            net.append(double_conv(cin, cin))
            net.append(stride_conv(cin, out_dim[i]))
        net.append(double_conv(out_dim[-1], out_dim[-1]))
        self.net = nn.ModuleList(net)

    # ---- the MI355X-native path: gpnerf_volume.hip ---------------------------------------------------------
    def _folded_bn(self, bn):
        """inference-mode BatchNorm1d as y = x * scale + shift; cached until a parameter or running statistic changes
        (five tiny launches per convolution otherwise, every frame)."""
        ts = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
        key = tuple((t.data_ptr(), t._version) for t in ts)
        cache = self.__dict__.setdefault("_bn_cache", {})
        hit = cache.get(id(bn))
        if hit is None or hit[0] != key:
            with torch.no_grad():
                scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                hit = (key, scale.float().contiguous(), (bn.bias - bn.running_mean * scale).float().contiguous())
            cache[id(bn)] = hit
        return hit[1], hit[2]

    def _packed_weight(self, mod, dev):
        """gpnerf_sparse_pack_weight image of a conv's [3,3,3,Cin,Cout] weight on `dev`, re-packed when the parameter changes."""
        key = (str(dev), mod.weight.data_ptr(), mod.weight._version)
        cache = self.__dict__.setdefault("_wp_cache", {})
        hit = cache.get(id(mod))
        if hit is None or hit[0] != key:
            lib = L.lib()
            w = np.ascontiguousarray(mod.weight.detach().float().cpu().numpy().reshape(27, mod.cin, mod.cout))
            if self._split16(mod):
                # split-precision image (f16 hi | lo + an fp32 copy); weights beyond its range (|w| >= 15.99) keep the fp32 form
                packed = np.zeros(int(lib.gpnerf_sparse_packed_weight16_bytes(mod.cin)), np.uint8)
                rc = lib.gpnerf_sparse_pack_weight16(w.ctypes.data_as(L.FP), mod.cin, mod.cout, packed.ctypes.data_as(C.c_void_p))
                if rc == 0:
                    hit = (key, torch.from_numpy(packed).to(dev), True)
                    cache[id(mod)] = hit
                    return hit[1], True
            packed = np.zeros(int(lib.gpnerf_sparse_packed_weight_floats(mod.cin)), np.float32)
            L.check(lib.gpnerf_sparse_pack_weight(w.ctypes.data_as(L.FP), mod.cin, mod.cout, packed.ctypes.data_as(L.FP)),
                    "gpnerf_sparse_pack_weight")
            hit = (key, torch.from_numpy(packed).to(dev), False)
            cache[id(mod)] = hit
        return hit[1], hit[2]

    @staticmethod
    def _split16(mod):
        """the split-precision kernel's shapes (gpnerf_sparse_conv3_mfma16); GPNERF_SPARSE_FP32=1 under GPNERF_DEBUG=1 keeps the fp32 form"""
        import os
        if os.environ.get("GPNERF_DEBUG") == "1" and os.environ.get("GPNERF_SPARSE_FP32") == "1":
            return False
        return mod.cin in (16, 32) and mod.cout <= 32

    def plan_levels(self, coord, out_sh, channels=None):
        """The STRUCTURE of a frame's pyramid, which depends on the vertices' voxel coordinates only: the full-resolution index
        grid, for each of the 4 levels its coarse site list + index grid (gpnerf_sparse_down_sites), and the zeroed dense volumes.
        No features are touched, so Renderer.render enqueues this on a side stream while the image encoder runs (~150 us of
        small launches off the critical path); `dense_levels_hip(..., plan=)` then only runs the convolutions and the scatters."""
        if any(int(v) % 16 for v in out_sh):
            # the reference's datasets round out_sh up to a multiple of 32 (ZjumocapDataset.py:243-254); with odd sizes spconv's
            # strided output shape (n - 1) // 2 + 1 and the n // 2 used below would part ways
            raise L.GpnerfError(f"out_sh {tuple(out_sh)} must be a multiple of 16 in every dimension")
        if not coord.is_cuda:
            raise L.GpnerfError("the HIP volume builder needs GPU tensors (no CPU fallback)")
        lib = L.lib()
        dev = coord.device
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        dims = tuple(int(v) for v in out_sh)
        coords = coord[:, 1:].to(torch.int32).contiguous()
        m0 = coords.shape[0]
        ch = [int(c) for c in (channels or [self.net[2 * i + 2][3].cout for i in range(self.n_layers)])]
        if self.n_layers > L.PYRAMID_MAX_LEVELS:
            raise L.GpnerfError(f"the pyramid runner is built for at most {L.PYRAMID_MAX_LEVELS} levels")
        # every buffer of the frame's pyramid, and ONE native call that lays its structure out (gpnerf_sparse_pyramid_plan: the
        # full-resolution index, per level the coarse sites + their grid + the zeroed dense volume: ~30 launches)
        p = L.GpnerfPyramid()
        p.n_levels, p.m0 = self.n_layers, m0
        p.dims0[:] = dims
        keep = {"coords0": coords, "grid0": torch.empty(dims, device=dev, dtype=torch.int32),
                "dup": torch.empty((9 * max(m0, 1),), device=dev, dtype=torch.int32), "levels": []}
        p.coords0, p.grid0, p.dup_scratch = coords.data_ptr(), keep["grid0"].data_ptr(), keep["dup"].data_ptr()
        m_cap, d = m0, dims
        for i in range(self.n_layers):
            d = tuple(n // 2 for n in d)
            cap = int(min(d[0] * d[1] * d[2], 8 * m_cap))           # a fine site reaches at most 2^3 coarse sites
            grid = torch.empty(d, device=dev, dtype=torch.int32)
            oc = torch.empty((max(cap, 1), 3), device=dev, dtype=torch.int32)
            om = torch.empty((1,), device=dev, dtype=torch.int32)
            vol = torch.empty(d + (ch[i],), device=dev, dtype=torch.float32)
            p.dims[i][:] = d
            p.cap[i], p.ch[i] = cap, ch[i]
            p.grid[i], p.coords[i], p.m[i], p.vol[i] = grid.data_ptr(), oc.data_ptr(), om.data_ptr(), vol.data_ptr()
            keep["levels"].append((grid, d, oc, om, cap, vol))
            m_cap = cap
        L.check(lib.gpnerf_sparse_pyramid_plan(C.byref(p), st), "gpnerf_sparse_pyramid_plan")
        keep["pyramid"], keep["dims0"], keep["max_rows"] = p, dims, max([m0] + [lv[4] for lv in keep["levels"]])
        return keep

    def dense_levels_hip(self, code, coord, out_sh, batch_size=1, plan=None):
        """code [M,C] per-vertex features, coord [M,4] (batch, d, h, w), out_sh (D,H,W) -> the 4 dense levels of
        SparseConvNet.py:105-111, computed by the HIP sparse-convolution kernels and returned directly in the render kernel's
        channels-last layout: a list of 4 tensors [D_k,H_k,W_k,C] (tagged `_gpnerf_ndhwc`).  plan: `plan_levels(coord, out_sh)`
        made earlier (on whatever stream, as long as this stream waits for it)."""
        if int(batch_size) != 1:
            raise ValueError("the per-ray path renders one frame at a time (BaseRender.py:336 asserts batch 1)")
        if any(int(v) % 16 for v in out_sh):
            raise L.GpnerfError(f"out_sh {tuple(out_sh)} must be a multiple of 16 in every dimension")       # (see plan_levels)
        if self.training:
            raise L.GpnerfError("the HIP volume builder folds BatchNorm running statistics: call .eval() first")
        if not code.is_cuda:
            raise L.GpnerfError("the HIP volume builder needs GPU tensors (no CPU fallback)")
        if plan is None:
            plan = self.plan_levels(coord, out_sh)
        lib = L.lib()
        dev = code.device
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        convs = self._conv_table(dev)
        p = plan["pyramid"]
        if [lv[5].shape[-1] for lv in plan["levels"]] != [int(convs[4 + 3 * i].cout) for i in range(self.n_layers)]:
            raise L.GpnerfError("plan_levels: channel count of the planned volume does not match the network")
        x = code.detach().float().contiguous()
        if x.shape[0] != p.m0 or x.shape[1] != convs[0].cin:
            raise L.GpnerfError(f"code {tuple(x.shape)} does not match the plan's {p.m0} rows / the network's {convs[0].cin} input channels")
        # two feature buffers the convolutions alternate between, then ONE native call for the ~30 launches of the pyramid
        feat = torch.empty((3, max(plan["max_rows"], 1), 32), device=dev, dtype=torch.float32)
        p.feat_a, p.feat_b, p.feat_c = feat[0].data_ptr(), feat[1].data_ptr(), feat[2].data_ptr()
        L.check(lib.gpnerf_sparse_pyramid_run(C.byref(p), x.data_ptr(), int(x.shape[1]), convs, len(convs), st), "gpnerf_sparse_pyramid_run")
        levels = []
        for lv in plan["levels"]:
            lv[5]._gpnerf_ndhwc = True
            levels.append(lv[5])
        plan["feat"] = feat                                   # alive until the plan goes (stream order covers the rest)
        return levels

    def _conv_table(self, dev):
        """GpnerfSparseConv[2 + 3 * n_layers] in network order (weights packed / BatchNorm folded on a parameter change only); the
        tensors it points to are kept alive with it."""
        mods = [(False, self.net[0][0], self.net[0][1]), (False, self.net[0][3], self.net[0][4])]
        for i in range(self.n_layers):
            sc, dc = self.net[2 * i + 1], self.net[2 * i + 2]
            mods += [(True, sc[0], sc[1]), (False, dc[0], dc[1]), (False, dc[3], dc[4])]
        key = (str(dev),) + tuple((m.weight.data_ptr(), m.weight._version, bn.weight._version, bn.bias._version, bn.running_mean._version,
                                   bn.running_var._version, bn.running_mean.data_ptr()) for _, m, bn in mods)
        hit = self.__dict__.get("_conv_table_cache")
        hit = hit.v if hit is not None else None
        if hit is not None and hit[0] == key:
            return hit[1]
        table = (L.GpnerfSparseConv * len(mods))()
        alive = []
        for n_conv, (t, (strided, mod, bn)) in enumerate(zip(table, mods)):
            if mod.cout > 32:
                raise L.GpnerfError("the sparse convolutions are built for at most 32 output channels")
            scale, shift = self._folded_bn(bn)
            if mod.cin % 8 == 0 and mod.cin <= 32:
                w, split16 = self._packed_weight(mod, dev)
                form = 2 if split16 else 1
            else:
                w, form = mod.weight.detach().float().contiguous(), 0
            t.strided, t.cin, t.cout, t.form = int(strided), mod.cin, mod.cout, form
            t.weight, t.bn_scale, t.bn_shift = w.data_ptr(), scale.data_ptr(), shift.data_ptr()
            alive += [w, scale, shift]
            if n_conv < 2:                # the two vertex-level convolutions: spconv's own [27][cin][cout] layout for the shared-voxel rows
                raw = w if form == 0 else mod.weight.detach().float().contiguous().to(dev)
                t.weight_raw = raw.data_ptr()
                alive.append(raw)
        self.__dict__["_conv_table_cache"] = _NotCopied((key, table, alive))
        return table

    # ---- the reference's two calls (SparseConvNet.py:105-143) --------------------------------------------------------
    def _levels_of(self, x):
        """The 4 channels-last levels of sparse tensor `x`, built once per tensor object (the reference rebuilds them on
        every call: encode, then forward inside test_forward, demo_render.py:155,295)."""
        hit = self.__dict__.get("_levels_cache")
        if hit is not None and hit[0] is x:
            return hit[1]
        from . import frame as F_
        pre = getattr(x, "dense_levels", None)
        if pre is not None:
            fr = F_.Frame.for_volumes(list(pre), None)
        else:
            fr = F_.Frame.for_volumes(self.dense_levels_hip(x.features, x.indices, x.spatial_shape, x.batch_size), None)
        self.__dict__["_levels_cache"] = (x, fr)
        return fr

    def forward(self, x, grid_coords=None):
        """SparseConvNet.forward (:105-124): with grid_coords [B,1,1,P,3] the trilinear samples of the 4 levels, [B,128,P];
        without, the list of dense levels [1,C,D,H,W] (views of the channels-last volumes)."""
        from . import frame as F_
        fr = self._levels_of(x)
        if grid_coords is None:
            return [v.permute(3, 0, 1, 2).unsqueeze(0) for v in fr.vols]
        P = grid_coords.shape[-2]
        feat = F_.sample_volume(fr, grid_coords.reshape(-1, 3))
        return feat.view(grid_coords.shape[0], P, 128).permute(0, 2, 1)

    def encode(self, x, threshold=0.1):
        """SparseConvNet.encode (:126-143): sets `features` (the dense levels), `masks3d` (occupancy at level-1 size:
        channel sums, nearest-upsampled, summed over levels; gpnerf_build_occupancy) and `mask_xyz` (the occupied voxels'
        (x,y,z) indices times 2, float, in torch.where's d-major order)."""
        fr = self._levels_of(x)
        self.features = [v.permute(3, 0, 1, 2).unsqueeze(0) for v in fr.vols]
        self.masks3d = fr.build_occupancy()
        self.mask_xyz = torch.stack(torch.where(self.masks3d > threshold), dim=0).permute(1, 0).flip(-1).float() * 2.0
