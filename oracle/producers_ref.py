"""CPU restatements of the per-frame producers (TEST INFRASTRUCTURE -- only tests/ import this; the product never does).

The product computes these on the GPU only (gp-nerf_amd/volume.py, encoder.py -> HIP kernels / MIOpen).  What is here is the
checker: plain torch-CPU formulations that read the parameters off the product's modules (same state_dict keys as the
reference), so one set of weights drives both sides.

* `attention(module, q, k, v)` -- libs/nerfheads/networks/MultiHeadAttention.py:61-98.  PINNED: tests/test_attention.py holds it
  to vectors captured from the reference's module (tests/golden/attention_*.npz).
* `encoder(net, x)` -- libs/encoders/UNet.py:17-53,107-131,213-234.  PINNED: tests/test_encoder.py holds it to vectors captured
  from the reference's ResUNet (tests/golden/encoder_*.npz).
* `dense_levels(net, code, coord, out_sh)` -- libs/nerfheads/networks/SparseConvNet.py:22-111 on the published algorithm of
  spconv v1.2.1 (SubMConv3d / SparseConv3d by coordinate rulebook, `.dense()`).  spconv is absent from the reference tree and
  not installable here: **parity unpinned**; tests/test_volume_builder.py checks this restatement against the dense
  conv3d-with-mask definition it must agree with.

spconv v1.2.1 semantics this restatement (and csrc/gpnerf_volume.hip behind it) assumes -- the reference pins the version in
README.md:27-33 (`git checkout abf0acf30f5526ea93e687e3f424f62d9cd8313a`).  The library is NOT in this image and there is no
network: the locations below are RECALLED from the published repository (traveller59/spconv at that commit), they were not read
here, and nothing in this list has been executed against spconv.  A maintainer with spconv installed can settle every item with
the ten-line check at the end.

 1. Weight layout and tap index.  `SparseConvolution.__init__` (spconv/conv.py) allocates `weight` as
    `[kd, kh, kw, Cin, Cout]` (the reference's state_dict agrees: `xyzc_net.net.7.0.weight` is (3, 3, 3, 32, 32)) and
    `indice_conv` (spconv/ops.py -> src/spconv/spconv_ops.cc `indiceConv`) views it as `filters.view(-1, Cin, Cout)`: tap
    i = (kd * 3 + kh) * 3 + kw, kd slowest.  Here: `weight.view(k * k * k, cin, cout)` and `offs` from
    `meshgrid(..., indexing="ij")` in the same order.
 2. Tap index <-> spatial offset.  include/spconv/geometry.h `getValidOutPos` pairs input position p with output o under
    offset index built from `(p - o * stride + padding) / dilation` per axis, row-major: tap k reads the input at
    p = o * stride - padding + k  (cross-correlation, as torch.nn.Conv3d; NOT the flipped convolution).  spconv's own
    test/test_conv.py checks SparseConv3d against a dense Conv3d whose weight is `filters.transpose(4, 3, 0, 1, 2)` -- the same
    permutation tests/test_volume_builder.py uses for its F.conv3d reference.  A flipped or axis-swapped tap order fails that
    test here by 0.5+ (test_a_permuted_tap_order_is_caught): the dense check discriminates.
 3. SubMConv3d (k = 3).  `getIndicePairsSubM`: output sites = input sites, same row order; for every input row j at position p
    and every tap k, the pair (in = j, out = row of the active site at p + padding - k) is recorded when that site exists;
    `indiceConv` applies the centre tap to all rows at once (`filters[kernelVolume / 2]`) and skips it in the loop.  Same sums.
 4. SubMConv3d with k = 1 (`single_conv`, SparseConvNet.py:22-33; defined but not used by SparseConvNet.__init__):
    `SparseConvolution.forward` takes the `conv1x1` shortcut, `features @ weight.view(Cin, Cout)` on EVERY row -- rows that share a
    voxel stay separate rows.  (`sparse_conv3d(subm=True)` with k = 1 routes the centre tap through the lookup instead: not the
    shortcut's semantics on shared voxels; nothing in the network exercises it.)
 5. SparseConv3d (stride 2, padding 1, k = 3; `stride_conv`, SparseConvNet.py:85-92).  `getIndicePairsConv`: every INPUT ROW
    generates its pairs, so several rows in one voxel (SMPL vertices quantised to 5 mm: the vertex level has them) ALL add into
    the outputs they reach; output sites = the distinct reachable positions, spatial shape floor((n + 2 p - k) / s) + 1 per axis.
    Row order of the outputs differs between spconv's CPU and CUDA paths and does not matter to `.dense()`.
 6. Rows that share a position and the k = 3 submanifold convolution -- the network's FIRST block, net[0] = double_conv on the
    vertex level (SparseConvNet.py:96-97,106), is exactly that.  src/spconv/indice.cc `create_submconv_indice_pair_cpu` fills the
    position -> row grid with every row in turn (the LAST row of a voxel stays: its owner; the CUDA path elects the owner by a write
    race, so the reference itself is not deterministic on such voxels), then lets EVERY input row j emit, per tap, the pair
    (j -> owner of the active position it feeds); src/spconv/spconv_ops.cc `indiceConv` applies the centre tap as
    `mm_out(output, features, filters[centre])` on all rows and skips it in the pair loop.  Hence
        owner row:      f W_c + sum over the other taps of (the SUM of all rows in that neighbour voxel) W_k
        any other row:  f W_c only.
    `subm_conv3d_rulebook` restates this and csrc/gpnerf_volume.hip implements it (subm_shared_rows_kernel) since round 5; rounds
    2-4 let every row compute through one-representative lookups instead (`sparse_conv3d(subm=True)`), which agrees wherever no
    voxel is shared and differs materially where one is (tools/probes/duplicate_voxels.py).
 7. `.dense()` (spconv/__init__.py `SparseConvTensor.dense` -> `scatter_nd`): zeros + index assignment of the feature rows at their
    coordinates, `[N, C, D, H, W]`.
 8. BatchNorm1d(eps=1e-3) + ReLU act on the feature rows (spconv.SparseSequential applies plain modules to `.features`).

What is and is not recall-only (round 6).  Items 1-3, 5, 7 and 8 together say "a sparse convolution is the dense cross-correlation
of the densified input, restricted to the active (submanifold) / reachable (strided) sites, with BatchNorm + ReLU on those sites" --
which is spconv's own documented contract and the property its test suite checks against torch.nn.Conv3d.  On inputs with ONE row
per voxel that contract fixes every output, and the HIP builder is held to it directly, without this file's sparse code:
tests/test_gpu_sparse_conv.py::test_the_builder_is_the_dense_conv3d_pyramid_where_no_voxel_is_shared builds the whole four-level
pyramid with F.conv3d in float64 on fifty random person-shaped vertex sets.  What REMAINS recall-only is item 6 alone -- what
spconv v1.2.1 does with several rows in one voxel in the first submanifold block (owner row / other rows; the CUDA path's owner
election is a write race, so the reference itself is not deterministic there) -- and item 4, which the network never exercises.
On a real SMPL frame ~7 % of the level-0 sites are shared voxels (tools/probes/duplicate_voxels.py: 1 236 of 18 183).

Settling it with spconv at hand (not possible here):
    x = spconv.SparseConvTensor(feat, coord_int32, shape, 1); conv = spconv.SparseConv3d(C, C, 3, 2, padding=1, bias=False)
    ref = sparse_conv3d(SparseTensor(feat, coord[:, 1:].long(), shape), conv.weight, 2, 1).dense()
    assert torch.allclose(conv(x).dense(), ref, atol=1e-5)        # and the same for SubMConv3d(k=3) on duplicate-free coords
"""
import torch
import torch.nn.functional as F


# ---- MultiHeadAttention.forward ----------------------------------------------------------------------------------------
def attention(m, q, k, v, mask=None):
    """m: the product's MultiHeadAttention (parameters w_qs, w_ks, w_vs, fc, layer_norm).  Returns (out, attn)."""
    b, lq, lk = q.size(0), q.size(1), k.size(1)
    residual = q
    qh = m.w_qs(q).view(b, lq, m.n_head, m.d_k).transpose(1, 2)
    kh = m.w_ks(k).view(k.size(0), lk, m.n_head, m.d_k).transpose(1, 2)
    vh = m.w_vs(v).view(v.size(0), v.size(1), m.n_head, m.d_v).transpose(1, 2)
    attn = torch.matmul(qh / (m.d_k ** 0.5), kh.transpose(2, 3))
    if mask is not None:
        attn = attn.masked_fill(mask.unsqueeze(1) == 0, -1e9)
    attn = F.softmax(attn, dim=-1)
    out = torch.matmul(attn, vh).transpose(1, 2).contiguous().view(b, lq, -1)
    out = m.fc(out)
    if m.sum_flag:
        out = m.layer_norm(out + residual)
    return out, attn


# ---- ResUNet.forward ---------------------------------------------------------------------------------------------------
def _unit(u, x):
    y = u.bn2(u.conv2(F.relu(u.bn1(u.conv1(x)))))
    return F.relu(y + (x if u.downsample is None else u.downsample(x)))


def _cne(c, x):
    return F.elu(c.bn(c.conv(x)))


def _up(u, x):
    return _cne(u.conv, F.interpolate(x, scale_factor=u.scale, mode="bilinear", align_corners=True))


def _skip(skip, up):
    dy, dx = up.shape[2] - skip.shape[2], up.shape[3] - skip.shape[3]
    if dy or dx:
        skip = F.pad(skip, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))
    return torch.cat([up, skip], dim=1)


def encoder(net, x):
    """net: the product's ResUNet; x [V,3,H,W] -> [V,out_ch,H/4,W/4] with stock torch operators."""
    x = F.relu(net.bn1(net.conv1(x)))
    x1 = x
    for u in net.layer1:
        x1 = _unit(u, x1)
    x2 = x1
    for u in net.layer2:
        x2 = _unit(u, x2)
    x3 = x2
    for u in net.layer3:
        x3 = _unit(u, x3)
    x = _cne(net.iconv3, _skip(x2, _up(net.upconv3, x3)))
    x = _cne(net.iconv2, _skip(x1, _up(net.upconv2, x)))
    return net.out_conv(x)


# ---- sparse 3-D convolution by rulebook --------------------------------------------------------------------------------
class SparseTensor:
    """features [M,C]; coords [M,3] (d,h,w) int64; spatial shape (D,H,W).  Batch size 1."""

    def __init__(self, features, coords, shape):
        self.features, self.coords, self.shape = features, coords, tuple(int(s) for s in shape)

    def keys(self):
        D, H, W = self.shape
        return (self.coords[:, 0] * H + self.coords[:, 1]) * W + self.coords[:, 2]

    def dense(self):
        """[1,C,D,H,W], zeros where inactive (spconv's .dense(), SparseConvNet.py:111)."""
        D, H, W = self.shape
        C = self.features.shape[1]
        out = torch.zeros((D * H * W, C), dtype=self.features.dtype, device=self.features.device)
        out[self.keys()] = self.features
        return out.view(D, H, W, C).permute(3, 0, 1, 2).unsqueeze(0).contiguous()


def _lookup(sorted_keys, order, query):
    """index into the original rows of the entry whose key equals `query` (the highest row among duplicates), or -1."""
    pos = (torch.searchsorted(sorted_keys, query, right=True) - 1).clamp_(min=0)
    hit = sorted_keys[pos] == query
    return torch.where(hit, order[pos], torch.full_like(pos, -1))


def subm_conv3d_rulebook(x, weight):
    """SubMConv3d as spconv v1.2.1's CPU rulebook treats rows that SHARE a voxel (header item 6, RECALLED -- see there): the grid maps
    a position to the LAST row written there (the owner); every input row i generates the pairs (i -> owner of each active
    neighbour position); the centre tap is applied to every row's own features.  So an owner row receives its neighbours' rows --
    ALL of them, summed --, a non-owner row only its own centre term.  Equal to sparse_conv3d(subm=True) when no voxel is shared.
    (spconv's CUDA path elects the owner by a write race: the reference itself is not deterministic on such voxels.)"""
    k = weight.shape[0]
    cin, cout = weight.shape[3], weight.shape[4]
    pad = k // 2
    D, H, W = x.shape
    dev = x.coords.device
    offs = torch.stack(torch.meshgrid(torch.arange(k), torch.arange(k), torch.arange(k), indexing="ij"), -1).view(-1, 3).to(dev)
    Wk = weight.view(k * k * k, cin, cout)
    sk, order = torch.sort(x.keys(), stable=True)
    centre = (k * k * k) // 2
    out = x.features @ Wk[centre]
    lim = torch.tensor([D, H, W], device=dev)
    for i in range(k * k * k):
        if i == centre:
            continue
        q = x.coords + pad - offs[i]                            # the output position input row j feeds through tap i: p = q - pad + k
        ok = ((q >= 0) & (q < lim)).all(1)
        key = (q[:, 0] * H + q[:, 1]) * W + q[:, 2]
        o = _lookup(sk, order, torch.where(ok, key, torch.full_like(key, -1)))
        sel = (o >= 0) & ok
        if sel.any():
            out.index_add_(0, o[sel], x.features[sel] @ Wk[i])
    return SparseTensor(out, x.coords, x.shape)


def sparse_conv3d(x, weight, stride=1, padding=0, subm=False):
    """weight [k,k,k,Cin,Cout] as spconv v1.x stores it; out[o] = sum_k W[k] in[o*stride - pad + k]."""
    k, s = weight.shape[0], stride
    cin, cout = weight.shape[3], weight.shape[4]
    pad = (k // 2) if subm else padding
    D, H, W = x.shape
    dev = x.coords.device
    offs = torch.stack(torch.meshgrid(torch.arange(k), torch.arange(k), torch.arange(k), indexing="ij"), -1).view(-1, 3).to(dev)
    Wk = weight.view(k * k * k, cin, cout)
    if subm:
        sk, order = torch.sort(x.keys(), stable=True)
        out = torch.zeros((x.coords.shape[0], cout), dtype=x.features.dtype, device=dev)
        for i in range(k * k * k):
            nb = x.coords - pad + offs[i]                       # input position feeding output site through tap i
            ok = ((nb >= 0) & (nb < torch.tensor([D, H, W], device=dev))).all(1)
            q = (nb[:, 0] * H + nb[:, 1]) * W + nb[:, 2]
            j = _lookup(sk, order, torch.where(ok, q, torch.full_like(q, -1)))
            sel = (j >= 0) & ok
            if sel.any():
                out[sel] += x.features[j[sel]] @ Wk[i]
        return SparseTensor(out, x.coords, x.shape)
    oD, oH, oW = [(n + 2 * pad - k) // s + 1 for n in (D, H, W)]
    pairs_o, pairs_i, pairs_k = [], [], []
    lim = torch.tensor([oD, oH, oW], device=dev)
    for i in range(k * k * k):
        num = x.coords + pad - offs[i]                           # o*stride = p + pad - k
        o = torch.div(num, s, rounding_mode="floor")
        ok = ((num % s) == 0).all(1) & ((o >= 0) & (o < lim)).all(1)
        idx = torch.nonzero(ok).squeeze(1)
        pairs_o.append((o[idx, 0] * oH + o[idx, 1]) * oW + o[idx, 2])
        pairs_i.append(idx)
        pairs_k.append(torch.full_like(idx, i))
    okeys, iidx, kidx = torch.cat(pairs_o), torch.cat(pairs_i), torch.cat(pairs_k)
    ukeys, inv = torch.unique(okeys, sorted=True, return_inverse=True)
    out = torch.zeros((ukeys.numel(), cout), dtype=x.features.dtype, device=dev)
    for i in range(k * k * k):
        m = kidx == i
        if m.any():
            out.index_add_(0, inv[m], x.features[iidx[m]] @ Wk[i])
    oc = torch.stack([ukeys // (oH * oW), (ukeys // oW) % oH, ukeys % oW], 1)
    return SparseTensor(out, oc, (oD, oH, oW))


def _sequential(seq, x, rulebook_duplicates=True):
    """conv -> BatchNorm1d -> ReLU chains on the active features (spconv.SparseSequential); seq: the product's module list."""
    for m in seq:
        if hasattr(m, "subm"):
            if rulebook_duplicates and m.subm and m.weight.shape[0] == 3:
                x = subm_conv3d_rulebook(x, m.weight)
            else:
                x = sparse_conv3d(x, m.weight, m.stride, m.padding, m.subm)
        else:
            x = SparseTensor(m(x.features), x.coords, x.shape)
    return x


def dense_levels(net, code, coord, out_sh, rulebook_duplicates=True):
    """net: the product's SparseConvNet; code [M,C] per-vertex features, coord [M,4] (batch, d, h, w), out_sh (D,H,W) ->
    list of 4 dense levels [1,C_k,D/2^k,H/2^k,W/2^k] (SparseConvNet.py:105-111).
    rulebook_duplicates (default, and what csrc/gpnerf_volume.hip implements since round 5): the two submanifold convolutions of
    the VERTEX level (the only one whose rows can share a voxel) as subm_conv3d_rulebook -- what this file's header recalls of
    spconv's own handling of shared voxels.  False: one representative row per voxel for every lookup and every row computed
    through lookups (rounds 2-4's reading; tools/probes/duplicate_voxels.py measures the difference: material wherever voxels are
    shared -- 1 236 of 18 183 level-0 sites of the body-like frame)."""
    x = SparseTensor(code, coord[:, 1:].long(), out_sh)
    x = _sequential(net.net[0], x, rulebook_duplicates)
    levels = []
    for i in range(net.n_layers):
        x = _sequential(net.net[2 * i + 1], x)
        x = _sequential(net.net[2 * i + 2], x)
        levels.append(x.dense())
    return levels
