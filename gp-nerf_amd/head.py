"""`build_head(cfg)` / `NeRFHead` with the reference's interface and parameter names
(libs/nerfheads/trainhead.py:27-177), computing on the HIP path.

The modules below own exactly the parameters the reference's do, under the same
state_dict keys, so `load_state_dict(ckpt['state_dict'], strict=True)`
of a reference checkpoint works (tools/inference.py:67-74).  They do NOT evaluate the
per-ray layers with torch: `NeRFHead.forward` packs the parameters into the kernels' LDS image
and calls gpnerf_sample_volume + gpnerf_head_forward.  There is no PyTorch fallback.

Per-frame pieces that are outside the per-ray path (SMPL code attention, sparse volume
builder) live in `volume.py`.
"""
import torch
import torch.nn as nn

from . import _lib as L
from . import frame as F_
from . import volume as V_


def weights_init(m):
    """kaiming-normal weights, zero bias on every Linear (trainhead.py:13-17)."""
    if isinstance(m, nn.Linear):
        nn.init.kaiming_normal_(m.weight.data)
        if m.bias is not None:
            nn.init.zeros_(m.bias.data)


class NeRFSigmaHead(nn.Module):
    """Parameters of trainhead.py:27-41: `c`, `xyzc_attn`, `xyzc_net`, `out_geometry_fc`."""

    def __init__(self, in_feat_ch=32, n_smpl=6890, code_dim=16, attn_n_heads=4, spconv_n_layers=4,
                 spconv_out_dim=(32, 32, 32, 32)):
        super().__init__()
        self.n_smpl = n_smpl
        self.c = nn.Embedding(n_smpl, code_dim)
        self.xyzc_attn = V_.MultiHeadAttention(attn_n_heads, code_dim, code_dim // attn_n_heads, code_dim // attn_n_heads,
                                               kv_dim=in_feat_ch, sum=False)
        self.xyzc_net = V_.SparseConvNet(n_layers=spconv_n_layers, in_dim=code_dim, out_dim=list(spconv_out_dim))
        self.out_geometry_fc = nn.Sequential(nn.Linear(sum(spconv_out_dim), 64), nn.ELU(inplace=True))
        self.out_geometry_fc.apply(weights_init)

    def build_volumes(self, sp_input, smpl_feat_sampled):
        """Embedding -> attention over the V views -> sparse conv net -> 4 dense levels
        (trainhead.py:48-56, SparseConvNet.py:105-111).  Per frame, not per ray.  sp_input["plan"]: the pyramid's structure laid
        out earlier (SparseConvNet.plan_levels; Renderer.render does that while the image encoder runs)."""
        code = self.c.weight                        # = self.c(arange(n_smpl)) (trainhead.py:48): every row, in order
        fused = self.xyzc_attn.fuse_vertices(code, smpl_feat_sampled.flatten(0, 1))          # HIP only: raises on CPU / in training
        return self.xyzc_net.dense_levels_hip(fused, sp_input["coord"], sp_input["out_sh"], sp_input["batch_size"], plan=sp_input.get("plan"))


    def _blob(self, device):
        """Head image carrying this module's Linear(128,64) only (the other layers zero): enough for gpnerf_sigma_features."""
        lin = self.out_geometry_fc[0]
        key = (str(device), lin.weight.data_ptr(), lin.weight._version, lin.bias.data_ptr(), lin.bias._version)
        if self.__dict__.get("_blob_key") != key:
            sd = {}
            for short, name in L.HEAD_FIELDS:
                sd[name + ".weight"] = torch.zeros(L.HEAD_SHAPES[short])
                sd[name + ".bias"] = torch.zeros(L.HEAD_SHAPES[short][0])
            sd["sigmahead.out_geometry_fc.0.weight"], sd["sigmahead.out_geometry_fc.0.bias"] = lin.weight, lin.bias
            self.__dict__["_blob_t"], self.__dict__["_blob_key"] = F_.pack_head(sd, device), key
        return self.__dict__["_blob_t"]

    def _sample(self, x, grid_coords):
        """xyzc_net(x, grid)[B,128,P] -> [P,128] rows (trainhead.py:56-57,63-68)."""
        return self.xyzc_net(x, grid_coords[:, None, None].float()).permute(0, 2, 1).reshape(-1, 128)

    def forward(self, sp_input, grid_coords, smpl_feat_sampled, mask):
        """trainhead.py:43-59: per-frame volumes from the SMPL features, sampled at grid_coords [1,P,3], through
        Linear(128,64)+ELU.  Returned with the reference's (odd) view [-1, n_samples, 1]."""
        x = sp_input.get("xyzc")
        if x is None:
            code = self.c(torch.arange(0, self.n_smpl, device=grid_coords.device))
            fused = self.xyzc_attn.fuse_vertices(code, smpl_feat_sampled.flatten(0, 1))
            x = V_.SparseConvTensor(fused, sp_input["coord"], sp_input["out_sh"], sp_input["batch_size"], sp_input.get("volumes"))
        vol_feat = self._sample(x, grid_coords)
        zeros = torch.zeros((vol_feat.shape[0], L.VIEWS, 35), device=vol_feat.device)
        sigma_feat, _ = F_.sigma_features(self._blob(vol_feat.device), vol_feat, zeros)
        return sigma_feat.view(-1, mask.shape[1], 1)

    def test_forward(self, sp_input, grid_coords, rgb_feat, mask):
        """trainhead.py:61-76: sigma_feat [R,S,64] and globalfeat [R,S,1,134] = [sigma_feat, mean_v, var_v] of rgb_feat
        [R,S,V,35]; the volumes come from sp_input['xyzc'] (the tensor `encode` was called with, demo_render.py:154-165)."""
        R, S = rgb_feat.shape[:2]
        vol_feat = self._sample(sp_input["xyzc"], grid_coords)
        sf, gf = F_.sigma_features(self._blob(vol_feat.device), vol_feat, rgb_feat.reshape(R * S, L.VIEWS, 35))
        return sf.view(R, S, 64), gf.view(R, S, 1, 134)


class NeRFRGBHead(nn.Module):
    """Parameters of trainhead.py:82-115: base_fc, vis_fc, rgb_fc, out_geometry_fc."""

    def __init__(self, in_feat_ch=32):
        super().__init__()
        if in_feat_ch != L.CH:
            raise L.GpnerfError(f"the HIP kernels are built for {L.CH} feature channels, got in_feat_ch={in_feat_ch}")
        e = lambda: nn.ELU(inplace=True)
        self.base_fc = nn.Sequential(nn.Linear((in_feat_ch + 3) * 3, 64), e(), nn.Linear(64, 32), e())
        self.vis_fc = nn.Sequential(nn.Linear(32, 32), e(), nn.Linear(32, 32), e())
        self.rgb_fc = nn.Sequential(nn.Linear(96, 32), e(), nn.Linear(32, 16), e(), nn.Linear(16, 3))
        self.out_geometry_fc = nn.Sequential(nn.Linear(64 + (in_feat_ch + 3) * 2, 64), e(), nn.Linear(64, 32), e(),
                                             nn.Linear(32, 16), e(), nn.Linear(16, 1), nn.ReLU())
        for m in (self.out_geometry_fc, self.base_fc, self.vis_fc, self.rgb_fc):
            m.apply(weights_init)

    def _blob(self, device):
        """Head image with this module's 11 layers (Linear(128,64) of the sigma head zero): what gpnerf_rgb_head_forward stages."""
        ps = [p for seq in (self.base_fc, self.vis_fc, self.rgb_fc, self.out_geometry_fc) for p in seq.parameters()]
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in ps)
        if self.__dict__.get("_blob_key") != key:
            sd = {"sigmahead.out_geometry_fc.0.weight": torch.zeros(64, 128), "sigmahead.out_geometry_fc.0.bias": torch.zeros(64)}
            for _, name in L.HEAD_FIELDS:
                if name.startswith("rgbhead."):
                    mod, idx = name[len("rgbhead."):].rsplit(".", 1)
                    lin = getattr(self, mod)[int(idx)]
                    sd[name + ".weight"], sd[name + ".bias"] = lin.weight, lin.bias
            self.__dict__["_blob_t"], self.__dict__["_blob_key"] = F_.pack_head(sd, device), key
        return self.__dict__["_blob_t"]

    def forward(self, rgb_feat, sigma_feat, mask):
        """trainhead.py:118-145: rgb_feat [R,S,V,35], sigma_feat (any view of [R,S,64]), mask [R,S,V,1] ->
        (rgb_in [R,S,V,3], rgb_out [R,S,3], sigma_out [R,S,1])."""
        R, S = rgb_feat.shape[:2]
        raw = F_.rgb_head_forward(self._blob(rgb_feat.device), sigma_feat.reshape(R * S, 64), rgb_feat.reshape(R * S, L.VIEWS, 35),
                                  mask.reshape(R * S, L.VIEWS)).view(R, S, 4)
        return rgb_feat[..., :3], raw[..., :3], raw[..., 3:]


class NeRFHead(nn.Module):
    """trainhead.py:148-163.  forward(sp_input, grid_coords, smpl_feat_sampled, rgb_feat, mask) -> (raw, rgb_in)."""

    def __init__(self, in_feat_ch=32, n_smpl=6890, code_dim=16, attn_n_heads=4, spconv_n_layers=4,
                 spconv_out_dim=(32, 32, 32, 32), use_rgbhead=True):
        super().__init__()
        self.sigmahead = NeRFSigmaHead(in_feat_ch=in_feat_ch, n_smpl=n_smpl, code_dim=code_dim, attn_n_heads=attn_n_heads,
                                       spconv_n_layers=spconv_n_layers, spconv_out_dim=spconv_out_dim)
        self.use_rgbhead = use_rgbhead
        self.rgbhead = NeRFRGBHead(in_feat_ch=in_feat_ch)
        self._blob = None
        self._blob_key = None

    # ---- parameters -> LDS image ------------------------------------------------------------
    def per_ray_state(self):
        sd = {}
        for _, name in L.HEAD_FIELDS:
            mod_name, idx = name.rsplit(".", 1)
            seq = self.get_submodule(mod_name)
            lin = seq[int(idx)]
            sd[name + ".weight"], sd[name + ".bias"] = lin.weight, lin.bias
        return sd

    def head_blob(self, device):
        """Packed image of the per-ray layers, re-packed only when a parameter changed (storage pointer or version counter of one
        of the 24 Parameter objects: load_state_dict, .to(), in-place edits).  The objects are looked up once -- walking the module
        tree costs the host ~0.1 ms per call, twice per frame, with the device idle -- and again whenever one of them was REPLACED
        (their ids are part of the key)."""
        plist = self.__dict__.get("_gpnerf_plist")
        if plist is None or any(getattr(m, n, None) is not p for (m, n, p) in plist):
            plist = []
            for _, name in L.HEAD_FIELDS:
                mod_name, idx = name.rsplit(".", 1)
                lin = self.get_submodule(mod_name)[int(idx)]
                plist += [(lin, "weight", lin.weight), (lin, "bias", lin.bias)]
            self.__dict__["_gpnerf_plist"] = plist
        key = (str(device),) + tuple((p.data_ptr(), p._version) for (_, _, p) in plist)
        if self._blob is None or self._blob_key != key:
            self._blob = F_.pack_head(self.per_ray_state(), device)
            self._blob_key = key
        return self._blob

    # ---- the reference's call ----------------------------------------------------------------
    def forward(self, sp_input, grid_coords, smpl_feat_sampled, rgb_feat, mask):
        """
        grid_coords [1, R*S, 3]; rgb_feat [R,S,V,35]; mask [R,S,V,1]   (BaseRender.py:125-140)
        sp_input['volumes'] (4 dense levels) short-cuts the per-frame volume builder.
        returns raw [R,S,4] = (rgb, sigma) and rgb_in [R,S,V,3].
        """
        dev = rgb_feat.device
        R, S = rgb_feat.shape[:2]
        vols = sp_input.get("volumes") if isinstance(sp_input, dict) else None
        if vols is None:
            vols = self.sigmahead.build_volumes(sp_input, smpl_feat_sampled)
        blob = self.head_blob(dev)
        fr = F_.Frame.for_volumes(vols, blob)
        vol_feat = F_.sample_volume(fr, grid_coords.reshape(-1, 3))
        raw = F_.head_forward(blob, vol_feat, rgb_feat.reshape(R * S, L.VIEWS, 35), mask.reshape(R * S, L.VIEWS))
        return raw.view(R, S, 4), rgb_feat[..., :3]


def build_head(cfg):
    """Same cfg keys as trainhead.py:166-177."""
    return NeRFHead(in_feat_ch=cfg.encoder.out_ch, use_rgbhead=cfg.head.rgb.use_rgbhead, n_smpl=cfg.head.sigma.n_smpl,
                    code_dim=cfg.head.sigma.code_dim, attn_n_heads=cfg.head.sigma.n_heads,
                    spconv_n_layers=cfg.head.sigma.n_layers, spconv_out_dim=cfg.head.sigma.outdims)
