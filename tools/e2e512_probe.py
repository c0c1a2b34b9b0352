"""Where does the error of the config-5-sized end-to-end case come from?  GPU encoder vs the torch-CPU restatement (full maps),
and the per-ray kernel fed with either set of feature maps, against tests/golden/e2e_512_survey.npz.  Diagnostic (uses oracle/)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gp-nerf_amd", "plugins")]
from golden_cases import load, scene_of
from oracle import producers_ref as ref
from test_gpu_renderer import cfg, batch_of, load_head
syn = importlib.import_module("gp-nerf_amd.synthetic")
hip_render = importlib.import_module("hip_render")
z, meta = load("e2e_512_survey")
sc = scene_of(meta)
sc["src_imgs"] = syn.make_encoder_images(512, 512, meta["seed"])[None]
c = cfg(n_samples=meta["n_samples"]); c.encoder.file = "hip_encoder"
r = hip_render.build_render(c).to("cuda:0").eval()
load_head(r, sc)
r.encoder.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_encoder_weights(meta["seed"]).items()}, strict=True)
b = batch_of(sc, with_products=False)
b["volumes"] = [torch.from_numpy(v).to("cuda:0") for v in sc["volumes"]]
b["mask_at_box"] = torch.from_numpy(sc["mask_at_box"]).to("cuda:0")
st = int(z["ray_stride"])
with torch.no_grad():
    cpu_net = importlib.import_module("gp-nerf_amd.encoder").ResUNet()
    cpu_net.load_state_dict({k: v.cpu() for k, v in r.encoder.state_dict().items()})
    fm_ref = ref.encoder(cpu_net.eval(), torch.from_numpy(sc["src_imgs"][0]))
    r.encoder.precision = "split"
    fm_gpu = r.encoder(b["src_imgs"][0])
    d = (fm_gpu.cpu() - fm_ref).abs()
    print("encoder gpu vs cpu restatement: max", float(d.max()), "mean", float(d.mean()), "absmax value", float(fm_ref.abs().max()))
    print("  per view max", [float(d[v].max()) for v in range(3)])
    print("  cpu restatement vs fixture subset", float((fm_ref[:, :, ::4, ::4] - torch.from_numpy(z["featmaps_sub"])).abs().max()))
    r.encoder.precision = "fp32"
    fm_exact = r.encoder(b["src_imgs"][0])
    r.encoder.precision = "split"
    de = (fm_exact.cpu() - fm_ref).abs()
    print("fp32-precision encoder vs cpu restatement: max", float(de.max()), "mean", float(de.mean()))
    for name, fm in (("gpu-encoder precision split", None), ("gpu encoder precision fp32 (default)", fm_exact), ("cpu-restatement featmaps", fm_ref.to("cuda:0"))):
        bb = dict(b)
        if fm is not None:
            bb["featmaps"] = fm
        for split in (False, True):
            r.split_f16 = split
            ret = r.render(bb)
            e = {k: float(np.abs(ret[k][0, ::st].cpu().numpy().reshape(z[k].shape) - z[k]).max()) for k in ("rgb_map", "depth_map", "acc_map", "rgb_in_map")}
            print(name, "split" if split else "fp32", e)
    r.split_f16 = False
