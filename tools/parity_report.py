import sys, importlib, numpy as np, torch
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from golden_cases import *
fm=importlib.import_module('gp-nerf_amd.frame')
def to_dev(a): return torch.from_numpy(np.ascontiguousarray(a)).to('cuda:0')
def e(a,b):
    a=np.asarray(a,np.float64);b=np.asarray(b,np.float64);m=~(np.isnan(a)|np.isnan(b));return np.abs(a[m]-b[m]).max() if m.any() else 0
for name in case_names():
    z,meta=load(name); sc=scene_of(meta)
    blob=fm.pack_head(sc['head'],torch.device('cuda:0'))
    fr=fm.Frame(to_dev(sc['src_imgs'][0]),to_dev(sc['featmaps']),[to_dev(v) for v in sc['volumes']],to_dev(sc['src_Ks'][0]),to_dev(sc['src_poses'][0]),sc['Rh'][0],sc['Th'][0],sc['bounds'][0,0],sc['voxel_size'],sc['out_sh'][0],blob)
    rays=to_dev(np.concatenate([sc['ray_o'][0],sc['ray_d'][0],sc['near'][0][:,None],sc['far'][0][:,None]],1))
    out=[]
    for sp in (False,True):
        g={k:v.cpu().numpy() for k,v in fm.render_fused(fr,rays,meta['n_samples'],neg_ray=meta['neg_ray'],split_f16=sp).items()}
        out.append((e(g['rgb_map'],z['rgb_map']),e(g['depth_map'],z['depth_map'])))
    print(f"{name:20s} fp32 rgb {out[0][0]:.1e} depth {out[0][1]:.1e} | split rgb {out[1][0]:.1e} depth {out[1][1]:.1e}")
