# the stem kernel with its weights in LDS (product) or read from global memory by every wave (-DGPNERF_STEM_WGLOBAL: 24 KB of LDS,
# several workgroups per CU): encoder time and the stem's own time per build (builds on the GPU box)
cd "$GRAFT_REPO_ROOT"; mkdir -p /tmp/ab
C=gp-nerf_amd/csrc
hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -c -o /tmp/ab/conv_wg.o $C/gpnerf_conv.hip &&
hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/ab/lib_wg.so $C/gpnerf_kernels.o $C/gpnerf_volume.o /tmp/ab/conv_wg.o
cat > /tmp/ab/stem_time.py <<'PY'
import importlib, os, sys, torch
sys.path[:0] = [os.environ["GRAFT_REPO_ROOT"]]
enc = importlib.import_module("gp-nerf_amd.encoder")
dev = "cuda:0"
conv = torch.nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False, padding_mode="reflect").to(dev)
norm = torch.nn.InstanceNorm2d(64, affine=True).to(dev)
x = torch.randn((3, 3, 512, 512), device=dev).contiguous(memory_format=torch.channels_last)
def t(fn, per_graph=20, replays=10):
    with torch.no_grad():
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): fn()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(per_graph): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(replays): g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (per_graph * replays) * 1e3
print(f"stem 3x512x512 -> 64 channels: conv + sums + table {t(lambda: enc._conv_norm(conv, norm, x)):.1f} us")
PY
for v in product wg product wg; do
  if [ $v = product ]; then L=$PWD/$C/libgpnerf_hip.so; else L=/tmp/ab/lib_wg.so; fi
  echo "== $v"; GPNERF_DEBUG=1 GPNERF_LIB_PATH=$L python /tmp/ab/stem_time.py 2>&1 | tail -1; GPNERF_DEBUG=1 GPNERF_LIB_PATH=$L python tools/probes/encoder_time.py 2>&1 | tail -1 | cut -c1-42
done
