"""Device time of gpnerf_vertex_attention on 6 890 vertices x 3 views x 32 channels (graph-timed, 20 per graph)."""
import ctypes as C, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
L = importlib.import_module("gp-nerf_amd._lib")
lib = L.lib()
dev = "cuda:0"
n, d, V = 6890, 32, 3
q = torch.randn((n, d), device=dev); kv = torch.randn((n, V, d), device=dev)
w = [torch.randn((d, d), device=dev) * 0.2 for _ in range(4)]
out = torch.empty((n, d), device=dev)
def fn():
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.gpnerf_vertex_attention(q.data_ptr(), kv.data_ptr(), w[0].data_ptr(), w[1].data_ptr(), w[2].data_ptr(), w[3].data_ptr(),
                                        n, d, d, 4, V, out.data_ptr(), st), "att")
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): fn()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): fn()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): g.replay()
e1.record(); torch.cuda.synchronize()
print(f"blocks {os.environ.get('GPNERF_ATT_BLOCKS', 'default')}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us, checksum {float(out.double().sum()):.6f}")
