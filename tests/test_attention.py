"""The volume builder's vertex-code attention (trainhead.py:48-52) against vectors captured from the reference's
MultiHeadAttention (tests/golden/make_golden.py run_attention_case): the oracle's torch restatement on CPU (which that pins),
the fused HIP kernel -- the product's only path -- on the GPU."""
import hashlib
import importlib

import numpy as np
import pytest
import torch

from golden_cases import assert_close, attention_case_names, load

syn = importlib.import_module("gp-nerf_amd.synthetic")
vol = importlib.import_module("gp-nerf_amd.volume")


def _case(meta):
    state, code, feat = syn.make_attention_case(meta["n"], meta["code_dim"], meta["seed"])
    h = hashlib.sha256()
    for a in [code, feat] + [state[k] for k in sorted(state)]:
        h.update(np.ascontiguousarray(a).tobytes())
    assert h.hexdigest() == meta["inputs_sha256"]
    d = meta["code_dim"]
    m = vol.MultiHeadAttention(4, d, d // 4, d // 4, kv_dim=32, sum=False)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)      # the reference's keys
    return m.eval(), torch.from_numpy(code), torch.from_numpy(feat)


@pytest.mark.parametrize("name", attention_case_names())
def test_attention_restatement_matches_reference_golden_cpu(name):
    from oracle import producers_ref as ref
    z, meta = load(name)
    m, code, feat = _case(meta)                 # the product's module: the reference's parameter names, loaded strictly
    with torch.no_grad():
        out = ref.attention(m, code.unsqueeze(1), feat, feat)[0].squeeze(1).numpy()
    assert_close(out, z["out"], 2e-6, "attention")


@pytest.mark.gpu
@pytest.mark.parametrize("name", attention_case_names())
def test_fused_attention_kernel_matches_reference_golden(name):
    z, meta = load(name)
    m, code, feat = _case(meta)
    m = m.to("cuda:0")
    with torch.no_grad():
        out = m.fuse_vertices(code.to("cuda:0"), feat.to("cuda:0")).cpu().numpy()
        out2, attn = m(code.to("cuda:0").unsqueeze(1), feat.to("cuda:0"), feat.to("cuda:0"))      # the reference's call form
    assert_close(out, z["out"], 2e-5, "attention (HIP)")
    assert attn is None and out2.shape == (code.shape[0], 1, code.shape[1]) and np.array_equal(out2[:, 0].cpu().numpy(), out)


def test_fused_attention_rejects_unsupported_shapes():
    L = importlib.import_module("gp-nerf_amd._lib")
    lib = L.lib()
    p = 0x1000
    assert lib.gpnerf_vertex_attention(p, p, p, p, p, p, 8, 24, 32, 4, 3, p, None) == -1     # d_k = 6: not a power of two
    assert lib.gpnerf_vertex_attention(p, p, p, p, p, p, 8, 128, 32, 4, 3, p, None) == -1    # d_model > 64
    assert lib.gpnerf_vertex_attention(p, p, p, p, p, p, 8, 32, 32, 4, 5, p, None) == -1     # views > 4
    assert lib.gpnerf_vertex_attention(None, p, p, p, p, p, 0, 32, 32, 4, 3, p, None) == 0    # nothing to do


@pytest.mark.gpu
@pytest.mark.parametrize("n_head,views,n", [(4, 3, 6890), (8, 3, 100), (2, 4, 33), (1, 2, 64), (4, 1, 31)])
def test_matrix_core_attention_for_every_head_size_and_view_count(n_head, views, n):
    """gpnerf_vertex_attention at d_model = kv_dim = 32 (the matrix-core form): heads of 4, 8, 16 and 32 channels, 1-4 views, a last
    tile that is not full -- against the formula in float64 (MultiHeadAttention.py:61-98, sum=False, one query per vertex)."""
    L = importlib.import_module("gp-nerf_amd._lib")
    lib = L.lib()
    g = torch.Generator().manual_seed(n_head * 100 + views)
    d = 32
    q = torch.randn((n, d), generator=g)
    kv = torch.randn((n, views, d), generator=g)
    w = [torch.randn((d, d), generator=g) * 0.25 for _ in range(4)]
    qd, kd = q.double(), kv.double()
    d_k = d // n_head
    qh = (qd @ w[0].double().T).view(n, n_head, d_k) / d_k ** 0.5
    kh = (kd @ w[1].double().T).view(n, views, n_head, d_k)
    vh = (kd @ w[2].double().T).view(n, views, n_head, d_k)
    att = torch.softmax(torch.einsum("nhd,nvhd->nhv", qh, kh), dim=-1)
    ref = (torch.einsum("nhv,nvhd->nhd", att, vh).reshape(n, d) @ w[3].double().T).float()
    dev = "cuda:0"
    qg, kg, wg = q.to(dev), kv.to(dev), [t.to(dev).contiguous() for t in w]
    out = torch.empty((n, d), device=dev)
    L.check(lib.gpnerf_vertex_attention(qg.data_ptr(), kg.data_ptr(), wg[0].data_ptr(), wg[1].data_ptr(), wg[2].data_ptr(), wg[3].data_ptr(),
                                        n, d, d, n_head, views, out.data_ptr(), None), "gpnerf_vertex_attention")
    torch.cuda.synchronize()
    assert float((out.cpu() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
