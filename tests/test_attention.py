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
