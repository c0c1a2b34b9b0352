"""CPU: the throughput twin (oracle/gpnerf_cpu_blocked.c, bench.py's `cpu_baseline` kind "port-blocked") against the oracle and,
through it, against the reference's golden vectors."""
import numpy as np
import pytest

from golden_cases import assert_close, load, scene_of


@pytest.mark.parametrize("name", ["base_s32", "base_s64", "base_s8", "neg_s32", "allmasked_s32", "partial_s32", "stretch_s32", "nonsquare_s16"])
def test_blocked_twin_matches_the_oracle_and_the_reference(name, oracle):
    from oracle import blocked
    z, meta = load(name)
    sc = scene_of(meta)
    S = meta["n_samples"]
    got = blocked.render(blocked.Frame(sc), oracle.rays_of(sc), S, neg_ray=meta["neg_ray"])
    ref = oracle.render(sc, S, neg_ray=meta["neg_ray"])
    for k in ("rgb_map", "depth_map", "acc_map", "rgb_in_map", "weights"):
        tol = 5e-5 if k == "depth_map" else 1e-5          # depth is ~3: the same relative error as the other maps
        assert_close(got[k], ref[k], tol, k + " vs the oracle")
        assert_close(got[k], z[k], 2 * tol, k + " vs the reference's vector")
    assert np.array_equal(got["ray_mask"], ref["ray_mask"])
    assert_close(got["disp_map"], ref["disp_map"], 1e-4, "disp_map")


def test_blocked_twin_ragged_blocks(oracle, syn):
    """ray counts that do not fill a block, sample counts that are not a multiple of the vector length"""
    from oracle import blocked
    for S, n in ((1, 5), (7, 31), (24, 3), (100, 9)):
        sc = syn.make_scene(H=16, W=16, seed=40 + S, fill="full", pose="random", aabb_half=(0.12, 0.16, 0.05), bias_std=0.1, max_rays=n)
        got = blocked.render(blocked.Frame(sc), oracle.rays_of(sc), S)
        ref = oracle.render(sc, S)
        for k in ("rgb_map", "depth_map", "acc_map", "weights"):
            assert_close(got[k], ref[k], 5e-5 if k == "depth_map" else 1e-5, f"{k} S={S} n={n}")
