"""Evaluator: PSNR by the reference's formula, SSIM against a scipy restatement of the published
skimage algorithm (scikit-image is not installed here: unpinned)."""
import importlib
import math
import types

import numpy as np
import pytest
import torch
from scipy.ndimage import uniform_filter

ev = importlib.import_module("gp-nerf_amd.evaluator")


def ssim_restated(a, b):
    """compare_ssim(a, b, multichannel=True), float64 images -> data_range 2, win 7, sample covariance, crop 3."""
    vals = []
    for c in range(a.shape[2]):
        x, y = a[..., c].astype(np.float64), b[..., c].astype(np.float64)
        ux, uy = uniform_filter(x, 7), uniform_filter(y, 7)
        n = 49.0 / 48.0
        vx, vy = n * (uniform_filter(x * x, 7) - ux * ux), n * (uniform_filter(y * y, 7) - uy * uy)
        vxy = n * (uniform_filter(x * y, 7) - ux * uy)
        c1, c2 = (0.01 * 2) ** 2, (0.03 * 2) ** 2
        s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
        vals.append(s[3:-3, 3:-3].mean())
    return float(np.mean(vals))


def _case(H=40, W=56, seed=0):
    g = np.random.Generator(np.random.PCG64(seed))
    mask = np.zeros((H, W), bool)
    mask[6:31, 9:44] = g.uniform(size=(25, 35)) < 0.8
    mask[6, 9] = mask[30, 43] = True
    n = int(mask.sum())
    gt = g.uniform(0, 1, (n, 3)).astype(np.float32)
    pred = np.clip(gt + 0.05 * g.standard_normal((n, 3)).astype(np.float32), 0, 1)
    cfg = types.SimpleNamespace(dataset=types.SimpleNamespace(H=H * 2, W=W * 2, ratio=0.5))
    batch = {"mask_at_box": torch.from_numpy(mask.reshape(1, -1)), "rgb": torch.from_numpy(gt)[None]}
    return cfg, batch, mask, pred, gt


def test_psnr_and_ssim_follow_the_reference_formulas():
    cfg, batch, mask, pred, gt = _case()
    e = ev.Evaluator(cfg, "seq")
    e.evaluate({"rgb_map": torch.from_numpy(pred)[None]}, batch)
    mse = np.mean((pred - gt) ** 2)
    assert abs(e.mse[0] - mse) < 1e-8
    assert abs(e.psnr[0] - (-10 * np.log(mse) / np.log(10))) < 1e-5     # the reference means in float32, this in float64
    H, W = mask.shape
    a, b = np.zeros((H, W, 3)), np.zeros((H, W, 3))
    a[mask], b[mask] = pred, gt
    assert abs(e.ssim[0] - ssim_restated(a[6:31, 9:44], b[6:31, 9:44])) < 1e-9
    m = e.summarize()
    assert set(m) == {"mse", "psnr", "ssim"} and e.mse == []


def test_pred_img_branch_equals_rgb_map_branch():
    cfg, batch, mask, pred, gt = _case(seed=3)
    img = np.zeros(mask.shape + (3,), np.float32)
    img[mask] = pred
    a, b = ev.Evaluator(cfg, "s"), ev.Evaluator(cfg, "s")
    a.evaluate({"rgb_map": torch.from_numpy(pred)[None]}, batch)
    b.evaluate({"pred_img": img}, batch)
    assert a.psnr == b.psnr and a.ssim == b.ssim


def test_identical_images_and_bounding_rect():
    x = torch.rand(16, 20, 3)
    assert abs(ev.ssim_images(x, x) - 1.0) < 1e-12
    assert ev.psnr_metric(x, x) == math.inf
    m = torch.zeros(10, 12, dtype=torch.bool)
    assert ev.mask_bounding_rect(m) == (0, 0, 0, 0)
    m[2, 5] = m[7, 3] = True
    assert ev.mask_bounding_rect(m) == (3, 2, 3, 6)
    with pytest.raises(ValueError):
        ev.ssim_images(torch.rand(5, 20, 3), torch.rand(5, 20, 3))


def test_ssim_closed_forms():
    """skimage is not in the image, so `ssim_images` cannot be pinned to it; what CAN be checked without it is that the
    restatement IS the published formula (Wang et al. 2004, skimage's defaults for float images: 7x7 uniform window, sample
    covariance, K1 = 0.01, K2 = 0.03, L = 2) on inputs whose SSIM has a closed form:
      * identical images: every window gives (2 mu^2 + c1)(2 s^2 + c2) / ((2 mu^2 + c1)(2 s^2 + c2)) = 1;
      * a constant image a against a + c: all (co)variances vanish, SSIM = (2 a (a + c) + c1) / (a^2 + (a + c)^2 + c1), c1 = (K1 L)^2;
      * x against its negative -x (zero-mean windows are not needed): numerator (-2 mu^2 + c1)(-2 s^2 + c2), i.e. for a checkerboard of
        +-v, whose every 7x7 window has the same |mean| = v / 49 and sample variance s^2 = (49 v^2 - v^2 / 49) / 48."""
    c1, c2 = (0.01 * 2.0) ** 2, (0.03 * 2.0) ** 2
    x = torch.rand(20, 24, 3)
    assert abs(ev.ssim_images(x, x) - 1.0) < 1e-12
    for a, c in ((0.3, 0.1), (0.0, 0.5), (0.7, -0.7), (-0.2, 0.05)):
        want = (2 * a * (a + c) + c1) / (a * a + (a + c) ** 2 + c1)
        got = ev.ssim_images(torch.full((9, 11, 3), a, dtype=torch.float64), torch.full((9, 11, 3), a + c, dtype=torch.float64))
        assert abs(got - want) < 1e-12, (a, c, got, want)
    v = 0.25
    yy, xx = torch.meshgrid(torch.arange(15), torch.arange(17), indexing="ij")
    board = (v * (1 - 2 * ((yy + xx) % 2))).double()[..., None].repeat(1, 1, 3)
    mu2 = (v / 49.0) ** 2                                    # 25 cells of one sign, 24 of the other
    s2 = (49 * v * v - 49 * mu2) / 48.0                      # sum x^2 - n mu^2 over n - 1
    want = ((-2 * mu2 + c1) * (-2 * s2 + c2)) / ((2 * mu2 + c1) * (2 * s2 + c2))
    assert abs(ev.ssim_images(board, -board) - want) < 1e-12
    # symmetric, and bounded by 1
    y = torch.rand(20, 24, 3)
    assert abs(ev.ssim_images(x, y) - ev.ssim_images(y, x)) < 1e-12 and ev.ssim_images(x, y) < 1.0


def test_psnr_matches_the_reference_evaluators_value():
    """tests/golden/e2e_64x64_s32.npz carries the value libs/evaluators/if_nerf.py Evaluator.psnr_metric returned for the
    reference's own render against a seeded ground truth; the device-side formula reproduces it on the same arrays."""
    import os
    import numpy as np
    import torch
    from golden_cases import GOLDEN_DIR
    ev = importlib.import_module("gp-nerf_amd.evaluator")
    z = np.load(os.path.join(GOLDEN_DIR, "e2e_64x64_s32.npz"))
    got = ev.psnr_metric(torch.from_numpy(z["rgb_map"]), torch.from_numpy(z["rgb_gt"]))
    assert abs(got - float(z["psnr"])) < 1e-4           # the reference averages in float32, this in float64
    e = ev.Evaluator(types.SimpleNamespace(dataset=types.SimpleNamespace(H=64, W=64, ratio=1.0)), "seq")
    e.evaluate({"rgb_map": torch.from_numpy(z["rgb_map"])[None]}, {"rgb": torch.from_numpy(z["rgb_gt"])[None], "mask_at_box": torch.ones((1, 64 * 64), dtype=torch.bool)})
    m = e.summarize()
    assert abs(m["psnr"] - float(z["psnr"])) < 1e-4 and abs(m["mse"] - float(z["mse"])) < 1e-8
