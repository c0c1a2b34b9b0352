"""`Evaluator` with the reference's interface (libs/evaluators/if_nerf.py:8-88), computing on the tensors' device
(SURVEY.md §8f-4): `evaluate(output, batch)` accumulates MSE / PSNR / SSIM of one rendered view, `summarize()` returns
and prints their means and resets.

PSNR = -10 log10(mean((pred - gt)^2)) over the `mask_at_box` pixels (if_nerf.py:15-18,59-63).
SSIM follows the call the reference makes, `skimage.measure.compare_ssim(pred, gt, multichannel=True)` on the
bounding-box crop of the mask with both images zero outside it (if_nerf.py:20-47): 7x7 uniform window, sample
covariance (x 49/48), K1 = 0.01, K2 = 0.03, data range 2 (skimage's range for float images), mean over the map cropped by
3 pixels and over channels.  scikit-image is not installed in this image, so that part is a restatement of the published
algorithm: **parity unpinned** (tests check it against a scipy `uniform_filter` restatement).
Image writing (cfg.test.save_imgs) is I/O and out of scope.

`evaluate_loop` is the evaluation loop around it (libs/trainers/BaseTrainer.py:255-280 `Trainer.evaluate`): per frame
`render.render(batch)` -> `Evaluator.evaluate`, the render time summed from `ret["rtime"]`, the means from `summarize()`;
pinned to the reference's own loop over three frames (tests/golden/loop_demo_3frames.npz).
"""
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

_WIN, _K1, _K2, _RANGE = 7, 0.01, 0.03, 2.0


def psnr_metric(pred, gt):
    """pred, gt: tensors of equal shape -> python float (if_nerf.py:15-18)."""
    mse = torch.mean((pred.double() - gt.double()) ** 2).item()
    return -10.0 * math.log(mse) / math.log(10.0) if mse > 0 else float("inf")


def ssim_images(a, b):
    """Mean SSIM of two [H,W,C] images (skimage compare_ssim defaults, multichannel).  float64 on the inputs' device."""
    if a.shape != b.shape or a.dim() != 3:
        raise ValueError(f"expected two [H,W,C] images, got {tuple(a.shape)} and {tuple(b.shape)}")
    if min(a.shape[0], a.shape[1]) < _WIN:
        raise ValueError("win_size exceeds image extent")      # skimage raises the same
    x = a.double().permute(2, 0, 1).unsqueeze(1)              # [C,1,H,W]
    y = b.double().permute(2, 0, 1).unsqueeze(1)
    box = lambda t: F.avg_pool2d(t, _WIN, stride=1)           # valid window means == the map cropped by (win-1)/2
    ux, uy = box(x), box(y)
    norm = _WIN * _WIN / (_WIN * _WIN - 1.0)
    vx, vy, vxy = norm * (box(x * x) - ux * ux), norm * (box(y * y) - uy * uy), norm * (box(x * y) - ux * uy)
    c1, c2 = (_K1 * _RANGE) ** 2, (_K2 * _RANGE) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    return s.mean(dim=(1, 2, 3)).mean().item()


def mask_bounding_rect(mask):
    """(x, y, w, h) of the non-zero pixels of a [H,W] bool tensor (what cv2.boundingRect returns); (0,0,0,0) if empty."""
    rows = torch.nonzero(mask.any(dim=1)).flatten()
    cols = torch.nonzero(mask.any(dim=0)).flatten()
    if rows.numel() == 0:
        return 0, 0, 0, 0
    y0, y1, x0, x1 = int(rows[0]), int(rows[-1]), int(cols[0]), int(cols[-1])
    return x0, y0, x1 - x0 + 1, y1 - y0 + 1


class Evaluator:
    def __init__(self, cfg, seq_name):
        self.cfg, self.seq_name = cfg, seq_name
        self.mse, self.psnr, self.ssim = [], [], []

    def _hw(self):
        d = self.cfg.dataset
        return int(d.H * d.ratio), int(d.W * d.ratio)

    def psnr_metric(self, img_pred, img_gt):
        return psnr_metric(torch.as_tensor(img_pred), torch.as_tensor(img_gt))

    def ssim_metric(self, rgb_pred, rgb_gt, batch):
        H, W = self._hw()
        rgb_pred, rgb_gt = torch.as_tensor(rgb_pred), torch.as_tensor(rgb_gt)
        mask = torch.as_tensor(batch["mask_at_box"][0]).reshape(H, W).to(rgb_pred.device).bool()
        pred = torch.zeros((H, W, 3), dtype=torch.float64, device=rgb_pred.device)
        gt = torch.zeros_like(pred)
        pred[mask] = rgb_pred.double()
        gt[mask] = rgb_gt.to(rgb_pred.device).double()
        x, y, w, h = mask_bounding_rect(mask)
        return ssim_images(pred[y:y + h, x:x + w], gt[y:y + h, x:x + w])

    def evaluate(self, output, batch):
        if "pred_img" not in output:
            rgb_pred = torch.as_tensor(output["rgb_map"][0]).detach()
        else:                                                     # progressive renderer (demo_render.py:359-364)
            H, W = self._hw()
            mask = torch.as_tensor(batch["mask_at_box"][0]).reshape(H, W).bool()
            img = torch.as_tensor(output["pred_img"])
            rgb_pred = img[mask.to(img.device)]
        rgb_gt = torch.as_tensor(batch["rgb"][0]).detach().to(rgb_pred.device)
        self.mse.append(torch.mean((rgb_pred.double() - rgb_gt.double()) ** 2).item())
        self.psnr.append(psnr_metric(rgb_pred, rgb_gt))
        self.ssim.append(self.ssim_metric(rgb_pred, rgb_gt, batch))

    def summarize(self):
        metrics = {"mse": float(np.mean(self.mse)), "psnr": float(np.mean(self.psnr)), "ssim": float(np.mean(self.ssim))}
        result_dir = getattr(self.cfg, "result_dir", None)
        if result_dir:                                            # if_nerf.py:72-80 keeps the per-view MSE list
            path = os.path.join(result_dir, self.seq_name, "metrics.npy")
            os.makedirs(os.path.dirname(path), exist_ok=True)
            np.save(path, self.mse)
        for k in ("mse", "psnr", "ssim"):
            print(f"{k}: {metrics[k]}")
        self.mse, self.psnr, self.ssim = [], [], []
        return metrics


def evaluate_loop(render, eval_loader, cfg, device=None, quiet=False, pipeline=None):
    """`Trainer.evaluate` (libs/trainers/BaseTrainer.py:255-280) without its image writing: for every batch of `eval_loader`
    move it to `device` (`_read_inputs`, :89-97), `ret = render.render(batch)` (the reference calls `.module.render` on its
    DataParallel wrapper; a wrapped model is unwrapped here too), `Evaluator.evaluate(ret, batch)`, `total_time += ret["rtime"]`;
    then `summarize()` when the head renders colour.  Returns {"count", "total_time", "avg_time", "metrics" (summarize()'s dict or
    None), "mse", "psnr", "ssim" (the per-frame lists), "wall_time" (the loop's own clock)} -- the reference prints the average and
    returns nothing.
    pipeline (not in the reference, whose loop is strictly serial): frame t + 1 is fetched, moved to the device and PREFETCHED
    (Renderer.prefetch: encoder graph, volume builder, frame glue on a second stream) right after frame t's per-ray kernel has been
    enqueued, so the device goes from one frame's per-ray kernel straight into the next frame's producers while the host evaluates
    frame t.  Default: on when the renderer offers `prefetch` and is neither progressive nor sharded.  Same bits per frame."""
    model = getattr(render, "module", render)
    model.eval()
    evaluator = Evaluator(cfg, cfg.test.test_seq)
    count, total_time = 0, 0.0
    if pipeline is None:
        pipeline = hasattr(model, "prefetch") and not getattr(model, "progressive", False) and getattr(model, "shard_group", None) is None

    def move(v):
        if device is None:
            return v
        if isinstance(v, (list, tuple)):
            return [b.to(device) for b in v]
        if isinstance(v, dict):
            return {k: b.to(device) for k, b in v.items()}
        return v.to(device)

    import time as _time
    t_loop = _time.time()
    if not pipeline:
        for data in eval_loader:
            with torch.no_grad():
                val = {k: move(v) for k, v in data.items()}
                ret = model.render(val)
                evaluator.evaluate(ret, val)
            total_time += ret["rtime"]                       # the dense renderer of the reference returns no "rtime": KeyError there
            count += 1
    else:
        it = iter(eval_loader)

        def fetch():
            data = next(it, None)
            if data is None:
                return None
            val = {k: move(v) for k, v in data.items()}
            if hasattr(model, "host_consts"):
                # the frame's small constants go to the host NOW, from the loader's CPU tensors when it hands out those (no copy at
                # all), otherwise while no per-ray kernel is running: inside prefetch() the copy would wait for that kernel
                src = data if all(not (isinstance(v, torch.Tensor) and v.is_cuda) for v in data.values()) else val
                val["_gpnerf_consts"] = model.host_consts(src)
            return val

        with torch.no_grad():
            val = fetch()
            pre = model.prefetch(val) if val is not None else None
            while val is not None:
                nxt = fetch()
                ret = model.render(val, prefetched=pre, next_batch=nxt)
                pre = ret.pop("next_prefetched", None)
                evaluator.evaluate(ret, val)
                total_time += ret["rtime"]
                count += 1
                val = nxt
    wall = _time.time() - t_loop
    per_frame = {"mse": list(evaluator.mse), "psnr": list(evaluator.psnr), "ssim": list(evaluator.ssim)}
    if quiet:                                                 # (summarize() prints its three means, as the reference's does)
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            metrics = evaluator.summarize() if cfg.head.rgb.use_rgbhead else None
    else:
        metrics = evaluator.summarize() if cfg.head.rgb.use_rgbhead else None
    if not quiet:
        print(f"avg total render time: {total_time / max(count, 1)}s per sample")
    return dict(count=count, total_time=total_time, avg_time=total_time / max(count, 1), metrics=metrics, wall_time=wall, **per_frame)
