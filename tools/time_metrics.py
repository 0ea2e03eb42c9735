"""Diagnostic: Trainer.compute_depth_losses at 640x192 b12 against a 375x1242 ground truth with 4 % valid pixels:
the HIP call (no host synchronisation) against the reference's sequence in eager torch ops on the GPU (boolean-mask
gathers synchronise the host, the medians sort).  Wall time per call including the final read-back of the metrics."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mde_amd.layers as L  # noqa: E402

DEV = "cuda:0"


def eager(depth, depth_gt):
    pred = torch.clamp(F.interpolate(depth, [375, 1242], mode="bilinear", align_corners=False), 1e-3, 80).detach()
    mask = depth_gt > 0
    crop = torch.zeros_like(mask)
    crop[:, :, 153:371, 44:1197] = 1
    mask = mask * crop
    gt, pred = depth_gt[mask], pred[mask]
    pred = torch.clamp(pred * (torch.median(gt) / torch.median(pred)), min=1e-3, max=80)
    thresh = torch.max(gt / pred, pred / gt)
    out = [torch.mean(torch.abs(gt - pred) / gt), torch.mean((gt - pred) ** 2 / gt), torch.sqrt(((gt - pred) ** 2).mean()),
           torch.sqrt(((torch.log(gt) - torch.log(pred)) ** 2).mean()), (thresh < 1.25).float().mean(),
           (thresh < 1.25 ** 2).float().mean(), (thresh < 1.25 ** 3).float().mean()]
    return torch.stack(out)


def wall_us(fn, n=100):
    for _ in range(10):
        fn().cpu()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn().cpu()
    return 1e6 * (time.perf_counter() - t0) / n


def gpu_us(fn, n=100):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def main():
    B = 12
    depth = 1.0 / (0.01 + 9.99 * torch.rand(B, 1, 192, 640, device=DEV))
    gt = (1.0 + 70.0 * torch.rand(B, 1, 375, 1242, device=DEV) ** 2) * (torch.rand(B, 1, 375, 1242, device=DEV) < 0.04)
    a, b = L.depth_metrics(depth, gt)[:7], eager(depth, gt)
    print("max relative difference hip vs eager: %.2e" % float(((a - b).abs() / b.abs()).max()))
    print("depth metrics 640x192 b%d: HIP %.0f us per call with read-back (%.0f us back to back on the stream), "
          "eager torch %.0f us per call" % (B, wall_us(lambda: L.depth_metrics(depth, gt)),
                                            gpu_us(lambda: L.depth_metrics(depth, gt)), wall_us(lambda: eager(depth, gt))))


main()
