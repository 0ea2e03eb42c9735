"""Diagnostic: coarse-scale feature warp, forward + backward, at the three sizes of Train_distillation_new.py:1159-1182
(batch 12, 128 channels): the fused pass, the module-by-module kernels of this library, and eager torch ops.
GPU time per call from torch events (default stream), median of several rounds."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mde_amd.layers as L  # noqa: E402
import mde_amd.synth as synth  # noqa: E402

DEV = "cuda:0"


def eager(feats, disp, K, inv_K, T):
    B, C, H, W = feats.shape
    depth = 1 / (0.01 + (10.0 - 0.01) * disp)
    ys, xs = torch.meshgrid(torch.arange(H, device=DEV, dtype=torch.float32), torch.arange(W, device=DEV, dtype=torch.float32),
                            indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(H * W, device=DEV)], 0).unsqueeze(0).expand(B, 3, H * W)
    cam = depth.view(B, 1, -1) * torch.matmul(inv_K[:, :3, :3], pix)
    cam = torch.cat([cam, torch.ones(B, 1, H * W, device=DEV)], 1)
    P = torch.matmul(K, T)[:, :3, :]
    c = torch.matmul(P, cam)
    uv = (c[:, :2] / (c[:, 2:3] + 1e-7)).view(B, 2, H, W).permute(0, 2, 3, 1)
    uv = (uv / torch.tensor([W - 1, H - 1], device=DEV, dtype=torch.float32) - 0.5) * 2
    return F.grid_sample(feats, uv, padding_mode="border", align_corners=False)


def modules(feats, disp, K, inv_K, T):
    B, C, H, W = feats.shape
    _, depth = L.disp_to_depth(disp, 0.1, 100.0)
    return L.grid_sample_border(feats, L.Project3D(B, H, W)(L.BackprojectDepth(B, H, W)(depth, inv_K), K, T))


def gpu_us(fn, n=50):
    for _ in range(10):
        fn()
    times = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times.append(1e3 * e0.elapsed_time(e1) / n)
    return sorted(times)[len(times) // 2]


def main():
    B, C = 12, 128
    for H, W in ((24, 80), (12, 40), (6, 20)):
        data = synth.make_batch(B, H, W, [0, -1], seed=1)
        K, inv_K = data["K"].to(DEV), data["inv_K"].to(DEV)
        g_out = torch.randn(B, C, H, W, device=DEV)
        row = []
        for name, fn in (("fused", lambda f, d, t: L.warp_features(f, d, K, inv_K, t)), ("modules", lambda f, d, t: modules(f, d, K, inv_K, t)),
                         ("eager torch", lambda f, d, t: eager(f, d, K, inv_K, t))):
            def step():
                f = torch.randn(B, C, H, W, device=DEV).requires_grad_(True) if False else feats
                for x in (feats, disp, T):
                    x.grad = None
                fn(feats, disp, T).backward(g_out)
            feats = torch.randn(B, C, H, W, device=DEV, requires_grad=True)
            disp = data[("disp", 0)].to(DEV).requires_grad_(True)
            T = data[("cam_T_cam", -1)].to(DEV).requires_grad_(True)
            row.append("%s %.0f us" % (name, gpu_us(step)))
        print("%dx%d b%d C%d fwd+bwd: %s" % (H, W, B, C, ", ".join(row)))


main()
