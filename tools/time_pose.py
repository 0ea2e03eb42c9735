"""Diagnostic: pose preparation of one training step (two sources, batch 12), forward + backward, on the HIP
library (one launch each way) against the same maths in eager torch ops on the GPU (what the reference runs,
Python/layers.py:28-103: about twenty small kernels per source and direction)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mde_amd.layers as L  # noqa: E402

DEV = "cuda:0"


def eager_pose(vec, translation, invert=False):
    angle = torch.norm(vec, 2, 2, True)
    axis = vec / (angle + 1e-7)
    ca, sa = torch.cos(angle), torch.sin(angle)
    C = 1 - ca
    x, y, z = (axis[..., i].unsqueeze(1) for i in range(3))
    xs, ys, zs, xC, yC, zC = x * sa, y * sa, z * sa, x * C, y * C, z * C
    xyC, yzC, zxC = x * yC, y * zC, z * xC
    rot = torch.zeros((vec.shape[0], 4, 4), device=vec.device)
    rot[:, 0, 0] = torch.squeeze(x * xC + ca)
    rot[:, 0, 1] = torch.squeeze(xyC - zs)
    rot[:, 0, 2] = torch.squeeze(zxC + ys)
    rot[:, 1, 0] = torch.squeeze(xyC + zs)
    rot[:, 1, 1] = torch.squeeze(y * yC + ca)
    rot[:, 1, 2] = torch.squeeze(yzC - xs)
    rot[:, 2, 0] = torch.squeeze(zxC - ys)
    rot[:, 2, 1] = torch.squeeze(yzC + xs)
    rot[:, 2, 2] = torch.squeeze(z * zC + ca)
    rot[:, 3, 3] = 1
    t = translation.clone()
    if invert:
        rot = rot.transpose(1, 2)
        t = t * -1
    T = torch.zeros(t.shape[0], 4, 4, device=t.device)
    for i in range(4):
        T[:, i, i] = 1
    T[:, :3, 3, None] = t.contiguous().view(-1, 3, 1)
    return torch.matmul(rot, T) if invert else torch.matmul(T, rot)


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


def main():
    B = 12
    out = 0.01 * torch.randn(B, 3, 1, 6, device=DEV)
    g = [torch.randn(B, 4, 4, device=DEV) for _ in range(2)]

    def run(fn_all):
        o = out.clone().requires_grad_(True)
        aa, tr = o[..., :3], o[..., 3:]
        Ts = fn_all(aa, tr)
        torch.autograd.backward(Ts, g)
        return o.grad

    hip = lambda aa, tr: L.poses_from_parameters([aa[:, 0], aa[:, 1]], [tr[:, 0], tr[:, 1]], [False, False])
    eager = lambda aa, tr: [eager_pose(aa[:, 0], tr[:, 0]), eager_pose(aa[:, 1], tr[:, 1])]
    a, b = run(hip), run(eager)
    print("max |grad difference| hip vs eager: %.2e" % float((a - b).abs().max()))
    print("pose preparation fwd+bwd, 2 sources, batch %d: HIP %.1f us, eager torch %.1f us (host-side time per call, "
          "GPU idle otherwise)" % (B, timeit(lambda: run(hip)), timeit(lambda: run(eager))))


main()
