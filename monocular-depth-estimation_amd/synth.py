"""Deterministic synthetic KITTI-shaped inputs for the view-synthesis loss.

A counter-based hash generator (splitmix64 finaliser over ``seed``/``index``)
replaces every library RNG, so the development container and the GPU box build
bit-identical tensors from the same (seed, shape) without sharing files.

Conventions follow the reference data loader (Python/data_loaders.py:333-341,
384-399): images are min-max style values in [0,1); the intrinsics are the KITTI
normalised matrix scaled by ``width // 2`` and ``height // 2`` and
``inv_K = pinv(K)``.
"""
from __future__ import annotations

from typing import Dict, Sequence

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix(z: np.ndarray) -> np.ndarray:
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _MASK
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def hash_bits(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """n 64-bit words, a pure function of (seed, stream, index)."""
    with np.errstate(over="ignore"):
        base = _splitmix(np.array([seed], dtype=np.uint64) ^ _splitmix(np.array([stream + 0x51], dtype=np.uint64)))
        idx = np.arange(n, dtype=np.uint64)
        return _splitmix(idx ^ base)


def uniform(seed: int, shape: Sequence[int], stream: int = 0) -> np.ndarray:
    """float64 array, U[0,1) with 24 random bits (exactly representable in fp32)."""
    n = int(np.prod(shape))
    bits = hash_bits(seed, n, stream) >> np.uint64(40)
    return (bits.astype(np.float64) / float(1 << 24)).reshape(shape)


def normal(seed: int, shape: Sequence[int], stream: int = 0) -> np.ndarray:
    """float64 N(0,1) by Box-Muller over two hashed uniforms."""
    n = int(np.prod(shape))
    u1 = (hash_bits(seed, n, 2 * stream + 101) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    u2 = (hash_bits(seed, n, 2 * stream + 102) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return (r * np.cos(2.0 * np.pi * u2)).reshape(shape)


def box_blur(a: np.ndarray, k: int = 5) -> np.ndarray:
    """k x k mean filter over the last two axes, edge-replicated."""
    if k <= 1:
        return a
    r = k // 2
    pad = [(0, 0)] * (a.ndim - 2) + [(r, r), (r, r)]
    p = np.pad(a, pad, mode="edge")
    h, w = a.shape[-2:]
    out = np.zeros_like(a)
    for dy in range(k):
        for dx in range(k):
            out += p[..., dy:dy + h, dx:dx + w]
    return out / float(k * k)


def kitti_intrinsics(batch: int, height: int, width: int):
    """K and pinv(K) as [B,4,4] float32 (data_loaders.py:333-336,388-396)."""
    K = np.array([[0.58, 0, 0.5, 0],
                  [0, 1.92, 0.5, 0],
                  [0, 0, 1, 0],
                  [0, 0, 0, 1]], dtype=np.float32)
    K[0, :] *= width // 2
    K[1, :] *= height // 2
    inv_K = np.linalg.pinv(K)
    Kt = torch.from_numpy(K).unsqueeze(0).repeat(batch, 1, 1).contiguous()
    iKt = torch.from_numpy(inv_K.astype(np.float32)).unsqueeze(0).repeat(batch, 1, 1).contiguous()
    return Kt, iKt


def _rodrigues(axisangle: np.ndarray, translation: np.ndarray) -> np.ndarray:
    """float32 [B,4,4] = translate(t) @ rotate(axisangle), the non-inverted form the
    trainer builds for pose_model_input == "all" (Trainer.py:414)."""
    b = axisangle.shape[0]
    v = axisangle.astype(np.float32)
    ang = np.sqrt((v * v).sum(-1, keepdims=True)).astype(np.float32)
    ax = v / (ang + np.float32(1e-7))
    ca, sa = np.cos(ang)[:, 0], np.sin(ang)[:, 0]
    C = 1 - ca
    x, y, z = ax[:, 0], ax[:, 1], ax[:, 2]
    M = np.zeros((b, 4, 4), dtype=np.float32)
    M[:, 0, 0] = x * x * C + ca
    M[:, 0, 1] = x * y * C - z * sa
    M[:, 0, 2] = z * x * C + y * sa
    M[:, 1, 0] = x * y * C + z * sa
    M[:, 1, 1] = y * y * C + ca
    M[:, 1, 2] = y * z * C - x * sa
    M[:, 2, 0] = z * x * C - y * sa
    M[:, 2, 1] = y * z * C + x * sa
    M[:, 2, 2] = z * z * C + ca
    M[:, 3, 3] = 1
    M[:, :3, 3] = translation.astype(np.float32)
    return M


def smooth_disp(batch: int, height: int, width: int, seed: int = 0, head: int = 0, n_patches: int = 6) -> torch.Tensor:
    """[B,1,H,W] float32 disparity of a street-like scene: a ground plane below a horizon (disparity grows
    linearly with the image row, as a plane seen by a pinhole camera does), a far background above it and a few
    fronto-parallel rectangles of constant disparity standing on the ground.  Piecewise smooth, the opposite extreme of
    make_batch's blurred noise: neighbouring pixels sample the source frames next to each other.  Heads differ by
    a few per cent in the ground slope and the patch disparities."""
    ys = np.arange(height, dtype=np.float64)[:, None] / float(height - 1)
    out = np.empty((batch, 1, height, width), dtype=np.float64)
    u = uniform(seed + 5, (batch, n_patches, 5), stream=60)
    wob = uniform(seed + 5, (batch, 2 + n_patches), stream=61 + head)
    for b in range(batch):
        horizon = 0.35 + 0.1 * u[b, 0, 0]
        slope = (0.75 + 0.1 * wob[b, 0]) / (1.0 - horizon)
        d = np.where(ys > horizon, 0.06 + slope * (ys - horizon), 0.06) + np.zeros((1, width))
        for p in range(n_patches):
            cx, foot, w_rel, h_rel, _ = u[b, p]
            foot_y = horizon + (1.0 - horizon) * (0.15 + 0.8 * foot)        # where the patch touches the ground
            dp = (0.06 + slope * (foot_y - horizon)) * (1.0 + 0.03 * (wob[b, 2 + p] - 0.5))
            x0 = int(cx * width * 0.9)
            x1 = min(width, x0 + 8 + int(w_rel * width * 0.15))
            y1 = int(foot_y * (height - 1))
            y0 = max(0, y1 - 4 - int(h_rel * height * 0.4))
            d[y0:y1 + 1, x0:x1] = np.maximum(d[y0:y1 + 1, x0:x1], dp)        # nearer surfaces occlude
        out[b, 0] = np.clip(d, 0.05, 0.95)
    return torch.from_numpy(out.astype(np.float32)).contiguous()


def make_batch(batch: int, height: int, width: int, frame_ids: Sequence = (0, -1, 1),
               seed: int = 0, pose_sigma: float = 0.01, n_heads: int = 1,
               image_dtype: torch.dtype = torch.float32, blur: int = 5, disp: str = "noise") -> Dict:
    """Synthetic inputs for one step.

    Returns a dict with
      ("image", f)   [B,3,H,W]  image_dtype, values exactly representable in fp32
      ("disp", h)    [B,1,H,W]  float32 in (0.05, 0.95), one per head
      "K", "inv_K"   [B,4,4]    float32
      ("cam_T_cam", f) [B,4,4]  float32 for temporal sources, "stereo_T" for 's'
      "noise"        [B,S,H,W]  float32 N(0,1) (tie-break noise of Trainer.py:589)
    ``disp``: "noise" = box-blurred uniform noise (the benchmark's default), "smooth" = smooth_disp().
    """
    out: Dict = {}
    for i, f in enumerate(frame_ids):
        img = box_blur(uniform(seed, (batch, 3, height, width), stream=10 + i), blur)
        out[("image", f)] = torch.from_numpy(img.astype(np.float32)).to(image_dtype).contiguous()
    if disp not in ("noise", "smooth"):
        raise ValueError("disp is 'noise' or 'smooth'")
    for h in range(n_heads):
        if disp == "smooth":
            out[("disp", h)] = smooth_disp(batch, height, width, seed, h)
            continue
        d = box_blur(uniform(seed + 1, (batch, 1, height, width), stream=40 + h), blur)
        d = 0.05 + 0.9 * d
        out[("disp", h)] = torch.from_numpy(d.astype(np.float32)).contiguous()
    out["K"], out["inv_K"] = kitti_intrinsics(batch, height, width)
    sources = list(frame_ids[1:])
    for i, f in enumerate(sources):
        if f == "s":
            T = np.tile(np.eye(4, dtype=np.float32), (batch, 1, 1))
            T[:, 0, 3] = 0.1
            out["stereo_T"] = torch.from_numpy(T)
        else:
            aa = normal(seed + 2, (batch, 3), stream=2 * i) * pose_sigma
            tr = normal(seed + 2, (batch, 3), stream=2 * i + 1) * pose_sigma
            out[("cam_T_cam", f)] = torch.from_numpy(_rodrigues(aa, tr))
    out["noise"] = torch.from_numpy(
        normal(seed + 3, (batch, len(sources), height, width), stream=7).astype(np.float32)).contiguous()
    return out
