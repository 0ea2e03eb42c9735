"""Drop-in counterparts of the reference's ``Python/layers.py`` primitives on the HIP library.

Same names, argument order and return values as the reference:

    disp_to_depth(disp, min_depth, max_depth) -> (scaled_disp, depth)          layers.py:16
    BackprojectDepth(batch_size, height, width).forward(depth, inv_K)          layers.py:139
    Project3D(batch_size, height, width, eps=1e-7).forward(points, K, T)       layers.py:177
    SSIM().forward(x, y)                                                       layers.py:229
    get_smooth_loss(disp, img)                                                 layers.py:211
    grid_sample_border(image, grid)   == F.grid_sample(..., padding_mode="border")  Trainer.py:467

All of them run hand-written HIP kernels through the C-ABI, take part in autograd, accept
float64 inputs (cast to float32 on load: the reference's images are float64) and, unlike the
reference, take the batch size from the input instead of the constructor.

    transformation_from_parameters(axisangle, translation, invert=False)       layers.py:28
    poses_from_parameters([...], [...], inverts, t_scale)   every source pose of a step in one launch
    mean_inverse_depth(disp, min_depth, max_depth)                             Trainer.py:451-452
    warp_features(feats, disp, K, inv_K, T)          fused coarse-scale feature warp, Train_distillation_new.py:1159-1182
    depth_metrics(depth, depth_gt)                   Trainer.compute_depth_losses + compute_depth_errors, Trainer.py:622-662
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn

from . import _abi

_lib = _abi.lib


def _prep(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise ValueError("%s must be a GPU tensor (there is no CPU implementation)" % name)
    return t.float().contiguous()


def _stream(t: torch.Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


def _ptr(t):
    return t.data_ptr() if t is not None else None


# ------------------------------------------------------------------ disp_to_depth
class _DispToDepth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, min_depth, max_depth):
        scaled, depth = torch.empty_like(disp), torch.empty_like(disp)
        _abi.check(_lib.vsl_disp_to_depth_fwd(disp.data_ptr(), scaled.data_ptr(), depth.data_ptr(), disp.numel(),
                                              min_depth, max_depth, _stream(disp)))
        ctx.save_for_backward(disp)
        ctx.rng = (min_depth, max_depth)
        return scaled, depth

    @staticmethod
    def backward(ctx, g_scaled, g_depth):
        (disp,) = ctx.saved_tensors
        g = torch.empty_like(disp)
        gs = g_scaled.contiguous() if g_scaled is not None else None
        gd = g_depth.contiguous() if g_depth is not None else None
        _abi.check(_lib.vsl_disp_to_depth_bwd(disp.data_ptr(), _ptr(gs), _ptr(gd), g.data_ptr(), disp.numel(),
                                              ctx.rng[0], ctx.rng[1], _stream(disp)))
        return g, None, None


def disp_to_depth(disp, min_depth, max_depth):
    """Sigmoid disparity -> (scaled disparity, depth), layers.py:16-25."""
    return _DispToDepth.apply(_prep(disp, "disp"), float(min_depth), float(max_depth))


# ------------------------------------------------------------------ BackprojectDepth
class _Backproject(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, inv_K):
        B, _, H, W = depth.shape
        pts = torch.empty(B, 4, H * W, dtype=torch.float32, device=depth.device)
        _abi.check(_lib.vsl_backproject_fwd(depth.data_ptr(), inv_K.data_ptr(), pts.data_ptr(), B, H, W, _stream(depth)))
        ctx.save_for_backward(inv_K)
        ctx.shape = (B, H, W)
        return pts

    @staticmethod
    def backward(ctx, g_pts):
        (inv_K,) = ctx.saved_tensors
        B, H, W = ctx.shape
        g_pts = g_pts.contiguous()
        g = torch.empty(B, 1, H, W, dtype=torch.float32, device=g_pts.device)
        _abi.check(_lib.vsl_backproject_bwd(g_pts.data_ptr(), inv_K.data_ptr(), g.data_ptr(), B, H, W, _stream(g_pts)))
        return g, None


class BackprojectDepth(nn.Module):
    """Depth image -> homogeneous camera points [B,4,H*W], layers.py:139-174."""

    def __init__(self, batch_size, height, width):
        super().__init__()
        self.batch_size, self.height, self.width = batch_size, height, width

    def forward(self, depth, inv_K):
        depth = _prep(depth, "depth")
        if depth.dim() != 4 or depth.shape[1] != 1:
            raise ValueError("depth must be [B,1,H,W]")
        return _Backproject.apply(depth, _prep(inv_K, "inv_K"))


# ------------------------------------------------------------------ Project3D
class _Project(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, K, T, H, W, eps):
        B = points.shape[0]
        grid = torch.empty(B, H, W, 2, dtype=torch.float32, device=points.device)
        _abi.check(_lib.vsl_project_fwd(points.data_ptr(), K.data_ptr(), T.data_ptr(), grid.data_ptr(), B, H, W, eps,
                                        _stream(points)))
        ctx.save_for_backward(points, K, T)
        ctx.cfg = (B, H, W, eps)
        return grid

    @staticmethod
    def backward(ctx, g_grid):
        points, K, T = ctx.saved_tensors
        B, H, W, eps = ctx.cfg
        g_grid = g_grid.contiguous()
        need = ctx.needs_input_grad
        g_pts = torch.empty_like(points) if need[0] else None
        g_K = torch.empty_like(K) if need[1] else None
        g_T = torch.empty_like(T) if need[2] else None
        n = _lib.vsl_project_bwd_workspace_bytes(B, H, W)
        ws = torch.empty(n, dtype=torch.uint8, device=points.device)
        _abi.check(_lib.vsl_project_bwd(points.data_ptr(), K.data_ptr(), T.data_ptr(), g_grid.data_ptr(), _ptr(g_pts),
                                        _ptr(g_T), _ptr(g_K), ws.data_ptr(), n, B, H, W, eps, _stream(points)))
        return g_pts, g_K, g_T, None, None, None


class Project3D(nn.Module):
    """Camera points -> normalised sampling grid [B,H,W,2], layers.py:177-202."""

    def __init__(self, batch_size, height, width, eps=1e-7):
        super().__init__()
        self.batch_size, self.height, self.width, self.eps = batch_size, height, width, eps

    def forward(self, points, K, T):
        points = _prep(points, "points")
        if points.dim() != 3 or points.shape[1] != 4 or points.shape[2] != self.height * self.width:
            raise ValueError("points must be [B,4,H*W] for the module's height and width")
        return _Project.apply(points, _prep(K, "K"), _prep(T, "T"), self.height, self.width, float(self.eps))


# ------------------------------------------------------------------ grid_sample (border)
class _GridSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, grid):
        B, C, Hi, Wi = image.shape
        _, Ho, Wo, _ = grid.shape
        out = torch.empty(B, C, Ho, Wo, dtype=torch.float32, device=image.device)
        _abi.check(_lib.vsl_grid_sample_fwd(image.data_ptr(), grid.data_ptr(), out.data_ptr(), B, C, Hi, Wi, Ho, Wo,
                                            _stream(image)))
        ctx.save_for_backward(image, grid)
        return out

    @staticmethod
    def backward(ctx, g_out):
        image, grid = ctx.saved_tensors
        B, C, Hi, Wi = image.shape
        _, Ho, Wo, _ = grid.shape
        g_out = g_out.contiguous()
        g_img = torch.empty_like(image) if ctx.needs_input_grad[0] else None
        g_grid = torch.empty_like(grid) if ctx.needs_input_grad[1] else None
        _abi.check(_lib.vsl_grid_sample_bwd(image.data_ptr(), grid.data_ptr(), g_out.data_ptr(), _ptr(g_grid), _ptr(g_img),
                                            B, C, Hi, Wi, Ho, Wo, _stream(image)))
        return g_img, g_grid


def grid_sample_border(image, grid):
    """``F.grid_sample(image, grid, padding_mode="border")`` (bilinear, align_corners=False), Trainer.py:467."""
    image, grid = _prep(image, "image"), _prep(grid, "grid")
    if image.dim() != 4 or grid.dim() != 4 or grid.shape[3] != 2 or grid.shape[0] != image.shape[0]:
        raise ValueError("image must be [B,C,H,W] and grid [B,Ho,Wo,2]")
    return _GridSample.apply(image, grid)


# ------------------------------------------------------------------ SSIM
class _Ssim(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        B, C, H, W = x.shape
        out = torch.empty_like(x)
        _abi.check(_lib.vsl_ssim_fwd(x.data_ptr(), y.data_ptr(), out.data_ptr(), B * C, H, W, _stream(x)))
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        B, C, H, W = x.shape
        g = g.contiguous()
        gx = gy = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            _abi.check(_lib.vsl_ssim_bwd(x.data_ptr(), y.data_ptr(), g.data_ptr(), gx.data_ptr(), B * C, H, W, _stream(x)))
        if ctx.needs_input_grad[1]:     # the measure is symmetric in its two arguments
            gy = torch.empty_like(y)
            _abi.check(_lib.vsl_ssim_bwd(y.data_ptr(), x.data_ptr(), g.data_ptr(), gy.data_ptr(), B * C, H, W, _stream(x)))
        return gx, gy


class SSIM(nn.Module):
    """Per-pixel ``clamp((1 - SSIM(x, y)) / 2, 0, 1)`` over a reflect-padded 3x3 window, layers.py:229-262."""

    def forward(self, x, y):
        x, y = _prep(x, "x"), _prep(y, "y")
        if x.shape != y.shape or x.dim() != 4:
            raise ValueError("x and y must be [B,C,H,W] of the same shape")
        return _Ssim.apply(x, y)


# ------------------------------------------------------------------ get_smooth_loss
class _Smooth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, img):
        B, _, H, W = disp.shape
        C = img.shape[1]
        loss = torch.empty((), dtype=torch.float32, device=disp.device)
        n = _lib.vsl_smooth_workspace_bytes(B, H, W)
        ws = torch.empty(n, dtype=torch.uint8, device=disp.device)
        _abi.check(_lib.vsl_smooth_fwd(disp.data_ptr(), img.data_ptr(), loss.data_ptr(), ws.data_ptr(), n, B, C, H, W,
                                       _stream(disp)))
        ctx.save_for_backward(disp, img)
        return loss

    @staticmethod
    def backward(ctx, g):
        disp, img = ctx.saved_tensors
        B, _, H, W = disp.shape
        g = g.contiguous().float()
        gd = torch.empty_like(disp)
        _abi.check(_lib.vsl_smooth_bwd(disp.data_ptr(), img.data_ptr(), g.data_ptr(), gd.data_ptr(), B, img.shape[1], H, W,
                                       _stream(disp)))
        return gd, None


def get_smooth_loss(disp, img):
    """Edge-aware first-order smoothness of a disparity image, layers.py:211-226 (0-dim tensor)."""
    disp, img = _prep(disp, "disp"), _prep(img, "img")
    if disp.dim() != 4 or disp.shape[1] != 1 or img.dim() != 4 or img.shape[0] != disp.shape[0] \
            or img.shape[2:] != disp.shape[2:]:
        raise ValueError("disp must be [B,1,H,W] and img [B,C,H,W]")
    return _Smooth.apply(disp, img)


# ------------------------------------------------------------------ pose preparation (layers.py:28-103)
def _rows3(t: torch.Tensor, name: str) -> torch.Tensor:
    """[B,1,3] or [B,3] (possibly a slice of a [B,F,1,3] network output) as a [B,3] view with unit inner stride."""
    if not t.is_cuda:
        raise ValueError("%s must be a GPU tensor (there is no CPU implementation)" % name)
    if t.dim() not in (2, 3) or t.shape[-1] != 3 or (t.dim() == 3 and t.shape[1] != 1):
        raise ValueError("%s must be [B,1,3] or [B,3]" % name)
    r = t.float().reshape(t.shape[0], 3)
    if r.stride(1) != 1 or (r.shape[0] > 1 and r.stride(0) < 3):
        r = r.contiguous()
    return r


class _Poses(torch.autograd.Function):
    """All source poses of a batch in one launch: vsl_pose_fwd / vsl_pose_bwd."""

    @staticmethod
    def forward(ctx, inverts, t_scale, *params):
        n = len(params) // 2
        aa = [_rows3(params[2 * s], "axisangle") for s in range(n)]
        tr = [_rows3(params[2 * s + 1], "translation") for s in range(n)]
        B = aa[0].shape[0]
        if any(x.shape[0] != B for x in aa + tr):
            raise ValueError("all pose parameters must have the same batch size")

        def common_stride(rows):
            st = {3 if B == 1 else r.stride(0) for r in rows}
            if len(st) > 1:
                return [r.contiguous() for r in rows], 3
            return rows, st.pop()

        aa, aa_stride = common_stride(aa)
        tr, tr_stride = common_stride(tr)
        if t_scale is not None:
            t_scale = _prep(t_scale, "t_scale").reshape(-1)
            if t_scale.numel() != B:
                raise ValueError("t_scale must hold one factor per image")
        d = _abi.PoseDesc()
        d.B, d.n_src, d.aa_stride, d.tr_stride = B, n, aa_stride, tr_stride
        d.invert_mask = sum(1 << s for s in range(n) if inverts[s])
        for s in range(n):
            d.axisangle[s], d.translation[s] = aa[s].data_ptr(), tr[s].data_ptr()
        d.t_scale = _ptr(t_scale)
        out = [torch.empty(B, 4, 4, dtype=torch.float32, device=aa[0].device) for _ in range(n)]
        tab = (ctypes.c_void_p * _abi.MAX_SOURCES)(*[o.data_ptr() for o in out])
        _abi.check(_lib.vsl_pose_fwd(ctypes.byref(d), tab, _stream(aa[0])))
        ctx.desc, ctx.keep, ctx.n = d, (aa, tr, t_scale), n
        ctx.shapes = [p.shape for p in params]
        return tuple(out)

    @staticmethod
    def backward(ctx, *g_T):
        d, n = ctx.desc, ctx.n
        aa, tr, t_scale = ctx.keep
        B, dev = d.B, aa[0].device
        g = [(x if x is not None else torch.zeros(B, 4, 4, device=dev)).float().contiguous() for x in g_T]
        g_aa = [torch.empty(B, 3, device=dev) for _ in range(n)]
        g_tr = [torch.empty(B, 3, device=dev) for _ in range(n)]
        g_sc = torch.empty(B, device=dev) if t_scale is not None else None
        tab = lambda ts: (ctypes.c_void_p * _abi.MAX_SOURCES)(*[t.data_ptr() for t in ts])
        _abi.check(_lib.vsl_pose_bwd(ctypes.byref(d), tab(g), tab(g_aa), tab(g_tr), _ptr(g_sc), _stream(aa[0])))
        grads = []
        for s in range(n):
            grads += [g_aa[s].reshape(ctx.shapes[2 * s]), g_tr[s].reshape(ctx.shapes[2 * s + 1])]
        return (None, g_sc) + tuple(grads)


class _PosesPoseCnn(torch.autograd.Function):
    """The "posecnn" pose model's preparation (Trainer.py:446-455): mean inverse depth of the disparity, scaled
    translations and all pose matrices in two launches (vsl_pose_fwd_posecnn); the backward chains vsl_pose_bwd and
    vsl_mean_inv_depth_bwd, so the gradient reaches the pose parameters and the disparity."""

    @staticmethod
    def forward(ctx, inverts, min_depth, max_depth, disp, *params):
        n = len(params) // 2
        aa = [_rows3(params[2 * s], "axisangle") for s in range(n)]
        tr = [_rows3(params[2 * s + 1], "translation") for s in range(n)]
        disp = _prep(disp, "disp")
        if disp.dim() != 4 or disp.shape[1] != 1:
            raise ValueError("disp must be [B,1,H,W]")
        B, _, H, W = disp.shape
        if any(x.shape[0] != B for x in aa + tr):
            raise ValueError("all pose parameters must have the disparity's batch size")

        def common_stride(rows):
            st = {3 if B == 1 else r.stride(0) for r in rows}
            if len(st) > 1:
                return [r.contiguous() for r in rows], 3
            return rows, st.pop()

        aa, aa_stride = common_stride(aa)
        tr, tr_stride = common_stride(tr)
        d = _abi.PoseDesc()
        d.B, d.n_src, d.aa_stride, d.tr_stride = B, n, aa_stride, tr_stride
        d.invert_mask = sum(1 << s for s in range(n) if inverts[s])
        for s in range(n):
            d.axisangle[s], d.translation[s] = aa[s].data_ptr(), tr[s].data_ptr()
        out = [torch.empty(B, 4, 4, dtype=torch.float32, device=disp.device) for _ in range(n)]
        tab = (ctypes.c_void_p * _abi.MAX_SOURCES)(*[o.data_ptr() for o in out])
        scale = torch.empty(B, dtype=torch.float32, device=disp.device)
        nbytes = _lib.vsl_mean_inv_depth_workspace_bytes(B)
        ws = torch.empty(nbytes // 8, dtype=torch.float64, device=disp.device)
        _abi.check(_lib.vsl_pose_fwd_posecnn(ctypes.byref(d), disp.data_ptr(), H, W, float(min_depth), float(max_depth),
                                             scale.data_ptr(), tab, ws.data_ptr(), nbytes, _stream(disp)))
        d.t_scale = scale.data_ptr()       # what vsl_pose_bwd reads
        ctx.desc, ctx.keep, ctx.n = d, (aa, tr, scale, disp), n
        ctx.shapes = [p.shape for p in params]
        ctx.rng = (float(min_depth), float(max_depth))
        return tuple(out)

    @staticmethod
    def backward(ctx, *g_T):
        d, n = ctx.desc, ctx.n
        aa, tr, scale, disp = ctx.keep
        B, dev = d.B, disp.device
        g = [(x if x is not None else torch.zeros(B, 4, 4, device=dev)).float().contiguous() for x in g_T]
        g_aa = [torch.empty(B, 3, device=dev) for _ in range(n)]
        g_tr = [torch.empty(B, 3, device=dev) for _ in range(n)]
        g_sc = torch.empty(B, device=dev)
        tab = lambda ts: (ctypes.c_void_p * _abi.MAX_SOURCES)(*[t.data_ptr() for t in ts])
        _abi.check(_lib.vsl_pose_bwd(ctypes.byref(d), tab(g), tab(g_aa), tab(g_tr), g_sc.data_ptr(), _stream(disp)))
        g_disp = None
        if ctx.needs_input_grad[3]:
            g_disp = torch.empty_like(disp)
            _abi.check(_lib.vsl_mean_inv_depth_bwd(disp.data_ptr(), g_sc.data_ptr(), g_disp.data_ptr(), 0, B, disp.shape[2],
                                                   disp.shape[3], ctx.rng[0], ctx.rng[1], _stream(disp)))
        grads = []
        for s in range(n):
            grads += [g_aa[s].reshape(ctx.shapes[2 * s]), g_tr[s].reshape(ctx.shapes[2 * s + 1])]
        return (None, None, None, g_disp) + tuple(grads)


def poses_from_parameters(axisangles, translations, inverts=None, t_scale=None, posecnn_disp=None, depth_range=(0.1, 100.0)):
    """``[transformation_from_parameters(a, t, inv) for a, t, inv in zip(...)]`` (Trainer.py:393,414) in one
    launch; ``t_scale`` [B] multiplies every translation first (the mean inverse depth of Trainer.py:455).
    ``posecnn_disp`` [B,1,H,W] instead of ``t_scale``: that factor is the mean of 1/depth of this disparity
    (``depth_range`` = (min_depth, max_depth)), finished inside the pose launch itself -- two launches for
    Trainer.py:446-455 where mean_inverse_depth() + poses_from_parameters(t_scale=...) take three; same values."""
    n = len(axisangles)
    if n < 1 or n > _abi.MAX_SOURCES or len(translations) != n:
        raise ValueError("1..4 (axisangle, translation) pairs")
    inverts = tuple(bool(v) for v in (inverts if inverts is not None else [False] * n))
    if len(inverts) != n:
        raise ValueError("one invert flag per pose")
    params = []
    for a, t in zip(axisangles, translations):
        params += [a, t]
    if posecnn_disp is not None:
        if t_scale is not None:
            raise ValueError("give t_scale or posecnn_disp, not both")
        return list(_PosesPoseCnn.apply(inverts, depth_range[0], depth_range[1], posecnn_disp, *params))
    return list(_Poses.apply(inverts, t_scale, *params))


def transformation_from_parameters(axisangle, translation, invert=False):
    """(axisangle [B,1,3], translation [B,1,3]) -> 4x4 pose [B,4,4], layers.py:28-45."""
    return poses_from_parameters([axisangle], [translation], [invert])[0]


class _MeanInvDepth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, min_depth, max_depth):
        disp = _prep(disp, "disp")
        if disp.dim() != 4 or disp.shape[1] != 1:
            raise ValueError("disp must be [B,1,H,W]")
        B, _, H, W = disp.shape
        out = torch.empty(B, dtype=torch.float32, device=disp.device)
        nbytes = _lib.vsl_mean_inv_depth_workspace_bytes(B)
        ws = torch.empty(nbytes // 8, dtype=torch.float64, device=disp.device)
        _abi.check(_lib.vsl_mean_inv_depth_fwd(disp.data_ptr(), out.data_ptr(), ws.data_ptr(), nbytes, B, H, W,
                                               float(min_depth), float(max_depth), _stream(disp)))
        ctx.save_for_backward(disp)
        ctx.rng = (float(min_depth), float(max_depth))
        return out

    @staticmethod
    def backward(ctx, g):
        (disp,) = ctx.saved_tensors
        B, _, H, W = disp.shape
        g_disp = torch.empty_like(disp)
        _abi.check(_lib.vsl_mean_inv_depth_bwd(disp.data_ptr(), g.float().contiguous().data_ptr(), g_disp.data_ptr(), 0,
                                               B, H, W, ctx.rng[0], ctx.rng[1], _stream(disp)))
        return g_disp, None, None


def mean_inverse_depth(disp, min_depth, max_depth):
    """Per-image mean of 1/depth, ``(1 / depth).mean(3, True).mean(2, True)`` of Trainer.py:451-452, as [B]."""
    return _MeanInvDepth.apply(disp, min_depth, max_depth)


# ------------------------------------------------------------------ feature warping (Train_distillation_new.py:1159-1182)
class _WarpFeatures(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, disp, K, inv_K, T, min_depth, max_depth, eps):
        feats, disp = _prep(feats, "feats"), _prep(disp, "disp")
        K, inv_K, T = _prep(K, "K"), _prep(inv_K, "inv_K"), _prep(T, "T")
        if feats.dim() != 4 or disp.dim() != 4 or disp.shape[1] != 1 or disp.shape[0] != feats.shape[0] \
                or disp.shape[2:] != feats.shape[2:]:
            raise ValueError("feats must be [B,C,H,W] and disp [B,1,H,W] of the same size")
        B, C, H, W = feats.shape
        if any(tuple(m.shape) != (B, 4, 4) for m in (K, inv_K, T)):
            raise ValueError("K, inv_K and T must be [B,4,4]")
        d = _abi.WarpDesc()
        d.B, d.C, d.H, d.W = B, C, H, W
        d.min_depth, d.max_depth, d.eps = float(min_depth), float(max_depth), float(eps)
        d.feats, d.disp, d.K, d.inv_K, d.T = (t.data_ptr() for t in (feats, disp, K, inv_K, T))
        out = torch.empty_like(feats)
        sample = torch.empty(B, H, W, 2, dtype=torch.float32, device=feats.device)
        _abi.check(_lib.vsl_warp_features_fwd(ctypes.byref(d), out.data_ptr(), sample.data_ptr(), _stream(feats)))
        ctx.save_for_backward(feats, disp, K, inv_K, T)
        ctx.desc = d
        ctx.mark_non_differentiable(sample)
        return out, sample

    @staticmethod
    def backward(ctx, g_out, _g_sample):
        feats, disp, K, inv_K, T = ctx.saved_tensors
        d = ctx.desc
        g_out = g_out.float().contiguous()
        need = ctx.needs_input_grad
        g_feats = torch.empty_like(feats) if need[0] else None
        g_disp = torch.empty_like(disp) if need[1] else None
        g_T = torch.empty_like(T) if need[4] else None
        nbytes = _lib.vsl_warp_features_workspace_bytes(ctypes.byref(d))
        ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=feats.device)
        _abi.check(_lib.vsl_warp_features_bwd(ctypes.byref(d), g_out.data_ptr(), _ptr(g_feats), _ptr(g_disp), _ptr(g_T),
                                              ws.data_ptr(), ws.numel() * 8, _stream(feats)))
        return g_feats, g_disp, None, None, g_T, None, None, None


def warp_features(feats, disp, K, inv_K, T, min_depth=0.1, max_depth=100.0, eps=1e-7, return_sample=False):
    """``F.grid_sample(feats, Project3D(BackprojectDepth(disp_to_depth(disp)[1], inv_K), K, T), padding_mode="border")``
    in one pass, as Train_distillation_new.py:1159-1182 builds ``prev_maps[("feats", k)]`` at 24x80, 12x40 and 6x20;
    ``disp`` is the disparity already interpolated to the size of ``feats``.  Gradients reach ``feats`` (atomic
    scatter), ``disp`` and ``T``."""
    out, sample = _WarpFeatures.apply(feats, disp, K, inv_K, T, min_depth, max_depth, eps)
    return (out, sample) if return_sample else out


class _WarpMulti(torch.autograd.Function):
    """Several feature maps through one set of sampling coordinates (vsl_warp_multi_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, n, min_depth, max_depth, eps, disp, K, inv_K, T, *feats):
        feats = [_prep(f, "feats") for f in feats]
        disp, K, inv_K, T = _prep(disp, "disp"), _prep(K, "K"), _prep(inv_K, "inv_K"), _prep(T, "T")
        B, _, H, W = feats[0].shape
        if any(f.dim() != 4 or f.shape[0] != B or tuple(f.shape[2:]) != (H, W) for f in feats) \
                or tuple(disp.shape) != (B, 1, H, W):
            raise ValueError("every feature map must be [B,C,H,W] and disp [B,1,H,W] of one size")
        if any(tuple(m.shape) != (B, 4, 4) for m in (K, inv_K, T)):
            raise ValueError("K, inv_K and T must be [B,4,4]")
        d = _abi.WarpMultiDesc()
        d.B, d.H, d.W, d.n = B, H, W, n
        d.min_depth, d.max_depth, d.eps = float(min_depth), float(max_depth), float(eps)
        outs = [torch.empty_like(f) for f in feats]
        tab = (ctypes.c_void_p * _abi.MAX_WARP_TENSORS)()
        for i, f in enumerate(feats):
            d.C[i], d.feats[i], tab[i] = f.shape[1], f.data_ptr(), outs[i].data_ptr()
        d.disp, d.K, d.inv_K, d.T = disp.data_ptr(), K.data_ptr(), inv_K.data_ptr(), T.data_ptr()
        sample = torch.empty(B, H, W, 2, dtype=torch.float32, device=disp.device)
        _abi.check(_lib.vsl_warp_multi_fwd(ctypes.byref(d), tab, sample.data_ptr(), _stream(disp)))
        ctx.save_for_backward(disp, K, inv_K, T, *feats)
        ctx.desc = d
        ctx.mark_non_differentiable(sample)
        # an output nobody differentiates reaches backward as None, not as a tensor of zeros: vsl_warp_multi_bwd then
        # skips that tensor's gather and scatter altogether (a NULL entry of its g_out table)
        ctx.set_materialize_grads(False)
        return (sample,) + tuple(outs)

    @staticmethod
    def backward(ctx, _g_sample, *g_outs):
        disp, K, inv_K, T = ctx.saved_tensors[:4]
        feats = ctx.saved_tensors[4:]
        d = ctx.desc
        need = ctx.needs_input_grad          # (n, min, max, eps, disp, K, inv_K, T, *feats)
        g_outs = [g.float().contiguous() if g is not None else None for g in g_outs]
        g_feats = [torch.empty_like(f) if (need[8 + i] and g_outs[i] is not None) else None for i, f in enumerate(feats)]
        if all(g is None for g in g_outs):
            return (None,) * (8 + len(feats))
        go, gf = (ctypes.c_void_p * _abi.MAX_WARP_TENSORS)(), (ctypes.c_void_p * _abi.MAX_WARP_TENSORS)()
        for i in range(len(feats)):
            go[i], gf[i] = _ptr(g_outs[i]), _ptr(g_feats[i])
        g_disp = torch.empty_like(disp) if need[4] else None
        g_T = torch.empty_like(T) if need[7] else None
        nbytes = _lib.vsl_warp_multi_workspace_bytes(ctypes.byref(d))
        ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=disp.device)
        _abi.check(_lib.vsl_warp_multi_bwd(ctypes.byref(d), go, gf, _ptr(g_disp), _ptr(g_T), ws.data_ptr(), ws.numel() * 8,
                                           _stream(disp)))
        return (None, None, None, None, g_disp, None, None, g_T) + tuple(g_feats)


def warp_features_multi(feats, disp, K, inv_K, T, min_depth=0.1, max_depth=100.0, eps=1e-7):
    """``warp_features`` for 1..4 feature maps that share one sampling grid -- the colour frame and the 25-channel
    affinity map of a source frame (Python/Trainer.py:467-473) -- in ONE launch.  Returns ``(sample, [warped...])``."""
    feats = list(feats)
    if not 1 <= len(feats) <= _abi.MAX_WARP_TENSORS:
        raise ValueError("need 1..%d feature maps" % _abi.MAX_WARP_TENSORS)
    res = _WarpMulti.apply(len(feats), min_depth, max_depth, eps, disp, K, inv_K, T, *feats)
    return res[0], list(res[1:])


class _AffinityLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, rand, keep_below):
        pred, target, rand = _prep(pred, "pred"), _prep(target, "target"), _prep(rand, "rand")
        if target.shape != pred.shape or rand.shape != pred.shape:
            raise ValueError("pred, target and rand must have one shape")
        n = pred.numel()
        nbytes = _lib.vsl_affinity_loss_workspace_bytes(n)
        ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=pred.device)
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        _abi.check(_lib.vsl_affinity_loss_fwd(pred.data_ptr(), target.data_ptr(), rand.data_ptr(), float(keep_below), n,
                                              loss.data_ptr(), ws.data_ptr(), ws.numel() * 8, _stream(pred)))
        ctx.save_for_backward(pred, target, rand)
        ctx.keep_below = float(keep_below)
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        pred, target, rand = ctx.saved_tensors
        g_pred = torch.empty_like(pred)
        g = g_loss.float().contiguous().reshape(1)
        _abi.check(_lib.vsl_affinity_loss_bwd(pred.data_ptr(), target.data_ptr(), rand.data_ptr(), ctx.keep_below, pred.numel(),
                                              g.data_ptr(), g_pred.data_ptr(), _stream(pred)))
        return g_pred, None, None, None


def affinity_reprojection_loss(pred, target, rand, keep_below=0.5):
    """``Trainer.compute_affinity_reprojection_loss`` (Python/Trainer.py:497-516) as one forward and one backward
    kernel: mean((pred - target * drop)**2), drop = pred where ``rand < keep_below`` else 1; ``rand`` is the
    reference's ``torch.rand_like(target)``.  Differentiable with respect to ``pred``."""
    return _AffinityLoss.apply(pred, target, rand, keep_below)


# ------------------------------------------------------------------ depth metrics (Trainer.py:622-662, layers.py:265-283)
def depth_metrics(depth, depth_gt, crop=(153, 371, 44, 1197), clamp=(1e-3, 80.0)):
    """The masked, median-scaled depth metrics of ``Trainer.compute_depth_losses`` in one stream-ordered call
    (no gather, no sort, no host synchronisation).  Returns a float32 tensor of ``_abi.METRIC_SLOTS`` values:
    abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3, their sum, the masked count and the two medians.  Carries no
    gradient: the reference detaches the prediction (Trainer.py:630)."""
    depth, depth_gt = _prep(depth.detach(), "depth"), _prep(depth_gt, "depth_gt")
    if depth.dim() != 4 or depth.shape[1] != 1 or depth_gt.dim() != 4 or depth_gt.shape[1] != 1 \
            or depth_gt.shape[0] != depth.shape[0]:
        raise ValueError("depth must be [B,1,H,W] and depth_gt [B,1,Hg,Wg]")
    d = _abi.DepthMetricsDesc()
    d.B, d.H, d.W = depth.shape[0], depth.shape[2], depth.shape[3]
    d.Hg, d.Wg = depth_gt.shape[2], depth_gt.shape[3]
    d.crop_y0, d.crop_y1, d.crop_x0, d.crop_x1 = (int(v) for v in crop)
    d.clamp_min, d.clamp_max = float(clamp[0]), float(clamp[1])
    d.depth, d.depth_gt = depth.data_ptr(), depth_gt.data_ptr()
    nbytes = _lib.vsl_depth_metrics_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=depth.device)
    out = torch.empty(_abi.METRIC_SLOTS, dtype=torch.float32, device=depth.device)
    _abi.check(_lib.vsl_depth_metrics(ctypes.byref(d), out.data_ptr(), ws.data_ptr(), nbytes, _stream(depth)))
    return out


# ------------------------------------------------------------------ the reference's two helpers, for completeness
def rot_from_axisangle(vec):
    """Axis-angle [B,1,3] -> rotation [B,4,4], layers.py:64-103 (pose of zero translation)."""
    return transformation_from_parameters(vec, torch.zeros_like(vec))


def get_translation_matrix(translation_vector):
    """Translation [B,1,3] -> [B,4,4], layers.py:48-61 (pose of zero rotation)."""
    return transformation_from_parameters(torch.zeros_like(translation_vector), translation_vector)
