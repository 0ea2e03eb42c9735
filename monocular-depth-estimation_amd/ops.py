"""Autograd bindings of the fused view-synthesis loss (C-ABI: include/vsloss.h).

``fused_view_synthesis_loss`` evaluates, for one disparity tensor, what
``Trainer.generate_images_pred`` + ``Trainer.compute_losses`` compute
(Python/Trainer.py:419-620 of the reference) and is differentiable with respect
to the disparity and the pose matrices.  The forward launch already prepares
the gradients; ``backward`` is a single elementwise kernel that applies the
upstream gradient.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import torch

from . import _abi


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise ValueError("%s must live on the GPU (the loss has no CPU implementation)" % name)
    if t.dtype != torch.float32:
        t = t.float()          # the reference feeds float64 images (data_loaders.py:338-341); cast on load
    return t.contiguous()


# Descriptor templates per call signature.  Filling a ctypes structure costs a fraction of a microsecond per field and a
# step touches ~80 of them; a training loop calls with the same shapes and options every step, so the constant fields
# are filled once per signature and every call copies the template (one memcpy) and sets only the pointers.  The
# copies are per call: a forward of the same signature between another call's forward and backward cannot disturb it.
_PLANS = {}


class _Plan:
    __slots__ = ("desc", "heads", "nbytes", "stride")


def _loss_plan(B, H, W, S, flags, cfg, scale_weights):
    key = ("single", B, H, W, S, flags, cfg["min_depth"], cfg["max_depth"], cfg["warp_scale"], cfg["inv_num_scales"],
           tuple(scale_weights))
    plan = _PLANS.get(key)
    if plan is None:
        plan = _Plan()
        desc = _abi.LossDesc()
        desc.B, desc.H, desc.W, desc.S = B, H, W, S
        desc.flags = flags
        desc.min_depth, desc.max_depth, desc.eps = cfg["min_depth"], cfg["max_depth"], 1e-7
        desc.ssim_weight, desc.l1_weight, desc.warp_scale = 0.85, 0.15, cfg["warp_scale"]
        desc.n_scales = len(scale_weights)
        for i, w in enumerate(scale_weights):
            desc.scale_weight[i] = w
        desc.inv_num_scales = cfg["inv_num_scales"]
        plan.desc, plan.heads = desc, None
        plan.nbytes = _abi.lib.vsl_loss_workspace_bytes(ctypes.byref(desc))
        plan.stride = (plan.nbytes + 255) // 256 * 256
        if len(_PLANS) > 256:       # signatures that keep changing (a sweep over shapes): start over
            _PLANS.clear()
        _PLANS[key] = plan
    return plan


def _set_common_pointers(desc, cfg, poses, keep):
    S = len(poses)
    desc.target = cfg["target"].data_ptr()
    desc.K, desc.inv_K = cfg["K"].data_ptr(), cfg["inv_K"].data_ptr()
    for s in range(S):
        desc.source[s] = cfg["sources"][s].data_ptr()
        desc.T[s] = poses[s].data_ptr()
    if cfg["noise"] is not None:
        desc.noise = cfg["noise"].data_ptr()
        keep.append(cfg["noise"])
    desc.seed = cfg["seed"]


class _FusedLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, disp, *poses):
        B, _, H, W = disp.shape
        S = len(poses)
        flags = cfg["flags"] | (0 if cfg["want_grad"] else _abi.FLAG_NO_GRAD)
        plan = _loss_plan(B, H, W, S, flags, cfg, cfg["scale_weights"])
        desc = _abi.LossDesc.from_buffer_copy(plan.desc)
        nbytes = plan.nbytes
        keep = [disp, cfg["target"], cfg["K"], cfg["inv_K"]] + list(cfg["sources"]) + list(poses)
        desc.disp = disp.data_ptr()
        _set_common_pointers(desc, cfg, poses, keep)

        ws = torch.empty(nbytes, dtype=torch.uint8, device=disp.device)
        losses = torch.empty(_abi.LOSS_SLOTS, dtype=torch.float32, device=disp.device)
        automask = bool(cfg["flags"] & _abi.FLAG_AUTOMASK)
        ident = torch.empty(B, H, W, dtype=torch.float32, device=disp.device) if automask else None
        stream = torch.cuda.current_stream(disp.device).cuda_stream
        extra = ()
        if cfg.get("materialise"):
            # the tensors Trainer.generate_images_pred keeps (Trainer.py:436,464,467), from the same launch
            head = _abi.Head()
            head.disp, head.workspace, head.workspace_bytes = desc.disp, ws.data_ptr(), nbytes
            head.losses = losses.data_ptr()
            head.identity_selection = ident.data_ptr() if ident is not None else None
            head.n_scales = desc.n_scales
            for i in range(desc.n_scales):
                head.scale_weight[i] = desc.scale_weight[i]
            depth = torch.empty(B, 1, H, W, dtype=torch.float32, device=disp.device)
            sample = torch.empty(S, B, H, W, 2, dtype=torch.float32, device=disp.device)
            color = torch.empty(S, B, 3, H, W, dtype=torch.float32, device=disp.device)
            head.depth = depth.data_ptr()
            for s in range(S):
                head.sample[s], head.color[s] = sample[s].data_ptr(), color[s].data_ptr()
            _abi.check(_abi.forward_heads(desc, ctypes.byref(head), 1, stream))
            extra = (depth, sample, color)
        else:
            _abi.check(_abi.lib.vsl_loss_forward(ctypes.byref(desc), ws.data_ptr(), nbytes, losses.data_ptr(),
                                                 ident.data_ptr() if ident is not None else None, stream))
        ctx.desc, ctx.ws, ctx.nbytes, ctx.keep = desc, ws, nbytes, keep
        ctx.shape = (B, H, W, S)
        if ident is None:
            ident = torch.empty(0, device=disp.device)
        ctx.mark_non_differentiable(ident, *extra)
        return (losses, ident) + extra

    @staticmethod
    def backward(ctx, g_losses, _g_ident, *_g_extra):
        B, H, W, S = ctx.shape
        dev = ctx.ws.device
        need = ctx.needs_input_grad  # (cfg, disp, *poses)
        if g_losses.dtype != torch.float32 or not g_losses.is_contiguous():
            g_losses = g_losses.contiguous().float()
        grad_disp = torch.empty(B, 1, H, W, dtype=torch.float32, device=dev) if need[1] else None
        grad_T = torch.empty(S, B, 4, 4, dtype=torch.float32, device=dev) if any(need[2:2 + S]) else None
        ptrs = (ctypes.c_void_p * _abi.MAX_SOURCES)()
        for s in range(S):
            ptrs[s] = grad_T[s].data_ptr() if need[2 + s] else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        _abi.check(_abi.lib.vsl_loss_backward(ctypes.byref(ctx.desc), ctx.ws.data_ptr(), ctx.nbytes,
                                              g_losses.data_ptr(),
                                              grad_disp.data_ptr() if grad_disp is not None else None,
                                              ptrs, stream))
        return (None, grad_disp) + tuple(grad_T[s] if need[2 + s] else None for s in range(S))


def fused_view_synthesis_loss(disp: torch.Tensor, target: torch.Tensor, sources: Sequence[torch.Tensor],
                              K: torch.Tensor, inv_K: torch.Tensor, poses: Sequence[torch.Tensor],
                              noise: Optional[torch.Tensor] = None, *, automask: bool = True,
                              avg_reprojection: bool = False, no_ssim: bool = False,
                              scale_weights: Sequence[float] = (1e-3,), inv_num_scales: float = 1.0,
                              warp_scale: float = 1.0, min_depth: float = 0.1, max_depth: float = 100.0,
                              seed: int = 0, materialise: bool = False):
    """One fused pass of the view-synthesis loss.

    disp [B,1,H,W]; target and every source [B,3,H,W]; K, inv_K and every pose [B,4,4].
    Returns ``(losses, identity_selection)``: ``losses`` is a float32 vector laid out as
    ``_abi.LOSS_*`` (total, photometric mean, smoothness, then one "loss/{scale}" per entry
    of ``scale_weights``); ``identity_selection`` is [B,H,W] (empty without auto-masking).
    With ``materialise`` the same launch also writes what ``Trainer.generate_images_pred`` keeps
    (Python/Trainer.py:436,464,467) and the call returns ``(losses, identity_selection, depth [B,1,H,W],
    sample [S,B,H,W,2], color [S,B,3,H,W])``, the extra tensors detached.
    """
    if disp.dim() != 4 or disp.shape[1] != 1:
        raise ValueError("disp must be [B,1,H,W]")
    S = len(sources)
    if S < 1 or S > _abi.MAX_SOURCES or len(poses) != S:
        raise ValueError("need 1..%d sources and one pose per source" % _abi.MAX_SOURCES)
    if not 1 <= len(scale_weights) <= _abi.MAX_SCALES:
        raise ValueError("need 1..%d scale weights" % _abi.MAX_SCALES)
    B, _, H, W = disp.shape
    for t in [target] + list(sources):
        if tuple(t.shape) != (B, 3, H, W):
            raise ValueError("images must be [B,3,H,W] matching disp")
    n_id = 1 if avg_reprojection else S
    if noise is not None and automask and tuple(noise.shape) != (B, n_id, H, W):
        raise ValueError("noise must be [B,%d,H,W]" % n_id)
    flags = (_abi.FLAG_AUTOMASK if automask else 0) | (_abi.FLAG_AVG_REPROJECTION if avg_reprojection else 0) \
        | (_abi.FLAG_NO_SSIM if no_ssim else 0)
    poses = [_f32c(p, "pose") for p in poses]
    disp = _f32c(disp, "disp")
    want_grad = torch.is_grad_enabled() and (disp.requires_grad or any(p.requires_grad for p in poses))
    cfg = dict(flags=flags, want_grad=want_grad, min_depth=float(min_depth), max_depth=float(max_depth),
               warp_scale=float(warp_scale), scale_weights=[float(w) for w in scale_weights],
               inv_num_scales=float(inv_num_scales), target=_f32c(target, "target"),
               sources=[_f32c(s, "source") for s in sources], K=_f32c(K, "K"), inv_K=_f32c(inv_K, "inv_K"),
               noise=_f32c(noise, "noise") if (noise is not None and automask) else None, seed=int(seed),
               materialise=bool(materialise))
    out = _FusedLossFn.apply(cfg, disp, *poses)
    return (out[0], (out[1] if automask else None)) + tuple(out[2:])


def _heads_plan(B, H, W, S, n_heads, flags, cfg):
    key = ("heads", B, H, W, S, n_heads, flags, cfg["min_depth"], cfg["max_depth"], cfg["warp_scale"],
           cfg["inv_num_scales"], tuple(cfg["forward_only"]), tuple(tuple(w) for w in cfg["scale_weights"]))
    plan = _PLANS.get(key)
    if plan is None:
        plan = _Plan()
        desc = _abi.LossDesc()
        desc.B, desc.H, desc.W, desc.S = B, H, W, S
        desc.flags = flags
        desc.min_depth, desc.max_depth, desc.eps = cfg["min_depth"], cfg["max_depth"], 1e-7
        desc.ssim_weight, desc.l1_weight, desc.warp_scale = 0.85, 0.15, cfg["warp_scale"]
        desc.n_scales, desc.inv_num_scales = 1, cfg["inv_num_scales"]
        plan.desc = desc
        plan.nbytes = _abi.lib.vsl_loss_workspace_bytes(ctypes.byref(desc))
        plan.stride = (plan.nbytes + 255) // 256 * 256
        heads = (_abi.Head * n_heads)()
        for h in range(n_heads):
            hd = heads[h]
            hd.workspace_bytes = plan.nbytes
            hd.flags = _abi.HEAD_NO_GRAD if cfg["forward_only"][h] else 0
            sw = cfg["scale_weights"][h]
            hd.n_scales = len(sw)
            for i, w in enumerate(sw):
                hd.scale_weight[i] = w
        plan.heads = heads
        if len(_PLANS) > 256:
            _PLANS.clear()
        _PLANS[key] = plan
    return plan


class _FusedHeadsFn(torch.autograd.Function):
    """Several disparity heads over the same images and poses in one launch (vsl_loss_*_heads)."""

    @staticmethod
    def forward(ctx, cfg, n_heads, *tensors):
        disps, poses = tensors[:n_heads], tensors[n_heads:]
        B, _, H, W = disps[0].shape
        S = len(poses)
        flags = cfg["flags"] | (0 if cfg["want_grad"] else _abi.FLAG_NO_GRAD)
        plan = _heads_plan(B, H, W, S, n_heads, flags, cfg)
        desc = _abi.LossDesc.from_buffer_copy(plan.desc)
        heads = type(plan.heads).from_buffer_copy(plan.heads)
        keep = list(disps) + list(poses) + [cfg["target"], cfg["K"], cfg["inv_K"]] + list(cfg["sources"])
        _set_common_pointers(desc, cfg, poses, keep)
        dev = disps[0].device
        automask = bool(cfg["flags"] & _abi.FLAG_AUTOMASK)
        # one allocation for the workspaces of all heads (each a 256-byte-aligned slice), one for the outputs
        ws = torch.empty(n_heads * plan.stride, dtype=torch.uint8, device=dev)
        losses = torch.empty(n_heads, _abi.LOSS_SLOTS, dtype=torch.float32, device=dev)
        ident = torch.empty(n_heads if automask else 0, B, H, W, dtype=torch.float32, device=dev)
        ws_ptr, loss_ptr, ident_ptr = ws.data_ptr(), losses.data_ptr(), ident.data_ptr()
        for h in range(n_heads):
            hd = heads[h]
            hd.disp, hd.workspace = disps[h].data_ptr(), ws_ptr + h * plan.stride
            hd.losses = loss_ptr + 4 * _abi.LOSS_SLOTS * h
            hd.identity_selection = ident_ptr + 4 * B * H * W * h if automask else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        _abi.check(_abi.forward_heads(desc, heads, n_heads, stream))
        ctx.desc, ctx.heads, ctx.ws, ctx.keep = desc, heads, ws, keep
        ctx.shape = (B, H, W, S, n_heads)
        ctx.live = [not f for f in cfg["forward_only"]]
        ctx.mark_non_differentiable(ident)
        return losses, ident

    @staticmethod
    def backward(ctx, g_losses, _g_ident):
        B, H, W, S, n_heads = ctx.shape
        dev = ctx.ws.device
        need = ctx.needs_input_grad  # (cfg, n_heads, disps..., poses...)
        if g_losses.dtype != torch.float32 or not g_losses.is_contiguous():
            g_losses = g_losses.contiguous().float()
        live = ctx.live
        want_d = [bool(need[2 + h]) and live[h] for h in range(n_heads)]
        want_T = [bool(need[2 + n_heads + s]) for s in range(S)]
        idx = [h for h in range(n_heads) if live[h]]
        # one tensor for the disparity gradients of all heads, one for the pose gradients (head-major)
        grad_disp = torch.empty(n_heads, B, 1, H, W, dtype=torch.float32, device=dev) if any(want_d) else None
        grad_T = torch.empty(n_heads, S, B, 4, 4, dtype=torch.float32, device=dev) if (any(want_T) and idx) else None
        gl_ptr = g_losses.data_ptr()
        gd_ptr = grad_disp.data_ptr() if grad_disp is not None else 0
        gt_ptr = grad_T.data_ptr() if grad_T is not None else 0
        for h in range(n_heads):
            hd = ctx.heads[h]
            hd.grad_losses = gl_ptr + 4 * _abi.LOSS_SLOTS * h
            hd.grad_disp = gd_ptr + 4 * B * H * W * h if want_d[h] else None
            for s in range(S):
                hd.grad_T[s] = gt_ptr + 64 * B * (h * S + s) if (grad_T is not None and want_T[s] and live[h]) else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        _abi.check(_abi.backward_heads(ctx.desc, ctx.heads, n_heads, stream))
        # every head with gradients differentiates the same pose matrices: their gradients add up (one reduction
        # over the head axis for all sources)
        gT = [None] * S
        if grad_T is not None:
            if all(want_T):
                tot = grad_T.sum(0) if len(idx) == n_heads else grad_T[idx].sum(0)
                gT = list(tot.unbind(0))
            else:
                gT = [grad_T[idx, s].sum(0) if want_T[s] else None for s in range(S)]
        return (None, None) + tuple(grad_disp[h] if want_d[h] else None for h in range(n_heads)) + tuple(gT)


def fused_view_synthesis_loss_heads(disps: Sequence[torch.Tensor], target: torch.Tensor, sources: Sequence[torch.Tensor],
                                    K: torch.Tensor, inv_K: torch.Tensor, poses: Sequence[torch.Tensor],
                                    noise: Optional[torch.Tensor] = None, *, automask: bool = True,
                                    avg_reprojection: bool = False, no_ssim: bool = False,
                                    scale_weights: Optional[Sequence[Sequence[float]]] = None,
                                    inv_num_scales: float = 1.0, warp_scale: float = 1.0, min_depth: float = 0.1,
                                    max_depth: float = 100.0, seed: int = 0,
                                    forward_only: Optional[Sequence[bool]] = None):
    """``fused_view_synthesis_loss`` for several disparity tensors over the same images and poses in one
    launch.  ``scale_weights[h]`` are head h's smoothness weights (default: [1e-3 / 2**h]).  Returns
    ``(losses [n_heads, LOSS_SLOTS], identity_selection [n_heads, B, H, W] or None)``; the result of every
    head is bit-identical to a separate ``fused_view_synthesis_loss`` call.  ``forward_only[h]`` marks a head
    that gets no gradient (a teacher head next to the student's): its gradient work is skipped."""
    n_heads = len(disps)
    if not 1 <= n_heads <= _abi.MAX_HEADS:
        raise ValueError("need 1..%d heads" % _abi.MAX_HEADS)
    S = len(sources)
    if S < 1 or S > _abi.MAX_SOURCES or len(poses) != S:
        raise ValueError("need 1..%d sources and one pose per source" % _abi.MAX_SOURCES)
    B, _, H, W = disps[0].shape
    for d in disps:
        if d.dim() != 4 or tuple(d.shape) != (B, 1, H, W):
            raise ValueError("every disp must be [B,1,H,W] of one shape")
    for t in [target] + list(sources):
        if tuple(t.shape) != (B, 3, H, W):
            raise ValueError("images must be [B,3,H,W] matching disp")
    if scale_weights is None:
        scale_weights = [[1e-3 / 2 ** h] for h in range(n_heads)]
    if len(scale_weights) != n_heads or any(not 1 <= len(w) <= _abi.MAX_SCALES for w in scale_weights):
        raise ValueError("need 1..%d scale weights for every head" % _abi.MAX_SCALES)
    n_id = 1 if avg_reprojection else S
    if noise is not None and automask and tuple(noise.shape) != (B, n_id, H, W):
        raise ValueError("noise must be [B,%d,H,W]" % n_id)
    forward_only = [False] * n_heads if forward_only is None else [bool(f) for f in forward_only]
    if len(forward_only) != n_heads:
        raise ValueError("forward_only needs one entry per head")
    flags = (_abi.FLAG_AUTOMASK if automask else 0) | (_abi.FLAG_AVG_REPROJECTION if avg_reprojection else 0) \
        | (_abi.FLAG_NO_SSIM if no_ssim else 0)
    poses = [_f32c(p, "pose") for p in poses]
    disps = [_f32c(d.detach() if forward_only[h] else d, "disp") for h, d in enumerate(disps)]
    want_grad = torch.is_grad_enabled() and any(t.requires_grad for t in disps + poses)
    cfg = dict(flags=flags, want_grad=want_grad, min_depth=float(min_depth), max_depth=float(max_depth),
               warp_scale=float(warp_scale), scale_weights=[[float(w) for w in ws] for ws in scale_weights],
               inv_num_scales=float(inv_num_scales), target=_f32c(target, "target"),
               sources=[_f32c(s, "source") for s in sources], K=_f32c(K, "K"), inv_K=_f32c(inv_K, "inv_K"),
               noise=_f32c(noise, "noise") if (noise is not None and automask) else None, seed=int(seed),
               forward_only=forward_only)
    losses, ident = _FusedHeadsFn.apply(cfg, n_heads, *disps, *poses)
    return losses, (ident if automask else None)
