"""Counterpart of the loss methods of the reference trainers on the HIP library.

``ViewSynthesisLoss`` exposes the call surface of ``Trainer`` (Python/Trainer.py) that the
hot path consists of, with the same names, dict keys and return values:

    generate_images_pred(inputs, outputs) -> None            Trainer.py:419-479
    compute_reprojection_loss(pred, target) -> [B,1,H,W]      Trainer.py:481-495
    compute_losses(inputs, outputs) -> {"loss/{s}", "loss"}   Trainer.py:518-620
    compute_losses_single(inputs, outputs)                    Trainer_distilation.py:1113-1199
    compute_losses_distillation(inputs, outputs, ...)         Trainer_distilation.py:1208-1330 (0.9 scale)
    compute_losses_per_scale(inputs, outputs)                 the same loop with one disparity per scale

``compute_losses`` runs ONE fused kernel pass from the disparity and the poses (it does not
need ``generate_images_pred`` to have run); ``generate_images_pred`` materialises
("depth",0), ("sample",f), ("color",f) through the per-module kernels for callers that read
them (validation dumps, depth metrics).  ``opt`` is the reference's option namespace; the
fields read are: scales, frame_ids, height, width, min_depth, max_depth, v1_multiscale,
disable_automasking, avg_reprojection, no_ssim, disparity_smoothness, pose_model_type.

    pose_outputs(axisangle, translation) -> {("cam_T_cam",f), ...}   Trainer.py:388-393,408-414
    compute_depth_losses(inputs, outputs, losses) -> (sum, [7 metrics])   Trainer.py:622-662
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import _abi
from .layers import (BackprojectDepth, Project3D, SSIM, affinity_reprojection_loss, depth_metrics, disp_to_depth,
                     get_smooth_loss, grid_sample_border, mean_inverse_depth, poses_from_parameters, warp_features,
                     warp_features_multi)
from .ops import fused_view_synthesis_loss, fused_view_synthesis_loss_heads


class ViewSynthesisLoss:
    def __init__(self, opt, num_scales: int = 1, photometric_scale: float = 1.0, seed: int = 0,
                 use_affinity: bool = False):
        self.opt = opt
        self.use_affinity = use_affinity      # Trainer.py:26
        self.val_mode = False                 # Trainer.py:25: the affinity term is skipped while validating
        self.num_scales = num_scales          # the reference fixes this to 1 (Trainer.py:33)
        self.photometric_scale = photometric_scale
        self.seed = seed
        self._step = 0
        self.backproject_depth = BackprojectDepth(None, opt.height, opt.width)
        self.project_3d = Project3D(None, opt.height, opt.width)
        self.ssim = SSIM()

    # ------------------------------------------------------------------ helpers
    def _sources(self):
        return list(self.opt.frame_ids[1:])

    def _pose(self, inputs, outputs, f):
        return inputs["stereo_T"] if f == "s" else outputs[("cam_T_cam", f)]

    def _poses(self, inputs, outputs, disp):
        """The pose matrix of every source as the warp uses it.  With the "posecnn" pose model the reference
        rebuilds it from the parameters, the translation multiplied by the image's mean inverse depth
        (Trainer.py:446-455): one reduction launch and one pose launch here (the pose launch finishes the mean itself),
        differentiable down to disp."""
        srcs = self._sources()
        if getattr(self.opt, "pose_model_type", "shared") != "posecnn":
            return [self._pose(inputs, outputs, f) for f in srcs]
        if "s" in srcs:
            raise ValueError("pose_model_type='posecnn' has no stereo branch in the reference (Trainer.py:446-455)")
        return poses_from_parameters([outputs[("axisangle", f)][:, 0] for f in srcs],
                                     [outputs[("translation", f)][:, 0] for f in srcs],
                                     [f < 0 for f in srcs], posecnn_disp=disp,
                                     depth_range=(self.opt.min_depth, self.opt.max_depth))

    # ------------------------------------------------------------------ Trainer.py:388-393,408-414
    def pose_outputs(self, axisangle, translation) -> Dict:
        """The dict entries Trainer.predict_pose adds for the temporal sources, all matrices in one launch.
        "all" input mode (Trainer.py:408-414): one pose-network output pair [B,F,1,3], source i uses slice i,
        no inversion.  "pairs" mode (Trainer.py:388-393): dicts frame_id -> [B,1,1,3] network outputs, slice 0,
        inverted for negative frame ids."""
        srcs = [f for f in self._sources() if f != "s"]
        out = {}
        if isinstance(axisangle, dict):
            aa = [axisangle[f][:, 0] for f in srcs]
            tr = [translation[f][:, 0] for f in srcs]
            inv = [f < 0 for f in srcs]
            for f in srcs:
                out[("axisangle", f)], out[("translation", f)] = axisangle[f], translation[f]
        else:
            aa = [axisangle[:, i] for i, _ in enumerate(srcs)]
            tr = [translation[:, i] for i, _ in enumerate(srcs)]
            inv = [False] * len(srcs)
            for f in srcs:
                out[("axisangle", f)], out[("translation", f)] = axisangle, translation
        for f, T in zip(srcs, poses_from_parameters(aa, tr, inv)):
            out[("cam_T_cam", f)] = T
        return out

    def _full_res_disp(self, outputs):
        disp = outputs[("disp", 0)]
        if not getattr(self.opt, "v1_multiscale", False) and tuple(disp.shape[2:]) != (self.opt.height, self.opt.width):
            disp = F.interpolate(disp, [self.opt.height, self.opt.width], mode="bilinear", align_corners=False)
        return disp

    # ------------------------------------------------------------------ Trainer.py:419-479
    def generate_images_pred(self, inputs: Dict, outputs: Dict) -> None:
        disp = self._full_res_disp(outputs)
        _, depth = disp_to_depth(disp, self.opt.min_depth, self.opt.max_depth)
        outputs[("depth", 0)] = depth
        affinity = self.use_affinity and not self.val_mode
        points = None if affinity else self.backproject_depth(depth, inputs["inv_K"])
        for f, T in zip(self._sources(), self._poses(inputs, outputs, disp)):
            if affinity:
                # Trainer.py:467-473: the colour frame and the 25-channel affinity map of the source sampled with one
                # grid -- one launch for both (coordinates from disp and T inside it, gradients to both)
                grid, (color, aff) = warp_features_multi(
                    [inputs[("image", f)].float(), inputs[("affinity", f)][:, 0].float()], disp, inputs["K"], inputs["inv_K"], T,
                    self.opt.min_depth, self.opt.max_depth)
                outputs[("sample", f)], outputs[("color", f)], outputs[("pred_affinity", f)] = grid, color, aff
            else:
                grid = self.project_3d(points, inputs["K"], T)
                outputs[("sample", f)] = grid
                outputs[("color", f)] = grid_sample_border(inputs[("image", f)], grid)
            if not self.opt.disable_automasking:
                outputs[("color_identity", f)] = inputs[("image", f)]

    # ------------------------------------------------------------------ Trainer.py:481-495
    def compute_reprojection_loss(self, pred, target):
        pred = pred.float()
        target = target.float().to(pred.device)
        l1 = torch.abs(target - pred).mean(1, True)
        if self.opt.no_ssim:
            return l1
        return 0.85 * self.ssim(pred, target).mean(1, True) + 0.15 * l1

    # ------------------------------------------------------------------ Trainer.py:497-516
    def compute_affinity_reprojection_loss(self, input_, t_seg, target, weight, rand: Optional[torch.Tensor] = None):
        """Squared error between the warped affinity map and the target frame's, the target multiplied by the
        prediction where ``rand < weight[1]``; ``t_seg`` does not enter the value (Trainer.py:508-514).  One
        forward and one backward kernel (vsl_affinity_loss_fwd / _bwd); ``rand`` is the reference's
        ``torch.rand_like(target)`` (drawn here when not given)."""
        target = target.float().to(input_.device)
        rand = torch.rand_like(target) if rand is None else rand.float().to(input_.device)
        return affinity_reprojection_loss(input_, target, rand, float(weight[1]))

    # ------------------------------------------------------------------ Trainer_distilation.py:1076-1110
    def compute_affinity_huber_loss(self, input_, t_seg, target, weight=None):
        """The distillation trainer's variant of the affinity loss: Huber (delta 1, mean) between the warped map
        and the target frame's, times the fraction of pixels whose target channels sum to less than 24;
        ``t_seg`` and ``weight`` do not enter the value.  Elementwise torch ops on the device."""
        target = target.to(input_.device)
        huber = F.huber_loss(input_, target.to(input_.dtype))
        frac = (target.sum(1, keepdim=True) < 24).to(input_.dtype).mean()
        return huber * frac

    def _affinity_term(self, inputs, outputs, disp, poses, drop_rand):
        """Sum over the sources of the affinity loss (Trainer.py:546-548), each map warped by the fused
        disp -> depth -> backproject -> project -> sample pass so that the gradient reaches disp and the poses."""
        total = 0
        for f, T in zip(self._sources(), poses):
            pred = warp_features(inputs[("affinity", f)][:, 0], disp, inputs["K"], inputs["inv_K"], T,
                                 self.opt.min_depth, self.opt.max_depth)
            outputs[("pred_affinity", f)] = pred
            total = total + self.compute_affinity_reprojection_loss(pred, inputs.get(("segmentation", 0)),
                                                                    inputs[("affinity", 0)][:, 0], [1.5, 0.5], drop_rand)
        return total

    # ------------------------------------------------------------------ Trainer.py:518-620
    def compute_losses(self, inputs: Dict, outputs: Dict, noise: Optional[torch.Tensor] = None,
                       photometric_scale: Optional[float] = None,
                       drop_rand: Optional[torch.Tensor] = None, materialise: bool = False) -> Dict[str, torch.Tensor]:
        """``materialise``: the same launch also fills outputs[("depth",0)], ("sample",f), ("color",f) and
        ("color_identity",f) -- what generate_images_pred leaves for validation dumps and the depth metrics
        (Trainer.py:129-132,627) -- so that no second pass over the per-module kernels is needed (detached)."""
        opt = self.opt
        srcs = self._sources()
        disp = self._full_res_disp(outputs)
        automask = not opt.disable_automasking
        scale_w = [opt.disparity_smoothness / (2 ** s) for s in opt.scales]
        self._step += 1
        poses = self._poses(inputs, outputs, disp)
        res = fused_view_synthesis_loss(
            disp, inputs[("image", 0)], [inputs[("image", f)] for f in srcs], inputs["K"], inputs["inv_K"],
            poses, noise,
            automask=automask, avg_reprojection=opt.avg_reprojection, no_ssim=opt.no_ssim,
            scale_weights=scale_w, inv_num_scales=1.0 / self.num_scales,
            warp_scale=self.photometric_scale if photometric_scale is None else photometric_scale,
            min_depth=opt.min_depth, max_depth=opt.max_depth, seed=self.seed + self._step, materialise=materialise)
        out, ident = res[0], res[1]
        if materialise:
            outputs[("depth", 0)] = res[2]
            for i, f in enumerate(srcs):
                outputs[("sample", f)], outputs[("color", f)] = res[3][i], res[4][i]
                if automask:
                    outputs[("color_identity", f)] = inputs[("image", f)]
        if automask:
            outputs["identity_selection"] = ident
        losses = {"loss/{}".format(s): out[_abi.LOSS_SCALE0 + i] for i, s in enumerate(opt.scales)}
        losses["loss"] = out[_abi.LOSS_TOTAL]
        if self.use_affinity and not self.val_mode:
            # Trainer.py:606: the same affinity term is added to every scale's loss
            aff = self._affinity_term(inputs, outputs, disp, poses, drop_rand)
            losses = {k: v + aff for k, v in losses.items() if k != "loss"}
            losses["loss"] = out[_abi.LOSS_TOTAL] + aff * (len(opt.scales) / self.num_scales)
        return losses

    # ------------------------------------------------------------------ Trainer.py:622-662
    def compute_depth_losses(self, inputs: Dict, outputs: Dict, losses: Optional[Dict] = None):
        """(sum of the seven metrics, [abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3]) as 0-dim device tensors, from
        outputs[("depth",0)] (or, when the fused loss ran without materialising it, from the disparity) and
        inputs["depth_gt"]; the target size is depth_gt's (375 x 1242 in the reference), the crop is the reference's."""
        depth = outputs.get(("depth", 0))
        if depth is None:
            _, depth = disp_to_depth(self._full_res_disp(outputs).detach(), self.opt.min_depth, self.opt.max_depth)
        m = depth_metrics(depth, inputs["depth_gt"])
        return m[_abi.METRIC_SUM], [m[i] for i in range(7)]

    def compute_losses_per_scale(self, inputs: Dict, outputs: Dict, noise: Optional[torch.Tensor] = None,
                                 photometric_scale: Optional[float] = None) -> Dict[str, torch.Tensor]:
        """The loop of Trainer.py:524-619 for a trainer that feeds a DIFFERENT disparity to every scale
        (outputs[("disp", s)], upsampled to full resolution as Trainer.py:431 does) instead of the
        reference's outputs[("disp", 0)] everywhere: all scales run as heads of one launch.  Same keys."""
        opt = self.opt
        srcs = self._sources()
        if getattr(opt, "pose_model_type", "shared") == "posecnn":
            # every head would need its own pose matrices (the translation is scaled by that head's mean inverse
            # depth, Trainer.py:446-455) while the heads of one launch share theirs
            raise ValueError("compute_losses_per_scale does not support pose_model_type='posecnn'; "
                             "call compute_losses once per scale")
        disps = []
        for s in opt.scales:
            d = outputs[("disp", s)]
            if tuple(d.shape[2:]) != (opt.height, opt.width):
                d = F.interpolate(d, [opt.height, opt.width], mode="bilinear", align_corners=False)
            disps.append(d)
        automask = not opt.disable_automasking
        self._step += 1
        out, ident = fused_view_synthesis_loss_heads(
            disps, inputs[("image", 0)], [inputs[("image", f)] for f in srcs], inputs["K"], inputs["inv_K"],
            [self._pose(inputs, outputs, f) for f in srcs], noise,
            automask=automask, avg_reprojection=opt.avg_reprojection, no_ssim=opt.no_ssim,
            scale_weights=[[opt.disparity_smoothness / (2 ** s)] for s in opt.scales],
            warp_scale=self.photometric_scale if photometric_scale is None else photometric_scale,
            min_depth=opt.min_depth, max_depth=opt.max_depth, seed=self.seed + self._step)
        if automask:
            outputs["identity_selection"] = ident[-1]      # the reference keeps the last scale's (Trainer.py:602)
        losses = {"loss/{}".format(s): out[i, _abi.LOSS_TOTAL] for i, s in enumerate(opt.scales)}
        losses["loss"] = out[:, _abi.LOSS_TOTAL].sum() / self.num_scales
        return losses

    # Trainer_distilation.py:1113-1199 is the same computation
    def compute_losses_single(self, inputs, outputs, noise=None):
        return self.compute_losses(inputs, outputs, noise, photometric_scale=1.0)

    # Trainer_distilation.py:1208-1330 with the encoder-feature Huber terms left to the caller
    def compute_losses_distillation(self, inputs, outputs, outputs_teacher=None, features=None, teacher_features=None,
                                    buffered_features=None, noise=None):
        return self.compute_losses(inputs, outputs, noise, photometric_scale=0.9)

    # ------------------------------------------------------------------ un-fused reference order, for cross-checks
    def compute_losses_unfused(self, inputs: Dict, outputs: Dict, noise: Optional[torch.Tensor] = None):
        """Trainer.py:518-620 step by step on the per-module kernels (needs generate_images_pred first)."""
        opt = self.opt
        srcs = self._sources()
        target = inputs[("image", 0)].float()
        disp = outputs[("disp", 0)]
        losses, total = {}, 0
        for scale in opt.scales:
            warped = torch.cat([self.compute_reprojection_loss(outputs[("color", f)], target) * self.photometric_scale
                                for f in srcs], 1)
            if opt.avg_reprojection:
                warped = warped.mean(1, keepdim=True)
            if not opt.disable_automasking:
                ident = torch.cat([self.compute_reprojection_loss(inputs[("image", f)], target) for f in srcs], 1)
                if opt.avg_reprojection:
                    ident = ident.mean(1, keepdim=True)
                z = noise if noise is not None else torch.randn(ident.shape, device=ident.device)
                combined = torch.cat((ident + z.to(ident.device) * 0.00001, warped), dim=1)
            else:
                combined = warped
            if combined.shape[1] == 1:
                chosen = combined
            else:
                chosen, idx = torch.min(combined, dim=1)
            if not opt.disable_automasking:
                outputs["identity_selection"] = (idx > ident.shape[1] - 1).float()
            loss = chosen.mean()
            mean_disp = disp.mean(2, True).mean(3, True)
            loss = loss + opt.disparity_smoothness * get_smooth_loss(disp / (mean_disp + 1e-7), target) / (2 ** scale)
            total = total + loss
            losses["loss/{}".format(scale)] = loss
        losses["loss"] = total / self.num_scales
        return losses
