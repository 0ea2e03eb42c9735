"""Counterpart of the reference trainer's per-batch path around the loss, in the reference's order:

    process_batch(inputs) -> (outputs, losses)         Python/Trainer.py:328-358
    predict_pose(inputs, features) -> outputs          Python/Trainer.py:360-416
    train_step(inputs) -> (outputs, losses)            the loop body of run_epoch, Python/Trainer.py:217-243

The networks are the caller's torch modules (``models["encoder"]``, ``models["depth"]``, ``models["pose"]`` and,
for ``pose_model_type == "separate_resnet"``, ``models["pose_encoder"]``) with the reference's interfaces:
the encoder maps a batch of images to a list of feature maps, the depth decoder maps that list to a dict with
``("disp", scale)`` entries, the pose network returns ``(axisangle, translation)`` of shape [B,F,1,3].
``NetworkPass`` runs all of them for one batch as ONE module, so that data-parallel training wraps a single
``DistributedDataParallel`` (one gradient reducer, one set of buckets) around it:

    net = DistributedDataParallel(NetworkPass(opt, models), device_ids=[dev])
    step = TrainStep(opt, net, optimizer)

Everything between the network outputs and the scalar loss runs on the HIP library: every pose matrix of the
step in one launch, then the fused view-synthesis loss; nothing in between is materialised unless
``materialise`` is set (the reference always calls ``generate_images_pred``; its outputs are only read by
validation dumps and the depth metrics).  The loss needs no collective of its own.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple, Union

import torch

from .trainer_loss import ViewSynthesisLoss


class NetworkPass(torch.nn.Module):
    """The network part of Trainer.process_batch + Trainer.predict_pose (Python/Trainer.py:330-351,360-416
    without the pose-matrix maths): ``forward({frame_id: image_aug})`` returns the depth decoder's dict plus
    ``"axisangle"`` / ``"translation"`` -- tensors [B,F,1,3] in "all" input mode, dicts frame_id -> [B,1,1,3]
    in "pairs" mode."""

    def __init__(self, opt, models: Dict[str, torch.nn.Module], use_pose_net: bool = True):
        super().__init__()
        self.opt = opt
        self.models = torch.nn.ModuleDict(models)
        self.use_pose_net = use_pose_net                      # Trainer.py:24
        self.num_pose_frames = 2 if getattr(opt, "pose_model_input", "all") == "pairs" else len(opt.frame_ids)

    def forward(self, image_aug: Dict) -> Dict:
        opt = self.opt
        frames = list(opt.frame_ids)
        if getattr(opt, "pose_model_type", "shared") == "shared":
            batch = image_aug[0].shape[0]
            all_features = self.models["encoder"](torch.cat([image_aug[i] for i in frames]).float())
            all_features = [torch.split(f, batch) for f in all_features]
            features = {k: [f[i] for f in all_features] for i, k in enumerate(frames)}
        else:
            # The reference prints "separate encoders not implemented" here and then fails on the undefined
            # `outputs` (Trainer.py:347-351).  Its pose branches for these types exist (Trainer.py:373-384,
            # 397-405) and need only the target frame's depth: encode frame 0 alone.
            features = {0: self.models["encoder"](image_aug[0].float())}
        out = dict(self.models["depth"](features[0]))
        if self.use_pose_net:
            out["axisangle"], out["translation"] = self.pose_parameters(image_aug, features)
        return out

    # ------------------------------------------------------------------ Trainer.py:360-416
    def pose_parameters(self, image_aug: Dict, features: Dict):
        opt = self.opt
        kind = getattr(opt, "pose_model_type", "shared")
        temporal = [f for f in opt.frame_ids[1:] if f != "s"]
        if self.num_pose_frames == 2:
            # one pass of the pose network per source frame, frames in temporal order (Trainer.py:364-393)
            feats = features if kind == "shared" else image_aug
            aa, tr = {}, {}
            for f in temporal:
                pose_inputs = [feats[f], feats[0]] if f < 0 else [feats[0], feats[f]]
                if kind == "separate_resnet":
                    pose_inputs = [self.models["pose_encoder"](torch.cat(pose_inputs, 1).float())]
                elif kind == "posecnn":
                    pose_inputs = torch.cat(pose_inputs, 1).float()
                aa[f], tr[f] = self.models["pose"](pose_inputs)
            return aa, tr
        # all frames into the pose network at once (Trainer.py:395-414)
        if kind in ("separate_resnet", "posecnn"):
            pose_inputs = torch.cat([image_aug[i] for i in opt.frame_ids if i != "s"], 1).float()
            if kind == "separate_resnet":
                pose_inputs = [self.models["pose_encoder"](pose_inputs)]
        else:
            pose_inputs = [features[i] for i in opt.frame_ids if i != "s"]
        return self.models["pose"](pose_inputs)


def _to_float32(v):
    if isinstance(v, dict):
        return {k: _to_float32(x) for k, x in v.items()}
    return v.float() if torch.is_tensor(v) and v.is_floating_point() else v


class TrainStep:
    def __init__(self, opt, models: Union[Dict[str, torch.nn.Module], torch.nn.Module],
                 optimizer: Optional[torch.optim.Optimizer] = None, num_scales: int = 1, materialise: bool = False,
                 use_pose_net: Optional[bool] = None, per_scale: bool = False, autocast_dtype: Optional[torch.dtype] = None):
        """``models``: the dict of networks, or a module with NetworkPass's forward (e.g. a NetworkPass wrapped in
        DistributedDataParallel).  ``per_scale``: feed ("disp", s) of every scale to the loss
        (compute_losses_per_scale) instead of the reference's ("disp", 0) everywhere.  ``autocast_dtype``: run the
        networks under torch.autocast with that type (e.g. torch.bfloat16); their outputs are cast back to float32, the
        loss always computes in float32.  ``use_pose_net`` (default True) belongs to the NetworkPass this constructor
        builds from a dict; a ready module carries its own setting, and passing a different one here is an error."""
        self.opt = opt
        if isinstance(models, dict):
            self.net = NetworkPass(opt, models, True if use_pose_net is None else use_pose_net)
        else:
            self.net = models
            inner = self.network_pass
            if use_pose_net is not None and hasattr(inner, "use_pose_net") and bool(inner.use_pose_net) != bool(use_pose_net):
                raise ValueError("use_pose_net=%s contradicts the module's own setting (%s): set it on the NetworkPass"
                                 % (use_pose_net, inner.use_pose_net))
        self.optimizer = optimizer
        self.materialise = materialise
        self.per_scale = per_scale
        self.autocast_dtype = autocast_dtype
        self.loss = ViewSynthesisLoss(opt, num_scales=num_scales)

    # what the reference trainer keeps as attributes of its own (Trainer.py:24,31-37), read through to the network module
    @property
    def network_pass(self):
        """The NetworkPass behind ``net`` (unwrapped from DistributedDataParallel's ``.module``)."""
        return getattr(self.net, "module", self.net)

    @property
    def models(self):
        return self.network_pass.models

    @property
    def use_pose_net(self):
        return self.network_pass.use_pose_net

    @property
    def num_pose_frames(self):
        return self.network_pass.num_pose_frames

    def predict_pose(self, inputs: Dict, features: Dict) -> Dict:
        """Trainer.predict_pose (Python/Trainer.py:360-416): pose network, then every pose matrix in one launch."""
        images = {f: inputs[("image_aug", f)] for f in self.opt.frame_ids}
        aa, tr = self.network_pass.pose_parameters(images, features)
        return self.loss.pose_outputs(aa, tr)

    # ------------------------------------------------------------------ Trainer.py:328-358
    def process_batch(self, inputs: Dict, noise: Optional[torch.Tensor] = None) -> Tuple[Dict, Dict]:
        images = {f: inputs[("image_aug", f)] for f in self.opt.frame_ids}
        if self.autocast_dtype is None:
            outputs = dict(self.net(images))
        else:
            with torch.autocast(images[self.opt.frame_ids[0]].device.type, dtype=self.autocast_dtype):
                outputs = dict(self.net(images))
            outputs = {k: _to_float32(v) for k, v in outputs.items()}
        if "axisangle" in outputs:
            outputs.update(self.loss.pose_outputs(outputs.pop("axisangle"), outputs.pop("translation")))
        if self.per_scale:
            if self.materialise:
                self.loss.generate_images_pred(inputs, outputs)   # per-module kernels: the heads call has one pose set, many disparities
            losses = self.loss.compute_losses_per_scale(inputs, outputs, noise)
        else:
            # ("depth",0), ("sample",f), ("color",f) come out of the loss launch itself when asked for
            losses = self.loss.compute_losses(inputs, outputs, noise, materialise=self.materialise)
        return outputs, losses

    # ------------------------------------------------------------------ Trainer.py:217-243
    def train_step(self, inputs: Dict, noise: Optional[torch.Tensor] = None) -> Tuple[Dict, Dict]:
        if self.optimizer is None:
            raise ValueError("train_step needs an optimizer")
        outputs, losses = self.process_batch(inputs, noise)
        self.optimizer.zero_grad()
        losses["loss"].backward()
        self.optimizer.step()
        return outputs, losses
