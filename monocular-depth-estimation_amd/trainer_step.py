"""Counterpart of the reference trainer's per-batch path around the loss, in the reference's order:

    process_batch(inputs) -> (outputs, losses)         Python/Trainer.py:328-358
    predict_pose(inputs, features) -> outputs          Python/Trainer.py:360-416
    train_step(inputs) -> (outputs, losses)            the loop body of run_epoch, Python/Trainer.py:217-243

The networks are the caller's torch modules (``models["encoder"]``, ``models["depth"]``, ``models["pose"]`` and,
for ``pose_model_type == "separate_resnet"``, ``models["pose_encoder"]``) with the reference's interfaces:
the encoder maps a batch of images to a list of feature maps, the depth decoder maps that list to a dict with
``("disp", scale)`` entries, the pose network returns ``(axisangle, translation)`` of shape [B,F,1,3].  Everything
between the network outputs and the scalar loss runs on the HIP library: every pose matrix of the step in one
launch, then the fused view-synthesis loss; nothing in between is materialised unless ``materialise`` is set
(the reference always calls ``generate_images_pred``; its outputs are only read by validation dumps and the
depth metrics).  For data-parallel training wrap the modules in ``DistributedDataParallel`` before passing
them in: the loss needs no collective of its own.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from .trainer_loss import ViewSynthesisLoss


class TrainStep:
    def __init__(self, opt, models: Dict[str, torch.nn.Module], optimizer: Optional[torch.optim.Optimizer] = None,
                 num_scales: int = 1, materialise: bool = False, use_pose_net: bool = True):
        self.opt = opt
        self.models = models
        self.optimizer = optimizer
        self.materialise = materialise
        self.use_pose_net = use_pose_net                      # Trainer.py:24
        self.num_input_frames = len(opt.frame_ids)            # Trainer.py:31
        self.num_pose_frames = 2 if getattr(opt, "pose_model_input", "all") == "pairs" else self.num_input_frames
        self.loss = ViewSynthesisLoss(opt, num_scales=num_scales)

    # ------------------------------------------------------------------ Trainer.py:328-358
    def process_batch(self, inputs: Dict, noise: Optional[torch.Tensor] = None) -> Tuple[Dict, Dict]:
        opt = self.opt
        if getattr(opt, "pose_model_type", "shared") != "shared":
            raise NotImplementedError("separate encoders not implemented (Trainer.py:347-348)")
        frames = list(opt.frame_ids)
        batch = inputs[("image_aug", 0)].shape[0]
        all_colour_aug = torch.cat([inputs[("image_aug", i)] for i in frames])
        all_features = self.models["encoder"](all_colour_aug.float())
        all_features = [torch.split(f, batch) for f in all_features]
        features = {k: [f[i] for f in all_features] for i, k in enumerate(frames)}
        outputs = self.models["depth"](features[0])
        if self.use_pose_net:
            outputs.update(self.predict_pose(inputs, features))
        if self.materialise:
            self.loss.generate_images_pred(inputs, outputs)
        losses = self.loss.compute_losses(inputs, outputs, noise)
        return outputs, losses

    # ------------------------------------------------------------------ Trainer.py:360-416
    def predict_pose(self, inputs: Dict, features: Dict) -> Dict:
        opt = self.opt
        kind = getattr(opt, "pose_model_type", "shared")
        temporal = [f for f in opt.frame_ids[1:] if f != "s"]
        if self.num_pose_frames == 2:
            # one pass of the pose network per source frame, frames in temporal order (Trainer.py:364-393)
            feats = ({f: features[f] for f in opt.frame_ids} if kind == "shared"
                     else {f: inputs[("image_aug", f)] for f in opt.frame_ids})
            aa, tr = {}, {}
            for f in temporal:
                pose_inputs = [feats[f], feats[0]] if f < 0 else [feats[0], feats[f]]
                if kind == "separate_resnet":
                    pose_inputs = [self.models["pose_encoder"](torch.cat(pose_inputs, 1))]
                elif kind == "posecnn":
                    pose_inputs = torch.cat(pose_inputs, 1)
                aa[f], tr[f] = self.models["pose"](pose_inputs)
            return self.loss.pose_outputs(aa, tr)
        # all frames into the pose network at once (Trainer.py:395-414)
        if kind in ("separate_resnet", "posecnn"):
            pose_inputs = torch.cat([inputs[("image_aug", i)] for i in opt.frame_ids if i != "s"], 1)
            if kind == "separate_resnet":
                pose_inputs = [self.models["pose_encoder"](pose_inputs)]
        else:
            pose_inputs = [features[i] for i in opt.frame_ids if i != "s"]
        axisangle, translation = self.models["pose"](pose_inputs)
        return self.loss.pose_outputs(axisangle, translation)

    # ------------------------------------------------------------------ Trainer.py:217-243
    def train_step(self, inputs: Dict, noise: Optional[torch.Tensor] = None) -> Tuple[Dict, Dict]:
        if self.optimizer is None:
            raise ValueError("train_step needs an optimizer")
        outputs, losses = self.process_batch(inputs, noise)
        self.optimizer.zero_grad()
        losses["loss"].backward()
        self.optimizer.step()
        return outputs, losses
