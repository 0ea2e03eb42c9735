"""Synthetic end-to-end training loop in the reference's per-batch order (Python/Trainer.py:213-243, 328-358)
around the HIP view-synthesis loss: encoder -> depth decoder -> pose network -> pose matrices -> fused loss ->
backward -> Adam step, one process per GPU with DistributedDataParallel (RCCL) when started under torchrun.

The three networks below are small stand-ins with the reference's interfaces (a list of feature maps, a dict
with ("disp", scale), (axisangle, translation) of shape [B,F,1,3]); the reference's ResNet encoder and its
decoders are ordinary torch modules and can be passed to TrainStep instead.

    python examples/synthetic_train.py --steps 20
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 examples/synthetic_train.py
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class TinyEncoder(nn.Module):
    def __init__(self, width=16):
        super().__init__()
        self.c1 = nn.Conv2d(3, width, 3, 2, 1)
        self.c2 = nn.Conv2d(width, 2 * width, 3, 2, 1)
        self.c3 = nn.Conv2d(2 * width, 4 * width, 3, 2, 1)

    def forward(self, x):
        f1 = F.elu(self.c1(x))
        f2 = F.elu(self.c2(f1))
        f3 = F.elu(self.c3(f2))
        return [f3, f2, f1]


class TinyDepthDecoder(nn.Module):
    def __init__(self, scales, width=16):
        super().__init__()
        self.scales = list(scales)
        self.up2 = nn.Conv2d(4 * width, 2 * width, 3, 1, 1)
        self.up1 = nn.Conv2d(4 * width, width, 3, 1, 1)
        self.out = nn.Conv2d(2 * width, 1, 3, 1, 1)

    def forward(self, feats):
        f3, f2, f1 = feats
        x = F.elu(self.up2(F.interpolate(f3, scale_factor=2.0, mode="nearest")))
        x = F.elu(self.up1(F.interpolate(torch.cat([x, f2], 1), scale_factor=2.0, mode="nearest")))
        disp = torch.sigmoid(self.out(F.interpolate(torch.cat([x, f1], 1), scale_factor=2.0, mode="nearest")))
        return {("disp", s): disp for s in self.scales}


class TinyPoseDecoder(nn.Module):
    """(axisangle, translation), each [B, F, 1, 3], from the first feature map of every frame."""

    def __init__(self, num_frames, width=16):
        super().__init__()
        self.num_frames = num_frames
        self.conv = nn.Conv2d(4 * width * num_frames, 6 * num_frames, 1)

    def forward(self, input_features):
        x = torch.cat([f[0] for f in input_features], 1)
        out = 0.01 * self.conv(x).mean(3).mean(2).view(-1, self.num_frames, 1, 6)
        return out[..., :3], out[..., 3:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, one GPU per rank) or gloo (rehearsal: ranks may share a GPU)")
    ap.add_argument("--verbose", action="store_true", help="print a line per stage and step (every rank)")
    ap.add_argument("--share-gpu", action="store_true", help="every rank uses GPU 0 (rehearsal of the N > 1 path on one card)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", 0 if args.share_gpu else local)
    torch.cuda.set_device(dev)
    if world > 1:
        if args.backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(args.backend)

    def say(msg):
        if args.verbose:
            print("[rank %d] %s" % (rank, msg), flush=True)

    if args.verbose:      # a rank that makes no progress for 45 s prints the Python stack of every thread
        import faulthandler
        faulthandler.dump_traceback_later(45, repeat=False)
    say("process group ready")
    import mde_amd.synth as synth
    from mde_amd.trainer_step import TrainStep

    opt = types.SimpleNamespace(
        scales=[0, 1, 2, 3], frame_ids=[0, -1, 1], height=args.height, width=args.width, min_depth=0.1, max_depth=100.0,
        v1_multiscale=False, disable_automasking=False, avg_reprojection=False, no_ssim=False,
        disparity_smoothness=1e-3, pose_model_type="shared", pose_model_input="all", learning_rate=1e-4)
    torch.manual_seed(0)
    models = {"encoder": TinyEncoder().to(dev), "depth": TinyDepthDecoder(opt.scales).to(dev),
              "pose": TinyPoseDecoder(len(opt.frame_ids)).to(dev)}
    params = [p for m in models.values() for p in m.parameters()]
    if world > 1:
        models = {k: nn.parallel.DistributedDataParallel(m, device_ids=[dev.index]) for k, m in models.items()}
    say("models wrapped")
    step = TrainStep(opt, models, torch.optim.Adam(params, opt.learning_rate))

    data = synth.make_batch(args.batch, args.height, args.width, opt.frame_ids, seed=100 + rank)
    inputs = {"K": data["K"].to(dev), "inv_K": data["inv_K"].to(dev)}
    for f in opt.frame_ids:
        inputs[("image", f)] = data[("image", f)].to(dev)
        inputs[("image_aug", f)] = inputs[("image", f)]

    first = last = None
    for i in range(args.warmup + args.steps):
        if i == args.warmup:
            if world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        _, losses = step.train_step(inputs)
        say("step %d done" % i)
        if i == 0:
            first = losses["loss"].item()
    last = losses["loss"].item()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"what": "synthetic end-to-end step with stand-in networks", "n_gpus": world, "steps": args.steps,
                          "frames_per_s": world * args.batch * args.steps / t.item(), "ms_per_step": 1e3 * t.item() / args.steps,
                          "first_loss": first, "last_loss": last}))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
