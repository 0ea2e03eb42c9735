"""TrainStep (process_batch / predict_pose / train_step in the reference's order, Python/Trainer.py:217-243,
328-416) on the GPU with small stand-in networks: the loss equals the oracle's on the same network outputs,
gradients reach every trainable parameter, and a few optimizer steps reduce the loss."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import vsl_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(pose_model_input="all", B=2, H=32, W=96):
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import synthetic_train as ex
    import mde_amd.synth as synth
    from mde_amd.trainer_step import TrainStep
    opt = types.SimpleNamespace(
        scales=[0, 1], frame_ids=[0, -1, 1], height=H, width=W, min_depth=0.1, max_depth=100.0, v1_multiscale=False,
        disable_automasking=False, avg_reprojection=False, no_ssim=False, disparity_smoothness=1e-3,
        pose_model_type="shared", pose_model_input=pose_model_input)
    torch.manual_seed(3)
    n_pose = 2 if pose_model_input == "pairs" else len(opt.frame_ids)
    models = {"encoder": ex.TinyEncoder().to(DEV), "depth": ex.TinyDepthDecoder(opt.scales).to(DEV),
              "pose": ex.TinyPoseDecoder(n_pose).to(DEV)}
    if pose_model_input == "pairs":        # one (axisangle, translation) per call, as networks.PoseDecoder(2, 1) gives
        inner = models["pose"]
        models["pose"] = _FirstFrame(inner).to(DEV)
    params = [p for m in models.values() for p in m.parameters()]
    step = TrainStep(opt, models, torch.optim.Adam(params, 1e-3))
    data = synth.make_batch(B, H, W, opt.frame_ids, seed=21)
    inputs = {"K": data["K"].to(DEV), "inv_K": data["inv_K"].to(DEV)}
    for f in opt.frame_ids:
        inputs[("image", f)] = data[("image", f)].to(DEV)
        inputs[("image_aug", f)] = inputs[("image", f)]
    return opt, models, params, step, data, inputs


class _FirstFrame(torch.nn.Module):
    def __init__(self, inner):
        super().__init__()
        self.inner = inner

    def forward(self, feats):
        a, t = self.inner(feats)
        return a[:, :1], t[:, :1]


@pytest.mark.parametrize("mode", ["all", "pairs"])
def test_process_batch_matches_oracle_on_the_same_network_outputs(mode):
    opt, models, params, step, data, inputs = _setup(mode)
    noise = data["noise"].to(DEV)
    outputs, losses = step.process_batch(inputs, noise)
    losses["loss"].backward()
    assert set(losses) == {"loss", "loss/0", "loss/1"}
    for f in (-1, 1):
        assert outputs[("cam_T_cam", f)].shape == (2, 4, 4)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params)
    assert sum(float(p.grad.abs().sum()) for p in models["pose"].parameters()) > 0
    # the oracle on the network outputs (float32 disparity, parameters), gradients with respect to those outputs
    cfg = O.LossConfig(frame_ids=opt.frame_ids, scales=opt.scales, height=opt.height, width=opt.width, automask=True)
    o_in = {"K": data["K"], "inv_K": data["inv_K"]}
    for f in opt.frame_ids:
        o_in[("image", f)] = data[("image", f)].double()
    disp = outputs[("disp", 0)].detach().cpu().requires_grad_(True)
    o_out = {("disp", 0): disp}
    for i, f in enumerate(opt.frame_ids[1:]):
        aa = outputs[("axisangle", f)].detach().cpu()
        tr = outputs[("translation", f)].detach().cpu()
        k = 0 if mode == "pairs" else i
        o_out[("cam_T_cam", f)] = O.transformation_from_parameters(aa[:, k], tr[:, k], invert=(mode == "pairs" and f < 0))
    ref = O.loss_forward_backward(o_in, o_out, cfg, data["noise"])
    for k in losses:
        assert abs(losses[k].item() - ref[k].item()) <= 1e-5 * abs(ref[k].item()), k


def test_train_steps_reduce_the_loss():
    opt, models, params, step, data, inputs = _setup("all")
    before = [p.detach().clone() for p in params]
    values = [step.train_step(inputs, data["noise"].to(DEV))[1]["loss"].item() for _ in range(8)]
    assert all(np.isfinite(values))
    assert values[-1] < values[0]
    assert any(not torch.equal(a, b) for a, b in zip(before, params))


def _setup_typed(kind, mode, B=2, H=32, W=96):
    """pose_model_type "posecnn" / "separate_resnet" (Python/Trainer.py:369-384,397-405): the pose network sees the
    stacked images (through its own encoder for "separate_resnet"), the depth network only frame 0."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import synthetic_train as ex
    import networks as N
    import mde_amd.synth as synth
    from mde_amd.trainer_step import TrainStep
    opt = types.SimpleNamespace(
        scales=[0], frame_ids=[0, -1, 1], height=H, width=W, min_depth=0.1, max_depth=100.0, v1_multiscale=False,
        disable_automasking=False, avg_reprojection=False, no_ssim=False, disparity_smoothness=1e-3,
        pose_model_type=kind, pose_model_input=mode)
    torch.manual_seed(5)
    n_in = 2 if mode == "pairs" else 3
    models = {"encoder": ex.TinyEncoder().to(DEV), "depth": ex.TinyDepthDecoder(opt.scales).to(DEV)}
    if kind == "posecnn":
        models["pose"] = N.PoseCNN(n_in).to(DEV)
    else:
        models["pose_encoder"] = ex.TinyEncoder(cin=3 * n_in).to(DEV)
        models["pose"] = ex.TinyPoseDecoder(1).to(DEV) if mode == "pairs" else _AllFrames(ex.TinyPoseDecoder(1), 2).to(DEV)
    params = [p for m in models.values() for p in m.parameters()]
    step = TrainStep(opt, models, torch.optim.Adam(params, 1e-3))
    data = synth.make_batch(B, H, W, opt.frame_ids, seed=22)
    inputs = {"K": data["K"].to(DEV), "inv_K": data["inv_K"].to(DEV)}
    for f in opt.frame_ids:
        inputs[("image", f)] = data[("image", f)].to(DEV)
        inputs[("image_aug", f)] = inputs[("image", f)]
    return opt, models, params, step, data, inputs


class _AllFrames(torch.nn.Module):
    """[B,1,1,6]-style pose head widened to n frames (separate_resnet, "all" mode: one encoder pass, n poses)."""

    def __init__(self, inner, n):
        super().__init__()
        self.inner, self.n = inner, n

    def forward(self, feats):
        a, t = self.inner(feats)
        scale = torch.arange(1, self.n + 1, device=a.device, dtype=a.dtype).view(1, self.n, 1, 1)
        return a * scale, t * scale


@pytest.mark.parametrize("kind,mode", [("posecnn", "pairs"), ("posecnn", "all"), ("separate_resnet", "pairs"),
                                       ("separate_resnet", "all")])
def test_process_batch_with_image_fed_pose_networks(kind, mode):
    opt, models, params, step, data, inputs = _setup_typed(kind, mode)
    noise = data["noise"].to(DEV)
    outputs, losses = step.process_batch(inputs, noise)
    losses["loss"].backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params)
    assert sum(float(p.grad.abs().sum()) for p in models["pose"].parameters()) > 0
    # the oracle on the same network outputs
    cfg = O.LossConfig(frame_ids=opt.frame_ids, scales=opt.scales, height=opt.height, width=opt.width, automask=True,
                       pose_model_type="posecnn" if kind == "posecnn" else "shared")
    o_in = {"K": data["K"], "inv_K": data["inv_K"]}
    for f in opt.frame_ids:
        o_in[("image", f)] = data[("image", f)].double()
    o_out = {("disp", 0): outputs[("disp", 0)].detach().cpu().requires_grad_(True)}
    for i, f in enumerate(opt.frame_ids[1:]):
        aa = outputs[("axisangle", f)].detach().cpu()
        tr = outputs[("translation", f)].detach().cpu()
        if kind == "posecnn":       # the warp rebuilds T from slice 0 of the parameters for every source, inverted
            o_out[("axisangle", f)], o_out[("translation", f)] = aa, tr      # for negative ids (Trainer.py:446-455)
        k = 0 if mode == "pairs" else i
        o_out[("cam_T_cam", f)] = O.transformation_from_parameters(aa[:, k], tr[:, k], invert=(mode == "pairs" and f < 0))
    ref = O.loss_forward_backward(o_in, o_out, cfg, data["noise"])
    for k in losses:
        assert abs(losses[k].item() - ref[k].item()) <= 1e-5 * abs(ref[k].item()), k


def test_train_step_under_ddp_rccl_world_size_one():
    """RCCL initialisation and DistributedDataParallel's gradient hooks around the custom autograd.Function on the
    real device: one rank, backend "nccl", three optimizer steps; losses equal the un-wrapped run (to 1e-6: the
    convolution backward of the stand-in networks is not bit-reproducible from run to run)."""
    import torch.distributed as td
    from mde_amd.trainer_step import NetworkPass, TrainStep
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    runs = []
    for wrap in (False, True):
        opt, models, params, _, data, inputs = _setup("all")
        net = NetworkPass(opt, models)
        if wrap:
            td.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
            net = torch.nn.parallel.DistributedDataParallel(net, device_ids=[0])
        try:
            step = TrainStep(opt, net, torch.optim.Adam(params, 1e-3))
            runs.append([step.train_step(inputs, data["noise"].to(DEV))[1]["loss"].item() for _ in range(3)])
            torch.cuda.synchronize()
            if wrap:
                # the collectives bench.py --gpus N and the sharded driver issue, on the same RCCL group
                import mde_amd.dist as D
                D.barrier()
                assert D.max_over_ranks(1.25, torch.device(DEV)) == 1.25
                g = [torch.full((5,), 3.0, device=DEV), torch.full((2, 3), -1.0, device=DEV)]
                D.all_reduce_mean_(g, bucket_bytes=16)
                assert float(g[0].sum()) == 15.0 and float(g[1].sum()) == -6.0
        finally:
            if wrap:
                td.destroy_process_group()
    assert all(np.isfinite(runs[1])) and runs[1][-1] < runs[1][0]
    assert runs[0][0] == runs[1][0] and np.allclose(runs[0], runs[1], rtol=1e-6, atol=0)


def test_train_step_with_autocast_networks_and_materialised_outputs():
    """Networks under bfloat16 autocast, loss in float32; with materialise the loss launch itself leaves
    ("depth",0), ("sample",f), ("color",f) in `outputs` (Python/Trainer.py:436,464,467), equal to the per-module chain."""
    from mde_amd.trainer_step import TrainStep
    opt, models, params, _, data, inputs = _setup("all")
    step = TrainStep(opt, models, torch.optim.Adam(params, 1e-3), materialise=True, autocast_dtype=torch.bfloat16)
    noise = data["noise"].to(DEV)
    outputs, losses = step.process_batch(inputs, noise)
    assert outputs[("disp", 0)].dtype == torch.float32 and outputs[("cam_T_cam", -1)].dtype == torch.float32
    assert torch.isfinite(losses["loss"]).all()
    losses["loss"].backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params)
    ref = dict(outputs)
    for k in [("depth", 0)] + [(n, f) for n in ("sample", "color") for f in (-1, 1)]:
        ref.pop(k)
    step.loss.generate_images_pred(inputs, ref)                      # per-module kernels on the same network outputs
    assert torch.equal(outputs[("depth", 0)], ref[("depth", 0)])
    for f in (-1, 1):
        assert torch.equal(outputs[("sample", f)], ref[("sample", f)])
        assert torch.equal(outputs[("color", f)], ref[("color", f)].float())
    values = [step.train_step(inputs, noise)[1]["loss"].item() for _ in range(6)]
    assert all(np.isfinite(values)) and values[-1] < values[0]
