"""Host-side behaviour of TrainStep that needs no GPU: the attributes the reference trainer keeps on itself
(Trainer.py:24,31-37) read through to the network module, and a contradicting use_pose_net is an error."""
import types

import pytest
import torch


def _opt():
    return types.SimpleNamespace(scales=[0], frame_ids=[0, -1, 1], height=24, width=80, min_depth=0.1, max_depth=100.0,
                                 v1_multiscale=False, disable_automasking=False, avg_reprojection=False, no_ssim=False,
                                 disparity_smoothness=1e-3, pose_model_type="shared", pose_model_input="all")


def test_trainstep_reads_through_to_the_network_module():
    from mde_amd.trainer_step import NetworkPass, TrainStep
    nets = {"encoder": torch.nn.Identity(), "depth": torch.nn.Identity(), "pose": torch.nn.Identity()}
    step = TrainStep(_opt(), nets)
    assert step.use_pose_net is True and step.num_pose_frames == 3 and set(step.models.keys()) == set(nets)
    net = NetworkPass(_opt(), nets, use_pose_net=False)
    assert TrainStep(_opt(), net).use_pose_net is False
    assert TrainStep(_opt(), net, use_pose_net=False).network_pass is net
    with pytest.raises(ValueError, match="use_pose_net"):
        TrainStep(_opt(), net, use_pose_net=True)


def test_trainstep_unwraps_a_wrapped_module():
    from mde_amd.trainer_step import NetworkPass, TrainStep

    class Wrapper(torch.nn.Module):            # stands in for DistributedDataParallel: the wrapped module under .module
        def __init__(self, m):
            super().__init__()
            self.module = m

    nets = {"encoder": torch.nn.Identity(), "depth": torch.nn.Identity(), "pose": torch.nn.Identity()}
    inner = NetworkPass(_opt(), nets, use_pose_net=False)
    step = TrainStep(_opt(), Wrapper(inner))
    assert step.network_pass is inner and step.use_pose_net is False
    with pytest.raises(ValueError):
        TrainStep(_opt(), Wrapper(inner), use_pose_net=True)
