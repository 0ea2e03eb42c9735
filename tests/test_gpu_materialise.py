"""The optional outputs of the fused forward launch -- outputs[("depth",0)], ("sample",f), ("color",f) of
Trainer.generate_images_pred (Python/Trainer.py:436,464,467) -- against the reference fixtures and against the
per-module kernels, and the losses / gradients of a launch with outputs against one without."""
import types

import numpy as np
import pytest
import torch

from golden_util import case_names, fid_key, load_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _opt(cfg, H, W):
    return types.SimpleNamespace(
        scales=list(cfg["scales"]), frame_ids=list(cfg["frame_ids"]), height=H, width=W, min_depth=0.1, max_depth=100.0,
        v1_multiscale=False, disable_automasking=not cfg["automask"], avg_reprojection=cfg["avg_reprojection"],
        no_ssim=cfg["no_ssim"], disparity_smoothness=1e-3, pose_model_type="shared")


@pytest.mark.parametrize("name", ["b2_24x80_automask", "b2_24x80_stereo", "b1_13x37_noautomask_avg", "b2_48x160_automask",
                                  "b3_16x33_sharp", "b2_24x80_large_motion", "b2_24x80_distil_scale09"])
def test_outputs_of_the_fused_launch_match_fixture_and_modules(name):
    from mde_amd.trainer_loss import ViewSynthesisLoss
    if name not in case_names():
        pytest.skip("fixture %s not present" % name)
    cfg, z = load_case(name)
    fids = list(cfg["frame_ids"])
    B, _, H, W = z["disp"].shape
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    inputs = {"K": t(z["K"]), "inv_K": t(z["inv_K"])}
    for f in fids:
        inputs[("image", f)] = t(z["image_%s" % fid_key(f)]).float()
    vs = ViewSynthesisLoss(_opt(cfg, H, W), photometric_scale=0.9 if cfg["variant"] == "distil" else 1.0)
    noise = t(z["noise"]) if cfg["automask"] else None

    def outputs_for(requires_grad):
        out = {("disp", 0): t(z["disp"]).requires_grad_(requires_grad)}
        for f in fids[1:]:
            T = t(z["T_%s" % fid_key(f)]).requires_grad_(requires_grad)
            if f == "s":
                inputs["stereo_T"] = T
            else:
                out[("cam_T_cam", f)] = T
        return out

    plain = outputs_for(True)
    l_plain = vs.compute_losses(inputs, plain, noise)
    l_plain["loss"].backward()
    fused = outputs_for(True)
    l_fused = vs.compute_losses(inputs, fused, noise, materialise=True)
    l_fused["loss"].backward()
    # the launch that also writes outputs computes the same loss and gradients, bit for bit
    assert torch.equal(l_plain["loss"], l_fused["loss"])
    assert torch.equal(plain[("disp", 0)].grad, fused[("disp", 0)].grad)
    # per-module kernels
    mod = outputs_for(False)
    with torch.no_grad():
        vs.generate_images_pred(inputs, mod)
    assert torch.equal(fused[("depth", 0)], mod[("depth", 0)])
    assert np.array_equal(fused[("depth", 0)].cpu().numpy(), z["depth"])                         # bit-exact vs the reference
    for f in fids[1:]:
        k = fid_key(f)
        assert torch.equal(fused[("sample", f)], mod[("sample", f)])
        assert np.array_equal(fused[("sample", f)].cpu().numpy(), z["sample_%s" % k])            # bit-exact vs the reference
        ref_color = z["color_%s" % k]                                                            # float64 in the reference
        got = fused[("color", f)].cpu().numpy()
        assert np.abs(got - ref_color).max() <= 2e-6, np.abs(got - ref_color).max()
        assert np.abs(got - mod[("color", f)].cpu().numpy()).max() <= 5e-7
        if cfg["automask"]:
            assert fused[("color_identity", f)] is inputs[("image", f)]
    assert not fused[("depth", 0)].requires_grad


def test_outputs_at_a_ragged_benchmark_shape_and_in_forward_only_mode():
    """640x192 b2 (tiles that end past the right edge) under torch.no_grad(): the forward-only tile program writes the
    same outputs as the full one."""
    import mde_amd.synth as synth
    from mde_amd.ops import fused_view_synthesis_loss
    B, H, W, fids = 2, 192, 640, [0, -1, 1]
    data = synth.make_batch(B, H, W, fids, seed=5)
    g = lambda k: data[k].to(DEV)
    args = (g(("image", 0)), [g(("image", f)) for f in fids[1:]], g("K"), g("inv_K"))
    poses = [g(("cam_T_cam", f)) for f in fids[1:]]
    full = fused_view_synthesis_loss(g(("disp", 0)).requires_grad_(True), *args, poses, g("noise"), materialise=True)
    with torch.no_grad():
        fwd = fused_view_synthesis_loss(g(("disp", 0)), *args, poses, g("noise"), materialise=True)
    for a, b in zip(full[2:], fwd[2:]):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    assert torch.equal(full[1], fwd[1])
