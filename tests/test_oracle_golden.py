"""The CPU oracle against the fixtures produced by the reference itself.

The oracle uses the same PyTorch primitives in the same order as the reference,
so agreement is expected to the last bits of float64 / float32.  Its three tiny
matrix products run either through torch.matmul like the reference ("blas") or in
the spelled-out order that reproduces the fixtures' bits on every host ("fixed",
the default): forward values are bit-identical in both modes, gradients are
identical in "blas" mode on the generating host and within 3e-6 in "fixed" mode
(the backward of the spelled-out products rounds differently)."""
import numpy as np
import pytest
import torch

import json
import os

from golden_util import GOLDEN_DIR, build_dicts, case_names, load_case
from oracle import vsl_oracle as O


def make_cfg(cfg):
    variant = cfg["variant"]
    return O.LossConfig(
        frame_ids=[("s" if f == "s" else int(f)) for f in cfg["frame_ids"]], scales=cfg["scales"],
        height=cfg["H"], width=cfg["W"], automask=cfg["automask"], avg_reprojection=cfg["avg_reprojection"],
        no_ssim=cfg["no_ssim"], warp_scale=0.9 if variant == "distil" else 1.0)


@pytest.fixture(params=["fixed", "blas"])
def contraction_mode(request):
    old = O.FIXED_ORDER
    O.FIXED_ORDER = request.param == "fixed"
    yield request.param
    O.FIXED_ORDER = old


@pytest.mark.parametrize("name", case_names())
def test_oracle_matches_reference_fixture(name, contraction_mode):
    torch.set_num_threads(1)
    gtol = 3e-6 if contraction_mode == "fixed" else 1e-6
    cfg, z = load_case(name)
    ocfg = make_cfg(cfg)
    inputs, outputs, disp, poses, noise, fids = build_dicts(cfg, z)
    losses = O.loss_forward_backward(inputs, outputs, ocfg, noise)

    np.testing.assert_array_equal(outputs[("depth", 0)].detach().numpy(), z["depth"])
    for f in fids[1:]:
        np.testing.assert_array_equal(outputs[("sample", f)].detach().numpy(), z["sample_%s" % f])
        np.testing.assert_allclose(outputs[("color", f)].detach().numpy(), z["color_%s" % f], rtol=0, atol=1e-15)
        with torch.no_grad():
            rp = O.reprojection_error(outputs[("color", f)], inputs[("image", 0)], ocfg.no_ssim)
        np.testing.assert_allclose(rp.numpy(), z["rp_%s" % f], rtol=1e-13, atol=1e-15)
    if cfg["automask"]:
        np.testing.assert_array_equal(outputs["identity_selection"].numpy(), z["identity_selection"])
    np.testing.assert_allclose(losses["loss"].item(), float(z["loss"]), rtol=1e-13)
    for s in cfg["scales"]:
        np.testing.assert_allclose(losses["loss/%d" % s].item(), float(z["loss_%d" % s]), rtol=1e-13)
    g = disp.grad.numpy()
    assert np.linalg.norm(g - z["grad_disp"]) <= gtol * np.linalg.norm(z["grad_disp"])
    for f in fids[1:]:
        gt, gr = poses[f].grad.numpy(), z["grad_T_%s" % f]
        assert np.linalg.norm(gt - gr) <= gtol * np.linalg.norm(gr)


def test_backproject_points_match_fixture():
    cfg, z = load_case("b2_24x80_automask")
    depth = torch.from_numpy(z["depth"])
    pts = O.backproject(depth, torch.from_numpy(z["inv_K"]))
    np.testing.assert_array_equal(pts.numpy(), z["points"])


def test_pose_helper_matches_synth():
    import mde_amd.synth as synth
    aa = torch.from_numpy(synth.normal(3, (4, 1, 3)).astype(np.float32)) * 0.1
    tr = torch.from_numpy(synth.normal(4, (4, 1, 3)).astype(np.float32)) * 0.1
    M = O.transformation_from_parameters(aa, tr)
    M2 = synth._rodrigues(aa[:, 0].numpy(), tr[:, 0].numpy())
    np.testing.assert_allclose(M.numpy(), M2, atol=1e-6)
    Mi = O.transformation_from_parameters(aa, tr, invert=True)
    np.testing.assert_allclose(torch.matmul(M, Mi).numpy(), np.tile(np.eye(4), (4, 1, 1)), atol=1e-6)


def test_pose_parameters_match_reference_fixture():
    """layers.transformation_from_parameters of the reference on slices of a [B,F,1,3] output, both invert
    settings, values and gradients (fixture written by oracle/make_golden.py from the reference import)."""
    z = np.load(os.path.join(GOLDEN_DIR, "pose", "pose_params.npz"))
    F_ = z["axisangle"].shape[1]
    for i in range(F_):
        for inv in (0, 1):
            aa = torch.from_numpy(z["axisangle"]).clone().requires_grad_(True)
            tr = torch.from_numpy(z["translation"]).clone().requires_grad_(True)
            T = O.transformation_from_parameters(aa[:, i], tr[:, i], invert=bool(inv))
            (T * torch.from_numpy(z["g_T"][i, inv])).sum().backward()
            np.testing.assert_allclose(T.detach().numpy(), z["T_%d_%d" % (i, inv)], rtol=0, atol=1e-7)
            np.testing.assert_allclose(aa.grad.numpy(), z["g_axisangle_%d_%d" % (i, inv)], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(tr.grad.numpy(), z["g_translation_%d_%d" % (i, inv)], rtol=1e-6, atol=1e-7)


def test_posecnn_mode_matches_reference_fixture():
    """Trainer.generate_images_pred in the "posecnn" mode (Trainer.py:446-455): T from the parameters and the
    per-image mean inverse depth; loss and the gradients of disparity and pose parameters."""
    z = np.load(os.path.join(GOLDEN_DIR, "pose", "b2_24x80_posecnn.npz"))
    c = json.loads(str(z["config"]))
    fids = c["frame_ids"]
    cfg = O.LossConfig(frame_ids=fids, scales=c["scales"], height=c["H"], width=c["W"], automask=True,
                       pose_model_type="posecnn")
    inputs = {"K": torch.from_numpy(z["K"]), "inv_K": torch.from_numpy(z["inv_K"])}
    for f in fids:
        inputs[("image", f)] = torch.from_numpy(z["image_%s" % f]).double()
    disp = torch.from_numpy(z["disp"]).clone().requires_grad_(True)
    outputs = {("disp", 0): disp}
    params = {}
    for f in fids[1:]:
        aa = torch.from_numpy(z["axisangle_%s" % f]).clone().requires_grad_(True)
        tr = torch.from_numpy(z["translation_%s" % f]).clone().requires_grad_(True)
        params[f] = (aa, tr)
        outputs[("axisangle", f)], outputs[("translation", f)] = aa, tr
        outputs[("cam_T_cam", f)] = O.transformation_from_parameters(aa[:, 0], tr[:, 0], invert=(f < 0))
    losses = O.loss_forward_backward(inputs, outputs, cfg, torch.from_numpy(z["noise"]))
    assert abs(losses["loss"].item() - float(z["loss"])) <= 1e-7 * abs(float(z["loss"]))
    np.testing.assert_array_equal(outputs["identity_selection"].numpy(), z["identity_selection"])
    tol = 1e-5

    def close(a, b):
        return np.linalg.norm(a - b) <= tol * np.linalg.norm(b)

    assert close(disp.grad.numpy(), z["grad_disp"])
    for f in fids[1:]:
        assert close(params[f][0].grad.numpy(), z["g_axisangle_%s" % f])
        assert close(params[f][1].grad.numpy(), z["g_translation_%s" % f])


@pytest.mark.parametrize("name", ["feat_12x40", "feat_6x20", "feat_24x80", "feat_13x37_large_motion"])
def test_feature_warp_matches_reference_fixture(name):
    """Coarse-scale feature warping (Train_distillation_new.py:1159-1182) replayed on the reference's layers:
    grid bit-identical, warped features and gradients to float32 rounding."""
    z = np.load(os.path.join(GOLDEN_DIR, "warp", name + ".npz"))
    feats = torch.from_numpy(z["feats"]).clone().requires_grad_(True)
    disp = torch.from_numpy(z["disp"]).clone().requires_grad_(True)
    T = torch.from_numpy(z["T"]).clone().requires_grad_(True)
    out, grid = O.warp_features(feats, disp, torch.from_numpy(z["K"]), torch.from_numpy(z["inv_K"]), T)
    np.testing.assert_array_equal(grid.detach().numpy(), z["sample"])
    np.testing.assert_allclose(out.detach().numpy(), z["out"], rtol=0, atol=1e-6)
    (out * torch.from_numpy(z["g_out"])).sum().backward()
    for got, key in ((feats.grad, "g_feats"), (disp.grad, "g_disp"), (T.grad, "g_T")):
        err = np.linalg.norm(got.numpy() - z[key]) / np.linalg.norm(z[key])
        assert err <= 1e-5, "%s %.2e" % (key, err)


def test_affinity_term_matches_reference_fixture():
    """Trainer.py:468-473,497-516,546-548,606 with use_affinity=True: the warped 25-channel maps, the loss per scale
    and the gradients of disparity and poses."""
    z = np.load(os.path.join(GOLDEN_DIR, "affinity", "b2_24x80_affinity.npz"))
    c = json.loads(str(z["config"]))
    fids = c["frame_ids"]
    cfg = O.LossConfig(frame_ids=fids, scales=c["scales"], height=c["H"], width=c["W"], automask=True, use_affinity=True)
    inputs = {"K": torch.from_numpy(z["K"]), "inv_K": torch.from_numpy(z["inv_K"])}
    for f in fids:
        inputs[("image", f)] = torch.from_numpy(z["image_%s" % f]).double()
        inputs[("affinity", f)] = torch.from_numpy(z["affinity_%s" % f])
    disp = torch.from_numpy(z["disp"]).clone().requires_grad_(True)
    outputs = {("disp", 0): disp}
    poses = {}
    for f in fids[1:]:
        poses[f] = torch.from_numpy(z["T_%s" % f]).clone().requires_grad_(True)
        outputs[("cam_T_cam", f)] = poses[f]
    losses = O.loss_forward_backward(inputs, outputs, cfg, torch.from_numpy(z["noise"]), torch.from_numpy(z["drop_rand"]))
    assert abs(losses["loss"].item() - float(z["loss"])) <= 1e-7 * abs(float(z["loss"]))
    for s_ in c["scales"]:
        assert abs(losses["loss/%d" % s_].item() - float(z["loss_%d" % s_])) <= 1e-7 * abs(float(z["loss_%d" % s_]))
    for f in fids[1:]:
        np.testing.assert_allclose(outputs[("pred_affinity", f)].detach().numpy(), z["pred_affinity_%s" % f], rtol=0, atol=1e-6)
        err = np.linalg.norm(poses[f].grad.numpy() - z["grad_T_%s" % f]) / np.linalg.norm(z["grad_T_%s" % f])
        assert err <= 1e-5, "grad_T %s %.2e" % (f, err)
    err = np.linalg.norm(disp.grad.numpy() - z["grad_disp"]) / np.linalg.norm(z["grad_disp"])
    assert err <= 1e-5, "grad_disp %.2e" % err


@pytest.mark.parametrize("name", ["b2_48x160", "b3_24x80_even_count"])
def test_depth_metrics_match_reference_fixture(name):
    """Trainer.compute_depth_losses (Trainer.py:622-662): same torch ops, so the seven metrics agree to rounding."""
    z = np.load(os.path.join(GOLDEN_DIR, "metrics", name + ".npz"))
    total, metrics = O.depth_losses(torch.from_numpy(z["depth"]), torch.from_numpy(z["depth_gt"]))
    np.testing.assert_allclose(np.array([float(v) for v in metrics]), z["metrics"], rtol=1e-6)
    assert abs(float(total) - float(z["total"])) <= 1e-6 * abs(float(z["total"]))


def test_affinity_huber_variant_matches_reference_fixture():
    """Trainer_distilation.py:1076-1110 called directly on a float64 prediction: value and gradient."""
    z = np.load(os.path.join(GOLDEN_DIR, "affinity", "huber_b2_24x80.npz"))
    pred = torch.from_numpy(z["pred"]).double().requires_grad_(True)
    loss = O.affinity_huber_loss(pred, torch.from_numpy(z["target"]))
    loss.backward()
    assert abs(float(loss) - float(z["loss"])) <= 1e-12
    np.testing.assert_allclose(pred.grad.numpy(), z["grad_pred"], rtol=0, atol=1e-15)
