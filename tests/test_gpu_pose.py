"""GPU parity of the pose preparation (vsl_pose_fwd/bwd, vsl_mean_inv_depth_fwd/bwd through the C-ABI)
against the fixtures written from the reference's layers.transformation_from_parameters and from its
"posecnn" branch of Trainer.generate_images_pred, and against the CPU oracle.
Tolerance: 1e-5 relative (L2) on every matrix / gradient tensor, stated per assertion."""
import json
import os
import types

import numpy as np
import pytest
import torch

from golden_util import GOLDEN_DIR
from oracle import vsl_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5


def rel(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, "pose", name + ".npz"), allow_pickle=False)


@pytest.mark.parametrize("inv", [0, 1])
@pytest.mark.parametrize("i", [0, 1])
def test_transformation_from_parameters_matches_reference_fixture(i, inv):
    """One source at a time, on slices [:, i] of the [B,F,1,3] network outputs (strided rows)."""
    import mde_amd.layers as L
    z = load("pose_params")
    aa = torch.from_numpy(z["axisangle"]).to(DEV).requires_grad_(True)
    tr = torch.from_numpy(z["translation"]).to(DEV).requires_grad_(True)
    T = L.transformation_from_parameters(aa[:, i], tr[:, i], invert=bool(inv))
    assert T.shape == (aa.shape[0], 4, 4)
    assert rel(T, z["T_%d_%d" % (i, inv)]) <= TOL
    np.testing.assert_array_equal(T[:, 3].detach().cpu().numpy(), np.tile([0, 0, 0, 1], (aa.shape[0], 1)))
    (T * torch.from_numpy(z["g_T"][i, inv]).to(DEV)).sum().backward()
    assert rel(aa.grad, z["g_axisangle_%d_%d" % (i, inv)]) <= TOL
    assert rel(tr.grad, z["g_translation_%d_%d" % (i, inv)]) <= TOL
    # per image too: the rows span rotation angles from 0.01 to 2 rad and one exact zero
    for b in range(aa.shape[0]):
        ref = z["g_axisangle_%d_%d" % (i, inv)][b, i]
        assert rel(aa.grad[b, i], ref) <= 5 * TOL, "image %d" % b
    assert torch.isfinite(aa.grad).all()


def test_all_sources_in_one_launch_match_single_calls():
    """Mixed invert flags, both slices at once: equal to the one-by-one results bit for bit, gradients add up."""
    import mde_amd.layers as L
    z = load("pose_params")
    aa = torch.from_numpy(z["axisangle"]).to(DEV).requires_grad_(True)
    tr = torch.from_numpy(z["translation"]).to(DEV).requires_grad_(True)
    Ts = L.poses_from_parameters([aa[:, 0], aa[:, 1], aa[:, 1]], [tr[:, 0], tr[:, 1], tr[:, 0]], [True, False, True])
    g = [torch.from_numpy(z["g_T"][0, 1]).to(DEV), torch.from_numpy(z["g_T"][1, 0]).to(DEV),
         torch.from_numpy(z["g_T"][1, 1]).to(DEV)]
    sum((T * w).sum() for T, w in zip(Ts, g)).backward()
    aa2 = torch.from_numpy(z["axisangle"]).to(DEV).requires_grad_(True)
    tr2 = torch.from_numpy(z["translation"]).to(DEV).requires_grad_(True)
    singles = [L.transformation_from_parameters(aa2[:, 0], tr2[:, 0], True),
               L.transformation_from_parameters(aa2[:, 1], tr2[:, 1], False),
               L.transformation_from_parameters(aa2[:, 1], tr2[:, 0], True)]
    sum((T * w).sum() for T, w in zip(singles, g)).backward()
    for a, b in zip(Ts, singles):
        assert torch.equal(a, b)
    assert rel(aa.grad, aa2.grad) <= 1e-6 and rel(tr.grad, tr2.grad) <= 1e-6
    assert rel(Ts[0], z["T_0_1"]) <= TOL and rel(Ts[1], z["T_1_0"]) <= TOL


def test_translation_scale_and_its_gradient_match_oracle():
    import mde_amd.layers as L
    z = load("pose_params")
    B = z["axisangle"].shape[0]
    sc = torch.linspace(0.5, 6.0, B)
    g_T = torch.from_numpy(z["g_T"][0])     # [2, B, 4, 4]
    leaves = {}
    results = {}
    for where in ("gpu", "cpu"):
        dev = DEV if where == "gpu" else "cpu"
        aa = torch.from_numpy(z["axisangle"]).to(dev).requires_grad_(True)
        tr = torch.from_numpy(z["translation"]).to(dev).requires_grad_(True)
        s = sc.clone().to(dev).requires_grad_(True)
        if where == "gpu":
            Ts = L.poses_from_parameters([aa[:, 0], aa[:, 1]], [tr[:, 0], tr[:, 1]], [True, False], t_scale=s)
        else:
            Ts = [O.transformation_from_parameters(aa[:, 0], tr[:, 0] * s.view(B, 1, 1), True),
                  O.transformation_from_parameters(aa[:, 1], tr[:, 1] * s.view(B, 1, 1), False)]
        sum((T * g_T[k].to(dev)).sum() for k, T in enumerate(Ts)).backward()
        leaves[where] = (aa.grad, tr.grad, s.grad)
        results[where] = Ts
    for a, b in zip(results["gpu"], results["cpu"]):
        assert rel(a, b) <= TOL
    for a, b in zip(leaves["gpu"], leaves["cpu"]):
        assert rel(a, b) <= TOL


def test_mean_inverse_depth_and_gradient():
    import mde_amd.layers as L
    z = load("b2_24x80_posecnn")
    disp = torch.from_numpy(z["disp"]).to(DEV).requires_grad_(True)
    m = L.mean_inverse_depth(disp, 0.1, 100.0)
    assert rel(m, z["mean_inv_depth"].reshape(-1)) <= TOL
    w = torch.tensor([0.7, -1.3], device=DEV)
    (m * w).sum().backward()
    d = torch.from_numpy(z["disp"]).requires_grad_(True)
    _, depth = O.disp_to_depth(d, 0.1, 100.0)
    ((1 / depth).mean(3, True).mean(2, True).reshape(-1) * w.cpu()).sum().backward()
    assert rel(disp.grad, d.grad) <= TOL
    # an odd size that does not divide into the partial-sum chunks
    d2 = torch.rand(3, 1, 37, 53, device=DEV)
    ref = (1 / O.disp_to_depth(d2.cpu(), 0.1, 100.0)[1]).double().mean((1, 2, 3))
    assert rel(L.mean_inverse_depth(d2, 0.1, 100.0), ref) <= TOL


def _posecnn_float64(z, c):
    """Gradients of the pose parameters from the CPU oracle evaluated entirely in float64."""
    fids = c["frame_ids"]
    cfg = O.LossConfig(frame_ids=fids, scales=c["scales"], height=c["H"], width=c["W"], automask=True,
                       pose_model_type="posecnn")
    inputs = {"K": torch.from_numpy(z["K"]).double(), "inv_K": torch.from_numpy(z["inv_K"]).double()}
    for f in fids:
        inputs[("image", f)] = torch.from_numpy(z["image_%s" % f]).double()
    outputs = {("disp", 0): torch.from_numpy(z["disp"]).double().requires_grad_(True)}
    params = {}
    for f in fids[1:]:
        params[f] = tuple(torch.from_numpy(z["%s_%s" % (n, f)]).double().requires_grad_(True)
                          for n in ("axisangle", "translation"))
        outputs[("axisangle", f)], outputs[("translation", f)] = params[f]
        outputs[("cam_T_cam", f)] = O.transformation_from_parameters(params[f][0][:, 0], params[f][1][:, 0], f < 0)
    old = O.FIXED_ORDER
    O.FIXED_ORDER = False        # the spelled-out float32 contractions do not apply to float64
    try:
        O.loss_forward_backward(inputs, outputs, cfg, torch.from_numpy(z["noise"]))
    finally:
        O.FIXED_ORDER = old
    return {f: (params[f][0].grad.numpy(), params[f][1].grad.numpy()) for f in fids[1:]}


def test_posecnn_loss_matches_reference_fixture():
    """compute_losses with opt.pose_model_type == "posecnn": pose matrices rebuilt from the parameters and the
    mean inverse depth (Trainer.py:446-455), gradients down to disparity, axis-angle and translation."""
    from mde_amd.trainer_loss import ViewSynthesisLoss
    z = load("b2_24x80_posecnn")
    c = json.loads(str(z["config"]))
    fids = c["frame_ids"]
    opt = types.SimpleNamespace(scales=c["scales"], frame_ids=fids, height=c["H"], width=c["W"], min_depth=0.1,
                                max_depth=100.0, v1_multiscale=False, disable_automasking=False, avg_reprojection=False,
                                no_ssim=False, disparity_smoothness=1e-3, pose_model_type="posecnn")
    vs = ViewSynthesisLoss(opt)
    inputs = {"K": torch.from_numpy(z["K"]).to(DEV), "inv_K": torch.from_numpy(z["inv_K"]).to(DEV)}
    for f in fids:
        inputs[("image", f)] = torch.from_numpy(z["image_%s" % f]).to(DEV)
    disp = torch.from_numpy(z["disp"]).to(DEV).requires_grad_(True)
    outputs = {("disp", 0): disp}
    params = {}
    for f in fids[1:]:
        aa = torch.from_numpy(z["axisangle_%s" % f]).to(DEV).requires_grad_(True)
        tr = torch.from_numpy(z["translation_%s" % f]).to(DEV).requires_grad_(True)
        params[f] = (aa, tr)
        outputs[("axisangle", f)], outputs[("translation", f)] = aa, tr
    losses = vs.compute_losses(inputs, outputs, noise=torch.from_numpy(z["noise"]).to(DEV))
    losses["loss"].backward()
    assert abs(losses["loss"].item() - float(z["loss"])) <= TOL * abs(float(z["loss"]))
    # (the fixture's seed was chosen so that no pixel's two best candidates are closer than 1e-6: the device's
    # sinf / cosf differ from the host's libm in the last bits, which moves sampling coordinates by an ulp)
    flips = int((torch.from_numpy(z["identity_selection"]).to(DEV) != outputs["identity_selection"]).sum())
    assert flips == 0, "%d identity_selection flips" % flips
    err = rel(disp.grad, z["grad_disp"])
    assert err <= TOL, "grad_disp %.3e" % err
    # Pose-parameter gradients are sums of heavy-tailed per-pixel terms scaled by the mean inverse depth; the
    # reference's own float32 result is 2.5e-5 .. 5.4e-5 away from a float64 evaluation of the same graph, so
    # 1e-5 between two float32 evaluations is not attainable here.  Bar: the device result is no further from
    # the float64 evaluation than twice the reference's float32 result is (+1e-5), and within 1e-4 of the fixture.
    exact = _posecnn_float64(z, c)
    for f in fids[1:]:
        for k, name in enumerate(("axisangle", "translation")):
            ref32 = z["g_%s_%s" % (name, f)]
            e_dev, e_ref = rel(params[f][k].grad, exact[f][k]), rel(ref32, exact[f][k])
            assert e_dev <= 2 * e_ref + TOL, "%s %s: device %.2e, reference %.2e from float64" % (name, f, e_dev, e_ref)
            e_fix = rel(params[f][k].grad, ref32)
            assert e_fix <= 1e-4, "%s %s: %.2e from the fixture" % (name, f, e_fix)
    # the un-fused path (generate_images_pred) lands on the reference's sampling coordinates
    vs.generate_images_pred(inputs, outputs)
    for f in fids[1:]:
        es = rel(outputs[("sample", f)], z["sample_%s" % f])
        assert es <= TOL, "sample %s %.3e" % (f, es)


def test_pose_outputs_all_and_pairs_modes():
    """Trainer.predict_pose's dict entries: "all" mode slices one [B,F,1,3] output without inversion
    (Trainer.py:408-414), "pairs" mode inverts negative frame ids (Trainer.py:388-393)."""
    from mde_amd.trainer_loss import ViewSynthesisLoss
    z = load("pose_params")
    opt = types.SimpleNamespace(scales=[0], frame_ids=[0, -1, 1], height=24, width=80, min_depth=0.1, max_depth=100.0,
                                v1_multiscale=False, disable_automasking=False, avg_reprojection=False, no_ssim=False,
                                disparity_smoothness=1e-3)
    vs = ViewSynthesisLoss(opt)
    aa = torch.from_numpy(z["axisangle"]).to(DEV)
    tr = torch.from_numpy(z["translation"]).to(DEV)
    out = vs.pose_outputs(aa, tr)
    assert rel(out[("cam_T_cam", -1)], z["T_0_0"]) <= TOL and rel(out[("cam_T_cam", 1)], z["T_1_0"]) <= TOL
    assert out[("axisangle", 1)] is aa and out[("translation", -1)] is tr
    pairs = vs.pose_outputs({-1: aa[:, :1], 1: aa[:, 1:]}, {-1: tr[:, :1], 1: tr[:, 1:]})
    assert rel(pairs[("cam_T_cam", -1)], z["T_0_1"]) <= TOL and rel(pairs[("cam_T_cam", 1)], z["T_1_0"]) <= TOL


def test_pose_argument_errors():
    import mde_amd.layers as L
    a = torch.zeros(2, 1, 3, device=DEV)
    with pytest.raises(ValueError):
        L.transformation_from_parameters(a.cpu(), a.cpu())
    with pytest.raises(ValueError):
        L.transformation_from_parameters(torch.zeros(2, 2, 3, device=DEV), a)
    with pytest.raises(ValueError):
        L.poses_from_parameters([a] * 5, [a] * 5)
    with pytest.raises(ValueError):
        L.poses_from_parameters([a], [torch.zeros(3, 1, 3, device=DEV)])
    with pytest.raises(ValueError):
        L.poses_from_parameters([a], [a], t_scale=torch.ones(5, device=DEV))


def test_posecnn_preparation_in_two_launches_equals_the_three_launch_chain():
    """poses_from_parameters(posecnn_disp=...) -- the mean inverse depth finished inside the pose launch
    (vsl_pose_fwd_posecnn) -- against mean_inverse_depth() followed by poses_from_parameters(t_scale=...): the same
    pose matrices and the same gradients of the pose parameters and the disparity, bit for bit."""
    import mde_amd.layers as L
    z = load("b2_24x80_posecnn")
    B = z["disp"].shape[0]
    g = [torch.randn(B, 4, 4, device=DEV, generator=torch.Generator(device=DEV).manual_seed(7 + i)) for i in range(2)]

    def run(two_launch):
        disp = torch.from_numpy(z["disp"]).to(DEV).requires_grad_(True)
        aa = [torch.from_numpy(z["axisangle_%d" % f]).to(DEV).requires_grad_(True) for f in (-1, 1)]
        tr = [torch.from_numpy(z["translation_%d" % f]).to(DEV).requires_grad_(True) for f in (-1, 1)]
        a_list, t_list, inv = [a[:, 0] for a in aa], [t[:, 0] for t in tr], [True, False]
        if two_launch:
            Ts = L.poses_from_parameters(a_list, t_list, inv, posecnn_disp=disp, depth_range=(0.1, 100.0))
        else:
            Ts = L.poses_from_parameters(a_list, t_list, inv, t_scale=L.mean_inverse_depth(disp, 0.1, 100.0))
        sum((T * w).sum() for T, w in zip(Ts, g)).backward()
        return [T.detach() for T in Ts], disp.grad, torch.stack([a.grad for a in aa]), torch.stack([t.grad for t in tr])

    T2, gd2, ga2, gt2 = run(True)
    T3, gd3, ga3, gt3 = run(False)
    for a, b in zip(T2, T3):
        assert torch.equal(a, b)
    assert torch.equal(gd2, gd3) and torch.equal(ga2, ga3) and torch.equal(gt2, gt3)
    assert float(gd2.abs().max()) > 0 and float(gt2.abs().max()) > 0
