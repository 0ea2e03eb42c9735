"""GPU parity of the fused coarse-scale feature warp (vsl_warp_features_fwd/bwd through the C-ABI) against the
fixtures replayed from the reference's layers (Train_distillation_new.py:1159-1182) and against the module-by-module
kernels.  Sampling grid bit-exact; warped features and gradients 1e-5 relative (L2); the gradient of the features
is accumulated with float atomics, so its last bits vary from run to run."""
import ctypes
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN_DIR
from oracle import vsl_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5
CASES = ["feat_12x40", "feat_6x20", "feat_24x80", "feat_13x37_large_motion"]


def rel(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, "warp", name + ".npz"), allow_pickle=False)


def leaves(z, dev=DEV):
    t = {k: torch.from_numpy(z[k]).to(dev) for k in ("feats", "disp", "K", "inv_K", "T", "g_out")}
    for k in ("feats", "disp", "T"):
        t[k].requires_grad_(True)
    return t


@pytest.mark.parametrize("name", CASES)
def test_fused_feature_warp_matches_reference_fixture(name):
    import mde_amd.layers as L
    z = load(name)
    t = leaves(z)
    out, sample = L.warp_features(t["feats"], t["disp"], t["K"], t["inv_K"], t["T"], return_sample=True)
    flips = int((sample.cpu().numpy() != z["sample"]).sum())
    assert flips == 0, "%d sampling coordinates differ in their bits" % flips
    err = rel(out, z["out"])
    assert err <= TOL, "warped features %.2e" % err
    (out * t["g_out"]).sum().backward()
    errs = {k: rel(t[n].grad, z[k]) for n, k in (("feats", "g_feats"), ("disp", "g_disp"), ("T", "g_T"))}
    if name.endswith("large_motion"):
        # points next to the camera plane: 1/z^2 amplifies float32 rounding in the reference's own g_disp / g_T;
        # bounded against a float64 evaluation instead (no further from it than twice the reference is, +1e-5)
        f64 = {k: torch.from_numpy(z[k]).double() for k in ("feats", "disp", "K", "inv_K", "T")}
        for k in ("feats", "disp", "T"):
            f64[k].requires_grad_(True)
        old, O.FIXED_ORDER = O.FIXED_ORDER, False
        try:
            o64, _ = O.warp_features(f64["feats"], f64["disp"], f64["K"], f64["inv_K"], f64["T"])
        finally:
            O.FIXED_ORDER = old
        (o64 * torch.from_numpy(z["g_out"]).double()).sum().backward()
        for n, k in (("disp", "g_disp"), ("T", "g_T")):
            e_dev, e_ref = rel(t[n].grad, f64[n].grad), rel(z[k], f64[n].grad)
            assert e_dev <= 2 * e_ref + TOL, "%s: device %.2e, reference %.2e from float64" % (k, e_dev, e_ref)
        assert errs["g_feats"] <= TOL, "g_feats %.2e" % errs["g_feats"]
    else:
        assert all(e <= TOL for e in errs.values()), str(errs)


def test_fused_feature_warp_equals_module_by_module_kernels():
    """128 channels at 24x80, batch 12 (the size of Train_distillation_new.py:1159-1164): the fused pass against
    disp_to_depth -> BackprojectDepth -> Project3D -> grid_sample_border on the per-module kernels."""
    import mde_amd.layers as L
    import mde_amd.synth as synth
    B, C, H, W = 12, 128, 24, 80
    data = synth.make_batch(B, H, W, [0, -1], seed=61)
    g = torch.Generator().manual_seed(5)
    feats0 = torch.randn(B, C, H, W, generator=g)
    g_out = torch.randn(B, C, H, W, generator=g).to(DEV)
    res = {}
    for mode in ("fused", "modules"):
        feats = feats0.to(DEV).requires_grad_(True)
        disp = data[("disp", 0)].to(DEV).requires_grad_(True)
        T = data[("cam_T_cam", -1)].to(DEV).requires_grad_(True)
        K, inv_K = data["K"].to(DEV), data["inv_K"].to(DEV)
        if mode == "fused":
            out, sample = L.warp_features(feats, disp, K, inv_K, T, return_sample=True)
        else:
            _, depth = L.disp_to_depth(disp, 0.1, 100.0)
            sample = L.Project3D(B, H, W)(L.BackprojectDepth(B, H, W)(depth, inv_K), K, T)
            out = L.grid_sample_border(feats, sample)
        (out * g_out).sum().backward()
        res[mode] = (sample.detach(), out.detach(), feats.grad, disp.grad, T.grad)
    assert torch.equal(res["fused"][0], res["modules"][0])
    assert torch.equal(res["fused"][1], res["modules"][1])
    for i, what in ((2, "g_feats"), (3, "g_disp"), (4, "g_T")):
        err = rel(res["fused"][i], res["modules"][i])
        assert err <= TOL, "%s %.2e" % (what, err)


def test_odd_channel_count_and_no_gradient_for_constants():
    import mde_amd.layers as L
    z = load("feat_12x40")
    t = leaves(z)
    feats = t["feats"][:, :13].detach().contiguous().requires_grad_(True)      # 13 channels: a partial channel group
    out = L.warp_features(feats, t["disp"].detach(), t["K"], t["inv_K"], t["T"].detach())
    err = rel(out, z["out"][:, :13])
    assert err <= TOL, "%.2e" % err
    (out * t["g_out"][:, :13]).sum().backward()
    ref = torch.from_numpy(z["feats"][:, :13]).clone().requires_grad_(True)
    o, _ = O.warp_features(ref, torch.from_numpy(z["disp"]), torch.from_numpy(z["K"]), torch.from_numpy(z["inv_K"]),
                           torch.from_numpy(z["T"]))
    (o * torch.from_numpy(z["g_out"][:, :13])).sum().backward()
    err = rel(feats.grad, ref.grad)
    assert err <= TOL, "%.2e" % err


def test_warp_argument_errors():
    import mde_amd._abi as abi
    import mde_amd.layers as L
    f = torch.zeros(2, 4, 6, 8, device=DEV)
    d = torch.zeros(2, 1, 6, 8, device=DEV)
    M = torch.eye(4, device=DEV).repeat(2, 1, 1)
    with pytest.raises(ValueError):
        L.warp_features(f, torch.zeros(2, 1, 5, 8, device=DEV), M, M, M)
    with pytest.raises(ValueError):
        L.warp_features(f, d, M[:1], M, M)
    with pytest.raises(ValueError):
        L.warp_features(f.cpu(), d.cpu(), M.cpu(), M.cpu(), M.cpu())
    desc = abi.WarpDesc()
    desc.B, desc.C, desc.H, desc.W = 2, 4, 1, 8
    assert abi.lib.vsl_warp_features_fwd(ctypes.byref(desc), None, None, None) == -1


def test_affinity_term_matches_reference_fixture():
    """ViewSynthesisLoss(use_affinity=True).compute_losses against the reference's Trainer with use_affinity=True
    (Trainer.py:468-473,497-516,546-548,606): warped 25-channel maps, loss per scale, gradients of disparity and poses."""
    import json
    import types
    from mde_amd.trainer_loss import ViewSynthesisLoss
    z = np.load(os.path.join(GOLDEN_DIR, "affinity", "b2_24x80_affinity.npz"), allow_pickle=False)
    c = json.loads(str(z["config"]))
    fids = c["frame_ids"]
    opt = types.SimpleNamespace(scales=c["scales"], frame_ids=fids, height=c["H"], width=c["W"], min_depth=0.1,
                                max_depth=100.0, v1_multiscale=False, disable_automasking=False, avg_reprojection=False,
                                no_ssim=False, disparity_smoothness=1e-3)
    vs = ViewSynthesisLoss(opt, use_affinity=True)
    inputs = {"K": torch.from_numpy(z["K"]).to(DEV), "inv_K": torch.from_numpy(z["inv_K"]).to(DEV),
              ("segmentation", 0): torch.from_numpy(z["segmentation"]).to(DEV)}
    for f in fids:
        inputs[("image", f)] = torch.from_numpy(z["image_%s" % f]).to(DEV)
        inputs[("affinity", f)] = torch.from_numpy(z["affinity_%s" % f]).to(DEV)
    disp = torch.from_numpy(z["disp"]).to(DEV).requires_grad_(True)
    outputs = {("disp", 0): disp}
    poses = {}
    for f in fids[1:]:
        poses[f] = torch.from_numpy(z["T_%s" % f]).to(DEV).requires_grad_(True)
        outputs[("cam_T_cam", f)] = poses[f]
    losses = vs.compute_losses(inputs, outputs, noise=torch.from_numpy(z["noise"]).to(DEV),
                               drop_rand=torch.from_numpy(z["drop_rand"]).to(DEV))
    losses["loss"].backward()
    assert abs(losses["loss"].item() - float(z["loss"])) <= TOL * abs(float(z["loss"]))
    for s_ in c["scales"]:
        assert abs(losses["loss/%d" % s_].item() - float(z["loss_%d" % s_])) <= TOL * abs(float(z["loss_%d" % s_]))
    flips = int((outputs["identity_selection"].cpu().numpy() != z["identity_selection"]).sum())
    assert flips == 0
    for f in fids[1:]:
        e = rel(outputs[("pred_affinity", f)], z["pred_affinity_%s" % f])
        assert e <= TOL, "pred_affinity %s %.2e" % (f, e)
        e = rel(poses[f].grad, z["grad_T_%s" % f])
        assert e <= TOL, "grad_T %s %.2e" % (f, e)
    e = rel(disp.grad, z["grad_disp"])
    assert e <= TOL, "grad_disp %.2e" % e
    # validation mode skips the term (Trainer.py:468,546)
    vs.val_mode = True
    plain = vs.compute_losses(inputs, {("disp", 0): disp.detach(), **{("cam_T_cam", f): poses[f].detach() for f in fids[1:]}},
                              noise=torch.from_numpy(z["noise"]).to(DEV))
    assert plain["loss"].item() < losses["loss"].item()


@pytest.mark.gpu
def test_affinity_huber_variant_matches_reference_fixture():
    """ViewSynthesisLoss.compute_affinity_huber_loss against Trainer_distilation.py:1076-1110 run on the CPU in
    float64: value and gradient within 1e-5 relative."""
    import types
    from mde_amd.trainer_loss import ViewSynthesisLoss
    z = np.load(os.path.join(GOLDEN_DIR, "affinity", "huber_b2_24x80.npz"), allow_pickle=False)
    opt = types.SimpleNamespace(scales=[0], frame_ids=[0, -1, 1], height=24, width=80, min_depth=0.1, max_depth=100.0,
                                disable_automasking=False, avg_reprojection=False, no_ssim=False,
                                disparity_smoothness=1e-3, pose_model_type="shared")
    vs = ViewSynthesisLoss(opt, use_affinity=True)
    pred = torch.from_numpy(z["pred"]).to(DEV).requires_grad_(True)
    loss = vs.compute_affinity_huber_loss(pred, None, torch.from_numpy(z["target"]).to(DEV), [1.5, 0.5])
    loss.backward()
    assert abs(float(loss) - float(z["loss"])) <= TOL * abs(float(z["loss"]))
    e = rel(pred.grad, z["grad_pred"])
    assert e <= TOL, "grad %.2e" % e


def test_colour_and_affinity_in_one_launch_match_fixture_and_separate_calls():
    """generate_images_pred with the affinity term on (Python/Trainer.py:467-473): the colour frame (3 channels) and
    the affinity map (25 channels) of a source frame go through ONE launch that shares the sampling coordinates;
    values against the reference fixture and, bit for bit, against one single-tensor launch per map; gradients of
    a loss over both outputs against the sum of the separate launches' gradients."""
    import json
    import types
    from mde_amd.layers import warp_features, warp_features_multi
    from mde_amd.trainer_loss import ViewSynthesisLoss
    z = np.load(os.path.join(GOLDEN_DIR, "affinity", "b2_24x80_affinity.npz"), allow_pickle=False)
    c = json.loads(str(z["config"]))
    fids = c["frame_ids"]
    t = lambda k: torch.from_numpy(z[k]).to(DEV)
    K, inv_K = t("K"), t("inv_K")
    f = fids[1]
    img, aff = t("image_%s" % f).float(), t("affinity_%s" % f)[:, 0].float()
    disp = t("disp").requires_grad_(True)
    T = t("T_%s" % f).requires_grad_(True)
    grid, (color, paff) = warp_features_multi([img, aff], disp, K, inv_K, T)
    assert rel(paff, z["pred_affinity_%s" % f]) <= TOL
    w1 = torch.linspace(0.5, 1.5, color.numel(), device=DEV).view_as(color)
    w2 = torch.linspace(-1.0, 1.0, paff.numel(), device=DEV).view_as(paff)
    ((color * w1).sum() + (paff * w2).sum()).backward()
    g_disp, g_T = disp.grad.clone(), T.grad.clone()
    disp.grad = T.grad = None
    color1, grid1 = warp_features(img, disp, K, inv_K, T, return_sample=True)
    paff1 = warp_features(aff, disp, K, inv_K, T)
    assert torch.equal(grid, grid1) and torch.equal(color, color1) and torch.equal(paff, paff1)
    ((color1 * w1).sum() + (paff1 * w2).sum()).backward()
    # (one chain through the projection for the summed tap derivatives against two chains summed afterwards: float32 order)
    assert rel(g_disp, disp.grad.cpu().numpy()) <= TOL and rel(g_T, T.grad.cpu().numpy()) <= TOL
    # only one of the two outputs carries a gradient: the other tensor's groups contribute nothing
    disp.grad = T.grad = None
    _, (color2, paff2) = warp_features_multi([img, aff], disp, K, inv_K, T)
    (paff2 * w2).sum().backward()
    ga = disp.grad.clone()
    disp.grad = T.grad = None
    (warp_features(aff, disp, K, inv_K, T) * w2).sum().backward()
    assert torch.equal(ga, disp.grad)
    # the trainer surface: ("color",f), ("sample",f), ("pred_affinity",f) of every source from one launch each
    opt = types.SimpleNamespace(scales=c["scales"], frame_ids=fids, height=c["H"], width=c["W"], min_depth=0.1,
                                max_depth=100.0, v1_multiscale=False, disable_automasking=False, avg_reprojection=False,
                                no_ssim=False, disparity_smoothness=1e-3)
    vs = ViewSynthesisLoss(opt, use_affinity=True)
    inputs = {"K": K, "inv_K": inv_K}
    outputs = {("disp", 0): t("disp")}
    for g in fids:
        inputs[("image", g)], inputs[("affinity", g)] = t("image_%s" % g), t("affinity_%s" % g)
    for g in fids[1:]:
        outputs[("cam_T_cam", g)] = t("T_%s" % g)
    vs.generate_images_pred(inputs, outputs)
    for g in fids[1:]:
        assert rel(outputs[("pred_affinity", g)], z["pred_affinity_%s" % g]) <= TOL
        assert tuple(outputs[("color", g)].shape) == tuple(inputs[("image", g)].shape)


def test_warp_multi_bwd_with_a_null_upstream_gradient_entry():
    """vsl_warp_multi_bwd through the C-ABI with a NULL entry in its g_out table ("this tensor has no upstream
    gradient"): same disparity and pose gradients, bit for bit, as the call that passes a tensor of zeros there, and the
    skipped tensor's own gradient comes back as zeros."""
    import ctypes
    import mde_amd._abi as abi
    import mde_amd.synth as synth
    B, H, W = 2, 24, 80
    data = synth.make_batch(B, H, W, [0, -1], seed=5)
    t = {k: v.to(DEV) for k, v in data.items() if torch.is_tensor(v)}
    feats = [t[("image", -1)], torch.rand(B, 25, H, W, device=DEV)]
    d = abi.WarpMultiDesc()
    d.B, d.H, d.W, d.n = B, H, W, 2
    d.min_depth, d.max_depth, d.eps = 0.1, 100.0, 1e-7
    for i, f in enumerate(feats):
        d.C[i], d.feats[i] = f.shape[1], f.data_ptr()
    d.disp, d.K, d.inv_K, d.T = t[("disp", 0)].data_ptr(), t["K"].data_ptr(), t["inv_K"].data_ptr(), t[("cam_T_cam", -1)].data_ptr()
    nbytes = abi.lib.vsl_warp_multi_workspace_bytes(ctypes.byref(d))
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=DEV)
    g1 = torch.randn(B, 25, H, W, device=DEV)
    stream = torch.cuda.current_stream().cuda_stream

    def run(g0):
        go, gf = (ctypes.c_void_p * abi.MAX_WARP_TENSORS)(), (ctypes.c_void_p * abi.MAX_WARP_TENSORS)()
        go[0], go[1] = (g0.data_ptr() if g0 is not None else None), g1.data_ptr()
        gfe = [torch.full_like(feats[0], 7.0), torch.zeros_like(feats[1])]
        gf[0], gf[1] = gfe[0].data_ptr(), gfe[1].data_ptr()
        g_disp, g_T = torch.empty(B, 1, H, W, device=DEV), torch.empty(B, 4, 4, device=DEV)
        abi.check(abi.lib.vsl_warp_multi_bwd(ctypes.byref(d), go, gf, g_disp.data_ptr(), g_T.data_ptr(), ws.data_ptr(),
                                             ws.numel() * 8, stream))
        torch.cuda.synchronize()
        return g_disp, g_T, gfe

    gd_null, gT_null, gf_null = run(None)
    gd_zero, gT_zero, _ = run(torch.zeros_like(feats[0]))
    assert torch.equal(gd_null, gd_zero) and torch.equal(gT_null, gT_zero)
    assert float(gd_null.abs().max()) > 0
    assert float(gf_null[0].abs().max()) == 0.0 and float(gf_null[1].abs().max()) > 0      # no upstream gradient: zeros


def test_affinity_loss_kernel_matches_torch_ops_and_fixture():
    """vsl_affinity_loss_fwd / _bwd against the reference's op sequence (Python/Trainer.py:497-516) on the fixture's
    warped map, target map and pinned torch.rand_like tensor: value and gradient with respect to the prediction."""
    from mde_amd.layers import affinity_reprojection_loss
    z = np.load(os.path.join(GOLDEN_DIR, "affinity", "b2_24x80_affinity.npz"), allow_pickle=False)
    pred = torch.from_numpy(z["pred_affinity_-1"]).float().to(DEV).requires_grad_(True)
    target = torch.from_numpy(z["affinity_0"])[:, 0].float().to(DEV)
    rand = torch.from_numpy(z["drop_rand"]).float().to(DEV)
    loss = affinity_reprojection_loss(pred, target, rand, 0.5)
    (3.0 * loss).backward()
    ref_pred = pred.detach().double().requires_grad_(True)
    keep = (rand.double() < 0.5).double()
    drop = torch.where(keep == 1.0, ref_pred, torch.ones_like(ref_pred))
    ref = torch.mean((ref_pred - target.double() * drop) ** 2)
    (3.0 * ref).backward()
    assert abs(loss.item() - ref.item()) <= 1e-6 * abs(ref.item())
    assert rel(pred.grad, ref_pred.grad.cpu().numpy()) <= 1e-6
    again = affinity_reprojection_loss(pred.detach(), target, rand, 0.5)
    assert again.item() == loss.item()                       # deterministic
