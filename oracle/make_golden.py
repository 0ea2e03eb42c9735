"""Generate the golden fixtures under tests/golden from the REFERENCE ITSELF.

Runs only in the development container (it needs /root/reference); the GPU box
never sees the reference, only the .npz files this writes.  The reference's
``layers.py``, ``Trainer.py`` and ``Trainer_distilation.py`` are loaded by file
path (never via sys.path: the reference ships a ``collections.py`` that would
shadow the standard library), third-party imports that are not installed are
replaced by empty modules, ``Tensor.cuda`` is neutralised because this container
has no GPU, and ``torch.randn`` is pinned to the recorded tie-break noise while
``compute_losses`` runs (Trainer.py:589 draws it from the CPU generator).

    python oracle/make_golden.py               # rewrites tests/golden/*.npz
    python oracle/make_golden.py --pose-only   # only the pose-preparation fixtures
    python oracle/make_golden.py --warp-only   # only the feature-warping fixtures
    python oracle/make_golden.py --affinity-only
    python oracle/make_golden.py --metrics-only

Every file holds: the inputs (fp32 values; the reference is fed their float64
copies, as its data loader would produce), every intermediate the reference
keeps in ``outputs``, the per-source photometric error maps, the loss values
and the gradients with respect to the disparity and each pose matrix.
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/Python"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import mde_amd.synth as synth  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    torch.Tensor.cuda = lambda self, *a, **k: self
    layers = _load("layers", os.path.join(REF, "layers.py"))
    for n in ["networks", "maskrcnn_benchmark", "maskrcnn_benchmark.config", "cv2", "eval_utils"]:
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules["maskrcnn_benchmark.config"].cfg = None
    sys.modules["eval_utils"].evaluate_depth = None
    trainer = _load("ref_trainer", os.path.join(REF, "Trainer.py"))
    distil = _load("ref_trainer_distil", os.path.join(REF, "Trainer_distilation.py"))
    return layers, trainer, distil


def make_trainer(cls, layers, B, H, W, c):
    tr = object.__new__(cls)
    tr.opt = types.SimpleNamespace(
        scales=list(c["scales"]), frame_ids=list(c["frame_ids"]), v1_multiscale=False, height=H, width=W,
        min_depth=0.1, max_depth=100.0, pose_model_type=c.get("pose_model_type", "shared"),
        disable_automasking=not c["automask"], predictive_mask=False,
        avg_reprojection=c["avg_reprojection"], no_ssim=c["no_ssim"], disparity_smoothness=1e-3)
    tr.num_scales = 1
    tr.use_affinity = False
    tr.val_mode = False
    tr.epoch = 0
    tr.backproject_depth = layers.BackprojectDepth(B, H, W)
    tr.project_3d = layers.Project3D(B, H, W)
    tr.ssim = layers.SSIM()
    return tr


CASES = [
    # name, B, H, W, frame_ids, flags
    dict(name="b2_24x80_automask", B=2, H=24, W=80, frame_ids=[0, -1, 1], automask=True),
    dict(name="b2_24x80_noautomask", B=2, H=24, W=80, frame_ids=[0, -1, 1], automask=False),
    dict(name="b1_13x37_automask", B=1, H=13, W=37, frame_ids=[0, -1, 1], automask=True, seed=5),
    dict(name="b1_13x37_noautomask_avg", B=1, H=13, W=37, frame_ids=[0, -1, 1], automask=False,
         avg_reprojection=True, seed=6),
    dict(name="b2_24x80_automask_avg", B=2, H=24, W=80, frame_ids=[0, -1, 1], automask=True,
         avg_reprojection=True, seed=7),
    dict(name="b2_24x80_automask_nossim", B=2, H=24, W=80, frame_ids=[0, -1, 1], automask=True,
         no_ssim=True, seed=8),
    dict(name="b2_24x80_stereo", B=2, H=24, W=80, frame_ids=[0, -1, 1, "s"], automask=True, seed=9),
    dict(name="b1_20x50_single_source", B=1, H=20, W=50, frame_ids=[0, 1], automask=False, seed=10),
    dict(name="b1_20x50_single_source_automask", B=1, H=20, W=50, frame_ids=[0, 1], automask=True, seed=11),
    dict(name="b2_24x80_scales4", B=2, H=24, W=80, frame_ids=[0, -1, 1], automask=True,
         scales=[0, 1, 2, 3], seed=12),
    dict(name="b2_24x80_large_motion", B=2, H=24, W=80, frame_ids=[0, -1, 1], automask=True,
         pose_sigma=0.25, seed=13),
    dict(name="b2_48x160_automask", B=2, H=48, W=160, frame_ids=[0, -1, 1], automask=True, seed=14),
    dict(name="b3_16x33_sharp", B=3, H=16, W=33, frame_ids=[0, -1, 1], automask=True, seed=15, blur=1),
    dict(name="b2_24x80_distil_scale09", B=2, H=24, W=80, frame_ids=[0, -1, 1], automask=True,
         variant="distil", seed=16),
    dict(name="b2_24x80_distil_single", B=2, H=24, W=80, frame_ids=[0, -1, 1], automask=True,
         variant="distil_single", seed=17),
]


def run_case(c, layers, trainer, distil, out_dir):
    c = dict(dict(avg_reprojection=False, no_ssim=False, scales=[0], seed=0, pose_sigma=0.01,
                  variant="trainer", blur=5), **c)
    B, H, W = c["B"], c["H"], c["W"]
    fids = c["frame_ids"]
    srcs = fids[1:]
    data = synth.make_batch(B, H, W, fids, seed=c["seed"], pose_sigma=c["pose_sigma"], blur=c["blur"])
    n_id = 1 if c["avg_reprojection"] else len(srcs)
    noise = data["noise"][:, :n_id].contiguous()

    cls = trainer.Trainer if c["variant"] == "trainer" else distil.Trainer
    tr = make_trainer(cls, layers, B, H, W, c)

    inputs = {"K": data["K"], "inv_K": data["inv_K"]}
    for f in fids:
        inputs[("image", f)] = data[("image", f)].double()
    disp = data[("disp", 0)].clone().requires_grad_(True)
    outputs = {("disp", 0): disp}
    poses = {}
    for f in srcs:
        if f == "s":
            poses[f] = data["stereo_T"].clone().requires_grad_(True)
            inputs["stereo_T"] = poses[f]
        else:
            poses[f] = data[("cam_T_cam", f)].clone().requires_grad_(True)
            outputs[("cam_T_cam", f)] = poses[f]

    real_randn = torch.randn
    torch.randn = lambda *a, **k: noise.clone()
    try:
        if c["variant"] == "trainer":
            tr.generate_images_pred(inputs, outputs)
            losses = tr.compute_losses(inputs, outputs)
        elif c["variant"] == "distil":
            teacher = {("disp", 0): data[("disp", 0)].clone()}
            tr.generate_images_pred(inputs, outputs)
            losses = tr.compute_losses(inputs, outputs, teacher, None, None, None)
        else:
            tr.generate_images_pred_single(inputs, outputs)
            losses = tr.compute_losses_single(inputs, outputs)
    finally:
        torch.randn = real_randn
    losses["loss"].backward()

    rec = {"config": np.array(json.dumps({k: v for k, v in c.items()}))}
    for f in fids:
        rec["image_%s" % f] = data[("image", f)].numpy()
    rec["disp"] = data[("disp", 0)].numpy()
    rec["K"] = data["K"].numpy()
    rec["inv_K"] = data["inv_K"].numpy()
    rec["noise"] = noise.numpy()
    rec["depth"] = outputs[("depth", 0)].detach().numpy()
    target = inputs[("image", 0)]
    with torch.no_grad():
        pts = tr.backproject_depth(outputs[("depth", 0)], inputs["inv_K"])
        rec["points"] = pts.numpy()
    for f in srcs:
        rec["T_%s" % f] = poses[f].detach().numpy()
        rec["sample_%s" % f] = outputs[("sample", f)].detach().numpy()
        rec["color_%s" % f] = outputs[("color", f)].detach().numpy()
        with torch.no_grad():
            rec["rp_%s" % f] = tr.compute_reprojection_loss(outputs[("color", f)], target).numpy()
            rec["ssim_%s" % f] = tr.ssim(outputs[("color", f)], target).numpy()
            rec["rp_identity_%s" % f] = tr.compute_reprojection_loss(inputs[("image", f)], target).numpy()
        rec["grad_T_%s" % f] = poses[f].grad.numpy()
    if c["automask"]:
        rec["identity_selection"] = outputs["identity_selection"].numpy()
    with torch.no_grad():
        md = disp.mean(2, True).mean(3, True)
        rec["smooth"] = layers.get_smooth_loss(disp / (md + 1e-7), target).numpy()
    rec["loss"] = losses["loss"].detach().numpy()
    for s in c["scales"]:
        rec["loss_%d" % s] = losses["loss/%d" % s].detach().numpy()
    rec["grad_disp"] = disp.grad.numpy()
    path = os.path.join(out_dir, c["name"] + ".npz")
    np.savez_compressed(path, **rec)
    return path, float(rec["loss"]), float(np.abs(rec["grad_disp"]).sum())


def run_pose_cases(layers, trainer, out_dir):
    """Pose preparation: layers.transformation_from_parameters on slices of a [B,F,1,3] network output, both
    invert settings, with gradients for a fixed upstream g_T; and one loss case in the "posecnn" mode, where
    Trainer.generate_images_pred rebuilds T from the parameters and the mean inverse depth (Trainer.py:446-455)."""
    B, F = 6, 2
    scale = torch.tensor([0.01, 0.01, 0.3, 0.3, 2.0, 1.0]).view(B, 1, 1, 1)
    aa0 = torch.from_numpy(synth.normal(21, (B, F, 1, 3)).astype(np.float32)) * scale
    tr0 = torch.from_numpy(synth.normal(22, (B, F, 1, 3)).astype(np.float32)) * scale
    aa0[5, 0] = 0.0                                   # zero rotation: angle = 0, axis = 0 / 1e-7
    g_T = torch.from_numpy(synth.normal(23, (F, 2, B, 4, 4)).astype(np.float32))
    rec = {"axisangle": aa0.numpy(), "translation": tr0.numpy(), "g_T": g_T.numpy()}
    for i in range(F):
        for inv in (0, 1):
            aa, tr = aa0.clone().requires_grad_(True), tr0.clone().requires_grad_(True)
            T = layers.transformation_from_parameters(aa[:, i], tr[:, i], invert=bool(inv))
            (T * g_T[i, inv]).sum().backward()
            rec["T_%d_%d" % (i, inv)] = T.detach().numpy()
            rec["g_axisangle_%d_%d" % (i, inv)] = aa.grad.numpy()
            rec["g_translation_%d_%d" % (i, inv)] = tr.grad.numpy()
    os.makedirs(os.path.join(out_dir, "pose"), exist_ok=True)
    np.savez_compressed(os.path.join(out_dir, "pose", "pose_params.npz"), **rec)

    # The seed is the first one from 18 on for which no pixel's two smallest candidates of the per-pixel minimum
    # are closer than 1e-6: the device evaluates sin / cos and the mean inverse depth with other last bits than
    # this host, and a closer tie could legitimately select the other candidate there.
    for seed in range(18, 60):
        c = dict(name="b2_24x80_posecnn", B=2, H=24, W=80, frame_ids=[0, -1, 1], automask=True, seed=seed,
                 avg_reprojection=False, no_ssim=False, scales=[0], pose_model_type="posecnn")
        Bc, H, W, fids = c["B"], c["H"], c["W"], c["frame_ids"]
        data = synth.make_batch(Bc, H, W, fids, seed=c["seed"])
        noise = data["noise"].contiguous()
        tr_ = make_trainer(trainer.Trainer, layers, Bc, H, W, c)
        inputs = {"K": data["K"], "inv_K": data["inv_K"]}
        for f in fids:
            inputs[("image", f)] = data[("image", f)].double()
        disp = data[("disp", 0)].clone().requires_grad_(True)
        outputs = {("disp", 0): disp}
        params = {}
        for j, f in enumerate(fids[1:]):
            aa = (torch.from_numpy(synth.normal(30 + j, (Bc, 1, 1, 3)).astype(np.float32)) * 0.01).requires_grad_(True)
            tv = (torch.from_numpy(synth.normal(40 + j, (Bc, 1, 1, 3)).astype(np.float32)) * 0.01).requires_grad_(True)
            params[f] = (aa, tv)
            outputs[("axisangle", f)], outputs[("translation", f)] = aa, tv
            outputs[("cam_T_cam", f)] = layers.transformation_from_parameters(aa[:, 0], tv[:, 0], invert=(f < 0))
        real_randn = torch.randn
        torch.randn = lambda *a, **k: noise.clone()
        try:
            tr_.generate_images_pred(inputs, outputs)
            losses = tr_.compute_losses(inputs, outputs)
        finally:
            torch.randn = real_randn
        with torch.no_grad():
            tgt = inputs[("image", 0)]
            cand = [tr_.compute_reprojection_loss(inputs[("image", f)], tgt) for f in fids[1:]]
            cand = torch.cat(cand, 1) + noise.double() * 0.00001
            cand = torch.cat([cand] + [tr_.compute_reprojection_loss(outputs[("color", f)], tgt) for f in fids[1:]], 1)
            two = torch.topk(cand, 2, dim=1, largest=False).values
            margin = float((two[:, 1] - two[:, 0]).min())
        if margin >= 1e-6:
            break
    else:
        raise RuntimeError("no seed without near-ties found")
    losses["loss"].backward()
    rec = {"config": np.array(json.dumps(c)), "disp": data[("disp", 0)].numpy(), "K": data["K"].numpy(),
           "inv_K": data["inv_K"].numpy(), "noise": noise.numpy(), "loss": losses["loss"].detach().numpy(),
           "grad_disp": disp.grad.numpy(), "identity_selection": outputs["identity_selection"].numpy()}
    with torch.no_grad():
        rec["mean_inv_depth"] = (1 / outputs[("depth", 0)]).mean(3, True).mean(2, True).numpy()
    for f in fids:
        rec["image_%s" % f] = data[("image", f)].numpy()
    for f in fids[1:]:
        rec["axisangle_%s" % f] = params[f][0].detach().numpy()
        rec["translation_%s" % f] = params[f][1].detach().numpy()
        rec["g_axisangle_%s" % f] = params[f][0].grad.numpy()
        rec["g_translation_%s" % f] = params[f][1].grad.numpy()
        rec["sample_%s" % f] = outputs[("sample", f)].detach().numpy()
    np.savez_compressed(os.path.join(out_dir, "pose", c["name"] + ".npz"), **rec)
    print("pose/pose_params.npz, %s.npz  seed=%d margin=%.2e loss=%.10f" % (c["name"], seed, margin, float(rec["loss"])))


def run_affinity_case(layers, trainer, out_dir):
    """Trainer.generate_images_pred + compute_losses with use_affinity=True (Trainer.py:468-473,497-516,546-548,606):
    a 25-channel map per frame warped with the colour grid, squared error against the target frame's map with the
    random drop-out of :502-506 (torch.rand_like pinned to a recorded tensor)."""
    c = dict(name="b2_24x80_affinity", B=2, H=24, W=80, frame_ids=[0, -1, 1], automask=True, seed=70,
             avg_reprojection=False, no_ssim=False, scales=[0, 1])
    B, H, W, fids = c["B"], c["H"], c["W"], c["frame_ids"]
    data = synth.make_batch(B, H, W, fids, seed=c["seed"])
    noise = data["noise"].contiguous()
    tr = make_trainer(trainer.Trainer, layers, B, H, W, c)
    tr.use_affinity = True
    inputs = {"K": data["K"], "inv_K": data["inv_K"]}
    rec = {}
    for j, f in enumerate(fids):
        inputs[("image", f)] = data[("image", f)].double()
        a = torch.from_numpy(synth.uniform(80 + j, (B, 1, 25, H, W)).astype(np.float32))
        a = torch.nn.functional.avg_pool3d(torch.nn.functional.pad(a, (1, 1, 1, 1, 0, 0), mode="replicate"), (1, 3, 3), 1)
        inputs[("affinity", f)] = a.contiguous()
        rec["affinity_%s" % f] = inputs[("affinity", f)].numpy()
    inputs[("segmentation", 0)] = (torch.from_numpy(synth.uniform(90, (B, 4, H, W)).astype(np.float32)) > 0.5).float()
    rec["segmentation"] = inputs[("segmentation", 0)].numpy()
    drop_rand = torch.from_numpy(synth.uniform(91, (B, 25, H, W)).astype(np.float32))
    disp = data[("disp", 0)].clone().requires_grad_(True)
    outputs = {("disp", 0): disp}
    poses = {}
    for f in fids[1:]:
        poses[f] = data[("cam_T_cam", f)].clone().requires_grad_(True)
        outputs[("cam_T_cam", f)] = poses[f]
    real_randn, real_rand_like = torch.randn, torch.rand_like
    torch.randn = lambda *a, **k: noise.clone()
    torch.rand_like = lambda t, **k: drop_rand.clone().to(t.dtype)
    try:
        tr.generate_images_pred(inputs, outputs)
        losses = tr.compute_losses(inputs, outputs)
    finally:
        torch.randn, torch.rand_like = real_randn, real_rand_like
    losses["loss"].backward()
    rec.update({"config": np.array(json.dumps(c)), "disp": data[("disp", 0)].numpy(), "K": data["K"].numpy(),
                "inv_K": data["inv_K"].numpy(), "noise": noise.numpy(), "drop_rand": drop_rand.numpy(),
                "loss": losses["loss"].detach().numpy(), "grad_disp": disp.grad.numpy(),
                "identity_selection": outputs["identity_selection"].numpy()})
    for s_ in c["scales"]:
        rec["loss_%d" % s_] = losses["loss/%d" % s_].detach().numpy()
    for f in fids:
        rec["image_%s" % f] = data[("image", f)].numpy()
    for f in fids[1:]:
        rec["T_%s" % f] = poses[f].detach().numpy()
        rec["grad_T_%s" % f] = poses[f].grad.numpy()
        rec["pred_affinity_%s" % f] = outputs[("pred_affinity", f)].detach().numpy().astype(np.float32)
    os.makedirs(os.path.join(out_dir, "affinity"), exist_ok=True)
    np.savez_compressed(os.path.join(out_dir, "affinity", c["name"] + ".npz"), **rec)
    print("affinity/%s.npz  loss=%.10f" % (c["name"], float(rec["loss"])))


def run_affinity_huber_case(distil, out_dir):
    """Trainer_distilation.Trainer.compute_affinity_reprojection_loss (:1076-1110) called directly (no caller exists
    in that trainer): a float64 prediction as grid_sample of a float64 map returns it, a near-binary float32 target
    whose channel sums straddle the threshold 24 of :1084."""
    B, H, W = 2, 24, 80
    box = lambda a: torch.nn.functional.avg_pool3d(torch.nn.functional.pad(a, (1, 1, 1, 1, 0, 0), mode="replicate"), (1, 3, 3), 1)[:, 0]
    pred = box(torch.from_numpy(synth.uniform(120, (B, 1, 25, H, W)).astype(np.float32)))
    pred = (pred * 3.0 - 1.0).double().clone().requires_grad_(True)     # |pred - target| crosses delta = 1
    target = (box(torch.from_numpy(synth.uniform(121, (B, 1, 25, H, W)).astype(np.float32))) > 0.36).float().contiguous()
    tr = object.__new__(distil.Trainer)
    loss = tr.compute_affinity_reprojection_loss(pred, None, target, [1.5, 0.5])
    loss.backward()
    frac = float((target.sum(1) < 24).float().mean())
    lin = float(((pred.detach() - target).abs() >= 1).float().mean())
    os.makedirs(os.path.join(out_dir, "affinity"), exist_ok=True)
    np.savez_compressed(os.path.join(out_dir, "affinity", "huber_b2_24x80.npz"), pred=pred.detach().numpy().astype(np.float32),
                        target=target.numpy(), loss=loss.detach().numpy(), grad_pred=pred.grad.numpy())
    print("affinity/huber_b2_24x80.npz  loss=%.10f  pixels below the threshold %.1f %%  linear branch %.1f %%" % (
        float(loss), 100 * frac, 100 * lin))


def run_depth_metric_cases(layers, trainer, out_dir):
    """Trainer.compute_depth_losses (Trainer.py:622-662) on a predicted depth and a sparse 375 x 1242 ground truth
    (LiDAR-like: a few percent of the pixels carry a value)."""
    os.makedirs(os.path.join(out_dir, "metrics"), exist_ok=True)
    for name, B, H, W, density, seed in (("b2_48x160", 2, 48, 160, 0.05, 110), ("b3_24x80_even_count", 3, 24, 80, 0.02, 111)):
        tr = object.__new__(trainer.Trainer)
        disp = synth.make_batch(B, H, W, [0], seed=seed)[("disp", 0)]
        _, depth = layers.disp_to_depth(disp, 0.1, 100.0)
        gt = 1.0 + 60.0 * torch.from_numpy(synth.uniform(seed + 1, (B, 1, 375, 1242)).astype(np.float32)) ** 2
        keep = torch.from_numpy(synth.uniform(seed + 2, (B, 1, 375, 1242)).astype(np.float32)) < density
        gt = gt * keep
        if name.endswith("even_count"):   # make the masked count even: torch.median then takes the lower middle element
            m = (gt > 0)
            m[:, :, :153] = False
            m[:, :, 371:] = False
            m[:, :, :, :44] = False
            m[:, :, :, 1197:] = False
            if int(m.sum()) % 2 == 1:
                idx = m.nonzero()[0]
                gt[tuple(idx)] = 0.0
        total, metrics = tr.compute_depth_losses({"depth_gt": gt.clone()}, {("depth", 0): depth.clone()}, {})
        gmask = (gt > 0)
        np.savez_compressed(os.path.join(out_dir, "metrics", name + ".npz"), depth=depth.numpy(), depth_gt=gt.numpy(),
                            metrics=np.array([float(v) for v in metrics], np.float64), total=float(total))
        print("metrics/%s.npz  %s" % (name, " ".join("%.6f" % float(v) for v in metrics)))


def run_feature_warp_cases(layers, out_dir):
    """Coarse-scale feature warping as Train_distillation_new.py:1159-1182 spells it out (the trainer module itself
    pulls in networks that are not installed, so the lines are replayed here on the reference's own layers):
    F.interpolate(disp) -> disp_to_depth -> BackprojectDepth(B,h,w) -> Project3D(B,h,w) -> F.grid_sample(feats, grid,
    padding_mode="border"), float32 throughout, with the full-resolution K / inv_K at every size."""
    import torch.nn.functional as F
    os.makedirs(os.path.join(out_dir, "warp"), exist_ok=True)
    B, Hf, Wf, C = 2, 24, 80, 20
    for name, (h, w), sigma, seed in (("feat_12x40", (12, 40), 0.01, 50), ("feat_6x20", (6, 20), 0.01, 51),
                                      ("feat_24x80", (24, 80), 0.01, 52), ("feat_13x37_large_motion", (13, 37), 0.2, 53)):
        data = synth.make_batch(B, Hf, Wf, [0, -1], seed=seed, pose_sigma=sigma)
        feats = torch.from_numpy(synth.normal(seed + 100, (B, C, h, w)).astype(np.float32))
        feats = F.avg_pool2d(F.pad(feats, (1, 1, 1, 1), mode="replicate"), 3, 1).clone().requires_grad_(True)
        disp = data[("disp", 0)]
        disp_k = F.interpolate(disp, [h, w], mode="bilinear", align_corners=False).clone().requires_grad_(True)
        T = data[("cam_T_cam", -1)].clone().requires_grad_(True)
        _, depth_k = layers.disp_to_depth(disp_k, 0.1, 100.0)
        cam = layers.BackprojectDepth(B, h, w)(depth_k, data["inv_K"])
        pix = layers.Project3D(B, h, w)(cam, data["K"], T)
        out = F.grid_sample(feats, pix, padding_mode="border")
        g_out = torch.from_numpy(synth.normal(seed + 200, (B, C, h, w)).astype(np.float32))
        (out * g_out).sum().backward()
        np.savez_compressed(os.path.join(out_dir, "warp", name + ".npz"), feats=feats.detach().numpy(),
                            disp=disp_k.detach().numpy(), K=data["K"].numpy(), inv_K=data["inv_K"].numpy(),
                            T=T.detach().numpy(), sample=pix.detach().numpy(), out=out.detach().numpy(), g_out=g_out.numpy(),
                            g_feats=feats.grad.numpy(), g_disp=disp_k.grad.numpy(), g_T=T.grad.numpy())
        print("warp/%s.npz  |out|=%.6f  out-of-range samples %.1f %%" % (
            name, float(out.abs().mean()), 100 * float((pix.abs() > 1).any(-1).float().mean())))


def main():
    torch.set_num_threads(1)   # deterministic summation order on the CPU
    layers, trainer, distil = load_reference()
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    if "--affinity-only" in sys.argv:
        run_affinity_case(layers, trainer, out_dir)
        run_affinity_huber_case(distil, out_dir)
        return
    if "--metrics-only" in sys.argv:
        run_depth_metric_cases(layers, trainer, out_dir)
        return
    if "--warp-only" in sys.argv:
        run_feature_warp_cases(layers, out_dir)
        return
    run_pose_cases(layers, trainer, out_dir)
    if "--pose-only" in sys.argv:
        return
    run_feature_warp_cases(layers, out_dir)
    run_affinity_case(layers, trainer, out_dir)
    run_affinity_huber_case(distil, out_dir)
    run_depth_metric_cases(layers, trainer, out_dir)
    for c in CASES:
        path, loss, g = run_case(c, layers, trainer, distil, out_dir)
        print("%-40s loss=%.10f |grad_disp|_1=%.6e  %6.1f KB" % (
            os.path.basename(path), loss, g, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
