"""Per-module path (generate_images_pred + compute_losses_unfused + backward on the layers.py kernels) against the
fused call, one head at 640x192 batch 12 (GPU box)."""
import os, sys, types, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mde_amd.synth as synth
from mde_amd.trainer_loss import ViewSynthesisLoss

B, H, W, fids = 12, 192, 640, [0, -1, 1]
dev = "cuda:0"
data = synth.make_batch(B, H, W, fids, seed=1)
opt = types.SimpleNamespace(scales=[0], frame_ids=fids, height=H, width=W, min_depth=0.1, max_depth=100.0, v1_multiscale=False,
                            disable_automasking=False, avg_reprojection=False, no_ssim=False, disparity_smoothness=1e-3)
inputs = {"K": data["K"].to(dev), "inv_K": data["inv_K"].to(dev)}
for f in fids:
    inputs[("image", f)] = data[("image", f)].to(dev)
noise = data["noise"].to(dev)
vs = ViewSynthesisLoss(opt)


def step(fused):
    disp = data[("disp", 0)].to(dev).requires_grad_(True)
    outputs = {("disp", 0): disp}
    for f in fids[1:]:
        outputs[("cam_T_cam", f)] = data[("cam_T_cam", f)].to(dev).requires_grad_(True)
    if fused:
        losses = vs.compute_losses(inputs, outputs, noise)
    else:
        vs.generate_images_pred(inputs, outputs)
        losses = vs.compute_losses_unfused(inputs, outputs, noise)
    losses["loss"].backward()


for fused in (True, False):
    for _ in range(3):
        step(fused)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        step(fused)
    torch.cuda.synchronize()
    print("%s path: %.2f ms per head (host-inclusive, autograd)" % ("fused" if fused else "per-module", (time.perf_counter() - t0) / n * 1e3))
