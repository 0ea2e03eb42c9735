"""GPU parity of the depth metrics (vsl_depth_metrics through the C-ABI) against fixtures from the reference's
Trainer.compute_depth_losses (Trainer.py:622-662) and against the CPU oracle.  Medians: bit-exact against the oracle
whenever the upsampled prediction agrees in its bits, else within 1e-6; metrics 1e-5 relative."""
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


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, "metrics", name + ".npz"), allow_pickle=False)


@pytest.mark.parametrize("name", ["b2_48x160", "b3_24x80_even_count"])
def test_depth_metrics_match_reference_fixture(name):
    import mde_amd._abi as abi
    import mde_amd.layers as L
    z = load(name)
    depth, gt = torch.from_numpy(z["depth"]).to(DEV), torch.from_numpy(z["depth_gt"]).to(DEV)
    m = L.depth_metrics(depth, gt).cpu().double().numpy()
    for i, ref in enumerate(z["metrics"]):
        assert abs(m[i] - ref) <= TOL * abs(ref), "%s: %.8g vs %.8g" % (abi.METRIC_NAMES[i], m[i], ref)
    assert abs(m[abi.METRIC_SUM] - float(z["total"])) <= TOL * abs(float(z["total"]))
    # count and medians against the oracle's masked arrays
    mask = torch.from_numpy(z["depth_gt"]) > 0
    crop = torch.zeros_like(mask)
    crop[:, :, 153:371, 44:1197] = True
    mask &= crop
    assert int(m[abi.METRIC_COUNT]) == int(mask.sum())
    if name.endswith("even_count"):
        assert int(mask.sum()) % 2 == 0
    gt_med = torch.median(torch.from_numpy(z["depth_gt"])[mask]).item()
    assert m[abi.METRIC_MEDIAN_GT] == gt_med                      # a selection: exact
    pred = torch.clamp(torch.nn.functional.interpolate(torch.from_numpy(z["depth"]), [375, 1242], mode="bilinear",
                                                       align_corners=False), 1e-3, 80)[mask]
    assert abs(m[abi.METRIC_MEDIAN_PRED] - torch.median(pred).item()) <= 1e-6 * torch.median(pred).item()


def test_trainer_surface_and_full_size():
    """compute_depth_losses at 640x192 b12 from the disparity alone (the fused loss does not materialise the depth),
    against the oracle; two calls give identical bits (integer atomics and fixed-order sums only)."""
    import types
    import mde_amd.synth as synth
    from mde_amd.trainer_loss import ViewSynthesisLoss
    B, H, W = 12, 192, 640
    opt = types.SimpleNamespace(scales=[0], frame_ids=[0, -1, 1], height=H, width=W, min_depth=0.1, max_depth=100.0,
                                v1_multiscale=False, disable_automasking=False, avg_reprojection=False, no_ssim=False,
                                disparity_smoothness=1e-3)
    vs = ViewSynthesisLoss(opt)
    disp = synth.make_batch(B, H, W, [0], seed=5)[("disp", 0)]
    g = torch.Generator().manual_seed(9)
    gt = (1.0 + 70.0 * torch.rand(B, 1, 375, 1242, generator=g) ** 2) * (torch.rand(B, 1, 375, 1242, generator=g) < 0.04)
    total, metrics = vs.compute_depth_losses({"depth_gt": gt.to(DEV)}, {("disp", 0): disp.to(DEV)}, {})
    total2, metrics2 = vs.compute_depth_losses({"depth_gt": gt.to(DEV)}, {("disp", 0): disp.to(DEV)}, {})
    assert torch.equal(torch.stack(metrics), torch.stack(metrics2)) and torch.equal(total, total2)
    ref_total, ref = O.depth_losses(O.disp_to_depth(disp, 0.1, 100.0)[1], gt)
    for i in range(7):
        assert abs(metrics[i].item() - ref[i].item()) <= TOL * abs(ref[i].item()), i
    assert abs(total.item() - ref_total.item()) <= TOL * abs(ref_total.item())


def test_other_sizes_crop_clipping_and_empty_mask():
    import mde_amd._abi as abi
    import mde_amd.layers as L
    g = torch.Generator().manual_seed(3)
    depth = 0.5 + 30 * torch.rand(2, 1, 17, 23, generator=g)
    gt = (1.0 + 50 * torch.rand(2, 1, 40, 61, generator=g)) * (torch.rand(2, 1, 40, 61, generator=g) < 0.3)
    m = L.depth_metrics(depth.to(DEV), gt.to(DEV), crop=(5, 1000, 3, 55)).cpu()     # y1 beyond the image: clipped like a slice
    ref_total, ref = O.depth_losses(depth, gt, crop=(5, 1000, 3, 55))
    for i in range(7):
        assert abs(m[i].item() - ref[i].item()) <= TOL * abs(ref[i].item()), i
    empty = L.depth_metrics(depth.to(DEV), torch.zeros_like(gt).to(DEV)).cpu()
    assert empty[abi.METRIC_COUNT].item() == 0 and torch.isnan(empty[:7]).all()
    with pytest.raises(ValueError):
        L.depth_metrics(depth.to(DEV), gt[:1].to(DEV))
    d = abi.DepthMetricsDesc()
    assert abi.lib.vsl_depth_metrics(ctypes.byref(d), None, None, 0, None) == -1
