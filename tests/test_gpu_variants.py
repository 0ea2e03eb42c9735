"""Every compiled variant of the tile kernels (1..4 source frames x min / avg reprojection, with and without
auto-masking and SSIM) on the GPU against the CPU build of the same tile program (tests/emu), on shapes that
exercise partial tiles.  Tolerance 1e-5 relative; identity_selection exact.  The golden fixtures cover the
reference's own configurations; this covers the instantiations they do not reach."""
import numpy as np
import pytest
import torch

import emu_driver as E

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("S", [1, 2, 3, 4])
@pytest.mark.parametrize("avg", [False, True])
@pytest.mark.parametrize("automask,no_ssim", [(True, False), (False, False), (True, True)])
def test_variant_matches_cpu_build(S, avg, automask, no_ssim):
    import mde_amd.synth as synth
    from mde_amd.ops import fused_view_synthesis_loss
    B, H, W = 2, 45, 100                       # 4 x 4 tiles, the last row and column partial
    fids = [0, -1, 1, 2, -2][:S + 1]
    data = synth.make_batch(B, H, W, fids, seed=50 + S, blur=3)
    n_id = 1 if avg else S
    noise = data["noise"][:, :n_id].contiguous()
    srcs = fids[1:]
    sw = [1e-3, 5e-4]
    r = E.run_emu(data[("disp", 0)].numpy(), data[("image", 0)].numpy(), [data[("image", f)].numpy() for f in srcs],
                  data["K"].numpy(), data["inv_K"].numpy(), [data[("cam_T_cam", f)].numpy() for f in srcs],
                  noise.numpy() if automask else None, automask=automask, avg=avg, no_ssim=no_ssim, scale_weights=sw)
    disp = data[("disp", 0)].to(DEV).requires_grad_(True)
    poses = [data[("cam_T_cam", f)].to(DEV).requires_grad_(True) for f in srcs]
    losses, ident = fused_view_synthesis_loss(
        disp, data[("image", 0)].to(DEV), [data[("image", f)].to(DEV) for f in srcs], data["K"].to(DEV),
        data["inv_K"].to(DEV), poses, noise.to(DEV) if automask else None, automask=automask, avg_reprojection=avg,
        no_ssim=no_ssim, scale_weights=sw)
    losses[0].backward()
    assert abs(losses[0].item() - r["loss"]) <= TOL * abs(r["loss"])
    assert abs(losses[2].item() - r["smooth"]) <= TOL * abs(r["smooth"])
    assert rel(disp.grad.cpu().numpy().reshape(-1), np.asarray(r["grad_disp"]).reshape(-1)) <= TOL
    for i in range(S):
        assert rel(poses[i].grad.cpu().numpy(), r["grad_T"][i]) <= TOL
    if automask:
        assert np.array_equal(ident.cpu().numpy().reshape(-1), np.asarray(r["ident_sel"]).reshape(-1))
    # forward-only program of the same variant
    with torch.no_grad():
        l2, i2 = fused_view_synthesis_loss(
            disp.detach(), data[("image", 0)].to(DEV), [data[("image", f)].to(DEV) for f in srcs], data["K"].to(DEV),
            data["inv_K"].to(DEV), [p.detach() for p in poses], noise.to(DEV) if automask else None, automask=automask,
            avg_reprojection=avg, no_ssim=no_ssim, scale_weights=sw)
    assert torch.allclose(l2, losses.detach(), rtol=2e-6, atol=0)
    if automask:
        assert torch.equal(i2, ident)


@pytest.mark.gpu
@pytest.mark.parametrize("fids,avg,automask", [
    ([0, -2, -1, 1, 2], False, True),      # four sources, per-pixel minimum
    ([0, -2, -1, 1, 2], True, True),       # four sources, mean over sources
    ([0, -2, -1, 1, 2], False, False),
    ([0, -1, 1, 2], True, True),           # three sources, mean over sources
    ([0, -1, 1, 2], True, False),
    ([0, -1], True, True),                 # one source with the mean-over-sources program
])
def test_three_and_four_sources_against_the_oracle(fids, avg, automask):
    """Every vsl_fused_kernel<S, AVG> instantiation that the reference fixtures do not reach (the fixtures stop at
    three sources with the per-pixel minimum) against the CPU oracle, which takes any frame list (Trainer.py:538-605
    loops over opt.frame_ids[1:]): ragged shape, two tiles wide and three high, loss / gradients within 1e-5."""
    import numpy as np
    import torch
    import mde_amd.synth as synth
    from mde_amd.ops import fused_view_synthesis_loss
    from oracle import vsl_oracle as O
    B, H, W = 2, 29, 47
    data = synth.make_batch(B, H, W, fids, seed=900 + len(fids) + 10 * avg, blur=3)
    S = len(fids) - 1
    noise = data["noise"][:, :1].contiguous() if avg else data["noise"]
    dev = torch.device("cuda:0")
    disp = data[("disp", 0)].to(dev).requires_grad_(True)
    poses = [data[("cam_T_cam", f)].to(dev).requires_grad_(True) for f in fids[1:]]
    losses, ident = fused_view_synthesis_loss(
        disp, data[("image", 0)].to(dev), [data[("image", f)].to(dev) for f in fids[1:]],
        data["K"].to(dev), data["inv_K"].to(dev), poses, noise.to(dev) if automask else None,
        automask=automask, avg_reprojection=avg)
    losses[0].backward()
    cfg = O.LossConfig(frame_ids=fids, height=H, width=W, automask=automask, avg_reprojection=avg)
    inputs = {"K": data["K"], "inv_K": data["inv_K"]}
    for f in fids:
        inputs[("image", f)] = data[("image", f)].double()
    d_ref = data[("disp", 0)].clone().requires_grad_(True)
    outputs = {("disp", 0): d_ref}
    for f in fids[1:]:
        outputs[("cam_T_cam", f)] = data[("cam_T_cam", f)].clone().requires_grad_(True)
    ref = O.loss_forward_backward(inputs, outputs, cfg, noise)
    rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))
    assert abs(losses[0].item() - ref["loss"].item()) <= 1e-5 * abs(ref["loss"].item())
    assert rel(disp.grad.cpu().numpy(), d_ref.grad.numpy()) <= 1e-5
    for i, f in enumerate(fids[1:]):
        assert rel(poses[i].grad.cpu().numpy(), outputs[("cam_T_cam", f)].grad.numpy()) <= 2e-5, (f, S)
    if automask:
        assert np.array_equal(ident.cpu().numpy(), outputs["identity_selection"].numpy())
