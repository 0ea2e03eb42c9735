"""The benchmark configurations at their full sizes: HIP path (through the C-ABI) against the CPU
oracle on the same seeded inputs, plus size-independent properties (shared identity minimum,
forward-only heads, determinism).  Tolerances: loss 1e-5 relative, gradients 1e-5 relative (L2),
identity_selection exact wherever the two smallest candidates differ by more than float32 resolution."""
import ctypes
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def oracle_run(data, fids, B, H, W, head=0, warp_scale=1.0, scales=(0,), automask=True, avg=False, no_ssim=False,
               noise_override=None):
    from oracle import vsl_oracle as O
    cfg = O.LossConfig(frame_ids=fids, scales=list(scales), height=H, width=W, automask=automask, warp_scale=warp_scale,
                       avg_reprojection=avg, no_ssim=no_ssim)
    inputs = {"K": data["K"], "inv_K": data["inv_K"]}
    for f in fids:
        inputs[("image", f)] = data[("image", f)].double()
    disp = data[("disp", head)].clone().requires_grad_(True)
    outputs = {("disp", 0): disp}
    poses = {}
    for f in fids[1:]:
        T = (data["stereo_T"] if f == "s" else data[("cam_T_cam", f)]).clone().requires_grad_(True)
        poses[f] = T
        if f == "s":
            inputs["stereo_T"] = T
        else:
            outputs[("cam_T_cam", f)] = T
    noise = data["noise"][:, :1].contiguous() if avg else data["noise"]
    if noise_override is not None:
        noise = noise_override
    losses = O.loss_forward_backward(inputs, outputs, cfg, noise)
    # Pixels whose derivative is decided by a comparison that float32 cannot resolve (the reference evaluates the
    # warp and the errors in float64, the kernels in float32); they are compared separately:
    #  - |target - warped| of some channel below float32 resolution: sign of the L1 term (Trainer.py:487), a jump
    #    of 2 * 0.15/3 in the derivative of that pixel;
    #  - the two smallest candidates of the per-pixel minimum (Trainer.py:598) closer than float32 resolution:
    #    the winner may differ, which changes the SSIM gradient of the 3x3 neighbourhood.
    amb = torch.zeros(B, H, W, dtype=torch.bool)
    tie_px = torch.zeros(B, H, W, dtype=torch.bool)  # the per-pixel minimum itself is undecided in float32
    with torch.no_grad():
        cand, ident = [], []
        for f in fids[1:]:
            amb |= ((inputs[("image", 0)] - outputs[("color", f)]).abs() < 2e-7).any(1)
            cand.append(O.reprojection_error(outputs[("color", f)], inputs[("image", 0)], no_ssim) * warp_scale)
            ident.append(O.reprojection_error(inputs[("image", f)], inputs[("image", 0)], no_ssim))
        cand, ident = torch.cat(cand, 1), torch.cat(ident, 1)
        if avg:
            cand, ident = cand.mean(1, keepdim=True), ident.mean(1, keepdim=True)
        if automask:
            cand = torch.cat([cand, ident + noise.double() * 1e-5], 1)
        if cand.shape[1] > 1:
            two = torch.topk(cand, 2, dim=1, largest=False).values
            tie = ((two[:, 1] - two[:, 0]) < 2e-7).float().unsqueeze(1)
            tie_px = tie[:, 0] > 0
            # with SSIM the winner of a pixel's window enters the gradient of its 3x3 neighbourhood; L1 alone is local
            amb |= (tie if no_ssim else torch.nn.functional.max_pool2d(tie, 3, 1, 1))[:, 0] > 0
    sel = outputs["identity_selection"] if automask else None
    amb.tie_px = tie_px
    return losses, disp.grad, {f: poses[f].grad for f in fids[1:]}, sel, amb


def gpu_run(data, fids, head=0, warp_scale=1.0, scales=(0,), noise=True, automask=True, avg=False, no_ssim=False,
            noise_override=None):
    from mde_amd.ops import fused_view_synthesis_loss
    disp = data[("disp", head)].to(DEV).requires_grad_(True)
    poses = [(data["stereo_T"] if f == "s" else data[("cam_T_cam", f)]).to(DEV).requires_grad_(True) for f in fids[1:]]
    losses, ident = fused_view_synthesis_loss(
        disp, data[("image", 0)].to(DEV), [data[("image", f)].to(DEV) for f in fids[1:]], data["K"].to(DEV),
        data["inv_K"].to(DEV), poses,
        noise_override.to(DEV) if noise_override is not None else
        (data["noise"][:, :1].contiguous() if avg else data["noise"]).to(DEV) if noise else None,
        automask=automask, avg_reprojection=avg, no_ssim=no_ssim,
        scale_weights=[1e-3 / 2 ** s for s in scales], warp_scale=warp_scale)
    losses[0].backward()
    torch.cuda.synchronize()
    return losses.detach().cpu(), disp.grad.cpu(), [p.grad.cpu() for p in poses], ident.cpu() if automask else None


CONFIGS = {
    # name: (B, H, W, frame_ids, photometric scale, scales)
    "c1_640x192_b2": (2, 192, 640, [0, -1, 1], 1.0, (0,)),
    "c2_640x192_b12_scales4": (12, 192, 640, [0, -1, 1], 1.0, (0, 1, 2, 3)),
    "c3_1024x320_b12_stereo": (12, 320, 1024, [0, -1, 1, "s"], 1.0, (0,)),
    "c5_1280x384_b8_distillation": (8, 384, 1280, [0, -1, 1], 0.9, (0,)),
    # the option flags of the path at a benchmark size (options.py:120-138)
    "c2_640x192_b4_avg_reprojection": (4, 192, 640, [0, -1, 1], 1.0, (0,), dict(avg=True)),
    "c2_640x192_b4_no_ssim": (4, 192, 640, [0, -1, 1], 1.0, (0,), dict(no_ssim=True)),
    "c2_640x192_b4_no_automask": (4, 192, 640, [0, -1, 1], 1.0, (0,), dict(automask=False)),
    "c3_1024x320_b2_stereo_avg_no_automask": (2, 320, 1024, [0, -1, 1, "s"], 1.0, (0,), dict(avg=True, automask=False)),
    # camera motion ten times the benchmark's: a large share of the samples falls outside the source frame, where
    # grid_sample's border padding clips the coordinate and stops its gradient (Trainer.py:467)
    "c2_640x192_b4_large_motion": (4, 192, 640, [0, -1, 1], 1.0, (0,), dict(pose_sigma=0.1)),
}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_config_matches_oracle(name):
    import mde_amd.synth as synth
    B, H, W, fids, kappa, scales = CONFIGS[name][:6]
    flags = dict(CONFIGS[name][6]) if len(CONFIGS[name]) > 6 else {}
    torch.set_num_threads(16)
    data = synth.make_batch(B, H, W, fids, seed=7, pose_sigma=flags.pop("pose_sigma", 0.01))
    automask, avg = flags.get("automask", True), flags.get("avg", False)
    ref, g_ref, gT_ref, sel_ref, amb = oracle_run(data, fids, B, H, W, warp_scale=kappa, scales=scales, **flags)
    losses, g, gT, sel = gpu_run(data, fids, warp_scale=kappa, scales=scales, **flags)
    assert abs(losses[0].item() - ref["loss"].item()) <= TOL * abs(ref["loss"].item())
    for i, s in enumerate(scales):
        assert abs(losses[3 + i].item() - ref["loss/%d" % s].item()) <= TOL * abs(ref["loss/%d" % s].item())
    n_amb = int(amb.sum())
    assert n_amb <= max(2, int(1e-4 * amb.numel())), "too many float32-ambiguous pixels: %d" % n_amb
    keep = (~amb).unsqueeze(1).float()
    err_g = rel((g * keep).numpy(), (g_ref * keep).numpy())
    assert err_g <= TOL, "grad_disp relative error %.3e (%d ambiguous pixels left out)" % (err_g, n_amb)
    if n_amb:   # at the left-out pixels the two results differ by at most the jump of the L1 derivative
        gap = float(((g - g_ref).abs() * amb.unsqueeze(1).float()).max())
        assert gap <= 0.11 * float(g_ref.abs().max()) + 1e-12
    # Pose gradients image by image.  Raw: an image that holds an ambiguous pixel may differ by one flipped decision,
    # and because the per-pixel contributions to a pose gradient are heavy-tailed one flip can move the image's 4x4
    # gradient by a few 1e-3 -- the size of the reference's own float32 inconsistency (5e-3 between two association
    # orders of its projection).  Every image without such a pixel must meet 1e-5, every other 5e-3.
    n_amb_img = amb.flatten(1).sum(1)
    raw = {}
    for i, f in enumerate(fids[1:]):
        errs = [rel(gT[i][b].numpy(), gT_ref[f][b].numpy()) for b in range(B)]
        raw[f] = max(errs)
        msg = "grad_T[%s] relative errors %s" % (f, ["%.1e" % e for e in errs])
        assert all(errs[b] <= (TOL if n_amb_img[b] == 0 else 5e-3) for b in range(B)), msg
    # Masked: both sides evaluated once more with the ambiguous pixels (grown by the 5x5 footprint of the SSIM
    # backward) taken out of the gradient.  The tie-break noise is an input of both: -1e9 there makes the identity
    # term win (Trainer.py:587-599), so those pixels carry no gradient to disparity or pose on either side, and
    # nothing that float32 cannot resolve is left.  Then every image must meet 1e-4 (the bar for a masked
    # comparison; measured a few 1e-6).
    if automask and n_amb and not avg:
        grown = torch.nn.functional.max_pool2d(amb.float().unsqueeze(1), 5, 1, 2) > 0          # [B,1,H,W]
        noise2 = torch.where(grown.expand_as(data["noise"]), torch.full_like(data["noise"], -1e9), data["noise"])
        _, g2_ref, gT2_ref, _, amb2 = oracle_run(data, fids, B, H, W, warp_scale=kappa, scales=scales, noise_override=noise2, **flags)
        _, g2, gT2, _ = gpu_run(data, fids, warp_scale=kappa, scales=scales, noise_override=noise2, **flags)
        assert int((amb2 & ~grown[:, 0]).sum()) == 0
        for i, f in enumerate(fids[1:]):
            errs = [rel(gT2[i][b].numpy(), gT2_ref[f][b].numpy()) for b in range(B)]
            print("grad_T[%s] %s: raw max %.1e (reference against itself: 5e-3), masked max %.1e" % (f, name, raw[f], max(errs)))
            assert max(errs) <= 1e-4, "masked grad_T[%s] relative errors %s" % (f, ["%.1e" % e for e in errs])
        assert rel(g2.numpy(), g2_ref.numpy()) <= TOL
    if sel_ref is not None:
        # identity_selection must be exact wherever float32 can decide the minimum: a flip is only tolerated at a pixel
        # whose two smallest candidates are closer than 2e-7 in the float64 oracle (a handful per million at most)
        differs = torch.from_numpy(sel.numpy() != sel_ref.numpy())
        flips, undecided = int((differs & ~amb.tie_px).sum()), int((differs & amb.tie_px).sum())
        print("identity_selection %s: %d flips at float32-undecidable pixels (of %d such pixels)" % (name, undecided, int(amb.tie_px.sum())))
        assert flips == 0, "%d identity_selection flips at decidable pixels" % flips
        assert undecided <= 2


def _desc(abi, t, B, H, W, fids, head, flags, idmin=None, warp_scale=1.0):
    d = abi.LossDesc()
    d.B, d.H, d.W, d.S = B, H, W, len(fids) - 1
    d.flags = flags
    d.min_depth, d.max_depth, d.eps = 0.1, 100.0, 1e-7
    d.ssim_weight, d.l1_weight, d.warp_scale = 0.85, 0.15, warp_scale
    d.n_scales, d.inv_num_scales = 1, 1.0
    d.scale_weight[0] = 1e-3
    d.disp, d.target = t[("disp", head)].data_ptr(), t[("image", 0)].data_ptr()
    for s, f in enumerate(fids[1:]):
        d.source[s] = t[("image", f)].data_ptr()
        d.T[s] = t[("cam_T_cam", f)].data_ptr()
    d.K, d.inv_K = t["K"].data_ptr(), t["inv_K"].data_ptr()
    d.noise = t["noise"].data_ptr()
    d.identity_min = idmin.data_ptr() if idmin is not None else None
    return d


def test_two_heads_share_identity_minimum_and_forward_only_head():
    """Distillation layout (1280x384, batch 8): a student head with gradients and a teacher head evaluated
    forward-only, both reading one identity minimum computed once.  Sharing must not change a bit, and the
    forward-only result must equal the forward part of a full evaluation."""
    import mde_amd._abi as abi
    import mde_amd.synth as synth
    B, H, W, fids = 8, 384, 1280, [0, -1, 1]
    data = synth.make_batch(B, H, W, fids, seed=11, n_heads=2)
    t = {k: v.to(DEV) for k, v in data.items() if torch.is_tensor(v)}
    stream = torch.cuda.current_stream().cuda_stream
    idmin = torch.empty(B, H, W, device=DEV)
    abi.check(abi.lib.vsl_identity_min(ctypes.byref(_desc(abi, t, B, H, W, fids, 0, abi.FLAG_AUTOMASK)), idmin.data_ptr(), stream))

    def run(head, flags, shared):
        d = _desc(abi, t, B, H, W, fids, head, flags, idmin if shared else None, warp_scale=0.9)
        n = abi.lib.vsl_loss_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(n, dtype=torch.uint8, device=DEV)
        losses = torch.zeros(abi.LOSS_SLOTS, device=DEV)
        sel = torch.empty(B, H, W, device=DEV)
        abi.check(abi.lib.vsl_loss_forward(ctypes.byref(d), ws.data_ptr(), n, losses.data_ptr(), sel.data_ptr(), stream))
        g = None
        if not flags & abi.FLAG_NO_GRAD:
            g = torch.empty(B, 1, H, W, device=DEV)
            abi.check(abi.lib.vsl_loss_backward(ctypes.byref(d), ws.data_ptr(), n, None, g.data_ptr(), None, stream))
        torch.cuda.synchronize()
        return losses.cpu().numpy(), sel.cpu().numpy(), None if g is None else g.cpu().numpy()

    l_own, s_own, g_own = run(0, abi.FLAG_AUTOMASK, shared=False)
    l_sh, s_sh, g_sh = run(0, abi.FLAG_AUTOMASK, shared=True)
    assert np.array_equal(l_own, l_sh) and np.array_equal(s_own, s_sh) and np.array_equal(g_own, g_sh)
    l_again, _, g_again = run(0, abi.FLAG_AUTOMASK, shared=True)
    assert np.array_equal(l_sh, l_again) and np.array_equal(g_sh, g_again)          # deterministic
    l_t_full, s_t_full, _ = run(1, abi.FLAG_AUTOMASK, shared=True)
    l_t_fwd, s_t_fwd, _ = run(1, abi.FLAG_AUTOMASK | abi.FLAG_NO_GRAD, shared=True)
    # the forward-only program sums the same per-pixel values in another order: equal to float32 summation noise
    np.testing.assert_allclose(l_t_fwd, l_t_full, rtol=2e-6, atol=0)
    assert np.array_equal(s_t_full, s_t_fwd)
    assert not np.array_equal(l_t_full, l_sh)                                        # the heads do differ

    with pytest.raises(abi.VslError):
        d = _desc(abi, t, B, H, W, fids, 1, abi.FLAG_AUTOMASK | abi.FLAG_NO_GRAD, idmin)
        n = abi.lib.vsl_loss_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(n, dtype=torch.uint8, device=DEV)
        abi.check(abi.lib.vsl_loss_backward(ctypes.byref(d), ws.data_ptr(), n, None, ws.data_ptr(), None, stream))


def test_in_kernel_noise_changes_only_rare_ties():
    """Without a noise tensor the kernels draw the tie-break noise themselves: same statistics, so the loss
    moves by far less than the 1e-5 noise scale allows and the result is reproducible for a fixed seed."""
    import mde_amd.synth as synth
    B, H, W, fids = 4, 192, 640, [0, -1, 1]
    data = synth.make_batch(B, H, W, fids, seed=3)
    a = gpu_run(data, fids, noise=True)
    b = gpu_run(data, fids, noise=False)
    c = gpu_run(data, fids, noise=False)
    assert abs(a[0][0].item() - b[0][0].item()) < 1e-4 * a[0][0].item()
    assert torch.equal(b[0], c[0]) and torch.equal(b[1], c[1])


@pytest.mark.parametrize("shape,fids", [((2, 48, 160), [0, -1, 1]), ((12, 192, 640), [0, -1, 1]), ((2, 40, 70), [0, -1, 1, "s"])])
def test_merged_heads_equal_separate_calls(shape, fids):
    """vsl_loss_forward_heads / vsl_loss_backward_heads: four heads in one launch give, bit for bit, the
    losses, selections and disparity gradients of four separate calls; the pose gradient is the sum over heads."""
    import mde_amd.synth as synth
    from mde_amd.ops import fused_view_synthesis_loss, fused_view_synthesis_loss_heads
    B, H, W = shape
    n_heads = 4
    data = synth.make_batch(B, H, W, fids, seed=5, n_heads=n_heads)
    img = lambda f: data[("image", f)].to(DEV)
    pose = lambda f: (data["stereo_T"] if f == "s" else data[("cam_T_cam", f)]).to(DEV)
    noise = data["noise"].to(DEV)
    sw = [[1e-3 / 2 ** h] for h in range(n_heads)]
    up = torch.linspace(0.5, 2.0, n_heads * 11, device=DEV).reshape(n_heads, 11)   # some upstream gradient

    disps = [data[("disp", h)].to(DEV).requires_grad_(True) for h in range(n_heads)]
    poses = [pose(f).requires_grad_(True) for f in fids[1:]]
    losses, ident = fused_view_synthesis_loss_heads(disps, img(0), [img(f) for f in fids[1:]], data["K"].to(DEV),
                                                    data["inv_K"].to(DEV), poses, noise, scale_weights=sw)
    (losses * up).sum().backward()

    gT_sum = [torch.zeros(B, 4, 4, device=DEV) for _ in fids[1:]]
    for h in range(n_heads):
        d = data[("disp", h)].to(DEV).requires_grad_(True)
        ps = [pose(f).requires_grad_(True) for f in fids[1:]]
        l1, i1 = fused_view_synthesis_loss(d, img(0), [img(f) for f in fids[1:]], data["K"].to(DEV),
                                           data["inv_K"].to(DEV), ps, noise, scale_weights=sw[h])
        (l1 * up[h]).sum().backward()
        assert torch.equal(l1, losses[h]) and torch.equal(i1, ident[h])
        assert torch.equal(d.grad, disps[h].grad)
        for s in range(len(ps)):
            gT_sum[s] += ps[s].grad
    for s in range(len(poses)):
        err = float((poses[s].grad - gT_sum[s]).norm() / gT_sum[s].norm())
        assert err < 1e-6, err


def test_student_and_forward_only_teacher_in_one_launch():
    """Distillation layout through the merged call: head 0 (student) with gradients, head 1 (teacher) marked
    forward only.  Both match their separate evaluations bit for bit; gradients come from the student alone."""
    import mde_amd.synth as synth
    from mde_amd.ops import fused_view_synthesis_loss, fused_view_synthesis_loss_heads
    B, H, W, fids = 4, 192, 640, [0, -1, 1]
    data = synth.make_batch(B, H, W, fids, seed=13, n_heads=2)
    img = lambda f: data[("image", f)].to(DEV)
    noise = data["noise"].to(DEV)
    common = dict(scale_weights=[[1e-3], [1e-3]], warp_scale=0.9)

    disps = [data[("disp", h)].to(DEV).requires_grad_(True) for h in range(2)]
    poses = [data[("cam_T_cam", f)].to(DEV).requires_grad_(True) for f in fids[1:]]
    losses, ident = fused_view_synthesis_loss_heads(disps, img(0), [img(f) for f in fids[1:]], data["K"].to(DEV),
                                                    data["inv_K"].to(DEV), poses, noise, forward_only=[False, True], **common)
    losses[0, 0].backward()
    assert disps[1].grad is None

    d0 = data[("disp", 0)].to(DEV).requires_grad_(True)
    p0 = [data[("cam_T_cam", f)].to(DEV).requires_grad_(True) for f in fids[1:]]
    l0, i0 = fused_view_synthesis_loss(d0, img(0), [img(f) for f in fids[1:]], data["K"].to(DEV), data["inv_K"].to(DEV),
                                       p0, noise, scale_weights=[1e-3], warp_scale=0.9)
    l0[0].backward()
    assert torch.equal(l0, losses[0]) and torch.equal(i0, ident[0]) and torch.equal(d0.grad, disps[0].grad)
    for a, b in zip(p0, poses):
        assert torch.equal(a.grad, b.grad)
    with torch.no_grad():
        l1, i1 = fused_view_synthesis_loss(data[("disp", 1)].to(DEV), img(0), [img(f) for f in fids[1:]], data["K"].to(DEV),
                                           data["inv_K"].to(DEV), [p.detach() for p in poses], noise, scale_weights=[1e-3],
                                           warp_scale=0.9)
    # (a launch without any gradient runs the forward-only program: same per-pixel values, other summation order)
    assert torch.allclose(l1, losses[1], rtol=2e-6, atol=0) and torch.equal(i1, ident[1])


def test_batch_offsets_just_below_the_index_limit():
    """B*3*H*W = 2 146 959 360, 524 288 below the documented 2^31 limit (39 images of 8960x2048): the first, a
    middle and the last image evaluated on their own give the same identity_selection bit for bit and, up to the
    1/B of the batch mean, the same disparity and pose gradients as inside the batch.  About 55 GB of device memory."""
    from mde_amd.ops import fused_view_synthesis_loss
    B, H, W = 39, 2048, 8960
    assert B * 3 * H * W < 2 ** 31 and (B + 1) * 3 * H * W >= 2 ** 31
    if torch.cuda.get_device_properties(0).total_memory < 120 * 2 ** 30:
        pytest.skip("needs a device with more than 120 GB")
    g = torch.Generator(device=DEV).manual_seed(17)
    pool = torch.nn.AvgPool2d(5, 1, 2, count_include_pad=False)
    imgs = [pool(torch.rand(B, 3, H, W, device=DEV, generator=g)) for _ in range(3)]
    disp = 0.05 + 0.9 * pool(torch.rand(B, 1, H, W, device=DEV, generator=g))
    noise = torch.randn(B, 2, H, W, device=DEV, generator=g)
    K = torch.tensor([[0.58 * (W // 2), 0, 0.5 * (W // 2), 0], [0, 1.92 * (H // 2), 0.5 * (H // 2), 0], [0, 0, 1, 0], [0, 0, 0, 1]],
                     device=DEV).repeat(B, 1, 1)
    inv_K = torch.linalg.pinv(K)
    T = [torch.eye(4, device=DEV).repeat(B, 1, 1) for _ in range(2)]
    for s, t in enumerate(T):
        t[:, :3, 3] = 0.02 * torch.randn(B, 3, device=DEV, generator=g)
        t[:, 0, 1] = 0.003 * (s + 1)
        t[:, 1, 0] = -0.003 * (s + 1)

    def run(sl):
        d = disp[sl].clone().requires_grad_(True)
        poses = [t[sl].clone().requires_grad_(True) for t in T]
        losses, ident = fused_view_synthesis_loss(d, imgs[0][sl], [imgs[1][sl], imgs[2][sl]], K[sl], inv_K[sl], poses, noise[sl])
        losses[0].backward()
        return losses.detach(), ident, d.grad, [p.grad for p in poses]

    l_all, sel_all, g_all, gT_all = run(slice(0, B))
    assert torch.isfinite(l_all[:4]).all() and 0.0 < float(sel_all.mean()) < 1.0
    for b in (0, B // 2, B - 1):
        l_one, sel_one, g_one, gT_one = run(slice(b, b + 1))
        assert torch.equal(sel_one[0], sel_all[b]), "identity_selection of image %d" % b
        err = float((g_all[b] * B - g_one[0]).norm() / g_one[0].norm())
        assert err <= TOL, "grad_disp of image %d: %.2e" % (b, err)     # the batch-mean factors round differently
        for s in range(2):
            err = float((gT_all[s][b] * B - gT_one[s][0]).norm() / gT_one[s][0].norm())
            assert err <= 1e-5, "grad_T[%d] of image %d: %.2e" % (s, b, err)
        del l_one, sel_one, g_one, gT_one
    # one image more is refused, not mis-indexed
    with pytest.raises(Exception):
        big = torch.empty(B + 1, 1, H, W, device=DEV)
        fused_view_synthesis_loss(big, imgs[0], [imgs[1], imgs[2]], K, inv_K, T, None)


def test_loss_slots_of_a_large_batch_equal_the_mean_over_its_halves():
    """The loss slots are means over the batch, assembled by whichever workgroup of the reduction arrives last: a
    batch of 20 must give the mean of its two halves of 10, slot by slot, and the same gradients."""
    import mde_amd.synth as synth
    from mde_amd.ops import fused_view_synthesis_loss
    B, H, W, fids = 20, 48, 160, [0, -1, 1]
    data = synth.make_batch(B, H, W, fids, seed=23)
    t = {k: v.to(DEV) for k, v in data.items() if torch.is_tensor(v)}

    def run(sl):
        d = t[("disp", 0)][sl].clone().requires_grad_(True)
        poses = [t[("cam_T_cam", f)][sl].clone().requires_grad_(True) for f in fids[1:]]
        losses, _ = fused_view_synthesis_loss(d, t[("image", 0)][sl], [t[("image", f)][sl] for f in fids[1:]], t["K"][sl],
                                              t["inv_K"][sl], poses, t["noise"][sl])
        losses[0].backward()
        return losses.detach().double().cpu(), d.grad.cpu(), [p.grad.cpu() for p in poses]

    l_all, g_all, gT_all = run(slice(0, B))
    l_a, g_a, gT_a = run(slice(0, B // 2))
    l_b, g_b, gT_b = run(slice(B // 2, B))
    want = 0.5 * (l_a + l_b)
    assert torch.all((l_all - want).abs() <= 1e-6 * want.abs() + 1e-12), (l_all, want)
    assert rel(g_all.numpy() * 2, torch.cat([g_a, g_b]).numpy()) <= 1e-6
    for s in range(2):
        assert rel(gT_all[s].numpy() * 2, torch.cat([gT_a[s], gT_b[s]]).numpy()) <= 1e-6


def test_in_kernel_tie_break_noise_is_standard_normal_and_keyed_by_seed():
    """With source == target the identity error is exactly 0, so vsl_identity_min returns min over sources of
    1e-5 * z (Trainer.py:587-590): with one source that is the noise itself.  The kernel's own generator (no noise
    tensor passed) must give N(0,1) values (mean, variance, fourth moment, tail), the same values for the same
    seed, different ones for another seed, and no correlation between neighbouring pixels or images."""
    import mde_amd._abi as abi
    import mde_amd.synth as synth
    B, H, W, fids = 4, 192, 640, [0, 1]
    data = synth.make_batch(B, H, W, fids, seed=17)
    t = {k: v.to(DEV) for k, v in data.items() if torch.is_tensor(v)}
    t[("image", 1)] = t[("image", 0)].clone()
    stream = torch.cuda.current_stream().cuda_stream

    def draw(seed):
        d = _desc(abi, t, B, H, W, fids, 0, abi.FLAG_AUTOMASK)
        d.noise, d.seed = None, seed
        out = torch.empty(B, H, W, device=DEV)
        abi.check(abi.lib.vsl_identity_min(ctypes.byref(d), out.data_ptr(), stream))
        torch.cuda.synchronize()
        return (out.double() / 1e-5).cpu()

    z, z_again, z_other = draw(99), draw(99), draw(100)
    assert torch.equal(z, z_again)
    n = z.numel()
    assert abs(float(z.mean())) < 5.0 / n ** 0.5
    assert abs(float(z.var()) - 1.0) < 0.01
    assert abs(float((z ** 4).mean()) - 3.0) < 0.05
    assert 0.0020 < float((z.abs() > 3.0).double().mean()) < 0.0034          # 0.0027 for N(0,1)
    assert float(z.abs().max()) < 6.0
    for a, b in ((z[:, :, 1:], z[:, :, :-1]), (z[:, 1:], z[:, :-1]), (z[1:], z[:-1]), (z, z_other)):
        c = float((a * b).mean())
        assert abs(c) < 5.0 / a.numel() ** 0.5, c


def test_in_kernel_noise_of_two_source_planes_is_independent():
    """Two sources equal to the target: vsl_identity_min returns 1e-5 * min(z0, z1).  The kernel draws the noise of a
    pixel's two source planes from one Box-Muller transform (cosine and sine branch); the minimum of two independent
    N(0,1) values has mean -1/sqrt(pi) and variance 1 - 1/pi (correlated planes would move both towards 0 and 1)."""
    import mde_amd._abi as abi
    import mde_amd.synth as synth
    B, H, W, fids = 4, 192, 640, [0, -1, 1]
    data = synth.make_batch(B, H, W, fids, seed=17)
    t = {k: v.to(DEV) for k, v in data.items() if torch.is_tensor(v)}
    t[("image", -1)] = t[("image", 0)].clone()
    t[("image", 1)] = t[("image", 0)].clone()
    d = _desc(abi, t, B, H, W, fids, 0, abi.FLAG_AUTOMASK)
    d.noise, d.seed = None, 5
    out = torch.empty(B, H, W, device=DEV)
    abi.check(abi.lib.vsl_identity_min(ctypes.byref(d), out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    m = (out.double() / 1e-5).cpu()
    n = m.numel()
    assert abs(float(m.mean()) + math.pi ** -0.5) < 5.0 / n ** 0.5
    assert abs(float(m.var()) - (1.0 - 1.0 / math.pi)) < 0.01
    c = float(((m[:, :, 1:] - m.mean()) * (m[:, :, :-1] - m.mean())).mean())
    assert abs(c) < 5.0 / n ** 0.5, c


def test_directional_derivative_agrees_with_the_gradient_at_benchmark_size():
    """Oracle-free check of forward against backward at 640x192 b12: along v = sign(grad_disp) the central
    difference (L(d + e v) - L(d - e v)) / 2e must equal sum |grad_disp|.  The loss has kinks (|.|, the per-pixel
    minimum, the SSIM clamp) and is evaluated in float32, so the agreement is a few parts in a thousand; a wrong
    backward term (a missing factor, a sign, a dropped halo contribution) would show at the percent level or more."""
    import mde_amd._abi as abi
    import mde_amd.synth as synth
    from mde_amd.ops import fused_view_synthesis_loss
    B, H, W, fids = 12, 192, 640, [0, -1, 1]
    data = synth.make_batch(B, H, W, fids, seed=23)
    img = lambda f: data[("image", f)].to(DEV)
    args = (img(0), [img(f) for f in fids[1:]], data["K"].to(DEV), data["inv_K"].to(DEV),
            [data[("cam_T_cam", f)].to(DEV) for f in fids[1:]], data["noise"].to(DEV))
    d = data[("disp", 0)].to(DEV).requires_grad_(True)
    out, _ = fused_view_synthesis_loss(d, *args, scale_weights=[1e-3])
    out[abi.LOSS_TOTAL].backward()
    v = torch.sign(d.grad)
    expect = float(d.grad.double().abs().sum())
    eps = 1e-4
    with torch.no_grad():
        lp = float(fused_view_synthesis_loss(d.detach() + eps * v, *args, scale_weights=[1e-3])[0][abi.LOSS_TOTAL])
        lm = float(fused_view_synthesis_loss(d.detach() - eps * v, *args, scale_weights=[1e-3])[0][abi.LOSS_TOTAL])
    got = (lp - lm) / (2 * eps)
    print("directional derivative %.6e, sum |grad| %.6e, ratio %.5f" % (got, expect, got / expect))
    assert abs(got / expect - 1.0) < 0.02


def test_directional_derivative_agrees_with_the_pose_gradient_at_benchmark_size():
    """The same check for the pose matrices (the 12 entries of [R|t] of every image and source, perturbed as free
    variables along the sign of their gradient): central difference against sum |grad_T|."""
    import mde_amd._abi as abi
    import mde_amd.synth as synth
    from mde_amd.ops import fused_view_synthesis_loss
    B, H, W, fids = 12, 192, 640, [0, -1, 1]
    data = synth.make_batch(B, H, W, fids, seed=29)
    img = lambda f: data[("image", f)].to(DEV)
    d = data[("disp", 0)].to(DEV)
    poses = [data[("cam_T_cam", f)].to(DEV).requires_grad_(True) for f in fids[1:]]
    run = lambda ps: fused_view_synthesis_loss(d, img(0), [img(f) for f in fids[1:]], data["K"].to(DEV), data["inv_K"].to(DEV),
                                               ps, data["noise"].to(DEV), scale_weights=[1e-3])[0][abi.LOSS_TOTAL]
    run(poses).backward()
    vs = [torch.sign(p.grad) for p in poses]
    expect = sum(float(p.grad.double().abs().sum()) for p in poses)
    eps = 5e-6
    with torch.no_grad():
        lp = float(run([p.detach() + eps * v for p, v in zip(poses, vs)]))
        lm = float(run([p.detach() - eps * v for p, v in zip(poses, vs)]))
    got = (lp - lm) / (2 * eps)
    print("directional derivative %.6e, sum |grad_T| %.6e, ratio %.5f" % (got, expect, got / expect))
    assert abs(got / expect - 1.0) < 0.02


def test_a_step_captured_in_a_hip_graph_replays_bit_for_bit():
    """The entry points allocate nothing and only enqueue work on the caller's stream, so a whole step (identity
    pass, fused kernel, finalize, gradient assembly of two heads) can be captured in a HIP graph
    (torch.cuda.CUDAGraph) and replayed: every output equals the eager calls bit for bit."""
    import bench
    B, H, W, fids, heads = 2, 192, 640, [0, -1, 1], 2
    dev = torch.device(DEV)
    data, t = bench.build_inputs(B, H, W, fids, heads, dev)
    step = bench.LossStep(t, B, H, W, fids, heads, dev)
    outputs = lambda: step.losses + step.gdisp + [g for gs in step.gT for g in gs] + step.ident
    step(torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    eager = [x.clone() for x in outputs()]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=torch.cuda.Stream(dev)):
        step(torch.cuda.current_stream(dev).cuda_stream)
    for x in outputs():
        x.fill_(float("nan"))
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(eager, outputs()):
        assert torch.equal(a, b)


@pytest.mark.parametrize("fids,flags", [
    ([0, 1], {}),
    ([0, -1, 1, "s"], {}),
    ([0, -1, 1], {"avg_reprojection": True}),
    ([0, -1, 1, "s"], {"avg_reprojection": True}),
    ([0, -1, 1], {"no_ssim": True}),
    ([0, -1, 1], {"automask": False}),
])
def test_directional_derivative_of_the_other_kernel_variants(fids, flags):
    """Forward against backward without the oracle for the other instantiations of the tile program (one and
    three sources, averaged reprojection, L1 only, no auto-masking), 4 x 192 x 640: disparity and pose directions
    together, central difference against sum |grad|."""
    import mde_amd._abi as abi
    import mde_amd.synth as synth
    from mde_amd.ops import fused_view_synthesis_loss
    B, H, W = 4, 192, 640
    data = synth.make_batch(B, H, W, fids, seed=31 + len(fids))
    img = lambda f: data[("image", f)].to(DEV)
    pose = lambda f: (data["stereo_T"] if f == "s" else data[("cam_T_cam", f)]).to(DEV)
    S = len(fids) - 1
    noise = data["noise"].to(DEV)
    if flags.get("avg_reprojection"):
        noise = noise[:, :1].contiguous()
    run = lambda d, ps: fused_view_synthesis_loss(d, img(0), [img(f) for f in fids[1:]], data["K"].to(DEV), data["inv_K"].to(DEV),
                                                  ps, noise, scale_weights=[1e-3], **flags)[0][abi.LOSS_TOTAL]
    d = data[("disp", 0)].to(DEV).requires_grad_(True)
    poses = [pose(f).requires_grad_(True) for f in fids[1:]]
    run(d, poses).backward()
    for what, eps in (("disp", 1e-4), ("pose", 5e-6)):
        vd = torch.sign(d.grad) if what == "disp" else torch.zeros_like(d)
        vp = [torch.sign(p.grad) if what == "pose" else torch.zeros_like(p) for p in poses]
        expect = float(d.grad.double().abs().sum()) if what == "disp" else sum(float(p.grad.double().abs().sum()) for p in poses)
        with torch.no_grad():
            lp = float(run(d.detach() + eps * vd, [p.detach() + eps * v for p, v in zip(poses, vp)]))
            lm = float(run(d.detach() - eps * vd, [p.detach() - eps * v for p, v in zip(poses, vp)]))
        got = (lp - lm) / (2 * eps)
        print("%s %s: directional derivative %.6e, sum |grad| %.6e, ratio %.5f" % (fids, what, got, expect, got / expect))
        assert abs(got / expect - 1.0) < 0.03, (what, got / expect)
