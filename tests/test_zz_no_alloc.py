"""vsl_loss_forward_static: the forward pass for callers that cannot allocate device memory.

Called through the C-ABI twice in a row on one stream -- the second time with a smaller input, so that
whatever the first call left in the library's own scratch array is still there -- and compared bit for
bit with vsl_loss_forward (forward only) on a caller-owned workspace.  The oracle pins the value once."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _desc(abi, t, B, H, W, fids, flags, with_noise=True):
    d = abi.LossDesc()
    d.B, d.H, d.W, d.S = B, H, W, len(fids) - 1
    d.flags = flags
    d.min_depth, d.max_depth, d.eps = 0.1, 100.0, 1e-7
    d.ssim_weight, d.l1_weight, d.warp_scale = 0.85, 0.15, 1.0
    d.n_scales, d.inv_num_scales = 2, 1.0
    d.scale_weight[0], d.scale_weight[1] = 1e-3, 5e-4
    d.disp, d.target = t[("disp", 0)].data_ptr(), t[("image", 0)].data_ptr()
    for s, f in enumerate(fids[1:]):
        d.source[s] = t[("image", f)].data_ptr()
        d.T[s] = t[("cam_T_cam", f)].data_ptr()
    d.K, d.inv_K = t["K"].data_ptr(), t["inv_K"].data_ptr()
    d.noise = t["noise"].data_ptr() if with_noise else None
    d.seed = 77
    return d


def _both(abi, B, H, W, fids, seed, flags):
    import mde_amd.synth as synth
    data = synth.make_batch(B, H, W, fids, seed=seed, n_heads=1)
    t = {k: v.to(DEV) for k, v in data.items() if torch.is_tensor(v)}
    stream = torch.cuda.current_stream().cuda_stream
    d = _desc(abi, t, B, H, W, fids, flags | abi.FLAG_NO_GRAD)
    n = abi.lib.vsl_loss_workspace_bytes(ctypes.byref(d))
    assert n <= abi.STATIC_SCRATCH_BYTES
    ws = torch.empty(n, dtype=torch.uint8, device=DEV)
    l_ref, s_ref = torch.zeros(abi.LOSS_SLOTS, device=DEV), torch.full((B, H, W), -1.0, device=DEV)
    abi.check(abi.lib.vsl_loss_forward(ctypes.byref(d), ws.data_ptr(), n, l_ref.data_ptr(), s_ref.data_ptr(), stream))
    # the variant under test gets the caller's flags as they are (it forces forward-only itself)
    d2 = _desc(abi, t, B, H, W, fids, flags)
    l_new, s_new = torch.zeros(abi.LOSS_SLOTS, device=DEV), torch.full((B, H, W), -1.0, device=DEV)
    abi.check(abi.lib.vsl_loss_forward_static(ctypes.byref(d2), l_new.data_ptr(), s_new.data_ptr(), stream))
    torch.cuda.synchronize()
    return data, l_ref.cpu().numpy(), s_ref.cpu().numpy(), l_new.cpu().numpy(), s_new.cpu().numpy()


def test_static_scratch_forward_matches_workspace_forward_twice():
    import mde_amd._abi as abi
    from oracle import vsl_oracle as O
    fids = [0, -1, 1]
    # first call: the largest benchmark shape that fits the scratch array; second call: a smaller, ragged one
    for (B, H, W, seed) in ((2, 192, 640, 3), (1, 37, 75, 4)):
        data, l_ref, s_ref, l_new, s_new = _both(abi, B, H, W, fids, seed, abi.FLAG_AUTOMASK)
        assert np.isfinite(l_new).all() and l_new[abi.LOSS_TOTAL] > 0
        assert np.array_equal(l_new, l_ref), (l_new, l_ref)
        assert np.array_equal(s_new, s_ref)
        assert set(np.unique(s_new)) <= {0.0, 1.0}
    # the small case once more against the CPU oracle (float32 tolerance of the loss value: 1e-5 relative)
    cfg = O.LossConfig(frame_ids=fids, height=H, width=W, automask=True, scales=[0, 1])
    inputs = {"K": data["K"], "inv_K": data["inv_K"]}
    for f in fids:
        inputs[("image", f)] = data[("image", f)].double()
    outputs = {("disp", 0): data[("disp", 0)].clone()}
    for f in fids[1:]:
        outputs[("cam_T_cam", f)] = data[("cam_T_cam", f)].clone()
    with torch.no_grad():
        O.view_synthesis(inputs, outputs, cfg)
        ref = O.compute_losses(inputs, outputs, cfg, data["noise"])
    want = float(ref["loss"])
    assert abs(float(l_new[abi.LOSS_TOTAL]) - want) <= 1e-5 * abs(want), (l_new, want)


def test_static_scratch_rejects_oversized_shapes():
    import mde_amd._abi as abi
    import mde_amd.synth as synth
    B, H, W, fids = 12, 192, 640, [0, -1, 1]
    data = synth.make_batch(1, 16, 32, fids, seed=1, n_heads=1)       # never read: the size check comes first
    t = {k: v.to(DEV) for k, v in data.items() if torch.is_tensor(v)}
    d = _desc(abi, t, B, H, W, fids, abi.FLAG_AUTOMASK)
    assert abi.lib.vsl_loss_workspace_bytes(ctypes.byref(d)) > abi.STATIC_SCRATCH_BYTES
    out = torch.zeros(abi.LOSS_SLOTS, device=DEV)
    rc = abi.lib.vsl_loss_forward_static(ctypes.byref(d), out.data_ptr(), None, None)
    assert rc == -2 and b"VSL_STATIC_SCRATCH_BYTES" in abi.lib.vsl_last_error_string()


def test_static_scratch_forward_inside_a_hip_graph():
    """The contract that gives the entry point its purpose: it allocates and frees nothing, so it can be captured.
    One call outside the capture first (the library looks up its scratch array's address once per device), then the
    call is captured into a HIP graph on a side stream -- a capture fails on any allocating call -- and replayed twice;
    every replay must reproduce the eager result bit for bit, also after the outputs were overwritten."""
    import mde_amd._abi as abi
    import mde_amd.synth as synth
    B, H, W, fids = 2, 64, 160, [0, -1, 1]
    data = synth.make_batch(B, H, W, fids, seed=9, n_heads=1)
    t = {k: v.to(DEV) for k, v in data.items() if torch.is_tensor(v)}
    d = _desc(abi, t, B, H, W, fids, abi.FLAG_AUTOMASK)
    losses, sel = torch.zeros(abi.LOSS_SLOTS, device=DEV), torch.full((B, H, W), -1.0, device=DEV)
    abi.check(abi.lib.vsl_loss_forward_static(ctypes.byref(d), losses.data_ptr(), sel.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    want_l, want_s = losses.clone(), sel.clone()
    assert want_l[abi.LOSS_TOTAL].item() > 0

    side = torch.cuda.Stream(DEV)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        abi.check(abi.lib.vsl_loss_forward_static(ctypes.byref(d), losses.data_ptr(), sel.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream))
    for _ in range(2):
        losses.zero_()
        sel.fill_(-1.0)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(losses, want_l) and torch.equal(sel, want_s)
