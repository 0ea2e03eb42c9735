"""CPU-side checks of the boundary: the shared library loads, exports every symbol the
header declares, the ctypes mirror matches the C struct, and argument validation works
without touching a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import mde_amd._abi as abi
    declared = set()
    for name in ("vsloss.h", "vsloss_debug.h"):
        declared |= set(re.findall(r"\b(vsl_[a-z_]+)\s*\(", open(os.path.join(ROOT, "include", name)).read()))
    boundary = open(os.path.join(ROOT, "include", "vsloss.h")).read()
    assert "vsl_debug" not in boundary          # diagnostics live in their own header
    assert declared == set(abi.EXPORTS)
    for name in declared:
        assert hasattr(abi.lib, name), name
    assert abi.lib.vsl_version() >= 210


def test_descriptor_mirror_and_workspace_query():
    import mde_amd._abi as abi
    d = abi.LossDesc()
    assert ctypes.sizeof(d) == 208
    assert ctypes.sizeof(abi.Head) == 200
    d.B, d.H, d.W, d.S = 12, 192, 640, 2
    n = abi.lib.vsl_loss_workspace_bytes(ctypes.byref(d))
    assert n >= 2 * 12 * 192 * 640 * 4 and n % 256 == 0
    d.B = 0
    assert abi.lib.vsl_loss_workspace_bytes(ctypes.byref(d)) == 0


def test_argument_validation_without_gpu():
    import mde_amd._abi as abi
    d = abi.LossDesc()
    d.B, d.H, d.W, d.S = 1, 1, 8, 1          # H < 2
    rc = abi.lib.vsl_loss_forward(ctypes.byref(d), None, 0, None, None, None)
    assert rc == -1 and b"H >= 2" in abi.lib.vsl_last_error_string()
    d.H, d.S = 8, 9                            # too many sources
    assert abi.lib.vsl_loss_forward(ctypes.byref(d), None, 0, None, None, None) == -1
    d.S, d.n_scales, d.min_depth, d.max_depth = 1, 1, 0.1, 100.0
    assert abi.lib.vsl_loss_forward(ctypes.byref(d), None, 0, None, None, None) == -1   # NULL tensors
    d.H, d.W = 2, 1 << 24                      # a row index times W must fit the 24-bit multiply of the kernels
    rc = abi.lib.vsl_loss_forward(ctypes.byref(d), None, 0, None, None, None)
    assert rc == -1 and b"2^24" in abi.lib.vsl_last_error_string()
    d.H, d.W = 1 << 14, 1 << 14                # H*W = 2^28: offsets within a plane no longer fit 28 bits
    rc = abi.lib.vsl_loss_forward(ctypes.byref(d), None, 0, None, None, None)
    assert rc == -1 and b"2^28" in abi.lib.vsl_last_error_string()
    with pytest.raises(abi.VslError):
        abi.check(-1)


def test_heads_calls_reject_a_caller_built_against_another_struct_layout():
    """vsl_head was 128 bytes before it carried the optional outputs: a caller compiled against that header must get an
    error from the size-checked calls, not have bytes 128..199 of its table read as output pointers."""
    import mde_amd._abi as abi
    d = abi.LossDesc()
    d.B, d.H, d.W, d.S = 1, 8, 8, 1
    heads = (abi.Head * 1)()
    for fn in (abi.lib.vsl_loss_forward_heads_checked, abi.lib.vsl_loss_backward_heads_checked):
        rc = fn(ctypes.byref(d), ctypes.sizeof(abi.LossDesc), heads, 128, 1, None)
        assert rc == -1 and b"recompile" in abi.lib.vsl_last_error_string()
        rc = fn(ctypes.byref(d), 200, heads, ctypes.sizeof(abi.Head), 1, None)
        assert rc == -1 and b"structure sizes differ" in abi.lib.vsl_last_error_string()
    # with matching sizes the call goes on to the ordinary argument checks (NULL tensors here)
    rc = abi.forward_heads(d, heads, 1, None)
    assert rc == -1 and b"structure sizes" not in abi.lib.vsl_last_error_string()


def test_inv_num_scales_and_identity_min_limits_without_gpu():
    """A zeroed inv_num_scales would silently zero every loss: rejected.  vsl_identity_min applies the same shape
    limits as the loss entry points."""
    import mde_amd._abi as abi
    fake = 0x1000
    err = lambda: abi.lib.vsl_last_error_string()
    d = abi.LossDesc()
    d.B, d.H, d.W, d.S, d.n_scales, d.min_depth, d.max_depth = 1, 8, 8, 1, 1, 0.1, 100.0
    d.target = d.K = d.inv_K = d.disp = fake
    d.source[0] = d.T[0] = fake
    for bad in (0.0, -1.0, float("inf"), float("nan")):
        d.inv_num_scales = bad
        assert abi.lib.vsl_loss_forward(ctypes.byref(d), fake, 1 << 30, fake, None, None) == -1
        assert b"inv_num_scales" in err()
    d.inv_num_scales = 1.0
    d.H, d.W = 2, 1 << 24
    assert abi.lib.vsl_identity_min(ctypes.byref(d), fake, None) == -1 and b"2^24" in err()
    d.H, d.W = 1 << 14, 1 << 14
    assert abi.lib.vsl_identity_min(ctypes.byref(d), fake, None) == -1 and b"2^28" in err()
    d.B, d.H, d.W = 1 << 12, 1 << 10, 1 << 10
    assert abi.lib.vsl_identity_min(ctypes.byref(d), fake, None) == -1 and b"2^31" in err()
    d.B, d.H, d.W, d.S = 1, 8, 8, 5
    assert abi.lib.vsl_identity_min(ctypes.byref(d), fake, None) == -1
    d.S = 1
    assert abi.lib.vsl_identity_min(ctypes.byref(d), None, None) == -1 and b"NULL" in err()
    assert abi.lib.vsl_identity_min(None, fake, None) == -1


def test_no_kernel_uses_scratch_memory():
    """The build records the compiler's per-kernel resources; a private array spilled to scratch memory is written
    to HBM by every wave, so none is allowed (the build itself refuses such a library)."""
    path = os.path.join(ROOT, "monocular-depth-estimation_amd", "libvsloss.resources.txt")
    if not os.path.exists(path):
        pytest.skip("library was not built by build.py in this tree")
    rows = [ln.split() for ln in open(path) if not ln.startswith("#")]
    assert len(rows) > 40
    assert [r[0] for r in rows if int(r[3]) or int(r[4])] == []


def test_head_validation_without_gpu():
    """vsl_loss_forward_heads / _backward_heads reject malformed head tables before anything is launched."""
    import mde_amd._abi as abi
    d = abi.LossDesc()
    d.B, d.H, d.W, d.S = 2, 16, 32, 1
    d.min_depth, d.max_depth, d.inv_num_scales = 0.1, 100.0, 1.0
    fake = 0x1000                                   # never dereferenced: validation fails first
    d.target = d.K = d.inv_K = fake
    d.source[0] = d.T[0] = fake
    need = abi.lib.vsl_loss_workspace_bytes(ctypes.byref(d))
    heads = (abi.Head * 2)()

    def fill(h, ws=fake * 16, nbytes=need, n_scales=1):
        h.disp, h.losses, h.workspace, h.workspace_bytes, h.n_scales = fake, fake, ws, nbytes, n_scales
        h.scale_weight[0] = 1e-3

    for h in heads:
        fill(h)
    err = lambda: abi.lib.vsl_last_error_string()
    assert abi.lib.vsl_loss_forward_heads(ctypes.byref(d), heads, 0, None) == -1 and b"heads" in err()
    assert abi.lib.vsl_loss_forward_heads(ctypes.byref(d), heads, abi.MAX_HEADS + 1, None) == -1
    fill(heads[1], nbytes=need - 1)
    assert abi.lib.vsl_loss_forward_heads(ctypes.byref(d), heads, 2, None) == -2 and b"workspace" in err()
    fill(heads[1], ws=fake * 16 + 8)
    assert abi.lib.vsl_loss_forward_heads(ctypes.byref(d), heads, 2, None) == -1 and b"aligned" in err()
    fill(heads[1], n_scales=9)
    assert abi.lib.vsl_loss_forward_heads(ctypes.byref(d), heads, 2, None) == -1 and b"n_scales" in err()
    fill(heads[1])
    heads[1].disp = None
    assert abi.lib.vsl_loss_forward_heads(ctypes.byref(d), heads, 2, None) == -1
    d.flags = abi.FLAG_NO_GRAD
    fill(heads[1])
    assert abi.lib.vsl_loss_backward_heads(ctypes.byref(d), heads, 2, None) == -1 and b"NO_GRAD" in err()
    assert abi.lib.vsl_loss_forward_heads(None, heads, 2, None) == -1


def test_pose_descriptor_validation_without_gpu():
    """vsl_pose_fwd / vsl_pose_bwd / vsl_mean_inv_depth_* reject malformed arguments before any launch."""
    import mde_amd._abi as abi
    assert ctypes.sizeof(abi.PoseDesc) == 96
    fake = 0x1000
    d = abi.PoseDesc()
    tab = (ctypes.c_void_p * abi.MAX_SOURCES)(fake, fake, fake, fake)
    err = lambda: abi.lib.vsl_last_error_string()
    d.B, d.n_src, d.aa_stride, d.tr_stride = 2, 0, 3, 3
    assert abi.lib.vsl_pose_fwd(ctypes.byref(d), tab, None) == -1 and b"n_src" in err()
    d.n_src = 5
    assert abi.lib.vsl_pose_fwd(ctypes.byref(d), tab, None) == -1
    d.n_src, d.aa_stride = 2, 2
    assert abi.lib.vsl_pose_fwd(ctypes.byref(d), tab, None) == -1 and b"stride" in err()
    d.aa_stride = 6
    assert abi.lib.vsl_pose_fwd(ctypes.byref(d), tab, None) == -1 and b"null" in err()      # parameter pointers unset
    d.axisangle[0] = d.axisangle[1] = d.translation[0] = d.translation[1] = fake
    assert abi.lib.vsl_pose_fwd(ctypes.byref(d), None, None) == -1
    assert abi.lib.vsl_pose_bwd(ctypes.byref(d), None, tab, tab, None, None) == -1
    d.B = 0
    assert abi.lib.vsl_pose_fwd(ctypes.byref(d), tab, None) == -1
    assert abi.lib.vsl_pose_fwd(None, tab, None) == -1
    assert abi.lib.vsl_mean_inv_depth_workspace_bytes(12) == 12 * 32 * 8
    assert abi.lib.vsl_mean_inv_depth_fwd(fake, fake, fake, 8, 12, 192, 640, 0.1, 100.0, None) == -2
    assert abi.lib.vsl_mean_inv_depth_fwd(fake, fake, fake, 1 << 20, 12, 0, 640, 0.1, 100.0, None) == -1
    assert abi.lib.vsl_mean_inv_depth_bwd(fake, None, fake, 0, 12, 192, 640, 0.1, 100.0, None) == -1


def test_warp_descriptor_validation_without_gpu():
    import mde_amd._abi as abi
    assert ctypes.sizeof(abi.WarpDesc) == 72
    fake = 0x1000
    d = abi.WarpDesc()
    err = lambda: abi.lib.vsl_last_error_string()
    assert abi.lib.vsl_warp_features_workspace_bytes(ctypes.byref(d)) == 0
    d.B, d.C, d.H, d.W = 12, 128, 24, 80
    d.min_depth, d.max_depth, d.eps = 0.1, 100.0, 1e-7
    need = abi.lib.vsl_warp_features_workspace_bytes(ctypes.byref(d))
    assert need == (12 * 16 * 1920 * 2 + 12 * 8 * 12) * 4
    assert abi.lib.vsl_warp_features_fwd(ctypes.byref(d), fake, None, None) == -1 and b"NULL" in err()
    d.feats = d.disp = d.K = d.inv_K = d.T = fake
    assert abi.lib.vsl_warp_features_fwd(ctypes.byref(d), None, None, None) == -1
    assert abi.lib.vsl_warp_features_bwd(ctypes.byref(d), fake, None, None, None, fake, need - 1, None) == -2
    assert abi.lib.vsl_warp_features_bwd(ctypes.byref(d), fake, None, None, None, fake + 4, need, None) == -1
    d.W = 1
    assert abi.lib.vsl_warp_features_fwd(ctypes.byref(d), fake, None, None) == -1
    d.W, d.max_depth = 80, 0.05
    assert abi.lib.vsl_warp_features_fwd(ctypes.byref(d), fake, None, None) == -1 and b"depth" in err()


def test_depth_metrics_descriptor_validation_without_gpu():
    import mde_amd._abi as abi
    assert ctypes.sizeof(abi.DepthMetricsDesc) == 64
    fake = 0x1000
    d = abi.DepthMetricsDesc()
    err = lambda: abi.lib.vsl_last_error_string()
    assert abi.lib.vsl_depth_metrics_workspace_bytes(ctypes.byref(d)) == 0
    d.B, d.H, d.W, d.Hg, d.Wg = 12, 192, 640, 375, 1242
    d.crop_y0, d.crop_y1, d.crop_x0, d.crop_x1 = 153, 371, 44, 1197
    need = abi.lib.vsl_depth_metrics_workspace_bytes(ctypes.byref(d))
    assert need >= 12 * 218 * 1153 * 4 and need % 256 == 0
    assert abi.lib.vsl_depth_metrics(ctypes.byref(d), fake, fake * 16, need, None) == -1 and b"clamp" in err()
    d.clamp_min, d.clamp_max = 1e-3, 80.0
    assert abi.lib.vsl_depth_metrics(ctypes.byref(d), fake, fake * 16, need, None) == -1 and b"NULL" in err()
    d.depth = d.depth_gt = fake
    assert abi.lib.vsl_depth_metrics(ctypes.byref(d), fake, fake * 16, need - 1, None) == -2
    assert abi.lib.vsl_depth_metrics(ctypes.byref(d), fake, fake * 16 + 8, need, None) == -1 and b"aligned" in err()
    assert abi.lib.vsl_depth_metrics(ctypes.byref(d), None, fake * 16, need, None) == -1
