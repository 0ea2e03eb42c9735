"""ctypes binding of libvsloss.so -- the C-ABI boundary declared in include/vsloss.h.

The library is hand-written HIP for gfx950 and there is no other implementation
behind these functions: if the shared object is missing or fails to load, the
import raises, it never falls back to a CPU path.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VSL_LIBRARY", os.path.join(_HERE, "libvsloss.so"))  # override: A/B builds of the same ABI

MAX_SOURCES = 4
MAX_SCALES = 8
MAX_HEADS = 8
FLAG_AUTOMASK, FLAG_AVG_REPROJECTION, FLAG_NO_SSIM, FLAG_NO_GRAD = 1, 2, 4, 8
HEAD_NO_GRAD = 1
LOSS_TOTAL, LOSS_PHOTO, LOSS_SMOOTH, LOSS_SCALE0 = 0, 1, 2, 3
LOSS_SLOTS = 3 + MAX_SCALES
STATIC_SCRATCH_BYTES = 4 << 20

MIN_VERSION = 210
EXPORTS = ("vsl_loss_workspace_bytes", "vsl_loss_forward", "vsl_loss_backward", "vsl_identity_min",
           "vsl_loss_forward_heads", "vsl_loss_backward_heads", "vsl_loss_forward_heads_checked",
           "vsl_loss_backward_heads_checked", "vsl_loss_forward_static",
           "vsl_disp_to_depth_fwd", "vsl_disp_to_depth_bwd", "vsl_backproject_fwd", "vsl_backproject_bwd",
           "vsl_project_fwd", "vsl_project_bwd_workspace_bytes", "vsl_project_bwd",
           "vsl_grid_sample_fwd", "vsl_grid_sample_bwd", "vsl_ssim_fwd", "vsl_ssim_bwd",
           "vsl_smooth_workspace_bytes", "vsl_smooth_fwd", "vsl_smooth_bwd",
           "vsl_pose_fwd", "vsl_pose_bwd", "vsl_pose_fwd_posecnn", "vsl_depth_metrics_workspace_bytes", "vsl_depth_metrics",
           "vsl_warp_features_workspace_bytes", "vsl_warp_features_fwd", "vsl_warp_features_bwd",
           "vsl_warp_multi_workspace_bytes", "vsl_warp_multi_fwd", "vsl_warp_multi_bwd",
           "vsl_affinity_loss_workspace_bytes", "vsl_affinity_loss_fwd", "vsl_affinity_loss_bwd",
           "vsl_mean_inv_depth_workspace_bytes", "vsl_mean_inv_depth_fwd", "vsl_mean_inv_depth_bwd",
           "vsl_debug_fused_occupancy", "vsl_debug_quot", "vsl_debug_time_fused", "vsl_last_error_string", "vsl_version")


class LossDesc(ctypes.Structure):
    """Mirror of ``vsl_loss_desc`` (include/vsloss.h)."""
    _fields_ = [
        ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("S", ctypes.c_int32),
        ("flags", ctypes.c_uint32),
        ("min_depth", ctypes.c_float), ("max_depth", ctypes.c_float), ("eps", ctypes.c_float),
        ("ssim_weight", ctypes.c_float), ("l1_weight", ctypes.c_float), ("warp_scale", ctypes.c_float),
        ("n_scales", ctypes.c_int32), ("scale_weight", ctypes.c_float * MAX_SCALES),
        ("inv_num_scales", ctypes.c_float),
        ("disp", ctypes.c_void_p), ("target", ctypes.c_void_p), ("source", ctypes.c_void_p * MAX_SOURCES),
        ("K", ctypes.c_void_p), ("inv_K", ctypes.c_void_p), ("T", ctypes.c_void_p * MAX_SOURCES),
        ("noise", ctypes.c_void_p), ("seed", ctypes.c_uint64), ("identity_min", ctypes.c_void_p),
    ]


class Head(ctypes.Structure):
    """Mirror of ``vsl_head`` (include/vsloss.h)."""
    _fields_ = [
        ("disp", ctypes.c_void_p), ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
        ("losses", ctypes.c_void_p), ("identity_selection", ctypes.c_void_p),
        ("n_scales", ctypes.c_int32), ("scale_weight", ctypes.c_float * MAX_SCALES), ("flags", ctypes.c_uint32),
        ("grad_losses", ctypes.c_void_p), ("grad_disp", ctypes.c_void_p), ("grad_T", ctypes.c_void_p * MAX_SOURCES),
        ("depth", ctypes.c_void_p), ("sample", ctypes.c_void_p * MAX_SOURCES), ("color", ctypes.c_void_p * MAX_SOURCES),
    ]


class PoseDesc(ctypes.Structure):
    """Mirror of ``vsl_pose_desc`` (include/vsloss.h)."""
    _fields_ = [
        ("B", ctypes.c_int32), ("n_src", ctypes.c_int32), ("aa_stride", ctypes.c_int32), ("tr_stride", ctypes.c_int32),
        ("invert_mask", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
        ("axisangle", ctypes.c_void_p * MAX_SOURCES), ("translation", ctypes.c_void_p * MAX_SOURCES),
        ("t_scale", ctypes.c_void_p),
    ]


class WarpDesc(ctypes.Structure):
    """Mirror of ``vsl_warp_desc`` (include/vsloss.h)."""
    _fields_ = [
        ("B", ctypes.c_int32), ("C", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
        ("min_depth", ctypes.c_float), ("max_depth", ctypes.c_float), ("eps", ctypes.c_float), ("reserved", ctypes.c_uint32),
        ("feats", ctypes.c_void_p), ("disp", ctypes.c_void_p), ("K", ctypes.c_void_p), ("inv_K", ctypes.c_void_p),
        ("T", ctypes.c_void_p),
    ]


MAX_WARP_TENSORS = 4


class WarpMultiDesc(ctypes.Structure):
    """Mirror of ``vsl_warp_multi_desc`` (include/vsloss.h)."""
    _fields_ = [
        ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("n", ctypes.c_int32),
        ("min_depth", ctypes.c_float), ("max_depth", ctypes.c_float), ("eps", ctypes.c_float), ("reserved", ctypes.c_uint32),
        ("C", ctypes.c_int32 * MAX_WARP_TENSORS), ("feats", ctypes.c_void_p * MAX_WARP_TENSORS),
        ("disp", ctypes.c_void_p), ("K", ctypes.c_void_p), ("inv_K", ctypes.c_void_p), ("T", ctypes.c_void_p),
    ]


METRIC_NAMES = ("abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3")
METRIC_SUM, METRIC_COUNT, METRIC_MEDIAN_GT, METRIC_MEDIAN_PRED, METRIC_SLOTS = 7, 8, 9, 10, 11


class DepthMetricsDesc(ctypes.Structure):
    """Mirror of ``vsl_depth_metrics_desc`` (include/vsloss.h)."""
    _fields_ = [
        ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("Hg", ctypes.c_int32), ("Wg", ctypes.c_int32),
        ("crop_y0", ctypes.c_int32), ("crop_y1", ctypes.c_int32), ("crop_x0", ctypes.c_int32), ("crop_x1", ctypes.c_int32),
        ("clamp_min", ctypes.c_float), ("clamp_max", ctypes.c_float), ("reserved", ctypes.c_uint32),
        ("depth", ctypes.c_void_p), ("depth_gt", ctypes.c_void_p),
    ]


class VslError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libvsloss.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950) -- there is no fallback implementation")
    lib = ctypes.CDLL(LIB_PATH)
    lib.vsl_loss_workspace_bytes.restype = ctypes.c_size_t
    lib.vsl_loss_workspace_bytes.argtypes = [ctypes.POINTER(LossDesc)]
    lib.vsl_loss_forward.restype = ctypes.c_int
    lib.vsl_loss_forward.argtypes = [ctypes.POINTER(LossDesc), ctypes.c_void_p, ctypes.c_size_t,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.vsl_loss_backward.restype = ctypes.c_int
    lib.vsl_loss_backward.argtypes = [ctypes.POINTER(LossDesc), ctypes.c_void_p, ctypes.c_size_t,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p),
                                      ctypes.c_void_p]
    lib.vsl_loss_forward_static.restype = ctypes.c_int
    lib.vsl_loss_forward_static.argtypes = [ctypes.POINTER(LossDesc), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    for name in ("vsl_loss_forward_heads", "vsl_loss_backward_heads"):
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.POINTER(LossDesc), ctypes.POINTER(Head), ctypes.c_int, ctypes.c_void_p]
    for name in ("vsl_loss_forward_heads_checked", "vsl_loss_backward_heads_checked"):
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.POINTER(LossDesc), ctypes.c_size_t, ctypes.POINTER(Head), ctypes.c_size_t, ctypes.c_int,
                       ctypes.c_void_p]
    lib.vsl_debug_time_fused.restype = ctypes.c_int
    lib.vsl_debug_time_fused.argtypes = [ctypes.POINTER(LossDesc), ctypes.POINTER(Head), ctypes.c_int, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_float), ctypes.c_void_p]
    lib.vsl_identity_min.restype = ctypes.c_int
    lib.vsl_identity_min.argtypes = [ctypes.POINTER(LossDesc), ctypes.c_void_p, ctypes.c_void_p]
    P, I, F, LL, SZ = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong, ctypes.c_size_t
    sigs = {
        "vsl_disp_to_depth_fwd": [P, P, P, LL, F, F, P],
        "vsl_disp_to_depth_bwd": [P, P, P, P, LL, F, F, P],
        "vsl_backproject_fwd": [P, P, P, I, I, I, P],
        "vsl_backproject_bwd": [P, P, P, I, I, I, P],
        "vsl_project_fwd": [P, P, P, P, I, I, I, F, P],
        "vsl_project_bwd": [P, P, P, P, P, P, P, P, SZ, I, I, I, F, P],
        "vsl_grid_sample_fwd": [P, P, P, I, I, I, I, I, I, P],
        "vsl_grid_sample_bwd": [P, P, P, P, P, I, I, I, I, I, I, P],
        "vsl_ssim_fwd": [P, P, P, I, I, I, P],
        "vsl_ssim_bwd": [P, P, P, P, I, I, I, P],
        "vsl_smooth_fwd": [P, P, P, P, SZ, I, I, I, I, P],
        "vsl_smooth_bwd": [P, P, P, P, I, I, I, I, P],
        "vsl_debug_fused_occupancy": [I, I],
        "vsl_debug_quot": [P, P, P, LL, P],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int
        fn.argtypes = args
    for name in ("vsl_project_bwd_workspace_bytes", "vsl_smooth_workspace_bytes"):
        fn = getattr(lib, name)
        fn.restype = ctypes.c_size_t
        fn.argtypes = [I, I, I]
    PT = ctypes.POINTER(ctypes.c_void_p)
    lib.vsl_pose_fwd.restype = ctypes.c_int
    lib.vsl_pose_fwd.argtypes = [ctypes.POINTER(PoseDesc), PT, P]
    lib.vsl_pose_fwd_posecnn.restype = ctypes.c_int
    lib.vsl_pose_fwd_posecnn.argtypes = [ctypes.POINTER(PoseDesc), P, I, I, F, F, P, PT, P, SZ, P]
    lib.vsl_pose_bwd.restype = ctypes.c_int
    lib.vsl_pose_bwd.argtypes = [ctypes.POINTER(PoseDesc), PT, PT, PT, P, P]
    lib.vsl_warp_features_workspace_bytes.restype = ctypes.c_size_t
    lib.vsl_warp_features_workspace_bytes.argtypes = [ctypes.POINTER(WarpDesc)]
    lib.vsl_warp_features_fwd.restype = ctypes.c_int
    lib.vsl_warp_features_fwd.argtypes = [ctypes.POINTER(WarpDesc), P, P, P]
    lib.vsl_warp_features_bwd.restype = ctypes.c_int
    lib.vsl_warp_features_bwd.argtypes = [ctypes.POINTER(WarpDesc), P, P, P, P, P, SZ, P]
    lib.vsl_warp_multi_workspace_bytes.restype = ctypes.c_size_t
    lib.vsl_warp_multi_workspace_bytes.argtypes = [ctypes.POINTER(WarpMultiDesc)]
    lib.vsl_warp_multi_fwd.restype = ctypes.c_int
    lib.vsl_warp_multi_fwd.argtypes = [ctypes.POINTER(WarpMultiDesc), PT, P, P]
    lib.vsl_warp_multi_bwd.restype = ctypes.c_int
    lib.vsl_warp_multi_bwd.argtypes = [ctypes.POINTER(WarpMultiDesc), PT, PT, P, P, P, SZ, P]
    lib.vsl_affinity_loss_workspace_bytes.restype = ctypes.c_size_t
    lib.vsl_affinity_loss_workspace_bytes.argtypes = [LL]
    lib.vsl_affinity_loss_fwd.restype = ctypes.c_int
    lib.vsl_affinity_loss_fwd.argtypes = [P, P, P, F, LL, P, P, SZ, P]
    lib.vsl_affinity_loss_bwd.restype = ctypes.c_int
    lib.vsl_affinity_loss_bwd.argtypes = [P, P, P, F, LL, P, P, P]
    lib.vsl_depth_metrics_workspace_bytes.restype = ctypes.c_size_t
    lib.vsl_depth_metrics_workspace_bytes.argtypes = [ctypes.POINTER(DepthMetricsDesc)]
    lib.vsl_depth_metrics.restype = ctypes.c_int
    lib.vsl_depth_metrics.argtypes = [ctypes.POINTER(DepthMetricsDesc), P, P, SZ, P]
    lib.vsl_mean_inv_depth_workspace_bytes.restype = ctypes.c_size_t
    lib.vsl_mean_inv_depth_workspace_bytes.argtypes = [I]
    lib.vsl_mean_inv_depth_fwd.restype = ctypes.c_int
    lib.vsl_mean_inv_depth_fwd.argtypes = [P, P, P, SZ, I, I, I, F, F, P]
    lib.vsl_mean_inv_depth_bwd.restype = ctypes.c_int
    lib.vsl_mean_inv_depth_bwd.argtypes = [P, P, P, I, I, I, I, F, F, P]
    lib.vsl_last_error_string.restype = ctypes.c_char_p
    lib.vsl_last_error_string.argtypes = []
    lib.vsl_version.restype = ctypes.c_int
    lib.vsl_version.argtypes = []
    if lib.vsl_version() < MIN_VERSION:  # an older build lays vsl_head out differently
        raise ImportError("libvsloss.so is version %d, this binding needs >= %d: rebuild it" % (lib.vsl_version(), MIN_VERSION))
    return lib


lib = _load()


def forward_heads(desc, heads, n_heads, stream) -> int:
    """vsl_loss_forward_heads with this binding's structure sizes checked against the library's."""
    return lib.vsl_loss_forward_heads_checked(ctypes.byref(desc), ctypes.sizeof(LossDesc), heads, ctypes.sizeof(Head),
                                              n_heads, stream)


def backward_heads(desc, heads, n_heads, stream) -> int:
    return lib.vsl_loss_backward_heads_checked(ctypes.byref(desc), ctypes.sizeof(LossDesc), heads, ctypes.sizeof(Head),
                                               n_heads, stream)


def check(rc: int) -> None:
    if rc != 0:
        raise VslError("libvsloss error %d: %s" % (rc, lib.vsl_last_error_string().decode()))
