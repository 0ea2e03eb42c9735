// libvsloss.so -- pose preparation in front of the view-synthesis loss: (axis-angle, translation) -> 4x4
// camera motion for every source frame of a batch in ONE launch, forward and backward
// (Python/layers.py:28-103), and the per-image mean inverse depth that the "posecnn" pose model
// multiplies the translation with (Python/Trainer.py:446-455).  In eager PyTorch this is about
// twenty [B,4,4]-sized kernels per source frame and direction; here it is one launch each way.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/vsloss.h"

namespace vsl_err {
int fail(int code, const std::string& msg);  // defined in vsl_loss.hip
}

namespace {

using vsl_err::fail;

static_assert(sizeof(vsl_pose_desc) == 96, "vsl_pose_desc layout is part of the ABI");

int launch_status() {
  if (hipError_t e = hipGetLastError(); e != hipSuccess)
    return fail(VSL_ERR_LAUNCH_FAILED, std::string("kernel launch failed: ") + hipGetErrorString(e));
  return VSL_OK;
}

struct PoseArgs {
  const float* axisangle[VSL_MAX_SOURCES];
  const float* translation[VSL_MAX_SOURCES];
  const float* t_scale;  // [B] or null
  int B, n_src, aa_stride, tr_stride;
  unsigned invert_mask;
  // posecnn form: the translation scale is the image's mean inverse depth, of which only the per-chunk partial sums
  // exist yet ([B][kMidChunks] doubles); the pose launch finishes the mean itself and leaves it in t_scale_out
  const double* mid_partial;
  float* t_scale_out;
  int HW;
};

struct PoseOut {
  float* T[VSL_MAX_SOURCES];
};

struct PoseGrad {
  const float* g_T[VSL_MAX_SOURCES];
  float* g_axisangle[VSL_MAX_SOURCES];
  float* g_translation[VSL_MAX_SOURCES];
  float* g_t_scale;  // [B] or null; sum over the sources
};

// Rotation of layers.py:64-103 in the reference's operation order; also returns what the backward needs.
struct Rot {
  float R[3][3];
  float x, y, z, ca, sa, C, angle;
};

__device__ __forceinline__ Rot rotation(const float v[3]) {
  Rot o;
  o.angle = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);  // torch.norm(vec, 2, 2, True), layers.py:69
  const float den = o.angle + 1e-7f;                            // layers.py:70
  o.x = v[0] / den;
  o.y = v[1] / den;
  o.z = v[2] / den;
  o.ca = cosf(o.angle);
  o.sa = sinf(o.angle);
  o.C = 1.0f - o.ca;
  const float xs = o.x * o.sa, ys = o.y * o.sa, zs = o.z * o.sa;
  const float xC = o.x * o.C, yC = o.y * o.C, zC = o.z * o.C;
  const float xyC = o.x * yC, yzC = o.y * zC, zxC = o.z * xC;
  o.R[0][0] = o.x * xC + o.ca;
  o.R[0][1] = xyC - zs;
  o.R[0][2] = zxC + ys;
  o.R[1][0] = xyC + zs;
  o.R[1][1] = o.y * yC + o.ca;
  o.R[1][2] = yzC - xs;
  o.R[2][0] = zxC - ys;
  o.R[2][1] = yzC + xs;
  o.R[2][2] = o.z * zC + o.ca;
  return o;
}

constexpr int kMidChunks = 32;   // partial sums of the mean inverse depth per image

// one thread per (source, image)
__global__ void k_pose_fwd(PoseArgs a, PoseOut out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.B * a.n_src) return;
  const int s = i / a.B, b = i % a.B;
  const float* aa = nullptr;
  const float* tr = nullptr;
  float* T = nullptr;
#pragma unroll
  for (int k = 0; k < VSL_MAX_SOURCES; ++k)
    if (k == s) {
      aa = a.axisangle[k] + (size_t)b * a.aa_stride;
      tr = a.translation[k] + (size_t)b * a.tr_stride;
      T = out.T[k] + (size_t)b * 16;
    }
  const float v[3] = {aa[0], aa[1], aa[2]};
  const Rot r = rotation(v);
  float sc = a.t_scale ? a.t_scale[b] : 1.0f;
  if (a.mid_partial) {  // the same sum, in the same order and rounding, as k_mid_final
    double m = 0.0;
    for (int i = 0; i < kMidChunks; ++i) m += a.mid_partial[(size_t)b * kMidChunks + i];
    sc = (float)(m / (double)a.HW);
    if (s == 0) a.t_scale_out[b] = sc;
  }
  float t[3] = {tr[0] * sc, tr[1] * sc, tr[2] * sc};  // Trainer.py:455
  float M[3][4];
  if ((a.invert_mask >> s) & 1u) {
    // layers.py:34-41: R^T @ translation(-t); the products with the zeros and ones of the 4x4 factors are exact
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int q = 0; q < 3; ++q) M[p][q] = r.R[q][p];
      M[p][3] = r.R[0][p] * -t[0] + r.R[1][p] * -t[1] + r.R[2][p] * -t[2];
    }
  } else {
    // layers.py:43: translation(t) @ R
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int q = 0; q < 3; ++q) M[p][q] = r.R[p][q];
      M[p][3] = t[p];
    }
  }
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) T[p * 4 + q] = M[p][q];
  T[12] = 0.0f;
  T[13] = 0.0f;
  T[14] = 0.0f;
  T[15] = 1.0f;
}

// Backward of k_pose_fwd.  g_t_scale sums over the sources, so one thread handles every source of its image.
__global__ void k_pose_bwd(PoseArgs a, PoseGrad g) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  float g_scale = 0.0f;
#pragma unroll
  for (int s = 0; s < VSL_MAX_SOURCES; ++s) {
    if (s >= a.n_src) break;
    const float* aa = a.axisangle[s] + (size_t)b * a.aa_stride;
    const float* tr = a.translation[s] + (size_t)b * a.tr_stride;
    const float* gT = g.g_T[s] + (size_t)b * 16;
    const float v[3] = {aa[0], aa[1], aa[2]};
    const Rot r = rotation(v);
    const float sc = a.t_scale ? a.t_scale[b] : 1.0f;
    const float tv[3] = {tr[0], tr[1], tr[2]};
    float gR[3][3], gt[3];  // gradients of R and of the scaled translation
    if ((a.invert_mask >> s) & 1u) {
      // M[:3,:3] = Q, M[:3,3] = -Q t with Q = R^T
      const float gm[3] = {gT[3], gT[7], gT[11]};
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q) gR[q][p] = gT[p * 4 + q] - gm[p] * (tv[q] * sc);
#pragma unroll
      for (int q = 0; q < 3; ++q) gt[q] = -(r.R[q][0] * gm[0] + r.R[q][1] * gm[1] + r.R[q][2] * gm[2]);
    } else {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int q = 0; q < 3; ++q) gR[p][q] = gT[p * 4 + q];
        gt[p] = gT[p * 4 + 3];
      }
    }
    if (g.g_translation[s]) {
      float* o = g.g_translation[s] + (size_t)b * 3;
#pragma unroll
      for (int q = 0; q < 3; ++q) o[q] = gt[q] * sc;
    }
    g_scale += gt[0] * tv[0] + gt[1] * tv[1] + gt[2] * tv[2];

    if (g.g_axisangle[s]) {
      const float x = r.x, y = r.y, z = r.z, C = r.C, sa = r.sa, ca = r.ca;
      const float sxy = gR[0][1] + gR[1][0], szx = gR[0][2] + gR[2][0], syz = gR[1][2] + gR[2][1];
      const float g_ca = gR[0][0] + gR[1][1] + gR[2][2];
      const float g_C = x * x * gR[0][0] + y * y * gR[1][1] + z * z * gR[2][2] + x * y * sxy + z * x * szx + y * z * syz;
      const float g_sa = z * (gR[1][0] - gR[0][1]) + y * (gR[0][2] - gR[2][0]) + x * (gR[2][1] - gR[1][2]);
      const float gax[3] = {2.0f * x * C * gR[0][0] + y * C * sxy + z * C * szx + sa * (gR[2][1] - gR[1][2]),
                            2.0f * y * C * gR[1][1] + x * C * sxy + z * C * syz + sa * (gR[0][2] - gR[2][0]),
                            2.0f * z * C * gR[2][2] + x * C * szx + y * C * syz + sa * (gR[1][0] - gR[0][1])};
      // C = 1 - cos, axis = vec / (angle + 1e-7), angle = |vec| (gradient 0 at vec = 0, as torch.norm gives)
      const float den = r.angle + 1e-7f;
      float g_angle = -sa * (g_ca - g_C) + ca * g_sa;
      g_angle -= (gax[0] * v[0] + gax[1] * v[1] + gax[2] * v[2]) / (den * den);
      const float k = r.angle > 0.0f ? g_angle / r.angle : 0.0f;
      float* o = g.g_axisangle[s] + (size_t)b * 3;
#pragma unroll
      for (int q = 0; q < 3; ++q) o[q] = gax[q] / den + k * v[q];
    }
  }
  if (g.g_t_scale) g.g_t_scale[b] = a.t_scale ? g_scale : 0.0f;
}

// ---------------------------------------------------------------------------
// mean inverse depth per image (Trainer.py:451-452): mean over the image of 1 / depth, depth = 1 / (a + c disp)
// ---------------------------------------------------------------------------
constexpr int kMidThreads = 256;

__device__ __forceinline__ double block_sum(double v, double* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) sh[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;  // valid in thread 0
}

__global__ void k_mid_partial(const float* __restrict__ disp, double* __restrict__ partial, int HW, float a, float c) {
  __shared__ double sh[kMidThreads / 64];
  const int b = blockIdx.y;
  const float* d = disp + (size_t)b * HW;
  const int per = (HW + kMidChunks - 1) / kMidChunks;
  const int lo = blockIdx.x * per, hi = min(HW, lo + per);
  double acc = 0.0;
  for (int p = lo + threadIdx.x; p < hi; p += blockDim.x) {
    const float depth = __fdiv_rn(1.0f, __fadd_rn(a, __fmul_rn(c, d[p])));  // layers.py:23-24
    acc += (double)__fdiv_rn(1.0f, depth);                                    // Trainer.py:451
  }
  const double t = block_sum(acc, sh);
  if (threadIdx.x == 0) partial[(size_t)b * kMidChunks + blockIdx.x] = t;
}

__global__ void k_mid_final(const double* __restrict__ partial, float* __restrict__ out, int B, int HW) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double t = 0.0;
  for (int i = 0; i < kMidChunks; ++i) t += partial[(size_t)b * kMidChunks + i];
  out[b] = (float)(t / (double)HW);
}

__global__ void k_mid_bwd(const float* __restrict__ disp, const float* __restrict__ g_out, float* __restrict__ g_disp, int HW,
                          float a, float c, int accumulate) {
  const int b = blockIdx.y;
  const float g = g_out[b] / (float)HW;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    const size_t i = (size_t)b * HW + p;
    // d (1/depth) / d disp through depth = 1/scaled: (1/depth^2) (1/scaled^2) c, the product the reference's autograd forms
    const float sd = a + c * disp[i];
    const float depth = 1.0f / sd;
    const float v = g * c / ((depth * depth) * (sd * sd));
    g_disp[i] = accumulate ? g_disp[i] + v : v;
  }
}

int check_pose(const vsl_pose_desc* d) {
  if (!d) return fail(VSL_ERR_INVALID_ARGUMENT, "pose descriptor is null");
  if (d->B < 1 || d->B > 65535 * 64) return fail(VSL_ERR_INVALID_ARGUMENT, "pose: B out of range");
  if (d->n_src < 1 || d->n_src > VSL_MAX_SOURCES) return fail(VSL_ERR_INVALID_ARGUMENT, "pose: n_src must be 1..4");
  if (d->aa_stride < 3 || d->tr_stride < 3) return fail(VSL_ERR_INVALID_ARGUMENT, "pose: row strides must be >= 3 floats");
  for (int s = 0; s < d->n_src; ++s)
    if (!d->axisangle[s] || !d->translation[s]) return fail(VSL_ERR_INVALID_ARGUMENT, "pose: null parameter pointer");
  return VSL_OK;
}

PoseArgs pose_args(const vsl_pose_desc* d) {
  PoseArgs a{};
  for (int s = 0; s < d->n_src; ++s) {
    a.axisangle[s] = d->axisangle[s];
    a.translation[s] = d->translation[s];
  }
  a.t_scale = d->t_scale;
  a.B = d->B;
  a.n_src = d->n_src;
  a.aa_stride = d->aa_stride;
  a.tr_stride = d->tr_stride;
  a.invert_mask = d->invert_mask;
  return a;
}

}  // namespace

extern "C" {

int vsl_pose_fwd(const vsl_pose_desc* d, float* const* T, void* stream) {
  if (int rc = check_pose(d)) return rc;
  if (!T) return fail(VSL_ERR_INVALID_ARGUMENT, "pose: output table is null");
  PoseOut o{};
  for (int s = 0; s < d->n_src; ++s) {
    if (!T[s]) return fail(VSL_ERR_INVALID_ARGUMENT, "pose: null output pointer");
    o.T[s] = T[s];
  }
  const int n = d->B * d->n_src;
  hipLaunchKernelGGL(k_pose_fwd, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, pose_args(d), o);
  return launch_status();
}

int vsl_pose_bwd(const vsl_pose_desc* d, const float* const* g_T, float* const* g_axisangle, float* const* g_translation,
                 float* g_t_scale, void* stream) {
  if (int rc = check_pose(d)) return rc;
  if (!g_T) return fail(VSL_ERR_INVALID_ARGUMENT, "pose: gradient table is null");
  PoseGrad g{};
  for (int s = 0; s < d->n_src; ++s) {
    if (!g_T[s]) return fail(VSL_ERR_INVALID_ARGUMENT, "pose: null upstream gradient");
    g.g_T[s] = g_T[s];
    g.g_axisangle[s] = g_axisangle ? g_axisangle[s] : nullptr;
    g.g_translation[s] = g_translation ? g_translation[s] : nullptr;
  }
  g.g_t_scale = g_t_scale;
  hipLaunchKernelGGL(k_pose_bwd, dim3((d->B + 63) / 64), dim3(64), 0, (hipStream_t)stream, pose_args(d), g);
  return launch_status();
}

size_t vsl_mean_inv_depth_workspace_bytes(int B) { return B > 0 ? (size_t)B * kMidChunks * sizeof(double) : 0; }

int vsl_mean_inv_depth_fwd(const float* disp, float* out, void* workspace, size_t workspace_bytes, int B, int H, int W,
                           float min_depth, float max_depth, void* stream) {
  if (!disp || !out || !workspace) return fail(VSL_ERR_INVALID_ARGUMENT, "mean_inv_depth: null pointer");
  if (B < 1 || B > 65535 || H < 1 || W < 1 || (long long)H * W >= (1LL << 28))
    return fail(VSL_ERR_INVALID_ARGUMENT, "mean_inv_depth: shape out of range");
  if (!(min_depth > 0.0f) || !(max_depth > min_depth)) return fail(VSL_ERR_INVALID_ARGUMENT, "mean_inv_depth: depth range");
  if (workspace_bytes < vsl_mean_inv_depth_workspace_bytes(B)) return fail(VSL_ERR_WORKSPACE_TOO_SMALL, "mean_inv_depth: workspace");
  if ((uintptr_t)workspace % 8) return fail(VSL_ERR_INVALID_ARGUMENT, "mean_inv_depth: workspace must be 8-byte aligned");
  const float a = 1.0f / max_depth, c = 1.0f / min_depth - 1.0f / max_depth;
  hipLaunchKernelGGL(k_mid_partial, dim3(kMidChunks, B), dim3(kMidThreads), 0, (hipStream_t)stream, disp, (double*)workspace,
                     H * W, a, c);
  hipLaunchKernelGGL(k_mid_final, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const double*)workspace, out, B,
                     H * W);
  return launch_status();
}

int vsl_pose_fwd_posecnn(const vsl_pose_desc* d, const float* disp, int H, int W, float min_depth, float max_depth,
                         float* t_scale_out, float* const* T, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_pose(d)) return rc;
  if (!T || !disp || !t_scale_out || !workspace) return fail(VSL_ERR_INVALID_ARGUMENT, "pose (posecnn): null pointer");
  if (d->B > 65535 || H < 1 || W < 1 || (long long)H * W >= (1LL << 28))
    return fail(VSL_ERR_INVALID_ARGUMENT, "pose (posecnn): shape out of range");
  if (!(min_depth > 0.0f) || !(max_depth > min_depth)) return fail(VSL_ERR_INVALID_ARGUMENT, "pose (posecnn): depth range");
  if (workspace_bytes < vsl_mean_inv_depth_workspace_bytes(d->B)) return fail(VSL_ERR_WORKSPACE_TOO_SMALL, "pose (posecnn): workspace");
  if ((uintptr_t)workspace % 8) return fail(VSL_ERR_INVALID_ARGUMENT, "pose (posecnn): workspace must be 8-byte aligned");
  PoseOut o{};
  for (int s = 0; s < d->n_src; ++s) {
    if (!T[s]) return fail(VSL_ERR_INVALID_ARGUMENT, "pose: null output pointer");
    o.T[s] = T[s];
  }
  const float a = 1.0f / max_depth, c = 1.0f / min_depth - 1.0f / max_depth;
  hipLaunchKernelGGL(k_mid_partial, dim3(kMidChunks, d->B), dim3(kMidThreads), 0, (hipStream_t)stream, disp, (double*)workspace,
                     H * W, a, c);
  PoseArgs pa = pose_args(d);
  pa.t_scale = nullptr;
  pa.mid_partial = (const double*)workspace;
  pa.t_scale_out = t_scale_out;
  pa.HW = H * W;
  const int n = d->B * d->n_src;
  hipLaunchKernelGGL(k_pose_fwd, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, pa, o);
  return launch_status();
}

int vsl_mean_inv_depth_bwd(const float* disp, const float* g_out, float* g_disp, int accumulate, int B, int H, int W,
                           float min_depth, float max_depth, void* stream) {
  if (!disp || !g_out || !g_disp) return fail(VSL_ERR_INVALID_ARGUMENT, "mean_inv_depth: null pointer");
  if (B < 1 || B > 65535 || H < 1 || W < 1 || (long long)H * W >= (1LL << 28))
    return fail(VSL_ERR_INVALID_ARGUMENT, "mean_inv_depth: shape out of range");
  if (!(min_depth > 0.0f) || !(max_depth > min_depth)) return fail(VSL_ERR_INVALID_ARGUMENT, "mean_inv_depth: depth range");
  const float a = 1.0f / max_depth, c = 1.0f / min_depth - 1.0f / max_depth;
  const int HW = H * W;
  int blocks = (HW + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(k_mid_bwd, dim3(blocks, B), dim3(256), 0, (hipStream_t)stream, disp, g_out, g_disp, HW, a, c, accumulate);
  return launch_status();
}

}  // extern "C"
