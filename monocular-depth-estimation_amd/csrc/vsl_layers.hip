// libvsloss.so -- the per-module kernels behind the drop-in layers.py surface
// (disp_to_depth, BackprojectDepth, Project3D, grid_sample warp, SSIM, get_smooth_loss),
// forward and backward.  The fused loss (vsl_loss.hip) is the hot path; these serve callers
// that use the modules one by one (Trainer.generate_images_pred, feature warping, validation).
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/vsloss.h"
#include "vsl_fused.hpp"

namespace vsl_err {
int fail(int code, const std::string& msg);  // defined in vsl_loss.hip
}

namespace {

using namespace vsl;
using vsl_err::fail;

int launch_status() {
  if (hipError_t e = hipGetLastError(); e != hipSuccess)
    return fail(VSL_ERR_LAUNCH_FAILED, std::string("kernel launch failed: ") + hipGetErrorString(e));
  return VSL_OK;
}

inline int blocks_for(long long n, int threads) {
  long long b = (n + threads - 1) / threads;
  return (int)(b < 1 ? 1 : (b > 65535LL * 16 ? 65535LL * 16 : b));
}

// ---------------------------------------------------------------------------
// disp_to_depth (layers.py:16-25)
// ---------------------------------------------------------------------------
__global__ void k_disp_to_depth_fwd(const float* __restrict__ disp, float* __restrict__ scaled, float* __restrict__ depth,
                                    long long n, float a, float c) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float sd = VSL_FADD(a, VSL_FMUL(c, disp[i]));
    if (scaled) scaled[i] = sd;
    if (depth) depth[i] = VSL_FDIV(1.0f, sd);
  }
}

__global__ void k_disp_to_depth_bwd(const float* __restrict__ disp, const float* __restrict__ g_scaled,
                                    const float* __restrict__ g_depth, float* __restrict__ g_disp, long long n, float a,
                                    float c) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float sd = a + c * disp[i];
    float g = 0.0f;
    if (g_scaled) g += g_scaled[i] * c;
    if (g_depth) g -= g_depth[i] * c / (sd * sd);
    g_disp[i] = g;
  }
}

// ---------------------------------------------------------------------------
// BackprojectDepth (layers.py:161-174): points[b,:,p] = (depth * inv_K[:3,:3] (x,y,1), 1)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void ray(const float* ik, float fx, float fy, float r[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float v = VSL_FMUL(ik[i * 4 + 0], fx);
    v = fmaf(ik[i * 4 + 1], fy, v);
    r[i] = fmaf(ik[i * 4 + 2], 1.0f, v);
  }
}

__global__ void k_backproject_fwd(const float* __restrict__ depth, const float* __restrict__ invK, float* __restrict__ pts,
                                  int H, int W) {
  const int b = blockIdx.y;
  const int HW = H * W;
  const float* ik = invK + 16 * b;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    float r[3];
    ray(ik, (float)(p % W), (float)(p / W), r);
    const float d = depth[(size_t)b * HW + p];
    float* o = pts + (size_t)b * 4 * HW + p;
#pragma unroll
    for (int i = 0; i < 3; ++i) o[(size_t)i * HW] = VSL_FMUL(d, r[i]);
    o[(size_t)3 * HW] = 1.0f;
  }
}

__global__ void k_backproject_bwd(const float* __restrict__ g_pts, const float* __restrict__ invK,
                                  float* __restrict__ g_depth, int H, int W) {
  const int b = blockIdx.y;
  const int HW = H * W;
  const float* ik = invK + 16 * b;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    float r[3];
    ray(ik, (float)(p % W), (float)(p / W), r);
    const float* g = g_pts + (size_t)b * 4 * HW + p;
    g_depth[(size_t)b * HW + p] = g[0] * r[0] + g[(size_t)HW] * r[1] + g[(size_t)2 * HW] * r[2];
  }
}

// ---------------------------------------------------------------------------
// Project3D (layers.py:188-202)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void load_proj(const float* K, const float* T, int b, float P[12]) {
#pragma unroll
  for (int e = 0; e < 12; ++e) P[e] = proj_entry(K + 16 * b, T + 16 * b, e / 4, e % 4);
}

__global__ void k_project_fwd(const float* __restrict__ pts, const float* __restrict__ K, const float* __restrict__ T,
                              float* __restrict__ grid, int H, int W, float eps) {
  const int b = blockIdx.y;
  const int HW = H * W;
  float P[12];
  load_proj(K, T, b, P);
  const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    const float* x = pts + (size_t)b * 4 * HW + p;
    const float X[4] = {x[0], x[(size_t)HW], x[(size_t)2 * HW], x[(size_t)3 * HW]};
    float cam[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float a = VSL_FMUL(P[i * 4 + 0], X[0]);
      a = fmaf(P[i * 4 + 1], X[1], a);
      a = fmaf(P[i * 4 + 2], X[2], a);
      cam[i] = fmaf(P[i * 4 + 3], X[3], a);
    }
    const float den = VSL_FADD(cam[2], eps);
    const float u = VSL_FDIV(cam[0], den), v = VSL_FDIV(cam[1], den);
    float2v o;
    o.x = VSL_FMUL(VSL_FSUB(VSL_FDIV(u, wm1), 0.5f), 2.0f);
    o.y = VSL_FMUL(VSL_FSUB(VSL_FDIV(v, hm1), 0.5f), 2.0f);
    *reinterpret_cast<float2v*>(grid + ((size_t)b * HW + p) * 2) = o;
  }
}

// g_pts [B,4,HW]; per-block sums of d L / d P into partials[b][block][12]
__global__ void k_project_bwd(const float* __restrict__ pts, const float* __restrict__ K, const float* __restrict__ T,
                              const float* __restrict__ g_grid, float* __restrict__ g_pts, float* __restrict__ partials,
                              int H, int W, float eps) {
  __shared__ float red[12][4];
  const int b = blockIdx.y;
  const int HW = H * W;
  float P[12];
  load_proj(K, T, b, P);
  const float sx = 2.0f / (float)(W - 1), sy = 2.0f / (float)(H - 1);
  float acc[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) acc[e] = 0.0f;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    const float* x = pts + (size_t)b * 4 * HW + p;
    const float X[4] = {x[0], x[(size_t)HW], x[(size_t)2 * HW], x[(size_t)3 * HW]};
    float cam[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) cam[i] = P[i * 4 + 0] * X[0] + P[i * 4 + 1] * X[1] + P[i * 4 + 2] * X[2] + P[i * 4 + 3] * X[3];
    const float id = 1.0f / (cam[2] + eps);
    const float2v g = *reinterpret_cast<const float2v*>(g_grid + ((size_t)b * HW + p) * 2);
    const float gu = g.x * sx, gv = g.y * sy;
    const float gc[3] = {gu * id, gv * id, -(gu * cam[0] + gv * cam[1]) * id * id};
    float* o = g_pts + (size_t)b * 4 * HW + p;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (g_pts) o[(size_t)j * HW] = P[0 * 4 + j] * gc[0] + P[1 * 4 + j] * gc[1] + P[2 * 4 + j] * gc[2];
#pragma unroll
      for (int i = 0; i < 3; ++i) acc[i * 4 + j] = fmaf(gc[i], X[j], acc[i * 4 + j]);
    }
  }
#pragma unroll
  for (int e = 0; e < 12; ++e) {
    float v = acc[e];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if ((threadIdx.x & 63) == 0) red[e][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 12) {
    const float s = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    partials[((size_t)b * gridDim.x + blockIdx.x) * 12 + threadIdx.x] = s;
  }
}

// one block per image: dP = sum of block partials; g_T = K[:3,:]^T dP; g_K[:3,:] = dP T^T
__global__ void k_project_bwd_finish(const float* __restrict__ partials, int nblocks, const float* __restrict__ K,
                                     const float* __restrict__ T, float* __restrict__ g_T, float* __restrict__ g_K) {
  __shared__ double dP[12];
  const int b = blockIdx.x;
  if (threadIdx.x < 12) {
    double s = 0.0;
    for (int i = 0; i < nblocks; ++i) s += (double)partials[((size_t)b * nblocks + i) * 12 + threadIdx.x];
    dP[threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const int k = threadIdx.x / 4, j = threadIdx.x % 4;
    if (g_T) {
      double g = 0.0;
      for (int i = 0; i < 3; ++i) g += (double)K[16 * b + i * 4 + k] * dP[i * 4 + j];
      g_T[16 * b + threadIdx.x] = (float)g;
    }
    if (g_K) {  // P[i][j] = sum_k K[i][k] T[k][j]  ->  dK[i][k] = sum_j dP[i][j] T[k][j]
      double g = 0.0;
      if (k < 3)
        for (int jj = 0; jj < 4; ++jj) g += dP[k * 4 + jj] * (double)T[16 * b + j * 4 + jj];
      g_K[16 * b + threadIdx.x] = (float)g;  // here k is the row i of K, j its column
    }
  }
}

// ---------------------------------------------------------------------------
// grid_sample: bilinear, padding_mode="border", align_corners=False (Trainer.py:467)
// ---------------------------------------------------------------------------
__global__ void k_grid_sample_fwd(const float* __restrict__ img, const float* __restrict__ grid, float* __restrict__ out,
                                  int C, int Hi, int Wi, int Ho, int Wo) {
  const int b = blockIdx.y;
  const int HWo = Ho * Wo, HWi = Hi * Wi;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HWo; p += gridDim.x * blockDim.x) {
    const float2v g = *reinterpret_cast<const float2v*>(grid + ((size_t)b * HWo + p) * 2);
    const Taps tx = axis_taps(g.x, Wi), ty = axis_taps(g.y, Hi);
    const int o00 = ty.i0 * Wi + tx.i0, o01 = ty.i0 * Wi + tx.i1, o10 = ty.i1 * Wi + tx.i0, o11 = ty.i1 * Wi + tx.i1;
    const float w00 = (1.0f - tx.t) * (1.0f - ty.t), w01 = tx.t * (1.0f - ty.t);
    const float w10 = (1.0f - tx.t) * ty.t, w11 = tx.t * ty.t;
    for (int c = 0; c < C; ++c) {
      const float* pl = img + ((size_t)b * C + c) * HWi;
      out[((size_t)b * C + c) * HWo + p] = pl[o00] * w00 + pl[o01] * w01 + pl[o10] * w10 + pl[o11] * w11;
    }
  }
}

// g_grid [B,Ho,Wo,2]; optionally scatters d L / d img with float atomics (g_img pre-zeroed by the host)
__global__ void k_grid_sample_bwd(const float* __restrict__ img, const float* __restrict__ grid,
                                  const float* __restrict__ g_out, float* __restrict__ g_grid, float* __restrict__ g_img,
                                  int C, int Hi, int Wi, int Ho, int Wo) {
  const int b = blockIdx.y;
  const int HWo = Ho * Wo, HWi = Hi * Wi;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HWo; p += gridDim.x * blockDim.x) {
    const float2v g = *reinterpret_cast<const float2v*>(grid + ((size_t)b * HWo + p) * 2);
    const Taps tx = axis_taps(g.x, Wi), ty = axis_taps(g.y, Hi);
    const int o00 = ty.i0 * Wi + tx.i0, o01 = ty.i0 * Wi + tx.i1, o10 = ty.i1 * Wi + tx.i0, o11 = ty.i1 * Wi + tx.i1;
    const float w00 = (1.0f - tx.t) * (1.0f - ty.t), w01 = tx.t * (1.0f - ty.t);
    const float w10 = (1.0f - tx.t) * ty.t, w11 = tx.t * ty.t;
    float gix = 0.0f, giy = 0.0f;
    for (int c = 0; c < C; ++c) {
      const float go = g_out[((size_t)b * C + c) * HWo + p];
      const float* pl = img + ((size_t)b * C + c) * HWi;
      const float v00 = pl[o00], v01 = pl[o01], v10 = pl[o10], v11 = pl[o11];
      gix = fmaf(go, (1.0f - ty.t) * (v01 - v00) + ty.t * (v11 - v10), gix);
      giy = fmaf(go, (1.0f - tx.t) * (v10 - v00) + tx.t * (v11 - v01), giy);
      if (g_img) {
        float* gp = g_img + ((size_t)b * C + c) * HWi;
        atomicAdd(gp + o00, go * w00);
        atomicAdd(gp + o01, go * w01);
        atomicAdd(gp + o10, go * w10);
        atomicAdd(gp + o11, go * w11);
      }
    }
    if (g_grid) {
      float2v o;
      o.x = gix * tx.live * (0.5f * (float)Wi);
      o.y = giy * ty.live * (0.5f * (float)Hi);
      *reinterpret_cast<float2v*>(g_grid + ((size_t)b * HWo + p) * 2) = o;
    }
  }
}

// ---------------------------------------------------------------------------
// SSIM (layers.py:229-262): LDS tile with a halo of 2, reflection padding
// ---------------------------------------------------------------------------
constexpr int ST_W = 32, ST_H = 8;  // 256 threads, one output pixel each

// value of (x, y) windows centred at image pixel (qy, qx) from the LDS tiles (origin = tile - 2)
__device__ __forceinline__ SsimOut ssim_at(const float* sx, const float* sy, int pitch, int ly, int lx, bool want_coef) {
  const float cx = sx[ly * pitch + lx], cy = sy[ly * pitch + lx];
  float s1 = 0.0f, s2 = 0.0f, s12 = 0.0f, t1 = 0.0f, t2 = 0.0f;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const float e = sx[(ly + dy) * pitch + lx + dx] - cx, d = sy[(ly + dy) * pitch + lx + dx] - cy;
      s1 += e;
      s2 = fmaf(e, e, s2);
      s12 = fmaf(e, d, s12);
      t1 += d;
      t2 = fmaf(d, d, t2);
    }
  constexpr float i9 = 1.0f / 9.0f;
  const float myp = t1 * i9;
  const float vyc = fmaf(t2, i9, -myp * myp) + kC2;
  return ssim_formula(cx, cy, s1, s2, s12, myp, vyc, want_coef);
}

template <bool BWD>
__global__ __launch_bounds__(ST_W* ST_H) void k_ssim(const float* __restrict__ x, const float* __restrict__ y,
                                                     const float* __restrict__ g_out, float* __restrict__ out,
                                                     int H, int W) {
  constexpr int PW = ST_W + 4, PH = ST_H + 4, PN = PW * PH;
  constexpr int CW = ST_W + 2, CH = ST_H + 2, CN = CW * CH;
  __shared__ float sx[PN], sy[PN];
  __shared__ float cA[BWD ? CN : 1], cB[BWD ? CN : 1], cG[BWD ? CN : 1];
  const size_t plane = (size_t)blockIdx.z * H * W;
  const int x0 = blockIdx.x * ST_W, y0 = blockIdx.y * ST_H;
  const int tid = threadIdx.y * ST_W + threadIdx.x;
  for (int p = tid; p < PN; p += ST_W * ST_H) {
    const int ly = p / PW, lx = p % PW;
    const int gy = reflect_clamp(y0 - 2 + ly, H), gx = reflect_clamp(x0 - 2 + lx, W);
    sx[p] = x[plane + (size_t)gy * W + gx];
    sy[p] = y[plane + (size_t)gy * W + gx];
  }
  __syncthreads();
  const int jx = x0 + threadIdx.x, jy = y0 + threadIdx.y;
  if (!BWD) {
    if (jx < W && jy < H) out[plane + (size_t)jy * W + jx] = ssim_at(sx, sy, PW, threadIdx.y + 2, threadIdx.x + 2, false).f;
    return;
  }
  // coefficients of every window centred in the tile grown by one, weighted by the upstream gradient
  for (int p = tid; p < CN; p += ST_W * ST_H) {
    const int cy_ = p / CW, cx_ = p % CW;
    const int qy = y0 - 1 + cy_, qx = x0 - 1 + cx_;
    float A = 0.0f, Bc = 0.0f, G = 0.0f;
    if (qy >= 0 && qy < H && qx >= 0 && qx < W) {
      const SsimOut o = ssim_at(sx, sy, PW, cy_ + 1, cx_ + 1, true);
      const float w = g_out[plane + (size_t)qy * W + qx] * (1.0f / 9.0f);
      A = o.A * w;
      Bc = o.B * w;
      G = o.G * w;
    }
    cA[p] = A;
    cB[p] = Bc;
    cG[p] = G;
  }
  __syncthreads();
  if (jx < W && jy < H) {
    float a = 0.0f, bq = 0.0f, g = 0.0f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const float fy = 1.0f + (float)(jy == 1 && dy == 0) + (float)(jy == H - 2 && dy == 2);
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const float m = fy * (1.0f + (float)(jx == 1 && dx == 0) + (float)(jx == W - 2 && dx == 2));
        const int p = (threadIdx.y + dy) * CW + threadIdx.x + dx;
        a = fmaf(m, cA[p], a);
        bq = fmaf(m, cB[p], bq);
        g = fmaf(m, cG[p], g);
      }
    }
    const int l = (threadIdx.y + 2) * PW + threadIdx.x + 2;
    out[plane + (size_t)jy * W + jx] = a + bq * sx[l] + g * sy[l];
  }
}

// ---------------------------------------------------------------------------
// get_smooth_loss (layers.py:211-226)
// ---------------------------------------------------------------------------
// partial sums per block: sum_x w|d disp|, sum_y w|d disp|; optional gradient planes
template <bool BWD>
__global__ __launch_bounds__(256) void k_smooth(const float* __restrict__ disp, const float* __restrict__ img,
                                                const float* __restrict__ g_loss, float* __restrict__ out, int C, int H,
                                                int W, float inv_nx, float inv_ny) {
  __shared__ float red[2][4];
  const int b = blockIdx.y;
  const int HW = H * W;
  const float* d = disp + (size_t)b * HW;
  const float* im = img + (size_t)b * C * HW;
  const float inv_c = 1.0f / (float)C;
  float sxs = 0.0f, sys = 0.0f;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    const int yy = p / W, xx = p % W;
    const float d0 = d[p];
    float e = 0.0f;
    // right, left, down, up
    const int off[4] = {1, -1, W, -W};
    const bool has[4] = {xx + 1 < W, xx > 0, yy + 1 < H, yy > 0};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (!has[t]) continue;
      float a = 0.0f;
      for (int c = 0; c < C; ++c) a += fabsf(im[(size_t)c * HW + p] - im[(size_t)c * HW + p + off[t]]);
      const float w = VSL_EXP(-a * inv_c);
      const float diff = (t == 0 || t == 2) ? d0 - d[p + off[t]] : d[p + off[t]] - d0;
      if (t == 0) sxs = fmaf(w, fabsf(diff), sxs);
      if (t == 2) sys = fmaf(w, fabsf(diff), sys);
      const float sg = sgn(diff) * w * (t < 2 ? inv_nx : inv_ny);
      e += (t == 0 || t == 2) ? sg : -sg;
    }
    if (BWD) out[(size_t)b * HW + p] = e * g_loss[0];
  }
  if (!BWD) {
    float v[2] = {sxs, sys};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v[k] += __shfl_xor(v[k], m, 64);
      if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 2) {
      const float s = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
      out[((size_t)b * gridDim.x + blockIdx.x) * 2 + threadIdx.x] = s;
    }
  }
}

__global__ void k_smooth_finish(const float* __restrict__ partials, int n, float inv_nx, float inv_ny, float* __restrict__ loss) {
  __shared__ double r0[256], r1[256];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    a += (double)partials[2 * i];
    b += (double)partials[2 * i + 1];
  }
  r0[threadIdx.x] = a;
  r1[threadIdx.x] = b;
  __syncthreads();
  for (int m = 128; m >= 1; m >>= 1) {
    if ((int)threadIdx.x < m) {
      r0[threadIdx.x] += r0[threadIdx.x + m];
      r1[threadIdx.x] += r1[threadIdx.x + m];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(r0[0] * inv_nx + r1[0] * inv_ny);
}

int check_hw(int B, int H, int W) {
  if (B < 1 || B > 65535 || H < 1 || W < 1 || (long long)H * W >= (1ll << 28))
    return fail(VSL_ERR_INVALID_ARGUMENT, "need 1 <= B <= 65535 and 1 <= H*W < 2^28");
  return VSL_OK;
}


// ---------------------------------------------------------------------------
// Feature warping at a coarse scale (Python/Train_distillation_new.py:1159-1182): disp -> depth -> backproject ->
// project -> bilinear sample of a C-channel feature map that carries a gradient, in one pass.  Coordinates follow
// the same rounding recipe as the kernels above, so they equal the module-by-module result bit for bit.
// Grid: (pixel blocks, channel groups, images); a thread handles one pixel and `cg` consecutive channels.
// ---------------------------------------------------------------------------
// Up to VSL_MAX_WARP_TENSORS feature maps share the sampling coordinates of one launch (the colour frame and the
// 25-channel affinity map of a source, Python/Trainer.py:467-473): their channel groups are numbered one tensor
// after the other, gstart[t] is the first group of tensor t.
constexpr int kWarpMaxT = VSL_MAX_WARP_TENSORS;
struct WarpArgs {
  const float* feats[kWarpMaxT];
  int C[kWarpMaxT];
  int gstart[kWarpMaxT + 1];
  const float* disp;
  const float* K;
  const float* inv_K;
  const float* T;
  int H, W, cg;
  float a_disp, c_disp, eps;
};
struct WarpPtrs {  // one pointer per tensor: outputs, upstream gradients or gradient accumulators (entries may be null)
  float* p[kWarpMaxT];
};
// which tensor channel group g belongs to, and that tensor's fields, through compile-time indices only (a run-time
// index into a by-value argument struct would move it to scratch memory)
struct WarpSel {
  const float* feats;
  float* ptr;
  int C, local_group;
};
__device__ __forceinline__ WarpSel warp_select(const WarpArgs& a, const WarpPtrs& q, int g) {
  // every field is read before the first comparison and picked with selects of values: with the reads inside the
  // conditional assignments the compiler merges them into one read through a selected address, which again is a
  // run-time index into the argument structs
  const float* f[kWarpMaxT];
  float* o[kWarpMaxT];
  int C[kWarpMaxT], start[kWarpMaxT];
#pragma unroll
  for (int t = 0; t < kWarpMaxT; ++t) {
    f[t] = a.feats[t];
    o[t] = q.p[t];
    C[t] = a.C[t];
    start[t] = a.gstart[t];
  }
  WarpSel r{f[0], o[0], C[0], g};
#pragma unroll
  for (int t = 1; t < kWarpMaxT; ++t) {
    const bool in = g >= start[t];
    r.feats = in ? f[t] : r.feats;
    r.ptr = in ? o[t] : r.ptr;
    r.C = in ? C[t] : r.C;
    r.local_group = in ? g - start[t] : r.local_group;
  }
  return r;
}

struct WarpPixel {
  float X[3];    // back-projected point
  float ray[3];  // inv_K[:3,:3] (x, y, 1)
  float cam[3];  // P (X, 1)
  float sd;      // scaled disparity
  float gx, gy;  // normalised sampling coordinates
};

__device__ __forceinline__ WarpPixel warp_pixel(const WarpArgs& a, const float P[12], int b, int p) {
  WarpPixel o;
  const float d = a.disp[(size_t)b * a.H * a.W + p];
  o.sd = VSL_FADD(a.a_disp, VSL_FMUL(a.c_disp, d));
  const float depth = VSL_FDIV(1.0f, o.sd);
  ray(a.inv_K + 16 * b, (float)(p % a.W), (float)(p / a.W), o.ray);
#pragma unroll
  for (int i = 0; i < 3; ++i) o.X[i] = VSL_FMUL(depth, o.ray[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float v = VSL_FMUL(P[i * 4 + 0], o.X[0]);
    v = fmaf(P[i * 4 + 1], o.X[1], v);
    v = fmaf(P[i * 4 + 2], o.X[2], v);
    o.cam[i] = fmaf(P[i * 4 + 3], 1.0f, v);
  }
  const float den = VSL_FADD(o.cam[2], a.eps);
  const float u = VSL_FDIV(o.cam[0], den), v = VSL_FDIV(o.cam[1], den);
  o.gx = VSL_FMUL(VSL_FSUB(VSL_FDIV(u, (float)(a.W - 1)), 0.5f), 2.0f);
  o.gy = VSL_FMUL(VSL_FSUB(VSL_FDIV(v, (float)(a.H - 1)), 0.5f), 2.0f);
  return o;
}

__global__ __launch_bounds__(256) void k_warp_fwd(WarpArgs a, WarpPtrs outs, float* __restrict__ sample) {
  const int b = blockIdx.z, g = blockIdx.y;
  const WarpSel sel = warp_select(a, outs, g);
  const int HW = a.H * a.W;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  float P[12];
  load_proj(a.K, a.T, b, P);
  const WarpPixel w = warp_pixel(a, P, b, p);
  if (sample && g == 0) {
    float2v o;
    o.x = w.gx;
    o.y = w.gy;
    *reinterpret_cast<float2v*>(sample + ((size_t)b * HW + p) * 2) = o;
  }
  const Taps tx = axis_taps(w.gx, a.W), ty = axis_taps(w.gy, a.H);
  const int o00 = ty.i0 * a.W + tx.i0, o01 = ty.i0 * a.W + tx.i1, o10 = ty.i1 * a.W + tx.i0, o11 = ty.i1 * a.W + tx.i1;
  const float w00 = (1.0f - tx.t) * (1.0f - ty.t), w01 = tx.t * (1.0f - ty.t);
  const float w10 = (1.0f - tx.t) * ty.t, w11 = tx.t * ty.t;
  const int c1 = min(sel.C, (sel.local_group + 1) * a.cg);
  for (int c = sel.local_group * a.cg; c < c1; ++c) {
    const float* pl = sel.feats + ((size_t)b * sel.C + c) * HW;
    sel.ptr[((size_t)b * sel.C + c) * HW + p] = pl[o00] * w00 + pl[o01] * w01 + pl[o10] * w10 + pl[o11] * w11;
  }
}

// Backward, first pass: the channel group's share of d L / d (ix, iy) of its pixel into partial[b][g][p] and the
// scatter of d L / d feats (float atomics; g_feats zeroed by the host).
__global__ __launch_bounds__(256) void k_warp_bwd_taps(WarpArgs a, WarpPtrs g_outs, WarpPtrs g_featss,
                                                       float* __restrict__ partial) {
  const int b = blockIdx.z, g = blockIdx.y;
  const WarpSel sel = warp_select(a, g_outs, g);
  const float* g_out = sel.ptr;  // null: this tensor has no upstream gradient
  float* g_feats = warp_select(a, g_featss, g).ptr;
  const int HW = a.H * a.W;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  float P[12];
  load_proj(a.K, a.T, b, P);
  const WarpPixel w = warp_pixel(a, P, b, p);
  const Taps tx = axis_taps(w.gx, a.W), ty = axis_taps(w.gy, a.H);
  const int o00 = ty.i0 * a.W + tx.i0, o01 = ty.i0 * a.W + tx.i1, o10 = ty.i1 * a.W + tx.i0, o11 = ty.i1 * a.W + tx.i1;
  const float w00 = (1.0f - tx.t) * (1.0f - ty.t), w01 = tx.t * (1.0f - ty.t);
  const float w10 = (1.0f - tx.t) * ty.t, w11 = tx.t * ty.t;
  float gix = 0.0f, giy = 0.0f;
  const int c1 = g_out ? min(sel.C, (sel.local_group + 1) * a.cg) : 0;
  for (int c = sel.local_group * a.cg; c < c1; ++c) {
    const float go = g_out[((size_t)b * sel.C + c) * HW + p];
    const float* pl = sel.feats + ((size_t)b * sel.C + c) * HW;
    const float v00 = pl[o00], v01 = pl[o01], v10 = pl[o10], v11 = pl[o11];
    gix = fmaf(go, (1.0f - ty.t) * (v01 - v00) + ty.t * (v11 - v10), gix);
    giy = fmaf(go, (1.0f - tx.t) * (v10 - v00) + tx.t * (v11 - v01), giy);
    if (g_feats) {
      float* gp = g_feats + ((size_t)b * sel.C + c) * HW;
      atomicAdd(gp + o00, go * w00);
      atomicAdd(gp + o01, go * w01);
      atomicAdd(gp + o10, go * w10);
      atomicAdd(gp + o11, go * w11);
    }
  }
  float2v o;
  o.x = gix;
  o.y = giy;
  *reinterpret_cast<float2v*>(partial + (((size_t)b * gridDim.y + g) * HW + p) * 2) = o;
}

// Backward, second pass: sum the groups in fixed order, chain through sampling, projection, back-projection and
// disp_to_depth to d L / d disp, and reduce d L / d P per block into sums[b][block][12].
__global__ __launch_bounds__(256) void k_warp_bwd_chain(WarpArgs a, const float* __restrict__ partial, int groups,
                                                        float* __restrict__ g_disp, float* __restrict__ sums) {
  __shared__ float red[12][4];
  const int b = blockIdx.y;
  const int HW = a.H * a.W;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  float P[12];
  load_proj(a.K, a.T, b, P);
  float acc[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) acc[e] = 0.0f;
  if (p < HW) {
    const WarpPixel w = warp_pixel(a, P, b, p);
    const Taps tx = axis_taps(w.gx, a.W), ty = axis_taps(w.gy, a.H);
    float gix = 0.0f, giy = 0.0f;
    for (int g = 0; g < groups; ++g) {
      const float2v v = *reinterpret_cast<const float2v*>(partial + (((size_t)b * groups + g) * HW + p) * 2);
      gix += v.x;
      giy += v.y;
    }
    // d ix / d gx = W/2 (align_corners=False), d gx / d u = 2/(W-1)
    const float gu = gix * tx.live * ((float)a.W / (float)(a.W - 1)), gv = giy * ty.live * ((float)a.H / (float)(a.H - 1));
    const float id = 1.0f / (w.cam[2] + a.eps);
    const float gc[3] = {gu * id, gv * id, -(gu * w.cam[0] + gv * w.cam[1]) * id * id};
    float g_depth = 0.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float gX = P[0 * 4 + j] * gc[0] + P[1 * 4 + j] * gc[1] + P[2 * 4 + j] * gc[2];
      g_depth = fmaf(gX, w.ray[j], g_depth);
#pragma unroll
      for (int i = 0; i < 3; ++i) acc[i * 4 + j] = gc[i] * w.X[j];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i * 4 + 3] = gc[i];
    if (g_disp) g_disp[(size_t)b * HW + p] = -g_depth * a.c_disp / (w.sd * w.sd);
  }
#pragma unroll
  for (int e = 0; e < 12; ++e) {
    float v = acc[e];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if ((threadIdx.x & 63) == 0) red[e][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 12) {
    const float s = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    sums[((size_t)b * gridDim.x + blockIdx.x) * 12 + threadIdx.x] = s;
  }
}

constexpr int kWarpGroupChannels = 8;
static_assert(sizeof(vsl_warp_desc) == 72, "vsl_warp_desc layout is part of the ABI");

int warp_groups(int C) { return (C + kWarpGroupChannels - 1) / kWarpGroupChannels; }
static_assert(sizeof(vsl_warp_multi_desc) == 112, "vsl_warp_multi_desc layout is part of the ABI");

int check_warp(const vsl_warp_desc* d) {
  if (!d) return fail(VSL_ERR_INVALID_ARGUMENT, "warp descriptor is null");
  if (int rc = check_hw(d->B, d->H, d->W)) return rc;
  if (d->H < 2 || d->W < 2) return fail(VSL_ERR_INVALID_ARGUMENT, "warp_features: need H >= 2 and W >= 2");
  if (d->C < 1 || d->C > 65535 * kWarpGroupChannels) return fail(VSL_ERR_INVALID_ARGUMENT, "warp_features: C out of range");
  if ((long long)d->B * d->C * d->H * d->W >= (1LL << 31)) return fail(VSL_ERR_INVALID_ARGUMENT, "warp_features: B*C*H*W must be < 2^31");
  if (!(d->min_depth > 0.0f) || !(d->max_depth > d->min_depth)) return fail(VSL_ERR_INVALID_ARGUMENT, "warp_features: depth range");
  if (!d->feats || !d->disp || !d->K || !d->inv_K || !d->T) return fail(VSL_ERR_INVALID_ARGUMENT, "warp_features: NULL tensor");
  return VSL_OK;
}

// the single-tensor descriptor as a one-entry multi-tensor descriptor
vsl_warp_multi_desc as_multi(const vsl_warp_desc* d) {
  vsl_warp_multi_desc m{};
  m.B = d->B;
  m.H = d->H;
  m.W = d->W;
  m.n = 1;
  m.min_depth = d->min_depth;
  m.max_depth = d->max_depth;
  m.eps = d->eps;
  m.C[0] = d->C;
  m.feats[0] = d->feats;
  m.disp = d->disp;
  m.K = d->K;
  m.inv_K = d->inv_K;
  m.T = d->T;
  return m;
}

int check_warp_multi(const vsl_warp_multi_desc* d) {
  if (!d) return fail(VSL_ERR_INVALID_ARGUMENT, "warp descriptor is null");
  if (int rc = check_hw(d->B, d->H, d->W)) return rc;
  if (d->H < 2 || d->W < 2) return fail(VSL_ERR_INVALID_ARGUMENT, "warp_features: need H >= 2 and W >= 2");
  if (d->n < 1 || d->n > VSL_MAX_WARP_TENSORS) return fail(VSL_ERR_INVALID_ARGUMENT, "warp_features: need 1..VSL_MAX_WARP_TENSORS tensors");
  long long groups = 0;
  for (int t = 0; t < d->n; ++t) {
    if (d->C[t] < 1) return fail(VSL_ERR_INVALID_ARGUMENT, "warp_features: C out of range");
    if ((long long)d->B * d->C[t] * d->H * d->W >= (1LL << 31)) return fail(VSL_ERR_INVALID_ARGUMENT, "warp_features: B*C*H*W must be < 2^31");
    if (!d->feats[t]) return fail(VSL_ERR_INVALID_ARGUMENT, "warp_features: NULL tensor");
    groups += warp_groups(d->C[t]);
  }
  if (groups > 65535) return fail(VSL_ERR_INVALID_ARGUMENT, "warp_features: C out of range");
  if (!(d->min_depth > 0.0f) || !(d->max_depth > d->min_depth)) return fail(VSL_ERR_INVALID_ARGUMENT, "warp_features: depth range");
  if (!d->disp || !d->K || !d->inv_K || !d->T) return fail(VSL_ERR_INVALID_ARGUMENT, "warp_features: NULL tensor");
  return VSL_OK;
}

WarpArgs warp_args(const vsl_warp_multi_desc* d) {
  WarpArgs a{};
  int g = 0;
  for (int t = 0; t < kWarpMaxT; ++t) {
    a.feats[t] = t < d->n ? d->feats[t] : nullptr;
    a.C[t] = t < d->n ? d->C[t] : 0;
    a.gstart[t] = t < d->n ? g : 0x7fffffff;   // tensors that do not exist start beyond every group
    if (t < d->n) g += warp_groups(d->C[t]);
  }
  a.gstart[kWarpMaxT] = g;                      // total number of groups
  a.disp = d->disp;
  a.K = d->K;
  a.inv_K = d->inv_K;
  a.T = d->T;
  a.H = d->H;
  a.W = d->W;
  a.cg = kWarpGroupChannels;
  a.a_disp = (float)(1.0 / (double)d->max_depth);
  a.c_disp = (float)(1.0 / (double)d->min_depth - 1.0 / (double)d->max_depth);
  a.eps = d->eps;
  return a;
}

size_t warp_multi_workspace(const vsl_warp_multi_desc* d) {
  const size_t HW = (size_t)d->H * d->W;
  const size_t blocks = (HW + 255) / 256;
  size_t groups = 0;
  for (int t = 0; t < d->n && t < VSL_MAX_WARP_TENSORS; ++t) groups += warp_groups(d->C[t]);
  return ((size_t)d->B * groups * HW * 2 + (size_t)d->B * blocks * 12) * sizeof(float);
}

int warp_multi_fwd(const vsl_warp_multi_desc* d, float* const* out, float* sample, void* stream) {
  if (int rc = check_warp_multi(d)) return rc;
  if (!out) return fail(VSL_ERR_INVALID_ARGUMENT, "warp_features: NULL output");
  WarpPtrs o{};
  for (int t = 0; t < d->n; ++t) {
    if (!out[t]) return fail(VSL_ERR_INVALID_ARGUMENT, "warp_features: NULL output");
    o.p[t] = out[t];
  }
  const WarpArgs a = warp_args(d);
  const int bx = (d->H * d->W + 255) / 256;
  hipLaunchKernelGGL(k_warp_fwd, dim3(bx, a.gstart[kWarpMaxT], d->B), dim3(256), 0, (hipStream_t)stream, a, o, sample);
  return launch_status();
}

int warp_multi_bwd(const vsl_warp_multi_desc* d, const float* const* g_out, float* const* g_feats, float* g_disp, float* g_T,
                   void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_warp_multi(d)) return rc;
  if (!g_out || !workspace) return fail(VSL_ERR_INVALID_ARGUMENT, "warp_features backward: NULL tensor");
  if (workspace_bytes < warp_multi_workspace(d))
    return fail(VSL_ERR_WORKSPACE_TOO_SMALL, "warp_features backward: workspace too small");
  if ((uintptr_t)workspace % 8) return fail(VSL_ERR_INVALID_ARGUMENT, "warp_features backward: workspace must be 8-byte aligned");
  const size_t HW = (size_t)d->H * d->W;
  WarpPtrs go{}, gf{};
  bool any = false;
  for (int t = 0; t < d->n; ++t) {
    go.p[t] = const_cast<float*>(g_out[t]);
    gf.p[t] = (g_feats && g_out[t]) ? g_feats[t] : nullptr;
    any = any || g_out[t] != nullptr;
    if (g_feats && g_feats[t] &&
        hipMemsetAsync(g_feats[t], 0, (size_t)d->B * d->C[t] * HW * sizeof(float), (hipStream_t)stream) != hipSuccess)
      return fail(VSL_ERR_LAUNCH_FAILED, "warp_features backward: memset failed");
  }
  if (!any) return fail(VSL_ERR_INVALID_ARGUMENT, "warp_features backward: NULL tensor");
  const WarpArgs a = warp_args(d);
  const int bx = (int)((HW + 255) / 256), groups = a.gstart[kWarpMaxT];
  float* partial = (float*)workspace;
  float* sums = partial + (size_t)d->B * groups * HW * 2;
  hipLaunchKernelGGL(k_warp_bwd_taps, dim3(bx, groups, d->B), dim3(256), 0, (hipStream_t)stream, a, go, gf, partial);
  if (g_disp || g_T) {
    hipLaunchKernelGGL(k_warp_bwd_chain, dim3(bx, d->B), dim3(256), 0, (hipStream_t)stream, a, (const float*)partial, groups,
                       g_disp, sums);
    if (g_T)
      hipLaunchKernelGGL(k_project_bwd_finish, dim3(d->B), dim3(64), 0, (hipStream_t)stream, (const float*)sums, bx, d->K, d->T,
                         g_T, (float*)nullptr);
  }
  return launch_status();
}

// ---------------------------------------------------------------------------
// Affinity reprojection loss (Python/Trainer.py:497-516): mean((pred - target * drop)^2), drop = pred where
// rand < keep_below, else 1.  Forward: float64 partial sum per workgroup in fixed order, finished by one workgroup.
// ---------------------------------------------------------------------------
constexpr int kAffBlocks = 512;

__device__ __forceinline__ float affinity_residual(float p, float t, float r, float keep_below, float& slope) {
  const bool keep = r < keep_below;
  slope = keep ? 1.0f - t : 1.0f;        // d (pred - target * drop) / d pred, through drop = pred as autograd does
  return keep ? p - t * p : p - t;
}

__global__ __launch_bounds__(256) void k_affinity_partial(const float* __restrict__ pred, const float* __restrict__ target,
                                                          const float* __restrict__ rnd, float keep_below, long long n,
                                                          double* __restrict__ partial) {
  __shared__ double red[256];
  double acc = 0.0;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float slope;
    const float e = affinity_residual(pred[i], target[i], rnd[i], keep_below, slope);
    acc += (double)(e * e);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int m = 128; m >= 1; m >>= 1) {
    if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void k_affinity_finish(const double* __restrict__ partial, int nblocks, double inv_n,
                                                         float* __restrict__ loss) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) acc += partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int m = 128; m >= 1; m >>= 1) {
    if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(red[0] * inv_n);
}

__global__ __launch_bounds__(256) void k_affinity_bwd(const float* __restrict__ pred, const float* __restrict__ target,
                                                      const float* __restrict__ rnd, float keep_below, long long n,
                                                      const float* __restrict__ g_loss, float two_over_n,
                                                      float* __restrict__ g_pred) {
  const float g = g_loss[0] * two_over_n;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float slope;
    const float e = affinity_residual(pred[i], target[i], rnd[i], keep_below, slope);
    g_pred[i] = g * e * slope;
  }
}

int affinity_blocks(long long n) {
  const long long b = (n + 255) / 256;
  return (int)(b < kAffBlocks ? b : kAffBlocks);
}

}  // namespace

extern "C" {

size_t vsl_warp_features_workspace_bytes(const vsl_warp_desc* d) {
  if (!d || d->B < 1 || d->C < 1 || d->H < 1 || d->W < 1) return 0;
  const vsl_warp_multi_desc m = as_multi(d);
  return warp_multi_workspace(&m);
}

int vsl_warp_features_fwd(const vsl_warp_desc* d, float* out, float* sample, void* stream) {
  if (int rc = check_warp(d)) return rc;
  const vsl_warp_multi_desc m = as_multi(d);
  return warp_multi_fwd(&m, &out, sample, stream);
}

int vsl_warp_features_bwd(const vsl_warp_desc* d, const float* g_out, float* g_feats, float* g_disp, float* g_T,
                          void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_warp(d)) return rc;
  if (!g_out || !workspace) return fail(VSL_ERR_INVALID_ARGUMENT, "warp_features backward: NULL tensor");
  const vsl_warp_multi_desc m = as_multi(d);
  return warp_multi_bwd(&m, &g_out, &g_feats, g_disp, g_T, workspace, workspace_bytes, stream);
}

size_t vsl_warp_multi_workspace_bytes(const vsl_warp_multi_desc* d) {
  if (!d || d->B < 1 || d->H < 1 || d->W < 1 || d->n < 1 || d->n > VSL_MAX_WARP_TENSORS) return 0;
  return warp_multi_workspace(d);
}

int vsl_warp_multi_fwd(const vsl_warp_multi_desc* d, float* const* out, float* sample, void* stream) {
  return warp_multi_fwd(d, out, sample, stream);
}

int vsl_warp_multi_bwd(const vsl_warp_multi_desc* d, const float* const* g_out, float* const* g_feats, float* g_disp, float* g_T,
                       void* workspace, size_t workspace_bytes, void* stream) {
  return warp_multi_bwd(d, g_out, g_feats, g_disp, g_T, workspace, workspace_bytes, stream);
}

int vsl_disp_to_depth_fwd(const float* disp, float* scaled, float* depth, long long n, float min_depth, float max_depth,
                          void* stream) {
  if (!disp || n < 0 || !(min_depth > 0.0f) || !(max_depth > min_depth))
    return fail(VSL_ERR_INVALID_ARGUMENT, "disp_to_depth: bad arguments");
  if (n == 0) return VSL_OK;
  const float a = (float)(1.0 / (double)max_depth), c = (float)(1.0 / (double)min_depth - 1.0 / (double)max_depth);
  hipLaunchKernelGGL(k_disp_to_depth_fwd, dim3(blocks_for(n, 256) > 4096 ? 4096 : blocks_for(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, disp, scaled, depth, n, a, c);
  return launch_status();
}

int vsl_disp_to_depth_bwd(const float* disp, const float* g_scaled, const float* g_depth, float* g_disp, long long n,
                          float min_depth, float max_depth, void* stream) {
  if (!disp || !g_disp || n < 0) return fail(VSL_ERR_INVALID_ARGUMENT, "disp_to_depth backward: bad arguments");
  if (n == 0) return VSL_OK;
  const float a = (float)(1.0 / (double)max_depth), c = (float)(1.0 / (double)min_depth - 1.0 / (double)max_depth);
  hipLaunchKernelGGL(k_disp_to_depth_bwd, dim3(blocks_for(n, 256) > 4096 ? 4096 : blocks_for(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, disp, g_scaled, g_depth, g_disp, n, a, c);
  return launch_status();
}

int vsl_backproject_fwd(const float* depth, const float* inv_K, float* points, int B, int H, int W, void* stream) {
  if (int rc = check_hw(B, H, W)) return rc;
  if (!depth || !inv_K || !points) return fail(VSL_ERR_INVALID_ARGUMENT, "backproject: NULL tensor");
  const int bx = blocks_for((long long)H * W, 256) > 1024 ? 1024 : blocks_for((long long)H * W, 256);
  hipLaunchKernelGGL(k_backproject_fwd, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, depth, inv_K, points, H, W);
  return launch_status();
}

int vsl_backproject_bwd(const float* g_points, const float* inv_K, float* g_depth, int B, int H, int W, void* stream) {
  if (int rc = check_hw(B, H, W)) return rc;
  if (!g_points || !inv_K || !g_depth) return fail(VSL_ERR_INVALID_ARGUMENT, "backproject backward: NULL tensor");
  const int bx = blocks_for((long long)H * W, 256) > 1024 ? 1024 : blocks_for((long long)H * W, 256);
  hipLaunchKernelGGL(k_backproject_bwd, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, g_points, inv_K, g_depth, H, W);
  return launch_status();
}

int vsl_project_fwd(const float* points, const float* K, const float* T, float* grid, int B, int H, int W, float eps,
                    void* stream) {
  if (int rc = check_hw(B, H, W)) return rc;
  if (H < 2 || W < 2) return fail(VSL_ERR_INVALID_ARGUMENT, "project: need H >= 2 and W >= 2");
  if (!points || !K || !T || !grid) return fail(VSL_ERR_INVALID_ARGUMENT, "project: NULL tensor");
  const int bx = blocks_for((long long)H * W, 256) > 1024 ? 1024 : blocks_for((long long)H * W, 256);
  hipLaunchKernelGGL(k_project_fwd, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, points, K, T, grid, H, W, eps);
  return launch_status();
}

size_t vsl_project_bwd_workspace_bytes(int B, int H, int W) {
  if (B < 1 || H < 1 || W < 1) return 0;
  const int bx = blocks_for((long long)H * W, 256) > 256 ? 256 : blocks_for((long long)H * W, 256);
  return (size_t)B * bx * 12 * sizeof(float);
}

int vsl_project_bwd(const float* points, const float* K, const float* T, const float* g_grid, float* g_points, float* g_T,
                    float* g_K, void* workspace, size_t workspace_bytes, int B, int H, int W, float eps, void* stream) {
  if (int rc = check_hw(B, H, W)) return rc;
  if (!points || !K || !T || !g_grid || !workspace) return fail(VSL_ERR_INVALID_ARGUMENT, "project backward: NULL tensor");
  if (workspace_bytes < vsl_project_bwd_workspace_bytes(B, H, W))
    return fail(VSL_ERR_WORKSPACE_TOO_SMALL, "project backward: workspace too small");
  const int bx = blocks_for((long long)H * W, 256) > 256 ? 256 : blocks_for((long long)H * W, 256);
  hipLaunchKernelGGL(k_project_bwd, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, points, K, T, g_grid, g_points,
                     (float*)workspace, H, W, eps);
  if (g_T || g_K)
    hipLaunchKernelGGL(k_project_bwd_finish, dim3(B), dim3(64), 0, (hipStream_t)stream, (const float*)workspace, bx, K, T,
                       g_T, g_K);
  return launch_status();
}

int vsl_grid_sample_fwd(const float* image, const float* grid, float* out, int B, int C, int Hi, int Wi, int Ho, int Wo,
                        void* stream) {
  if (int rc = check_hw(B, Hi, Wi)) return rc;
  if (int rc = check_hw(B, Ho, Wo)) return rc;
  if (C < 1 || !image || !grid || !out) return fail(VSL_ERR_INVALID_ARGUMENT, "grid_sample: bad arguments");
  const int bx = blocks_for((long long)Ho * Wo, 256) > 2048 ? 2048 : blocks_for((long long)Ho * Wo, 256);
  hipLaunchKernelGGL(k_grid_sample_fwd, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, image, grid, out, C, Hi, Wi, Ho, Wo);
  return launch_status();
}

int vsl_grid_sample_bwd(const float* image, const float* grid, const float* g_out, float* g_grid, float* g_image, int B,
                        int C, int Hi, int Wi, int Ho, int Wo, void* stream) {
  if (int rc = check_hw(B, Hi, Wi)) return rc;
  if (int rc = check_hw(B, Ho, Wo)) return rc;
  if (C < 1 || !image || !grid || !g_out) return fail(VSL_ERR_INVALID_ARGUMENT, "grid_sample backward: bad arguments");
  if (g_image && hipMemsetAsync(g_image, 0, (size_t)B * C * Hi * Wi * sizeof(float), (hipStream_t)stream) != hipSuccess)
    return fail(VSL_ERR_LAUNCH_FAILED, "grid_sample backward: memset failed");
  const int bx = blocks_for((long long)Ho * Wo, 256) > 2048 ? 2048 : blocks_for((long long)Ho * Wo, 256);
  hipLaunchKernelGGL(k_grid_sample_bwd, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, image, grid, g_out, g_grid,
                     g_image, C, Hi, Wi, Ho, Wo);
  return launch_status();
}

int vsl_ssim_fwd(const float* x, const float* y, float* out, int planes, int H, int W, void* stream) {
  if (planes < 1 || planes > 65535 || H < 2 || W < 2 || !x || !y || !out)
    return fail(VSL_ERR_INVALID_ARGUMENT, "ssim: need 1 <= planes <= 65535, H >= 2, W >= 2, non-NULL tensors");
  hipLaunchKernelGGL((k_ssim<false>), dim3((W + ST_W - 1) / ST_W, (H + ST_H - 1) / ST_H, planes), dim3(ST_W, ST_H), 0,
                     (hipStream_t)stream, x, y, (const float*)nullptr, out, H, W);
  return launch_status();
}

int vsl_ssim_bwd(const float* x, const float* y, const float* g_out, float* g_x, int planes, int H, int W, void* stream) {
  if (planes < 1 || planes > 65535 || H < 2 || W < 2 || !x || !y || !g_out || !g_x)
    return fail(VSL_ERR_INVALID_ARGUMENT, "ssim backward: need 1 <= planes <= 65535, H >= 2, W >= 2, non-NULL tensors");
  hipLaunchKernelGGL((k_ssim<true>), dim3((W + ST_W - 1) / ST_W, (H + ST_H - 1) / ST_H, planes), dim3(ST_W, ST_H), 0,
                     (hipStream_t)stream, x, y, g_out, g_x, H, W);
  return launch_status();
}

size_t vsl_smooth_workspace_bytes(int B, int H, int W) {
  if (B < 1 || H < 1 || W < 1) return 0;
  const int bx = blocks_for((long long)H * W, 256) > 128 ? 128 : blocks_for((long long)H * W, 256);
  return (size_t)B * bx * 2 * sizeof(float);
}

int vsl_smooth_fwd(const float* disp, const float* img, float* loss, void* workspace, size_t workspace_bytes, int B, int C,
                   int H, int W, void* stream) {
  if (int rc = check_hw(B, H, W)) return rc;
  if (H < 2 || W < 2 || C < 1 || !disp || !img || !loss || !workspace)
    return fail(VSL_ERR_INVALID_ARGUMENT, "smooth: need H >= 2, W >= 2, C >= 1, non-NULL tensors");
  if (workspace_bytes < vsl_smooth_workspace_bytes(B, H, W)) return fail(VSL_ERR_WORKSPACE_TOO_SMALL, "smooth: workspace too small");
  const int bx = blocks_for((long long)H * W, 256) > 128 ? 128 : blocks_for((long long)H * W, 256);
  const float inv_nx = (float)(1.0 / ((double)B * H * (W - 1))), inv_ny = (float)(1.0 / ((double)B * (H - 1) * W));
  hipLaunchKernelGGL((k_smooth<false>), dim3(bx, B), dim3(256), 0, (hipStream_t)stream, disp, img, (const float*)nullptr,
                     (float*)workspace, C, H, W, inv_nx, inv_ny);
  hipLaunchKernelGGL(k_smooth_finish, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, B * bx, inv_nx,
                     inv_ny, loss);
  return launch_status();
}

int vsl_smooth_bwd(const float* disp, const float* img, const float* g_loss, float* g_disp, int B, int C, int H, int W,
                   void* stream) {
  if (int rc = check_hw(B, H, W)) return rc;
  if (H < 2 || W < 2 || C < 1 || !disp || !img || !g_loss || !g_disp)
    return fail(VSL_ERR_INVALID_ARGUMENT, "smooth backward: need H >= 2, W >= 2, C >= 1, non-NULL tensors");
  const int bx = blocks_for((long long)H * W, 256) > 1024 ? 1024 : blocks_for((long long)H * W, 256);
  const float inv_nx = (float)(1.0 / ((double)B * H * (W - 1))), inv_ny = (float)(1.0 / ((double)B * (H - 1) * W));
  hipLaunchKernelGGL((k_smooth<true>), dim3(bx, B), dim3(256), 0, (hipStream_t)stream, disp, img, g_loss, g_disp, C, H, W,
                     inv_nx, inv_ny);
  return launch_status();
}

size_t vsl_affinity_loss_workspace_bytes(long long n) { return n < 1 ? 0 : (size_t)affinity_blocks(n) * sizeof(double); }

int vsl_affinity_loss_fwd(const float* pred, const float* target, const float* rnd, float keep_below, long long n, float* loss,
                          void* workspace, size_t workspace_bytes, void* stream) {
  if (!pred || !target || !rnd || !loss || !workspace || n < 1) return fail(VSL_ERR_INVALID_ARGUMENT, "affinity loss: bad arguments");
  if (workspace_bytes < vsl_affinity_loss_workspace_bytes(n)) return fail(VSL_ERR_WORKSPACE_TOO_SMALL, "affinity loss: workspace too small");
  if ((uintptr_t)workspace % 8) return fail(VSL_ERR_INVALID_ARGUMENT, "affinity loss: workspace must be 8-byte aligned");
  const int nb = affinity_blocks(n);
  hipLaunchKernelGGL(k_affinity_partial, dim3(nb), dim3(256), 0, (hipStream_t)stream, pred, target, rnd, keep_below, n, (double*)workspace);
  hipLaunchKernelGGL(k_affinity_finish, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, nb, 1.0 / (double)n, loss);
  return launch_status();
}

int vsl_affinity_loss_bwd(const float* pred, const float* target, const float* rnd, float keep_below, long long n,
                          const float* g_loss, float* g_pred, void* stream) {
  if (!pred || !target || !rnd || !g_loss || !g_pred || n < 1) return fail(VSL_ERR_INVALID_ARGUMENT, "affinity loss backward: bad arguments");
  hipLaunchKernelGGL(k_affinity_bwd, dim3(affinity_blocks(n) * 4), dim3(256), 0, (hipStream_t)stream, pred, target, rnd, keep_below, n,
                     g_loss, (float)(2.0 / (double)n), g_pred);
  return launch_status();
}

}  // extern "C"
