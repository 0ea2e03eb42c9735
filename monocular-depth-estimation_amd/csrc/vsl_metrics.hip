// libvsloss.so -- depth metrics of a training step (Python/Trainer.py:622-662, Python/layers.py:265-283):
// bilinear upsampling of the predicted depth to the ground-truth size, clamp, mask (gt > 0 inside the Garg/Eigen
// crop), median scaling over the whole batch and the seven masked means.  The reference gathers the masked
// elements (a host synchronisation) and sorts twice for the medians; here the medians come from a three-pass
// radix select over the float bit patterns (all values are positive, so the patterns order like the values)
// and nothing synchronises the host.  Counts use integer atomics, sums are reduced in a fixed order: results
// are deterministic.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/vsloss.h"

namespace vsl_err {
int fail(int code, const std::string& msg);  // defined in vsl_loss.hip
}

namespace {

using vsl_err::fail;

int launch_status() {
  if (hipError_t e = hipGetLastError(); e != hipSuccess)
    return fail(VSL_ERR_LAUNCH_FAILED, std::string("kernel launch failed: ") + hipGetErrorString(e));
  return VSL_OK;
}

static_assert(sizeof(vsl_depth_metrics_desc) == 64, "vsl_depth_metrics_desc layout is part of the ABI");

constexpr int kBins = 2048;       // 11 + 11 + 10 bits
constexpr int kThreads = 256;
constexpr int kTargetBlocks = 512;  // blocks per launch: every block flushes its histogram with global atomics on the
                                    // same few dozen addresses, so more blocks cost more than they gain

struct SelectState {
  unsigned hist[2][kBins];  // [0] ground truth, [1] prediction
  unsigned prefix[2];       // key bits fixed so far
  unsigned rank[2];         // rank of the median among the keys that share the prefix
  unsigned n;               // masked elements
  float median[2];
};

struct MetricArgs {
  const float* depth;     // [B,1,H,W]
  const float* depth_gt;  // [B,1,Hg,Wg]
  float* pred;            // [B,ch,cw] upsampled + clamped prediction inside the crop, -1 where masked out
  SelectState* st;
  double* partial;        // [B * gridDim.x][7]
  int H, W, Hg, Wg, y0, x0, ch, cw;
  float lo, hi;
};

__device__ __forceinline__ unsigned digit(unsigned key, int pass) {
  return pass == 0 ? key >> 21 : pass == 1 ? (key >> 10) & 0x7FFu : key & 0x3FFu;
}
__device__ __forceinline__ bool matches(unsigned key, unsigned prefix, int pass) {
  return pass == 0 ? true : pass == 1 ? (key >> 21) == prefix : (key >> 10) == prefix;
}

// F.interpolate(depth, [Hg, Wg], mode="bilinear", align_corners=False) at output pixel (y, x)
__device__ __forceinline__ float upsample(const float* d, int H, int W, int Hg, int Wg, int y, int x) {
  const float rh = (float)H / (float)Hg, rw = (float)W / (float)Wg;
  const float sy = fmaxf(rh * ((float)y + 0.5f) - 0.5f, 0.0f), sx = fmaxf(rw * ((float)x + 0.5f) - 0.5f, 0.0f);
  const int y1 = min((int)sy, H - 1), x1 = min((int)sx, W - 1);
  const int yp = y1 < H - 1 ? 1 : 0, xp = x1 < W - 1 ? 1 : 0;
  const float ly = sy - (float)y1, lx = sx - (float)x1;
  const float hy = 1.0f - ly, hx = 1.0f - lx;
  const float* r0 = d + (size_t)y1 * W + x1;
  const float* r1 = r0 + (size_t)yp * W;
  return hy * (hx * r0[0] + lx * r0[xp]) + ly * (hx * r1[0] + lx * r1[xp]);
}

// pass 0: prediction inside the crop, mask, count, first histogram; passes 1, 2: next digit of the keys that share the prefix
template <int PASS>
__global__ __launch_bounds__(kThreads) void k_hist(MetricArgs a) {
  __shared__ unsigned h[2][kBins];
  for (int i = threadIdx.x; i < 2 * kBins; i += kThreads) (&h[0][0])[i] = 0u;
  __syncthreads();
  const int b = blockIdx.y;
  const int n = a.ch * a.cw;
  const unsigned p0 = PASS ? a.st->prefix[0] : 0u, p1 = PASS ? a.st->prefix[1] : 0u;
  unsigned count = 0;
  for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
    const int y = a.y0 + i / a.cw, x = a.x0 + i % a.cw;
    const float gt = a.depth_gt[((size_t)b * a.Hg + y) * a.Wg + x];
    float pr;
    if (PASS == 0) {
      pr = -1.0f;
      if (gt > 0.0f) pr = fminf(fmaxf(upsample(a.depth + (size_t)b * a.H * a.W, a.H, a.W, a.Hg, a.Wg, y, x), a.lo), a.hi);
      a.pred[(size_t)b * n + i] = pr;
    } else {
      pr = a.pred[(size_t)b * n + i];
    }
    if (pr < 0.0f) continue;
    ++count;
    const unsigned kg = __float_as_uint(gt), kp = __float_as_uint(pr);
    if (matches(kg, p0, PASS)) atomicAdd(&h[0][digit(kg, PASS)], 1u);
    if (matches(kp, p1, PASS)) atomicAdd(&h[1][digit(kp, PASS)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * kBins; i += kThreads) {
    const unsigned v = (&h[0][0])[i];
    if (v) atomicAdd(&a.st->hist[0][0] + i, v);
  }
  if (PASS == 0) {
    __shared__ unsigned cnt[kThreads / 64];
    for (int m = 32; m >= 1; m >>= 1) count += __shfl_xor(count, m, 64);
    if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = count;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned t = 0;
      for (int w = 0; w < kThreads / 64; ++w) t += cnt[w];
      if (t) atomicAdd(&a.st->n, t);
    }
  }
}

// one block of two waves, one per array: locate the bin that holds the median, extend the prefix, clear the
// histograms.  Every lane sums 32 consecutive bins, a wave scan finds the lane whose range crosses the rank and
// that lane walks its 32 bins.
template <int PASS>
__global__ __launch_bounds__(128) void k_scan(SelectState* st) {
  __shared__ unsigned h[2][kBins];
  for (int i = threadIdx.x; i < 2 * kBins; i += 128) (&h[0][0])[i] = (&st->hist[0][0])[i];
  __syncthreads();
  const int which = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned n = st->n;
  const unsigned rank = PASS ? st->rank[which] : (n ? (n - 1u) / 2u : 0u);  // torch.median: the lower middle element
  constexpr int PER = kBins / 64;
  unsigned mine = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) mine += h[which][lane * PER + i];
  unsigned incl = mine;  // inclusive scan over the lanes
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  const unsigned before = incl - mine;
  if (before <= rank && rank < incl) {  // exactly one lane when n > 0
    unsigned acc = before, bin = (unsigned)(lane * PER);
    for (int i = 0; i < PER; ++i) {
      const unsigned c = h[which][lane * PER + i];
      if (acc + c > rank) {
        bin = (unsigned)(lane * PER + i);
        break;
      }
      acc += c;
    }
    st->rank[which] = rank - acc;
    const unsigned prefix = PASS == 0 ? bin : PASS == 1 ? (st->prefix[which] << 11) | bin : (st->prefix[which] << 10) | bin;
    st->prefix[which] = prefix;
    if (PASS == 2) st->median[which] = __uint_as_float(prefix);
  }
  if (PASS == 2 && n == 0 && lane == 0) st->median[which] = __uint_as_float(0x7FC00000u);
  if (PASS < 2)
    for (int i = threadIdx.x; i < 2 * kBins; i += 128) (&st->hist[0][0])[i] = 0u;
}

// the seven sums of layers.py:265-283 over the masked elements, prediction scaled by median(gt) / median(pred)
__global__ __launch_bounds__(kThreads) void k_terms(MetricArgs a) {
  __shared__ double red[7][kThreads / 64];
  const int b = blockIdx.y;
  const int n = a.ch * a.cw;
  const float ratio = a.st->median[0] / a.st->median[1];  // Trainer.py:646
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
    float pr = a.pred[(size_t)b * n + i];
    if (pr < 0.0f) continue;
    const int y = a.y0 + i / a.cw, x = a.x0 + i % a.cw;
    const float gt = a.depth_gt[((size_t)b * a.Hg + y) * a.Wg + x];
    pr = fminf(fmaxf(pr * ratio, a.lo), a.hi);  // Trainer.py:648
    const float thresh = fmaxf(gt / pr, pr / gt);
    const float diff = gt - pr, dl = logf(gt) - logf(pr);
    acc[0] += (double)(fabsf(diff) / gt);
    acc[1] += (double)(diff * diff / gt);
    acc[2] += (double)(diff * diff);
    acc[3] += (double)(dl * dl);
    acc[4] += thresh < 1.25f ? 1.0 : 0.0;
    acc[5] += thresh < 1.5625f ? 1.0 : 0.0;     // 1.25 ** 2
    acc[6] += thresh < 1.953125f ? 1.0 : 0.0;   // 1.25 ** 3
  }
#pragma unroll
  for (int e = 0; e < 7; ++e) {
    double v = acc[e];
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if ((threadIdx.x & 63) == 0) red[e][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 7) {
    double s = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) s += red[threadIdx.x][w];
    a.partial[((size_t)b * gridDim.x + blockIdx.x) * 7 + threadIdx.x] = s;
  }
}

// metrics[0..6] = abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3; [7] = their sum (Trainer.py:659); [8] = masked count;
// [9], [10] = the two medians.  One block of 256 threads: strided sums of the block partials, then an LDS tree.
__global__ __launch_bounds__(256) void k_metrics_final(const double* __restrict__ partial, int nblocks, const SelectState* st,
                                                       float* __restrict__ out) {
  __shared__ double red[7][256];
  __shared__ float m[7];
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < nblocks; i += 256)
#pragma unroll
    for (int e = 0; e < 7; ++e) acc[e] += partial[(size_t)i * 7 + e];
#pragma unroll
  for (int e = 0; e < 7; ++e) red[e][threadIdx.x] = acc[e];
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o)
#pragma unroll
      for (int e = 0; e < 7; ++e) red[e][threadIdx.x] += red[e][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 7) {
    const double mean = red[threadIdx.x][0] / (double)st->n;  // 0 / 0 = NaN for an empty mask, where the reference raises
    const float v = (threadIdx.x == 2 || threadIdx.x == 3) ? sqrtf((float)mean) : (float)mean;
    m[threadIdx.x] = v;
    out[threadIdx.x] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[7] = m[0] + m[1] + m[2] + m[3] + m[4] + m[5] + m[6];
    out[8] = (float)st->n;
    out[9] = st->median[0];
    out[10] = st->median[1];
  }
}

struct Geometry {
  int y0, y1, x0, x1, ch, cw, bx;
  size_t pred_bytes, state_off, partial_off, total;
};

Geometry geometry(const vsl_depth_metrics_desc* d) {
  Geometry g{};
  // Python slice semantics of crop_mask[:, :, y0:y1, x0:x1] (Trainer.py:641)
  auto clip = [](int v, int n) { return v < 0 ? 0 : (v > n ? n : v); };
  g.y0 = clip(d->crop_y0, d->Hg);
  g.y1 = clip(d->crop_y1, d->Hg);
  g.x0 = clip(d->crop_x0, d->Wg);
  g.x1 = clip(d->crop_x1, d->Wg);
  g.ch = g.y1 > g.y0 ? g.y1 - g.y0 : 0;
  g.cw = g.x1 > g.x0 ? g.x1 - g.x0 : 0;
  const long long n = (long long)g.ch * g.cw;
  long long bx = (n + kThreads - 1) / kThreads;
  const long long cap = kTargetBlocks / d->B > 1 ? kTargetBlocks / d->B : 1;
  g.bx = (int)(bx < 1 ? 1 : (bx > cap ? cap : bx));
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  g.pred_bytes = up((size_t)d->B * (size_t)n * sizeof(float));
  g.state_off = g.pred_bytes;
  g.partial_off = g.state_off + up(sizeof(SelectState));
  g.total = g.partial_off + up((size_t)d->B * g.bx * 7 * sizeof(double));
  return g;
}

int check_metrics(const vsl_depth_metrics_desc* d) {
  if (!d) return fail(VSL_ERR_INVALID_ARGUMENT, "depth metrics descriptor is null");
  if (d->B < 1 || d->B > 65535 || d->H < 1 || d->W < 1 || d->Hg < 1 || d->Wg < 1)
    return fail(VSL_ERR_INVALID_ARGUMENT, "depth_metrics: shape out of range");
  if ((long long)d->H * d->W >= (1LL << 28) || (long long)d->Hg * d->Wg >= (1LL << 28) ||
      (long long)d->B * d->Hg * d->Wg >= (1LL << 31))
    return fail(VSL_ERR_INVALID_ARGUMENT, "depth_metrics: image too large");
  if (!(d->clamp_min > 0.0f) || !(d->clamp_max > d->clamp_min))
    return fail(VSL_ERR_INVALID_ARGUMENT, "depth_metrics: need 0 < clamp_min < clamp_max");
  if (!d->depth || !d->depth_gt) return fail(VSL_ERR_INVALID_ARGUMENT, "depth_metrics: NULL tensor");
  return VSL_OK;
}

}  // namespace

extern "C" {

size_t vsl_depth_metrics_workspace_bytes(const vsl_depth_metrics_desc* d) {
  if (!d || d->B < 1 || d->Hg < 1 || d->Wg < 1) return 0;
  return geometry(d).total;
}

int vsl_depth_metrics(const vsl_depth_metrics_desc* d, float* metrics, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_metrics(d)) return rc;
  if (!metrics || !workspace) return fail(VSL_ERR_INVALID_ARGUMENT, "depth_metrics: NULL output or workspace");
  const Geometry g = geometry(d);
  if (workspace_bytes < g.total) return fail(VSL_ERR_WORKSPACE_TOO_SMALL, "depth_metrics: workspace too small");
  if ((uintptr_t)workspace % 256) return fail(VSL_ERR_INVALID_ARGUMENT, "depth_metrics: workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  char* base = (char*)workspace;
  MetricArgs a{};
  a.depth = d->depth;
  a.depth_gt = d->depth_gt;
  a.pred = (float*)base;
  a.st = (SelectState*)(base + g.state_off);
  a.partial = (double*)(base + g.partial_off);
  a.H = d->H;
  a.W = d->W;
  a.Hg = d->Hg;
  a.Wg = d->Wg;
  a.y0 = g.y0;
  a.x0 = g.x0;
  a.ch = g.ch;
  a.cw = g.cw;
  a.lo = d->clamp_min;
  a.hi = d->clamp_max;
  if (hipMemsetAsync(a.st, 0, sizeof(SelectState), s) != hipSuccess) return fail(VSL_ERR_LAUNCH_FAILED, "depth_metrics: memset failed");
  const dim3 grid(g.bx, d->B);
  hipLaunchKernelGGL(k_hist<0>, grid, dim3(kThreads), 0, s, a);
  hipLaunchKernelGGL(k_scan<0>, dim3(1), dim3(128), 0, s, a.st);
  hipLaunchKernelGGL(k_hist<1>, grid, dim3(kThreads), 0, s, a);
  hipLaunchKernelGGL(k_scan<1>, dim3(1), dim3(128), 0, s, a.st);
  hipLaunchKernelGGL(k_hist<2>, grid, dim3(kThreads), 0, s, a);
  hipLaunchKernelGGL(k_scan<2>, dim3(1), dim3(128), 0, s, a.st);
  hipLaunchKernelGGL(k_terms, grid, dim3(kThreads), 0, s, a);
  hipLaunchKernelGGL(k_metrics_final, dim3(1), dim3(256), 0, s, (const double*)a.partial, g.bx * d->B, (const SelectState*)a.st,
                     metrics);
  return launch_status();
}

}  // extern "C"
