// libvsloss.so -- kernels and C-ABI entry points of the fused view-synthesis loss.
//
//   vsl_ident_kernel     identity-reprojection minimum per pixel (once per batch)
//   vsl_fused_kernel     one workgroup per 28x16 tile; the tile program of vsl_fused.hpp
//   vsl_finalize_kernel  one workgroup per image: deterministic float64 reduction of the
//                        tile records, per-image smoothness factors, pose gradients; the
//                        last workgroup to finish writes the loss scalars
//   vsl_grad_kernel      elementwise: d total / d disp from the photometric and the
//                        smoothness parts, scaled by the upstream gradient
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vsloss.h"
#include "../../include/vsloss_debug.h"
#include "vsl_fused.hpp"

namespace vsl_err {
thread_local std::string g_err;
// records the message of the calling thread's last error and returns the code
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
}  // namespace vsl_err

namespace {

using namespace vsl;
using vsl_err::fail;

static_assert(sizeof(vsl_loss_desc) == 208 && sizeof(vsl_head) == 200, "C-ABI struct layout changed");
constexpr int kTW = kTileW;
constexpr int kIdTH = kIdentTileH, kIdNT = tile_threads(kIdentTileH);
// Resident workgroups per CU the fused kernel is compiled for: as many as its LDS planes allow (160 KiB per CU), at
// most five (the register budget of five waves per SIMD, 96, is the smallest the tile program fits without spills).
template <int S, bool AVG>
constexpr int fused_blocks_per_cu() {
  constexpr int TH = fused_tile_h(S, AVG);
  constexpr int by_lds = 160 * 1024 / (Fused<fused_tile_w(S, AVG), TH, tile_threads(TH), S, AVG>::LDS_FLOATS * (int)sizeof(float));
  return by_lds > 5 ? 5 : by_lds;
}

#ifdef VSL_STAMPS
unsigned long long* g_stamps = nullptr;
size_t g_stamp_bytes = 0;
#endif

// --------------------------------------------------------------------------
// fused tile kernel
// --------------------------------------------------------------------------
// Diagnostic build only (-DVSL_STAMPS): lane 0 of every wave records the shader clock at the
// phase boundaries into a buffer of its own (kp.gedge is redirected by the host tool); the
// shipped library contains no stamp.
#ifdef VSL_STAMPS
#define VSL_STAMP(i)                                                                            \
  do {                                                                                          \
    unsigned long long t_;                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    if ((threadIdx.x & 63) == 0)                                                                \
      kp.stamps[((size_t)block_linear * 8 + (threadIdx.x >> 6)) * 16 + (i)] = t_;      \
  } while (0)
#else
#define VSL_STAMP(i)
#endif

// Workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so workgroups i and i+8 share an
// L2.  The 1-D grid of 8*chunk workgroups is mapped so that every XCD walks one contiguous run of tiles
// (x fastest, then y, then image): horizontally and vertically adjacent tiles, which share 128-byte lines
// of the images (a 28-pixel tile row is 112 bytes), then meet in the same L2 at about the same time.
// Returns the tile's linear index (x fastest), or -1 for the padding workgroups past the last tile.
// The three divisors of that mapping, prepared on the host with floor(2^32 / d) each: a uniform integer division
// costs the compiler about 35 vector and scalar instructions per wave, the multiply-high form six scalar ones.
struct TileDiv {
  unsigned tiles_x, per_image, per_head;  // tiles per tile row, per image, per head
  unsigned m_x, m_image, m_head;          // min(floor(2^32 / d), 2^32 - 1)
};
// t / d and t % d for any t < 2^32: the multiply-high quotient is floor(t / d) or one less
__device__ __forceinline__ unsigned div_prepared(unsigned t, unsigned d, unsigned m, unsigned& rem) {
  unsigned q = __umulhi(t, m);
  unsigned r = t - q * d;
  if (r >= d) {
    ++q;
    r -= d;
  }
  rem = r;
  return q;
}
__device__ __forceinline__ int swizzled_tile(const TileDiv& td, int tile_w, int tile_h, int n_heads, int& head, int& b, int& x0,
                                             int& y0) {
  const unsigned chunk = gridDim.x >> 3;
#ifdef VSL_NO_SWIZZLE
  unsigned t = blockIdx.x;
#else
  unsigned t = (blockIdx.x & 7u) * chunk + (blockIdx.x >> 3);
#endif
  if (t >= td.per_head * (unsigned)n_heads) return -1;
  unsigned in_head, in_image, tx;
  head = (int)div_prepared(t, td.per_head, td.m_head, in_head);
  b = (int)div_prepared(in_head, td.per_image, td.m_image, in_image);
  const unsigned ty = div_prepared(in_image, td.tiles_x, td.m_x, tx);
  x0 = (int)tx * tile_w;
  y0 = (int)ty * tile_h;
  return (int)in_head;
}

// Per-head part of the fused kernel's arguments: several disparity heads over the same images and poses run
// as one grid (head-major tile order), which fills the CUs better than one launch per head.
struct HeadArgs {
  int n_heads;
  const float* disp[VSL_MAX_HEADS];
  float* gphoto[VSL_MAX_HEADS];
  float* gedge[VSL_MAX_HEADS];
  float* ident_sel[VSL_MAX_HEADS];
  float* partials[VSL_MAX_HEADS];
  unsigned int* counter[VSL_MAX_HEADS];
  float cst_ssim[VSL_MAX_HEADS], cst_l1[VSL_MAX_HEADS];
  int want_grad[VSL_MAX_HEADS];
  int index[VSL_MAX_HEADS];  // the head's position in the caller's table (a launch may cover a subset of the heads)
  TileDiv td;
};

// timing probe only (-DVSL_PROBE_NOBAR, results wrong): the tile program without its barriers, i.e. the cost of the
// phase coupling between the four waves of a workgroup
#ifdef VSL_PROBE_NOBAR
#define VSL_TILE_SYNC() __builtin_amdgcn_sched_barrier(0)
#else
#define VSL_TILE_SYNC() __syncthreads()
#endif

template <int S, bool AVG, bool OUT>
__global__ __launch_bounds__(tile_threads(fused_tile_h(S, AVG)), (fused_blocks_per_cu<S, AVG>()))
void vsl_fused_kernel(const KParams kp_in, const HeadArgs ha) {
  constexpr int TH = fused_tile_h(S, AVG), NT = tile_threads(TH);
  constexpr int TW = fused_tile_w(S, AVG);
  using F = Fused<TW, TH, NT, S, AVG, 2, OUT>;
  extern __shared__ __align__(16) float lds[];
  const int tid = threadIdx.x;
  typename F::Block bk;
  int head = 0;
  const int block_linear = swizzled_tile(ha.td, TW, TH, ha.n_heads, head, bk.b, bk.x0, bk.y0);
  if (block_linear < 0) return;
  const KParams& kp = kp_in;
  // the head's pointers and constants (a patched copy of the argument struct would end up in scratch memory)
  const HeadView hv{ha.disp[head],     ha.gphoto[head],  ha.gedge[head],    ha.ident_sel[head],
                    ha.partials[head], ha.counter[head], ha.cst_ssim[head], ha.cst_l1[head],
                    ha.want_grad[head], ha.index[head]};

  ThreadState<S> st;
#ifdef VSL_STAMPS
  if ((threadIdx.x & 63) == 0) {
    unsigned long long rt = __builtin_amdgcn_s_memrealtime();
    unsigned int hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* o = kp.stamps + ((size_t)block_linear * 8 + (threadIdx.x >> 6)) * 16;
    o[9] = rt;
    o[11] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
  // timing probes only (-DVSL_PROBE_SKIP=<bits>, results wrong): 1 = no phase 1, 2 = no phase 2, 4 = no phase 3
#ifndef VSL_PROBE_SKIP
#define VSL_PROBE_SKIP 0
#endif
  VSL_STAMP(0);
  F::phase0(kp, hv, bk, tid, st);
  VSL_STAMP(1);
  if constexpr (!(VSL_PROBE_SKIP & 1)) {
    F::phase1(kp, hv, bk, tid, lds, st);
  } else {
    st.photo = 0.0f;
    st.live = 0u;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int q = 0; q < S; ++q)
#pragma unroll
        for (int c = 0; c < 3; ++c) st.dp[k][q][c][0] = st.dp[k][q][c][1] = (float)tid;
  }
  VSL_STAMP(2);
  VSL_TILE_SYNC();
  VSL_STAMP(3);
  if constexpr (!(VSL_PROBE_SKIP & 2)) F::phase2(kp, hv, bk, tid, lds, st);
  VSL_STAMP(4);
  VSL_TILE_SYNC();
  VSL_STAMP(5);
  if constexpr (!(VSL_PROBE_SKIP & 4)) {
    F::phase3(kp, hv, bk, tid, lds, st);
  } else {
#pragma unroll
    for (int k = 0; k < F::NPART; ++k) st.part[k] = st.photo + st.dp[0][0][0][0] * (float)k;
  }
  VSL_STAMP(6);
  VSL_TILE_SYNC();
  VSL_STAMP(7);
  F::phase4_store(tid, lds, st);
  VSL_TILE_SYNC();
  F::phase4_sum_a(tid, lds);
  VSL_TILE_SYNC();
  F::phase4_sum_b(hv, block_linear, tid, lds);
  VSL_STAMP(8);
#ifdef VSL_STAMPS
  if ((threadIdx.x & 63) == 0)
    kp.stamps[((size_t)block_linear * 8 + (threadIdx.x >> 6)) * 16 + 10] = __builtin_amdgcn_s_memrealtime();
#endif
}

// --------------------------------------------------------------------------
// forward-only tile kernel (every head of the launch without gradients: validation, teacher heads)
// --------------------------------------------------------------------------
template <int S, bool AVG, bool OUT>
__global__ __launch_bounds__(tile_threads(fwd_tile_h(S, AVG))) void vsl_forward_kernel(const KParams kp_in, const HeadArgs ha) {
  constexpr int TH = fwd_tile_h(S, AVG), NT = tile_threads(TH);
  constexpr int TW = fused_tile_w(S, AVG);
  using F = Fused<TW, TH, NT, S, AVG, 1, OUT>;
  __shared__ __align__(16) float lds[F::LDS_FLOATS];
  const int tid = threadIdx.x;
  typename F::Block bk;
  int head = 0;
  const int block_linear = swizzled_tile(ha.td, TW, TH, ha.n_heads, head, bk.b, bk.x0, bk.y0);
  if (block_linear < 0) return;
  const KParams& kp = kp_in;
  const HeadView hv{ha.disp[head],     ha.gphoto[head],  ha.gedge[head],    ha.ident_sel[head],
                    ha.partials[head], ha.counter[head], ha.cst_ssim[head], ha.cst_l1[head],
                    0, ha.index[head]};
  ThreadState<S> st;
  F::phase0(kp, hv, bk, tid, st);
  F::phase1(kp, hv, bk, tid, lds, st);
  __syncthreads();
  F::phase_forward(kp, hv, bk, tid, lds, st);
  __syncthreads();
  F::phase4_store(tid, lds, st);
  __syncthreads();
  F::phase4_sum_a(tid, lds);
  __syncthreads();
  F::phase4_sum_b(hv, block_linear, tid, lds);
}

// --------------------------------------------------------------------------
// identity minimum
// --------------------------------------------------------------------------
// optional output pointers of every head (vsl_head.depth / sample / color), handed to the tile programs through a table
// in the workspace: the launch in front of them stores it
struct OutArgs {
  int n_heads;
  float* ptr[VSL_MAX_HEADS][kOutPtrs];
  TileDiv td;  // of the identity launch's own tiling
};
__device__ __forceinline__ void store_out_table(const KParams& kp, const OutArgs& oa, int tid, int nt) {
  if (!kp.out_table) return;
  for (int i = tid; i < oa.n_heads * kOutPtrs; i += nt) kp.out_table[i] = oa.ptr[i / kOutPtrs][i % kOutPtrs];
}

template <int S, bool AVG>
__global__ __launch_bounds__(kIdNT) void vsl_ident_kernel(const KParams kp, const OutArgs oa) {
  using I = Ident<kTW, kIdTH, kIdNT, S, AVG>;
  __shared__ __align__(16) float lds[I::LDS_FLOATS];
  typename I::Block bk;
  int head = 0;
  // this launch precedes the tile programs of the step: its first workgroup also writes the per-image matrices
  if (blockIdx.x == 0) store_out_table(kp, oa, threadIdx.x, kIdNT);
  if (kp.prep && blockIdx.x == 0)
    for (int i = threadIdx.x; i < kp.B * kPrepStride; i += kIdNT) kp.prep[i] = prep_entry(kp, i / kPrepStride, i % kPrepStride);
  if (swizzled_tile(oa.td, kTW, kIdTH, 1, head, bk.b, bk.x0, bk.y0) < 0) return;
  I::load(kp, bk, threadIdx.x, lds);
  __syncthreads();
  I::compute(kp, bk, threadIdx.x, lds);
}

// the per-image matrices alone, for steps that launch no identity pass (no auto-masking, or a shared identity minimum)
__global__ void vsl_prep_kernel(const KParams kp, const OutArgs oa) {
  if (blockIdx.x == 0) store_out_table(kp, oa, threadIdx.x, blockDim.x);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kp.B * kPrepStride; i += gridDim.x * blockDim.x)
    kp.prep[i] = prep_entry(kp, i / kPrepStride, i % kPrepStride);
}

// --------------------------------------------------------------------------
// finalize
// --------------------------------------------------------------------------
struct FinalizeHead {
  int n_scales;
  int tiles_per_image;  // of the tile program that wrote this head's records (the forward-only one tiles differently)
  float scale_weight[VSL_MAX_SCALES];
  float photo_w, smooth_w;  // weights of the two terms in the total
  const float* partials;
  double* img;        // [B][4]: photo sum, smoothness term, c1, c2
  float* gT;          // [kMaxSrc][B][16]
  float* c12;         // [B][2] float copies of c1, c2 for the gradient kernel
  unsigned int* counter;
  float* losses;      // [VSL_LOSS_SLOTS]
};

struct FinalizeParams {
  int B, H, W, S;
  double inv_nx, inv_ny;
  const float* K;
  FinalizeHead head[VSL_MAX_HEADS];
};

// grid (B, heads): one workgroup reduces the tile records of one image of one head
__global__ __launch_bounds__(1024) void vsl_finalize_kernel(const FinalizeParams fp) {
  __shared__ double red[16][64];
  __shared__ double tot[64];
  const FinalizeHead& fh = fp.head[blockIdx.y];
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int v = tid & 63, r = tid >> 6;  // value index, row group (16 tile records in flight per step)
  const int npart = num_partials(fp.S);
  double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  // the image's intrinsics for the pose gradients below: requested now, used after the reduction
  float kq[3] = {0.0f, 0.0f, 0.0f};
  if (tid < fp.S * 16) {
#pragma unroll
    for (int i = 0; i < 3; ++i) kq[i] = fp.K[16 * b + i * 4 + (tid % 16) / 4];
  }
  if (v < npart) {
    // Row group r takes records r, r + 16, r + 32, ...; accumulator i % 8 takes the i-th of them, in ascending order
    // (a fixed association order).  The kernel is latency bound -- 48 workgroups on an otherwise idle chip -- so up
    // to 24 loads are in flight per thread at once: one memory round trip for the 368 tiles of a 640x192 image
    // instead of three (9.2 -> 6.8 us).  The loads are unconditional (past the end: the last record again, not
    // added): as predicated loads the compiler gives each a branch of its own and they run one after the other (12 us).
    const float* base = fh.partials + (size_t)b * fh.tiles_per_image * kPartStride + v;
    const int n = fh.tiles_per_image;
    for (int t = r; t < n; t += 16 * 24) {
      float x[24];
#pragma unroll
      for (int i = 0; i < 24; ++i) x[i] = base[(size_t)(t + 16 * i < n ? t + 16 * i : n - 1) * kPartStride];
#pragma unroll
      for (int i = 0; i < 24; ++i)
        if (t + 16 * i < n) acc[i % 8] += (double)x[i];
    }
  }
  red[r][v] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (tid < 64) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[i][tid];
    tot[tid] = s;
  }
  __syncthreads();

  const double HW = (double)fp.H * fp.W;
  if (tid < fp.S * 16) {
    // d total / d T[k][j] = sum_{i<3} K[i][k] * dP[i][j]   (P = (K @ T)[:3,:], layers.py:189)
    const int s = tid / 16, j = tid % 4;  // kq[i] = K[b][i][k] with k = (tid % 16) / 4
    double g = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) g += (double)kq[i] * tot[4 + 12 * s + i * 4 + j];
    fh.gT[((size_t)s * fp.B + b) * 16 + (tid % 16)] = (float)g;
  }
  if (tid == 0) {
    const double mean = tot[1] / HW;
    const double me = mean + 1e-7;  // Trainer.py:610
    const double edge = tot[2] * fp.inv_nx + tot[3] * fp.inv_ny;
    const double c1 = (double)fh.smooth_w / me;
    const double c2 = (double)fh.smooth_w * edge / (me * me * HW);
    fh.img[b * 4 + 0] = tot[0];
    fh.img[b * 4 + 1] = edge / me;
    fh.c12[b * 2 + 0] = (float)c1;
    fh.c12[b * 2 + 1] = (float)c2;
    // publish, then take a ticket; the workgroup of the head's last image assembles the scalars
    __threadfence();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int ticket = atomicAdd(fh.counter, 1u);
    const int last = (ticket == (unsigned int)(fp.B - 1));
    if (last) {
      __threadfence();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      double photo = 0.0, smooth = 0.0;
      for (int i0 = 0; i0 < fp.B; i0 += 8) {  // eight images' records requested together, summed in image order
        double ph[8], sm[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int i = i0 + j < fp.B ? i0 + j : fp.B - 1;
          ph[j] = __hip_atomic_load(&fh.img[i * 4 + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          sm[j] = __hip_atomic_load(&fh.img[i * 4 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (i0 + j < fp.B) {
            photo += ph[j];
            smooth += sm[j];
          }
      }
      photo /= (double)fp.B * HW;
      fh.losses[VSL_LOSS_TOTAL] = (float)((double)fh.photo_w * photo + (double)fh.smooth_w * smooth);
      fh.losses[VSL_LOSS_PHOTO] = (float)photo;
      fh.losses[VSL_LOSS_SMOOTH] = (float)smooth;
      for (int i = 0; i < VSL_MAX_SCALES; ++i)
        fh.losses[VSL_LOSS_SCALE0 + i] = (i < fh.n_scales) ? (float)(photo + (double)fh.scale_weight[i] * smooth) : 0.0f;
    }
  }
}

// --------------------------------------------------------------------------
// gradient assembly
// --------------------------------------------------------------------------
struct GradHead {
  const float* gphoto;
  const float* gedge;
  const float* c12;
  const float* gT;
  const float* grad_losses;  // [VSL_LOSS_SLOTS] upstream gradients or null
  int n_scales;
  float scale_weight[VSL_MAX_SCALES];
  float photo_w, smooth_w;  // weights baked into gphoto / c1, c2 by the forward pass
  float* grad_disp;
  float* grad_T[kMaxSrc];
};

struct GradParams {
  int B, HW, S;
  GradHead head[VSL_MAX_HEADS];
};

// grid (chunks, B, heads)
__global__ __launch_bounds__(256) void vsl_grad_kernel(const GradParams gp) {
  const GradHead& gh = gp.head[blockIdx.z];
  // Every loss slot is a combination of the photometric mean and the smoothness term, so the
  // upstream gradients collapse to one factor for each of the two stored gradient parts.
  float ap = 1.0f, as = 1.0f;
  if (gh.grad_losses) {
    const float* g = gh.grad_losses;
    float wp = g[VSL_LOSS_TOTAL] * gh.photo_w + g[VSL_LOSS_PHOTO];
    float ws = g[VSL_LOSS_TOTAL] * gh.smooth_w + g[VSL_LOSS_SMOOTH];
    for (int i = 0; i < gh.n_scales; ++i) {
      wp += g[VSL_LOSS_SCALE0 + i];
      ws += g[VSL_LOSS_SCALE0 + i] * gh.scale_weight[i];
    }
    ap = wp / gh.photo_w;
    as = gh.smooth_w != 0.0f ? ws / gh.smooth_w : 0.0f;
  }
  const int b = blockIdx.y;
  if (gh.grad_disp) {
    const float c1 = as * gh.c12[b * 2 + 0], c2 = as * gh.c12[b * 2 + 1];
    const size_t off = (size_t)b * gp.HW;
    if ((gp.HW & 3) == 0) {  // 16 bytes per lane (the planes are 256-byte aligned, so is every image in them)
      const float4* gph = reinterpret_cast<const float4*>(gh.gphoto + off);
      const float4* ged = reinterpret_cast<const float4*>(gh.gedge + off);
      const bool aligned_out = (reinterpret_cast<uintptr_t>(gh.grad_disp + off) & 15) == 0;
      for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < gp.HW / 4; i += gridDim.x * blockDim.x) {
        const float4 p = gph[i], e = ged[i];
        const float4 r = make_float4(ap * p.x + (c1 * e.x - c2), ap * p.y + (c1 * e.y - c2), ap * p.z + (c1 * e.z - c2),
                                     ap * p.w + (c1 * e.w - c2));
        if (aligned_out) {
          reinterpret_cast<float4*>(gh.grad_disp + off)[i] = r;
        } else {
          float* o = gh.grad_disp + off + 4 * (size_t)i;
          o[0] = r.x;
          o[1] = r.y;
          o[2] = r.z;
          o[3] = r.w;
        }
      }
    } else {
      for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < gp.HW; i += gridDim.x * blockDim.x)
        gh.grad_disp[off + i] = ap * gh.gphoto[off + i] + (c1 * gh.gedge[off + i] - c2);
    }
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < gp.S * 16) {
    const int s = threadIdx.x / 16, e = threadIdx.x % 16;
    if (gh.grad_T[s]) gh.grad_T[s][b * 16 + e] = ap * gh.gT[((size_t)s * gp.B + b) * 16 + e];
  }
}

// --------------------------------------------------------------------------
// library-owned scratch for vsl_loss_forward_static (callers that may not allocate)
// --------------------------------------------------------------------------
// Every GPU the library runs on holds its own copy of this array until the process ends.  In host code the
// array's name is only a handle for hipGetSymbolAddress: the device address it returns (one per device) is what
// kernels and memory operations get.
constexpr size_t kStaticScratchBytes = VSL_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

constexpr int kMaxDevices = 64;
std::atomic<void*> g_scratch_ptr[kMaxDevices];
std::mutex g_scratch_mutex;

// device address of `scratch` on the calling thread's current device (looked up once per device), or null
void* static_scratch_ptr() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    fail(VSL_ERR_LAUNCH_FAILED, "vsl_loss_forward_static: no current device (or device index >= 64)");
    return nullptr;
  }
  void* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
  if (p) return p;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  p = g_scratch_ptr[dev].load(std::memory_order_relaxed);
  if (p) return p;
  const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
  if (e != hipSuccess || !p) {
    fail(VSL_ERR_LAUNCH_FAILED, std::string("hipGetSymbolAddress(scratch) failed: ") + hipGetErrorString(e));
    return nullptr;
  }
  g_scratch_ptr[dev].store(p, std::memory_order_release);
  return p;
}

// --------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------
inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }
inline unsigned swizzled_grid(int tiles) { return 8u * (unsigned)((tiles + 7) / 8); }  // see swizzled_tile()
inline unsigned div_multiplier(unsigned d) {
  const unsigned long long m = (1ull << 32) / d;
  return m > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)m;
}

struct Layout {
  int tiles_x, tiles_y, tiles_per_image, blocks;
  size_t off_partials, off_gphoto, off_gedge, off_idmin, off_img, off_gT, off_c12, off_counter, off_prep, off_outs, total;
};

inline int desc_tile_h(const vsl_loss_desc* d) {
  return fused_tile_h(d->S, (d->flags & VSL_FLAG_AVG_REPROJECTION) != 0);
}
inline int desc_tile_w(const vsl_loss_desc* d) {
  return fused_tile_w(d->S, (d->flags & VSL_FLAG_AVG_REPROJECTION) != 0);
}

Layout make_layout(int B, int H, int W, int th, int tw) {
  Layout L;
  L.tiles_x = (W + tw - 1) / tw;
  L.tiles_y = (H + th - 1) / th;
  L.tiles_per_image = L.tiles_x * L.tiles_y;
  L.blocks = L.tiles_per_image * B;
  size_t o = 0;
  L.off_counter = o;
  o += 256;
  L.off_partials = o;
  o += align256((size_t)L.blocks * kPartStride * sizeof(float));
  L.off_gphoto = o;
  o += align256((size_t)B * H * W * sizeof(float));
  L.off_gedge = o;
  o += align256((size_t)B * H * W * sizeof(float));
  L.off_idmin = o;
  o += align256((size_t)B * H * W * sizeof(float));
  L.off_img = o;
  o += align256((size_t)B * 4 * sizeof(double));
  L.off_gT = o;
  o += align256((size_t)kMaxSrc * B * 16 * sizeof(float));
  L.off_c12 = o;
  o += align256((size_t)B * 2 * sizeof(float));
  L.off_prep = o;
  o += align256((size_t)B * kPrepStride * sizeof(float));
  L.off_outs = o;
  o += align256((size_t)VSL_MAX_HEADS * kOutPtrs * sizeof(float*));
  L.total = o;
  return L;
}

TileDiv make_tile_div(const Layout& L) {
  TileDiv td;
  td.tiles_x = (unsigned)L.tiles_x;
  td.per_image = (unsigned)L.tiles_per_image;
  td.per_head = (unsigned)L.blocks;
  td.m_x = div_multiplier(td.tiles_x);
  td.m_image = div_multiplier(td.per_image);
  td.m_head = div_multiplier(td.per_head);
  return td;
}

// Limits every launch over the images of a descriptor relies on: 16-bit grid dimensions for the image index, 32-bit
// element offsets into the image tensors, 28-bit offsets within a plane and 24-bit operands of the row-times-width
// multiplies.
int check_shape(const vsl_loss_desc* d) {
  if (!d) return fail(VSL_ERR_INVALID_ARGUMENT, "desc is NULL");
  if (d->B < 1 || d->H < 2 || d->W < 2) return fail(VSL_ERR_INVALID_ARGUMENT, "need B >= 1, H >= 2, W >= 2");
  if (d->B > 65535) return fail(VSL_ERR_INVALID_ARGUMENT, "B exceeds 65535");
  if ((long long)d->B * d->H * d->W * 3 >= (1ll << 31))
    return fail(VSL_ERR_INVALID_ARGUMENT, "B*3*H*W must stay below 2^31");
  if ((long long)d->H * d->W >= (1ll << 28) || d->H >= (1 << 24) || d->W >= (1 << 24))
    return fail(VSL_ERR_INVALID_ARGUMENT, "H*W must stay below 2^28 and H, W below 2^24");
  if (d->S < 1 || d->S > VSL_MAX_SOURCES) return fail(VSL_ERR_INVALID_ARGUMENT, "S must be 1..VSL_MAX_SOURCES");
  return VSL_OK;
}

int check_desc(const vsl_loss_desc* d) {
  if (int rc = check_shape(d)) return rc;
  if (!(d->min_depth > 0.0f) || !(d->max_depth > d->min_depth))
    return fail(VSL_ERR_INVALID_ARGUMENT, "need 0 < min_depth < max_depth");
  // a zeroed or unset factor would silently return all-zero losses and gradients
  if (!(d->inv_num_scales > 0.0f) || !std::isfinite(d->inv_num_scales))
    return fail(VSL_ERR_INVALID_ARGUMENT, "inv_num_scales must be positive and finite (1 / num_scales)");
  if (!d->target || !d->K || !d->inv_K) return fail(VSL_ERR_INVALID_ARGUMENT, "NULL input tensor");
  for (int s = 0; s < d->S; ++s)
    if (!d->source[s] || !d->T[s]) return fail(VSL_ERR_INVALID_ARGUMENT, "NULL source image or pose");
  return VSL_OK;
}

template <int S, bool AVG, bool OUT>
int launch_fused(const KParams& kp, const HeadArgs& ha, dim3 grid, hipStream_t stream) {
  constexpr int TH = fused_tile_h(S, AVG), NT = tile_threads(TH);
  using F = Fused<fused_tile_w(S, AVG), TH, NT, S, AVG>;
  constexpr size_t lds_bytes = (size_t)F::LDS_FLOATS * sizeof(float);
  static_assert(lds_bytes <= 160 * 1024, "tile does not fit the 160 KiB LDS of a CU");
  // the attribute belongs to the (function, device) pair: remember it per device of the calling thread
  static thread_local unsigned long long configured = 0;  // bit d = done on device d (devices >= 64: every call)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail(VSL_ERR_LAUNCH_FAILED, "hipGetDevice failed");
  if (dev >= 64 || !((configured >> dev) & 1ull)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&vsl_fused_kernel<S, AVG, OUT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return fail(VSL_ERR_LAUNCH_FAILED, "hipFuncSetAttribute(max dynamic LDS) failed");
    if (dev < 64) configured |= 1ull << dev;
  }
  size_t launch_lds = lds_bytes;
#ifdef VSL_STAMPS
  if (const char* e = getenv("VSL_DEBUG_EXTRA_LDS")) launch_lds += (size_t)atoi(e);  // diagnostic: fewer resident blocks
#endif
  hipLaunchKernelGGL((vsl_fused_kernel<S, AVG, OUT>), grid, dim3(NT), launch_lds, stream, kp, ha);
  return VSL_OK;
}

template <int S>
void launch_forward_s(const KParams& kp, const HeadArgs& ha, bool avg, dim3 grid, hipStream_t stream) {
  constexpr int NTF = tile_threads(fwd_tile_h(S, false)), NTA = tile_threads(fwd_tile_h(S, true));
  const bool out = kp.out_table != nullptr;
  if (avg && out)
    hipLaunchKernelGGL((vsl_forward_kernel<S, true, true>), grid, dim3(NTA), 0, stream, kp, ha);
  else if (avg)
    hipLaunchKernelGGL((vsl_forward_kernel<S, true, false>), grid, dim3(NTA), 0, stream, kp, ha);
  else if (out)
    hipLaunchKernelGGL((vsl_forward_kernel<S, false, true>), grid, dim3(NTF), 0, stream, kp, ha);
  else
    hipLaunchKernelGGL((vsl_forward_kernel<S, false, false>), grid, dim3(NTF), 0, stream, kp, ha);
}

template <int S>
int launch_fused_s(const KParams& kp, const HeadArgs& ha, bool avg, dim3 grid, hipStream_t stream) {
  const bool out = kp.out_table != nullptr;
  if (avg) return out ? launch_fused<S, true, true>(kp, ha, grid, stream) : launch_fused<S, true, false>(kp, ha, grid, stream);
  return out ? launch_fused<S, false, true>(kp, ha, grid, stream) : launch_fused<S, false, false>(kp, ha, grid, stream);
}

template <int S>
void launch_ident_s(const KParams& kp, const OutArgs& oa, bool avg, dim3 grid, hipStream_t stream) {
  if (avg)
    hipLaunchKernelGGL((vsl_ident_kernel<S, true>), grid, dim3(kIdNT), 0, stream, kp, oa);
  else
    hipLaunchKernelGGL((vsl_ident_kernel<S, false>), grid, dim3(kIdNT), 0, stream, kp, oa);
}

void launch_ident(const KParams& kp, const OutArgs& oa, bool avg, dim3 grid, hipStream_t stream) {
  switch (kp.S) {
    case 1: launch_ident_s<1>(kp, oa, avg, grid, stream); break;
    case 2: launch_ident_s<2>(kp, oa, avg, grid, stream); break;
    case 3: launch_ident_s<3>(kp, oa, avg, grid, stream); break;
    default: launch_ident_s<4>(kp, oa, avg, grid, stream); break;
  }
}

// the part of the kernel parameters that only depends on the descriptor
KParams base_params(const vsl_loss_desc* d) {
  KParams kp{};
  kp.B = d->B;
  kp.H = d->H;
  kp.W = d->W;
  kp.S = d->S;
  kp.automask = (d->flags & VSL_FLAG_AUTOMASK) != 0;
  kp.no_ssim = (d->flags & VSL_FLAG_NO_SSIM) != 0;
  kp.eps = d->eps;
  kp.ssim_w3 = d->ssim_weight / 3.0f;
  kp.l1_w3 = d->l1_weight / 3.0f;
  kp.warp_scale = d->warp_scale;
  kp.tgt = d->target;
  for (int s = 0; s < kMaxSrc; ++s) {
    kp.src[s] = s < d->S ? d->source[s] : nullptr;
    kp.T[s] = s < d->S ? d->T[s] : nullptr;
  }
  kp.noise = d->noise;
  kp.seed = d->seed;
  return kp;
}

}  // namespace

extern "C" {

const char* vsl_last_error_string(void) { return vsl_err::g_err.c_str(); }

int vsl_version(void) { return 210; }  // 0.2.0: vsl_head grew to 200 bytes (optional outputs); 0.2.1: *_heads_checked

size_t vsl_loss_workspace_bytes(const vsl_loss_desc* d) {
  if (!d || d->B < 1 || d->H < 1 || d->W < 1) return 0;
  return make_layout(d->B, d->H, d->W, desc_tile_h(d), desc_tile_w(d)).total;
}

}  // extern "C"

namespace {

int check_heads(const vsl_loss_desc* d, const vsl_head* heads, int n_heads, bool backward) {
  if (!heads || n_heads < 1 || n_heads > VSL_MAX_HEADS) return fail(VSL_ERR_INVALID_ARGUMENT, "need 1..VSL_MAX_HEADS heads");
  const size_t need = make_layout(d->B, d->H, d->W, desc_tile_h(d), desc_tile_w(d)).total;
  for (int h = 0; h < n_heads; ++h) {
    const vsl_head& hd = heads[h];
    if (hd.n_scales < 1 || hd.n_scales > VSL_MAX_SCALES) return fail(VSL_ERR_INVALID_ARGUMENT, "n_scales must be 1..VSL_MAX_SCALES");
    if (!hd.workspace) return fail(VSL_ERR_INVALID_ARGUMENT, "workspace must not be NULL");
    if (((uintptr_t)hd.workspace & 255) != 0) return fail(VSL_ERR_INVALID_ARGUMENT, "workspace must be 256-byte aligned");
    if (hd.workspace_bytes < need) return fail(VSL_ERR_WORKSPACE_TOO_SMALL, "workspace smaller than vsl_loss_workspace_bytes()");
    if (!backward && (!hd.disp || !hd.losses)) return fail(VSL_ERR_INVALID_ARGUMENT, "disp and losses must not be NULL");
  }
  return VSL_OK;
}

void head_weights(const vsl_loss_desc* d, const vsl_head& hd, float& photo_w, float& smooth_w) {
  float smooth_sum = 0.0f;
  for (int i = 0; i < hd.n_scales; ++i) smooth_sum += hd.scale_weight[i];
  photo_w = (float)hd.n_scales * d->inv_num_scales;
  smooth_w = smooth_sum * d->inv_num_scales;
}

// ev0 / ev1 (optional) are recorded on the stream directly around the fused kernel launch
int forward_impl(const vsl_loss_desc* d, const vsl_head* heads, int n_heads, void* stream_, hipEvent_t ev0, hipEvent_t ev1) {
  if (int rc = check_desc(d)) return rc;
  if (int rc = check_heads(d, heads, n_heads, false)) return rc;
  const Layout L = make_layout(d->B, d->H, d->W, desc_tile_h(d), desc_tile_w(d));
  // the forward-only program's own tiling (never more tiles than the full program's: the records fit the same region)
  const Layout LF = make_layout(d->B, d->H, d->W, fwd_tile_h(d->S, (d->flags & VSL_FLAG_AVG_REPROJECTION) != 0), desc_tile_w(d));
  hipStream_t stream = (hipStream_t)stream_;

  const bool avg = (d->flags & VSL_FLAG_AVG_REPROJECTION) != 0;
  const bool automask = (d->flags & VSL_FLAG_AUTOMASK) != 0;
  const bool no_ssim = (d->flags & VSL_FLAG_NO_SSIM) != 0;
  const bool want_grad = (d->flags & VSL_FLAG_NO_GRAD) == 0;
  const double npix = (double)d->B * d->H * d->W;

  KParams kp = base_params(d);
  // python floats: min_disp = 1/max_depth, (max_disp - min_disp), then cast to float32 (layers.py:21-23)
  kp.a_disp = (float)(1.0 / (double)d->max_depth);
  kp.c_disp = (float)(1.0 / (double)d->min_depth - 1.0 / (double)d->max_depth);
  kp.inv_nx = (float)(1.0 / ((double)d->B * d->H * (d->W - 1)));
  kp.inv_ny = (float)(1.0 / ((double)d->B * (d->H - 1) * d->W));
  kp.ix_scale = (float)d->W / (float)(d->W - 1);
  kp.iy_scale = (float)d->H / (float)(d->H - 1);
  kp.K = d->K;
  kp.invK = d->inv_K;
  kp.want_grad = want_grad;

  HeadArgs ha{};
  FinalizeParams fp{};
  ha.n_heads = n_heads;
  fp.B = d->B;
  fp.H = d->H;
  fp.W = d->W;
  fp.S = d->S;
  fp.inv_nx = 1.0 / ((double)d->B * d->H * (d->W - 1));
  fp.inv_ny = 1.0 / ((double)d->B * (d->H - 1) * d->W);
  fp.K = d->K;
  for (int h = 0; h < n_heads; ++h) {
    const vsl_head& hd = heads[h];
    char* ws = (char*)hd.workspace;
    float photo_w, smooth_w;
    head_weights(d, hd, photo_w, smooth_w);
    const double base = (double)d->warp_scale * photo_w / npix / (avg ? (double)d->S : 1.0);
    ha.disp[h] = hd.disp;
    ha.gphoto[h] = (float*)(ws + L.off_gphoto);
    ha.gedge[h] = (float*)(ws + L.off_gedge);
    ha.ident_sel[h] = hd.identity_selection;
    ha.partials[h] = (float*)(ws + L.off_partials);
    ha.counter[h] = (unsigned int*)(ws + L.off_counter);
    ha.cst_ssim[h] = no_ssim ? 0.0f : (float)(base * d->ssim_weight / 27.0);
    ha.cst_l1[h] = no_ssim ? (float)(base / 3.0) : (float)(base * d->l1_weight / 3.0);
    ha.want_grad[h] = want_grad && !(hd.flags & VSL_HEAD_NO_GRAD);
    ha.index[h] = h;
    FinalizeHead& fh = fp.head[h];
    fh.n_scales = hd.n_scales;
    fh.tiles_per_image = ha.want_grad[h] ? L.tiles_per_image : LF.tiles_per_image;
    for (int i = 0; i < VSL_MAX_SCALES; ++i) fh.scale_weight[i] = i < hd.n_scales ? hd.scale_weight[i] : 0.0f;
    fh.photo_w = photo_w;
    fh.smooth_w = smooth_w;
    fh.partials = ha.partials[h];
    fh.img = (double*)(ws + L.off_img);
    fh.gT = (float*)(ws + L.off_gT);
    fh.c12 = (float*)(ws + L.off_c12);
    fh.counter = ha.counter[h];
    fh.losses = hd.losses;
  }

#ifdef VSL_STAMPS
  {
    const size_t need = (size_t)L.blocks * n_heads * 8 * 16 * sizeof(unsigned long long);
    if (g_stamp_bytes < need) {
      if (g_stamps) (void)hipFree(g_stamps);
      if (hipMalloc((void**)&g_stamps, need) != hipSuccess) return fail(VSL_ERR_LAUNCH_FAILED, "stamp buffer");
      g_stamp_bytes = need;
    }
    kp.stamps = g_stamps;
  }
#endif
  // per-image projection matrices, shared by all heads: written by the identity pass when there is one
  kp.prep = (float*)((char*)heads[0].workspace + L.off_prep);
  // optional outputs (depth, sampling grids, warped frames): their pointers travel through a table in the workspace
  OutArgs oa{};
  oa.n_heads = n_heads;
  bool any_out = false;
  for (int h = 0; h < n_heads; ++h) {
    oa.ptr[h][0] = heads[h].depth;
    for (int s = 0; s < kMaxSrc; ++s) {
      oa.ptr[h][1 + s] = s < d->S ? heads[h].sample[s] : nullptr;
      oa.ptr[h][1 + kMaxSrc + s] = s < d->S ? heads[h].color[s] : nullptr;
    }
    for (int i = 0; i < kOutPtrs; ++i) any_out = any_out || oa.ptr[h][i] != nullptr;
  }
  kp.out_table = any_out ? (float**)((char*)heads[0].workspace + L.off_outs) : nullptr;
  if (automask && !d->identity_min) {  // computed once into the first head's workspace and read by every head
    kp.idmin = (float*)((char*)heads[0].workspace + L.off_idmin);
    const Layout LI = make_layout(d->B, d->H, d->W, kIdTH, kTW);
    oa.td = make_tile_div(LI);
    launch_ident(kp, oa, avg, dim3(swizzled_grid(LI.blocks)), stream);
  } else {
    if (automask) kp.idmin = const_cast<float*>(d->identity_min);
    hipLaunchKernelGGL(vsl_prep_kernel, dim3((d->B * kPrepStride + 255) / 256 > 64 ? 64 : (d->B * kPrepStride + 255) / 256), dim3(256), 0,
                       stream, kp, oa);
  }
  int rc = VSL_OK;
  if (ev0) (void)hipEventRecord(ev0, stream);
  // Heads with gradients go through the full tile program, forward-only heads (a teacher head beside a student's)
  // through the forward-only one (halo 1, no coefficient planes: about half the time per head): two grids when
  // a call has both kinds.
  HeadArgs hg{}, hf{};
  hg.td = make_tile_div(L);
  hf.td = make_tile_div(LF);
  auto take = [&](HeadArgs& dst, int h) {
    const int i = dst.n_heads++;
    dst.disp[i] = ha.disp[h];
    dst.gphoto[i] = ha.gphoto[h];
    dst.gedge[i] = ha.gedge[h];
    dst.ident_sel[i] = ha.ident_sel[h];
    dst.partials[i] = ha.partials[h];
    dst.counter[i] = ha.counter[h];
    dst.cst_ssim[i] = ha.cst_ssim[h];
    dst.cst_l1[i] = ha.cst_l1[h];
    dst.want_grad[i] = ha.want_grad[h];
    dst.index[i] = h;
  };
  for (int h = 0; h < n_heads; ++h) take(ha.want_grad[h] ? hg : hf, h);
  if (hg.n_heads) {
    const dim3 gg(swizzled_grid(L.blocks * hg.n_heads));
    switch (d->S) {
      case 1: rc = launch_fused_s<1>(kp, hg, avg, gg, stream); break;
      case 2: rc = launch_fused_s<2>(kp, hg, avg, gg, stream); break;
      case 3: rc = launch_fused_s<3>(kp, hg, avg, gg, stream); break;
      default: rc = launch_fused_s<4>(kp, hg, avg, gg, stream); break;
    }
  }
  if (hf.n_heads && rc == VSL_OK) {
    const dim3 gf(swizzled_grid(LF.blocks * hf.n_heads));
    switch (d->S) {
      case 1: launch_forward_s<1>(kp, hf, avg, gf, stream); break;
      case 2: launch_forward_s<2>(kp, hf, avg, gf, stream); break;
      case 3: launch_forward_s<3>(kp, hf, avg, gf, stream); break;
      default: launch_forward_s<4>(kp, hf, avg, gf, stream); break;
    }
  }
  if (ev1) (void)hipEventRecord(ev1, stream);
  if (rc) return rc;

  hipLaunchKernelGGL(vsl_finalize_kernel, dim3(d->B, n_heads), dim3(1024), 0, stream, fp);
  if (hipError_t e = hipGetLastError(); e != hipSuccess)
    return fail(VSL_ERR_LAUNCH_FAILED, std::string("kernel launch failed: ") + hipGetErrorString(e));
  return VSL_OK;
}

int backward_impl(const vsl_loss_desc* d, const vsl_head* heads, int n_heads, void* stream_) {
  if (int rc = check_desc(d)) return rc;
  if (d->flags & VSL_FLAG_NO_GRAD) return fail(VSL_ERR_INVALID_ARGUMENT, "forward ran with VSL_FLAG_NO_GRAD");
  if (int rc = check_heads(d, heads, n_heads, true)) return rc;
  const Layout L = make_layout(d->B, d->H, d->W, desc_tile_h(d), desc_tile_w(d));
  GradParams gp{};
  gp.B = d->B;
  gp.HW = d->H * d->W;
  gp.S = d->S;
  for (int h = 0; h < n_heads; ++h) {
    const vsl_head& hd = heads[h];
    const char* ws = (const char*)hd.workspace;
    GradHead& gh = gp.head[h];
    if (hd.flags & VSL_HEAD_NO_GRAD) {  // forward-only head: nothing was prepared, nothing is written
      if (hd.grad_disp || hd.grad_T[0] || hd.grad_T[1] || hd.grad_T[2] || hd.grad_T[3])
        return fail(VSL_ERR_INVALID_ARGUMENT, "gradient requested from a VSL_HEAD_NO_GRAD head");
      continue;
    }
    gh.gphoto = (const float*)(ws + L.off_gphoto);
    gh.gedge = (const float*)(ws + L.off_gedge);
    gh.c12 = (const float*)(ws + L.off_c12);
    gh.gT = (const float*)(ws + L.off_gT);
    gh.grad_losses = hd.grad_losses;
    gh.n_scales = hd.n_scales;
    for (int i = 0; i < VSL_MAX_SCALES; ++i) gh.scale_weight[i] = i < hd.n_scales ? hd.scale_weight[i] : 0.0f;
    head_weights(d, hd, gh.photo_w, gh.smooth_w);
    gh.grad_disp = hd.grad_disp;
    for (int s = 0; s < kMaxSrc; ++s) gh.grad_T[s] = s < d->S ? hd.grad_T[s] : nullptr;
  }
  int bx = ((gp.HW & 3) == 0 ? gp.HW / 4 + 255 : gp.HW + 255) / 256;
  if (bx > 512) bx = 512;
  hipLaunchKernelGGL(vsl_grad_kernel, dim3(bx, d->B, n_heads), dim3(256), 0, (hipStream_t)stream_, gp);
  if (hipError_t e = hipGetLastError(); e != hipSuccess)
    return fail(VSL_ERR_LAUNCH_FAILED, std::string("kernel launch failed: ") + hipGetErrorString(e));
  return VSL_OK;
}

// the single-head entry points describe their head inside the descriptor
vsl_head head_of_desc(const vsl_loss_desc* d, void* workspace, size_t workspace_bytes) {
  vsl_head hd{};
  hd.disp = d->disp;
  hd.workspace = workspace;
  hd.workspace_bytes = workspace_bytes;
  hd.n_scales = d->n_scales;
  for (int i = 0; i < VSL_MAX_SCALES; ++i) hd.scale_weight[i] = d->scale_weight[i];
  return hd;
}

// diagnostic: the shared-divisor division of the tile program on arrays, two quotients per prepared divisor
__global__ void vsl_debug_quot_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ q, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const Divisor d = divisor(b[i]);
    q[i] = quot(a[i], d);
  }
}

}  // namespace

extern "C" {

#ifdef VSL_STAMPS
// diagnostic build: copy the phase time stamps of the last fused launch to the host
long long vsl_debug_read_stamps(void* host_out, long long max_bytes) {
  if (!g_stamps) return 0;
  (void)hipDeviceSynchronize();
  const size_t n = g_stamp_bytes < (size_t)max_bytes ? g_stamp_bytes : (size_t)max_bytes;
  if (hipMemcpy(host_out, g_stamps, n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (long long)n;
}
#endif

// diagnostic: resident workgroups per CU the runtime predicts for the fused kernel
int vsl_debug_quot(const float* a, const float* b, float* q, long long n, void* stream) {
  if (!a || !b || !q || n < 0) return fail(VSL_ERR_INVALID_ARGUMENT, "debug_quot: bad arguments");
  if (n == 0) return VSL_OK;
  const long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(vsl_debug_quot_kernel, dim3((unsigned)(blocks > 65535 ? 65535 : blocks)), dim3(256), 0, (hipStream_t)stream,
                     a, b, q, n);
  if (hipError_t e = hipGetLastError(); e != hipSuccess)
    return fail(VSL_ERR_LAUNCH_FAILED, std::string("kernel launch failed: ") + hipGetErrorString(e));
  return VSL_OK;
}

int vsl_debug_fused_occupancy(int S, int avg) {
  int n = -1;
  auto q = [&](auto kern, int nt, size_t lds) { (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, nt, lds); };
#define VSL_OCC(S_, A_)                                                                     \
  q(vsl_fused_kernel<S_, A_, false>, tile_threads(fused_tile_h(S_, A_)),                     \
    Fused<fused_tile_w(S_, A_), fused_tile_h(S_, A_), tile_threads(fused_tile_h(S_, A_)), S_, A_>::LDS_FLOATS * sizeof(float))
  switch (S * 2 + (avg ? 1 : 0)) {
    case 2: VSL_OCC(1, false); break;
    case 3: VSL_OCC(1, true); break;
    case 4: VSL_OCC(2, false); break;
    case 5: VSL_OCC(2, true); break;
    case 6: VSL_OCC(3, false); break;
    case 7: VSL_OCC(3, true); break;
    case 8: VSL_OCC(4, false); break;
    case 9: VSL_OCC(4, true); break;
    default: break;
  }
#undef VSL_OCC
  return n;
}

int vsl_loss_forward(const vsl_loss_desc* d, void* workspace, size_t workspace_bytes, float* losses,
                     float* identity_selection, void* stream_) {
  if (!d) return fail(VSL_ERR_INVALID_ARGUMENT, "desc is NULL");
  vsl_head hd = head_of_desc(d, workspace, workspace_bytes);
  hd.losses = losses;
  hd.identity_selection = identity_selection;
  return forward_impl(d, &hd, 1, stream_, nullptr, nullptr);
}

int vsl_loss_forward_static(const vsl_loss_desc* d, float* losses, float* identity_selection, void* stream_) {
  if (!d) return fail(VSL_ERR_INVALID_ARGUMENT, "desc is NULL");
  if (int rc = check_desc(d)) return rc;
  // forward only: the gradient planes would have to outlive the call in memory the caller cannot name
  vsl_loss_desc fd = *d;
  fd.flags |= VSL_FLAG_NO_GRAD;
  const size_t need = make_layout(fd.B, fd.H, fd.W, desc_tile_h(&fd), desc_tile_w(&fd)).total;
  if (need > kStaticScratchBytes)
    return fail(VSL_ERR_WORKSPACE_TOO_SMALL,
                "vsl_loss_forward_static: this shape needs " + std::to_string(need) + " bytes of scratch, the library holds " +
                    std::to_string(kStaticScratchBytes) + " (VSL_STATIC_SCRATCH_BYTES); use vsl_loss_forward with a workspace");
  void* ws = static_scratch_ptr();
  if (!ws) return VSL_ERR_LAUNCH_FAILED;
  vsl_head hd = head_of_desc(&fd, ws, kStaticScratchBytes);
  hd.losses = losses;
  hd.identity_selection = identity_selection;
  return forward_impl(&fd, &hd, 1, stream_, nullptr, nullptr);
}

int vsl_loss_forward_heads(const vsl_loss_desc* d, const vsl_head* heads, int n_heads, void* stream_) {
  if (!d) return fail(VSL_ERR_INVALID_ARGUMENT, "desc is NULL");
  return forward_impl(d, heads, n_heads, stream_, nullptr, nullptr);
}

int vsl_loss_backward_heads(const vsl_loss_desc* d, const vsl_head* heads, int n_heads, void* stream_) {
  if (!d) return fail(VSL_ERR_INVALID_ARGUMENT, "desc is NULL");
  return backward_impl(d, heads, n_heads, stream_);
}

// the caller's idea of the two structures against the library's: a binary compiled against another layout is refused
// instead of having its table misread (and written through)
static int check_struct_sizes(size_t desc_bytes, size_t head_bytes) {
  if (desc_bytes != sizeof(vsl_loss_desc) || head_bytes != sizeof(vsl_head))
    return fail(VSL_ERR_INVALID_ARGUMENT,
                "structure sizes differ: caller vsl_loss_desc " + std::to_string(desc_bytes) + " / vsl_head " +
                    std::to_string(head_bytes) + " bytes, library " + std::to_string(sizeof(vsl_loss_desc)) + " / " +
                    std::to_string(sizeof(vsl_head)) + " (recompile the caller against this library's vsloss.h)");
  return VSL_OK;
}

int vsl_loss_forward_heads_checked(const vsl_loss_desc* d, size_t desc_bytes, const vsl_head* heads, size_t head_bytes,
                                   int n_heads, void* stream_) {
  if (int rc = check_struct_sizes(desc_bytes, head_bytes)) return rc;
  return vsl_loss_forward_heads(d, heads, n_heads, stream_);
}

int vsl_loss_backward_heads_checked(const vsl_loss_desc* d, size_t desc_bytes, const vsl_head* heads, size_t head_bytes,
                                    int n_heads, void* stream_) {
  if (int rc = check_struct_sizes(desc_bytes, head_bytes)) return rc;
  return vsl_loss_backward_heads(d, heads, n_heads, stream_);
}

// diagnostic / benchmarking: average duration in ms of the fused tile kernel alone over `reps` forward
// calls, measured with HIP events recorded on `stream` directly around that launch.  The calls are enqueued back
// to back, each with an event pair of its own, and the host waits once at the end: waiting after every call lets
// the chip idle between launches, and the launches then run some 4 % longer than in a stream of steps.
int vsl_debug_time_fused(const vsl_loss_desc* d, const vsl_head* heads, int n_heads, int reps, float* avg_ms, void* stream_) {
  if (!d) return fail(VSL_ERR_INVALID_ARGUMENT, "desc is NULL");
  if (reps < 1 || reps > 4096 || !avg_ms) return fail(VSL_ERR_INVALID_ARGUMENT, "1 <= reps <= 4096 and avg_ms required");
  std::vector<hipEvent_t> ev(2 * (size_t)reps, nullptr);
  int rc = VSL_OK;
  for (auto& e : ev)
    if (hipEventCreate(&e) != hipSuccess) {
      rc = fail(VSL_ERR_LAUNCH_FAILED, "hipEventCreate failed");
      break;
    }
  int done = 0;
  for (; done < reps && rc == VSL_OK; ++done) rc = forward_impl(d, heads, n_heads, stream_, ev[2 * done], ev[2 * done + 1]);
  double total = 0.0;
  if (rc == VSL_OK) {
    if (hipEventSynchronize(ev[2 * (size_t)reps - 1]) != hipSuccess) rc = fail(VSL_ERR_LAUNCH_FAILED, "event timing failed");
    for (int i = 0; i < reps && rc == VSL_OK; ++i) {
      float ms = 0.0f;
      if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) != hipSuccess) rc = fail(VSL_ERR_LAUNCH_FAILED, "event timing failed");
      total += ms;
    }
  } else {
    (void)hipStreamSynchronize((hipStream_t)stream_);
  }
  for (auto& e : ev)
    if (e) (void)hipEventDestroy(e);
  *avg_ms = (float)(total / reps);
  return rc;
}

int vsl_identity_min(const vsl_loss_desc* d, float* identity_min, void* stream_) {
  if (int rc = check_shape(d)) return rc;
  if (!d->target || !identity_min) return fail(VSL_ERR_INVALID_ARGUMENT, "NULL tensor");
  for (int s = 0; s < d->S; ++s)
    if (!d->source[s]) return fail(VSL_ERR_INVALID_ARGUMENT, "NULL source image");
  KParams kp = base_params(d);
  kp.idmin = identity_min;
  const Layout L = make_layout(d->B, d->H, d->W, kIdTH, kTW);
  OutArgs oa{};
  oa.td = make_tile_div(L);
  launch_ident(kp, oa, (d->flags & VSL_FLAG_AVG_REPROJECTION) != 0, dim3(swizzled_grid(L.blocks)), (hipStream_t)stream_);
  if (hipError_t e = hipGetLastError(); e != hipSuccess)
    return fail(VSL_ERR_LAUNCH_FAILED, std::string("kernel launch failed: ") + hipGetErrorString(e));
  return VSL_OK;
}

int vsl_loss_backward(const vsl_loss_desc* d, const void* workspace, size_t workspace_bytes,
                      const float* grad_losses, float* grad_disp, float* const* grad_T, void* stream_) {
  if (!d) return fail(VSL_ERR_INVALID_ARGUMENT, "desc is NULL");
  vsl_head hd = head_of_desc(d, const_cast<void*>(workspace), workspace_bytes);
  hd.grad_losses = grad_losses;
  hd.grad_disp = grad_disp;
  for (int s = 0; s < VSL_MAX_SOURCES; ++s) hd.grad_T[s] = (grad_T && s < d->S) ? grad_T[s] : nullptr;
  return backward_impl(d, &hd, 1, stream_);
}

}  // extern "C"
