// Tile programs of the view-synthesis loss (forward + gradient in one pass).
//
// Ident  -- identity-reprojection error of every un-warped source, plus tie-break noise,
//           minimum over sources.  Independent of disparity and pose: one launch per batch
//           serves every disparity head.                    (Trainer.py:553-566,587-590)
//
// Fused  -- one workgroup owns a TW x TH tile of target pixels of one batch element and runs
//           barrier-separated phases over LDS planes:
//   phase 0  projection matrices P = (K T)[:3,:] and inv_K into LDS        (layers.py:189)
//   phase 1  every position of the tile grown by 2 (region R2): disparity -> depth ->
//            back-projection -> projection -> bilinear sample of each source; target and
//            warped values go to LDS, the tap derivatives of the thread's own two interior
//            pixels stay in registers.          (layers.py:16-25,161-202, Trainer.py:467)
//   phase 2  photometric error of every warped source on the tile grown by 1 (region R1):
//            3x3 SSIM with reflection padding + L1, running minimum / argmin against the
//            identity minimum, SSIM gradient coefficients of the winner into LDS.
//                                     (layers.py:245-262, Trainer.py:481-495,592-605)
//   phase 3  interior pixels: 3x3 gather of the coefficients -> dL/dwarped -> bilinear /
//            projection chain -> dL/ddisp and the 3x4 projection gradient; edge-aware
//            smoothness terms.                  (layers.py:211-226, Trainer.py:609-614)
//   phase 4  deterministic block reduction of the partial sums.
//
// The phases are plain functions of (thread id, LDS base, per-thread state) so that the same
// source also compiles as an ordinary C++ program that runs the threads of a block one after
// the other (the CPU test build: address-sanitizer and parity checks; not shipped).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define VSL_DEV __device__ __forceinline__
#define VSL_FMUL(a, b) __fmul_rn((a), (b))
#define VSL_FADD(a, b) __fadd_rn((a), (b))
#define VSL_FSUB(a, b) __fsub_rn((a), (b))
#define VSL_FDIV(a, b) __fdiv_rn((a), (b))
#define VSL_EXP(x) __expf(x)
// keeps the instruction scheduler from interleaving the unrolled per-channel / per-source bodies
// (interleaving them multiplies the number of live registers)
#ifdef VSL_NO_FENCE
#define VSL_SCHED_FENCE()
#else
#define VSL_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#else
// host build: compile with -ffp-contract=off so that a*b+c is never fused
#define VSL_DEV inline
#define VSL_FMUL(a, b) ((a) * (b))
#define VSL_FADD(a, b) ((a) + (b))
#define VSL_FSUB(a, b) ((a) - (b))
#define VSL_FDIV(a, b) ((a) / (b))
#define VSL_EXP(x) expf(x)
#define VSL_SCHED_FENCE()
#endif

#ifndef VSL_NC
#define VSL_NC 0
#endif

namespace vsl {

constexpr int kMaxSrc = 4;
// Tile geometry shared by the HIP kernels and the CPU test build.  A workgroup owns a kTileW x TH tile and has
// one thread per horizontal pixel pair of the tile grown by 2: 16 * (TH + 4) threads.  VSL_TILE_H selects the
// tile height of the two-source configuration; 12 (four 4-wave workgroups per CU) measured as fast as 28 (two
// 8-wave workgroups per CU, 14 % less halo work but longer barrier waits).  The configurations with more LDS
// planes always use 12 so that their planes fit.
#ifndef VSL_TILE_H
#define VSL_TILE_H 12
#endif
constexpr int kTileW = 28, kIdentTileH = 12;
#ifndef VSL_TILE_H3
#define VSL_TILE_H3 12
#endif
constexpr int fused_tile_h(int S, bool avg) { return (S <= 2 && !avg) ? VSL_TILE_H : VSL_TILE_H3; }
// Tile width of the fused programs.  Three sources need 13 planes over R2: at 28 columns the LDS of a workgroup is
// 42.4 KiB and three workgroups fit a CU, at 26 columns it is 39.7 KiB and four fit.  Three per CU is the faster
// choice: the 168 registers of three waves per SIMD hold the 36 tap derivatives (Fused::CARRY, 147 used), which
// saves phase 3 its second round of gathers, and a 28-wide tile has 7 % more outputs per workgroup (1024x320 b12
// stereo: 253 -> 224 us).
#ifndef VSL_TILE_W3
#define VSL_TILE_W3 28
#endif
#ifndef VSL_TILE_W2
#define VSL_TILE_W2 kTileW
#endif
constexpr int fused_tile_w(int S, bool avg) { return (S == 3 && !avg) ? VSL_TILE_W3 : ((S <= 2 && !avg) ? VSL_TILE_W2 : kTileW); }
constexpr int tile_threads(int th) { return 16 * (th + 4); }
// Tile height of the forward-only program (halo 1, no coefficient planes: 36 KB of LDS and 62 registers at 28 rows, so
// four eight-wave workgroups fill a CU's 32 wave slots).  Taller tiles cut its halo work from 1.25 to 1.15 x the output
// pixels (1280x384 b8, forward-only head: 87.3 -> 82.3 us); the full program loses a factor of two with the same tiles
// (two of its waves then share a SIMD and are always in the same phase), so the two programs tile the image differently.
// Two sources, per-pixel minimum only: the other instantiations (more planes, or more registers) keep the full program's height.
#ifndef VSL_FWD_TILE_H
#define VSL_FWD_TILE_H 28
#endif
constexpr int fwd_tile_h(int S, bool avg) { return (S == 2 && !avg) ? VSL_FWD_TILE_H : fused_tile_h(S, avg); }
constexpr float kC1 = 0.01f * 0.01f;
constexpr float kC2 = 0.03f * 0.03f;

// number of per-block partial sums: photo, sum(disp), Sx, Sy, then 12 per source
constexpr int num_partials(int S) { return 4 + 12 * S; }
constexpr int kPartStride = 4 + 12 * kMaxSrc;  // stride of one block's record in the workspace
constexpr int kPrepStride = 64;                 // floats per image in KParams::prep: 12 per source, then inv_K[:3,:3]
constexpr int kPrepInvK = 12 * kMaxSrc;
constexpr int kOutPtrs = 1 + 2 * kMaxSrc;       // optional outputs of a head: depth, sample grid and warped frame per source

struct KParams {
  int B, H, W, S;
  int automask, no_ssim;
  float a_disp, c_disp;  // scaled = a + c * disp
  float eps;
  float cst_ssim;  // weight of (sum A + x sum B + y sum G) in dL/dwarped
  float cst_l1;    // weight of -sign(y - x) in dL/dwarped for the selected source
  float ssim_w3, l1_w3;  // ssim_weight/3, l1_weight/3 (channel means folded in)
  float warp_scale;
  float inv_nx, inv_ny;  // 1/(B*H*(W-1)), 1/(B*(H-1)*W)
  float ix_scale, iy_scale;  // W/(W-1), H/(H-1): d(sample position)/d(pixel coordinate u, v) of the warp
  const float* disp;
  const float* tgt;
  const float* src[kMaxSrc];
  const float* K;
  const float* invK;
  const float* T[kMaxSrc];
  const float* noise;
  unsigned long long seed;
  float* idmin;      // [B,H,W] min over sources of the identity-reprojection error (+ tie-break noise):
                     // written by the identity tile program, read by the fused one
  float* gphoto;     // [B,H,W] d(photo_w * photo)/d disp
  float* gedge;      // [B,H,W] edge part of the smoothness gradient (before the per-image factors)
  float* ident_sel;  // [B,H,W] or null
  float* partials;   // [blocks][kPartStride]
  unsigned int* counter;  // arrival counter of the finalize kernel, cleared here
  float* prep;       // [B][kPrepStride] per-image projection matrices and inv_K, written by prepare_matrices() in a
                     // launch before the tile programs, or null: every workgroup then forms them itself
  int want_grad;
  float** out_table;  // [heads][kOutPtrs] output pointers (each may be null), written by the launch before the tile
                      // programs; null: no head wants outputs
#ifdef VSL_STAMPS
  unsigned long long* stamps;  // diagnostic build: [blocks][waves][16] shader-clock samples
#endif
};

// The per-head part of the arguments as the fused tile program sees it (several disparity heads share one
// KParams; the kernel picks the head of its tile).
struct HeadView {
  const float* disp;
  float* gphoto;
  float* gedge;
  float* ident_sel;
  float* partials;
  unsigned int* counter;
  float cst_ssim, cst_l1;
  int want_grad;  // 0: forward only (no coefficients, no gradient planes)
  int head;       // index into KParams::out_table
};
VSL_DEV HeadView head_view(const KParams& kp) {
  return HeadView{kp.disp, kp.gphoto, kp.gedge, kp.ident_sel, kp.partials, kp.counter, kp.cst_ssim, kp.cst_l1, kp.want_grad, 0};
}

// ----------------------------------------------------------------------------
// small helpers
// ----------------------------------------------------------------------------
VSL_DEV int reflect_clamp(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  if (i < 0) i = 0;
  if (i > n - 1) i = n - 1;
  return i;
}

VSL_DEV float fast_rcp(float x) {
#if defined(__HIPCC__) && !defined(VSL_EXACT_RCP)
  return __builtin_amdgcn_rcpf(x);  // 1 ulp; the results that must be bit-exact use VSL_FDIV
#else
  return 1.0f / x;
#endif
}

// sign(v) in {-1, 0, 1}.  On the device: two exact scalings by 2^126 take every non-zero finite value (denormals
// included) beyond 1 in magnitude, the median with -1 and 1 clamps it: three instructions instead of two
// compares, two selects and a subtraction; same values.
// Correctly rounded float32 division with the divisor's part of the work done once.  On the device `a / b`
// expands to: scale both operands, reciprocal, two refinement steps of the reciprocal, quotient, two residual
// corrections, special-case fix-up -- 11 instructions, of which the reciprocal and its refinement depend on the
// divisor only.  Divisor prepares them; quot() is the rest (6 instructions).  The operand scaling is left out: it
// only acts when an operand or the quotient leaves about [2^-96, 2^96], which the quantities divided here
// (camera coordinates by depth + eps, pixel coordinates by W-1 / H-1, 1 by the scaled disparity) cannot reach
// unless they are zero, infinite or NaN, and those the fix-up instruction resolves as IEEE division does.
// tests/test_gpu_layers.py checks it bit for bit against float32 division.  Host build: plain division.
struct Divisor {
  float b, r;
};
VSL_DEV Divisor divisor(float b) {
#if defined(__HIPCC__)
  float r = __builtin_amdgcn_rcpf(b);
  const float e = fmaf(-b, r, 1.0f);
  r = fmaf(e, r, r);
  return Divisor{b, r};
#else
  return Divisor{b, 0.0f};
#endif
}
VSL_DEV float quot(float a, const Divisor& d) {
#if defined(__HIPCC__)
  float q = a * d.r;
  float e = fmaf(-d.b, q, a);
  q = fmaf(e, d.r, q);
  e = fmaf(-d.b, q, a);
  q = fmaf(e, d.r, q);
  return __builtin_amdgcn_div_fixupf(q, d.b, a);
#else
  return a / d.b;
#endif
}

VSL_DEV float sgn(float v) {
#if defined(__HIPCC__)
  return __builtin_amdgcn_fmed3f(v * 0x1p126f * 0x1p126f, -1.0f, 1.0f);
#else
  return (float)(v > 0.0f) - (float)(v < 0.0f);
#endif
}

// Tie-break noise when the caller passes no noise tensor: a counter-based generator, N(0,1) per (seed, plane,
// pixel).  The part that depends on the seed and the plane (image x source) is uniform over a workgroup and runs
// on the scalar unit (splitmix64 finaliser); per pixel a 32-bit mixer (two multiplies) gives 24 bits for the radius
// and, after one more multiply, 24 bits for the angle of a Box-Muller transform.  On the device the logarithm,
// square root and cosine are the hardware's approximations (the cosine instruction takes revolutions): the values
// only separate candidates that are equal to within 1e-5, so a few ulp do not matter; the cost does, this runs
// once per pixel and source.
VSL_DEV unsigned long long noise_key(unsigned long long seed, unsigned plane) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)plane + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// Both outputs of one Box-Muller transform: two independent N(0,1) values per (key, pixel).  The identity tile program
// draws the noise of two source planes of a pixel from one transform (one hash, logarithm and square root for both).
struct Normal2 {
  float a, b;
};
VSL_DEV Normal2 hash_normal2(unsigned long long key, unsigned pix) {
  unsigned x = pix ^ (unsigned)key;
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  x += (unsigned)(key >> 32);
  const unsigned y = x * 0x9E3779B1u;
  const float u1 = ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0,1]
  const float u2 = (float)(y >> 8) * (1.0f / 16777216.0f);           // [0,1)
#if defined(__HIP_DEVICE_COMPILE__)
  // -2 ln(u1) = -2 ln(2) log2(u1)
  const float r = __builtin_amdgcn_sqrtf(-1.38629436112f * __builtin_amdgcn_logf(u1));
  return Normal2{r * __builtin_amdgcn_cosf(u2), r * __builtin_amdgcn_sinf(u2)};
#else
  const float r = sqrtf(-2.0f * logf(u1));
  return Normal2{r * cosf(6.28318530718f * u2), r * sinf(6.28318530718f * u2)};
#endif
}
VSL_DEV float hash_normal(unsigned long long key, unsigned pix) { return hash_normal2(key, pix).a; }

// Projection matrix P = (K @ T)[:3,:] with separate multiply and add roundings
// (what a batched 4x4 matmul on the CPU produces), layers.py:189.
VSL_DEV float proj_entry(const float* K, const float* T, int i, int j) {
  float acc = VSL_FMUL(K[i * 4 + 0], T[0 * 4 + j]);
  for (int q = 1; q < 4; ++q) acc = VSL_FADD(acc, VSL_FMUL(K[i * 4 + q], T[q * 4 + j]));
  return acc;
}

// kp.src[s] / kp.T[s] through compile-time indices only: a run-time index into the by-value argument struct
// (a source loop the compiler did not unroll) would move the whole struct into scratch memory
VSL_DEV const float* src_image(const KParams& kp, int s) {
  const float* p = kp.src[0];
  if (s == 1) p = kp.src[1];
  if (s == 2) p = kp.src[2];
  if (s == 3) p = kp.src[3];
  return p;
}
VSL_DEV const float* src_pose(const KParams& kp, int s) {
  const float* p = kp.T[0];
  if (s == 1) p = kp.T[1];
  if (s == 2) p = kp.T[2];
  if (s == 3) p = kp.T[3];
  return p;
}

// Entry `e` of image b's record in KParams::prep: P = (K @ T)[:3,:] of every source, then inv_K[:3,:3].
VSL_DEV float prep_entry(const KParams& kp, int b, int e) {
  if (e < kPrepInvK) {
    const int s = e / 12, q = e % 12;
    if (s >= kp.S) return 0.0f;
    return proj_entry(kp.K + 16 * b, src_pose(kp, s) + 16 * b, q / 4, q % 4);
  }
  const int q = e - kPrepInvK;
  return q < 9 ? kp.invK[16 * b + (q / 3) * 4 + (q % 3)] : 0.0f;
}

struct alignas(8) float2v {
  float x, y;
};

// two horizontally adjacent texels fetched with one 8-byte load (the address is only 4-byte aligned)
struct __attribute__((packed, aligned(4))) TexelPair {
  float lo, hi;
};
VSL_DEV TexelPair load_texel_pair(const float* p) { return *reinterpret_cast<const TexelPair*>(p); }
// The same from a uniform plane base and a 32-bit byte offset (planes are smaller than 2^30 bytes: H*W < 2^28 is
// checked on the host), which the compiler can issue as scalar base + 32-bit vector offset without 64-bit
// address arithmetic per load; one offset register serves the three colour planes.
VSL_DEV float load_f32_at(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
VSL_DEV void store_f32_at(float* base, unsigned byte_off, float v) {
  *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
VSL_DEV TexelPair load_texel_pair_at(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const TexelPair*>(reinterpret_cast<const char*>(base) + byte_off);
}

// 3 rows x NC consecutive floats starting at float offset `base` of an LDS plane (any alignment)
template <int NC>
VSL_DEV void load_cols(const float* plane, int base, int pitch, float w[3][NC]) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int j = 0; j < NC; ++j) w[r][j] = plane[base + r * pitch + j];
}

#if defined(__HIPCC__)
// the value the lane to the left / right (within its group of 16 lanes, i.e. the neighbouring pixel pair of the same
// tile row) holds: one data-parallel-primitive move, usually folded into the instruction that consumes it
VSL_DEV float lane_left(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true));
}
VSL_DEV float lane_right(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xF, 0xF, true));
}
#endif

// y * W + x for y, W < 2^24 (checked on the host): the masks let the compiler pick the 24-bit multiply-add, a
// full-rate instruction, where the 32-bit multiply runs at a quarter of that.
VSL_DEV unsigned row_major(unsigned y, unsigned W, unsigned x) { return (y & 0xFFFFFFu) * (W & 0xFFFFFFu) + x; }

struct Taps {
  int i0, i1;
  float t;
  float live;  // 1 where d(sample)/d(coordinate) passes (0 < coordinate < n-1)
};

// grid_sample(align_corners=False, padding_mode="border") along one axis:
// unnormalise, clip, floor and fraction in float64 from the float32 grid value.
VSL_DEV Taps axis_taps(float g, int n) {
  // ((g + 1) n - 1) / 2 with the exact halving folded into the factor: same roundings, one operation less
  double c = ((double)g + 1.0) * ((double)n * 0.5) - 0.5;
  Taps tp;
  tp.live = (c > 0.0 && c < (double)(n - 1)) ? 1.0f : 0.0f;
  double cc = fmin(fmax(c, 0.0), (double)(n - 1));
  double f = floor(cc);
  tp.t = (float)(cc - f);
  int i0 = (int)f;
  if (i0 < 0) i0 = 0;
  if (i0 > n - 1) i0 = n - 1;
  tp.i0 = i0;
  tp.i1 = (i0 + 1 > n - 1) ? n - 1 : i0 + 1;
  return tp;
}

// The same for the tile program, which wants the first tap, whether a second one follows it and whether the
// derivative passes.  The clipped coordinate is never NaN (fmax(NaN, 0) is 0) and lies in [0, n-1], so its integer
// part needs no further clamping and truncation equals floor.
struct TileTaps {
  int i0;
  float t;
  bool step;  // the second tap is i0 + 1 (else i0 = n-1 and the second tap repeats it with weight 0)
  bool live;
};
VSL_DEV TileTaps tile_taps(float g, int n) {
  const double c = ((double)g + 1.0) * ((double)n * 0.5) - 0.5;
  const double top = (double)(n - 1);
  TileTaps tp;
  tp.live = c > 0.0 && c < top;
  const double cc = fmin(fmax(c, 0.0), top);
  tp.i0 = (int)cc;
#if defined(__HIP_DEVICE_COMPILE__)
  tp.t = (float)__builtin_amdgcn_fract(cc);  // cc - floor(cc) is exact, and so is the fraction instruction
#else
  tp.t = (float)(cc - floor(cc));
#endif
  tp.step = tp.i0 < n - 1;
  return tp;
}

// What phase 3 needs from phase 1 for one (pixel, source).  Where the registers allow it (Fused::CARRY: up to
// three sources) the six tap derivatives d warped_c / d ix, d iy themselves stay in registers (ThreadState::dp).
// The other programs have no room for them: three registers say where the four bilinear taps are and what their
// fractions are, and phase 3 fetches the taps again (cache hits, but 12 more gathers per thread and source through
// the vector L1, which is the busiest unit of the kernel: without them the two-source kernel is 10 % faster) and
// rebuilds the derivatives.
struct Chain {
  float tx, ty;  // bilinear fractions; tx is relative to the texel pair below (1 when x0 = W-1)
  int packed;    // bits 0..27 offset of texel (y0, min(x0, W-2)) in the image plane: the 8 bytes there hold both
                 // x taps; 29: y1 = y0+1, 30: d/dx passes (0 < ix < W-1), 31: d/dy passes
};

// Projection matrices P = (K T)[:3,:] of every source and the 3x3 part of inv_K: the same for every thread of a
// workgroup, so on the GPU they live in scalar registers (one lane computes each entry, v_readlane broadcasts).
template <int S>
struct Mats {
  float P[S][12];
  float ik[9];
};

template <int S>
struct ThreadState {
  Mats<S> m;
  Chain ch[2][S];            // programs that fetch the taps again in phase 3
  float dp[2][S][3][2];      // programs that carry them: d warped_c / d ix, d iy of (pixel, source, channel), before the clip masks
  unsigned live;             // with dp: bit 2 (k S + s) = d/dx passes (0 < ix < W-1), bit 2 (k S + s) + 1 = d/dy passes
  float photo;                  // sum of the selected photometric error over this thread's interior pixels
  float part[num_partials(S)];  // filled at the end of phase 3
};

// ----------------------------------------------------------------------------
// tile program
// ----------------------------------------------------------------------------
// ----------------------------------------------------------------------------
// shared window / SSIM helpers (a thread works on two horizontally adjacent pixels)
// ----------------------------------------------------------------------------
// 3 rows x 4 columns starting at float offset `base` (even) of an LDS plane
VSL_DEV void load_win(const float* plane, int base, int pitch, float w[3][4]) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float2v a = *reinterpret_cast<const float2v*>(plane + base + r * pitch);
    const float2v c = *reinterpret_cast<const float2v*>(plane + base + r * pitch + 2);
    w[r][0] = a.x;
    w[r][1] = a.y;
    w[r][2] = c.x;
    w[r][3] = c.y;
  }
}

// left window = columns 0..2, right window = columns 1..3 of per-column sums
VSL_DEV void two_windows(const float col[4], float out[2]) {
  const float mid = col[1] + col[2];
  out[0] = mid + col[0];
  out[1] = mid + col[3];
}

// Target-side window sums of both pixels, relative to the common centre yw[1][1].
// Moments are shift invariant, so any nearby value works as the centre; taking
// differences first keeps E[y^2] - mu^2 free of cancellation in float32.
VSL_DEV void y_stats(const float yw[3][4], float sy[2], float syy[2]) {
  const float cy = yw[1][1];
  float c1[4], c2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float e0 = yw[0][j] - cy, e1 = yw[1][j] - cy, e2 = yw[2][j] - cy;
    c1[j] = (e0 + e1) + e2;
    c2[j] = fmaf(e2, e2, fmaf(e1, e1, e0 * e0));
  }
  float s1[2], s2[2];
  two_windows(c1, s1);
  two_windows(c2, s2);
  // handed on as: centre-relative mean  myp = s1/9  and  variance + C2  (both shared by every source)
  constexpr float i9 = 1.0f / 9.0f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    sy[k] = s1[k] * i9;
    syy[k] = fmaf(s2[k], i9, -sy[k] * sy[k]) + kC2;
  }
}

struct SsimOut {
  float f;        // clamp((1 - SSIM)/2, 0, 1)
  float A, B, G;  // d f / d x_j = (A + B x_j + G y_j) / 9 for every j of the window
};

// layers.py:252-262 from centre-relative sums (cx, cy = the common centre values)
// (myp, vyc) = the target side from y_stats: centre-relative mean and variance + C2
VSL_DEV SsimOut ssim_formula(float cx, float cy, float sx, float sxx, float sxy, float myp, float vyc, bool want_coef) {
  constexpr float i9 = 1.0f / 9.0f;
  const float mxp = sx * i9;
  const float mux = cx + mxp, muy = cy + myp;
  // variance and covariance as (sum of products - sum * mean) / 9, the division folded into the next step:
  // d2 = var_x + (var_y + C2), n2 = 2 cov + C2
  const float d2 = fmaf(fmaf(-sx, mxp, sxx), i9, vyc);
  const float n2 = fmaf(fmaf(-sx, myp, sxy), 2.0f * i9, kC2);
  const float n1 = fmaf(2.0f * mux, muy, kC1);
  const float d1 = fmaf(mux, mux, fmaf(muy, muy, kC1));
  const float inv = fast_rcp(d1 * d2);
  const float val = fmaf(-0.5f * n1, n2 * inv, 0.5f);
  SsimOut o;
  o.f = fminf(fmaxf(val, 0.0f), 1.0f);
  o.A = o.B = o.G = 0.0f;
  if (want_coef) {
    // the clamp passes the gradient on [0,1] only, i.e. where it left the value alone: every coefficient carries
    // the factor 1 / (d1 d2), so zeroing that one factor zeroes them all
    const float ginv = (o.f == val) ? inv : 0.0f;
    const float t = n2 * ginv;  // n2 / (d1 d2)
    const float inv_d1 = inv * d2, inv_d2 = inv * d1;
    const float a = -t * (muy - n1 * mux * inv_d1);
    const float bq = n1 * t * inv_d2;
    const float g = -n1 * ginv;
    o.A = a - bq * mux - g * muy;
    o.B = bq;
    o.G = g;
  }
  return o;
}

// SSIM term of both pixels for one (image, channel)
VSL_DEV void ssim_pair(const float xw[3][4], const float yw[3][4], const float sy[2], const float syy[2],
                       bool want_coef, SsimOut out[2]) {
  const float cx = xw[1][1], cy = yw[1][1];
  float c1[4], c2[4], c3[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float e0 = xw[0][j] - cx, e1 = xw[1][j] - cx, e2 = xw[2][j] - cx;
    const float d0 = yw[0][j] - cy, d1 = yw[1][j] - cy, d2 = yw[2][j] - cy;
    c1[j] = (e0 + e1) + e2;
    c2[j] = fmaf(e2, e2, fmaf(e1, e1, e0 * e0));
    c3[j] = fmaf(e2, d2, fmaf(e1, d1, e0 * d0));
  }
  float sx[2], sxx[2], sxy[2];
  two_windows(c1, sx);
  two_windows(c2, sxx);
  two_windows(c3, sxy);
#pragma unroll
  for (int k = 0; k < 2; ++k) out[k] = ssim_formula(cx, cy, sx[k], sxx[k], sxy[k], sy[k], syy[k], want_coef);
}

// The same two steps with the target-side differences d = y - y_centre formed once and shared between the
// target statistics and every source frame (same values, same operation order per result).
VSL_DEV void y_diffs(const float yw[3][4], float d[3][4]) {
  const float cy = yw[1][1];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) d[r][j] = yw[r][j] - cy;
  d[1][1] = 0.0f;  // the centre itself: spelled out so that no instruction is spent on it or on its products
}
VSL_DEV void y_stats_d(const float d[3][4], float sy[2], float syy[2]) {
  float c1[4], c2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j == 1) {  // column of the centre: its middle entry is zero
      c1[j] = d[0][j] + d[2][j];
      c2[j] = fmaf(d[2][j], d[2][j], d[0][j] * d[0][j]);
    } else {
      c1[j] = (d[0][j] + d[1][j]) + d[2][j];
      c2[j] = fmaf(d[2][j], d[2][j], fmaf(d[1][j], d[1][j], d[0][j] * d[0][j]));
    }
  }
  float s1[2], s2[2];
  two_windows(c1, s1);
  two_windows(c2, s2);
  constexpr float i9 = 1.0f / 9.0f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    sy[k] = s1[k] * i9;
    syy[k] = fmaf(s2[k], i9, -sy[k] * sy[k]) + kC2;
  }
}
VSL_DEV void ssim_pair_d(const float xw[3][4], float cy, const float d[3][4], const float sy[2], const float syy[2],
                         bool want_coef, SsimOut out[2]) {
  const float cx = xw[1][1];
  float c1[4], c2[4], c3[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float e0 = xw[0][j] - cx, e2 = xw[2][j] - cx;
    if (j == 1) {  // column of the centre: its middle entry is zero on both sides
      c1[j] = e0 + e2;
      c2[j] = fmaf(e2, e2, e0 * e0);
      c3[j] = fmaf(e2, d[2][j], e0 * d[0][j]);
    } else {
      const float e1 = xw[1][j] - cx;
      c1[j] = (e0 + e1) + e2;
      c2[j] = fmaf(e2, e2, fmaf(e1, e1, e0 * e0));
      c3[j] = fmaf(e2, d[2][j], fmaf(e1, d[1][j], e0 * d[0][j]));
    }
  }
  float sx[2], sxx[2], sxy[2];
  two_windows(c1, sx);
  two_windows(c2, sxx);
  two_windows(c3, sxy);
#pragma unroll
  for (int k = 0; k < 2; ++k) out[k] = ssim_formula(cx, cy, sx[k], sxx[k], sxy[k], sy[k], syy[k], want_coef);
}

// ----------------------------------------------------------------------------
// identity tile program: min over sources of the un-warped photometric error
// (+ tie-break noise), Trainer.py:553-566,587-590.  Independent of disparity and
// pose, so one launch serves every disparity head of the same batch.
// ----------------------------------------------------------------------------
template <int TW, int TH, int NT, int S, bool AVG>
struct Ident {
  static constexpr int RW = TW + 2, RH = TH + 2, PN = RW * RH;
  static constexpr int ROWP = 16;
  static constexpr int LDS_FLOATS = (3 + 3 * S) * PN;
  static_assert(RW / 2 <= ROWP && ROWP * RH <= NT && PN % 2 == 0 && TW % 2 == 0, "tile / thread mismatch");

  struct Block {
    int b, x0, y0;
  };

  static VSL_DEV void load(const KParams& kp, const Block& bk, int tid, float* lds) {
    // one pixel pair of the tile grown by 1 per thread (row = tid / ROWP, pair = tid % ROWP), 32-bit offsets
    // from uniform plane bases
    const int ry = tid / ROWP, rx0 = 2 * (tid % ROWP);
    if (ry >= RH || rx0 >= RW) return;
    const unsigned HW = (unsigned)kp.H * (unsigned)kp.W;
    const float* tgtb = kp.tgt + (size_t)bk.b * 3 * HW;
    const int py = reflect_clamp(bk.y0 - 1 + ry, kp.H);
    // Tiles whose grown region lies inside the image horizontally (all but the first and last of a row) need no
    // reflection in x: the two pixels of a pair are neighbours in memory and come from one 8-byte load per plane.
    if (bk.x0 >= 1 && bk.x0 + TW + 1 <= kp.W) {
      const unsigned off = row_major((unsigned)py, (unsigned)kp.W, (unsigned)(bk.x0 - 1 + rx0)) << 2;
      float* row = lds + ry * RW + rx0;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const TexelPair t = load_texel_pair_at(tgtb + (size_t)c * HW, off);
        *reinterpret_cast<float2v*>(row + c * PN) = float2v{t.lo, t.hi};
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const TexelPair v = load_texel_pair_at(src_image(kp, s) + ((size_t)bk.b * 3 + c) * HW, off);
          *reinterpret_cast<float2v*>(row + (3 + s * 3 + c) * PN) = float2v{v.lo, v.hi};
        }
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int px = reflect_clamp(bk.x0 - 1 + rx0 + k, kp.W);
      const unsigned pix = row_major((unsigned)py, (unsigned)kp.W, (unsigned)px);
      const int p = ry * RW + rx0 + k;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        lds[c * PN + p] = (tgtb + (size_t)c * HW)[pix];
#pragma unroll
        for (int s = 0; s < S; ++s) lds[(3 + s * 3 + c) * PN + p] = (src_image(kp, s) + ((size_t)bk.b * 3 + c) * HW)[pix];
      }
    }
  }

  static VSL_DEV void compute(const KParams& kp, const Block& bk, int tid, const float* lds) {
    const int iy = tid / ROWP, ip = tid % ROWP;
    if (iy >= TH || ip >= TW / 2) return;
    const int gy = bk.y0 + iy, gx0 = bk.x0 + 2 * ip;
    if (gy >= kp.H || gx0 >= kp.W) return;
    const int base = iy * RW + 2 * ip;
    float err[S][2];  // photometric error, one channel's share after the other (see channel_error)
#pragma unroll
    for (int s = 0; s < S; ++s) err[s][0] = err[s][1] = 0.0f;
    const float w_l1 = kp.no_ssim ? 1.0f / 3.0f : kp.l1_w3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float yw[3][4], dy[3][4], sy[2], syy[2];
      load_win(lds + c * PN, base, RW, yw);
      if (!kp.no_ssim) {
        y_diffs(yw, dy);
        y_stats_d(dy, sy, syy);
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        float xw[3][4];
        load_win(lds + (3 + s * 3 + c) * PN, base, RW, xw);
        err[s][0] = fmaf(w_l1, fabsf(yw[1][1] - xw[1][1]), err[s][0]);
        err[s][1] = fmaf(w_l1, fabsf(yw[1][2] - xw[1][2]), err[s][1]);
        if (!kp.no_ssim) {
          SsimOut o[2];
          ssim_pair_d(xw, yw[1][1], dy, sy, syy, false, o);
          err[s][0] = fmaf(kp.ssim_w3, o[0].f, err[s][0]);
          err[s][1] = fmaf(kp.ssim_w3, o[1].f, err[s][1]);
        }
      }
    }
    const size_t HW = (size_t)kp.H * kp.W;
    constexpr int NI = AVG ? 1 : S;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (gx0 + k >= kp.W) continue;
      const size_t pix = (size_t)gy * kp.W + gx0 + k;
      float best = INFINITY, idsum = 0.0f;
      float zs[S];  // (!AVG) tie-break noise of every source plane at this pixel
      if (!AVG) {
        if (kp.noise) {
#pragma unroll
          for (int s = 0; s < S; ++s) zs[s] = kp.noise[(size_t)(bk.b * NI + s) * HW + pix];
        } else {
          // planes 2j and 2j+1 share one transform, keyed by the even plane
#pragma unroll
          for (int s = 0; s < S; s += 2) {
            const Normal2 z2 = hash_normal2(noise_key(kp.seed, (unsigned)(bk.b * NI + s)), (unsigned)pix);
            zs[s] = z2.a;
            if (s + 1 < S) zs[s + 1] = z2.b;
          }
        }
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const float e = err[s][k];
        if (AVG) {
          idsum += e;
        } else {
          const float v = fmaf(zs[s], 0.00001f, e);
          if (v < best) best = v;
        }
      }
      if (AVG) {
        const float z = kp.noise ? kp.noise[(size_t)bk.b * HW + pix] : hash_normal(noise_key(kp.seed, (unsigned)bk.b), (unsigned)pix);
        best = fmaf(z, 0.00001f, idsum * (1.0f / S));
      }
      kp.idmin[(size_t)bk.b * HW + pix] = best;
    }
  }
};

// ----------------------------------------------------------------------------
// fused tile program
// ----------------------------------------------------------------------------
// HALO = 2: the full program (loss and everything for the backward).  HALO = 1: forward only -- phases 0 and 1
// on the tile grown by 1, then phase_forward instead of phases 2 and 3 (no coefficients, no gradients).
// OUT: the instantiation that also writes the optional outputs of phase 1 (vsl_head.depth / sample / color); the
// plain one carries no trace of them (their tests alone cost the two-source kernel 22 spilled scalar registers and 11 %).
template <int TW, int TH, int NT, int S, bool AVG, int HALO = 2, bool OUT = false>
struct Fused {
  static_assert(TW % 2 == 0, "tile width must be even");
  static_assert(HALO == 1 || HALO == 2, "halo is 1 (forward only) or 2");
  static constexpr int R2W = TW + 2 * HALO, R2H = TH + 2 * HALO, P2 = R2W * R2H;
  static constexpr int R1W = TW + 2, R1H = TH + 2, P1 = R1W * R1H;
  // Threads map to pixel pairs as (row = tid / ROWP, pair = tid % ROWP) in every phase.  ROWP = 16
  // makes each 16-lane LDS access group lie inside one tile row, which keeps the 8-byte window
  // reads and coefficient writes free of bank conflicts (a row is at most 32 dwords).
  static constexpr int ROWP = 16;
  static constexpr int NSET = AVG ? S : 1;
  static constexpr int NPART = num_partials(S);
  // The full program keeps the tap derivatives of phase 1 in registers for phase 3 where the register budget of its
  // workgroups per CU allows (two sources: 128 of the 128 registers of four waves per SIMD; three sources: 147 of the
  // 168 of three; one source with the mean over sources sits at the 96 of five workgroups per CU already); the
  // others fetch the taps again.
  static constexpr bool CARRY = HALO == 2 && (S == 2 || ((S == 1 || S == 3) && !AVG));
  static_assert(R2W / 2 <= ROWP && ROWP * R2H <= NT, "one R2 pair per thread, at most ROWP pairs per row");
  static_assert(P2 % 2 == 0 && P1 % 2 == 0, "planes must keep 8-byte alignment");

  // LDS layout in floats
  static constexpr int OFF_TGT = 0;  // 3 planes over R2
  static constexpr int OFF_WRP = OFF_TGT + 3 * P2;      // S*3 planes over R2
  static constexpr int OFF_DSP = OFF_WRP + S * 3 * P2;  // 1 plane over R2
  static constexpr int OFF_COEF = OFF_DSP + P2;         // NSET*9 coefficient planes + selection plane over R1
  static constexpr int COEF_FLOATS = HALO == 2 ? (NSET * 9 + 1) * P1 : 0;
  // row pitch = 8 mod 64: the (row, piece) items of a wave in phase4_sum_a then read 64 different banks
  static constexpr int RED_PITCH = ((NT + 63) / 64) * 64 + 8;
  static constexpr int NRED = HALO == 2 ? NPART : 4;  // partial sums that can be non-zero
  static constexpr int RED_SPLIT = 8;
  // Full program: a sum only has entries where they can be non-zero.  The photometric sum and the 12 gradient sums
  // per source are held by the threads of the tile rows (BIG = ROWP x TH entries; the photometric one by thread rows
  // 1..TH, which took the interior rows of R1 in phase 2), the three smoothness sums by the threads beyond them
  // (SMALL entries).  Pitches = 8 mod 64 floats, so that the items of a wave in phase4_sum_a read 64 different banks.
  // With three sources this is what lets the planes, not the reduction, set the LDS footprint (four workgroups per CU).
  static constexpr int NBIG = 1 + 12 * S;
  static constexpr int BIG = ROWP * TH, SMALL = NT - ROWP * TH;
  static constexpr int BIG_PITCH = ((BIG + 63) / 64) * 64 + 8, SMALL_PITCH = ((SMALL + 63) / 64) * 64 + 8;
  static constexpr int RED_FLOATS = HALO == 2 ? NBIG * BIG_PITCH + 3 * SMALL_PITCH : NRED * RED_PITCH;
  static constexpr int LDS_MAIN = OFF_COEF + COEF_FLOATS;
  static constexpr int RED_TOTAL = RED_FLOATS + (HALO == 2 ? NBIG + 3 : NRED) * RED_SPLIT;
  static constexpr int LDS_FLOATS = LDS_MAIN > RED_TOTAL ? LDS_MAIN : RED_TOTAL;
  static_assert(HALO == 1 || (SMALL > 0 && SMALL % ROWP == 0 && BIG % ROWP == 0), "the full program needs threads beside the tile rows");

  struct Block {
    int b, x0, y0;
  };
  // The launcher sends every head that wants gradients through the full program and every other head through the
  // forward-only one, so inside the full program the answer is known: phase 3 and the smoothness term take it as a
  // constant (32 vector and 73 scalar instructions fewer per wave than with a run-time test).  Where the constant
  // would cost registers the program does not have, the uniform branch on HeadView::want_grad stays: the one-source
  // mean-over-sources program as a whole (it sits at the 96 registers of five waves per SIMD), and phase 2 (the
  // coefficients) of every program but the plain one- and two-source ones (the others spill with it).
  static VSL_DEV bool wants_grad(const HeadView& hv) {
    if constexpr (HALO == 2 && !(S == 1 && AVG))
      return true;
    else
      return hv.want_grad != 0;
  }
  // the same question in phase 2 (coefficients or not): a constant only where the registers allow it
  static VSL_DEV bool wants_coef(const HeadView& hv) {
    if constexpr (HALO == 2 && !AVG && !OUT && S <= 2)
      return true;
    else
      return hv.want_grad != 0;
  }

  // Scheduling fence between the unrolled per-channel / per-source bodies: it keeps the scheduler from interleaving
  // them, which multiplies the live registers.  The three-source program runs three workgroups per CU and has the
  // registers (147 of 168 with or without the fences): there the scheduler is left alone, which is 2.3 % faster.
  static VSL_DEV void sched_fence() {
    if constexpr (!(HALO == 2 && S == 3 && !AVG)) VSL_SCHED_FENCE();
  }

  // ---- phase 0: projection matrices (layers.py:189) and inv_K into wave-uniform registers -----------
  static VSL_DEV void phase0(const KParams& kp, const HeadView& hv, const Block& bk, int tid, ThreadState<S>& st) {
#if defined(__HIPCC__)
    // Prepared by an earlier launch on the same stream (the launcher always sets KParams::prep). The image index
    // is uniform, so these are scalar loads straight into wave-uniform registers; they come before any store of
    // this kernel.
    const float* pp = kp.prep + (size_t)bk.b * kPrepStride;
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int e = 0; e < 12; ++e) st.m.P[s][e] = pp[s * 12 + e];
#pragma unroll
    for (int e = 0; e < 9; ++e) st.m.ik[e] = pp[kPrepInvK + e];
    if (tid == 0 && bk.b == 0 && bk.x0 == 0 && bk.y0 == 0 && hv.counter) *hv.counter = 0u;
#else
    if (tid == 0 && bk.b == 0 && bk.x0 == 0 && bk.y0 == 0 && hv.counter) *hv.counter = 0u;
    for (int s = 0; s < S; ++s)
      for (int e = 0; e < 12; ++e) st.m.P[s][e] = proj_entry(kp.K + 16 * bk.b, src_pose(kp, s) + 16 * bk.b, e / 4, e % 4);
    for (int e = 0; e < 9; ++e) st.m.ik[e] = kp.invK[16 * bk.b + (e / 3) * 4 + (e % 3)];
#endif
  }

  // Projection matrix of source s where a phase uses it.  Up to two sources: the entries phase 0 put into scalar registers.
  // Three or four sources: 36 or 48 entries held for the whole kernel do not fit the scalar registers beside the pointers
  // and constants (the three-source program spilled 30 of them into vector-register lanes and fetched them back 78 times),
  // so every use loads its source's 12 entries from the prepared record again: scalar loads just ahead of each source in
  // phase 1, vector loads of one address in phase 3 (the kernel has stored to global memory by then, which rules scalar
  // loads out); no spilled scalar register is left and the program is 1.5 % faster.
  static VSL_DEV void source_matrix(const KParams& kp, const Block& bk, const ThreadState<S>& st, int s, float P[12]) {
#if defined(__HIPCC__)
    if constexpr (S >= 3) {
      const float* pp = kp.prep + (size_t)bk.b * kPrepStride + s * 12;
#pragma unroll
      for (int e = 0; e < 12; ++e) P[e] = pp[e];
      return;
    }
#endif
#pragma unroll
    for (int e = 0; e < 12; ++e) P[e] = st.m.P[s][e];
  }

  // ---- optional outputs of phase 1 (vsl_head.depth / sample / color): OUT instantiations only ----
  // R2 position (ry, rx) is a pixel of the tile, not a halo position or one past the image's edge
  static VSL_DEV bool is_tile_pixel(const KParams& kp, const Block& bk, int ry, int rx) {
    return ry >= HALO && ry < HALO + TH && rx >= HALO && rx < HALO + TW && bk.y0 - HALO + ry < kp.H && bk.x0 - HALO + rx < kp.W;
  }
  static VSL_DEV void emit_depth(const KParams& kp, const HeadView& hv, const Block& bk, int ry, int rx, unsigned pix, float D) {
    float* out = kp.out_table[hv.head * kOutPtrs];
    if (out && is_tile_pixel(kp, bk, ry, rx)) store_f32_at(out + (size_t)bk.b * kp.H * kp.W, pix << 2, D);
  }
  static VSL_DEV void emit_sample(const KParams& kp, const HeadView& hv, const Block& bk, int ry, int rx, unsigned pix, int s,
                                  float gx, float gy) {
    float* out = kp.out_table[hv.head * kOutPtrs + 1 + s];
    if (out && is_tile_pixel(kp, bk, ry, rx))
      *reinterpret_cast<float2v*>(out + 2 * ((size_t)bk.b * kp.H * kp.W + pix)) = float2v{gx, gy};
  }
  static VSL_DEV void emit_color(const KParams& kp, const HeadView& hv, const Block& bk, int ry, int rx, unsigned pix, int s,
                                 int c, float v) {
    float* out = kp.out_table[hv.head * kOutPtrs + 1 + kMaxSrc + s];
    if (out && is_tile_pixel(kp, bk, ry, rx)) store_f32_at(out + ((size_t)bk.b * 3 + c) * kp.H * kp.W, pix << 2, v);
  }

  // ---- phase 1: two horizontally adjacent R2 positions per thread -------------------
  // Every thread warps exactly one pair of R2 and keeps where its taps were; the threads whose
  // pair is interior use that again in phase 3.
  static VSL_DEV void phase1(const KParams& kp, const HeadView& hv, const Block& bk, int tid, float* lds, ThreadState<S>& st) {
    st.photo = 0.0f;
    // Which two positions of its row a thread warps.  The full program takes the horizontally adjacent pair
    // (2 lane, 2 lane + 1) that the same thread differentiates in phase 3.  The forward-only program keeps nothing
    // per thread, so it takes (lane, lane + ROWP) instead: the four lanes of a quad then gather for
    // four adjacent pixels rather than every other one, and their taps fall into fewer cache lines.
#if defined(VSL_PROBE_STRIDE1)
    constexpr bool kStride1 = true;  // timing probe only: phase 3 then pairs the carried derivatives with the wrong pixels
#else
    constexpr bool kStride1 = HALO == 1;
#endif
    const int trow = tid / ROWP, rx0 = kStride1 ? tid % ROWP : 2 * (tid % ROWP);
    constexpr int kStep = kStride1 ? ROWP : 1;
#if defined(VSL_PROBE_NOVHALO)
    // timing probe only (results wrong): the rows of the vertical halo are not warped, as if a neighbouring step of a
    // column march had left them in the planes
    const bool probe_skip = HALO == 2 && trow >= TH;
#else
    constexpr bool probe_skip = false;
#endif
    if (trow >= R2H || rx0 >= R2W || probe_skip) {
      // No position for this thread (regions narrower than 2 * ROWP columns).  Phase 3 keeps every lane of a tile row
      // alive for its lane exchange and lets it fetch the taps its record names: the record must name a valid texel.
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int s = 0; s < S; ++s) {
          if constexpr (CARRY) {
#pragma unroll
            for (int c = 0; c < 3; ++c) st.dp[k][s][c][0] = st.dp[k][s][c][1] = 0.0f;
          } else {
            st.ch[k][s] = Chain{0.0f, 0.0f, 0};
          }
        }
      if constexpr (CARRY) st.live = 0u;
      return;
    }
    if constexpr (CARRY) st.live = 0u;
    // Full program: thread row t warps R2 row t + 2 (wrapping round), so that thread rows 0..TH-1 -- whole waves --
    // hold the tap records of the interior rows, which the same threads differentiate in phase 3, and the halo
    // rows fall to the last wave, which phase 3 gives the smoothness term instead.
    const int ry = HALO == 2 ? (trow + 2 < R2H ? trow + 2 : trow + 2 - R2H) : trow;
    const int H = kp.H, W = kp.W;
    const unsigned HW = (unsigned)H * (unsigned)W;
    // per-image bases are uniform over the workgroup; per-lane offsets stay unsigned 32-bit
    const float* dispb = hv.disp + (size_t)bk.b * HW;
    const float* tgtb = kp.tgt + (size_t)bk.b * 3 * HW;
    const int py = reflect_clamp(bk.y0 - HALO + ry, H);
    const float* ik = st.m.ik;
    const Divisor by_wm1 = divisor((float)(W - 1)), by_hm1 = divisor((float)(H - 1));
    const float fy = (float)py;
    // rays share the y part: inv_K[i][1]*y is added after inv_K[i][0]*x in the reference's order
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      // (a second position past the row's end repeats the last one: same values into the same cells)
      const int rx = kStride1 ? (rx0 + k * kStep < R2W ? rx0 + k * kStep : R2W - 1) : rx0 + k;
      const int px = reflect_clamp(bk.x0 - HALO + rx, W);
      const int p2 = ry * R2W + rx;
      const unsigned pix = row_major((unsigned)py, (unsigned)W, (unsigned)px);
      const float d = load_f32_at(dispb, pix << 2);
      lds[OFF_DSP + p2] = d;
#pragma unroll
      for (int c = 0; c < 3; ++c) lds[OFF_TGT + c * P2 + p2] = load_f32_at(tgtb + (size_t)c * HW, pix << 2);

      // disp_to_depth and the back-projected ray, layers.py:21-25,163-169
      const float sd = VSL_FADD(kp.a_disp, VSL_FMUL(kp.c_disp, d));
      const float D = quot(1.0f, divisor(sd));
      if constexpr (OUT) emit_depth(kp, hv, bk, ry, rx, pix, D);
      const float fx = (float)px;
      float X[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        float r = VSL_FMUL(ik[i * 3 + 0], fx);
        r = fmaf(ik[i * 3 + 1], fy, r);
        r = fmaf(ik[i * 3 + 2], 1.0f, r);
        X[i] = VSL_FMUL(D, r);
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        float P[12];
        source_matrix(kp, bk, st, s, P);
        float cam[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          float a = VSL_FMUL(P[i * 4 + 0], X[0]);
          a = fmaf(P[i * 4 + 1], X[1], a);
          a = fmaf(P[i * 4 + 2], X[2], a);
          a = fmaf(P[i * 4 + 3], 1.0f, a);
          cam[i] = a;
        }
        // layers.py:196-201
        const float den = VSL_FADD(cam[2], kp.eps);
        const Divisor by_den = divisor(den);
        const float u = quot(cam[0], by_den), v = quot(cam[1], by_den);
        const float gx = VSL_FMUL(VSL_FSUB(quot(u, by_wm1), 0.5f), 2.0f);
        const float gy = VSL_FMUL(VSL_FSUB(quot(v, by_hm1), 0.5f), 2.0f);
        if constexpr (OUT) emit_sample(kp, hv, bk, ry, rx, pix, s, gx, gy);
        const TileTaps tx = tile_taps(gx, W), ty = tile_taps(gy, H);
        const float* sp = src_image(kp, s) + (size_t)bk.b * 3 * HW;
        // both x taps of a row come from one 8-byte load at min(x0, W-2)
        const bool xs = tx.step, ys = ty.step;
        const unsigned o0 = row_major((unsigned)ty.i0, (unsigned)W, (unsigned)(tx.i0 < W - 2 ? tx.i0 : W - 2));
        const unsigned o1 = o0 + (ys ? (unsigned)W : 0u);
        // At the last column (x0 = W-1, fraction 0) the pair holds texels W-2 and W-1: fraction 1 on that pair is
        // the same sample, weight for weight, and needs no selection between the two halves.
        const float txe = xs ? tx.t : 1.0f;
        const float w00 = (1.0f - txe) * (1.0f - ty.t), w01 = txe * (1.0f - ty.t);
        const float w10 = (1.0f - txe) * ty.t, w11 = txe * ty.t;
#if defined(VSL_PROBE_RGBX)
        // timing probe only (results wrong): the four taps as four 16-byte loads from a texel-interleaved [H,W,4] copy of
        // the source instead of six 8-byte loads from three planes.  There is no such copy here: the loads go to the
        // same image's planes at the interleaved layout's offsets, wrapped into the image's 3 H W floats.
        struct alignas(16) Texel4 { float v[4]; };
        const unsigned wrap = 3u * HW - 8u;
        unsigned q0 = o0 << 2;
        q0 = q0 >= wrap ? q0 - wrap : q0;
        unsigned q1 = q0 + (ys ? 4u * (unsigned)W : 0u);
        q1 = q1 >= wrap ? q1 - wrap : q1;
        const Texel4 t00 = *reinterpret_cast<const Texel4*>(sp + q0), t01 = *reinterpret_cast<const Texel4*>(sp + q0 + 4);
        const Texel4 t10 = *reinterpret_cast<const Texel4*>(sp + q1), t11 = *reinterpret_cast<const Texel4*>(sp + q1 + 4);
#endif
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#if defined(VSL_PROBE_RGBX)
          const TexelPair a{t00.v[c], t01.v[c]}, b{t10.v[c], t11.v[c]};
#else
          const float* pc = sp + (size_t)c * HW;
          const TexelPair a = load_texel_pair_at(pc, o0 << 2), b = load_texel_pair_at(pc, o1 << 2);
#endif
          const float wv = a.lo * w00 + a.hi * w01 + b.lo * w10 + b.hi * w11;
          lds[OFF_WRP + (s * 3 + c) * P2 + p2] = wv;
          if constexpr (OUT) emit_color(kp, hv, bk, ry, rx, pix, s, c, wv);
          if constexpr (CARRY) {
            // by every thread, also those of the halo rows that never use them: a branch here would end the basic
            // block, and the compiler then waits for each pair of gathers before it issues the next (measured +5 %)
            // d/dix = (1-ty)(v01-v00) + ty(v11-v10), d/diy = (1-tx)(v10-v00) + tx(v11-v01): both are the difference
            // along their own axis at the first tap plus the fraction along the other axis times the mixed difference
            const float v00 = a.lo, v01 = a.hi, v10 = b.lo, v11 = b.hi;
            const float dx0 = v01 - v00, dy0 = v10 - v00, dxy = (v11 - v10) - dx0;
            st.dp[k][s][c][0] = fmaf(ty.t, dxy, dx0);
            st.dp[k][s][c][1] = fmaf(txe, dxy, dy0);
          }
        }
        if constexpr (CARRY) {
          st.live |= ((unsigned)tx.live | ((unsigned)ty.live << 1)) << (2 * (k * S + s));
        } else {
          Chain& ch = st.ch[k][s];
          ch.tx = txe;
          ch.ty = ty.t;
          ch.packed = (int)(o0 | ((unsigned)ys << 29) | ((unsigned)tx.live << 30) |
                            ((unsigned)ty.live << 31));
        }
      }
    }
  }

  // ---- phase 2: photometric error on R1, running argmin, coefficients -----------
  static VSL_DEV void phase2(const KParams& kp, const HeadView& hv, const Block& bk, int tid, float* lds, ThreadState<S>& st) {
    // no_ssim is uniform over the launch: two specialised bodies instead of a test inside every channel loop
    if (kp.no_ssim)
      phase2_t<true>(kp, hv, bk, tid, lds, st);
    else
      phase2_t<false>(kp, hv, bk, tid, lds, st);
  }
  template <bool NOSSIM>
  static VSL_DEV void phase2_t(const KParams& kp, const HeadView& hv, const Block& bk, int tid, float* lds, ThreadState<S>& st) {
#if defined(VSL_PROBE_NOVHALO)
    if (tid / ROWP >= TH) return;  // timing probe only: the two halo rows of R1 are not evaluated
    const int qy = tid / ROWP + 1, qx0 = 2 * (tid % ROWP);
#else
    const int qy = tid / ROWP, qx0 = 2 * (tid % ROWP);
#endif
    if (qy >= R1H || qx0 >= R1W) return;
    const int gy = bk.y0 - 1 + qy, gx0 = bk.x0 - 1 + qx0;
    const bool rowv = gy >= 0 && gy < kp.H;
    const bool rowi = qy >= 1 && qy <= TH;
    bool valid[2], interior[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      valid[k] = rowv && gx0 + k >= 0 && gx0 + k < kp.W;
      interior[k] = valid[k] && rowi && (qx0 + k) >= 1 && (qx0 + k) <= TW;
    }
    const int base = qy * R2W + qx0;
    const int q1 = qy * R1W + qx0;
    float* coef = lds + OFF_COEF;               // [NSET][9][P1]
    float* sel = lds + OFF_COEF + NSET * 9 * P1;  // [P1] winning source (-1 = none); AVG: 1 = warped wins

    const size_t HW = (size_t)kp.H * kp.W;
    float best[2] = {INFINITY, INFINITY};
    if (kp.automask) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (valid[k]) best[k] = load_f32_at(kp.idmin + (size_t)bk.b * HW, row_major((unsigned)gy, (unsigned)kp.W, (unsigned)(gx0 + k)) << 2);
    }
    float win[2] = {-1.0f, -1.0f};
    float rsum[2] = {0.0f, 0.0f};
    // (!AVG) the winner's nine coefficients of both pixels stay in registers and go to LDS once, as pairs.  Where no
    // source wins (identity term smaller, or a position outside the image) the planes hold the first source's
    // coefficients: finite values that phase 3 multiplies by a zero mask, since the selection plane says -1 there --
    // so the first source needs no selection and nothing has to be cleared.
    float wc[2][9];
    if constexpr (S <= 3) {
      // Channel by channel: the target window, its differences to the centre and its statistics are formed once and
      // serve every source frame; the per-source sums over the channels keep their order (c ascending).
      // The error of (source, pixel) is accumulated channel by channel, one register each:
      // 0.85/3 * SSIM term + 0.15/3 * |difference| per channel (Trainer.py:486-493; |difference| / 3 without SSIM).
      float err[S][2];
      constexpr float w_l1_only = 1.0f / 3.0f;
      // (!AVG) the first source's coefficients go to the LDS planes as they are formed -- they are what the planes
      // hold unless a later source wins -- so only the later sources' sets wait in registers for the decision
      constexpr int S0 = AVG ? 0 : 1;
      float cf[S - S0 > 0 ? S - S0 : 1][2][9];
#pragma unroll
      for (int s = 0; s < S; ++s) err[s][0] = err[s][1] = 0.0f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float yw[3][4], dy[3][4], sy[2], syy[2];
        load_win(lds + OFF_TGT + c * P2, base, R2W, yw);
        if (!NOSSIM) {
          y_diffs(yw, dy);
          y_stats_d(dy, sy, syy);
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
          float xw[3][4];
          load_win(lds + OFF_WRP + (s * 3 + c) * P2, base, R2W, xw);
          err[s][0] = fmaf(NOSSIM ? w_l1_only : kp.l1_w3, fabsf(yw[1][1] - xw[1][1]), err[s][0]);
          err[s][1] = fmaf(NOSSIM ? w_l1_only : kp.l1_w3, fabsf(yw[1][2] - xw[1][2]), err[s][1]);
          SsimOut o[2];
          if (!NOSSIM) {
            ssim_pair_d(xw, yw[1][1], dy, sy, syy, wants_coef(hv), o);
            err[s][0] = fmaf(kp.ssim_w3, o[0].f, err[s][0]);
            err[s][1] = fmaf(kp.ssim_w3, o[1].f, err[s][1]);
          } else {
            o[0].A = o[0].B = o[0].G = o[1].A = o[1].B = o[1].G = 0.0f;
          }
          if (s < S0) {
            *reinterpret_cast<float2v*>(coef + (c * 3 + 0) * P1 + q1) = float2v{o[0].A, o[1].A};
            *reinterpret_cast<float2v*>(coef + (c * 3 + 1) * P1 + q1) = float2v{o[0].B, o[1].B};
            *reinterpret_cast<float2v*>(coef + (c * 3 + 2) * P1 + q1) = float2v{o[0].G, o[1].G};
          } else {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              cf[s - S0][k][c * 3 + 0] = o[k].A;
              cf[s - S0][k][c * 3 + 1] = o[k].B;
              cf[s - S0][k][c * 3 + 2] = o[k].G;
            }
          }
        }
        sched_fence();
      }
      bool later[2] = {false, false};  // a source after the first wins the pixel
#pragma unroll
      for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          float e = err[s][k];
          e *= kp.warp_scale;
          if (AVG) {
            rsum[k] += e;
#pragma unroll
            for (int j = 0; j < 9; ++j) coef[(s * 9 + j) * P1 + q1 + k] = valid[k] ? cf[s][k][j] : 0.0f;
          } else {
            const bool take = valid[k] && e < best[k];
            best[k] = take ? e : best[k];
            win[k] = take ? (float)s : win[k];
            if (s >= 1) {
              later[k] = later[k] || take;
#pragma unroll
              for (int j = 0; j < 9; ++j) wc[k][j] = (s == 1 || take) ? cf[s - S0][k][j] : wc[k][j];
            }
          }
        }
      }
      if (!AVG && S > 1) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (later[k]) {
#pragma unroll
            for (int j = 0; j < 9; ++j) coef[j * P1 + q1 + k] = wc[k][j];
          }
      }
    } else {
      // four sources: the per-source coefficient sets of the channel-outer order no longer fit the register budget
      // of three workgroups per CU, so the sources are walked one after the other
      // target-side window sums once per channel
      float sy[3][2], syy[3][2];
      if (!NOSSIM) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float yw[3][4];
          load_win(lds + OFF_TGT + c * P2, base, R2W, yw);
          y_stats(yw, sy[c], syy[c]);
          VSL_SCHED_FENCE();
        }
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        float err[2] = {0.0f, 0.0f};
        float cf[2][9];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float yw[3][4], xw[3][4];
          load_win(lds + OFF_TGT + c * P2, base, R2W, yw);
          load_win(lds + OFF_WRP + (s * 3 + c) * P2, base, R2W, xw);
          err[0] = fmaf(NOSSIM ? 1.0f / 3.0f : kp.l1_w3, fabsf(yw[1][1] - xw[1][1]), err[0]);
          err[1] = fmaf(NOSSIM ? 1.0f / 3.0f : kp.l1_w3, fabsf(yw[1][2] - xw[1][2]), err[1]);
          if (!NOSSIM) {
            SsimOut o[2];
            ssim_pair(xw, yw, sy[c], syy[c], hv.want_grad != 0, o);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              err[k] = fmaf(kp.ssim_w3, o[k].f, err[k]);
              cf[k][c * 3 + 0] = o[k].A;
              cf[k][c * 3 + 1] = o[k].B;
              cf[k][c * 3 + 2] = o[k].G;
            }
          } else {
#pragma unroll
            for (int k = 0; k < 2; ++k) cf[k][c * 3 + 0] = cf[k][c * 3 + 1] = cf[k][c * 3 + 2] = 0.0f;
          }
          VSL_SCHED_FENCE();
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          float e = err[k];
          e *= kp.warp_scale;
          if (AVG) {
            rsum[k] += e;
#pragma unroll
            for (int j = 0; j < 9; ++j) coef[(s * 9 + j) * P1 + q1 + k] = valid[k] ? cf[k][j] : 0.0f;
          } else {
            const bool take = valid[k] && e < best[k];
            best[k] = take ? e : best[k];
            win[k] = take ? (float)s : win[k];
#pragma unroll
            for (int j = 0; j < 9; ++j) wc[k][j] = (s == 0 || take) ? cf[k][j] : wc[k][j];
          }
        }
      }
      if (!AVG) {
#pragma unroll
        for (int j = 0; j < 9; ++j) *reinterpret_cast<float2v*>(coef + j * P1 + q1) = float2v{wc[0][j], wc[1][j]};
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (AVG) {
        const float e = rsum[k] * (1.0f / S);
        if (valid[k] && e < best[k]) {
          best[k] = e;
          win[k] = 1.0f;
        }
      }
      sel[q1 + k] = win[k];
      if (interior[k]) {
        st.photo += best[k];
        if (kp.automask && hv.ident_sel)
          store_f32_at(hv.ident_sel + (size_t)bk.b * HW, row_major((unsigned)gy, (unsigned)kp.W, (unsigned)(gx0 + k)) << 2,
                       (win[k] >= 0.0f) ? 1.0f : 0.0f);
      }
    }
  }

  // ---- phase 3: gradients of the interior pair ---------------------------------
  static VSL_DEV void phase3(const KParams& kp, const HeadView& hv, const Block& bk, int tid, const float* lds, ThreadState<S>& st) {
#pragma unroll
    for (int k = 0; k < NPART; ++k) st.part[k] = 0.0f;
    st.part[0] = st.photo;
    const int H = kp.H, W = kp.W;
    // thread row t warped R2 row t + 2 in phase 1, the interior row t; rows from TH on -- the whole last wave --
    // held halo rows only
    const int row = tid / ROWP, ip = tid % ROWP - 1;
    if (row >= TH) {
      phase3_smooth(kp, hv, bk, tid, lds, st);
      return;
    }
    const int iy = row;
    // one combined test (no short-circuit): nested early exits make the compiler re-zero all partial sums at every level
    const bool off_tile = (unsigned)ip >= (unsigned)(TW / 2);  // covers ip < 0 as well
    const bool off_image = bk.y0 + iy >= kp.H;
    const bool off_row = bk.x0 + 2 * ip >= kp.W;
    // On the device the threads of a tile row form the column sums of the coefficient gather together (each its own
    // two columns, see below), so the pairs outside the tile or the image stay until that is done.
    const bool inactive = off_tile | off_image | off_row;
#if !defined(__HIPCC__)
    if (inactive) return;
#endif
    Chain chn[2][S];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int s = 0; s < S; ++s) {
        chn[k][s] = st.ch[k][s];
      }
    const int jy = bk.y0 + iy;
    const int jx0 = bk.x0 + 2 * ip;
    // pairs kept alive for the lane exchange below take part in the rest as pixels that do not exist: no
    // contribution to any sum, no store (an early exit for them after the exchange made the compiler zero all partial
    // sums once more on that path)
    const bool ok[2] = {!inactive, !inactive && jx0 + 1 < W};
    const float* coef = lds + OFF_COEF;
    const float* sel = lds + OFF_COEF + NSET * 9 * P1;
    const int qb = iy * R1W + 2 * ip;  // R1 offset of the window's top-left (even)
    const int p2 = (2 + iy) * R2W + 2 + 2 * ip;
    const size_t HW = (size_t)H * W;

    float g_disp[2] = {0.0f, 0.0f};
    if (wants_grad(hv)) {
      // Reflection-padding fold (backward of layers.py:240,249-250): a window centred on the image
      // border counts twice for the pixel next to the border.  Only tiles that contain column 1 /
      // W-2 or row 1 / H-2 need the weights; the test is uniform over the workgroup.
      const bool border = bk.x0 == 0 || bk.y0 == 0 || (bk.x0 <= W - 2 && W - 2 < bk.x0 + TW) ||
                          (bk.y0 <= H - 2 && H - 2 < bk.y0 + TH);
      float fy[3] = {1.0f, 1.0f, 1.0f}, fxm[2][3] = {{1.0f, 1.0f, 1.0f}, {1.0f, 1.0f, 1.0f}};
      if (border) {
        fy[0] += (float)(jy == 1);
        fy[2] += (float)(jy == H - 2);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          fxm[k][0] += (float)(jx0 + k == 1);
          fxm[k][2] += (float)(jx0 + k == W - 2);
        }
      }
      const float* ikm = st.m.ik;

      float gD[2] = {0.0f, 0.0f};
      float gcs[S][2][3];  // d L / d cam of (source, pixel); the 3x4 projection gradient is built from it at the end

      // ---- d L / d warped_c of both pixels for every source: 3x3 gather of the winners' coefficients ----
      // Every coefficient plane is read once; its window is split between the sources by 0/1 masks of the
      // selection plane (x the row fold on border tiles).  Column sums first, then the two overlapping
      // windows of the pixel pair.
      // A pixel pair needs the sums of four window columns (its own two and one on either side).  The host build
      // forms all four; on the device a thread forms the two columns of its own pixels and takes the outer two from
      // the neighbouring lanes, which hold exactly those sums: half the multiply-adds, masks and LDS reads.
#if defined(__HIPCC__)
      constexpr int NC = 2, J0 = 1;
#else
      constexpr int NC = 4, J0 = 0;
#endif
      float g[S][2][3];
      {
        float mf[S][3][NC];
        {
          float sw[3][NC];
          load_cols<NC>(sel, qb + J0, R1W, sw);
#pragma unroll
          for (int s = 0; s < S; ++s)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
              for (int j = 0; j < NC; ++j) {
                if constexpr (!AVG && S == 2)  // the plane holds -1, 0 or 1: one instruction per mask
                  mf[s][i][j] = s == 0 ? 1.0f - fabsf(sw[i][j]) : fminf(fmaxf(sw[i][j], 0.0f), 1.0f);  // clamp: one instruction
                else
                  mf[s][i][j] = AVG ? (float)(sw[i][j] > 0.0f) : (float)(sw[i][j] == (float)s);
              }
        }
        // (row 1 of the masks is never folded, so mf[s][1][1 + k - J0] stays the mask of the pixel's own window)
        if (border) {
#pragma unroll
          for (int s = 0; s < (AVG ? 1 : S); ++s)
#pragma unroll
            for (int j = 0; j < NC; ++j) {
              mf[s][0][j] *= fy[0];
              mf[s][2][j] *= fy[2];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float xv[S][2], yv[2];
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            yv[k] = lds[OFF_TGT + c * P2 + p2 + k];
#pragma unroll
            for (int s = 0; s < S; ++s) xv[s][k] = lds[OFF_WRP + (s * 3 + c) * P2 + p2 + k];
          }
          float gs[S][2];  // sum A + x sum B + y sum G
#pragma unroll
          for (int s = 0; s < S; ++s) gs[s][0] = gs[s][1] = 0.0f;
          if (!kp.no_ssim) {
#pragma unroll
            for (int e = 0; e < 3; ++e) {
              float cw[3][NC];
              if (!AVG) load_cols<NC>(coef + (c * 3 + e) * P1, qb + J0, R1W, cw);
#pragma unroll
              for (int s = 0; s < S; ++s) {
                if (AVG) load_cols<NC>(coef + (s * 9 + c * 3 + e) * P1, qb + J0, R1W, cw);
                const int ms = AVG ? 0 : s;  // avg_reprojection: one mask (warped beats identity) for all sources
                float own[NC];
#pragma unroll
                for (int j = 0; j < NC; ++j)
                  own[j] = fmaf(mf[ms][2][j], cw[2][j], fmaf(mf[ms][1][j], cw[1][j], mf[ms][0][j] * cw[0][j]));
#if defined(__HIPCC__)
                const float col[4] = {lane_left(own[1]), own[0], own[1], lane_right(own[0])};
#else
                const float col[4] = {own[0], own[1], own[2], own[3]};
#endif
                float a2[2];
                if (!border) {
                  two_windows(col, a2);
                } else {
#pragma unroll
                  for (int k = 0; k < 2; ++k) a2[k] = fmaf(fxm[k][0], col[k], fmaf(fxm[k][2], col[k + 2], col[k + 1]));
                }
#pragma unroll
                for (int k = 0; k < 2; ++k)
                  gs[s][k] = (e == 0) ? a2[k] : fmaf(a2[k], (e == 1) ? xv[s][k] : yv[k], gs[s][k]);
              }
            }
          }
          if constexpr (AVG) {
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
              for (int k = 0; k < 2; ++k)
                g[s][k][c] = hv.cst_ssim * gs[s][k] - hv.cst_l1 * mf[0][1][1 + k - J0] * sgn(yv[k] - xv[s][k]);
          } else {
            // At most one source wins a pixel, and only the winner's L1 term is non-zero: one sign per pixel and
            // channel, of the difference to the winner (to source 0 where nobody wins, times a zero mask), instead
            // of one per source.  Same values: the masked products are exact.
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              float xw = xv[0][k];
#pragma unroll
              for (int s = 1; s < S; ++s) xw = mf[s][1][1 + k - J0] != 0.0f ? xv[s][k] : xw;
              const float sg = sgn(yv[k] - xw);
#pragma unroll
              for (int s = 0; s < S; ++s) g[s][k][c] = fmaf(hv.cst_ssim, gs[s][k], -((hv.cst_l1 * mf[s][1][1 + k - J0]) * sg));
            }
          }
          sched_fence();
        }
      }

      float D[2], r[2][3];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float d = lds[OFF_DSP + p2 + k];
        D[k] = fast_rcp(fmaf(kp.c_disp, d, kp.a_disp));  // gradient only: a plain reciprocal is enough
        const float fx = (float)(jx0 + k), fyv = (float)jy;
#pragma unroll
        for (int i = 0; i < 3; ++i) r[k][i] = fmaf(ikm[i * 3 + 2], 1.0f, fmaf(ikm[i * 3 + 1], fyv, ikm[i * 3 + 0] * fx));
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        // tap derivatives d warped_c / d ix, d iy (zero outside the clip range) and the chain to (ix, iy)
        float gix[2] = {0.0f, 0.0f}, giy[2] = {0.0f, 0.0f};
        if constexpr (CARRY) {
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const unsigned bits = st.live >> (2 * (k * S + s));
            const float lx = (float)(bits & 1u), ly = (float)((bits >> 1) & 1u);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              gix[k] = fmaf(g[s][k][c], st.dp[k][s][c][0], gix[k]);
              giy[k] = fmaf(g[s][k][c], st.dp[k][s][c][1], giy[k]);
            }
            gix[k] *= lx;
            giy[k] *= ly;
          }
        } else {
          // fetch the bilinear taps of both pixels again (cache hits)
          float tv[2][3][4];
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const unsigned pk = (unsigned)chn[k][s].packed;
            const unsigned o0 = pk & 0x0FFFFFFFu;
            const unsigned o1 = o0 + (((pk >> 29) & 1u) ? (unsigned)W : 0u);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const float* pc = src_image(kp, s) + ((size_t)bk.b * 3 + c) * HW;
              const TexelPair a = load_texel_pair_at(pc, o0 << 2), b = load_texel_pair_at(pc, o1 << 2);
              tv[k][c][0] = a.lo;
              tv[k][c][1] = a.hi;
              tv[k][c][2] = b.lo;
              tv[k][c][3] = b.hi;
            }
          }
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const Chain ch = chn[k][s];
            const float lx = (float)((ch.packed >> 30) & 1), ly = (float)((ch.packed >> 31) & 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const float v00 = tv[k][c][0], v01 = tv[k][c][1], v10 = tv[k][c][2], v11 = tv[k][c][3];
              const float dx0 = v01 - v00, dy0 = v10 - v00, dxy = (v11 - v10) - dx0;  // as in phase 1
              const float dpx = fmaf(ch.ty, dxy, dx0);
              const float dpy = fmaf(ch.tx, dxy, dy0);
              gix[k] = fmaf(g[s][k][c], dpx, gix[k]);
              giy[k] = fmaf(g[s][k][c], dpy, giy[k]);
            }
            gix[k] *= lx;
            giy[k] *= ly;
          }
        }
        float P[12];
        source_matrix(kp, bk, st, s, P);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          if (!ok[k]) {
            gcs[s][k][0] = gcs[s][k][1] = gcs[s][k][2] = 0.0f;
            continue;
          }
          // d ix / d gx = W/2, d gx / d u = 2/(W-1)
          const float gu = gix[k] * kp.ix_scale, gv = giy[k] * kp.iy_scale;
          // projection of this pixel again (layers.py:193-196); gradient only, so a plain reciprocal
          float cam[3];
#pragma unroll
          for (int i = 0; i < 3; ++i)
            cam[i] = fmaf(P[i * 4 + 0], D[k] * r[k][0], fmaf(P[i * 4 + 1], D[k] * r[k][1], fmaf(P[i * 4 + 2], D[k] * r[k][2], P[i * 4 + 3])));
          const float id = fast_rcp(cam[2] + kp.eps);
          const float gc[3] = {gu * id, gv * id, -(gu * cam[0] + gv * cam[1]) * id * id};
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const float gX = P[0 * 4 + i] * gc[0] + P[1 * 4 + i] * gc[1] + P[2 * 4 + i] * gc[2];
            gD[k] = fmaf(gX, r[k][i], gD[k]);
          }
#pragma unroll
          for (int i = 0; i < 3; ++i) gcs[s][k][i] = gc[i];
        }
        sched_fence();
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) g_disp[k] = gD[k] * (-kp.c_disp * D[k] * D[k]);
      // d L / d P[i][j] = sum over pixels of gc[i] * (D r, 1)[j]
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
          for (int j = 0; j < 3; ++j)
            st.part[4 + 12 * s + i * 4 + j] = gcs[s][0][i] * (D[0] * r[0][j]) + gcs[s][1][i] * (D[1] * r[1][j]);
          st.part[4 + 12 * s + i * 4 + 3] = gcs[s][0][i] + gcs[s][1][i];
        }
    }

    if (wants_grad(hv)) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (ok[k]) store_f32_at(hv.gphoto + (size_t)bk.b * HW, row_major((unsigned)jy, (unsigned)W, (unsigned)(jx0 + k)) << 2, g_disp[k]);
    }
  }

  // Edge-aware smoothness of the tile, layers.py:211-226 with the 1/(mean+1e-7) factor applied later.  It needs
  // only the disparity and target planes, so the wave that has no interior pixels in phase 3 (R2 rows from TH on)
  // computes it for the whole tile while the other waves do the photometric gradient.
  static VSL_DEV void phase3_smooth(const KParams& kp, const HeadView& hv, const Block& bk, int tid, const float* lds, ThreadState<S>& st) {
    constexpr int NSM = NT - TH * ROWP;  // threads of the idle rows
    for (int q = tid - TH * ROWP; q < TH * (TW / 2); q += NSM) smooth_pair(kp, hv, bk, lds, st, q / (TW / 2), q % (TW / 2));
  }

  // smoothness terms of the interior pixel pair (iy, ip): sums into st.part[1..3], edge gradient to hv.gedge
  static VSL_DEV void smooth_pair(const KParams& kp, const HeadView& hv, const Block& bk, const float* lds, ThreadState<S>& st,
                                  int iy, int ip) {
    const int H = kp.H, W = kp.W;
    const size_t HW = (size_t)H * W;
    const int jy = bk.y0 + iy, jx0 = bk.x0 + 2 * ip;
    if (jy >= H || jx0 >= W) return;
    const int p2 = (HALO + iy) * R2W + HALO + 2 * ip;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int jx = jx0 + k;
      if (jx >= W) continue;
      const int pp = p2 + k;
      const float d0 = lds[OFF_DSP + pp];
      float ax[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // sum_c |I - I_neighbour| for right, left, down, up
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* tp = lds + OFF_TGT + c * P2;
        const float i0 = tp[pp];
        ax[0] += fabsf(i0 - tp[pp + 1]);
        ax[1] += fabsf(tp[pp - 1] - i0);
        ax[2] += fabsf(i0 - tp[pp + R2W]);
        ax[3] += fabsf(tp[pp - R2W] - i0);
      }
      const float dn[4] = {lds[OFF_DSP + pp + 1], lds[OFF_DSP + pp - 1], lds[OFF_DSP + pp + R2W],
                           lds[OFF_DSP + pp - R2W]};
      const bool has[4] = {jx + 1 < W, jx > 0, jy + 1 < H, jy > 0};
      float e = 0.0f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (!has[t]) continue;
        const float w = VSL_EXP(-ax[t] * (1.0f / 3.0f));
        const float inv_n = (t < 2) ? kp.inv_nx : kp.inv_ny;
        // right/down edges are owned by this pixel (first operand), left/up by the neighbour
        const float diff = (t == 0 || t == 2) ? d0 - dn[t] : dn[t] - d0;
        const float sg = sgn(diff) * w * inv_n;
        e += (t == 0 || t == 2) ? sg : -sg;
        if (t == 0) st.part[2] = fmaf(w, fabsf(diff), st.part[2]);
        if (t == 2) st.part[3] = fmaf(w, fabsf(diff), st.part[3]);
      }
      st.part[1] += d0;
      if (wants_grad(hv)) store_f32_at(hv.gedge + (size_t)bk.b * HW, row_major((unsigned)jy, (unsigned)W, (unsigned)jx) << 2, e);
    }
  }

  // ---- forward only (HALO = 1): photometric error, selection and smoothness of the interior pair ----
  // Same arithmetic per pixel as phase 2 (Trainer.py:481-495,551-605), without coefficients; replaces phases 2, 3.
  static VSL_DEV void phase_forward(const KParams& kp, const HeadView& hv, const Block& bk, int tid, const float* lds,
                                    ThreadState<S>& st) {
#pragma unroll
    for (int k = 0; k < NPART; ++k) st.part[k] = 0.0f;
    const int iy = tid / ROWP, ip = tid % ROWP;
    if (iy >= TH || ip >= TW / 2) return;
    const int jy = bk.y0 + iy, jx0 = bk.x0 + 2 * ip;
    if (jy >= kp.H || jx0 >= kp.W) return;
    const bool ok[2] = {true, jx0 + 1 < kp.W};
    const int base = iy * R2W + 2 * ip;  // top-left of the pair's windows in the planes over the tile grown by 1
    const size_t HW = (size_t)kp.H * kp.W;
    const size_t pix0 = (size_t)bk.b * HW + (size_t)jy * kp.W + jx0;
    float best[2] = {INFINITY, INFINITY};
    if (kp.automask) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (ok[k]) best[k] = kp.idmin[pix0 + k];
    }
    bool warped_wins[2] = {false, false};
    float rsum[2] = {0.0f, 0.0f};
    // channel by channel, the target window's differences and statistics shared by all sources (as in phase 2)
    float err[S][2];
#pragma unroll
    for (int s = 0; s < S; ++s) err[s][0] = err[s][1] = 0.0f;
    const float w_l1 = kp.no_ssim ? 1.0f / 3.0f : kp.l1_w3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float yw[3][4], dy[3][4], sy[2], syy[2];
      load_win(lds + OFF_TGT + c * P2, base, R2W, yw);
      if (!kp.no_ssim) {
        y_diffs(yw, dy);
        y_stats_d(dy, sy, syy);
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        float xw[3][4];
        load_win(lds + OFF_WRP + (s * 3 + c) * P2, base, R2W, xw);
        err[s][0] = fmaf(w_l1, fabsf(yw[1][1] - xw[1][1]), err[s][0]);
        err[s][1] = fmaf(w_l1, fabsf(yw[1][2] - xw[1][2]), err[s][1]);
        if (!kp.no_ssim) {
          SsimOut o[2];
          ssim_pair_d(xw, yw[1][1], dy, sy, syy, false, o);
          err[s][0] = fmaf(kp.ssim_w3, o[0].f, err[s][0]);
          err[s][1] = fmaf(kp.ssim_w3, o[1].f, err[s][1]);
        }
      }
      sched_fence();
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float e = err[s][k];
        e *= kp.warp_scale;
        if (AVG) {
          rsum[k] += e;
        } else if (ok[k] && e < best[k]) {
          best[k] = e;
          warped_wins[k] = true;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (AVG) {
        const float e = rsum[k] * (1.0f / S);
        if (ok[k] && e < best[k]) {
          best[k] = e;
          warped_wins[k] = true;
        }
      }
      if (ok[k]) {
        st.part[0] += best[k];
        if (kp.automask && hv.ident_sel) hv.ident_sel[pix0 + k] = warped_wins[k] ? 1.0f : 0.0f;
      }
    }
    smooth_pair(kp, hv, bk, lds, st, iy, ip);
  }

  // ---- phase 4: deterministic block reduction ----------------------------------
  static VSL_DEV void phase4_store(int tid, float* lds, const ThreadState<S>& st) {
    if constexpr (HALO == 2) {
      const int trow = tid / ROWP;
      if (trow >= 1 && trow <= TH) lds[tid - ROWP] = st.part[0];
      // The first three stores pick value and slot with selects: as two branches storing st.part[4 + j] and
      // st.part[1 + j], the compiler merges them into one store through a selected address of st.part, which
      // moves the array to scratch memory (16 B per lane written to HBM by every tile).
      const bool big = trow < TH;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float v = big ? st.part[4 + j] : st.part[1 + j];
        lds[big ? (1 + j) * BIG_PITCH + tid : NBIG * BIG_PITCH + j * SMALL_PITCH + tid - BIG] = v;
      }
      if (big) {
#pragma unroll
        for (int i = 4; i < NBIG; ++i) lds[i * BIG_PITCH + tid] = st.part[3 + i];
      }
    } else {
#pragma unroll
      for (int k = 0; k < NRED; ++k) lds[k * RED_PITCH + tid] = st.part[k];
    }
  }
  // after a barrier: item (row, part) sums every RED_SPLIT-th element of a row in a fixed order
  static VSL_DEV void phase4_sum_a(int tid, float* lds) {
    if constexpr (HALO == 2) {
      // pairs 0..7 of every thread row into one accumulator, pairs 8..15 into another, rows in ascending order
      for (int item = tid; item < (NBIG + 3) * RED_SPLIT; item += NT) {
        const int r = item / RED_SPLIT, part = item % RED_SPLIT;
        const float* e = lds + (r < NBIG ? r * BIG_PITCH : NBIG * BIG_PITCH + (r - NBIG) * SMALL_PITCH) + part;
        const int rows = r < NBIG ? BIG / ROWP : SMALL / ROWP;
        float s0 = 0.0f, s1 = 0.0f;
        for (int m = 0; m < rows; ++m) {
          s0 += e[m * ROWP];
          s1 += e[m * ROWP + RED_SPLIT];
        }
        lds[RED_FLOATS + item] = s0 + s1;
      }
      return;
    }
    for (int item = tid; item < NRED * RED_SPLIT; item += NT) {
      const int row = item / RED_SPLIT, part = item % RED_SPLIT;
      const float* r = lds + row * RED_PITCH;
      float s0 = 0.0f, s1 = 0.0f;
      int t = part;
      for (; t + RED_SPLIT < NT; t += 2 * RED_SPLIT) {
        s0 += r[t];
        s1 += r[t + RED_SPLIT];
      }
      if (t < NT) s0 += r[t];
      lds[RED_FLOATS + item] = s0 + s1;
    }
  }
  // after another barrier: thread k < NPART adds the RED_SPLIT pieces of sum k (sums that cannot be
  // non-zero in the forward-only program are written as zeros: the finalize kernel reads all of them)
  static VSL_DEV void phase4_sum_b(const HeadView& hv, int block_linear, int tid, const float* lds) {
    if (tid >= NPART) return;
    float s = 0.0f;
    if constexpr (HALO == 2) {
      // sum k: 0 photometric, 1..3 smoothness, 4.. gradient sums
      const int r = tid == 0 ? 0 : (tid < 4 ? NBIG + tid - 1 : tid - 3);
      const float* p = lds + RED_FLOATS + r * RED_SPLIT;
#pragma unroll
      for (int i = 0; i < RED_SPLIT; ++i) s += p[i];
    } else if (tid < NRED) {
      const float* r = lds + RED_FLOATS + tid * RED_SPLIT;
#pragma unroll
      for (int i = 0; i < RED_SPLIT; ++i) s += r[i];
    }
    hv.partials[(size_t)block_linear * kPartStride + tid] = s;
  }
};

}  // namespace vsl
