// Diagnostic: issue cost of the vector-ALU instruction classes the tile kernels use, at 1, 2 and 4 waves per SIMD,
// with 8 independent chains per wave and with one dependent chain per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_ __attribute__((ext_vector_type(2)));

enum { FMA32, PKFMA32, FMA64, RCP32, MULLO, CVT64, DPPMOV, CNDMASK, FMA32_3SRC, NMODES };
static const char* kNames[NMODES] = {"v_fmac_f32", "v_pk_fma_f32", "v_fma_f64", "v_rcp_f32", "v_mul_lo_u32",
                                     "v_cvt_f64_f32+back", "v_mov_dpp", "v_cndmask", "v_fma_f32 3 vgpr"};

template <int MODE, int CHAINS>
__global__ void __launch_bounds__(1024) k(float* out, int iters) {
  float a[8];
  float2_ p[8];
  double d[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = threadIdx.x * 0.001f + i + 1.0f;
    p[i] = float2_{a[i], a[i] + 1.f};
    d[i] = a[i];
    u[i] = threadIdx.x * 7 + i;
  }
  const float m = 1.0001f, c = 0.5f;
  const float2_ m2 = {m, m}, c2 = {c, c};
  float x = out[0], y = out[1];  // run-time values in vector registers
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 64 / CHAINS; ++r)
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) {
        if (MODE == FMA32) a[i] = __builtin_fmaf(a[i], m, c);
        if (MODE == PKFMA32) p[i] = __builtin_elementwise_fma(p[i], m2, c2);
        if (MODE == FMA64) d[i] = __builtin_fma(d[i], 1.0001, 0.5);
        if (MODE == RCP32) a[i] = __builtin_amdgcn_rcpf(a[i]);
        if (MODE == MULLO) u[i] = u[i] * (u[i] | 3u);
        if (MODE == CVT64) { d[i] = (double)a[i]; asm volatile("" : "+v"(d[i])); a[i] = (float)d[i]; }
        if (MODE == DPPMOV) a[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0x111, 0xf, 0xf, false));
        if (MODE == CNDMASK) a[i] = a[i] > x ? y : a[i];
        if (MODE == FMA32_3SRC) a[i] = __builtin_fmaf(a[i], x, y);
      }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)d[i] + (float)u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int CHAINS>
void run(int threads, float* d) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, CHAINS>), dim3(256), dim3(threads), 0, 0, d, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, CHAINS>), dim3(256), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_wave = (double)iters * 64 * (MODE == CVT64 ? 2 : 1);
  const double waves_per_simd = threads / 64.0 / 4.0;
  const double ns = ms * 1e6 / (instr_per_wave * waves_per_simd);
  printf("%-20s chains %d  waves/SIMD %.0f: %8.3f ms -> %6.2f ns per wave-instr per SIMD = %5.2f cycles @2.4GHz; per wave %5.2f cycles\n",
         kNames[MODE], CHAINS, waves_per_simd, ms, ns, ns * 2.4, ns * 2.4 * waves_per_simd);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

template <int MODE>
void sweep(float* d) {
  for (int t : {256, 512, 1024}) run<MODE, 8>(t, d);
  for (int t : {256, 512, 1024}) run<MODE, 1>(t, d);
}

int main() {
  float* d;
  if (hipMalloc(&d, 256 * 1024 * 4) != hipSuccess) return 1;
  hipMemset(d, 0, 256 * 1024 * 4);
  sweep<FMA32>(d);
  sweep<PKFMA32>(d);
  sweep<FMA64>(d);
  sweep<RCP32>(d);
  sweep<MULLO>(d);
  sweep<CVT64>(d);
  sweep<DPPMOV>(d);
  sweep<CNDMASK>(d);
  sweep<FMA32_3SRC>(d);
  hipFree(d);
  return 0;
}
