// Diagnostic: VALU issue rate of plain vs packed f32 FMA at 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_ __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, int iters) {
  float a[8];
  float2_ p[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = float2_{a[i], a[i] + 1.f}; }
  const float m = 1.0001f, c = 0.5f;
  const float2_ m2 = {m, m}, c2 = {c, c};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], m, c);
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], m2, c2);
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(int threads, const char* tag) {
  float* d;
  hipMalloc(&d, 256 * 1024 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr_per_wave = (double)iters * 64;
  double waves_per_simd = threads / 64.0 / 4.0;
  double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * waves_per_simd);
  printf("%-10s threads/CU %4d (%.0f waves/SIMD): %.3f ms -> %.2f ns per wave-instr per SIMD = %.2f cycles @2.4GHz; %s flops/instr/lane\n", tag, threads,
         waves_per_simd, ms, ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4, MODE ? "4" : "2");
  hipFree(d);
}
int main() {
  for (int t : {256, 512, 1024}) run<0>(t, "v_fma");
  for (int t : {256, 512, 1024}) run<1>(t, "v_pk_fma");
  return 0;
}
