// Diagnostic: what a wave-wide global load costs in the vector L1 (TCP) for the address patterns of the bilinear
// gathers.  One kernel launch per pattern (so that rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD can
// give accesses per instruction), data resident in a 64 KB window, 4 waves per SIMD on every CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_ __attribute__((ext_vector_type(2)));
typedef float float4_ __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(1024) k(const float* __restrict__ src, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = lane >> 4, col = lane & 15;
  unsigned off;  // in floats
  if (MODE == 0) off = lane;                                   // dword, contiguous
  if (MODE == 1) off = 2 * lane;                               // dwordx2, contiguous, 8-byte aligned
  if (MODE == 2) off = 2 * lane + 1;                           // dwordx2, contiguous, misaligned by 4 bytes
  if (MODE == 3) off = lane;                                   // dwordx2 at 4-byte stride (overlapping pairs)
  if (MODE == 4) off = row * 640 + 2 * col;                    // dwordx2, four image rows of 16 aligned pairs
  if (MODE == 5) off = row * 640 + 2 * col + 1;                // the same, misaligned by 4 bytes
  if (MODE == 6) off = row * 640 + 2 * col + ((col * 5) >> 4); // the same with a slowly growing sub-pixel drift
  if (MODE == 7) off = 2 * lane;                               // dwordx4 at 8-byte stride (overlapping)
  if (MODE == 8) off = 4 * lane;                               // dwordx4 contiguous
  if (MODE == 9) off = row * 640 + 2 * col;                    // dwordx4, four rows, 8-byte stride
  if (MODE == 10) off = row * 640 + 2 * col + 1;               // dwordx4, four rows, 8-byte stride, misaligned
  if (MODE == 11) off = row * 640 + col;                       // dword, four rows of 16
  off += wave * 2560;
  float acc = 0.0f;
  for (int it = 0; it < iters; ++it) {
    const unsigned o = off + (unsigned)((it & 7) * 16);  // stays inside the window, changes so that loads are not hoisted
    if (MODE == 0 || MODE == 11) {
      acc += __builtin_nontemporal_load(src + o);
    } else if (MODE <= 6) {
      const float2_ v = *reinterpret_cast<const float2_*>(src + o);
      acc += v.x + v.y;
    } else {
      const float4_ v = *reinterpret_cast<const float4_*>(src + o);
      acc += v.x + v.y + v.z + v.w;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
void run(const float* src, float* out, const char* name) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, src, out, 200);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, src, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  // 16 waves per CU issue `iters` loads each
  printf("%2d %-62s %7.3f ms  %6.2f ns per wave-load per CU\n", MODE, name, ms, ms * 1e6 / (16.0 * iters));
}

int main() {
  float *src, *out;
  if (hipMalloc(&src, 1 << 20) != hipSuccess || hipMalloc(&out, 256 * 1024 * 4) != hipSuccess) return 1;
  (void)hipMemset(src, 0, 1 << 20);
  for (int w = 0; w < 30; ++w) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, 0, src, out, 20000);  // clocks settle
  (void)hipDeviceSynchronize();
  run<0>(src, out, "dword, contiguous");
  run<1>(src, out, "dwordx2, contiguous, aligned");
  run<2>(src, out, "dwordx2, contiguous, misaligned by 4 bytes");
  run<3>(src, out, "dwordx2 at 4-byte stride (overlapping pairs)");
  run<4>(src, out, "dwordx2, four rows of 16 aligned pairs");
  run<5>(src, out, "dwordx2, four rows of 16 pairs, misaligned by 4 bytes");
  run<6>(src, out, "dwordx2, four rows of 16 pairs with sub-pixel drift");
  run<7>(src, out, "dwordx4 at 8-byte stride (overlapping)");
  run<8>(src, out, "dwordx4, contiguous");
  run<9>(src, out, "dwordx4, four rows, 8-byte stride");
  run<10>(src, out, "dwordx4, four rows, 8-byte stride, misaligned by 4 bytes");
  run<11>(src, out, "dword, four rows of 16");
  return 0;
}
