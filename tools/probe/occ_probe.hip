// Diagnostic: how many workgroups of a given shape does one CU of this GPU really hold?
// Each block spins ~20 us and records (CU id, start, end) from s_memrealtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <map>

template <int NREG>
__global__ void __launch_bounds__(1024) probe(unsigned long long* out, int spin_us, float* sink) {
  extern __shared__ float lds[];
  float r[NREG];
#pragma unroll
  for (int i = 0; i < NREG; ++i) r[i] = threadIdx.x * 0.5f + i;
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t = t0;
  while (t - t0 < (unsigned long long)spin_us * 100ull) {
#pragma unroll
    for (int i = 0; i < NREG; ++i) r[i] = fmaf(r[i], 1.0001f, 0.5f);
    t = __builtin_amdgcn_s_memrealtime();
  }
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < NREG; ++i) acc += r[i];
  if (acc == 12345.678f) sink[0] = acc + lds[threadIdx.x];
  if (threadIdx.x == 0) {
    unsigned int hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[blockIdx.x * 3 + 0] = t0;
    out[blockIdx.x * 3 + 1] = t;
    out[blockIdx.x * 3 + 2] = ((unsigned long long)(xcc & 0xF) << 32) | hw;
  }
}

template <int NREG>
void run(int threads, size_t lds, const char* tag) {
  const int blocks = 256 * 12;
  unsigned long long* d;
  float* sink;
  hipMalloc(&d, blocks * 3 * 8);
  hipMalloc(&sink, 4096);
  hipFuncSetAttribute((const void*)probe<NREG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int api = -1;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, probe<NREG>, threads, lds);
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, (const void*)probe<NREG>);
  hipLaunchKernelGGL(probe<NREG>, dim3(blocks), dim3(threads), lds, 0, d, 20, sink);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 3);
  hipMemcpy(h.data(), d, blocks * 3 * 8, hipMemcpyDeviceToHost);
  std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
  for (int b = 0; b < blocks; ++b) {
    unsigned long long hw = h[b * 3 + 2];
    unsigned long long cu = ((hw >> 32) << 16) | ((hw >> 8) & 0xFF) | (((hw >> 13) & 7) << 8);  // xcc, se, sh+cu
    ev[cu].push_back({h[b * 3 + 0], 1});
    ev[cu].push_back({h[b * 3 + 1], -1});
  }
  int mx = 0, mn = 1 << 30;
  for (auto& kv : ev) {
    std::sort(kv.second.begin(), kv.second.end());
    int cur = 0, m = 0;
    for (auto& e : kv.second) { cur += e.second; m = std::max(m, cur); }
    mx = std::max(mx, m);
    mn = std::min(mn, m);
  }
  printf("%-28s threads %4d lds %6zu vgprs %3d  api %d  measured max/CU %d (min over CUs %d) CUs %zu\n", tag, threads, lds,
         fa.numRegs, api, mx, mn, ev.size());
  hipFree(d);
  hipFree(sink);
}

int main() {
  run<100>(320, 47400, "5w 128v 47K");
  run<100>(320, 1024, "5w 128v 1K");
  run<100>(256, 47400, "4w 128v 47K");
  run<100>(256, 1024, "4w 128v 1K");
  run<100>(384, 47400, "6w 128v 47K");
  run<70>(320, 47400, "5w ~96v 47K");
  run<70>(320, 1024, "5w ~96v 1K");
  run<40>(320, 47400, "5w ~64v 47K");
  run<40>(320, 32000, "5w ~64v 32K");
  run<40>(320, 1024, "5w ~64v 1K");
  run<40>(256, 32000, "4w ~64v 32K");
  run<40>(256, 40000, "4w ~64v 40K");
  run<40>(256, 53000, "4w ~64v 53K");
  run<40>(512, 53000, "8w ~64v 53K");
  run<40>(640, 53000, "10w ~64v 53K");
  return 0;
}
