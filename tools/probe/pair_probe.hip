// Diagnostic: which adjacent vector-ALU instructions of one wave share an issue slot on gfx950.
// Every pattern is a short instruction sequence repeated inside a loop; the result is SIMD cycles per vector
// instruction at 4 waves per SIMD (a slot of about 4 cycles that holds one instruction reads ~4, two read ~2).
#include <hip/hip_runtime.h>
#include <cstdio>

#define R2(x) x x
#define R4(x) R2(x) R2(x)
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)
#define R32(x) R16(x) R16(x)
#define R512(x) R32(R16(x))
#define R1024(x) R32(R32(x))

#define OPERANDS                                                                                          \
  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(d0), "+v"(d1)                                              \
  : "v"(x), "v"(y), "s"(sx)                                                                                \
  : "vcc", "s20", "s21", "s22", "s23", "v60", "v61", "v62", "v64"

#define PATTERN(ID, NVALU, REPS, TEXT)                                                                     \
  if (mode == ID) {                                                                                        \
    n_valu = NVALU;                                                                                        \
    for (int it = 0; it < iters; ++it) asm volatile(REPS(TEXT) OPERANDS);                                  \
  }

__global__ void __launch_bounds__(1024) k(float* out, const float* in, int iters, int mode, int* n_out) {
  float a0 = threadIdx.x * 0.001f + 1.0f, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f;
  double d0 = a0, d1 = a1;
  const float x = in[0] + 1.0001f, y = in[1] + 0.5f;   // in[] holds zeros
  const int sx = __builtin_amdgcn_readfirstlane(__float_as_int(x));
  int n_valu = 0;
  asm volatile("v_cmp_gt_f32 vcc, %0, %1\n s_mov_b64 s[20:21], vcc\n v_mov_b32 v61, 0" ::"v"(a0), "v"(a1) : "vcc", "s20", "s21", "v61");
  // NVALU counts the vector instructions of one repetition unit times the repetitions
  PATTERN(0, 32, R16, "v_fmac_f32 %0, %6, %7\n v_fmac_f32 %1, %6, %7\n")
  PATTERN(1, 32, R8, "v_fmac_f32 %0, %6, %7\n v_fmac_f32 %0, %6, %7\n v_fmac_f32 %1, %6, %7\n v_fmac_f32 %1, %6, %7\n")
  PATTERN(2, 32, R16, "v_mul_f32 %0, %0, %6\n v_add_f32 %1, %1, %7\n")
  PATTERN(3, 32, R16, "v_fmac_f32 %0, %6, %7\n v_mov_b32 %1, %6\n")
  PATTERN(4, 32, R16, "v_mov_b32 %0, %6\n v_mov_b32 %1, %7\n")
  PATTERN(5, 32, R16, "v_add_u32 %0, %0, %6\n v_add_u32 %1, %1, %7\n")
  PATTERN(6, 32, R16, "v_fmac_f32 %0, %6, %7\n s_nop 0\n v_fmac_f32 %1, %6, %7\n s_nop 0\n")
  PATTERN(7, 32, R16, "v_fmac_f32 %0, %6, %7\n s_mov_b32 s20, 0\n v_fmac_f32 %1, %6, %7\n s_mov_b32 s21, 0\n")
  PATTERN(8, 32, R16, "v_fmac_f32 %0, %8, %7\n v_fmac_f32 %1, %8, %7\n")
  PATTERN(9, 32, R16, "v_add_f32 %0, 0x3f8ccccd, %0\n v_add_f32 %1, 0x3f8ccccd, %1\n")
  PATTERN(10, 32, R8, "v_fmac_f32 %0, %1, %7\n v_fmac_f32 %1, %2, %7\n v_fmac_f32 %2, %3, %7\n v_fmac_f32 %3, %0, %7\n")
  PATTERN(11, 32, R16, "v_cndmask_b32 %0, %0, %6, vcc\n v_cndmask_b32 %1, %1, %7, vcc\n")
  PATTERN(12, 32, R16, "v_cmp_gt_f32 s[20:21], %0, %6\n v_cmp_gt_f32 s[22:23], %1, %7\n")
  PATTERN(13, 32, R16, "v_max_f32 %0, %0, %6\n v_min_f32 %1, %1, %7\n")
  PATTERN(14, 32, R16, "v_fmac_f32 v60, %6, %7\n v_fmac_f32 v64, %6, %7\n")
  PATTERN(15, 32, R16, "v_fmac_f32 v60, %6, %7\n v_fmac_f32 v61, %6, %7\n")
  PATTERN(16, 48, R16, "v_fmac_f32 %0, %6, %7\n v_fmac_f32 %1, %6, %7\n v_fmac_f32 %2, %6, %7\n")
  PATTERN(17, 32, R16, "v_fmac_f32 %0, %1, %7\n v_fmac_f32 %1, %0, %7\n")
  PATTERN(18, 32, R16, "v_fma_f32 %0, %0, %6, %7\n v_fma_f32 %1, %1, %6, %7\n")
  PATTERN(19, 32, R16, "v_mul_f32 %0, -%0, |%6|\n v_mul_f32 %1, -%1, |%7|\n")
  PATTERN(20, 32, R8, "v_rcp_f32 %0, %0\n v_fmac_f32 %1, %6, %7\n v_fmac_f32 %2, %6, %7\n v_fmac_f32 %3, %6, %7\n")
  PATTERN(21, 32, R16, "v_fma_f64 %4, %4, %4, %4\n v_fmac_f32 %1, %6, %7\n")
  PATTERN(22, 32, R16, "v_cmp_gt_f32 vcc, %0, %6\n v_cndmask_b32 %1, %1, %7, vcc\n")
  PATTERN(23, 32, R16, "v_fmac_f32 %0, %6, %7\n v_cndmask_b32 %1, %1, %7, vcc\n")
  PATTERN(24, 32, R16, "v_sub_f32 %0, %0, %6\n v_mul_f32 %1, %1, %7\n")
  PATTERN(25, 32, R16, "v_fmac_f32 %0, %6, %7\n v_add_u32 %1, %1, %7\n")
  PATTERN(26, 32, R16, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32 %1, %6, %7\n")
  PATTERN(27, 32, R16, "v_fma_f64 %4, %4, %4, %4\n v_fma_f64 %5, %5, %5, %5\n")
  PATTERN(28, 32, R16, "v_lshlrev_b32 %0, 1, %0\n v_and_b32 %1, %1, %7\n")
  PATTERN(29, 32, R16, "v_fmac_f32 %0, %6, %7\n v_cmp_gt_f32 s[22:23], %1, %7\n")
  PATTERN(30, 32, R16, "v_cvt_f32_i32 %0, %0\n v_cvt_i32_f32 %1, %1\n")
  PATTERN(31, 32, R16, "v_mad_u32_u24 %0, %0, %6, %7\n v_mad_u32_u24 %1, %1, %6, %7\n")
  PATTERN(32, 32, R32, "v_fma_f32 %0, %0, %8, 0.5\n")
  PATTERN(33, 32, R32, "v_fma_f32 %0, %0, %6, %7\n")
  PATTERN(34, 32, R32, "v_fmac_f32 %0, %6, %7\n")
  PATTERN(35, 32, R32, "v_mul_f32 %0, %0, %6\n")
  PATTERN(36, 32, R32, "v_mul_f32 %0, %6, %0\n")
  PATTERN(37, 32, R16, "v_fma_f32 %0, %0, %8, 0.5\n v_fma_f32 %1, %1, %8, 0.5\n")
  PATTERN(38, 32, R16, "v_mul_f32 %0, %1, %6\n v_mul_f32 %1, %0, %7\n")
  PATTERN(39, 32, R16, "v_add_f32 %0, %0, %6\n v_add_f32 %0, %0, %7\n")
  PATTERN(40, 32, R16, "v_max_f32 %0, %0, %6\n v_fmac_f32 %1, %6, %7\n")
  PATTERN(41, 32, R16, "v_and_b32 %0, %0, %6\n v_fmac_f32 %1, %6, %7\n")
  PATTERN(42, 32, R16, "v_cndmask_b32 %0, %0, %6, vcc\n v_cndmask_b32 %1, %1, %7, vcc\n")
  PATTERN(43, 32, R16, "v_cndmask_b32_e64 %0, %0, %6, s[20:21]\n v_cndmask_b32_e64 %1, %1, %7, s[20:21]\n")
  PATTERN(44, 24, R8, "v_fmac_f32 %0, %6, %7\n v_fmac_f32 %1, %6, %7\n v_fmac_f32 %2, %6, %7\n ds_read_b32 v62, v61\n")
  PATTERN(45, 32, R32, "v_fma_f32 %0, %0, %8, 0.5\n")
  PATTERN(46, 32, R16, "v_fma_f32 %0, %0, %8, 0.5\n v_fma_f32 %1, %1, %8, 0.5\n")
  PATTERN(47, 32, R32, "v_fmac_f32 %0, %8, %7\n")
  PATTERN(48, 32, R32, "v_mul_f32 %0, %8, %0\n")
  PATTERN(49, 32, R32, "v_fma_f32 %0, %0, %6, 0.5\n")
  PATTERN(50, 32, R32, "v_mul_f32 %0, 0x3f800347, %0\n")
  PATTERN(51, 32, R8, "v_fma_f32 %0, %0, %8, 0.5\n v_fma_f32 %1, %1, %8, 0.5\n v_fma_f32 %2, %2, %8, 0.5\n v_fma_f32 %3, %3, %8, 0.5\n")
  PATTERN(52, 32, R16, "v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n")
  PATTERN(53, 32, R16, "v_mul_f32 %0, %8, %0\n v_add_f32 %0, %6, %0\n")
  PATTERN(54, 32, R16, "v_mul_f32 %0, %8, %0\n v_max_f32 %1, %1, %7\n")
  PATTERN(55, 32, R16, "v_mul_f32 %0, %8, %0\n v_cmp_gt_f32 s[22:23], %1, %7\n")
  PATTERN(56, 32, R16, "v_max_f32 %0, %0, %6\n v_cmp_gt_f32 s[22:23], %1, %7\n")
  PATTERN(57, 32, R16, "v_max_f32 %0, %0, %6\n v_and_b32 %1, %1, %7\n")
  PATTERN(58, 32, R16, "v_cndmask_b32_e64 %0, %0, %6, s[20:21]\n v_mul_f32 %1, %8, %1\n")
  PATTERN(59, 32, R16, "v_sub_u32 %0, %0, %6\n v_sub_u32 %1, %1, %7\n")
  PATTERN(60, 32, R16, "v_fmaak_f32 %0, %0, %6, 0x3f800347\n v_fmaak_f32 %1, %1, %7, 0x3f800347\n")
  PATTERN(61, 48, R16, "v_mul_f32 %0, %8, %0\n v_fmac_f32 %1, %6, %7\n v_fmac_f32 %2, %6, %7\n")
  PATTERN(62, 32, R16, "v_cvt_f32_i32 %0, %0\n v_fmac_f32 %1, %6, %7\n")
  PATTERN(63, 32, R16, "v_mad_u32_u24 %0, %0, %6, %7\n v_fmac_f32 %1, %6, %7\n")
  PATTERN(64, 32, R8, "v_mul_lo_u32 %0, %0, %6\n v_fmac_f32 %1, %6, %7\n v_fmac_f32 %2, %6, %7\n v_fmac_f32 %3, %6, %7\n")
  PATTERN(65, 32, R8, "v_fma_f64 %4, %4, %4, %4\n v_fmac_f32 %1, %6, %7\n v_fmac_f32 %2, %6, %7\n v_fmac_f32 %3, %6, %7\n")
  PATTERN(66, 32, R4, "v_rcp_f32 %0, %0\n v_fmac_f32 %1, %6, %7\n v_fmac_f32 %2, %6, %7\n v_fmac_f32 %3, %6, %7\n v_fmac_f32 %1, %6, %7\n v_fmac_f32 %2, %6, %7\n v_fmac_f32 %3, %6, %7\n v_fmac_f32 %1, %6, %7\n")
  PATTERN(67, 32, R8, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32 %1, %6, %7\n v_fmac_f32 %2, %6, %7\n v_fmac_f32 %3, %6, %7\n")
  PATTERN(68, 32, R16, "v_cmp_gt_f32 vcc, %0, %6\n v_fmac_f32 %1, %6, %7\n")
  PATTERN(69, 24, R8, "s_mov_b32 s20, 0\n v_fmac_f32 %1, %6, %7\n v_fmac_f32 %2, %6, %7\n v_fmac_f32 %3, %6, %7\n")
  PATTERN(70, 24, R8, "s_waitcnt lgkmcnt(0)\n v_fmac_f32 %1, %6, %7\n v_fmac_f32 %2, %6, %7\n v_fmac_f32 %3, %6, %7\n")
  PATTERN(71, 32, R16, "v_fma_f64 %4, %4, %4, %4\n v_add_f64 %5, %5, %5\n")
  PATTERN(72, 32, R16, "v_cvt_f64_f32 %4, %0\n v_fmac_f32 %1, %6, %7\n")
  PATTERN(73, 32, R8, "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32 %1, %6, %7\n v_fmac_f32 %2, %6, %7\n v_fmac_f32 %3, %6, %7\n")
  PATTERN(74, 32, R16, "v_med3_f32 %0, %0, %6, %7\n v_fmac_f32 %1, %6, %7\n")
  PATTERN(75, 32, R16, "v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n")
  // long straight-line bodies (2048 instructions, the size of the tile kernels): does instruction fetch keep up?
  PATTERN(76, 2048, R1024, "v_fmac_f32 %0, %6, %7\n v_fmac_f32 %1, %6, %7\n")
  PATTERN(77, 2048, R1024, "v_fma_f32 %0, %0, %6, %7\n v_fma_f32 %1, %1, %6, %7\n")
  PATTERN(78, 2048, R1024, "v_fmaak_f32 %0, %0, %6, 0x3f800347\n v_fmaak_f32 %1, %1, %7, 0x3f800347\n")
  PATTERN(79, 2048, R512, "v_fma_f32 %0, %0, %6, %7\n v_fmac_f32 %1, %6, %7\n v_fma_f32 %2, %2, %6, %7\n v_fmac_f32 %3, %6, %7\n")
  if (threadIdx.x == 0 && blockIdx.x == 0) *n_out = n_valu;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (float)d0 + (float)d1;
}

static const char* kNames[] = {
    "fmac A / fmac B", "fmac A A B B", "mul A / add B", "fmac A / mov B", "mov A / mov B", "add_u32 A / add_u32 B",
    "fmac A / s_nop / fmac B / s_nop", "fmac A / s_mov / fmac B / s_mov", "fmac with SGPR operand", "add with literal",
    "fmac ring (write-after-read)", "cndmask A / cndmask B (vcc)", "cmp -> sgpr pair, A / B", "max A / min B",
    "fmac v60 / fmac v64", "fmac v60 / fmac v61", "fmac A / B / C", "fmac A<-B / fmac B<-A (dependent)", "fma e64 A / B",
    "mul with neg/abs A / B", "rcp A / fmac B C D", "fma_f64 / fmac_f32", "cmp vcc / cndmask vcc (dependent)",
    "fmac A / cndmask B", "sub A / mul B", "fmac A / add_u32 B", "mov_dpp A / fmac B", "fma_f64 A / fma_f64 B",
    "lshlrev A / and B", "fmac A / cmp B -> sgpr", "cvt_f32_i32 A / cvt_i32_f32 B", "mad_u32_u24 A / B",
    "fma A<-A sgpr, 0.5 (one chain)", "fma A<-A vgprs (one chain)", "fmac A (one chain)", "mul A<-A src0 (one chain)",
    "mul A<-A src1 (one chain)", "fma sgpr, 0.5 A / B", "mul A<-B / mul B<-A", "add A / add A", "max A / fmac B", "and A / fmac B",
    "cndmask vcc A / B again", "cndmask e64 sgpr mask A / B", "fmac A B C / ds_read (3 valu of 4)",
    "fma A<-A real sgpr, 0.5 (one chain)", "fma real sgpr, 0.5 A / B", "fmac A, sgpr (one chain)", "mul A<-A, sgpr (one chain)",
    "fma A<-A vgpr, 0.5 (one chain)", "mul A<-A literal (one chain)", "fma real sgpr A / B / C / D", "mul sgpr A / B",
    "mul A sgpr / add A vgpr (one chain)",
    "mul sgpr A / max B", "mul sgpr A / cmp B -> sgpr", "max A / cmp B -> sgpr", "max A / and B", "cndmask sgpr-mask A / mul sgpr B",
    "sub_u32 A / B", "fmaak literal A / B", "mul sgpr A / fmac B / fmac C", "cvt A / fmac B", "mad_u32_u24 A / fmac B",
    "mul_lo_u32 A / fmac B C D", "fma_f64 A / fmac B C D", "rcp A / 7 fmac", "mov_dpp A / fmac B C D", "cmp vcc A / fmac B",
    "s_mov / fmac A B C (3 valu of 4)", "s_waitcnt / fmac A B C (3 valu)", "fma_f64 A / add_f64 B", "cvt_f64_f32 A / fmac B",
    "add_f32_dpp A / fmac B C D", "med3 A / fmac B", "mul sgpr A / mul sgpr B (same sgpr)",
    "2048 x fmac (4-byte encoding)", "2048 x fma e64 (8-byte encoding)", "2048 x fmaak (8 bytes with literal)", "2048 x fma e64 / fmac mixed (6 bytes avg)"};

int main() {
  float* d;
  int* n;
  if (hipMalloc(&d, 256 * 1024 * 4) != hipSuccess || hipMalloc(&n, 4) != hipSuccess) return 1;
  (void)hipMemset(d, 0, 256 * 1024 * 4);
  float* zin;
  if (hipMalloc(&zin, 256) != hipSuccess) return 1;
  (void)hipMemset(zin, 0, 256);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 8000;
  for (int w = 0; w < 100; ++w) hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, d, zin, iters, 0, n);  // clocks settle
  (void)hipDeviceSynchronize();
  for (int threads : {1024, 256}) {
    for (int mode = 0; mode < (int)(sizeof(kNames) / sizeof(kNames[0])); ++mode) {
      hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, zin, 100, mode, n);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, zin, mode >= 76 ? iters / 64 : iters, mode, n);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      int n_valu = 0;
      (void)hipMemcpy(&n_valu, n, 4, hipMemcpyDeviceToHost);
      const double waves = threads / 256.0;
      const double ns = ms * 1e6 / ((double)(mode >= 76 ? iters / 64 : iters) * n_valu * waves);
      printf("%2d %-40s waves/SIMD %.0f: %7.3f ms  %5.2f ns per vector instruction per SIMD\n", mode, kNames[mode], waves, ms, ns);
    }
  }
  return 0;
}
