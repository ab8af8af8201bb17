// Issue cost of the vector instructions the s3 attention body is made of (gfx950): cycles per instruction of a
// stream of independent instructions, one wave alone on its SIMD (64 threads) and two waves per SIMD (512 threads).
//   hipcc -O3 --offload-arch=gfx950 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x
#define BODY(name, text)                                                                                       \
  __global__ void name(unsigned long long *out, float seed) {                                                   \
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
    float b0 = seed * 2, b1 = b0 + 1, b2 = b0 + 2, b3 = b0 + 3, b4 = b0 + 4, b5 = b0 + 5, b6 = b0 + 6, b7 = b0 + 7;          \
    unsigned long long t0, t1;                                                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                                \
    for (int it = 0; it < 64; ++it) {                                                                           \
      asm volatile(REP8(text) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),  \
                   "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));               \
    }                                                                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                                \
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;                                               \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 == 12345.678f) out[63] = 1; \
  }
// each text = 8 independent instructions
BODY(k_add, "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_add_f32 %2, %2, %10\n v_add_f32 %3, %3, %11\n v_add_f32 %4, %4, %12\n v_add_f32 %5, %5, %13\n v_add_f32 %6, %6, %14\n v_add_f32 %7, %7, %15\n")
BODY(k_max3, "v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %9, %10\n v_max3_f32 %2, %2, %10, %11\n v_max3_f32 %3, %3, %11, %12\n v_max3_f32 %4, %4, %12, %13\n v_max3_f32 %5, %5, %13, %14\n v_max3_f32 %6, %6, %14, %15\n v_max3_f32 %7, %7, %15, %8\n")
BODY(k_exp, "v_exp_f32 %0, %8\n v_exp_f32 %1, %9\n v_exp_f32 %2, %10\n v_exp_f32 %3, %11\n v_exp_f32 %4, %12\n v_exp_f32 %5, %13\n v_exp_f32 %6, %14\n v_exp_f32 %7, %15\n")
BODY(k_cvtpk, "v_cvt_pk_f16_f32 %0, %8, %9\n v_cvt_pk_f16_f32 %1, %9, %10\n v_cvt_pk_f16_f32 %2, %10, %11\n v_cvt_pk_f16_f32 %3, %11, %12\n v_cvt_pk_f16_f32 %4, %12, %13\n v_cvt_pk_f16_f32 %5, %13, %14\n v_cvt_pk_f16_f32 %6, %14, %15\n v_cvt_pk_f16_f32 %7, %15, %8\n")
BODY(k_cvt, "v_cvt_f32_f16 %0, %8\n v_cvt_f32_f16 %1, %9\n v_cvt_f32_f16 %2, %10\n v_cvt_f32_f16 %3, %11\n v_cvt_f32_f16 %4, %12\n v_cvt_f32_f16 %5, %13\n v_cvt_f32_f16 %6, %14\n v_cvt_f32_f16 %7, %15\n")
BODY(k_mix, "v_fma_mix_f32 %0, %8, -1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %9, -1.0, %10 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %10, -1.0, %11 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %11, -1.0, %12 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %4, %12, -1.0, %13 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %13, -1.0, %14 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, %14, -1.0, %15 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %15, -1.0, %8 op_sel_hi:[1,0,0]\n")
BODY(k_cnd, "v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %9, %10, vcc\n v_cndmask_b32 %2, %10, %11, vcc\n v_cndmask_b32 %3, %11, %12, vcc\n v_cndmask_b32 %4, %12, %13, vcc\n v_cndmask_b32 %5, %13, %14, vcc\n v_cndmask_b32 %6, %14, %15, vcc\n v_cndmask_b32 %7, %15, %8, vcc\n")

BODY(k_sub, "v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %9\n v_sub_f32 %2, %2, %10\n v_sub_f32 %3, %3, %11\n v_sub_f32 %4, %4, %12\n v_sub_f32 %5, %5, %13\n v_sub_f32 %6, %6, %14\n v_sub_f32 %7, %7, %15\n")
BODY(k_mul, "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %9\n v_mul_f32 %2, %2, %10\n v_mul_f32 %3, %3, %11\n v_mul_f32 %4, %4, %12\n v_mul_f32 %5, %5, %13\n v_mul_f32 %6, %6, %14\n v_mul_f32 %7, %7, %15\n")
BODY(k_fmac, "v_fmac_f32 %0, %8, %8\n v_fmac_f32 %1, %9, %9\n v_fmac_f32 %2, %10, %10\n v_fmac_f32 %3, %11, %11\n v_fmac_f32 %4, %12, %12\n v_fmac_f32 %5, %13, %13\n v_fmac_f32 %6, %14, %14\n v_fmac_f32 %7, %15, %15\n")
BODY(k_fma, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %9, %10\n v_fma_f32 %2, %2, %10, %11\n v_fma_f32 %3, %3, %11, %12\n v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %13, %14\n v_fma_f32 %6, %6, %14, %15\n v_fma_f32 %7, %7, %15, %8\n")
BODY(k_max, "v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %9\n v_max_f32 %2, %2, %10\n v_max_f32 %3, %3, %11\n v_max_f32 %4, %4, %12\n v_max_f32 %5, %5, %13\n v_max_f32 %6, %6, %14\n v_max_f32 %7, %7, %15\n")
BODY(k_and, "v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %9\n v_and_b32 %2, %2, %10\n v_and_b32 %3, %3, %11\n v_and_b32 %4, %4, %12\n v_and_b32 %5, %5, %13\n v_and_b32 %6, %6, %14\n v_and_b32 %7, %7, %15\n")
BODY(k_mov, "v_mov_b32 %0, %8\n v_mov_b32 %1, %9\n v_mov_b32 %2, %10\n v_mov_b32 %3, %11\n v_mov_b32 %4, %12\n v_mov_b32 %5, %13\n v_mov_b32 %6, %14\n v_mov_b32 %7, %15\n")
BODY(k_cmp, "v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %9\n v_cmp_lt_f32 vcc, %2, %10\n v_cmp_lt_f32 vcc, %3, %11\n v_cmp_lt_f32 vcc, %4, %12\n v_cmp_lt_f32 vcc, %5, %13\n v_cmp_lt_f32 vcc, %6, %14\n v_cmp_lt_f32 vcc, %7, %15\n")
BODY(k_ldexp, "v_ldexp_f32 %0, %0, %8\n v_ldexp_f32 %1, %1, %9\n v_ldexp_f32 %2, %2, %10\n v_ldexp_f32 %3, %3, %11\n v_ldexp_f32 %4, %4, %12\n v_ldexp_f32 %5, %5, %13\n v_ldexp_f32 %6, %6, %14\n v_ldexp_f32 %7, %7, %15\n")
BODY(k_exp16, "v_exp_f16 %0, %8\n v_exp_f16 %1, %9\n v_exp_f16 %2, %10\n v_exp_f16 %3, %11\n v_exp_f16 %4, %12\n v_exp_f16 %5, %13\n v_exp_f16 %6, %14\n v_exp_f16 %7, %15\n")
BODY(k_pkmul, "v_pk_mul_f16 %0, %0, %8\n v_pk_mul_f16 %1, %1, %9\n v_pk_mul_f16 %2, %2, %10\n v_pk_mul_f16 %3, %3, %11\n v_pk_mul_f16 %4, %4, %12\n v_pk_mul_f16 %5, %5, %13\n v_pk_mul_f16 %6, %6, %14\n v_pk_mul_f16 %7, %7, %15\n")
BODY(k_cvtf16, "v_cvt_f16_f32 %0, %8\n v_cvt_f16_f32 %1, %9\n v_cvt_f16_f32 %2, %10\n v_cvt_f16_f32 %3, %11\n v_cvt_f16_f32 %4, %12\n v_cvt_f16_f32 %5, %13\n v_cvt_f16_f32 %6, %14\n v_cvt_f16_f32 %7, %15\n")
BODY(k_med3, "v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %9, %10\n v_med3_f32 %2, %2, %10, %11\n v_med3_f32 %3, %3, %11, %12\n v_med3_f32 %4, %4, %12, %13\n v_med3_f32 %5, %5, %13, %14\n v_med3_f32 %6, %6, %14, %15\n v_med3_f32 %7, %7, %15, %8\n")
BODY(k_subrev, "v_subrev_f32 %0, %8, %0\n v_subrev_f32 %1, %9, %1\n v_subrev_f32 %2, %10, %2\n v_subrev_f32 %3, %11, %3\n v_subrev_f32 %4, %12, %4\n v_subrev_f32 %5, %13, %5\n v_subrev_f32 %6, %14, %6\n v_subrev_f32 %7, %15, %7\n")
BODY(k_addlit, "v_add_f32 %0, 0x42f60000, %0\n v_add_f32 %1, 0x42f60000, %1\n v_add_f32 %2, 0x42f60000, %2\n v_add_f32 %3, 0x42f60000, %3\n v_add_f32 %4, 0x42f60000, %4\n v_add_f32 %5, 0x42f60000, %5\n v_add_f32 %6, 0x42f60000, %6\n v_add_f32 %7, 0x42f60000, %7\n")

typedef __attribute__((ext_vector_type(2))) float f32x2;
__global__ void k_pkadd(unsigned long long *out, float seed) {
  f32x2 a0 = {seed, seed + 1}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  f32x2 b = {seed * 2, seed * 3};
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < 64; ++it) {
    asm volatile(REP8("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
  f32x2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (s[0] + s[1] == 12345.678f) out[63] = 1;
}

int main() {
  unsigned long long *d;
  hipMalloc(&d, 64 * 8);
  struct { const char *name; void (*fn)(unsigned long long *, float); } ks[] = {
      {"v_add_f32", k_add}, {"v_max3_f32", k_max3}, {"v_exp_f32", k_exp}, {"v_cvt_pk_f16_f32", k_cvtpk}, {"v_cvt_f32_f16", k_cvt},
      {"v_fma_mix_f32", k_mix}, {"v_cndmask_b32", k_cnd}, {"v_pk_add_f32", k_pkadd}, {"v_sub_f32", k_sub}, {"v_mul_f32", k_mul}, {"v_fmac_f32", k_fmac}, {"v_fma_f32", k_fma}, {"v_max_f32", k_max}, {"v_and_b32", k_and}, {"v_mov_b32", k_mov}, {"v_cmp_lt_f32", k_cmp}, {"v_ldexp_f32", k_ldexp}, {"v_exp_f16", k_exp16}, {"v_pk_mul_f16", k_pkmul}, {"v_cvt_f16_f32", k_cvtf16}, {"v_med3_f32", k_med3}, {"v_subrev_f32", k_subrev}, {"v_add_f32 literal", k_addlit}};
  for (auto &k : ks)
    for (int threads : {64, 512}) {
      std::vector<unsigned long long> h(64);
      hipLaunchKernelGGL(k.fn, dim3(1), dim3(threads), 0, 0, d, 1.5f);
      hipLaunchKernelGGL(k.fn, dim3(1), dim3(threads), 0, 0, d, 1.5f);
      hipMemcpy(h.data(), d, 64 * 8, hipMemcpyDeviceToHost);
      double worst = 0;
      for (int w = 0; w < threads / 64; ++w) worst = worst > (double)h[w] ? worst : (double)h[w];
      printf("%-18s %3d threads: %.2f cycles per instruction (slowest wave)\n", k.name, threads, worst / (64.0 * 64));
    }
  return 0;
}
