// Issue rate of v_mfma_f32_16x16x4_f32 on gfx950: ns and nominal cycles (2.4 GHz) per MFMA and SIMD, for NACC independent
// accumulators per wave, W waves per SIMD, on one CU and on the whole chip (power / clock effects).
//   hipcc -O3 --offload-arch=gfx950 -o mfma_f32_rate mfma_f32_rate.hip && ./mfma_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NACC>
__global__ void k(float *out, float seed, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){seed, seed, seed, seed};
  float a = seed + threadIdx.x, b = seed * 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}
template <int NACC>
void run(int blocks, int threads, const char *what) {
  float *out; hipMalloc(&out, 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<blocks, threads>>>(out, 1.f, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC><<<blocks, threads>>>(out, 1.f, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_simd = threads / 64 / 4.0;     // one workgroup per CU
  const double mfma_per_simd = (double)iters * 8 * NACC * waves_per_simd;
  const double ns = ms * 1e6 / mfma_per_simd;
  printf("%-28s NACC=%d blocks=%4d threads=%4d: %.2f ns per MFMA and SIMD = %.1f cycles at 2.4 GHz  (%.1f TFLOP/s chip-equivalent)\n",
         what, NACC, blocks, threads, ns, ns * 2.4, 2048.0 / ns * 1024 / 1000);
  hipFree(out);
}
int main() {
  run<1>(1, 256, "one CU, 1 wave/SIMD");
  run<2>(1, 256, "one CU, 1 wave/SIMD");
  run<4>(1, 256, "one CU, 1 wave/SIMD");
  run<8>(1, 256, "one CU, 1 wave/SIMD");
  run<4>(1, 512, "one CU, 2 waves/SIMD");
  run<1>(256, 256, "chip, 1 wave/SIMD");
  run<2>(256, 256, "chip, 1 wave/SIMD");
  run<4>(256, 256, "chip, 1 wave/SIMD");
  run<8>(256, 256, "chip, 1 wave/SIMD");
  run<4>(256, 512, "chip, 2 waves/SIMD");
  run<2>(256, 1024, "chip, 4 waves/SIMD");
  return 0;
}
