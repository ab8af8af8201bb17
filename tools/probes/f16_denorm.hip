// Probe: does v_mfma_f32_16x16x32_f16 keep f16 subnormal INPUTS (and does the f32 -> f16 conversion produce them)?
// x3.h / gemm.h PREC 3 rely on it for the low halves of small operands.   hipcc --offload-arch=gfx950 f16_denorm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(float a_val, float b_val, float *out) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)a_val; b[j] = (_Float16)b_val; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
}
int main() {
  float *d; hipMalloc(&d, 8);
  const float vals[] = {1e-3f, 3e-5f, 1e-5f, 1e-6f, 1e-7f, 6e-8f};
  for (float v : vals) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, v, 1.0f, d);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("a = %.3e  f16(a) = %.6e   mfma sum_k a*1 (k=32) = %.6e   expected %.6e\n", v, h[1], h[0], 32.0 * h[1]);
  }
  return 0;
}
