#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../../aline_amd/csrc/common.h"
__global__ void k(const float *in, float *out) {
  const int lane = threadIdx.x;
  float v = in[lane];
  out[lane] = wave_scan_sum(v);
  out[64 + lane] = wave_sum_dpp(v);
  out[128 + lane] = wave_max_dpp(v);
}
int main() {
  float h[64], o[192], *d, *od;
  for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37) % 11) - 3.f;
  hipMalloc(&d, 256); hipMalloc(&od, 768);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, od);
  hipMemcpy(o, od, 768, hipMemcpyDeviceToHost);
  float run = 0, mx = -1e30; int bad = 0;
  for (int i = 0; i < 64; ++i) { run += h[i]; mx = fmaxf(mx, h[i]); if (o[i] != run) ++bad; }
  for (int i = 0; i < 64; ++i) { if (o[64 + i] != run) ++bad; if (o[128 + i] != mx) ++bad; }
  printf("bad %d total %g max %g\n", bad, run, mx);
  return bad != 0;
}
