// tailbwd kernels on a ragged last tile: M rows against the same data padded to a multiple of 16 (extra rows: dY = 0).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -I aline_amd/csrc -o tail_ragged tools/probes/tail_ragged.hip
#include "tail_bwd.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
int main(int argc, char **argv) {
  const long M = argc > 1 ? atol(argv[1]) : 4770, MP = (M + 15) / 16 * 16;
  std::vector<float> h(MP * 32), hdy(MP * 32);
  for (auto &v : h) v = (rand() % 2001 - 1000) * 1e-3f;
  for (long i = 0; i < MP * 32; ++i) hdy[i] = i < M * 32 ? (rand() % 2001 - 1000) * 1e-3f : 0.f;
  float *X, *A, *dY, *Y[2], *dA[2], *dU[2], *W, *G[2];
  hipMalloc(&X, MP * 128); hipMalloc(&A, MP * 128); hipMalloc(&dY, MP * 128);
  hipMemcpy(X, h.data(), MP * 128, hipMemcpyHostToDevice); hipMemcpy(A, h.data(), MP * 128, hipMemcpyHostToDevice); hipMemcpy(dY, hdy.data(), MP * 128, hipMemcpyHostToDevice);
  const int NW = 32 * 32 + 32 + 128 * 32 + 128 + 32 * 128 + 32 + 4 * 32;
  hipMalloc(&W, NW * 4); hipMemcpy(W, h.data(), NW * 4, hipMemcpyHostToDevice);
  for (int k = 0; k < 2; ++k) { hipMalloc(&Y[k], MP * 128); hipMalloc(&dA[k], MP * 128); hipMalloc(&dU[k], MP * 128); hipMalloc(&G[k], NW * 4); hipMemset(G[k], 0, NW * 4); hipMemset(Y[k], 0, MP * 128); hipMemset(dA[k], 0, MP * 128); hipMemset(dU[k], 0, MP * 128); }
  hipFuncSetAttribute(reinterpret_cast<const void *>(&tailbwd::tail_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, tailbwd::LDS_FLOATS * 4);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&tailbwd::tail_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, tailbwd::LDS_FLOATS_FWD * 4);
  for (int k = 0; k < 2; ++k) {
    tailbwd::Args a{};
    a.X = X; a.A = A; a.dY = dY; a.Y = Y[k]; a.dA = dA[k]; a.dU = dU[k]; a.M = k ? MP : M;
    float *w = W, *g = G[k];
    a.wo = w; a.dwo = g; w += 1024; g += 1024; a.bo = w; a.dbo = g; w += 32; g += 32;
    a.w1 = w; a.dw1 = g; w += 4096; g += 4096; a.b1 = w; a.db1 = g; w += 128; g += 128;
    a.w2 = w; a.dw2 = g; w += 4096; g += 4096; a.b2 = w; a.db2 = g; w += 32; g += 32;
    a.g1 = w; a.dg1 = g; w += 32; g += 32; a.e1 = w; a.de1 = g; w += 32; g += 32;
    a.g2 = w; a.dg2 = g; w += 32; g += 32; a.e2 = w; a.de2 = g;
    const long groups = ((a.M + 15) / 16 + 3) / 4;
    tailbwd::tail_kernel<false><<<(unsigned)groups, tailbwd::THREADS, tailbwd::LDS_FLOATS_FWD * 4>>>(a);
    tailbwd::tail_kernel<true><<<(unsigned)std::min<long>(groups, 256), tailbwd::THREADS, tailbwd::LDS_FLOATS * 4>>>(a);
  }
  hipDeviceSynchronize();
  auto cmp = [&](const char *nm, float *p0, float *p1, long n) {
    std::vector<float> a(n), b(n);
    hipMemcpy(a.data(), p0, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), p1, n * 4, hipMemcpyDeviceToHost);
    double mx = 0, ref = 0; long at = -1;
    for (long i = 0; i < n; ++i) { const double d = fabs((double)a[i] - b[i]); if (d > mx) { mx = d; at = i; } ref = fmax(ref, fabs(b[i])); }
    printf("%-6s max |diff| = %.3e (max |ref| %.3e) at %ld (row %ld)\n", nm, mx, ref, at, at / 32);
  };
  cmp("Y", Y[0], Y[1], M * 32); cmp("dA", dA[0], dA[1], M * 32); cmp("dU", dU[0], dU[1], M * 32); cmp("grads", G[0], G[1], NW);
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
