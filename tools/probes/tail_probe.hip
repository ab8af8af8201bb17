// Stand-alone timing of tailbwd::tail_kernel (forward / backward) on random data, for remove-one-part experiments:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -I aline_amd/csrc -o tail_probe tools/probes/tail_probe.hip
//   ./tail_probe [rows] [fwd_grid] [bwd_grid]
#include "tail_bwd_pc.h"     // (tail_bwd.h + the archived producer / consumer experiment)
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char **argv) {
  const long M = argc > 1 ? atol(argv[1]) : 6090000;
  const int gf = argc > 2 ? atoi(argv[2]) : 768, gb = argc > 3 ? atoi(argv[3]) : 256;
  std::vector<float> h(M * 32);
  for (auto &v : h) v = (rand() % 2001 - 1000) * 1e-3f;
  float *X, *A, *dY, *Y, *dA, *dU, *W, *G;
  hipMalloc(&X, M * 128); hipMalloc(&A, M * 128); hipMalloc(&dY, M * 128); hipMalloc(&Y, M * 128); hipMalloc(&dA, M * 128); hipMalloc(&dU, M * 128);
  hipMemcpy(X, h.data(), M * 128, hipMemcpyHostToDevice); hipMemcpy(A, h.data(), M * 128, hipMemcpyHostToDevice); hipMemcpy(dY, h.data(), M * 128, hipMemcpyHostToDevice);
  const int NW = 32 * 32 + 32 + 128 * 32 + 128 + 32 * 128 + 32 + 4 * 32;
  hipMalloc(&W, NW * 4); hipMalloc(&G, NW * 4);
  hipMemcpy(W, h.data(), NW * 4, hipMemcpyHostToDevice); hipMemset(G, 0, NW * 4);
  tailbwd::Args a{};
  a.X = X; a.A = A; a.dY = dY; a.Y = Y; a.dA = dA; a.dU = dU; a.M = M;
  float *w = W, *g = G;
  a.wo = w; a.dwo = g; w += 1024; g += 1024; a.bo = w; a.dbo = g; w += 32; g += 32;
  a.w1 = w; a.dw1 = g; w += 4096; g += 4096; a.b1 = w; a.db1 = g; w += 128; g += 128;
  a.w2 = w; a.dw2 = g; w += 4096; g += 4096; a.b2 = w; a.db2 = g; w += 32; g += 32;
  a.g1 = w; a.dg1 = g; w += 32; g += 32; a.e1 = w; a.de1 = g; w += 32; g += 32;
  a.g2 = w; a.dg2 = g; w += 32; g += 32; a.e2 = w; a.de2 = g;
  hipFuncSetAttribute(reinterpret_cast<const void *>(&tailbwd::tail_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, tailbwd::LDS_FLOATS * 4);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&tailbwd::tail_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, tailbwd::LDS_FLOATS_FWD * 4);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&tailbwd::tail_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, tailbwd::LDS_FLOATS_ACC * 4);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&tailbwd::tail_bwd_pc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, tailbwd::LDS_FLOATS_PC * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pass = 0; pass < 2; ++pass) {
    float ms;
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) tailbwd::tail_kernel<false><<<gf, tailbwd::THREADS, tailbwd::LDS_FLOATS_FWD * 4>>>(a);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    if (pass) printf("forward  grid %4d: %.3f ms\n", gf, ms / 3);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) tailbwd::tail_kernel<true><<<gb, tailbwd::THREADS, tailbwd::LDS_FLOATS * 4>>>(a);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    if (pass) printf("backward grid %4d: %.3f ms\n", gb, ms / 3);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) tailbwd::tail_kernel<true, true><<<gb, 64 * tailbwd::WAVES_ACC, tailbwd::LDS_FLOATS_ACC * 4>>>(a);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    if (pass) printf("backward, LDS accumulators, grid %4d x 8 waves: %.3f ms\n", gb, ms / 3);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) tailbwd::tail_bwd_pc_kernel<<<gb, 512, tailbwd::LDS_FLOATS_PC * 4>>>(a);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    if (pass) printf("backward, producer / consumer wave pairs, grid %4d x 8 waves: %.3f ms\n", gb, ms / 3);
  }
#if defined(TAIL_STAMPS)
  {
    unsigned long long *st; hipMalloc(&st, 64); hipMemset(st, 0, 64);
    a.stamps = st;
    tailbwd::tail_kernel<true><<<gb, tailbwd::THREADS, tailbwd::LDS_FLOATS * 4>>>(a);
    unsigned long long h[8]; hipMemcpy(h, st, 64, hipMemcpyDeviceToHost);
    const char *nm[7] = {"load + u1 + LN1", "h (W1) + ReLU", "u2 (W2) + LN2", "LN2 bwd + transposes", "FFN bwd chunks (dh, dx1, dW2, dW1)", "LN1 bwd + da + stores", "transposes + dWo"};
    double tot = 0; for (int i = 0; i < 7; ++i) tot += h[i];
    for (int i = 0; i < 7; ++i) printf("  %-36s %5.1f %%  (%llu ticks)\n", nm[i], 100.0 * h[i] / tot, h[i]);
  }
#endif
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
