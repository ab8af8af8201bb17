// Stand-alone check of the observation behind clear_words_kernel (aline_hip.hip): a hipMemsetAsync node captured into a HIP graph
// appeared to fill its 64 bytes with the kernel arguments of an eager launch enqueued right behind the replay (ROCm 7.2, MI355X;
// tools/ws_debug.py, round 3).  Here: a graph of { memset(buf, 0, 64 B); kernel } replayed N times, each replay followed by an eager
// launch whose arguments are recognisable; after every K replays the 16 words are read back and must be zero.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/graph_memset_repro tools/probes/graph_memset_repro.hip && tools/probes/graph_memset_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ void graph_work(unsigned *p) { p[64 + threadIdx.x] += 1u; }
__global__ void eager_touch(unsigned *p, unsigned long long seed, unsigned long long off, float a, float b) {
  if (seed == 1ull) p[200] = (unsigned)off + (unsigned)(a + b);      // (never true: the arguments only have to exist)
}

int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 20000, K = 50;
  unsigned *buf;
  CHECK(hipMalloc(&buf, 4096));
  CHECK(hipMemset(buf, 0, 4096));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  hipGraph_t g;
  hipGraphExec_t ge;
  CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  CHECK(hipMemsetAsync(buf, 0, 64, s));
  hipLaunchKernelGGL(graph_work, dim3(1), dim3(64), 0, s, buf);
  CHECK(hipStreamEndCapture(s, &g));
  CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  unsigned host[16];
  long bad = 0;
  for (int i = 0; i < N; ++i) {
    CHECK(hipGraphLaunch(ge, s));
    hipLaunchKernelGGL(eager_touch, dim3(1), dim3(64), 0, s, buf, 0x5eed5eed00000000ull + i, 0x0ff5e70000000000ull + 4 * i, 1.5f, 2.5f);
    if (i % K == K - 1) {
      CHECK(hipMemcpyAsync(host, buf, 64, hipMemcpyDeviceToHost, s));
      CHECK(hipStreamSynchronize(s));
      for (int w = 0; w < 16; ++w)
        if (host[w]) { if (bad < 5) printf("replay %d: word %d = 0x%08x\n", i, w, host[w]); ++bad; }
    }
  }
  CHECK(hipStreamSynchronize(s));
  printf("{\"replays\": %d, \"checks\": %d, \"nonzero_words_seen\": %ld}\n", N, N / K, bad);
  return bad ? 1 : 0;
}
