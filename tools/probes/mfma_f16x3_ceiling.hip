// Probe: what does the f16 matrix pipe of THIS chip sustain, on random operands, in the instruction pattern of the reference-
// precision layer kernels (x3 / x5 / s3: every product three v_mfma_f32_16x16x32_f16 passes hi*hi + hi*lo + lo*hi)?
// The 2.5 PFLOP/s the bench line prices against is 1024 SIMDs x 1024 FLOP/clk at 2.4 GHz; under matrix load the chip lowers its
// clock (MI355X_MICROARCH.md, DVFS give-back), so the number a perfect kernel could reach is lower and device-dependent.
// Variants (all: 256 workgroups per CU-slot round, persistent for ~1 s each, random f16 operands in [-1, 1]):
//   0  operands in registers, 1 wave / SIMD (4 waves per workgroup), 16 independent accumulators
//   1  operands in registers, 2 waves / SIMD
//   2  A operand pairs re-read from LDS (2 x ds_read_b128 per 3 MFMAs: the x3 chunk loop), 2 waves / SIMD, no stream
//   3  variant 2 + the weight stream: every 48 MFMAs per wave the workgroup pulls one 32 KB chunk L2 -> LDS by
//      buffer-less global_load_lds (4 pieces of 1 KB per wave) into a ring of 4, one barrier per chunk
//      = the loop structure of x3::layer_kernel with everything but its MFMAs, fragment reads and stream removed
//   4  variant 3 with TWO token tiles per wave (a fragment pair feeds 6 MFMAs, 32 KB per 96 MFMAs) at one wave per SIMD
//      (the 512-register budget: 128 operand + 128 accumulator registers) = the structure of x5::layer_kernel
// Output: one line per variant: MFMA TFLOP/s, the fraction of 2.5 PF, /3 = the algorithmic ceiling of a 3-pass product, and the
// in-kernel clock (delta s_memtime / delta s_memrealtime x 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_f16x3_ceiling tools/probes/mfma_f16x3_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int CHUNK_BYTES = 32 * 1024, RING = 4;

struct Args {
  const u32x4 *w;         // weight image: nchunks x 32 KB of f16 fragment pairs (random)
  int nchunks;
  long iters;             // chunks per wave
  float *out;
  unsigned long long *clk;   // per workgroup: memtime delta, memrealtime delta
};

__device__ __forceinline__ f16x8 rnd_frag(unsigned s) {
  f16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s = s * 1664525u + 1013904223u;
    v[j] = (_Float16)(((int)(s >> 9) & 0xffff) * (1.f / 32768.f) - 1.f);
  }
  return v;
}

template <int VARIANT, int THREADS, int TILES>
__global__ __launch_bounds__(THREADS) void probe_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int WAVES = THREADS / 64;
  f16x8 xh[TILES][8], xl[TILES][8];
#pragma unroll
  for (int t = 0; t < TILES; ++t)
#pragma unroll
    for (int k = 0; k < 8; ++k) { xh[t][k] = rnd_frag(tid * 977u + k * 131u + blockIdx.x + 77u * t); xl[t][k] = rnd_frag(tid * 613u + k * 17u + 7u * blockIdx.x + 5u * t) * (_Float16)0.001f; }
  f32x4 y[TILES][16];
#pragma unroll
  for (int t = 0; t < TILES; ++t)
#pragma unroll
    for (int m = 0; m < 16; ++m) y[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (VARIANT >= 2) {     // fill the ring with random fragments once (variant 2 never refreshes it)
    for (int i = tid; i < RING * CHUNK_BYTES / 16; i += THREADS)
      reinterpret_cast<u32x4 *>(lds)[i] = a.w[(i + (long)blockIdx.x * 64) % ((long)a.nchunks * CHUNK_BYTES / 16)];
    __syncthreads();
  }
  f16x8 ah = rnd_frag(tid * 31u + 5u), al = rnd_frag(tid * 57u + 3u) * (_Float16)0.001f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (long it8 = 0; it8 < a.iters; it8 += 8) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {      // (static k-step: a dynamic index into xh / xl would put them in scratch)
      const long it = it8 + ks;
      const int slot = ks % RING;
      if (VARIANT >= 3) {
        // request chunk it + 3 into slot (it + 3) % RING: 32 pieces of 1 KB, WAVES waves -> 32 / WAVES pieces per wave
        const long c = (it + 3 + blockIdx.x * 7) % a.nchunks;
        const unsigned char *src = reinterpret_cast<const unsigned char *>(a.w) + c * CHUNK_BYTES;
        unsigned char *dst = lds + ((ks + 3) % RING) * CHUNK_BYTES;
#pragma unroll
        for (int p = 0; p < 32 / WAVES; ++p) {
          const int piece = wave * (32 / WAVES) + p;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + piece * 1024 + lane * 16),
                                           (__attribute__((address_space(3))) void *)(dst + piece * 1024), 16, 0, 0);
        }
      }
      const u32x4 *frag = reinterpret_cast<const u32x4 *>(lds + slot * CHUNK_BYTES) + lane;
#pragma unroll
      for (int p = 0; p < 16; ++p) {        // 16 pairs per chunk: one k-step x 16 output tiles -> 48 MFMAs
        if (VARIANT >= 2) {
          ah = __builtin_bit_cast(f16x8, frag[(2 * p) * 64]);
          al = __builtin_bit_cast(f16x8, frag[(2 * p + 1) * 64]);
        }
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
          y[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xh[t][ks], y[t][p], 0, 0, 0);
          y[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl[t][ks], y[t][p], 0, 0, 0);
          y[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh[t][ks], y[t][p], 0, 0, 0);
        }
      }
      if (VARIANT >= 3) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (32 / WAVES)) : "memory");      // chunk it + 1 has landed (two younger chunks in flight)
        __syncthreads();
      }
      asm volatile("" ::: "memory");          // (the fragment reads of the next chunk stay behind this chunk's MFMAs: no spills)
      __builtin_amdgcn_sched_barrier(0);
    }
    if ((it8 & 63) == 56) {      // keep the accumulators finite
#pragma unroll
      for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int m = 0; m < 16; ++m) y[t][m] = y[t][m] * 1e-3f;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (VARIANT >= 3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < TILES; ++t)
#pragma unroll
    for (int m = 0; m < 16; ++m) s += y[t][m][0] + y[t][m][1] + y[t][m][2] + y[t][m][3];
  a.out[(long)blockIdx.x * THREADS + tid] = s;
  if (tid == 0) { a.clk[2 * blockIdx.x] = t1 - t0; a.clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int VARIANT, int THREADS, int TILES>
static void run(const char *name, Args a, int ncu, double seconds) {
  const size_t smem = VARIANT >= 2 ? RING * CHUNK_BYTES : 0;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&probe_kernel<VARIANT, THREADS, TILES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  // calibrate: one short launch, then launches of ~0.25 s until `seconds` have passed
  a.iters = 2000;
  hipLaunchKernelGGL((probe_kernel<VARIANT, THREADS, TILES>), dim3(ncu), dim3(THREADS), smem, 0, a);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((probe_kernel<VARIANT, THREADS, TILES>), dim3(ncu), dim3(THREADS), smem, 0, a);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  a.iters = (long)(2000 * 250.0 / ms) / 8 * 8;
  double total_ms = 0, last_ms = 0;
  int launches = 0;
  while (total_ms < seconds * 1e3) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe_kernel<VARIANT, THREADS, TILES>), dim3(ncu), dim3(THREADS), smem, 0, a);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    total_ms += ms; last_ms = ms; ++launches;
  }
  std::vector<unsigned long long> clk(2 * ncu);
  CHECK(hipMemcpy(clk.data(), a.clk, clk.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> ghz(ncu);
  for (int i = 0; i < ncu; ++i) ghz[i] = (double)clk[2 * i] / (double)clk[2 * i + 1] * 0.1;
  std::sort(ghz.begin(), ghz.end());
  const double flop = (double)ncu * (THREADS / 64) * a.iters * 48.0 * TILES * 16 * 16 * 32 * 2;
  const double tf = flop / (last_ms * 1e-3) / 1e12;
  printf("{\"variant\": %d, \"name\": \"%s\", \"waves_per_simd\": %d, \"mfma_tflops\": %.1f, \"frac_of_2.5PF\": %.3f, \"algorithmic_3pass_frac\": %.3f, "
         "\"clock_ghz_median\": %.3f, \"ms\": %.1f, \"launches\": %d}\n",
         VARIANT, name, THREADS / 256, tf, tf / 2500.0, tf / 7500.0, ghz[ncu / 2], last_ms, launches);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 1.5;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  Args a{};
  a.nchunks = 80;                     // one layer image of the d = 256 model: 2.6 MB, L2-resident
  std::vector<unsigned short> h((size_t)a.nchunks * CHUNK_BYTES / 2);
  unsigned s = 12345u;
  for (auto &v : h) {                 // random f16, |v| in [2^-11, 2): sign, biased exponent 4 .. 15, random mantissa
    s = s * 1664525u + 1013904223u;
    v = (unsigned short)(((s >> 16) & 0x8000u) | ((4u + ((s >> 20) % 12u)) << 10) | ((s >> 6) & 0x3ffu));
  }
  u32x4 *w;
  CHECK(hipMalloc(&w, h.size() * 2));
  CHECK(hipMemcpy(w, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  a.w = w;
  CHECK(hipMalloc(&a.out, (size_t)ncu * 512 * 4));
  CHECK(hipMalloc(&a.clk, (size_t)ncu * 16));
  printf("# %s, %d CUs; peak priced at 2500 TFLOP/s (f16 / bf16 dense)\n", prop.name, ncu);
  run<0, 256, 1>("registers", a, ncu, seconds);
  run<1, 512, 1>("registers", a, ncu, seconds);
  run<2, 512, 1>("A pairs from LDS", a, ncu, seconds);
  run<3, 512, 1>("A pairs from LDS + 32 KB / 48 MFMA weight stream (x3 structure)", a, ncu, seconds);
  run<4, 256, 2>("two tiles per wave: A pairs from LDS + 32 KB / 96 MFMA weight stream (x5 structure)", a, ncu, seconds);
  return 0;
}
