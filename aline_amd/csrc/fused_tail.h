// Token-local tail of one encoder layer for the small-width model (d = 32, F = 128), generic pipeline:
//   x1 = LN1(x + Wo a + bo);   y = LN2(x1 + W2 relu(W1 x1 + b1) + b2)          (model/encoder.py:128-141, post-norm)
// in ONE kernel: a = attention output [M, 32], x = layer input [M, 32], y = layer output.  It replaces the
// out-projection GEMM, residual + LayerNorm, both FFN GEMMs and the second residual + LayerNorm of the per-op
// pipeline (5 launches, ~2.3 KB of HBM traffic per token row) by 384 B per row; the [M, 128] hidden activations
// never exist.  Same register scheme and arithmetic as the fused rollout kernel (fused_rollout.h: transposed
// activations in the MFMA accumulator layout, exact-fp32 16x16x4 MFMAs for the out-projection, exact 3-way
// split-bf16 products for the FFN), and the same packed layer image (pack_weights_kernel) held in LDS.
#pragma once
#include "fused_rollout.h"

namespace fused {

struct TailArgs {
  const float *A, *X;     // [M, 32] attention output, layer input
  float *Y;               // [M, 32] layer output (may alias X)
  long M;
  const float *wimg;      // packed image of this layer (LAYER_FLOATS)
};

// 8 waves, 2 workgroups per CU (LDS: 2 x 67 KB) = 4 waves per SIMD at <= 128 registers
constexpr int TAIL_THREADS = 512;

__global__ __launch_bounds__(TAIL_THREADS, 4) void layer_tail_kernel(TailArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tok = lane & 15, g = lane >> 4;
  {
    // layer image -> LDS: all loads in flight before the first store
    const f32x4 *src = reinterpret_cast<const f32x4 *>(a.wimg);
    f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
    constexpr int TOT = LAYER_FLOATS / 4, BATCH = 6;
    for (int q0 = tid; q0 < TOT; q0 += BATCH * TAIL_THREADS) {
      f32x4 buf[BATCH];
#pragma unroll
      for (int i = 0; i < BATCH; ++i) { const int q = q0 + i * TAIL_THREADS; if (q < TOT) buf[i] = src[q]; }
#pragma unroll
      for (int i = 0; i < BATCH; ++i) { const int q = q0 + i * TAIL_THREADS; if (q < TOT) dst[q] = buf[i]; }
    }
  }
  __syncthreads();
  constexpr int NT = 2;
  const long ngroups = (a.M + 16 * NT - 1) / (16 * NT);        // a wave takes groups of NT token tiles
  const long gstep = (long)gridDim.x * (TAIL_THREADS / 64);
  for (long grp = (long)blockIdx.x * (TAIL_THREADS / 64) + wave; grp < ngroups; grp += gstep) {
    // the image is loop-invariant: without an opaque base the compiler hoists every weight fragment of the layer
    // into registers (16 split fragments = 192 VGPRs) and spills
    int zoff = 0;
    asm volatile("" : "+v"(zoff));
    const float *Wl = lds + zoff, *prm = Wl + PRM_BASE;
    f32x4 x[NT][2], o[NT][2];
    long row[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      row[t] = (grp * NT + t) * 16 + tok;
      const long r = min(row[t], a.M - 1);
      x[t][0] = ld4(a.X + r * D + 4 * g);
      x[t][1] = ld4(a.X + r * D + 16 + 4 * g);
      o[t][0] = ld4(a.A + r * D + 4 * g);
      o[t][1] = ld4(a.A + r * D + 16 + 4 * g);
    }
    // ---- x1 = LN1(x + Wo o + bo) ---------------------------------------------------------------------------
    f32x4 x1[NT][2];
    {
      const Frag w0 = ld_frag(Wl + FO * FRAG, lane), w1 = ld_frag(Wl + (FO + 1) * FRAG, lane);
      const f32x4 b0 = ld4(prm + PB_O + 4 * g), b1 = ld4(prm + PB_O + 16 + 4 * g);
#pragma unroll
      for (int t = 0; t < NT; ++t) { x1[t][0] = b0 + x[t][0]; x1[t][1] = b1 + x[t][1]; }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int t = 0; t < NT; ++t) MFMA4(x1[t][0], w0.lo[j], o[t][0][j]);
#pragma unroll
        for (int t = 0; t < NT; ++t) MFMA4(x1[t][1], w1.lo[j], o[t][0][j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int t = 0; t < NT; ++t) MFMA4(x1[t][0], w0.hi[j], o[t][1][j]);
#pragma unroll
        for (int t = 0; t < NT; ++t) MFMA4(x1[t][1], w1.hi[j], o[t][1][j]);
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) layer_norm(x1[t], prm + PLN1W, prm + PLN1B, g);
    // ---- y = LN2(x1 + W2 relu(W1 x1 + b1) + b2), hidden in 32-wide chunks; split-bf16 products ------------
    {
      const float *Wf = Wl + FFN_BASE;
      const f32x4 b0 = ld4(prm + PB_2 + 4 * g), b1 = ld4(prm + PB_2 + 16 + 4 * g);
      Frag3 x1f[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        x[t][0] = b0 + x1[t][0]; x[t][1] = b1 + x1[t][1];
        x1f[t] = split_acc(x1[t][0], x1[t][1]);
      }
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const Frag3 u0 = ld_frag3(Wf + (F1 + 2 * kb) * FRAG3, lane), u1 = ld_frag3(Wf + (F1 + 2 * kb + 1) * FRAG3, lane);
        const f32x4 hb0 = ld4(prm + PB_1 + 32 * kb + 4 * g), hb1 = ld4(prm + PB_1 + 32 * kb + 16 + 4 * g);
        f32x4 hd[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t) { hd[t][0] = hb0; hd[t][1] = hb1; mma6x2(hd[t][0], hd[t][1], u0, u1, x1f[t]); }
        const Frag3 d0 = ld_frag3(Wf + (F2 + kb) * FRAG3, lane), d1 = ld_frag3(Wf + (F2 + 4 + kb) * FRAG3, lane);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { hd[t][0][r] = relu_nn(hd[t][0][r]); hd[t][1][r] = relu_nn(hd[t][1][r]); }
          const Frag3 hf = split_acc(hd[t][0], hd[t][1]);
          mma6x2(x[t][0], x[t][1], d0, d1, hf);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      layer_norm(x[t], prm + PLN2W, prm + PLN2B, g);
      if (row[t] < a.M) {
        *reinterpret_cast<f32x4 *>(a.Y + row[t] * D + 4 * g) = x[t][0];
        *reinterpret_cast<f32x4 *>(a.Y + row[t] * D + 16 + 4 * g) = x[t][1];
      }
    }
  }
}

}  // namespace fused
