// Experiment archive (NOT part of libaline_hip.so since round 4): the token-local tail backward as a producer / consumer pair of
// waves per SIMD.  Round 2 measured 3.56 ms per call against 3.65 ms of tailbwd::tail_kernel<true> at the time (DESIGN.md
// section 7: the consumer's MFMAs arrive while the producer is in its own MFMA phases, not in its LayerNorm phases); it was an
// opt-in of the library until round 3 (ALINE_DBG_BWD_TAIL_PC) and is kept here for tools/probes/tail_probe.hip only.
#pragma once
#include "tail_bwd.h"

namespace tailbwd {

// ---- backward of the tail as a producer / consumer pair of waves per SIMD (an experiment, opt-in: ALINE_BWD_TAIL_PC=1) ----
// tail_kernel<true> runs one wave per SIMD (434 registers) and its vector phases (LayerNorms, ReLU gates, transposes) are
// covered by nobody's MFMAs: matrix pipe busy 60 %.  Here a workgroup is 4 PRODUCER waves (forward recompute + the dX chain
// of a tile, 288 MFMAs, no accumulators) and 4 CONSUMER waves (the 144 weight-gradient MFMAs of the tile the producer finished
// one round earlier, 36 resident accumulator tiles), 224 registers each, a pair per SIMD.  The producer leaves the twelve
// 16 x 32 operand blocks of its tile (du2 | x1 | h, dh per 32 hidden units | du1 | a) in the pair's LDS buffer in the T
// layout; the consumer reads them N-wise -- the T -> N transpose of tail_kernel, split over two waves.  Two workgroup
// barriers per round (buffer free / buffer full); every wave runs the same number of rounds, so the counts always match.
// Measured (tools/probes/tail_probe.hip, 6.09 M rows): 3.56 ms against 3.65 of tail_kernel<true>; the producer alone 3.05
// (its 288 MFMAs are 1.53 ms of matrix-pipe time), the consumer alone 1.07 (0.76): the forward + dX chain of a tile is a
// dependent sequence of short MFMA groups, LDS operand reads and vector phases that one wave cannot keep the pipe busy with,
// and the partner's 144 MFMAs fill only a fifth of the gaps.
constexpr int PC_BLOCKS = 12;                                  // 16 x 32 blocks per tile: du2, x1, h0..3, dh0..3, du1, a
constexpr int PC_BUF = PC_BLOCKS * 16 * PW;                   // floats per pair
constexpr int LDS_FLOATS_PC = L_SCR + 4 * PC_BUF;             // 151.8 KB
constexpr int B_DU2 = 0, B_X1 = 1, B_H = 2, B_DH = 6, B_DU1 = 10, B_A = 11;

__device__ __forceinline__ void put_block(float *buf, int blk, const f32x4 &v0, const f32x4 &v1, int tok, int g) {
  *reinterpret_cast<f32x4 *>(buf + blk * 16 * PW + tok * PW + 4 * g) = v0;
  *reinterpret_cast<f32x4 *>(buf + blk * 16 * PW + tok * PW + 16 + 4 * g) = v1;
}
__device__ __forceinline__ void get_block_n(f32x4 (&out)[2], const float *buf, int blk, int tok, int g) {
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[mt][r] = buf[blk * 16 * PW + (4 * g + r) * PW + 16 * mt + tok];
}

__global__ __launch_bounds__(512) void tail_bwd_pc_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tok = lane & 15, g = lane >> 4;
#if defined(TAIL_PC_ADJACENT)          // (probe variant: waves 2 p, 2 p + 1 are a pair -- they land on different SIMDs: 4.78 ms)
  const int pair = wave >> 1;
  const bool producer = (wave & 1) == 0;
#else                                  // waves 0-3 produce, 4-7 consume: wave w and w + 4 share SIMD w
  const int pair = wave & 3;
  const bool producer = wave < 4;
#endif
  for (int i = tid; i < D * D; i += 512) lds[L_WO + (i >> 5) * PW + (i & 31)] = a.wo[i];
  for (int i = tid; i < F * D; i += 512) lds[L_W1 + (i >> 5) * PW + (i & 31)] = a.w1[i];
  for (int i = tid; i < D * F; i += 512) lds[L_W2 + (i >> 7) * PW2 + (i & 127)] = a.w2[i];
  if (tid < F) lds[L_PRM + P_B1 + tid] = a.b1[tid];
  if (tid < D) {
    lds[L_PRM + P_BO + tid] = a.bo[tid]; lds[L_PRM + P_B2 + tid] = a.b2[tid];
    lds[L_PRM + P_G1 + tid] = a.g1[tid]; lds[L_PRM + P_E1 + tid] = a.e1[tid];
    lds[L_PRM + P_G2 + tid] = a.g2[tid]; lds[L_PRM + P_E2 + tid] = a.e2[tid];
  }
  __syncthreads();
  const long ntiles = (a.M + 15) / 16;
  const long tstep = (long)gridDim.x * 4;
  const int rounds = (int)((ntiles + tstep - 1) / tstep) + 1;      // the consumer trails by one round
  float *const buf = lds + L_SCR + pair * PC_BUF;

  if (producer) {
    __builtin_amdgcn_s_setprio(3);             // the producer's chain is the critical path: the consumer fills its bubbles
    f32x4 gG1[2], gE1[2], gG2[2], gE2[2];      // LayerNorm parameters, T layout (partial over rows)
#pragma unroll
    for (int i = 0; i < 2; ++i) gG1[i] = gE1[i] = gG2[i] = gE2[i] = fused::zero4();
    for (int rd = 0; rd < rounds; ++rd) {
      int zoff = 0;
      asm volatile("" : "+v"(zoff));
      const float *W = lds + zoff, *prm = W + L_PRM;
      const long tile = (long)rd * tstep + (long)blockIdx.x * 4 + pair;
      const long row = tile * 16 + tok;
#if defined(TAIL_PC_NO_PRODUCE)
      if (a.M > 0) { __syncthreads(); __syncthreads(); continue; }
#endif
      const bool ok = row < a.M;                      // (rows of the trailing round and of a ragged last tile: dy = 0)
      const long rc = ok ? row : a.M - 1;
      f32x4 x[2], at[2], dy[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        x[mt] = ld4(a.X + rc * D + 16 * mt + 4 * g);
        at[mt] = ld4(a.A + rc * D + 16 * mt + 4 * g);
        dy[mt] = ok ? ld4(a.dY + rc * D + 16 * mt + 4 * g) : fused::zero4();
      }
      // forward
      f32x4 n1[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) n1[mt] = ld4(prm + P_BO + 16 * mt + 4 * g) + x[mt];
      mm_fwd<2, 2>(n1, W + L_WO, PW, at, tok, g);
      const float rstd1 = normalise(n1);
      f32x4 x1[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) x1[mt] = n1[mt] * ld4(prm + P_G1 + 16 * mt + 4 * g) + ld4(prm + P_E1 + 16 * mt + 4 * g);
      f32x4 h[8];
#pragma unroll
      for (int ob = 0; ob < 8; ++ob) h[ob] = ld4(prm + P_B1 + 16 * ob + 4 * g);
      mm_fwd<8, 2>(h, W + L_W1, PW, x1, tok, g);
#pragma unroll
      for (int ob = 0; ob < 8; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[ob][r] = relu_nn(h[ob][r]);
      f32x4 n2[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) n2[mt] = ld4(prm + P_B2 + 16 * mt + 4 * g) + x1[mt];
      mm_fwd<2, 8>(n2, W + L_W2, PW2, h, tok, g);
      const float rstd2 = normalise(n2);
      // backward (dX chain); dh of all four chunks stays in registers until the buffer is free
      f32x4 du2[2];
      ln_backward(du2, dy, n2, rstd2, prm + P_G2, gG2, gE2, g);
      f32x4 dx1[2] = {du2[0], du2[1]};
      f32x4 dh[8];
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        f32x4 d2[2] = {fused::zero4(), fused::zero4()};
        mm_bwd<2, 2>(d2, W + L_W2 + 32 * kc, PW2, du2, tok, g);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) d2[j][r] = h[2 * kc + j][r] > 0.f ? d2[j][r] : 0.f;
        mm_bwd<2, 2>(dx1, W + L_W1 + 32 * kc * PW, PW, d2, tok, g);
        dh[2 * kc] = d2[0]; dh[2 * kc + 1] = d2[1];
      }
      f32x4 du1[2];
      ln_backward(du1, dx1, n1, rstd1, prm + P_G1, gG1, gE1, g);
      f32x4 da[2] = {fused::zero4(), fused::zero4()};
      mm_bwd<2, 2>(da, W + L_WO, PW, du1, tok, g);
      if (ok) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          *reinterpret_cast<f32x4 *>(a.dA + row * D + 16 * mt + 4 * g) = da[mt];
          *reinterpret_cast<f32x4 *>(a.dU + row * D + 16 * mt + 4 * g) = du1[mt];
        }
      }
      __syncthreads();                       // the consumer has read the previous tile: the buffer is free
      put_block(buf, B_DU2, du2[0], du2[1], tok, g);
      put_block(buf, B_X1, x1[0], x1[1], tok, g);
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        put_block(buf, B_H + kc, h[2 * kc], h[2 * kc + 1], tok, g);
        put_block(buf, B_DH + kc, dh[2 * kc], dh[2 * kc + 1], tok, g);
      }
      put_block(buf, B_DU1, du1[0], du1[1], tok, g);
      put_block(buf, B_A, at[0], at[1], tok, g);
      __syncthreads();                       // the buffer is full
    }
    // LayerNorm parameter gradients of this wave
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v[4] = {gG1[i][r], gE1[i][r], gG2[i][r], gE2[i][r]};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) v[q] += __shfl_xor(v[q], o, WAVE);      // over the 16 rows of the lane group
        if (tok == 0) {
          unsafeAtomicAdd(a.dg1 + 16 * i + 4 * g + r, v[0]);
          unsafeAtomicAdd(a.de1 + 16 * i + 4 * g + r, v[1]);
          unsafeAtomicAdd(a.dg2 + 16 * i + 4 * g + r, v[2]);
          unsafeAtomicAdd(a.de2 + 16 * i + 4 * g + r, v[3]);
        }
      }
    __syncthreads();                         // (pairs with the consumers' staging barriers below: three of them)
    __syncthreads();
    __syncthreads();
  } else {
    f32x4 gWo[2][2], gW1[8][2], gW2[2][8];
    float gBo[2] = {0.f, 0.f}, gB2[2] = {0.f, 0.f}, gB1[8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) gWo[i][j] = fused::zero4();
#pragma unroll
      for (int j = 0; j < 8; ++j) { gW1[j][i] = fused::zero4(); gW2[i][j] = fused::zero4(); }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) gB1[j] = 0.f;
    for (int rd = 0; rd < rounds; ++rd) {
#if defined(TAIL_PC_NO_CONSUME)
      if (false) {
#else
      if (rd > 0) {
#endif
        int zoff = 0;
        asm volatile("" : "+v"(zoff));
        const float *bf = buf + zoff;
        f32x4 du2N[2], x1N[2];
        get_block_n(du2N, bf, B_DU2, tok, g);
        get_block_n(x1N, bf, B_X1, tok, g);
        gB2[0] += sum4(du2N[0]); gB2[1] += sum4(du2N[1]);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          f32x4 hN[2], dhN[2];
          get_block_n(hN, bf, B_H + kc, tok, g);
          get_block_n(dhN, bf, B_DH + kc, tok, g);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            mm_dw4(gW2[0][2 * kc + j], gW2[1][2 * kc + j], gW1[2 * kc + j][0], gW1[2 * kc + j][1], du2N[0], hN[j], du2N[1], hN[j],
                   dhN[j], x1N[0], dhN[j], x1N[1]);
            gB1[2 * kc + j] += sum4(dhN[j]);
          }
        }
        f32x4 du1N[2], aN[2];
        get_block_n(du1N, bf, B_DU1, tok, g);
        get_block_n(aN, bf, B_A, tok, g);
        mm_dw4(gWo[0][0], gWo[0][1], gWo[1][0], gWo[1][1], du1N[0], aN[0], du1N[0], aN[1], du1N[1], aN[0], du1N[1], aN[1]);
        gBo[0] += sum4(du1N[0]); gBo[1] += sum4(du1N[1]);
      }
      __syncthreads();                       // this wave is done with the buffer
      __syncthreads();                       // the next tile is in the buffer
    }
    // the four consumers' gradients: LDS staging over the (now idle) weight images, then one atomic per element
    __syncthreads();
    const int t = pair * 64 + lane;      // 0..255 over the four consumer waves
    for (int i = t; i < G_TOT; i += 256) lds[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) atomicAdd(&lds[G_WO + (16 * i + 4 * g + r) * D + 16 * j + tok], gWo[i][j][r]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          atomicAdd(&lds[G_W1 + (16 * j + 4 * g + r) * D + 16 * i + tok], gW1[j][i][r]);
          atomicAdd(&lds[G_W2 + (16 * i + 4 * g + r) * F + 16 * j + tok], gW2[i][j][r]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      atomicAdd(&lds[G_PRM + P_BO + 16 * i + tok], gBo[i]);
      atomicAdd(&lds[G_PRM + P_B2 + 16 * i + tok], gB2[i]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&lds[G_PRM + P_B1 + 16 * j + tok], gB1[j]);
    __syncthreads();
    for (int i = t; i < D * D; i += 256) unsafeAtomicAdd(a.dwo + i, lds[G_WO + i]);
    for (int i = t; i < F * D; i += 256) unsafeAtomicAdd(a.dw1 + i, lds[G_W1 + i]);
    for (int i = t; i < D * F; i += 256) unsafeAtomicAdd(a.dw2 + i, lds[G_W2 + i]);
    if (t < F) unsafeAtomicAdd(a.db1 + t, lds[G_PRM + P_B1 + t]);
    if (t < D) {
      unsafeAtomicAdd(a.dbo + t, lds[G_PRM + P_BO + t]);
      unsafeAtomicAdd(a.db2 + t, lds[G_PRM + P_B2 + t]);
    }
  }
}

}  // namespace tailbwd
