// Token-local tail of one encoder layer for the small-width model (d = 32, F = 128) in the TRAINING backward:
//   u1 = x + Wo a + bo;  x1 = LN1(u1);  h = relu(W1 x1 + b1);  u2 = x1 + W2 h + b2;  y = LN2(u2)
// (model/encoder.py:128-141, post-norm; `loss.backward()` of train_aline.py:124-132 through it).
//
// tail_kernel<false>: forward recompute, y from (x, a) -- exact fp32, [M, 128] hidden activations never exist.
// tail_kernel<true>:  forward recompute AND backward of the same 16-row tile in the registers of one wave:
//   in  x, a, dy = dLoss/dy                       (384 B per token row)
//   out da = dLoss/da, du1 = dLoss/du1            (256 B per token row; du1 is also the residual branch into x)
//   and the parameter gradients dWo, dbo, dW1, db1, dW2, db2, LN1 / LN2 weight and bias.
// It replaces, per layer, 5 forward launches (out-projection GEMM, add + LN, two FFN GEMMs, add + LN) that saved
// U1 / X1 / Hid / U2 (3.4 KB per token row) and 8 backward launches (two LN backward, three dW and three dX GEMMs)
// that re-read them: ~7 KB of HBM traffic per token row and layer become 640 B.
//
// Register scheme.  Two layouts of a 16-row x 16-feature block in the 4 registers of a lane (tok = lane & 15, g = lane >> 4):
//   T layout (token on lane): reg r of lane (tok, g) = V[row tok][feature 4 g + r]  -- the MFMA accumulator of Y^T = W X^T and,
//     register for register, the B operand of the next product (fused_rollout.h): all forward products and all dX products
//     (dX^T = W^T dY^T) chain in it; the A operand is a weight fragment read from a padded row-major LDS image.
//   N layout (feature on lane): reg r of lane (f, g) = V[row 4 g + r][feature f]  -- what a weight gradient needs:
//     dW[i][j] = sum_rows dY[row][i] X[row][j] is four 16x16x4 MFMAs with A = dY_N[r], B = X_N[r] (k = row).
//   T -> N is a 16 x 32 transpose through a private LDS scratch of the wave (one ds_write_b128 pair, eight ds_read_b32).
// The 36 dW accumulator tiles (144 registers) of a wave stay resident over all its tiles and are added to global memory once.
// All products are exact fp32 (v_mfma_f32_16x16x4_f32): the acquisition-head gradients are a small difference of large
// REINFORCE terms and do not survive a split-bf16 forward.
#pragma once
#include "fused_rollout.h"

namespace tailbwd {

constexpr int D = 32, F = 128;
#if !defined(TAIL_PW)
#define TAIL_PW 36
#define TAIL_PW2 132
#endif
constexpr int PW = TAIL_PW, PW2 = TAIL_PW2;                      // LDS row pitches (floats): conflict-free b128 row reads and column reads
constexpr int L_WO = 0, L_W1 = L_WO + D * PW, L_W2 = L_W1 + F * PW, L_PRM = L_W2 + D * PW2;
constexpr int P_BO = 0, P_B1 = 32, P_B2 = 160, P_G1 = 192, P_E1 = 224, P_G2 = 256, P_E2 = 288, NPRM = 320;
constexpr int L_SCR = L_PRM + NPRM;
constexpr int SCR = 2 * 16 * PW;       // two 16 x 32 blocks per wave (backward only)
constexpr int WAVES = 4, THREADS = 64 * WAVES;
// backward only: transposed copies of the three images behind the scratch, so that the dX products read their A fragments
// like the forward ones (one ds_read_b128 per four MFMAs instead of one ds_read_b32 per MFMA)
constexpr int L_T = L_SCR + WAVES * SCR, L_WOT = L_T, L_W1T = L_WOT + D * PW, L_W2T = L_W1T + D * PW2, L_TEND = L_W2T + F * PW;
constexpr int LDS_FLOATS_FWD = L_SCR, LDS_FLOATS = L_TEND;       // 41.2 KB forward, 99.5 KB backward
// backward with the weight-gradient accumulators in LDS (8 waves of <= 256 registers, two per SIMD): image | 8 scratches | sums
constexpr int WAVES_ACC = 8, L_ACC = L_SCR + WAVES_ACC * SCR;
// gradient staging, reusing the image region after the tile loop
constexpr int G_WO = 0, G_W1 = G_WO + D * D, G_W2 = G_W1 + F * D, G_PRM = G_W2 + D * F, G_TOT = G_PRM + NPRM;
static_assert(G_TOT <= L_SCR, "gradient staging must fit below the scratch");
constexpr int LDS_FLOATS_ACC = L_ACC + G_TOT;      // 116 KB

struct Args {
  const float *X, *A;        // [M, 32] layer input, attention output
  const float *dY;           // [M, 32] gradient wrt the layer output                (backward)
  float *Y;                  // [M, 32] layer output                                 (forward)
  float *dA, *dU;            // [M, 32] gradient wrt a, gradient wrt u1              (backward)
  long M;
  const float *wo, *bo, *w1, *b1, *w2, *b2, *g1, *e1, *g2, *e2;
  float *dwo, *dbo, *dw1, *db1, *dw2, *db2, *dg1, *de1, *dg2, *de2;     // accumulated into (+=)
  unsigned *da_absmax;          // tail16_kernel, optional: max |dA| written, as bits (the gradient scale of attn_block_bwd16_kernel)
  const unsigned *dy_max_bits;  // tail16_kernel: bits of max |dY| (absmax_bits_kernel): the power-of-two scale of every gradient in the tile program
#if defined(TAIL_STAMPS)      // (tools/probes/tail_probe.hip: s_memtime deltas per phase, wave 0 of workgroup 0)
  unsigned long long *stamps;
#endif
};
#if defined(TAIL_STAMPS)
#define TAIL_NSTAMP 8
struct Stamps {
  unsigned long long t_prev, acc[TAIL_NSTAMP];
  __device__ __forceinline__ void start() { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory"); }
  __device__ __forceinline__ void lap(int k) {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    acc[k] += t - t_prev;
    t_prev = t;
  }
};
#define TAIL_LAP(k, ...) do { asm volatile("" :: __VA_ARGS__); st.lap(k); } while (0)
#else
#define TAIL_LAP(k, ...) do { } while (0)
#endif

using fused::ld4;
using fused::group_sum;

// An MFMA whose place among the other MFMAs is the one written here: hipcc's scheduler otherwise groups the MFMAs of one
// accumulator back to back (a dependent v_mfma_f32_16x16x4_f32 issues every ~50 cycles instead of 32; measured 52 cycles
// per MFMA in tail_kernel<false>).  Everything that is not an MFMA may still move across (mask 0x7F6).
#define MFMAO(acc, a, b)                                              \
  do {                                                                \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);   \
    __builtin_amdgcn_sched_barrier(0x7F6);                            \
  } while (0)

// acc[ob] += W[16 ob + tok][16 kb + 4 g + r] * in[kb][r]   (Y^T = W X^T; W row-major [out][in] in LDS)
template <int NOB, int NKB>
__device__ __forceinline__ void mm_fwd(f32x4 (&acc)[NOB], const float *W, int pitch, const f32x4 (&in)[NKB], int tok, int g) {
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    f32x4 w[NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) w[ob] = ld4(W + (16 * ob + tok) * pitch + 16 * kb + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) MFMAO(acc[ob], w[ob][r], in[kb][r]);
  }
}
// acc[ib] += W[16 kb + 4 g + r][16 ib + tok] * dy[kb][r]    (dX^T = W^T dY^T; same image, column reads)
template <int NIB, int NKB>
__device__ __forceinline__ void mm_bwd(f32x4 (&acc)[NIB], const float *W, int pitch, const f32x4 (&dy)[NKB], int tok, int g) {
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float w[NIB];
#pragma unroll
      for (int ib = 0; ib < NIB; ++ib) w[ib] = W[(16 * kb + 4 * g + r) * pitch + 16 * ib + tok];
#pragma unroll
      for (int ib = 0; ib < NIB; ++ib) MFMAO(acc[ib], w[ib], dy[kb][r]);
    }
}
// dW tile += sum over the 16 rows of the tile  a_N[.][i] * b_N[.][j]
__device__ __forceinline__ void mm_dw(f32x4 &acc, const f32x4 &aN, const f32x4 &bN) {
#pragma unroll
  for (int r = 0; r < 4; ++r) MFMAO(acc, aN[r], bN[r]);
}
// T layout -> N layout of a 16 x 32 block through the wave's scratch (DS operations of a wave execute in order)
__device__ __forceinline__ void to_n(f32x4 (&out)[2], const f32x4 &in0, const f32x4 &in1, float *scr, int tok, int g) {
  *reinterpret_cast<f32x4 *>(scr + tok * PW + 4 * g) = in0;
  *reinterpret_cast<f32x4 *>(scr + tok * PW + 16 + 4 * g) = in1;
  asm volatile("" ::: "memory");
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[mt][r] = scr[(4 * g + r) * PW + 16 * mt + tok];
  asm volatile("" ::: "memory");
}
// two blocks in one LDS round trip (the wave's scratch holds two 16 x 32 blocks)
__device__ __forceinline__ void to_n2(f32x4 (&o0)[2], f32x4 (&o1)[2], const f32x4 &a0, const f32x4 &a1, const f32x4 &b0,
                                      const f32x4 &b1, float *scr, int tok, int g) {
  *reinterpret_cast<f32x4 *>(scr + tok * PW + 4 * g) = a0;
  *reinterpret_cast<f32x4 *>(scr + tok * PW + 16 + 4 * g) = a1;
  *reinterpret_cast<f32x4 *>(scr + 16 * PW + tok * PW + 4 * g) = b0;
  *reinterpret_cast<f32x4 *>(scr + 16 * PW + tok * PW + 16 + 4 * g) = b1;
  asm volatile("" ::: "memory");
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o0[mt][r] = scr[(4 * g + r) * PW + 16 * mt + tok];
      o1[mt][r] = scr[16 * PW + (4 * g + r) * PW + 16 * mt + tok];
    }
  asm volatile("" ::: "memory");
}
// four weight-gradient tiles at once, MFMA by MFMA over independent accumulators (a dependent 16x16x4 MFMA waits 40 cycles)
__device__ __forceinline__ void mm_dw4(f32x4 &c0, f32x4 &c1, f32x4 &c2, f32x4 &c3, const f32x4 &a0, const f32x4 &b0,
                                       const f32x4 &a1, const f32x4 &b1, const f32x4 &a2, const f32x4 &b2, const f32x4 &a3,
                                       const f32x4 &b3) {
#pragma unroll
  for (int r = 0; r < 4; ++r) { MFMAO(c0, a0[r], b0[r]); MFMAO(c1, a1[r], b1[r]); MFMAO(c2, a2[r], b2[r]); MFMAO(c3, a3[r], b3[r]); }
}
// u <- (u - mean) * rstd over the 32 features of each row (eps 1e-5, biased variance); returns rstd
__device__ __forceinline__ float normalise(f32x4 (&u)[2]) {
  float s = 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) s += u[mt][r];
  const float mean = group_sum(s) * (1.f / D);
  float ss = 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { u[mt][r] -= mean; ss = fmaf(u[mt][r], u[mt][r], ss); }
  const float rstd = rsqrtf(group_sum(ss) * (1.f / D) + 1e-5f);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) u[mt][r] *= rstd;
  return rstd;
}
// LayerNorm backward: dy (gradient wrt gamma n + beta) -> du; the parameter gradients accumulate per lane (T layout)
__device__ __forceinline__ void ln_backward(f32x4 (&du)[2], const f32x4 (&dy)[2], const f32x4 (&n)[2], float rstd,
                                            const float *gamma, f32x4 (&dgam)[2], f32x4 (&dbet)[2], int g) {
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const f32x4 gv = ld4(gamma + 16 * mt + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dgam[mt][r] = fmaf(dy[mt][r], n[mt][r], dgam[mt][r]);
      dbet[mt][r] += dy[mt][r];
      const float dn = dy[mt][r] * gv[r];
      du[mt][r] = dn;
      s1 += dn;
      s2 = fmaf(dn, n[mt][r], s2);
    }
  }
  const float m1 = group_sum(s1) * (1.f / D), m2 = group_sum(s2) * (1.f / D);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) du[mt][r] = (du[mt][r] - m1 - n[mt][r] * m2) * rstd;
}
__device__ __forceinline__ float sum4(const f32x4 &v) { return (v[0] + v[1]) + (v[2] + v[3]); }
// tile (rb, cb) of a row-major [.., ld] LDS matrix += an MFMA accumulator tile ([16 rb + 4 g + r][16 cb + tok])
__device__ __forceinline__ void lds_add_tile(float *base, int ld, int rb, int cb, const f32x4 &t, int tok, int g) {
#pragma unroll
  for (int r = 0; r < 4; ++r) atomicAdd(&base[(16 * rb + 4 * g + r) * ld + 16 * cb + tok], t[r]);
}

// LDSACC (backward only; an experiment kept for tools/probes/tail_probe.hip, NOT used by the library): the 36 weight-gradient
// tiles are not kept in registers (144 of them: 434 registers, one wave per SIMD, ~470 VGPR <-> AGPR moves per tile) but added
// to a workgroup-wide LDS accumulator after every group of products, so that a wave fits 256 registers and two waves share a
// SIMD.  Measured: 18.6 ms instead of 3.7 -- ds_add_f32 retires a few lanes per cycle, 9 216 lane-adds per tile swamp the LDS.
template <bool BWD, bool LDSACC = false>
__global__ __launch_bounds__(LDSACC ? 64 * WAVES_ACC : THREADS) void tail_kernel(Args a) {
  static_assert(BWD || !LDSACC, "LDS accumulation is a backward option");
  constexpr int WAVES = LDSACC ? WAVES_ACC : tailbwd::WAVES, THREADS = 64 * WAVES;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tok = lane & 15, g = lane >> 4;
  if (LDSACC) for (int i = tid; i < G_TOT; i += THREADS) lds[L_ACC + i] = 0.f;
  for (int i = tid; i < D * D; i += THREADS) lds[L_WO + (i >> 5) * PW + (i & 31)] = a.wo[i];
  for (int i = tid; i < F * D; i += THREADS) lds[L_W1 + (i >> 5) * PW + (i & 31)] = a.w1[i];
  for (int i = tid; i < D * F; i += THREADS) lds[L_W2 + (i >> 7) * PW2 + (i & 127)] = a.w2[i];
  if (tid < F) lds[L_PRM + P_B1 + tid] = a.b1[tid];
  if (tid < D) {
    lds[L_PRM + P_BO + tid] = a.bo[tid]; lds[L_PRM + P_B2 + tid] = a.b2[tid];
    lds[L_PRM + P_G1 + tid] = a.g1[tid]; lds[L_PRM + P_E1 + tid] = a.e1[tid];
    lds[L_PRM + P_G2 + tid] = a.g2[tid]; lds[L_PRM + P_E2 + tid] = a.e2[tid];
  }
  constexpr bool TIMG = BWD && !LDSACC;      // Wo^T [32][36], W1^T [32][132], W2^T [128][36]
  if (TIMG) {
    for (int i = tid; i < D * D; i += THREADS) lds[L_WOT + (i & 31) * PW + (i >> 5)] = a.wo[i];
    for (int i = tid; i < F * D; i += THREADS) lds[L_W1T + (i & 31) * PW2 + (i >> 5)] = a.w1[i];
    for (int i = tid; i < D * F; i += THREADS) lds[L_W2T + (i & 127) * PW + (i >> 7)] = a.w2[i];
  }
  __syncthreads();

  // weight-gradient accumulators of this wave (MFMA accumulator layout: [16 ib + 4 g + r][16 jb + tok])
  f32x4 gWo[2][2], gW1[8][2], gW2[2][8];
  f32x4 gG1[2], gE1[2], gG2[2], gE2[2];      // LayerNorm parameters, T layout (feature 16 mt + 4 g + r, partial over rows)
  float gBo[2], gB1[8], gB2[2];              // biases, N layout (feature 16 blk + tok, partial over lane groups)
  if (BWD) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) gWo[i][j] = fused::zero4();
#pragma unroll
      for (int j = 0; j < 8; ++j) { gW1[j][i] = fused::zero4(); gW2[i][j] = fused::zero4(); }
      gG1[i] = gE1[i] = gG2[i] = gE2[i] = fused::zero4();
      gBo[i] = gB2[i] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) gB1[j] = 0.f;
  }

#if defined(TAIL_STAMPS)
  Stamps st{};
  st.start();
#endif
  const long ntiles = (a.M + 15) / 16;
  const long tstep = (long)gridDim.x * WAVES;
  // the tile after this one is loaded while this one is computed (one wave per SIMD: nobody else hides the latency)
  f32x4 nx[2], na[2], ndy[2];
  auto load_tile = [&](long tile) {
    const long r = min(tile * 16 + tok, a.M - 1);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      nx[mt] = ld4(a.X + r * D + 16 * mt + 4 * g);
      na[mt] = ld4(a.A + r * D + 16 * mt + 4 * g);
      if (BWD) ndy[mt] = ld4(a.dY + r * D + 16 * mt + 4 * g);
    }
  };
  constexpr bool PREF = !LDSACC;      // two waves per SIMD hide the load latency themselves, and the 24 registers are spills there
  if (PREF && (long)blockIdx.x * WAVES + wave < ntiles) load_tile((long)blockIdx.x * WAVES + wave);
  for (long tile = (long)blockIdx.x * WAVES + wave; tile < ntiles; tile += tstep) {
    // opaque base: the images are loop-invariant, and hoisted weight fragments would take every register
    int zoff = 0;
    asm volatile("" : "+v"(zoff));
    const float *W = lds + zoff, *prm = W + L_PRM;
    float *scr = lds + zoff + L_SCR + wave * SCR;
    const long row = tile * 16 + tok;
    const bool ok = row < a.M;
    f32x4 x[2], at[2], dy[2];
    if (!PREF) load_tile(tile);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      x[mt] = nx[mt];
      at[mt] = na[mt];
      if (BWD) dy[mt] = ok ? ndy[mt] : fused::zero4();
    }
    if (PREF && tile + tstep < ntiles) load_tile(tile + tstep);
    // ---- forward ---------------------------------------------------------------------------------------------
    f32x4 n1[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) n1[mt] = ld4(prm + P_BO + 16 * mt + 4 * g) + x[mt];
    mm_fwd<2, 2>(n1, W + L_WO, PW, at, tok, g);
    const float rstd1 = normalise(n1);
    TAIL_LAP(0, "v"(n1[0]), "v"(n1[1]));
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
    TAIL_LAP(1, "v"(h[0]), "v"(h[7]));
    f32x4 n2[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) n2[mt] = ld4(prm + P_B2 + 16 * mt + 4 * g) + x1[mt];
    mm_fwd<2, 8>(n2, W + L_W2, PW2, h, tok, g);
    const float rstd2 = normalise(n2);
    TAIL_LAP(2, "v"(n2[0]), "v"(n2[1]));
    if (!BWD) {
      if (ok) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          *reinterpret_cast<f32x4 *>(a.Y + row * D + 16 * mt + 4 * g) =
              n2[mt] * ld4(prm + P_G2 + 16 * mt + 4 * g) + ld4(prm + P_E2 + 16 * mt + 4 * g);
      }
      continue;
    }
    // ---- backward --------------------------------------------------------------------------------------------
    f32x4 du2[2];
    ln_backward(du2, dy, n2, rstd2, prm + P_G2, gG2, gE2, g);
    f32x4 du2N[2], x1N[2];
    to_n2(du2N, x1N, du2[0], du2[1], x1[0], x1[1], scr, tok, g);
    gB2[0] += sum4(du2N[0]); gB2[1] += sum4(du2N[1]);
    TAIL_LAP(3, "v"(du2N[0]), "v"(x1N[1]));
    f32x4 dx1[2] = {du2[0], du2[1]};
    // 32 hidden units at a time, software-pipelined so that the MFMAs of independent products follow each other:
    //   dx1 += W1^T dh(kc)  |  dh(kc + 1) = W2^T du2  |  dW2, dW1 of chunk kc  -- the ReLU gate of chunk kc + 1 and the LDS
    // transposes of chunk kc are vector / LDS work beside them (written one product after the other, every 16-MFMA group
    // drained the pipe before the next could start)
    auto dh_chunk = [&](f32x4 (&dh)[2], int kc) {
      dh[0] = dh[1] = fused::zero4();
      if (TIMG) mm_fwd<2, 2>(dh, W + L_W2T + 32 * kc * PW, PW, du2, tok, g);
      else mm_bwd<2, 2>(dh, W + L_W2 + 32 * kc, PW2, du2, tok, g);
    };
    f32x4 dhc[2];
    dh_chunk(dhc, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) dhc[j][r] = h[j][r] > 0.f ? dhc[j][r] : 0.f;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      f32x4 hN[2], dhN[2];
      to_n2(hN, dhN, h[2 * kc], h[2 * kc + 1], dhc[0], dhc[1], scr, tok, g);
      if (TIMG) mm_fwd<2, 2>(dx1, W + L_W1T + 32 * kc, PW2, dhc, tok, g);
      else mm_bwd<2, 2>(dx1, W + L_W1 + 32 * kc * PW, PW, dhc, tok, g);
      f32x4 dhn[2];
      if (kc < 3) dh_chunk(dhn, kc + 1);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (LDSACC) {
          f32x4 t0 = fused::zero4(), t1 = fused::zero4(), t2 = fused::zero4(), t3 = fused::zero4();
          mm_dw4(t0, t1, t2, t3, du2N[0], hN[j], du2N[1], hN[j], dhN[j], x1N[0], dhN[j], x1N[1]);
          float *acc = lds + zoff + L_ACC;
          lds_add_tile(acc + G_W2, F, 0, 2 * kc + j, t0, tok, g);
          lds_add_tile(acc + G_W2, F, 1, 2 * kc + j, t1, tok, g);
          lds_add_tile(acc + G_W1, D, 2 * kc + j, 0, t2, tok, g);
          lds_add_tile(acc + G_W1, D, 2 * kc + j, 1, t3, tok, g);
        } else {
          mm_dw4(gW2[0][2 * kc + j], gW2[1][2 * kc + j], gW1[2 * kc + j][0], gW1[2 * kc + j][1], du2N[0], hN[j], du2N[1], hN[j],
                 dhN[j], x1N[0], dhN[j], x1N[1]);
        }
        gB1[2 * kc + j] += sum4(dhN[j]);
      }
      if (kc < 3) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) dhc[j][r] = h[2 * kc + 2 + j][r] > 0.f ? dhn[j][r] : 0.f;
      }
    }
    TAIL_LAP(4, "v"(dx1[0]), "v"(dx1[1]));
    f32x4 du1[2];
    ln_backward(du1, dx1, n1, rstd1, prm + P_G1, gG1, gE1, g);
    f32x4 da[2] = {fused::zero4(), fused::zero4()};
    if (TIMG) mm_fwd<2, 2>(da, W + L_WOT, PW, du1, tok, g);
    else mm_bwd<2, 2>(da, W + L_WO, PW, du1, tok, g);
    if (ok) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        *reinterpret_cast<f32x4 *>(a.dA + row * D + 16 * mt + 4 * g) = da[mt];
        *reinterpret_cast<f32x4 *>(a.dU + row * D + 16 * mt + 4 * g) = du1[mt];
      }
    }
    TAIL_LAP(5, "v"(da[0]), "v"(da[1]));
    f32x4 du1N[2], aN[2];
    to_n2(du1N, aN, du1[0], du1[1], at[0], at[1], scr, tok, g);
    if (LDSACC) {
      f32x4 t0 = fused::zero4(), t1 = fused::zero4(), t2 = fused::zero4(), t3 = fused::zero4();
      mm_dw4(t0, t1, t2, t3, du1N[0], aN[0], du1N[0], aN[1], du1N[1], aN[0], du1N[1], aN[1]);
      float *acc = lds + zoff + L_ACC;
      lds_add_tile(acc + G_WO, D, 0, 0, t0, tok, g);
      lds_add_tile(acc + G_WO, D, 0, 1, t1, tok, g);
      lds_add_tile(acc + G_WO, D, 1, 0, t2, tok, g);
      lds_add_tile(acc + G_WO, D, 1, 1, t3, tok, g);
    } else {
      mm_dw4(gWo[0][0], gWo[0][1], gWo[1][0], gWo[1][1], du1N[0], aN[0], du1N[0], aN[1], du1N[1], aN[0], du1N[1], aN[1]);
    }
    gBo[0] += sum4(du1N[0]); gBo[1] += sum4(du1N[1]);
    TAIL_LAP(6, "v"(gWo[0][0]), "v"(gWo[1][1]));
  }
#if defined(TAIL_STAMPS)
  if (BWD && a.stamps && blockIdx.x == 0 && tid == 0) for (int q = 0; q < TAIL_NSTAMP; ++q) a.stamps[q] = st.acc[q];
#endif
  if (!BWD) return;

  // ---- the workgroup's gradients: LDS staging, then one atomic per element ------------------------------------
  float *const stg = lds + (LDSACC ? L_ACC : 0);
  __syncthreads();
  if (!LDSACC) {
    for (int i = tid; i < G_TOT; i += THREADS) stg[i] = 0.f;
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (!LDSACC) {
#pragma unroll
        for (int j = 0; j < 2; ++j) atomicAdd(&stg[G_WO + (16 * i + 4 * g + r) * D + 16 * j + tok], gWo[i][j][r]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          atomicAdd(&stg[G_W1 + (16 * j + 4 * g + r) * D + 16 * i + tok], gW1[j][i][r]);
          atomicAdd(&stg[G_W2 + (16 * i + 4 * g + r) * F + 16 * j + tok], gW2[i][j][r]);
        }
      }
      atomicAdd(&stg[G_PRM + P_G1 + 16 * i + 4 * g + r], gG1[i][r]);
      atomicAdd(&stg[G_PRM + P_E1 + 16 * i + 4 * g + r], gE1[i][r]);
      atomicAdd(&stg[G_PRM + P_G2 + 16 * i + 4 * g + r], gG2[i][r]);
      atomicAdd(&stg[G_PRM + P_E2 + 16 * i + 4 * g + r], gE2[i][r]);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    atomicAdd(&stg[G_PRM + P_BO + 16 * i + tok], gBo[i]);
    atomicAdd(&stg[G_PRM + P_B2 + 16 * i + tok], gB2[i]);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) atomicAdd(&stg[G_PRM + P_B1 + 16 * j + tok], gB1[j]);
  __syncthreads();
  for (int i = tid; i < D * D; i += THREADS) unsafeAtomicAdd(a.dwo + i, stg[G_WO + i]);
  for (int i = tid; i < F * D; i += THREADS) unsafeAtomicAdd(a.dw1 + i, stg[G_W1 + i]);
  for (int i = tid; i < D * F; i += THREADS) unsafeAtomicAdd(a.dw2 + i, stg[G_W2 + i]);
  if (tid < F) unsafeAtomicAdd(a.db1 + tid, stg[G_PRM + P_B1 + tid]);
  if (tid < D) {
    unsafeAtomicAdd(a.dbo + tid, stg[G_PRM + P_BO + tid]);
    unsafeAtomicAdd(a.db2 + tid, stg[G_PRM + P_B2 + tid]);
    unsafeAtomicAdd(a.dg1 + tid, stg[G_PRM + P_G1 + tid]);
    unsafeAtomicAdd(a.de1 + tid, stg[G_PRM + P_E1 + tid]);
    unsafeAtomicAdd(a.dg2 + tid, stg[G_PRM + P_G2 + tid]);
    unsafeAtomicAdd(a.de2 + tid, stg[G_PRM + P_E2 + tid]);
  }
}



// ---- round 4: the backward tile program on the f16 matrix pipe (tail16_kernel) -------------------------------------------------------
// The same program, layouts and accumulators as tail_kernel<true>; every group of four v_mfma_f32_16x16x4_f32 over one 16 x 16 operand
// block (k = 16: 128 pipe cycles) becomes the 3-term f16 split of the forward kernels on v_mfma_f32_16x16x16_f16 (hi * hi + hi * lo +
// lo * hi): the T / N blocks of the fp32 program ARE the k = 4 g .. 4 g + 3 operand slices of that instruction, register for register.
//   * weights: the LDS images hold, in the 16 bytes of four consecutive k, the f16 hi halves and the f16 lo halves of W * 2^8 (one
//     ds_read_b128 per fragment as before; 2^8 keeps the lo halves of small weights normal; the accumulators are divided where they are read);
//   * activations and gradients are split where they are produced (8 vector instructions per block, reused by every output tile);
//   * the backward is LINEAR in dY: dY is multiplied once, at the load, by the power of two that puts max |dY| into [2^4, 2^5) (a
//     reduction pass over dY in front of the launch: Args.dy_max_bits) -- f16 holds such values with 22 bits down to 2^-14 of the maximum
//     and leaves 2^11 of head room for the tile program's own growth (LayerNorm backward: rstd * gamma per stage) -- and every result
//     (dA, dU1, the 36 dW tiles, bias and LayerNorm gradients) is divided by it at the end.  An overflow would make the results
//     non-finite, never silently wrong; the training loop refuses a non-finite gradient norm.
// 3.4 ms -> see DESIGN.md section 7 for the measured figure; tail_kernel<true> stays as the exact-fp32 reference (ALINE_DBG_BWD_GRAD_F32).
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
struct H8 { h4 hi, lo; };
__device__ __forceinline__ H8 split4(const f32x4 &v) {
  H8 o;
  o.hi = __builtin_convertvector(v, h4);
  o.lo = __builtin_convertvector(v - __builtin_convertvector(o.hi, f32x4), h4);
  return o;
}
__device__ __forceinline__ H8 as_h8(const f32x4 &raw) {      // an LDS weight fragment: words 0, 1 = hi halves, 2, 3 = lo halves
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const u2 a = {__float_as_uint(raw[0]), __float_as_uint(raw[1])}, b = {__float_as_uint(raw[2]), __float_as_uint(raw[3])};
  H8 o;
  o.hi = __builtin_bit_cast(h4, a);
  o.lo = __builtin_bit_cast(h4, b);
  return o;
}
#define MFMA16O(acc, a, b)                                               \
  do {                                                                   \
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, acc, 0, 0, 0);     \
    __builtin_amdgcn_sched_barrier(0x7F6);                               \
  } while (0)
constexpr float WSC = 256.f, WINV16 = 1.f / 256.f;
// acc[ob] += 2^8 W[16 ob + tok][16 kb + 4 g + r] * in[kb][r]
template <int NOB, int NKB>
__device__ __forceinline__ void mm_fwd16(f32x4 (&acc)[NOB], const float *W, int pitch, const H8 (&in)[NKB], int tok, int g) {
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    H8 w[NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) w[ob] = as_h8(ld4(W + (16 * ob + tok) * pitch + 16 * kb + 4 * g));
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) MFMA16O(acc[ob], w[ob].lo, in[kb].hi);
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) MFMA16O(acc[ob], w[ob].hi, in[kb].lo);
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) MFMA16O(acc[ob], w[ob].hi, in[kb].hi);
  }
}
__device__ __forceinline__ void mm_dw4_16(f32x4 &c0, f32x4 &c1, f32x4 &c2, f32x4 &c3, const H8 &a0, const H8 &b0, const H8 &a1,
                                          const H8 &b1, const H8 &a2, const H8 &b2, const H8 &a3, const H8 &b3) {
  MFMA16O(c0, a0.lo, b0.hi); MFMA16O(c1, a1.lo, b1.hi); MFMA16O(c2, a2.lo, b2.hi); MFMA16O(c3, a3.lo, b3.hi);
  MFMA16O(c0, a0.hi, b0.lo); MFMA16O(c1, a1.hi, b1.lo); MFMA16O(c2, a2.hi, b2.lo); MFMA16O(c3, a3.hi, b3.lo);
  MFMA16O(c0, a0.hi, b0.hi); MFMA16O(c1, a1.hi, b1.hi); MFMA16O(c2, a2.hi, b2.hi); MFMA16O(c3, a3.hi, b3.hi);
}
// image of a [rows][cols] matrix M(r, c) = src[r * sr + c * sc]: 16 bytes per four consecutive c = (hi x 4 | lo x 4) of 2^8 M
__device__ __forceinline__ void pack_image16(float *dst, int pitch, const float *src, int rows, int cols, int sr, int sc, int tid, int nthr) {
  const int gpr = cols >> 2;
  for (int i = tid; i < rows * gpr; i += nthr) {
    const int r = i / gpr, q = i - r * gpr;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = src[(long)r * sr + (long)(4 * q + j) * sc] * WSC;
    const H8 h = split4(v);
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 a = __builtin_bit_cast(u2, h.hi), b = __builtin_bit_cast(u2, h.lo);
    *reinterpret_cast<f32x4 *>(dst + r * pitch + 4 * q) = (f32x4){__uint_as_float(a[0]), __uint_as_float(a[1]), __uint_as_float(b[0]), __uint_as_float(b[1])};
  }
}

// the power of two gs with max * gs in [2^4, 2^5) for a gradient tensor whose max |.| has the bits b (1 for an all-zero / non-finite one)
__device__ __forceinline__ float grad_scale16(unsigned b, float &ginv) {
  const int e = (int)(b >> 23);
  ginv = 1.f;
  if (b == 0 || e >= 255) return 1.f;
  int k = 4 + 127 - e;
  k = k > 126 ? 126 : (k < -126 ? -126 : k);
  ginv = __uint_as_float((unsigned)(127 - k) << 23);
  return __uint_as_float((unsigned)(127 + k) << 23);
}

__global__ __launch_bounds__(THREADS) void tail16_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tok = lane & 15, g = lane >> 4;
  pack_image16(lds + L_WO, PW, a.wo, D, D, D, 1, tid, THREADS);
  pack_image16(lds + L_W1, PW, a.w1, F, D, D, 1, tid, THREADS);
  pack_image16(lds + L_W2, PW2, a.w2, D, F, F, 1, tid, THREADS);
  pack_image16(lds + L_WOT, PW, a.wo, D, D, 1, D, tid, THREADS);        // Wo^T [32][36]
  pack_image16(lds + L_W1T, PW2, a.w1, D, F, 1, D, tid, THREADS);       // W1^T [32][132]
  pack_image16(lds + L_W2T, PW, a.w2, F, D, 1, F, tid, THREADS);        // W2^T [128][36]
  if (tid < F) lds[L_PRM + P_B1 + tid] = a.b1[tid];
  if (tid < D) {
    lds[L_PRM + P_BO + tid] = a.bo[tid]; lds[L_PRM + P_B2 + tid] = a.b2[tid];
    lds[L_PRM + P_G1 + tid] = a.g1[tid]; lds[L_PRM + P_E1 + tid] = a.e1[tid];
    lds[L_PRM + P_G2 + tid] = a.g2[tid]; lds[L_PRM + P_E2 + tid] = a.e2[tid];
  }
  __syncthreads();
  // the scale of the gradients: max |dY| * gs in [2^4, 2^5)
  float ginv;
  const float gs = grad_scale16(*a.dy_max_bits, ginv);
  f32x4 gWo[2][2], gW1[8][2], gW2[2][8];
  f32x4 gG1[2], gE1[2], gG2[2], gE2[2];
  float gBo[2], gB1[8], gB2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) gWo[i][j] = fused::zero4();
#pragma unroll
    for (int j = 0; j < 8; ++j) { gW1[j][i] = fused::zero4(); gW2[i][j] = fused::zero4(); }
    gG1[i] = gE1[i] = gG2[i] = gE2[i] = fused::zero4();
    gBo[i] = gB2[i] = 0.f;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) gB1[j] = 0.f;

  const long ntiles = (a.M + 15) / 16;
  const long tstep = (long)gridDim.x * WAVES;
  f32x4 nx[2], na[2], ndy[2];
  auto load_tile = [&](long tile) {
    const long r = min(tile * 16 + tok, a.M - 1);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      nx[mt] = ld4(a.X + r * D + 16 * mt + 4 * g);
      na[mt] = ld4(a.A + r * D + 16 * mt + 4 * g);
      ndy[mt] = ld4(a.dY + r * D + 16 * mt + 4 * g);
    }
  };
  float da_max = 0.f;
  if ((long)blockIdx.x * WAVES + wave < ntiles) load_tile((long)blockIdx.x * WAVES + wave);
  for (long tile = (long)blockIdx.x * WAVES + wave; tile < ntiles; tile += tstep) {
    int zoff = 0;
    asm volatile("" : "+v"(zoff));
    const float *W = lds + zoff, *prm = W + L_PRM;
    float *scr = lds + zoff + L_SCR + wave * SCR;
    const long row = tile * 16 + tok;
    const bool ok = row < a.M;
    f32x4 x[2], at[2], dy[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      x[mt] = nx[mt];
      at[mt] = na[mt];
      dy[mt] = ok ? ndy[mt] * gs : fused::zero4();
    }
    if (tile + tstep < ntiles) load_tile(tile + tstep);
    // ---- forward ---------------------------------------------------------------------------------------------
    H8 atS[2] = {split4(at[0]), split4(at[1])};
    f32x4 n1[2] = {fused::zero4(), fused::zero4()};
    mm_fwd16<2, 2>(n1, W + L_WO, PW, atS, tok, g);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) n1[mt] = n1[mt] * WINV16 + ld4(prm + P_BO + 16 * mt + 4 * g) + x[mt];
    const float rstd1 = normalise(n1);
    f32x4 x1[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) x1[mt] = n1[mt] * ld4(prm + P_G1 + 16 * mt + 4 * g) + ld4(prm + P_E1 + 16 * mt + 4 * g);
    H8 x1S[2] = {split4(x1[0]), split4(x1[1])};
    f32x4 h[8];
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) h[ob] = fused::zero4();
    mm_fwd16<8, 2>(h, W + L_W1, PW, x1S, tok, g);
    H8 hS[8];
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) {
      const f32x4 b1v = ld4(prm + P_B1 + 16 * ob + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) h[ob][r] = relu_nn(fmaf(h[ob][r], WINV16, b1v[r]));
      hS[ob] = split4(h[ob]);
    }
    f32x4 n2[2] = {fused::zero4(), fused::zero4()};
    mm_fwd16<2, 8>(n2, W + L_W2, PW2, hS, tok, g);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) n2[mt] = n2[mt] * WINV16 + ld4(prm + P_B2 + 16 * mt + 4 * g) + x1[mt];
    const float rstd2 = normalise(n2);
    // ---- backward (every gradient below carries the factor gs) ---------------------------------------------------
    f32x4 du2[2];
    ln_backward(du2, dy, n2, rstd2, prm + P_G2, gG2, gE2, g);
    f32x4 du2N[2], x1N[2];
    to_n2(du2N, x1N, du2[0], du2[1], x1[0], x1[1], scr, tok, g);
    gB2[0] += sum4(du2N[0]); gB2[1] += sum4(du2N[1]);
    const H8 du2S[2] = {split4(du2[0]), split4(du2[1])};
    const H8 du2NS[2] = {split4(du2N[0]), split4(du2N[1])}, x1NS[2] = {split4(x1N[0]), split4(x1N[1])};
    f32x4 dx1a[2] = {fused::zero4(), fused::zero4()};      // W1^T dh, in units of 2^8
    auto dh_chunk = [&](f32x4 (&dh)[2], int kc) {
      dh[0] = dh[1] = fused::zero4();
      mm_fwd16<2, 2>(dh, W + L_W2T + 32 * kc * PW, PW, du2S, tok, g);
    };
    f32x4 dhc[2];
    dh_chunk(dhc, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) dhc[j][r] = h[j][r] > 0.f ? dhc[j][r] * WINV16 : 0.f;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      f32x4 hN[2], dhN[2];
      to_n2(hN, dhN, h[2 * kc], h[2 * kc + 1], dhc[0], dhc[1], scr, tok, g);
      const H8 dhcS[2] = {split4(dhc[0]), split4(dhc[1])};
      mm_fwd16<2, 2>(dx1a, W + L_W1T + 32 * kc, PW2, dhcS, tok, g);
      f32x4 dhn[2];
      if (kc < 3) dh_chunk(dhn, kc + 1);
      const H8 hNS[2] = {split4(hN[0]), split4(hN[1])}, dhNS[2] = {split4(dhN[0]), split4(dhN[1])};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        mm_dw4_16(gW2[0][2 * kc + j], gW2[1][2 * kc + j], gW1[2 * kc + j][0], gW1[2 * kc + j][1], du2NS[0], hNS[j], du2NS[1], hNS[j],
                  dhNS[j], x1NS[0], dhNS[j], x1NS[1]);
        gB1[2 * kc + j] += sum4(dhN[j]);
      }
      if (kc < 3) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) dhc[j][r] = h[2 * kc + 2 + j][r] > 0.f ? dhn[j][r] * WINV16 : 0.f;
      }
    }
    f32x4 dx1[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) dx1[mt] = du2[mt] + dx1a[mt] * WINV16;
    f32x4 du1[2];
    ln_backward(du1, dx1, n1, rstd1, prm + P_G1, gG1, gE1, g);
    const H8 du1S[2] = {split4(du1[0]), split4(du1[1])};
    f32x4 da[2] = {fused::zero4(), fused::zero4()};
    mm_fwd16<2, 2>(da, W + L_WOT, PW, du1S, tok, g);
    if (ok) {
      const float dsc = WINV16 * ginv;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const f32x4 dav = da[mt] * dsc;
        *reinterpret_cast<f32x4 *>(a.dA + row * D + 16 * mt + 4 * g) = dav;
        *reinterpret_cast<f32x4 *>(a.dU + row * D + 16 * mt + 4 * g) = du1[mt] * ginv;
        da_max = fmaxf(fmaxf(da_max, fmaxf(fabsf(dav[0]), fabsf(dav[1]))), fmaxf(fabsf(dav[2]), fabsf(dav[3])));
      }
    }
    f32x4 du1N[2], aN[2];
    to_n2(du1N, aN, du1[0], du1[1], at[0], at[1], scr, tok, g);
    const H8 du1NS[2] = {split4(du1N[0]), split4(du1N[1])}, aNS[2] = {split4(aN[0]), split4(aN[1])};
    mm_dw4_16(gWo[0][0], gWo[0][1], gWo[1][0], gWo[1][1], du1NS[0], aNS[0], du1NS[0], aNS[1], du1NS[1], aNS[0], du1NS[1], aNS[1]);
    gBo[0] += sum4(du1N[0]); gBo[1] += sum4(du1N[1]);
  }
  if (a.da_absmax) {
    da_max = wave_max(da_max);
    if (lane == 0 && da_max > 0.f) atomicMax(a.da_absmax, __float_as_uint(da_max));
  }
  // ---- the workgroup's gradients (divided by the gradient scale): LDS staging, then one atomic per element -----------------
  float *const stg = lds;
  __syncthreads();
  for (int i = tid; i < G_TOT; i += THREADS) stg[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) atomicAdd(&stg[G_WO + (16 * i + 4 * g + r) * D + 16 * j + tok], gWo[i][j][r] * ginv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        atomicAdd(&stg[G_W1 + (16 * j + 4 * g + r) * D + 16 * i + tok], gW1[j][i][r] * ginv);
        atomicAdd(&stg[G_W2 + (16 * i + 4 * g + r) * F + 16 * j + tok], gW2[i][j][r] * ginv);
      }
      atomicAdd(&stg[G_PRM + P_G1 + 16 * i + 4 * g + r], gG1[i][r] * ginv);
      atomicAdd(&stg[G_PRM + P_E1 + 16 * i + 4 * g + r], gE1[i][r] * ginv);
      atomicAdd(&stg[G_PRM + P_G2 + 16 * i + 4 * g + r], gG2[i][r] * ginv);
      atomicAdd(&stg[G_PRM + P_E2 + 16 * i + 4 * g + r], gE2[i][r] * ginv);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    atomicAdd(&stg[G_PRM + P_BO + 16 * i + tok], gBo[i] * ginv);
    atomicAdd(&stg[G_PRM + P_B2 + 16 * i + tok], gB2[i] * ginv);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) atomicAdd(&stg[G_PRM + P_B1 + 16 * j + tok], gB1[j] * ginv);
  __syncthreads();
  for (int i = tid; i < D * D; i += THREADS) unsafeAtomicAdd(a.dwo + i, stg[G_WO + i]);
  for (int i = tid; i < F * D; i += THREADS) unsafeAtomicAdd(a.dw1 + i, stg[G_W1 + i]);
  for (int i = tid; i < D * F; i += THREADS) unsafeAtomicAdd(a.dw2 + i, stg[G_W2 + i]);
  if (tid < F) unsafeAtomicAdd(a.db1 + tid, stg[G_PRM + P_B1 + tid]);
  if (tid < D) {
    unsafeAtomicAdd(a.dbo + tid, stg[G_PRM + P_BO + tid]);
    unsafeAtomicAdd(a.db2 + tid, stg[G_PRM + P_B2 + tid]);
    unsafeAtomicAdd(a.dg1 + tid, stg[G_PRM + P_G1 + tid]);
    unsafeAtomicAdd(a.de1 + tid, stg[G_PRM + P_E1 + tid]);
    unsafeAtomicAdd(a.dg2 + tid, stg[G_PRM + P_G2 + tid]);
    unsafeAtomicAdd(a.de2 + tid, stg[G_PRM + P_E2 + tid]);
  }
}

}  // namespace tailbwd
