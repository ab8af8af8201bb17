// Masked set-attention backward on the matrix pipe for the small-width model (d = 32, 4 heads of 8, <= 48 keys per instance):
//   P = softmax(Q K^T / sqrt(hd)) over the visible keys;  dV = P^T dO;  dS = P (dO V^T - delta);  dQ = dS K;  dK = dS^T Q
// (model/encoder.py:83-126 mask, nn.MultiheadAttention; `loss.backward()` of train_aline.py:124-132 through it).
// One workgroup per (step, episode) instance, one 16-row token tile per wave at a time, all products exact fp32
// (v_mfma_f32_16x16x4_f32) in the two register layouts of tail_bwd.h:
//   S^T, dP^T  [keys x rows]  = Kblk q^T, Vblk dO^T : q / dO in the T layout are the B operands, the K / V rows of the key tile
//                               (from LDS, the other heads' channels zeroed) the A operands; keys land on the register axis,
//                               so softmax and delta reduce over registers + the 4 lane groups;
//   dQ^T       [chan x rows]  = Kblk^T dS^T         : dS^T is, register for register, the B operand;
//   dV^T, dK^T [chan x keys]  = dO_N^T P_N, q_N^T dS_N (k = token row): N-layout operands; P^T / dS^T tiles go through a
//                               16 x 32 transpose in the wave's LDS scratch; accumulators stay resident over the wave's tiles.
// The VALU kernel it replaces (attention_bwd_kernel<8>) ran at 62 % of the fp32 vector peak: 3.5 ms per call at the headline shape.
#pragma once
#include "tail_bwd.h"

namespace abwd {

constexpr int D = 32, HD = 8, H = 4, KTMAX = 3, MAXK = 16 * KTMAX;
constexpr int PK = 36;
constexpr int WAVES = 4, THREADS = 64 * WAVES;

using fused::ld4;
using fused::group_sum;
using fused::group_max;
using fused::zero4;

__global__ __launch_bounds__(THREADS) void attention_bwd_mfma_kernel(Geo g, const float *__restrict__ QKV,
                                                                     const float *__restrict__ dA, float *__restrict__ dQKV) {
  __shared__ __attribute__((aligned(16))) float Ks[MAXK * PK], Vs[MAXK * PK], dKs[MAXK * D], dVs[MAXK * D], scrs[WAVES * 16 * PK];
  __shared__ int keyrow[MAXK];
  __shared__ int wave_cnt[WAVES];
  __shared__ int s_base;
  const int b = blockIdx.x;
  if (b >= g.B) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tok = lane & 15, gq = lane >> 4;
  const int n_t = g.n_td + g.n_th;
  // ---- key list: context points in slot order, then the visible targets -----------------------------------------
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < g.P; c0 += THREADS) {
    const int row = c0 + tid;
    const bool key = row < g.P && is_ctx(g, b, row);
    const unsigned long long bal = __ballot(key);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    const int k = off + __popcll(bal & ((1ull << lane) - 1ull));
    if (key && k < MAXK) keyrow[k] = row;
    __syncthreads();
    if (tid == 0) s_base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  const int n_ck = min(s_base, MAXK);
  __syncthreads();
  if (tid == 0) {
    int n = n_ck;
    for (int j = 0; j < n_t; ++j)
      if ((!g.tmask || g.tmask[j]) && n < MAXK) keyrow[n++] = g.P + j;
    s_base = n;
  }
  __syncthreads();
  const int n_ak = s_base;
  const int nkt = (n_ak + 15) >> 4;
  const long ep = (long)b * g.N;
  for (int i = tid; i < 16 * nkt * 8; i += THREADS) {
    const int j = i >> 3, c4 = (i & 7) * 4;
    f32x4 kv = zero4(), vv = zero4();
    if (j < n_ak) {
      const float *src = QKV + (ep + keyrow[j]) * 3 * D + c4;
      kv = ld4(src + D);
      vv = ld4(src + 2 * D);
    }
    *reinterpret_cast<f32x4 *>(Ks + j * PK + c4) = kv;
    *reinterpret_cast<f32x4 *>(Vs + j * PK + c4) = vv;
    *reinterpret_cast<f32x4 *>(dKs + j * D + c4) = zero4();
    *reinterpret_cast<f32x4 *>(dVs + j * D + c4) = zero4();
  }
  __syncthreads();

  const float scale = rsqrtf((float)HD);
  float *scr = scrs + wave * 16 * PK;
  f32x4 dKt[2][KTMAX], dVt[2][KTMAX];      // [chan 16 mt + 4 g + r][key 16 kt + tok], summed over this wave's tiles
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int kt = 0; kt < KTMAX; ++kt) { dKt[mt][kt] = zero4(); dVt[mt][kt] = zero4(); }

  const int ntile = (g.N + 15) >> 4;
  for (int tile = wave; tile < ntile; tile += WAVES) {
    const int row = tile * 16 + tok;
    const bool ok = row < g.N;
    const int rc = ok ? row : g.N - 1;
    const bool isq = ok && row < g.P && !is_ctx(g, b, row);
    const int nk = !ok ? 0 : (isq ? n_ak : n_ck);
    f32x4 q[2], go[2], dq[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      q[mt] = ld4(QKV + (ep + rc) * 3 * D + 16 * mt + 4 * gq) * scale;
      go[mt] = ok ? ld4(dA + (ep + rc) * D + 16 * mt + 4 * gq) : zero4();
      dq[mt] = zero4();
    }
    f32x4 qN[2], goN[2];
    tailbwd::to_n(qN, q[0], q[1], scr, tok, gq);
    tailbwd::to_n(goN, go[0], go[1], scr, tok, gq);
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const int mt = h >> 1, hg = h & 1;
      const bool mine_g = (gq >> 1) == hg;      // this lane group's channels 16 mt + 4 g + r belong to head h (k axis)
      const bool mine_c = (tok >> 3) == hg;     // channel 16 mt + tok belongs to head h (i axis)
      f32x4 s[KTMAX], dp[KTMAX];
#pragma unroll
      for (int kt = 0; kt < KTMAX; ++kt) {
        s[kt] = zero4(); dp[kt] = zero4();
        if (kt < nkt) {
          const f32x4 kf = mine_g ? ld4(Ks + (16 * kt + tok) * PK + 16 * mt + 4 * gq) : zero4();
          const f32x4 vf = mine_g ? ld4(Vs + (16 * kt + tok) * PK + 16 * mt + 4 * gq) : zero4();
#pragma unroll
          for (int r = 0; r < 4; ++r) { MFMA4(s[kt], kf[r], q[mt][r]); MFMA4(dp[kt], vf[r], go[mt][r]); }
        }
      }
      // softmax over the keys (register axis x lane groups), delta, score gradient
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < KTMAX; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) if (16 * kt + 4 * gq + r < nk) mx = fmaxf(mx, s[kt][r]);
      mx = group_max(mx);
      float l = 0.f, delta = 0.f;
#pragma unroll
      for (int kt = 0; kt < KTMAX; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = 16 * kt + 4 * gq + r < nk ? __expf(s[kt][r] - mx) : 0.f;
          s[kt][r] = e;
          l += e;
          delta = fmaf(e, dp[kt][r], delta);
        }
      l = group_sum(l);
      const float inv = l > 0.f ? 1.f / l : 0.f;
      delta = group_sum(delta) * inv;
#pragma unroll
      for (int kt = 0; kt < KTMAX; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = s[kt][r] * inv;
          s[kt][r] = p;
          dp[kt][r] = p * (dp[kt][r] - delta);      // dS^T
        }
      const f32x4 goA = mine_c ? goN[mt] : zero4(), qA = mine_c ? qN[mt] : zero4();
#pragma unroll
      for (int kt = 0; kt < KTMAX; ++kt) {
        if (kt < nkt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float ka = mine_c ? Ks[(16 * kt + 4 * gq + r) * PK + 16 * mt + tok] : 0.f;
            MFMA4(dq[mt], ka, dp[kt][r]);
          }
          f32x4 pn[2];
          tailbwd::to_n(pn, s[kt], dp[kt], scr, tok, gq);      // pn[0] = P_N, pn[1] = dS_N  (key on lane, row on (g, r))
#pragma unroll
          for (int r = 0; r < 4; ++r) { MFMA4(dVt[mt][kt], goA[r], pn[0][r]); MFMA4(dKt[mt][kt], qA[r], pn[1][r]); }
        }
      }
    }
    if (ok) {
      float *out = dQKV + (ep + row) * 3 * D + 16 * 0 + 4 * gq;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        *reinterpret_cast<f32x4 *>(out + 16 * mt) = dq[mt] * scale;
        *reinterpret_cast<f32x4 *>(out + D + 16 * mt) = zero4();       // K / V gradients of key rows are written below
        *reinterpret_cast<f32x4 *>(out + 2 * D + 16 * mt) = zero4();
      }
    }
  }
  // ---- dK / dV of the key rows: sum over the waves through LDS -----------------------------------------------------
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int kt = 0; kt < KTMAX; ++kt)
      if (kt < nkt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          atomicAdd(&dKs[(16 * kt + tok) * D + 16 * mt + 4 * gq + r], dKt[mt][kt][r]);
          atomicAdd(&dVs[(16 * kt + tok) * D + 16 * mt + 4 * gq + r], dVt[mt][kt][r]);
        }
      }
  __syncthreads();      // (also orders the zero stores of every wave's tiles before the key-row stores)
  for (int i = tid; i < n_ak * 8; i += THREADS) {
    const int j = i >> 3, c4 = (i & 7) * 4;
    float *dst = dQKV + (ep + keyrow[j]) * 3 * D + c4;
    *reinterpret_cast<f32x4 *>(dst + D) = ld4(dKs + j * D + c4);          // q already carries the 1 / sqrt(hd)
    *reinterpret_cast<f32x4 *>(dst + 2 * D) = ld4(dVs + j * D + c4);
  }
}

}  // namespace abwd
