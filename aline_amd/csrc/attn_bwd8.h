// Masked set-attention backward at head_dim 8 for up to 160 keys on the f16 matrix pipe (round 4): the training backward of the d = 32
// model beyond the 48 keys of the fused attention block (cfg3: al_mix, 100 data targets + up to 50 context points; the fp32 VALU
// `attention_bwd_kernel<8>` was 31 of that step's 68 ms).  Same inputs and outputs as that kernel.
//   P = softmax_keys(Q K^T / sqrt(hd)) over the keys a row may see;  dV = P^T dO;  dP = dO V^T;  dS = P (dP - delta),
//   delta_i = dO_i . O_i;  dQ = dS K / sqrt(hd);  dK = dS^T Q / sqrt(hd)                              (model/encoder.py:8-46 backwards)
// Structure of attn_bwd_wide.h (one workgroup per instance, one wave per head, S with the token rows on the register axis and the keys on
// the lanes, P / dS straight from the accumulators as B operands of dV^T / dK^T, one 16 x 16 transpose per (row tile, key tile) for dQ^T,
// dK^T / dV^T of every key tile resident over the row tiles, no atomics), with two differences:
//   * every product is the 3-term f16 split on v_mfma_f32_16x16x16_f16 (tail_bwd.h).  The four accumulator registers of a lane ARE the
//     k = 4 g .. 4 g + 3 slice of that instruction's B operand, as they were four k-steps of the fp32 one.  A head has 8 channels: they sit
//     in k (or M) positions 0 .. 7, positions 8 .. 15 are zeros -- half of every tile, as at fp32, at 3 x 16 pipe cycles per tile instead of
//     2 x 32 (S, dP) or 4 x 32 (dV, dK, dQ).  dO is scaled by the power of two of max |dA| at the staging (the backward is linear in it).
//   * a row tile's scores of all its key tiles stay in registers (4 per tile) through maximum, 2^(s - max) and sum; dP / dS exist for ONE key
//     tile at a time;
//   * TWO waves per head (eight per workgroup, two per SIMD) share the head's K / V operands -- packed (hi | lo) quads in LDS, 30 KB per head at
//     160 keys -- and take alternate row tiles; each keeps its own dK^T / dV^T tiles (80 registers at ten key tiles), the second wave of a
//     pair adds its sums to the rows the first has stored.  (First version: one wave per head with the operands in registers, 350
//     registers, one wave per SIMD: 11.8 ms per call at the cfg3 shape against 10.4 ms of the VALU kernel -- every dependent step exposed.)
#pragma once
#include "tail_bwd.h"
#include "kernels.h"

namespace abw8 {

using tailbwd::H8;
using tailbwd::split4;
constexpr int HD = 8, PT = 12, HEADS = 4, WAVES = 2 * HEADS;      // staged row pitch (floats): 48-byte rows
// LDS (floats) per wave: Q | dO | O rows of a tile [16][12], the transpose slot [16][16], delta [16];  per head: KB | VB | KA operands [3][NKT][64][4]
__host__ __device__ constexpr int wave_lds_floats() { return 3 * 16 * PT + 256 + 16; }
__host__ __device__ constexpr int head_lds_floats(int nkt) { return 3 * nkt * 256; }
__host__ __device__ inline size_t lds_bytes(int nkt, int N) {
  const int npad = (N + 15) / 16 * 16;
  return (size_t)(16 * nkt + 2 * npad + 8) * 4 + (size_t)(WAVES * wave_lds_floats() + HEADS * head_lds_floats(nkt)) * 4;
}
__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ void mfma3(f32x4 &acc, const H8 &a, const H8 &b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a.lo, b.hi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a.hi, b.lo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a.hi, b.hi, acc, 0, 0, 0);
}

// d = 32, 4 heads of 8 (one wave each); dA_max_bits: bits of max |dA| (the tail kernel reduces what it writes)
template <int NKT>
__global__ __launch_bounds__(64 * WAVES) void attention_bwd8_kernel(Geo g, const float *__restrict__ QKV, const float *__restrict__ dA,
                                                                    const float *__restrict__ Aout, float *__restrict__ dQKV,
                                                                    const unsigned *__restrict__ dA_max_bits) {
  constexpr int D = 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int b = blockIdx.x;
  const int npad = (g.N + 15) / 16 * 16, n_t = g.n_td + g.n_th;
  int *keyrow = reinterpret_cast<int *>(lds);            // [16 NKT]
  int *nvis = keyrow + 16 * NKT;                          // [npad] keys visible to a row (0 for the padding rows)
  int *kidx = nvis + npad;                                // [npad] position of a row in the key list, or -1
  int *cnt = kidx + npad;                                 // [8] n_ck, n_ak
  float *wl = reinterpret_cast<float *>(cnt + 8) + (size_t)wave * wave_lds_floats();
  float *Qs = wl, *Gs = Qs + 16 * PT, *Os = Gs + 16 * PT, *Ts = Os + 16 * PT, *Dl = Ts + 256;
  const int h = wave & (HEADS - 1), half = wave >> 2;     // two waves per head: alternate row tiles
  float *KBs = reinterpret_cast<float *>(cnt + 8) + (size_t)WAVES * wave_lds_floats() + (size_t)h * head_lds_floats(NKT);
  float *VBs = KBs + NKT * 256, *KAs = VBs + NKT * 256;
  const long ep = (long)b * g.N;
  // ---- key list (context rows in slot order, then the visible targets) and the per-row visibility, by wave 0 ----------------------
  if (wave == 0) {
    int n = 0;
    for (int c0 = 0; c0 < npad; c0 += 64) {
      const int row = c0 + lane;
      const bool key = row < g.P && is_ctx(g, b, row);
      const unsigned long long bal = __ballot(key);
      const int k = n + __popcll(bal & ((1ull << lane) - 1ull));
      if (row < npad) kidx[row] = (key && k < 16 * NKT) ? k : -1;
      if (key && k < 16 * NKT) keyrow[k] = row;
      n += __popcll(bal);
    }
    n = min(n, 16 * NKT);
    const int n_ck = n;
    for (int c0 = 0; c0 < n_t; c0 += 64) {
      const int j = c0 + lane;
      const bool key = j < n_t && (!g.tmask || g.tmask[j]);
      const unsigned long long bal = __ballot(key);
      const int k = n + __popcll(bal & ((1ull << lane) - 1ull));
      if (key && k < 16 * NKT) { keyrow[k] = g.P + j; kidx[g.P + j] = k; }
      n += __popcll(bal);
    }
    n = min(n, 16 * NKT);
    if (lane == 0) { cnt[0] = n_ck; cnt[1] = n; }
    for (int k = n + lane; k < 16 * NKT; k += 64) keyrow[k] = -1;
  }
  __syncthreads();
  const int n_ck = cnt[0], n_ak = cnt[1];
  for (int row = tid; row < npad; row += blockDim.x) {
    const bool isq = row < g.P && kidx[row] < 0;          // (a point row that is not a context key is a remaining query)
    nvis[row] = row < g.N ? (isq ? n_ak : n_ck) : 0;
  }
  __syncthreads();
  const int nkt_all = (n_ak + 15) >> 4, nkt_ctx = (n_ck + 15) >> 4;
  float ginv;
  const float gs = tailbwd::grad_scale16(*dA_max_bits, ginv);
  const float scale = rsqrtf((float)HD) * 1.44269504088896340736f;      // scores in base-2 units; dQ / dK get 1 / sqrt(hd) = scale * ln 2 at the end
  const float oscale = rsqrtf((float)HD) * ginv, kscale = 0.69314718055994530942f * ginv;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const bool mine = fg < 2;                                // lane groups 0, 1 carry the head's 8 channels on a k (or M = fr < 8) axis
  const int qc = h * HD, kc = D + h * HD, vc = 2 * D + h * HD;
  // ---- K / V of the head's key rows, once: KB / VB [kt]: B operands of S / dP (k = channel 4 g .. 4 g + 3, n = key fr);  KA [kt] -> LDS: A operand
  // of dQ^T (m = channel fr, k = keys 4 g .. 4 g + 3 of the tile)
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  auto pack = [](const f32x4 &v) {
    const H8 s8 = split4(v);
    const u2 ph = __builtin_bit_cast(u2, s8.hi), pl = __builtin_bit_cast(u2, s8.lo);
    return (f32x4){__uint_as_float(ph[0]), __uint_as_float(ph[1]), __uint_as_float(pl[0]), __uint_as_float(pl[1])};
  };
  f32x4 dKT[NKT], dVT[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    dKT[kt] = z4; dVT[kt] = z4;
    if (kt < nkt_all && half == 0) {
      const int kr = keyrow[16 * kt + fr];
      const float *kp = QKV + (ep + max(kr, 0)) * 3 * D;
      f32x4 kv = z4, vv = z4;
      if (kr >= 0 && mine) { kv = *reinterpret_cast<const f32x4 *>(kp + kc + 4 * fg); vv = *reinterpret_cast<const f32x4 *>(kp + vc + 4 * fg); }
      *reinterpret_cast<f32x4 *>(KBs + (kt * 64 + lane) * 4) = pack(kv);
      *reinterpret_cast<f32x4 *>(VBs + (kt * 64 + lane) * 4) = pack(vv);
      f32x4 ka = z4;
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        const int kr2 = keyrow[16 * kt + 4 * fg + sp];
        if (kr2 >= 0 && fr < HD) ka[sp] = QKV[(ep + kr2) * 3 * D + kc + fr];
      }
      *reinterpret_cast<f32x4 *>(KAs + (kt * 64 + lane) * 4) = pack(ka);
    }
  }
  __syncthreads();
  // ---- row tiles -----------------------------------------------------------------------------------------------------------------
  for (int r0 = 16 * half; r0 < npad; r0 += 32) {
    // stage Q (scaled), dO (scaled by gs), O of the 16 rows: two lanes per row of 8 floats
    if (lane < 32) {
      const int rr = lane >> 1, c4 = 4 * (lane & 1), row = r0 + rr;
      f32x4 q = z4, go = z4, oo = z4;
      if (row < g.N) {
        q = *reinterpret_cast<const f32x4 *>(QKV + (ep + row) * 3 * D + qc + c4);
        go = *reinterpret_cast<const f32x4 *>(dA + (ep + row) * D + qc + c4);
        oo = *reinterpret_cast<const f32x4 *>(Aout + (ep + row) * D + qc + c4);
      }
      *reinterpret_cast<f32x4 *>(Qs + rr * PT + c4) = q * scale;
      *reinterpret_cast<f32x4 *>(Gs + rr * PT + c4) = go * gs;
      *reinterpret_cast<f32x4 *>(Os + rr * PT + c4) = oo;
    }
    // (the wave's own LDS traffic is ordered: no barrier)  delta of row fr (scaled by gs): two channels per lane group
    {
      float dl = fmaf(Gs[fr * PT + 2 * fg], Os[fr * PT + 2 * fg], Gs[fr * PT + 2 * fg + 1] * Os[fr * PT + 2 * fg + 1]);
      dl += __shfl_xor(dl, 16, 64);
      dl += __shfl_xor(dl, 32, 64);
      if (fg == 0) Dl[fr] = dl;
    }
    // operands of the tile: QA / GA: A of S / dP (m = row fr, k = channels 4 g ..: groups 0, 1);  QTA / GTA: A of dK^T / dV^T (m = channel fr < 8, k = rows 4 g ..)
    f32x4 qa = z4, ga = z4, qta = z4, gta = z4;
    if (mine) { qa = *reinterpret_cast<const f32x4 *>(Qs + fr * PT + 4 * fg); ga = *reinterpret_cast<const f32x4 *>(Gs + fr * PT + 4 * fg); }
    if (fr < HD) {
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) { qta[sp] = Qs[(4 * fg + sp) * PT + fr]; gta[sp] = Gs[(4 * fg + sp) * PT + fr]; }
    }
    const H8 QA = split4(qa), GA = split4(ga), QTA = split4(qta), GTA = split4(gta);
    const int4 nv = *reinterpret_cast<const int4 *>(nvis + r0 + 4 * fg);
    const f32x4 dl4 = *reinterpret_cast<const f32x4 *>(Dl + 4 * fg);
    const int nkt = r0 < g.P ? nkt_all : nkt_ctx;          // (wave-uniform) a tile without point rows sees the context keys only
    // sweep 1: the scores of every key tile (kept: 4 registers per tile), their maximum per row; then e = 2^(s - max) in place and the sum
    const int nvr[4] = {nv.x, nv.y, nv.z, nv.w};
    f32x4 S[NKT];
    f32x4 mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, sum = z4;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt < nkt) {
        S[kt] = z4;
        mfma3(S[kt], QA, tailbwd::as_h8(*reinterpret_cast<const f32x4 *>(KBs + (kt * 64 + lane) * 4)));
        const int key = 16 * kt + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          S[kt][r] = key < nvr[r] ? S[kt][r] : -INFINITY;
          mx[r] = fmaxf(mx[r], S[kt][r]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) mx[r] = group16_max(mx[r]);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt < nkt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = mx[r] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(S[kt][r] - mx[r]);      // (a masked score: 2^-inf = 0; a row without keys: 0)
          S[kt][r] = e;
          sum[r] += e;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float l = group16_sum(sum[r]); sum[r] = l > 0.f ? 1.f / l : 0.f; }
    // sweep 2: P, dS;  dV^T += dO^T P,  dK^T += Q^T dS;  dQ^T += K^T dS^T through the transpose slot
    f32x4 dQT = z4;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt < nkt) {
        f32x4 dP = z4;
        mfma3(dP, GA, tailbwd::as_h8(*reinterpret_cast<const f32x4 *>(VBs + (kt * 64 + lane) * 4)));
        f32x4 P, dS;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          P[r] = S[kt][r] * sum[r];
          dS[r] = P[r] * (dP[r] - dl4[r]);
        }
        const H8 PS = split4(P), dSS = split4(dS);
        mfma3(dVT[kt], GTA, PS);
        mfma3(dKT[kt], QTA, dSS);
        // transpose dS: write [row 4 g + r][key fr], read [row fr][keys 4 g .. 4 g + 3]
#pragma unroll
        for (int r = 0; r < 4; ++r) Ts[(4 * fg + r) * 16 + fr] = dS[r];
        const H8 dST = split4(*reinterpret_cast<const f32x4 *>(Ts + fr * 16 + 4 * fg));
        const H8 KA = tailbwd::as_h8(*reinterpret_cast<const f32x4 *>(KAs + (kt * 64 + lane) * 4));
        mfma3(dQT, KA, dST);
      }
    }
    // dQ of the tile (channels 4 g + r (g < 2) of row fr), zeros for the K / V gradient slices of the rows that are not keys
    const int row = r0 + fr;
    if (row < g.N && mine) {
      float *out = dQKV + (ep + row) * 3 * D;
      *reinterpret_cast<f32x4 *>(out + qc + 4 * fg) = dQT * oscale;
      if (kidx[row] < 0) {
        *reinterpret_cast<f32x4 *>(out + kc + 4 * fg) = z4;
        *reinterpret_cast<f32x4 *>(out + vc + 4 * fg) = z4;
      }
    }
  }
  // ---- dK / dV of the head's key rows (Q was staged with the base-2 scale: dK takes ln 2): the first wave of a pair stores its sums, the
  // second adds its own behind a barrier (same CU: the rows come back from its L1 / L2) ------------------------------------------------
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == half) {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        if (kt < nkt_all) {
          const int kr = keyrow[16 * kt + fr];
          if (kr >= 0 && mine) {
            float *out = dQKV + (ep + kr) * 3 * D;
            f32x4 dk = dKT[kt] * kscale, dv = dVT[kt] * ginv;
            if (pass == 1) { dk += *reinterpret_cast<const f32x4 *>(out + kc + 4 * fg); dv += *reinterpret_cast<const f32x4 *>(out + vc + 4 * fg); }
            *reinterpret_cast<f32x4 *>(out + kc + 4 * fg) = dk;
            *reinterpret_cast<f32x4 *>(out + vc + 4 * fg) = dv;
          }
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace abw8
