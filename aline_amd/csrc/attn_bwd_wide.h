// Masked set-attention backward at head_dim 32 / 64 on the fp32 matrix pipe (round 4) -- the per-op backward of the wide models
// (d = 256 / 8 heads of 32: the roofline variant; d = 512 / 8 heads of 64: the psychometric configuration).  Replaces the fp32 VALU
// `attention_bwd_kernel<32 | 64>` (backward.h; 15 TFLOP/s at the d = 256 headline shape, 14 % of that training step) where an
// episode holds <= 16 NKT keys; same inputs, same outputs, exact fp32 products (v_mfma_f32_16x16x4_f32), no atomics.
//   P = softmax_keys(Q K^T / sqrt(hd)) over the keys a row may see;  dV = P^T dO;  dP = dO V^T;  dS = P (dP - delta),
//   delta_i = dO_i . O_i;  dQ = dS K / sqrt(hd);  dK = dS^T Q / sqrt(hd)                              (model/encoder.py:8-46 backwards)
//
// One workgroup per instance, one wave per head (heads beyond the wave count in turn).  Head_dim is a whole number of 4-deep k-steps and
// of 16-wide output tiles, so no MFMA multiplies padding (at head_dim 8 half of every tile would be: DESIGN.md history section 7).
// Orientation: S = Q K^T with the token rows on the register axis and the keys on the lanes (D[row 4 g + r][key lane & 15]).  Then
//   * dV^T [ch, key] = sum_rows dO^T [ch, row] P [row, key] and dK^T likewise take P / dS STRAIGHT from the accumulators as B operands:
//     k-step s' of the MFMA reads accumulator register s' -- lane group g then contributes row 4 g + s' -- and the A operand
//     (dO^T or Q^T, read from the staged rows) is indexed the same way.  dK^T / dV^T of all key tiles stay in registers over all row tiles.
//   * dQ^T [ch, row] = sum_keys K^T [ch, key] dS^T [key, row] needs dS with the rows on the lanes: one 16 x 16 transpose per (row tile,
//     key tile) through a 1 KB LDS slot of the wave (4 ds_write_b32, 1 ds_read_b128).
//   * the softmax statistics of a row are reductions over the 16 lanes of a group (xor shuffles) and the key tiles; delta comes from the
//     staged dO / O rows.
// Q, dO, O rows of a tile are staged in LDS with coalesced float4 loads (a wave reads 8 whole rows per instruction) and read back in the
// operand layouts; K / V of the head's key rows are loaded once per head into registers in both operand layouts.
#pragma once
#include "common.h"
#include "kernels.h"
#include "tail_bwd.h"

namespace abww {

constexpr int MAXW = 8;                       // waves per workgroup
template <int HD> __host__ __device__ constexpr int pitch() { return HD + 4; }      // staged row pitch (floats): 16-byte aligned rows
// LDS (floats) per wave: Q | dO | O rows of a tile [16][pitch], the transpose slot [16][16], row statistics [3][16]
template <int HD> __host__ __device__ constexpr int wave_lds_floats() { return 3 * 16 * pitch<HD>() + 256 + 48; }
// workgroup: key list [16 NKT] ints, visible-key count per row [Npad] ints, key index of a row (or -1) [Npad] ints, then the waves
__host__ __device__ inline size_t lds_bytes(int hd, int nkt, int N, int waves) {
  const int npad = (N + 15) / 16 * 16;
  return (size_t)(16 * nkt + 2 * npad + 8) * 4 + (size_t)waves * (hd == 32 ? wave_lds_floats<32>() : wave_lds_floats<64>()) * 4;
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

// (head_dim 64: the operands of two key tiles + a row tile are ~270 registers: four waves per workgroup with the whole 512-entry file)
template <int HD, int NKT>
__global__ __launch_bounds__(HD == 64 ? 256 : 64 * MAXW) void attention_bwd_wide_kernel(Geo g, int d, const float *__restrict__ QKV, const float *__restrict__ dA,
                                                                       const float *__restrict__ Aout, float *__restrict__ dQKV,
                                                                       unsigned *out_absmax, int key_rows_only) {
  constexpr int NS = HD / 4, NCT = HD / 16, PT = pitch<HD>();
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NWV = blockDim.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int b = blockIdx.x;
  const int npad = (g.N + 15) / 16 * 16, n_t = g.n_td + g.n_th, H = d / HD;
  int *keyrow = reinterpret_cast<int *>(lds);            // [16 NKT]
  int *nvis = keyrow + 16 * NKT;                          // [npad] keys visible to a row (0 for the padding rows)
  int *kidx = nvis + npad;                                // [npad] position of a row in the key list, or -1
  int *cnt = kidx + npad;                                 // [8] n_ck, n_ak
  float *wl = reinterpret_cast<float *>(cnt + 8) + (size_t)wave * wave_lds_floats<HD>();
  float *Qs = wl, *Gs = Qs + 16 * PT, *Os = Gs + 16 * PT, *Ts = Os + 16 * PT, *St = Ts + 256;
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
  const float scale = rsqrtf((float)HD);
  unsigned omax = 0;
  auto track = [&](const f32x4 &v) {
    omax = max(max(omax, __float_as_uint(v[0]) & 0x7fffffffu), max(__float_as_uint(v[1]) & 0x7fffffffu, max(__float_as_uint(v[2]) & 0x7fffffffu, __float_as_uint(v[3]) & 0x7fffffffu)));
  };
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (int h = wave; h < H; h += NWV) {
    const int qc = h * HD, kc = d + h * HD, vc = 2 * d + h * HD;
    // ---- K / V of the head's key rows, once, in both operand layouts -------------------------------------------------------------
    // KB / VB [kt][s]: B operand of S / dP (k = channel 4 s + g, n = key lane & 15);  KA [kt][ct][s']: A operand of dQ^T (m = channel
    // 16 ct + lane & 15, k -> key 4 g + s' of the tile)
    float KB[NKT][NS], VB[NKT][NS], KA[NKT][NCT][4];
    f32x4 dKT[NKT][NCT], dVT[NKT][NCT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const int kr = keyrow[16 * kt + fr];
      const float *kp = QKV + (ep + max(kr, 0)) * 3 * d;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        KB[kt][s] = kr >= 0 ? kp[kc + 4 * s + fg] : 0.f;
        VB[kt][s] = kr >= 0 ? kp[vc + 4 * s + fg] : 0.f;
      }
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        const int kr2 = keyrow[16 * kt + 4 * fg + sp];
        const float *kp2 = QKV + (ep + max(kr2, 0)) * 3 * d + kc;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) KA[kt][ct][sp] = kr2 >= 0 ? kp2[16 * ct + fr] : 0.f;
      }
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) { dKT[kt][ct] = z4; dVT[kt][ct] = z4; }
    }
    // ---- row tiles -----------------------------------------------------------------------------------------------------------------
    for (int r0 = 0; r0 < npad; r0 += 16) {
      // stage Q (scaled), dO, O of the 16 rows: a wave instruction moves 4 rows of HD floats (HD = 32: 8 lanes per row)
      constexpr int LPR = HD / 4, RPI = 64 / LPR;       // lanes per row, rows per instruction
#pragma unroll
      for (int i = 0; i < 16 / RPI; ++i) {
        const int rr = RPI * i + lane / LPR, c4 = 4 * (lane % LPR), row = r0 + rr;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f), go = q, oo = q;
        if (row < g.N) {
          q = *reinterpret_cast<const float4 *>(QKV + (ep + row) * 3 * d + qc + c4);
          go = *reinterpret_cast<const float4 *>(dA + (ep + row) * d + qc + c4);
          oo = *reinterpret_cast<const float4 *>(Aout + (ep + row) * d + qc + c4);
        }
        q.x *= scale; q.y *= scale; q.z *= scale; q.w *= scale;
        *reinterpret_cast<float4 *>(Qs + rr * PT + c4) = q;
        *reinterpret_cast<float4 *>(Gs + rr * PT + c4) = go;
        *reinterpret_cast<float4 *>(Os + rr * PT + c4) = oo;
      }
      // (the wave's own LDS traffic is ordered: no barrier)
      // delta of row fr: partial over the channels of lane group g, then over the groups; to St[32 + row]
      {
        float dl = 0.f;
#pragma unroll
        for (int c = 0; c < HD / 4; ++c) dl = fmaf(Gs[fr * PT + fg * (HD / 4) + c], Os[fr * PT + fg * (HD / 4) + c], dl);
        dl += __shfl_xor(dl, 16, 64);
        dl += __shfl_xor(dl, 32, 64);
        if (fg == 0) St[32 + fr] = dl;
      }
      // operands of the tile
      float QA[NS], GA[NS], QTA[NCT][4], GTA[NCT][4];
#pragma unroll
      for (int s = 0; s < NS; ++s) { QA[s] = Qs[fr * PT + 4 * s + fg]; GA[s] = Gs[fr * PT + 4 * s + fg]; }
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) { QTA[ct][sp] = Qs[(4 * fg + sp) * PT + 16 * ct + fr]; GTA[ct][sp] = Gs[(4 * fg + sp) * PT + 16 * ct + fr]; }
      const int4 nv = *reinterpret_cast<const int4 *>(nvis + r0 + 4 * fg);
      const f32x4 dl4 = *reinterpret_cast<const f32x4 *>(St + 32 + 4 * fg);
      // S and dP of every key tile
      f32x4 S[NKT], dP[NKT];
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        S[kt] = z4; dP[kt] = z4;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          S[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(QA[s], KB[kt][s], S[kt], 0, 0, 0);
          dP[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(GA[s], VB[kt][s], dP[kt], 0, 0, 0);
        }
      }
      // softmax over the visible keys of each row (rows 4 g + r on the registers, keys on the 16 lanes of the group and the tiles)
      f32x4 mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const int key = 16 * kt + fr;
        S[kt][0] = key < nv.x ? S[kt][0] : -INFINITY; S[kt][1] = key < nv.y ? S[kt][1] : -INFINITY;
        S[kt][2] = key < nv.z ? S[kt][2] : -INFINITY; S[kt][3] = key < nv.w ? S[kt][3] : -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) mx[r] = fmaxf(mx[r], S[kt][r]);
      }
      f32x4 sum = z4;
#pragma unroll
      for (int r = 0; r < 4; ++r) mx[r] = group16_max(mx[r]);
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = mx[r] == -INFINITY ? 0.f : __expf(S[kt][r] - mx[r]);      // (padding rows see no key: all zero)
          S[kt][r] = e;
          sum[r] += e;
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float l = group16_sum(sum[r]); sum[r] = l > 0.f ? 1.f / l : 0.f; }
      // P, dS;  dV^T += dO^T P,  dK^T += Q^T dS;  dQ^T += K^T dS^T through the transpose slot
      f32x4 dQT[NCT];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) dQT[ct] = z4;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        f32x4 P, dS;
#pragma unroll
        for (int r = 0; r < 4; ++r) { P[r] = S[kt][r] * sum[r]; dS[r] = P[r] * (dP[kt][r] - dl4[r]); }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int sp = 0; sp < 4; ++sp) {
            dVT[kt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(GTA[ct][sp], P[sp], dVT[kt][ct], 0, 0, 0);
            dKT[kt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(QTA[ct][sp], dS[sp], dKT[kt][ct], 0, 0, 0);
          }
        // transpose dS: write [row 4 g + r][key fr], read [row fr][keys 4 g .. 4 g + 3]
#pragma unroll
        for (int r = 0; r < 4; ++r) Ts[(4 * fg + r) * 16 + fr] = dS[r];
        const f32x4 dST = *reinterpret_cast<const f32x4 *>(Ts + fr * 16 + 4 * fg);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int sp = 0; sp < 4; ++sp)
            dQT[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(KA[kt][ct][sp], dST[sp], dQT[ct], 0, 0, 0);
      }
      // dQ of the tile (channels 16 ct + 4 g + r of row fr), zeros for the K / V gradient slices of the rows that are not keys
      const int row = r0 + fr;
      if (row < g.N) {
        float *out = dQKV + (ep + row) * 3 * d;
        const bool notkey = kidx[row] < 0;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const f32x4 v = dQT[ct] * scale;
          track(v);
          *reinterpret_cast<f32x4 *>(out + qc + 16 * ct + 4 * fg) = v;
          if (notkey && !key_rows_only) {      // (key_rows_only: the in-projection's products read the K | V slices of the key rows alone)
            *reinterpret_cast<f32x4 *>(out + kc + 16 * ct + 4 * fg) = z4;
            *reinterpret_cast<f32x4 *>(out + vc + 16 * ct + 4 * fg) = z4;
          }
        }
      }
    }
    // ---- dK / dV of the head's key rows (Q was staged scaled: dK carries the 1 / sqrt(hd)) ------------------------------------------
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const int kr = keyrow[16 * kt + fr];
      if (kr >= 0) {
        float *out = dQKV + (ep + kr) * 3 * d;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          track(dKT[kt][ct]); track(dVT[kt][ct]);
          *reinterpret_cast<f32x4 *>(out + kc + 16 * ct + 4 * fg) = dKT[kt][ct];
          *reinterpret_cast<f32x4 *>(out + vc + 16 * ct + 4 * fg) = dVT[kt][ct];
        }
      }
    }
  }
  if (out_absmax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) omax = max(omax, (unsigned)__shfl_xor((int)omax, o, 64));
    if (lane == 0 && omax) atomicMax(out_absmax, omax);
  }
}

// ---- the same kernel on the f16 matrix pipe (round 4, end; tail_bwd.h has the scheme) -----------------------------------------------------
// Every run of four v_mfma_f32_16x16x4_f32 over one 16-deep operand block is the 3-term f16 split on v_mfma_f32_16x16x16_f16: the four
// scalars a lane held for the four k-steps are the k = 4 g .. 4 g + 3 slice of that instruction (register counts unchanged: a (hi | lo)
// quad is four registers too); P / dS still come straight from the accumulators.  dO is multiplied by the power of two of max |dA| (the
// out-projection's dX product reduces what it stores: dA_max_bits) at the staging, dQ / dK / dV are divided by it where they leave.
template <int HD, int NKT>
__global__ __launch_bounds__(HD == 64 ? 256 : 64 * MAXW) void attention_bwd_wide16_kernel(Geo g, int d, const float *__restrict__ QKV, const float *__restrict__ dA,
                                                                         const float *__restrict__ Aout, float *__restrict__ dQKV,
                                                                         unsigned *out_absmax, int key_rows_only, const unsigned *__restrict__ dA_max_bits) {
  using tailbwd::H8;
  using tailbwd::split4;
  constexpr int NG = HD / 16, NCT = HD / 16, PT = pitch<HD>();      // 16-deep k groups of S / dP, 16-wide channel tiles of dK^T / dV^T / dQ^T
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NWV = blockDim.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int b = blockIdx.x;
  const int npad = (g.N + 15) / 16 * 16, n_t = g.n_td + g.n_th, H = d / HD;
  int *keyrow = reinterpret_cast<int *>(lds);            // [16 NKT]
  int *nvis = keyrow + 16 * NKT;                          // [npad] keys visible to a row (0 for the padding rows)
  int *kidx = nvis + npad;                                // [npad] position of a row in the key list, or -1
  int *cnt = kidx + npad;                                 // [8] n_ck, n_ak
  float *wl = reinterpret_cast<float *>(cnt + 8) + (size_t)wave * wave_lds_floats<HD>();
  float *Qs = wl, *Gs = Qs + 16 * PT, *Os = Gs + 16 * PT, *Ts = Os + 16 * PT, *St = Ts + 256;
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
  const float scale = rsqrtf((float)HD);
  float ginv;
  const float gs = tailbwd::grad_scale16(*dA_max_bits, ginv);
  auto mfma3 = [](f32x4 &acc, const H8 &a, const H8 &b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a.lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a.hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a.hi, b.hi, acc, 0, 0, 0);
  };
  unsigned omax = 0;
  auto track = [&](const f32x4 &v) {
    omax = max(max(omax, __float_as_uint(v[0]) & 0x7fffffffu), max(__float_as_uint(v[1]) & 0x7fffffffu, max(__float_as_uint(v[2]) & 0x7fffffffu, __float_as_uint(v[3]) & 0x7fffffffu)));
  };
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (int h = wave; h < H; h += NWV) {
    const int qc = h * HD, kc = d + h * HD, vc = 2 * d + h * HD;
    // ---- K / V of the head's key rows, once, in both operand layouts -------------------------------------------------------------
    // KB / VB [kt][kg]: B operand of S / dP (k = channels 16 kg + 4 g .. + 3, n = key lane & 15);  KA [kt][ct]: A operand of dQ^T (m = channel
    // 16 ct + lane & 15, k = keys 4 g .. 4 g + 3 of the tile)
    H8 KB[NKT][NG], VB[NKT][NG], KA[NKT][NCT];
    f32x4 dKT[NKT][NCT], dVT[NKT][NCT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const int kr = keyrow[16 * kt + fr];
      const float *kp = QKV + (ep + max(kr, 0)) * 3 * d;
#pragma unroll
      for (int kg = 0; kg < NG; ++kg) {
        const f32x4 kv = kr >= 0 ? *reinterpret_cast<const f32x4 *>(kp + kc + 16 * kg + 4 * fg) : z4;
        const f32x4 vv = kr >= 0 ? *reinterpret_cast<const f32x4 *>(kp + vc + 16 * kg + 4 * fg) : z4;
        KB[kt][kg] = split4(kv); VB[kt][kg] = split4(vv);
      }
      f32x4 ka[NCT];
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        const int kr2 = keyrow[16 * kt + 4 * fg + sp];
        const float *kp2 = QKV + (ep + max(kr2, 0)) * 3 * d + kc;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) ka[ct][sp] = kr2 >= 0 ? kp2[16 * ct + fr] : 0.f;
      }
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) KA[kt][ct] = split4(ka[ct]);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) { dKT[kt][ct] = z4; dVT[kt][ct] = z4; }
    }
    // ---- row tiles -----------------------------------------------------------------------------------------------------------------
    for (int r0 = 0; r0 < npad; r0 += 16) {
      // stage Q (scaled), dO, O of the 16 rows: a wave instruction moves 4 rows of HD floats (HD = 32: 8 lanes per row)
      constexpr int LPR = HD / 4, RPI = 64 / LPR;       // lanes per row, rows per instruction
#pragma unroll
      for (int i = 0; i < 16 / RPI; ++i) {
        const int rr = RPI * i + lane / LPR, c4 = 4 * (lane % LPR), row = r0 + rr;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f), go = q, oo = q;
        if (row < g.N) {
          q = *reinterpret_cast<const float4 *>(QKV + (ep + row) * 3 * d + qc + c4);
          go = *reinterpret_cast<const float4 *>(dA + (ep + row) * d + qc + c4);
          oo = *reinterpret_cast<const float4 *>(Aout + (ep + row) * d + qc + c4);
        }
        q.x *= scale; q.y *= scale; q.z *= scale; q.w *= scale;
        go.x *= gs; go.y *= gs; go.z *= gs; go.w *= gs;
        *reinterpret_cast<float4 *>(Qs + rr * PT + c4) = q;
        *reinterpret_cast<float4 *>(Gs + rr * PT + c4) = go;
        *reinterpret_cast<float4 *>(Os + rr * PT + c4) = oo;
      }
      // (the wave's own LDS traffic is ordered: no barrier)
      // delta of row fr: partial over the channels of lane group g, then over the groups; to St[32 + row]
      {
        float dl = 0.f;
#pragma unroll
        for (int c = 0; c < HD / 4; ++c) dl = fmaf(Gs[fr * PT + fg * (HD / 4) + c], Os[fr * PT + fg * (HD / 4) + c], dl);
        dl += __shfl_xor(dl, 16, 64);
        dl += __shfl_xor(dl, 32, 64);
        if (fg == 0) St[32 + fr] = dl;
      }
      // operands of the tile
      H8 QA[NG], GA[NG], QTA[NCT], GTA[NCT];
#pragma unroll
      for (int kg = 0; kg < NG; ++kg) {
        QA[kg] = split4(*reinterpret_cast<const f32x4 *>(Qs + fr * PT + 16 * kg + 4 * fg));
        GA[kg] = split4(*reinterpret_cast<const f32x4 *>(Gs + fr * PT + 16 * kg + 4 * fg));
      }
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        f32x4 qt, gt;
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) { qt[sp] = Qs[(4 * fg + sp) * PT + 16 * ct + fr]; gt[sp] = Gs[(4 * fg + sp) * PT + 16 * ct + fr]; }
        QTA[ct] = split4(qt); GTA[ct] = split4(gt);
      }
      const int4 nv = *reinterpret_cast<const int4 *>(nvis + r0 + 4 * fg);
      const f32x4 dl4 = *reinterpret_cast<const f32x4 *>(St + 32 + 4 * fg);
      // S and dP of every key tile
      f32x4 S[NKT], dP[NKT];
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        S[kt] = z4; dP[kt] = z4;
#pragma unroll
        for (int kg = 0; kg < NG; ++kg) { mfma3(S[kt], QA[kg], KB[kt][kg]); mfma3(dP[kt], GA[kg], VB[kt][kg]); }
      }
      // softmax over the visible keys of each row (rows 4 g + r on the registers, keys on the 16 lanes of the group and the tiles)
      f32x4 mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const int key = 16 * kt + fr;
        S[kt][0] = key < nv.x ? S[kt][0] : -INFINITY; S[kt][1] = key < nv.y ? S[kt][1] : -INFINITY;
        S[kt][2] = key < nv.z ? S[kt][2] : -INFINITY; S[kt][3] = key < nv.w ? S[kt][3] : -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) mx[r] = fmaxf(mx[r], S[kt][r]);
      }
      f32x4 sum = z4;
#pragma unroll
      for (int r = 0; r < 4; ++r) mx[r] = group16_max(mx[r]);
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = mx[r] == -INFINITY ? 0.f : __expf(S[kt][r] - mx[r]);      // (padding rows see no key: all zero)
          S[kt][r] = e;
          sum[r] += e;
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float l = group16_sum(sum[r]); sum[r] = l > 0.f ? 1.f / l : 0.f; }
      // P, dS;  dV^T += dO^T P,  dK^T += Q^T dS;  dQ^T += K^T dS^T through the transpose slot
      f32x4 dQT[NCT];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) dQT[ct] = z4;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        f32x4 P, dS;
#pragma unroll
        for (int r = 0; r < 4; ++r) { P[r] = S[kt][r] * sum[r]; dS[r] = P[r] * (dP[kt][r] - dl4[r]); }
        const H8 PS = split4(P), dSS = split4(dS);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) { mfma3(dVT[kt][ct], GTA[ct], PS); mfma3(dKT[kt][ct], QTA[ct], dSS); }
        // transpose dS: write [row 4 g + r][key fr], read [row fr][keys 4 g .. 4 g + 3]
#pragma unroll
        for (int r = 0; r < 4; ++r) Ts[(4 * fg + r) * 16 + fr] = dS[r];
        const H8 dST = split4(*reinterpret_cast<const f32x4 *>(Ts + fr * 16 + 4 * fg));
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) mfma3(dQT[ct], KA[kt][ct], dST);
      }
      // dQ of the tile (channels 16 ct + 4 g + r of row fr), zeros for the K / V gradient slices of the rows that are not keys
      const int row = r0 + fr;
      if (row < g.N) {
        float *out = dQKV + (ep + row) * 3 * d;
        const bool notkey = kidx[row] < 0;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const f32x4 v = dQT[ct] * (scale * ginv);
          track(v);
          *reinterpret_cast<f32x4 *>(out + qc + 16 * ct + 4 * fg) = v;
          if (notkey && !key_rows_only) {      // (key_rows_only: the in-projection's products read the K | V slices of the key rows alone)
            *reinterpret_cast<f32x4 *>(out + kc + 16 * ct + 4 * fg) = z4;
            *reinterpret_cast<f32x4 *>(out + vc + 16 * ct + 4 * fg) = z4;
          }
        }
      }
    }
    // ---- dK / dV of the head's key rows (Q was staged scaled: dK carries the 1 / sqrt(hd)) ------------------------------------------
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const int kr = keyrow[16 * kt + fr];
      if (kr >= 0) {
        float *out = dQKV + (ep + kr) * 3 * d;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const f32x4 dkv = dKT[kt][ct] * ginv, dvv = dVT[kt][ct] * ginv;
          track(dkv); track(dvv);
          *reinterpret_cast<f32x4 *>(out + kc + 16 * ct + 4 * fg) = dkv;
          *reinterpret_cast<f32x4 *>(out + vc + 16 * ct + 4 * fg) = dvv;
        }
      }
    }
  }
  if (out_absmax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) omax = max(omax, (unsigned)__shfl_xor((int)omax, o, 64));
    if (lane == 0 && omax) atomicMax(out_absmax, omax);
  }
}

}  // namespace abww
