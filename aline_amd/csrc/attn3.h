// Masked set-attention of the generic pipeline on the matrix pipe (model/encoder.py:8-46), for the reference-precision
// f16x3 mode and the bf16 modes at head_dim 32 / 64 (cfg5: d = 512, 8 heads of 64): the VALU kernel (kernels.h
// attention_kernel) runs at the fp32 vector peak there -- 95 us per call at cfg5 -- and is 15 % of that rollout.
// Every product is the 3-term f16 split of x3.h (fp32-grade).  One workgroup per (episode, head), as attention_kernel:
// K of the key rows as A fragments and V^T fragments are staged in LDS once, then each wave takes 16-row token tiles:
// S^T = K Q^T, softmax over the keys of a token (exp2 domain), O^T = V^T P.  Inputs are the fp32 rows the GEMMs wrote:
// Q [M, d] and the compact K | V rows [B * max_keys, 2 d] of the key list (key_list_kernel), up to 64 keys.
#pragma once
#include "x3.h"

namespace attn3 {

using img::group_sum4;
using img::u32x4;
using x3::f16x8;
using x3::group_max4;
using x3::mfma3;
using x3::split2;
using x3::split_frag;

constexpr int MAX_KT = 4;     // key tiles of 16: up to 64 keys

template <int HD>
__global__ __launch_bounds__(256) void attention_kernel(Geo g, int d, const float *__restrict__ Q, const float *__restrict__ KVc,
                                                        const int *__restrict__ kcnt, float *__restrict__ Aout, int max_keys) {
  constexpr int NKS = HD / 32, NCT = HD / 16;
  __shared__ __attribute__((aligned(16))) u32x4 Kf[MAX_KT * NKS * 2 * 64];          // [kt][ks][hi | lo][lane]
  __shared__ __attribute__((aligned(16))) u32x4 Vf[NCT * (MAX_KT / 2) * 2 * 64];    // [channel tile][key block of 32][hi | lo][lane]
  const int H = d / HD, b = (blockIdx.x / (8 * H)) * 8 + blockIdx.x % 8, h = (blockIdx.x / 8) % H;
  if (b >= g.B) return;
  const int tid = threadIdx.x, lane = tid & 63, tok = lane & 15, gq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_ck = kcnt[2 * b], n_ak = kcnt[2 * b + 1];
  const int nkt = (n_ak + 15) >> 4, nkb = (nkt + 1) >> 1;
  const float *kv = KVc + (long)b * max_keys * 2 * d + h * HD;      // key j: K at kv + j * 2 d, V at + d
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  // K fragments: lane (key, g) holds channels 32 ks + 4 g + (0..3) and 32 ks + 16 + 4 g + (0..3) of its key row
  for (int job = wave; job < nkt * NKS; job += 4) {
    const int kt = job / NKS, ks = job - kt * NKS, key = 16 * kt + tok;
    f32x4 lo4 = z4, hi4 = z4;
    if (key < n_ak) {
      const float *p = kv + (long)key * 2 * d + 32 * ks + 4 * gq;
      lo4 = *reinterpret_cast<const f32x4 *>(p); hi4 = *reinterpret_cast<const f32x4 *>(p + 16);
    }
    f16x8 fh, fl;
    split_frag(lo4, hi4, fh, fl);
    Kf[((kt * NKS + ks) * 2) * 64 + lane] = __builtin_bit_cast(u32x4, fh);
    Kf[((kt * NKS + ks) * 2 + 1) * 64 + lane] = __builtin_bit_cast(u32x4, fl);
  }
  // V^T fragments: lane (c, g) holds V[key 32 kb + 16 (j >> 2) + 4 g + (j & 3)][channel 16 i + c], j = 0..7
  for (int job = wave; job < NCT * nkb; job += 4) {
    const int i = job / nkb, kb = job - i * nkb;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int key = 32 * kb + 16 * (j >> 2) + 4 * gq + (j & 3);
      v[j] = key < n_ak ? kv[(long)key * 2 * d + d + 16 * i + tok] : 0.f;
    }
    unsigned hh[4], ll[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) split2(v[2 * w], v[2 * w + 1], hh[w], ll[w]);
    Vf[((i * (MAX_KT / 2) + kb) * 2) * 64 + lane] = (u32x4){hh[0], hh[1], hh[2], hh[3]};
    Vf[((i * (MAX_KT / 2) + kb) * 2 + 1) * 64 + lane] = (u32x4){ll[0], ll[1], ll[2], ll[3]};
  }
  __syncthreads();
  const float scale = rsqrtf((float)HD) * 1.44269504088896340736f;   // softmax runs in exp2
  const long ep = (long)b * g.N;
  const int ntiles = (g.N + 15) >> 4;
  for (int tile = wave; tile < ntiles; tile += 4) {
    const int r = 16 * tile + tok, rc = min(r, g.N - 1);
    const bool isq = rc < g.P && !is_ctx(g, b, rc);
    const int nv4 = (isq ? n_ak : n_ck) - 4 * gq;
    f16x8 qh[NKS], ql[NKS];
    {
      const float *qp = Q + (ep + rc) * d + h * HD + 4 * gq;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const f32x4 lo4 = *reinterpret_cast<const f32x4 *>(qp + 32 * ks) * scale, hi4 = *reinterpret_cast<const f32x4 *>(qp + 32 * ks + 16) * scale;
        split_frag(lo4, hi4, qh[ks], ql[ks]);
      }
    }
    f32x4 s[MAX_KT];
#pragma unroll
    for (int kt = 0; kt < MAX_KT; ++kt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) s[kt][q] = (16 * kt + q) < nv4 ? 0.f : -INFINITY;
      if (kt < nkt) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const f16x8 kh = __builtin_bit_cast(f16x8, Kf[((kt * NKS + ks) * 2) * 64 + lane]);
          const f16x8 kl = __builtin_bit_cast(f16x8, Kf[((kt * NKS + ks) * 2 + 1) * 64 + lane]);
          mfma3(s[kt], kh, kl, qh[ks], ql[ks]);
        }
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < MAX_KT; ++kt) mx = fmaxf(mx, fmaxf(fmaxf(s[kt][0], s[kt][1]), fmaxf(s[kt][2], s[kt][3])));
    mx = group_max4(mx, x3::opaque_inf());
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < MAX_KT; ++kt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s[kt][q] = __builtin_amdgcn_exp2f(s[kt][q] - mx);
        sum += s[kt][q];
      }
    const float inv = __builtin_amdgcn_rcpf(group_sum4(sum));
    f32x4 o[NCT];
#pragma unroll
    for (int i = 0; i < NCT; ++i) o[i] = z4;
#pragma unroll
    for (int kb = 0; kb < MAX_KT / 2; ++kb) {
      if (kb < nkb) {
        f16x8 ph, pl;
        split_frag(s[2 * kb], s[2 * kb + 1], ph, pl);
#pragma unroll
        for (int i = 0; i < NCT; ++i) {
          const f16x8 vh = __builtin_bit_cast(f16x8, Vf[((i * (MAX_KT / 2) + kb) * 2) * 64 + lane]);
          const f16x8 vl = __builtin_bit_cast(f16x8, Vf[((i * (MAX_KT / 2) + kb) * 2 + 1) * 64 + lane]);
          mfma3(o[i], vh, vl, ph, pl);
        }
      }
    }
    if (r < g.N) {
      float *op = Aout + (ep + r) * d + h * HD + 4 * gq;
#pragma unroll
      for (int i = 0; i < NCT; ++i) *reinterpret_cast<f32x4 *>(op + 16 * i) = o[i] * inv;
    }
  }
}

}  // namespace attn3
