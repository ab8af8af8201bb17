// Fused T-step rollout for the small-width model (d=32, F=128, H=4: config/encoder/encoder.yaml).
//
// One workgroup (4 waves) owns one episode for the WHOLE acquisition loop: embeddings, encoder,
// acquisition head, design selection, context update and the GMM posterior never leave the CU.
//   * token rows live in registers in the MFMA accumulator layout, TRANSPOSED: an activation tile
//     is X^T [features x 16 tokens]; lane (tok = lane & 15, g = lane >> 4) holds features
//     16*mt + 4*g + r (mt = acc tile, r = register).  Every linear layer is Y^T = W X^T, so the
//     accumulator of one product is, register for register, the B operand of the next one: the
//     k index of a 32-feature block is permuted as  pi(kb, g, j) = 32 kb + 16 (j>>2) + 4 g + (j&3)
//     and the weight (A operand) fragments are pre-permuted the same way at pack time.  No LDS
//     round trip, no cross-lane movement between layers; LayerNorm / softmax reduce over the 4
//     lane groups with two xor-shuffles.
//   * nobody attends to query rows (model/encoder.py:107,121): per layer the K/V of the <= 32 key
//     rows (context + selected targets) are computed once per episode by a pre-pass and kept in
//     LDS as head-block-structured MFMA fragments; every token tile then does
//     S^T = Kblk Q^T, softmax over keys (register axis), O^T = Vblk P^T.
//   * per-layer weights stream from a packed, L2-resident image into LDS (48 KB fp32 / layer).
//   * the static-slot state (role per slot, E = x-embedding (+ y-embedding once a point joins
//     the context)) lives in LDS across the T steps; design selection and the update that
//     replaces Task.update_batch (tasks/base_task.py:133-154) run in-kernel.
// Arithmetic: fp32 MFMA (v_mfma_f32_16x16x4_f32), the reference-precision mode.
#pragma once
#include "common.h"

namespace fused {

constexpr int D = 32, F = 128, H = 4, HD = 8;
constexpr int NKMAX = 32;       // key rows (context + selected targets) per episode
constexpr int MAXROWS = 256;    // token rows per episode (16 tiles)
constexpr int MAXNT = 8;        // target rows
constexpr int ES = 36;          // LDS row stride (floats) of E / Xk / Zt
constexpr int FRAG = 512;       // floats per operand fragment: 64 lanes x 8 k-elements
constexpr int NFRAG_LAYER = 24; // Wq 2, Wk 2, Wv 2, Wo 2, W1 8, W2 8
constexpr int FQ = 0, FK = 2, FV = 4, FO = 6, F1 = 8, F2 = 16;
// per-layer parameter block (floats) after the fragments
constexpr int PB_Q = 0, PB_K = 32, PB_V = 64, PB_O = 96, PB_1 = 128, PB_2 = 256, PLN1W = 288,
              PLN1B = 320, PLN2W = 352, PLN2B = 384, PARAMS = 416;
constexpr int LAYER_FLOATS = NFRAG_LAYER * FRAG + PARAMS;     // 12704
// acquisition head image: W1 8 fragments + b1[128] + w2[128] + b2 (padded to 4)
constexpr int HEAD_FLOATS = 8 * FRAG + 128 + 128 + 4;          // 4356

// ---- weight packing (once per rollout; weights are constant during a rollout) -------------------
struct PackArgs {
  int L;
  const float *in_proj_w[8], *in_proj_b[8], *out_proj_w[8], *out_proj_b[8], *lin1_w[8], *lin1_b[8],
      *lin2_w[8], *lin2_b[8], *n1w[8], *n1b[8], *n2w[8], *n2b[8];
  const float *acq_w1, *acq_b1, *acq_w2, *acq_b2;
  float *out;   // [L * LAYER_FLOATS + HEAD_FLOATS]
};

// element (lane, j) of fragment (mt, kb) of a [rows, K] row-major weight: W[16 mt + (lane & 15)][pi]
__device__ __forceinline__ float frag_elem(const float *W, int K, int mt, int kb, int e) {
  const int sub = e >> 8, lane = (e >> 2) & 63, jj = e & 3;   // image = [sub-plane][lane][4]
  const int j = sub * 4 + jj, g = lane >> 4;
  const int k = 32 * kb + 16 * (j >> 2) + 4 * g + (j & 3);
  return W[(16 * mt + (lane & 15)) * K + k];
}

__global__ void pack_weights_kernel(PackArgs a) {
  const int total = a.L * LAYER_FLOATS + HEAD_FLOATS;
  // 1/sqrt(hd) and log2(e) folded into Wq, bq: the kernel's softmax is exp2(s - max)
  const float qscale = rsqrtf((float)HD) * 1.44269504088896340736f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (i < a.L * LAYER_FLOATS) {
      const int l = i / LAYER_FLOATS, o = i % LAYER_FLOATS;
      if (o < NFRAG_LAYER * FRAG) {
        const int f = o / FRAG, e = o % FRAG;
        if (f < FO) {            // in_proj rows: q 0..31, k 32..63, v 64..95   (K = 32)
          const int which = f >> 1, mt = f & 1;
          v = frag_elem(a.in_proj_w[l] + which * D * D, D, mt, 0, e);
          if (which == 0) v *= qscale;
        } else if (f < F1) {
          v = frag_elem(a.out_proj_w[l], D, f - FO, 0, e);
        } else if (f < F2) {
          v = frag_elem(a.lin1_w[l], D, f - F1, 0, e);          // [128, 32]: 8 m-tiles
        } else {
          const int q = f - F2;                                  // [32, 128]: (mt, kb) = (q / 4, q % 4)
          v = frag_elem(a.lin2_w[l], F, q >> 2, q & 3, e);
        }
      } else {
        const int p = o - NFRAG_LAYER * FRAG;
        if (p < PB_K) v = a.in_proj_b[l][p] * qscale;
        else if (p < PB_O) v = a.in_proj_b[l][p];                // k, v biases (offsets 32..95)
        else if (p < PB_1) v = a.out_proj_b[l][p - PB_O];
        else if (p < PB_2) v = a.lin1_b[l][p - PB_1];
        else if (p < PLN1W) v = a.lin2_b[l][p - PB_2];
        else if (p < PLN1B) v = a.n1w[l][p - PLN1W];
        else if (p < PLN2W) v = a.n1b[l][p - PLN1B];
        else if (p < PLN2B) v = a.n2w[l][p - PLN2W];
        else v = a.n2b[l][p - PLN2B];
      }
    } else {
      const int o = i - a.L * LAYER_FLOATS;
      if (o < 8 * FRAG) v = frag_elem(a.acq_w1, D, o / FRAG, 0, o % FRAG);
      else if (o < 8 * FRAG + 128) v = a.acq_b1[o - 8 * FRAG];
      else if (o < 8 * FRAG + 256) v = a.acq_w2[o - 8 * FRAG - 128];
      else if (o == 8 * FRAG + 256) v = a.acq_b2[0];
    }
    a.out[i] = v;
  }
}

// ---- device helpers -------------------------------------------------------------------------------
struct Frag { f32x4 lo, hi; };   // k-elements j = 0..3 / 4..7 of this lane

__device__ __forceinline__ Frag ld_frag(const float *base, int lane) {
  Frag f;
  f.lo = *reinterpret_cast<const f32x4 *>(base + lane * 4);
  f.hi = *reinterpret_cast<const f32x4 *>(base + 256 + lane * 4);
  return f;
}
__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

// acc += A(16 x 32) * B(32 x 16) as eight exact-fp32 16x16x4 MFMAs
__device__ __forceinline__ void mma_block(f32x4 &acc, const Frag &A, const Frag &B) {
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A.lo[j], B.lo[j], acc, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A.hi[j], B.hi[j], acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_half(f32x4 &acc, const f32x4 &A, const f32x4 &B) {
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j], B[j], acc, 0, 0, 0);
}

// reduce over the 4 lane groups (lanes l, l^16, l^32, l^48 hold the same token) with the gfx950
// row/half swaps (VALU, no LDS crossbar): v_permlane16_swap(x, x) leaves {x0,x0,x2,x2} and
// {x1,x1,x3,x3} (rows of 16 lanes), v_permlane32_swap(x, x) leaves {lo,lo} and {hi,hi}.
__device__ __forceinline__ float group_sum(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float group_max(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// LayerNorm over the 32 features of each token (eps 1e-5, biased variance): x <- LN(x) * w + b
__device__ __forceinline__ void layer_norm(f32x4 (&x)[2], const float *w, const float *b, int g) {
  float s = 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) s += x[mt][r];
  const float mean = group_sum(s) * (1.f / D);
  float ss = 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { float t = x[mt][r] - mean; ss = fmaf(t, t, ss); }
  const float rstd = rsqrtf(group_sum(ss) * (1.f / D) + 1e-5f);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const f32x4 wv = ld4(w + 16 * mt + 4 * g), bv = ld4(b + 16 * mt + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) x[mt][r] = (x[mt][r] - mean) * rstd * wv[r] + bv[r];
  }
}


#define MFMA4(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0)
__device__ __forceinline__ f32x4 zero4() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }

// One encoder layer for NT (1 or 2) token tiles of one wave (x[tile][acc tile]).  Independent
// accumulators (2 acc tiles x NT tiles) are interleaved MFMA by MFMA so that a 16x16x4 fp32 MFMA
// (32-cycle issue, 40-cycle dependent latency) never waits for its own accumulator; with NT = 2 every
// weight / K / V fragment read from LDS feeds both tiles.  nv[t]: keys visible to this lane's token.
template <int NT>
__device__ __forceinline__ void layer_tiles(f32x4 (&x)[NT][2], const float *Wl, const float *prm,
                                            const float *Kb, const float *Vb, int lane, int g,
                                            const int (&nv)[NT], bool two_kt) {
  // ---- q = Wq x + bq (scaled) ------------------------------------------------------------------------
  f32x4 q[NT][2];
  {
    const Frag w0 = ld_frag(Wl + FQ * FRAG, lane), w1 = ld_frag(Wl + (FQ + 1) * FRAG, lane);
    const f32x4 b0 = ld4(prm + PB_Q + 4 * g), b1 = ld4(prm + PB_Q + 16 + 4 * g);
#pragma unroll
    for (int t = 0; t < NT; ++t) { q[t][0] = b0; q[t][1] = b1; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int t = 0; t < NT; ++t) MFMA4(q[t][0], w0.lo[j], x[t][0][j]);
#pragma unroll
      for (int t = 0; t < NT; ++t) MFMA4(q[t][1], w1.lo[j], x[t][0][j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int t = 0; t < NT; ++t) MFMA4(q[t][0], w0.hi[j], x[t][1][j]);
#pragma unroll
      for (int t = 0; t < NT; ++t) MFMA4(q[t][1], w1.hi[j], x[t][1][j]);
    }
  }
  // ---- S^T = Kblk q for the 4 heads (only the head's 16-channel sub-plane is non-zero) ----------------
  f32x4 s[NT][H][2];
  {
    f32x4 kf[H][2];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      kf[h][0] = ld4(Kb + (h * 2 + 0) * 256 + lane * 4);
      kf[h][1] = two_kt ? ld4(Kb + (h * 2 + 1) * 256 + lane * 4) : zero4();
#pragma unroll
      for (int t = 0; t < NT; ++t) { s[t][h][0] = zero4(); s[t][h][1] = zero4(); }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int h = 0; h < H; ++h)
#pragma unroll
        for (int t = 0; t < NT; ++t) MFMA4(s[t][h][0], kf[h][0][j], q[t][h >> 1][j]);
    if (two_kt) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
          for (int t = 0; t < NT; ++t) MFMA4(s[t][h][1], kf[h][1][j], q[t][h >> 1][j]);
    }
  }
  // ---- masked softmax over the key axis (registers x lane groups); p stays unnormalised ---------------
  float inv[NT][H];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    bool ok[2][4];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) ok[kt][r] = (16 * kt + 4 * g + r) < nv[t];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[t][h][kt][r] = ok[kt][r] ? s[t][h][kt][r] : -INFINITY;
          mx = fmaxf(mx, s[t][h][kt][r]);
        }
      mx = group_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[t][h][kt][r] = __builtin_amdgcn_exp2f(s[t][h][kt][r] - mx);
          sum += s[t][h][kt][r];
        }
      inv[t][h] = __builtin_amdgcn_rcpf(group_sum(sum));
    }
  }
  // ---- O^T = Vblk P^T (head h feeds the 8 channels of acc tile h>>1), then normalise ------------------
  f32x4 o[NT][2];
  {
    f32x4 vf[H][2];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float *vp = Vb + ((h >> 1) * 2 + (h & 1)) * FRAG;
      vf[h][0] = ld4(vp + lane * 4);
      vf[h][1] = two_kt ? ld4(vp + 256 + lane * 4) : zero4();
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) { o[t][0] = zero4(); o[t][1] = zero4(); }
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int t = 0; t < NT; ++t) MFMA4(o[t][0], vf[e][0][j], s[t][e][0][j]);
#pragma unroll
        for (int t = 0; t < NT; ++t) MFMA4(o[t][1], vf[2 + e][0][j], s[t][2 + e][0][j]);
      }
    if (two_kt) {
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int t = 0; t < NT; ++t) MFMA4(o[t][0], vf[e][1][j], s[t][e][1][j]);
#pragma unroll
          for (int t = 0; t < NT; ++t) MFMA4(o[t][1], vf[2 + e][1][j], s[t][2 + e][1][j]);
        }
    }
    // rows 4g+r of acc tile mt belong to head 2 mt + (g >> 1)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const float f = (g >> 1) ? inv[t][2 * mt + 1] : inv[t][2 * mt];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[t][mt][r] *= f;
      }
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
  // ---- x = LN2(x1 + W2 relu(W1 x1 + b1) + b2), hidden streamed in 32-wide chunks ----------------------------
  {
    const f32x4 b0 = ld4(prm + PB_2 + 4 * g), b1 = ld4(prm + PB_2 + 16 + 4 * g);
#pragma unroll
    for (int t = 0; t < NT; ++t) { x[t][0] = b0 + x1[t][0]; x[t][1] = b1 + x1[t][1]; }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const Frag u0 = ld_frag(Wl + (F1 + 2 * kb) * FRAG, lane), u1 = ld_frag(Wl + (F1 + 2 * kb + 1) * FRAG, lane);
      const f32x4 hb0 = ld4(prm + PB_1 + 32 * kb + 4 * g), hb1 = ld4(prm + PB_1 + 32 * kb + 16 + 4 * g);
      f32x4 hd[NT][2];
#pragma unroll
      for (int t = 0; t < NT; ++t) { hd[t][0] = hb0; hd[t][1] = hb1; }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int t = 0; t < NT; ++t) MFMA4(hd[t][0], u0.lo[j], x1[t][0][j]);
#pragma unroll
        for (int t = 0; t < NT; ++t) MFMA4(hd[t][1], u1.lo[j], x1[t][0][j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int t = 0; t < NT; ++t) MFMA4(hd[t][0], u0.hi[j], x1[t][1][j]);
#pragma unroll
        for (int t = 0; t < NT; ++t) MFMA4(hd[t][1], u1.hi[j], x1[t][1][j]);
      }
      const Frag d0 = ld_frag(Wl + (F2 + kb) * FRAG, lane), d1 = ld_frag(Wl + (F2 + 4 + kb) * FRAG, lane);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) hd[t][mt][r] = fmaxf(hd[t][mt][r], 0.f);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int t = 0; t < NT; ++t) MFMA4(x[t][0], d0.lo[j], hd[t][0][j]);
#pragma unroll
        for (int t = 0; t < NT; ++t) MFMA4(x[t][1], d1.lo[j], hd[t][0][j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int t = 0; t < NT; ++t) MFMA4(x[t][0], d0.hi[j], hd[t][1][j]);
#pragma unroll
        for (int t = 0; t < NT; ++t) MFMA4(x[t][1], d1.hi[j], hd[t][1][j]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) layer_norm(x[t], prm + PLN2W, prm + PLN2B, g);
}

// acquisition MLP (model/head.py:27-33) for NT tiles: logit = w2 . relu(W1 z + b1) + b2
template <int NT>
__device__ __forceinline__ void acq_tiles(const f32x4 (&z)[NT][2], const float *Wl, int lane, int g,
                                          float (&lg)[NT]) {
  const float *hb1 = Wl + 8 * FRAG, *hw2 = hb1 + 128, *hb2 = hw2 + 128;
  float p[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) p[t] = 0.f;
#pragma unroll
  for (int mp = 0; mp < 4; ++mp) {
    const Frag u0 = ld_frag(Wl + (2 * mp) * FRAG, lane), u1 = ld_frag(Wl + (2 * mp + 1) * FRAG, lane);
    const f32x4 hb0 = ld4(hb1 + 32 * mp + 4 * g), hbb = ld4(hb1 + 32 * mp + 16 + 4 * g);
    const f32x4 w20 = ld4(hw2 + 32 * mp + 4 * g), w21 = ld4(hw2 + 32 * mp + 16 + 4 * g);
    f32x4 hd[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) { hd[t][0] = hb0; hd[t][1] = hbb; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int t = 0; t < NT; ++t) MFMA4(hd[t][0], u0.lo[j], z[t][0][j]);
#pragma unroll
      for (int t = 0; t < NT; ++t) MFMA4(hd[t][1], u1.lo[j], z[t][0][j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int t = 0; t < NT; ++t) MFMA4(hd[t][0], u0.hi[j], z[t][1][j]);
#pragma unroll
      for (int t = 0; t < NT; ++t) MFMA4(hd[t][1], u1.hi[j], z[t][1][j]);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p[t] = fmaf(fmaxf(hd[t][0][r], 0.f), w20[r], p[t]);
        p[t] = fmaf(fmaxf(hd[t][1][r], 0.f), w21[r], p[t]);
      }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) lg[t] = group_sum(p[t]) + hb2[0];
}

struct RolloutArgs {
  int B, P, n_ctx0, n_th, T, L, C;
  const float *wpack;           // packed layer + head images
  const float *Ex, *Ey;         // [B, P, 32] cached point embeddings (rollout_init)
  const float *theta_tokens;    // [n_th, 32]
  const uint8_t *tmask;         // [n_th] or null
  const float *target_all;      // [B, n_th] or null
  const float *gmm_w1[16], *gmm_b1[16], *gmm_w2[16], *gmm_b2[16];
  float std_min;
  int mode;                     // ALINE_SELECT_*
  const float *uniform;         // [T, B]
  const int64_t *forced;        // [B, T]
  int *role;                    // [B, P] out (final roles, for export)
  int64_t *idx; int *slot; float *log_prob;           // [B, T]
  float *target_ll;                                   // [T, B, n_th]
  float *zt;                                          // [T, B, P - n_ctx0] or null
  float *post_mean, *post_std, *post_weight;          // [T, B, n_th, C] or null
  int stagger_sleeps;           // s_sleep(127) iterations (8128 cycles each) for odd residency slots
  unsigned long long *stamps;   // diagnostic build only: per-phase cycle sums [8 waves x 16]
};

// LDS carve (floats).  Everything lives in one dynamic array (16-byte aligned carve offsets).
constexpr int L_W = 0;                               // layer image / head image
constexpr int L_KB = L_W + LAYER_FLOATS;             // Kblk: 8 fragments x 256 floats (one sub-plane each)
constexpr int L_VB = L_KB + 8 * 256;                 // Vblk: 4 fragments x 512 floats
constexpr int L_XK = L_VB + 4 * FRAG;                // Xk [NKMAX][ES]
constexpr int L_ZT = L_XK + NKMAX * ES;              // Zt [MAXNT][ES]
constexpr int L_LOGIT = L_ZT + MAXNT * ES;           // logits / probs [MAXROWS]
constexpr int L_RAW = L_LOGIT + MAXROWS;             // GMM raw outputs [MAXNT][16][4]
constexpr int L_INT = L_RAW + MAXNT * 16 * 4;        // ints: role[MAXROWS], kidx[MAXROWS], qslot[MAXROWS], misc[16]
constexpr int L_TOTAL = L_INT + 3 * MAXROWS + 16;
constexpr size_t LDS_BYTES = (size_t)L_TOTAL * 4;

// Diagnostic stamps (template STAMP = true builds only; never in the shipped instantiation): wave w
// of workgroup 0 accumulates s_memtime deltas per phase into a.stamps[w * 16 + phase].
#define STAMP_PHASE(ph)                                                           \
  if constexpr (STAMP) {                                                          \
    unsigned long long _t;                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");    \
    if (blockIdx.x == 0 && lane == 0) a.stamps[wave * 16 + (ph)] += _t - t_prev;  \
    t_prev = _t;                                                                  \
  }

template <bool STAMP>
__global__ __launch_bounds__(256, 2) void rollout_f32_kernel(RolloutArgs a) {
  unsigned long long t_prev = 0;
  if constexpr (STAMP) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory"); }
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *Wl = lds + L_W, *Kb = lds + L_KB, *Vb = lds + L_VB, *Xk = lds + L_XK,
        *Zt = lds + L_ZT, *logit = lds + L_LOGIT, *raw = lds + L_RAW;
  int *role = reinterpret_cast<int *>(lds + L_INT);
  int *kidx = role + MAXROWS, *qslot = kidx + MAXROWS, *misc = qslot + MAXROWS;
  // misc: 0 n_ck, 1 n_ak, 2.. wave counts

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tok = lane & 15, g = lane >> 4;
  const int P = a.P, n_th = a.n_th, N = P + n_th;
  const int ntiles = (N + 15) >> 4;
  const int zw = P - a.n_ctx0;
  // Balanced contiguous tile ranges: wave wv owns tiles [t0, t0 + tcnt), tcnt = ntiles/4 (+1 for the
  // first ntiles%4 waves).  The waves that own the extra tile rotate with the workgroup (and with its
  // residency slot) so that co-resident workgroups do not stack their heavy waves on one SIMD.
  const int wv = (wave + blockIdx.x + (blockIdx.x >> 8)) & 3;
  const int tcnt = ntiles / 4 + (wv < (ntiles & 3) ? 1 : 0);
  const int t0 = wv * (ntiles / 4) + min(wv, ntiles & 3);

  // ---- episode state: roles.  The step-invariant embeddings stay in HBM/L2 (Ex, Ey: 26 KB per
  // episode and step): X^(0)[row] = Ex[row] (+ Ey[row] once the point has joined the context).
  for (int r = tid; r < MAXROWS; r += 256) role[r] = r < P ? (r < a.n_ctx0 ? r + 1 : 0) : -1;
  for (int i = tid; i < NKMAX * ES; i += 256) Xk[i] = 0.f;
  __syncthreads();

  // Two workgroups share a CU (and its matrix pipes).  They run the same program, so without help
  // they would sit in the MFMA-free phases (selection, GMM, weight streaming) at the same time.
  // Workgroups of the second residency slot (dispatch is round-robin over 256 CUs) start half a
  // step late, so one workgroup's MFMA-dense layer passes cover the other's scalar phases.  Speed
  // only: nothing depends on the placement.
  if (a.stagger_sleeps > 0 && ((blockIdx.x >> 8) & 1)) {
    for (int i = 0; i < a.stagger_sleeps; ++i) __builtin_amdgcn_s_sleep(127);
  }

  for (int t = 0; t < a.T; ++t) {
    // ---- key list: context slots in slot order, then the selected targets ----------------------
    {
      const int r = tid;                                   // MAXROWS == blockDim
      const bool ck = r < P && role[r] > 0;
      const unsigned long long bal = __ballot(ck);
      if (lane == 0) misc[2 + wave] = __popcll(bal);
      __syncthreads();
      int off = 0;
      for (int w = 0; w < wave; ++w) off += misc[2 + w];
      const int nck = misc[2] + misc[3] + misc[4] + misc[5];
      int k = ck ? off + __popcll(bal & ((1ull << lane) - 1ull)) : -1;
      if (r >= P && r < N) {
        int nsel = 0, mine = -1;
        for (int j = 0; j < n_th; ++j) {
          const bool sel = !a.tmask || a.tmask[j];
          if (j == r - P && sel) mine = nsel;
          nsel += sel ? 1 : 0;
        }
        if (mine >= 0) k = nck + mine;
        if (r == P) { misc[0] = nck; misc[1] = nck + nsel; }
      }
      kidx[r] = k;
      __syncthreads();
    }
    const int n_ck = misc[0], n_ak = misc[1];
    STAMP_PHASE(0)   // key list

    // ---- X^(0): token tiles of this wave from E (tile ti = wave + 4 i) ----------------------------
    f32x4 x[4][2];   // [local tile][acc tile]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 16 * (t0 + i) + tok;
      f32x4 v0 = zero4(), v1 = zero4();
      if (i < tcnt && row < N) {
        const float *src = row < P ? a.Ex + ((long)b * P + row) * D : a.theta_tokens + (long)(row - P) * D;
        v0 = ld4(src + 4 * g);
        v1 = ld4(src + 16 + 4 * g);
        if (row < P && role[row] > 0) {
          const float *ey = a.Ey + ((long)b * P + row) * D;
          v0 += ld4(ey + 4 * g);
          v1 += ld4(ey + 16 + 4 * g);
        }
        const int k = kidx[row];
        if (k >= 0) {
          *reinterpret_cast<f32x4 *>(Xk + k * ES + 4 * g) = v0;
          *reinterpret_cast<f32x4 *>(Xk + k * ES + 16 + 4 * g) = v1;
        }
      }
      x[i][0] = v0;
      x[i][1] = v1;
    }
    STAMP_PHASE(1)   // x0 load
    for (int l = 0; l < a.L; ++l) {
      // ---- stream layer l's packed image into LDS ------------------------------------------------
      {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.wpack + (long)l * LAYER_FLOATS);
        f32x4 *dst = reinterpret_cast<f32x4 *>(Wl);
        for (int i = tid; i < LAYER_FLOATS / 4; i += 256) dst[i] = src[i];
      }
      __syncthreads();   // weights + Xk visible
      STAMP_PHASE(2)   // weight stream + barrier
      const float *prm = Wl + NFRAG_LAYER * FRAG;

      // ---- pre-pass: K^T (waves 0,1) and V (waves 2,3) of key tile kt = wave & 1 ------------------
      {
        const int kt = wave & 1;
        if (kt == 0 || n_ak > 16) {
          const int key = 16 * kt + tok;
          Frag xf;
          xf.lo = ld4(Xk + key * ES + 4 * g);
          xf.hi = ld4(Xk + key * ES + 16 + 4 * g);
          if (wave < 2) {
            // K^T[c, key] = Wk x_key + bk : rows = channels
            f32x4 kacc[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
              kacc[mt] = ld4(prm + PB_K + 16 * mt + 4 * g);
              mma_block(kacc[mt], ld_frag(Wl + (FK + mt) * FRAG, lane), xf);
            }
            // head-block fragments: head h uses sub-plane h>>1, lanes with (g>>1) == (h&1)
#pragma unroll
            for (int h = 0; h < H; ++h) {
              f32x4 v = ((g >> 1) == (h & 1)) ? kacc[h >> 1] : (f32x4){0.f, 0.f, 0.f, 0.f};
              *reinterpret_cast<f32x4 *>(Kb + (h * 2 + kt) * 256 + lane * 4) = v;
            }
          } else {
            // V[key, c] = x_key Wv^T + bv : rows = keys, lane column = channel
            f32x4 vacc[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              const float bv = prm[PB_V + 16 * nt + tok];
              vacc[nt] = (f32x4){bv, bv, bv, bv};
              mma_block(vacc[nt], xf, ld_frag(Wl + (FV + nt) * FRAG, lane));
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                f32x4 v = ((tok >> 3) == e) ? vacc[nt] : (f32x4){0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4 *>(Vb + (nt * 2 + e) * FRAG + kt * 256 + lane * 4) = v;
              }
          }
        }
      }
      __syncthreads();   // K/V fragments visible
      STAMP_PHASE(3)   // pre-pass + barrier

      // ---- main pass: local tiles (0,1) as a pair, then (2,3) as a pair or tile 2 alone ------------
      {
        const bool two_kt = n_ak > 16;
        const bool publish = l + 1 < a.L;
        auto nvis = [&](int row) { return (row < P && role[row] == 0) ? n_ak : n_ck; };
        auto pub = [&](int row, const f32x4 &v0, const f32x4 &v1) {
          const int k = (publish && row < N) ? kidx[row] : -1;
          if (k >= 0) {
            *reinterpret_cast<f32x4 *>(Xk + k * ES + 4 * g) = v0;
            *reinterpret_cast<f32x4 *>(Xk + k * ES + 16 + 4 * g) = v1;
          }
        };
        const int rowb = 16 * t0 + tok;
        if (tcnt >= 2) {
          f32x4 xp[2][2] = {{x[0][0], x[0][1]}, {x[1][0], x[1][1]}};
          const int nv[2] = {nvis(rowb), nvis(rowb + 16)};
          layer_tiles<2>(xp, Wl, prm, Kb, Vb, lane, g, nv, two_kt);
          pub(rowb, xp[0][0], xp[0][1]); pub(rowb + 16, xp[1][0], xp[1][1]);
          x[0][0] = xp[0][0]; x[0][1] = xp[0][1]; x[1][0] = xp[1][0]; x[1][1] = xp[1][1];
        } else if (tcnt == 1) {
          f32x4 xp[1][2] = {{x[0][0], x[0][1]}};
          const int nv[1] = {nvis(rowb)};
          layer_tiles<1>(xp, Wl, prm, Kb, Vb, lane, g, nv, two_kt);
          pub(rowb, xp[0][0], xp[0][1]);
          x[0][0] = xp[0][0]; x[0][1] = xp[0][1];
        }
        if (tcnt == 4) {
          f32x4 xp[2][2] = {{x[2][0], x[2][1]}, {x[3][0], x[3][1]}};
          const int nv[2] = {nvis(rowb + 32), nvis(rowb + 48)};
          layer_tiles<2>(xp, Wl, prm, Kb, Vb, lane, g, nv, two_kt);
          pub(rowb + 32, xp[0][0], xp[0][1]); pub(rowb + 48, xp[1][0], xp[1][1]);
          x[2][0] = xp[0][0]; x[2][1] = xp[0][1]; x[3][0] = xp[1][0]; x[3][1] = xp[1][1];
        } else if (tcnt == 3) {
          f32x4 xp[1][2] = {{x[2][0], x[2][1]}};
          const int nv[1] = {nvis(rowb + 32)};
          layer_tiles<1>(xp, Wl, prm, Kb, Vb, lane, g, nv, two_kt);
          pub(rowb + 32, xp[0][0], xp[0][1]);
          x[2][0] = xp[0][0]; x[2][1] = xp[0][1];
        }
      }
      STAMP_PHASE(4)   // main pass (this wave's tiles)
      __syncthreads();   // everyone done with this layer's weights / K / V
      STAMP_PHASE(5)   // wait for the slowest wave
    }

    // ---- acquisition head (model/head.py:27-33) on every tile; z of the target rows -> Zt --------
    {
      const f32x4 *src = reinterpret_cast<const f32x4 *>(a.wpack + (long)a.L * LAYER_FLOATS);
      f32x4 *dst = reinterpret_cast<f32x4 *>(Wl);
      for (int i = tid; i < HEAD_FLOATS / 4; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    STAMP_PHASE(6)   // head image stream
    {
      auto emit = [&](int row, float lgt, const f32x4 &v0, const f32x4 &v1) {
        if (g == 0 && row < MAXROWS) logit[row] = lgt;
        if (row >= P && row < N) {
          *reinterpret_cast<f32x4 *>(Zt + (row - P) * ES + 4 * g) = v0;
          *reinterpret_cast<f32x4 *>(Zt + (row - P) * ES + 16 + 4 * g) = v1;
        }
      };
      const int rowb = 16 * t0 + tok;
      if (tcnt >= 2) {
        const f32x4 zp[2][2] = {{x[0][0], x[0][1]}, {x[1][0], x[1][1]}};
        float lg[2];
        acq_tiles<2>(zp, Wl, lane, g, lg);
        emit(rowb, lg[0], zp[0][0], zp[0][1]); emit(rowb + 16, lg[1], zp[1][0], zp[1][1]);
      } else if (tcnt == 1) {
        const f32x4 zp[1][2] = {{x[0][0], x[0][1]}};
        float lg[1];
        acq_tiles<1>(zp, Wl, lane, g, lg);
        emit(rowb, lg[0], zp[0][0], zp[0][1]);
      }
      if (tcnt == 4) {
        const f32x4 zp[2][2] = {{x[2][0], x[2][1]}, {x[3][0], x[3][1]}};
        float lg[2];
        acq_tiles<2>(zp, Wl, lane, g, lg);
        emit(rowb + 32, lg[0], zp[0][0], zp[0][1]); emit(rowb + 48, lg[1], zp[1][0], zp[1][1]);
      } else if (tcnt == 3) {
        const f32x4 zp[1][2] = {{x[2][0], x[2][1]}};
        float lg[1];
        acq_tiles<1>(zp, Wl, lane, g, lg);
        emit(rowb + 32, lg[0], zp[0][0], zp[0][1]);
      }
    }
    STAMP_PHASE(7)   // acquisition MLP
    __syncthreads();
    STAMP_PHASE(8)   // barrier after acquisition

    if (wave == 0) {
      // ---- softmax over the remaining queries + design selection (model/head.py:347-362) ---------
      int nq = 0;
      for (int c0 = 0; c0 < P; c0 += 64) {
        const int p = c0 + lane;
        const bool isq = p < P && role[p] == 0;
        const unsigned long long bal = __ballot(isq);
        if (isq) qslot[nq + __popcll(bal & ((1ull << lane) - 1ull))] = p;
        nq += __popcll(bal);
      }
      float mx = -INFINITY;
      for (int i = lane; i < nq; i += 64) mx = fmaxf(mx, logit[qslot[i]]);
      mx = wave_max(mx);
      float pv[4], sum = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int i = lane + 64 * c;
        pv[c] = i < nq ? __expf(logit[qslot[i]] - mx) : 0.f;
        sum += pv[c];
      }
      sum = wave_sum(sum);
      const float inv = 1.f / sum;
      float *prob = logit;   // compacted probabilities overwrite the logits (all reads done above)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        pv[c] *= inv;
        const int i = lane + 64 * c;
        if (i < nq) prob[i] = pv[c];
      }
      if (a.zt) {
        float *zo = a.zt + ((long)t * a.B + b) * zw;
        for (int i = lane; i < zw; i += 64) zo[i] = i < nq ? prob[i] : 0.f;
      }
      int choice = 0;
      float val = 0.f;
      if (a.mode == 0) {
        float best = -1.f; int bi = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int i = lane + 64 * c;
          if (i < nq && pv[c] > best) { best = pv[c]; bi = i; }
        }
        for (int o = 32; o > 0; o >>= 1) {
          const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
          if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        choice = bi; val = best;
      } else {
        float tot = wave_sum(pv[0] + pv[1] + pv[2] + pv[3]);
        if (a.mode == 2) {
          choice = (int)a.forced[(long)b * a.T + t];
          choice = min(max(choice, 0), nq - 1);
        } else {
          const float u = a.uniform[(long)t * a.B + b] * tot;
          float run = 0.f; int found = nq - 1; bool done = false;
          for (int c0 = 0; c0 < nq && !done; c0 += 64) {
            const int i = c0 + lane;
            float v = i < nq ? prob[i] : 0.f, incl = v;
            for (int o = 1; o < 64; o <<= 1) { const float tt = __shfl_up(incl, o, 64); if (lane >= o) incl += tt; }
            const bool hit = i < nq && (run + incl) > u;
            const unsigned long long bal = __ballot(hit);
            if (bal) { found = c0 + __ffsll((long long)bal) - 1; done = true; }
            run += __shfl(incl, 63, 64);
          }
          choice = found;
        }
        // Categorical(probs).log_prob: probs / probs.sum(), clamped to [eps, 1 - eps]
        val = fminf(fmaxf(prob[choice] / tot, 1.1920929e-07f), 1.f - 1.1920929e-07f);
      }
      const int sl = qslot[choice];
      const int order = (P - nq) + 1;
      if (lane == 0) {
        const long o = (long)b * a.T + t;
        if (a.idx) a.idx[o] = choice;
        if (a.slot) a.slot[o] = sl;
        if (a.log_prob) a.log_prob[o] = logf(val);
        role[sl] = order;
      }
      // the chosen point joins the context: from the next step on its row adds Ey (embedder.py:156)
    } else {
      // ---- GMM heads on the target rows (model/head.py:172-177), fp32 FMA: n_t rows are few -----
      for (int c = wave - 1; c < a.C; c += 3) {
        const float *w1 = a.gmm_w1[c], *b1 = a.gmm_b1[c], *w2 = a.gmm_w2[c];
        f32x4 wr[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int q4 = 0; q4 < 8; ++q4) wr[u][q4] = ld4(w1 + (lane + 64 * u) * D + 4 * q4);
        const float bb0 = b1[lane], bb1 = b1[lane + 64];
        float w2v[3][2];
#pragma unroll
        for (int j = 0; j < 3; ++j) { w2v[j][0] = w2[j * F + lane]; w2v[j][1] = w2[j * F + lane + 64]; }
        for (int r = 0; r < n_th; ++r) {
          float h0 = bb0, h1 = bb1;
#pragma unroll
          for (int q4 = 0; q4 < 8; ++q4) {
            const f32x4 zv = ld4(Zt + r * ES + 4 * q4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { h0 = fmaf(wr[0][q4][e], zv[e], h0); h1 = fmaf(wr[1][q4][e], zv[e], h1); }
          }
          h0 = fmaxf(h0, 0.f); h1 = fmaxf(h1, 0.f);
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const float sres = wave_sum(h0 * w2v[j][0] + h1 * w2v[j][1]);
            if (lane == 0) raw[(r * 16 + c) * 4 + j] = sres + a.gmm_b2[c][j];
          }
        }
      }
    }
    STAMP_PHASE(9)   // selection (wave 0) / GMM heads (waves 1-3)
    __syncthreads();
    STAMP_PHASE(10)  // barrier after selection / GMM

    // ---- GMM parameter maps + compute_ll (head.py:176-177, utils/eval.py:200-207) -------------------
    if (wave == 1) {
      for (int r = 0; r < n_th; ++r) {
        const bool act = lane < a.C;
        const float r0 = act ? raw[(r * 16 + lane) * 4 + 0] : 0.f;
        const float r1 = act ? raw[(r * 16 + lane) * 4 + 1] : 0.f;
        const float r2 = act ? raw[(r * 16 + lane) * 4 + 2] : -INFINITY;
        const float sd = softplus_f(r1) + a.std_min;
        const float m2 = wave_max(r2);
        const float e = act ? __expf(r2 - m2) : 0.f;
        const float wgt = e / wave_sum(e);
        const long orow = ((long)t * a.B + b) * n_th + r;
        if (act) {
          if (a.post_mean) a.post_mean[orow * a.C + lane] = r0;
          if (a.post_std) a.post_std[orow * a.C + lane] = sd;
          if (a.post_weight) a.post_weight[orow * a.C + lane] = wgt;
        }
        if (a.target_ll && a.target_all) {
          const float v = a.target_all[(long)b * n_th + r];
          const float z = (v - r0) / sd;
          const float lp = act ? (-0.5f * z * z - logf(sd) - 0.91893853320467274178f + logf(wgt)) : -INFINITY;
          const float m3 = wave_max(lp);
          const float se = wave_sum(act ? __expf(lp - m3) : 0.f);
          if (lane == 0) a.target_ll[orow] = m3 + logf(se);
        }
      }
    }
    // (the barrier at the top of the next step's key-list build orders role / E updates)
    __syncthreads();
    STAMP_PHASE(11)  // GMM epilogue + barrier
  }
  for (int r = tid; r < P; r += 256) a.role[(long)b * P + r] = role[r];
}

}  // namespace fused
