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
constexpr int FRAG = 512;       // floats per fp32 operand fragment: 64 lanes x 8 k-elements
constexpr int FRAG3 = 768;      // 32-bit words per split-bf16 fragment: 3 planes x 64 lanes x 8 bf16
constexpr int FQ = 0, FK = 2, FV = 4, FO = 6;        // fp32 fragments (attention projections)
constexpr int NF32 = 8;                               // Wq 2, Wk 2, Wv 2, Wo 2
constexpr int FFN_BASE = NF32 * FRAG;                 // then 16 split-bf16 fragments: W1 8, W2 8
constexpr int F1 = 0, F2 = 8;                         // indices into the FFN fragment block
constexpr int PRM_BASE = FFN_BASE + 16 * FRAG3;
// per-layer parameter block (floats) after the fragments
constexpr int PB_Q = 0, PB_K = 32, PB_V = 64, PB_O = 96, PB_1 = 128, PB_2 = 256, PLN1W = 288,
              PLN1B = 320, PLN2W = 352, PLN2B = 384, PARAMS = 416;
constexpr int LAYER_FLOATS = PRM_BASE + PARAMS;                 // 16800 (67.2 KB)
// acquisition head image: W1 8 split-bf16 fragments + b1[128] + w2[128] + b2 (padded to 4)
constexpr int HEAD_FLOATS = 8 * FRAG3 + 128 + 128 + 4;         // 6404

// ---- weight packing (once per rollout; weights are constant during a rollout) -------------------
struct PackArgs {
  int L;
  const float *in_proj_w[8], *in_proj_b[8], *out_proj_w[8], *out_proj_b[8], *lin1_w[8], *lin1_b[8],
      *lin2_w[8], *lin2_b[8], *n1w[8], *n1b[8], *n2w[8], *n2b[8];
  const float *acq_w1, *acq_b1, *acq_w2, *acq_b2;
  int C;                                        // side images (fused_side.h), after the head image:
  const float *gmm_w1[16];                      //   C x 8 split-bf16 fragments of the GMM first layers
  const float *x_w2, *y_w2;                     //   8 + 8 fragments of the point embedders' second layers
  float *out;   // [L * LAYER_FLOATS + HEAD_FLOATS + (C + 2) * 8 * FRAG3]
  int layers_only;   // 1: pack the L layer images only (layer_tail_kernel of the generic pipeline: no head pointers needed)
                     // 2: pack the C GMM first-layer images only, at out[0 ..)
};
constexpr int SIDE_FRAGS = 8 * FRAG3;          // one [128 x 32] or [32 x 128] weight as 8 fragments

// element (lane, j) of fragment (mt, kb) of a [rows, K] row-major weight: W[16 mt + (lane & 15)][pi]
__device__ __forceinline__ float frag_val(const float *W, int K, int mt, int kb, int lane, int j) {
  const int g = lane >> 4;
  const int k = 32 * kb + 16 * (j >> 2) + 4 * g + (j & 3);
  return W[(16 * mt + (lane & 15)) * K + k];
}
__device__ __forceinline__ float frag_elem(const float *W, int K, int mt, int kb, int e) {
  const int sub = e >> 8, lane = (e >> 2) & 63, jj = e & 3;   // fp32 image = [sub-plane][lane][4]
  return frag_val(W, K, mt, kb, lane, sub * 4 + jj);
}
// exact 3-way split of an fp32 value into bf16 planes: a == hi + mid + lo (8 + 8 + 8 mantissa bits)
__device__ __forceinline__ unsigned short split3_plane(float a, int plane) {
  const unsigned short h = f2bf(a);
  if (plane == 0) return h;
  const float r = a - bf2f(h);
  const unsigned short m = f2bf(r);
  if (plane == 1) return m;
  return f2bf(r - bf2f(m));
}
// 32-bit word e of a split-bf16 fragment image [plane][lane][4 words]: elements j = 2w, 2w + 1
__device__ __forceinline__ float frag3_word(const float *W, int K, int mt, int kb, int e) {
  const int plane = e >> 8, lane = (e >> 2) & 63, w = e & 3;
  const unsigned lo = split3_plane(frag_val(W, K, mt, kb, lane, 2 * w), plane);
  const unsigned hi = split3_plane(frag_val(W, K, mt, kb, lane, 2 * w + 1), plane);
  return __uint_as_float(lo | (hi << 16));
}

__global__ void pack_weights_kernel(PackArgs a) {
  if (a.layers_only == 2) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.C * SIDE_FRAGS; i += gridDim.x * blockDim.x) {
      const int img = i / SIDE_FRAGS, e = i % SIDE_FRAGS;
      a.out[i] = frag3_word(a.gmm_w1[img], D, e / FRAG3, 0, e % FRAG3);
    }
    return;
  }
  const int core = a.L * LAYER_FLOATS + HEAD_FLOATS;
  const int total = a.layers_only == 1 ? a.L * LAYER_FLOATS : core + (a.C + 2) * SIDE_FRAGS;
  // 1/sqrt(hd) and log2(e) folded into Wq, bq: the kernel's softmax is exp2(s - max)
  const float qscale = rsqrtf((float)HD) * 1.44269504088896340736f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (i < a.L * LAYER_FLOATS) {
      const int l = i / LAYER_FLOATS, o = i % LAYER_FLOATS;
      if (o < FFN_BASE) {
        const int f = o / FRAG, e = o % FRAG;
        if (f < FO) {            // in_proj rows: q 0..31, k 32..63, v 64..95   (K = 32)
          const int which = f >> 1, mt = f & 1;
          v = frag_elem(a.in_proj_w[l] + which * D * D, D, mt, 0, e);
          if (which == 0) v *= qscale;
        } else {
          v = frag_elem(a.out_proj_w[l], D, f - FO, 0, e);
        }
      } else if (o < PRM_BASE) {
        const int q = (o - FFN_BASE) / FRAG3, e = (o - FFN_BASE) % FRAG3;
        if (q < F2) v = frag3_word(a.lin1_w[l], D, q, 0, e);                       // [128, 32]: 8 m-tiles
        else v = frag3_word(a.lin2_w[l], F, (q - F2) >> 2, (q - F2) & 3, e);       // [32, 128]: (mt, kb)
      } else {
        const int p = o - PRM_BASE;
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
    } else if (i >= core) {
      const int o = i - core, img = o / SIDE_FRAGS, e = o % SIDE_FRAGS;
      if (img < a.C) v = frag3_word(a.gmm_w1[img], D, e / FRAG3, 0, e % FRAG3);                 // [128, 32]
      else v = frag3_word(img == a.C ? a.x_w2 : a.y_w2, F, (e / FRAG3) >> 2, (e / FRAG3) & 3, e % FRAG3);   // [32, 128]
    } else {
      const int o = i - a.L * LAYER_FLOATS;
      if (o < 8 * FRAG3) v = frag3_word(a.acq_w1, D, o / FRAG3, 0, o % FRAG3);
      else if (o < 8 * FRAG3 + 128) v = a.acq_b1[o - 8 * FRAG3];
      else if (o < 8 * FRAG3 + 256) v = a.acq_w2[o - 8 * FRAG3 - 128];
      else if (o == 8 * FRAG3 + 256) v = a.acq_b2[0];
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

// ---- split-bf16 ("bf16x6") products for the weight matmuls of the FFN and the acquisition MLP ---------
// An fp32 value is EXACTLY hi + mid + lo with three bf16 (8 + 8 + 8 mantissa bits).  A product of two
// such operands keeps the six terms down to 2^-16 relative (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo,
// lo*hi) -- what is dropped is <= 3 * 2^-24, the size of fp32 rounding itself -- on the bf16 matrix
// pipe: 6 x 16 cycles per 16x16x32 block instead of 8 x 32 cycles of v_mfma_f32_16x16x4_f32.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
struct Frag3 { bf16x8 p[3]; };

__device__ __forceinline__ Frag3 ld_frag3(const float *base, int lane) {
  Frag3 f;
#pragma unroll
  for (int p = 0; p < 3; ++p)
    f.p[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(base + p * 256 + lane * 4));
  return f;
}
// two fp32 values -> three packed bf16 pairs (v_cvt_pk_bf16_f32 rounds to nearest even)
__device__ __forceinline__ void split3_pair(float a0, float a1, unsigned &h, unsigned &m, unsigned &l) {
  const f32x2 a = {a0, a1};
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2));
  const f32x2 hf = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
  const f32x2 r = a - hf;
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  const f32x2 mf = {__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
  const f32x2 r2 = r - mf;
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}
// B operand of a 32-feature block from the two accumulator tiles that hold it (k order j = 0..7)
__device__ __forceinline__ Frag3 split_acc(const f32x4 &lo, const f32x4 &hi) {
  unsigned hh[4], mm[4], ll[4];
  split3_pair(lo[0], lo[1], hh[0], mm[0], ll[0]);
  split3_pair(lo[2], lo[3], hh[1], mm[1], ll[1]);
  split3_pair(hi[0], hi[1], hh[2], mm[2], ll[2]);
  split3_pair(hi[2], hi[3], hh[3], mm[3], ll[3]);
  const u32x4 h = {hh[0], hh[1], hh[2], hh[3]}, m = {mm[0], mm[1], mm[2], mm[3]}, l = {ll[0], ll[1], ll[2], ll[3]};
  Frag3 f;
  f.p[0] = __builtin_bit_cast(bf16x8, h);
  f.p[1] = __builtin_bit_cast(bf16x8, m);
  f.p[2] = __builtin_bit_cast(bf16x8, l);
  return f;
}
#define MFMA16(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0)
// acc += A B over one 32-deep block, six bf16 passes, small terms first
__device__ __forceinline__ void mma6(f32x4 &acc, const Frag3 &A, const Frag3 &B) {
  MFMA16(acc, A.p[0], B.p[2]); MFMA16(acc, A.p[2], B.p[0]); MFMA16(acc, A.p[1], B.p[1]);
  MFMA16(acc, A.p[0], B.p[1]); MFMA16(acc, A.p[1], B.p[0]); MFMA16(acc, A.p[0], B.p[0]);
}
// the same for two independent accumulators sharing the B operand, interleaved
__device__ __forceinline__ void mma6x2(f32x4 &acc0, f32x4 &acc1, const Frag3 &A0, const Frag3 &A1, const Frag3 &B) {
  MFMA16(acc0, A0.p[0], B.p[2]); MFMA16(acc1, A1.p[0], B.p[2]);
  MFMA16(acc0, A0.p[2], B.p[0]); MFMA16(acc1, A1.p[2], B.p[0]);
  MFMA16(acc0, A0.p[1], B.p[1]); MFMA16(acc1, A1.p[1], B.p[1]);
  MFMA16(acc0, A0.p[0], B.p[1]); MFMA16(acc1, A1.p[0], B.p[1]);
  MFMA16(acc0, A0.p[1], B.p[0]); MFMA16(acc1, A1.p[1], B.p[0]);
  MFMA16(acc0, A0.p[0], B.p[0]); MFMA16(acc1, A1.p[0], B.p[0]);
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
  // ---- S^T = Kblk q for the 4 heads (only the head's 16-channel sub-plane is non-zero).  The key mask
  // is the accumulator's initial value (0 / -inf per key row), so masking costs no instruction per head.
  f32x4 s[NT][H][2];
  {
    f32x4 kf[H][2];
    f32x4 mb[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mb[t][kt][r] = (16 * kt + 4 * g + r) < nv[t] ? 0.f : -INFINITY;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      kf[h][0] = ld4(Kb + (h * 2 + 0) * 256 + lane * 4);
      kf[h][1] = two_kt ? ld4(Kb + (h * 2 + 1) * 256 + lane * 4) : zero4();
#pragma unroll
      for (int t = 0; t < NT; ++t) { s[t][h][0] = mb[t][0]; s[t][h][1] = mb[t][1]; }
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
  // ---- softmax over the key axis (registers x lane groups); p stays unnormalised.  While at most 16 keys
  // exist the second key tile is all -inf and is skipped entirely ------------------------------------------
  float inv[NT][H];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float mx = fmaxf(fmaxf(s[t][h][0][0], s[t][h][0][1]), fmaxf(s[t][h][0][2], s[t][h][0][3]));
      if (two_kt) mx = fmaxf(mx, fmaxf(fmaxf(s[t][h][1][0], s[t][h][1][1]), fmaxf(s[t][h][1][2], s[t][h][1][3])));
      mx = group_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[t][h][0][r] = __builtin_amdgcn_exp2f(s[t][h][0][r] - mx); sum += s[t][h][0][r]; }
      if (two_kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[t][h][1][r] = __builtin_amdgcn_exp2f(s[t][h][1][r] - mx); sum += s[t][h][1][r]; }
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
  // ---- x = LN2(x1 + W2 relu(W1 x1 + b1) + b2), hidden streamed in 32-wide chunks; split-bf16 products ----
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
  for (int t = 0; t < NT; ++t) layer_norm(x[t], prm + PLN2W, prm + PLN2B, g);
}

// acquisition MLP (model/head.py:27-33) for NT tiles: logit = w2 . relu(W1 z + b1) + b2
template <int NT>
__device__ __forceinline__ void acq_tiles(const f32x4 (&z)[NT][2], const float *Wl, int lane, int g,
                                          float (&lg)[NT]) {
  const float *hb1 = Wl + 8 * FRAG3, *hw2 = hb1 + 128, *hb2 = hw2 + 128;
  float p[NT];
  Frag3 zf[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { p[t] = 0.f; zf[t] = split_acc(z[t][0], z[t][1]); }
#pragma unroll
  for (int mp = 0; mp < 4; ++mp) {
    const Frag3 u0 = ld_frag3(Wl + (2 * mp) * FRAG3, lane), u1 = ld_frag3(Wl + (2 * mp + 1) * FRAG3, lane);
    const f32x4 hb0 = ld4(hb1 + 32 * mp + 4 * g), hbb = ld4(hb1 + 32 * mp + 16 + 4 * g);
    const f32x4 w20 = ld4(hw2 + 32 * mp + 4 * g), w21 = ld4(hw2 + 32 * mp + 16 + 4 * g);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4 h0 = hb0, h1 = hbb;
      mma6x2(h0, h1, u0, u1, zf[t]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p[t] = fmaf(relu_nn(h0[r]), w20[r], p[t]);
        p[t] = fmaf(relu_nn(h1[r]), w21[r], p[t]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) lg[t] = group_sum(p[t]) + hb2[0];
}

struct RolloutArgs {
  int B, P, n_ctx0, n_th, T, L;
  const float *wpack;           // packed layer + head images
  const float *Ex, *Ey;         // [B, P, 32] cached point embeddings (rollout_init)
  const float *theta_tokens;    // [n_th, 32]
  const uint8_t *tmask;         // [n_th] or null
  int mode;                     // ALINE_SELECT_*
  const float *uniform;         // [T, B]
  const int64_t *forced;        // [B, T]
  int *role;                    // [B, P] out (final roles, for export)
  int64_t *idx; int *slot; float *log_prob;           // [B, T]
  float *zt;                                          // [T, B, P - n_ctx0] or null
  float *ztg;                   // [T, B, n_th, 32] encoder outputs of the target rows (GMM runs after the loop)
  unsigned long long *stamps;   // diagnostic build only: per-phase cycle sums [8 waves x 16]
};

// ---- wave-wide reductions on the VALU (DPP row ops + row/half swaps), no LDS crossbar -------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wsum(float v) {     // every lane gets the 64-lane sum
  v += dpp_f<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);    // row_half_mirror
  v += dpp_f<0x140>(v);    // row_mirror  -> 16-lane row sums
  return group_sum(v);     // across the four rows
}
__device__ __forceinline__ float wmax(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  return group_max(v);
}

// Workgroup geometry: 12 waves = 4 episode lanes x 3 waves.  Wave w serves episode lane e = w & 3 as
// its sub-wave j = w >> 2, so the three waves of an episode share one SIMD (waves i, i + 4, i + 8) and
// every SIMD carries one whole episode: ntiles tiles split 5/4/4 -> 13 tiles per SIMD, balanced.
// With B = 1000 episodes this is 250 workgroups: the whole batch is resident at once on 256 CUs.
constexpr int EPW = 4, WPE = 3, NTHREADS = EPW * WPE * 64, ETH = WPE * 64;
constexpr int MAXT = 5;   // tiles per wave (at most 15 tiles / 3 waves: N <= 240 rows)

// LDS carve (floats).  One dynamic array, 16-byte aligned offsets.
//   [0, LAYER_FLOATS)                 current layer image / head image (shared by the 4 episodes)
//   per episode (EP_FLOATS each):
//     KB  8 fragments x 256           K head-block fragments of the current layer
//     VB  4 fragments x 512           V head-block fragments
//         (head phase reuses KB: logits[256])
//     XK  [NKMAX][ES]                 inputs of the key rows for the next pre-pass
//     INT role u8[256] | kidx i8[256] | (spare 256 B) | misc int[16]
constexpr int EP_KB = 0, EP_VB = EP_KB + 8 * 256, EP_XK = EP_VB + 4 * FRAG, EP_INT = EP_XK + NKMAX * ES;
constexpr int EP_FLOATS = EP_INT + (3 * MAXROWS) / 4 + 16;
constexpr int HP_LOGIT = 0, HP_ZT = HP_LOGIT + MAXROWS, HP_RAW = HP_ZT + MAXNT * ES;   // inside KB
static_assert(HP_RAW + MAXNT * 16 * 4 <= 8 * 256, "head-phase scratch must fit in the K fragment area");
constexpr int L_TOTAL = LAYER_FLOATS + EPW * EP_FLOATS;
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
__global__ __launch_bounds__(NTHREADS, 3) void rollout_f32_kernel(RolloutArgs a) {
  unsigned long long t_prev = 0;
  if constexpr (STAMP) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory"); }
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int e = wave & 3, j = wave >> 2, tid3 = j * 64 + lane;
  const int tok = lane & 15, g = lane >> 4;
  const int b = blockIdx.x * EPW + e;
  const bool valid = b < a.B;

  float *Wl = lds;
  float *ep = lds + LAYER_FLOATS + e * EP_FLOATS;
  float *Kb = ep + EP_KB, *Vb = ep + EP_VB, *Xk = ep + EP_XK;
  float *logit = Kb + HP_LOGIT;
  unsigned char *role = reinterpret_cast<unsigned char *>(ep + EP_INT);   // 0 query, k>0 k-th context, 255 n/a
  signed char *kidx = reinterpret_cast<signed char *>(role + MAXROWS);
  int *misc = reinterpret_cast<int *>(role + 3 * MAXROWS);
  float *fmisc = reinterpret_cast<float *>(misc + 8);
  // misc: 0 n_ck, 1 n_ak, 2..4 per-wave counts, 5 nq, 6 choice;  fmisc: 0..2 per-wave partials, 3 total

  const int P = a.P, n_th = a.n_th, N = P + n_th;
  const int ntiles = (N + 15) >> 4;
  const int zw = P - a.n_ctx0;
  const int tcnt = ntiles / WPE + (j < (ntiles % WPE) ? 1 : 0);
  const int t0 = j * (ntiles / WPE) + min(j, ntiles % WPE);

  // ---- episode state: roles.  The step-invariant embeddings stay in HBM/L2 (Ex, Ey: 26 KB per
  // episode and step): X^(0)[row] = Ex[row] (+ Ey[row] once the point has joined the context).
  for (int r = tid3; r < MAXROWS; r += ETH) role[r] = r < P ? (r < a.n_ctx0 ? r + 1 : 0) : 255;
  for (int i = tid3; i < NKMAX * ES; i += ETH) Xk[i] = 0.f;
  __syncthreads();

  for (int t = 0; t < a.T; ++t) {
    // ---- key list: context slots in slot order, then the selected targets ----------------------
    // rows tid3 and tid3 + 192; chunks in row order: (j=0,p=0) (1,0) (2,0) (0,1)
    {
      const int r0 = tid3, r1 = tid3 + ETH;
      const bool c0 = r0 < P && role[r0] != 0 && role[r0] != 255;
      const bool c1 = r1 < P && role[r1] != 0 && role[r1] != 255;
      const unsigned long long b0 = __ballot(c0), b1 = __ballot(c1);
      if (lane == 0) { misc[2 + j] = __popcll(b0); if (j == 0) misc[5] = __popcll(b1); }
      __syncthreads();
      const int cnt0 = misc[2], cnt1 = misc[3], cnt2 = misc[4], cnt3 = misc[5];
      const int nck = cnt0 + cnt1 + cnt2 + cnt3;
      const int off0 = j == 0 ? 0 : (j == 1 ? cnt0 : cnt0 + cnt1);
      const unsigned long long below = (1ull << lane) - 1ull;
      int k0 = c0 ? off0 + __popcll(b0 & below) : -1;
      int k1 = c1 ? cnt0 + cnt1 + cnt2 + __popcll(b1 & below) : -1;
      auto target_key = [&](int r) {
        int nsel = 0, mine = -1;
        for (int q = 0; q < n_th; ++q) {
          const bool sel = !a.tmask || a.tmask[q];
          if (q == r - P && sel) mine = nsel;
          nsel += sel ? 1 : 0;
        }
        if (r == P) { misc[0] = nck; misc[1] = nck + nsel; }
        return mine >= 0 ? nck + mine : -1;
      };
      if (r0 >= P && r0 < N) k0 = target_key(r0);
      if (r1 >= P && r1 < N) k1 = target_key(r1);
      if (r0 < MAXROWS) kidx[r0] = (signed char)k0;
      if (r1 < MAXROWS) kidx[r1] = (signed char)k1;
      __syncthreads();
    }
    const int n_ck = misc[0], n_ak = misc[1];
    STAMP_PHASE(0)   // key list

    // ---- X^(0): this wave's token tiles --------------------------------------------------------------
    // all loads are issued before the first use so that their HBM/L2 latencies overlap
    f32x4 x[MAXT][2];
    {
      f32x4 ey[MAXT][2];
      int kk[MAXT];
#pragma unroll
      for (int i = 0; i < MAXT; ++i) {
        int row = 16 * (t0 + i) + tok;
        // keep the address arithmetic inside the step loop: hoisted to kernel entry it gets spilled and
        // every reload drains the outstanding loads (s_waitcnt vmcnt(0)), serialising the whole prologue
        asm volatile("" : "+v"(row));
        const bool ok = valid && i < tcnt && row < N;
        const float *src = ok ? (row < P ? a.Ex + ((long)b * P + row) * D : a.theta_tokens + (long)(row - P) * D)
                              : a.theta_tokens;
        x[i][0] = ld4(src + 4 * g);
        x[i][1] = ld4(src + 16 + 4 * g);
        const bool ctx = ok && row < P && role[row] != 0;
        const float *eyp = ctx ? a.Ey + ((long)b * P + row) * D : a.theta_tokens;
        ey[i][0] = ld4(eyp + 4 * g);
        ey[i][1] = ld4(eyp + 16 + 4 * g);
        if (!ctx) { ey[i][0] = zero4(); ey[i][1] = zero4(); }
        if (!ok) { x[i][0] = zero4(); x[i][1] = zero4(); }
        kk[i] = ok ? kidx[row] : -1;
      }
#pragma unroll
      for (int i = 0; i < MAXT; ++i) {
        x[i][0] += ey[i][0];
        x[i][1] += ey[i][1];
        if (kk[i] >= 0) {
          *reinterpret_cast<f32x4 *>(Xk + kk[i] * ES + 4 * g) = x[i][0];
          *reinterpret_cast<f32x4 *>(Xk + kk[i] * ES + 16 + 4 * g) = x[i][1];
        }
      }
    }
    STAMP_PHASE(1)   // x0 load

    for (int l = 0; l < a.L; ++l) {
      // ---- stream layer l's packed image into LDS (all 768 threads) ------------------------------
      {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.wpack + (long)l * LAYER_FLOATS);
        f32x4 *dst = reinterpret_cast<f32x4 *>(Wl);
        constexpr int NIT = (LAYER_FLOATS / 4 + NTHREADS - 1) / NTHREADS;
        f32x4 buf[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) { const int q = tid + i * NTHREADS; if (q < LAYER_FLOATS / 4) buf[i] = src[q]; }
#pragma unroll
        for (int i = 0; i < NIT; ++i) { const int q = tid + i * NTHREADS; if (q < LAYER_FLOATS / 4) dst[q] = buf[i]; }
      }
      __syncthreads();   // weights + Xk visible
      STAMP_PHASE(2)   // weight stream + barrier
      const float *prm = Wl + PRM_BASE;

      // ---- pre-pass: K^T / V of the key tiles.  items: 0 K kt0, 1 V kt0, 2 K kt1, 3 V kt1 ------------
      if (valid) {
        const int nitems = n_ak > 16 ? 4 : 2;
        for (int it = j; it < nitems; it += WPE) {
          const int kt = it >> 1;
          const int key = 16 * kt + tok;
          Frag xf;
          xf.lo = ld4(Xk + key * ES + 4 * g);
          xf.hi = ld4(Xk + key * ES + 16 + 4 * g);
          if ((it & 1) == 0) {
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
              f32x4 v = ((g >> 1) == (h & 1)) ? kacc[h >> 1] : zero4();
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
              for (int hh = 0; hh < 2; ++hh) {
                f32x4 v = ((tok >> 3) == hh) ? vacc[nt] : zero4();
                *reinterpret_cast<f32x4 *>(Vb + (nt * 2 + hh) * FRAG + kt * 256 + lane * 4) = v;
              }
          }
        }
      }
      __syncthreads();   // K/V fragments visible
      STAMP_PHASE(3)   // pre-pass + barrier

      // ---- main pass: this wave's tiles, one at a time (registers rotate so that every index is static)
      {
        const bool two_kt = n_ak > 16;
        const bool publish = l + 1 < a.L;
#pragma unroll 1
        for (int it = 0; it < MAXT; ++it) {
          f32x4 xp[1][2] = {{x[0][0], x[0][1]}};
          if (valid && it < tcnt) {
            const int row = 16 * (t0 + it) + tok;
            const int nv[1] = {(row < P && role[row] == 0) ? n_ak : n_ck};
            layer_tiles<1>(xp, Wl, prm, Kb, Vb, lane, g, nv, two_kt);
            const int k = (publish && row < N) ? kidx[row] : -1;
            if (k >= 0) {   // key rows publish x^(l+1) for the next layer's pre-pass
              *reinterpret_cast<f32x4 *>(Xk + k * ES + 4 * g) = xp[0][0];
              *reinterpret_cast<f32x4 *>(Xk + k * ES + 16 + 4 * g) = xp[0][1];
            }
          }
#pragma unroll
          for (int i = 0; i + 1 < MAXT; ++i) { x[i][0] = x[i + 1][0]; x[i][1] = x[i + 1][1]; }
          x[MAXT - 1][0] = xp[0][0];
          x[MAXT - 1][1] = xp[0][1];
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
      for (int i = tid; i < HEAD_FLOATS / 4; i += NTHREADS) dst[i] = src[i];
    }
    __syncthreads();
    STAMP_PHASE(6)   // head image stream
#pragma unroll 1
    for (int it = 0; it < MAXT; ++it) {
      const f32x4 zp[1][2] = {{x[0][0], x[0][1]}};
      if (valid && it < tcnt) {
        const int row = 16 * (t0 + it) + tok;
        float lg[1];
        acq_tiles<1>(zp, Wl, lane, g, lg);
        if (g == 0 && row < MAXROWS) logit[row] = lg[0];
        if (row >= P && row < N) {
          float *zo = a.ztg + (((long)t * a.B + b) * n_th + (row - P)) * D;
          *reinterpret_cast<f32x4 *>(zo + 4 * g) = zp[0][0];
          *reinterpret_cast<f32x4 *>(zo + 16 + 4 * g) = zp[0][1];
        }
      }
#pragma unroll
      for (int i = 0; i + 1 < MAXT; ++i) { x[i][0] = x[i + 1][0]; x[i][1] = x[i + 1][1]; }
      x[MAXT - 1][0] = zp[0][0];
      x[MAXT - 1][1] = zp[0][1];
    }
    STAMP_PHASE(7)   // acquisition MLP
    __syncthreads();
    STAMP_PHASE(8)   // barrier after acquisition

    // ---- softmax over the remaining queries + design selection (model/head.py:347-362), by the three
    // waves of the episode: thread tid3 owns slots tid3 and tid3 + 192 --------------------------------
    {
      const int s0 = tid3, s1 = tid3 + ETH;
      const bool q0 = valid && s0 < P && role[s0] == 0;
      const bool q1 = valid && s1 < P && role[s1] == 0;
      const float l0 = q0 ? logit[s0] : -INFINITY, l1 = q1 ? logit[s1] : -INFINITY;
      const unsigned long long b0 = __ballot(q0), b1 = __ballot(q1);
      const float mw = wmax(fmaxf(l0, l1));
      if (lane == 0) { misc[2 + j] = __popcll(b0); if (j == 0) misc[5] = __popcll(b1); fmisc[j] = mw; }
      __syncthreads();
      const int cnt0 = misc[2], cnt1 = misc[3], cnt2 = misc[4], cnt3 = misc[5];
      const int nq = cnt0 + cnt1 + cnt2 + cnt3;
      const float mx = fmaxf(fmaxf(fmisc[0], fmisc[1]), fmisc[2]);
      const unsigned long long below = (1ull << lane) - 1ull;
      const int rk0 = (j == 0 ? 0 : (j == 1 ? cnt0 : cnt0 + cnt1)) + __popcll(b0 & below);   // compacted index
      const int rk1 = cnt0 + cnt1 + cnt2 + __popcll(b1 & below);
      const float e0 = q0 ? __expf(l0 - mx) : 0.f, e1 = q1 ? __expf(l1 - mx) : 0.f;
      const float sw = wsum(e0 + e1);
      __syncthreads();                      // everyone has read fmisc (max) before it is reused
      if (lane == 0) fmisc[j] = sw;
      __syncthreads();
      const float inv = 1.f / (fmisc[0] + fmisc[1] + fmisc[2]);
      const float p0 = e0 * inv, p1 = e1 * inv;
      if (valid && a.zt) {
        float *zo = a.zt + ((long)t * a.B + b) * zw;
        if (q0) zo[rk0] = p0;
        if (q1) zo[rk1] = p1;
        for (int i = nq + tid3; i < zw; i += ETH) zo[i] = 0.f;
      }
      // choose: each thread proposes (value, compacted index) or "no"; winner found by reductions
      int choice = -1;       // compacted index of the chosen query (episode-uniform after this block)
      float pch = 0.f;
      if (a.mode == 0) {
        // argmax with first-index tie-break (torch.max)
        float bv = fmaxf(p0, p1);
        const float wbest = wmax(q0 || q1 ? bv : -1.f);
        __syncthreads();
        if (lane == 0) fmisc[j] = wbest;
        __syncthreads();
        const float best = fmaxf(fmaxf(fmisc[0], fmisc[1]), fmisc[2]);
        int cand = 0x7fffffff;
        if (q0 && p0 == best) cand = rk0;
        if (q1 && p1 == best) cand = min(cand, rk1);
        // min over the wave, then over the 3 waves
        for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
        if (lane == 0) misc[2 + j] = cand;
        __syncthreads();
        choice = min(min(misc[2], misc[3]), misc[4]);
        pch = best;
      } else if (a.mode == 2) {
        choice = valid ? (int)a.forced[(long)b * a.T + t] : 0;
        choice = min(max(choice, 0), nq - 1);
      } else {
        // inverse CDF in compacted (= slot) order: inclusive prefix sums of p over slots
        // chunk order: (j=0,s0) (1,s0) (2,s0) (0,s1): wave-level scans + chunk offsets
        float inc0 = p0, inc1 = p1;
        for (int o = 1; o < 64; o <<= 1) {
          const float u0 = __shfl_up(inc0, o, 64), u1 = __shfl_up(inc1, o, 64);
          if (lane >= o) { inc0 += u0; inc1 += u1; }
        }
        __syncthreads();
        if (lane == 63) { fmisc[j] = inc0; if (j == 0) fmisc[3] = inc1; }
        __syncthreads();
        const float c0 = fmisc[0], c1 = fmisc[1], c2 = fmisc[2], c3 = fmisc[3];
        const float tot = c0 + c1 + c2 + c3;
        const float u = (valid ? a.uniform[(long)t * a.B + b] : 0.f) * tot;
        const float base0 = j == 0 ? 0.f : (j == 1 ? c0 : c0 + c1);
        int cand = 0x7fffffff;
        if (q0 && base0 + inc0 > u) cand = rk0;
        if (q1 && (c0 + c1 + c2) + inc1 > u) cand = min(cand, rk1);
        for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
        if (lane == 0) misc[2 + j] = cand;
        __syncthreads();
        choice = min(min(misc[2], misc[3]), misc[4]);
        if (choice == 0x7fffffff) choice = nq - 1;
      }
      // the owner of the chosen compacted index publishes slot / log-prob and updates the role
      const bool own0 = q0 && rk0 == choice, own1 = q1 && rk1 == choice;
      if (own0 || own1) {
        const int sl = own0 ? s0 : s1;
        float val = own0 ? p0 : p1;
        if (a.mode != 0) {
          // Categorical(probs).log_prob: probs / probs.sum(), clamped to [eps, 1 - eps]
          val = fminf(fmaxf(val, 1.1920929e-07f), 1.f - 1.1920929e-07f);
        }
        const long o = (long)b * a.T + t;
        if (a.idx) a.idx[o] = choice;
        if (a.slot) a.slot[o] = sl;
        if (a.log_prob) a.log_prob[o] = logf(val);
        role[sl] = (unsigned char)((P - nq) + 1);
      }
      (void)pch;
    }
    STAMP_PHASE(9)   // selection

    // (the GMM posterior of all T steps runs after the loop on the saved target-row encodings: it
    //  does not feed back into the rollout, model/head.py:365 / train_aline.py:92-95)
    __syncthreads();
    STAMP_PHASE(11)  // GMM epilogue + barrier
  }
  if (valid)
    for (int r = tid3; r < P; r += ETH) {
      const int rl = role[r];
      a.role[(long)b * P + r] = rl;
    }
}

}  // namespace fused
