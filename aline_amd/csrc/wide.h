// Wide path: d = 256 (head_dim 32), any F multiple of 64, bf16 MFMA with fp32 accumulate -- the
// matrix-core-bound regime of the same step (north star: ">= 40 % of the bf16 MFMA roofline at d >= 256").
//
// Same operand scheme as fused_rollout.h, scaled up: a token tile is X^T [features x 16 tokens]; every
// linear layer is Y^T = W X^T, so an accumulator tile is the B operand of the next product (k order
// pi(ks,g,j) = 32 ks + 16 (j>>2) + 4 g + (j&3)) and weights are A fragments pre-permuted at pack time.
// What changes at d = 256: a layer's weights (1.5 MB bf16) no longer fit in LDS, so they are STREAMED:
// 32 KB chunks of 32 fragments go global -> registers -> LDS (double buffered, one barrier per chunk) and
// are consumed by all 8 waves of the workgroup, each of which keeps 32 tokens (2 column tiles) on chip:
// its input as 64 VGPRs of bf16 B fragments, its 256-feature output as 128 fp32 accumulator VGPRs.
// Activations travel between kernels as bf16 TILE IMAGES: for every 16 consecutive token rows the 8 B-operand
// fragments of X^T (k-step ks: 64 lanes x 16 B = 1 KB), i.e. exactly what a wave loads as MFMA operands and
// exactly what an epilogue holds in its accumulators -- every load/store of a wave is one contiguous KB.
// Head h of Q/K/V (head_dim 32 = one k-step) is fragment ks = h of a tile.  LayerNorm, softmax, biases: fp32.
//
//   wide_block_kernel<WB_QKV>   Q,K,V = in_proj X               (3 passes over the resident input)
//   wide_attention_kernel       masked set-attention, one workgroup per episode, one wave per head
//   wide_block_kernel<WB_OUT>   X1 = LN1(X + Wo A + bo)
//   wide_block_kernel<WB_FFN>   X  = LN2(X1 + W2 relu(W1 X1 + b1) + b2): the hidden layer never leaves
//                               the CU (32-unit chunks: 16 fragments of W1, 16 of W2)
//   wide_block_kernel<WB_ACQ>   acquisition logits = w2 . relu(W1a z + b1a) + b2a
//   wide_block_kernel<WB_GMM>   one GMM head: raw[0..2] = W2 relu(W1 z + b1) + b2  (same code, 3 outputs per token)
#pragma once
#include "common.h"
#include "kernels.h"

namespace wide {

constexpr int D = 256, HD = 32, H = 8, NMT = D / 16, NKS = D / 32;
constexpr int CHUNK_FRAGS = 32, FRAG_W = 256, CHUNK_W = CHUNK_FRAGS * FRAG_W;   // 32-bit words
constexpr int NTHREADS = 512;             // attention kernel
#ifndef WIDE_NT
#define WIDE_NT 2                         // token tiles per wave in the block kernels
#endif
constexpr int NT = WIDE_NT, BTHREADS = NT == 2 ? 512 : 256, BWAVES = BTHREADS / 64;
constexpr int WTOK = 16 * NT, WG_TOK = BWAVES * WTOK, NST = CHUNK_W / 4 / BTHREADS;
enum { WB_QKV = 0, WB_OUT = 1, WB_FFN = 2, WB_ACQ = 3, WB_GMM = 4 };

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
#define WMFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0)

// layer image (32-bit words): [QKV 12 chunks][OUT 4 chunks][FFN F/32 chunks], then fp32 params
__host__ __device__ inline int layer_chunks(int F) { return 12 + 4 + F / 32; }
__host__ __device__ inline int layer_params(int F) { return 3 * D + D + F + D + 4 * D; }   // bqkv, bo, b1, b2, ln1w, ln1b, ln2w, ln2b
__host__ __device__ inline long layer_words(int F) { return (long)layer_chunks(F) * CHUNK_W + layer_params(F); }
__host__ __device__ inline int head_chunks(int F) { return F / 64; }
__host__ __device__ inline long head_words(int F) { return (long)head_chunks(F) * CHUNK_W + 2 * F + 4; }   // b1a, w2a, b2a
__host__ __device__ inline long gmm_words(int F) { return (long)head_chunks(F) * CHUNK_W + 4 * F + 4; }    // b1, w2[3][F], b2[3]
__host__ __device__ inline long emb_words(int F) { return (long)head_chunks(F) * CHUNK_W; }               // W2 [256, F] of a point embedder

struct PackArgs {
  int L, F;
  const float *in_proj_w[8], *in_proj_b[8], *out_proj_w[8], *out_proj_b[8], *lin1_w[8], *lin1_b[8], *lin2_w[8],
      *lin2_b[8], *n1w[8], *n1b[8], *n2w[8], *n2b[8];
  const float *acq_w1, *acq_b1, *acq_w2, *acq_b2;
  int C;                         // GMM heads packed after the acquisition head (each: F/64 chunks of W1 | b1 | w2[3][F] | b2[3])
  const float *gmm_w1[16], *gmm_b1[16], *gmm_w2[16], *gmm_b2[16];
  const float *emb_w2[2];        // second layers of the x / y point embedders, packed last (chunk c: hidden groups 2c, 2c+1 x 16 output tiles)
  unsigned *out;
};

// word w (0..3) of lane `lane` of the fragment (rows row0.., k-step ks) of a [*, K] row-major weight
__device__ __forceinline__ unsigned frag_word(const float *W, int K, int row0, int ks, int lane, int w, float scale) {
  const int g = lane >> 4, j0 = 2 * w;
  const int k0 = 32 * ks + 16 * (j0 >> 2) + 4 * g + (j0 & 3);
  const float *p = W + (long)(row0 + (lane & 15)) * K + k0;
  return pack_bf16(p[0] * scale, p[1] * scale);
}

__global__ void pack_kernel(PackArgs a) {
  const long lw = layer_words(a.F), total = a.L * lw + head_words(a.F) + a.C * gmm_words(a.F) + 2 * emb_words(a.F);
  const float qscale = rsqrtf((float)HD) * 1.44269504088896340736f;   // softmax runs in exp2
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    unsigned v = 0;
    if (i < a.L * lw) {
      const int l = i / lw;
      const long o = i % lw;
      const long nfw = (long)layer_chunks(a.F) * CHUNK_W;
      if (o < nfw) {
        const int chunk = o / CHUNK_W, fid = (o % CHUNK_W) / FRAG_W, e = o % FRAG_W, lane = e >> 2, w = e & 3;
        if (chunk < 12) {               // QKV pass p, chunk cc: fragments (ks_local, mt)
          const int p = chunk / 4, cc = chunk % 4, ks = 2 * cc + fid / 16, mt = fid % 16;
          v = frag_word(a.in_proj_w[l], D, 256 * p + 16 * mt, ks, lane, w, p == 0 ? qscale : 1.f);
        } else if (chunk < 16) {
          const int cc = chunk - 12, ks = 2 * cc + fid / 16, mt = fid % 16;
          v = frag_word(a.out_proj_w[l], D, 16 * mt, ks, lane, w, 1.f);
        } else {
          const int c = chunk - 16;
          if (fid < 16) v = frag_word(a.lin1_w[l], D, 32 * c + 16 * (fid & 1), fid >> 1, lane, w, 1.f);   // (ks, mt')
          else v = frag_word(a.lin2_w[l], a.F, 16 * (fid - 16), c, lane, w, 1.f);                           // (mt, ks = c)
        }
      } else {
        const int p = o - nfw;
        float f;
        if (p < 3 * D) f = a.in_proj_b[l][p] * (p < D ? qscale : 1.f);
        else if (p < 4 * D) f = a.out_proj_b[l][p - 3 * D];
        else if (p < 4 * D + a.F) f = a.lin1_b[l][p - 4 * D];
        else {
          const int q = p - 4 * D - a.F;
          f = q < D ? a.lin2_b[l][q] : q < 2 * D ? a.n1w[l][q - D] : q < 3 * D ? a.n1b[l][q - 2 * D]
            : q < 4 * D ? a.n2w[l][q - 3 * D] : a.n2b[l][q - 4 * D];
        }
        v = __float_as_uint(f);
      }
    } else if (i >= a.L * lw + head_words(a.F) + a.C * gmm_words(a.F)) {
      const long oe = i - (a.L * lw + head_words(a.F) + a.C * gmm_words(a.F));
      const int which = oe / emb_words(a.F);
      const long o = oe % emb_words(a.F);
      const int chunk = o / CHUNK_W, fid = (o % CHUNK_W) / FRAG_W, e = o % FRAG_W, lane = e >> 2, w = e & 3;
      v = frag_word(a.emb_w2[which], a.F, 16 * (fid % 16), 2 * chunk + fid / 16, lane, w, 1.f);   // (mt, k-step = hidden group)
    } else if (i >= a.L * lw + head_words(a.F)) {
      const long og = i - a.L * lw - head_words(a.F);
      const int c = og / gmm_words(a.F);
      const long o = og % gmm_words(a.F);
      const long nfw = (long)head_chunks(a.F) * CHUNK_W;
      if (o < nfw) {                    // same fragment order as the acquisition head
        const int chunk = o / CHUNK_W, fid = (o % CHUNK_W) / FRAG_W, e = o % FRAG_W, lane = e >> 2, w = e & 3;
        const int grp = 2 * chunk + fid / 16, f16 = fid % 16;
        v = frag_word(a.gmm_w1[c], D, 32 * grp + 16 * (f16 & 1), f16 >> 1, lane, w, 1.f);
      } else {
        const int p = o - nfw;
        const float f = p < a.F ? a.gmm_b1[c][p] : p < 4 * a.F ? a.gmm_w2[c][p - a.F] : p < 4 * a.F + 3 ? a.gmm_b2[c][p - 4 * a.F] : 0.f;
        v = __float_as_uint(f);
      }
    } else {
      const long o = i - a.L * lw;
      const long nfw = (long)head_chunks(a.F) * CHUNK_W;
      if (o < nfw) {                    // acquisition W1a [F, 256]: chunk c = hidden groups 2c, 2c+1; (grp, ks, mt')
        const int chunk = o / CHUNK_W, fid = (o % CHUNK_W) / FRAG_W, e = o % FRAG_W, lane = e >> 2, w = e & 3;
        const int grp = 2 * chunk + fid / 16, f16 = fid % 16;
        v = frag_word(a.acq_w1, D, 32 * grp + 16 * (f16 & 1), f16 >> 1, lane, w, 1.f);
      } else {
        const int p = o - nfw;
        const float f = p < a.F ? a.acq_b1[p] : p < 2 * a.F ? a.acq_w2[p - a.F] : p == 2 * a.F ? a.acq_b2[0] : 0.f;
        v = __float_as_uint(f);
      }
    }
    a.out[i] = v;
  }
}

// tile image addressing, in 16-byte pieces: token row -> (tile = row / 16, tok = row % 16); piece (ks, g) of a
// token holds features 32 ks + 4 g + (0..3) and 32 ks + 16 + 4 g + (0..3)  (the B fragment of lane 16 g + tok)
__host__ __device__ inline long tile_rows(long M) { return (M + 15) / 16 * 16; }
__device__ __forceinline__ long piece(long row, int ks, int g) { return ((row >> 4) * NKS + ks) * 64 + g * 16 + (row & 15); }

// X0 (bf16 tile image) from the cached fp32 point embeddings: Ex (+ Ey on context rows), theta tokens
__global__ void assemble_bf16_kernel(Geo g, const float *__restrict__ Ex, const float *__restrict__ Ey,
                                     int ey_rows, const float *__restrict__ theta_tokens,
                                     u32x4 *__restrict__ X) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // one thread = one 16-byte piece, image order
  const long M = (long)g.B * g.N;
  if (i >= tile_rows(M) * (D / 8)) return;
  const int lane = i & 63, ks = (i >> 6) % NKS, gq = lane >> 4;
  const long r = (i >> 6) / NKS * 16 + (lane & 15);
  u32x4 o = {0u, 0u, 0u, 0u};
  if (r < M) {
    const int b = r / g.N, row = r % g.N, c = 32 * ks + 4 * gq;
    f32x4 lo, hi;
    if (row < g.P + g.n_td) {
      const float *e = Ex + ((long)b * (g.P + g.n_td) + row) * D + c;
      lo = *reinterpret_cast<const f32x4 *>(e); hi = *reinterpret_cast<const f32x4 *>(e + 16);
      if (row < g.P && is_ctx(g, b, row)) {
        const float *y = Ey + ((long)b * ey_rows + row) * D + c;
        lo += *reinterpret_cast<const f32x4 *>(y); hi += *reinterpret_cast<const f32x4 *>(y + 16);
      }
    } else {
      const float *t = theta_tokens + (row - g.P - g.n_td) * D + c;
      lo = *reinterpret_cast<const f32x4 *>(t); hi = *reinterpret_cast<const f32x4 *>(t + 16);
    }
    o = (u32x4){pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3]), pack_bf16(hi[0], hi[1]), pack_bf16(hi[2], hi[3])};
  }
  X[i] = o;
}

// The step's input image changes in ONE row per episode from step to step: the point chosen at step t (role ==
// order of entry == `order`) becomes a context point, i.e. its row becomes Ex + Ey.  One workgroup per episode.
__global__ __launch_bounds__(64) void patch_context_row_kernel(Geo g, int order, const float *__restrict__ Ex,
                                                               const float *__restrict__ Ey, int ey_rows,
                                                               u32x4 *__restrict__ X) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int slot = -1;
  for (int p = lane; p < g.P; p += 64)
    if (g.role[(long)b * g.P + p] == order) slot = p;
  slot = __reduce_max_sync(~0ull, slot);
  if (slot < 0 || lane >= D / 8) return;
  const int ks = lane >> 2, gq = lane & 3, c = 32 * ks + 4 * gq;
  const float *e = Ex + ((long)b * (g.P + g.n_td) + slot) * D + c, *y = Ey + ((long)b * ey_rows + slot) * D + c;
  const f32x4 lo = *reinterpret_cast<const f32x4 *>(e) + *reinterpret_cast<const f32x4 *>(y);
  const f32x4 hi = *reinterpret_cast<const f32x4 *>(e + 16) + *reinterpret_cast<const f32x4 *>(y + 16);
  X[piece((long)b * g.N + slot, ks, gq)] =
      (u32x4){pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3]), pack_bf16(hi[0], hi[1]), pack_bf16(hi[2], hi[3])};
}

// dense fp32 rows out of a tile image: out[r] = token (r / rows_per_ep) * ep_stride + off + r % rows_per_ep
__global__ void image_rows_to_f32_kernel(const u32x4 *__restrict__ X, int rows_per_ep, int ep_stride,
                                         int off, long rows, float *__restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // one thread = one piece (8 features)
  if (i >= rows * (D / 8)) return;
  const long r = i / (D / 8);
  const int pc = i % (D / 8), ks = pc >> 2, gq = pc & 3;
  const long src = (r / rows_per_ep) * ep_stride + off + (r % rows_per_ep);
  const u32x4 v = X[piece(src, ks, gq)];
  float *o = out + r * D + 32 * ks + 4 * gq;
  *reinterpret_cast<f32x4 *>(o) = (f32x4){bf_lo(v[0]), bf_hi(v[0]), bf_lo(v[1]), bf_hi(v[1])};
  *reinterpret_cast<f32x4 *>(o + 16) = (f32x4){bf_lo(v[2]), bf_hi(v[2]), bf_lo(v[3]), bf_hi(v[3])};
}

struct BlockArgs {
  const u32x4 *X;                // tile image of the input of the matmul chain
  const u32x4 *Xres;             // tile image of the residual (WB_OUT), else unused
  u32x4 *Y;                      // output tile image (WB_QKV: three images Q | K | V, tile_rows(M) * 32 pieces apart)
  float *logits;                 // WB_ACQ: [M]; WB_GMM: raw[row * out_stride + out_off + j], j < 3
  int out_stride, out_off;
  const unsigned *wimg;          // first chunk of this block's weights
  const float *prm;              // this layer's (or the head's) fp32 parameter block
  int M, F;
};

// B fragments of a 16-token tile: 8 contiguous KB
__device__ __forceinline__ void load_xfrags(const u32x4 *X, long row, int g, bf16x8 (&xb)[NKS]) {
  const u32x4 *xr = X + piece(row, 0, g);
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) xb[ks] = __builtin_bit_cast(bf16x8, xr[ks * 64]);
}
__device__ __forceinline__ bf16x8 acc_to_frag(const f32x4 &lo, const f32x4 &hi) {
  const u32x4 v = {pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3]), pack_bf16(hi[0], hi[1]), pack_bf16(hi[2], hi[3])};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ float group_sum4(float v) {   // over the 4 lane groups holding one token
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <int MODE>
__global__ __launch_bounds__(BTHREADS) void wide_block_kernel(BlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];      // [2][CHUNK_W] + params (floats)
  unsigned *buf0 = lds, *buf1 = lds + CHUNK_W;
  float *ps = reinterpret_cast<float *>(lds + 2 * CHUNK_W);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tok = lane & 15, g = lane >> 4;
  const long base = (long)blockIdx.x * WG_TOK + wave * WTOK;
  long row[NT];
  bool ok[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) { row[ct] = min(base + 16 * ct + tok, (long)a.M - 1); ok[ct] = base + 16 * ct + tok < a.M; }

  // parameters of this block into LDS
  const int nprm = MODE == WB_QKV ? 3 * D : MODE == WB_OUT ? 3 * D : MODE == WB_FFN ? a.F + 3 * D : MODE == WB_GMM ? 4 * a.F + 4 : 2 * a.F + 4;
  for (int i = tid; i < nprm; i += BTHREADS) {
    float v;
    if (MODE == WB_QKV) v = a.prm[i];                                                  // bq | bk | bv
    else if (MODE == WB_OUT) v = i < D ? a.prm[3 * D + i] : a.prm[4 * D + a.F + D + (i - D)];       // bo | ln1w | ln1b
    else if (MODE == WB_FFN) v = i < a.F ? a.prm[4 * D + i] : i < a.F + D ? a.prm[4 * D + a.F + (i - a.F)]
                                 : a.prm[4 * D + a.F + 3 * D + (i - a.F - D)];              // b1 | b2 | ln2w | ln2b
    else v = a.prm[i];                                                                  // b1a | w2a | b2a
    ps[i] = v;
  }

  bf16x8 xb[NT][NKS];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) load_xfrags(a.X, row[ct], g, xb[ct]);

  const int nchunk = MODE == WB_QKV ? 12 : MODE == WB_OUT ? 4 : MODE == WB_FFN ? a.F / 32 : a.F / 64;   // (WB_ACQ, WB_GMM: 64 hidden units per chunk)
  const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(a.wimg);
  u32x4 st[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) st[i] = wsrc[tid + i * BTHREADS];

  f32x4 y[NMT][NT];
  constexpr int NOUT = MODE == WB_GMM ? 3 : 1;
  float plog[NT][NOUT];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct)
#pragma unroll
    for (int j = 0; j < NOUT; ++j) plog[ct][j] = 0.f;
  if (MODE != WB_ACQ && MODE != WB_GMM) {
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) y[mt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  for (int c = 0; c < nchunk; ++c) {
    unsigned *buf = (c & 1) ? buf1 : buf0;
#pragma unroll
    for (int i = 0; i < NST; ++i) reinterpret_cast<u32x4 *>(buf)[tid + i * BTHREADS] = st[i];
    __syncthreads();
    if (c + 1 < nchunk) {
#pragma unroll
      for (int i = 0; i < NST; ++i) st[i] = wsrc[(long)(c + 1) * (CHUNK_W / 4) + tid + i * BTHREADS];
    }
    const bf16x8 *fr = reinterpret_cast<const bf16x8 *>(buf) + lane;        // fragment f at fr[f * 64]

    if (MODE == WB_QKV || MODE == WB_OUT) {
      // chunk = k-steps 2cc, 2cc+1 of the 16 output tiles
      const int cc = MODE == WB_QKV ? c & 3 : c;
#pragma unroll
      for (int kl = 0; kl < 2; ++kl)
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
          const bf16x8 A = fr[(kl * 16 + mt) * 64];
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) WMFMA(y[mt][ct], A, xb[ct][2 * cc + kl]);
        }
      if (MODE == WB_QKV && cc == 3) {
        // end of pass p: bias, store bf16 at column block p, reset the accumulators
        const int p = c >> 2;
        u32x4 *img = a.Y + (long)p * tile_rows(a.M) * (D / 8);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const f32x4 b0 = *reinterpret_cast<const f32x4 *>(ps + p * D + 32 * ks + 4 * g);
          const f32x4 b1 = *reinterpret_cast<const f32x4 *>(ps + p * D + 32 * ks + 16 + 4 * g);
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) {
            if (ok[ct])
              img[piece(row[ct], ks, g)] =
                  __builtin_bit_cast(u32x4, acc_to_frag(y[2 * ks][ct] + b0, y[2 * ks + 1][ct] + b1));
            y[2 * ks][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
            y[2 * ks + 1][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
      }
    } else if (MODE == WB_FFN) {
      f32x4 h[2][NT];
      {
        const f32x4 b0 = *reinterpret_cast<const f32x4 *>(ps + 32 * c + 4 * g);
        const f32x4 b1 = *reinterpret_cast<const f32x4 *>(ps + 32 * c + 16 + 4 * g);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) { h[0][ct] = b0; h[1][ct] = b1; }
      }
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const bf16x8 A0 = fr[(2 * ks) * 64], A1 = fr[(2 * ks + 1) * 64];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) { WMFMA(h[0][ct], A0, xb[ct][ks]); WMFMA(h[1][ct], A1, xb[ct][ks]); }
      }
      bf16x8 hb[NT];
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
#ifdef ALINE_RELU_SITE_FFN
        for (int r = 0; r < 4; ++r) { h[0][ct][r] = relu_int(h[0][ct][r]); h[1][ct][r] = relu_int(h[1][ct][r]); }
#else
        for (int r = 0; r < 4; ++r) { h[0][ct][r] = relu_nn(h[0][ct][r]); h[1][ct][r] = relu_nn(h[1][ct][r]); }
#endif
        hb[ct] = acc_to_frag(h[0][ct], h[1][ct]);
      }
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) {
        const bf16x8 A = fr[(16 + mt) * 64];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) WMFMA(y[mt][ct], A, hb[ct]);
      }
    } else {   // WB_ACQ / WB_GMM: two groups of 32 hidden units per chunk
#pragma unroll
      for (int grp = 0; grp < 2; ++grp) {
        const int hbase = 64 * c + 32 * grp;
        f32x4 h[2][NT];
        const f32x4 b0 = *reinterpret_cast<const f32x4 *>(ps + hbase + 4 * g);
        const f32x4 b1 = *reinterpret_cast<const f32x4 *>(ps + hbase + 16 + 4 * g);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) { h[0][ct] = b0; h[1][ct] = b1; }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const bf16x8 A0 = fr[(grp * 16 + 2 * ks) * 64], A1 = fr[(grp * 16 + 2 * ks + 1) * 64];
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) { WMFMA(h[0][ct], A0, xb[ct][ks]); WMFMA(h[1][ct], A1, xb[ct][ks]); }
        }
#pragma unroll
        for (int j = 0; j < NOUT; ++j) {
          const f32x4 w0 = *reinterpret_cast<const f32x4 *>(ps + (1 + j) * a.F + hbase + 4 * g);
          const f32x4 w1 = *reinterpret_cast<const f32x4 *>(ps + (1 + j) * a.F + hbase + 16 + 4 * g);
#pragma unroll
          for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#if defined(ALINE_RELU_SITE_ACQ_AFTER)   // (diagnostic: wait states between the integer max and its consumer)
              { float r0 = relu_int(h[0][ct][r]), r1 = relu_int(h[1][ct][r]);
                asm volatile("s_nop 3" : "+v"(r0), "+v"(r1));
                plog[ct][j] = fmaf(r0, w0[r], plog[ct][j]);
                plog[ct][j] = fmaf(r1, w1[r], plog[ct][j]); }
#elif defined(ALINE_RELU_SITE_ACQ_NOPK)  // (diagnostic: integer max, but no packed FMA: each product is its own statement)
              { float r0 = relu_int(h[0][ct][r]), r1 = relu_int(h[1][ct][r]);
                float p0 = plog[ct][j];
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "v"(r0), "v"(w0[r]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "v"(r1), "v"(w1[r]));
                plog[ct][j] = p0; }
#elif defined(ALINE_RELU_SITE_ACQ_GUARD)     // (diagnostic: integer ReLU with NaNs mapped to 0 first, as fmaxf does)
              plog[ct][j] = fmaf(relu_int(h[0][ct][r] != h[0][ct][r] ? 0.f : h[0][ct][r]), w0[r], plog[ct][j]);
              plog[ct][j] = fmaf(relu_int(h[1][ct][r] != h[1][ct][r] ? 0.f : h[1][ct][r]), w1[r], plog[ct][j]);
#elif defined(ALINE_RELU_SITE_ACQ)
              plog[ct][j] = fmaf(relu_int(h[0][ct][r]), w0[r], plog[ct][j]);
              plog[ct][j] = fmaf(relu_int(h[1][ct][r]), w1[r], plog[ct][j]);
#else
              plog[ct][j] = fmaf(relu_nn(h[0][ct][r]), w0[r], plog[ct][j]);
              plog[ct][j] = fmaf(relu_nn(h[1][ct][r]), w1[r], plog[ct][j]);
#endif
            }
        }
      }
    }
  }

  if (MODE == WB_ACQ || MODE == WB_GMM) {
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
      for (int j = 0; j < NOUT; ++j) {
        const float v = group_sum4(plog[ct][j]) + ps[(1 + NOUT) * a.F + j];
        if (g == 0 && ok[ct]) {
          if (MODE == WB_ACQ) a.logits[row[ct]] = v;
          else a.logits[row[ct] * a.out_stride + a.out_off + j] = v;
        }
      }
    return;
  }
  if (MODE == WB_QKV) return;

  // ---- residual + LayerNorm over the 256 features of each token (fp32), bf16 store ------------------------
  const float *bo = ps + (MODE == WB_OUT ? 0 : a.F);
  const float *lw = bo + D, *lb = lw + D;
  if (MODE == WB_OUT) {     // the residual is the layer input, not this block's A operand
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) load_xfrags(a.Xres, row[ct], g, xb[ct]);
  }
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    float s = 0.f;
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
      const f32x4 bv = *reinterpret_cast<const f32x4 *>(bo + 16 * mt + 4 * g);
      const u32x4 xr = __builtin_bit_cast(u32x4, xb[ct][mt >> 1]);     // features 32 ks + 16 (mt&1) + 4 g + r
      const unsigned w0 = xr[2 * (mt & 1)], w1 = xr[2 * (mt & 1) + 1];
      y[mt][ct][0] += bv[0] + bf_lo(w0); y[mt][ct][1] += bv[1] + bf_hi(w0);
      y[mt][ct][2] += bv[2] + bf_lo(w1); y[mt][ct][3] += bv[3] + bf_hi(w1);
      s += (y[mt][ct][0] + y[mt][ct][1]) + (y[mt][ct][2] + y[mt][ct][3]);
    }
    const float mean = group_sum4(s) * (1.f / D);
    float ss = 0.f;
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float t = y[mt][ct][r] - mean; ss = fmaf(t, t, ss); }
    const float rstd = rsqrtf(group_sum4(ss) * (1.f / D) + 1e-5f);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      f32x4 o[2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int mt = 2 * ks + hf;
        const f32x4 wv = *reinterpret_cast<const f32x4 *>(lw + 16 * mt + 4 * g);
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(lb + 16 * mt + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[hf][r] = (y[mt][ct][r] - mean) * rstd * wv[r] + bv[r];
      }
      if (ok[ct]) a.Y[piece(row[ct], ks, g)] = __builtin_bit_cast(u32x4, acc_to_frag(o[0], o[1]));
    }
  }
}

// ---- point embedder  E[row] = W2 relu(W1 x[row] + b1) + b2  (model/embedder.py:47-57), d = 256 --------------
// K = dim_x / dim_y <= 8 inputs: the hidden units are FMA work done directly in the B-fragment layout (lane =
// token, 8 hidden units per lane and 32-unit group), the second layer runs on the streamed W2 fragments.  The
// [rows, F] hidden activations of the generic path (0.8 GB at the headline shape) never exist.  fp32 rows out.
struct EmbedArgs {
  Src3 src; int rows_per_ep, B, K, F;
  const float *w1, *b1, *b2;                     // [F, K], [F], [256]
  const unsigned *wimg;                          // packed W2
  float *out;                                    // [B * rows_per_ep, 256]
};
__global__ __launch_bounds__(BTHREADS) void wide_embed_kernel(EmbedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];      // [2][CHUNK_W] | w1 [F * K] | b1 [F] | b2 [256]
  unsigned *buf0 = lds, *buf1 = lds + CHUNK_W;
  float *w1s = reinterpret_cast<float *>(lds + 2 * CHUNK_W), *b1s = w1s + a.F * a.K, *b2s = b1s + a.F;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tok = lane & 15, g = lane >> 4;
  const long M = (long)a.B * a.rows_per_ep, base = (long)blockIdx.x * WG_TOK + wave * WTOK;
  for (int i = tid; i < a.F * a.K; i += BTHREADS) w1s[i] = a.w1[i];
  for (int i = tid; i < a.F; i += BTHREADS) b1s[i] = a.b1[i];
  for (int i = tid; i < D; i += BTHREADS) b2s[i] = a.b2[i];
  float xv[NT][8];
  long row[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    row[ct] = base + 16 * ct + tok;
    const long r = min(row[ct], M - 1);
    const int b = r / a.rows_per_ep, p = r % a.rows_per_ep;
    const float *x;
    if (p < a.src.n[0]) x = a.src.p[0] + ((long)b * a.src.n[0] + p) * a.K;
    else if (p < a.src.n[0] + a.src.n[1]) x = a.src.p[1] + ((long)b * a.src.n[1] + (p - a.src.n[0])) * a.K;
    else x = a.src.p[2] + ((long)b * a.src.n[2] + (p - a.src.n[0] - a.src.n[1])) * a.K;
#pragma unroll
    for (int k = 0; k < 8; ++k) xv[ct][k] = k < a.K ? x[k] : 0.f;
  }
  const int nchunk = a.F / 64;
  const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(a.wimg);
  u32x4 st[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) st[i] = wsrc[tid + i * BTHREADS];
  f32x4 y[NMT][NT];
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) y[mt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < nchunk; ++c) {
    unsigned *buf = (c & 1) ? buf1 : buf0;
#pragma unroll
    for (int i = 0; i < NST; ++i) reinterpret_cast<u32x4 *>(buf)[tid + i * BTHREADS] = st[i];
    __syncthreads();
    if (c + 1 < nchunk) {
#pragma unroll
      for (int i = 0; i < NST; ++i) st[i] = wsrc[(long)(c + 1) * (CHUNK_W / 4) + tid + i * BTHREADS];
    }
    const bf16x8 *fr = reinterpret_cast<const bf16x8 *>(buf) + lane;
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
      const int hbase = 64 * c + 32 * grp;
      bf16x8 hb[NT];                                 // hidden units hbase + 4 g + (0..3) and hbase + 16 + 4 g + (0..3)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        float h[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int u = hbase + 16 * (j >> 2) + 4 * g + (j & 3);
          float acc = b1s[u];
          for (int k = 0; k < a.K; ++k) acc = fmaf(xv[ct][k], w1s[u * a.K + k], acc);
#ifdef ALINE_RELU_SITE_EMB
          h[j] = relu_int(acc);
#else
          h[j] = relu_nn(acc);
#endif
        }
        const u32x4 v = {pack_bf16(h[0], h[1]), pack_bf16(h[2], h[3]), pack_bf16(h[4], h[5]), pack_bf16(h[6], h[7])};
        hb[ct] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) {
        const bf16x8 A = fr[(grp * 16 + mt) * 64];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) WMFMA(y[mt][ct], A, hb[ct]);
      }
    }
  }
#pragma unroll
  for (int ct = 0; ct < NT; ++ct)
    if (row[ct] < M) {
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt)
        *reinterpret_cast<f32x4 *>(a.out + row[ct] * D + 16 * mt + 4 * g) = y[mt][ct] + *reinterpret_cast<const f32x4 *>(b2s + 16 * mt + 4 * g);
    }
}

// GMM parameter maps + mixture log-likelihood from the raw head outputs raw[row][3 c + j] (model/head.py:152-186,
// 251-266; utils/eval.py:200-207): mean_c = raw[c][0], std_c = softplus(raw[c][1]) + std_min, weight = softmax_c(raw[c][2])
struct GmmRawArgs {
  const float *raw; int raw_stride; long rows; int C; float std_min;
  int nblk; long blk_stride;                     // raw = sum of nblk partial arrays blk_stride apart (0 / 1: a single one)
  float *mean, *sd, *wgt;                        // [rows, C] or null
  const float *value; long value_mod;            // value[(value_row0 + row) % value_mod] or null
  long value_row0;
  float *ll;                                     // [rows] or null
  unsigned *range_flag;                          // f16 range guard (common.h): raised when a log-likelihood is not finite; may be null
};
__global__ void gmm_raw_finish_kernel(GmmRawArgs a) {
  const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= a.rows) return;
  float r[48];
  for (int e = 0; e < 3 * a.C; ++e) {
    float s = a.raw[row * a.raw_stride + e];
    for (int k = 1; k < a.nblk; ++k) s += a.raw[k * a.blk_stride + row * a.raw_stride + e];
    r[e] = s;
  }
  float mxw = -INFINITY;
  for (int c = 0; c < a.C; ++c) mxw = fmaxf(mxw, r[3 * c + 2]);
  float sw = 0.f;
  for (int c = 0; c < a.C; ++c) sw += __expf(r[3 * c + 2] - mxw);
  const float v = (a.ll && a.value) ? a.value[(a.value_row0 + row) % a.value_mod] : 0.f;
  float mx2 = -INFINITY, lps[16];
  for (int c = 0; c < a.C; ++c) {
    const float mean = r[3 * c], sd = softplus_f(r[3 * c + 1]) + a.std_min, w = __expf(r[3 * c + 2] - mxw) / sw;
    if (a.mean) a.mean[row * a.C + c] = mean;
    if (a.sd) a.sd[row * a.C + c] = sd;
    if (a.wgt) a.wgt[row * a.C + c] = w;
    const float zz = (v - mean) / sd;
    lps[c] = -0.5f * zz * zz - logf(sd) - 0.91893853320467274178f + logf(w);
    mx2 = fmaxf(mx2, lps[c]);
  }
  if (a.ll && a.value) {
    float se = 0.f;
    for (int c = 0; c < a.C; ++c) se += __expf(lps[c] - mx2);
    const float ll = mx2 + logf(se);
    a.ll[row] = ll;
    if (!(fabsf(ll) <= 3.4e38f)) range_raise(a.range_flag, ALINE_RANGE_ACT);
  }
}

// ---- masked set-attention on the Q/K/V tile images -------------------------------------------------------
// One workgroup per episode, wave h = head h (head_dim 32 = one MFMA k-step: no zero padding).  The head's
// K fragments (keys x 32 channels) and V^T fragments (32 channels x keys) live in registers for the whole
// episode; V^T is built through an LDS transpose of the key rows.  Up to 64 keys.
constexpr int WNK = 64;
struct AttnArgs {
  Geo g;
  const u32x4 *Q, *K, *V;        // tile images over the B*N token rows
  u32x4 *A;                      // tile image of the concatenated head outputs
};

__global__ __launch_bounds__(NTHREADS) void wide_attention_kernel(AttnArgs a) {
  __shared__ unsigned short Vs[WNK][D + 8];      // V rows of the key tokens (all heads)
  __shared__ int keyrow[WNK];
  __shared__ int wave_cnt[8];
  __shared__ int s_base, s_nck, s_nak;
  const Geo &g = a.g;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tok = lane & 15, gg = lane >> 4;
  const int n_t = g.n_td + g.n_th;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < g.P; c0 += NTHREADS) {
    const int row = c0 + tid;
    const bool key = row < g.P && is_ctx(g, b, row);
    const unsigned long long bal = __ballot(key);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    const int k = off + __popcll(bal & ((1ull << lane) - 1ull));
    if (key && k < WNK) keyrow[k] = row;
    __syncthreads();
    if (tid == 0) { int s = 0; for (int w = 0; w < 8; ++w) s += wave_cnt[w]; s_base += s; }
    __syncthreads();
  }
  if (tid == 0) {
    int n = min(s_base, WNK);
    s_nck = n;
    for (int j = 0; j < n_t; ++j)
      if ((!g.tmask || g.tmask[j]) && n < WNK) keyrow[n++] = g.P + j;
    s_nak = n;
  }
  __syncthreads();
  const int n_ck = s_nck, n_ak = s_nak;
  const long ep = (long)b * g.N;
  // V rows -> LDS (zero rows beyond n_ak)
  for (int i = tid; i < WNK * (D / 8); i += NTHREADS) {
    const int j = i / (D / 8), pc = i % (D / 8), ks = pc >> 2, gq = pc & 3;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (j < n_ak) v = a.V[piece(ep + keyrow[j], ks, gq)];
    *reinterpret_cast<u32x2 *>(&Vs[j][32 * ks + 4 * gq]) = (u32x2){v[0], v[1]};
    *reinterpret_cast<u32x2 *>(&Vs[j][32 * ks + 16 + 4 * gq]) = (u32x2){v[2], v[3]};
  }
  __syncthreads();
  const int h = wave;
  const int nkt = (n_ak + 15) >> 4;            // key tiles in use (<= 4)
  // K fragments of this head: rows = keys 16 kt + tok, k = channel pi(0, g, j)
  bf16x8 kf[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int j = 16 * kt + tok;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (j < n_ak) v = a.K[piece(ep + keyrow[j], h, gg)];
    kf[kt] = __builtin_bit_cast(bf16x8, v);
  }
  // V^T fragments: rows = channel 16 mt + tok, k-step s covers keys 32 s + 16 (j>>2) + 4 g + (j&3)
  bf16x8 vf[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      unsigned short e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = Vs[32 * s + 16 * (j >> 2) + 4 * gg + (j & 3)][HD * h + 16 * mt + tok];
      const u32x4 v = {(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16),
                       (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16)};
      vf[mt][s] = __builtin_bit_cast(bf16x8, v);
    }
  const int ntiles = (g.N + 15) >> 4;
  for (int ti = 0; ti < ntiles; ++ti) {
    const int row = 16 * ti + tok;
    const bool valid = row < g.N;
    const int rr = valid ? row : g.N - 1;
    const bool isq = rr < g.P && !is_ctx(g, b, rr);
    const int nv = isq ? n_ak : n_ck;
    // Q^T fragment of this head (already scaled by log2(e)/sqrt(hd))
    const bf16x8 qf = __builtin_bit_cast(bf16x8, a.Q[piece(ep + rr, h, gg)]);
    f32x4 s[4];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s[kt][r] = (16 * kt + 4 * gg + r) < nv ? 0.f : -INFINITY;
      if (kt < nkt) {
        WMFMA(s[kt], kf[kt], qf);
        mx = fmaxf(mx, fmaxf(fmaxf(s[kt][0], s[kt][1]), fmaxf(s[kt][2], s[kt][3])));
      }
    }
    {
      auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
      r = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[kt][r] = kt < nkt ? __builtin_amdgcn_exp2f(s[kt][r] - mx) : 0.f;
        sum += s[kt][r];
      }
    const float inv = __builtin_amdgcn_rcpf(group_sum4(sum));
    f32x4 o[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    const bf16x8 p0 = acc_to_frag(s[0], s[1]);
    WMFMA(o[0], vf[0][0], p0);
    WMFMA(o[1], vf[1][0], p0);
    if (nkt > 2) {
      const bf16x8 p1 = acc_to_frag(s[2], s[3]);
      WMFMA(o[0], vf[0][1], p1);
      WMFMA(o[1], vf[1][1], p1);
    }
    if (valid) a.A[piece(ep + row, h, gg)] = __builtin_bit_cast(u32x4, acc_to_frag(o[0] * inv, o[1] * inv));
  }
}

}  // namespace wide
