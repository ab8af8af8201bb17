// x3 path: d = 256 (8 heads of 32) at REFERENCE precision on the f16 matrix pipe.
//
// Every matrix product of the step (model/embedder.py second layers excepted: they run once per rollout on the
// exact-fp32 GEMM) is computed as an fp32-grade 3-term split on v_mfma_f32_16x16x32_f16:
//     a = a_hi + a_lo,  b = b_hi + b_lo   (a_hi = f16(a), a_lo = f16(a - a_hi): 22 significant bits)
//     a b ~= a_hi b_hi + a_hi b_lo + a_lo b_hi        (dropped: a_lo b_lo <= 2^-24 |a b|, the size of fp32 rounding)
// accumulated in fp32.  Measured on the reference fixtures this is indistinguishable from the exact-fp32 pipeline
// (posterior log-likelihood within 2e-5, tests/test_x3_gpu.py), at 3 MFMA passes on the 2.5 PFLOP/s pipe instead of
// the 157 TFLOP/s fp32 MFMA.  Weights are pre-scaled by 2^8 at pack time (exact; keeps both halves of a weight in
// f16's normal range), products are scaled back in the epilogues.
//
// Operand scheme (as wide.h): a token tile is X^T [features x 16 tokens]; every linear is Y^T = W X^T, so an
// accumulator tile is the B operand of the next product with k order pi(ks,g,j) = 32 ks + 16 (j>>2) + 4 g + (j&3);
// weights are A fragments pre-permuted at pack time, as PAIRS (hi fragment 1 KB | lo fragment 1 KB).
//
// Kernels (one launch each per layer and step; activations travel as split-f16 TILE IMAGES, episodes padded to whole
// 16-row tiles, so tiles never straddle episodes and every wave-level load / store is whole KBs):
//   keys_kernel    per step: the key list of every episode (context rows, then the visible target rows)
//   kv_all_kernel / kv_split_kernel   K / V of the key rows only, written as the A fragments the attention needs
//   layer_kernel   Q projection, masked set-attention, out-projection, LN1, FFN, LN2 of a token tile, all in
//                  registers: 8 waves x one 16-token tile, weights streamed through LDS by LDS-DMA (32 KB chunks
//                  of 16 pairs, 3 buffers); persistent workgroups walk the tile list, the weight stream is cyclic
//   head_kernel    acquisition logits / one GMM head (hidden layer in registers, [F -> 1|3] in fp32 FMA)
// (no include guard: x3.h includes this file once per width, with X3_NS / X3_D / X3_HD / X3_THREADS set)

namespace X3_NS {

using img::u32x4;
using img::u32x2;
using img::group_sum4;

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int D = X3_D, HD = X3_HD, H = D / HD, NMT = D / 16, NKS = D / 32, WNK = 64;
constexpr int KPH = HD / 32, MPH = HD / 16;           // k-steps / 16-channel tiles of one head
constexpr int CPK = NMT / 16;                         // chunks per k-step of a product with D outputs (16 pairs per chunk)
constexpr int PC = NKS * CPK;                         // chunks of a D x D product
constexpr int W1C = NKS / 8;                          // chunks of the W1 pairs of 32 hidden units (2 tiles x NKS k-steps)
constexpr int THREADS = X3_THREADS, WAVES = THREADS / 64;
constexpr int CHUNK_PAIRS = 16, CHUNK_BYTES = CHUNK_PAIRS * 2048, CHUNK_WORDS = CHUNK_BYTES / 4;
constexpr int NBUF = 4, PD = 3;                       // LDS ring, chunks issued ahead of the one in use
constexpr int PIECES_PER_WAVE = CHUNK_BYTES / 1024 / WAVES;
constexpr float WSCALE = 256.f, WINV = 1.f / 256.f;
constexpr long KV_VOFF = NKS * 512;                   // u32x4 per episode: K pairs [channel k-step NKS][kt 4] | V^T pairs [i NMT][s 2]
constexpr long KV_EP = KV_VOFF + NMT * 256;

#ifndef XMFMA
#define XMFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0)
#endif
// acc += (ah + al) (bh + bl) without the lo*lo term; small terms first
__device__ __forceinline__ void mfma3(f32x4 &acc, const f16x8 &ah, const f16x8 &al, const f16x8 &bh, const f16x8 &bl) {
  XMFMA(acc, al, bh);
  XMFMA(acc, ah, bl);
  XMFMA(acc, ah, bh);
}

// (a, b) -> packed f16 hi halves, packed f16 lo halves (a - hi, b - hi: exact in fp32, then rounded; round to nearest
// even both times).  The residuals are one v_fma_mix_f32 each (its f16 operand read straight out of the packed register):
// 4 instructions per pair instead of the 6 of convert-back-and-subtract.
__device__ __forceinline__ void split2(float a, float b, unsigned &hi, unsigned &lo) {
  const f32x2 v = {a, b};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(a));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(b));
  const f32x2 r = {r0, r1};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}
// two accumulator tiles (features 16 m + 4 g + r, 16 (m+1) + 4 g + r) -> the hi / lo B fragments of their k-step
__device__ __forceinline__ void split_frag(const f32x4 &a, const f32x4 &b, f16x8 &hi, f16x8 &lo) {
  unsigned h0, h1, h2, h3, l0, l1, l2, l3;
  split2(a[0], a[1], h0, l0);
  split2(a[2], a[3], h1, l1);
  split2(b[0], b[1], h2, l2);
  split2(b[2], b[3], h3, l3);
  hi = __builtin_bit_cast(f16x8, (u32x4){h0, h1, h2, h3});
  lo = __builtin_bit_cast(f16x8, (u32x4){l0, l1, l2, l3});
}
// the fp32 values a fragment pair stands for, elements 4 hf .. 4 hf + 3: hi + lo, one v_fma_mix_f32 each (both f16
// operands read straight out of the packed registers)
__device__ __forceinline__ f32x4 frag_value(const f16x8 &hi, const f16x8 &lo, int hf) {
  const u32x4 h = __builtin_bit_cast(u32x4, hi), l = __builtin_bit_cast(u32x4, lo);
  f32x4 v;
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    float a, b;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(a) : "v"(h[2 * hf + w]), "v"(l[2 * hf + w]));
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(b) : "v"(h[2 * hf + w]), "v"(l[2 * hf + w]));
    v[2 * w] = a; v[2 * w + 1] = b;
  }
  return v;
}
// max(a, b) as v_med3_f32(a, b, +inf): llvm.maxnum (fmaxf) must quiet signalling NaNs in IEEE mode, so hipcc puts a canonicalising
// `v_max_f32 x, x, x` in front of every fmaxf operand it cannot prove quiet -- MFMA accumulators and permlane outputs, i.e. every
// operand of the softmax maxima: 6 of the ~14 instructions of a head's max reduction.  The median form is one instruction, is visible to
// the compiler's hazard recogniser (unlike inline asm behind an MFMA) and ignores a NaN operand like maxnum does.
// (the +inf travels in a register the optimiser cannot see through: with the literal it folds the median back into maxnum)
__device__ __forceinline__ float opaque_inf() { float v = __builtin_inff(); asm("" : "+v"(v)); return v; }
__device__ __forceinline__ float vmax(float a, float b, float inf) { return __builtin_amdgcn_fmed3f(a, b, inf); }
__device__ __forceinline__ float group_max4(float v, float inf) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = vmax(__uint_as_float(r[0]), __uint_as_float(r[1]), inf);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return vmax(__uint_as_float(r[0]), __uint_as_float(r[1]), inf);
}

// ---- images ---------------------------------------------------------------------------------------------------
// activation image, in 16-byte pieces: [tile][ks][hi | lo][lane = 16 g + row % 16]; a piece holds features
// 32 ks + 4 g + (0..3) and 32 ks + 16 + 4 g + (0..3) of its token row (= the B fragment element order)
__host__ __device__ inline long img_pieces(long tiles) { return tiles * (NKS * 2 * 64); }
__device__ __forceinline__ long xpiece(long tile, int ks, int hl, int lane) { return ((tile * NKS + ks) * 2 + hl) * 64 + lane; }

// weight image of a layer (32-bit words): chunks [Q PC][K PC][V PC][OUT PC][FFN: per 32 hidden units W1 (W1C chunks) | W2 (CPK)],
// then fp32 parameters bq (pre-scaled) bk bv | bo | b1 | b2 | ln1w ln1b ln2w ln2b.  A chunk is 16 fragment pairs; chunk c of a
// product with D outputs holds k-step c / CPK, output tiles 16 (c % CPK) + p; pair q = 16 j + p of a W1 group: k-step q >> 1,
// hidden tile q & 1.
__host__ __device__ inline int layer_chunks(int F) { return 4 * PC + (F / 32) * (W1C + CPK); }
__host__ __device__ inline int layer_params(int F) { return 9 * D + F; }
__host__ __device__ inline long layer_words(int F) { return (long)layer_chunks(F) * CHUNK_WORDS + layer_params(F); }
// head image (acquisition head, GMM heads): F/32 chunks of W1, then b1 [F] | w2 [3][F] | b2 [4]
__host__ __device__ inline int head_chunks(int F) { return (F / 32) * W1C; }
__host__ __device__ inline int head_params(int F) { return 4 * F + 4; }
__host__ __device__ inline int head_lds_params(int F) { return 4 * F; }
__host__ __device__ inline long head_words(int F) { return (long)head_chunks(F) * CHUNK_WORDS + head_params(F); }
__host__ __device__ inline long image_words(int L, int F, int C) { return (long)L * layer_words(F) + (long)(1 + C) * head_words(F); }

struct PackArgs {
  int L, F, C;
  const float *in_proj_w[8], *in_proj_b[8], *out_proj_w[8], *out_proj_b[8], *lin1_w[8], *lin1_b[8], *lin2_w[8],
      *lin2_b[8], *n1w[8], *n1b[8], *n2w[8], *n2b[8];
  const float *acq_w1, *acq_b1, *acq_w2, *acq_b2;
  const float *gmm_w1[16], *gmm_b1[16], *gmm_w2[16], *gmm_b2[16];
  unsigned *out;
  unsigned *range_flag;
  int time_token;      // model.time_token: the acquisition head's W1 is [F, d + 1] (model/head.py:24-25); its last column goes to the free
                       // slot [2 F, 3 F) of that head's parameters and becomes a bias term per step (HeadArgs.tau)
};

// word e (0..511) of the pair (rows row0.., k-step ks) of the row-major weight W [*, K]: hi fragment, lo fragment
__device__ __forceinline__ unsigned pair_word(const float *W, int K, int row0, int ks, int e, float scale, unsigned *range_flag = nullptr) {
  const int hl = e >> 8, lane = (e & 255) >> 2, w = e & 3, g = lane >> 4, j0 = 2 * w;
  const int k0 = 32 * ks + 16 * (j0 >> 2) + 4 * g + (j0 & 3);
  const float *p = W + (long)(row0 + (lane & 15)) * K + k0;
  unsigned hi, lo;
  if (f16_out_of_range(p[0] * scale) || f16_out_of_range(p[1] * scale)) range_raise(range_flag, ALINE_RANGE_WEIGHT);
  split2(p[0] * scale, p[1] * scale, hi, lo);
  return hl ? lo : hi;
}

__global__ void pack_kernel(PackArgs a) {
  const long lw = layer_words(a.F), hw = head_words(a.F), total = image_words(a.L, a.F, a.C);
  const float qscale = rsqrtf((float)HD) * 1.44269504088896340736f;   // softmax runs in exp2
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    unsigned v = 0;
    if (i < a.L * lw) {
      const int l = i / lw;
      const long o = i % lw, nfw = (long)layer_chunks(a.F) * CHUNK_WORDS;
      if (o < nfw) {
        const int ch = o / CHUNK_WORDS, cw = o % CHUNK_WORDS, p = cw >> 9, e = cw & 511;
        if (ch < 4 * PC) {
          const int mat = ch / PC, c = ch % PC, ks = c / CPK, row0 = 16 * (16 * (c % CPK) + p);
          if (mat < 3) v = pair_word(a.in_proj_w[l] + (long)mat * D * D, D, row0, ks, e, (mat == 0 ? qscale : 1.f) * WSCALE, a.range_flag);
          else v = pair_word(a.out_proj_w[l], D, row0, ks, e, WSCALE, a.range_flag);
        } else {
          const int c = (ch - 4 * PC) / (W1C + CPK), j = (ch - 4 * PC) % (W1C + CPK);
          if (j < W1C) { const int q = 16 * j + p; v = pair_word(a.lin1_w[l], D, 32 * c + 16 * (q & 1), q >> 1, e, WSCALE, a.range_flag); }
          else v = pair_word(a.lin2_w[l], a.F, 16 * (16 * (j - W1C) + p), c, e, WSCALE, a.range_flag);
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
    } else {
      const long oh = i - a.L * lw;
      const int k = oh / hw;                          // 0: acquisition head, 1 + c: GMM head c
      const long o = oh % hw, nfw = (long)head_chunks(a.F) * CHUNK_WORDS;
      const float *w1 = k == 0 ? a.acq_w1 : a.gmm_w1[k - 1], *b1 = k == 0 ? a.acq_b1 : a.gmm_b1[k - 1];
      const float *w2 = k == 0 ? a.acq_w2 : a.gmm_w2[k - 1], *b2 = k == 0 ? a.acq_b2 : a.gmm_b2[k - 1];
      const int nout = k == 0 ? 1 : 3;
      const int ldw1 = D + ((k == 0 && a.time_token) ? 1 : 0);
      if (o < nfw) {
        const int ch = o / CHUNK_WORDS, cw = o % CHUNK_WORDS, p = cw >> 9, e = cw & 511, q = 16 * (ch % W1C) + p;
        v = pair_word(w1, ldw1, 32 * (ch / W1C) + 16 * (q & 1), q >> 1, e, WSCALE, a.range_flag);
      } else {
        const int p = o - nfw;
        float f = p < a.F ? b1[p] : p < (1 + nout) * a.F ? w2[p - a.F] : (p >= 4 * a.F && p < 4 * a.F + nout) ? b2[p - 4 * a.F] : 0.f;
        if (k == 0 && a.time_token && p >= 2 * a.F && p < 3 * a.F) f = w1[(long)(p - 2 * a.F) * ldw1 + D];      // the time column
        v = __float_as_uint(f);
      }
    }
    a.out[i] = v;
  }
}

// ---- X0 image from the cached fp32 point embeddings (model/embedder.py:128-214): Ex (+ Ey on context rows), theta
// tokens; rows beyond N of an episode's last tile are zero.  One thread = one (piece hi, piece lo).
struct AsmArgs {
  Geo g; int tpe;
  const float *Ex, *Ey; int ey_rows; const float *theta_tokens;
  u32x4 *X;
  unsigned *range_flag;
};
// the layer-0 input is the one split operand no LayerNorm has bounded: checked where it is assembled (off the hot path)
__device__ __forceinline__ void range_check8(unsigned *flag, const f32x4 &a, const f32x4 &b) {
  float m = fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3])));
  m = fmaxf(m, fmaxf(fmaxf(fabsf(b[0]), fabsf(b[1])), fmaxf(fabsf(b[2]), fabsf(b[3]))));
  const float nanp = (a[0] + a[1] + a[2] + a[3] + b[0] + b[1] + b[2] + b[3]) * 0.f;     // NaN iff any is NaN / inf
  if (!(m < 65504.f) || nanp != nanp) range_raise(flag, ALINE_RANGE_ACT);
}
__device__ __forceinline__ void store_split8(u32x4 *X, long tile, int ks, int lane, const f32x4 &lo4, const f32x4 &hi4) {
  f16x8 h, l;
  split_frag(lo4, hi4, h, l);
  X[xpiece(tile, ks, 0, lane)] = __builtin_bit_cast(u32x4, h);
  X[xpiece(tile, ks, 1, lane)] = __builtin_bit_cast(u32x4, l);
}
__device__ __forceinline__ void embed_row8(const AsmArgs &a, int b, int row, int c, f32x4 &lo, f32x4 &hi) {
  const Geo &g = a.g;
  const int eb = g.inst_B > 0 ? b % g.inst_B : b;      // (instance mode of the backward: b is a (step, episode) pair, the embeddings are per episode)
  if (row < g.P + g.n_td) {
    const float *e = a.Ex + ((long)eb * (g.P + g.n_td) + row) * D + c;
    lo = *reinterpret_cast<const f32x4 *>(e); hi = *reinterpret_cast<const f32x4 *>(e + 16);
    if (row < g.P && is_ctx(g, b, row)) {
      const float *y = a.Ey + ((long)eb * a.ey_rows + row) * D + c;
      lo += *reinterpret_cast<const f32x4 *>(y); hi += *reinterpret_cast<const f32x4 *>(y + 16);
    }
  } else {
    const float *t = a.theta_tokens + (row - g.P - g.n_td) * D + c;
    lo = *reinterpret_cast<const f32x4 *>(t); hi = *reinterpret_cast<const f32x4 *>(t + 16);
  }
}
__global__ void assemble_kernel(AsmArgs a) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long tiles = (long)a.g.B * a.tpe;
  if (i >= tiles * NKS * 64) return;
  const int lane = i & 63, ks = (i >> 6) % NKS, gq = lane >> 4;
  const long tile = (i >> 6) / NKS;
  const int b = tile / a.tpe, row = (int)(tile % a.tpe) * 16 + (lane & 15);
  f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
  if (row < a.g.N) embed_row8(a, b, row, 32 * ks + 4 * gq, lo, hi);
  range_check8(a.range_flag, lo, hi);
  store_split8(a.X, tile, ks, lane, lo, hi);
}
// between steps the input image changes in ONE row per episode: the point chosen at the previous step (role ==
// `order`) became a context point, its row becomes Ex + Ey.  One wave per episode.
__global__ __launch_bounds__(64) void patch_row_kernel(AsmArgs a, int order) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int slot = -1;
  for (int p = lane; p < a.g.P; p += 64)
    if (a.g.role[(long)b * a.g.P + p] == order) slot = p;
  slot = __reduce_max_sync(~0ull, slot);
  if (slot < 0 || lane >= D / 8) return;
  const int ks = lane >> 2, gq = lane & 3;
  f32x4 lo, hi;
  embed_row8(a, b, slot, 32 * ks + 4 * gq, lo, hi);
  range_check8(a.range_flag, lo, hi);
  store_split8(a.X, (long)b * a.tpe + (slot >> 4), ks, gq * 16 + (slot & 15), lo, hi);
}

// ---- key list of every episode (model/encoder.py:83-126): context rows in slot order, then the visible targets --
// Also, for the K / V kernel: keypos [B][16 tpe] = position of a token row in its episode's key list (-1: not a key) -- the
// layer kernel files the output rows that are keys into the KEY IMAGE of the next layer with it --, and the key image of
// layer 0 itself: the key rows of the input image gathered into [B][WNK / 16 key tiles] tiles of the usual piece layout, so
// that kv_kernel reads whole KBs instead of 64 scattered 16-byte pieces per key row (a row of the token image shares each
// of its 64-byte lines with three other rows: 4x read amplification; the gather was most of kv_kernel's time in round 2).
__global__ __launch_bounds__(256) void keys_kernel(Geo g, int tpe, int *__restrict__ keyrow, int *__restrict__ kcnt,
                                                   short *__restrict__ keypos, const u32x4 *__restrict__ X0, u32x4 *__restrict__ KX) {
  __shared__ int wave_cnt[4];
  __shared__ int s_base, s_n;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_base = 0;
  for (int i = tid; i < 16 * tpe; i += 256) keypos[(long)b * 16 * tpe + i] = -1;
  __syncthreads();
  for (int c0 = 0; c0 < g.P; c0 += 256) {
    const int row = c0 + tid;
    const bool key = row < g.P && is_ctx(g, b, row);
    const unsigned long long bal = __ballot(key);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    const int k = off + __popcll(bal & ((1ull << lane) - 1ull));
    if (key && k < WNK) { keyrow[b * WNK + k] = row; keypos[(long)b * 16 * tpe + row] = (short)k; }
    __syncthreads();
    if (tid == 0) s_base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (tid == 0) {
    int n = min(s_base, WNK);
    kcnt[2 * b] = n;
    const int n_t = g.n_td + g.n_th;
    for (int j = 0; j < n_t; ++j)
      if ((!g.tmask || g.tmask[j]) && n < WNK) { keyrow[b * WNK + n] = g.P + j; keypos[(long)b * 16 * tpe + g.P + j] = (short)n; ++n; }
    kcnt[2 * b + 1] = n;
    s_n = n;
  }
  __syncthreads();
  // layer 0's key image: key k -> tile b * (WNK / 16) + k / 16, row k % 16; PPR = 8 NKS pieces (ks, hi | lo, g) of 16 bytes per key row
  constexpr int PPR = 8 * NKS;
  const int n = s_n;
  for (int i0 = tid; i0 < n * PPR; i0 += 256 * 8) {          // 8 pieces per thread in flight (the loads miss L2: one at a time took 28 us)
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(i0 + 256 * u, n * PPR - 1), k = i / PPR, e = i % PPR, ks = e >> 3, hl = (e >> 2) & 1, gq = e & 3;
      const int row = keyrow[b * WNK + k];
      v[u] = X0[xpiece((long)b * tpe + (row >> 4), ks, hl, 16 * gq + (row & 15))];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + 256 * u, k = i / PPR, e = i % PPR, ks = e >> 3, hl = (e >> 2) & 1, gq = e & 3;
      if (i < n * PPR) KX[xpiece((long)b * (WNK / 16) + (k >> 4), ks, hl, 16 * gq + (k & 15))] = v[u];
    }
  }
}

// ---- the weight stream ------------------------------------------------------------------------------------------
__device__ __forceinline__ void glds16(const void *gsrc, void *ldst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                   (__attribute__((address_space(3))) void *)ldst, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {   // all but the N youngest vector-memory operations of this wave are done
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

__device__ __forceinline__ unsigned lds_addr(const void *p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char *)(p);
}

// Cyclic stream of 32 KB chunks through a ring of NBUF = 4 LDS buffers, issued PD = 3 chunks ahead of the one in use.
// Every wave moves its 4 KB share of a chunk with 4 LDS-DMA instructions.  sync(), once per chunk u: this wave's
// pieces of the chunks <= u + 1 have landed (counted vmcnt -- only chunk u + 2 may still be in flight; operations
// issued later only make the wait more conservative), then the workgroup barrier: every wave's pieces of chunks u and
// u + 1 are visible and nobody reads chunk u - 1 any more, so its buffer may take chunk u + 3 (issue(), once per wave
// and chunk, any time before the next sync).  Making chunk u + 1 readable already lets the last MFMA batch of chunk u
// prefetch the first fragments of chunk u + 1, so no wave starts a chunk by waiting for LDS.
// What the stream costs (timing experiments, tools/x3_variants.sh, layer kernel at the headline shape): 165 us of a 770 us
// launch disappear with the LDS-DMA instructions removed, none with their vmcnt wait removed, 30 us with the barrier
// removed; issuing from four waves only (one per SIMD) or from every wave in a different MFMA batch than its SIMD
// partner changes nothing (+-4 %) -- part of the cost is clock: the chip holds 2.1 GHz with the stream, 2.35 without.
// In-kernel phase stamps (X3_STAMPS diagnostic build only, tools/x3_stamps.py): s_memtime deltas accumulated per wave.
//   0 LDS wait (touch)   1 MFMA batch (+ next reads)   2 DMA issue   3 vmcnt wait   4 barrier   5 hidden / epilogue VALU
//   6 attention   7 layer norms + image I/O   8 whole kernel
#ifdef X3_STAMPS
#define X3_NSTAMP 9
struct Stamps {
  unsigned long long t_prev, acc[X3_NSTAMP];
  __device__ __forceinline__ void start() { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory"); }
  __device__ __forceinline__ void lap(int k) {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    acc[k] += t - t_prev;
    t_prev = t;
  }
};
#define X3_LAP(st, k) (st).stamps.lap(k)
#else
#define X3_LAP(st, k)
#endif

template <class SrcFn>
struct Stream {
#ifdef X3_STAMPS
  Stamps stamps;
#endif
  SrcFn src;              // position in the cyclic sequence -> first byte of the chunk
  char *ring;
  int seq_len, s_issue, b_issue, b_use;
  unsigned lane_off, wave_off;
  static constexpr int dma_slot = 0;   // the MFMA batch (0..3) of a chunk with which a wave issues its pieces of the stream
  __device__ __forceinline__ void issue() {
    // buffer_load ... lds with the chunk's base in a scalar buffer descriptor and a constant 32-bit lane offset: no per-piece
    // vector address arithmetic (global_load_lds took a 64-bit per-lane address per piece); the whole rollout 80.2 -> 77.3 ms
    // (kv_kernel / head_kernel, which are stream-bound, gain; the layer kernel is unchanged: profiles/r03_x3_timing_experiments.txt)
    {
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(src(s_issue)), 0, CHUNK_BYTES, 0x00020000);
      char *d = ring + b_issue * CHUNK_BYTES + wave_off;
#pragma unroll
      for (int i = 0; i < PIECES_PER_WAVE; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(d + i * 1024), 16, lane_off, wave_off + i * 1024, 0, 0);
    }
    s_issue = s_issue + 1 == seq_len ? 0 : s_issue + 1;
    b_issue = (b_issue + 1) & (NBUF - 1);
  }
  __device__ __forceinline__ void start() {
#pragma unroll
    for (int i = 0; i < PD; ++i) issue();
  }
  __device__ __forceinline__ void sync() {
    wait_vmcnt<PIECES_PER_WAVE *(PD - 2)>();
    X3_LAP(*this, 3);
    __builtin_amdgcn_s_barrier();
    X3_LAP(*this, 4);
  }
  // LDS byte address of this lane's 16 bytes of fragment 0 of the chunk in use / of the next one
  __device__ __forceinline__ const f16x8 *cur() const { return reinterpret_cast<const f16x8 *>(ring + b_use * CHUNK_BYTES + lane_off); }
  __device__ __forceinline__ const f16x8 *nxt() const { return reinterpret_cast<const f16x8 *>(ring + ((b_use + 1) & (NBUF - 1)) * CHUNK_BYTES + lane_off); }
  __device__ __forceinline__ void advance() { b_use = (b_use + 1) & (NBUF - 1); }
  __device__ __forceinline__ void finish() { wait_vmcnt<0>(); }   // no LDS-DMA may outlive the workgroup
};
template <class SrcFn>
__device__ __forceinline__ Stream<SrcFn> make_stream(SrcFn src, char *ring, int seq_len, int tid) {
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  Stream<SrcFn> s{
#ifdef X3_STAMPS
                  Stamps{},
#endif
                  src, ring, seq_len, 0, 0, 0, (unsigned)(tid & 63) * 16u, (unsigned)wave * (unsigned)(PIECES_PER_WAVE * 1024)};
  return s;
}

// The 16 fragment pairs of a chunk go through a register ring in 4 batches of 4 pairs (2 x 32 VGPRs): the 8
// ds_read_b128 of batch k + 1 (after the last batch: of the first batch of the NEXT chunk, which sync() has already made
// readable) are issued in front of the 12 MFMAs of batch k and land while they run.  The sched_group_barriers pin that
// order (left alone the scheduler sinks every read to its use and the wave waits a full LDS round trip per fragment).
struct FragRing { f16x8 hi[2][4], lo[2][4]; };
__device__ __forceinline__ void fetch_batch(FragRing &r, int slot, const f16x8 *fr, int batch) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r.hi[slot][j] = fr[(2 * (4 * batch + j)) * 64];
    r.lo[slot][j] = fr[(2 * (4 * batch + j) + 1) * 64];
  }
}
// The compiler's wait for a batch goes where its registers are first used.  touch_batch() is that use, placed BEFORE the
// reads of the following batch are issued: with an LDS-DMA pending hipcc waits lgkmcnt(0) instead of a counted wait, and
// a wait placed after the next batch's reads would drain those too (a full LDS round trip per batch).
__device__ __forceinline__ void touch_batch(FragRing &r, int slot) {
  asm volatile("" : "+v"(r.hi[slot][0]), "+v"(r.lo[slot][0]), "+v"(r.hi[slot][1]), "+v"(r.lo[slot][1]),
                    "+v"(r.hi[slot][2]), "+v"(r.lo[slot][2]), "+v"(r.hi[slot][3]), "+v"(r.lo[slot][3]));
}
__device__ __forceinline__ void prime(FragRing &r, const f16x8 *base) { fetch_batch(r, 0, base, 0); }
// One chunk: body(p, A_hi, A_lo) consumes pair p with exactly 3 MFMAs.  Precondition: ring slot 0 holds (or is being
// read with) batch 0 of `cur`, and the sync() of this chunk has been passed.  Ends with the sync() of the next chunk.
template <bool PREFETCH_NEXT, class St, class Body>
__device__ __forceinline__ void chunk_pipe(St &st, FragRing &r, const f16x8 *cur, const f16x8 *nxt, Body body) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    __builtin_amdgcn_sched_barrier(0);
    touch_batch(r, k & 1);
#ifdef X3_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    X3_LAP(st, 0);
    __builtin_amdgcn_sched_barrier(0);
#if X3_INTERLEAVE
    // The fragment reads of the next batch and the wave's LDS-DMA pieces go BETWEEN the MFMAs of the batch instead of in front of /
    // behind them (a wave that issues 8 reads or 8 pieces in a row leaves the matrix pipe of its SIMD idle meanwhile; at d = 512 it
    // has no partner wave to fill it): a read behind each of the first eight MFMAs; in the batch that carries the stream, two reads
    // behind each of the first four and a piece behind each of the following ones.  Same-box A/B (profiles/r03_x3_timing_experiments.txt):
    // x5 cfg5 35.2 -> 33.2 ms (F = 128), 101.3 -> 92.5 ms (F = 2048); x3 layer kernel 697 -> 661 us.
    const bool reads = k < 3 || PREFETCH_NEXT;
    if (k < 3) fetch_batch(r, (k + 1) & 1, cur, k + 1);
    else if (PREFETCH_NEXT) fetch_batch(r, 0, nxt, 0);
    if (k == st.dma_slot) st.issue();
#pragma unroll
    for (int j = 0; j < 4; ++j) body(4 * k + j, r.hi[k & 1][j], r.lo[k & 1][j]);
    constexpr int NP = PIECES_PER_WAVE < 8 ? PIECES_PER_WAVE : 8;
#define X3_MFMA1() __builtin_amdgcn_sched_group_barrier(0x008, 1, 0)
#define X3_READ(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0)
#define X3_DMA1() __builtin_amdgcn_sched_group_barrier(0x020, 1, 0)
    if (k != st.dma_slot) {
      if (reads) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { X3_MFMA1(); X3_READ(1); }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
      }
    } else {                                         // two reads behind each of four MFMAs, then the pieces
#pragma unroll
      for (int i = 0; i < 4; ++i) { X3_MFMA1(); if (reads) X3_READ(2); }
#pragma unroll
      for (int i = 0; i < NP; ++i) { X3_MFMA1(); X3_DMA1(); }
      if constexpr (NP < 8) __builtin_amdgcn_sched_group_barrier(0x008, 8 - NP, 0);
    }
#undef X3_MFMA1
#undef X3_READ
#undef X3_DMA1
    __builtin_amdgcn_sched_barrier(0);
    X3_LAP(st, 1);          // (stamps: the DMA pieces are inside the batch now, lap 2 stays empty)
#else
    if (k < 3) {
      fetch_batch(r, (k + 1) & 1, cur, k + 1);
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
    } else if (PREFETCH_NEXT) {
      fetch_batch(r, 0, nxt, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) body(4 * k + j, r.hi[k & 1][j], r.lo[k & 1][j]);
    __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
    __builtin_amdgcn_sched_barrier(0);
    X3_LAP(st, 1);
    if (k == st.dma_slot) { st.issue(); X3_LAP(st, 2); }
#endif
  }
  __builtin_amdgcn_sched_barrier(0);
  st.sync();
}
// Convention: a kernel calls st.sync() once after st.start(); from then on every chunk carries the sync that publishes
// the chunk after the next one, so a phase starts with its first chunk already readable.
// A run of N chunks (one phase of a kernel): pair p of chunk i is consumed by body(i, p, A_hi, A_lo).
template <int N, class St, class Body>
__device__ __forceinline__ void chunk_run(St &st, FragRing &r, Body body) {
  prime(r, st.cur());
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const f16x8 *cur = st.cur(), *nx = st.nxt();
    st.advance();
    if (i + 1 < N) chunk_pipe<true>(st, r, cur, nx, [&](int p, const f16x8 &ah, const f16x8 &al) { body(i, p, ah, al); });
    else chunk_pipe<false>(st, r, cur, nx, [&](int p, const f16x8 &ah, const f16x8 &al) { body(i, p, ah, al); });
  }
}
// The same with hook(i) behind chunk i (global loads that should be in flight while the following chunks compute)
template <int N, class St, class Body, class Hook>
__device__ __forceinline__ void chunk_run_hook(St &st, FragRing &r, Body body, Hook hook) {
  prime(r, st.cur());
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const f16x8 *cur = st.cur(), *nx = st.nxt();
    st.advance();
    if (i + 1 < N) chunk_pipe<true>(st, r, cur, nx, [&](int p, const f16x8 &ah, const f16x8 &al) { body(i, p, ah, al); });
    else chunk_pipe<false>(st, r, cur, nx, [&](int p, const f16x8 &ah, const f16x8 &al) { body(i, p, ah, al); });
    hook(i);
  }
}
// A wave without a tile in this round: it computes nothing and reads no fragments (its SIMD partner then has the matrix pipe
// to itself), but keeps its share of the weight stream and the chunk barriers going -- exactly one issue / sync / advance per chunk.
template <class St>
__device__ __forceinline__ void idle_chunks(St &st, int nchunks) {
#pragma unroll 1
  for (int i = 0; i < nchunks; ++i) {
    st.advance();
    st.issue();
    st.sync();
  }
}
// FFN-shaped run over F/32 groups of 32 hidden units: W1C chunks = the group's W1 pairs (pair q: k-step q >> 1, hidden tile
// q & 1) -> hidden units in registers; CPK chunks = the group's W2 pairs -> consume(c, output tile, A_hi, A_lo) with the hidden
// fragment pair made by `hidden(c, h0, h1)`.  Fragment prefetch runs across all chunk boundaries of the run.
template <class St, class Hidden, class Consume>
__device__ __forceinline__ void ffn_run(St &st, FragRing &r, int ngroups, const f16x8 (&xh)[NKS], const f16x8 (&xl)[NKS],
                                        Hidden hidden, Consume consume) {
  prime(r, st.cur());
#pragma unroll 1
  for (int c = 0; c < ngroups; ++c) {
    f32x4 h0 = {0.f, 0.f, 0.f, 0.f}, h1 = h0;
#pragma unroll
    for (int j = 0; j < W1C; ++j) {
      const f16x8 *cur = st.cur(), *nx = st.nxt();
      st.advance();
      chunk_pipe<true>(st, r, cur, nx, [&](int p, const f16x8 &ah, const f16x8 &al) {
        const int q = 16 * j + p;
        if (q & 1) mfma3(h1, ah, al, xh[q >> 1], xl[q >> 1]);
        else mfma3(h0, ah, al, xh[q >> 1], xl[q >> 1]);
      });
    }
    hidden(c, h0, h1);
#pragma unroll
    for (int j = 0; j < CPK; ++j) {
      const f16x8 *cur = st.cur(), *nx = st.nxt();
      st.advance();
      chunk_pipe<true>(st, r, cur, nx, [&](int p, const f16x8 &ah, const f16x8 &al) { consume(c, 16 * j + p, ah, al); });
    }
  }
}

__device__ __forceinline__ void load_tile(const u32x4 *X, long tile, int lane_idx, f16x8 (&xh)[NKS], f16x8 (&xl)[NKS]) {
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    xh[ks] = __builtin_bit_cast(f16x8, X[xpiece(tile, ks, 0, lane_idx)]);
    xl[ks] = __builtin_bit_cast(f16x8, X[xpiece(tile, ks, 1, lane_idx)]);
  }
}

// The fp32 parameter vectors of a kernel sit in LDS behind the 128 KB chunk ring and are read as f32x4 at (word offset + 4 g).  Their
// byte offsets are beyond the 16-bit immediate of ds_read, so, left alone, hipcc materialises one VGPR address per vector read, hoists
// them all out of the tile loop and spills them (36 registers at d = 512, 11 at d = 256; every reload is a scratch load whose wait is
// vmcnt(0), i.e. it also drains the K / V and residual loads in flight).  LaneParams is ONE per-lane base the optimiser cannot see
// through: every read is base + immediate.
typedef __attribute__((address_space(3))) const f32x4 lds_cf32x4;
struct LaneParams {
  unsigned base;          // LDS byte address of prm + 4 g
  __device__ __forceinline__ f32x4 operator()(int word) const { return *(lds_cf32x4 *)(uintptr_t)(base + 4u * (unsigned)word); }
};
__device__ __forceinline__ LaneParams lane_params(const float *prm, int g) {
  unsigned b = lds_addr(prm) + 16u * (unsigned)g;
  asm volatile("" : "+v"(b));
  return LaneParams{b};
}

// v = LayerNorm(v) over the D features of each token (NMT tiles x 4 registers x 4 lane groups), fp32, two passes
// returns the reciprocal standard deviation (NaN iff an input was NaN / inf: the f16 range guard accumulates it)
__device__ __forceinline__ float layer_norm(f32x4 (&v)[NMT], const LaneParams &P, int lw, int lb) {   // lw / lb: word offsets of weight / bias
  float s = 0.f;
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) s += (v[mt][0] + v[mt][1]) + (v[mt][2] + v[mt][3]);
  const float mean = group_sum4(s) * (1.f / D);
  float q = 0.f;
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[mt][r] -= mean; q = fmaf(v[mt][r], v[mt][r], q); }
  const float rstd = __builtin_amdgcn_rsqf(group_sum4(q) * (1.f / D) + 1e-5f);     // (v_rsq_f32: 1 ulp)
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) {
    const f32x4 wv = P(lw + 16 * mt), bv = P(lb + 16 * mt);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[mt][r] = fmaf(v[mt][r] * rstd, wv[r], bv[r]);
  }
  return rstd;
}

// ---- K / V of the key rows ----------------------------------------------------------------------------------------
struct KvArgs {
  Geo g; int tpe, nkt2, ngroups;
  const u32x4 *X;                 // KEY IMAGE of this layer: [B][WNK / 16] tiles of the key rows in key-list order (keys_kernel / the previous layer)
  const unsigned *img;            // this layer's weight image
  int F;
  const int *keyrow, *kcnt;
  u32x4 *KV;
};

// A workgroup computes K and V of its group of WAVES key tiles (all 2 PC chunks of Wk | Wv per group).
__global__ __launch_bounds__(THREADS) void kv_all_kernel(KvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float *prm = reinterpret_cast<float *>(lds + NBUF * CHUNK_BYTES);      // bk | bv
  const int tid = threadIdx.x, lane = tid & 63, tok = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float *gprm = reinterpret_cast<const float *>(a.img + (long)layer_chunks(a.F) * CHUNK_WORDS);
  for (int i = tid; i < 2 * D; i += THREADS) prm[i] = gprm[D + i];
  const char *wbase = reinterpret_cast<const char *>(a.img) + (long)PC * CHUNK_BYTES;     // K chunks, then V chunks
  auto st = make_stream([wbase](int s) { return wbase + (long)s * CHUNK_BYTES; }, lds, 2 * PC, tid);
  st.start();
  __syncthreads();
  st.sync();
  FragRing ring;
  const long ntiles = (long)a.g.B * a.nkt2;
  for (int grp = blockIdx.x; grp < a.ngroups; grp += gridDim.x) {
    const long tile = (long)grp * WAVES + wave;
    const bool valid = tile < ntiles;
    const long tl = valid ? tile : ntiles - 1;
    const int b = tl / a.nkt2, kt = tl % a.nkt2, key = 16 * kt + tok;
    const bool live = key < a.kcnt[2 * b + 1];          // (slots behind the key count hold stale rows: zeroed, the attention masks them)
    f16x8 xh[NKS], xl[NKS];
    load_tile(a.X, (long)b * (WNK / 16) + kt, lane, xh, xl);
    if (!live) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) { xh[ks] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0}; xl[ks] = xh[ks]; }
    }
    u32x4 *kv = a.KV + (long)b * KV_EP;
    f32x4 y[NMT];
    // K^T = Wk KX^T: rows = channels, columns = keys -> A fragments of S^T = K Q^T, pair (channel k-step h, kt) -- a head is KPH k-steps
#pragma unroll
    for (int m = 0; m < NMT; ++m) y[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    chunk_run<PC>(st, ring, [&](int cc, int p, const f16x8 &ah, const f16x8 &al) { mfma3(y[16 * (cc % CPK) + p], ah, al, xh[cc / CPK], xl[cc / CPK]); });
#pragma unroll
    for (int h = 0; h < NKS; ++h) {
      const f32x4 b0 = *reinterpret_cast<const f32x4 *>(prm + 32 * h + 4 * g), b1 = *reinterpret_cast<const f32x4 *>(prm + 32 * h + 16 + 4 * g);
      f16x8 fh, fl;
      split_frag(y[2 * h] * WINV + b0, y[2 * h + 1] * WINV + b1, fh, fl);
      if (valid) {
        kv[((h * 4 + kt) * 2) * 64 + lane] = __builtin_bit_cast(u32x4, fh);
        kv[((h * 4 + kt) * 2 + 1) * 64 + lane] = __builtin_bit_cast(u32x4, fl);
      }
    }
    // V = KX Wv^T with the MFMA operands swapped (rows = keys, columns = channels): accumulator tile i holds
    // V[key 16 kt + 4 g + r][channel 16 i + tok] -- half (kt & 1) of this lane's piece of the V^T pair (i, kt / 2)
#pragma unroll
    for (int m = 0; m < NMT; ++m) y[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    chunk_run<PC>(st, ring, [&](int cc, int p, const f16x8 &bh, const f16x8 &bl) { mfma3(y[16 * (cc % CPK) + p], xh[cc / CPK], xl[cc / CPK], bh, bl); });
#pragma unroll
    for (int i = 0; i < NMT; ++i) {
      const float bv = prm[D + 16 * i + tok];
      unsigned h0, l0, h1, l1;
      split2(y[i][0] * WINV + bv, y[i][1] * WINV + bv, h0, l0);
      split2(y[i][2] * WINV + bv, y[i][3] * WINV + bv, h1, l1);
      if (valid) {
        u32x2 *ph = reinterpret_cast<u32x2 *>(kv + KV_VOFF + ((i * 2 + (kt >> 1)) * 2) * 64 + lane) + (kt & 1);
        u32x2 *pl = reinterpret_cast<u32x2 *>(kv + KV_VOFF + ((i * 2 + (kt >> 1)) * 2 + 1) * 64 + lane) + (kt & 1);
        *ph = (u32x2){h0, h1};
        *pl = (u32x2){l0, l1};
      }
    }
  }
  st.finish();
}

// The same split into jobs -- (K | V) x (CPK halves of the D output channels): 16 output tiles and NKS chunks each --, one job per
// workgroup (blockIdx % (2 CPK)), for launches with fewer groups of key tiles than CUs: at d = 512 / B = 256 there are 128 groups, each
// streaming 2 MB of weights through one CU's LDS-DMA (60 us per launch, 14 % of the cfg5 rollout); split four ways every CU streams.
__global__ __launch_bounds__(THREADS) void kv_split_kernel(KvArgs a) {
  constexpr bool SPLIT = true;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float *prm = reinterpret_cast<float *>(lds + NBUF * CHUNK_BYTES);      // bk | bv
  const int tid = threadIdx.x, lane = tid & 63, tok = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float *gprm = reinterpret_cast<const float *>(a.img + (long)layer_chunks(a.F) * CHUNK_WORDS);
  for (int i = tid; i < 2 * D; i += THREADS) prm[i] = gprm[D + i];
  constexpr int NJ = 2 * CPK;
  const int job0 = SPLIT ? (int)(blockIdx.x % NJ) : 0;
  const int grp0 = SPLIT ? (int)(blockIdx.x / NJ) : (int)blockIdx.x, gstride = SPLIT ? (int)(gridDim.x / NJ) : (int)gridDim.x;
  const char *wbase = reinterpret_cast<const char *>(a.img) + (long)PC * CHUNK_BYTES;     // K chunks, then V chunks
  // stream position s -> job (SPLIT: job0; else s / NKS), k-step s % NKS -> chunk (K | V) PC + CPK k + half
  auto st = make_stream([wbase, job0](int s) {
    const int job = SPLIT ? job0 : s / NKS, k = s % NKS;
    return wbase + (long)((job / CPK) * PC + CPK * k + job % CPK) * CHUNK_BYTES;
  }, lds, (SPLIT ? 1 : NJ) * NKS, tid);
  st.start();
  __syncthreads();
  st.sync();
  FragRing ring;
  const long ntiles = (long)a.g.B * a.nkt2;
  for (int grp = grp0; grp < a.ngroups; grp += gstride) {
    const long tile = (long)grp * WAVES + wave;
    const bool valid = tile < ntiles;
    const long tl = valid ? tile : ntiles - 1;
    const int b = tl / a.nkt2, kt = tl % a.nkt2, key = 16 * kt + tok;
    const bool live = key < a.kcnt[2 * b + 1];          // (slots behind the key count hold stale rows: zeroed, the attention masks them)
    f16x8 xh[NKS], xl[NKS];
    load_tile(a.X, (long)b * (WNK / 16) + kt, lane, xh, xl);
    if (!live) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) { xh[ks] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0}; xl[ks] = xh[ks]; }
    }
    u32x4 *kv = a.KV + (long)b * KV_EP;
#pragma unroll
    for (int job = 0; job < NJ; ++job) {      // (unrolled: the job, and with it every register index and parameter offset, is a constant)
      if (SPLIT && job != job0) continue;
      const int half = job % CPK;
      f32x4 y[16];
#pragma unroll
      for (int m = 0; m < 16; ++m) y[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (job < CPK) {
        // K^T = Wk KX^T: rows = channels, columns = keys -> A fragments of S^T = K Q^T, pair (channel k-step f, kt) -- a head is KPH k-steps
        chunk_run<NKS>(st, ring, [&](int cc, int p, const f16x8 &ah, const f16x8 &al) { mfma3(y[p], ah, al, xh[cc], xl[cc]); });
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int f = 8 * half + j;
          const f32x4 b0 = *reinterpret_cast<const f32x4 *>(prm + 32 * f + 4 * g), b1 = *reinterpret_cast<const f32x4 *>(prm + 32 * f + 16 + 4 * g);
          f16x8 fh, fl;
          split_frag(y[2 * j] * WINV + b0, y[2 * j + 1] * WINV + b1, fh, fl);
          if (valid) {
            kv[((f * 4 + kt) * 2) * 64 + lane] = __builtin_bit_cast(u32x4, fh);
            kv[((f * 4 + kt) * 2 + 1) * 64 + lane] = __builtin_bit_cast(u32x4, fl);
          }
        }
      } else {
        // V = KX Wv^T with the MFMA operands swapped (rows = keys, columns = channels): accumulator tile i holds
        // V[key 16 kt + 4 g + r][channel 16 i + tok] -- half (kt & 1) of this lane's piece of the V^T pair (i, kt / 2)
        chunk_run<NKS>(st, ring, [&](int cc, int p, const f16x8 &bh, const f16x8 &bl) { mfma3(y[p], xh[cc], xl[cc], bh, bl); });
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int i = 16 * half + j;
          const float bv = prm[D + 16 * i + tok];
          unsigned h0, l0, h1, l1;
          split2(y[j][0] * WINV + bv, y[j][1] * WINV + bv, h0, l0);
          split2(y[j][2] * WINV + bv, y[j][3] * WINV + bv, h1, l1);
          if (valid) {
            u32x2 *ph = reinterpret_cast<u32x2 *>(kv + KV_VOFF + ((i * 2 + (kt >> 1)) * 2) * 64 + lane) + (kt & 1);
            u32x2 *pl = reinterpret_cast<u32x2 *>(kv + KV_VOFF + ((i * 2 + (kt >> 1)) * 2 + 1) * 64 + lane) + (kt & 1);
            *ph = (u32x2){h0, h1};
            *pl = (u32x2){l0, l1};
          }
        }
      }
    }
  }
  st.finish();
}

// ---- one encoder layer of a token tile ----------------------------------------------------------------------------
struct LayerArgs {
  Geo g; int tpe, ngroups;
  const u32x4 *XIN; u32x4 *XOUT;
  const unsigned *img; int F;
  const u32x4 *KV; const int *kcnt;
  u32x4 *zimg; long zrow0;        // last layer: the rows of the target tokens also go to this (dense-row) image
  unsigned *range_flag;           // f16 range guard (common.h)
  const short *keypos; u32x4 *KXout;   // not the last layer: output rows that are keys also go to the next layer's key image
  // layer_save_kernel (the forward recompute of the per-op backward, round 4): fp32 rows [instance * N + token row][..] of what that
  // backward reads -- the attention output A, U1 = X + Wo A + bo, X1 = LN1(U1), the hidden units relu(W1 X1 + b1) [.., F], U2 = X1 + W2 h + b2
  // and the layer output Y = LN2(U2)
  float *svA, *svU1, *svX1, *svHid, *svU2, *svY;
  float *svQ; long svQ_ld;      // the (unscaled) Q rows, into the [.., 3 d] buffer whose K | V slices a row-gather GEMM fills for the key rows
#ifdef X3_STAMPS
  unsigned long long *stamps;     // [8 waves][X3_NSTAMP] of workgroup 0
#endif
};

// masked set-attention of one token tile against NKT key tiles, head by head (model/encoder.py:8-46): S^T = K Q^T in
// the exp2 domain (scale folded into Wq), softmax over the keys of a token (4 NKT values per lane x 4 lane groups),
// O^T = V^T P.  qh / ql [h] go in as the Q^T fragment pair of head h and come out as the pair of the normalised head
// output (= k-step h of the out-projection).
// The K / V fragment pairs of one head come from L2 (KV buffer of the episode, 2 NKT + 4 NS pieces of 16 bytes per lane): the
// pairs of head h + 1 are requested before head h is computed (a dependent L2 round trip per head was 11 % of the kernel:
// profiles/r02_x3_layer_kernel_stamps.txt), those of head 0 by the caller before the Q projection's epilogue.
template <int NKT>
struct HeadKV {
  static constexpr int NS = NKT > 2 ? 2 : 1;
  u32x4 k[2 * NKT * KPH], v[2 * MPH * NS];     // K pairs ((j, kt): hi, lo), channel k-step KPH h + j; V^T pairs ((ii, si): hi, lo), channel tile MPH h + ii
  __device__ __forceinline__ void load(const u32x4 *kv, int h) {
#pragma unroll
    for (int j = 0; j < KPH; ++j)
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        k[2 * (j * NKT + kt)] = kv[(((h * KPH + j) * 4 + kt) * 2) * 64];
        k[2 * (j * NKT + kt) + 1] = kv[(((h * KPH + j) * 4 + kt) * 2 + 1) * 64];
      }
#pragma unroll
    for (int ii = 0; ii < MPH; ++ii)
#pragma unroll
      for (int si = 0; si < NS; ++si) {
        const u32x4 *vp = kv + KV_VOFF + (((MPH * h + ii) * 2 + si) * 2) * 64;
        v[(ii * NS + si) * 2] = vp[0]; v[(ii * NS + si) * 2 + 1] = vp[64];
      }
  }
};
template <int NKT>
__device__ __forceinline__ void attention_tile(f16x8 (&qh)[NKS], f16x8 (&ql)[NKS], const u32x4 *kv, int nv, HeadKV<NKT> &first) {
  constexpr int NS = NKT > 2 ? 2 : 1;
  f32x4 mb[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) mb[kt][r] = (16 * kt + r) < nv ? 0.f : -INFINITY;
  // K / V pairs requested AHEAD heads in front of the one being computed.  d = 256: one (two waves per SIMD hide the rest; two ahead
  // measured 1 % slower).  d = 512 (one wave per SIMD): a head is ~0.3 kcycles of work behind a round trip of ~5 kcycles, and with one
  // head ahead the waits were 16 % of the kernel (in-kernel stamps) -- as many heads as the registers the input tile has just freed can
  // hold (a head's pairs are 64 registers up to 32 keys, 128 beyond) are in flight at once; the scheduling barriers keep the requests
  // where they are written (left alone the scheduler sinks every load towards its use).
  constexpr int AHEAD = X3_KV_AHEAD > 1 && NKT > 2 ? 1 : X3_KV_AHEAD, NB = AHEAD + 1;
  const float inf = opaque_inf();
  HeadKV<NKT> buf[NB];
  buf[0] = first;
#pragma unroll
  for (int h = 1; h < AHEAD; ++h) buf[h].load(kv, h);
  if (AHEAD > 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int h = 0; h < H; ++h) {
    HeadKV<NKT> &c = buf[h % NB];
    if (h + AHEAD < H) buf[(h + AHEAD) % NB].load(kv, h + AHEAD);
    if (AHEAD > 1) __builtin_amdgcn_sched_barrier(0);
    f32x4 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      s[kt] = mb[kt];
#pragma unroll
      for (int j = 0; j < KPH; ++j)
        mfma3(s[kt], __builtin_bit_cast(f16x8, c.k[2 * (j * NKT + kt)]), __builtin_bit_cast(f16x8, c.k[2 * (j * NKT + kt) + 1]), qh[h * KPH + j], ql[h * KPH + j]);
    }
    float mx = vmax(vmax(s[0][0], s[0][1], inf), vmax(s[0][2], s[0][3], inf), inf);
#pragma unroll
    for (int kt = 1; kt < NKT; ++kt) mx = vmax(mx, vmax(vmax(s[kt][0], s[kt][1], inf), vmax(s[kt][2], s[kt][3], inf), inf), inf);
    mx = group_max4(mx, inf);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[kt][r] = __builtin_amdgcn_exp2f(s[kt][r] - mx);
        sum += s[kt][r];
      }
    const float inv = __builtin_amdgcn_rcpf(group_sum4(sum));                          // (v_rcp_f32: 1 ulp)
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 o[MPH];
#pragma unroll
    for (int ii = 0; ii < MPH; ++ii) o[ii] = z4;
#pragma unroll
    for (int si = 0; si < NS; ++si) {
      f16x8 ph, pl;
      split_frag(s[2 * si], (2 * si + 1 < NKT) ? s[(2 * si + 1 < NKT) ? 2 * si + 1 : 0] : z4, ph, pl);
#pragma unroll
      for (int ii = 0; ii < MPH; ++ii)
        mfma3(o[ii], __builtin_bit_cast(f16x8, c.v[(ii * NS + si) * 2]), __builtin_bit_cast(f16x8, c.v[(ii * NS + si) * 2 + 1]), ph, pl);
    }
#pragma unroll
    for (int j = 0; j < KPH; ++j) split_frag(o[2 * j] * inv, o[2 * j + 1] * inv, qh[h * KPH + j], ql[h * KPH + j]);
  }
}

// fp32 side outputs of layer_save_kernel: written once, read by other kernels much later -- streaming stores
__device__ __forceinline__ void sv_store(float *p, const f32x4 &v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p)); }

template <bool LAST, bool SAVE>
__device__ __forceinline__ void layer_body(const LayerArgs &a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float *prm = reinterpret_cast<float *>(lds + NBUF * CHUNK_BYTES);   // bq bk bv | bo | b1 | b2 | ln1w ln1b ln2w ln2b
  const Geo &G = a.g;
  const int tid = threadIdx.x, lane = tid & 63, tok = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int F = a.F, np = layer_params(F), n_t = G.n_td + G.n_th;
  {
    const float *gprm = reinterpret_cast<const float *>(a.img + (long)layer_chunks(F) * CHUNK_WORDS);
    for (int i = tid; i < np; i += THREADS) prm[i] = gprm[i];
  }
  const LaneParams P = lane_params(prm, g);
  const int o_bo = 3 * D, o_b1 = 4 * D, o_b2 = o_b1 + F, o_ln1w = o_b2 + D, o_ln1b = o_ln1w + D, o_ln2w = o_ln1b + D, o_ln2b = o_ln2w + D;   // word offsets
  // stream order of one tile group: Q (chunks 0 .. PC - 1), OUT (3 PC .. 4 PC - 1), FFN (4 PC ..)
  const char *wbase = reinterpret_cast<const char *>(a.img);
  const int seq = 2 * PC + (F / 32) * (W1C + CPK);
  auto st = make_stream([wbase](int s) { return wbase + (long)(s < PC ? s : s + 2 * PC) * CHUNK_BYTES; }, lds, seq, tid);
  st.start();
  __syncthreads();
#ifdef X3_STAMPS
  unsigned long long t_begin;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_begin)::"memory");
  st.stamps.start();
#endif
  st.sync();
  FragRing ring;
  float range_chk = 0.f;
  // Tile rounds.  Full rounds: workgroup b takes the 8 consecutive tiles of group rd * G + b, one per wave (two waves per
  // SIMD).  What is left after the full rounds (712 of 13 000 tiles at the headline shape: 0.35 of a round) is spread over ALL
  // workgroups, `ktail` waves each -- waves 0 .. ktail - 1 sit on different SIMDs while ktail <= 4, so a tail tile has the
  // matrix pipe of its SIMD to itself instead of sharing it in a few fully occupied workgroups while the other CUs idle.
  const long ntiles = (long)G.B * a.tpe;
  const int NG = gridDim.x;
  const long per_round = (long)NG * WAVES;
  const int full = (int)(ntiles / per_round);
  const long rem = ntiles - (long)full * per_round;
  const int ktail = (int)((rem + NG - 1) / NG);
  const bool wg_tail = rem > 0 && (long)blockIdx.x * ktail < rem;              // (workgroup-uniform) a tail round for this workgroup
  const long tail_tile = (long)full * per_round + (long)blockIdx.x * ktail + wave;
  const bool my_tail = wg_tail && wave < ktail && tail_tile < ntiles;
  const int my_rounds = full + (my_tail ? 1 : 0);                              // rounds in which this wave computes a tile
  auto tile_of = [&](int rd) -> long {                                         // (rd < my_rounds)
    return rd < full ? ((long)rd * NG + blockIdx.x) * WAVES + wave : tail_tile;
  };
  auto rows_of = [&](long tile, int &b, int &r, int &lidx) {
    b = (int)(tile / a.tpe);
    r = 16 * (int)(tile % a.tpe) + tok;
    lidx = 16 * g + (min(r, G.N - 1) & 15);
  };
  f16x8 xh[NKS], xl[NKS];
  if (my_rounds > 0) { int b0, r0, l0; rows_of(tile_of(0), b0, r0, l0); load_tile(a.XIN, tile_of(0), l0, xh, xl); }
  for (int rd = 0; rd < my_rounds; ++rd) {
    const long tl = tile_of(rd);
    int b, r, lidx;
    rows_of(tl, b, r, lidx);
    const int rc = min(r, G.N - 1);
    const bool rowok = r < G.N;
    const int n_ck = a.kcnt[2 * b], n_ak = a.kcnt[2 * b + 1];
    const bool isq = rc < G.P && !is_ctx(G, b, rc);
    const u32x4 *kv = a.KV + (long)b * KV_EP + lane;
    const int nkt = __builtin_amdgcn_readfirstlane((n_ak + 15) >> 4);     // (uniform per wave: one episode per tile)
    X3_LAP(st, 7);

    f32x4 y[NMT];
    // ---- Q^T = Wq X^T (pre-scaled by log2(e) / sqrt(hd)) ----------------------------------------------------------
#pragma unroll
    for (int m = 0; m < NMT; ++m) y[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    chunk_run<PC>(st, ring, [&](int cc, int p, const f16x8 &ah, const f16x8 &al) { mfma3(y[16 * (cc % CPK) + p], ah, al, xh[cc / CPK], xl[cc / CPK]); });
    // ---- attention: context and target rows see the context keys, query rows also the visible targets -------------
    f16x8 qh[NKS], ql[NKS];          // Q^T fragment pairs by channel k-step (head h: k-steps KPH h ..), then the attention output's
    auto q_epilogue = [&]() {
#pragma unroll
      for (int h = 0; h < NKS; ++h) {
        const f32x4 c0 = P(32 * h), c1 = P(32 * h + 16);
        if (SAVE && rowok) {
          constexpr float QINV = (HD == 64 ? 8.f : 5.65685424949238f) * 0.69314718055994531f;      // sqrt(HD) / log2(e): undoes pack_kernel's qscale
          float *qp = a.svQ + ((long)b * G.N + rc) * a.svQ_ld + 32 * h + 4 * g;
          sv_store(qp, (y[2 * h] * WINV + c0) * QINV);
          sv_store(qp + 16, (y[2 * h + 1] * WINV + c1) * QINV);
        }
        split_frag(y[2 * h] * WINV + c0, y[2 * h + 1] * WINV + c1, qh[h], ql[h]);
      }
      if (X3_KV_AHEAD > 1) __builtin_amdgcn_sched_barrier(0);      // (the Q accumulators are dead before more K / V pairs are requested)
      X3_LAP(st, 5);
    };
    {
      const int nv = (isq ? n_ak : n_ck) - 4 * g;          // key 16 kt + 4 g + r is visible iff 16 kt + r < nv
      // (head 0's K / V pairs are requested before the epilogue of the Q projection and land while it runs)
      if (nkt <= 1) { HeadKV<1> k0; k0.load(kv, 0); q_epilogue(); attention_tile<1>(qh, ql, kv, nv, k0); }
      else if (nkt == 2) { HeadKV<2> k0; k0.load(kv, 0); q_epilogue(); attention_tile<2>(qh, ql, kv, nv, k0); }
      else { HeadKV<4> k0; k0.load(kv, 0); q_epilogue(); attention_tile<4>(qh, ql, kv, nv, k0); }
      X3_LAP(st, 6);
    }
    const long srow = (long)b * G.N + rc;          // (SAVE) fp32 row of this lane's token
    if (SAVE && rowok) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        sv_store(a.svA + srow * D + 32 * ks + 4 * g, frag_value(qh[ks], ql[ks], 0));
        sv_store(a.svA + srow * D + 32 * ks + 16 + 4 * g, frag_value(qh[ks], ql[ks], 1));
      }
    }
    auto save_rows = [&](float *dst, const f32x4 (&v)[NMT]) {
      if (SAVE && rowok) {
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) sv_store(dst + srow * D + 16 * mt + 4 * g, v[mt]);
      }
    };
    // ---- X1 = LN1(X + bo + Wo A): the residual X comes back from L2 a k-step per chunk, into the registers the consumed
    // fragments of the attention output free (it was not kept through the attention: registers) -------------------------
#pragma unroll
    for (int m = 0; m < NMT; ++m) y[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    chunk_run_hook<PC>(st, ring, [&](int cc, int p, const f16x8 &ah, const f16x8 &al) { mfma3(y[16 * (cc % CPK) + p], ah, al, qh[cc / CPK], ql[cc / CPK]); },
                       [&](int cc) {
                         if (cc % CPK == CPK - 1) {        // k-step cc / CPK of the attention output is consumed
                           xh[cc / CPK] = __builtin_bit_cast(f16x8, a.XIN[xpiece(tl, cc / CPK, 0, lidx)]);
                           xl[cc / CPK] = __builtin_bit_cast(f16x8, a.XIN[xpiece(tl, cc / CPK, 1, lidx)]);
                         }
                       });
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
      y[mt] = y[mt] * WINV + P(o_bo + 16 * mt) + frag_value(xh[mt >> 1], xl[mt >> 1], mt & 1);
    save_rows(a.svU1, y);
    range_chk += layer_norm(y, P, o_ln1w, o_ln1b);
    save_rows(a.svX1, y);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) split_frag(y[2 * ks], y[2 * ks + 1], xh[ks], xl[ks]);
    X3_LAP(st, 7);
    // ---- X = LN2(X1 + b2 + W2 relu(W1 X1 + b1)): 32 hidden units per chunk pair ---------------------------------------
#pragma unroll
    for (int m = 0; m < NMT; ++m) y[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      f16x8 hbh, hbl;
      ffn_run(st, ring, F / 32, xh, xl,
              [&](int c, f32x4 &h0, f32x4 &h1) {
                const f32x4 c0 = P(o_b1 + 32 * c), c1 = P(o_b1 + 32 * c + 16);
                h0 = h0 * WINV + c0; h1 = h1 * WINV + c1;
#pragma unroll
                for (int r = 0; r < 4; ++r) { h0[r] = relu_nn(h0[r]); h1[r] = relu_nn(h1[r]); }
                if (SAVE && rowok) {
                  sv_store(a.svHid + srow * F + 32 * c + 4 * g, h0);
                  sv_store(a.svHid + srow * F + 32 * c + 16 + 4 * g, h1);
                }
                split_frag(h0, h1, hbh, hbl);
                X3_LAP(st, 5);
              },
              [&](int, int m, const f16x8 &ah, const f16x8 &al) { mfma3(y[m], ah, al, hbh, hbl); });
    }
    // (The next round's tile is loaded at the top of the round: the Q projection consumes a k-step per chunk, so only the first
    //  k-step's latency is exposed.  Requesting the whole tile here, behind LN2 and the stores, was measured 2 % SLOWER -- 64 more
    //  live registers through LN2 --, requesting its first k-step only: no change.  profiles/r03_x3_timing_experiments.txt)
    const long tn = tile_of(rd + 1 < my_rounds ? rd + 1 : rd);
    int bn, rn, ln;
    rows_of(tn, bn, rn, ln);
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
      y[mt] = y[mt] * WINV + P(o_b2 + 16 * mt) + frag_value(xh[mt >> 1], xl[mt >> 1], mt & 1);
    save_rows(a.svU2, y);
    range_chk += layer_norm(y, P, o_ln2w, o_ln2b);
    save_rows(a.svY, y);
    const bool ztgt = LAST && a.zimg && rowok && r >= G.P;
    const long zr = a.zrow0 + (long)b * n_t + (r - G.P);
    const int kp = (!LAST && rowok) ? a.keypos[(long)b * 16 * a.tpe + r] : -1;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      f16x8 oh, ol;
      split_frag(y[2 * ks], y[2 * ks + 1], oh, ol);
      if (rowok) {
        a.XOUT[xpiece(tl, ks, 0, lidx)] = __builtin_bit_cast(u32x4, oh);
        a.XOUT[xpiece(tl, ks, 1, lidx)] = __builtin_bit_cast(u32x4, ol);
      }
      if (ztgt) {
        a.zimg[xpiece(zr >> 4, ks, 0, 16 * g + (int)(zr & 15))] = __builtin_bit_cast(u32x4, oh);
        a.zimg[xpiece(zr >> 4, ks, 1, 16 * g + (int)(zr & 15))] = __builtin_bit_cast(u32x4, ol);
      }
      if (!LAST && kp >= 0) {
        a.KXout[xpiece((long)b * (WNK / 16) + (kp >> 4), ks, 0, 16 * g + (kp & 15))] = __builtin_bit_cast(u32x4, oh);
        a.KXout[xpiece((long)b * (WNK / 16) + (kp >> 4), ks, 1, 16 * g + (kp & 15))] = __builtin_bit_cast(u32x4, ol);
      }
    }
    load_tile(a.XIN, tn, ln, xh, xl);       // (after the last round: the same tile once more, from L2, dropped)
    X3_LAP(st, 7);
  }
  if (wg_tail && !my_tail) idle_chunks(st, seq);       // the tail round of a wave without a tile: stream + barriers only
  st.finish();
  range_check_nan(a.range_flag, range_chk);
#ifdef X3_STAMPS
  if (a.stamps && blockIdx.x == 0 && lane == 0) {
    unsigned long long t_end;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end)::"memory");
    st.stamps.acc[8] = t_end - t_begin;
    for (int k = 0; k < X3_NSTAMP; ++k) a.stamps[wave * X3_NSTAMP + k] = st.stamps.acc[k];
  }
#endif
}
template <bool LAST>
__global__ __launch_bounds__(THREADS) void layer_kernel(LayerArgs a) { layer_body<LAST, false>(a); }
// the same layer with the fp32 side outputs the per-op backward reads (LayerArgs.sv*): its forward recompute at the speed of the
// rollout's own kernel instead of four generic GEMMs, two LayerNorm kernels and the attention kernel per layer
__global__ __launch_bounds__(THREADS) void layer_save_kernel(LayerArgs a) { layer_body<false, true>(a); }

// ---- acquisition head (NOUT = 1, model/head.py:27-33) / one GMM head (NOUT = 3, model/head.py:152-186) --------------
// out[row * out_stride + out_off + j] = w2[j] . relu(W1 z + b1) + b2[j] over the 16-row tiles of an image
struct HeadArgs {
  const u32x4 *X; long ntiles, M;
  const unsigned *img; int F, ngroups;
  float *out; int out_stride, out_off;
  float tau;      // acquisition head with a time token: this step's t / T (b1 + tau W1[:, d] is the hidden layer's bias), else 0
};
template <int NOUT>
__global__ __launch_bounds__(THREADS) void head_kernel(HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float *prm = reinterpret_cast<float *>(lds + NBUF * CHUNK_BYTES);      // b1 [F] | w2 [3][F]  (b2 [4] stays in the image: at F = 2048 the ring + 4 F floats are the 160 KB)
  const int tid = threadIdx.x, lane = tid & 63, tok = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int F = a.F;
  const float *gprm = reinterpret_cast<const float *>(a.img + (long)head_chunks(F) * CHUNK_WORDS);
  {
    for (int i = tid; i < head_lds_params(F); i += THREADS) prm[i] = (a.tau != 0.f && i < F) ? fmaf(a.tau, gprm[2 * F + i], gprm[i]) : gprm[i];
  }
  const char *wbase = reinterpret_cast<const char *>(a.img);
  auto st = make_stream([wbase](int s) { return wbase + (long)s * CHUNK_BYTES; }, lds, head_chunks(F), tid);
  st.start();
  __syncthreads();
  st.sync();
  FragRing ring;
  for (int grp = blockIdx.x; grp < a.ngroups; grp += gridDim.x) {
    const long tile = (long)grp * WAVES + wave;
    const bool valid = tile < a.ntiles;
    const long tl = valid ? tile : a.ntiles - 1;
    const long row = 16 * tl + tok;
    f16x8 xh[NKS], xl[NKS];
    load_tile(a.X, tl, lane, xh, xl);
    float plog[NOUT];
#pragma unroll
    for (int jo = 0; jo < NOUT; ++jo) plog[jo] = 0.f;
    prime(ring, st.cur());
#pragma unroll 1
    for (int c = 0; c < F / 32; ++c) {
      f32x4 h0 = {0.f, 0.f, 0.f, 0.f}, h1 = h0;
#pragma unroll
      for (int j = 0; j < W1C; ++j) {
        const f16x8 *cur = st.cur(), *nx = st.nxt();
        st.advance();
        auto w1 = [&](int p, const f16x8 &ah, const f16x8 &al) {
          const int q = 16 * j + p;
          if (q & 1) mfma3(h1, ah, al, xh[q >> 1], xl[q >> 1]);
          else mfma3(h0, ah, al, xh[q >> 1], xl[q >> 1]);
        };
        chunk_pipe<true>(st, ring, cur, nx, w1);
      }
      const f32x4 c0 = *reinterpret_cast<const f32x4 *>(prm + 32 * c + 4 * g), c1 = *reinterpret_cast<const f32x4 *>(prm + 32 * c + 16 + 4 * g);
      h0 = h0 * WINV + c0; h1 = h1 * WINV + c1;
#pragma unroll
      for (int jo = 0; jo < NOUT; ++jo) {
        const f32x4 w0 = *reinterpret_cast<const f32x4 *>(prm + (1 + jo) * F + 32 * c + 4 * g);
        const f32x4 w1 = *reinterpret_cast<const f32x4 *>(prm + (1 + jo) * F + 32 * c + 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          plog[jo] = fmaf(relu_nn(h0[r]), w0[r], plog[jo]);
          plog[jo] = fmaf(relu_nn(h1[r]), w1[r], plog[jo]);
        }
      }
    }
#pragma unroll
    for (int jo = 0; jo < NOUT; ++jo) {
      const float v = group_sum4(plog[jo]) + gprm[4 * F + jo];
      if (g == 0 && valid && row < a.M) a.out[row * a.out_stride + a.out_off + jo] = v;
    }
  }
  st.finish();
}

}  // namespace X3_NS
