// s3 path: d = 32 (4 heads of 8) at REFERENCE precision, any embedding mode (theta / data / mix), up to 160 keys.
//
// One launch per design step: a workgroup owns EPW = 2 whole episodes and runs every encoder layer of the step for
// them -- only workgroup barriers separate the layers, the K / V of an episode never leave LDS, the activations make
// one round trip per layer through an L2-resident work image.  Arithmetic as x3.h: every product a 3-term f16 split on
// v_mfma_f32_16x16x32_f16 with fp32 accumulation; weights pre-scaled by 2^8 at pack time.
//
// Per layer and workgroup:
//   weights   the layer's 8 + F/8 fragment pairs + fp32 parameters (50 KB at F = 128), LDS-DMA'd into LDS
//   K / V     of the key rows (context rows, then the visible targets; model/encoder.py:83-126): key tiles spread
//             over the 8 waves, results written to LDS as the A fragments the attention consumes
//   tokens    2 x ceil(N / 16) token tiles spread over the 8 waves; a tile's Q projection, masked set-attention
//             (key-tile loop, all scores of two heads in registers), out-projection, LN1, FFN, LN2 stay in registers;
//             the last layer also emits the acquisition logits (model/head.py:27-33) and the target-row encodings
// With hd = 8 a k-step of the MFMA spans all four heads: the scores of head h use the Q^T fragment with the other
// heads' channels zeroed (2 registers x 2 lane groups survive), K stays whole and is read once per head PAIR; the PV
// product of a head pair shares the V^T fragment and keeps, per lane group, the accumulator of the head it belongs to.
#pragma once
#include "x3.h"

namespace s3 {

using img::group_sum4;
using img::u32x2;
using img::u32x4;
using x3::f16x8;
using x3::frag_value;
using x3::split_frag;
using x3::glds16;
using x3::group_max4;
using x3::vmax;
using x3::mfma3;
using x3::split2;
using x3::wait_vmcnt;

constexpr int D = 32, H = 4, HD = 8, F_MAX = 128;
constexpr int THREADS = 512, WAVES = THREADS / 64;                          // (pack / GMM kernels; the step kernel: 8 or 16 waves)
constexpr int NKP_MAX = 5, NKT_MAX = 2 * NKP_MAX, NK_MAX = 16 * NKT_MAX;     // key-tile pairs / tiles / keys per episode
constexpr int EPW_MAX = 16;                                                  // episodes per workgroup
constexpr float WSCALE = 256.f, WINV = 1.f / 256.f;
constexpr int PAIR_BYTES = 2048, PAIR_WORDS = 512;

__host__ __device__ inline int round_kb(int bytes) { return (bytes + 1023) & ~1023; }
// layer image: pairs Q0 Q1 K0 K1 V0 V1 O0 O1 | W1 [F/16] | W2 [(c, m): 2 c + m], then fp32
// bq (pre-scaled) bk bv bo | b1 [F] | b2 ln1w ln1b ln2w ln2b; padded to whole KB (LDS-DMA pieces)
__host__ __device__ inline int layer_pairs(int F) { return 8 + F / 8; }
__host__ __device__ inline int layer_params(int F) { return 9 * D + F; }
__host__ __device__ inline int layer_bytes(int F) { return round_kb(layer_pairs(F) * PAIR_BYTES + layer_params(F) * 4); }
// head image (acquisition head / one GMM head): W1 pairs [F/16], then b1 [F] | w2 [3][F] | b2 [4]
__host__ __device__ inline int head_pairs(int F) { return F / 16; }
__host__ __device__ inline int head_params(int F) { return 4 * F + 4; }
__host__ __device__ inline int head_bytes(int F) { return round_kb(head_pairs(F) * PAIR_BYTES + head_params(F) * 4); }
__host__ __device__ inline long image_words(int L, int F, int C) { return ((long)L * layer_bytes(F) + (long)(1 + C) * head_bytes(F)) / 4; }

// activation image in 16-byte pieces: [tile][hi | lo][lane = 16 g + row % 16]; a piece holds features 4 g + (0..3) and
// 16 + 4 g + (0..3) of its token row (= the B fragment element order of the single k-step)
__host__ __device__ inline long img_pieces(long tiles) { return tiles * 128; }
__device__ __forceinline__ long xpiece(long tile, int hl, int lane) { return (tile * 2 + hl) * 64 + lane; }

// LDS map of the step kernel (bytes): layer image | acquisition-head image | per episode K pairs [2 nkp] and V^T pairs
// [m 2][nkp] | per episode key list [32 nkp] | misc.  nkp = key-tile PAIRS kept per episode (the rollout's bound), epw =
// episodes per workgroup: both chosen by the host (step_plan).
constexpr int W_OFF = 0, W_BYTES = 51200;                 // layer_bytes(F_MAX)
constexpr int HEAD_OFF = W_OFF + W_BYTES, HEAD_BYTES = 19456;   // head_bytes(F_MAX)
constexpr int KV_OFF = HEAD_OFF + HEAD_BYTES;
constexpr int MISC_INTS = 2 * EPW_MAX + 32;               // n_ck [epw] | n_ak [epw] | wave counts [4][4] | running [4] | queues [2]
constexpr int SEL_PMAX = 256;                             // the in-kernel design selection holds an episode's logits in LDS: [epw][SEL_PMAX] floats behind misc
__host__ __device__ inline int kv_ep_bytes(int nkp) { return nkp * 8192; }
__host__ __device__ inline int step_lds_bytes(int epw, int nkp, bool sel = false) {
  return KV_OFF + epw * (kv_ep_bytes(nkp) + 32 * nkp * 4) + MISC_INTS * 4 + (sel ? epw * (SEL_PMAX * 4 + SEL_PMAX / 8) : 0);      // logits + context bit masks
}
constexpr int LDS_LIMIT = 160 * 1024;
static_assert(KV_OFF + (NKT_MAX * 4096 + NK_MAX * 4) * 2 + MISC_INTS * 4 <= LDS_LIMIT, "two episodes of 160 keys must fit");

// ---- weights -> fragment pairs -----------------------------------------------------------------------------------
struct PackArgs {
  int L, F, C;
  const float *in_proj_w[8], *in_proj_b[8], *out_proj_w[8], *out_proj_b[8], *lin1_w[8], *lin1_b[8], *lin2_w[8],
      *lin2_b[8], *n1w[8], *n1b[8], *n2w[8], *n2b[8];
  const float *acq_w1, *acq_b1, *acq_w2, *acq_b2;
  const float *gmm_w1[16], *gmm_b1[16], *gmm_w2[16], *gmm_b2[16];
  unsigned *out;
  unsigned *range_flag;
  int time_token;      // model.time_token: the acquisition head's W1 is [F, d + 1] (model/head.py:24-25); its last column goes to the
                       // free slot [2 F, 3 F) of the head's parameters and is folded into the bias per step (StepArgs.tau)
};

__global__ void pack_kernel(PackArgs a) {
  const long lw = layer_bytes(a.F) / 4, hw = head_bytes(a.F) / 4, total = image_words(a.L, a.F, a.C);
  const float qscale = rsqrtf((float)HD) * 1.44269504088896340736f;   // softmax runs in exp2
  const int F = a.F;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    unsigned v = 0;
    if (i < a.L * lw) {
      const int l = i / lw;
      const long o = i % lw, nfw = (long)layer_pairs(F) * PAIR_WORDS;
      if (o < nfw) {
        const int p = o / PAIR_WORDS, e = o % PAIR_WORDS;
        if (p < 6) v = x3::pair_word(a.in_proj_w[l] + (long)(p >> 1) * D * D, D, 16 * (p & 1), 0, e, (p < 2 ? qscale : 1.f) * WSCALE, a.range_flag);
        else if (p < 8) v = x3::pair_word(a.out_proj_w[l], D, 16 * (p & 1), 0, e, WSCALE, a.range_flag);
        else if (p < 8 + F / 16) v = x3::pair_word(a.lin1_w[l], D, 16 * (p - 8), 0, e, WSCALE, a.range_flag);
        else {
          const int q = p - 8 - F / 16;
          v = x3::pair_word(a.lin2_w[l], F, 16 * (q & 1), q >> 1, e, WSCALE, a.range_flag);
        }
      } else {
        const int p = o - nfw;
        float f = 0.f;
        if (p < 3 * D) f = a.in_proj_b[l][p] * (p < D ? qscale : 1.f);
        else if (p < 4 * D) f = a.out_proj_b[l][p - 3 * D];
        else if (p < 4 * D + F) f = a.lin1_b[l][p - 4 * D];
        else if (p < layer_params(F)) {
          const int q = p - 4 * D - F;
          f = q < D ? a.lin2_b[l][q] : q < 2 * D ? a.n1w[l][q - D] : q < 3 * D ? a.n1b[l][q - 2 * D]
            : q < 4 * D ? a.n2w[l][q - 3 * D] : a.n2b[l][q - 4 * D];
        }
        v = __float_as_uint(f);
      }
    } else {
      const long oh = i - a.L * lw;
      const int k = oh / hw;                          // 0: acquisition head, 1 + c: GMM head c
      const long o = oh % hw, nfw = (long)head_pairs(F) * PAIR_WORDS;
      const float *w1 = k == 0 ? a.acq_w1 : a.gmm_w1[k - 1], *b1 = k == 0 ? a.acq_b1 : a.gmm_b1[k - 1];
      const float *w2 = k == 0 ? a.acq_w2 : a.gmm_w2[k - 1], *b2 = k == 0 ? a.acq_b2 : a.gmm_b2[k - 1];
      const int nout = k == 0 ? 1 : 3;
      const int ldw1 = D + ((k == 0 && a.time_token) ? 1 : 0);
      if (o < nfw) {
        v = x3::pair_word(w1, ldw1, 16 * (int)(o / PAIR_WORDS), 0, (int)(o % PAIR_WORDS), WSCALE, a.range_flag);
      } else {
        const int p = o - nfw;
        float f = p < F ? b1[p] : p < (1 + nout) * F ? w2[p - F] : (p >= 4 * F && p < 4 * F + nout) ? b2[p - 4 * F] : 0.f;
        if (k == 0 && a.time_token && p >= 2 * F && p < 3 * F) f = w1[(long)(p - 2 * F) * ldw1 + D];      // the time column
        v = __float_as_uint(f);
      }
    }
    a.out[i] = v;
  }
}

// ---- X0 image from the cached fp32 point embeddings (model/embedder.py:128-214) ----------------------------------
struct AsmArgs {
  Geo g; int tpe;
  const float *Ex, *Ey; int ey_rows; const float *theta_tokens;
  u32x4 *X;
  unsigned *range_flag;            // f16 range guard (common.h): the layer-0 input is checked where it is assembled
};
__device__ __forceinline__ void embed_row8(const AsmArgs &a, int b, int row, int c, f32x4 &lo, f32x4 &hi) {
  const Geo &g = a.g;
  if (row < g.P + g.n_td) {
    const float *e = a.Ex + ((long)b * (g.P + g.n_td) + row) * D + c;
    lo = *reinterpret_cast<const f32x4 *>(e); hi = *reinterpret_cast<const f32x4 *>(e + 16);
    if (row < g.P && is_ctx(g, b, row)) {
      const float *y = a.Ey + ((long)b * a.ey_rows + row) * D + c;
      lo += *reinterpret_cast<const f32x4 *>(y); hi += *reinterpret_cast<const f32x4 *>(y + 16);
    }
  } else {
    const float *t = a.theta_tokens + (row - g.P - g.n_td) * D + c;
    lo = *reinterpret_cast<const f32x4 *>(t); hi = *reinterpret_cast<const f32x4 *>(t + 16);
  }
}
__device__ __forceinline__ void store_split8(u32x4 *X, long tile, int lane, const f32x4 &lo4, const f32x4 &hi4) {
  f16x8 h, l;
  split_frag(lo4, hi4, h, l);
  X[xpiece(tile, 0, lane)] = __builtin_bit_cast(u32x4, h);
  X[xpiece(tile, 1, lane)] = __builtin_bit_cast(u32x4, l);
}
__global__ void assemble_kernel(AsmArgs a) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long tiles = (long)a.g.B * a.tpe;
  if (i >= tiles * 64) return;
  const int lane = i & 63, gq = lane >> 4;
  const long tile = i >> 6;
  const int b = tile / a.tpe, row = (int)(tile % a.tpe) * 16 + (lane & 15);
  f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
  if (row < a.g.N) embed_row8(a, b, row, 4 * gq, lo, hi);
  x3::range_check8(a.range_flag, lo, hi);
  store_split8(a.X, tile, lane, lo, hi);
}

// ---- helpers of the step kernel ------------------------------------------------------------------------------------
// `bytes` (whole KB) from global to LDS, 1 KB pieces round-robin over the waves
__device__ __forceinline__ void dma_copy(const char *src, char *dst, int bytes, int wave, int lane) {
  for (int piece = wave; piece < (bytes >> 10); piece += WAVES) glds16(src + piece * 1024 + lane * 16, dst + piece * 1024);
}
struct Frag { f16x8 hi, lo; };
__device__ __forceinline__ Frag lds_pair(const char *base, int pair, int lane) {
  const char *p = base + pair * PAIR_BYTES + lane * 16;
  return Frag{*reinterpret_cast<const f16x8 *>(p), *reinterpret_cast<const f16x8 *>(p + 1024)};
}
__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ float relu_s(float v) { return fmaxf(v, 0.f); }

// ---- point embedders (model/embedder.py:47-57): E = W2 relu(W1 x + b1) + b2 for every point / target-data row ---------
// One kernel instead of the generic pair (hidden layer kernel + GEMM): the F hidden units of a 16-row tile stay in
// registers (first layer on the VALU: K = dim_x or dim_y <= 8 inputs), the second layer is the usual 3-term f16 split.
// A workgroup packs the W2 fragment pairs it needs into LDS itself (16 KB at F = 128).
struct EmbArgs {
  Src3 src; int rows_per_ep, B, K;
  const float *w1, *b1, *w2, *b2;       // [F, K], [F], [D, F], [D]
  float *E;                             // [B * rows_per_ep, D] fp32 rows
  unsigned *range_flag;                 // f16 range guard: W2, and (through the assembled input image) the hidden units
};
// bid / nwg: this workgroup's index among the nwg that embed `a` (embed_kernel embeds two row sets -- x and y -- in one launch)
template <int F>
__device__ __forceinline__ void embed_body(const EmbArgs &a, unsigned bid, unsigned nwg, unsigned *wimg, float *prm) {
  constexpr int NH = F / 16, PAIRS = F / 16;
  const int tid = threadIdx.x, lane = tid & 63, tok = lane & 15, g = lane >> 4, wave = tid >> 6;
  for (int e = tid; e < PAIRS * PAIR_WORDS; e += 256) {      // pair (c, m) = k-step c of output tile m
    const int p = e / PAIR_WORDS;
    wimg[e] = x3::pair_word(a.w2, F, 16 * (p & 1), p >> 1, e % PAIR_WORDS, WSCALE, bid == 0 ? a.range_flag : nullptr);
  }
  for (int i = tid; i < D; i += 256) prm[i] = a.b2[i];
  for (int i = tid; i < F; i += 256) prm[D + i] = a.b1[i];
  for (int i = tid; i < F * a.K; i += 256) prm[D + F + i] = a.w1[i];
  __syncthreads();
  const float *b2 = prm, *b1 = prm + D, *w1 = prm + D + F;
  const long total = (long)a.B * a.rows_per_ep, ntiles = (total + 15) / 16;
  const int K = a.K;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (long tile = (long)bid * 4 + wave; tile < ntiles; tile += (long)nwg * 4) {
    const long row = 16 * tile + tok, rr = min(row, total - 1);
    const int b = rr / a.rows_per_ep, p = rr % a.rows_per_ep;
    const float *x;
    if (p < a.src.n[0]) x = a.src.p[0] + ((long)b * a.src.n[0] + p) * K;
    else if (p < a.src.n[0] + a.src.n[1]) x = a.src.p[1] + ((long)b * a.src.n[1] + (p - a.src.n[0])) * K;
    else x = a.src.p[2] + ((long)b * a.src.n[2] + (p - a.src.n[0] - a.src.n[1])) * K;
    float xin[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) xin[k] = k < K ? x[k] : 0.f;
    f32x4 hid[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int u = 16 * i + 4 * g + r;
        float acc = b1[u];
#pragma unroll
        for (int k = 0; k < 8; ++k)      // (compile-time indices: a runtime-indexed xin[] would live in scratch memory)
          if (k < K) acc = fmaf(xin[k], w1[u * K + k], acc);
        hid[i][r] = relu_s(acc);
      }
    f32x4 y0 = z4, y1 = z4;
#pragma unroll
    for (int c = 0; c < NH / 2; ++c) {
      f16x8 hh, hl;
      split_frag(hid[2 * c], hid[2 * c + 1], hh, hl);
      const Frag v0 = lds_pair(reinterpret_cast<const char *>(wimg), 2 * c, lane), v1 = lds_pair(reinterpret_cast<const char *>(wimg), 2 * c + 1, lane);
      mfma3(y0, v0.hi, v0.lo, hh, hl);
      mfma3(y1, v1.hi, v1.lo, hh, hl);
    }
    if (row < total) {
      float *e = a.E + row * D;
      *reinterpret_cast<f32x4 *>(e + 4 * g) = y0 * WINV + ld4(b2 + 4 * g);
      *reinterpret_cast<f32x4 *>(e + 16 + 4 * g) = y1 * WINV + ld4(b2 + 16 + 4 * g);
    }
  }
}
// the first nwg_a workgroups embed `a` (the x rows), the rest `b` (the y rows; nwg_a == gridDim.x: one row set only)
template <int F>
__global__ __launch_bounds__(256) void embed_kernel(EmbArgs a, EmbArgs b, unsigned nwg_a) {
  __shared__ __attribute__((aligned(16))) unsigned wimg[(F / 16) * PAIR_WORDS];
  __shared__ float prm[D + F + F * 8];                       // b2 | b1 | w1 [F][K]
  if (blockIdx.x < nwg_a) embed_body<F>(a, blockIdx.x, nwg_a, wimg, prm);
  else embed_body<F>(b, blockIdx.x - nwg_a, gridDim.x - nwg_a, wimg, prm);
}

// (a, b) = LayerNorm over the 32 features of each token: 8 values per lane x 4 lane groups, fp32, two passes
__device__ __forceinline__ float layer_norm32(f32x4 &a, f32x4 &b, const float *w, const float *bb, int g) {
  const float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
  const float mean = group_sum4(s) * (1.f / D);
  float q = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) { a[r] -= mean; b[r] -= mean; q = fmaf(a[r], a[r], q); q = fmaf(b[r], b[r], q); }
  const float rstd = __builtin_amdgcn_rsqf(group_sum4(q) * (1.f / D) + 1e-5f);     // (v_rsq_f32: 1 ulp)
  const f32x4 w0 = ld4(w + 4 * g), w1 = ld4(w + 16 + 4 * g), b0 = ld4(bb + 4 * g), b1 = ld4(bb + 16 + 4 * g);
#pragma unroll
  for (int r = 0; r < 4; ++r) { a[r] = fmaf(a[r] * rstd, w0[r], b0[r]); b[r] = fmaf(b[r] * rstd, w1[r], b1[r]); }
  return rstd;      // NaN iff an input was NaN / inf (the f16 range guard accumulates it)
}

// softmax numerators of one head over NKT key tiles (exp2 domain), in place; returns 1 / denominator
template <int NKT>
__device__ __forceinline__ float softmax_tiles(f32x4 (&s)[NKT], float inf) {
  float m0 = vmax(s[0][0], s[0][1], inf), m1 = vmax(s[0][2], s[0][3], inf);      // (vmax: no canonicalising moves in front of the MFMA results, x3.h)
#pragma unroll
  for (int kt = 1; kt < NKT; ++kt) {
    m0 = vmax(vmax(m0, s[kt][0], inf), s[kt][1], inf);
    m1 = vmax(vmax(m1, s[kt][2], inf), s[kt][3], inf);
  }
  const float mx = group_max4(vmax(m0, m1, inf), inf);
  // (plain v_sub / v_add: full rate; the packed forms are not, and cost register moves; the sums start from the first pair, not from 0)
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) s[kt][r] = __builtin_amdgcn_exp2f(s[kt][r] - mx);
  float acc0 = s[0][0] + s[0][1], acc1 = s[0][2] + s[0][3];
#pragma unroll
  for (int kt = 1; kt < NKT; ++kt) {
    acc0 += s[kt][0] + s[kt][1];
    acc1 += s[kt][2] + s[kt][3];
  }
  return __builtin_amdgcn_rcpf(group_sum4(acc0 + acc1));                         // (v_rcp_f32: 1 ulp)
}

// masked set-attention of one token tile against NKT key tiles (model/encoder.py:8-46): S^T = K Q_h^T in the exp2
// domain (scale folded into Wq), softmax over the keys of a token, O^T = V^T P.  kv: this episode's K / V^T pairs in
// LDS (+ lane * 16; V^T pairs [m][nkp] at kv_v); nv4 = (number of keys the lane's token sees) - 4 g.
template <int NKT, int CAP>
__device__ __forceinline__ void attention(const f16x8 &qh, const f16x8 &ql, const char *kv, int nv4, int g, f32x4 (&o)[2]) {
  constexpr int kv_v = CAP * 4096, nkp = CAP;     // LDS slot of an episode: K pairs [2 CAP] | V^T pairs [m][CAP]
  constexpr int NKP = (NKT + 1) / 2;              // key tiles NKT (an odd count: the last V^T pair is half used), pairs NKP
  const u32x4 qhu = __builtin_bit_cast(u32x4, qh), qlu = __builtin_bit_cast(u32x4, ql);
  const bool glo = g < 2;
  unsigned ma = glo ? 0xFFFFFFFFu : 0u;      // lane groups 0, 1 carry the first head of a pair, 2, 3 the second
  asm("" : "+v"(ma));                        // (opaque: the optimiser otherwise turns the full-rate v_and back into v_cndmask selects)
  const unsigned mbm = ~ma;
  const float inf = x3::opaque_inf();
  f32x4 mb[NKT];                                               // key mask of this lane's token: the initial accumulator of both heads' scores
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) mb[kt][r] = (16 * kt + r) < nv4 ? 0.f : -INFINITY;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    // heads 2 m (channels 16 m + 0..7: lane groups 0, 1) and 2 m + 1 (lane groups 2, 3): registers 2 m, 2 m + 1
    u32x4 ah = {0u, 0u, 0u, 0u}, al = ah, bh = ah, bl = ah;
    ah[2 * m] = qhu[2 * m] & ma; ah[2 * m + 1] = qhu[2 * m + 1] & ma;
    al[2 * m] = qlu[2 * m] & ma; al[2 * m + 1] = qlu[2 * m + 1] & ma;
    bh[2 * m] = qhu[2 * m] & mbm; bh[2 * m + 1] = qhu[2 * m + 1] & mbm;
    bl[2 * m] = qlu[2 * m] & mbm; bl[2 * m + 1] = qlu[2 * m + 1] & mbm;
    const f16x8 qah = __builtin_bit_cast(f16x8, ah), qal = __builtin_bit_cast(f16x8, al);
    const f16x8 qbh = __builtin_bit_cast(f16x8, bh), qbl = __builtin_bit_cast(f16x8, bl);
    f32x4 sa[NKT], sb[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const f16x8 kh = *reinterpret_cast<const f16x8 *>(kv + kt * PAIR_BYTES), kl = *reinterpret_cast<const f16x8 *>(kv + kt * PAIR_BYTES + 1024);
      sa[kt] = mb[kt]; sb[kt] = mb[kt];
      mfma3(sa[kt], kh, kl, qah, qal);
      mfma3(sb[kt], kh, kl, qbh, qbl);
    }
    // the V^T pairs of this head pair are requested before the softmax arithmetic and land while it runs
    f16x8 vh[NKP], vl[NKP];
#pragma unroll
    for (int kb = 0; kb < NKP; ++kb) {
      const char *vp = kv + kv_v + (m * nkp + kb) * PAIR_BYTES;
      vh[kb] = *reinterpret_cast<const f16x8 *>(vp); vl[kb] = *reinterpret_cast<const f16x8 *>(vp + 1024);
    }
    const float inva = softmax_tiles<NKT>(sa, inf), invb = softmax_tiles<NKT>(sb, inf);
    f32x4 oa = {0.f, 0.f, 0.f, 0.f}, ob = oa;
#pragma unroll
    for (int kb = 0; kb < NKP; ++kb) {
      f16x8 ph, pl;
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      split_frag(sa[2 * kb], 2 * kb + 1 < NKT ? sa[2 * kb + 1 < NKT ? 2 * kb + 1 : 0] : z4, ph, pl);
      mfma3(oa, vh[kb], vl[kb], ph, pl);
      split_frag(sb[2 * kb], 2 * kb + 1 < NKT ? sb[2 * kb + 1 < NKT ? 2 * kb + 1 : 0] : z4, ph, pl);
      mfma3(ob, vh[kb], vl[kb], ph, pl);
    }
    {
      const float inv = glo ? inva : invb;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[m][r] = (glo ? oa[r] : ob[r]) * inv;
    }
  }
}

// In-kernel phase stamps (S3_STAMPS diagnostic build only, tools/s3_stamps.py): s_memtime deltas accumulated per wave.
//   0 prologue (DMA issue, patched row, key lists, first barrier)   1 K / V tiles   2 barrier behind K / V
//   3 tile load + Q projection   4 attention   5 out-projection .. LN2, stores, logits   6 barrier behind the tiles
//   7 next layer's weights (DMA + wait + barrier)   8 whole kernel
#ifdef S3_STAMPS
#define S3_NSTAMP 9
struct Stamps {
  unsigned long long t_prev, acc[S3_NSTAMP];
  __device__ __forceinline__ void start() { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory"); }
  __device__ __forceinline__ void lap(int k) {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    acc[k] += t - t_prev;
    t_prev = t;
  }
};
#define S3_LAP(k) stamps.lap(k)
#else
#define S3_LAP(k)
#endif

// ---- one design step of EPW episodes: every encoder layer + the acquisition logits ----------------------------------
// Design selection of the step (model/head.py:36-60: softmax over the remaining candidates, argmax / Categorical sample / forced index,
// log-probability, role update) at the end of the step kernel (P <= SEL_PMAX), one wave per episode of the workgroup, instead of a launch
// of its own (acq_select_wave_kernel: 5.5 us x T of the 2.8 ms headline rollout).  Same arithmetic, same order of operations as that kernel.
struct SelArgs {
  int on, mode;                                  // ALINE_SELECT_*
  const float *uniform;                          // [B] of this step
  const int64_t *forced; int forced_stride;
  int64_t *idx; int idx_stride;
  int *slot; int slot_stride;
  float *log_prob; int lp_stride;
  float *zt; int zt_stride, zt_width;
  int *role_out;
  unsigned *range_flag;
};

struct StepArgs {
  Geo g; int tpe, L, F;
  int epw;                         // episodes per workgroup (the LDS slot of an episode holds the variant's MAXNKP key-tile pairs)
  int nk2;                         // key tiles (even) to compute this step: 2 ceil(bound of the key count / 32) <= 2 nkp
  int order;                       // > 0: role value of the point chosen at the previous step (its X0 row becomes Ex + Ey)
  const unsigned *img;             // L layer images, then the head images
  u32x4 *X0, *XW;                  // input image (patched in place) / work image
  AsmArgs emb;                     // sources of the patched row
  float *logits; int NP;           // logits[b * NP + row]
  u32x4 *zimg; long zrow0;         // dense-row image of the target rows of all steps (null: not wanted)
  u32x4 *zq; long zq_row0;         // dense-row image of the P point rows of all steps (null: posterior_out_query not wanted)
  float tau;                       // time token of this step (t / T or (T - t) / T: model/head.py:342-345), 0 without one
  const SelArgs *selp;             // in-kernel design selection: this step's arguments in device memory (sel_args_kernel), or null: the selection is a launch of its own.
                                   // (as kernel arguments they cost the tile loop its scalar registers: 98 SGPR spills, the rollout 6 % slower)
  float *sv; long sv_rows, sv_row0;   // aline_rollout.saved_acts (null: not wanted): [2 L + 1][sv_rows][32] fp32; this step's rows start at sv_row0
#ifdef S3_STAMPS
  unsigned long long *stamps;      // [8 waves][S3_NSTAMP] of workgroup 0
#endif
};

// the T argument sets of a rollout's in-kernel selections: the per-step pointers are the rollout's [.., T] arrays advanced by the step
__global__ void sel_args_kernel(SelArgs base, int T, long B, SelArgs *__restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  SelArgs q = base;
  if (q.uniform) q.uniform += (size_t)t * B;
  if (q.forced) q.forced += t;
  if (q.idx) q.idx += t;
  if (q.slot) q.slot += t;
  if (q.log_prob) q.log_prob += t;
  if (q.zt) q.zt += (size_t)t * B * q.zt_stride;
  out[t] = q;
}

// one wave: the selection of episode b from its logits in LDS (acq_select_wave_kernel, kernels.h, with lg[] read from LDS)
__device__ __forceinline__ void select_episode(const Geo &G, const SelArgs &a, int b, const float *lgs, const unsigned long long *cmask, int lane) {
  // cmask[c]: bit l = point 64 c + l is a context point (the ballots of the step's prologue): no role reads, no ballots here
  const int P = G.P;
  const float u01 = a.mode == 1 ? a.uniform[b] : 0.f;      // (requested first: it lands while the softmax runs)
  const unsigned long long below = (1ull << lane) - 1ull;
  float lg[4], pr[4];
  bool isq[4];
  int ci[4], nq = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int p = 64 * c + lane;
    const int nv = min(max(P - 64 * c, 0), 64);
    const unsigned long long valid = nv >= 64 ? ~0ull : ((1ull << nv) - 1ull);
    const unsigned long long bal = valid & ~cmask[c];       // the remaining candidates of this chunk
    lg[c] = p < P ? lgs[p] : -INFINITY;
    isq[c] = (bal >> lane) & 1ull;
    ci[c] = nq + __popcll(bal & below);
    nq += __popcll(bal);
  }
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < 4; ++c) mx = fmaxf(mx, isq[c] ? lg[c] : -INFINITY);
  mx = wave_max_dpp(mx);
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) { pr[c] = isq[c] ? __expf(lg[c] - mx) : 0.f; sum += pr[c]; }
  sum = wave_sum_dpp(sum);
  if (lane == 0 && !(sum <= 3.4e38f)) range_raise(a.range_flag, ALINE_RANGE_ACT);     // a NaN / +inf logit
  const float inv = 1.f / sum;
  float tot = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) { pr[c] *= inv; tot += pr[c]; }
  tot = wave_sum_dpp(tot);
  if (a.zt) {
    float *z = a.zt + (long)b * a.zt_stride;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (isq[c] && ci[c] < a.zt_width) z[ci[c]] = pr[c];
    for (int i = nq + lane; i < a.zt_width; i += 64) z[i] = 0.f;
  }
  int choice = 0;
  if (a.mode == 0) {            // argmax, first maximal index (torch.max semantics)
    float best = -1.f; int bi = 0x7fffffff;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (isq[c] && (pr[c] > best || (pr[c] == best && ci[c] < bi))) { best = pr[c]; bi = ci[c]; }
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    choice = bi == 0x7fffffff ? 0 : bi;
  } else if (a.mode == 2) {
    choice = (int)a.forced[(long)b * a.forced_stride];
    choice = min(max(choice, 0), nq - 1);
  } else {                      // inverse CDF of Categorical(probs = zt / sum zt)
    const float u = u01 * tot;
    float incl[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) incl[c] = wave_scan_sum(pr[c]);      // (four independent DPP scans)
    float run = 0.f; int found = nq - 1; bool done = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const unsigned long long bal = __ballot(isq[c] && (run + incl[c]) > u);
      if (!done && bal) { found = __builtin_amdgcn_readlane(ci[c], __builtin_amdgcn_readfirstlane(__ffsll((long long)bal) - 1)); done = true; }
      run += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(incl[c]), 63));
    }
    choice = found;
  }
  float val = 0.f; int sl = 0;      // (exactly one lane of one chunk holds the chosen candidate: read it, no reduction)
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const unsigned long long hit = __ballot(isq[c] && ci[c] == choice);
    if (hit) {
      const int ln = __builtin_amdgcn_readfirstlane(__ffsll((long long)hit) - 1);
      val = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pr[c]), ln));
      sl = 64 * c + ln;
    }
  }
  if (a.mode != 0) val = val / tot;       // Categorical(probs).log_prob uses probs / probs.sum() ...
  if (lane == 0) {
    if (a.mode != 0) val = fminf(fmaxf(val, 1.1920929e-07f), 1.f - 1.1920929e-07f);      // ... clamped to [eps, 1 - eps]
    if (a.idx) a.idx[(long)b * a.idx_stride] = choice;
    if (a.log_prob) a.log_prob[(long)b * a.lp_stride] = logf(val);
    if (a.slot) a.slot[(long)b * a.slot_stride] = sl;
    if (a.role_out) a.role_out[(long)b * P + sl] = (P - nq) + 1;
  }
}

// SAVE: the instantiation of training rollouts (StepArgs.sv: layer inputs / attention outputs kept for the backward); the inference
// instantiation carries none of its registers or branches.
template <int F, int NW, int MAXNKP, bool PREFETCH, bool SAVE = false>
__global__ __launch_bounds__(NW * 64) void step_kernel(StepArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const Geo &G = a.g;
  const int tid = threadIdx.x, lane = tid & 63, tok = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tpe = a.tpe, n_t = G.n_td + G.n_th, epw = a.epw;
  constexpr int nkp = MAXNKP;                      // key-tile pairs an episode's LDS slot holds
  constexpr int NH = F / 16;                       // hidden tiles (= W1 pairs = W2 pairs)
  const int lbytes = layer_bytes(F);
  char *wl = lds + W_OFF, *whd = lds + HEAD_OFF;
  constexpr int kv_ep = nkp * 8192, kv_v = nkp * 4096, nkcap = 32 * nkp;
  int *keyrow = reinterpret_cast<int *>(lds + KV_OFF + epw * kv_ep), *misc = keyrow + epw * nkcap;
  int *n_ck = misc, *n_ak = misc + EPW_MAX, *wcnt = misc + 2 * EPW_MAX, *run = wcnt + 16, *queue = run + 4;
  float *lgs = reinterpret_cast<float *>(misc + MISC_INTS);         // (in-kernel selection) logits [epw][SEL_PMAX]
  unsigned long long *cmask = reinterpret_cast<unsigned long long *>(lgs + epw * SEL_PMAX);      // ... and context bit masks [epw][SEL_PMAX / 64] (MISC_INTS is even: 8-byte aligned)
  const char *gimg = reinterpret_cast<const char *>(a.img);
#ifdef S3_STAMPS
  Stamps stamps{};
  stamps.start();
  const unsigned long long t_begin = stamps.t_prev;
#endif
  for (int piece = wave; piece < (lbytes >> 10); piece += NW) glds16(gimg + piece * 1024 + lane * 16, wl + piece * 1024);
  for (int piece = wave; piece < (head_bytes(F) >> 10); piece += NW) glds16(gimg + (long)a.L * lbytes + piece * 1024 + lane * 16, whd + piece * 1024);
  if (tid == 0) { queue[0] = 0; queue[1] = 0; }
  // per episode, by ONE WAVE (ballot compaction: no workgroup barriers; the 256-thread version with three barriers per 256 slots
  // and a second pass for the episodes beyond NW / 4 was 7 % of the step at the headline shape): the patched row, then the key
  // list (context rows in slot order, then the visible targets)
  for (int e = wave; e < epw; e += NW) {
    const int b_e = blockIdx.x * epw + e;
    const bool ep_ok = b_e < G.B;
    const int bh = min(b_e, G.B - 1);
    int *list = keyrow + e * nkcap;
    int n = 0, slot = -1;
    for (int c0 = 0; c0 < G.P; c0 += 64) {
      const int p = c0 + lane;
      const int rl = (ep_ok && p < G.P) ? (G.role ? G.role[(long)bh * G.P + p] : (p < G.n_ctx ? 1 : 0)) : 0;
      if (a.order > 0 && rl == a.order) slot = p;
      const bool key = rl > 0;
      const unsigned long long bal = __ballot(key);
      const int k = n + __popcll(bal & ((1ull << lane) - 1ull));
      if (key && k < nkcap) list[k] = p;
      n += __popcll(bal);
      if (a.selp && lane == 0 && c0 < SEL_PMAX) cmask[e * (SEL_PMAX / 64) + (c0 >> 6)] = bal;
    }
    n = min(n, nkcap);
    if (lane == 0) n_ck[e] = n;
    for (int c0 = 0; c0 < n_t; c0 += 64) {
      const int j = c0 + lane;
      const bool key = ep_ok && j < n_t && (!G.tmask || G.tmask[j]);
      const unsigned long long bal = __ballot(key);
      const int k = n + __popcll(bal & ((1ull << lane) - 1ull));
      if (key && k < nkcap) list[k] = G.P + j;
      n += __popcll(bal);
    }
    if (lane == 0) n_ak[e] = min(n, nkcap);
    if (a.order > 0 && ep_ok) {      // the point chosen at the previous step enters the context: its X0 row becomes Ex + Ey
      slot = __reduce_max_sync(~0ull, slot);
      if (slot >= 0 && lane < 4) {
        f32x4 lo, hi;
        embed_row8(a.emb, bh, slot, 4 * lane, lo, hi);
        x3::range_check8(a.emb.range_flag, lo, hi);
        store_split8(a.X0, (long)bh * tpe + (slot >> 4), lane * 16 + (slot & 15), lo, hi);
      }
    }
  }
  wait_vmcnt<0>();
  __syncthreads();
  if (a.tau != 0.f && tid < F) {      // W1 [z | t] = W1[:, :d] z + t W1[:, d]: the step's time token is a bias (read in the last layer, barriers away)
    float *hp = reinterpret_cast<float *>(whd + head_pairs(F) * PAIR_BYTES);
    hp[tid] = fmaf(a.tau, hp[2 * F + tid], hp[tid]);
  }
  S3_LAP(0);
  float range_chk = 0.f;

  for (int l = 0; l < a.L; ++l) {
    const bool last = l == a.L - 1;
    const u32x4 *xin = l == 0 ? a.X0 : a.XW;
    const float *prm = reinterpret_cast<const float *>(wl + layer_pairs(F) * PAIR_BYTES);
    const float *bq = prm, *bk = prm + D, *bv = prm + 2 * D, *bo = prm + 3 * D, *b1 = prm + 4 * D, *b2 = b1 + F;
    const float *ln1w = b2 + D, *ln1b = ln1w + D, *ln2w = ln1b + D, *ln2b = ln2w + D;
    // ---- K / V of the key rows -> LDS ---------------------------------------------------------------------------
    // jobs: (episode, key tile, K | V) -- K and V of a key tile as separate jobs: twice as many, half as long (at the headline
    // shape 16 jobs for the 12 waves instead of 8: the phase in front of the barrier is latency-bound)
    const int nk2 = a.nk2;
    for (;;) {
      int job = 0;
      if (lane == 0) job = atomicAdd(queue, 1);
      job = __builtin_amdgcn_readfirstlane(job);
      if (job >= 2 * epw * nk2) break;
      const int want_v = job & 1;
      int e = 0, kt = job >> 1;
      while (kt >= nk2) { kt -= nk2; ++e; }
      const int b = min(blockIdx.x * epw + e, G.B - 1);
      const int key = 16 * kt + tok;
      const int row = key < n_ak[e] ? keyrow[e * nkcap + key] : -1;
      f16x8 xh = {0, 0, 0, 0, 0, 0, 0, 0}, xl = xh;
      if (row >= 0) {
        const long tl = (long)b * tpe + (row >> 4);
        xh = __builtin_bit_cast(f16x8, xin[xpiece(tl, 0, 16 * g + (row & 15))]);
        xl = __builtin_bit_cast(f16x8, xin[xpiece(tl, 1, 16 * g + (row & 15))]);
      }
      char *kv = lds + KV_OFF + e * kv_ep + lane * 16;
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      if (!want_v) {   // K^T = Wk KX^T: rows = channels, columns = keys -> the A fragment pair of key tile kt
        f32x4 y0 = z4, y1 = z4;
        const Frag w0 = lds_pair(wl, 2, lane), w1 = lds_pair(wl, 3, lane);
        mfma3(y0, w0.hi, w0.lo, xh, xl);
        mfma3(y1, w1.hi, w1.lo, xh, xl);
        f16x8 fh, fl;
        split_frag(y0 * WINV + ld4(bk + 4 * g), y1 * WINV + ld4(bk + 16 + 4 * g), fh, fl);
        *reinterpret_cast<f16x8 *>(kv + kt * PAIR_BYTES) = fh;
        *reinterpret_cast<f16x8 *>(kv + kt * PAIR_BYTES + 1024) = fl;
      } else {
        // V = KX Wv^T with the operands swapped (rows = keys, columns = channels): accumulator i holds
        // V[key 16 kt + 4 g + r][channel 16 i + tok] = half (kt & 1) of this lane's piece of the V^T pair (i, kt / 2)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          f32x4 v = z4;
          const Frag w = lds_pair(wl, 4 + i, lane);
          mfma3(v, xh, xl, w.hi, w.lo);
          const float bvv = bv[16 * i + tok];
          unsigned h0, l0, h1, l1;
          split2(v[0] * WINV + bvv, v[1] * WINV + bvv, h0, l0);
          split2(v[2] * WINV + bvv, v[3] * WINV + bvv, h1, l1);
          char *vp = kv + kv_v + (i * nkp + (kt >> 1)) * PAIR_BYTES + 8 * (kt & 1);
          *reinterpret_cast<u32x2 *>(vp) = (u32x2){h0, h1};
          *reinterpret_cast<u32x2 *>(vp + 1024) = (u32x2){l0, l1};
        }
      }
    }
    S3_LAP(1);
    __syncthreads();
    if (tid == 0) queue[0] = 0;
    S3_LAP(2);
    // ---- token tiles: a work queue (the waves of a SIMD do not run at the same pace), the tiles with candidate rows (which
    // may see the targets) first; the next tile's rows are requested before the current tile is computed -------------
    const int n_heavy = min((G.P + 15) >> 4, tpe), n_light = tpe - n_heavy, n_jobs = epw * tpe;
    auto tile_of = [&](int job, int &e, int &j) {       // (scalar: a subtract loop over <= 16 episodes instead of a division)
      const bool heavy = job < epw * n_heavy;
      const int per = heavy ? n_heavy : n_light;
      int q = heavy ? job : job - epw * n_heavy;
      e = 0;
      while (q >= per) { q -= per; ++e; }
      j = heavy ? q : n_heavy + q;
    };
    auto next_job = [&]() {
      int job = 0;
      if (lane == 0) job = atomicAdd(queue + 1, 1);
      return __builtin_amdgcn_readfirstlane(job);
    };
    auto load_rows = [&](int job, f16x8 &h, f16x8 &lo_) {
      int e, j;
      tile_of(min(job, n_jobs - 1), e, j);
      const long tl = (long)min(blockIdx.x * epw + e, G.B - 1) * tpe + j;
      h = __builtin_bit_cast(f16x8, xin[xpiece(tl, 0, lane)]);
      lo_ = __builtin_bit_cast(f16x8, xin[xpiece(tl, 1, lane)]);
    };
    int job = next_job();
    f16x8 xh_next, xl_next;
    if (PREFETCH) load_rows(job, xh_next, xl_next);
    while (job < n_jobs) {
      int e, j;
      tile_of(job, e, j);
      const int b = blockIdx.x * epw + e;
      f16x8 xh, xl;
      if (PREFETCH) { xh = xh_next; xl = xl_next; }
      else load_rows(job, xh, xl);
      job = next_job();
      if (PREFETCH) load_rows(job, xh_next, xl_next);
      if (b >= G.B) continue;
      const long tl = (long)b * tpe + j;
      const int r = 16 * j + tok, rc = min(r, G.N - 1);
      const int nck = n_ck[e], nak = n_ak[e];
      const bool isq = rc < G.P && !is_ctx(G, b, rc);
      const bool hasq = 16 * j < G.P;                 // (wave-uniform) some row of the tile may see the targets
      const int nkp_t = max(1, ((hasq ? nak : nck) + 31) >> 5), nkt_t = max(1, ((hasq ? nak : nck) + 15) >> 4);
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      // Q^T = Wq X^T (pre-scaled by log2(e) / sqrt(hd))
      f16x8 qh, ql;
      {
        f32x4 y0 = z4, y1 = z4;
        const Frag w0 = lds_pair(wl, 0, lane), w1 = lds_pair(wl, 1, lane);
        mfma3(y0, w0.hi, w0.lo, xh, xl);
        mfma3(y1, w1.hi, w1.lo, xh, xl);
        split_frag(y0 * WINV + ld4(bq + 4 * g), y1 * WINV + ld4(bq + 16 + 4 * g), qh, ql);
      }
      S3_LAP(3);
      // training rollouts: the layer's input and (below) attention output of the tile's rows as fp32 rows, for the backward
      float *svrow = (SAVE && a.sv && r < G.N) ? a.sv + (a.sv_row0 + (long)b * G.N + r) * D + 4 * g : nullptr;
      if (svrow) {
        float *px = svrow + (long)l * a.sv_rows * D;
        *reinterpret_cast<f32x4 *>(px) = frag_value(xh, xl, 0);
        *reinterpret_cast<f32x4 *>(px + 16) = frag_value(xh, xl, 1);
      }
      f32x4 o[2];
      {
        const char *kv = lds + KV_OFF + e * kv_ep + lane * 16;
        const int nv4 = (isq ? nak : nck) - 4 * g;
        // (the host never asks for more key tiles than the variant holds.)  The 12- / 16-wave variant (<= 64 keys) has a body per
        // key-tile count -- at the headline shape the first 14 steps need one tile --, the 8-wave variant per pair
        if constexpr (MAXNKP == 2) {
          if (nkt_t >= 4) attention<4, MAXNKP>(qh, ql, kv, nv4, g, o);
          else if (nkt_t == 3) attention<3, MAXNKP>(qh, ql, kv, nv4, g, o);
          else if (nkt_t == 2) attention<2, MAXNKP>(qh, ql, kv, nv4, g, o);
          else attention<1, MAXNKP>(qh, ql, kv, nv4, g, o);
        } else {
          // (per PAIR of key tiles: bodies for the odd counts were measured at cfg3, 1 % for 40 % more code)
          if (MAXNKP >= 5 && nkp_t >= 5) attention<10, MAXNKP>(qh, ql, kv, nv4, g, o);
          else if (MAXNKP >= 4 && nkp_t == 4) attention<8, MAXNKP>(qh, ql, kv, nv4, g, o);
          else if (MAXNKP >= 3 && nkp_t == 3) attention<6, MAXNKP>(qh, ql, kv, nv4, g, o);
          else if (nkp_t == 2) attention<4, MAXNKP>(qh, ql, kv, nv4, g, o);
          else attention<2, MAXNKP>(qh, ql, kv, nv4, g, o);
        }
      }
#ifdef S3_STAMPS
      asm volatile("" :: "v"(o[0]), "v"(o[1]));
#endif
      S3_LAP(4);
      if (svrow) {
        float *pa = svrow + (long)(a.L + 1 + l) * a.sv_rows * D;
        *reinterpret_cast<f32x4 *>(pa) = o[0];
        *reinterpret_cast<f32x4 *>(pa + 16) = o[1];
      }
      // X1 = LN1(X + bo + Wo A)
      f32x4 x1a, x1b;
      f16x8 x1h, x1l;
      {
        f16x8 ah, al;
        split_frag(o[0], o[1], ah, al);
        f32x4 y0 = z4, y1 = z4;
        const Frag w0 = lds_pair(wl, 6, lane), w1 = lds_pair(wl, 7, lane);
        mfma3(y0, w0.hi, w0.lo, ah, al);
        mfma3(y1, w1.hi, w1.lo, ah, al);
        x1a = y0 * WINV + ld4(bo + 4 * g) + frag_value(xh, xl, 0);
        x1b = y1 * WINV + ld4(bo + 16 + 4 * g) + frag_value(xh, xl, 1);
        range_chk += layer_norm32(x1a, x1b, ln1w, ln1b, g);
        split_frag(x1a, x1b, x1h, x1l);
      }
      // X2 = LN2(X1 + b2 + W2 relu(W1 X1 + b1)): the hidden units of the tile in registers, GH tiles of 16 at a time (all of them
      // in the 256-register variant; four in the 128-register variant, which otherwise spills)
      f32x4 y0 = z4, y1 = z4;
      {
        constexpr int GH = (NW > 8 && NH > 4) ? (NH % 4 == 0 ? 4 : 2) : NH;
#pragma unroll
        for (int h0i = 0; h0i < NH; h0i += GH) {
          f32x4 hid[GH];
#pragma unroll
          for (int i = 0; i < GH; ++i) {
            const Frag u = lds_pair(wl, 8 + h0i + i, lane);
            hid[i] = z4;
            mfma3(hid[i], u.hi, u.lo, x1h, x1l);
          }
          f16x8 hh[GH / 2], hl[GH / 2];
#pragma unroll
          for (int c = 0; c < GH / 2; ++c) {
            const int cc = h0i / 2 + c;
            f32x4 h0 = hid[2 * c] * WINV + ld4(b1 + 32 * cc + 4 * g), h1 = hid[2 * c + 1] * WINV + ld4(b1 + 32 * cc + 16 + 4 * g);
#pragma unroll
            for (int q = 0; q < 4; ++q) { h0[q] = relu_s(h0[q]); h1[q] = relu_s(h1[q]); }
            split_frag(h0, h1, hh[c], hl[c]);
          }
#pragma unroll
          for (int c = 0; c < GH / 2; ++c) {
            const int cc = h0i / 2 + c;
            const Frag v0 = lds_pair(wl, 8 + NH + 2 * cc, lane), v1 = lds_pair(wl, 8 + NH + 2 * cc + 1, lane);
            mfma3(y0, v0.hi, v0.lo, hh[c], hl[c]);
            mfma3(y1, v1.hi, v1.lo, hh[c], hl[c]);
          }
          if (GH < NH) __builtin_amdgcn_sched_barrier(0);     // keep the groups apart (the scheduler would merge their loads)
        }
      }
      f32x4 x2a = y0 * WINV + ld4(b2 + 4 * g) + x1a, x2b = y1 * WINV + ld4(b2 + 16 + 4 * g) + x1b;
      range_chk += layer_norm32(x2a, x2b, ln2w, ln2b, g);
      f16x8 oh, ol;
      split_frag(x2a, x2b, oh, ol);
      if (!last) {
        a.XW[xpiece(tl, 0, lane)] = __builtin_bit_cast(u32x4, oh);
        a.XW[xpiece(tl, 1, lane)] = __builtin_bit_cast(u32x4, ol);
      } else {
        if (svrow) {      // X_L: what the heads read (the fp32 value of the split pair they are given)
          float *px = svrow + (long)a.L * a.sv_rows * D;
          *reinterpret_cast<f32x4 *>(px) = frag_value(oh, ol, 0);
          *reinterpret_cast<f32x4 *>(px + 16) = frag_value(oh, ol, 1);
        }
        if (a.zimg && r >= G.P && r < G.N) {
          const long zr = a.zrow0 + (long)b * n_t + (r - G.P);
          a.zimg[xpiece(zr >> 4, 0, 16 * g + (int)(zr & 15))] = __builtin_bit_cast(u32x4, oh);
          a.zimg[xpiece(zr >> 4, 1, 16 * g + (int)(zr & 15))] = __builtin_bit_cast(u32x4, ol);
        }
        if (a.zq && r < G.P) {
          const long zr = a.zq_row0 + (long)b * G.P + r;
          a.zq[xpiece(zr >> 4, 0, 16 * g + (int)(zr & 15))] = __builtin_bit_cast(u32x4, oh);
          a.zq[xpiece(zr >> 4, 1, 16 * g + (int)(zr & 15))] = __builtin_bit_cast(u32x4, ol);
        }
        if (hasq) {   // acquisition logits of the candidate rows (model/head.py:27-33)
          const float *hp = reinterpret_cast<const float *>(whd + head_pairs(F) * PAIR_BYTES);   // b1 | w2 | .. | b2
          float plog = 0.f;
          constexpr int GH = (NW > 8 && NH > 4) ? (NH % 4 == 0 ? 4 : 2) : NH;
#pragma unroll
          for (int h0i = 0; h0i < NH; h0i += GH) {
            f32x4 hid[GH];
#pragma unroll
            for (int i = 0; i < GH; ++i) {
              const Frag u = lds_pair(whd, h0i + i, lane);
              hid[i] = z4;
              mfma3(hid[i], u.hi, u.lo, oh, ol);
            }
#pragma unroll
            for (int i = 0; i < GH; ++i) {
              const f32x4 hv = hid[i] * WINV + ld4(hp + 16 * (h0i + i) + 4 * g), wv = ld4(hp + F + 16 * (h0i + i) + 4 * g);
#pragma unroll
              for (int q = 0; q < 4; ++q) plog = fmaf(relu_s(hv[q]), wv[q], plog);
            }
            if (GH < NH) __builtin_amdgcn_sched_barrier(0);
          }
          const float v = group_sum4(plog) + hp[4 * F];
          if (g == 0 && r < G.P) {
            if (a.selp) lgs[e * SEL_PMAX + r] = v;
            else a.logits[(long)b * a.NP + r] = v;
          }
        }
      }
      S3_LAP(5);
    }
    __syncthreads();
    if (tid == 0) queue[1] = 0;
    S3_LAP(6);
    if (!last) {
      for (int piece = wave; piece < (lbytes >> 10); piece += NW) glds16(gimg + (long)(l + 1) * lbytes + piece * 1024 + lane * 16, wl + piece * 1024);
      wait_vmcnt<0>();
      __syncthreads();
      S3_LAP(7);
    }
  }
  range_check_nan(a.emb.range_flag, range_chk);
  // the design selection of the workgroup's episodes, one wave each (behind the last layer's barrier: every logit is in LDS).  Inside the
  // layer loop -- the waves that run out of tiles selecting for the finished episodes -- it cost the tile loop 69 - 98 scalar-register
  // spills and the rollout 6 % (profiles/r04_s3_select_in_kernel.txt); here the loop's registers are dead
  if (a.selp) {
    const int n_valid = min(epw, G.B - (int)blockIdx.x * epw);
    for (int e = wave; e < n_valid; e += NW) select_episode(G, *a.selp, blockIdx.x * epw + e, lgs + e * SEL_PMAX, cmask + e * (SEL_PMAX / 64), lane);
  }
#ifdef S3_STAMPS
  if (a.stamps && blockIdx.x == 0 && lane == 0) {
    stamps.acc[8] = stamps.t_prev - t_begin;
    for (int k = 0; k < S3_NSTAMP; ++k) a.stamps[wave * S3_NSTAMP + k] = stamps.acc[k];
  }
#endif
}

// ---- the C GMM heads (model/head.py:152-186) over the dense-row image of all steps' target rows --------------------
// raw_c[j] = w2_c[j] . relu(W1_c z + b1_c) + b2_c[j]; a wave keeps GT tiles in registers and walks the components,
// whose images take turns in LDS; the raw outputs of the workgroup's 512 rows collect in LDS and one thread per row
// turns them into the mixture parameters (mean, softplus std + std_min, softmax weight: head.py:264-265 with
// dim_y == 1) and the log-likelihood of the row's target value (utils/eval.py:200-207).
constexpr int GT = 4, GROWS = WAVES * GT * 16;
struct GmmArgs {
  const u32x4 *Z; long ntiles, M;
  const unsigned *img; int C; float std_min;      // img: the first GMM head image
  float *mean, *sd, *wgt;                         // [M, C] or null
  const float *value; long value_mod;             // value[row % value_mod] or null
  float *ll;                                      // [M] or null
  unsigned *range_flag;                           // f16 range guard: raised when a log-likelihood is not finite
};
template <int F>
__global__ __launch_bounds__(THREADS) void gmm_kernel(GmmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, tok = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NH = F / 16;
  const int hb = head_bytes(F);
  float *raw = reinterpret_cast<float *>(lds + hb);          // [3 C][GROWS]
  const long tile0 = ((long)blockIdx.x * WAVES + wave) * GT;
  f16x8 zh[GT], zl[GT];
#pragma unroll
  for (int i = 0; i < GT; ++i) {
    const long tl = min(tile0 + i, a.ntiles - 1);
    zh[i] = __builtin_bit_cast(f16x8, a.Z[xpiece(tl, 0, lane)]);
    zl[i] = __builtin_bit_cast(f16x8, a.Z[xpiece(tl, 1, lane)]);
  }
  const float *hp = reinterpret_cast<const float *>(lds + head_pairs(F) * PAIR_BYTES);   // b1 | w2 [3] | b2
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < a.C; ++c) {
    __syncthreads();
    dma_copy(reinterpret_cast<const char *>(a.img) + (long)c * hb, lds, hb, wave, lane);
    wait_vmcnt<0>();
    __syncthreads();
#pragma unroll
    for (int i = 0; i < GT; ++i) {
      f32x4 hid[NH];
#pragma unroll
      for (int t = 0; t < NH; ++t) {
        const Frag u = lds_pair(lds, t, lane);
        hid[t] = z4;
        mfma3(hid[t], u.hi, u.lo, zh[i], zl[i]);
      }
      float p0 = 0.f, p1 = 0.f, p2 = 0.f;
#pragma unroll
      for (int t = 0; t < NH; ++t) {
        const f32x4 hv = hid[t] * WINV + ld4(hp + 16 * t + 4 * g);
        const f32x4 w0 = ld4(hp + F + 16 * t + 4 * g), w1 = ld4(hp + 2 * F + 16 * t + 4 * g), w2 = ld4(hp + 3 * F + 16 * t + 4 * g);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float rl = relu_s(hv[q]);
          p0 = fmaf(rl, w0[q], p0); p1 = fmaf(rl, w1[q], p1); p2 = fmaf(rl, w2[q], p2);
        }
      }
      p0 = group_sum4(p0) + hp[4 * F]; p1 = group_sum4(p1) + hp[4 * F + 1]; p2 = group_sum4(p2) + hp[4 * F + 2];
      if (g == 0) {
        const int rl = (wave * GT + i) * 16 + tok;
        raw[(3 * c) * GROWS + rl] = p0; raw[(3 * c + 1) * GROWS + rl] = p1; raw[(3 * c + 2) * GROWS + rl] = p2;
      }
    }
  }
  __syncthreads();
  const long row = (long)blockIdx.x * GROWS + tid;
  if (row >= a.M) return;
  const float *rr = raw + tid;
  float mxw = -INFINITY;
  for (int c = 0; c < a.C; ++c) mxw = fmaxf(mxw, rr[(3 * c + 2) * GROWS]);
  float sw = 0.f;
  for (int c = 0; c < a.C; ++c) sw += __expf(rr[(3 * c + 2) * GROWS] - mxw);
  const bool want_ll = a.ll && a.value;
  const float v = want_ll ? a.value[row % a.value_mod] : 0.f;
  float mx2 = -INFINITY;
  for (int c = 0; c < a.C; ++c) {
    const float mean = rr[(3 * c) * GROWS], sd = softplus_f(rr[(3 * c + 1) * GROWS]) + a.std_min, w = __expf(rr[(3 * c + 2) * GROWS] - mxw) / sw;
    if (a.mean) a.mean[row * a.C + c] = mean;
    if (a.sd) a.sd[row * a.C + c] = sd;
    if (a.wgt) a.wgt[row * a.C + c] = w;
    const float zz = (v - mean) / sd;
    const float lp = -0.5f * zz * zz - logf(sd) - 0.91893853320467274178f + logf(w);
    raw[(3 * c) * GROWS + tid] = lp;     // (this thread's own column)
    mx2 = fmaxf(mx2, lp);
  }
  if (want_ll) {
    float se = 0.f;
    for (int c = 0; c < a.C; ++c) se += __expf(rr[(3 * c) * GROWS] - mx2);
    const float ll = mx2 + logf(se);
    a.ll[row] = ll;
    if (!(fabsf(ll) <= 3.4e38f)) range_raise(a.range_flag, ALINE_RANGE_ACT);
  }
}

}  // namespace s3
