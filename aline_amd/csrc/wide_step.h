// Wide path, fused step: ONE kernel runs the whole encoder stack (L layers) and the acquisition head of one
// rollout step for d = 256 / 8 heads in bf16 MFMA -- activations never leave the CU between layers.
//
//   workgroup = one episode (N <= 256 token rows = up to 16 tiles of 16), 4 waves = one per SIMD, each wave
//   owns 4 token tiles for the whole step:
//     xb[4][8]   its X^T as bf16 B fragments                      (128 VGPRs)
//     y[16][4]   the 256-feature fp32 output of the running linear (256 accumulator registers)
//   Weights stream through LDS in 32 KB chunks of 32 fragments (global_load_lds, 2 buffers, one barrier per
//   chunk); per layer the stream is  K(4) V(4) Q(4) OUT(4) FFN(F/32), then the head's F/64 chunks.
//
//   set-attention inside the workgroup (model/encoder.py:8-46, 83-126): the keys of an episode are its
//   context rows (+ the target rows for query tokens), at most 64.  Their rows of X are exchanged through LDS
//   (KX, a 4-tile image); K^T = Wk KX^T is produced as the A fragments of S^T = K Q^T, and V = KX Wv^T comes
//   from the SAME weight and X fragments with the MFMA operands swapped, which lands directly in the layout of
//   the A fragments of O^T = V^T P.  K and V are computed for the key rows only.
//   Residuals are folded into the accumulator init; LayerNorm / softmax / biases are fp32.
#pragma once
#include "wide.h"

namespace wide {

constexpr int ST = 256, SNT = 4;                 // threads, token tiles per wave
constexpr int KF_PIECES = H * 4 * 64;            // K fragments [h][kt]          (32 KB)
constexpr int KX_PIECES = 4 * NKS * 64;          // key-row image [kt][ks]       (32 KB), aliased by V^T [nt][s]

struct StepArgs {
  Geo g;
  const u32x4 *XIN;       // tile image over the B*N token rows: this step's assembled input (patched in one row per
                          // episode between steps, not re-assembled)
  u32x4 *X0;              // scratch tile image: each layer's output (the residual of the next layer's attention block)
  const unsigned *img;    // packed weights: L layer images, then the head image (pack_kernel)
  int L, F;
  float *logits;          // [B*N] acquisition logits
  float *zt;              // [B, n_t, 256] fp32 encodings of the target rows (last layer), or null
  u32x4 *zimg; long zrow0;   // tile image of the same rows in bf16 (row zrow0 + b * n_t + j), or null: the input of the
                          // post-loop GMM head kernels
  unsigned long long *stamps;   // diagnostic instantiation only: per-phase s_memtime sums [4 waves x 16]
};

// Diagnostic stamps (STAMP = true instantiation only, ALINE_WIDE_STAMPS=1): wave w of workgroup 0 accumulates
// s_memtime deltas per phase into a.stamps[w * 16 + phase]
#define WSTAMP(ph)                                                                \
  if constexpr (STAMP) {                                                          \
    unsigned long long _t;                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");    \
    if (blockIdx.x == 0 && lane == 0) a.stamps[wave * 16 + (ph)] += _t - t_prev;  \
    t_prev = _t;                                                                  \
  }

__host__ __device__ inline int step_params(int F) { return layer_params(F); }
__host__ __device__ inline size_t step_lds_bytes(int F) {
  return (size_t)(2 * CHUNK_W) * 4 + (size_t)(KF_PIECES + KX_PIECES) * 16 + (size_t)2 * step_params(F) * 4 + 256 * 2 + 64;
}

__device__ __forceinline__ void glds16(const void *gsrc, void *ldst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                   (__attribute__((address_space(3))) void *)ldst, 16, 0, 0);
}
// 32 KB chunk -> LDS buffer; every wave-instruction moves one contiguous KB
// One 1 KB-per-wave piece (of 8) of a 32 KB chunk: uniform base + one 32-bit lane offset (saddr form); the LDS
// destination is the wave-uniform base the hardware adds lane * 16 to.
__device__ __forceinline__ void issue_piece(const char *src, char *dst, int piece, unsigned lane_off, unsigned wave_off) {
#ifdef WIDE_NO_GLDS   // (timing experiment only: the weights never arrive)
  return;
#endif
  glds16(src + piece * (ST * 16) + lane_off, dst + piece * (ST * 16) + wave_off);
}
__device__ __forceinline__ void issue_floats(const float *src, float *dst, int n, unsigned lane_off, unsigned wave_off) {   // n % 4 == 0
  for (int i0 = 0; i0 < n / 4; i0 += ST)
    if (i0 + (int)(lane_off >> 4) < n / 4)
      glds16(reinterpret_cast<const char *>(src) + (size_t)i0 * 16 + lane_off, reinterpret_cast<char *>(dst) + (size_t)i0 * 16 + wave_off);
}

__device__ __forceinline__ f32x4 ldsf4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  return (f32x4){relu_nn(v[0]), relu_nn(v[1]), relu_nn(v[2]), relu_nn(v[3])};
}
// ReLU on packed bf16: as 16-bit integers the negative values (sign bit set, -0 included) are negative -> max(x, 0)
typedef __attribute__((ext_vector_type(8))) short s16x8;
__device__ __forceinline__ bf16x8 relu_frag(bf16x8 f) {
  const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  return __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, f), z));
}
__device__ __forceinline__ float group_max4(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// The weight stream, one 4 KB piece per step, 8 steps per chunk: step k stores piece k of chunk q+1 (read from
// global memory four steps ago into staging registers) into the LDS buffer chunk q-1 vacated, and refills the
// staging slot with the piece four steps further down the stream (chunk q+1 pieces 4..7, then chunk q+2 pieces
// 0..3).  Plain loads + ds_write: an LDS-DMA (global_load_lds) costs this kernel 190-360 issue cycles per piece
// among the ds_reads and MFMAs of a chunk, these two instructions a few tens.
#ifndef WIDE_STREAM
#define WIDE_STREAM 2
#endif
#if WIDE_STREAM == 1
#define STAGE_STEP(K)                                                                                       \
  {                                                                                                         \
    *reinterpret_cast<u32x4 *>(ndst + (K) * (ST * 16) + lane_off) = stg[(K) & 3];                           \
    stg[(K) & 3] = *reinterpret_cast<const u32x4 *>(((K) < 4 ? nsrc1 : nsrc2) + (((K) + 4) & 7) * (ST * 16) + lane_off); \
  }
#define STAGE_GROUPS __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#elif WIDE_STREAM == 0
#define STAGE_STEP(K) issue_piece(nsrc1, ndst, (K), lane_off, wave_off);
#define STAGE_GROUPS __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
#else   // two LDS-DMA pieces per step in the first four steps of a chunk: the copy gets half a chunk of head start
#define STAGE_STEP(K) if ((K) < 4) { issue_piece(nsrc1, ndst, 2 * (K), lane_off, wave_off); issue_piece(nsrc1, ndst, 2 * (K) + 1, lane_off, wave_off); }
#define STAGE_GROUPS __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
#endif

// Fragments f = 0..NF-1 of the current chunk (LDS index IDX(f)), NM MFMAs each, in batches of 4: the reads of
// batch k+1 are issued as one burst right after the FIRST fragment of batch k, so the s_waitcnt lgkmcnt(0) the
// compiler puts in front of batch k+1 (with an LDS-DMA pending it never counts) finds them complete three
// fragments later (one wave per SIMD: nothing else hides LDS latency).
// Each batch also runs ONE step (P0 + k) of the weight stream.  The sched_group_barriers pin the issue order,
// the sched_barriers fence the region.
#define FRAG_PIPE(NF, NM, IDX, P0, BODY)                                 \
  {                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                   \
    bf16x8 ring_[2][4];                                                  \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) ring_[0][j_] = fr[(IDX(j_)) * 64]; \
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                   \
    _Pragma("unroll") for (int k_ = 0; k_ < NF / 4; ++k_) {              \
      { const int f = 4 * k_; const bf16x8 A = ring_[k_ & 1][0]; BODY }  \
      __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);                \
      if (k_ + 1 < NF / 4) {                                             \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) ring_[(k_ + 1) & 1][j_] = fr[(IDX(4 * (k_ + 1) + j_)) * 64]; \
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);               \
      }                                                                  \
      STAGE_STEP((P0) + k_)                                              \
      STAGE_GROUPS                                                       \
      _Pragma("unroll") for (int j_ = 1; j_ < 4; ++j_) { const int f = 4 * k_ + j_; const bf16x8 A = ring_[k_ & 1][j_]; BODY } \
      __builtin_amdgcn_sched_group_barrier(0x008, 3 * NM, 0);            \
    }                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                   \
  }

// The 256 accumulators of y exactly fill the AGPR file.  Left to itself the register allocator copies all of
// them to VGPRs (and from there to scratch) at the exit of an MFMA loop; these opaque moves keep every element in
// the accumulator file until the instruction that needs it.  (Opaque to the hazard recogniser as well: callers
// put mfma_drain() between the last MFMA and the first acc_rd.)
__device__ __forceinline__ float acc_rd(float a) { float v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a)); return v; }
__device__ __forceinline__ float acc_wr(float v) { float a; asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v)); return a; }
__device__ __forceinline__ f32x4 acc_rd4(const f32x4 &a) { return (f32x4){acc_rd(a[0]), acc_rd(a[1]), acc_rd(a[2]), acc_rd(a[3])}; }
__device__ __forceinline__ f32x4 acc_wr4(const f32x4 &v) { return (f32x4){acc_wr(v[0]), acc_wr(v[1]), acc_wr(v[2]), acc_wr(v[3])}; }
// y owns the whole accumulator file: the small accumulators of the other MFMAs (scores, P V, hidden units) are
// pinned to VGPRs, or the allocator evicts parts of y to scratch to make room for them in AGPRs
#define IN_VGPR(x) asm volatile("" : "+v"(x))
// While an LDS-DMA is pending the compiler turns every LDS wait into lgkmcnt(0) (one full LDS round trip per
// ds_read).  Outside the chunk pipelines, first let the pending pieces of the next chunk land (vmcnt(0); they
// were issued half a chunk or more ago), after which LDS reads are counted and pipelined normally again.
__device__ __forceinline__ void stream_landed() { __builtin_amdgcn_s_waitcnt(0x0F70); }
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory"); }

// One token tile of  out = bf16(LayerNorm(y + bias + residual)):  the tile's 64 accumulators leave the AGPR file
// exactly here, bias and residual (bf16 fragments res[ks]) are added in fp32, the result comes back as the B
// fragments of the next linear.  With ZT the fp32 row also goes to zrow (target tokens of the last layer).
template <bool ZT, bool RES>
__device__ __forceinline__ void ln_tile(const f32x4 (&yt)[NMT], const bf16x8 (&res)[NKS], const float *bias, const float *lw,
                                        const float *lb, int g, float *zrow, bf16x8 (&out)[NKS]) {
  f32x4 v[NMT];
  float s = 0.f, q = 0.f;                             // one pass: sum and sum of squares (fp32, 256 values of O(1..10))
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) {
    v[mt] = acc_rd4(yt[mt]);
    if (RES) {                                        // (otherwise bias and residual were the accumulators' initial value)
      const f32x4 bv = ldsf4(bias + 16 * mt + 4 * g);
      const u32x4 xr = __builtin_bit_cast(u32x4, res[mt >> 1]);
      const unsigned w0 = xr[2 * (mt & 1)], w1 = xr[2 * (mt & 1) + 1];
      v[mt] += (f32x4){bv[0] + bf_lo(w0), bv[1] + bf_hi(w0), bv[2] + bf_lo(w1), bv[3] + bf_hi(w1)};
    }
    s += (v[mt][0] + v[mt][1]) + (v[mt][2] + v[mt][3]);
#pragma unroll
    for (int r = 0; r < 4; ++r) q = fmaf(v[mt][r], v[mt][r], q);
  }
  const float mean = group_sum4(s) * (1.f / D);
  const float var = fmaxf(group_sum4(q) * (1.f / D) - mean * mean, 0.f);
  const float rstd = rsqrtf(var + 1e-5f), nmr = -mean * rstd;
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    f32x4 o[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int mt = 2 * ks + hf;
      const f32x4 wv = ldsf4(lw + 16 * mt + 4 * g), bv = ldsf4(lb + 16 * mt + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[hf][r] = fmaf(fmaf(v[mt][r], rstd, nmr), wv[r], bv[r]);
      if (ZT && zrow) *reinterpret_cast<f32x4 *>(zrow + 16 * mt + 4 * g) = o[hf];
    }
    out[ks] = acc_to_frag(o[0], o[1]);
  }
}

// Attention of this wave's 4 token tiles against NKT key tiles, head by head: S^T = K Q^T (scores arrive in the
// exp2 domain, the mask is the accumulator init), softmax over the keys of a token (16 NKT values per lane + one
// cross-group reduction), O^T = V^T P.  qa[h][ct] goes in as the Q^T fragment and comes out as the fragment of the
// normalised head output (k-step h of the OUT projection).
template <int NKT>
__device__ __forceinline__ void attention_tiles(bf16x8 (&qa)[H][SNT], const bf16x8 *kf, const bf16x8 *vf, const int (&nv)[SNT]) {
  // the additive key mask of a token tile (0 / -inf per (key tile, r)) is the same for all heads: with 1 or 2 key
  // tiles it is kept in registers for the whole attention block (8 VGPRs per key tile and token tile) and is the
  // C operand of the score MFMA; with 4 key tiles it is rebuilt per head (registers)
  constexpr bool KEEP = NKT <= 2;
  f32x4 mb[KEEP ? SNT : 1][NKT];
  if (KEEP) {
#pragma unroll
    for (int ct = 0; ct < SNT; ++ct)
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mb[ct][kt][r] = (16 * kt + r) < nv[ct] ? 0.f : -INFINITY;
  }
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int ct = 0; ct < SNT; ++ct) {
      f32x4 s[NKT];
      if (!KEEP) {
        int lim = nv[ct];
        asm volatile("" : "+v"(lim));                 // keep the mask values out of the loop-invariant set
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mb[0][kt][r] = (16 * kt + r) < lim ? 0.f : -INFINITY;
      }
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[(h * 4 + kt) * 64], qa[h][ct], mb[KEEP ? ct : 0][kt], 0, 0, 0);
        IN_VGPR(s[kt]);
      }
      float mx = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
      for (int kt = 1; kt < NKT; ++kt) mx = fmaxf(mx, fmaxf(fmaxf(s[kt][0], s[kt][1]), fmaxf(s[kt][2], s[kt][3])));
      mx = group_max4(mx);
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[kt][r] = __builtin_amdgcn_exp2f(s[kt][r] - mx);
          sum += s[kt][r];
        }
      const float inv = __builtin_amdgcn_rcpf(group_sum4(sum));
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      const bf16x8 p0 = acc_to_frag(s[0], NKT > 1 ? s[NKT > 1 ? 1 : 0] : z4);
      f32x4 o0 = z4, o1 = z4;
      IN_VGPR(o0); IN_VGPR(o1);
      WMFMA(o0, vf[((2 * h) * 2) * 64], p0);
      WMFMA(o1, vf[((2 * h + 1) * 2) * 64], p0);
      IN_VGPR(o0); IN_VGPR(o1);
      if (NKT > 2) {
        const bf16x8 p1 = acc_to_frag(s[NKT > 2 ? 2 : 0], s[NKT > 2 ? 3 : 0]);
        WMFMA(o0, vf[((2 * h) * 2 + 1) * 64], p1);
        WMFMA(o1, vf[((2 * h + 1) * 2 + 1) * 64], p1);
        IN_VGPR(o0); IN_VGPR(o1);
      }
      qa[h][ct] = acc_to_frag(o0 * inv, o1 * inv);
      IN_VGPR(qa[h][ct]);                             // materialise here (left alone, the normalisation sinks into the OUT chunks)
      __builtin_amdgcn_sched_barrier(0);              // no interleaving across iterations: it only costs registers
    }
}

// Address of the first piece of token row `row` in a tile image, formed at the point of use from a 32-bit piece
// index the optimiser cannot see through (hoisted out of the layer loop, the 64-bit addresses of all tiles and
// k-steps end up in scratch, and every reload serialises the loads behind it).
template <typename T>
__device__ __forceinline__ T *tile_row(T *img, long row, int g) {
  unsigned pc = (unsigned)piece(row, 0, g);
  asm volatile("" : "+v"(pc));
  return img + pc;
}

template <bool STAMP>
__global__ __launch_bounds__(ST) void wide_step_kernel(StepArgs a) {
  unsigned long long t_prev = 0;
  if constexpr (STAMP) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory"); }
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  unsigned *const wbuf0 = lds, *const wbuf1 = lds + CHUNK_W;
  u32x4 *const Kf = reinterpret_cast<u32x4 *>(lds + 2 * CHUNK_W);
  u32x4 *const KX = Kf + KF_PIECES;                 // V^T fragments alias the key-row image
  const int np = step_params(a.F);
  float *const pb0 = reinterpret_cast<float *>(KX + KX_PIECES), *const pb1 = pb0 + np;
  short *const keyidx = reinterpret_cast<short *>(pb1 + np);
  int *const misc = reinterpret_cast<int *>(keyidx + 256);

  const Geo &G = a.g;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, tok = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lane_off = (unsigned)tid * 16u, wave_off = (unsigned)wave * 1024u;
  const long ep = (long)b * G.N;
  const int F = a.F, n_t = G.n_td + G.n_th;
  const long lw = layer_words(F);
  const long nfw = (long)layer_chunks(F) * CHUNK_W;
  const int nlc = 16 + F / 32, nq = a.L * nlc + F / 64;
  const int row0 = 64 * wave + tok;                  // this lane's token row in tile ct: row0 + 16 ct

  // ---- the weight stream -----------------------------------------------------------------------------
  int q = 0;                                         // next chunk of the sequence to be consumed
  auto chunk_src = [&](int qq) -> const char * {
    const unsigned *p;
    if (qq < a.L * nlc) {
      const int l = qq / nlc, i = qq % nlc;
      const int c = i < 8 ? i + 4 : i < 12 ? i - 8 : i;          // stream order K V Q OUT FFN over the Q K V image
      p = a.img + l * lw + (long)c * CHUNK_W;
    } else {
      p = a.img + a.L * lw + (long)(qq - a.L * nlc) * CHUNK_W;
    }
    return reinterpret_cast<const char *>(p);
  };
  auto params_src = [&](int l) -> const float * {
    return reinterpret_cast<const float *>(l < a.L ? a.img + l * lw + nfw : a.img + a.L * lw + (long)head_chunks(F) * CHUNK_W);
  };
  // Barrier: chunk q is complete in LDS and every wave is done with chunk q-1, whose buffer (ndst) the caller
  // refills with chunk q+1 in 8 STAGE_STEPs WHILE it computes on chunk q (past the end: chunk nq-1 again).
  const char *nsrc1, *nsrc2;
  char *ndst;
  u32x4 stg[4];
  auto begin_chunk = [&]() -> const bf16x8 * {
    if (a.stamps && a.stamps[63]) { WSTAMP(14) }     // (diagnostic split: compute | barrier wait)
#ifndef WIDE_NO_BARRIER   // (timing experiment only)
    __syncthreads();
#endif
    if (a.stamps && a.stamps[63]) { WSTAMP(12) }
    nsrc1 = chunk_src(min(q + 1, nq - 1));
    nsrc2 = chunk_src(min(q + 2, nq - 1));   // (register-staged stream variant only)
    (void)nsrc2;
    ndst = reinterpret_cast<char *>(((q + 1) & 1) ? wbuf1 : wbuf0);
    const bf16x8 *fr = reinterpret_cast<const bf16x8 *>((q & 1) ? wbuf1 : wbuf0) + lane;
    ++q;
    return fr;
  };

  {
    const char *s0 = chunk_src(0), *s1 = chunk_src(1);
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_piece(s0, reinterpret_cast<char *>(wbuf0), i, lane_off, wave_off);
#if WIDE_STREAM == 1
#pragma unroll
    for (int i = 0; i < 4; ++i) stg[i] = *reinterpret_cast<const u32x4 *>(s1 + i * (ST * 16) + lane_off);
#else
    (void)s1; (void)stg;
#endif
  }
  issue_floats(params_src(0), pb0, np, lane_off, wave_off);

  // ---- episode geometry: key list -----------------------------------------------------------------------------
  int n_ck, n_ak;
  {
    const int row = tid;
    const bool ctx = row < G.P && is_ctx(G, b, row);
    const unsigned long long bal = __ballot(ctx);
    if (lane == 0) misc[wave] = __popcll(bal);
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += misc[w];
    n_ck = misc[0] + misc[1] + misc[2] + misc[3];
    int nvis = 0, rank = 0;
    for (int i = 0; i < n_t; ++i) {
      const bool vis = !G.tmask || G.tmask[i];
      if (i < row - G.P) rank += vis;
      nvis += vis;
    }
    n_ak = n_ck + nvis;
    int j = -1;
    if (ctx) j = off + __popcll(bal & ((1ull << lane) - 1ull));
    else if (row >= G.P && row < G.N && (!G.tmask || G.tmask[row - G.P])) j = n_ck + rank;
    if (j >= WNK) j = -1;
    keyidx[row] = (short)j;
    n_ck = __builtin_amdgcn_readfirstlane(min(n_ck, WNK));
    n_ak = __builtin_amdgcn_readfirstlane(min(n_ak, WNK));
  }

  bf16x8 xb[SNT][NKS];
#pragma unroll
  for (int ct = 0; ct < SNT; ++ct) {
    const u32x4 *xr = tile_row(a.XIN, ep + min(row0 + 16 * ct, G.N - 1), g);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) xb[ct][ks] = __builtin_bit_cast(bf16x8, xr[ks * 64]);
  }
  __syncthreads();                                    // keyidx visible

  f32x4 y[NMT][SNT];
  WSTAMP(0)   // setup: key list, X0 load

#pragma unroll 1
  for (int l = 0; l < a.L; ++l) {
    const float *prm = (l & 1) ? pb1 : pb0;           // bq bk bv | bo | b1 | b2 | ln1w ln1b ln2w ln2b

    // ---- key rows of X -> LDS (zero rows pad the last key tile) ----------------------------------------------
    // (the key-row image shares LDS with the X1 copies of the previous layer's FFN, which slower waves may
    // still be reading in their LN2)
    if (l > 0) __syncthreads();
#pragma unroll
    for (int ct = 0; ct < SNT; ++ct) {
      const int r = row0 + 16 * ct;
      const int j = r < G.N ? keyidx[r] : -1;
      if (j >= 0) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) KX[((j >> 4) * NKS + ks) * 64 + g * 16 + (j & 15)] = __builtin_bit_cast(u32x4, xb[ct][ks]);
      }
    }
    for (int i = tid; i < (WNK - n_ak) * (D / 8); i += ST) {
      const int j = n_ak + i / (D / 8), pc = i % (D / 8);
      KX[((j >> 4) * NKS + (pc >> 2)) * 64 + (pc & 3) * 16 + (j & 15)] = (u32x4){0u, 0u, 0u, 0u};
    }

    WSTAMP(1)   // key rows -> LDS
    // ---- K^T = Wk KX^T for this wave's 64 features (heads 2w, 2w+1), all key tiles ---------------------------
    {
      f32x4 kacc[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) kacc[i][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int cc = 0; cc < 4; ++cc) {
        const bf16x8 *fr = begin_chunk();
        if (cc == 0) issue_floats(params_src(l + 1), (l & 1) ? pb0 : pb1, l + 1 < a.L ? np : 2 * F + 4, lane_off, wave_off);
        const bf16x8 *kx = reinterpret_cast<const bf16x8 *>(KX) + lane;
#pragma unroll
        for (int kl = 0; kl < 2; ++kl) {
          const int ks = 2 * cc + kl;
          bf16x8 bx[4];
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) bx[kt] = kx[(kt * NKS + ks) * 64];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bf16x8 A = fr[(kl * 16 + 4 * wave + i) * 64];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) WMFMA(kacc[i][kt], A, bx[kt]);
            STAGE_STEP(4 * kl + i)
          }
        }
      }
#pragma unroll
      for (int hl = 0; hl < 2; ++hl) {
        const f32x4 b0 = ldsf4(prm + D + 64 * wave + 32 * hl + 4 * g), b1 = ldsf4(prm + D + 64 * wave + 32 * hl + 16 + 4 * g);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
          Kf[((2 * wave + hl) * 4 + kt) * 64 + lane] = __builtin_bit_cast(u32x4, acc_to_frag(kacc[2 * hl][kt] + b0, kacc[2 * hl + 1][kt] + b1));
      }
    }
    WSTAMP(2)   // K projection
    // ---- V = KX Wv^T (operands swapped: rows = keys), this wave's 64 features --------------------------------
    {
      f32x4 vacc[4][4];                               // [kt][feature tile]
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int i = 0; i < 4; ++i) vacc[kt][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int cc = 0; cc < 4; ++cc) {
        const bf16x8 *fr = begin_chunk();
        const bf16x8 *kx = reinterpret_cast<const bf16x8 *>(KX) + lane;
#pragma unroll
        for (int kl = 0; kl < 2; ++kl) {
          const int ks = 2 * cc + kl;
          bf16x8 ax[4];
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) ax[kt] = kx[(kt * NKS + ks) * 64];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bf16x8 Bw = fr[(kl * 16 + 4 * wave + i) * 64];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) WMFMA(vacc[kt][i], ax[kt], Bw);
            STAGE_STEP(4 * kl + i)
          }
        }
      }
      __syncthreads();                                // every wave is done reading KX: V^T may overwrite it
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float bv = prm[2 * D + 64 * wave + 16 * i + tok];
        const f32x4 b4 = {bv, bv, bv, bv};
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          KX[((4 * wave + i) * 2 + s2) * 64 + lane] = __builtin_bit_cast(u32x4, acc_to_frag(vacc[2 * s2][i] + b4, vacc[2 * s2 + 1][i] + b4));
      }
    }

    WSTAMP(3)   // V projection
    // ---- Q^T = Wq X^T (pre-scaled by log2(e)/sqrt(hd)) -------------------------------------------------------
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int ct = 0; ct < SNT; ++ct) y[mt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define IDX_LIN(f) (f)
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const bf16x8 *fr = begin_chunk();
      FRAG_PIPE(32, SNT, IDX_LIN, 0, { _Pragma("unroll") for (int ct = 0; ct < SNT; ++ct) WMFMA(y[f & 15][ct], A, xb[ct][2 * cc + (f >> 4)]); })
    }
    WSTAMP(4)   // Q projection
    // Q^T fragments per head (later overwritten by the attention output of that head).  X is dead from here to
    // LN1, which re-reads it (the residual) from the global tile image: the VGPRs hold Q / A fragments only.
    bf16x8 qa[H][SNT];
    stream_landed();
    mfma_drain();
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const f32x4 b0 = ldsf4(prm + 32 * h + 4 * g), b1 = ldsf4(prm + 32 * h + 16 + 4 * g);
#pragma unroll
      for (int ct = 0; ct < SNT; ++ct) qa[h][ct] = acc_to_frag(acc_rd4(y[2 * h][ct]) + b0, acc_rd4(y[2 * h + 1][ct]) + b1);
    }

    WSTAMP(5)   // Q fragments + residual init
    // ---- attention, one (head, token tile) at a time (K / V^T fragments from LDS; the Q chunks' barriers
    // published them).  Context and target rows see the n_ck context keys, query rows all n_ak keys.
    {
      const bf16x8 *kf = reinterpret_cast<const bf16x8 *>(Kf) + lane, *vf = reinterpret_cast<const bf16x8 *>(KX) + lane;
      int nv[SNT];
#pragma unroll
      for (int ct = 0; ct < SNT; ++ct) {
        const int r = min(row0 + 16 * ct, G.N - 1);
        nv[ct] = ((r < G.P && keyidx[r] < 0) ? n_ak : n_ck) - 4 * g;   // key 16 kt + 4 g + r is visible iff 16 kt + r < nv
      }
      if (n_ak <= 16) attention_tiles<1>(qa, kf, vf, nv);          // (wave-uniform: the softmax VALU work scales with
      else if (n_ak <= 32) attention_tiles<2>(qa, kf, vf, nv);     //  the key tiles; 1 or 2 cover rollouts up to T = 30)
      else attention_tiles<4>(qa, kf, vf, nv);
    }

    WSTAMP(6)   // attention
    // ---- X1 = LN1(X + bo + Wo A) ------------------------------------------------------------------------------------
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int ct = 0; ct < SNT; ++ct) y[mt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const bf16x8 *fr = begin_chunk();
      FRAG_PIPE(32, SNT, IDX_LIN, 0, { _Pragma("unroll") for (int ct = 0; ct < SNT; ++ct) WMFMA(y[f & 15][ct], A, qa[2 * cc + (f >> 4)][ct]); })
    }
    WSTAMP(7)   // OUT projection
    stream_landed();
    mfma_drain();
    {
      bf16x8 res[2][NKS];                             // the residual rows of the next tile are in flight during a tile's LN
      {
        const u32x4 *xr = tile_row(l > 0 ? a.X0 : a.XIN, ep + min(row0, G.N - 1), g);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) res[0][ks] = __builtin_bit_cast(bf16x8, xr[ks * 64]);
      }
#pragma unroll
      for (int ct = 0; ct < SNT; ++ct) {
        if (ct + 1 < SNT) {
          const u32x4 *xr = tile_row(l > 0 ? a.X0 : a.XIN, ep + min(row0 + 16 * (ct + 1), G.N - 1), g);
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) res[(ct + 1) & 1][ks] = __builtin_bit_cast(bf16x8, xr[ks * 64]);
        }
        f32x4 yt[NMT];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) yt[mt] = y[mt][ct];
        int po = 0;                                   // (opaque per tile: shared parameter loads would stay live across all tiles)
        asm volatile("" : "+v"(po));
        const float *pt = prm + po;
        ln_tile<false, true>(yt, res[ct & 1], pt + 3 * D, pt + 4 * D + F + D, pt + 4 * D + F + 2 * D, g, nullptr, xb[ct]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    WSTAMP(8)   // LN1

    // ---- X = LN2(X1 + W2 relu(W1 X1 + b1) + b2): 32 hidden units per chunk -----------------------------------------
    // Register budget: the 256 accumulators of y fill the AGPR file, so everything else shares 256 VGPRs.  The
    // X1 fragments of token tiles 2 and 3 (64 VGPRs) therefore move to the LDS region the K / V^T fragments no
    // longer need (wave-private 16 KB) and are re-read, two k-steps per batch, next to the W1 fragments.
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int ct = 0; ct < SNT; ++ct) y[mt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      u32x4 *xl = Kf + wave * (2 * NKS * 64) + lane;      // [tile 2..3][ks][lane]
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) xl[(c2 * NKS + ks) * 64] = __builtin_bit_cast(u32x4, xb[2 + c2][ks]);
    }
#pragma unroll 1
    for (int c = 0; c < F / 32; ++c) {
      const bf16x8 *fr = begin_chunk();
      const bf16x8 *xl = reinterpret_cast<const bf16x8 *>(Kf + wave * (2 * NKS * 64)) + lane;
      f32x4 hh[2][SNT];
      bf16x8 hb[SNT];
      const f32x4 hb0 = ldsf4(prm + 4 * D + 32 * c + 4 * g), hb1 = ldsf4(prm + 4 * D + 32 * c + 16 + 4 * g);
#pragma unroll
      for (int ct = 0; ct < SNT; ++ct) { hh[0][ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; hh[1][ct] = hh[0][ct]; }
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 rw[2][4], rx[2][4];                         // W fragments of a batch; X1 fragments [ks local][tile 2..3]
#pragma unroll
      for (int j = 0; j < 4; ++j) rw[0][j] = fr[j * 64];
#pragma unroll
      for (int j = 0; j < 4; ++j) rx[0][j] = xl[((j & 1) * NKS + (j >> 1)) * 64];
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {               // half 0: fragment 0; half 1: burst + stream step, fragments 1..3
          if (half == 1) {
            if (k + 1 < 8) {
#pragma unroll
              for (int j = 0; j < 4; ++j) rw[(k + 1) & 1][j] = fr[(4 * (k + 1) + j) * 64];
              if (k + 1 < 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) rx[(k + 1) & 1][j] = xl[((j & 1) * NKS + 2 * (k + 1) + (j >> 1)) * 64];
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
              } else {
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
              }
            }
            STAGE_STEP(k)
            STAGE_GROUPS
          }
#pragma unroll
          for (int j = half ? 1 : 0; j < (half ? 4 : 1); ++j) {
            const int f = 4 * k + j;
            const bf16x8 A = rw[k & 1][j];
            if (f < 16) {                                // W1 fragment: k-step f >> 1 (local k-step j >> 1), half f & 1
              WMFMA(hh[f & 1][0], A, xb[0][f >> 1]);
              WMFMA(hh[f & 1][1], A, xb[1][f >> 1]);
              WMFMA(hh[f & 1][2], A, rx[k & 1][2 * (j >> 1)]);
              WMFMA(hh[f & 1][3], A, rx[k & 1][2 * (j >> 1) + 1]);
            } else {
              if (f == 16) {
#pragma unroll
#ifdef WIDE_NO_PACK   // (timing experiment only)
                for (int ct = 0; ct < SNT; ++ct) hb[ct] = xb[0][ct];
#else
                for (int ct = 0; ct < SNT; ++ct) hb[ct] = relu_frag(acc_to_frag(hh[0][ct] + hb0, hh[1][ct] + hb1));
#endif
              }
#pragma unroll
              for (int ct = 0; ct < SNT; ++ct) WMFMA(y[f - 16][ct], A, hb[ct]);
            }
          }
          if (half) __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
          else __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    WSTAMP(9)   // FFN
    stream_landed();
    mfma_drain();
    {
      const bf16x8 *xl = reinterpret_cast<const bf16x8 *>(Kf + wave * (2 * NKS * 64)) + lane;
      const bool last = l == a.L - 1;
#pragma unroll
      for (int ct = 0; ct < SNT; ++ct) {
        bf16x8 res[NKS];                              // X1 of this tile: registers (tiles 0, 1) or the LDS copy (2, 3)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) res[ks] = ct < 2 ? xb[ct][ks] : xl[((ct - 2) * NKS + ks) * 64];
        f32x4 yt[NMT];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) yt[mt] = y[mt][ct];
        const int r = row0 + 16 * ct;
        float *zrow = (last && a.zt && r >= G.P && r < G.N) ? a.zt + ((long)b * n_t + (r - G.P)) * D : nullptr;
        int po = 0;
        asm volatile("" : "+v"(po));
        const float *pt = prm + po;
        ln_tile<true, true>(yt, res, pt + 4 * D + F, pt + 4 * D + F + 3 * D, pt + 4 * D + F + 4 * D, g, zrow, xb[ct]);
        if (last && a.zimg && r >= G.P && r < G.N) {
          u32x4 *zo = tile_row(a.zimg, a.zrow0 + (long)b * n_t + (r - G.P), g);
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) zo[ks * 64] = __builtin_bit_cast(u32x4, xb[ct][ks]);
        }
        if (!last && r < G.N) {                       // next layer's residual
          u32x4 *xo = tile_row(a.X0, ep + r, g);
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) xo[ks * 64] = __builtin_bit_cast(u32x4, xb[ct][ks]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    WSTAMP(10)  // LN2
  }

  // ---- acquisition head: logits = w2 . relu(W1a z + b1a) + b2a  (model/head.py:280-310) --------------------------------
  {
    const float *prm = (a.L & 1) ? pb1 : pb0;         // b1a | w2a | b2a
    float plog[SNT] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int c = 0; c < F / 64; ++c) {
      const bf16x8 *fr = begin_chunk();
#pragma unroll
      for (int grp = 0; grp < 2; ++grp) {
        const int hbase = 64 * c + 32 * grp;
        f32x4 hh[2][SNT];
        const f32x4 b0 = ldsf4(prm + hbase + 4 * g), b1 = ldsf4(prm + hbase + 16 + 4 * g);
#pragma unroll
        for (int ct = 0; ct < SNT; ++ct) { hh[0][ct] = b0; hh[1][ct] = b1; }
#define IDX_GRP(f) (grp * 16 + (f))
        FRAG_PIPE(16, SNT, IDX_GRP, 4 * grp, { _Pragma("unroll") for (int ct = 0; ct < SNT; ++ct) WMFMA(hh[f & 1][ct], A, xb[ct][f >> 1]); })
        const f32x4 w0 = ldsf4(prm + F + hbase + 4 * g), w1 = ldsf4(prm + F + hbase + 16 + 4 * g);
#pragma unroll
        for (int ct = 0; ct < SNT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            plog[ct] = fmaf(relu_nn(hh[0][ct][r]), w0[r], plog[ct]);
            plog[ct] = fmaf(relu_nn(hh[1][ct][r]), w1[r], plog[ct]);
          }
      }
    }
#pragma unroll
    for (int ct = 0; ct < SNT; ++ct) {
      const float v = group_sum4(plog[ct]) + prm[2 * F];
      const int r = row0 + 16 * ct;
      if (g == 0 && r < G.N) a.logits[ep + r] = v;
    }
  }
  WSTAMP(11)  // acquisition head
}

}  // namespace wide
