// Token-wise linear layers:  Y[m, n] = act( sum_k X[m, k] * W[n, k] + bias[n] (+ t * wcol[n]) )
//
// "NT" GEMM on the matrix cores: both operands are K-contiguous, which is exactly PyTorch's
// nn.Linear layout ([out, in] row-major), so weights are consumed where the optimiser keeps them.
// One workgroup = 4 waves (256 threads) computes a BM x BN tile with 16x16 MFMA fragments;
// K is walked in steps of 32 through LDS.  Three arithmetic policies (include/aline_hip.h):
//   F32    v_mfma_f32_16x16x4_f32   (exact fp32, 8 MFMAs per 32-deep step)
//   BF16   v_mfma_f32_16x16x32_bf16 (1 MFMA per step)
//   BF16X3 split operands hi+lo, 3 MFMAs per step (hi*hi + hi*lo + lo*hi), ~2^-16 relative error
//   F16X3  the same split in f16 (v_mfma_f32_16x16x32_f16): hi and lo carry 11 bits each, the dropped lo*lo term is
//          <= 2^-24 of the product -- fp32-grade results (reference fixtures: NLL within 2e-5, like F32) at 3 passes
//          of the 2.5 PFLOP/s pipe instead of 8 of the 157 TFLOP/s fp32 MFMA.  Weights are scaled by 2^8 on their way
//          into LDS (exact) so that both halves stay in f16's normal range; the epilogue scales back.
// Row maps let a launch read/write a sub-range of each episode's token rows (e.g. only the
// target rows) without a gather pass:  row(m) = (m / R) * G + off + (m % R).
// blockIdx.z selects a weight group (the C independent GMM heads run as one grouped launch).
#pragma once
#include "common.h"
#include <cstdint>

#define GEMM_MAX_GROUPS 16

struct GemmArgs {
  const float *X; int ldx; int R_in, G_in, off_in;
  const int *row_index;            // optional gather: GEMM row m reads X row row_index[m] (< 0: a zero row)
  const int *out_index;            // optional scatter: GEMM row m is stored to Y row out_index[m] (< 0: not stored); the rows must be distinct
  const float *W[GEMM_MAX_GROUPS]; const float *bias[GEMM_MAX_GROUPS]; int ldw;
  float *Y; int ldy; int R_out, G_out, off_out; int col_per_group;
  int M, N, K; int relu; int accum;   // accum: Y += result
  const float *tscalar; const float *tcol; int tcol_stride;  // optional rank-1 term (time token): + t * tcol[n * tcol_stride]
  const float *tvec; int tvec_div;                           // ... with a per-row t = tvec[m / tvec_div] instead of the scalar (backward: instances of several steps)
  const float *mask; int ldmask;   // optional ReLU gate of a backward product: Y[m, n] = 0 where mask[m, n] <= 0
  // optional fused second layer of a two-layer head: instead of storing Y, reduce it against red_w[grp] [red_nout, N]
  //   red_out[row * red_stride + grp * red_nout + j] (+)= sum_n Y[row, n] * red_w[grp][j, n]  (+ red_b[grp][j] once)
  // (hidden activations of the acquisition / GMM heads never reach memory; red_nout <= 3).  With more than one
  // column block per row (gemm_col_blocks(N) > 1) block y writes its partial sums at red_out + y * red_block_stride
  // and the consumer adds the blocks in order (no atomics: results stay bit-reproducible).
  const float *red_w[GEMM_MAX_GROUPS]; const float *red_b[GEMM_MAX_GROUPS]; int red_nout; float *red_out; int red_stride;
  long red_block_stride;
  unsigned *range_flag;            // F16X3: raised when an accumulator is not finite (an operand left f16's range); may be null
  // F16X3 only, optional: per-tensor power-of-two scaling of the X operand.  *xmax_bits = the bits of max |X| (absmax_bits_kernel);
  // the kernel multiplies X by 2^k on its way into LDS, k chosen so that the maximum lands in [2^14, 2^15) -- gradients of
  // 1e-8 .. 1e-5 would otherwise sit in f16's subnormals -- and the epilogue divides by it (exact: powers of two).  Null: no scaling.
  const unsigned *xmax_bits;
  // optional: max |Y| of what this launch stores (bits, atomicMax per wave) -- Y is the gradient operand of a following F16X3 product
  unsigned *out_absmax;
};

// max |x| over a [rows, cols] matrix of pitch ld, as the bit pattern of the non-negative float (unsigned order = float order; a NaN
// compares above every finite value and is reported as such): one atomicMax per wave into *out, which the caller zeroed.
__global__ __launch_bounds__(256) void absmax_bits_kernel(const float *__restrict__ x, long rows, int cols, int ld, unsigned *out) {
  const long n4 = rows * (cols / 4);
  unsigned m = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long r = i / (cols / 4);
    const int c = (int)(i - r * (cols / 4)) * 4;
    const float4 v = *reinterpret_cast<const float4 *>(x + r * ld + c);
    m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), max(__float_as_uint(v.y) & 0x7fffffffu, max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu)));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
// the scale 2^k for an operand whose max |x| has the bits `b`: max * 2^k in [2^14, 2^15); 1 for an all-zero or non-finite operand
// (a non-finite one shows up in the accumulators and raises the range flag there)
__device__ __forceinline__ float f16_operand_scale(unsigned b, float &inv) {
  const int e = (int)(b >> 23);                       // biased exponent of max |x|
  if (b == 0 || e >= 255) { inv = 1.f; return 1.f; }
  int k = 14 + 127 - e;                               // 2^(e - 127) <= max < 2^(e - 126)
  k = k > 126 ? 126 : (k < -126 ? -126 : k);
  inv = __uint_as_float((unsigned)(127 - k) << 23);
  return __uint_as_float((unsigned)(127 + k) << 23);
}

constexpr int GEMM_BK = 32;

template <int PREC> struct LdsTile;  // storage of a [ROWS x 32] operand tile in LDS
template <> struct LdsTile<0> { static constexpr int LD = 34; using T = float; static constexpr int PLANES = 1; };
template <> struct LdsTile<1> { static constexpr int LD = 32; using T = unsigned short; static constexpr int PLANES = 1; };
template <> struct LdsTile<2> { static constexpr int LD = 32; using T = unsigned short; static constexpr int PLANES = 2; };
template <> struct LdsTile<3> { static constexpr int LD = 32; using T = unsigned short; static constexpr int PLANES = 2; };
constexpr float GEMM_F16_WSCALE = 256.f;
// 16-bit operand tiles: rows of 32 k-values = 64 bytes, the four 16-byte chunks of a row XOR-swizzled by (row / 4) % 4:
// the 8-byte stores of a half wave (4 rows x 8 slots) and the 16-byte fragment reads of 16 rows both touch every bank once
// (a 40-element pitch made the reads conflict-free and the stores two-way conflicting: 47 % of the LDS-active cycles by PMC)
__device__ __forceinline__ int gemm_swz(int row, int k) { return row * 32 + ((((k >> 3) ^ (row >> 2)) & 3) << 3) + (k & 7); }

typedef __attribute__((ext_vector_type(8))) _Float16 gemm_f16x8;
__device__ __forceinline__ void split_f16(float a, unsigned short &hi, unsigned short &lo) {
  const _Float16 h = (_Float16)a;
  const _Float16 l = (_Float16)(a - (float)h);
  hi = __builtin_bit_cast(unsigned short, h);
  lo = __builtin_bit_cast(unsigned short, l);
}

__device__ __forceinline__ void split2_f16(float a, float b, unsigned &hi, unsigned &lo) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  typedef __attribute__((ext_vector_type(2))) float f2;
  const f2 v = {a, b};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2));
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(a));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(b));
  const f2 r = {r0, r1};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, h2));
}

// store 4 consecutive k-values of one tile row
template <int PREC>
__device__ __forceinline__ void lds_store4(typename LdsTile<PREC>::T *base, int plane_elems, int row,
                                           int k, float4 v, float scale = 1.f) {
  constexpr int LD = LdsTile<PREC>::LD;
  if constexpr (PREC == 3) {
    // packed pairs: hi = one v_cvt_pk_f16_f32 per pair, residuals one v_fma_mix_f32 each (the f16 operand read straight out
    // of the packed register), lo = one more v_cvt_pk: 4 instructions per pair instead of 8 scalar conversions
    unsigned h0, l0, h1, l1;
    split2_f16(v.x * scale, v.y * scale, h0, l0);
    split2_f16(v.z * scale, v.w * scale, h1, l1);
    typedef __attribute__((ext_vector_type(2))) unsigned gemm_u32x2;
    *reinterpret_cast<gemm_u32x2 *>(base + gemm_swz(row, k)) = (gemm_u32x2){h0, h1};
    *reinterpret_cast<gemm_u32x2 *>(base + plane_elems + gemm_swz(row, k)) = (gemm_u32x2){l0, l1};
    return;
  }
  if constexpr (PREC == 0) {
    float *p = base + row * LD + k;
    p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
  } else if constexpr (PREC == 1) {
    u16x4 h = {f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
    *reinterpret_cast<u16x4 *>(base + gemm_swz(row, k)) = h;
  } else {
    u16x4 h, l;
    unsigned short a, b;
    split_bf16(v.x, a, b); h[0] = a; l[0] = b;
    split_bf16(v.y, a, b); h[1] = a; l[1] = b;
    split_bf16(v.z, a, b); h[2] = a; l[2] = b;
    split_bf16(v.w, a, b); h[3] = a; l[3] = b;
    *reinterpret_cast<u16x4 *>(base + gemm_swz(row, k)) = h;
    *reinterpret_cast<u16x4 *>(base + plane_elems + gemm_swz(row, k)) = l;
  }
}

template <int PREC, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256, 3) void gemm_nt_kernel(GemmArgs a) {
  using LT = LdsTile<PREC>;
  using T = typename LT::T;
  constexpr int LD = LT::LD;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int A_ELEMS = BM * LD, B_ELEMS = BN * LD;
  __shared__ __attribute__((aligned(16))) T smem[(A_ELEMS + B_ELEMS) * LT::PLANES];
  T *As = smem;
  T *Bs = smem + A_ELEMS * LT::PLANES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  // Workgroup -> (row block, column block): the column blocks of one row block read the same A rows; they get ids 8
  // apart so that they run on the same XCD at about the same time (ids go round-robin over the 8 XCDs) and the rows
  // come out of that XCD's L2 after the first read instead of N / BN times out of HBM.
  int mb = blockIdx.x, nb = blockIdx.y;
  if (gridDim.y > 1) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y, per = 8 * gridDim.y;
    const int chunk = lin / per, within = lin - chunk * per;
    const int rows_in_chunk = min(8, (int)gridDim.x - chunk * 8);
    mb = chunk * 8 + within % rows_in_chunk;
    nb = within / rows_in_chunk;
  }
  const int m0 = mb * BM, n0 = nb * BN, grp = blockIdx.z;
  const float *__restrict__ W = a.W[grp];
  float xscale = 1.f, xinv = 1.f;
  if constexpr (PREC == 3) {
    if (a.xmax_bits) xscale = f16_operand_scale(*a.xmax_bits, xinv);
  }

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per-thread staging coordinates: 8 float4 per 32-wide row
  constexpr int A_IT = BM * 8 / 256, B_IT = BN * 8 / 256;
  const float *arow[A_IT];
  bool aok[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    int idx = tid + i * 256, row = idx >> 3;
    int m = m0 + row;
    aok[i] = m < a.M;
    int mm = aok[i] ? m : 0;
    long src = (long)(mm / a.R_in) * a.G_in + a.off_in + (mm % a.R_in);
    if (a.row_index) {
      const int ri = a.row_index[mm];
      aok[i] = aok[i] && ri >= 0;
      src = ri >= 0 ? ri : 0;
    }
    arow[i] = a.X + src * a.ldx + (idx & 7) * 4;
  }

  const int fr = lane & 15, fg = lane >> 4;
  // software pipeline: the global loads of k-step s + 1 are issued before the MFMAs of step s and land while they
  // run (one LDS buffer, two barriers per step; the operands of the next step wait in registers)
  float4 av[A_IT], bv[B_IT];
  auto load_step = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      av[i] = aok[i] ? *reinterpret_cast<const float4 *>(arow[i] + k0) : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      int idx = tid + i * 256, row = idx >> 3;
      bv[i] = *reinterpret_cast<const float4 *>(W + (long)(n0 + row) * a.ldw + k0 + (idx & 7) * 4);
    }
  };
  load_step(0);
  for (int k0 = 0; k0 < a.K; k0 += GEMM_BK) {
    __syncthreads();  // previous step's fragment reads are done
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int idx = tid + i * 256;
      lds_store4<PREC>(As, A_ELEMS, idx >> 3, (idx & 7) * 4, av[i], xscale);
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      int idx = tid + i * 256;
      lds_store4<PREC>(Bs, B_ELEMS, idx >> 3, (idx & 7) * 4, bv[i], PREC == 3 ? GEMM_F16_WSCALE : 1.f);
    }
    __syncthreads();
    if (k0 + GEMM_BK < a.K) load_step(k0 + GEMM_BK);

    if constexpr (PREC == 0) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        float af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = As[(wm * WM + i * 16 + fr) * LD + ks * 4 + fg];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = Bs[(wn * WN + j * 16 + fr) * LD + ks * 4 + fg];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    } else if constexpr (PREC == 3) {
      gemm_f16x8 ah[TM], bh[TN], al[TM], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const gemm_f16x8 *>(As + gemm_swz(wm * WM + i * 16 + fr, fg * 8));
        al[i] = *reinterpret_cast<const gemm_f16x8 *>(As + A_ELEMS + gemm_swz(wm * WM + i * 16 + fr, fg * 8));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const gemm_f16x8 *>(Bs + gemm_swz(wn * WN + j * 16 + fr, fg * 8));
        bl[j] = *reinterpret_cast<const gemm_f16x8 *>(Bs + B_ELEMS + gemm_swz(wn * WN + j * 16 + fr, fg * 8));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    } else {
      bf16x8 ah[TM], bh[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        ah[i] = *reinterpret_cast<const bf16x8 *>(As + gemm_swz(wm * WM + i * 16 + fr, fg * 8));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bh[j] = *reinterpret_cast<const bf16x8 *>(Bs + gemm_swz(wn * WN + j * 16 + fr, fg * 8));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      if constexpr (PREC == 2) {
        bf16x8 al[TM], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          al[i] = *reinterpret_cast<const bf16x8 *>(As + A_ELEMS + gemm_swz(wm * WM + i * 16 + fr, fg * 8));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bl[j] = *reinterpret_cast<const bf16x8 *>(Bs + B_ELEMS + gemm_swz(wn * WN + j * 16 + fr, fg * 8));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          }
      }
    }
  }

  // epilogue: C/D fragment layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg.  The tile goes
  // through LDS 32 rows at a time so that bias / ReLU / gate / accumulate and the store run on whole output rows
  // (float4 per lane, BN * 4 contiguous bytes per row) instead of 64-byte column segments.
  constexpr int LDC = BN + 4;
  constexpr int RP = 32 * LDC * sizeof(float) <= sizeof(smem) ? 32 : 16;     // rows per pass
  static_assert(RP * LDC * sizeof(float) <= sizeof(smem), "epilogue tile must fit the operand buffers");
  float *Cs = reinterpret_cast<float *>(smem);
  if constexpr (PREC == 3) {   // f16 range guard: an out-of-range operand shows as inf / NaN accumulators (0 * inf = NaN, 0 * NaN = NaN)
    float chk = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) chk = fmaf(acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3], 0.f, chk);
    range_check_nan(a.range_flag, chk);
  }
  const float *bias = a.bias[grp];
  const float tsc = a.tscalar ? a.tscalar[0] : 0.f;
  unsigned omax = 0;
  constexpr int C4 = BN / 4, PER_T = RP * C4 / 256;       // float4 per thread and pass (BN = 32: 1, 64: 2, 128: 4)
  static_assert(RP * C4 % 256 == 0, "whole float4 rounds per pass");
#pragma unroll
  for (int q = 0; q < BM / RP; ++q) {
    __syncthreads();                                       // fragment reads / previous pass done
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int rbase = wm * WM + 16 * i;
      if (rbase / RP == q) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            Cs[(rbase % RP + fg * 4 + r) * LDC + wn * WN + 16 * j + fr] = PREC == 3 ? acc[i][j][r] * ((1.f / GEMM_F16_WSCALE) * xinv) : acc[i][j][r];
      }
    }
    __syncthreads();
    if (a.red_out) {
      // fused second layer: SEGS = BN / 16 consecutive lanes own one row of the pass, 16 columns each
      constexpr int SEGS = BN / 16;
      const int row = tid / SEGS, seg = tid % SEGS;
      const int m = m0 + RP * q + row;
      if (row < RP && m < a.M) {
        float pj[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int n = n0 + 16 * seg + 4 * i;
          float4 v = *reinterpret_cast<const float4 *>(Cs + row * LDC + 16 * seg + 4 * i);
          if (bias) { v.x += bias[n]; v.y += bias[n + 1]; v.z += bias[n + 2]; v.w += bias[n + 3]; }
          if (a.tscalar || a.tvec) {
            const float tr = a.tvec ? a.tvec[m / a.tvec_div] : tsc;
            v.x += tr * a.tcol[(long)n * a.tcol_stride]; v.y += tr * a.tcol[(long)(n + 1) * a.tcol_stride];
            v.z += tr * a.tcol[(long)(n + 2) * a.tcol_stride]; v.w += tr * a.tcol[(long)(n + 3) * a.tcol_stride];
          }
          if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
#pragma unroll
          for (int j = 0; j < 3; ++j)
            if (j < a.red_nout) {
              const float *w = a.red_w[grp] + (long)j * a.N + n;
              pj[j] += v.x * w[0] + v.y * w[1] + v.z * w[2] + v.w * w[3];
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int o = SEGS / 2; o > 0; o >>= 1) pj[j] += __shfl_xor(pj[j], o, 64);
        if (seg == 0) {
          const long dst = (long)(m / a.R_out) * a.G_out + a.off_out + (m % a.R_out);
          float *op = a.red_out + nb * a.red_block_stride + dst * a.red_stride + grp * a.red_nout;
          for (int j = 0; j < a.red_nout; ++j) op[j] = pj[j] + (nb == 0 && a.red_b[grp] ? a.red_b[grp][j] : 0.f);
        }
      }
      continue;
    }
#pragma unroll
    for (int e = 0; e < PER_T; ++e) {
      const int idx = tid + e * 256, row = idx / C4, c4 = idx % C4;
      const int m = m0 + RP * q + row, n = n0 + 4 * c4;
      if (m >= a.M) continue;
      float4 v = *reinterpret_cast<const float4 *>(Cs + row * LDC + 4 * c4);
      if (bias) { v.x += bias[n]; v.y += bias[n + 1]; v.z += bias[n + 2]; v.w += bias[n + 3]; }
      if (a.tscalar || a.tvec) {
        const float tr = a.tvec ? a.tvec[m / a.tvec_div] : tsc;
        v.x += tr * a.tcol[(long)n * a.tcol_stride]; v.y += tr * a.tcol[(long)(n + 1) * a.tcol_stride];
        v.z += tr * a.tcol[(long)(n + 2) * a.tcol_stride]; v.w += tr * a.tcol[(long)(n + 3) * a.tcol_stride];
      }
      if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (a.mask) {
        const float4 g = *reinterpret_cast<const float4 *>(a.mask + (long)m * a.ldmask + n);
        if (!(g.x > 0.f)) v.x = 0.f;
        if (!(g.y > 0.f)) v.y = 0.f;
        if (!(g.z > 0.f)) v.z = 0.f;
        if (!(g.w > 0.f)) v.w = 0.f;
      }
      long dst = (long)(m / a.R_out) * a.G_out + a.off_out + (m % a.R_out);
      if (a.out_index) { dst = a.out_index[m]; if (dst < 0) continue; }
      float4 *yp = reinterpret_cast<float4 *>(a.Y + dst * a.ldy + grp * a.col_per_group + n);
      if (a.accum) { const float4 o = *yp; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
      *yp = v;
      omax = max(max(omax, __float_as_uint(v.x) & 0x7fffffffu), max(__float_as_uint(v.y) & 0x7fffffffu, max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu)));
    }
  }
  if (a.out_absmax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) omax = max(omax, (unsigned)__shfl_xor((int)omax, o, 64));
    if (lane == 0 && omax) atomicMax(a.out_absmax, omax);
  }
}

// (Round 4 measured a 256 x 256-block variant of this product -- 8 waves, wave tiles of 64 x 128, two LDS images, one barrier per
//  k-step: the structure of gemm_tn_f16_kernel in backward.h -- against this kernel on the d = 256 training step: 765 ms with it on
//  every eligible GEMM, 738 ms with it on the K >= 512 ones only, 712 - 719 ms without it.  One workgroup of 8 waves per CU whose
//  waves meet at a barrier every 768 MFMAs loses to three independent 4-wave workgroups at K = 256 .. 1024 (8 .. 32 k-steps per
//  block: prologue and the strided epilogue are not amortised as they are over the 100+ slabs of a weight-gradient block).  Not kept.)

// column blocks per output row for N outputs (tile width 128 / 64 / 32 by divisibility): a fused second layer
// (red_*) leaves that many partial sums per output
static inline int gemm_col_blocks(int N) { return N / (N % 128 == 0 ? 128 : N % 64 == 0 ? 64 : 32); }

template <int PREC>
static int launch_gemm_prec(const GemmArgs &a_in, int groups, hipStream_t st) {
  const GemmArgs &a = a_in;
  if (a.K % GEMM_BK != 0 || a.N % 32 != 0 || a.M <= 0) return -2;
  // the epilogue moves float4: 16-byte aligned output rows (and gate rows)
  if (!a.red_out && (a.ldy % 4 || a.col_per_group % 4 || (reinterpret_cast<uintptr_t>(a.Y) & 15))) return -2;
  if (a.red_out && (a.red_nout < 1 || a.red_nout > 3)) return -2;
  if (a.mask && (a.ldmask % 4 || (reinterpret_cast<uintptr_t>(a.mask) & 15))) return -2;
  dim3 block(256);
  if (!a.red_out && a.N % 96 == 0 && a.N % 128 != 0) {
    // in-projection of the small models (N = 3 d = 96 at d = 32): one column block, every input row is read once
    dim3 grid((a.M + 127) / 128, a.N / 96, groups);
    hipLaunchKernelGGL((gemm_nt_kernel<PREC, 128, 96, 4, 1>), grid, block, 0, st, a);
  } else if (a.N % 128 == 0) {
    dim3 grid((a.M + 127) / 128, a.N / 128, groups);
    hipLaunchKernelGGL((gemm_nt_kernel<PREC, 128, 128, 2, 2>), grid, block, 0, st, a);
  } else if (a.N % 64 == 0) {
    dim3 grid((a.M + 127) / 128, a.N / 64, groups);
    hipLaunchKernelGGL((gemm_nt_kernel<PREC, 128, 64, 4, 1>), grid, block, 0, st, a);
  } else {
    dim3 grid((a.M + 127) / 128, a.N / 32, groups);
    hipLaunchKernelGGL((gemm_nt_kernel<PREC, 128, 32, 4, 1>), grid, block, 0, st, a);
  }
  return 0;
}

static int launch_gemm(int prec, const GemmArgs &a, int groups, hipStream_t st) {
  switch (prec) {
    case 0: return launch_gemm_prec<0>(a, groups, st);
    case 1: return launch_gemm_prec<1>(a, groups, st);
    case 2: return launch_gemm_prec<2>(a, groups, st);
    case 3: return launch_gemm_prec<3>(a, groups, st);
  }
  return -1;
}
