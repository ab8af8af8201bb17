// Backward kernels of one design step (fp32).  The T steps of a rollout are independent given the
// roles (design selection is discrete: no gradient flows through it, train_aline.py:113-125), so the
// backward pass treats the T x B (step, episode) pairs as one batch of "instances", recomputes the
// forward with the generic pipeline while saving what it needs, and back-propagates once.
#pragma once
#include "common.h"
#include "kernels.h"
#include "gemm.h"

// dW[n, k] += sum_m dY[m, n] * X[m, k]   and (optionally) db[n] += sum_m dY[m, n]
// "TN" product over a very tall M.  One workgroup = one block of dW x one chunk of rows; its 4 waves take
// interleaved row groups, accumulate with exact-fp32 16x16x4 MFMAs (A = dY^T: lane n holds dY[m, n],
// k index = row m), are summed through LDS and added to global memory with float atomics.
struct GemmTnArgs {
  const float *dY; int ldy; int Ry, Gy, offy;   // row maps as in gemm.h
  const float *X; int ldx; int Rx, Gx, offx;
  float *dW; int ldw;                            // [N, K]
  float *db;                                     // [N] or null
  long M; int N, K; long mchunk;
  int swap;                                      // block kernel: roles exchanged (dY is the 32-wide operand), see gemm_dw
  // optional (TN = 8, no swap): one [128, K] gradient per column block of dY -- the C first layers of the GMM heads, whose hidden
  // gradients sit side by side in one [rows, C F] matrix, in ONE launch (blockIdx.y = component)
  float *dWg[16]; float *dbg[16]; int grouped;
  int gx, nby, nbz;                              // block kernel: row chunks, blocks along N (or components), blocks along K
  int walk;                                      // block kernel: row groups of >= 64 rows (or identity maps), M < 2^30: the walked loop
};

// Row m of a gathered operand lives at row (m / R) * G + off + m % R of its matrix (gemm.h).  The product below walks m in small fixed steps: the
// quotient / remainder pair is kept per lane and advanced without a division or a branch (steps <= R; R == G is the identity map m + off).
struct RowWalk {
  int R, G, off, r, q;
  __device__ __forceinline__ void init(long m, int R_, int G_, int off_) {
    if (R_ == G_) { R = 0x7fffffff; G = 0; off = off_; q = 0; r = (int)m; }
    else { R = R_; G = G_; off = off_; q = (int)(m / R_); r = (int)(m - (long)q * R_); }
  }
  __device__ __forceinline__ void step(int d) {
    r += d;
    const bool w = r >= R;
    r = w ? r - R : r;
    q += w ? 1 : 0;
  }
  __device__ __forceinline__ long src() const { return (long)q * G + (off + r); }
};
// The workgroup owns a [16 TN x 16 TK] block of dW for its row chunk (TN = 8, 6, 4, 2 by the divisibility of N; TK = 4 where the other
// dimension is a multiple of 64, else 2): a row of dY is read once per 16 TK columns of X, TN TK MFMAs per (TN + TK) loads, four row
// groups in flight per wave.  (TK was 2 everywhere: at d = 512 the in-projection's dY [M, 1536] was read 16 times -- 190 GB out of
// L2 per call -- and the dW products were 37 - 41 % of the training step of the wide models.)
template <int TN, int TK = 2>
__global__ __launch_bounds__(256, 2) void gemm_tn_block_kernel(GemmTnArgs a) {
  constexpr int BN = 16 * TN, BK = 16 * TK;
  __shared__ float red[BN][BK + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  // Workgroup order.  The nby x nbz blocks of dW that share a row chunk read the same rows of dY and X: they must run at the same time and on the
  // same XCD, or every one of them streams its rows from HBM again (chunk-fastest order, the first version: the 32 blocks of a [1024 x 256]
  // gradient were up to 1 487 workgroup ids apart and each streamed its rows from HBM: PMC FETCH_SIZE now 1.75 x the operands' size per launch).
  // Ids go round-robin over the 8 XCDs: ids congruent mod 8 inside a group of 8 nby nbz consecutive ones share a chunk.
  const unsigned nb_all = (unsigned)(a.nby * a.nbz), per = 8u * nb_all;
  const unsigned grp = blockIdx.x / per, within = blockIdx.x - grp * per;
  const long chunk = (long)grp * 8 + (within & 7u);
  const int nb = (int)(within >> 3);
  if (chunk >= a.gx) return;
  const int by = nb % a.nby, bz = nb / a.nby;
  const int n0 = by * BN, k0 = bz * BK;
  const long m_lo = chunk * a.mchunk, m_hi = min(a.M, m_lo + a.mchunk);
  f32x4 acc[TN][TK];
  float bsum[TN], bsum2[TK];
#pragma unroll
  for (int j = 0; j < TK; ++j) bsum2[j] = 0.f;
#pragma unroll
  for (int i = 0; i < TN; ++i) {
#pragma unroll
    for (int j = 0; j < TK; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bsum[i] = 0.f;
  }
  constexpr int UN = 4;                  // row groups (of 4 rows) in flight per wave
  long t_lo = m_lo;
  if (a.walk) {
    // The complete 64-row groups of the chunk, every lane in range: one basic block per group -- 4 (TN + TK) unmasked loads, then the MFMAs --
    // with the row maps walked.  (A two-register-set pipeline of half groups -- loads of one behind the MFMAs of the other -- was built twice and
    // spilled 26 - 50 registers at two waves per SIMD; a spill reload waits for every load in flight, and it ran 1.7 x slower.  In the general loop below every load sits behind its own range test and each row costs two 64-bit divisions
    // per operand: ~1 500 cycles of vector work per 128 MFMAs that nothing overlaps within the wave.)
    const long nfull = (m_hi - m_lo) / 64;
    t_lo = m_lo + 64 * nfull;
    RowWalk wy, wx;
    wy.init(m_lo + 16 * wave + fg, a.Ry, a.Gy, a.offy);
    wx.init(m_lo + 16 * wave + fg, a.Rx, a.Gx, a.offx);
    const float *const Ybase = a.dY + n0 + fr, *const Xbase = a.X + k0 + fr;
    // Rotated by a row group of 4: the operands of (group g + 1, rows 4 u ..) are requested right behind the MFMAs of (group g, rows 4 u ..),
    // into the registers those have just freed -- three MFMA phases (~3 000 cycles) ahead of their use, no second register set.
    float av[UN][TN], bv[UN][TK];
    auto load_u = [&](int u) {
      const float *py = Ybase + wy.src() * a.ldy, *px = Xbase + wx.src() * a.ldx;
#pragma unroll
      for (int i = 0; i < TN; ++i) av[u][i] = py[16 * i];
#pragma unroll
      for (int j = 0; j < TK; ++j) bv[u][j] = px[16 * j];
      wy.step(u + 1 < UN ? 4 : 52); wx.step(u + 1 < UN ? 4 : 52);        // rows +0, +4, +8, +12 | +64 ...
    };
    auto mul_u = [&](int u) {
#pragma unroll
      for (int j = 0; j < TK; ++j) bsum2[j] += bv[u][j];
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        bsum[i] += av[u][i];
#pragma unroll
        for (int j = 0; j < TK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][i], bv[u][j], acc[i][j], 0, 0, 0);
      }
    };
    if (nfull > 0) {
#pragma unroll
      for (int u = 0; u < UN; ++u) load_u(u);
      for (long gi = 0; gi + 1 < nfull; ++gi) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          __builtin_amdgcn_sched_barrier(0);
          mul_u(u);
          __builtin_amdgcn_sched_barrier(0);
          load_u(u);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < UN; ++u) mul_u(u);
    }
  }
  for (long mb = t_lo + 4 * UN * wave; mb < m_hi; mb += 16 * UN) {
    float av[UN][TN], bv[UN][TK];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long m = mb + 4 * u + fg;
      const bool ok = m < m_hi;
      const long mm = ok ? m : m_lo;
      const float *py = a.dY + ((mm / a.Ry) * a.Gy + a.offy + (mm % a.Ry)) * a.ldy + n0 + fr;
      const float *px = a.X + ((mm / a.Rx) * a.Gx + a.offx + (mm % a.Rx)) * a.ldx + k0 + fr;
#pragma unroll
      for (int i = 0; i < TN; ++i) av[u][i] = ok ? py[16 * i] : 0.f;
#pragma unroll
      for (int j = 0; j < TK; ++j) bv[u][j] = ok ? px[16 * j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int j = 0; j < TK; ++j) bsum2[j] += bv[u][j];
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        bsum[i] += av[u][i];
#pragma unroll
        for (int j = 0; j < TK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][i], bv[u][j], acc[i][j], 0, 0, 0);
      }
  }
  // The four waves' tiles are summed in LDS one wave after the other with plain stores / read-modify-writes (an element belongs to one lane of
  // a wave): as LDS float atomics this epilogue kept the LDS busy for 17 % of the kernel (PMC: 126 cycles per ds_add_f32 instruction).
  // acc[i][j][r]: n = 16 i + 4 fg + r, k = 16 j + fr
#pragma unroll 1
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float *p = &red[16 * i + 4 * fg + r][16 * j + fr];
            *p = w == 0 ? acc[i][j][r] : *p + acc[i][j][r];
          }
    }
    __syncthreads();
  }
  if (!a.swap) {
    float *dWo = a.grouped ? a.dWg[by] : a.dW + (long)n0 * a.ldw;
    float *dbo = a.grouped ? a.dbg[by] : (a.db ? a.db + n0 : nullptr);
    for (int e = threadIdx.x; e < BN * BK; e += 256) {
      const int n = e / BK, k = e % BK;
      atomicAdd(dWo + (long)n * a.ldw + k0 + k, red[n][k]);
    }
    if (dbo && bz == 0) {
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        float s = bsum[i];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (fg == 0) atomicAdd(dbo + 16 * i + fr, s);
      }
    }
  } else {
    // the wide operand is X: this block holds dW^T[n0.., k0..]; dW is [K_args, N_args] = [32-wide index, wide index]
    for (int e = threadIdx.x; e < BN * BK; e += 256) {
      const int k = e / BN, n = e % BN;
      atomicAdd(a.dW + (long)(k0 + k) * a.ldw + n0 + n, red[n][k]);
    }
    if (a.db && by == 0) {
#pragma unroll
      for (int j = 0; j < TK; ++j) {
        float s = bsum2[j];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (fg == 0) atomicAdd(a.db + k0 + 16 * j + fr, s);
      }
    }
  }
}

// Wide-row LayerNorm backward (d = 256 NV: 256, 512): one wave per row, NV float4 per lane -- 16 B / lane loads and stores where the
// kernel above moves 4 B / lane (it ran at 2.4 TB/s on the d = 256 training step; this one is HBM-bound like add_layernorm_wide_kernel).
// Same arithmetic; dw / db partials per lane, summed per workgroup through LDS, one atomic per feature and workgroup.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_wide_kernel(const float *__restrict__ dY, const float *__restrict__ U,
                                                                 const float *__restrict__ w, float *__restrict__ dU, float *dw, float *db,
                                                                 long rows, unsigned *out_absmax) {
  constexpr int d = 256 * NV;
  __shared__ float sdw[d], sdb[d];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < d; i += 256) { sdw[i] = 0.f; sdb[i] = 0.f; }
  __syncthreads();
  float4 wv[NV], pdw[NV], pdb[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    wv[i] = *reinterpret_cast<const float4 *>(w + 4 * (lane + 64 * i));
    pdw[i] = make_float4(0.f, 0.f, 0.f, 0.f); pdb[i] = pdw[i];
  }
  unsigned omax = 0;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    float4 u[NV], gy[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      u[i] = *reinterpret_cast<const float4 *>(U + row * d + 4 * (lane + 64 * i));
      gy[i] = *reinterpret_cast<const float4 *>(dY + row * d + 4 * (lane + 64 * i));
      s += (u[i].x + u[i].y) + (u[i].z + u[i].w);
    }
    const float mean = wave_sum(s) / d;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      u[i].x -= mean; u[i].y -= mean; u[i].z -= mean; u[i].w -= mean;
      ss += (u[i].x * u[i].x + u[i].y * u[i].y) + (u[i].z * u[i].z + u[i].w * u[i].w);
    }
    const float rstd = rsqrtf(wave_sum(ss) / d + 1e-5f);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float *x = reinterpret_cast<float *>(&u[i]), *gq = reinterpret_cast<float *>(&gy[i]);
      const float *wq = reinterpret_cast<const float *>(&wv[i]);
      float *aw = reinterpret_cast<float *>(&pdw[i]), *ab = reinterpret_cast<float *>(&pdb[i]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float xh = x[q] * rstd, gg = gq[q] * wq[q];
        aw[q] = fmaf(gq[q], xh, aw[q]);
        ab[q] += gq[q];
        x[q] = xh; gq[q] = gg;
        sg += gg;
        sgx = fmaf(gg, xh, sgx);
      }
    }
    const float mg = wave_sum(sg) / d, mgx = wave_sum(sgx) / d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float4 o = make_float4(rstd * (gy[i].x - mg - u[i].x * mgx), rstd * (gy[i].y - mg - u[i].y * mgx),
                                   rstd * (gy[i].z - mg - u[i].z * mgx), rstd * (gy[i].w - mg - u[i].w * mgx));
      *reinterpret_cast<float4 *>(dU + row * d + 4 * (lane + 64 * i)) = o;
      omax = max(max(omax, __float_as_uint(o.x) & 0x7fffffffu), max(__float_as_uint(o.y) & 0x7fffffffu, max(__float_as_uint(o.z) & 0x7fffffffu, __float_as_uint(o.w) & 0x7fffffffu)));
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float *aw = reinterpret_cast<const float *>(&pdw[i]), *ab = reinterpret_cast<const float *>(&pdb[i]);
#pragma unroll
    for (int q = 0; q < 4; ++q) { atomicAdd(&sdw[4 * (lane + 64 * i) + q], aw[q]); atomicAdd(&sdb[4 * (lane + 64 * i) + q], ab[q]); }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += 256) { atomicAdd(dw + c, sdw[c]); atomicAdd(db + c, sdb[c]); }
  if (out_absmax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) omax = max(omax, (unsigned)__shfl_xor((int)omax, o, 64));
    if (lane == 0 && omax) atomicMax(out_absmax, omax);
  }
}

// ---- weight-gradient product on the f16 matrix pipe (round 4) ---------------------------------------------------------------------
// out[a, b] += sum_m P[m, a0 + a] * Q[m, b0 + b] for a 256 x 256 block of (P columns) x (Q columns) and a chunk of rows, every product
// the 3-term f16 split of the forward (hi * hi + hi * lo + lo * hi, fp32 accumulate).  One of the operands is a gradient (1e-8 .. 1e-5):
// it is multiplied by the per-tensor power of two of gemm.h (xmax_bits) on its way into LDS and the block is divided by it at the end.
// 8 waves; a slab of 32 rows of both tiles (64 KB of fp32) is loaded as float4 (a wave reads 1 KB of one row per instruction), split and
// stored COLUMN-major -- [column][32 rows] f16, hi and lo planes, the 16-byte chunks swizzled as in gemm.h -- so that the fragment of a
// 16x16x32 MFMA (8 consecutive rows of one column per lane) is one ds_read_b128: the same LDS image and MFMA loop as the NT kernel.
// Wave (wi, wj) owns 64 x 128 of the block (32 accumulator tiles).  Two LDS images (128 KB): while the 768 MFMAs of slab s run out of
// one, slab s + 1 (loaded during slab s - 1) is split and stored into the other and the loads of slab s + 2 are issued -- one barrier
// per slab, no phase in which the matrix pipe waits for the staging (single-buffered: 0.76 ms per call, 25 % of the f16 peak).
// The exact-fp32 kernel above needs 8 MFMAs of 32 cycles where this one needs 3 of 16.
struct GemmTnF16Args {
  const float *P; int ldp;            // [M, ldp]: the operand whose columns index the ROWS of the block (a)
  const float *Q; int ldq;            // [M, ldq]: ... the COLUMNS of the block (b)
  int Rp, Gp, offp, Rq, Gq, offq;     // row maps as in gemm.h: product row m is row (m / R) * G + off + m % R of the operand (R == G: m + off)
  float *out; long sa, sb;            // out[a * sa + b * sb] += ...   (dW [N, K]: P = dY gives sa = ldw, sb = 1; P = X gives sa = 1, sb = ldw)
  float *colsum;                      // += column sums of the GRADIENT operand (the bias gradient), or null
  int grad_is_p;                      // which operand is the gradient (scaled; its column sums go to colsum)
  const unsigned *xmax_bits;          // bits of max |gradient operand| (absmax_bits_kernel)
  long M, mchunk; int gx, nba, nbb;   // rows, rows per workgroup, row chunks, blocks along a / b
  const int *row_index;               // optional: product row m is row row_index[m] of BOTH operands (< 0: no row) -- the key rows of every instance
};

__global__ __launch_bounds__(512, 1) void gemm_tn_f16_kernel(GemmTnF16Args a) {
  constexpr int TILE = 256, COLS = 2 * TILE, PLANE = COLS * 32;          // f16 elements per plane
  extern __shared__ __attribute__((aligned(16))) unsigned short tn_smem[];     // 2 images x (hi | lo) x 32 KB = 128 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  // workgroup order: the nba x nbb blocks of one row chunk get ids congruent mod 8 inside a group of 8 nba nbb consecutive ones: same XCD,
  // about the same time -- the rows they share come out of that L2 (as gemm_tn_block_kernel)
  const unsigned nb_all = (unsigned)(a.nba * a.nbb), per = 8u * nb_all;
  const unsigned grp = blockIdx.x / per, within = blockIdx.x - grp * per;
  const long chunk = (long)grp * 8 + (within & 7u);
  const int nb = (int)(within >> 3);
  if (chunk >= a.gx) return;
  const int ba = nb % a.nba, bb = nb / a.nba;
  const long m_lo = chunk * a.mchunk, m_hi = min(a.M, m_lo + a.mchunk);
  float xinv = 1.f;
  const float xscale = f16_operand_scale(*a.xmax_bits, xinv);
  // staging task of this thread: 4 consecutive columns x 8 rows of the combined [32 x 512] slab; waves 0, 2, 4, 6 load P, the others Q
  const int rg = __builtin_amdgcn_readfirstlane(wave >> 1);   // row group 0 .. 3 (rows 8 rg .. 8 rg + 7 of the slab)
  const bool is_p = (wave & 1) == 0;
  const int c4 = 4 * lane;                                    // first of the 4 columns inside the operand's tile
  const float *src = is_p ? a.P + (long)ba * TILE + c4 : a.Q + (long)bb * TILE + c4;
  const int ld = is_p ? a.ldp : a.ldq;
  const bool is_grad = is_p == (a.grad_is_p != 0);
  const float sc = is_grad ? xscale : 1.f;
  const int ccol = (is_p ? 0 : TILE) + c4;                    // column of the LDS image
  float4 v[8];
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  const int mapR = is_p ? a.Rp : a.Rq, mapG = is_p ? a.Gp : a.Gq, mapOff = is_p ? a.offp : a.offq;
  auto load_slab = [&](long m0) {
    const long first = m0 + 8 * rg;
    if (a.row_index) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ri = first + e < m_hi ? a.row_index[first + e] : -1;      // (wave-uniform)
        v[e] = ri >= 0 ? *reinterpret_cast<const float4 *>(src + (long)ri * ld) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      return;
    }
    long q = 0;
    int r = 0;
    if (mapR != mapG) { q = first / mapR; r = (int)(first - q * mapR); }      // one division per 8 rows, then a walked remainder
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const long m = first + e;
      const long row = mapR == mapG ? m + mapOff : q * mapG + mapOff + r;
      v[e] = m < m_hi ? *reinterpret_cast<const float4 *>(src + row * ld) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (++r == mapR) { r = 0; ++q; }
    }
  };
  auto store_slab = [&](unsigned short *smem) {
    if (is_grad) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { cs[0] += v[e].x; cs[1] += v[e].y; cs[2] += v[e].z; cs[3] += v[e].w; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      typedef __attribute__((ext_vector_type(4))) unsigned tn_u32x4;
      tn_u32x4 h, l;
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        const float x0 = (q == 0 ? v[2 * e2].x : q == 1 ? v[2 * e2].y : q == 2 ? v[2 * e2].z : v[2 * e2].w) * sc;
        const float x1 = (q == 0 ? v[2 * e2 + 1].x : q == 1 ? v[2 * e2 + 1].y : q == 2 ? v[2 * e2 + 1].z : v[2 * e2 + 1].w) * sc;
        unsigned hh, ll;
        split2_f16(x0, x1, hh, ll);
        h[e2] = hh; l[e2] = ll;
      }
      const int off = gemm_swz(ccol + q, 8 * rg);
      *reinterpret_cast<tn_u32x4 *>(smem + off) = h;
      *reinterpret_cast<tn_u32x4 *>(smem + PLANE + off) = l;
    }
  };
  // MFMA tiles of this wave: P tiles 4 wi .. 4 wi + 3 (rows of the block), Q tiles 8 wj .. 8 wj + 7 (columns)
  const int wi = wave & 3, wj = wave >> 2;
  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  load_slab(m_lo);
  store_slab(tn_smem);
  __syncthreads();
  if (m_lo + 32 < m_hi) load_slab(m_lo + 32);
  int buf = 0;
  for (long m0 = m_lo; m0 < m_hi; m0 += 32, buf ^= 1) {
    const unsigned short *smem = tn_smem + buf * 2 * PLANE;
    gemm_f16x8 ah[4], al[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int off = gemm_swz(64 * wi + 16 * i + fr, 8 * fg);
      ah[i] = *reinterpret_cast<const gemm_f16x8 *>(smem + off);
      al[i] = *reinterpret_cast<const gemm_f16x8 *>(smem + PLANE + off);
    }
    auto mul = [&](int j) {
      const int off = gemm_swz(TILE + 128 * wj + 16 * j + fr, 8 * fg);
      const gemm_f16x8 bh = *reinterpret_cast<const gemm_f16x8 *>(smem + off);
      const gemm_f16x8 bl = *reinterpret_cast<const gemm_f16x8 *>(smem + PLANE + off);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh, acc[i][j], 0, 0, 0);
      }
    };
#pragma unroll
    for (int j = 0; j < 2; ++j) mul(j);
    // the next slab (its loads were issued a slab ago) goes into the other image -- every wave left it at the last barrier --,
    // then the loads of the slab after it
    if (m0 + 32 < m_hi) store_slab(tn_smem + (buf ^ 1) * 2 * PLANE);
    if (m0 + 64 < m_hi) load_slab(m0 + 64);
#pragma unroll
    for (int j = 2; j < 8; ++j) mul(j);
    __syncthreads();
  }
  // every element of the block belongs to one lane of one wave: acc[i][j][r] is (a, b) = (64 wi + 16 i + 4 fg + r, 128 wj + 16 j + fr)
  float *o = a.out + ((long)ba * TILE + 64 * wi + 4 * fg) * a.sa + ((long)bb * TILE + 128 * wj + fr) * a.sb;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(o + (long)(16 * i + r) * a.sa + (long)(16 * j) * a.sb, acc[i][j][r] * xinv);
  // bias gradient: column sums of the gradient operand, once per row chunk (by the blocks with index 0 along the other operand)
  if (a.colsum && is_grad && (is_p ? bb == 0 : ba == 0)) {
    float *dst = a.colsum + (long)(is_p ? ba : bb) * TILE + c4;
#pragma unroll
    for (int q = 0; q < 4; ++q) atomicAdd(dst + q, cs[q]);
  }
}

// LayerNorm backward (post-norm block): y = LN(u) * w + b.
//   dU = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dY * w;  dw += dY * xhat;  db += dY
// One wave per row; per-workgroup partial dw/db through LDS, then one atomic per feature.
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float *__restrict__ dY,
                                                            const float *__restrict__ U,
                                                            const float *__restrict__ w, float *__restrict__ dU,
                                                            float *dw, float *db, long rows, int d,
                                                            int rows_per_block, unsigned *out_absmax = nullptr) {
  extern __shared__ float sm[];   // [2][d]
  float *sdw = sm, *sdb = sm + d;
  for (int i = threadIdx.x; i < 2 * d; i += 256) sm[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * rows_per_block;
  float pdw[8], pdb[8];
  unsigned omax = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { pdw[i] = 0.f; pdb[i] = 0.f; }
  for (long row = r0 + wave; row < min(rows, r0 + rows_per_block); row += 4) {
    float u[8], gy[8];
    float s = 0.f;
    int n = 0;
    for (int c = lane; c < d; c += 64, ++n) { u[n] = U[row * d + c]; gy[n] = dY[row * d + c]; s += u[n]; }
    const float mean = wave_sum(s) / d;
    float ss = 0.f;
    for (int i = 0; i < n; ++i) { const float t = u[i] - mean; ss += t * t; }
    const float rstd = rsqrtf(wave_sum(ss) / d + 1e-5f);
    float sg = 0.f, sgx = 0.f;
    n = 0;
    for (int c = lane; c < d; c += 64, ++n) {
      const float xh = (u[n] - mean) * rstd, g = gy[n] * w[c];
      pdw[n] += gy[n] * xh;
      pdb[n] += gy[n];
      u[n] = xh;
      gy[n] = g;
      sg += g;
      sgx += g * xh;
    }
    const float mg = wave_sum(sg) / d, mgx = wave_sum(sgx) / d;
    n = 0;
    for (int c = lane; c < d; c += 64, ++n) {
      const float du = rstd * (gy[n] - mg - u[n] * mgx);
      dU[row * d + c] = du;
      omax = max(omax, __float_as_uint(du) & 0x7fffffffu);
    }
  }
  if (out_absmax) {      // max |dU| of this launch: the scale of the F16X3 gradient products that read dU (gemm.h: xmax_bits)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) omax = max(omax, (unsigned)__shfl_xor((int)omax, o, 64));
    if (lane == 0 && omax) atomicMax(out_absmax, omax);
  }
  int n = 0;
  for (int c = lane; c < d; c += 64, ++n) { atomicAdd(&sdw[c], pdw[n]); atomicAdd(&sdb[c], pdb[n]); }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += 256) { atomicAdd(dw + c, sdw[c]); atomicAdd(db + c, sdb[c]); }
}

// Narrow-row variant of the above (d = 4 * LPR, see add_layernorm_narrow_kernel): 64 / LPR rows per wave step,
// dw / db partials in registers per lane (4 features), reduced over the workgroup's row lanes through LDS.
template <int LPR>
__global__ __launch_bounds__(256) void layernorm_bwd_narrow_kernel(const float *__restrict__ dY,
                                                                   const float *__restrict__ U,
                                                                   const float *__restrict__ w, float *__restrict__ dU,
                                                                   float *dw, float *db, long rows) {
  constexpr int d = 4 * LPR, RPB = 256 / LPR;
  __shared__ float red[2][RPB][d + 4];
  const int sub = threadIdx.x % LPR, rl = threadIdx.x / LPR;
  const float4 wv = *reinterpret_cast<const float4 *>(w + 4 * sub);
  float4 pdw = {0.f, 0.f, 0.f, 0.f}, pdb = {0.f, 0.f, 0.f, 0.f};
  for (long row = (long)blockIdx.x * RPB + rl; row < rows; row += (long)gridDim.x * RPB) {
    float4 u = *reinterpret_cast<const float4 *>(U + row * d + 4 * sub);
    const float4 gy = *reinterpret_cast<const float4 *>(dY + row * d + 4 * sub);
    const float mean = row_sum<LPR>(u.x + u.y + u.z + u.w) * (1.f / d);
    u.x -= mean; u.y -= mean; u.z -= mean; u.w -= mean;
    const float rstd = rsqrtf(row_sum<LPR>(u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w) * (1.f / d) + 1e-5f);
    const float4 xh = {u.x * rstd, u.y * rstd, u.z * rstd, u.w * rstd};
    const float4 g = {gy.x * wv.x, gy.y * wv.y, gy.z * wv.z, gy.w * wv.w};
    pdw.x += gy.x * xh.x; pdw.y += gy.y * xh.y; pdw.z += gy.z * xh.z; pdw.w += gy.w * xh.w;
    pdb.x += gy.x; pdb.y += gy.y; pdb.z += gy.z; pdb.w += gy.w;
    const float mg = row_sum<LPR>(g.x + g.y + g.z + g.w) * (1.f / d);
    const float mgx = row_sum<LPR>(g.x * xh.x + g.y * xh.y + g.z * xh.z + g.w * xh.w) * (1.f / d);
    const float4 o = {rstd * (g.x - mg - xh.x * mgx), rstd * (g.y - mg - xh.y * mgx), rstd * (g.z - mg - xh.z * mgx),
                      rstd * (g.w - mg - xh.w * mgx)};
    *reinterpret_cast<float4 *>(dU + row * d + 4 * sub) = o;
  }
  *reinterpret_cast<float4 *>(&red[0][rl][4 * sub]) = pdw;
  *reinterpret_cast<float4 *>(&red[1][rl][4 * sub]) = pdb;
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * d; e += 256) {
    const int which = e / d, c = e % d;
    float s = 0.f;
    for (int r = 0; r < RPB; ++r) s += red[which][r][c];
    atomicAdd((which ? db : dw) + c, s);
  }
}

// elementwise helpers
__global__ void add_inplace_kernel(float *__restrict__ a, const float *__restrict__ b, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] += b[i];
}
// time token (model/head.py:342-345): the acquisition MLP sees [z | t]; t of instance i (= step t0 + i / B of the rollout) is
// step / T, or (T - step) / T for the evaluation schedule (utils/eval.py:24)
__global__ void time_vec_kernel(float *__restrict__ tv, int I, int B, int t0, int TT, int reverse) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= I) return;
  const int step = t0 + i / B;
  tv[i] = reverse ? (float)(TT - step) / (float)TT : (float)step / (float)TT;
}
// gradient of the time column of the acquisition head's first layer: dW1[f, d] += sum_rows dHid[row, f] * t(row / rows_per_inst)
__global__ __launch_bounds__(256) void time_col_grad_kernel(const float *__restrict__ dHid, int F, long rows, int rows_per_inst,
                                                            const float *__restrict__ tv, float *__restrict__ dW1, int ldw, int col,
                                                            long rows_per_block) {
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (int f = threadIdx.x; f < F; f += 256) {
    float acc = 0.f;
    for (long r = r0; r < r1; ++r) acc = fmaf(dHid[r * F + f], tv[r / rows_per_inst], acc);
    atomicAdd(dW1 + (long)f * ldw + col, acc);
  }
}

__global__ void transpose_kernel(const float *__restrict__ src, int rows, int cols, int ld, float *__restrict__ dst) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // dst [cols, rows]
  if (i >= (long)rows * cols) return;
  const int c = i / rows, r = i % rows;
  dst[i] = src[(long)r * ld + c];
}

// Masked set-attention backward.  One workgroup per (instance, head), like attention_kernel.
//   dV_j = sum_i P_ij dO_i;  dS_ij = P_ij (dO_i.V_j - delta_i),  delta_i = sum_j P_ij dO_i.V_j;
//   dQ_i = scale * sum_j dS_ij K_j;  dK_j = scale * sum_i dS_ij Q_i
// Phase 1 (threads = token rows): softmax statistics, delta and dQ.  Phase 2 (threads = (key, channel
// pair)): every key row sums its own dK / dV over the token rows from the statistics kept in LDS -- no
// atomics.  dQKV rows of non-key tokens get zero K/V gradients.
template <int HD, bool REUSE = false>
__global__ __launch_bounds__(HD <= 16 ? 512 : 256) void attention_bwd_kernel(Geo g, int d, const float *__restrict__ QKV,
                                                            const float *__restrict__ dA,
                                                            float *__restrict__ dQKV, int max_keys,
                                                            const float *__restrict__ Aout = nullptr) {
  // Aout (the attention output the backward keeps for the out-projection gradient, or null): delta_i = dO_i . O_i is then known
  // before the key loop of phase 1, and dQ needs one accumulator per channel instead of two (sum e dp k and sum e k)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float *Ks = reinterpret_cast<float *>(smem_raw);                 // [max_keys][HD]
  float *Vs = Ks + (size_t)max_keys * HD;                          // [max_keys][HD]
  float *Qs = Vs + (size_t)max_keys * HD;                          // [N][HD]  scaled queries
  float *Gs = Qs + (size_t)g.N * HD;                               // [N][HD]  dO
  float *St = Gs + (size_t)g.N * HD;                               // [N][4]   max, 1/l, delta, #keys
  // dK / dV sums of the second phase.  REUSE (the launcher picks it at head_dim <= 16 while max_keys <= blockDim.x: a thread has one work item
  // then and copies its key's K / V row into registers first): they take the place of Ks / Vs -- 10 KB less LDS at the cfg3 shape, four
  // workgroups per CU instead of three (the kernel is latency-bound: profiles/r03_attention_bwd_hd8_cfg3_pmc.txt); otherwise their own
  // [2][max_keys][HD] block
  constexpr bool reuse = REUSE;
  float *dKx = St + (size_t)g.N * 4;
  float *dKs = reuse ? Ks : dKx, *dVs = reuse ? Vs : dKx + (size_t)max_keys * HD;
  int *keyrow = reinterpret_cast<int *>(dKx + (reuse ? 0 : (size_t)max_keys * 2 * HD));     // [max_keys]
  __shared__ int wave_cnt[8];
  __shared__ int s_base;
  // one workgroup per instance, the heads one after the other: the key list is built once and the four 32-byte head
  // slices of a QKV row are read by the same CU back to back (a workgroup per (instance, head) spent most of its time
  // in the prologue: 60 000 workgroups of ~5 us of arithmetic per call)
  const int H = d / HD, b = blockIdx.x;
  if (b >= g.B) return;
  // NT threads: the launcher rounds the token count up to whole waves (<= 512 at head_dim <= 16) -- phase 1 is a thread per token row,
  // phase 2 a thread per (key, row slice): 304 rows / 150 keys on 256 threads left 41 % of the lanes idle in both (cfg3)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NT = blockDim.x, NWV = NT >> 6;
  const int n_t = g.n_td + g.n_th;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < g.N; c0 += NT) {
    int row = c0 + tid;
    bool key = false;
    if (row < g.P) key = is_ctx(g, b, row);
    unsigned long long bal = __ballot(key);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (key) keyrow[off + __popcll(bal & ((1ull << lane) - 1ull))] = row;
    __syncthreads();
    if (tid == 0) { int tot = 0; for (int w = 0; w < NWV; ++w) tot += wave_cnt[w]; s_base += tot; }
    __syncthreads();
  }
  const int n_ck = s_base;
  __syncthreads();
  if (tid == 0) {
    int n = n_ck;
    for (int j = 0; j < n_t; ++j)
      if (!g.tmask || g.tmask[j]) keyrow[n++] = g.P + j;
    s_base = n;
  }
  __syncthreads();
  const int n_ak = s_base;
  const long ep = (long)b * g.N;
  for (int h = 0; h < H; ++h) {
  for (int i = tid; i < n_ak * HD; i += NT) {
    int j = i / HD, c = i % HD;
    const float *src = QKV + (ep + keyrow[j]) * 3 * d + h * HD + c;
    Ks[j * HD + c] = src[d];
    Vs[j * HD + c] = src[2 * d];
    if (!reuse) { dKs[j * HD + c] = 0.f; dVs[j * HD + c] = 0.f; }
  }
  __syncthreads();
  const float scale = rsqrtf((float)HD);
  for (int row = tid; row < g.N; row += NT) {
    const bool isq = row < g.P && !is_ctx(g, b, row);
    const int nk = isq ? n_ak : n_ck;
    float q[HD], go[HD], dq[HD];
    const float *qp = QKV + (ep + row) * 3 * d + h * HD;
    const float *gp = dA + (ep + row) * d + h * HD;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {          // rows are 16-byte aligned (d, HD multiples of 4): 128-bit moves
      const float4 qv = *reinterpret_cast<const float4 *>(qp + c), gv = *reinterpret_cast<const float4 *>(gp + c);
      q[c] = qv.x * scale; q[c + 1] = qv.y * scale; q[c + 2] = qv.z * scale; q[c + 3] = qv.w * scale;
      go[c] = gv.x; go[c + 1] = gv.y; go[c + 2] = gv.z; go[c + 3] = gv.w;
      dq[c] = dq[c + 1] = dq[c + 2] = dq[c + 3] = 0.f;
      *reinterpret_cast<float4 *>(Qs + row * HD + c) = make_float4(q[c], q[c + 1], q[c + 2], q[c + 3]);
      *reinterpret_cast<float4 *>(Gs + row * HD + c) = gv;
    }
    float mx = -INFINITY;
    for (int j = 0; j < nk; ++j) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) s = fmaf(q[c], Ks[j * HD + c], s);
      mx = fmaxf(mx, s);
    }
    float l = 0.f, delta = 0.f, inv;
    if (Aout) {
      // dq = sum_j p_j (dp_j - delta) k_j = (sum_j e_j (dp_j - delta) k_j) / l with delta = dO . O
      const float *ap = Aout + (ep + row) * d + h * HD;
#pragma unroll
      for (int c = 0; c < HD; ++c) delta = fmaf(go[c], ap[c], delta);
      for (int j = 0; j < nk; ++j) {
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) { s = fmaf(q[c], Ks[j * HD + c], s); dp = fmaf(go[c], Vs[j * HD + c], dp); }
        const float e = __expf(s - mx), w = e * (dp - delta);
        l += e;
#pragma unroll
        for (int c = 0; c < HD; ++c) dq[c] = fmaf(w, Ks[j * HD + c], dq[c]);
      }
      inv = 1.f / l;
#pragma unroll
      for (int c = 0; c < HD; ++c) dq[c] *= inv;
    } else {
      // dq = sum_j p_j (dp_j - delta) k_j = (sum_j e_j dp_j k_j - delta sum_j e_j k_j) / l  in one pass over the keys
      float pk[HD];
#pragma unroll
      for (int c = 0; c < HD; ++c) pk[c] = 0.f;
      for (int j = 0; j < nk; ++j) {
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) { s = fmaf(q[c], Ks[j * HD + c], s); dp = fmaf(go[c], Vs[j * HD + c], dp); }
        const float e = __expf(s - mx), edp = e * dp;
        l += e;
        delta += edp;
#pragma unroll
        for (int c = 0; c < HD; ++c) { dq[c] = fmaf(edp, Ks[j * HD + c], dq[c]); pk[c] = fmaf(e, Ks[j * HD + c], pk[c]); }
      }
      inv = 1.f / l;
      delta *= inv;
#pragma unroll
      for (int c = 0; c < HD; ++c) dq[c] = (dq[c] - delta * pk[c]) * inv;
    }
    *reinterpret_cast<float4 *>(St + row * 4) = make_float4(mx, inv, delta, (float)nk);
    float *out = dQKV + (ep + row) * 3 * d + h * HD;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      *reinterpret_cast<float4 *>(out + c) = make_float4(dq[c] * scale, dq[c + 1] * scale, dq[c + 2] * scale, dq[c + 3] * scale);
      *reinterpret_cast<float4 *>(out + d + c) = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4 *>(out + 2 * d + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
  // phase 2: thread = (key j, slice of the token rows): softmax weight and score gradient once per (row, key),
  // partial dK / dV rows in registers, summed per key through LDS
  const int nsl = max(1, NT / max(n_ak, 1));
  float kj[HD], vj[HD];
  if (reuse) {        // (one work item per thread) K / V rows into registers, then Ks / Vs become the dK / dV sums
    const int j0 = tid < n_ak * nsl ? tid % n_ak : 0;
#pragma unroll
    for (int c = 0; c < HD; ++c) { kj[c] = Ks[j0 * HD + c]; vj[c] = Vs[j0 * HD + c]; }
    __syncthreads();
    for (int i = tid; i < n_ak * HD; i += NT) { Ks[i] = 0.f; Vs[i] = 0.f; }
    __syncthreads();
  }
  for (int w = tid; w < n_ak * nsl; w += NT) {
    const int j = w % n_ak, sl = w / n_ak;
    float dk[HD], dv[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) {
      if (!reuse) { kj[c] = Ks[j * HD + c]; vj[c] = Vs[j * HD + c]; }
      dk[c] = 0.f; dv[c] = 0.f;
    }
    for (int row = sl; row < g.N; row += nsl) {
      if ((float)j >= St[row * 4 + 3]) continue;      // key not visible to this row
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) { s = fmaf(Qs[row * HD + c], kj[c], s); dp = fmaf(Gs[row * HD + c], vj[c], dp); }
      const float p = __expf(s - St[row * 4 + 0]) * St[row * 4 + 1];
      const float ds = p * (dp - St[row * 4 + 2]);
#pragma unroll
      for (int c = 0; c < HD; ++c) { dv[c] = fmaf(p, Gs[row * HD + c], dv[c]); dk[c] = fmaf(ds, Qs[row * HD + c], dk[c]); }
    }
#pragma unroll
    for (int c = 0; c < HD; ++c) { atomicAdd(&dKs[j * HD + c], dk[c]); atomicAdd(&dVs[j * HD + c], dv[c]); }
  }
  __syncthreads();
  for (int e = tid; e < n_ak * HD; e += NT) {
    const int j = e / HD, c = e % HD;
    float *dst = dQKV + (ep + keyrow[j]) * 3 * d + h * HD + c;
    dst[d] = dKs[j * HD + c];            // Qs already carries the 1/sqrt(hd)
    dst[2 * d] = dVs[j * HD + c];
  }
  __syncthreads();     // the staging arrays are reused by the next head
  }   // heads
}

// Acquisition head backward, first part (model/head.py:27-33 + log-softmax of the chosen design):
//   logits = hid . w2 + b2 over the point rows; p = softmax over the remaining queries;
//   dlogit[row] = g_logp * (1[row == chosen] - p[row]);  dhid = dlogit * w2 * (hid > 0);
//   dw2 += sum dlogit * hid;  db2 += sum dlogit.          One workgroup per instance; dhid overwrites hid.
struct AcqBwdArgs {
  Geo g; int F;
  float *hid;                  // [I*P, F] in: relu(z W1^T + b1), out: gradient wrt the pre-activation
  const float *w2, *b2;
  const float *g_logp;         // [B, T] dLoss/dlog_prob
  const int *slot;             // [B, T] chosen slot
  int T;
  float *dw2, *db2;
  unsigned *out_absmax;        // optional: max |gradient written to hid| as bits (the scale word of the f16 gradient products that read it)
};
__global__ __launch_bounds__(256) void acq_bwd_kernel(AcqBwdArgs a) {
  extern __shared__ float lds[];   // logits [P], dw2 partial [F]
  float *logit = lds, *sdw = lds + a.g.P;
  __shared__ float red[4];
  const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = a.g.P, F = a.F;
  const int b = i % a.g.inst_B, t = a.g.inst_t0 + i / a.g.inst_B;
  for (int f = tid; f < F; f += 256) sdw[f] = 0.f;
  for (int p = wave; p < P; p += 4) {
    const float *hp = a.hid + ((long)i * P + p) * F;
    float s = 0.f;
    for (int f = lane; f < F; f += 64) s = fmaf(hp[f], a.w2[f], s);
    s = wave_sum(s);
    if (lane == 0) logit[p] = s + a.b2[0];
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int p = tid; p < P; p += 256) if (!is_ctx(a.g, i, p)) mx = fmaxf(mx, logit[p]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int p = tid; p < P; p += 256) if (!is_ctx(a.g, i, p)) sum += __expf(logit[p] - mx);
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
  const float gl = a.g_logp[(long)b * a.T + t];
  const int chosen = a.slot[(long)b * a.T + t];
  __syncthreads();
  float dbl = 0.f;
  for (int p = tid; p < P; p += 256) {
    float dl = 0.f;
    if (!is_ctx(a.g, i, p)) dl = gl * ((p == chosen ? 1.f : 0.f) - __expf(logit[p] - mx) * inv);
    logit[p] = dl;     // each p is owned by one thread
    dbl += dl;
  }
  dbl = wave_sum(dbl);
  if (lane == 0) atomicAdd(a.db2, dbl);
  __syncthreads();
  // lane owns features f0 + lane + 64 n of one block of 512 features at a time.  (Rounds 1-3 handled the first block only: at
  // F = 1024 -- the d = 256 roofline variant -- the hidden units 512.. kept their FORWARD values as "gradient"; found by the
  // round-4 reference-autograd fixture grad_cfg2_d256.)
  float gmax = 0.f;
  for (int f0 = 0; f0 < F; f0 += 512) {
    float pw[8], w2r[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) { pw[n] = 0.f; w2r[n] = f0 + lane + 64 * n < F ? a.w2[f0 + lane + 64 * n] : 0.f; }
    for (int p = wave; p < P; p += 4) {
      float *hp = a.hid + ((long)i * P + p) * F;
      const float dl = logit[p];
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        const int f = f0 + lane + 64 * n;
        if (f < F) {
          const float hv = hp[f];
          pw[n] = fmaf(dl, hv, pw[n]);
          const float gv = hv > 0.f ? dl * w2r[n] : 0.f;
          hp[f] = gv;
          gmax = fmaxf(gmax, fabsf(gv));
        }
      }
    }
#pragma unroll
    for (int n = 0; n < 8; ++n)
      if (f0 + lane + 64 * n < F) atomicAdd(&sdw[f0 + lane + 64 * n], pw[n]);
  }
  if (a.out_absmax) {
    gmax = wave_max(gmax);
    if (lane == 0 && gmax > 0.f) atomicMax(a.out_absmax, __float_as_uint(gmax));
  }
  __syncthreads();
  for (int f = tid; f < F; f += 256) atomicAdd(a.dw2 + f, sdw[f]);
}

// GMM head backward, second layer + log-likelihood (model/head.py:172-177, utils/eval.py:200-207).
// One wave per target row; lane c = component.  hid [rows, C*F] (ReLU output) is overwritten by the
// gradient wrt the first layer's pre-activation; dw2/db2 accumulate with atomics.
struct GmmBwdArgs {
  float *hid; long rows; int C, F;
  const float *w2[16]; const float *b2[16];
  float *dw2[16]; float *db2[16];
  float std_min;
  const float *value; long value_mod;          // target value of row r: value[r % value_mod]
  const float *g_ll;                           // [rows] dLoss/d ll, or null
  const float *g_mean, *g_std, *g_wgt;         // [rows, C] dLoss/d mixture_{means,stds,weights}, or null
};
constexpr int GMM_BWD_ROWS = 256;  // rows per workgroup (64 per wave): weight-gradient partials stay in registers, are summed over the
                                   // waves in LDS and leave as one atomic per element and workgroup.  (1.65 ms per call at the headline shape,
                                   // 60 000 rows: neither the atomics nor the cross-lane reductions -- batching the 3 C reductions of a row made
                                   // it 2.0 ms; the kernel holds 424 registers and spills 236 SGPRs on its 4 x 16 pointer arguments.)
__global__ __launch_bounds__(256) void gmm_bwd_kernel(GmmBwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long r_lo = (long)blockIdx.x * GMM_BWD_ROWS, r_hi = min(a.rows, r_lo + GMM_BWD_ROWS);
  // lane owns hidden units lane, lane + 64 (F <= 128 here; wider heads fall back to per-row atomics below)
  const bool wide_f = a.F > 128;
  float pw[16][3][2];
#pragma unroll
  for (int c = 0; c < 16; ++c)
#pragma unroll
    for (int o = 0; o < 3; ++o) pw[c][o][0] = pw[c][o][1] = 0.f;
  float pb0 = 0.f, pb1 = 0.f, pb2 = 0.f;      // lane c: bias gradients of component c
  for (long row = r_lo + wave; row < r_hi; row += 4) {
    float raw0 = 0.f, raw1 = 0.f, raw2 = 0.f;
    for (int c = 0; c < a.C; ++c) {
      const float *hp = a.hid + (row * a.C + c) * a.F;
      const float *w = a.w2[c];
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
      for (int f = lane; f < a.F; f += 64) {
        const float hv = hp[f];
        s0 = fmaf(hv, w[f], s0); s1 = fmaf(hv, w[a.F + f], s1); s2 = fmaf(hv, w[2 * a.F + f], s2);
      }
      s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
      if (lane == c) { raw0 = s0 + a.b2[c][0]; raw1 = s1 + a.b2[c][1]; raw2 = s2 + a.b2[c][2]; }
    }
    const bool act = lane < a.C;
    const float mean = raw0, sd = softplus_f(raw1) + a.std_min;
    const float mxw = wave_max(act ? raw2 : -INFINITY);
    const float ew = act ? __expf(raw2 - mxw) : 0.f;
    const float wgt = ew / wave_sum(ew);
    const float v = a.value[row % a.value_mod];
    const float z = (v - mean) / sd;
    const float lp = act ? (-0.5f * z * z - logf(sd) - 0.91893853320467274178f + logf(wgt)) : -INFINITY;
    const float m2 = wave_max(lp);
    const float er = act ? __expf(lp - m2) : 0.f;
    const float resp = er / wave_sum(er);          // responsibilities
    const float gl = a.g_ll ? a.g_ll[row] : 0.f;
    // d ll / d raw
    float d0 = act ? gl * resp * z / sd : 0.f;                                        // mean
    float dsd = act ? gl * resp * (z * z - 1.f) / sd : 0.f;                           // sigma
    float d2 = act ? gl * (resp - wgt) : 0.f;                                         // mixture logits
    // direct upstream gradients of the GMM parameters (autograd through the caller's own compute_ll)
    if (a.g_mean && act) d0 += a.g_mean[row * a.C + lane];
    if (a.g_std && act) dsd += a.g_std[row * a.C + lane];
    if (a.g_wgt) {
      const float gw = act ? a.g_wgt[row * a.C + lane] : 0.f;
      const float dot = wave_sum(gw * wgt);
      if (act) d2 += wgt * (gw - dot);                                                // softmax backward
    }
    const float d1 = dsd * (1.f / (1.f + __expf(-raw1)));                             // softplus'
    pb0 += d0; pb1 += d1; pb2 += d2;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (c < a.C) {
        const float g0 = __shfl(d0, c, 64), g1 = __shfl(d1, c, 64), g2 = __shfl(d2, c, 64);
        float *hp = a.hid + (row * a.C + c) * a.F;
        const float *w = a.w2[c];
        if (wide_f) {
          for (int f = lane; f < a.F; f += 64) {
            const float hv = hp[f];
            if (hv > 0.f) {
              atomicAdd(a.dw2[c] + f, g0 * hv);
              atomicAdd(a.dw2[c] + a.F + f, g1 * hv);
              atomicAdd(a.dw2[c] + 2 * a.F + f, g2 * hv);
              hp[f] = g0 * w[f] + g1 * w[a.F + f] + g2 * w[2 * a.F + f];
            } else {
              hp[f] = 0.f;
            }
          }
        } else {
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            const int f = lane + 64 * n;
            if (f < a.F) {
              const float hv = hp[f];
              if (hv > 0.f) {
                pw[c][0][n] = fmaf(g0, hv, pw[c][0][n]);
                pw[c][1][n] = fmaf(g1, hv, pw[c][1][n]);
                pw[c][2][n] = fmaf(g2, hv, pw[c][2][n]);
                hp[f] = g0 * w[f] + g1 * w[a.F + f] + g2 * w[2 * a.F + f];
              } else {
                hp[f] = 0.f;
              }
            }
          }
        }
      }
    }
  }
  __shared__ float red[16 * 3 * 128 + 16 * 3];
  for (int e = threadIdx.x; e < 16 * 3 * 128 + 16 * 3; e += 256) red[e] = 0.f;
  __syncthreads();
  if (!wide_f) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int f = lane + 64 * n;
        if (c < a.C && f < a.F) {
#pragma unroll
          for (int o = 0; o < 3; ++o) atomicAdd(&red[(c * 3 + o) * 128 + f], pw[c][o][n]);
        }
      }
    }
  }
  if (lane < a.C) {
    atomicAdd(&red[16 * 3 * 128 + lane * 3 + 0], pb0); atomicAdd(&red[16 * 3 * 128 + lane * 3 + 1], pb1);
    atomicAdd(&red[16 * 3 * 128 + lane * 3 + 2], pb2);
  }
  __syncthreads();
  if (!wide_f)
    for (int e = threadIdx.x; e < a.C * 3 * 128; e += 256) {
      const int c = e / 384, o = (e / 128) % 3, f = e & 127;
      if (f < a.F) atomicAdd(a.dw2[c] + o * a.F + f, red[e]);
    }
  if (threadIdx.x < a.C * 3) atomicAdd(a.db2[threadIdx.x / 3] + threadIdx.x % 3, red[16 * 3 * 128 + threadIdx.x]);
}

// The same for F > 128 (the d = 256 / 512 models: F = 1024 ...), where gmm_bwd_kernel falls back to one global atomic per (row, component,
// hidden unit, output) -- 1.8e9 atomics and 15 ms per call at d = 256 / F = 1024 / 60 000 rows, 28 % of that model's training step.
// Two passes per workgroup of 64 rows: (1) a wave per row: raw head outputs, responsibilities, d ll / d raw of every component -> LDS;
// (2) a thread per hidden unit (f = tid, tid + 256, ...): for every component, the weight-gradient partials of its units summed over the
// rows in registers (one atomic per element and workgroup) and the hidden gradient written in place -- hidden rows are read coalesced.
constexpr int GMM_WIDE_ROWS = 64;
__global__ __launch_bounds__(256) void gmm_bwd_wide_kernel(GmmBwdArgs a) {
  __shared__ float dsh[GMM_WIDE_ROWS][3][16];
  __shared__ float pbs[16 * 3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long r_lo = (long)blockIdx.x * GMM_WIDE_ROWS, r_hi = min(a.rows, r_lo + GMM_WIDE_ROWS);
  if (threadIdx.x < 48) pbs[threadIdx.x] = 0.f;
  __syncthreads();
  float pb0 = 0.f, pb1 = 0.f, pb2 = 0.f;      // lane c: bias gradients of component c
  for (long row = r_lo + wave; row < r_hi; row += 4) {
    float raw0 = 0.f, raw1 = 0.f, raw2 = 0.f;
    for (int c = 0; c < a.C; ++c) {
      const float *hp = a.hid + (row * a.C + c) * a.F;
      const float *w = a.w2[c];
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
      for (int f = lane; f < a.F; f += 64) {
        const float hv = hp[f];
        s0 = fmaf(hv, w[f], s0); s1 = fmaf(hv, w[a.F + f], s1); s2 = fmaf(hv, w[2 * a.F + f], s2);
      }
      s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
      if (lane == c) { raw0 = s0 + a.b2[c][0]; raw1 = s1 + a.b2[c][1]; raw2 = s2 + a.b2[c][2]; }
    }
    const bool act = lane < a.C;
    const float mean = raw0, sd = softplus_f(raw1) + a.std_min;
    const float mxw = wave_max(act ? raw2 : -INFINITY);
    const float ew = act ? __expf(raw2 - mxw) : 0.f;
    const float wgt = ew / wave_sum(ew);
    const float v = a.value[row % a.value_mod];
    const float z = (v - mean) / sd;
    const float lp = act ? (-0.5f * z * z - logf(sd) - 0.91893853320467274178f + logf(wgt)) : -INFINITY;
    const float m2 = wave_max(lp);
    const float er = act ? __expf(lp - m2) : 0.f;
    const float resp = er / wave_sum(er);
    const float gl = a.g_ll ? a.g_ll[row] : 0.f;
    float d0 = act ? gl * resp * z / sd : 0.f;
    float dsd = act ? gl * resp * (z * z - 1.f) / sd : 0.f;
    float d2 = act ? gl * (resp - wgt) : 0.f;
    if (a.g_mean && act) d0 += a.g_mean[row * a.C + lane];
    if (a.g_std && act) dsd += a.g_std[row * a.C + lane];
    if (a.g_wgt) {
      const float gw = act ? a.g_wgt[row * a.C + lane] : 0.f;
      const float dot = wave_sum(gw * wgt);
      if (act) d2 += wgt * (gw - dot);
    }
    const float d1 = dsd * (1.f / (1.f + __expf(-raw1)));
    pb0 += d0; pb1 += d1; pb2 += d2;
    if (lane < 16) { dsh[row - r_lo][0][lane] = d0; dsh[row - r_lo][1][lane] = d1; dsh[row - r_lo][2][lane] = d2; }
  }
  if (lane < a.C) { atomicAdd(&pbs[lane * 3 + 0], pb0); atomicAdd(&pbs[lane * 3 + 1], pb1); atomicAdd(&pbs[lane * 3 + 2], pb2); }
  __syncthreads();
  const int nrow = (int)(r_hi - r_lo);
  for (int c = 0; c < a.C; ++c) {
    const float *w = a.w2[c];
    for (int f = threadIdx.x; f < a.F; f += 256) {
      const float w0 = w[f], w1 = w[a.F + f], w2 = w[2 * a.F + f];
      float p0 = 0.f, p1 = 0.f, p2 = 0.f;
      float *hp = a.hid + ((long)r_lo * a.C + c) * a.F + f;
      for (int rr = 0; rr < nrow; ++rr, hp += (long)a.C * a.F) {
        const float hv = *hp, g0 = dsh[rr][0][c], g1 = dsh[rr][1][c], g2 = dsh[rr][2][c];
        if (hv > 0.f) {
          p0 = fmaf(g0, hv, p0); p1 = fmaf(g1, hv, p1); p2 = fmaf(g2, hv, p2);
          *hp = g0 * w0 + g1 * w1 + g2 * w2;
        } else {
          *hp = 0.f;
        }
      }
      atomicAdd(a.dw2[c] + f, p0); atomicAdd(a.dw2[c] + a.F + f, p1); atomicAdd(a.dw2[c] + 2 * a.F + f, p2);
    }
  }
  if (threadIdx.x < a.C * 3) atomicAdd(a.db2[threadIdx.x / 3] + threadIdx.x % 3, pbs[threadIdx.x]);
}

// The same for F = 128 (lane owns hidden units lane, lane + 64), written around the latency of a row: gmm_bwd_kernel walks the
// components of a row one after the other, every one with its own global loads (twice) and its own cross-lane reductions:
// ~25 us per row on the one wave per SIMD its 424 registers allow (1.65 ms per call at the headline shape, VALU busy 9 %).
// Here the second-layer weights sit in LDS, the 2 C hidden values of a row are loaded in one batch (the next row's while this
// one is computed) and stay in registers for both passes, and the 3 C reductions of a row run side by side.
// (CMAX = 10 or 16 components: the register arrays are sized by it -- 218 instead of 296 registers, two waves per SIMD)
template <int CMAX, int ROWS>
__global__ __launch_bounds__(256, CMAX <= 10 ? 2 : 1) void gmm_bwd128_kernel(GmmBwdArgs a) {
  constexpr int F = 128;
  __shared__ float sw[16 * 3 * F + 16 * 4];       // w2 [C][3][F], b2 [C][4]; reused for the gradient sums at the end
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long r_lo = (long)blockIdx.x * ROWS, r_hi = min(a.rows, r_lo + ROWS);
  for (int e = threadIdx.x; e < a.C * 3 * F; e += 256) sw[e] = a.w2[e / (3 * F)][e % (3 * F)];
  if (threadIdx.x < a.C * 3) sw[16 * 3 * F + (threadIdx.x / 3) * 4 + threadIdx.x % 3] = a.b2[threadIdx.x / 3][threadIdx.x % 3];
  __syncthreads();
  float pw[CMAX][3][2];
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
#pragma unroll
    for (int o = 0; o < 3; ++o) pw[c][o][0] = pw[c][o][1] = 0.f;
  float pb0 = 0.f, pb1 = 0.f, pb2 = 0.f;      // lane c: bias gradients of component c
  float nh[CMAX][2];
  auto load_row = [&](long row) {
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < a.C) { const float *hp = a.hid + (row * a.C + c) * F; nh[c][0] = hp[lane]; nh[c][1] = hp[lane + 64]; }
  };
  if (r_lo + wave < r_hi) load_row(r_lo + wave);
  for (long row = r_lo + wave; row < r_hi; row += 4) {
    float hv[CMAX][2], ps[CMAX][3];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) { hv[c][0] = nh[c][0]; hv[c][1] = nh[c][1]; }
    if (row + 4 < r_hi) load_row(row + 4);
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < a.C) {
#pragma unroll
        for (int o = 0; o < 3; ++o) ps[c][o] = fmaf(hv[c][0], sw[(c * 3 + o) * F + lane], hv[c][1] * sw[(c * 3 + o) * F + lane + 64]);
      }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < a.C) {
#pragma unroll
          for (int o = 0; o < 3; ++o) ps[c][o] += __shfl_xor(ps[c][o], off, WAVE);
        }
    float raw0 = 0.f, raw1 = 0.f, raw2 = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < a.C && lane == c) {
        raw0 = ps[c][0] + sw[16 * 3 * F + c * 4]; raw1 = ps[c][1] + sw[16 * 3 * F + c * 4 + 1]; raw2 = ps[c][2] + sw[16 * 3 * F + c * 4 + 2];
      }
    const bool act = lane < a.C;
    const float mean = raw0, sd = softplus_f(raw1) + a.std_min;
    const float mxw = wave_max(act ? raw2 : -INFINITY);
    const float ew = act ? __expf(raw2 - mxw) : 0.f;
    const float wgt = ew / wave_sum(ew);
    const float v = a.value[row % a.value_mod];
    const float z = (v - mean) / sd;
    const float lp = act ? (-0.5f * z * z - logf(sd) - 0.91893853320467274178f + logf(wgt)) : -INFINITY;
    const float m2 = wave_max(lp);
    const float er = act ? __expf(lp - m2) : 0.f;
    const float resp = er / wave_sum(er);          // responsibilities
    const float gl = a.g_ll ? a.g_ll[row] : 0.f;
    float d0 = act ? gl * resp * z / sd : 0.f;                                        // mean
    float dsd = act ? gl * resp * (z * z - 1.f) / sd : 0.f;                           // sigma
    float d2 = act ? gl * (resp - wgt) : 0.f;                                         // mixture logits
    if (a.g_mean && act) d0 += a.g_mean[row * a.C + lane];
    if (a.g_std && act) dsd += a.g_std[row * a.C + lane];
    if (a.g_wgt) {
      const float gw = act ? a.g_wgt[row * a.C + lane] : 0.f;
      const float dot = wave_sum(gw * wgt);
      if (act) d2 += wgt * (gw - dot);                                                // softmax backward
    }
    const float d1 = dsd * (1.f / (1.f + __expf(-raw1)));                             // softplus'
    pb0 += d0; pb1 += d1; pb2 += d2;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < a.C) {
        const float g0 = __shfl(d0, c, WAVE), g1 = __shfl(d1, c, WAVE), g2 = __shfl(d2, c, WAVE);
        float *hp = a.hid + (row * a.C + c) * F;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int f = lane + 64 * n;
          const float h = hv[c][n];
          pw[c][0][n] = fmaf(g0, h, pw[c][0][n]);      // (h = relu output: 0 on the inactive units)
          pw[c][1][n] = fmaf(g1, h, pw[c][1][n]);
          pw[c][2][n] = fmaf(g2, h, pw[c][2][n]);
          hp[f] = h > 0.f ? g0 * sw[(c * 3) * F + f] + g1 * sw[(c * 3 + 1) * F + f] + g2 * sw[(c * 3 + 2) * F + f] : 0.f;
        }
      }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 16 * 3 * F + 16 * 4; e += 256) sw[e] = 0.f;
  __syncthreads();
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
    if (c < a.C) {
#pragma unroll
      for (int o = 0; o < 3; ++o) { atomicAdd(&sw[(c * 3 + o) * F + lane], pw[c][o][0]); atomicAdd(&sw[(c * 3 + o) * F + lane + 64], pw[c][o][1]); }
    }
  if (lane < a.C) { atomicAdd(&sw[16 * 3 * F + lane * 4], pb0); atomicAdd(&sw[16 * 3 * F + lane * 4 + 1], pb1); atomicAdd(&sw[16 * 3 * F + lane * 4 + 2], pb2); }
  __syncthreads();
  for (int e = threadIdx.x; e < a.C * 3 * F; e += 256) atomicAdd(a.dw2[e / (3 * F)] + e % (3 * F), sw[e]);
  if (threadIdx.x < a.C * 3) atomicAdd(a.db2[threadIdx.x / 3] + threadIdx.x % 3, sw[16 * 3 * F + (threadIdx.x / 3) * 4 + threadIdx.x % 3]);
}

// Wcat[i][c F + f] = W1_c[f][i]: the C first-layer weights [F, d] of the GMM heads as ONE [d, C F] operand, so that
// dz = sum_c dhid_c W1_c is one GEMM with K = C F instead of C accumulating ones
struct PackW1Args { const float *w1[16]; int C, F, d; float *out; };
__global__ void gmm_w1_pack_kernel(PackW1Args a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.C * a.F * a.d) return;
  const int row = i / (a.C * a.F), col = i % (a.C * a.F), c = col / a.F, f = col % a.F;
  a.out[i] = a.w1[c][f * a.d + row];
}

// Gradient of the step-invariant embeddings: X0[(t,b), row] = Ex[b, row] (+ Ey[b, p] while p is context),
// theta rows = tokens.   dEx[b,row] += sum_t dX0;  dEy[b,p] += sum_{t: ctx} dX0;  dtheta += sum_{t,b} dX0.
__global__ void assemble_bwd_kernel(Geo g, int d, int n_inst_t, const float *__restrict__ dX0,
                                    float *__restrict__ dEx, float *__restrict__ dEy, int ey_rows,
                                    float *__restrict__ dtheta) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int B = g.inst_B;
  long total = (long)B * g.N * d;
  if (i >= total) return;
  const int c = i % d;
  const long r = i / d;
  const int b = r / g.N, row = r % g.N;
  float sx = 0.f, sy = 0.f;
  for (int tt = 0; tt < n_inst_t; ++tt) {
    const long inst = (long)tt * B + b;
    const float v = dX0[(inst * g.N + row) * d + c];
    sx += v;
    if (row < g.P && is_ctx(g, (int)inst, row)) sy += v;
  }
  if (row < g.P + g.n_td) {
    dEx[((long)b * (g.P + g.n_td) + row) * d + c] += sx;
    if (row < g.P) dEy[((long)b * ey_rows + row) * d + c] += sy;
  } else {
    atomicAdd(dtheta + (row - g.P - g.n_td) * d + c, sx);
  }
}

// the same, 16 bytes per thread (d a multiple of 4)
__global__ void assemble_bwd4_kernel(Geo g, int d, int n_inst_t, const float *__restrict__ dX0,
                                     float *__restrict__ dEx, float *__restrict__ dEy, int ey_rows,
                                     float *__restrict__ dtheta) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int B = g.inst_B, d4 = d >> 2;
  if (i >= (long)B * g.N * d4) return;
  const int c = (int)(i % d4) * 4;
  const long r = i / d4;
  const int b = (int)(r / g.N), row = (int)(r % g.N);
  f32x4 sx = {0.f, 0.f, 0.f, 0.f}, sy = sx;
  for (int tt = 0; tt < n_inst_t; ++tt) {
    const long inst = (long)tt * B + b;
    const f32x4 v = *reinterpret_cast<const f32x4 *>(dX0 + (inst * g.N + row) * d + c);
    sx += v;
    if (row < g.P && is_ctx(g, (int)inst, row)) sy += v;
  }
  if (row < g.P + g.n_td) {
    f32x4 *px = reinterpret_cast<f32x4 *>(dEx + ((long)b * (g.P + g.n_td) + row) * d + c);
    *px += sx;
    if (row < g.P) { f32x4 *py = reinterpret_cast<f32x4 *>(dEy + ((long)b * ey_rows + row) * d + c); *py += sy; }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) atomicAdd(dtheta + (row - g.P - g.n_td) * d + c + q, sx[q]);
  }
}

// First layer of the point embedder, backward:  hid = relu(b1 + x W1^T) with tiny K.
// dhid [rows, F] (already masked by relu) -> dW1[f,k] += sum_r dhid[r,f] x[r,k], db1[f] += sum_r dhid[r,f]
__global__ __launch_bounds__(256) void embed_first_bwd_kernel(Src3 src, int rows_per_ep, int B, int K, int F,
                                                              const float *__restrict__ dhid,
                                                              float *dW1, float *db1, int rows_per_block) {
  // thread f-major: each thread owns one f and loops over the block's rows
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long rows = (long)B * rows_per_ep;
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    float gw[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gb = 0.f;
    for (long r = r0; r < min(rows, r0 + rows_per_block); ++r) {
      const int b = r / rows_per_ep, p = r % rows_per_ep;
      const float *x;
      if (p < src.n[0]) x = src.p[0] + ((long)b * src.n[0] + p) * K;
      else if (p < src.n[0] + src.n[1]) x = src.p[1] + ((long)b * src.n[1] + (p - src.n[0])) * K;
      else x = src.p[2] + ((long)b * src.n[2] + (p - src.n[0] - src.n[1])) * K;
      const float gv = dhid[r * F + f];
      gb += gv;
      for (int k = 0; k < K; ++k) gw[k] = fmaf(gv, x[k], gw[k]);
    }
    atomicAdd(db1 + f, gb);
    for (int k = 0; k < K; ++k) atomicAdd(dW1 + f * K + k, gw[k]);
  }
}

// The same for F <= 128:
__global__ __launch_bounds__(1024) void embed_first_bwd_groups_kernel(Src3 src, int rows_per_ep, int B, int K, int F,
                                                               const float *__restrict__ dhid,
                                                               float *dW1, float *db1, int rows_per_block) {
  // thread = (row group, f): blockDim.x / F groups share the block's rows (one thread per f walking all 256 rows of a block
  // was one memory latency per row: 0.18 ms per call with 0.4 waves per SIMD), partial sums meet in LDS, one atomic per
  // element and block as before.  F <= 128, K <= 8.
  __shared__ float part[8][128][9];
  const int ng = blockDim.x / F, grp = threadIdx.x / F, f = threadIdx.x % F;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long rows = (long)B * rows_per_ep, r1 = min(rows, r0 + rows_per_block);
  float gw[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gb = 0.f;
  if (grp < ng && grp < 8) {
    for (long r = r0 + grp; r < r1; r += min(ng, 8)) {
      const int b = r / rows_per_ep, p = r % rows_per_ep;
      const float *x;
      if (p < src.n[0]) x = src.p[0] + ((long)b * src.n[0] + p) * K;
      else if (p < src.n[0] + src.n[1]) x = src.p[1] + ((long)b * src.n[1] + (p - src.n[0])) * K;
      else x = src.p[2] + ((long)b * src.n[2] + (p - src.n[0] - src.n[1])) * K;
      const float gv = dhid[r * F + f];
      gb += gv;
      for (int k = 0; k < K; ++k) gw[k] = fmaf(gv, x[k], gw[k]);
    }
    part[grp][f][8] = gb;
    for (int k = 0; k < K; ++k) part[grp][f][k] = gw[k];
  }
  __syncthreads();
  if (grp == 0) {
    for (int q = 1; q < min(ng, 8); ++q) {
      gb += part[q][f][8];
      for (int k = 0; k < K; ++k) gw[k] += part[q][f][k];
    }
    atomicAdd(db1 + f, gb);
    for (int k = 0; k < K; ++k) atomicAdd(dW1 + f * K + k, gw[k]);
  }
}
