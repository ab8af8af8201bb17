// Acquisition head of the small-width model (d = 32, F = 128) in the TRAINING backward:
//   logit_p = w2 . relu(W1 z_p + b1) + b2  for the candidate rows,  log_prob = log_softmax over the remaining candidates
// (model/head.py:27-44; REINFORCE term of train_aline.py:113-125 through it).  Three kernels instead of five, and the
// [I P, 128] hidden activations (3 GB per chunk at the headline shape, written once and read / rewritten four times by the
// per-op pipeline: 15 GB, 5.4 ms) never exist:
//   logit_kernel    z -> logit of every token row (16-row tile per wave, hidden units in registers)
//   dlogit_kernel   one wave per (step, episode) instance: softmax over the remaining candidates, logit <- dLoss/dlogit
//   bwd_kernel      recomputes the hidden units of a tile, dz = W1^T dh for every row (target rows get zeros: this is also
//                   the initialisation of dLoss/dz), and keeps the dW1 / dw2 / db1 accumulators of the wave in registers.
// Register layouts and helpers: tail_bwd.h (T / N layouts, exact fp32 16x16x4 MFMAs).
#pragma once
#include "tail_bwd.h"

namespace acqb {

constexpr int D = 32, F = 128, PW = tailbwd::PW;
constexpr int L_W1 = 0, L_B1 = L_W1 + F * PW, L_W2 = L_B1 + F, L_SCR = L_W2 + F;
constexpr int WAVES = 4, THREADS = 64 * WAVES, SCR = 2 * 16 * PW;
constexpr int LDS_FLOATS_LOGIT = L_SCR, LDS_FLOATS = L_SCR + WAVES * SCR;
// gradient staging of bwd_kernel, reusing the image region: dW1 [128][32] | db1 [128] | dw2 [128]
static_assert(F * D + 2 * F <= L_SCR, "gradient staging must fit below the scratch");

struct Args {
  const float *Z;          // [M, 32] encoder output, every token row
  float *logit;            // [M] logits (logit_kernel) / dLoss/dlogit (after dlogit_kernel; 0 on non-candidate rows)
  float *dZ;               // [M, 32] (bwd_kernel) written for every row
  long M;
  const float *w1, *b1, *w2, *b2;
  float *dw1, *db1, *dw2;
  const unsigned *g_max_bits;   // bwd16_kernel: bits of max |dLoss/dlogit| (dlogit_kernel's DlArgs.out_absmax)
};
// bwd16_kernel (round 4: the products on the f16 matrix pipe, tail_bwd.h): the W1 image packed as f16 (hi | lo) quads, and its transpose
constexpr int PW2 = tailbwd::PW2, L_W1T = L_SCR + WAVES * SCR, LDS_FLOATS16 = L_W1T + D * PW2;

using fused::ld4;
using fused::group_sum;
using fused::zero4;

__device__ __forceinline__ void load_image(float *lds, const Args &a, int tid) {
  for (int i = tid; i < F * D; i += THREADS) lds[L_W1 + (i >> 5) * PW + (i & 31)] = a.w1[i];
  if (tid < F) { lds[L_B1 + tid] = a.b1[tid]; lds[L_W2 + tid] = a.w2[tid]; }
}

// F16: the hidden-unit product as the 3-term f16 split on v_mfma_f32_16x16x16_f16 (tail_bwd.h), as the rollout computed it
template <bool F16>
__global__ __launch_bounds__(THREADS) void logit_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tok = lane & 15, g = lane >> 4;
  if (F16) {
    tailbwd::pack_image16(lds + L_W1, PW, a.w1, F, D, D, 1, tid, THREADS);
    if (tid < F) { lds[L_B1 + tid] = a.b1[tid]; lds[L_W2 + tid] = a.w2[tid]; }
  } else load_image(lds, a, tid);
  __syncthreads();
  const float b2 = a.b2[0];
  const long ntiles = (a.M + 15) / 16;
  for (long tile = (long)blockIdx.x * WAVES + wave; tile < ntiles; tile += (long)gridDim.x * WAVES) {
    int zoff = 0;
    asm volatile("" : "+v"(zoff));
    const float *W = lds + zoff;
    const long row = tile * 16 + tok, rc = min(row, a.M - 1);
    const f32x4 z[2] = {ld4(a.Z + rc * D + 4 * g), ld4(a.Z + rc * D + 16 + 4 * g)};
    f32x4 h[8];
    if (F16) {
      const tailbwd::H8 zS[2] = {tailbwd::split4(z[0]), tailbwd::split4(z[1])};
#pragma unroll
      for (int ob = 0; ob < 8; ++ob) h[ob] = zero4();
      tailbwd::mm_fwd16<8, 2>(h, W + L_W1, PW, zS, tok, g);
#pragma unroll
      for (int ob = 0; ob < 8; ++ob) h[ob] = h[ob] * tailbwd::WINV16 + ld4(W + L_B1 + 16 * ob + 4 * g);
    } else {
#pragma unroll
      for (int ob = 0; ob < 8; ++ob) h[ob] = ld4(W + L_B1 + 16 * ob + 4 * g);
      tailbwd::mm_fwd<8, 2>(h, W + L_W1, PW, z, tok, g);
    }
    float s = 0.f;
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) {
      const f32x4 w2 = ld4(W + L_W2 + 16 * ob + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) s = fmaf(relu_nn(h[ob][r]), w2[r], s);
    }
    s = group_sum(s) + b2;
    if (g == 0 && row < a.M) a.logit[row] = s;
  }
}

struct DlArgs {
  Geo g;                   // instance mode
  float *logit;            // [I N] in: logits, out: dLoss/dlogit
  const float *g_logp;     // [B, T] dLoss/dlog_prob
  const int32_t *slot;     // [B, T] chosen slot
  int T;
  float *db2;
  unsigned *out_absmax;    // optional: max |dLoss/dlogit| as bits (the gradient scale of bwd16_kernel)
};
// one wave per instance at a time (persistent waves: the db2 sums leave as one atomic per wave, not per instance --
// 30 000 adds to one address were most of the 0.39 ms this kernel took)
__global__ __launch_bounds__(256) void dlogit_kernel(DlArgs a) {
  const int lane = threadIdx.x & 63;
  const int P = a.g.P, N = a.g.N;
  float dbl = 0.f, amax = 0.f;
  for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < a.g.B; i += gridDim.x * 4) {
    const int b = i % a.g.inst_B, t = a.g.inst_t0 + i / a.g.inst_B;
    float *lg = a.logit + (long)i * N;
    float mx = -INFINITY;
    for (int p = lane; p < P; p += 64) if (!is_ctx(a.g, i, p)) mx = fmaxf(mx, lg[p]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int p = lane; p < P; p += 64) if (!is_ctx(a.g, i, p)) sum += __expf(lg[p] - mx);
    const float inv = 1.f / wave_sum(sum);
    const float gl = a.g_logp[(long)b * a.T + t];
    const int chosen = a.slot[(long)b * a.T + t];
    for (int p = lane; p < N; p += 64) {
      float dl = 0.f;
      if (p < P && !is_ctx(a.g, i, p)) dl = gl * ((p == chosen ? 1.f : 0.f) - __expf(lg[p] - mx) * inv);
      lg[p] = dl;
      dbl += dl;
      amax = fmaxf(amax, fabsf(dl));
    }
  }
  dbl = wave_sum(dbl);
  if (lane == 0) atomicAdd(a.db2, dbl);
  if (a.out_absmax) {
    amax = wave_max(amax);
    if (lane == 0 && amax > 0.f) atomicMax(a.out_absmax, __float_as_uint(amax));
  }
}

__global__ __launch_bounds__(THREADS) void bwd_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tok = lane & 15, g = lane >> 4;
  load_image(lds, a, tid);
  __syncthreads();
  f32x4 gW1[8][2], gw2[8];      // dW1 tiles [16 ob + 4 g + r][16 jb + tok]; dw2 in the T layout (partial over the rows)
  float gB1[8];                 // db1 in the N layout (feature 16 ob + tok, partial over the lane groups)
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) { gW1[ob][0] = gW1[ob][1] = gw2[ob] = zero4(); gB1[ob] = 0.f; }
  const long ntiles = (a.M + 15) / 16;
  const long tstep = (long)gridDim.x * WAVES;
  // the rows of the tile after this one are requested while this one is computed (as in tailbwd::tail_kernel and gmmb::bwd_kernel)
  f32x4 nz[2];
  float ndl;
  auto load_tile = [&](long tile) {
    const long rn = min(tile * 16 + tok, a.M - 1);
    nz[0] = ld4(a.Z + rn * D + 4 * g); nz[1] = ld4(a.Z + rn * D + 16 + 4 * g);
    ndl = a.logit[rn];
  };
  if ((long)blockIdx.x * WAVES + wave < ntiles) load_tile((long)blockIdx.x * WAVES + wave);
  for (long tile = (long)blockIdx.x * WAVES + wave; tile < ntiles; tile += tstep) {
    int zoff = 0;
    asm volatile("" : "+v"(zoff));
    const float *W = lds + zoff;
    float *scr = lds + zoff + L_SCR + wave * SCR;
    const long row = tile * 16 + tok;
    const bool ok = row < a.M;
    const f32x4 z[2] = {nz[0], nz[1]};
    const float dl = ok ? ndl : 0.f;
    if (tile + tstep < ntiles) load_tile(tile + tstep);
    f32x4 h[8];
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) h[ob] = ld4(W + L_B1 + 16 * ob + 4 * g);
    tailbwd::mm_fwd<8, 2>(h, W + L_W1, PW, z, tok, g);
    // dw2 += dl h;  dh = dl w2 on the active units (h is overwritten by dh)
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) {
      const f32x4 w2 = ld4(W + L_W2 + 16 * ob + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float hv = relu_nn(h[ob][r]);
        gw2[ob][r] = fmaf(dl, hv, gw2[ob][r]);
        h[ob][r] = hv > 0.f ? dl * w2[r] : 0.f;
      }
    }
    f32x4 dz[2] = {zero4(), zero4()};
    tailbwd::mm_bwd<2, 8>(dz, W + L_W1, PW, h, tok, g);
    if (ok) {
      *reinterpret_cast<f32x4 *>(a.dZ + row * D + 4 * g) = dz[0];
      *reinterpret_cast<f32x4 *>(a.dZ + row * D + 16 + 4 * g) = dz[1];
    }
    f32x4 zN[2], dhN[2];
    tailbwd::to_n2(zN, dhN, z[0], z[1], h[0], h[1], scr, tok, g);
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      f32x4 nxt[2];
      if (kc < 3) tailbwd::to_n(nxt, h[2 * kc + 2], h[2 * kc + 3], scr, tok, g);
      tailbwd::mm_dw4(gW1[2 * kc][0], gW1[2 * kc][1], gW1[2 * kc + 1][0], gW1[2 * kc + 1][1], dhN[0], zN[0], dhN[0], zN[1],
                      dhN[1], zN[0], dhN[1], zN[1]);
      gB1[2 * kc] += tailbwd::sum4(dhN[0]);
      gB1[2 * kc + 1] += tailbwd::sum4(dhN[1]);
      if (kc < 3) { dhN[0] = nxt[0]; dhN[1] = nxt[1]; }
    }
  }
  // ---- the workgroup's gradients: LDS staging, then one atomic per element ---------------------------------------
  __syncthreads();
  for (int i = tid; i < F * D + 2 * F; i += THREADS) lds[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      atomicAdd(&lds[(16 * ob + 4 * g + r) * D + tok], gW1[ob][0][r]);
      atomicAdd(&lds[(16 * ob + 4 * g + r) * D + 16 + tok], gW1[ob][1][r]);
      atomicAdd(&lds[F * D + F + 16 * ob + 4 * g + r], gw2[ob][r]);
    }
    atomicAdd(&lds[F * D + 16 * ob + tok], gB1[ob]);
  }
  __syncthreads();
  for (int i = tid; i < F * D; i += THREADS) unsafeAtomicAdd(a.dw1 + i, lds[i]);
  if (tid < F) { unsafeAtomicAdd(a.db1 + tid, lds[F * D + tid]); unsafeAtomicAdd(a.dw2 + tid, lds[F * D + F + tid]); }
}

// bwd_kernel with every product a 3-term f16 split on v_mfma_f32_16x16x16_f16 (tail_bwd.h: tail16_kernel has the scheme): dLoss/dlogit is
// multiplied by the power of two of its maximum at the load, dz and the weight gradients are divided by it at the end.
__global__ __launch_bounds__(THREADS) void bwd16_kernel(Args a) {
  using tailbwd::H8;
  using tailbwd::split4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tok = lane & 15, g = lane >> 4;
  tailbwd::pack_image16(lds + L_W1, PW, a.w1, F, D, D, 1, tid, THREADS);
  tailbwd::pack_image16(lds + L_W1T, PW2, a.w1, D, F, 1, D, tid, THREADS);
  if (tid < F) { lds[L_B1 + tid] = a.b1[tid]; lds[L_W2 + tid] = a.w2[tid]; }
  __syncthreads();
  float ginv;
  const float gs = tailbwd::grad_scale16(*a.g_max_bits, ginv);
  f32x4 gW1[8][2], gw2[8];
  float gB1[8];
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) { gW1[ob][0] = gW1[ob][1] = gw2[ob] = zero4(); gB1[ob] = 0.f; }
  const long ntiles = (a.M + 15) / 16;
  const long tstep = (long)gridDim.x * WAVES;
  f32x4 nz[2];
  float ndl;
  auto load_tile = [&](long tile) {
    const long rn = min(tile * 16 + tok, a.M - 1);
    nz[0] = ld4(a.Z + rn * D + 4 * g); nz[1] = ld4(a.Z + rn * D + 16 + 4 * g);
    ndl = a.logit[rn];
  };
  if ((long)blockIdx.x * WAVES + wave < ntiles) load_tile((long)blockIdx.x * WAVES + wave);
  for (long tile = (long)blockIdx.x * WAVES + wave; tile < ntiles; tile += tstep) {
    int zoff = 0;
    asm volatile("" : "+v"(zoff));
    const float *W = lds + zoff;
    float *scr = lds + zoff + L_SCR + wave * SCR;
    const long row = tile * 16 + tok;
    const bool ok = row < a.M;
    const f32x4 z[2] = {nz[0], nz[1]};
    const float dl = ok ? ndl * gs : 0.f;
    if (tile + tstep < ntiles) load_tile(tile + tstep);
    const H8 zS[2] = {split4(z[0]), split4(z[1])};
    f32x4 h[8];
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) h[ob] = zero4();
    tailbwd::mm_fwd16<8, 2>(h, W + L_W1, PW, zS, tok, g);
    H8 dhS[8];
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) {
      const f32x4 w2 = ld4(W + L_W2 + 16 * ob + 4 * g), b1v = ld4(W + L_B1 + 16 * ob + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float hv = relu_nn(fmaf(h[ob][r], tailbwd::WINV16, b1v[r]));
        gw2[ob][r] = fmaf(dl, hv, gw2[ob][r]);
        h[ob][r] = hv > 0.f ? dl * w2[r] : 0.f;
      }
      dhS[ob] = split4(h[ob]);
    }
    f32x4 dz[2] = {zero4(), zero4()};
    tailbwd::mm_fwd16<2, 8>(dz, W + L_W1T, PW2, dhS, tok, g);
    if (ok) {
      const float dsc = tailbwd::WINV16 * ginv;
      *reinterpret_cast<f32x4 *>(a.dZ + row * D + 4 * g) = dz[0] * dsc;
      *reinterpret_cast<f32x4 *>(a.dZ + row * D + 16 + 4 * g) = dz[1] * dsc;
    }
    f32x4 zN[2], dhN[2];
    tailbwd::to_n2(zN, dhN, z[0], z[1], h[0], h[1], scr, tok, g);
    const H8 zNS[2] = {split4(zN[0]), split4(zN[1])};
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      f32x4 nxt[2];
      if (kc < 3) tailbwd::to_n(nxt, h[2 * kc + 2], h[2 * kc + 3], scr, tok, g);
      const H8 dhNS[2] = {split4(dhN[0]), split4(dhN[1])};
      tailbwd::mm_dw4_16(gW1[2 * kc][0], gW1[2 * kc][1], gW1[2 * kc + 1][0], gW1[2 * kc + 1][1], dhNS[0], zNS[0], dhNS[0], zNS[1],
                         dhNS[1], zNS[0], dhNS[1], zNS[1]);
      gB1[2 * kc] += tailbwd::sum4(dhN[0]);
      gB1[2 * kc + 1] += tailbwd::sum4(dhN[1]);
      if (kc < 3) { dhN[0] = nxt[0]; dhN[1] = nxt[1]; }
    }
  }
  __syncthreads();
  for (int i = tid; i < F * D + 2 * F; i += THREADS) lds[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      atomicAdd(&lds[(16 * ob + 4 * g + r) * D + tok], gW1[ob][0][r] * ginv);
      atomicAdd(&lds[(16 * ob + 4 * g + r) * D + 16 + tok], gW1[ob][1][r] * ginv);
      atomicAdd(&lds[F * D + F + 16 * ob + 4 * g + r], gw2[ob][r] * ginv);
    }
    atomicAdd(&lds[F * D + 16 * ob + tok], gB1[ob] * ginv);
  }
  __syncthreads();
  for (int i = tid; i < F * D; i += THREADS) unsafeAtomicAdd(a.dw1 + i, lds[i]);
  if (tid < F) { unsafeAtomicAdd(a.db1 + tid, lds[F * D + tid]); unsafeAtomicAdd(a.dw2 + tid, lds[F * D + F + tid]); }
}

}  // namespace acqb

// ---- GMM head (model/head.py:152-186, utils/eval.py:200-207) in the training backward, the same way ------------------------
//   raw_c = W2_c relu(W1_c z + b1_c) + b2_c  (3 numbers per component and target row) -> mean / softplus std / softmax weight
//   -> mixture log-likelihood.  The per-op pipeline kept the hidden units of all C components ([rows, C F]: 13 GB per step at
//   the cfg3 shape, 2.6 M target rows) and moved them six times.  Here they are recomputed per (16-row tile, component):
//   raw_kernel    grid (tiles, C): z of the target rows (gathered) -> raw [rows, C, 4]
//   draw_kernel   one thread per row: responsibilities etc. -> dLoss/draw [rows, C, 4], db2
//   bwd_kernel    grid (tiles, C): hidden units again, dw2_c / dW1_c / db1_c in registers, dz_c = W1_c^T dh -> dzc [C, rows, 32]
//   dzsum_kernel  dz[target rows] += sum_c dzc
namespace gmmb {

constexpr int D = acqb::D, F = acqb::F, PW = acqb::PW;
constexpr int L_W1 = 0, L_B1 = L_W1 + F * PW, L_W2 = L_B1 + F, L_SCR = L_W2 + 3 * F;      // W1 [128][36] | b1 | w2 [3][128]
constexpr int WAVES = 4, THREADS = 64 * WAVES, SCR = 2 * 16 * PW;
constexpr int LDS_FLOATS_RAW = L_SCR, LDS_FLOATS = L_SCR + WAVES * SCR;
static_assert(F * D + F + 3 * F <= L_SCR, "gradient staging must fit below the scratch");

struct Args {
  const float *Z;                // [I N, 32] encoder output
  int n_t, N, P;                 // target row q = i n_t + j  <->  token row i N + P + j
  long rows;                     // I n_t
  int C;
  const float *w1[16], *b1[16], *w2[16], *b2[16];
  float *dw1[16], *db1[16], *dw2[16], *db2[16];
  float *raw, *draw;             // [rows, C, 4]
  float *dzc;                    // [C, rows, 32]
  float *dZ;                     // [I N, 32] accumulated into on the target rows (dzsum_kernel)
  float std_min;
  const float *value; long value_mod;
  const float *g_ll, *g_mean, *g_std, *g_wgt;
  unsigned *draw_absmax;         // optional: max |draw| as bits, written by draw_kernel: the gradient scale of bwd16_kernel
};
constexpr int PW2 = tailbwd::PW2, L_W1T = L_SCR + WAVES * SCR, LDS_FLOATS16 = L_W1T + D * PW2;      // (bwd16_kernel: + the transposed W1 image)

using fused::ld4;
using fused::group_sum;
using fused::zero4;

__device__ __forceinline__ void load_image(float *lds, const Args &a, int c, int tid) {
  for (int i = tid; i < F * D; i += THREADS) lds[L_W1 + (i >> 5) * PW + (i & 31)] = a.w1[c][i];
  if (tid < F) lds[L_B1 + tid] = a.b1[c][tid];
  for (int i = tid; i < 3 * F; i += THREADS) lds[L_W2 + i] = a.w2[c][i];
}
__device__ __forceinline__ long token_row(const Args &a, long q) { return (q / a.n_t) * a.N + a.P + q % a.n_t; }

template <bool F16>
__global__ __launch_bounds__(THREADS) void raw_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tok = lane & 15, g = lane >> 4, c = blockIdx.y;
  if (F16) {
    tailbwd::pack_image16(lds + L_W1, PW, a.w1[c], F, D, D, 1, tid, THREADS);
    if (tid < F) lds[L_B1 + tid] = a.b1[c][tid];
    for (int i = tid; i < 3 * F; i += THREADS) lds[L_W2 + i] = a.w2[c][i];
  } else load_image(lds, a, c, tid);
  __syncthreads();
  const float b20 = a.b2[c][0], b21 = a.b2[c][1], b22 = a.b2[c][2];
  const long ntiles = (a.rows + 15) / 16;
  for (long tile = (long)blockIdx.x * WAVES + wave; tile < ntiles; tile += (long)gridDim.x * WAVES) {
    int zoff = 0;
    asm volatile("" : "+v"(zoff));
    const float *W = lds + zoff;
    const long q = tile * 16 + tok, zr = token_row(a, min(q, a.rows - 1));
    const f32x4 z[2] = {ld4(a.Z + zr * D + 4 * g), ld4(a.Z + zr * D + 16 + 4 * g)};
    f32x4 h[8];
    if (F16) {
      const tailbwd::H8 zS[2] = {tailbwd::split4(z[0]), tailbwd::split4(z[1])};
#pragma unroll
      for (int ob = 0; ob < 8; ++ob) h[ob] = zero4();
      tailbwd::mm_fwd16<8, 2>(h, W + L_W1, PW, zS, tok, g);
#pragma unroll
      for (int ob = 0; ob < 8; ++ob) h[ob] = h[ob] * tailbwd::WINV16 + ld4(W + L_B1 + 16 * ob + 4 * g);
    } else {
#pragma unroll
      for (int ob = 0; ob < 8; ++ob) h[ob] = ld4(W + L_B1 + 16 * ob + 4 * g);
      tailbwd::mm_fwd<8, 2>(h, W + L_W1, PW, z, tok, g);
    }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) {
      const f32x4 w0 = ld4(W + L_W2 + 16 * ob + 4 * g), w1 = ld4(W + L_W2 + F + 16 * ob + 4 * g), w2 = ld4(W + L_W2 + 2 * F + 16 * ob + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float hv = relu_nn(h[ob][r]);
        s0 = fmaf(hv, w0[r], s0); s1 = fmaf(hv, w1[r], s1); s2 = fmaf(hv, w2[r], s2);
      }
    }
    s0 = group_sum(s0) + b20; s1 = group_sum(s1) + b21; s2 = group_sum(s2) + b22;
    if (g == 0 && q < a.rows) *reinterpret_cast<f32x4 *>(a.raw + (q * a.C + c) * 4) = (f32x4){s0, s1, s2, 0.f};
  }
}

// one thread per target row (the arithmetic of gmm_bwd_kernel)
__global__ __launch_bounds__(256) void draw_kernel(Args a) {
  __shared__ float sdb[16 * 3];
  if (threadIdx.x < 48) sdb[threadIdx.x] = 0.f;
  __syncthreads();
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  const bool ok = q < a.rows;
  float d[16][3];
  if (ok) {
    float mean[16], sd[16], lw[16], r1[16];
    float mxw = -INFINITY;
    for (int c = 0; c < a.C; ++c) {
      const f32x4 rw = ld4(a.raw + (q * a.C + c) * 4);
      mean[c] = rw[0]; r1[c] = rw[1]; sd[c] = softplus_f(rw[1]) + a.std_min; lw[c] = rw[2];
      mxw = fmaxf(mxw, rw[2]);
    }
    float sw = 0.f;
    for (int c = 0; c < a.C; ++c) { lw[c] = __expf(lw[c] - mxw); sw += lw[c]; }
    const float v = a.value[q % a.value_mod];
    float lp[16], m2 = -INFINITY;
    for (int c = 0; c < a.C; ++c) {
      lw[c] /= sw;                                                   // mixture weight
      const float z = (v - mean[c]) / sd[c];
      lp[c] = -0.5f * z * z - logf(sd[c]) - 0.91893853320467274178f + logf(lw[c]);
      m2 = fmaxf(m2, lp[c]);
    }
    float se = 0.f;
    for (int c = 0; c < a.C; ++c) { lp[c] = __expf(lp[c] - m2); se += lp[c]; }
    const float gl = a.g_ll ? a.g_ll[q] : 0.f;
    float dot = 0.f;
    if (a.g_wgt) for (int c = 0; c < a.C; ++c) dot += a.g_wgt[q * a.C + c] * lw[c];
    for (int c = 0; c < a.C; ++c) {
      const float resp = lp[c] / se, z = (v - mean[c]) / sd[c];
      float d0 = gl * resp * z / sd[c], dsd = gl * resp * (z * z - 1.f) / sd[c], d2 = gl * (resp - lw[c]);
      if (a.g_mean) d0 += a.g_mean[q * a.C + c];
      if (a.g_std) dsd += a.g_std[q * a.C + c];
      if (a.g_wgt) d2 += lw[c] * (a.g_wgt[q * a.C + c] - dot);
      d[c][0] = d0; d[c][1] = dsd * (1.f / (1.f + __expf(-r1[c]))); d[c][2] = d2;
      *reinterpret_cast<f32x4 *>(a.draw + (q * a.C + c) * 4) = (f32x4){d[c][0], d[c][1], d[c][2], 0.f};
    }
  }
  if (a.draw_absmax) {
    float am = 0.f;
    if (ok) for (int c = 0; c < a.C; ++c) am = fmaxf(am, fmaxf(fabsf(d[c][0]), fmaxf(fabsf(d[c][1]), fabsf(d[c][2]))));
    am = wave_max(am);
    if ((threadIdx.x & 63) == 0 && am > 0.f) atomicMax(a.draw_absmax, __float_as_uint(am));
  }
  // db2: wave sums, then one LDS add per wave and value (48 values), one global atomic per block and value
  for (int c = 0; c < a.C; ++c)
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      const float sm = wave_sum(ok ? d[c][o] : 0.f);
      if ((threadIdx.x & 63) == 0) atomicAdd(&sdb[c * 3 + o], sm);
    }
  __syncthreads();
  if (threadIdx.x < a.C * 3) atomicAdd(a.db2[threadIdx.x / 3] + threadIdx.x % 3, sdb[threadIdx.x]);
}

__global__ __launch_bounds__(THREADS) void bwd_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tok = lane & 15, g = lane >> 4, c = blockIdx.y;
  load_image(lds, a, c, tid);
  __syncthreads();
  f32x4 gW1[8][2], gw2[3][8];   // dW1 tiles [16 ob + 4 g + r][16 jb + tok]; dw2 [o] in the T layout (partial over the rows)
  float gB1[8];
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) { gW1[ob][0] = gW1[ob][1] = gw2[0][ob] = gw2[1][ob] = gw2[2][ob] = zero4(); gB1[ob] = 0.f; }
  const long ntiles = (a.rows + 15) / 16;
  float *dzc = a.dzc + (long)c * a.rows * D;
  // the rows of the tile after this one are requested while this one is computed (one wave per SIMD: nobody else hides the round trip --
  // PMC at the cfg3 shape: matrix pipe 49 %, the wave waiting for an operand 41 % of its life before this)
  const long tstep = (long)gridDim.x * WAVES;
  f32x4 nz[2], ndr;
  auto load_tile = [&](long tile) {
    const long qn = min(tile * 16 + tok, a.rows - 1), zr = token_row(a, qn);
    nz[0] = ld4(a.Z + zr * D + 4 * g); nz[1] = ld4(a.Z + zr * D + 16 + 4 * g);
    ndr = ld4(a.draw + (qn * a.C + c) * 4);
  };
  if ((long)blockIdx.x * WAVES + wave < ntiles) load_tile((long)blockIdx.x * WAVES + wave);
  for (long tile = (long)blockIdx.x * WAVES + wave; tile < ntiles; tile += tstep) {
    int zoff = 0;
    asm volatile("" : "+v"(zoff));
    const float *W = lds + zoff;
    float *scr = lds + zoff + L_SCR + wave * SCR;
    const long q = tile * 16 + tok;
    const bool ok = q < a.rows;
    const f32x4 z[2] = {nz[0], nz[1]};
    f32x4 dr = ok ? ndr : zero4();
    if (tile + tstep < ntiles) load_tile(tile + tstep);
    f32x4 h[8];
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) h[ob] = ld4(W + L_B1 + 16 * ob + 4 * g);
    tailbwd::mm_fwd<8, 2>(h, W + L_W1, PW, z, tok, g);
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) {
      const f32x4 w0 = ld4(W + L_W2 + 16 * ob + 4 * g), w1 = ld4(W + L_W2 + F + 16 * ob + 4 * g), w2 = ld4(W + L_W2 + 2 * F + 16 * ob + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float hv = relu_nn(h[ob][r]);
        gw2[0][ob][r] = fmaf(dr[0], hv, gw2[0][ob][r]);
        gw2[1][ob][r] = fmaf(dr[1], hv, gw2[1][ob][r]);
        gw2[2][ob][r] = fmaf(dr[2], hv, gw2[2][ob][r]);
        h[ob][r] = hv > 0.f ? fmaf(dr[0], w0[r], fmaf(dr[1], w1[r], dr[2] * w2[r])) : 0.f;      // dh
      }
    }
    f32x4 dz[2] = {zero4(), zero4()};
    tailbwd::mm_bwd<2, 8>(dz, W + L_W1, PW, h, tok, g);
    if (ok) {
      *reinterpret_cast<f32x4 *>(dzc + q * D + 4 * g) = dz[0];
      *reinterpret_cast<f32x4 *>(dzc + q * D + 16 + 4 * g) = dz[1];
    }
    f32x4 zN[2], dhN[2];
    tailbwd::to_n2(zN, dhN, z[0], z[1], h[0], h[1], scr, tok, g);
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      f32x4 nxt[2];
      if (kc < 3) tailbwd::to_n(nxt, h[2 * kc + 2], h[2 * kc + 3], scr, tok, g);
      tailbwd::mm_dw4(gW1[2 * kc][0], gW1[2 * kc][1], gW1[2 * kc + 1][0], gW1[2 * kc + 1][1], dhN[0], zN[0], dhN[0], zN[1],
                      dhN[1], zN[0], dhN[1], zN[1]);
      gB1[2 * kc] += tailbwd::sum4(dhN[0]);
      gB1[2 * kc + 1] += tailbwd::sum4(dhN[1]);
      if (kc < 3) { dhN[0] = nxt[0]; dhN[1] = nxt[1]; }
    }
  }
  // ---- gradients of component c: LDS staging (dW1 [128][32] | db1 [128] | dw2 [3][128]), one atomic per element --------
  __syncthreads();
  for (int i = tid; i < F * D + F + 3 * F; i += THREADS) lds[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      atomicAdd(&lds[(16 * ob + 4 * g + r) * D + tok], gW1[ob][0][r]);
      atomicAdd(&lds[(16 * ob + 4 * g + r) * D + 16 + tok], gW1[ob][1][r]);
#pragma unroll
      for (int o = 0; o < 3; ++o) atomicAdd(&lds[F * D + F + o * F + 16 * ob + 4 * g + r], gw2[o][ob][r]);
    }
    atomicAdd(&lds[F * D + 16 * ob + tok], gB1[ob]);
  }
  __syncthreads();
  for (int i = tid; i < F * D; i += THREADS) unsafeAtomicAdd(a.dw1[c] + i, lds[i]);
  if (tid < F) unsafeAtomicAdd(a.db1[c] + tid, lds[F * D + tid]);
  for (int i = tid; i < 3 * F; i += THREADS) unsafeAtomicAdd(a.dw2[c] + i, lds[F * D + F + i]);
}

// bwd_kernel on the f16 matrix pipe (as acqb::bwd16_kernel): dLoss/draw scaled by the power of two of its maximum (draw_kernel)
__global__ __launch_bounds__(THREADS) void bwd16_kernel(Args a) {
  using tailbwd::H8;
  using tailbwd::split4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tok = lane & 15, g = lane >> 4, c = blockIdx.y;
  tailbwd::pack_image16(lds + L_W1, PW, a.w1[c], F, D, D, 1, tid, THREADS);
  tailbwd::pack_image16(lds + L_W1T, PW2, a.w1[c], D, F, 1, D, tid, THREADS);
  if (tid < F) lds[L_B1 + tid] = a.b1[c][tid];
  for (int i = tid; i < 3 * F; i += THREADS) lds[L_W2 + i] = a.w2[c][i];
  __syncthreads();
  float ginv;
  const float gs = tailbwd::grad_scale16(*a.draw_absmax, ginv);
  f32x4 gW1[8][2], gw2[3][8];
  float gB1[8];
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) { gW1[ob][0] = gW1[ob][1] = gw2[0][ob] = gw2[1][ob] = gw2[2][ob] = zero4(); gB1[ob] = 0.f; }
  const long ntiles = (a.rows + 15) / 16;
  float *dzc = a.dzc + (long)c * a.rows * D;
  const long tstep = (long)gridDim.x * WAVES;
  f32x4 nz[2], ndr;
  auto load_tile = [&](long tile) {
    const long qn = min(tile * 16 + tok, a.rows - 1), zr = token_row(a, qn);
    nz[0] = ld4(a.Z + zr * D + 4 * g); nz[1] = ld4(a.Z + zr * D + 16 + 4 * g);
    ndr = ld4(a.draw + (qn * a.C + c) * 4);
  };
  if ((long)blockIdx.x * WAVES + wave < ntiles) load_tile((long)blockIdx.x * WAVES + wave);
  for (long tile = (long)blockIdx.x * WAVES + wave; tile < ntiles; tile += tstep) {
    int zoff = 0;
    asm volatile("" : "+v"(zoff));
    const float *W = lds + zoff;
    float *scr = lds + zoff + L_SCR + wave * SCR;
    const long q = tile * 16 + tok;
    const bool ok = q < a.rows;
    const f32x4 z[2] = {nz[0], nz[1]};
    const f32x4 dr = ok ? ndr * gs : zero4();
    if (tile + tstep < ntiles) load_tile(tile + tstep);
    const H8 zS[2] = {split4(z[0]), split4(z[1])};
    f32x4 h[8];
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) h[ob] = zero4();
    tailbwd::mm_fwd16<8, 2>(h, W + L_W1, PW, zS, tok, g);
    H8 dhS[8];
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) {
      const f32x4 w0 = ld4(W + L_W2 + 16 * ob + 4 * g), w1 = ld4(W + L_W2 + F + 16 * ob + 4 * g), w2 = ld4(W + L_W2 + 2 * F + 16 * ob + 4 * g);
      const f32x4 b1v = ld4(W + L_B1 + 16 * ob + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float hv = relu_nn(fmaf(h[ob][r], tailbwd::WINV16, b1v[r]));
        gw2[0][ob][r] = fmaf(dr[0], hv, gw2[0][ob][r]);
        gw2[1][ob][r] = fmaf(dr[1], hv, gw2[1][ob][r]);
        gw2[2][ob][r] = fmaf(dr[2], hv, gw2[2][ob][r]);
        h[ob][r] = hv > 0.f ? fmaf(dr[0], w0[r], fmaf(dr[1], w1[r], dr[2] * w2[r])) : 0.f;      // dh
      }
      dhS[ob] = split4(h[ob]);
    }
    f32x4 dz[2] = {zero4(), zero4()};
    tailbwd::mm_fwd16<2, 8>(dz, W + L_W1T, PW2, dhS, tok, g);
    if (ok) {
      const float dsc = tailbwd::WINV16 * ginv;
      *reinterpret_cast<f32x4 *>(dzc + q * D + 4 * g) = dz[0] * dsc;
      *reinterpret_cast<f32x4 *>(dzc + q * D + 16 + 4 * g) = dz[1] * dsc;
    }
    f32x4 zN[2], dhN[2];
    tailbwd::to_n2(zN, dhN, z[0], z[1], h[0], h[1], scr, tok, g);
    const H8 zNS[2] = {split4(zN[0]), split4(zN[1])};
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      f32x4 nxt[2];
      if (kc < 3) tailbwd::to_n(nxt, h[2 * kc + 2], h[2 * kc + 3], scr, tok, g);
      const H8 dhNS[2] = {split4(dhN[0]), split4(dhN[1])};
      tailbwd::mm_dw4_16(gW1[2 * kc][0], gW1[2 * kc][1], gW1[2 * kc + 1][0], gW1[2 * kc + 1][1], dhNS[0], zNS[0], dhNS[0], zNS[1],
                         dhNS[1], zNS[0], dhNS[1], zNS[1]);
      gB1[2 * kc] += tailbwd::sum4(dhN[0]);
      gB1[2 * kc + 1] += tailbwd::sum4(dhN[1]);
      if (kc < 3) { dhN[0] = nxt[0]; dhN[1] = nxt[1]; }
    }
  }
  __syncthreads();
  for (int i = tid; i < F * D + F + 3 * F; i += THREADS) lds[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      atomicAdd(&lds[(16 * ob + 4 * g + r) * D + tok], gW1[ob][0][r] * ginv);
      atomicAdd(&lds[(16 * ob + 4 * g + r) * D + 16 + tok], gW1[ob][1][r] * ginv);
#pragma unroll
      for (int o = 0; o < 3; ++o) atomicAdd(&lds[F * D + F + o * F + 16 * ob + 4 * g + r], gw2[o][ob][r] * ginv);
    }
    atomicAdd(&lds[F * D + 16 * ob + tok], gB1[ob] * ginv);
  }
  __syncthreads();
  for (int i = tid; i < F * D; i += THREADS) unsafeAtomicAdd(a.dw1[c] + i, lds[i]);
  if (tid < F) unsafeAtomicAdd(a.db1[c] + tid, lds[F * D + tid]);
  for (int i = tid; i < 3 * F; i += THREADS) unsafeAtomicAdd(a.dw2[c] + i, lds[F * D + F + i]);
}

// dz[target rows] += sum over the components
__global__ void dzsum_kernel(Args a) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (row q, float4 of the 32 features)
  if (i >= a.rows * 8) return;
  const long q = i >> 3;
  const int c4 = (int)(i & 7) * 4;
  f32x4 s = zero4();
  for (int c = 0; c < a.C; ++c) s += ld4(a.dzc + ((long)c * a.rows + q) * D + c4);
  float *dst = a.dZ + token_row(a, q) * D + c4;
  *reinterpret_cast<f32x4 *>(dst) = ld4(dst) + s;
}

}  // namespace gmmb

