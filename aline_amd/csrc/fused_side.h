// Side kernels of the fused small-width path (d = 32, F = 128), same register-resident scheme as
// fused_rollout.h (token tiles in the MFMA accumulator layout, split-bf16 weight products):
//   embed_points_kernel   point embedder  Linear(k,128)-ReLU-Linear(128,32)  (model/embedder.py:47-57)
//   gmm_rows_kernel       C GMM heads + parameter maps + compute_ll on dense rows
//                         (model/head.py:152-186, 251-266; utils/eval.py:200-207)
#pragma once
#include "fused_rollout.h"

namespace fused {

// ---- point embedder: out[row, 0..31] = W2 relu(W1 x[row] + b1) + b2 -------------------------------
// One wave per 16-row tile.  The first layer has K = dim_x / dim_y <= 8 inputs: FMA work done directly
// in the accumulator layout (lane = token, 32 hidden units per lane); the second layer is 4 split-bf16
// blocks against W2 fragments packed into LDS once per workgroup.
struct EmbedArgs {
  const float *x; int K; long rows;            // [rows, K]
  const float *w1, *b1, *b2;                   // [128, K], [128], [32]
  const float *w2img;                          // W2 as 8 packed split-bf16 fragments (pack_weights_kernel)
  float *out;                                  // [rows, 32]
};

__global__ __launch_bounds__(256) void embed_points_kernel(EmbedArgs a) {
  __shared__ __attribute__((aligned(16))) float W2f[8 * FRAG3];   // (mt, kb) fragments, 24 KB
  __shared__ float w1s[F * 8], b1s[F], b2s[D];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tok = lane & 15, g = lane >> 4;
  for (int i = tid; i < 8 * FRAG3 / 4; i += 256)
    reinterpret_cast<f32x4 *>(W2f)[i] = reinterpret_cast<const f32x4 *>(a.w2img)[i];
  for (int i = tid; i < F * a.K; i += 256) w1s[i] = a.w1[i];
  for (int i = tid; i < F; i += 256) b1s[i] = a.b1[i];
  if (tid < D) b2s[tid] = a.b2[tid];
  __syncthreads();
  const long ntiles = (a.rows + 15) / 16;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    const long row = tile * 16 + tok;
    float xv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) xv[k] = (k < a.K && row < a.rows) ? a.x[row * a.K + k] : 0.f;
    f32x4 y[2];
    y[0] = ld4(b2s + 4 * g);
    y[1] = ld4(b2s + 16 + 4 * g);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      f32x4 h[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int u = 32 * kb + 16 * mt + 4 * g + r;
          float acc = b1s[u];
          for (int k = 0; k < a.K; ++k) acc = fmaf(xv[k], w1s[u * a.K + k], acc);
          h[mt][r] = relu_nn(acc);
        }
      const Frag3 hf = split_acc(h[0], h[1]);
      mma6x2(y[0], y[1], ld_frag3(W2f + (0 * 4 + kb) * FRAG3, lane), ld_frag3(W2f + (1 * 4 + kb) * FRAG3, lane), hf);
    }
    if (row < a.rows) {
      *reinterpret_cast<f32x4 *>(a.out + row * D + 4 * g) = y[0];
      *reinterpret_cast<f32x4 *>(a.out + row * D + 16 + 4 * g) = y[1];
    }
  }
}

// ---- GMM heads on dense rows ------------------------------------------------------------------------
// One workgroup = 4 waves x 1 tile of 16 rows; the C heads are walked one at a time: the head's first
// layer (8 split-bf16 fragments, 24 KB) is packed into LDS by the whole workgroup, every wave computes
// hidden^T = relu(W1 z + b1) for its tile (8 blocks x 6 passes) and folds the 3-output second layer
// into the accumulator layout with FMAs + one lane-group reduction per output.
struct GmmRowsArgs {
  const float *z; long rows;                   // [rows, 32]; input row of `row` = (row / zR) * zG + zoff + row % zR when zR > 0
  int zR, zG, zoff;                            // (the target rows of every episode inside a [B * N, 32] activation)
  int C; float std_min;
  const float *w1img;                          // C x 8 packed split-bf16 fragments of the first layers
  const float *b1[16], *w2[16], *b2[16];
  float *mean, *sd, *wgt;                      // [rows, C] or null
  const float *value; long value_row0, value_mod;   // value[(value_row0 + row) % value_mod]
  float *ll;                                   // [rows] or null
};

constexpr int GMM_TILES = 2, GMM_THREADS = 512;   // 8 waves x 2 token tiles: a head's first-layer image (24 KB) is staged once per 256 rows

__global__ __launch_bounds__(GMM_THREADS) void gmm_rows_kernel(GmmRowsArgs a) {
  __shared__ __attribute__((aligned(16))) float W1f[8 * FRAG3];   // 24 KB
  __shared__ float b1s[F], w2s[3 * F], b2s[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tok = lane & 15, g = lane >> 4;
  long row[GMM_TILES];
  bool ok[GMM_TILES];
  Frag3 zf[GMM_TILES];
#pragma unroll
  for (int t = 0; t < GMM_TILES; ++t) {
    row[t] = (((long)blockIdx.x * (GMM_THREADS / 64) + wave) * GMM_TILES + t) * 16 + tok;
    ok[t] = row[t] < a.rows;
    f32x4 z0 = zero4(), z1 = zero4();
    if (ok[t]) {
      const long zr = a.zR > 0 ? (row[t] / a.zR) * a.zG + a.zoff + row[t] % a.zR : row[t];
      z0 = ld4(a.z + zr * D + 4 * g); z1 = ld4(a.z + zr * D + 16 + 4 * g);
    }
    zf[t] = split_acc(z0, z1);
  }
  float raw[GMM_TILES][16][3];
#pragma unroll 1
  for (int c = 0; c < a.C; ++c) {
    __syncthreads();
    for (int i = tid; i < 8 * FRAG3 / 4; i += GMM_THREADS)
      reinterpret_cast<f32x4 *>(W1f)[i] = reinterpret_cast<const f32x4 *>(a.w1img + (long)c * SIDE_FRAGS)[i];
    for (int i = tid; i < F; i += GMM_THREADS) b1s[i] = a.b1[c][i];
    for (int i = tid; i < 3 * F; i += GMM_THREADS) w2s[i] = a.w2[c][i];
    if (tid < 3) b2s[tid] = a.b2[c][tid];
    __syncthreads();
    float p[GMM_TILES][3];
#pragma unroll
    for (int t = 0; t < GMM_TILES; ++t) p[t][0] = p[t][1] = p[t][2] = 0.f;
#pragma unroll
    for (int mp = 0; mp < 4; ++mp) {
      const Frag3 u0 = ld_frag3(W1f + (2 * mp) * FRAG3, lane), u1 = ld_frag3(W1f + (2 * mp + 1) * FRAG3, lane);
      const f32x4 hb0 = ld4(b1s + 32 * mp + 4 * g), hb1 = ld4(b1s + 32 * mp + 16 + 4 * g);
      f32x4 h0[GMM_TILES], h1[GMM_TILES];
#pragma unroll
      for (int t = 0; t < GMM_TILES; ++t) { h0[t] = hb0; h1[t] = hb1; mma6x2(h0[t], h1[t], u0, u1, zf[t]); }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int u0i = 32 * mp + 4 * g + r, u1i = u0i + 16;
        const float wa0 = w2s[u0i], wb0 = w2s[u1i], wa1 = w2s[F + u0i], wb1 = w2s[F + u1i], wa2 = w2s[2 * F + u0i],
                    wb2 = w2s[2 * F + u1i];
#pragma unroll
        for (int t = 0; t < GMM_TILES; ++t) {
          const float a0 = relu_nn(h0[t][r]), a1 = relu_nn(h1[t][r]);
          p[t][0] = fmaf(a0, wa0, p[t][0]); p[t][0] = fmaf(a1, wb0, p[t][0]);
          p[t][1] = fmaf(a0, wa1, p[t][1]); p[t][1] = fmaf(a1, wb1, p[t][1]);
          p[t][2] = fmaf(a0, wa2, p[t][2]); p[t][2] = fmaf(a1, wb2, p[t][2]);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < GMM_TILES; ++t) {
      const float r0 = group_sum(p[t][0]) + b2s[0], r1 = group_sum(p[t][1]) + b2s[1], r2 = group_sum(p[t][2]) + b2s[2];
      // static register file: write component c through a wave-uniform switch
#pragma unroll
      for (int cc = 0; cc < 16; ++cc)
        if (cc == c) { raw[t][cc][0] = r0; raw[t][cc][1] = r1; raw[t][cc][2] = r2; }
    }
  }
  if (g != 0) return;
  // parameter maps (head.py:176-177) and compute_ll (eval.py:200-207), one lane per row
#pragma unroll
  for (int t = 0; t < GMM_TILES; ++t) {
    if (!ok[t]) continue;
    float mxw = -INFINITY;
#pragma unroll
    for (int c = 0; c < 16; ++c) if (c < a.C) mxw = fmaxf(mxw, raw[t][c][2]);
    float sw = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) if (c < a.C) sw += __expf(raw[t][c][2] - mxw);
    const float v = (a.ll && a.value) ? a.value[a.value_mod > 0 ? (a.value_row0 + row[t]) % a.value_mod : row[t]] : 0.f;
    float lps[16], mx2 = -INFINITY;
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c < a.C) {
        const float mean = raw[t][c][0], sd = softplus_f(raw[t][c][1]) + a.std_min;
        const float w = __expf(raw[t][c][2] - mxw) / sw;
        if (a.mean) a.mean[row[t] * a.C + c] = mean;
        if (a.sd) a.sd[row[t] * a.C + c] = sd;
        if (a.wgt) a.wgt[row[t] * a.C + c] = w;
        const float zz = (v - mean) / sd;
        lps[c] = -0.5f * zz * zz - logf(sd) - 0.91893853320467274178f + logf(w);
        mx2 = fmaxf(mx2, lps[c]);
      }
    if (a.ll && a.value) {
      float se = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) if (c < a.C) se += __expf(lps[c] - mx2);
      a.ll[row[t]] = mx2 + logf(se);
    }
  }
}

}  // namespace fused
