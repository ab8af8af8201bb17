// Sequential nested-Monte-Carlo EIG bounds (loss/eig.py:174-209, utils/eval.py:42-80).
// Elementwise log-likelihoods + streaming logsumexp over the L contrastive samples: HBM-bound.
#pragma once
#include "common.h"

#define LOG_SQRT_2PI 0.91893853320467274178f

// HiddenLocation.log_likelihood (tasks/location_finding.py:110-130, :149-164)
__device__ __forceinline__ float location_ll(const float *__restrict__ th, const float *xi, float y,
                                             int K, int D, float noise, float base, float msig) {
  float inv_sum = 0.f;
  for (int k = 0; k < K; ++k) {
    float sq = 0.f;
    for (int c = 0; c < D; ++c) { float t = xi[c] - th[k * D + c]; sq = fmaf(t, t, sq); }
    inv_sum += 1.f / (msig + sq);
  }
  float mu = logf(base + inv_sum);
  float z = y - mu;
  return -(z * z) / (2.f * noise * noise) - logf(noise) - LOG_SQRT_2PI;
}

// S[l, b] += ll(y[b] | xi[b], theta[l, b])          EIGStepLoss.step (loss/eig.py:174-193)
__global__ __launch_bounds__(256) void eig_location_step_kernel(const float *__restrict__ theta,
                                                                const float *__restrict__ xi,
                                                                const float *__restrict__ y,
                                                                float *__restrict__ S, long L1, int B,
                                                                int K, int D, float noise, float base,
                                                                float msig) {
  const long total = L1 * B;
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  // (l, b) of element i is carried along the grid-stride loop: no 64-bit division per element
  int b = (int)(i % B);
  const int db = (int)(stride % B);
  if (K == 1 && D == 2) {               // the configured task (location_finding.yaml): theta is one float2 per (l, b)
    const float2 *th2 = reinterpret_cast<const float2 *>(theta);
    const float2 *xi2 = reinterpret_cast<const float2 *>(xi);
    const float inv2n2 = 1.f / (2.f * noise * noise), cst = logf(noise) + LOG_SQRT_2PI;
    for (; i < total; i += stride) {
      const float2 t = th2[i], x = xi2[b];
      const float dx = x.x - t.x, dy = x.y - t.y;
      const float z = y[b] - logf(base + 1.f / (msig + fmaf(dx, dx, dy * dy)));
      S[i] += -(z * z) * inv2n2 - cst;
      b += db; if (b >= B) b -= B;
    }
    return;
  }
  for (; i < total; i += stride) {
    float x[8];
    for (int c = 0; c < D; ++c) x[c] = xi[b * D + c];
    S[i] += location_ll(theta + i * K * D, x, y[b], K, D, noise, base, msig);
    b += db; if (b >= B) b -= B;
  }
}

// ---- CensoredSigmoidNormal.log_prob (distributions/censored_sigmoid_normal.py:47-86) -----------
__device__ __forceinline__ float logit_clamped(float v) {
  // torch SigmoidTransform._inverse: clamp to [finfo.tiny, 1 - finfo.eps]
  v = fminf(fmaxf(v, 1.17549435e-38f), 1.f - 1.1920929e-07f);
  return logf(v) - log1pf(-v);
}
__device__ __forceinline__ float normal_lp(float x, float mu, float sd) {
  float z = x - mu;
  return -(z * z) / (2.f * sd * sd) - logf(sd) - LOG_SQRT_2PI;
}
__device__ __forceinline__ float normal_cdf(float x, float mu, float sd) {
  return 0.5f * (1.f + erff((x - mu) / (sd * 1.41421356237309504880f)));
}
__device__ __forceinline__ float softplus_t(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_normal_lp(float v, float mu, float sd) {
  float x = logit_clamped(v);
  return normal_lp(x, mu, sd) + softplus_t(-x) + softplus_t(x);
}
__device__ __forceinline__ float csn_log_prob(float v, float mu, float sd, float lo, float hi) {
  const float crit = 2.f * 1.17549435e-38f;
  if (v > hi || v < lo) return -INFINITY;
  if (v == hi) {
    float lim = logit_clamped(hi);
    float ucdf = 1.f - normal_cdf(lim, mu, sd);
    if (ucdf < crit) return sigmoid_normal_lp(hi, mu, sd) - logf(crit + fabsf((lim - mu) / sd));
    return logf(ucdf);
  }
  if (v == lo) {
    float lim = logit_clamped(lo);
    float lcdf = normal_cdf(lim, mu, sd);
    if (lcdf < crit) return sigmoid_normal_lp(lo, mu, sd) - logf(crit + fabsf((lim - mu) / sd));
    return logf(lcdf);
  }
  return sigmoid_normal_lp(v, mu, sd);
}

// CESTask.log_likelihood (tasks/ces.py:96-115, :169-210)
__global__ __launch_bounds__(256) void eig_ces_step_kernel(const float *__restrict__ theta,
                                                           const float *__restrict__ xi,
                                                           const float *__restrict__ y,
                                                           float *__restrict__ S, long L1, int B,
                                                           float noise, float eps, int *nan_flag) {
  const long total = L1 * B;
  bool bad = false;
  const long stride = (long)gridDim.x * blockDim.x;
  long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int b = (int)(i0 % B);
  const int db = (int)(stride % B);
  for (long i = i0; i < total; i += stride, b = (b + db >= B ? b + db - B : b + db)) {
    const float *th = theta + i * 5;
    const float rho = th[0], a0 = th[1], a1 = th[2], a2 = th[3], u = expf(th[4]);
    float x[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) x[c] = fminf(fmaxf(xi[b * 6 + c], 0.01f), 100.f);
    const float ir = 1.f / rho;
    float u1 = powf(a0 * powf(x[0], rho) + a1 * powf(x[1], rho) + a2 * powf(x[2], rho), ir);
    float u2 = powf(a0 * powf(x[3], rho) + a1 * powf(x[4], rho) + a2 * powf(x[5], rho), ir);
    float mu = (u1 - u2) * u;
    float dd0 = x[0] - x[3], dd1 = x[1] - x[4], dd2 = x[2] - x[5];
    float sd = (1.f + sqrtf(dd0 * dd0 + dd1 * dd1 + dd2 * dd2)) * noise * u;
    float lp = csn_log_prob(y[b], mu, sd, eps, 1.f - eps);
    bad |= (lp != lp) || isinf(lp);
    S[i] += lp;
  }
  if (nan_flag && bad) atomicOr(nan_flag, 1);
}

// ---- EIGStepLoss.forward: logsumexp over l (loss/eig.py:195-209) ---------------------------------
// pass 1: per chunk of rows l in [1 + c*CH, ...) an online (max, sumexp) per column b
__global__ __launch_bounds__(256) void eig_lse_partial_kernel(const float *__restrict__ S, long L1, int B,
                                                              long chunk, float *__restrict__ part) {
  __shared__ float sm[4][64], ss[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int b = blockIdx.y * 64 + tx;
  const long l0 = 1 + (long)blockIdx.x * chunk;
  const long l1 = min(L1, l0 + chunk);
  float m = -INFINITY, s = 0.f;
  if (b < B) {
    long l = l0 + ty;
    for (; l + 12 < l1; l += 16) {       // four independent loads in flight per thread, one rescale per group
      const float v0 = S[l * B + b], v1 = S[(l + 4) * B + b], v2 = S[(l + 8) * B + b], v3 = S[(l + 12) * B + b];
      const float mn = fmaxf(fmaxf(m, fmaxf(v0, v1)), fmaxf(v2, v3));
      if (mn != -INFINITY)
        s = s * __expf(m - mn) + ((__expf(v0 - mn) + __expf(v1 - mn)) + (__expf(v2 - mn) + __expf(v3 - mn)));
      m = mn;
    }
    for (; l < l1; l += 4) {
      float v = S[l * B + b];
      float mn = fmaxf(m, v);
      s = (mn == -INFINITY) ? 0.f : s * __expf(m - mn) + __expf(v - mn);
      m = mn;
    }
  }
  sm[ty][tx] = m; ss[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && b < B) {
    for (int j = 1; j < 4; ++j) {
      float m2 = sm[j][tx], s2 = ss[j][tx];
      float mn = fmaxf(m, m2);
      if (mn != -INFINITY) s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
      m = mn;
    }
    part[((long)blockIdx.x * B + b) * 2 + 0] = m;
    part[((long)blockIdx.x * B + b) * 2 + 1] = s;
  }
}
// pass 2: combine the chunks; bounds of utils/eval.py:77-78
__global__ void eig_lse_combine_kernel(const float *__restrict__ S, const float *__restrict__ part,
                                       int nchunk, long L1, int B, float *pce, float *nmc) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float m = -INFINITY, s = 0.f;
  for (int c = 0; c < nchunk; ++c) {
    float m2 = part[((long)c * B + b) * 2], s2 = part[((long)c * B + b) * 2 + 1];
    float mn = fmaxf(m, m2);
    if (mn != -INFINITY) s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
    m = mn;
  }
  const float s0 = S[b];
  const float lse1 = m + logf(s);                 // l >= 1
  const float mx = fmaxf(lse1, s0);
  const float lse0 = mx + logf(__expf(lse1 - mx) + __expf(s0 - mx));   // l >= 0
  const float L = (float)(L1 - 1);
  if (pce) pce[b] = logf(L + 1.f) - (lse0 - s0);
  if (nmc) nmc[b] = logf(L) - (lse1 - s0);
}


// ---- batched Cholesky for the GP task sampler (tasks/gaussian_process.py:391-415) ------
// A = U^T U, U upper triangular, in place on the upper triangle of each [n, n] matrix (the strict lower
// triangle is zeroed).  One workgroup per matrix, left-looking by rows of U:
//   s_i = A[j][i] - sum_{k<j} U[k][i] U[k][j]   (i >= j; threads run over i: coalesced rows of U)
//   U[j][j] = sqrt(s_j);  U[j][i] = s_i / U[j][j]
__global__ __launch_bounds__(256) void cholesky_upper_kernel(float *__restrict__ A, int n, int *info) {
  extern __shared__ float colj[];              // U[0..j)[j]
  __shared__ float diag;
  float *M = A + (long)blockIdx.x * n * n;
  const int tid = threadIdx.x;
  for (int j = 0; j < n; ++j) {
    for (int k = tid; k < j; k += 256) colj[k] = M[(long)k * n + j];
    __syncthreads();
    for (int i = j + tid; i < n; i += 256) {
      float s = M[(long)j * n + i];
      for (int k = 0; k < j; ++k) s = fmaf(-M[(long)k * n + i], colj[k], s);
      M[(long)j * n + i] = s;
      if (i == j) {
        if (!(s > 0.f)) atomicOr(info, 1);
        diag = sqrtf(fmaxf(s, 1e-30f));
      }
    }
    __syncthreads();
    const float inv = 1.f / diag;
    for (int i = j + tid; i < n; i += 256) M[(long)j * n + i] = (i == j) ? diag : M[(long)j * n + i] * inv;
    for (int i = tid; i < j; i += 256) M[(long)j * n + i] = 0.f;     // strict lower part of row j
    __syncthreads();
  }
}
