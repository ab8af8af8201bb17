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
    // two independent elements per thread and iteration (more loads in flight); theta is streamed (non-temporal:
    // 1.6 GB per step, read once)
    for (; i + stride < total; i += 2 * stride) {
      int b1 = b + db; if (b1 >= B) b1 -= B;
      typedef __attribute__((ext_vector_type(2))) float v2f;
      const v2f *thv = reinterpret_cast<const v2f *>(theta);
      const v2f t0 = __builtin_nontemporal_load(thv + i), t1 = __builtin_nontemporal_load(thv + i + stride);
      const float s0 = S[i], s1 = S[i + stride];              // S stays cacheable: the logsumexp pass reads it next
      const float2 x0 = xi2[b], x1 = xi2[b1];
      const float dx0 = x0.x - t0.x, dy0 = x0.y - t0.y, dx1 = x1.x - t1.x, dy1 = x1.y - t1.y;
      const float z0 = y[b] - logf(base + 1.f / (msig + fmaf(dx0, dx0, dy0 * dy0)));
      const float z1 = y[b1] - logf(base + 1.f / (msig + fmaf(dx1, dx1, dy1 * dy1)));
      S[i] = s0 + (-(z0 * z0) * inv2n2 - cst);
      S[i + stride] = s1 + (-(z1 * z1) * inv2n2 - cst);
      b = b1 + db; if (b >= B) b -= B;
    }
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

// Same step with everything that depends on the design / outcome of episode b only (clamped design, its log2, the
// noise scale (1 + |x_a - x_b|) * noise, logit(y) and the softplus terms) tabulated in LDS once per workgroup, and
// x^rho = exp2(rho * log2 x), s^(1/rho) = exp2(log2 s / rho) on the native exp2 / log2 units:
//   lp = -z^2 / 2 - log(sd) - c + softplus(-x) + softplus(x),  z = (logit(y) - (U1 - U2) u) / sd,  sd = dn * u,
//   log(sd) = log(dn) + theta_4 exactly (u = exp(theta_4)).
// 11 transcendentals per (l, b) instead of 8 powf + ~12 more; censored outcomes (y at eps / 1 - eps) take the
// general csn_log_prob.  Table row = 24 floats; used when B * 96 bytes fits 48 KB.
constexpr int CES_ROW = 24;
// 2^a - 1 and log2(1 + r) without cancellation for small arguments (series below 2^-3 / 0.1, the native units above)
__device__ __forceinline__ float exp2m1_f(float a) {
  const float z = a * 0.69314718055994530942f;
  const float p = z * (1.f + z * (0.5f + z * (0.16666666666666666f + z * (0.041666666666666664f + z * (0.0083333333333333332f + z * 0.0013888888888888889f)))));
  return fabsf(a) < 0.125f ? p : __builtin_amdgcn_exp2f(a) - 1.f;
}
__device__ __forceinline__ float log2_1p_f(float r) {
  const float p = r * (1.f + r * (-0.5f + r * (0.33333333333333331f + r * (-0.25f + r * (0.2f + r * (-0.16666666666666666f + r * (0.14285714285714285f + r * -0.125f)))))));
  return fabsf(r) < 0.1f ? p * 1.44269504088896340736f : __builtin_amdgcn_logf(1.f + r);
}
__global__ __launch_bounds__(256) void eig_ces_step_table_kernel(const float *__restrict__ theta,
                                                                 const float *__restrict__ xi,
                                                                 const float *__restrict__ y,
                                                                 float *__restrict__ S, long L1, int B,
                                                                 float noise, float eps, int *nan_flag) {
  extern __shared__ float tab[];        // [B][CES_ROW]: x[6] | log2 x[6] | 1/dn | cst | logit(y) | kind | dn | y | log2(xa / xb)[3]
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float *r = tab + b * CES_ROW;
    float x[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) { x[c] = fminf(fmaxf(xi[b * 6 + c], 0.01f), 100.f); r[c] = x[c]; r[6 + c] = log2f(x[c]); }
    const float dd0 = x[0] - x[3], dd1 = x[1] - x[4], dd2 = x[2] - x[5];
    const float dn = (1.f + sqrtf(dd0 * dd0 + dd1 * dd1 + dd2 * dd2)) * noise;
    const float v = y[b], lo = eps, hi = 1.f - eps;
    const int kind = (v > hi || v < lo) ? 3 : (v == hi || v == lo) ? 1 : 0;
    const float xl = logit_clamped(v);
    r[12] = 1.f / dn;
    r[13] = softplus_t(-xl) + softplus_t(xl) - LOG_SQRT_2PI - logf(dn);
    r[14] = xl;
    r[15] = __int_as_float(kind);
    r[16] = dn;
    r[17] = v;
#pragma unroll
    for (int c = 0; c < 3; ++c) r[18 + c] = (float)log2((double)x[c] / (double)x[3 + c]);
  }
  __syncthreads();
  const long total = L1 * B;
  bool bad = false;
  const long stride = (long)gridDim.x * blockDim.x;
  long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int b = (int)(i0 % B);
  const int db = (int)(stride % B);
  for (long i = i0; i < total; i += stride, b = (b + db >= B ? b + db - B : b + db)) {
    const float *th = theta + i * 5;
    const float rho = __builtin_nontemporal_load(th), a0 = __builtin_nontemporal_load(th + 1),
                a1 = __builtin_nontemporal_load(th + 2), a2 = __builtin_nontemporal_load(th + 3),
                t4 = __builtin_nontemporal_load(th + 4);      // theta is streamed (read once per step)
    const float *r = tab + b * CES_ROW;
    const int kind = __float_as_int(r[15]);
    float lp;
    if (kind == 3) {
      lp = -INFINITY;
    } else {
      // U_a - U_b without the cancellation of two large utilities (designs with similar baskets are exactly where the
      // likelihood discriminates): with t_i = x_b,i^rho, e_i = (x_a,i / x_b,i)^rho - 1 and s_b = sum alpha_i t_i,
      //   s_a / s_b = 1 + (sum alpha_i t_i e_i) / s_b,   U_a - U_b = U_b ((s_a / s_b)^(1/rho) - 1)
      // every "- 1" / "1 +" taken by series for small arguments.  fp64 says: 1e-4 in the log-likelihood where the plain
      // fp32 evaluation -- the reference's own -- is 1e-2 .. 5e-2 off (tests/test_r2_gpu.py::test_ces_realistic_regime).
      const float ir = 1.f / rho;
      const float t0 = a0 * __builtin_amdgcn_exp2f(rho * r[9]), t1 = a1 * __builtin_amdgcn_exp2f(rho * r[10]),
                  t2 = a2 * __builtin_amdgcn_exp2f(rho * r[11]);
      const float s2 = t0 + t1 + t2;
      const float ds = t0 * exp2m1_f(rho * r[18]) + t1 * exp2m1_f(rho * r[19]) + t2 * exp2m1_f(rho * r[20]);
      const float u2 = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(s2) * ir);
      const float du = u2 * exp2m1_f(log2_1p_f(ds / s2) * ir);          // u1 - u2
      if (kind == 0) {
        // z = (logit(y) - (u1 - u2) u) / (dn u)
        const float z = (r[14] * __expf(-t4) - du) * r[12];
        lp = -0.5f * z * z - t4 + r[13];
      } else {                                     // outcome at a censoring limit: log cdf with its asymptotic tail
        const float u = expf(t4);
        lp = csn_log_prob(r[17], du * u, r[16] * u, eps, 1.f - eps);
      }
    }
    bad |= (lp != lp) || isinf(lp);
    S[i] += lp;                                   // S stays cacheable: the logsumexp pass reads it next
  }
  if (nan_flag && bad) atomicOr(nan_flag, 1);
}

// ---- EIGStepLoss.forward: logsumexp over l (loss/eig.py:195-209) ---------------------------------
// pass 1: per chunk of rows l in [1 + c*CH, ...) an online (max, sumexp) per column b
__global__ __launch_bounds__(256) void eig_lse_partial_kernel(const float *__restrict__ S, long L1, int B,
                                                              long chunk, float *__restrict__ part) {
  // a workgroup covers cols = min(B, 256) columns x rows = 256 / cols rows per pass: whole rows of S whenever
  // B <= 256 (a pass reads contiguous memory: no partially used cache lines), and for the small outer batches of the
  // evaluation protocol (B = 20, README.md:50) 240 of 256 lanes stay busy
  __shared__ float sm[256], ss[256];
  const int cols = min(B, 256), rows = 256 / cols;
  const int tx = threadIdx.x % cols, ty = threadIdx.x / cols;
  const int b = blockIdx.y * cols + tx;
  const long l0 = 1 + (long)blockIdx.x * chunk;
  const long l1 = min(L1, l0 + chunk);
  float m = -INFINITY, s = 0.f;
  if (b < B && ty < rows) {
    long l = l0 + ty;
    const long r1 = rows, r2 = 2 * rows, r3 = 3 * rows;
    for (; l + r3 < l1; l += 4 * rows) {       // four independent loads in flight per thread, one rescale per group
      const float v0 = S[l * B + b], v1 = S[(l + r1) * B + b], v2 = S[(l + r2) * B + b], v3 = S[(l + r3) * B + b];
      const float mn = fmaxf(fmaxf(m, fmaxf(v0, v1)), fmaxf(v2, v3));
      if (mn != -INFINITY)
        s = s * __expf(m - mn) + ((__expf(v0 - mn) + __expf(v1 - mn)) + (__expf(v2 - mn) + __expf(v3 - mn)));
      m = mn;
    }
    for (; l < l1; l += rows) {
      float v = S[l * B + b];
      float mn = fmaxf(m, v);
      s = (mn == -INFINITY) ? 0.f : s * __expf(m - mn) + __expf(v - mn);
      m = mn;
    }
  }
  sm[threadIdx.x] = m; ss[threadIdx.x] = s;
  __syncthreads();
  if (ty == 0 && b < B) {
    for (int j = 1; j < rows; ++j) {
      float m2 = sm[j * cols + tx], s2 = ss[j * cols + tx];
      float mn = fmaxf(m, m2);
      if (mn != -INFINITY) s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
      m = mn;
    }
    part[((long)blockIdx.x * B + b) * 2 + 0] = m;
    part[((long)blockIdx.x * B + b) * 2 + 1] = s;
  }
}
// pass 2: combine the chunks (one workgroup per column b); bounds of utils/eval.py:77-78
__global__ __launch_bounds__(256) void eig_lse_combine_kernel(const float *__restrict__ S, const float *__restrict__ part,
                                                              int nchunk, long L1, int B, float *pce, float *nmc) {
  __shared__ float sm[256], ss[256];
  const int b = blockIdx.x;
  float m = -INFINITY, s = 0.f;
  for (int c = threadIdx.x; c < nchunk; c += 256) {
    float m2 = part[((long)c * B + b) * 2], s2 = part[((long)c * B + b) * 2 + 1];
    float mn = fmaxf(m, m2);
    if (mn != -INFINITY) s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
    m = mn;
  }
  sm[threadIdx.x] = m; ss[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
      const float m2 = sm[threadIdx.x + w], s2 = ss[threadIdx.x + w];
      const float m1 = sm[threadIdx.x], s1 = ss[threadIdx.x];
      const float mn = fmaxf(m1, m2);
      sm[threadIdx.x] = mn;
      ss[threadIdx.x] = (mn == -INFINITY) ? 0.f : s1 * __expf(m1 - mn) + s2 * __expf(m2 - mn);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    m = sm[0]; s = ss[0];
    const float s0 = S[b];
    const float lse1 = m + logf(s);                 // l >= 1
    const float mx = fmaxf(lse1, s0);
    const float lse0 = mx + logf(__expf(lse1 - mx) + __expf(s0 - mx));   // l >= 0
    const float L = (float)(L1 - 1);
    if (pce) pce[b] = logf(L + 1.f) - (lse0 - s0);
    if (nmc) nmc[b] = logf(L) - (lse1 - s0);
  }
}


// ---- all T steps of a design history in ONE pass over theta (round 4) --------------------------------------------------------------
// compute_EIG_from_history (utils/eval.py:42-80) calls the step T times: every call reads theta (dim_theta x 4 B per (l, b)) and reads and
// writes the running sum S (8 B), and -- stepwise -- a logsumexp pass reads S again: 20 B per (l, b) and step, 3.2 - 4 GB per step at
// L = 1e6, B = 200.  The steps of a history are known up front, so one kernel reads theta ONCE, walks the T designs of its episode
// (xi, y of its episode from L1), carries S in a register and keeps one online (max, sum-exp) pair per step:
// 8 B per (l, b) for all T steps, the bound moves from HBM to the transcendental units (1 log + 1.25 exp + 1 rcp per (l, b, t)).
// part[row group][t][b] = (m, s) over that group's rows l >= 1;  s0[t][b] = S_t of row l = 0 (the "true" theta).
template <int K1D2>
__global__ __launch_bounds__(256) void eig_location_history_kernel(const float *__restrict__ theta, const float *__restrict__ xi,
                                                                   const float *__restrict__ y, long L1, int B, int T, int K, int D,
                                                                   float noise, float base, float msig, long R,
                                                                   float *__restrict__ part, float *__restrict__ s0) {
  // A thread owns column b and the rows l0 + ty, l0 + ty + rows, ... of its chunk; the (max, sum-exp) pairs of up to 32 steps live in
  // its registers (static indices: the step loop is unrolled), the designs / outcomes of its episode come from L1 (T x 12 B per
  // episode, the same addresses for every row).  No LDS, no barrier: occupancy is set by the ~110 registers alone.  (First version:
  // accumulators and designs in LDS, 133 KB per workgroup = one wave per SIMD: 13.2 ms at L = 1e6, B = 200, T = 30, latency-bound.)
  // flat mapping: thread gid owns episode b = gid % B and the rows 1 + rg, 1 + rg + R, ... with rg = gid / B (R row groups in all): every
  // lane of every wave works whatever B is (a [min(B, 256) columns] x [256 / columns rows] tiling left 56 of 256 lanes idle at B = 200)
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)B * R) return;
  const int b = (int)(gid % B);
  const long rg = gid / B;
  const float inv2n2 = 1.f / (2.f * noise * noise), cst = logf(noise) + LOG_SQRT_2PI;
  const long l1 = L1;
  // xi [B, T, D], y [B, T] episode-major: a lane's loads of the unrolled steps are ONE base register + immediate offsets (136 registers,
  // three waves per SIMD).  Step-major [T, D, B] (coalesced across lanes) was measured: runtime strides make the compiler precompute the 90
  // addresses of the unrolled steps (256 + 70 registers, one wave per SIMD) for the same 9.9 ms -- the lines stay in L1 either way.
  const float *xb = xi + (size_t)b * T * D, *yb = y + (size_t)b * T;
  auto ll_of = [&](const float *th, const float *x, float yv) -> float {
    float inv_sum = 0.f;
    for (int k = 0; k < (K1D2 ? 1 : K); ++k) {
      float sq = 0.f;
      for (int d = 0; d < (K1D2 ? 2 : D); ++d) { const float df = x[d] - th[k * D + d]; sq = fmaf(df, df, sq); }
      inv_sum += __builtin_amdgcn_rcpf(msig + sq);      // (v_rcp_f32, 1 ulp: the IEEE division sequence is 10 instructions)
    }
    // v_log_f32 (log2, 1 ulp) x ln 2 instead of the ~20-instruction logf: the kernel is bound by instruction issue, and the bounds
    // agree with the step kernels' (logf, IEEE division) to 1e-5 at L = 1e5 .. 1e6 (tests/test_r4_gpu.py)
    const float z = yv - __log2f(base + inv_sum) * 0.69314718055994530942f;
    return -(z * z) * inv2n2 - cst;
  };
  auto load_x = [&](int t, float *x) {
    if (K1D2) { const float2 v = reinterpret_cast<const float2 *>(xb)[t]; x[0] = v.x; x[1] = v.y; }
    else for (int d = 0; d < D; ++d) x[d] = xb[(size_t)t * D + d];
  };
  auto load_theta = [&](long l, float *th) {
    if (K1D2) { const float2 t2 = reinterpret_cast<const float2 *>(theta)[l * B + b]; th[0] = t2.x; th[1] = t2.y; }
    else for (int e = 0; e < K * D; ++e) th[e] = theta[(l * B + b) * K * D + e];
  };
  if (rg == 0) {                               // row l = 0 (the true theta): its running sums go to s0, not into the logsumexp over l >= 1
    float th[8], x[8];
    load_theta(0, th);
    float S = 0.f;
    for (int t = 0; t < T; ++t) { load_x(t, x); S += ll_of(th, x, yb[t]); s0[(size_t)t * B + b] = S; }
  }
  constexpr int TM = 32;                       // steps per register block (T <= 32: one pass over theta; longer histories: one per 32 steps)
  for (int t0 = 0; t0 < T; t0 += TM) {
    const int nt = min(TM, T - t0);
    float m[TM], sacc[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) { m[t] = -INFINITY; sacc[t] = 0.f; }
    // four rows of theta per thread and pass: four independent S chains, one rescale of the step's (max, sum) per group
    long l = 1 + rg;
    const long r1 = R;
    for (; l < l1; l += 4 * r1) {
      float th[4][8], S[4] = {0.f, 0.f, 0.f, 0.f};
      bool on[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { on[q] = l + q * r1 < l1; load_theta(on[q] ? l + q * r1 : l, th[q]); }
      for (int t = 0; t < t0; ++t) {           // (second and later passes: the steps before t0 only advance S)
        float x[8];
        load_x(t, x);
        const float yv = yb[t];
#pragma unroll
        for (int q = 0; q < 4; ++q) S[q] += ll_of(th[q], x, yv);
      }
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        if (t < nt) {
          float x[8];
          load_x(t0 + t, x);
          const float yv = yb[t0 + t];
#pragma unroll
          for (int q = 0; q < 4; ++q) { S[q] += ll_of(th[q], x, yv); if (!on[q]) S[q] = -INFINITY; }
          const float mn = fmaxf(fmaxf(m[t], fmaxf(S[0], S[1])), fmaxf(S[2], S[3]));      // (S[0] is always a live row: finite)
          sacc[t] = sacc[t] * __expf(m[t] - mn) + ((__expf(S[0] - mn) + __expf(S[1] - mn)) + (__expf(S[2] - mn) + __expf(S[3] - mn)));
          m[t] = mn;
          asm volatile("" ::: "memory");       // (keeps the loads of step t + 1 behind step t: hoisted, they cost 150 registers)
        }
      }
    }
    // every row group writes its own partial: part[rg][t][b]
#pragma unroll
    for (int t = 0; t < TM; ++t)
      if (t < nt) {
        float *pp = part + (((size_t)rg * T + (t0 + t)) * B + b) * 2;
        pp[0] = m[t]; pp[1] = sacc[t];
      }
  }
}
// combine the chunks: one WAVE per (t, b) -- its lanes walk the row groups 64 apart and merge their (max, sum-exp) pairs at the end (one
// thread per (t, b) walked them alone: 9 830 dependent round trips at B = 20, 5 of the CES history's 5.3 ms); bounds of utils/eval.py:77-78
__global__ __launch_bounds__(256) void eig_history_combine_kernel(const float *__restrict__ part, const float *__restrict__ s0, int nchunk,
                                                                  long L1, int B, int T, float *pce, float *nmc) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= (long)T * B) return;
  const int t = (int)(i / B), b = (int)(i % B);
  float m = -INFINITY, s = 0.f;
  for (int c = lane; c < nchunk; c += 64) {
    const float *pp = part + (((size_t)c * T + t) * B + b) * 2;
    const float m2 = pp[0], s2 = pp[1];
    const float mn = fmaxf(m, m2);
    if (mn != -INFINITY) s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
    m = mn;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    const float mn = fmaxf(m, m2);
    if (mn != -INFINITY) s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
    m = mn;
  }
  if (lane != 0) return;
  const float sv = s0[(size_t)t * B + b];
  const float lse1 = m + logf(s);                 // l >= 1
  const float mx = fmaxf(lse1, sv);
  const float lse0 = mx + logf(__expf(lse1 - mx) + __expf(sv - mx));   // l >= 0
  const float L = (float)(L1 - 1);
  if (pce) pce[(size_t)b * T + t] = logf(L + 1.f) - (lse0 - sv);      // [B, T] as torch.stack(..., dim=-1)
  if (nmc) nmc[(size_t)b * T + t] = logf(L) - (lse1 - sv);
}

// ---- CES: all T steps of a design history in one pass over theta (round 4, end) -------------------------------------------------------
// The same scheme as eig_location_history_kernel for CESTask.log_likelihood (tasks/ces.py:96-115, :169-210): a thread owns episode b and the
// rows 1 + rg, 1 + rg + R, ... of theta (five floats, read ONCE), walks the T designs of its episode -- the per-(episode, step) table of
// eig_ces_step_table_kernel (clamped designs, their log2, noise scale, logit(y), the censoring kind: CES_ROW floats) is built by every
// workgroup in LDS, B T rows -- with the running log-likelihood in a register and one online (max, sum-exp) pair per step.  The
// arithmetic of a step is the table kernel's (same series, same native exp2 / log2): bounds equal to the step kernels' to rounding.
// T <= 16 (CES histories are 10 steps); the host falls back to the step kernels when the table does not fit 64 KB.
__global__ __launch_bounds__(256) void eig_ces_history_kernel(const float *__restrict__ theta, const float *__restrict__ xi, const float *__restrict__ y,
                                                              long L1, int B, int T, float noise, float eps, long R,
                                                              float *__restrict__ part, float *__restrict__ s0, int *nan_flag) {
  extern __shared__ float tab[];        // [B][T] rows: x[6] | log2 x[6] | 1/dn | cst | logit(y) | kind | dn | y | log2(xa / xb)[3]
  // row pitch 25 floats, episode pitch odd: the lanes of a wave hold different episodes and read the same word of their rows -- with the
  // pitch of the step kernel (24, episodes 240 floats apart) they met in two banks
  constexpr int RP = CES_ROW + 1;
  const int EP = (T * RP) | 1;
  for (int i = threadIdx.x; i < B * T; i += blockDim.x) {
    float *r = tab + (size_t)(i / T) * EP + (size_t)(i % T) * RP;
    const float *xr = xi + (size_t)i * 6;      // xi [B, T, 6], y [B, T]
    float x[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) { x[c] = fminf(fmaxf(xr[c], 0.01f), 100.f); r[c] = x[c]; r[6 + c] = log2f(x[c]); }
    const float dd0 = x[0] - x[3], dd1 = x[1] - x[4], dd2 = x[2] - x[5];
    const float dn = (1.f + sqrtf(dd0 * dd0 + dd1 * dd1 + dd2 * dd2)) * noise;
    const float v = y[i], lo = eps, hi = 1.f - eps;
    const int kind = (v > hi || v < lo) ? 3 : (v == hi || v == lo) ? 1 : 0;
    const float xl = logit_clamped(v);
    r[12] = 1.f / dn;
    r[13] = softplus_t(-xl) + softplus_t(xl) - LOG_SQRT_2PI - logf(dn);
    r[14] = xl;
    r[15] = __int_as_float(kind);
    r[16] = dn;
    r[17] = v;
#pragma unroll
    for (int c = 0; c < 3; ++c) r[18 + c] = (float)log2((double)x[c] / (double)x[3 + c]);
  }
  __syncthreads();
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)B * R) return;
  const int b = (int)(gid % B);
  const long rg = gid / B;
  const float *rows = tab + (size_t)b * EP;
  auto lp_of = [&](const float *th, const float *r) -> float {      // (eig_ces_step_table_kernel's step)
    const float rho = th[0], a0 = th[1], a1 = th[2], a2 = th[3], t4 = th[4];
    const int kind = __float_as_int(r[15]);
    if (kind == 3) return -INFINITY;
    const float ir = 1.f / rho;
    const float t0 = a0 * __builtin_amdgcn_exp2f(rho * r[9]), t1 = a1 * __builtin_amdgcn_exp2f(rho * r[10]), t2 = a2 * __builtin_amdgcn_exp2f(rho * r[11]);
    const float s2 = t0 + t1 + t2;
    const float ds = t0 * exp2m1_f(rho * r[18]) + t1 * exp2m1_f(rho * r[19]) + t2 * exp2m1_f(rho * r[20]);
    const float u2 = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(s2) * ir);
    const float du = u2 * exp2m1_f(log2_1p_f(ds / s2) * ir);          // u1 - u2
    if (kind == 0) {
      const float z = (r[14] * __expf(-t4) - du) * r[12];
      return -0.5f * z * z - t4 + r[13];
    }
    const float u = expf(t4);                                          // outcome at a censoring limit
    return csn_log_prob(r[17], du * u, r[16] * u, eps, 1.f - eps);
  };
  bool bad = false;
  if (rg == 0) {                               // row l = 0 (the true theta)
    float th[5];
#pragma unroll
    for (int e = 0; e < 5; ++e) th[e] = theta[(size_t)b * 5 + e];
    float S = 0.f;
    for (int t = 0; t < T; ++t) { const float lp = lp_of(th, rows + t * RP); bad |= (lp != lp) || isinf(lp); S += lp; s0[(size_t)t * B + b] = S; }
  }
  // the (max, sum-exp) pair of every step lives in LDS ([T][256] pairs behind the table: the step loop is a real loop -- unrolled over 16
  // steps with two rows in flight the body was ~10 000 instructions, more than the instruction cache, and the kernel 2 x slower than the
  // ten step launches it replaces)
  float2 *ms = reinterpret_cast<float2 *>(tab + (((size_t)B * EP + 1) & ~(size_t)1)) + threadIdx.x;
  for (int t = 0; t < T; ++t) ms[t * 256] = make_float2(-INFINITY, 0.f);
  for (long l = 1 + rg; l < L1; l += 2 * R) {      // two rows per pass: two independent chains
    const bool on1 = l + R < L1;
    float tha[5], thb[5];
    const float *pa = theta + (l * B + b) * 5, *pb = theta + ((on1 ? l + R : l) * B + b) * 5;
#pragma unroll
    for (int e = 0; e < 5; ++e) { tha[e] = __builtin_nontemporal_load(pa + e); thb[e] = __builtin_nontemporal_load(pb + e); }
    float Sa = 0.f, Sb = 0.f;
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
      const float la = lp_of(tha, rows + t * RP), lb = lp_of(thb, rows + t * RP);
      bad |= (la != la) || isinf(la) || (on1 && ((lb != lb) || isinf(lb)));
      Sa += la; Sb = on1 ? Sb + lb : -INFINITY;
      float2 p = ms[t * 256];
      const float mn = fmaxf(p.x, fmaxf(Sa, Sb));
      if (mn != -INFINITY) p.y = p.y * __expf(p.x - mn) + __expf(Sa - mn) + __expf(Sb - mn);
      p.x = mn;
      ms[t * 256] = p;
    }
  }
  for (int t = 0; t < T; ++t) {
    const float2 p = ms[t * 256];
    float *pp = part + (((size_t)rg * T + t) * B + b) * 2;
    pp[0] = p.x; pp[1] = p.y;
  }
  if (nan_flag && bad) atomicOr(nan_flag, 1);
}

// ---- batched Cholesky for the GP task sampler (tasks/gaussian_process.py:391-415) ------
// A = U^T U, U upper triangular, in place on the upper triangle of each [n, n] matrix (the strict lower
// triangle is zeroed).  One workgroup per matrix, left-looking by rows of U:
//   s_i = A[j][i] - sum_{k<j} U[k][i] U[k][j]   (i >= j; threads run over i: coalesced rows of U)
//   U[j][j] = sqrt(s_j);  U[j][i] = s_i / U[j][j]
__global__ __launch_bounds__(256) void cholesky_upper_rowwise_kernel(float *__restrict__ A, int n, int *info) {
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

// Same factorisation, NB = 8 rows of U per sweep over the previous rows: every U[k][i] fetched feeds 8 FMAs, so the
// O(n^3 / 6) re-reads of the row-wise kernel (9.3 GB for 512 matrices of 301^2 by PMC) shrink by NB.  Thread t owns
// columns i = j0 + t + 256 c (c < CMAX); the block of U[k][j0 .. j0 + NB) multipliers sits in LDS.  Summation order
// over k is that of the row-wise kernel.  n <= 256 * CMAX.
constexpr int CHOL_NB = 8;
template <int CMAX>
__global__ __launch_bounds__(256) void cholesky_upper_kernel(float *__restrict__ A, int n, int *info) {
  constexpr int NB = CHOL_NB;
  extern __shared__ float sh[];                // colb [n][NB] | rowb [NB] | diag
  float *colb = sh, *rowb = sh + (size_t)n * NB, *diagp = rowb + NB;
  float *M = A + (long)blockIdx.x * n * n;
  const int tid = threadIdx.x;
  for (int j0 = 0; j0 < n; j0 += NB) {
    const int nbj = min(NB, n - j0);
    for (int e = tid; e < j0 * NB; e += 256) {
      const int k = e / NB, r = e % NB;
      colb[e] = r < nbj ? M[(long)k * n + j0 + r] : 0.f;
    }
    __syncthreads();
    float s[CMAX][NB];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      const int i = j0 + tid + 256 * c;
#pragma unroll
      for (int r = 0; r < NB; ++r) s[c][r] = (i < n && r < nbj) ? M[(long)(j0 + r) * n + i] : 0.f;
    }
    for (int k = 0; k < j0; ++k) {
      float mult[NB];
#pragma unroll
      for (int r = 0; r < NB; ++r) mult[r] = colb[k * NB + r];
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        const int i = j0 + tid + 256 * c;
        const float u = i < n ? M[(long)k * n + i] : 0.f;
#pragma unroll
        for (int r = 0; r < NB; ++r) s[c][r] = fmaf(-u, mult[r], s[c][r]);
      }
    }
    // factorise the NB x (n - j0) block row by row
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      if (r < nbj) {
        if (tid == r) {                      // column j0 + r is column c = 0 of thread r
          const float d = s[0][r];
          if (!(d > 0.f)) atomicOr(info, 1);
          diagp[0] = sqrtf(fmaxf(d, 1e-30f));
        }
        __syncthreads();
        const float dg = diagp[0], inv = 1.f / dg;
        float urow[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
          const int i = j0 + tid + 256 * c;
          urow[c] = 0.f;
          if (i < n && i >= j0 + r) {
            urow[c] = (i == j0 + r) ? dg : s[c][r] * inv;
            M[(long)(j0 + r) * n + i] = urow[c];
          }
        }
        if (tid > r && tid < nbj) rowb[tid] = urow[0];       // U[j0 + r][j0 + r'] for the rows r' > r of this block
        for (int i = tid; i < j0 + r; i += 256) M[(long)(j0 + r) * n + i] = 0.f;     // strict lower part of the row
        __syncthreads();
#pragma unroll
        for (int r2 = r + 1; r2 < NB; ++r2)
          if (r2 < nbj) {
            const float m2 = rowb[r2];
#pragma unroll
            for (int c = 0; c < CMAX; ++c) s[c][r2] = fmaf(-urow[c], m2, s[c][r2]);
          }
        __syncthreads();
      }
    }
  }
}
