// Non-GEMM kernels of one design step: point-embedder first layer, token assembly, masked
// set-attention, residual+LayerNorm, acquisition softmax + design selection (+ role update),
// GMM head epilogue (+ log-likelihood).  All fp32.
#pragma once
#include "common.h"

// Token geometry of one step.  Per episode the rows are
//   [ P candidate points | n_td target-data rows | n_th theta-token rows ]   (N rows)
// A point is a context point when is_ctx(b, p); otherwise it is a (remaining) query.
// Step API: the first n_ctx points are the context (role == nullptr).  Rollout API: role[b, p] > 0.
struct Geo {
  int B, P, n_td, n_th, N;
  int n_ctx;             // static context count (step API) / current count (rollout, informational)
  const int *role;       // [B, P] or nullptr
  const uint8_t *tmask;  // [n_td + n_th] or nullptr (= every target is visible to the queries)
  // instance mode (backward pass): "episode" index i is a (step, episode) pair, t = inst_t0 + i / inst_B,
  // b = i % inst_B; slot p is context at step t iff 0 < role[b, p] <= n_ctx0 + t (role = order of entry)
  int inst_B, inst_t0, n_ctx0;
};

__device__ __forceinline__ bool is_ctx(const Geo &g, int b, int p) {
  if (g.inst_B > 0) {
    const int r = g.role[(long)(b % g.inst_B) * g.P + p];
    return r > 0 && r <= g.n_ctx0 + g.inst_t0 + b / g.inst_B;
  }
  return g.role ? g.role[(long)b * g.P + p] > 0 : p < g.n_ctx;
}

// ---- G1a: first layer of the point embedder (model/embedder.py:47-57), K = dim_x / dim_y is
// tiny so this is FMA work:  hid[r, f] = relu(b1[f] + sum_k in[r, k] * w1[f, k]).
// Rows come from up to three source tensors laid end to end per episode (ctx | query | target_x).
struct Src3 { const float *p[3]; int n[3]; };
__global__ void embed_hidden_kernel(Src3 src, int rows_per_ep, int B, int K, int F,
                                    const float *__restrict__ w1, const float *__restrict__ b1,
                                    float *__restrict__ hid) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)B * rows_per_ep * F;
  if (i >= total) return;
  int f = i % F;
  long r = i / F;
  int b = r / rows_per_ep, p = r % rows_per_ep;
  const float *x;
  if (p < src.n[0]) x = src.p[0] + ((long)b * src.n[0] + p) * K;
  else if (p < src.n[0] + src.n[1]) x = src.p[1] + ((long)b * src.n[1] + (p - src.n[0])) * K;
  else x = src.p[2] + ((long)b * src.n[2] + (p - src.n[0] - src.n[1])) * K;
  float acc = b1[f];
  for (int k = 0; k < K; ++k) acc = fmaf(x[k], w1[f * K + k], acc);
  hid[i] = fmaxf(acc, 0.f);
}

// ---- G1b: token assembly (embedder.py:156-166 / :196-212):
// X[b, row] = Ex[b, row] (+ Ey[b, p] on context rows); theta rows = theta_tokens.
__global__ void assemble_kernel(Geo g, int d, const float *__restrict__ Ex, const float *__restrict__ Ey,
                                int ey_rows, const float *__restrict__ theta_tokens,
                                float *__restrict__ X) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)g.B * g.N * d;
  if (i >= total) return;
  int c = i % d;
  long r = i / d;
  int b = r / g.N, row = r % g.N;
  float v;
  const int eb = g.inst_B > 0 ? b % g.inst_B : b;   // embeddings are per episode, shared by its steps
  if (row < g.P + g.n_td) {
    v = Ex[((long)eb * (g.P + g.n_td) + row) * d + c];
    if (row < g.P && is_ctx(g, b, row)) v += Ey[((long)eb * ey_rows + row) * d + c];
  } else {
    v = theta_tokens[(row - g.P - g.n_td) * d + c];
  }
  X[i] = v;
}

// the same, 16 bytes per thread (d a multiple of 4): the scalar form ran at 2 TB/s on the 6.09 M rows of a backward chunk
__global__ void assemble4_kernel(Geo g, int d, const float *__restrict__ Ex, const float *__restrict__ Ey,
                                 int ey_rows, const float *__restrict__ theta_tokens, float *__restrict__ X) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int d4 = d >> 2;
  if (i >= (long)g.B * g.N * d4) return;
  const int c = (int)(i % d4) * 4;
  const long r = i / d4;
  const int b = (int)(r / g.N), row = (int)(r % g.N);
  const int eb = g.inst_B > 0 ? b % g.inst_B : b;
  f32x4 v;
  if (row < g.P + g.n_td) {
    v = *reinterpret_cast<const f32x4 *>(Ex + ((long)eb * (g.P + g.n_td) + row) * d + c);
    if (row < g.P && is_ctx(g, b, row)) v += *reinterpret_cast<const f32x4 *>(Ey + ((long)eb * ey_rows + row) * d + c);
  } else {
    v = *reinterpret_cast<const f32x4 *>(theta_tokens + (row - g.P - g.n_td) * d + c);
  }
  *reinterpret_cast<f32x4 *>(X + r * d + c) = v;
}

// Between two steps of a rollout the embedded input changes in ONE row per episode: the point chosen at the previous
// step (role == order) became a context point, its row becomes Ex + Ey.  One workgroup per episode.
__global__ __launch_bounds__(256) void patch_row_kernel(Geo g, int d, const float *__restrict__ Ex, const float *__restrict__ Ey,
                                                        int ey_rows, int order, float *__restrict__ X) {
  __shared__ int s_slot;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) s_slot = -1;
  __syncthreads();
  for (int p = tid; p < g.P; p += 256)
    if (g.role[(long)b * g.P + p] == order) s_slot = p;
  __syncthreads();
  const int slot = s_slot;
  if (slot < 0) return;
  for (int c = tid; c < d; c += 256)
    X[((long)b * g.N + slot) * d + c] = Ex[((long)b * (g.P + g.n_td) + slot) * d + c] + Ey[((long)b * ey_rows + slot) * d + c];
}

// ---- key rows of every episode (model/encoder.py:83-126): context points in slot order, then the visible targets.
// keyidx[b * max_keys + j] = global token row (b * N + row) of key j, -1 beyond the episode's keys; kcnt[2 b] = context
// keys, kcnt[2 b + 1] = all keys.  The K / V projections of the generic pipeline run on these rows only.
__global__ __launch_bounds__(256) void key_list_kernel(Geo g, int max_keys, int *__restrict__ keyidx, int *__restrict__ kcnt) {
  __shared__ int wave_cnt[4];
  __shared__ int s_base;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int *list = keyidx + (long)b * max_keys;
  if (tid == 0) s_base = 0;
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
    if (key && k < max_keys) list[k] = b * g.N + row;
    __syncthreads();
    if (tid == 0) s_base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  __shared__ int s_all;
  if (tid == 0) {
    int n = min(s_base, max_keys);
    kcnt[2 * b] = n;
    const int n_t = g.n_td + g.n_th;
    for (int j = 0; j < n_t; ++j)
      if ((!g.tmask || g.tmask[j]) && n < max_keys) list[n++] = b * g.N + g.P + j;
    kcnt[2 * b + 1] = n;
    s_all = n;
  }
  __syncthreads();
  for (int j = s_all + tid; j < max_keys; j += 256) list[j] = -1;
}

// ---- G2-G4: masked set-attention (model/encoder.py:8-46 and :83-126 without materialising the
// [N, N] mask).  Keys = context rows, plus (for query rows only) the selected target rows.
// One workgroup per (episode, head): K_h, V_h of the key rows are staged in LDS once, every
// thread then owns token rows and runs an online softmax over the keys.
template <int HD>
__global__ __launch_bounds__(512) void attention_kernel(Geo g, int d, const float *__restrict__ QKV,
                                                        float *__restrict__ Aout, int max_keys,
                                                        const float *__restrict__ KVc = nullptr, const int *__restrict__ kcnt = nullptr) {
  // KVc != null: Q rows are [M, d] at QKV, K | V of the key rows only are [B * max_keys, 2 d] at KVc in key-list order
  // with the counts in kcnt (key_list_kernel) -- no compaction here
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float *Ks = reinterpret_cast<float *>(smem_raw);          // [max_keys][HD]
  float *Vs = Ks + (size_t)max_keys * HD;                   // [max_keys][HD]
  int *keyrow = reinterpret_cast<int *>(Vs + (size_t)max_keys * HD);   // [max_keys]
  __shared__ int wave_cnt[8];      // blockDim.x = 256 .. 512: the smallest multiple of 64 covering the N token rows
  __shared__ int s_base;
  // the H heads of an episode read interleaved pieces of the same QKV rows: give them workgroup ids 8 apart so that
  // they run on the same XCD at about the same time and share the lines in its L2 (ids go round-robin over 8 XCDs)
  const int H = d / HD, b = (blockIdx.x / (8 * H)) * 8 + blockIdx.x % 8, h = (blockIdx.x / 8) % H;
  if (b >= g.B) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwave = nthr >> 6;
  const int n_t = g.n_td + g.n_th;
  const int qld = KVc ? d : 3 * d;
  // ordered compaction of the key rows: context points in slot order, then selected targets
  if (tid == 0) s_base = KVc ? kcnt[2 * b] : 0;
  __syncthreads();
  for (int c0 = 0; c0 < (KVc ? 0 : g.N); c0 += nthr) {
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
    if (tid == 0) for (int w = 0; w < nwave; ++w) s_base += wave_cnt[w];
    __syncthreads();
  }
  const int n_ck = s_base;   // context keys
  __syncthreads();
  if (tid == 0) {
    int n = n_ck;
    if (KVc) n = kcnt[2 * b + 1];
    else
      for (int j = 0; j < n_t; ++j)
        if (!g.tmask || g.tmask[j]) keyrow[n++] = g.P + j;
    s_base = n;
  }
  __syncthreads();
  const int n_ak = s_base;   // all keys (queries see these)
  const long ep = (long)b * g.N;
  for (int i = tid; i < n_ak * HD; i += nthr) {
    int j = i / HD, c = i % HD;
    if (KVc) {
      const float *src = KVc + ((long)b * max_keys + j) * 2 * d + h * HD + c;
      Ks[j * HD + c] = src[0];
      Vs[j * HD + c] = src[d];
    } else {
      const float *src = QKV + (ep + keyrow[j]) * 3 * d + h * HD + c;
      Ks[j * HD + c] = src[d];
      Vs[j * HD + c] = src[2 * d];
    }
  }
  __syncthreads();
  const float scale = rsqrtf((float)HD) * 1.44269504088896340736f;   // 1/sqrt(hd) * log2(e)
  for (int row = tid; row < g.N; row += nthr) {
    const bool isq = row < g.P && !is_ctx(g, b, row);
    const int nk = isq ? n_ak : n_ck;
    float q[HD], o[HD];
    const float *qp = QKV + (ep + row) * qld + h * HD;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {            // rows are 16-byte aligned (d, HD multiples of 4)
      const float4 qv = *reinterpret_cast<const float4 *>(qp + c);
      q[c] = qv.x * scale; q[c + 1] = qv.y * scale; q[c + 2] = qv.z * scale; q[c + 3] = qv.w * scale;
      o[c] = o[c + 1] = o[c + 2] = o[c + 3] = 0.f;
    }
    // online softmax over the keys in groups of four: one running-max rescale per group, exponentials in base 2
    // (log2(e) is folded into the query scale) on the bare v_exp_f32 (arguments are <= 0; the library exp2f spends four
    // more instructions per call on the denormal range)
    float mx = -INFINITY, l = 0.f;
    int j = 0;
    for (; j + 4 <= nk; j += 4) {
      float sc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) t = fmaf(q[c], Ks[(j + u) * HD + c], t);
        sc[u] = t;
      }
      const float mn = fmaxf(fmaxf(mx, fmaxf(sc[0], sc[1])), fmaxf(sc[2], sc[3]));
      const float corr = __builtin_amdgcn_exp2f(mx - mn);
      float pr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) pr[u] = __builtin_amdgcn_exp2f(sc[u] - mn);
      l = l * corr + ((pr[0] + pr[1]) + (pr[2] + pr[3]));
#pragma unroll
      for (int c = 0; c < HD; ++c) {
        float acc = o[c] * corr;
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = fmaf(pr[u], Vs[(j + u) * HD + c], acc);
        o[c] = acc;
      }
      mx = mn;
    }
    for (; j < nk; ++j) {
      float t = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) t = fmaf(q[c], Ks[j * HD + c], t);
      const float mn = fmaxf(mx, t);
      const float corr = __builtin_amdgcn_exp2f(mx - mn), pw = __builtin_amdgcn_exp2f(t - mn);
      l = l * corr + pw;
#pragma unroll
      for (int c = 0; c < HD; ++c) o[c] = fmaf(pw, Vs[j * HD + c], o[c] * corr);
      mx = mn;
    }
    float inv = 1.f / l;
    float *op = Aout + (ep + row) * d + h * HD;
#pragma unroll
    for (int c = 0; c < HD; c += 4)
      *reinterpret_cast<float4 *>(op + c) = make_float4(o[c] * inv, o[c + 1] * inv, o[c + 2] * inv, o[c + 3] * inv);
  }
}

// ---- G5/G6 epilogue: out = LayerNorm(a + b) * w + bias  (post-norm, eps 1e-5).  One wave / row.
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float *__restrict__ a,
                                                            const float *__restrict__ b2,
                                                            const float *__restrict__ w,
                                                            const float *__restrict__ bias,
                                                            float *__restrict__ out, long rows, int d,
                                                            float *__restrict__ usave = nullptr) {
  const int lane = threadIdx.x & 63;
  long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float *pa = a + row * d, *pb = b2 + row * d;
  float v[8];  // d <= 512
  float s = 0.f;
  int n = 0;
  for (int c = lane; c < d; c += 64, ++n) { v[n] = pa[c] + pb[c]; s += v[n]; if (usave) usave[row * d + c] = v[n]; }
  const float mean = wave_sum(s) / d;
  float ss = 0.f;
  for (int i = 0; i < n; ++i) { float t = v[i] - mean; ss += t * t; }
  const float rstd = rsqrtf(wave_sum(ss) / d + 1e-5f);
  n = 0;
  for (int c = lane; c < d; c += 64, ++n) out[row * d + c] = (v[n] - mean) * rstd * w[c] + bias[c];
}

// Wide-row variant (d = 256 NV: 256, 512): one wave per row, NV float4 per lane (16 B / lane loads and stores), a wave
// walks the rows grid-stride -- the scalar 4 B / lane kernel above runs at 3.9 TB/s at d = 512, this one is HBM-bound.
template <int NV>
__global__ __launch_bounds__(256) void add_layernorm_wide_kernel(const float *__restrict__ a, const float *__restrict__ b2,
                                                                 const float *__restrict__ w, const float *__restrict__ bias,
                                                                 float *__restrict__ out, long rows, float *__restrict__ usave) {
  constexpr int d = 256 * NV;
  const int lane = threadIdx.x & 63;
  float4 wv[NV], bv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    wv[i] = *reinterpret_cast<const float4 *>(w + 4 * (lane + 64 * i));
    bv[i] = *reinterpret_cast<const float4 *>(bias + 4 * (lane + 64 * i));
  }
  for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float4 x = *reinterpret_cast<const float4 *>(a + row * d + 4 * (lane + 64 * i));
      const float4 y = *reinterpret_cast<const float4 *>(b2 + row * d + 4 * (lane + 64 * i));
      v[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
      if (usave) *reinterpret_cast<float4 *>(usave + row * d + 4 * (lane + 64 * i)) = v[i];
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / d;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float t0 = v[i].x - mean, t1 = v[i].y - mean, t2 = v[i].z - mean, t3 = v[i].w - mean;
      ss += (t0 * t0 + t1 * t1) + (t2 * t2 + t3 * t3);
    }
    const float rstd = rsqrtf(wave_sum(ss) / d + 1e-5f);
#pragma unroll
    for (int i = 0; i < NV; ++i)
      *reinterpret_cast<float4 *>(out + row * d + 4 * (lane + 64 * i)) =
          make_float4((v[i].x - mean) * rstd * wv[i].x + bv[i].x, (v[i].y - mean) * rstd * wv[i].y + bv[i].y,
                      (v[i].z - mean) * rstd * wv[i].z + bv[i].z, (v[i].w - mean) * rstd * wv[i].w + bv[i].w);
  }
}

// Narrow-row variant (d = 4 * LPR, LPR = 8 or 16 lanes per row, float4 per lane): a wave normalises 64 / LPR
// rows at a time and the row statistics are LPR-lane xor reductions.  At d = 32 the one-wave-per-row kernel above
// leaves half of the lanes idle and runs at ~1 TB/s; this one is HBM-bound.
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}
template <int LPR>
__global__ __launch_bounds__(256) void add_layernorm_narrow_kernel(const float *__restrict__ a,
                                                                   const float *__restrict__ b2,
                                                                   const float *__restrict__ w,
                                                                   const float *__restrict__ bias,
                                                                   float *__restrict__ out, long rows,
                                                                   float *__restrict__ usave) {
  constexpr int d = 4 * LPR, RPB = 256 / LPR;
  const int sub = threadIdx.x % LPR;
  const float4 wv = *reinterpret_cast<const float4 *>(w + 4 * sub), bv = *reinterpret_cast<const float4 *>(bias + 4 * sub);
  for (long row = (long)blockIdx.x * RPB + threadIdx.x / LPR; row < rows; row += (long)gridDim.x * RPB) {
    const float4 x = *reinterpret_cast<const float4 *>(a + row * d + 4 * sub);
    const float4 y = *reinterpret_cast<const float4 *>(b2 + row * d + 4 * sub);
    float4 v = {x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w};
    if (usave) *reinterpret_cast<float4 *>(usave + row * d + 4 * sub) = v;
    const float mean = row_sum<LPR>(v.x + v.y + v.z + v.w) * (1.f / d);
    v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
    const float rstd = rsqrtf(row_sum<LPR>(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w) * (1.f / d) + 1e-5f);
    float4 o = {v.x * rstd * wv.x + bv.x, v.y * rstd * wv.y + bv.y, v.z * rstd * wv.z + bv.z, v.w * rstd * wv.w + bv.w};
    *reinterpret_cast<float4 *>(out + row * d + 4 * sub) = o;
  }
}

// ---- G7/G8 (+G11): acquisition softmax over the remaining queries, design selection, and (rollout
// API) the role update that replaces Task.update_batch (tasks/base_task.py:133-154).
// hid [B*P, F] = relu(z_q W1^T + b1) from the GEMM; logits = hid . w2 + b2 (model/head.py:27-33).
// One workgroup per episode.  zt / idx use the reference's *compacted* query numbering
// (order-preserving boolean-mask compaction, base_task.py:114-117).
struct SelectArgs {
  Geo g;
  int F;
  const float *hid, *w2, *b2;
  const float *logits; int logit_stride;   // precomputed logits[b * logit_stride + p] (wide path / fused head GEMM), or null
  int logit_nblk; long logit_blk_stride;   // ... as the sum of logit_nblk partial arrays logit_blk_stride apart
  int mode;                      // ALINE_SELECT_*
  const float *uniform;          // [B]
  const int64_t *forced; int forced_stride;   // forced[b * stride]
  int64_t *idx; int idx_stride;  // idx[b * stride]
  int *slot; int slot_stride;    // chosen slot (may be null)
  float *log_prob; int lp_stride;
  float *zt; int zt_stride;      // zt[b * stride + i], zero padded up to zt_width
  int zt_width;
  int *role_out;                 // rollout: role[b, chosen] = n_ctx_now + 1
  unsigned *range_flag;          // f16 range guard (common.h): raised when the logits are not finite (may be null)
};

__global__ __launch_bounds__(256) void acq_select_kernel(SelectArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float *logit = reinterpret_cast<float *>(smem_raw);   // [P]
  int *qslot = reinterpret_cast<int *>(logit + a.g.P);  // [P] compacted -> slot
  float *prob = logit + 2 * a.g.P;                      // [P] probabilities in compacted order
  __shared__ float red[4];
  __shared__ int wave_cnt[4];
  __shared__ int s_base, s_choice;
  __shared__ float s_val;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = a.g.P;
  // logits: one wave per row, lanes over F (or precomputed by the wide path)
  if (a.logits) {
    for (int p = tid; p < P; p += 256) {
      float s = a.logits[(long)b * a.logit_stride + p];
      for (int k = 1; k < a.logit_nblk; ++k) s += a.logits[k * a.logit_blk_stride + (long)b * a.logit_stride + p];
      logit[p] = s;
    }
  } else
  for (int p = wave; p < P; p += 4) {
    const float *hp = a.hid + ((long)b * P + p) * a.F;
    float s = 0.f;
    for (int f = lane; f < a.F; f += 64) s = fmaf(hp[f], a.w2[f], s);
    s = wave_sum(s);
    if (lane == 0) logit[p] = s + a.b2[0];
  }
  if (tid == 0) s_base = 0;
  __syncthreads();
  // ordered compaction of the remaining queries
  for (int c0 = 0; c0 < P; c0 += 256) {
    int p = c0 + tid;
    bool isq = p < P && !is_ctx(a.g, b, p);
    unsigned long long bal = __ballot(isq);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (isq) qslot[off + __popcll(bal & ((1ull << lane) - 1ull))] = p;
    __syncthreads();
    if (tid == 0) s_base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  const int nq = s_base;
  // softmax over the nq remaining queries (nn.Softmax(dim=-1), head.py:32)
  float mx = -INFINITY;
  for (int i = tid; i < nq; i += 256) mx = fmaxf(mx, logit[qslot[i]]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int i = tid; i < nq; i += 256) sum += __expf(logit[qslot[i]] - mx);
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  sum = red[0] + red[1] + red[2] + red[3];
  if (tid == 0 && !(sum <= 3.4e38f)) range_raise(a.range_flag, ALINE_RANGE_ACT);     // a NaN / +inf logit (fmaxf drops NaNs, the sum does not)
  const float inv = 1.f / sum;
  __syncthreads();
  // probabilities in compacted order (any P that fits LDS: the evaluation protocol runs n_query = 2000, README.md:45) and out to zt
  for (int i = tid; i < nq; i += 256) prob[i] = __expf(logit[qslot[i]] - mx) * inv;
  __syncthreads();
  if (a.zt)
    for (int i = tid; i < a.zt_width; i += 256)
      a.zt[(long)b * a.zt_stride + i] = i < nq ? prob[i] : 0.f;
  // selection (single wave: short and deterministic)
  if (wave == 0) {
    int choice = 0;
    float val = 0.f;
    if (a.mode == 0) {          // argmax, first maximal index (torch.max semantics)
      float best = -1.f; int bi = 0;
      for (int i = lane; i < nq; i += 64) if (prob[i] > best) { best = prob[i]; bi = i; }
      for (int o = 32; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o, 64); int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      choice = bi; val = best;
    } else if (a.mode == 2) {
      choice = (int)a.forced[(long)b * a.forced_stride];
      choice = min(max(choice, 0), nq - 1);
      val = prob[choice];
    } else {                    // inverse CDF of Categorical(probs = zt / sum zt)
      float tot = 0.f;
      for (int i = lane; i < nq; i += 64) tot += prob[i];
      tot = wave_sum(tot);
      const float u = a.uniform[b] * tot;
      // serial scan by chunks of 64 with a wave prefix sum
      float run = 0.f; int found = nq - 1; bool done = false;
      for (int c0 = 0; c0 < nq && !done; c0 += 64) {
        int i = c0 + lane;
        float v = i < nq ? prob[i] : 0.f, incl = v;
        for (int o = 1; o < 64; o <<= 1) { float t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        bool hit = i < nq && (run + incl) > u;
        unsigned long long bal = __ballot(hit);
        if (bal) { found = c0 + __ffsll((long long)bal) - 1; done = true; }
        run += __shfl(incl, 63, 64);
      }
      choice = found; val = prob[choice];
      // Categorical(probs).log_prob uses probs / probs.sum() clamped to [eps, 1-eps]
      val = val / tot;
    }
    if (a.mode == 2) {
      float tot = 0.f;
      for (int i = lane; i < nq; i += 64) tot += prob[i];
      tot = wave_sum(tot);
      val = val / tot;
    }
    if (lane == 0) { s_choice = choice; s_val = val; }
  }
  __syncthreads();
  if (tid == 0) {
    const int choice = s_choice;
    float v = s_val;
    if (a.mode != 0) v = fminf(fmaxf(v, 1.1920929e-07f), 1.f - 1.1920929e-07f);
    if (a.idx) a.idx[(long)b * a.idx_stride] = choice;
    if (a.log_prob) a.log_prob[(long)b * a.lp_stride] = logf(v);
    const int sl = qslot[choice];
    if (a.slot) a.slot[(long)b * a.slot_stride] = sl;
    if (a.role_out) a.role_out[(long)b * P + sl] = (P - nq) + 1;
  }
}

// The same selection for P <= 256 points with precomputed logits (the s3 / x3 / x5 rollouts): ONE WAVE per episode, four episodes per workgroup,
// no LDS and no barrier -- lane l holds the points l, l + 64, l + 128, l + 192; the compacted (remaining-query) order is the point order, so
// ballots give every query its place and the inverse-CDF scan walks the four 64-point chunks with a wave prefix sum.
__global__ __launch_bounds__(256) void acq_select_wave_kernel(SelectArgs a) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.g.B) return;
  const int P = a.g.P;
  const unsigned long long below = (1ull << lane) - 1ull;
  float lg[4], pr[4];
  bool isq[4];
  int ci[4], nq = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int p = 64 * c + lane;
    const bool valid = p < P;
    float s = -INFINITY;
    if (valid) {
      s = a.logits[(long)b * a.logit_stride + p];
      for (int k = 1; k < a.logit_nblk; ++k) s += a.logits[k * a.logit_blk_stride + (long)b * a.logit_stride + p];
    }
    lg[c] = s;
    isq[c] = valid && !is_ctx(a.g, b, p);
    const unsigned long long bal = __ballot(isq[c]);
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
    const float u = a.uniform[b] * tot;
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
  // probability and point slot of the chosen query (one lane holds it)
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

// (G9/G10: the second layers of the C GMM heads are reduced in the GEMM epilogue, gemm.h `red_*`; the parameter
// maps mean_c = raw[c][0], std_c = softplus(raw[c][1]) + std_min, weight = softmax_c(raw[c][2]) -- the reference's
// stack/movedim/flatten/chunk, head.py:264-265, with dim_y == 1 -- and compute_ll are img::gmm_raw_finish_kernel.)

// compute_ll on caller-provided GMM parameters (utils/eval.py:200-207).  One wave per row.
__global__ __launch_bounds__(256) void compute_ll_kernel(const float *__restrict__ value,
                                                         const float *__restrict__ means,
                                                         const float *__restrict__ stds,
                                                         const float *__restrict__ weights, long rows,
                                                         int C, float *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float m2 = -INFINITY, se = 0.f;
  const float v = value[row];
  float lp = -INFINITY;
  // C <= 64 in practice; loop for generality with an online logsumexp across chunks
  for (int c0 = 0; c0 < C; c0 += 64) {
    int c = c0 + lane;
    lp = -INFINITY;
    if (c < C) {
      float mu = means[row * C + c], sd = stds[row * C + c], w = weights[row * C + c];
      float z = (v - mu) / sd;
      lp = -0.5f * z * z - logf(sd) - 0.91893853320467274178f + logf(w);
    }
    float cm = wave_max(lp);
    float mn = fmaxf(m2, cm);
    float cs = wave_sum(c < C ? __expf(lp - mn) : 0.f);
    se = se * __expf(m2 - mn) + cs;
    m2 = mn;
  }
  if (lane == 0) out[row] = m2 + logf(se);
}

// rollout helpers -------------------------------------------------------------------------------
__global__ void role_init_kernel(int *role, int B, int P, int n_ctx0) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * P) return;
  int p = i % P;
  role[i] = p < n_ctx0 ? p + 1 : 0;
}

// Task.update_batch view of the static-slot state (tasks/base_task.py:133-154): context in order of
// entry, queries in slot order.  One workgroup per episode.
__global__ __launch_bounds__(256) void rollout_export_kernel(const int *role, const float *px,
                                                             const float *py, int B, int P, int n_ctx,
                                                             int dx, int dy, float *cx, float *cy,
                                                             float *qx, float *qy) {
  __shared__ int wave_cnt[4];
  __shared__ int s_base;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_base = 0;
  __syncthreads();
  const int nq = P - n_ctx;
  for (int c0 = 0; c0 < P; c0 += 256) {
    int p = c0 + tid;
    int r = p < P ? role[(long)b * P + p] : -1;
    if (r > 0 && r <= n_ctx) {
      for (int k = 0; k < dx; ++k) cx[((long)b * n_ctx + r - 1) * dx + k] = px[((long)b * P + p) * dx + k];
      for (int k = 0; k < dy; ++k) cy[((long)b * n_ctx + r - 1) * dy + k] = py[((long)b * P + p) * dy + k];
    }
    bool isq = r == 0;
    unsigned long long bal = __ballot(isq);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (isq) {
      int i = off + __popcll(bal & ((1ull << lane) - 1ull));
      if (i < nq) {
        if (qx) for (int k = 0; k < dx; ++k) qx[((long)b * nq + i) * dx + k] = px[((long)b * P + p) * dx + k];
        if (qy) for (int k = 0; k < dy; ++k) qy[((long)b * nq + i) * dy + k] = py[((long)b * P + p) * dy + k];
      }
    }
    __syncthreads();
    if (tid == 0) s_base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
}

// strided 2-D copy (packs the first `cols` columns of a [rows, ld] matrix)
__global__ void pack_cols_kernel(const float *src, int ld, int rows, int cols, float *dst) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * cols) return;
  dst[i] = src[(i / cols) * ld + (i % cols)];
}
