// Masked set-attention backward on the matrix pipe for the small-width model (d = 32, 4 heads of 8, <= 48 keys per instance):
//   P = softmax(Q K^T / sqrt(hd)) over the visible keys;  dV = P^T dO;  dS = P (dO V^T - delta);  dQ = dS K;  dK = dS^T Q
// (model/encoder.py:83-126 mask, nn.MultiheadAttention; `loss.backward()` of train_aline.py:124-132 through it).
// One workgroup per (step, episode) instance, one 16-row token tile per wave at a time, all products exact fp32
// (v_mfma_f32_16x16x4_f32) in the two register layouts of tail_bwd.h:
//   S^T, dP^T  [keys x rows]  = Kblk q^T, Vblk dO^T : q / dO in the T layout are the B operands, the K / V rows of the key tile
//                               (from LDS, the other heads' channels zeroed) the A operands; keys land on the register axis,
//                               so softmax and delta reduce over registers + the 4 lane groups;
//   dQ^T       [chan x rows]  = Kblk^T dS^T         : dS^T is, register for register, the B operand;
//   dV^T, dK^T [chan x keys]  = dO_N^T P_N, q_N^T dS_N (k = token row): N-layout operands; P^T / dS^T tiles go through a
//                               16 x 32 transpose in the wave's LDS scratch; accumulators stay resident over the wave's tiles.
// The VALU kernel it replaces (attention_bwd_kernel<8>) ran at 62 % of the fp32 vector peak: 3.5 ms per call at the headline shape.
#pragma once
#include "tail_bwd.h"

namespace abwd {

constexpr int D = 32, HD = 8, H = 4, KTMAX = 3, MAXK = 16 * KTMAX;
constexpr int PK = 36;
constexpr int WAVES = 4, THREADS = 64 * WAVES;

using fused::ld4;
using fused::group_sum;
using fused::group_max;
using fused::zero4;

__global__ __launch_bounds__(THREADS) void attention_bwd_mfma_kernel(Geo g, const float *__restrict__ QKV,
                                                                     const float *__restrict__ dA, float *__restrict__ dQKV) {
  __shared__ __attribute__((aligned(16))) float Ks[MAXK * PK], Vs[MAXK * PK], dKs[MAXK * D], dVs[MAXK * D], scrs[WAVES * 16 * PK];
  __shared__ int keyrow[MAXK];
  __shared__ int wave_cnt[WAVES];
  __shared__ int s_base;
  const int b = blockIdx.x;
  if (b >= g.B) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tok = lane & 15, gq = lane >> 4;
  const int n_t = g.n_td + g.n_th;
  // ---- key list: context points in slot order, then the visible targets -----------------------------------------
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < g.P; c0 += THREADS) {
    const int row = c0 + tid;
    const bool key = row < g.P && is_ctx(g, b, row);
    const unsigned long long bal = __ballot(key);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    const int k = off + __popcll(bal & ((1ull << lane) - 1ull));
    if (key && k < MAXK) keyrow[k] = row;
    __syncthreads();
    if (tid == 0) s_base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  const int n_ck = min(s_base, MAXK);
  __syncthreads();
  if (tid == 0) {
    int n = n_ck;
    for (int j = 0; j < n_t; ++j)
      if ((!g.tmask || g.tmask[j]) && n < MAXK) keyrow[n++] = g.P + j;
    s_base = n;
  }
  __syncthreads();
  const int n_ak = s_base;
  const int nkt = (n_ak + 15) >> 4;
  const long ep = (long)b * g.N;
  for (int i = tid; i < 16 * nkt * 8; i += THREADS) {
    const int j = i >> 3, c4 = (i & 7) * 4;
    f32x4 kv = zero4(), vv = zero4();
    if (j < n_ak) {
      const float *src = QKV + (ep + keyrow[j]) * 3 * D + c4;
      kv = ld4(src + D);
      vv = ld4(src + 2 * D);
    }
    *reinterpret_cast<f32x4 *>(Ks + j * PK + c4) = kv;
    *reinterpret_cast<f32x4 *>(Vs + j * PK + c4) = vv;
    *reinterpret_cast<f32x4 *>(dKs + j * D + c4) = zero4();
    *reinterpret_cast<f32x4 *>(dVs + j * D + c4) = zero4();
  }
  __syncthreads();

  const float scale = rsqrtf((float)HD);
  float *scr = scrs + wave * 16 * PK;
  f32x4 dKt[2][KTMAX], dVt[2][KTMAX];      // [chan 16 mt + 4 g + r][key 16 kt + tok], summed over this wave's tiles
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int kt = 0; kt < KTMAX; ++kt) { dKt[mt][kt] = zero4(); dVt[mt][kt] = zero4(); }

  const int ntile = (g.N + 15) >> 4;
  for (int tile = wave; tile < ntile; tile += WAVES) {
    const int row = tile * 16 + tok;
    const bool ok = row < g.N;
    const int rc = ok ? row : g.N - 1;
    const bool isq = ok && row < g.P && !is_ctx(g, b, row);
    const int nk = !ok ? 0 : (isq ? n_ak : n_ck);
    f32x4 q[2], go[2], dq[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      q[mt] = ld4(QKV + (ep + rc) * 3 * D + 16 * mt + 4 * gq) * scale;
      go[mt] = ok ? ld4(dA + (ep + rc) * D + 16 * mt + 4 * gq) : zero4();
      dq[mt] = zero4();
    }
    f32x4 qN[2], goN[2];
    tailbwd::to_n(qN, q[0], q[1], scr, tok, gq);
    tailbwd::to_n(goN, go[0], go[1], scr, tok, gq);
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const int mt = h >> 1, hg = h & 1;
      const bool mine_g = (gq >> 1) == hg;      // this lane group's channels 16 mt + 4 g + r belong to head h (k axis)
      const bool mine_c = (tok >> 3) == hg;     // channel 16 mt + tok belongs to head h (i axis)
      f32x4 s[KTMAX], dp[KTMAX];
#pragma unroll
      for (int kt = 0; kt < KTMAX; ++kt) {
        s[kt] = zero4(); dp[kt] = zero4();
        if (kt < nkt) {
          const f32x4 kf = mine_g ? ld4(Ks + (16 * kt + tok) * PK + 16 * mt + 4 * gq) : zero4();
          const f32x4 vf = mine_g ? ld4(Vs + (16 * kt + tok) * PK + 16 * mt + 4 * gq) : zero4();
#pragma unroll
          for (int r = 0; r < 4; ++r) { MFMAO(s[kt], kf[r], q[mt][r]); MFMAO(dp[kt], vf[r], go[mt][r]); }
        }
      }
      // softmax over the keys (register axis x lane groups), delta, score gradient
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < KTMAX; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) if (16 * kt + 4 * gq + r < nk) mx = fmaxf(mx, s[kt][r]);
      mx = group_max(mx);
      float l = 0.f, delta = 0.f;
#pragma unroll
      for (int kt = 0; kt < KTMAX; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = 16 * kt + 4 * gq + r < nk ? __expf(s[kt][r] - mx) : 0.f;
          s[kt][r] = e;
          l += e;
          delta = fmaf(e, dp[kt][r], delta);
        }
      l = group_sum(l);
      const float inv = l > 0.f ? 1.f / l : 0.f;
      delta = group_sum(delta) * inv;
#pragma unroll
      for (int kt = 0; kt < KTMAX; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = s[kt][r] * inv;
          s[kt][r] = p;
          dp[kt][r] = p * (dp[kt][r] - delta);      // dS^T
        }
      const f32x4 goA = mine_c ? goN[mt] : zero4(), qA = mine_c ? qN[mt] : zero4();
#pragma unroll
      for (int kt = 0; kt < KTMAX; ++kt) {
        if (kt < nkt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float ka = mine_c ? Ks[(16 * kt + 4 * gq + r) * PK + 16 * mt + tok] : 0.f;
            MFMAO(dq[mt], ka, dp[kt][r]);
          }
          f32x4 pn[2];
          tailbwd::to_n(pn, s[kt], dp[kt], scr, tok, gq);      // pn[0] = P_N, pn[1] = dS_N  (key on lane, row on (g, r))
#pragma unroll
          for (int r = 0; r < 4; ++r) { MFMAO(dVt[mt][kt], goA[r], pn[0][r]); MFMAO(dKt[mt][kt], qA[r], pn[1][r]); }
        }
      }
    }
    if (ok) {
      float *out = dQKV + (ep + row) * 3 * D + 16 * 0 + 4 * gq;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        *reinterpret_cast<f32x4 *>(out + 16 * mt) = dq[mt] * scale;
        *reinterpret_cast<f32x4 *>(out + D + 16 * mt) = zero4();       // K / V gradients of key rows are written below
        *reinterpret_cast<f32x4 *>(out + 2 * D + 16 * mt) = zero4();
      }
    }
  }
  // ---- dK / dV of the key rows: sum over the waves through LDS -----------------------------------------------------
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int kt = 0; kt < KTMAX; ++kt)
      if (kt < nkt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          atomicAdd(&dKs[(16 * kt + tok) * D + 16 * mt + 4 * gq + r], dKt[mt][kt][r]);
          atomicAdd(&dVs[(16 * kt + tok) * D + 16 * mt + 4 * gq + r], dVt[mt][kt][r]);
        }
      }
  __syncthreads();      // (also orders the zero stores of every wave's tiles before the key-row stores)
  for (int i = tid; i < n_ak * 8; i += THREADS) {
    const int j = i >> 3, c4 = (i & 7) * 4;
    float *dst = dQKV + (ep + keyrow[j]) * 3 * D + c4;
    *reinterpret_cast<f32x4 *>(dst + D) = ld4(dKs + j * D + c4);          // q already carries the 1 / sqrt(hd)
    *reinterpret_cast<f32x4 *>(dst + 2 * D) = ld4(dVs + j * D + c4);
  }
}


// ---- the whole attention block: in-projection + attention, backward ----------------------------------------------------------
//   q = Wq x + bq,  K / V = Wk / Wv x_key + b  (key rows only),  a = attention(q, K, V)
//   in   x = layer input [M, 32], dA = dLoss/da, dX = dLoss/du1 (the residual branch, written by the tail kernel)
//   out  dX += Wq^T dq  (every row)  + Wk^T dK + Wv^T dV  (key rows);  dWin, dbin accumulated into (+=)
// Neither QKV nor dQKV ([M, 96] each) exist: the per-op pipeline wrote / read them five times per layer (in-projection GEMM,
// attention backward, its dW and dX GEMMs).  Two kernels:
//   attn_block_bwd_kernel  persistent workgroups walk the instances; K / V of the instance's key rows come from the compact
//                          buffer of the forward recompute (one coalesced copy into LDS, one barrier), the token tiles run as
//                          described above with q recomputed and dX = dU1 + Wq^T dq, dWq in registers; dK / dV are summed over
//                          the waves in LDS and stored to a compact buffer [I * max_keys, 64].
//   kv_bwd_kernel          one wave per (instance, key tile): key rows' dX += Wk^T dK + Wv^T dV, dWk / dWv / biases.
// (The first version did the key list, the K / V projection and the key-row epilogue per instance inside the kernel: four
// dependent global round trips and seven barriers per instance, 1.69 of its 3.23 ms per call with the token tiles switched off.)
struct BlockArgs {
  Geo g;
  const float *X, *dA;
  float *dX;
  const float *win, *bin;      // in_proj_weight [96, 32], in_proj_bias [96]
  float *dwin, *dbin;
  const float *kvc;            // [I * max_keys, 64] K | V of the key rows, key-list order (forward recompute)
  float *dkvc;                 // [I * max_keys, 64] dK | dV sums of the key rows (attn_block_bwd_kernel -> kv_bwd_kernel)
  const int *keyidx, *kcnt;    // key_list_kernel: global token row of key j of instance b (-1 beyond), [I, 2] counts
  int max_keys;
  unsigned *dx_absmax;         // attn_block_bwd16_kernel, optional: max |dX| it writes, as bits (the next layer's tail16_kernel scales its dY by it; the
                               // key rows' K / V terms of kv_bwd_kernel come on top: the tile program's head room of 2^11 covers them)
  const unsigned *da_max_bits; // attn_block_bwd16_kernel: bits of max |dA| (tail16_kernel's Args.da_absmax): the scale of every gradient in it
};

__device__ __forceinline__ int load_role(const Geo &g, int b, int p) {
  if (g.inst_B > 0) return g.role[(long)(b % g.inst_B) * g.P + p];
  return g.role ? g.role[(long)b * g.P + p] : (p < g.n_ctx ? 1 : 0);
}
__device__ __forceinline__ bool role_is_ctx(const Geo &g, int b, int r) {
  if (g.inst_B > 0) return r > 0 && r <= g.n_ctx0 + g.inst_t0 + b / g.inst_B;
  return r > 0;
}

// LDS (floats): Wq image [32][36] | bq [32] | Ks, Vs, zeros [3][16 KT][36] | wave scratch [4][16][36] | dK / dV slots [4][2][16 KT][36]
constexpr int block_lds_floats(int KT) { return D * PK + D + 3 * 16 * KT * PK + WAVES * 16 * PK + WAVES * 2 * 16 * KT * PK; }

template <int KT>
__global__ __launch_bounds__(THREADS, KT <= 2 ? 2 : 1) void attn_block_bwd_kernel(BlockArgs a) {
  constexpr int MK = 16 * KT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *const Wi = lds, *const bi = Wi + D * PK, *const Ks = bi + D, *const Vs = Ks + MK * PK, *const Zs = Vs + MK * PK,
               *const scrs = Zs + MK * PK, *const slots = scrs + WAVES * 16 * PK;      // slots [wave][dK | dV][MK][PK]
  static_assert(2 * MK * PK >= D * D + D, "gradient staging reuses the key-row arrays");
  const Geo &g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tok = lane & 15, gq = lane >> 4;
  for (int i = tid; i < D * D; i += THREADS) Wi[(i >> 5) * PK + (i & 31)] = a.win[i];
  if (tid < D) bi[tid] = a.bin[tid];
  for (int i = tid; i < MK * PK; i += THREADS) Zs[i] = 0.f;
  // scores in base-2 units: q carries 1 / sqrt(hd) and log2(e), the softmax is exp2(s - max); dQ gets the plain
  // 1 / sqrt(hd) at the end and the dK sums (products with this q) are multiplied by ln 2 when they leave the registers
  const float scale = rsqrtf((float)HD), scale2 = scale * 1.44269504088896340736f, ln2 = 0.69314718055994530942f;
  float *scr = scrs + wave * 16 * PK;
  f32x4 gWq[2][2];
  float gBq[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) gWq[i][j] = zero4();
  const int ntile = (g.N + 15) >> 4;

  // K | V of an instance's key rows: 16 KT rows of 16 float4, KT per thread, fetched one instance ahead.  Rows beyond the
  // instance's keys hold the projection of a zero row (finite) and are masked; rows beyond max_keys are clamped.
  f32x4 nkv[KT];
  auto load_kv = [&](int b) {
#pragma unroll
    for (int u = 0; u < KT; ++u) {
      const int i = tid + u * THREADS, j = min(i >> 4, a.max_keys - 1), c4 = (i & 15) * 4;
      nkv[u] = ld4(a.kvc + ((long)b * a.max_keys + j) * 2 * D + c4);
    }
  };
  if ((int)blockIdx.x < g.B) load_kv(blockIdx.x);
  for (int b = blockIdx.x; b < g.B; b += gridDim.x) {
    const long ep = (long)b * g.N;
    // first tile of this wave: its loads fly while K / V arrive
    f32x4 nx[2], ngo[2];
    int nrole = 0;
    {
      const int row = min(wave * 16 + tok, g.N - 1);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        nx[mt] = ld4(a.X + (ep + row) * D + 16 * mt + 4 * gq);
        ngo[mt] = ld4(a.dA + (ep + row) * D + 16 * mt + 4 * gq);
      }
      if (row < g.P) nrole = load_role(g, b, row);
    }
    const int n_ck = min(a.kcnt[2 * b], MK), n_ak = min(a.kcnt[2 * b + 1], MK);
    const int nkt = (n_ak + 15) >> 4;
    __syncthreads();      // the previous instance is done with the arrays
#pragma unroll
    for (int u = 0; u < KT; ++u) {      // (loaded while the previous instance was computed)
      const int i = tid + u * THREADS, j = i >> 4, c4 = (i & 15) * 4;
      *reinterpret_cast<f32x4 *>((c4 < D ? Ks : Vs - D) + j * PK + c4) = nkv[u];
    }
    __syncthreads();
    if (b + (int)gridDim.x < g.B) load_kv(b + gridDim.x);

    f32x4 dKt[2][KT], dVt[2][KT];      // [chan 16 mt + 4 g + r][key 16 kt + tok], summed over this wave's tiles
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) { dKt[mt][kt] = zero4(); dVt[mt][kt] = zero4(); }
    for (int tile = wave; tile < ntile; tile += WAVES) {
      const int row = tile * 16 + tok;
      const bool ok = row < g.N;
      const int rc = ok ? row : g.N - 1;
      const bool isq = ok && row < g.P && !role_is_ctx(g, b, nrole);
      const int nk = isq ? n_ak : n_ck;      // (padding rows: dO = 0, they contribute nothing)
      f32x4 kmask[KT];                       // 0 on the visible keys of this row, -inf elsewhere: initial value of the scores
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) kmask[kt][r] = 16 * kt + 4 * gq + r < nk ? 0.f : -INFINITY;
      f32x4 x[2], q[2], go[2], dq[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        x[mt] = nx[mt];
        go[mt] = ok ? ngo[mt] : zero4();
        q[mt] = ld4(bi + 16 * mt + 4 * gq);
        dq[mt] = zero4();
      }
      if (tile + WAVES < ntile) {      // the next tile's rows
        const int nr = min(row + 16 * WAVES, g.N - 1);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          nx[mt] = ld4(a.X + (ep + nr) * D + 16 * mt + 4 * gq);
          ngo[mt] = ld4(a.dA + (ep + nr) * D + 16 * mt + 4 * gq);
        }
        nrole = nr < g.P ? load_role(g, b, nr) : 0;
      }
      tailbwd::mm_fwd<2, 2>(q, Wi, PK, x, tok, gq);
      q[0] *= scale2; q[1] *= scale2;
      f32x4 qN[2], goN[2];
      tailbwd::to_n(qN, q[0], q[1], scr, tok, gq);
      tailbwd::to_n(goN, go[0], go[1], scr, tok, gq);
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const int mt = h >> 1, hg = h & 1;
        const bool mine_g = (gq >> 1) == hg;      // this lane group's channels 16 mt + 4 g + r belong to head h (k axis)
        const bool mine_c = (tok >> 3) == hg;     // channel 16 mt + tok belongs to head h (i axis)
        // operands of the other heads come from the block of zeros: an address select instead of a branch around the loads
        const float *Kg = mine_g ? Ks : Zs, *Vg = mine_g ? Vs : Zs, *Kc = mine_c ? Ks : Zs;
        f32x4 s[KT], dp[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          s[kt] = kmask[kt]; dp[kt] = zero4();
          if (kt < nkt) {
            const f32x4 kf = ld4(Kg + (16 * kt + tok) * PK + 16 * mt + 4 * gq);
            const f32x4 vf = ld4(Vg + (16 * kt + tok) * PK + 16 * mt + 4 * gq);
            s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[0], q[mt][0], kmask[kt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0x7F6);
            dp[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[0], go[mt][0], zero4(), 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0x7F6);
#pragma unroll
            for (int r = 1; r < 4; ++r) { MFMAO(s[kt], kf[r], q[mt][r]); MFMAO(dp[kt], vf[r], go[mt][r]); }
          }
        }
        float mx = -3.0e38f;      // (a row without a visible key: every exp2 below is 0, inv = 0)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
        mx = group_max(mx);
        float l = 0.f, delta = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(s[kt][r] - mx);
            s[kt][r] = e;
            l += e;
            delta = fmaf(e, dp[kt][r], delta);
          }
        l = group_sum(l);
        const float inv = l > 0.f ? __builtin_amdgcn_rcpf(l) : 0.f;
        delta = group_sum(delta) * inv;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = s[kt][r] * inv;
            s[kt][r] = p;
            dp[kt][r] = p * (dp[kt][r] - delta);      // dS^T
          }
        const f32x4 goA = mine_c ? goN[mt] : zero4(), qA = mine_c ? qN[mt] : zero4();
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          if (kt < nkt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) MFMAO(dq[mt], Kc[(16 * kt + 4 * gq + r) * PK + 16 * mt + tok], dp[kt][r]);
            f32x4 pn[2];
            tailbwd::to_n(pn, s[kt], dp[kt], scr, tok, gq);      // pn[0] = P_N, pn[1] = dS_N  (key on lane, row on (g, r))
#pragma unroll
            for (int r = 0; r < 4; ++r) { MFMAO(dVt[mt][kt], goA[r], pn[0][r]); MFMAO(dKt[mt][kt], qA[r], pn[1][r]); }
          }
        }
      }
      dq[0] *= scale; dq[1] *= scale;
      f32x4 dxo[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) dxo[mt] = ok ? ld4(a.dX + (ep + rc) * D + 16 * mt + 4 * gq) : zero4();
      tailbwd::mm_bwd<2, 2>(dxo, Wi, PK, dq, tok, gq);          // dx = du1 + Wq^T dq
      if (ok) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) *reinterpret_cast<f32x4 *>(a.dX + (ep + row) * D + 16 * mt + 4 * gq) = dxo[mt];
      }
      f32x4 dqN[2], xN[2];
      tailbwd::to_n(dqN, dq[0], dq[1], scr, tok, gq);
      tailbwd::to_n(xN, x[0], x[1], scr, tok, gq);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        tailbwd::mm_dw(gWq[i][0], dqN[i], xN[0]);
        tailbwd::mm_dw(gWq[i][1], dqN[i], xN[1]);
        gBq[i] += tailbwd::sum4(dqN[i]);
      }
    }
    // ---- dK / dV of the key rows: every wave leaves its partial sums in its own LDS slot (plain 16-byte stores: LDS float
    // atomics retire a few lanes per cycle), the store to the compact buffer adds the four slots -------------------------
    {
      float *my = slots + wave * 2 * MK * PK;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
          if (kt < nkt) {
            *reinterpret_cast<f32x4 *>(my + (16 * kt + tok) * PK + 16 * mt + 4 * gq) = dKt[mt][kt] * ln2;
            *reinterpret_cast<f32x4 *>(my + MK * PK + (16 * kt + tok) * PK + 16 * mt + 4 * gq) = dVt[mt][kt];
          }
    }
    __syncthreads();
    for (int i = tid; i < n_ak * 16; i += THREADS) {
      const int j = i >> 4, c4 = (i & 15) * 4;
      const float *src = slots + (c4 < D ? 0 : MK * PK - D) + j * PK + c4;
      f32x4 v = ld4(src);
#pragma unroll
      for (int w = 1; w < WAVES; ++w) v += ld4(src + w * 2 * MK * PK);
      *reinterpret_cast<f32x4 *>(a.dkvc + ((long)b * a.max_keys + j) * 2 * D + c4) = v;
    }
  }
  // ---- the workgroup's Wq gradients: LDS staging, then one atomic per element ---------------------------------------
  __syncthreads();
  float *stg = Ks;      // Wq [32][32], then bq [32]
  for (int i = tid; i < D * D + D; i += THREADS) stg[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&stg[(16 * i + 4 * gq + r) * D + 16 * j + tok], gWq[i][j][r]);
    atomicAdd(&stg[D * D + 16 * i + tok], gBq[i]);
  }
  __syncthreads();
  for (int i = tid; i < D * D; i += THREADS) unsafeAtomicAdd(a.dwin + i, stg[i]);
  if (tid < D) unsafeAtomicAdd(a.dbin + tid, stg[D * D + tid]);
}

// ---- round 4: the same kernel on the f16 matrix pipe (tail_bwd.h: tail16_kernel has the scheme) ---------------------------------------
// Every group of four v_mfma_f32_16x16x4_f32 over a 16 x 16 operand block becomes the 3-term f16 split on v_mfma_f32_16x16x16_f16.
// K and V of the instance sit in LDS as (hi | lo) quads of four consecutive channels (the A operands of S^T and dP^T: one ds_read_b128
// as before), K a second time in fp32 (the K^T operand of dQ^T is read down a column and split in registers); Wq and Wq^T as packed
// images (x 2^8); q, dO, P, dS and the N-layout operands are split where they are produced.  dO is multiplied by the power of two of
// max |dA| (tail16_kernel reduces what it writes: BlockArgs.da_max_bits) at the load -- the backward is linear in it --, and dX, the
// dK / dV sums and the Wq gradients are divided by it where they leave.
// LDS (floats): Wq image [32][36] | Wq^T image | bq [32] | Kp, Vp, Kf, zeros [4][16 KT][36] | wave scratch [4][16][36] | dK / dV slots [4][2][16 KT][36]
constexpr int block16_lds_floats(int KT) { return 2 * D * PK + D + 4 * 16 * KT * PK + WAVES * 16 * PK + WAVES * 2 * 16 * KT * PK; }

template <int KT>
__global__ __launch_bounds__(THREADS, KT <= 2 ? 2 : 1) void attn_block_bwd16_kernel(BlockArgs a) {
  using tailbwd::H8;
  using tailbwd::split4;
  using tailbwd::as_h8;
  constexpr int MK = 16 * KT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *const Wi = lds, *const WiT = Wi + D * PK, *const bi = WiT + D * PK, *const Ks = bi + D, *const Vs = Ks + MK * PK, *const Kf = Vs + MK * PK,
               *const Zs = Kf + MK * PK, *const scrs = Zs + MK * PK, *const slots = scrs + WAVES * 16 * PK;      // slots [wave][dK | dV][MK][PK]
  static_assert(2 * MK * PK >= D * D + D, "gradient staging reuses the key-row arrays");
  const Geo &g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tok = lane & 15, gq = lane >> 4;
  tailbwd::pack_image16(Wi, PK, a.win, D, D, D, 1, tid, THREADS);
  tailbwd::pack_image16(WiT, PK, a.win, D, D, 1, D, tid, THREADS);
  if (tid < D) bi[tid] = a.bin[tid];
  float ginv, dx_max = 0.f;
  const float gs = tailbwd::grad_scale16(*a.da_max_bits, ginv);
  for (int i = tid; i < MK * PK; i += THREADS) Zs[i] = 0.f;
  // scores in base-2 units: q carries 1 / sqrt(hd) and log2(e), the softmax is exp2(s - max); dQ gets the plain
  // 1 / sqrt(hd) at the end and the dK sums (products with this q) are multiplied by ln 2 when they leave the registers
  const float scale = rsqrtf((float)HD), scale2 = scale * 1.44269504088896340736f, ln2 = 0.69314718055994530942f;
  float *scr = scrs + wave * 16 * PK;
  f32x4 gWq[2][2];
  float gBq[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) gWq[i][j] = zero4();
  const int ntile = (g.N + 15) >> 4;

  // K | V of an instance's key rows: 16 KT rows of 16 float4, KT per thread, fetched one instance ahead.  Rows beyond the
  // instance's keys hold the projection of a zero row (finite) and are masked; rows beyond max_keys are clamped.
  f32x4 nkv[KT];
  auto load_kv = [&](int b) {
#pragma unroll
    for (int u = 0; u < KT; ++u) {
      const int i = tid + u * THREADS, j = min(i >> 4, a.max_keys - 1), c4 = (i & 15) * 4;
      nkv[u] = ld4(a.kvc + ((long)b * a.max_keys + j) * 2 * D + c4);
    }
  };
  if ((int)blockIdx.x < g.B) load_kv(blockIdx.x);
  for (int b = blockIdx.x; b < g.B; b += gridDim.x) {
    const long ep = (long)b * g.N;
    // first tile of this wave: its loads fly while K / V arrive
    f32x4 nx[2], ngo[2];
    int nrole = 0;
    {
      const int row = min(wave * 16 + tok, g.N - 1);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        nx[mt] = ld4(a.X + (ep + row) * D + 16 * mt + 4 * gq);
        ngo[mt] = ld4(a.dA + (ep + row) * D + 16 * mt + 4 * gq);
      }
      if (row < g.P) nrole = load_role(g, b, row);
    }
    const int n_ck = min(a.kcnt[2 * b], MK), n_ak = min(a.kcnt[2 * b + 1], MK);
    const int nkt = (n_ak + 15) >> 4;
    __syncthreads();      // the previous instance is done with the arrays
#pragma unroll
    for (int u = 0; u < KT; ++u) {      // (loaded while the previous instance was computed)
      const int i = tid + u * THREADS, j = i >> 4, c4 = (i & 15) * 4;
      const H8 pk = split4(nkv[u]);
      typedef unsigned u2 __attribute__((ext_vector_type(2)));
      const u2 ph = __builtin_bit_cast(u2, pk.hi), pl = __builtin_bit_cast(u2, pk.lo);
      *reinterpret_cast<f32x4 *>((c4 < D ? Ks : Vs - D) + j * PK + c4) = (f32x4){__uint_as_float(ph[0]), __uint_as_float(ph[1]), __uint_as_float(pl[0]), __uint_as_float(pl[1])};
      if (c4 < D) *reinterpret_cast<f32x4 *>(Kf + j * PK + c4) = nkv[u];
    }
    __syncthreads();
    if (b + (int)gridDim.x < g.B) load_kv(b + gridDim.x);

    f32x4 dKt[2][KT], dVt[2][KT];      // [chan 16 mt + 4 g + r][key 16 kt + tok], summed over this wave's tiles
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) { dKt[mt][kt] = zero4(); dVt[mt][kt] = zero4(); }
    for (int tile = wave; tile < ntile; tile += WAVES) {
      const int row = tile * 16 + tok;
      const bool ok = row < g.N;
      const int rc = ok ? row : g.N - 1;
      const bool isq = ok && row < g.P && !role_is_ctx(g, b, nrole);
      const int nk = isq ? n_ak : n_ck;      // (padding rows: dO = 0, they contribute nothing)
      f32x4 kmask[KT];                       // 0 on the visible keys of this row, -inf elsewhere: initial value of the scores
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) kmask[kt][r] = 16 * kt + 4 * gq + r < nk ? 0.f : -INFINITY;
      f32x4 x[2], q[2], go[2], dq[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        x[mt] = nx[mt];
        go[mt] = ok ? ngo[mt] * gs : zero4();
        q[mt] = zero4();
        dq[mt] = zero4();
      }
      if (tile + WAVES < ntile) {      // the next tile's rows
        const int nr = min(row + 16 * WAVES, g.N - 1);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          nx[mt] = ld4(a.X + (ep + nr) * D + 16 * mt + 4 * gq);
          ngo[mt] = ld4(a.dA + (ep + nr) * D + 16 * mt + 4 * gq);
        }
        nrole = nr < g.P ? load_role(g, b, nr) : 0;
      }
      const H8 xS[2] = {split4(x[0]), split4(x[1])};
      tailbwd::mm_fwd16<2, 2>(q, Wi, PK, xS, tok, gq);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) q[mt] = (q[mt] * tailbwd::WINV16 + ld4(bi + 16 * mt + 4 * gq)) * scale2;
      const H8 qS[2] = {split4(q[0]), split4(q[1])}, goS[2] = {split4(go[0]), split4(go[1])};
      f32x4 qN[2], goN[2];
      tailbwd::to_n(qN, q[0], q[1], scr, tok, gq);
      tailbwd::to_n(goN, go[0], go[1], scr, tok, gq);
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const int mt = h >> 1, hg = h & 1;
        const bool mine_g = (gq >> 1) == hg;      // this lane group's channels 16 mt + 4 g + r belong to head h (k axis)
        const bool mine_c = (tok >> 3) == hg;     // channel 16 mt + tok belongs to head h (i axis)
        // operands of the other heads come from the block of zeros: an address select instead of a branch around the loads
        const float *Kg = mine_g ? Ks : Zs, *Vg = mine_g ? Vs : Zs, *Kc = mine_c ? Kf : Zs;
        f32x4 s[KT], dp[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          s[kt] = kmask[kt]; dp[kt] = zero4();
          if (kt < nkt) {
            const H8 kf = as_h8(ld4(Kg + (16 * kt + tok) * PK + 16 * mt + 4 * gq));
            const H8 vf = as_h8(ld4(Vg + (16 * kt + tok) * PK + 16 * mt + 4 * gq));
            MFMA16O(s[kt], kf.lo, qS[mt].hi); MFMA16O(dp[kt], vf.lo, goS[mt].hi);
            MFMA16O(s[kt], kf.hi, qS[mt].lo); MFMA16O(dp[kt], vf.hi, goS[mt].lo);
            MFMA16O(s[kt], kf.hi, qS[mt].hi); MFMA16O(dp[kt], vf.hi, goS[mt].hi);
          }
        }
        float mx = -3.0e38f;      // (a row without a visible key: every exp2 below is 0, inv = 0)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
        mx = group_max(mx);
        float l = 0.f, delta = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(s[kt][r] - mx);
            s[kt][r] = e;
            l += e;
            delta = fmaf(e, dp[kt][r], delta);
          }
        l = group_sum(l);
        const float inv = l > 0.f ? __builtin_amdgcn_rcpf(l) : 0.f;
        delta = group_sum(delta) * inv;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = s[kt][r] * inv;
            s[kt][r] = p;
            dp[kt][r] = p * (dp[kt][r] - delta);      // dS^T
          }
        const H8 goA = split4(mine_c ? goN[mt] : zero4()), qA = split4(mine_c ? qN[mt] : zero4());
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          if (kt < nkt) {
            f32x4 kc;
#pragma unroll
            for (int r = 0; r < 4; ++r) kc[r] = Kc[(16 * kt + 4 * gq + r) * PK + 16 * mt + tok];
            const H8 kcS = split4(kc), dsS = split4(dp[kt]);
            f32x4 pn[2];
            tailbwd::to_n(pn, s[kt], dp[kt], scr, tok, gq);      // pn[0] = P_N, pn[1] = dS_N  (key on lane, row on (g, r))
            const H8 pS = split4(pn[0]), dsN = split4(pn[1]);
            MFMA16O(dq[mt], kcS.lo, dsS.hi); MFMA16O(dVt[mt][kt], goA.lo, pS.hi); MFMA16O(dKt[mt][kt], qA.lo, dsN.hi);
            MFMA16O(dq[mt], kcS.hi, dsS.lo); MFMA16O(dVt[mt][kt], goA.hi, pS.lo); MFMA16O(dKt[mt][kt], qA.hi, dsN.lo);
            MFMA16O(dq[mt], kcS.hi, dsS.hi); MFMA16O(dVt[mt][kt], goA.hi, pS.hi); MFMA16O(dKt[mt][kt], qA.hi, dsN.hi);
          }
        }
      }
      dq[0] *= scale; dq[1] *= scale;
      f32x4 dxo[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) dxo[mt] = ok ? ld4(a.dX + (ep + rc) * D + 16 * mt + 4 * gq) : zero4();
      {
        const H8 dqS[2] = {split4(dq[0]), split4(dq[1])};
        f32x4 acc[2] = {zero4(), zero4()};
        tailbwd::mm_fwd16<2, 2>(acc, WiT, PK, dqS, tok, gq);          // dx = du1 + Wq^T dq
        const float dsc = tailbwd::WINV16 * ginv;
        dxo[0] += acc[0] * dsc; dxo[1] += acc[1] * dsc;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          dx_max = fmaxf(fmaxf(dx_max, fmaxf(fabsf(dxo[mt][0]), fabsf(dxo[mt][1]))), fmaxf(fabsf(dxo[mt][2]), fabsf(dxo[mt][3])));
      }
      if (ok) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) *reinterpret_cast<f32x4 *>(a.dX + (ep + row) * D + 16 * mt + 4 * gq) = dxo[mt];
      }
      f32x4 dqN[2], xN[2];
      tailbwd::to_n(dqN, dq[0], dq[1], scr, tok, gq);
      tailbwd::to_n(xN, x[0], x[1], scr, tok, gq);
      {
        const H8 dqNS[2] = {split4(dqN[0]), split4(dqN[1])}, xNS[2] = {split4(xN[0]), split4(xN[1])};
        tailbwd::mm_dw4_16(gWq[0][0], gWq[0][1], gWq[1][0], gWq[1][1], dqNS[0], xNS[0], dqNS[0], xNS[1], dqNS[1], xNS[0], dqNS[1], xNS[1]);
        gBq[0] += tailbwd::sum4(dqN[0]); gBq[1] += tailbwd::sum4(dqN[1]);
      }
    }
    // ---- dK / dV of the key rows: every wave leaves its partial sums in its own LDS slot (plain 16-byte stores: LDS float
    // atomics retire a few lanes per cycle), the store to the compact buffer adds the four slots -------------------------
    {
      float *my = slots + wave * 2 * MK * PK;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
          if (kt < nkt) {
            *reinterpret_cast<f32x4 *>(my + (16 * kt + tok) * PK + 16 * mt + 4 * gq) = dKt[mt][kt] * (ln2 * ginv);
            *reinterpret_cast<f32x4 *>(my + MK * PK + (16 * kt + tok) * PK + 16 * mt + 4 * gq) = dVt[mt][kt] * ginv;
          }
    }
    __syncthreads();
    for (int i = tid; i < n_ak * 16; i += THREADS) {
      const int j = i >> 4, c4 = (i & 15) * 4;
      const float *src = slots + (c4 < D ? 0 : MK * PK - D) + j * PK + c4;
      f32x4 v = ld4(src);
#pragma unroll
      for (int w = 1; w < WAVES; ++w) v += ld4(src + w * 2 * MK * PK);
      *reinterpret_cast<f32x4 *>(a.dkvc + ((long)b * a.max_keys + j) * 2 * D + c4) = v;
    }
  }
  if (a.dx_absmax) {
    dx_max = wave_max(dx_max);
    if (lane == 0 && dx_max > 0.f) atomicMax(a.dx_absmax, __float_as_uint(dx_max));
  }
  // ---- the workgroup's Wq gradients: LDS staging, then one atomic per element ---------------------------------------
  __syncthreads();
  float *stg = Ks;      // Wq [32][32], then bq [32]
  for (int i = tid; i < D * D + D; i += THREADS) stg[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&stg[(16 * i + 4 * gq + r) * D + 16 * j + tok], gWq[i][j][r] * ginv);
    atomicAdd(&stg[D * D + 16 * i + tok], gBq[i] * ginv);
  }
  __syncthreads();
  for (int i = tid; i < D * D; i += THREADS) unsafeAtomicAdd(a.dwin + i, stg[i]);
  if (tid < D) unsafeAtomicAdd(a.dbin + tid, stg[D * D + tid]);
}

// Key rows of the attention block, backward: dx[key row] += Wk^T dK + Wv^T dV, dWk / dWv += dK^T / dV^T x[key rows], biases.
// One wave per (instance, key tile) unit, no barriers; the 8 accumulator tiles of a wave stay in registers.
__global__ __launch_bounds__(THREADS) void kv_bwd_kernel(BlockArgs a) {
  __shared__ __attribute__((aligned(16))) float Wkv[2 * D * PK];      // Wk, Wv row-major [out][in]
  __shared__ float stg[2 * D * D + 2 * D];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tok = lane & 15, gq = lane >> 4;
  for (int i = tid; i < 2 * D * D; i += THREADS) Wkv[(i >> 5) * PK + (i & 31)] = a.win[D * D + i];
  for (int i = tid; i < 2 * D * D + 2 * D; i += THREADS) stg[i] = 0.f;
  __syncthreads();
  f32x4 gWk[2][2], gWv[2][2];
  float gBk[2] = {0.f, 0.f}, gBv[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) gWk[i][j] = gWv[i][j] = zero4();
  const int KTI = (a.max_keys + 15) >> 4;                    // key tiles per instance (upper bound)
  const long units = (long)a.g.B * KTI;
  for (long u = (long)blockIdx.x * WAVES + wave; u < units; u += (long)gridDim.x * WAVES) {
    const int b = (int)(u / KTI), kt = (int)(u % KTI);
    const int n_ak = min(a.kcnt[2 * b + 1], a.max_keys);
    if (16 * kt >= n_ak) continue;
    const long base = (long)b * a.max_keys;
    // T layout (key on lane): dK / dV rows of the tile -> the key rows' dx
    const int keyT = 16 * kt + tok;
    const bool okT = keyT < n_ak;
    const float *src = a.dkvc + (base + (okT ? keyT : 0)) * 2 * D;
    f32x4 dk[2], dv[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      dk[mt] = okT ? ld4(src + 16 * mt + 4 * gq) : zero4();
      dv[mt] = okT ? ld4(src + D + 16 * mt + 4 * gq) : zero4();
    }
    const int rowT = okT ? a.keyidx[base + keyT] : 0;        // global token row
    f32x4 acc[2] = {zero4(), zero4()};
    tailbwd::mm_bwd<2, 2>(acc, Wkv, PK, dk, tok, gq);
    tailbwd::mm_bwd<2, 2>(acc, Wkv + D * PK, PK, dv, tok, gq);
    if (okT) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        float *dst = a.dX + (long)rowT * D + 16 * mt + 4 * gq;
        *reinterpret_cast<f32x4 *>(dst) = ld4(dst) + acc[mt];
      }
    }
    // N layout (channel on lane, key on (g, r)): weight gradients
    f32x4 kN[2], vN[2], xkN[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 16 * kt + 4 * gq + r;
      const bool ok = key < n_ak;
      const float *s2 = a.dkvc + (base + (ok ? key : 0)) * 2 * D;
      const float *xr = a.X + (long)(ok ? a.keyidx[base + key] : 0) * D;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        kN[i][r] = ok ? s2[16 * i + tok] : 0.f;
        vN[i][r] = ok ? s2[D + 16 * i + tok] : 0.f;
        xkN[i][r] = ok ? xr[16 * i + tok] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      tailbwd::mm_dw4(gWk[i][0], gWk[i][1], gWv[i][0], gWv[i][1], kN[i], xkN[0], kN[i], xkN[1], vN[i], xkN[0], vN[i], xkN[1]);
      gBk[i] += tailbwd::sum4(kN[i]);
      gBv[i] += tailbwd::sum4(vN[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        atomicAdd(&stg[(16 * i + 4 * gq + r) * D + 16 * j + tok], gWk[i][j][r]);
        atomicAdd(&stg[D * D + (16 * i + 4 * gq + r) * D + 16 * j + tok], gWv[i][j][r]);
      }
    atomicAdd(&stg[2 * D * D + 16 * i + tok], gBk[i]);
    atomicAdd(&stg[2 * D * D + D + 16 * i + tok], gBv[i]);
  }
  __syncthreads();
  for (int i = tid; i < 2 * D * D; i += THREADS) unsafeAtomicAdd(a.dwin + D * D + i, stg[i]);
  if (tid < 2 * D) unsafeAtomicAdd(a.dbin + D + tid, stg[2 * D * D + tid]);
}

}  // namespace abwd
