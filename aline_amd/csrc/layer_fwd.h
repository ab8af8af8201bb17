// One encoder layer of the small-width model (d = 32, F = 128, 4 heads of 8, <= 48 keys per instance), forward recompute of
// the TRAINING backward, as ONE kernel:  x -> q = Wq x + bq -> a = masked set-attention(q, K, V) -> y = tail(x, a)
// (model/encoder.py:8-46, 83-126, 128-141).  It replaces the Q-projection GEMM, the K / V gather GEMM, the attention
// kernel and the tail kernel of the per-op recompute (2.25 ms per layer at the headline shape): x is read once, a (the
// tail backward needs it) and y are written once, q / K / V never leave the CU.
// Persistent workgroups walk the (step, episode) instances; per instance K / V of the key rows are copied from the compact
// buffer a small GEMM filled (one coalesced copy and one barrier: building the key list and projecting K / V inside this
// kernel cost more than the tiles -- four dependent global round trips and five barriers per instance); a wave owns a 16-row token tile from x to y in the T layout of tail_bwd.h:
//   S^T [keys x rows] = Kblk q^T (key mask = initial accumulator), softmax over registers + lane groups,
//   a^T [chan x rows] = Vblk^T P^T with P^T, register for register, as the B operand; then the tail in the same registers.
#pragma once
#include "attn_bwd_mfma.h"

namespace lfwd {

constexpr int D = 32, F = 128, HD = 8, H = 4, PK = abwd::PK, WAVES = 4, THREADS = 64 * WAVES;
using fused::ld4;
using fused::group_sum;
using fused::group_max;
using fused::zero4;
using namespace tailbwd;      // image offsets L_*, P_*, PW, PW2, mm_fwd, normalise

struct Args {
  Geo g;
  const float *X;            // [M, 32] layer input
  float *A, *Y;              // [M, 32] attention output (kept for the tail backward), layer output
  const float *win, *bin;    // in_proj_weight [96, 32], in_proj_bias [96]
  const float *kvc;          // [I * max_keys, 64] K | V of the key rows (key_list_kernel + row-gather GEMM), key-list order
  const int *kcnt;           // [I, 2] context keys, all keys
  int max_keys;
  const float *wo, *bo, *w1, *b1, *w2, *b2, *g1, *e1, *g2, *e2;
};

// LDS (floats): tail image [L_SCR] | Wq image [32][36] | bq [32] | Ks, Vs, zeros [3][16 KT][36]
constexpr int lds_floats(int KT) { return L_SCR + D * PK + D + 3 * 16 * KT * PK; }

template <int KT>
__global__ __launch_bounds__(THREADS, 2) void layer_fwd_kernel(Args a) {
  constexpr int MK = 16 * KT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *const Wi = lds + L_SCR, *const bi = Wi + D * PK, *const Ks = bi + D, *const Vs = Ks + MK * PK, *const Zs = Vs + MK * PK;
  const Geo &g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tok = lane & 15, gq = lane >> 4;
  for (int i = tid; i < D * D; i += THREADS) lds[L_WO + (i >> 5) * PW + (i & 31)] = a.wo[i];
  for (int i = tid; i < F * D; i += THREADS) lds[L_W1 + (i >> 5) * PW + (i & 31)] = a.w1[i];
  for (int i = tid; i < D * F; i += THREADS) lds[L_W2 + (i >> 7) * PW2 + (i & 127)] = a.w2[i];
  if (tid < F) lds[L_PRM + P_B1 + tid] = a.b1[tid];
  if (tid < D) {
    lds[L_PRM + P_BO + tid] = a.bo[tid]; lds[L_PRM + P_B2 + tid] = a.b2[tid];
    lds[L_PRM + P_G1 + tid] = a.g1[tid]; lds[L_PRM + P_E1 + tid] = a.e1[tid];
    lds[L_PRM + P_G2 + tid] = a.g2[tid]; lds[L_PRM + P_E2 + tid] = a.e2[tid];
  }
  for (int i = tid; i < D * D; i += THREADS) Wi[(i >> 5) * PK + (i & 31)] = a.win[i];
  if (tid < D) bi[tid] = a.bin[tid];
  for (int i = tid; i < MK * PK; i += THREADS) Zs[i] = 0.f;
  // scores in base-2 units: q carries 1 / sqrt(hd) and log2(e)
  const float scale2 = rsqrtf((float)HD) * 1.44269504088896340736f;
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
    f32x4 nx[2];
    int nrole = 0;
    {
      const int row = min(wave * 16 + tok, g.N - 1);
      nx[0] = ld4(a.X + (ep + row) * D + 4 * gq);
      nx[1] = ld4(a.X + (ep + row) * D + 16 + 4 * gq);
      if (row < g.P) nrole = abwd::load_role(g, b, row);
    }
    const int n_ck = min(a.kcnt[2 * b], MK), n_ak = min(a.kcnt[2 * b + 1], MK);
    const int nkt = (n_ak + 15) >> 4;
    __syncthreads();      // the previous instance is done with K / V (and the images are in place)
#pragma unroll
    for (int u = 0; u < KT; ++u) {      // (loaded while the previous instance was computed)
      const int i = tid + u * THREADS, j = i >> 4, c4 = (i & 15) * 4;
      *reinterpret_cast<f32x4 *>((c4 < D ? Ks : Vs - D) + j * PK + c4) = nkv[u];
    }
    __syncthreads();
    if (b + (int)gridDim.x < g.B) load_kv(b + gridDim.x);

    for (int tile = wave; tile < ntile; tile += WAVES) {
      int zoff = 0;
      asm volatile("" : "+v"(zoff));
      const float *W = lds + zoff, *prm = W + L_PRM;
      const int row = tile * 16 + tok;
      const bool ok = row < g.N;
      const bool isq = ok && row < g.P && !abwd::role_is_ctx(g, b, nrole);
      const int nk = isq ? n_ak : n_ck;
      f32x4 kmask[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) kmask[kt][r] = 16 * kt + 4 * gq + r < nk ? 0.f : -INFINITY;
      const f32x4 x[2] = {nx[0], nx[1]};
      if (tile + WAVES < ntile) {
        const int nr = min(row + 16 * WAVES, g.N - 1);
        nx[0] = ld4(a.X + (ep + nr) * D + 4 * gq);
        nx[1] = ld4(a.X + (ep + nr) * D + 16 + 4 * gq);
        nrole = nr < g.P ? abwd::load_role(g, b, nr) : 0;
      }
      f32x4 q[2] = {ld4(bi + 4 * gq), ld4(bi + 16 + 4 * gq)};
      mm_fwd<2, 2>(q, Wi, PK, x, tok, gq);
      q[0] *= scale2; q[1] *= scale2;
      f32x4 at[2] = {zero4(), zero4()};
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const int mt = h >> 1, hg = h & 1;
        const bool mine_g = (gq >> 1) == hg, mine_c = (tok >> 3) == hg;
        const float *Kg = mine_g ? Ks : Zs, *Vc = mine_c ? Vs : Zs;
        f32x4 s[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          s[kt] = kmask[kt];
          if (kt < nkt) {
            const f32x4 kf = ld4(Kg + (16 * kt + tok) * PK + 16 * mt + 4 * gq);
            s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[0], q[mt][0], kmask[kt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0x7F6);
#pragma unroll
            for (int r = 1; r < 4; ++r) MFMAO(s[kt], kf[r], q[mt][r]);
          }
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
        mx = group_max(mx);
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) { s[kt][r] = __builtin_amdgcn_exp2f(s[kt][r] - mx); l += s[kt][r]; }
        l = group_sum(l);
        const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          if (kt < nkt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) MFMAO(at[mt], Vc[(16 * kt + 4 * gq + r) * PK + 16 * mt + tok], s[kt][r] * inv);
          }
        }
      }
      if (ok) {
        *reinterpret_cast<f32x4 *>(a.A + (ep + row) * D + 4 * gq) = at[0];
        *reinterpret_cast<f32x4 *>(a.A + (ep + row) * D + 16 + 4 * gq) = at[1];
      }
      // ---- token-local tail (tail_bwd.h, forward) -------------------------------------------------------------------
      f32x4 n1[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) n1[mt] = ld4(prm + P_BO + 16 * mt + 4 * gq) + x[mt];
      mm_fwd<2, 2>(n1, W + L_WO, PW, at, tok, gq);
      normalise(n1);
      f32x4 x1[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) x1[mt] = n1[mt] * ld4(prm + P_G1 + 16 * mt + 4 * gq) + ld4(prm + P_E1 + 16 * mt + 4 * gq);
      f32x4 hid[8];
#pragma unroll
      for (int ob = 0; ob < 8; ++ob) hid[ob] = ld4(prm + P_B1 + 16 * ob + 4 * gq);
      mm_fwd<8, 2>(hid, W + L_W1, PW, x1, tok, gq);
#pragma unroll
      for (int ob = 0; ob < 8; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) hid[ob][r] = relu_nn(hid[ob][r]);
      f32x4 n2[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) n2[mt] = ld4(prm + P_B2 + 16 * mt + 4 * gq) + x1[mt];
      mm_fwd<2, 8>(n2, W + L_W2, PW2, hid, tok, gq);
      normalise(n2);
      if (ok) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          *reinterpret_cast<f32x4 *>(a.Y + (ep + row) * D + 16 * mt + 4 * gq) =
              n2[mt] * ld4(prm + P_G2 + 16 * mt + 4 * gq) + ld4(prm + P_E2 + 16 * mt + 4 * gq);
      }
    }
  }
}


// The same layer without the per-instance structure: a wave takes (instance, token tile) units from one flat list and reads
// the K / V fragments of its instance straight from the compact buffer (8 KB per instance, L2-resident: 13 tiles read it) --
// no LDS copy, no barrier after the start, every wave has the same number of tiles.
constexpr int LDS_FLOATS_FLAT = L_SCR + D * PK + D;

template <int KT>
__global__ __launch_bounds__(THREADS, 2) void layer_fwd_flat_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *const Wi = lds + L_SCR, *const bi = Wi + D * PK;
  const Geo &g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tok = lane & 15, gq = lane >> 4;
  for (int i = tid; i < D * D; i += THREADS) lds[L_WO + (i >> 5) * PW + (i & 31)] = a.wo[i];
  for (int i = tid; i < F * D; i += THREADS) lds[L_W1 + (i >> 5) * PW + (i & 31)] = a.w1[i];
  for (int i = tid; i < D * F; i += THREADS) lds[L_W2 + (i >> 7) * PW2 + (i & 127)] = a.w2[i];
  if (tid < F) lds[L_PRM + P_B1 + tid] = a.b1[tid];
  if (tid < D) {
    lds[L_PRM + P_BO + tid] = a.bo[tid]; lds[L_PRM + P_B2 + tid] = a.b2[tid];
    lds[L_PRM + P_G1 + tid] = a.g1[tid]; lds[L_PRM + P_E1 + tid] = a.e1[tid];
    lds[L_PRM + P_G2 + tid] = a.g2[tid]; lds[L_PRM + P_E2 + tid] = a.e2[tid];
  }
  for (int i = tid; i < D * D; i += THREADS) Wi[(i >> 5) * PK + (i & 31)] = a.win[i];
  if (tid < D) bi[tid] = a.bin[tid];
  __syncthreads();
  const float scale2 = rsqrtf((float)HD) * 1.44269504088896340736f;
  const int ntile = (g.N + 15) >> 4;
  const long units = (long)g.B * ntile, ustep = (long)gridDim.x * WAVES;
  for (long u = (long)blockIdx.x * WAVES + wave; u < units; u += ustep) {
    int zoff = 0;
    asm volatile("" : "+v"(zoff));
    const float *W = lds + zoff, *prm = W + L_PRM;
    const int b = (int)(u / ntile), tile = (int)(u % ntile);
    const long ep = (long)b * g.N;
    const int row = tile * 16 + tok;
    const bool ok = row < g.N;
    const int rc = ok ? row : g.N - 1;
    const f32x4 x[2] = {ld4(a.X + (ep + rc) * D + 4 * gq), ld4(a.X + (ep + rc) * D + 16 + 4 * gq)};
    const int role = rc < g.P ? abwd::load_role(g, b, rc) : 0;
    const int n_ck = min(a.kcnt[2 * b], 16 * KT), n_ak = min(a.kcnt[2 * b + 1], 16 * KT);
    const int nkt = (n_ak + 15) >> 4;
    // K rows (A operand of the scores: key on lane) and V columns (A operand of the output: channel on lane) of the instance
    f32x4 kf[2][KT], vc[2][KT];
    const float *kv = a.kvc + (long)b * a.max_keys * 2 * D;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        kf[mt][kt] = ld4(kv + (long)min(16 * kt + tok, a.max_keys - 1) * 2 * D + 16 * mt + 4 * gq);
#pragma unroll
        for (int r = 0; r < 4; ++r) vc[mt][kt][r] = kv[(long)min(16 * kt + 4 * gq + r, a.max_keys - 1) * 2 * D + D + 16 * mt + tok];
      }
    const bool isq = ok && row < g.P && !abwd::role_is_ctx(g, b, role);
    const int nk = isq ? n_ak : n_ck;
    f32x4 kmask[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) kmask[kt][r] = 16 * kt + 4 * gq + r < nk ? 0.f : -INFINITY;
    f32x4 q[2] = {ld4(bi + 4 * gq), ld4(bi + 16 + 4 * gq)};
    mm_fwd<2, 2>(q, Wi, PK, x, tok, gq);
    q[0] *= scale2; q[1] *= scale2;
    f32x4 at[2] = {zero4(), zero4()};
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const int mt = h >> 1, hg = h & 1;
      const bool mine_g = (gq >> 1) == hg, mine_c = (tok >> 3) == hg;
      f32x4 s[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        s[kt] = kmask[kt];
        if (kt < nkt) {
          const f32x4 kh = mine_g ? kf[mt][kt] : zero4();
          s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kh[0], q[mt][0], kmask[kt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0x7F6);
#pragma unroll
          for (int r = 1; r < 4; ++r) MFMAO(s[kt], kh[r], q[mt][r]);
        }
      }
      float mx = -3.0e38f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
      mx = group_max(mx);
      float l = 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[kt][r] = __builtin_amdgcn_exp2f(s[kt][r] - mx); l += s[kt][r]; }
      l = group_sum(l);
      const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        if (kt < nkt) {
          const f32x4 vh = mine_c ? vc[mt][kt] : zero4();
#pragma unroll
          for (int r = 0; r < 4; ++r) MFMAO(at[mt], vh[r], s[kt][r] * inv);
        }
      }
    }
    if (ok) {
      *reinterpret_cast<f32x4 *>(a.A + (ep + row) * D + 4 * gq) = at[0];
      *reinterpret_cast<f32x4 *>(a.A + (ep + row) * D + 16 + 4 * gq) = at[1];
    }
    // ---- token-local tail (tail_bwd.h, forward) -------------------------------------------------------------------
    f32x4 n1[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) n1[mt] = ld4(prm + P_BO + 16 * mt + 4 * gq) + x[mt];
    mm_fwd<2, 2>(n1, W + L_WO, PW, at, tok, gq);
    normalise(n1);
    f32x4 x1[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) x1[mt] = n1[mt] * ld4(prm + P_G1 + 16 * mt + 4 * gq) + ld4(prm + P_E1 + 16 * mt + 4 * gq);
    f32x4 hid[8];
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) hid[ob] = ld4(prm + P_B1 + 16 * ob + 4 * gq);
    mm_fwd<8, 2>(hid, W + L_W1, PW, x1, tok, gq);
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
#pragma unroll
      for (int r = 0; r < 4; ++r) hid[ob][r] = relu_nn(hid[ob][r]);
    f32x4 n2[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) n2[mt] = ld4(prm + P_B2 + 16 * mt + 4 * gq) + x1[mt];
    mm_fwd<2, 8>(n2, W + L_W2, PW2, hid, tok, gq);
    normalise(n2);
    if (ok) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        *reinterpret_cast<f32x4 *>(a.Y + (ep + row) * D + 16 * mt + 4 * gq) =
            n2[mt] * ld4(prm + P_G2 + 16 * mt + 4 * gq) + ld4(prm + P_E2 + 16 * mt + 4 * gq);
    }
  }
}

}  // namespace lfwd
