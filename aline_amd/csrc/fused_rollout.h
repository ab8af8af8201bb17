// Fused T-step rollout for the small-width model (d=32, F=128, H=4: config/encoder/encoder.yaml).
//
// One workgroup (4 waves) owns one episode for the WHOLE acquisition loop: embeddings, encoder,
// acquisition head, design selection, context update and the GMM posterior never leave the CU.
//   * token rows live in registers in the MFMA accumulator layout, TRANSPOSED: an activation tile
//     is X^T [features x 16 tokens]; lane (tok = lane & 15, g = lane >> 4) holds features
//     16*mt + 4*g + r (mt = acc tile, r = register).  Every linear layer is Y^T = W X^T, so the
//     accumulator of one product is, register for register, the B operand of the next one: the
//     k index of a 32-feature block is permuted as  pi(kb, g, j) = 32 kb + 16 (j>>2) + 4 g + (j&3)
//     and the weight (A operand) fragments are pre-permuted the same way at pack time.  No LDS
//     round trip, no cross-lane movement between layers; LayerNorm / softmax reduce over the 4
//     lane groups with two xor-shuffles.
//   * nobody attends to query rows (model/encoder.py:107,121): per layer the K/V of the <= 32 key
//     rows (context + selected targets) are computed once per episode by a pre-pass and kept in
//     LDS as head-block-structured MFMA fragments; every token tile then does
//     S^T = Kblk Q^T, softmax over keys (register axis), O^T = Vblk P^T.
//   * per-layer weights stream from a packed, L2-resident image into LDS (48 KB fp32 / layer).
//   * the static-slot state (role per slot, E = x-embedding (+ y-embedding once a point joins
//     the context)) lives in LDS across the T steps; design selection and the update that
//     replaces Task.update_batch (tasks/base_task.py:133-154) run in-kernel.
// Arithmetic: fp32 MFMA (v_mfma_f32_16x16x4_f32), the reference-precision mode.
#pragma once
#include "common.h"

namespace fused {

constexpr int D = 32, F = 128, H = 4, HD = 8;
constexpr int NKMAX = 32;       // key rows (context + selected targets) per episode
constexpr int MAXROWS = 256;    // token rows per episode (16 tiles)
constexpr int MAXNT = 8;        // target rows
constexpr int ES = 36;          // LDS row stride (floats) of E / Xk / Zt
constexpr int FRAG = 512;       // floats per operand fragment: 64 lanes x 8 k-elements
constexpr int NFRAG_LAYER = 24; // Wq 2, Wk 2, Wv 2, Wo 2, W1 8, W2 8
constexpr int FQ = 0, FK = 2, FV = 4, FO = 6, F1 = 8, F2 = 16;
// per-layer parameter block (floats) after the fragments
constexpr int PB_Q = 0, PB_K = 32, PB_V = 64, PB_O = 96, PB_1 = 128, PB_2 = 256, PLN1W = 288,
              PLN1B = 320, PLN2W = 352, PLN2B = 384, PARAMS = 416;
constexpr int LAYER_FLOATS = NFRAG_LAYER * FRAG + PARAMS;     // 12704
// acquisition head image: W1 8 fragments + b1[128] + w2[128] + b2 (padded to 4)
constexpr int HEAD_FLOATS = 8 * FRAG + 128 + 128 + 4;          // 4356

// ---- weight packing (once per rollout; weights are constant during a rollout) -------------------
struct PackArgs {
  int L;
  const float *in_proj_w[8], *in_proj_b[8], *out_proj_w[8], *out_proj_b[8], *lin1_w[8], *lin1_b[8],
      *lin2_w[8], *lin2_b[8], *n1w[8], *n1b[8], *n2w[8], *n2b[8];
  const float *acq_w1, *acq_b1, *acq_w2, *acq_b2;
  float *out;   // [L * LAYER_FLOATS + HEAD_FLOATS]
};

// element (lane, j) of fragment (mt, kb) of a [rows, K] row-major weight: W[16 mt + (lane & 15)][pi]
__device__ __forceinline__ float frag_elem(const float *W, int K, int mt, int kb, int e) {
  const int sub = e >> 8, lane = (e >> 2) & 63, jj = e & 3;   // image = [sub-plane][lane][4]
  const int j = sub * 4 + jj, g = lane >> 4;
  const int k = 32 * kb + 16 * (j >> 2) + 4 * g + (j & 3);
  return W[(16 * mt + (lane & 15)) * K + k];
}

__global__ void pack_weights_kernel(PackArgs a) {
  const int total = a.L * LAYER_FLOATS + HEAD_FLOATS;
  const float qscale = rsqrtf((float)HD);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (i < a.L * LAYER_FLOATS) {
      const int l = i / LAYER_FLOATS, o = i % LAYER_FLOATS;
      if (o < NFRAG_LAYER * FRAG) {
        const int f = o / FRAG, e = o % FRAG;
        if (f < FO) {            // in_proj rows: q 0..31, k 32..63, v 64..95   (K = 32)
          const int which = f >> 1, mt = f & 1;
          v = frag_elem(a.in_proj_w[l] + which * D * D, D, mt, 0, e);
          if (which == 0) v *= qscale;
        } else if (f < F1) {
          v = frag_elem(a.out_proj_w[l], D, f - FO, 0, e);
        } else if (f < F2) {
          v = frag_elem(a.lin1_w[l], D, f - F1, 0, e);          // [128, 32]: 8 m-tiles
        } else {
          const int q = f - F2;                                  // [32, 128]: (mt, kb) = (q / 4, q % 4)
          v = frag_elem(a.lin2_w[l], F, q >> 2, q & 3, e);
        }
      } else {
        const int p = o - NFRAG_LAYER * FRAG;
        if (p < PB_K) v = a.in_proj_b[l][p] * qscale;
        else if (p < PB_O) v = a.in_proj_b[l][p];                // k, v biases (offsets 32..95)
        else if (p < PB_1) v = a.out_proj_b[l][p - PB_O];
        else if (p < PB_2) v = a.lin1_b[l][p - PB_1];
        else if (p < PLN1W) v = a.lin2_b[l][p - PB_2];
        else if (p < PLN1B) v = a.n1w[l][p - PLN1W];
        else if (p < PLN2W) v = a.n1b[l][p - PLN1B];
        else if (p < PLN2B) v = a.n2w[l][p - PLN2W];
        else v = a.n2b[l][p - PLN2B];
      }
    } else {
      const int o = i - a.L * LAYER_FLOATS;
      if (o < 8 * FRAG) v = frag_elem(a.acq_w1, D, o / FRAG, 0, o % FRAG);
      else if (o < 8 * FRAG + 128) v = a.acq_b1[o - 8 * FRAG];
      else if (o < 8 * FRAG + 256) v = a.acq_w2[o - 8 * FRAG - 128];
      else if (o == 8 * FRAG + 256) v = a.acq_b2[0];
    }
    a.out[i] = v;
  }
}

// ---- device helpers -------------------------------------------------------------------------------
struct Frag { f32x4 lo, hi; };   // k-elements j = 0..3 / 4..7 of this lane

__device__ __forceinline__ Frag ld_frag(const float *base, int lane) {
  Frag f;
  f.lo = *reinterpret_cast<const f32x4 *>(base + lane * 4);
  f.hi = *reinterpret_cast<const f32x4 *>(base + 256 + lane * 4);
  return f;
}
__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

// acc += A(16 x 32) * B(32 x 16) as eight exact-fp32 16x16x4 MFMAs
__device__ __forceinline__ void mma_block(f32x4 &acc, const Frag &A, const Frag &B) {
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A.lo[j], B.lo[j], acc, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A.hi[j], B.hi[j], acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_half(f32x4 &acc, const f32x4 &A, const f32x4 &B) {
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j], B[j], acc, 0, 0, 0);
}

// reduce over the 4 lane groups (lanes l, l^16, l^32, l^48 hold the same token)
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float group_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

// LayerNorm over the 32 features of each token (eps 1e-5, biased variance): x <- LN(x) * w + b
__device__ __forceinline__ void layer_norm(f32x4 (&x)[2], const float *w, const float *b, int g) {
  float s = 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) s += x[mt][r];
  const float mean = group_sum(s) * (1.f / D);
  float ss = 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { float t = x[mt][r] - mean; ss = fmaf(t, t, ss); }
  const float rstd = rsqrtf(group_sum(ss) * (1.f / D) + 1e-5f);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const f32x4 wv = ld4(w + 16 * mt + 4 * g), bv = ld4(b + 16 * mt + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) x[mt][r] = (x[mt][r] - mean) * rstd * wv[r] + bv[r];
  }
}

struct RolloutArgs {
  int B, P, n_ctx0, n_th, T, L, C;
  const float *wpack;           // packed layer + head images
  const float *Ex, *Ey;         // [B, P, 32] cached point embeddings (rollout_init)
  const float *theta_tokens;    // [n_th, 32]
  const uint8_t *tmask;         // [n_th] or null
  const float *target_all;      // [B, n_th] or null
  const float *gmm_w1[16], *gmm_b1[16], *gmm_w2[16], *gmm_b2[16];
  float std_min;
  int mode;                     // ALINE_SELECT_*
  const float *uniform;         // [T, B]
  const int64_t *forced;        // [B, T]
  int *role;                    // [B, P] out (final roles, for export)
  int64_t *idx; int *slot; float *log_prob;           // [B, T]
  float *target_ll;                                   // [T, B, n_th]
  float *zt;                                          // [T, B, P - n_ctx0] or null
  float *post_mean, *post_std, *post_weight;          // [T, B, n_th, C] or null
};

// LDS carve (floats).  Everything lives in one dynamic array (16-byte aligned carve offsets).
constexpr int L_W = 0;                               // layer image / head image
constexpr int L_KB = L_W + LAYER_FLOATS;             // Kblk: 8 fragments x 256 floats (one sub-plane each)
constexpr int L_VB = L_KB + 8 * 256;                 // Vblk: 4 fragments x 512 floats
constexpr int L_E = L_VB + 4 * FRAG;                 // E [MAXROWS][ES]
constexpr int L_XK = L_E + MAXROWS * ES;             // Xk [NKMAX][ES]
constexpr int L_ZT = L_XK + NKMAX * ES;              // Zt [MAXNT][ES]
constexpr int L_LOGIT = L_ZT + MAXNT * ES;           // logits / probs [MAXROWS]
constexpr int L_RAW = L_LOGIT + MAXROWS;             // GMM raw outputs [MAXNT][16][4]
constexpr int L_INT = L_RAW + MAXNT * 16 * 4;        // ints: role[MAXROWS], kidx[MAXROWS], qslot[MAXROWS], misc[16]
constexpr int L_TOTAL = L_INT + 3 * MAXROWS + 16;
constexpr size_t LDS_BYTES = (size_t)L_TOTAL * 4;

__global__ __launch_bounds__(256, 1) void rollout_f32_kernel(RolloutArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *Wl = lds + L_W, *Kb = lds + L_KB, *Vb = lds + L_VB, *E = lds + L_E, *Xk = lds + L_XK,
        *Zt = lds + L_ZT, *logit = lds + L_LOGIT, *raw = lds + L_RAW;
  int *role = reinterpret_cast<int *>(lds + L_INT);
  int *kidx = role + MAXROWS, *qslot = kidx + MAXROWS, *misc = qslot + MAXROWS;
  // misc: 0 n_ck, 1 n_ak, 2.. wave counts

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tok = lane & 15, g = lane >> 4;
  const int P = a.P, n_th = a.n_th, N = P + n_th;
  const int ntiles = (N + 15) >> 4;
  const int zw = P - a.n_ctx0;

  // ---- episode state: roles, E = Ex (+ Ey on context rows), theta-token rows ---------------------
  for (int r = tid; r < MAXROWS; r += 256) role[r] = r < P ? (r < a.n_ctx0 ? r + 1 : 0) : -1;
  for (int i = tid; i < MAXROWS * 8; i += 256) {          // 8 float4 per row
    const int r = i >> 3, c4 = (i & 7) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < P) {
      v = ld4(a.Ex + ((long)b * P + r) * D + c4);
      if (r < a.n_ctx0) { f32x4 y = ld4(a.Ey + ((long)b * P + r) * D + c4); v += y; }
    } else if (r < N) {
      v = ld4(a.theta_tokens + (r - P) * D + c4);
    }
    *reinterpret_cast<f32x4 *>(E + r * ES + c4) = v;
  }
  for (int i = tid; i < NKMAX * ES; i += 256) Xk[i] = 0.f;
  __syncthreads();

  for (int t = 0; t < a.T; ++t) {
    // ---- key list: context slots in slot order, then the selected targets ----------------------
    {
      const int r = tid;                                   // MAXROWS == blockDim
      const bool ck = r < P && role[r] > 0;
      const unsigned long long bal = __ballot(ck);
      if (lane == 0) misc[2 + wave] = __popcll(bal);
      __syncthreads();
      int off = 0;
      for (int w = 0; w < wave; ++w) off += misc[2 + w];
      const int nck = misc[2] + misc[3] + misc[4] + misc[5];
      int k = ck ? off + __popcll(bal & ((1ull << lane) - 1ull)) : -1;
      if (r >= P && r < N) {
        int nsel = 0, mine = -1;
        for (int j = 0; j < n_th; ++j) {
          const bool sel = !a.tmask || a.tmask[j];
          if (j == r - P && sel) mine = nsel;
          nsel += sel ? 1 : 0;
        }
        if (mine >= 0) k = nck + mine;
        if (r == P) { misc[0] = nck; misc[1] = nck + nsel; }
      }
      kidx[r] = k;
      __syncthreads();
    }
    const int n_ck = misc[0], n_ak = misc[1];

    // ---- X^(0): token tiles of this wave from E (tile ti = wave + 4 i) ----------------------------
    f32x4 x[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 16 * (wave + 4 * i) + tok;
      x[i][0] = ld4(E + row * ES + 4 * g);
      x[i][1] = ld4(E + row * ES + 16 + 4 * g);
      const int k = kidx[row];
      if (k >= 0) {
        *reinterpret_cast<f32x4 *>(Xk + k * ES + 4 * g) = x[i][0];
        *reinterpret_cast<f32x4 *>(Xk + k * ES + 16 + 4 * g) = x[i][1];
      }
    }

    for (int l = 0; l < a.L; ++l) {
      // ---- stream layer l's packed image into LDS ------------------------------------------------
      {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.wpack + (long)l * LAYER_FLOATS);
        f32x4 *dst = reinterpret_cast<f32x4 *>(Wl);
        for (int i = tid; i < LAYER_FLOATS / 4; i += 256) dst[i] = src[i];
      }
      __syncthreads();   // weights + Xk visible
      const float *prm = Wl + NFRAG_LAYER * FRAG;

      // ---- pre-pass: K^T (waves 0,1) and V (waves 2,3) of key tile kt = wave & 1 ------------------
      {
        const int kt = wave & 1;
        if (kt == 0 || n_ak > 16) {
          const int key = 16 * kt + tok;
          Frag xf;
          xf.lo = ld4(Xk + key * ES + 4 * g);
          xf.hi = ld4(Xk + key * ES + 16 + 4 * g);
          if (wave < 2) {
            // K^T[c, key] = Wk x_key + bk : rows = channels
            f32x4 kacc[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
              kacc[mt] = ld4(prm + PB_K + 16 * mt + 4 * g);
              mma_block(kacc[mt], ld_frag(Wl + (FK + mt) * FRAG, lane), xf);
            }
            // head-block fragments: head h uses sub-plane h>>1, lanes with (g>>1) == (h&1)
#pragma unroll
            for (int h = 0; h < H; ++h) {
              f32x4 v = ((g >> 1) == (h & 1)) ? kacc[h >> 1] : (f32x4){0.f, 0.f, 0.f, 0.f};
              *reinterpret_cast<f32x4 *>(Kb + (h * 2 + kt) * 256 + lane * 4) = v;
            }
          } else {
            // V[key, c] = x_key Wv^T + bv : rows = keys, lane column = channel
            f32x4 vacc[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              const float bv = prm[PB_V + 16 * nt + tok];
              vacc[nt] = (f32x4){bv, bv, bv, bv};
              mma_block(vacc[nt], xf, ld_frag(Wl + (FV + nt) * FRAG, lane));
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                f32x4 v = ((tok >> 3) == e) ? vacc[nt] : (f32x4){0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4 *>(Vb + (nt * 2 + e) * FRAG + kt * 256 + lane * 4) = v;
              }
          }
        }
      }
      __syncthreads();   // K/V fragments visible

      // ---- main pass over this wave's token tiles --------------------------------------------------
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ti = wave + 4 * i;
        if (ti < ntiles) {
          const int row = 16 * ti + tok;
          const bool isq = row < P && role[row] == 0;
          const int nvalid = isq ? n_ak : n_ck;
          const Frag xf = {x[i][0], x[i][1]};
          // q = (Wq x + bq) / sqrt(hd)  (scale folded into the packed weights)
          f32x4 q[2], o[2];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            q[mt] = ld4(prm + PB_Q + 16 * mt + 4 * g);
            mma_block(q[mt], ld_frag(Wl + (FQ + mt) * FRAG, lane), xf);
            o[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int h = 0; h < H; ++h) {
            f32x4 s[2];
            s[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            s[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            mma_half(s[0], ld4(Kb + (h * 2 + 0) * 256 + lane * 4), q[h >> 1]);
            if (n_ak > 16) mma_half(s[1], ld4(Kb + (h * 2 + 1) * 256 + lane * 4), q[h >> 1]);
            // masked softmax over the key axis (registers x lane groups)
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int key = 16 * kt + 4 * g + r;
                s[kt][r] = key < nvalid ? s[kt][r] : -INFINITY;
                mx = fmaxf(mx, s[kt][r]);
              }
            mx = group_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
              for (int r = 0; r < 4; ++r) { s[kt][r] = __expf(s[kt][r] - mx); sum += s[kt][r]; }
            const float inv = 1.f / group_sum(sum);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
              for (int r = 0; r < 4; ++r) s[kt][r] *= inv;
            // O^T[c, tok] += Vblk[c, key] P^T[key, tok] for the 8 channels of head h
            const float *vf = Vb + ((h >> 1) * 2 + (h & 1)) * FRAG;
            mma_half(o[h >> 1], ld4(vf + lane * 4), s[0]);
            if (n_ak > 16) mma_half(o[h >> 1], ld4(vf + 256 + lane * 4), s[1]);
          }
          // x1 = LN1(x + Wo o + bo)
          f32x4 x1[2];
          {
            const Frag of = {o[0], o[1]};
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
              x1[mt] = ld4(prm + PB_O + 16 * mt + 4 * g) + x[i][mt];
              mma_block(x1[mt], ld_frag(Wl + (FO + mt) * FRAG, lane), of);
            }
          }
          layer_norm(x1, prm + PLN1W, prm + PLN1B, g);
          // x = LN2(x1 + W2 relu(W1 x1 + b1) + b2), hidden streamed in 32-wide chunks
          {
            const Frag x1f = {x1[0], x1[1]};
            f32x4 y[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) y[mt] = ld4(prm + PB_2 + 16 * mt + 4 * g) + x1[mt];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
              Frag hf;
              hf.lo = ld4(prm + PB_1 + 32 * kb + 4 * g);
              hf.hi = ld4(prm + PB_1 + 32 * kb + 16 + 4 * g);
              mma_block(hf.lo, ld_frag(Wl + (F1 + 2 * kb) * FRAG, lane), x1f);
              mma_block(hf.hi, ld_frag(Wl + (F1 + 2 * kb + 1) * FRAG, lane), x1f);
#pragma unroll
              for (int r = 0; r < 4; ++r) { hf.lo[r] = fmaxf(hf.lo[r], 0.f); hf.hi[r] = fmaxf(hf.hi[r], 0.f); }
#pragma unroll
              for (int mt = 0; mt < 2; ++mt) mma_block(y[mt], ld_frag(Wl + (F2 + mt * 4 + kb) * FRAG, lane), hf);
            }
            layer_norm(y, prm + PLN2W, prm + PLN2B, g);
            x[i][0] = y[0];
            x[i][1] = y[1];
          }
          // key rows publish x^(l+1) for the next layer's pre-pass
          if (l + 1 < a.L) {
            const int k = kidx[row];
            if (k >= 0) {
              *reinterpret_cast<f32x4 *>(Xk + k * ES + 4 * g) = x[i][0];
              *reinterpret_cast<f32x4 *>(Xk + k * ES + 16 + 4 * g) = x[i][1];
            }
          }
        }
      }
      __syncthreads();   // everyone done with this layer's weights / K / V
    }

    // ---- acquisition head (model/head.py:27-33) on every tile; z of the target rows -> Zt --------
    {
      const f32x4 *src = reinterpret_cast<const f32x4 *>(a.wpack + (long)a.L * LAYER_FLOATS);
      f32x4 *dst = reinterpret_cast<f32x4 *>(Wl);
      for (int i = tid; i < HEAD_FLOATS / 4; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    {
      const float *hb1 = Wl + 8 * FRAG, *hw2 = hb1 + 128, *hb2 = hw2 + 128;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ti = wave + 4 * i;
        if (ti < ntiles) {
          const int row = 16 * ti + tok;
          const Frag zf = {x[i][0], x[i][1]};
          float part = 0.f;
#pragma unroll
          for (int mt = 0; mt < 8; ++mt) {
            f32x4 hdn = ld4(hb1 + 16 * mt + 4 * g);
            mma_block(hdn, ld_frag(Wl + mt * FRAG, lane), zf);
            const f32x4 w2 = ld4(hw2 + 16 * mt + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) part = fmaf(fmaxf(hdn[r], 0.f), w2[r], part);
          }
          part = group_sum(part) + hb2[0];
          if (g == 0) logit[row] = part;
          if (row >= P && row < N) {
            *reinterpret_cast<f32x4 *>(Zt + (row - P) * ES + 4 * g) = x[i][0];
            *reinterpret_cast<f32x4 *>(Zt + (row - P) * ES + 16 + 4 * g) = x[i][1];
          }
        }
      }
    }
    __syncthreads();

    if (wave == 0) {
      // ---- softmax over the remaining queries + design selection (model/head.py:347-362) ---------
      int nq = 0;
      for (int c0 = 0; c0 < P; c0 += 64) {
        const int p = c0 + lane;
        const bool isq = p < P && role[p] == 0;
        const unsigned long long bal = __ballot(isq);
        if (isq) qslot[nq + __popcll(bal & ((1ull << lane) - 1ull))] = p;
        nq += __popcll(bal);
      }
      float mx = -INFINITY;
      for (int i = lane; i < nq; i += 64) mx = fmaxf(mx, logit[qslot[i]]);
      mx = wave_max(mx);
      float pv[4], sum = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int i = lane + 64 * c;
        pv[c] = i < nq ? __expf(logit[qslot[i]] - mx) : 0.f;
        sum += pv[c];
      }
      sum = wave_sum(sum);
      const float inv = 1.f / sum;
      float *prob = logit;   // compacted probabilities overwrite the logits (all reads done above)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        pv[c] *= inv;
        const int i = lane + 64 * c;
        if (i < nq) prob[i] = pv[c];
      }
      if (a.zt) {
        float *zo = a.zt + ((long)t * a.B + b) * zw;
        for (int i = lane; i < zw; i += 64) zo[i] = i < nq ? prob[i] : 0.f;
      }
      int choice = 0;
      float val = 0.f;
      if (a.mode == 0) {
        float best = -1.f; int bi = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int i = lane + 64 * c;
          if (i < nq && pv[c] > best) { best = pv[c]; bi = i; }
        }
        for (int o = 32; o > 0; o >>= 1) {
          const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
          if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        choice = bi; val = best;
      } else {
        float tot = wave_sum(pv[0] + pv[1] + pv[2] + pv[3]);
        if (a.mode == 2) {
          choice = (int)a.forced[(long)b * a.T + t];
          choice = min(max(choice, 0), nq - 1);
        } else {
          const float u = a.uniform[(long)t * a.B + b] * tot;
          float run = 0.f; int found = nq - 1; bool done = false;
          for (int c0 = 0; c0 < nq && !done; c0 += 64) {
            const int i = c0 + lane;
            float v = i < nq ? prob[i] : 0.f, incl = v;
            for (int o = 1; o < 64; o <<= 1) { const float tt = __shfl_up(incl, o, 64); if (lane >= o) incl += tt; }
            const bool hit = i < nq && (run + incl) > u;
            const unsigned long long bal = __ballot(hit);
            if (bal) { found = c0 + __ffsll((long long)bal) - 1; done = true; }
            run += __shfl(incl, 63, 64);
          }
          choice = found;
        }
        // Categorical(probs).log_prob: probs / probs.sum(), clamped to [eps, 1 - eps]
        val = fminf(fmaxf(prob[choice] / tot, 1.1920929e-07f), 1.f - 1.1920929e-07f);
      }
      const int sl = qslot[choice];
      const int order = (P - nq) + 1;
      if (lane == 0) {
        const long o = (long)b * a.T + t;
        if (a.idx) a.idx[o] = choice;
        if (a.slot) a.slot[o] = sl;
        if (a.log_prob) a.log_prob[o] = logf(val);
        role[sl] = order;
      }
      // the chosen point joins the context: E[slot] += y-embedding (embedder.py:156)
      if (lane < D) E[sl * ES + lane] += a.Ey[((long)b * P + sl) * D + lane];
    } else {
      // ---- GMM heads on the target rows (model/head.py:172-177), fp32 FMA: n_t rows are few -----
      for (int c = wave - 1; c < a.C; c += 3) {
        const float *w1 = a.gmm_w1[c], *b1 = a.gmm_b1[c], *w2 = a.gmm_w2[c];
        f32x4 wr[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int q4 = 0; q4 < 8; ++q4) wr[u][q4] = ld4(w1 + (lane + 64 * u) * D + 4 * q4);
        const float bb0 = b1[lane], bb1 = b1[lane + 64];
        float w2v[3][2];
#pragma unroll
        for (int j = 0; j < 3; ++j) { w2v[j][0] = w2[j * F + lane]; w2v[j][1] = w2[j * F + lane + 64]; }
        for (int r = 0; r < n_th; ++r) {
          float h0 = bb0, h1 = bb1;
#pragma unroll
          for (int q4 = 0; q4 < 8; ++q4) {
            const f32x4 zv = ld4(Zt + r * ES + 4 * q4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { h0 = fmaf(wr[0][q4][e], zv[e], h0); h1 = fmaf(wr[1][q4][e], zv[e], h1); }
          }
          h0 = fmaxf(h0, 0.f); h1 = fmaxf(h1, 0.f);
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const float sres = wave_sum(h0 * w2v[j][0] + h1 * w2v[j][1]);
            if (lane == 0) raw[(r * 16 + c) * 4 + j] = sres + a.gmm_b2[c][j];
          }
        }
      }
    }
    __syncthreads();

    // ---- GMM parameter maps + compute_ll (head.py:176-177, utils/eval.py:200-207) -------------------
    if (wave == 1) {
      for (int r = 0; r < n_th; ++r) {
        const bool act = lane < a.C;
        const float r0 = act ? raw[(r * 16 + lane) * 4 + 0] : 0.f;
        const float r1 = act ? raw[(r * 16 + lane) * 4 + 1] : 0.f;
        const float r2 = act ? raw[(r * 16 + lane) * 4 + 2] : -INFINITY;
        const float sd = softplus_f(r1) + a.std_min;
        const float m2 = wave_max(r2);
        const float e = act ? __expf(r2 - m2) : 0.f;
        const float wgt = e / wave_sum(e);
        const long orow = ((long)t * a.B + b) * n_th + r;
        if (act) {
          if (a.post_mean) a.post_mean[orow * a.C + lane] = r0;
          if (a.post_std) a.post_std[orow * a.C + lane] = sd;
          if (a.post_weight) a.post_weight[orow * a.C + lane] = wgt;
        }
        if (a.target_ll && a.target_all) {
          const float v = a.target_all[(long)b * n_th + r];
          const float z = (v - r0) / sd;
          const float lp = act ? (-0.5f * z * z - logf(sd) - 0.91893853320467274178f + logf(wgt)) : -INFINITY;
          const float m3 = wave_max(lp);
          const float se = wave_sum(act ? __expf(lp - m3) : 0.f);
          if (lane == 0) a.target_ll[orow] = m3 + logf(se);
        }
      }
    }
    // (the barrier at the top of the next step's key-list build orders role / E updates)
    __syncthreads();
  }
  for (int r = tid; r < P; r += 256) a.role[(long)b * P + r] = role[r];
}

}  // namespace fused
