// libaline_hip.so -- C ABI (include/aline_hip.h) over the gfx950 kernels.
// Host side only enqueues kernels on the caller's stream; no allocation, no synchronisation.
#include "../../include/aline_hip.h"
#include "common.h"
#include "gemm.h"
#include "kernels.h"
#include "eig.h"
#include "fused_rollout.h"
#include "fused_side.h"
#include "fused_tail.h"
#include "tile_image.h"
#include "x3.h"
#include "s3.h"
#include "attn3.h"
#include "backward.h"
#include "attn_bwd_wide.h"
#include "attn_bwd8.h"
#include "tail_bwd.h"
#include "attn_bwd_mfma.h"
#include "acq_head_bwd.h"
#include "layer_fwd.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <cstdlib>

namespace {

// Diagnostic word / knobs of include/aline_hip.h (aline_debug_set_flags / aline_debug_set_param): 0 in normal operation.
// The library never reads the environment; the Python mirror translates its ALINE_* debug variables into this word.
std::atomic<uint32_t> g_debug_flags{0};
std::atomic<int> g_debug_params[ALINE_DBG_NPARAMS] = {};
inline bool dbg(uint32_t bit) { return (g_debug_flags.load(std::memory_order_relaxed) & bit) != 0; }
inline int dbg_param(int key) { return g_debug_params[key].load(std::memory_order_relaxed); }

inline size_t align_up(size_t v, size_t a = 64) { return (v + a - 1) / a * a; }

#define CHECK_LAUNCH()                                   \
  do {                                                   \
    if (hipGetLastError() != hipSuccess) return ALINE_ELAUNCH; \
  } while (0)
#define TRY(x)                \
  do {                        \
    int _rc = (x);            \
    if (_rc != 0) return _rc; \
  } while (0)

constexpr int kRawStride = 48;            // floats per row of the raw GMM head outputs (3 per component, C <= 16)
constexpr size_t kGmmChunkRows = 32768;   // rows per post-loop GMM chunk (hidden = rows x C*F floats)

// Workspace plan (offsets in floats).  One plan serves the step API and the rollout API.
struct Plan {
  size_t Flag, xKX, xKpos, Ex, Ey, Hid, X, X1, QKV, A, Tm, Wacq, scalar, Wpack, Stamps, Ztg, xImg, xIn, xA, xB, xKV, xKeys, xKcnt, xLog, xZimg, xRaw, xRawQ, sImg, sX0, sXW, sLog, sZimg, sZq, sRaw, KeyIdx, Kcnt, X0, total;
  int qgmm_chunk;  // episodes per query-GMM chunk
};

Plan make_plan(const aline_model &m, int B, int P, int n_td, int ey_rows, bool query_gmm, int T = 0) {
  Plan p{};
  const size_t N = (size_t)P + n_td + m.n_theta, M = (size_t)B * N, d = m.d, F = m.F;
  const size_t n_t = (size_t)n_td + m.n_theta;
  const size_t Cc = (size_t)std::max(m.C, 1);
  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off += align_up(n); return o; };
  p.Flag = take(16);     // f16 range status word (aline_f16_range_offset() == 0 for every plan)
  p.Ex = take((size_t)B * (P + n_td) * d);
  p.Ey = take((size_t)B * ey_rows * d);
  size_t hid = std::max({(size_t)B * (P + n_td) * F, (size_t)B * ey_rows * F, M * F,
                         (size_t)B * n_t * Cc * F});
  p.qgmm_chunk = 0;
  if (query_gmm) {
    size_t per_ep = (size_t)P * Cc * F;
    size_t budget = (size_t)64 << 20;  // 64 Mi floats = 256 MiB of hidden activations per chunk
    p.qgmm_chunk = (int)std::max<size_t>(1, std::min<size_t>(B, budget / per_ep));
    hid = std::max(hid, per_ep * p.qgmm_chunk);
  }
  if (T > 0) hid = std::max(hid, std::min((size_t)T * B * n_t, kGmmChunkRows) * Cc * F);
  p.Hid = take(hid);
  p.X = take(M * d);
  p.X1 = take(M * d);
  p.QKV = take(M * 3 * d);
  p.A = take(M * d);
  p.Tm = take(M * d);
  p.KeyIdx = take(M);              // key rows of every episode (generic pipeline: K / V projections on these rows only)
  p.Kcnt = take((size_t)B * 2);
  p.X0 = take(T > 0 ? M * d : 0);   // rollouts: the embedded input, kept across the steps (one row per episode changes)
  p.Wacq = take(F * d);
  p.scalar = take(64);
  p.Wpack = take((size_t)ALINE_MAX_LAYERS * fused::LAYER_FLOATS + fused::HEAD_FLOATS +
                 (size_t)(ALINE_MAX_COMPONENTS + 2) * fused::SIDE_FRAGS);
  p.Stamps = take(512);   // diagnostic stamps of the fused kernel (8 x 16 x u64)
  p.Ztg = take((size_t)T * B * n_t * d);   // fused rollout: target-row encodings of all steps
  if (T > 0 && (m.d == x3::D || m.d == x5::D) && m.precision == ALINE_PREC_F16X3) {     // x3 / x5 path (x3.h): split-f16 tile images
    const bool w5 = m.d == x5::D;
    auto pieces = [w5](long tiles) { return (size_t)(w5 ? x5::img_pieces(tiles) : x3::img_pieces(tiles)) * 4; };
    const size_t tpe = (N + 15) / 16, img = pieces((long)B * tpe);
    p.xImg = take((size_t)(w5 ? x5::image_words(m.L, m.F, m.C) : x3::image_words(m.L, m.F, m.C)));
    p.xIn = take(img); p.xA = take(img); p.xB = take(img);
    p.xKV = take((size_t)B * (w5 ? x5::KV_EP : x3::KV_EP) * 4);
    p.xKeys = take((size_t)B * x3::WNK); p.xKcnt = take((size_t)B * 2);
    p.xKX = take(pieces((long)B * (x3::WNK / 16)));                           // key image (key rows of the layer input, key-list order)
    p.xKpos = take((size_t)B * tpe * 16 / 2 + 1);                             // short [B][16 tpe]
    p.xLog = take((size_t)B * tpe * 16);
    p.xZimg = take(pieces(((long)T * B * n_t + 15) / 16));
    p.xRaw = take((size_t)T * B * n_t * kRawStride);
    p.xRawQ = take(query_gmm ? (size_t)B * tpe * 16 * kRawStride : 0);       // raw GMM outputs of every token row of ONE step (posterior_out_query)
  }
  if (T > 0 && m.d == s3::D && m.precision == ALINE_PREC_F16X3) {           // s3 path (s3.h): d = 32 split-f16 tile images
    const size_t tpe = (N + 15) / 16, img = (size_t)s3::img_pieces((long)B * tpe) * 4;
    p.sImg = take((size_t)s3::image_words(m.L, m.F, m.C));
    p.sX0 = take(img); p.sXW = take(img);
    p.sLog = take((size_t)B * tpe * 16);
    p.sZimg = take((size_t)s3::img_pieces(((long)T * B * n_t + 15) / 16) * 4);
    p.sZq = take(query_gmm ? (size_t)s3::img_pieces(((long)T * B * P + 15) / 16) * 4 : 0);     // candidate rows of all steps (posterior_out_query)
    p.sRaw = take(1024);     // (diagnostic stamps of the S3_STAMPS build)
  }
  p.total = off;
  return p;
}

enum { ST_EMBED = 1, ST_ENC = 2, ST_HEAD = 4, ST_ALL = 7 };

// Only the fields (and weight pointers) of the requested stages are checked, so a stand-alone
// Embedder / Encoder / OutputHead module can call its own entry point with a partial struct.
int validate_model(const aline_model &m, int stages = ST_ALL) {
  if (m.d <= 0 || m.F <= 0) return ALINE_EINVAL;
  if (m.d % 32 || m.F % 32 || m.d > 512) return ALINE_EUNSUPPORTED;
  if (m.precision < 0 || m.precision > ALINE_PREC_F16X3) return ALINE_EINVAL;
  if (m.n_theta < 0) return ALINE_EINVAL;
  if (m.embedding_type == ALINE_EMB_DATA ? m.n_theta != 0 : m.n_theta <= 0) return ALINE_EINVAL;
  if (stages & ST_EMBED) {
    if (m.dim_y < 1 || m.dim_y > 8 || m.dim_x < 1 || m.dim_x > 8) return ALINE_EUNSUPPORTED;
    if (!m.x_w1 || !m.x_b1 || !m.x_w2 || !m.x_b2 || !m.y_w1 || !m.y_b1 || !m.y_w2 || !m.y_b2)
      return ALINE_EINVAL;
    if (m.n_theta > 0 && !m.theta_tokens) return ALINE_EINVAL;
  }
  if (stages & ST_ENC) {
    if (m.H <= 0 || m.L <= 0) return ALINE_EINVAL;
    if (m.L > ALINE_MAX_LAYERS || m.d % m.H) return ALINE_EUNSUPPORTED;
    const int hd = m.d / m.H;
    if (hd != 4 && hd != 8 && hd != 16 && hd != 32 && hd != 64 && hd != 128) return ALINE_EUNSUPPORTED;
    for (int l = 0; l < m.L; ++l)
      if (!m.in_proj_w[l] || !m.in_proj_b[l] || !m.out_proj_w[l] || !m.out_proj_b[l] || !m.lin1_w[l] ||
          !m.lin1_b[l] || !m.lin2_w[l] || !m.lin2_b[l] || !m.norm1_w[l] || !m.norm1_b[l] ||
          !m.norm2_w[l] || !m.norm2_b[l])
        return ALINE_EINVAL;
  }
  if (stages & ST_HEAD) {
    if (m.C <= 0) return ALINE_EINVAL;
    if (m.C > ALINE_MAX_COMPONENTS) return ALINE_EUNSUPPORTED;
    if (m.dim_y != 1) return ALINE_EUNSUPPORTED;  // GMM head is single-output (model/head.py:113 TODO)
    if (!m.acq_w1 || !m.acq_b1 || !m.acq_w2 || !m.acq_b2) return ALINE_EINVAL;
    for (int c = 0; c < m.C; ++c)
      if (!m.gmm_w1[c] || !m.gmm_b1[c] || !m.gmm_w2[c] || !m.gmm_b2[c]) return ALINE_EINVAL;
  }
  return ALINE_OK;
}

GemmArgs gemm_args(const float *X, int ldx, const float *W, const float *bias, int ldw, float *Y,
                   int ldy, int M, int N, int K, bool relu) {
  GemmArgs a{};
  a.X = X; a.ldx = ldx; a.R_in = 1; a.G_in = 1; a.off_in = 0;
  a.W[0] = W; a.bias[0] = bias; a.ldw = ldw;
  a.Y = Y; a.ldy = ldy; a.R_out = 1; a.G_out = 1; a.off_out = 0; a.col_per_group = 0;
  a.M = M; a.N = N; a.K = K; a.relu = relu ? 1 : 0;
  return a;
}

// the f16 range status word is cleared by a KERNEL node: a hipMemsetAsync captured into a HIP graph was observed (ROCm 7.2, MI355X)
// to fill the 64 bytes with the kernel arguments of an eager launch enqueued right behind the replay (tools/ws_debug.py: seed and
// offset of torch's uniform_ kernel appeared in the words after 200 back-to-back refresh + replay pairs)
__global__ void clear_words_kernel(unsigned *p) { p[threadIdx.x] = 0u; }

struct Ctx {
  const aline_model *m;
  Geo g;
  Plan pl;
  float *ws;
  hipStream_t st;
  float *at(size_t off) const { return ws + off; }
  unsigned *flag() const { return reinterpret_cast<unsigned *>(ws + pl.Flag); }     // f16 range guard (common.h)
  int clear_flag() const {
    hipLaunchKernelGGL(clear_words_kernel, dim3(1), dim3(16), 0, st, flag());
    return hipGetLastError() == hipSuccess ? ALINE_OK : ALINE_ELAUNCH;
  }
};

inline dim3 grid1d(size_t total, int block = 256) { return dim3((unsigned)((total + block - 1) / block)); }

// design selection of one step (model/head.py:347-362): LDS = logits | compacted slots | probabilities, 12 bytes per point slot.
// ALINE_MAX_POINTS (4096) covers the evaluation protocol's n_query = 2000 (README.md:45,50) within the default 64 KB of dynamic LDS.
int launch_acq_select(const Ctx &c, const SelectArgs &sel) {
  if (sel.g.P > ALINE_MAX_POINTS) return ALINE_EUNSUPPORTED;
  SelectArgs a = sel;
  a.range_flag = c.flag();
  if (a.logits && a.g.P <= 256 && a.g.inst_B == 0 && !dbg(ALINE_DBG_SELECT_WORKGROUP)) {      // one wave per episode (kernels.h)
    hipLaunchKernelGGL(acq_select_wave_kernel, dim3((unsigned)((a.g.B + 3) / 4)), dim3(256), 0, c.st, a);
    CHECK_LAUNCH();
    return ALINE_OK;
  }
  hipLaunchKernelGGL(acq_select_kernel, dim3(a.g.B), dim3(256), (size_t)a.g.P * 12, c.st, a);
  CHECK_LAUNCH();
  return ALINE_OK;
}

// forward GEMM in the model's precision, with the f16 range guard attached
int launch_gemm_fwd(const Ctx &c, GemmArgs a, int groups) {
  a.range_flag = c.flag();
  return launch_gemm(c.m->precision, a, groups, c.st);
}

// x/y point embedders -> Ex [B*(P+n_td), d], Ey [B*ey_rows, d]     (model/embedder.py:47-57)
int do_embed_points(const Ctx &c, Src3 xs, const float *ysrc, int ey_rows) {
  const aline_model &m = *c.m;
  const int rows_x = c.g.B * (c.g.P + c.g.n_td);
  float *hid = c.at(c.pl.Hid);
  hipLaunchKernelGGL(embed_hidden_kernel, grid1d((size_t)rows_x * m.F), dim3(256), 0, c.st, xs,
                     c.g.P + c.g.n_td, c.g.B, m.dim_x, m.F, m.x_w1, m.x_b1, hid);
  CHECK_LAUNCH();
  TRY(launch_gemm_fwd(c, gemm_args(hid, m.F, m.x_w2, m.x_b2, m.F, c.at(c.pl.Ex), m.d, rows_x,
                                         m.d, m.F, false), 1));
  CHECK_LAUNCH();
  const int rows_y = c.g.B * ey_rows;
  if (rows_y > 0) {
    Src3 ys{{ysrc, nullptr, nullptr}, {ey_rows, 0, 0}};
    hipLaunchKernelGGL(embed_hidden_kernel, grid1d((size_t)rows_y * m.F), dim3(256), 0, c.st, ys,
                       ey_rows, c.g.B, m.dim_y, m.F, m.y_w1, m.y_b1, hid);
    CHECK_LAUNCH();
    TRY(launch_gemm_fwd(c, gemm_args(hid, m.F, m.y_w2, m.y_b2, m.F, c.at(c.pl.Ey), m.d, rows_y,
                                           m.d, m.F, false), 1));
    CHECK_LAUNCH();
  }
  return ALINE_OK;
}

int do_assemble(const Ctx &c, int ey_rows, float *X) {
  const size_t total = (size_t)c.g.B * c.g.N * c.m->d;
  hipLaunchKernelGGL(assemble_kernel, grid1d(total), dim3(256), 0, c.st, c.g, c.m->d, c.at(c.pl.Ex),
                     c.at(c.pl.Ey), ey_rows, c.m->theta_tokens, X);
  CHECK_LAUNCH();
  return ALINE_OK;
}

template <int HD>
int launch_attention(const Ctx &c, const float *qkv, float *out, int max_keys, const float *kvc = nullptr, const int *kcnt = nullptr) {
  size_t smem = (size_t)max_keys * (2 * HD * sizeof(float) + sizeof(int));
  if (smem > 160 * 1024 - 1024) return ALINE_EUNSUPPORTED;
  if (smem > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&attention_kernel<HD>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const unsigned nthr = (unsigned)std::min(512, std::max(256, (c.g.N + 63) / 64 * 64));   // one token row per thread when N <= 512
  hipLaunchKernelGGL(attention_kernel<HD>, dim3((unsigned)((c.g.B + 7) / 8 * 8 * c.m->H)), dim3(nthr), smem, c.st, c.g, c.m->d,
                     qkv, out, max_keys, kvc, kcnt);
  CHECK_LAUNCH();
  return ALINE_OK;
}

// out = LayerNorm(a + b) * w + bias, optionally saving the pre-norm sum (narrow rows: d = 32 / 64)
static int launch_add_layernorm(hipStream_t st, const float *a, const float *b, const float *w, const float *bias,
                                float *out, long M, int d, float *usave) {
  if (d == 32 || d == 64) {
    const int rpb = d == 32 ? 32 : 16;
    const unsigned grid = (unsigned)std::min<long>((M + rpb - 1) / rpb, 256 * 32);
    if (d == 32) hipLaunchKernelGGL(add_layernorm_narrow_kernel<8>, dim3(grid), dim3(256), 0, st, a, b, w, bias, out, M, usave);
    else hipLaunchKernelGGL(add_layernorm_narrow_kernel<16>, dim3(grid), dim3(256), 0, st, a, b, w, bias, out, M, usave);
  } else if (d == 256 || d == 512) {
    const unsigned grid = (unsigned)std::min<long>((M + 3) / 4, 256 * 16);
    if (d == 256) hipLaunchKernelGGL(add_layernorm_wide_kernel<1>, dim3(grid), dim3(256), 0, st, a, b, w, bias, out, M, usave);
    else hipLaunchKernelGGL(add_layernorm_wide_kernel<2>, dim3(grid), dim3(256), 0, st, a, b, w, bias, out, M, usave);
  } else {
    hipLaunchKernelGGL(add_layernorm_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, a, b, w, bias, out, M, d, usave);
  }
  CHECK_LAUNCH();
  return ALINE_OK;
}

// Encoder.forward: L post-norm layers (model/encoder.py:128-141)
int do_encoder(const Ctx &c, const float *x_in, float *x_out, int max_keys) {
  const aline_model &m = *c.m;
  const int M = c.g.B * c.g.N, d = m.d, F = m.F;
  float *X = c.at(c.pl.X), *X1 = c.at(c.pl.X1), *QKV = c.at(c.pl.QKV), *A = c.at(c.pl.A),
        *Tm = c.at(c.pl.Tm), *Hid = c.at(c.pl.Hid);
  const float *cur = x_in;
  const int hd = d / m.H;
  // small-width model in reference precision: the token-local tail of a layer (out-projection, LN1, FFN, LN2) is
  // one kernel on the packed layer images of the fused path (fused_tail.h)
  const bool tail = (m.precision == ALINE_PREC_F32 || m.precision == ALINE_PREC_F16X3) && d == fused::D && F == fused::F && !dbg(ALINE_DBG_NO_LAYER_TAIL);
  float *wimg = c.at(c.pl.Wpack);
  if (tail) {
    fused::PackArgs pa{};
    pa.L = m.L; pa.layers_only = 1; pa.out = wimg;
    for (int l = 0; l < m.L; ++l) {
      pa.in_proj_w[l] = m.in_proj_w[l]; pa.in_proj_b[l] = m.in_proj_b[l];
      pa.out_proj_w[l] = m.out_proj_w[l]; pa.out_proj_b[l] = m.out_proj_b[l];
      pa.lin1_w[l] = m.lin1_w[l]; pa.lin1_b[l] = m.lin1_b[l];
      pa.lin2_w[l] = m.lin2_w[l]; pa.lin2_b[l] = m.lin2_b[l];
      pa.n1w[l] = m.norm1_w[l]; pa.n1b[l] = m.norm1_b[l];
      pa.n2w[l] = m.norm2_w[l]; pa.n2b[l] = m.norm2_b[l];
    }
    hipLaunchKernelGGL(fused::pack_weights_kernel, dim3(64), dim3(256), 0, c.st, pa);
    CHECK_LAUNCH();
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fused::layer_tail_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(fused::LAYER_FLOATS * sizeof(float)));
  }
  // Q for every token row, K / V for the key rows only (context points + visible targets: max_keys of the N rows of an
  // episode, e.g. 34 of 205 at cfg5): the key list is the same for all layers of a step
  const bool compact_kv = max_keys < c.g.N && !dbg(ALINE_DBG_FULL_QKV);
  int *keyidx = reinterpret_cast<int *>(c.at(c.pl.KeyIdx)), *kcnt = reinterpret_cast<int *>(c.at(c.pl.Kcnt));
  float *KVc = QKV + (size_t)M * d;
  const int Mk = c.g.B * max_keys;
  if (compact_kv) {
    hipLaunchKernelGGL(key_list_kernel, dim3(c.g.B), dim3(256), 0, c.st, c.g, max_keys, keyidx, kcnt);
    CHECK_LAUNCH();
  }
  for (int l = 0; l < m.L; ++l) {
    if (compact_kv) {
      TRY(launch_gemm_fwd(c, gemm_args(cur, d, m.in_proj_w[l], m.in_proj_b[l], d, QKV, d, M, d, d, false), 1));
      CHECK_LAUNCH();
      GemmArgs ka = gemm_args(cur, d, m.in_proj_w[l] + (size_t)d * d, m.in_proj_b[l] + d, d, KVc, 2 * d, Mk, 2 * d, d, false);
      ka.row_index = keyidx;
      TRY(launch_gemm_fwd(c, ka, 1));
      CHECK_LAUNCH();
    } else {
      TRY(launch_gemm_fwd(c, gemm_args(cur, d, m.in_proj_w[l], m.in_proj_b[l], d, QKV, 3 * d, M,
                                             3 * d, d, false), 1));
      CHECK_LAUNCH();
    }
    const float *kvc = compact_kv ? KVc : nullptr;
    // head_dim 32 / 64 outside the exact-fp32 mode, up to 64 keys: the attention on the matrix pipe (attn3.h)
    if (compact_kv && (hd == 32 || hd == 64) && max_keys <= 16 * attn3::MAX_KT && m.precision != ALINE_PREC_F32 &&
        !dbg(ALINE_DBG_VALU_ATTENTION)) {
      const dim3 grid((unsigned)((c.g.B + 7) / 8 * 8 * m.H));
      if (hd == 64) hipLaunchKernelGGL(attn3::attention_kernel<64>, grid, dim3(256), 0, c.st, c.g, d, QKV, KVc, kcnt, A, max_keys);
      else hipLaunchKernelGGL(attn3::attention_kernel<32>, grid, dim3(256), 0, c.st, c.g, d, QKV, KVc, kcnt, A, max_keys);
      CHECK_LAUNCH();
    } else
    switch (hd) {
      case 4: TRY(launch_attention<4>(c, QKV, A, max_keys, kvc, kcnt)); break;
      case 8: TRY(launch_attention<8>(c, QKV, A, max_keys, kvc, kcnt)); break;
      case 16: TRY(launch_attention<16>(c, QKV, A, max_keys, kvc, kcnt)); break;
      case 32: TRY(launch_attention<32>(c, QKV, A, max_keys, kvc, kcnt)); break;
      case 64: TRY(launch_attention<64>(c, QKV, A, max_keys, kvc, kcnt)); break;
      case 128: TRY(launch_attention<128>(c, QKV, A, max_keys, kvc, kcnt)); break;
      default: return ALINE_EUNSUPPORTED;
    }
    if (tail) {
      float *dst = (l == m.L - 1 && x_out) ? x_out : X;
      fused::TailArgs ta{A, cur, dst, (long)M, wimg + (size_t)l * fused::LAYER_FLOATS};
      const long groups = ((long)M + 31) / 32;
      const unsigned grid = (unsigned)std::min<long>((groups + 7) / 8, 512);     // 2 workgroups per CU, each loops
      hipLaunchKernelGGL(fused::layer_tail_kernel, dim3(grid), dim3(fused::TAIL_THREADS),
                         fused::LAYER_FLOATS * sizeof(float), c.st, ta);
      CHECK_LAUNCH();
      cur = dst;
      continue;
    }
    TRY(launch_gemm_fwd(c, gemm_args(A, d, m.out_proj_w[l], m.out_proj_b[l], d, Tm, d, M, d, d,
                                           false), 1));
    CHECK_LAUNCH();
    TRY(launch_add_layernorm(c.st, cur, Tm, m.norm1_w[l], m.norm1_b[l], X1, (long)M, d, nullptr));
    TRY(launch_gemm_fwd(c, gemm_args(X1, d, m.lin1_w[l], m.lin1_b[l], d, Hid, F, M, F, d, true), 1));
    CHECK_LAUNCH();
    TRY(launch_gemm_fwd(c, gemm_args(Hid, F, m.lin2_w[l], m.lin2_b[l], F, Tm, d, M, d, F, false), 1));
    CHECK_LAUNCH();
    float *dst = (l == m.L - 1 && x_out) ? x_out : X;
    TRY(launch_add_layernorm(c.st, X1, Tm, m.norm2_w[l], m.norm2_b[l], dst, (long)M, d, nullptr));
    cur = dst;
  }
  return ALINE_OK;
}

struct HeadIO {
  const float *time_t;
  const float *target_all;
  SelectArgs sel;                         // g/F/hid/w2/b2 filled by do_head
  float *post_mean, *post_std, *post_weight, *target_ll;   // [B*n_t(,C)]
  float *postq_mean, *postq_std, *postq_weight;            // [B*nq_rows, C]; rows q_off.. of each episode
  int q_off, q_rows;
};

// GMM head on `rows_per_ep` token rows starting at `row_off` of every episode in [b0, b0+nb)
// GMMTargetHead (model/head.py:152-186): C grouped first layers on the matrix cores with the [3, F] second layers
// reduced in the GEMM epilogue (raw[row][3 c + j]; the hidden activations are never stored), then the parameter
// maps / mixture log-likelihood per row.
static int gmm_heads(const Ctx &c, GemmArgs a, int rows, float *mean, float *sd, float *wgt, const float *value,
                     float *ll, long value_row0, long value_mod) {
  const aline_model &m = *c.m;
  if ((m.precision == ALINE_PREC_F32 || m.precision == ALINE_PREC_F16X3) && m.d == fused::D && m.F == fused::F && !dbg(ALINE_DBG_NO_LAYER_TAIL)) {
    // small-width model: all C heads, the parameter maps and compute_ll in one kernel on the transposed register
    // scheme of the fused path (fused_side.h), first layers as packed split-bf16 fragments
    float *side = c.at(c.pl.Wpack) + (size_t)ALINE_MAX_LAYERS * fused::LAYER_FLOATS + fused::HEAD_FLOATS;
    fused::PackArgs pa{};
    pa.layers_only = 2; pa.C = m.C; pa.out = side;
    for (int k = 0; k < m.C; ++k) pa.gmm_w1[k] = m.gmm_w1[k];
    hipLaunchKernelGGL(fused::pack_weights_kernel, dim3(60), dim3(256), 0, c.st, pa);
    CHECK_LAUNCH();
    fused::GmmRowsArgs ga{};
    ga.z = a.X; ga.rows = rows; ga.zR = a.R_in; ga.zG = a.G_in; ga.zoff = a.off_in;
    ga.C = m.C; ga.std_min = m.std_min; ga.w1img = side;
    for (int k = 0; k < m.C; ++k) { ga.b1[k] = m.gmm_b1[k]; ga.w2[k] = m.gmm_w2[k]; ga.b2[k] = m.gmm_b2[k]; }
    ga.mean = mean; ga.sd = sd; ga.wgt = wgt;
    ga.value = value; ga.value_row0 = value_row0; ga.value_mod = value_mod;
    ga.ll = (ll && value) ? ll : nullptr;
    hipLaunchKernelGGL(fused::gmm_rows_kernel, dim3((unsigned)((rows + fused::GMM_THREADS * fused::GMM_TILES / 4 - 1) / (fused::GMM_THREADS * fused::GMM_TILES / 4))), dim3(fused::GMM_THREADS), 0, c.st, ga);
    CHECK_LAUNCH();
    return ALINE_OK;
  }
  float *raw = c.at(c.pl.Hid);
  const int stride = (3 * m.C + 3) / 4 * 4;
  a.col_per_group = 0;
  for (int k = 0; k < m.C; ++k) { a.W[k] = m.gmm_w1[k]; a.bias[k] = m.gmm_b1[k]; a.red_w[k] = m.gmm_w2[k]; a.red_b[k] = m.gmm_b2[k]; }
  a.red_nout = 3; a.red_out = raw; a.red_stride = stride; a.red_block_stride = (long)rows * stride;
  TRY(launch_gemm_fwd(c, a, m.C));
  CHECK_LAUNCH();
  img::GmmRawArgs f{};
  f.range_flag = c.flag();
  f.raw = raw; f.raw_stride = stride; f.rows = rows; f.C = m.C; f.std_min = m.std_min;
  f.nblk = gemm_col_blocks(m.F); f.blk_stride = (long)rows * stride;
  f.mean = mean; f.sd = sd; f.wgt = wgt;
  f.value = value; f.ll = (ll && value) ? ll : nullptr; f.value_row0 = value_row0; f.value_mod = value_mod;
  hipLaunchKernelGGL(img::gmm_raw_finish_kernel, dim3((rows + 255) / 256), dim3(256), 0, c.st, f);
  CHECK_LAUNCH();
  return ALINE_OK;
}

int do_gmm(const Ctx &c, const float *Z, int b0, int nb, int row_off, int rows_per_ep, float *mean,
           float *sd, float *wgt, const float *value, float *ll) {
  const aline_model &m = *c.m;
  const int rows = nb * rows_per_ep;
  if (rows <= 0) return ALINE_OK;
  GemmArgs a = gemm_args(Z + (size_t)b0 * c.g.N * m.d, m.d, nullptr, nullptr, m.d, nullptr, m.C * m.F, rows,
                         m.F, m.d, true);
  a.R_in = rows_per_ep; a.G_in = c.g.N; a.off_in = row_off;
  const size_t o = (size_t)b0 * rows_per_ep;
  return gmm_heads(c, a, rows, mean ? mean + o * m.C : nullptr, sd ? sd + o * m.C : nullptr,
                   wgt ? wgt + o * m.C : nullptr, value ? value + o : nullptr, ll ? ll + o : nullptr, 0, rows);
}
int do_gmm_rows(const Ctx &c, const float *Zrows, int rows, float *mean, float *sd, float *wgt,
                const float *value, float *ll, long row0, long value_mod) {
  const aline_model &m = *c.m;
  if (rows <= 0) return ALINE_OK;
  GemmArgs a = gemm_args(Zrows, m.d, nullptr, nullptr, m.d, nullptr, m.C * m.F, rows, m.F, m.d, true);
  return gmm_heads(c, a, rows, mean, sd, wgt, value, ll, value_mod > 0 ? row0 : 0, value_mod > 0 ? value_mod : rows);
}
int do_acquisition(const Ctx &c, const float *Z, HeadIO io) {
  const aline_model &m = *c.m;
  const Geo &g = c.g;
  // acquisition MLP first layer on the P point rows of every episode
  const float *w1 = m.acq_w1;
  int ldw = m.d;
  if (m.time_token) {
    if (!io.time_t) return ALINE_EINVAL;
    hipLaunchKernelGGL(pack_cols_kernel, grid1d((size_t)m.F * m.d), dim3(256), 0, c.st, m.acq_w1,
                       m.d + 1, m.F, m.d, c.at(c.pl.Wacq));
    CHECK_LAUNCH();
    w1 = c.at(c.pl.Wacq);
  }
  // the [F] -> 1 second layer is reduced in the GEMM epilogue: logits[b * P + p], no hidden activations in memory
  float *logits = c.at(c.pl.Hid);
  GemmArgs a = gemm_args(Z, m.d, w1, m.acq_b1, ldw, nullptr, m.F, g.B * g.P, m.F, m.d, true);
  a.R_in = g.P; a.G_in = g.N; a.off_in = 0;
  if (m.time_token) { a.tscalar = io.time_t; a.tcol = m.acq_w1 + m.d; a.tcol_stride = m.d + 1; }
  a.red_w[0] = m.acq_w2; a.red_b[0] = m.acq_b2; a.red_nout = 1; a.red_out = logits; a.red_stride = 1;
  a.red_block_stride = (long)g.B * g.P;
  TRY(launch_gemm_fwd(c, a, 1));
  CHECK_LAUNCH();
  io.sel.g = g; io.sel.F = m.F; io.sel.hid = nullptr; io.sel.w2 = m.acq_w2; io.sel.b2 = m.acq_b2;
  io.sel.logits = logits; io.sel.logit_stride = g.P;
  io.sel.logit_nblk = gemm_col_blocks(m.F); io.sel.logit_blk_stride = (long)g.B * g.P;
  TRY(launch_acq_select(c, io.sel));
  CHECK_LAUNCH();
  return ALINE_OK;
}

// OutputHead.forward (model/head.py:319-393)
int do_head(const Ctx &c, const float *Z, HeadIO io) {
  const Geo &g = c.g;
  const int n_t = g.n_td + g.n_th;
  const bool want_sel = io.sel.idx || io.sel.log_prob || io.sel.zt || io.sel.slot || io.sel.role_out;
  if (want_sel) TRY(do_acquisition(c, Z, io));
  // posterior over the targets (+ compute_ll)
  if (io.post_mean || io.post_std || io.post_weight || io.target_ll)
    TRY(do_gmm(c, Z, 0, g.B, g.P + 0, n_t, io.post_mean, io.post_std, io.post_weight, io.target_all,
               io.target_ll));
  // posterior_out_query (head.py:366): chunked over episodes to bound the hidden activations
  if (io.postq_mean || io.postq_std || io.postq_weight) {
    if (c.pl.qgmm_chunk <= 0) return ALINE_EWORKSPACE;
    for (int b0 = 0; b0 < g.B; b0 += c.pl.qgmm_chunk) {
      int nb = std::min(c.pl.qgmm_chunk, g.B - b0);
      TRY(do_gmm(c, Z, b0, nb, io.q_off, io.q_rows, io.postq_mean, io.postq_std, io.postq_weight,
                 nullptr, nullptr));
    }
  }
  return ALINE_OK;
}

int step_geo(const aline_model &m, const aline_step &s, Geo &g) {
  if (s.B <= 0 || s.n_ctx < 1 || s.n_query < 1 || s.n_target_data < 0) return ALINE_EINVAL;
  if (m.embedding_type == ALINE_EMB_THETA && s.n_target_data != 0) return ALINE_EINVAL;
  g.B = s.B; g.P = s.n_ctx + s.n_query; g.n_td = s.n_target_data; g.n_th = m.n_theta;
  g.N = g.P + g.n_td + g.n_th; g.n_ctx = s.n_ctx; g.role = nullptr; g.tmask = s.target_mask;
  g.inst_B = 0; g.inst_t0 = 0; g.n_ctx0 = 0;
  return ALINE_OK;
}

bool wants_query_gmm(const aline_step &s) { return s.postq_mean || s.postq_std || s.postq_weight; }

int step_ctx(const aline_model *m, const aline_step *s, void *ws, size_t ws_bytes, void *stream,
             Ctx &c, int stages) {
  if (!m || !s || !ws) return ALINE_EINVAL;
  TRY(validate_model(*m, stages));
  TRY(step_geo(*m, *s, c.g));
  c.m = m;
  c.pl = make_plan(*m, s->B, c.g.P, c.g.n_td, s->n_ctx, wants_query_gmm(*s));
  if (ws_bytes < c.pl.total * sizeof(float)) return ALINE_EWORKSPACE;
  c.ws = static_cast<float *>(ws);
  c.st = static_cast<hipStream_t>(stream);
  return ALINE_OK;
}

int step_embed(const Ctx &c, const aline_step &s, float *X) {
  if (!s.context_x || !s.context_y || !s.query_x) return ALINE_EINVAL;
  if (c.g.n_td > 0 && !s.target_x) return ALINE_EINVAL;
  Src3 xs{{s.context_x, s.query_x, s.target_x}, {s.n_ctx, s.n_query, c.g.n_td}};
  TRY(do_embed_points(c, xs, s.context_y, s.n_ctx));
  return do_assemble(c, s.n_ctx, X);
}

HeadIO step_head_io(const aline_step &s, int n_ctx) {
  HeadIO io{};
  io.time_t = s.time_t;
  io.target_all = s.target_all;
  io.sel.mode = s.select_mode;
  io.sel.uniform = s.uniform;
  io.sel.forced = s.forced_idx; io.sel.forced_stride = 1;
  io.sel.idx = s.idx; io.sel.idx_stride = 1;
  io.sel.slot = nullptr; io.sel.slot_stride = 0;
  io.sel.log_prob = s.log_prob; io.sel.lp_stride = 1;
  io.sel.zt = s.zt; io.sel.zt_stride = s.n_query; io.sel.zt_width = s.n_query;
  io.sel.role_out = nullptr;
  io.post_mean = s.post_mean; io.post_std = s.post_std; io.post_weight = s.post_weight;
  io.target_ll = s.target_ll;
  io.postq_mean = s.postq_mean; io.postq_std = s.postq_std; io.postq_weight = s.postq_weight;
  io.q_off = n_ctx; io.q_rows = s.n_query;
  return io;
}

int check_select(int mode, const float *uniform, const int64_t *forced) {
  if (mode == ALINE_SELECT_SAMPLE && !uniform) return ALINE_EINVAL;
  if (mode == ALINE_SELECT_FORCED && !forced) return ALINE_EINVAL;
  if (mode < 0 || mode > 2) return ALINE_EINVAL;
  return ALINE_OK;
}

__global__ void set_scalar_kernel(float *p, float v) { p[0] = v; }

}  // namespace



// The time token of step t (model.time_token): t / T as the training loop feeds it (train_aline.py:82); time_token_T < 0 selects the
// schedule of the reference's evaluation loop, (T - t) / T (utils/eval.py:24)
static float step_time_token(const aline_rollout &r, int t) {
  const int TT = r.time_token_T > 0 ? r.time_token_T : r.time_token_T < 0 ? -r.time_token_T : r.T;
  return r.time_token_T < 0 ? (float)(TT - t) / (float)TT : (float)t / (float)TT;
}

static int device_cus() {      // compute units of the CURRENT device (cached per device: a process may drive several)
  static std::atomic<int> cache[64] = {};
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = cache[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    n = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    cache[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

// Launch shape of the s3 step kernel: 12 waves per workgroup (3 per SIMD, 170 registers: no spills; 16 waves at 128
// registers measured 1.5 % slower, ALINE_DBG_S3_WAVES = 16) while an episode never holds more than 64 keys, 8 waves (256
// registers: all scores of a head pair over 160 keys) otherwise; as many episodes per
// workgroup as keeps every CU busy and the token tiles spread evenly over the waves.
struct S3Shape { int nw, nkp, epw; unsigned nwg; size_t lds; bool sel; };     // nkp: key-tile pairs an episode's LDS slot holds; sel: the design selection runs inside the step kernel
static S3Shape s3_shape(const aline_model &m, const aline_rollout &r) {
  S3Shape s{};
  const int n_t = r.n_target_data + m.n_theta, N = r.P + n_t, tpe = (N + 15) / 16;
  const int nkeys = r.n_ctx0 + r.T - 1 + n_t;
  const int need = std::max(1, (nkeys + 31) / 32);
  s.nw = need <= 2 ? 12 : 8;
  if (const int w = dbg_param(ALINE_DBG_S3_WAVES)) s.nw = (w == 16 || w == 12) && need <= 2 ? w : 8;
  s.nkp = s.nw >= 12 ? 2 : s3::NKP_MAX;
  // the design selection inside the step kernel (s3.h: SelArgs) while an episode's logits fit its LDS line.  ALINE_DBG_S3_SELECT_KERNEL: the launch of its own
  s.sel = r.P <= s3::SEL_PMAX && r.role && !dbg(ALINE_DBG_S3_SELECT_KERNEL);
  const int per_ep = s3::kv_ep_bytes(s.nkp) + 32 * s.nkp * 4 + (s.sel ? s3::SEL_PMAX * 4 + s3::SEL_PMAX / 8 : 0);
  const int epw_max = std::max(1, std::min(s3::EPW_MAX, (s3::LDS_LIMIT - s3::KV_OFF - s3::MISC_INTS * 4) / per_ep));
  const int cus = device_cus();
  double best = 1e30;
  for (int e = 1; e <= epw_max; ++e) {
    const long nwg = (r.B + e - 1) / e;
    const double rounds = (double)((nwg + cus - 1) / cus);                    // workgroup rounds over the chip
    const double slots = std::ceil((double)e * tpe / s.nw) + 0.35;             // tile rounds of a workgroup (+ its fixed work)
    const double cost = rounds * slots;
    if (cost < best - 1e-9) { best = cost; s.epw = e; }
  }
  if (const int e = dbg_param(ALINE_DBG_S3_EPW)) s.epw = std::max(1, std::min(epw_max, e));
  s.nwg = (unsigned)((r.B + s.epw - 1) / s.epw);
  s.lds = (size_t)s3::step_lds_bytes(s.epw, s.nkp, s.sel);
  return s;
}

template <int F, int NW, int MAXNKP>
static int launch_s3_step_v(const Ctx &c, const S3Shape &sh, const s3::StepArgs &a) {
  constexpr bool PF = false;
  // (set at every launch: the attribute is per device, and a process may drive several)
  if constexpr (F == 128) {      // (the width of the fused backward: the only one whose rollouts are asked to keep their activations)
    if (a.sv) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&s3::step_kernel<F, NW, MAXNKP, PF, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh.lds);
      hipLaunchKernelGGL((s3::step_kernel<F, NW, MAXNKP, PF, true>), dim3(sh.nwg), dim3(NW * 64), sh.lds, c.st, a);
      CHECK_LAUNCH();
      return ALINE_OK;
    }
  }
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&s3::step_kernel<F, NW, MAXNKP, PF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh.lds);
  hipLaunchKernelGGL((s3::step_kernel<F, NW, MAXNKP, PF>), dim3(sh.nwg), dim3(NW * 64), sh.lds, c.st, a);
  CHECK_LAUNCH();
  return ALINE_OK;
}
template <int F>
static int launch_s3_step_f(const Ctx &c, const S3Shape &sh, const s3::StepArgs &a) {
  if (sh.nw == 12) return launch_s3_step_v<F, 12, 2>(c, sh, a);
  return sh.nw == 16 ? launch_s3_step_v<F, 16, 2>(c, sh, a) : launch_s3_step_v<F, 8, s3::NKP_MAX>(c, sh, a);
}
static int launch_s3_step(const Ctx &c, const S3Shape &sh, const s3::StepArgs &a) {
  switch (a.F) {
    case 32: return launch_s3_step_f<32>(c, sh, a);
    case 64: return launch_s3_step_f<64>(c, sh, a);
    case 96: return launch_s3_step_f<96>(c, sh, a);
    case 128: return launch_s3_step_f<128>(c, sh, a);
    default: return ALINE_EUNSUPPORTED;
  }
}
// the x rows and the y rows of a rollout in ONE launch (two were 2 x 42 us at the headline shape, most of it ramp: each workgroup packs its W2 pairs first)
template <int F>
static int launch_s3_embed_f(const Ctx &c, const s3::EmbArgs &a, const s3::EmbArgs &b) {
  const long ta = ((long)a.B * a.rows_per_ep + 15) / 16, tb = ((long)b.B * b.rows_per_ep + 15) / 16;
  // (every workgroup first packs the W2 fragment pairs into LDS: a few workgroups per CU that walk the tiles, not one per 4 tiles)
  const long cap = 2L * device_cus();
  const unsigned na = (unsigned)std::max<long>(1, std::min<long>((ta + 3) / 4, cap)), nb = (unsigned)std::max<long>(1, std::min<long>((tb + 3) / 4, cap));
  hipLaunchKernelGGL(s3::embed_kernel<F>, dim3(na + nb), dim3(256), 0, c.st, a, b, na);
  CHECK_LAUNCH();
  return ALINE_OK;
}
static int launch_s3_embed(const Ctx &c, int F, const s3::EmbArgs &a, const s3::EmbArgs &b) {
  switch (F) {
    case 32: return launch_s3_embed_f<32>(c, a, b);
    case 64: return launch_s3_embed_f<64>(c, a, b);
    case 96: return launch_s3_embed_f<96>(c, a, b);
    case 128: return launch_s3_embed_f<128>(c, a, b);
    default: return ALINE_EUNSUPPORTED;
  }
}
template <int F>
static int launch_s3_gmm_f(const Ctx &c, const s3::GmmArgs &a) {
  const size_t smem = (size_t)s3::head_bytes(F) + (size_t)3 * a.C * s3::GROWS * sizeof(float);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&s3::gmm_kernel<F>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const long per_wg = (long)s3::WAVES * s3::GT;
  hipLaunchKernelGGL(s3::gmm_kernel<F>, dim3((unsigned)((a.ntiles + per_wg - 1) / per_wg)), dim3(s3::THREADS), smem, c.st, a);
  CHECK_LAUNCH();
  return ALINE_OK;
}
static int launch_s3_gmm(const Ctx &c, int F, const s3::GmmArgs &a) {
  switch (F) {
    case 32: return launch_s3_gmm_f<32>(c, a);
    case 64: return launch_s3_gmm_f<64>(c, a);
    case 96: return launch_s3_gmm_f<96>(c, a);
    case 128: return launch_s3_gmm_f<128>(c, a);
    default: return ALINE_EUNSUPPORTED;
  }
}



extern "C" {

int aline_abi_version(void) { return ALINE_ABI_VERSION; }

const char *aline_error_string(int code) {
  switch (code) {
    case ALINE_OK: return "ok";
    case ALINE_EINVAL: return "invalid argument";
    case ALINE_EUNSUPPORTED: return "unsupported shape/configuration";
    case ALINE_EWORKSPACE: return "workspace too small";
    case ALINE_ELAUNCH: return "kernel launch failed";
    case ALINE_ERANGE: return "an F16X3 operand left f16's range (non-finite or >= 65504)";
  }
  return "unknown error";
}

size_t aline_step_workspace_bytes(const aline_model *m, const aline_step *s) {
  if (!m || !s || validate_model(*m, 0) != 0) return 0;
  Geo g;
  if (step_geo(*m, *s, g) != 0) return 0;
  return make_plan(*m, s->B, g.P, g.n_td, s->n_ctx, wants_query_gmm(*s)).total * sizeof(float);
}

int aline_embed_forward(const aline_model *m, const aline_step *s, void *ws, size_t ws_bytes,
                        void *stream) {
  Ctx c;
  TRY(step_ctx(m, s, ws, ws_bytes, stream, c, ST_EMBED));
  if (!s->embedding) return ALINE_EINVAL;
  TRY(c.clear_flag());
  return step_embed(c, *s, s->embedding);
}

int aline_encoder_forward(const aline_model *m, const aline_step *s, const float *x_in, void *ws,
                          size_t ws_bytes, void *stream) {
  Ctx c;
  TRY(step_ctx(m, s, ws, ws_bytes, stream, c, ST_ENC));
  if (!x_in || !s->encoding) return ALINE_EINVAL;
  return do_encoder(c, x_in, s->encoding, c.g.n_ctx + c.g.n_td + c.g.n_th);
}

int aline_head_forward(const aline_model *m, const aline_step *s, const float *z, void *ws,
                       size_t ws_bytes, void *stream) {
  Ctx c;
  TRY(step_ctx(m, s, ws, ws_bytes, stream, c, ST_HEAD));
  if (!z) return ALINE_EINVAL;
  if (s->idx || s->log_prob || s->zt) TRY(check_select(s->select_mode, s->uniform, s->forced_idx));
  return do_head(c, z, step_head_io(*s, s->n_ctx));
}

int aline_step_forward(const aline_model *m, const aline_step *s, void *ws, size_t ws_bytes,
                       void *stream) {
  Ctx c;
  TRY(step_ctx(m, s, ws, ws_bytes, stream, c, ST_ALL));
  TRY(check_select(s->select_mode, s->uniform, s->forced_idx));
  TRY(c.clear_flag());
  float *X0 = s->embedding ? s->embedding : c.at(c.pl.X);
  TRY(step_embed(c, *s, X0));
  TRY(do_encoder(c, X0, s->encoding, c.g.n_ctx + c.g.n_td + c.g.n_th));
  const float *Z = s->encoding ? s->encoding : c.at(c.pl.X);
  return do_head(c, Z, step_head_io(*s, s->n_ctx));
}

// ---- rollout (static slots) --------------------------------------------------------------------
static bool wants_postq(const aline_rollout &r) { return r.postq_mean || r.postq_std || r.postq_weight; }

static int rollout_ctx(const aline_model *m, const aline_rollout *r, void *ws, size_t ws_bytes,
                       void *stream, Ctx &c) {
  if (!m || !r || !ws) return ALINE_EINVAL;
  TRY(validate_model(*m));
  if (r->B <= 0 || r->P < 2 || r->n_ctx0 < 1 || r->n_ctx0 >= r->P || r->T < 1 ||
      r->n_ctx0 + r->T > r->P || !r->role || !r->point_x || !r->point_y)
    return ALINE_EINVAL;
  if (m->embedding_type == ALINE_EMB_THETA && r->n_target_data != 0) return ALINE_EINVAL;
  if (r->n_target_data > 0 && !r->target_x) return ALINE_EINVAL;
  c.g.B = r->B; c.g.P = r->P; c.g.n_td = r->n_target_data; c.g.n_th = m->n_theta;
  c.g.N = c.g.P + c.g.n_td + c.g.n_th; c.g.n_ctx = r->n_ctx0; c.g.role = r->role;
  c.g.tmask = r->target_mask;
  c.g.inst_B = 0; c.g.inst_t0 = 0; c.g.n_ctx0 = r->n_ctx0;
  c.m = m;
  c.pl = make_plan(*m, r->B, r->P, r->n_target_data, r->P, wants_postq(*r), r->T);
  if (ws_bytes < c.pl.total * sizeof(float)) return ALINE_EWORKSPACE;
  c.ws = static_cast<float *>(ws);
  c.st = static_cast<hipStream_t>(stream);
  return ALINE_OK;
}

size_t aline_rollout_workspace_bytes(const aline_model *m, const aline_rollout *r) {
  if (!m || !r || validate_model(*m, 0) != 0) return 0;
  return make_plan(*m, r->B, r->P, r->n_target_data, r->P, wants_postq(*r), r->T).total * sizeof(float);
}

int aline_rollout_init(const aline_model *m, const aline_rollout *r, void *ws, size_t ws_bytes,
                       void *stream) {
  Ctx c;
  TRY(rollout_ctx(m, r, ws, ws_bytes, stream, c));
  TRY(c.clear_flag());
  hipLaunchKernelGGL(role_init_kernel, grid1d((size_t)r->B * r->P), dim3(256), 0, c.st, r->role, r->B,
                     r->P, r->n_ctx0);
  CHECK_LAUNCH();
  // x- and y-embeddings of every slot are step-invariant: compute them once per rollout
  Src3 xs{{r->point_x, r->target_x, nullptr}, {r->P, r->n_target_data, 0}};
  return do_embed_points(c, xs, r->point_y, r->P);
}

int aline_rollout_step(const aline_model *m, const aline_rollout *r, int t, void *ws, size_t ws_bytes,
                       void *stream) {
  Ctx c;
  TRY(rollout_ctx(m, r, ws, ws_bytes, stream, c));
  if (t < 0 || t >= r->T) return ALINE_EINVAL;
  TRY(check_select(r->select_mode, r->uniform, r->forced_idx));
  const int n_t = c.g.n_td + c.g.n_th;
  c.g.n_ctx = r->n_ctx0 + t;
  float *X = c.at(c.pl.X), *X0 = c.at(c.pl.X0);
  if (t == 0) {
    TRY(do_assemble(c, r->P, X0));
  } else {        // only the row of the point chosen at step t - 1 changes (it became a context point: + Ey)
    hipLaunchKernelGGL(patch_row_kernel, dim3(r->B), dim3(256), 0, c.st, c.g, m->d, c.at(c.pl.Ex), c.at(c.pl.Ey), r->P,
                       r->n_ctx0 + t, X0);
    CHECK_LAUNCH();
  }
  TRY(do_encoder(c, X0, nullptr, r->n_ctx0 + t + n_t));
  HeadIO io{};
  if (m->time_token) {
    float *sc = c.at(c.pl.scalar);
    hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, c.st, sc, step_time_token(*r, t));
    CHECK_LAUNCH();
    io.time_t = sc;
  }
  io.target_all = r->target_all;
  io.sel.mode = r->select_mode;
  io.sel.uniform = r->uniform ? r->uniform + (size_t)t * r->B : nullptr;
  io.sel.forced = r->forced_idx ? r->forced_idx + t : nullptr; io.sel.forced_stride = r->T;
  io.sel.idx = r->idx ? r->idx + t : nullptr; io.sel.idx_stride = r->T;
  io.sel.slot = r->slot ? r->slot + t : nullptr; io.sel.slot_stride = r->T;
  io.sel.log_prob = r->log_prob ? r->log_prob + t : nullptr; io.sel.lp_stride = r->T;
  const int zw = r->P - r->n_ctx0;
  io.sel.zt = r->zt ? r->zt + (size_t)t * r->B * zw : nullptr; io.sel.zt_stride = zw; io.sel.zt_width = zw;
  io.sel.role_out = r->role;
  const size_t po = (size_t)t * r->B * n_t;
  io.post_mean = r->post_mean ? r->post_mean + po * m->C : nullptr;
  io.post_std = r->post_std ? r->post_std + po * m->C : nullptr;
  io.post_weight = r->post_weight ? r->post_weight + po * m->C : nullptr;
  io.target_ll = r->target_ll ? r->target_ll + po : nullptr;
  // posterior_out_query by slot (all P point rows; the rows that are context points hold unspecified values)
  const size_t pq = (size_t)t * r->B * r->P * m->C;
  io.postq_mean = r->postq_mean ? r->postq_mean + pq : nullptr;
  io.postq_std = r->postq_std ? r->postq_std + pq : nullptr;
  io.postq_weight = r->postq_weight ? r->postq_weight + pq : nullptr;
  io.q_off = 0; io.q_rows = r->P;
  return do_head(c, X, io);
}

// The fused per-episode kernel (fused_rollout.h) covers the small-width theta-mode models.
static bool fused_eligible(const aline_model &m, const aline_rollout &r) {
  if (wants_postq(r)) return false;     // posterior_out_query of every step: the s3 and generic paths
  if (dbg(ALINE_DBG_DISABLE_FUSED)) return false;
  if (m.precision != ALINE_PREC_F32) return false;
  if (m.d != fused::D || m.F != fused::F || m.H != fused::H || m.time_token) return false;
  if (m.embedding_type != ALINE_EMB_THETA || r.n_target_data != 0) return false;
  if (m.n_theta < 1 || m.n_theta > fused::MAXNT || r.P + m.n_theta > 16 * fused::WPE * fused::MAXT) return false;
  if (r.n_ctx0 + r.T - 1 + m.n_theta > fused::NKMAX) return false;
  return true;
}

static int rollout_fused(const aline_model *m, const aline_rollout *r, void *ws, size_t ws_bytes,
                         void *stream) {
  Ctx c;
  TRY(rollout_ctx(m, r, ws, ws_bytes, stream, c));
  TRY(check_select(r->select_mode, r->uniform, r->forced_idx));
  TRY(c.clear_flag());
  fused::PackArgs pa{};
  pa.L = m->L;
  for (int l = 0; l < m->L; ++l) {
    pa.in_proj_w[l] = m->in_proj_w[l]; pa.in_proj_b[l] = m->in_proj_b[l];
    pa.out_proj_w[l] = m->out_proj_w[l]; pa.out_proj_b[l] = m->out_proj_b[l];
    pa.lin1_w[l] = m->lin1_w[l]; pa.lin1_b[l] = m->lin1_b[l];
    pa.lin2_w[l] = m->lin2_w[l]; pa.lin2_b[l] = m->lin2_b[l];
    pa.n1w[l] = m->norm1_w[l]; pa.n1b[l] = m->norm1_b[l];
    pa.n2w[l] = m->norm2_w[l]; pa.n2b[l] = m->norm2_b[l];
  }
  pa.acq_w1 = m->acq_w1; pa.acq_b1 = m->acq_b1; pa.acq_w2 = m->acq_w2; pa.acq_b2 = m->acq_b2;
  pa.C = m->C;
  for (int k = 0; k < m->C; ++k) pa.gmm_w1[k] = m->gmm_w1[k];
  pa.x_w2 = m->x_w2; pa.y_w2 = m->y_w2;
  pa.out = c.at(c.pl.Wpack);
  hipLaunchKernelGGL(fused::pack_weights_kernel, dim3(128), dim3(256), 0, c.st, pa);
  CHECK_LAUNCH();
  const float *side = c.at(c.pl.Wpack) + (size_t)m->L * fused::LAYER_FLOATS + fused::HEAD_FLOATS;
  // step-invariant point embeddings (x- and y-embedder of every slot): fused embedder kernel
  {
    const long rows = (long)r->B * r->P;
    fused::EmbedArgs ex{};
    ex.x = r->point_x; ex.K = m->dim_x; ex.rows = rows; ex.w1 = m->x_w1; ex.b1 = m->x_b1; ex.b2 = m->x_b2;
    ex.w2img = side + (size_t)m->C * fused::SIDE_FRAGS; ex.out = c.at(c.pl.Ex);
    const unsigned grid = (unsigned)std::min<long>((rows + 63) / 64, 2048);
    hipLaunchKernelGGL(fused::embed_points_kernel, dim3(grid), dim3(256), 0, c.st, ex);
    fused::EmbedArgs ey = ex;
    ey.x = r->point_y; ey.K = m->dim_y; ey.w1 = m->y_w1; ey.b1 = m->y_b1; ey.b2 = m->y_b2;
    ey.w2img = side + (size_t)(m->C + 1) * fused::SIDE_FRAGS; ey.out = c.at(c.pl.Ey);
    hipLaunchKernelGGL(fused::embed_points_kernel, dim3(grid), dim3(256), 0, c.st, ey);
    CHECK_LAUNCH();
  }
  fused::RolloutArgs a{};
  a.B = r->B; a.P = r->P; a.n_ctx0 = r->n_ctx0; a.n_th = m->n_theta; a.T = r->T; a.L = m->L;
  a.wpack = c.at(c.pl.Wpack); a.Ex = c.at(c.pl.Ex); a.Ey = c.at(c.pl.Ey);
  a.theta_tokens = m->theta_tokens; a.tmask = r->target_mask;
  a.mode = r->select_mode; a.uniform = r->uniform; a.forced = r->forced_idx;
  a.role = r->role; a.idx = r->idx; a.slot = r->slot; a.log_prob = r->log_prob;
  a.zt = r->zt;
  a.ztg = c.at(c.pl.Ztg);
  // ALINE_FUSED_STAMPS=1 selects the diagnostic (s_memtime-stamped) instantiation; the stamps land
  // in the tail of the workspace scalar block and are never read by product code.
  if (r->ev_kernel_start) (void)hipEventRecord(static_cast<hipEvent_t>(r->ev_kernel_start), c.st);
  if (dbg(ALINE_DBG_FUSED_STAMPS)) {
    a.stamps = reinterpret_cast<unsigned long long *>(c.at(c.pl.Stamps));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fused::rollout_f32_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused::LDS_BYTES);
    hipLaunchKernelGGL(fused::rollout_f32_kernel<true>, dim3((r->B + fused::EPW - 1) / fused::EPW), dim3(fused::NTHREADS), fused::LDS_BYTES, c.st, a);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fused::rollout_f32_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused::LDS_BYTES);
    hipLaunchKernelGGL(fused::rollout_f32_kernel<false>, dim3((r->B + fused::EPW - 1) / fused::EPW), dim3(fused::NTHREADS), fused::LDS_BYTES, c.st, a);
  }
  CHECK_LAUNCH();
  if (r->ev_kernel_stop) (void)hipEventRecord(static_cast<hipEvent_t>(r->ev_kernel_stop), c.st);
  // GMM posterior + compute_ll of all T steps on the saved target-row encodings [T*B*n_t, d]
  if (r->post_mean || r->post_std || r->post_weight || r->target_ll) {
    fused::GmmRowsArgs ga{};
    ga.z = c.at(c.pl.Ztg); ga.rows = (long)r->T * r->B * m->n_theta; ga.C = m->C; ga.std_min = m->std_min;
    ga.w1img = side;
    for (int k = 0; k < m->C; ++k) { ga.b1[k] = m->gmm_b1[k]; ga.w2[k] = m->gmm_w2[k]; ga.b2[k] = m->gmm_b2[k]; }
    ga.mean = r->post_mean; ga.sd = r->post_std; ga.wgt = r->post_weight;
    ga.value = r->target_all; ga.value_row0 = 0; ga.value_mod = (long)r->B * m->n_theta;
    ga.ll = r->target_all ? r->target_ll : nullptr;
    hipLaunchKernelGGL(fused::gmm_rows_kernel, dim3((unsigned)((ga.rows + fused::GMM_THREADS * fused::GMM_TILES / 4 - 1) / (fused::GMM_THREADS * fused::GMM_TILES / 4))), dim3(fused::GMM_THREADS), 0, c.st, ga);
    CHECK_LAUNCH();
  }
  return ALINE_OK;
}



extern "C++" {
#define X3_NS x3
#include "x3_host.h"
#undef X3_NS
#define X3_NS x5
#include "x3_host.h"
#undef X3_NS
}

// The s3 path (s3.h): d = 32 / 4 heads at reference precision, any embedding mode, up to 160 keys per episode -- one
// launch per design step, a workgroup owns two whole episodes.
static bool s3_eligible(const aline_model &m, const aline_rollout &r) {
  if (dbg(ALINE_DBG_DISABLE_S3)) return false;
  if (m.precision != ALINE_PREC_F16X3 || m.d != s3::D || m.H != s3::H || m.F % 32 || m.F > s3::F_MAX) return false;
  if (r.n_ctx0 < 1 || r.n_ctx0 + r.T - 1 + r.n_target_data + m.n_theta > s3::NK_MAX) return false;
  return true;
}

static int rollout_s3(const aline_model *m, const aline_rollout *r, void *ws, size_t ws_bytes, void *stream) {
  Ctx c;
  TRY(rollout_ctx(m, r, ws, ws_bytes, stream, c));
  TRY(check_select(r->select_mode, r->uniform, r->forced_idx));
  TRY(c.clear_flag());
  const int n_t = c.g.n_td + c.g.n_th, N = c.g.N, F = m->F, tpe = (N + 15) / 16, NP = 16 * tpe;
  const long tiles = (long)r->B * tpe;
  using s3::u32x4;
  hipLaunchKernelGGL(role_init_kernel, grid1d((size_t)r->B * r->P), dim3(256), 0, c.st, r->role, r->B, r->P, r->n_ctx0);
  CHECK_LAUNCH();
  s3::PackArgs pa{};
  pa.L = m->L; pa.F = F; pa.C = m->C;
  for (int l = 0; l < m->L; ++l) {
    pa.in_proj_w[l] = m->in_proj_w[l]; pa.in_proj_b[l] = m->in_proj_b[l];
    pa.out_proj_w[l] = m->out_proj_w[l]; pa.out_proj_b[l] = m->out_proj_b[l];
    pa.lin1_w[l] = m->lin1_w[l]; pa.lin1_b[l] = m->lin1_b[l];
    pa.lin2_w[l] = m->lin2_w[l]; pa.lin2_b[l] = m->lin2_b[l];
    pa.n1w[l] = m->norm1_w[l]; pa.n1b[l] = m->norm1_b[l];
    pa.n2w[l] = m->norm2_w[l]; pa.n2b[l] = m->norm2_b[l];
  }
  pa.acq_w1 = m->acq_w1; pa.acq_b1 = m->acq_b1; pa.acq_w2 = m->acq_w2; pa.acq_b2 = m->acq_b2;
  for (int k = 0; k < m->C; ++k) { pa.gmm_w1[k] = m->gmm_w1[k]; pa.gmm_b1[k] = m->gmm_b1[k]; pa.gmm_w2[k] = m->gmm_w2[k]; pa.gmm_b2[k] = m->gmm_b2[k]; }
  unsigned *img = reinterpret_cast<unsigned *>(c.at(c.pl.sImg));
  pa.out = img; pa.range_flag = c.flag(); pa.time_token = m->time_token ? 1 : 0;
  hipLaunchKernelGGL(s3::pack_kernel, dim3(256), dim3(256), 0, c.st, pa);
  CHECK_LAUNCH();
  {   // step-invariant point embeddings (fp32 rows): x-embedder on the points + target-data rows, y-embedder on the points
    s3::EmbArgs ex{};
    ex.src = Src3{{r->point_x, r->target_x, nullptr}, {r->P, r->n_target_data, 0}};
    ex.rows_per_ep = r->P + r->n_target_data; ex.B = r->B; ex.K = m->dim_x;
    ex.w1 = m->x_w1; ex.b1 = m->x_b1; ex.w2 = m->x_w2; ex.b2 = m->x_b2; ex.E = c.at(c.pl.Ex); ex.range_flag = c.flag();
    s3::EmbArgs ey{};
    ey.src = Src3{{r->point_y, nullptr, nullptr}, {r->P, 0, 0}};
    ey.rows_per_ep = r->P; ey.B = r->B; ey.K = m->dim_y;
    ey.w1 = m->y_w1; ey.b1 = m->y_b1; ey.w2 = m->y_w2; ey.b2 = m->y_b2; ey.E = c.at(c.pl.Ey); ey.range_flag = c.flag();
    if (dbg(ALINE_DBG_S3_GENERIC_EMBED)) {       // (A/B: the generic hidden-layer kernel + GEMM pair)
      TRY(do_embed_points(c, ex.src, r->point_y, r->P));
    } else {
      TRY(launch_s3_embed(c, F, ex, ey));
    }
  }
  u32x4 *X0 = reinterpret_cast<u32x4 *>(c.at(c.pl.sX0)), *XW = reinterpret_cast<u32x4 *>(c.at(c.pl.sXW));
  u32x4 *Zimg = reinterpret_cast<u32x4 *>(c.at(c.pl.sZimg));
  float *logits = c.at(c.pl.sLog);
  s3::AsmArgs aa{};
  aa.g = c.g; aa.tpe = tpe; aa.Ex = c.at(c.pl.Ex); aa.Ey = c.at(c.pl.Ey); aa.ey_rows = r->P; aa.theta_tokens = m->theta_tokens; aa.X = X0; aa.range_flag = c.flag();
  hipLaunchKernelGGL(s3::assemble_kernel, grid1d((size_t)tiles * 64), dim3(256), 0, c.st, aa);
  CHECK_LAUNCH();
  const bool want_gmm = r->post_mean || r->post_std || r->post_weight || r->target_ll;
  const S3Shape sh = s3_shape(*m, *r);
  u32x4 *Zq = wants_postq(*r) ? reinterpret_cast<u32x4 *>(c.at(c.pl.sZq)) : nullptr;   // candidate rows of all steps
  const int zw = r->P - r->n_ctx0;
  // in-kernel selection: the T argument sets live where the logits of the launch of its own would (they stay in LDS)
  s3::SelArgs *selp = nullptr;
  if (sh.sel && (size_t)r->T * sizeof(s3::SelArgs) <= (size_t)r->B * tpe * 16 * sizeof(float)) {
    selp = reinterpret_cast<s3::SelArgs *>(logits);
    s3::SelArgs q{};
    q.on = 1; q.mode = r->select_mode; q.uniform = r->uniform;
    q.forced = r->forced_idx; q.forced_stride = r->T;
    q.idx = r->idx; q.idx_stride = r->T; q.slot = r->slot; q.slot_stride = r->T; q.log_prob = r->log_prob; q.lp_stride = r->T;
    q.zt = r->zt; q.zt_stride = zw; q.zt_width = zw; q.role_out = r->role; q.range_flag = c.flag();
    hipLaunchKernelGGL(s3::sel_args_kernel, dim3((r->T + 63) / 64), dim3(64), 0, c.st, q, (int)r->T, (long)r->B, selp);
    CHECK_LAUNCH();
  }
  for (int t = 0; t < r->T; ++t) {
    c.g.n_ctx = r->n_ctx0 + t;
    s3::StepArgs sa{};
    sa.g = c.g; sa.tpe = tpe; sa.L = m->L; sa.F = F; sa.order = t > 0 ? r->n_ctx0 + t : 0;
    sa.epw = sh.epw; sa.nk2 = 2 * std::min(sh.nkp, (r->n_ctx0 + t + n_t + 31) / 32);
    sa.img = img; sa.X0 = X0; sa.XW = XW; sa.emb = aa; sa.emb.g = c.g;
    sa.logits = logits; sa.NP = NP;
    sa.zimg = want_gmm ? Zimg : nullptr; sa.zrow0 = (long)t * r->B * n_t;
    sa.zq = Zq; sa.zq_row0 = (long)t * r->B * r->P;
    sa.sv = r->saved_acts; sa.sv_rows = (long)r->T * r->B * N; sa.sv_row0 = (long)t * r->B * N;
    sa.tau = m->time_token ? step_time_token(*r, t) : 0.f;
#ifdef S3_STAMPS
    sa.stamps = want_gmm ? nullptr : reinterpret_cast<unsigned long long *>(c.at(c.pl.sRaw));
#endif
    const bool timed = t == (r->ev_kernel_step > 0 ? r->ev_kernel_step - 1 : r->T - 1);      // bench.py times this launch of the dominant kernel
    if (timed && r->ev_kernel_start) (void)hipEventRecord(static_cast<hipEvent_t>(r->ev_kernel_start), c.st);
    sa.selp = selp ? selp + t : nullptr;
    TRY(launch_s3_step(c, sh, sa));
    if (timed && r->ev_kernel_stop) (void)hipEventRecord(static_cast<hipEvent_t>(r->ev_kernel_stop), c.st);
    CHECK_LAUNCH();
    if (selp) continue;
    SelectArgs sel{};
    sel.g = c.g; sel.F = F; sel.logits = logits; sel.logit_stride = NP;
    sel.mode = r->select_mode;
    sel.uniform = r->uniform ? r->uniform + (size_t)t * r->B : nullptr;
    sel.forced = r->forced_idx ? r->forced_idx + t : nullptr; sel.forced_stride = r->T;
    sel.idx = r->idx ? r->idx + t : nullptr; sel.idx_stride = r->T;
    sel.slot = r->slot ? r->slot + t : nullptr; sel.slot_stride = r->T;
    sel.log_prob = r->log_prob ? r->log_prob + t : nullptr; sel.lp_stride = r->T;
    sel.zt = r->zt ? r->zt + (size_t)t * r->B * zw : nullptr; sel.zt_stride = zw; sel.zt_width = zw;
    sel.role_out = r->role;
    TRY(launch_acq_select(c, sel));
    CHECK_LAUNCH();
  }
  if (want_gmm) {   // GMM heads of all T * B * n_t target rows, then the parameter maps + mixture log-likelihood
    const long per_step = (long)r->B * n_t, total = per_step * r->T;
    s3::GmmArgs ga{};
    ga.Z = Zimg; ga.ntiles = (total + 15) / 16; ga.M = total; ga.C = m->C; ga.std_min = m->std_min; ga.range_flag = c.flag();
    ga.img = img + ((long)m->L * s3::layer_bytes(F) + s3::head_bytes(F)) / 4;
    ga.mean = r->post_mean; ga.sd = r->post_std; ga.wgt = r->post_weight;
    ga.value = r->target_all; ga.value_mod = per_step; ga.ll = r->target_ll;
    TRY(launch_s3_gmm(c, F, ga));
  }
  if (Zq) {         // posterior_out_query (model/head.py:366): the same heads on the candidate rows of all steps
    const long total = (long)r->T * r->B * r->P;
    s3::GmmArgs ga{};
    ga.Z = Zq; ga.ntiles = (total + 15) / 16; ga.M = total; ga.C = m->C; ga.std_min = m->std_min;
    ga.img = img + ((long)m->L * s3::layer_bytes(F) + s3::head_bytes(F)) / 4;
    ga.mean = r->postq_mean; ga.sd = r->postq_std; ga.wgt = r->postq_weight;
    TRY(launch_s3_gmm(c, F, ga));
  }
  return ALINE_OK;
}

int aline_rollout_path(const aline_model *m, const aline_rollout *r) {
  if (!m || !r) return ALINE_EINVAL;
  TRY(validate_model(*m));
  if (fused_eligible(*m, *r)) return ALINE_PATH_FUSED;
  if (x3::eligible(*m, *r)) return ALINE_PATH_X3;
  if (x5::eligible(*m, *r)) return ALINE_PATH_X5;
  if (s3_eligible(*m, *r)) return ALINE_PATH_S3;
  return ALINE_PATH_GENERIC;
}

// aline_rollout.saved_acts: written by the s3 path, read by the backward of the model width whose tail backward needs nothing but
// a layer's input and attention output (tail_bwd.h)
static bool saved_acts_usable(const aline_model &m, const aline_rollout &r) {
  const bool fused_tail = m.d == tailbwd::D && m.F == tailbwd::F && !dbg(ALINE_DBG_NO_BWD_TAIL);      // (= bwd_fused_tail below)
  return r.T > 0 && fused_tail && aline_rollout_path(&m, &r) == ALINE_PATH_S3;
}
size_t aline_rollout_saved_acts_bytes(const aline_model *m, const aline_rollout *r) {
  if (!m || !r || validate_model(*m) != 0 || !saved_acts_usable(*m, *r)) return 0;
  const size_t N = (size_t)r->P + r->n_target_data + m->n_theta;
  return (size_t)(2 * m->L + 1) * r->T * r->B * N * m->d * sizeof(float);
}

int aline_rollout_kernel_name(const aline_model *m, const aline_rollout *r, char *buf, size_t n) {
  if (!buf || n == 0) return ALINE_EINVAL;
  const int path = aline_rollout_path(m, r);
  if (path < 0) return path;
  switch (path) {
    case ALINE_PATH_FUSED: snprintf(buf, n, "fused::rollout_f32_kernel<%s>", dbg(ALINE_DBG_FUSED_STAMPS) ? "true" : "false"); break;
    case ALINE_PATH_X3: snprintf(buf, n, "x3::layer_kernel<true>"); break;      // (the launch the event pair brackets: last layer of the last step)
    case ALINE_PATH_X5: snprintf(buf, n, "x5::layer_kernel<true>"); break;
    case ALINE_PATH_S3: {
      const S3Shape sh = s3_shape(*m, *r);
      snprintf(buf, n, "s3::step_kernel<%d, %d, %d, false, %s>", m->F, sh.nw, sh.nw == 8 ? s3::NKP_MAX : 2,
               (r->saved_acts && m->F == 128) ? "true" : "false");
      break;
    }
    default: snprintf(buf, n, "generic pipeline (no dominant kernel)"); break;
  }
  return path;
}

int aline_rollout_forward(const aline_model *m, const aline_rollout *r, void *ws, size_t ws_bytes,
                          void *stream) {
  switch (aline_rollout_path(m, r)) {
    case ALINE_PATH_FUSED: return rollout_fused(m, r, ws, ws_bytes, stream);
    case ALINE_PATH_X3: return x3::rollout(m, r, ws, ws_bytes, stream);
    case ALINE_PATH_X5: return x5::rollout(m, r, ws, ws_bytes, stream);
    case ALINE_PATH_S3: return rollout_s3(m, r, ws, ws_bytes, stream);
    default: break;     /* generic pipeline (its entry points report invalid arguments) */
  }
  TRY(aline_rollout_init(m, r, ws, ws_bytes, stream));
  for (int t = 0; t < r->T; ++t) TRY(aline_rollout_step(m, r, t, ws, ws_bytes, stream));
  return ALINE_OK;
}

int aline_rollout_export(const aline_rollout *r, int n_ctx, float *context_x, float *context_y,
                         float *query_x, float *query_y, int dim_x, int dim_y, void *stream) {
  if (!r || !r->role || !context_x || !context_y || n_ctx < 1 || n_ctx > r->P) return ALINE_EINVAL;
  hipLaunchKernelGGL(rollout_export_kernel, dim3(r->B), dim3(256), 0, static_cast<hipStream_t>(stream),
                     r->role, r->point_x, r->point_y, r->B, r->P, n_ctx, dim_x, dim_y, context_x,
                     context_y, query_x, query_y);
  CHECK_LAUNCH();
  return ALINE_OK;
}

// ---- objectives ---------------------------------------------------------------------------------
int aline_compute_ll(const float *value, const float *means, const float *stds, const float *weights,
                     int64_t rows, int C, float *out, void *stream) {
  if (!value || !means || !stds || !weights || !out || rows <= 0 || C <= 0) return ALINE_EINVAL;
  hipLaunchKernelGGL(compute_ll_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), value, means, stds, weights, (long)rows, C, out);
  CHECK_LAUNCH();
  return ALINE_OK;
}

int aline_eig_location_step(const float *theta, const float *xi, const float *y, float *S, int64_t L1,
                            int B, int K, int D, float noise_scale, float base_signal,
                            float max_signal, void *stream) {
  if (!theta || !xi || !y || !S || L1 < 2 || B <= 0 || K <= 0 || D <= 0 || D > 8) return ALINE_EINVAL;
  const size_t total = (size_t)L1 * B;
  unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(eig_location_step_kernel, dim3(blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), theta, xi, y, S, (long)L1, B, K, D, noise_scale,
                     base_signal, max_signal);
  CHECK_LAUNCH();
  return ALINE_OK;
}

int aline_eig_ces_step(const float *theta, const float *xi, const float *y, float *S, int64_t L1, int B,
                       float noise_scale, float epsilon, int32_t *nan_flag, void *stream) {
  if (!theta || !xi || !y || !S || L1 < 2 || B <= 0) return ALINE_EINVAL;
  const size_t total = (size_t)L1 * B;
  unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 16);
  const size_t tab_bytes = (size_t)B * CES_ROW * sizeof(float);
  if (tab_bytes <= 48 * 1024 && !dbg(ALINE_DBG_CES_GENERIC))
    hipLaunchKernelGGL(eig_ces_step_table_kernel, dim3(blocks), dim3(256), tab_bytes, static_cast<hipStream_t>(stream),
                       theta, xi, y, S, (long)L1, B, noise_scale, epsilon, nan_flag);
  else
    hipLaunchKernelGGL(eig_ces_step_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       theta, xi, y, S, (long)L1, B, noise_scale, epsilon, nan_flag);
  CHECK_LAUNCH();
  return ALINE_OK;
}

// rows of S per partial-logsumexp workgroup: about 2048 chunks over L (enough workgroups to fill the chip at the
// small outer batches of the evaluation protocol), 256 .. 4096 rows
static long eig_chunk_rows(int64_t L1) {
  const long want = ((L1 - 1 + 2047) / 2048 + 63) / 64 * 64;
  return std::min<long>(4096, std::max<long>(256, want));
}

size_t aline_eig_finalize_workspace_bytes(int64_t L1, int B) {
  const long ch = eig_chunk_rows(L1), nchunk = (L1 - 1 + ch - 1) / ch;
  return (size_t)std::max<long>(nchunk, 1) * B * 2 * sizeof(float);
}

int aline_eig_finalize(const float *S, int64_t L1, int B, float *pce, float *nmc, void *ws,
                       size_t ws_bytes, void *stream) {
  if (!S || L1 < 2 || B <= 0 || !ws) return ALINE_EINVAL;
  if (ws_bytes < aline_eig_finalize_workspace_bytes(L1, B)) return ALINE_EWORKSPACE;
  const long kEigChunk = eig_chunk_rows(L1);
  const int nchunk = (int)((L1 - 1 + kEigChunk - 1) / kEigChunk);
  float *part = static_cast<float *>(ws);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(eig_lse_partial_kernel, dim3(nchunk, (B + 255) / 256), dim3(256), 0, st, S, (long)L1,
                     B, kEigChunk, part);
  CHECK_LAUNCH();
  hipLaunchKernelGGL(eig_lse_combine_kernel, dim3(B), dim3(256), 0, st, S, part, nchunk,
                     (long)L1, B, pce, nmc);
  CHECK_LAUNCH();
  return ALINE_OK;
}

// All T steps of a design history in one pass over theta (eig.h: eig_location_history_kernel).  xi [B, T, D], y [B, T] (the history in
// order of acquisition), theta [L1, B, K, D] with row 0 the true parameter; pce / nmc [B, T]: the stepwise bounds of utils/eval.py:64-78.
// row groups of the history kernel: about 12 waves per CU worth of threads (three per SIMD), each with at least four rows
static long eig_history_groups(int64_t L1, int B) {
  const long want = (256l * 12 * 64 + B - 1) / B;
  return std::max<long>(1, std::min<long>(want, (L1 - 1 + 3) / 4));
}
size_t aline_eig_history_workspace_bytes(int64_t L1, int B, int T) {
  if (L1 < 2 || B <= 0 || T <= 0) return 0;
  return ((size_t)eig_history_groups(L1, B) * T * B * 2 + (size_t)T * B) * sizeof(float);
}

int aline_eig_location_history(const float *theta, const float *xi, const float *y, int64_t L1, int B, int T, int K, int D,
                               float noise_scale, float base_signal, float max_signal, float *pce, float *nmc, void *ws,
                               size_t ws_bytes, void *stream) {
  if (!theta || !xi || !y || L1 < 2 || B <= 0 || T <= 0 || K <= 0 || D <= 0 || K * D > 8 || !ws) return ALINE_EINVAL;
  if (ws_bytes < aline_eig_history_workspace_bytes(L1, B, T)) return ALINE_EWORKSPACE;
  const long R = eig_history_groups(L1, B);
  const size_t smem = 0;
  float *part = static_cast<float *>(ws), *s0 = part + (size_t)R * T * B * 2;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)(((long)B * R + 255) / 256));
  if (K == 1 && D == 2)
    hipLaunchKernelGGL(eig_location_history_kernel<1>, grid, dim3(256), smem, st, theta, xi, y, (long)L1, B, T, K, D, noise_scale,
                       base_signal, max_signal, R, part, s0);
  else
    hipLaunchKernelGGL(eig_location_history_kernel<0>, grid, dim3(256), smem, st, theta, xi, y, (long)L1, B, T, K, D, noise_scale,
                       base_signal, max_signal, R, part, s0);
  CHECK_LAUNCH();
  hipLaunchKernelGGL(eig_history_combine_kernel, grid1d((size_t)T * B * 64), dim3(256), 0, st, part, s0, (int)R, (long)L1, B, T, pce, nmc);
  CHECK_LAUNCH();
  return ALINE_OK;
}

// CES histories (tasks/ces.py:96-115): thetas [L1, B, 5], xi [B, T, 6], y [B, T] -> pce / nmc [B, T].  ALINE_EUNSUPPORTED when the per-(episode,
// step) table does not fit LDS or T > 16: the caller keeps the step kernels.
int aline_eig_ces_history(const float *theta, const float *xi, const float *y, int64_t L1, int B, int T, float noise_scale, float epsilon,
                          float *pce, float *nmc, int32_t *nan_flag, void *ws, size_t ws_bytes, void *stream) {
  if (!theta || !xi || !y || L1 < 2 || B <= 0 || T <= 0 || !ws) return ALINE_EINVAL;
  const size_t tab_only = ((size_t)B * ((T * (CES_ROW + 1)) | 1) + 2) * sizeof(float);      // padded pitches (eig.h)
  const size_t tab_bytes = tab_only + (size_t)T * 256 * 2 * sizeof(float);      // + the (max, sum-exp) pairs of the workgroup's threads
  if (T > 16 || tab_only > 64 * 1024) return ALINE_EUNSUPPORTED;
  if (ws_bytes < aline_eig_history_workspace_bytes(L1, B, T)) return ALINE_EWORKSPACE;
  const long R = eig_history_groups(L1, B);
  float *part = static_cast<float *>(ws), *s0 = part + (size_t)R * T * B * 2;
  hipStream_t st = static_cast<hipStream_t>(stream);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&eig_ces_history_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tab_bytes);
  hipLaunchKernelGGL(eig_ces_history_kernel, dim3((unsigned)(((long)B * R + 255) / 256)), dim3(256), tab_bytes, st, theta, xi, y, (long)L1, B, T,
                     noise_scale, epsilon, R, part, s0, nan_flag);
  CHECK_LAUNCH();
  hipLaunchKernelGGL(eig_history_combine_kernel, grid1d((size_t)T * B * 64), dim3(256), 0, st, part, s0, (int)R, (long)L1, B, T, pce, nmc);
  CHECK_LAUNCH();
  return ALINE_OK;
}

}  // extern "C"

// Diagnostic only: byte offset of the stamp block inside a rollout workspace (tools/stamps.py).
extern "C" size_t aline_debug_stamps_offset(const aline_model *m, const aline_rollout *r) {
  if (!m || !r) return 0;
  return make_plan(*m, r->B, r->P, r->n_target_data, r->P, false, r->T).Stamps * sizeof(float);
}

// Diagnostic only: byte offsets of the buffers the stamped diagnostic builds (X3_STAMPS / S3_STAMPS, tools/*_stamps.py) and
// read back; declared in include/aline_hip.h.
extern "C" size_t aline_debug_xraw_offset(const aline_model *m, const aline_rollout *r) {
  if (!m || !r) return 0;
  const Plan pl = make_plan(*m, r->B, r->P, r->n_target_data, r->P, false, r->T);
  return (m->d == s3::D ? pl.sRaw : pl.xRaw) * sizeof(float);
}

extern "C" uint32_t aline_debug_set_flags(uint32_t flags) { return g_debug_flags.exchange(flags); }
extern "C" uint32_t aline_debug_get_flags(void) { return g_debug_flags.load(); }
extern "C" int aline_debug_set_param(int key, int value) {
  if (key < 0 || key >= ALINE_DBG_NPARAMS) return ALINE_EINVAL;
  g_debug_params[key].store(value);
  return ALINE_OK;
}
extern "C" int aline_debug_get_param(int key) {
  if (key < 0 || key >= ALINE_DBG_NPARAMS) return 0;
  return g_debug_params[key].load();
}

// f16 range guard (include/aline_hip.h): the status word is the first word of every workspace plan
extern "C" size_t aline_f16_range_offset(void) { return 0; }
extern "C" int aline_f16_range_status(const void *ws, void *stream) {
  if (!ws) return ALINE_EINVAL;
  unsigned word = 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemcpyAsync(&word, ws, sizeof(word), hipMemcpyDeviceToHost, st) != hipSuccess) return ALINE_ELAUNCH;
  if (hipStreamSynchronize(st) != hipSuccess) return ALINE_ELAUNCH;
  return (int)(word & 0xFFu);
}

// ================================= backward (training) ============================================
// Arithmetic of the NT GEMMs of the backward pass (forward recompute and dX products): exact-fp32 MFMA.  ALINE_DBG_BWD_PREC = 3
// (3-term f16 split) was measured and is NOT used: upstream gradients are ~1e-8 .. 1e-5 (1 / (T B n_t) scaling), below
// f16's normal range, so the split loses them (tests/test_backward_gpu.py fails), and the step is bound by HBM traffic of
// the saved activations, not by the matrix pipe (67.7 -> 66.9 ms).
// The forward RECOMPUTE products of the per-op backward (activations x weights, exactly the products of the forward pass) take the
// forward's own arithmetic when that is the 3-term f16 split and the model is wide enough to be bound by its GEMMs (d >= 64: the
// small-width model has its fused exact-fp32 kernels and per-op A/B tests against them): 3 passes on the f16 pipe instead of the
// fp32 MFMA -- the recompute GEMMs were 40 % of the d = 256 training step.  The gradient products (dX, dW) stay exact fp32.
static int recompute_prec(const aline_model &m);
static int bwd_prec() {
  const int p = dbg_param(ALINE_DBG_BWD_PREC);
  return (p < 0 || p > 3) ? ALINE_PREC_F32 : p;
}

static int recompute_prec(const aline_model &m) {
  return (m.precision == ALINE_PREC_F16X3 && m.d >= 64 && !dbg(ALINE_DBG_BWD_RECOMPUTE_F32)) ? ALINE_PREC_F16X3 : bwd_prec();
}
// Round 4: the GRADIENT products of the per-op backward (dX = dY W, dW = dY^T X) of an F16X3 model at d >= 64 run the 3-term f16
// split too, with the gradient operand scaled by a per-tensor power of two (max |dY| -> [2^14, 2^15); gemm.h: xmax_bits) -- unscaled,
// gradients of 1e-8 .. 1e-5 sit in f16's subnormals (round 3 kept these products on the 157 TFLOP/s fp32 pipe for that reason).
// ALINE_DBG_BWD_GRAD_F32: exact fp32 as before (A/B runs, parity tests).
static bool bwd_grad_f16(const aline_model &m) {
  return m.precision == ALINE_PREC_F16X3 && m.d >= 64 && !dbg(ALINE_DBG_BWD_GRAD_F32) && dbg_param(ALINE_DBG_BWD_PREC) == 0;
}

namespace {

struct BwdPlan {
  size_t Xs, QKV, A, U1, X1, Hid, U2, HidA, HidG, dXa, dXb, dQKV, dHid, dTmp, Ex, Ey, EHx, EHy, dEx, dEy, Wt, KeyIdx, Kcnt, Tvec, Scale,
      total;
  int rc_width;                                   // 0, or the width (256 / 512) whose tile-image layer kernel recomputes the forward (x3_host.h: recompute)
  x3::RecomputePlan rc3; x5::RecomputePlan rc5;
};
// Forward recompute of the per-op backward on the rollout's own layer kernel (x3_host.h).  ALINE_DBG_NO_BWD_IMAGE_RECOMPUTE: the generic kernels.
static int image_recompute_width(const aline_model &m) {
  if (dbg(ALINE_DBG_NO_BWD_IMAGE_RECOMPUTE) || dbg(ALINE_DBG_BWD_RECOMPUTE_F32)) return 0;
  return x3::recompute_model_ok(m) ? x3::D : x5::recompute_model_ok(m) ? x5::D : 0;
}

// Fused token-local tail (tail_bwd.h) for the small-width model: U1 / X1 / Hid / U2 are never stored.  ALINE_BWD_TAIL=0
// switches back to the per-op pipeline (A/B measurements).
// (the switches are read at every call: tests compare both pipelines in one process)
static bool bwd_fused_tail(const aline_model &m) { return m.d == tailbwd::D && m.F == tailbwd::F && !dbg(ALINE_DBG_NO_BWD_TAIL); }

// Acquisition head backward without the [I P, F] hidden activations (acq_head_bwd.h).  ALINE_BWD_ACQ=0: the per-op kernels.
static bool fused_acq_head(const aline_model &m) { return m.d == acqb::D && m.F == acqb::F && !m.time_token && !dbg(ALINE_DBG_NO_BWD_ACQ); }

static bool fused_gmm_heads(const aline_model &m) { return m.d == gmmb::D && m.F == gmmb::F && m.C <= 16 && !dbg(ALINE_DBG_NO_BWD_GMM_FUSED); }

// In-projection + attention backward as one kernel (attn_bwd_mfma.h).  ALINE_BWD_ATTN_BLOCK=0: the per-op kernels.
static bool fused_attn_block(const aline_model &m, int max_keys) {
  return m.d == abwd::D && m.H == abwd::H && max_keys <= abwd::MAXK && !dbg(ALINE_DBG_NO_BWD_ATTN_BLOCK);
}

BwdPlan make_bwd_plan(const aline_model &m, int B, int P, int n_td, int tc) {
  BwdPlan p{};
  const bool ft = bwd_fused_tail(m);
  const size_t N = (size_t)P + n_td + m.n_theta, I = (size_t)B * tc, M = I * N, d = m.d, F = m.F, L = m.L;
  const size_t n_t = (size_t)n_td + m.n_theta, rows_x = (size_t)B * (P + n_td), rows_y = (size_t)B * P;
  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off += align_up(n); return o; };
  p.Xs = take((L + 1) * M * d);
  p.QKV = take(std::max(L * M * 3 * d, (2 * L + 3) * M * d));      // or: L + 1 compact K / V buffers (<= 2 M d each) + Q
  p.A = take(L * M * d);
  p.U1 = take(ft ? 0 : L * M * d);
  p.X1 = take(ft ? 0 : L * M * d);
  p.Hid = take(ft ? 0 : L * M * F);
  p.U2 = take(ft ? 0 : L * M * d);
  p.HidA = take(fused_acq_head(m) ? 0 : I * P * F);
  p.Tvec = take(I);
  // (fused GMM kernels: raw / draw [rows, C, 4] each + dz per component [C, rows, 32] instead of the hidden units [rows, C F])
  p.HidG = take(I * n_t * m.C * (fused_gmm_heads(m) ? 40 : F));
  p.dXa = take(M * d);
  p.dXb = take(M * d);
  p.dQKV = take(M * 3 * d);
  p.dHid = take(std::max(ft ? (size_t)0 : M * F, rows_x * F));
  p.dTmp = take(M * d);
  p.Ex = take(rows_x * d);
  p.Ey = take(rows_y * d);
  p.EHx = take(rows_x * F);
  p.EHy = take(rows_y * F);
  p.dEx = take(rows_x * d);
  p.dEy = take(rows_y * d);
  p.Wt = take(std::max({(size_t)3 * d * d, F * d, (size_t)m.C * F * d}));
  p.KeyIdx = take(M);      // (ints) key rows of every instance: K / V of the forward recompute on these rows only
  p.Kcnt = take(I * 2);
  p.Scale = take(16 * 16);      // (unsigned) ring of 16 max-|dY| words (one per 16-word line) of the scaled f16 gradient products
  p.rc_width = image_recompute_width(m);
  if (p.rc_width == x3::D) p.rc3 = x3::recompute_plan(m, (long)I, (int)N, take);
  else if (p.rc_width == x5::D) p.rc5 = x5::recompute_plan(m, (long)I, (int)N, take);
  p.total = off;
  return p;
}

struct BCtx {
  const aline_model *m;
  Geo g;
  BwdPlan pl;
  float *ws;
  hipStream_t st;
  mutable int scale_next = 0;      // next word of the scale ring (new_scale_word): per call, so that concurrent calls on other streams do not advance it
  float *at(size_t off) const { return ws + off; }
};

int transpose_to(const BCtx &c, const float *W, int rows, int cols, float *dst, int ld = 0) {
  hipLaunchKernelGGL(transpose_kernel, grid1d((size_t)rows * cols), dim3(256), 0, c.st, W, rows, cols, ld > 0 ? ld : cols, dst);
  CHECK_LAUNCH();
  return ALINE_OK;
}

// Scale words of the F16X3 gradient products (gemm.h: xmax_bits): max |dY| of a gradient tensor, as bits.  A word is taken from a ring
// of 16 (cleared by a kernel node in stream order), filled either by the PRODUCER of the tensor -- LayerNorm backward and the GEMM
// epilogue reduce what they store (one atomicMax per wave) -- or by a reduction pass over the stored tensor (grad_absmax: 190 us per
// call at the d = 256 headline chunk, 11 % of the step when every product ran its own), and handed to the products that read the tensor.
// A word lives until the ring comes round: 16 producers later (every consumer follows its producer within three calls).
unsigned *new_scale_word(const BCtx &c) {
  unsigned *w = reinterpret_cast<unsigned *>(c.at(c.pl.Scale)) + 16 * (c.scale_next++ & 15);
  hipLaunchKernelGGL(clear_words_kernel, dim3(1), dim3(16), 0, c.st, w);
  return w;
}
unsigned *grad_absmax(const BCtx &c, const float *dY, long rows, int cols, int ld) {
  unsigned *w = new_scale_word(c);
  const long n4 = rows * (cols / 4);
  hipLaunchKernelGGL(absmax_bits_kernel, dim3((unsigned)std::min<long>((n4 + 255) / 256, 2048)), dim3(256), 0, c.st, dY, rows, cols, ld, w);
  return w;
}

// A gradient product Y = dY . Wt^T of the per-op backward: the scaled f16 split (bwd_grad_f16) or the exact-fp32 policy.
// scale: the word of dY (null: reduce dY here); out_max: a cleared word that receives max |Y| (Y is itself a gradient operand), or null
int launch_grad_gemm(const BCtx &c, GemmArgs a, const unsigned *scale = nullptr, unsigned *out_max = nullptr) {
  a.out_absmax = out_max;
  if (bwd_grad_f16(*c.m) && a.K % 4 == 0 && (!a.row_index || scale) && a.R_in == a.G_in) {      // (a gathered operand: its scale word comes with it)
    a.xmax_bits = scale ? scale : grad_absmax(c, a.X, a.M, a.K, a.ldx);
    a.range_flag = nullptr;
    return launch_gemm(ALINE_PREC_F16X3, a, 1, c.st);
  }
  return launch_gemm(bwd_prec(), a, 1, c.st);
}

// dX[M, K] (+)= dY[M, N] . W[N, K]   (W in PyTorch layout [N, K]); Wt scratch holds W^T [K, N]
int gemm_dx(const BCtx &c, const float *dY, int ldy, const float *W, int N, int K, float *dX, int ldx, int M,
            bool accum, const float *relu_of = nullptr, const unsigned *scale = nullptr, unsigned *out_max = nullptr) {
  float *Wt = c.at(c.pl.Wt);
  TRY(transpose_to(c, W, N, K, Wt));
  GemmArgs a = gemm_args(dY, ldy, Wt, nullptr, N, dX, ldx, M, K, N, false);
  a.accum = accum ? 1 : 0;
  a.mask = relu_of; a.ldmask = ldx;      // gradient through the ReLU whose output is `relu_of` [M, K]
  TRY(launch_grad_gemm(c, a, scale, out_max));
  CHECK_LAUNCH();
  return ALINE_OK;
}

// dW[N, K] += dY^T X, db[N] += colsum(dY)
int gemm_dw(const BCtx &c, const float *dY, int ldy, const float *X, int ldx, float *dW, float *db, long M,
            int N, int K, int Ry = 1, int Gy = 1, int offy = 0, int Rx = 1, int Gx = 1, int offx = 0, int ldw = 0,
            const unsigned *scale = nullptr, const int *row_index = nullptr) {
  if (N % 32 || K % 32) return ALINE_EUNSUPPORTED;
  if (row_index && !(scale && bwd_grad_f16(*c.m) && N % 256 == 0 && K % 256 == 0)) return ALINE_EUNSUPPORTED;      // (callers check: the f16 block kernel only)
  if (bwd_grad_f16(*c.m) && Ry == Gy && N % 256 == 0 && K % 256 == 0 && ldy % 4 == 0 && ldx % 4 == 0 && M < (1l << 40)) {
    // the scaled 3-term f16 split on 256 x 256 blocks (backward.h: gemm_tn_f16_kernel); X may be row-mapped (the point rows of every
    // instance: the acquisition head's first layer), dY is dense
    const float *pY = dY + (long)offy * ldy, *pX = X;
    if (!(reinterpret_cast<uintptr_t>(pY) & 15) && !(reinterpret_cast<uintptr_t>(pX) & 15)) {
      GemmTnF16Args f{};
      f.Rp = 1; f.Gp = 1; f.offp = 0; f.Rq = Rx; f.Gq = Gx; f.offq = offx;
      f.xmax_bits = scale ? scale : grad_absmax(c, pY, M, N, ldy);
      const long ldo = ldw > 0 ? ldw : K;
      f.P = pY; f.ldp = ldy; f.Q = pX; f.ldq = ldx; f.out = dW; f.sa = ldo; f.sb = 1; f.grad_is_p = 1;
      f.nba = N / 256; f.nbb = K / 256;
      f.colsum = db; f.M = M; f.row_index = row_index;
      // rows per workgroup: one workgroup per CU at a time, so the launch is whole rounds of n_cu workgroups -- the row chunk is sized
      // so that chunks x blocks fills r rounds (r the smallest that keeps a chunk <= 4096 rows: 400 workgroups of 4096 rows on 256 CUs
      // were 1.56 rounds that cost 2)
      {
        int dev = 0, n_cu = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        const long nblk = (long)f.nba * f.nbb;
        long r = 1;
        while ((M * nblk + n_cu * r - 1) / (n_cu * r) > 4096) ++r;
        long per = (M * nblk + n_cu * r - 1) / (n_cu * r);          // rows per workgroup for exactly r rounds
        per = std::max<long>(256, (per + 31) / 32 * 32);
        f.mchunk = per;
      }
      f.gx = (int)((M + f.mchunk - 1) / f.mchunk);
      const unsigned grid = (unsigned)((f.gx + 7) / 8 * 8) * (unsigned)(f.nba * f.nbb);
      constexpr int TN16_LDS = 2 * 2 * 512 * 32 * 2;      // two images x (hi | lo) x [512 columns][32 rows] f16 = 128 KB
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tn_f16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TN16_LDS);
      hipLaunchKernelGGL(gemm_tn_f16_kernel, dim3(grid), dim3(512), TN16_LDS, c.st, f);
      CHECK_LAUNCH();
      return ALINE_OK;
    }
  }
  if (row_index) return ALINE_EUNSUPPORTED;      // (only the f16 block kernel takes an index list; its conditions did not hold)
  GemmTnArgs a{};
  a.dY = dY; a.ldy = ldy; a.Ry = Ry; a.Gy = Gy; a.offy = offy;
  a.X = X; a.ldx = ldx; a.Rx = Rx; a.Gx = Gx; a.offx = offx;
  a.dW = dW; a.ldw = ldw > 0 ? ldw : K; a.db = db; a.M = M; a.N = N; a.K = K;
  a.mchunk = 4096;
  // short products (the GMM heads see n_t target rows per instance: 60 000 rows at the headline shape) would run on M / 4096
  // = 15 workgroups (0.21 ms each, ten of them per step): shrink the chunk until ~512 workgroups exist
  {
    const long blocks_nk = (long)std::max(1, std::max(N, K) / 128) * (std::min(N, K) / 32);
    while (a.mchunk > 256 && ((M + a.mchunk - 1) / a.mchunk) * blocks_nk < 512) a.mchunk /= 2;
  }
  // the block kernel reads its wide operand once per 32 columns of the other one: let the wider matrix be the wide one
  if (K > N && K % 64 == 0) {
    std::swap(a.dY, a.X); std::swap(a.ldy, a.ldx); std::swap(a.Ry, a.Rx); std::swap(a.Gy, a.Gx); std::swap(a.offy, a.offx);
    std::swap(a.N, a.K);
    a.swap = 1;
  }
  const int Nw = a.N, Kn = a.K;
  const unsigned gx = (unsigned)((M + a.mchunk - 1) / a.mchunk);
  const bool tk4 = Kn % 64 == 0 && !dbg(ALINE_DBG_BWD_DW_TK2);      // 64 columns of the narrow operand per workgroup
  a.walk = (a.Ry == a.Gy || a.Ry >= 64) && (a.Rx == a.Gx || a.Rx >= 64) && M < (1l << 30) && !dbg(ALINE_DBG_NO_BWD_DW_WALK);
#define TNB_LAUNCH(TN, TK, NBY, NBZ) do { a.gx = (int)gx; a.nby = (NBY); a.nbz = (NBZ);                                          \
    const dim3 grid((gx + 7) / 8 * 8 * (unsigned)((NBY) * (NBZ)));                                                            \
    hipLaunchKernelGGL((gemm_tn_block_kernel<TN, TK>), grid, dim3(256), 0, c.st, a); } while (0)
  if (Nw % 128 == 0 && tk4) TNB_LAUNCH(8, 4, Nw / 128, Kn / 64);
  else if (Nw % 128 == 0) TNB_LAUNCH(8, 2, Nw / 128, Kn / 32);
  else if (Nw % 96 == 0) TNB_LAUNCH(6, 2, Nw / 96, Kn / 32);
  else if (Nw % 64 == 0) TNB_LAUNCH(4, 2, Nw / 64, Kn / 32);
  else TNB_LAUNCH(2, 2, Nw / 32, Kn / 32);
#undef TNB_LAUNCH
  CHECK_LAUNCH();
  return ALINE_OK;
}

int ln_bwd(const BCtx &c, const float *dY, const float *U, const float *w, float *dU, float *dw, float *db,
           long rows, unsigned **out_max = nullptr) {
  const int d = c.m->d;
  if (out_max) *out_max = nullptr;
  if (d == 32 || d == 64) {
    const int rows_per_pass = d == 32 ? 32 : 16;
    const unsigned grid = (unsigned)std::min<long>((rows + rows_per_pass - 1) / rows_per_pass, 256 * 8);
    if (d == 32) hipLaunchKernelGGL(layernorm_bwd_narrow_kernel<8>, dim3(grid), dim3(256), 0, c.st, dY, U, w, dU, dw, db, rows);
    else hipLaunchKernelGGL(layernorm_bwd_narrow_kernel<16>, dim3(grid), dim3(256), 0, c.st, dY, U, w, dU, dw, db, rows);
    CHECK_LAUNCH();
    return ALINE_OK;
  }
  const int rpb = 64;
  unsigned *mw = (out_max && bwd_grad_f16(*c.m)) ? new_scale_word(c) : nullptr;      // max |dU| for the F16X3 products that read dU
  if (out_max) *out_max = mw;
  if (d == 256 || d == 512) {      // 16 B per lane (backward.h: layernorm_bwd_wide_kernel)
    const unsigned grid = (unsigned)std::min<long>((rows + 3) / 4, 256 * 8);
    if (d == 256) hipLaunchKernelGGL(layernorm_bwd_wide_kernel<1>, dim3(grid), dim3(256), 0, c.st, dY, U, w, dU, dw, db, rows, mw);
    else hipLaunchKernelGGL(layernorm_bwd_wide_kernel<2>, dim3(grid), dim3(256), 0, c.st, dY, U, w, dU, dw, db, rows, mw);
    CHECK_LAUNCH();
    return ALINE_OK;
  }
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256),
                     (size_t)2 * c.m->d * sizeof(float), c.st, dY, U, w, dU, dw, db, rows, c.m->d, rpb, mw);
  CHECK_LAUNCH();
  return ALINE_OK;
}

// head_dim 32 / 64 on the fp32 matrix pipe (attn_bwd_wide.h); *out_max (when asked for) receives the scale word of dQKV
template <int HD, int NKT>
static int launch_attention_bwd_wide(const BCtx &c, const float *qkv, const float *dA, const float *aout, float *dqkv, unsigned *mw, int kro,
                                     const unsigned *da_scale = nullptr) {
  const int H = c.m->d / HD, waves = std::min(H, HD == 64 ? 4 : abww::MAXW);
  const size_t smem = abww::lds_bytes(HD, NKT, c.g.N, waves);
  if (smem > 160 * 1024) return ALINE_EUNSUPPORTED;
  if (da_scale) {      // the f16 matrix pipe, dO scaled by the power of two of max |dA| (attn_bwd_wide.h: attention_bwd_wide16_kernel)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&abww::attention_bwd_wide16_kernel<HD, NKT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((abww::attention_bwd_wide16_kernel<HD, NKT>), dim3((unsigned)c.g.B), dim3(64 * waves), smem, c.st, c.g, c.m->d, qkv, dA, aout, dqkv, mw, kro, da_scale);
    CHECK_LAUNCH();
    return ALINE_OK;
  }
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&abww::attention_bwd_wide_kernel<HD, NKT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL((abww::attention_bwd_wide_kernel<HD, NKT>), dim3((unsigned)c.g.B), dim3(64 * waves), smem, c.st, c.g, c.m->d, qkv, dA, aout, dqkv, mw, kro);
  CHECK_LAUNCH();
  return ALINE_OK;
}

template <int NKT>
static int launch_attention_bwd8(const BCtx &c, const float *qkv, const float *dA, const float *aout, float *dqkv, const unsigned *da_scale) {
  const size_t smem = abw8::lds_bytes(NKT, c.g.N);
  if (smem > 160 * 1024) return ALINE_EUNSUPPORTED;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&abw8::attention_bwd8_kernel<NKT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL((abw8::attention_bwd8_kernel<NKT>), dim3((unsigned)c.g.B), dim3(64 * abw8::WAVES), smem, c.st, c.g, qkv, dA, aout, dqkv, da_scale);
  CHECK_LAUNCH();
  return ALINE_OK;
}

template <int HD>
int launch_attention_bwd(const BCtx &c, const float *qkv, const float *dA, float *dqkv, int max_keys, const float *aout = nullptr,
                         unsigned **out_max = nullptr, bool key_rows_only = false, const unsigned *da_scale = nullptr) {
  const bool mfma_on = !dbg(ALINE_DBG_NO_BWD_ATTN_MFMA);      // 0: the VALU kernel (A/B measurements)
  if (out_max) *out_max = nullptr;
  if constexpr (HD == 8) {
    // d = 32 / 4 heads beyond the fused attention block's 48 keys (cfg3): the f16 matrix pipe (attn_bwd8.h), given the scale word of dA
    if (mfma_on && aout && da_scale && c.m->d == 32 && c.m->H == 4 && max_keys <= 160 && c.m->precision == ALINE_PREC_F16X3 &&
        !dbg(ALINE_DBG_BWD_GRAD_F32)) {
      if (max_keys <= 64) return launch_attention_bwd8<4>(c, qkv, dA, aout, dqkv, da_scale);
      if (max_keys <= 112) return launch_attention_bwd8<7>(c, qkv, dA, aout, dqkv, da_scale);
      return launch_attention_bwd8<10>(c, qkv, dA, aout, dqkv, da_scale);
    }
  }
  if constexpr (HD == 32 || HD == 64) {
    const int nkt = (max_keys + 15) / 16;
    if (mfma_on && aout && nkt <= 3 && c.m->d % HD == 0) {      // (<32, 4> spills 64 registers: beyond 48 keys the VALU kernel)
      unsigned *mw = (out_max && bwd_grad_f16(*c.m)) ? new_scale_word(c) : nullptr;
      if (out_max) *out_max = mw;
      const unsigned *wide_scale = (da_scale && bwd_grad_f16(*c.m)) ? da_scale : nullptr;      // F16X3 model + the scale word of dA: the f16 twin
      if constexpr (HD == 32) {
        switch (nkt) {
          case 1: return launch_attention_bwd_wide<32, 1>(c, qkv, dA, aout, dqkv, mw, key_rows_only ? 1 : 0, wide_scale);
          case 2: return launch_attention_bwd_wide<32, 2>(c, qkv, dA, aout, dqkv, mw, key_rows_only ? 1 : 0, wide_scale);
          default: return launch_attention_bwd_wide<32, 3>(c, qkv, dA, aout, dqkv, mw, key_rows_only ? 1 : 0, wide_scale);
        }
      } else {
        if (nkt == 1) return launch_attention_bwd_wide<64, 1>(c, qkv, dA, aout, dqkv, mw, key_rows_only ? 1 : 0, wide_scale);
        if (nkt == 2) return launch_attention_bwd_wide<64, 2>(c, qkv, dA, aout, dqkv, mw, key_rows_only ? 1 : 0, wide_scale);
        return launch_attention_bwd_wide<64, 3>(c, qkv, dA, aout, dqkv, mw, key_rows_only ? 1 : 0, wide_scale);      // (cfg5: 1 + 29 context + 4 targets = 34 keys)
      }
    }
  }
  if (mfma_on && HD == abwd::HD && c.m->d == abwd::D && max_keys <= abwd::MAXK) {
    hipLaunchKernelGGL(abwd::attention_bwd_mfma_kernel, dim3((unsigned)c.g.B), dim3(abwd::THREADS), 0, c.st, c.g, qkv, dA, dqkv);
    CHECK_LAUNCH();
    return ALINE_OK;
  }
  const int nthr = HD <= 16 ? std::min(512, std::max(256, (c.g.N + 63) / 64 * 64)) : 256;      // whole waves of token rows
  // K, V (+ the dK / dV sums where they cannot take their place: backward.h), key list | scaled Q, dO, row statistics
  const bool reuse = HD <= 16 && max_keys <= nthr;
  size_t smem = (size_t)max_keys * ((reuse ? 2 : 4) * HD * sizeof(float) + sizeof(int)) +
                (size_t)c.g.N * (2 * HD + 4) * sizeof(float);
  if (smem > 160 * 1024 - 1024) return ALINE_EUNSUPPORTED;
  const float *ao = dbg(ALINE_DBG_NO_BWD_ATTN_MFMA) ? (const float *)nullptr : aout;
  if constexpr (HD <= 16) {
    if (reuse) {
      if (smem > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&attention_bwd_kernel<HD, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)smem);
      hipLaunchKernelGGL((attention_bwd_kernel<HD, true>), dim3((unsigned)c.g.B), dim3(nthr), smem, c.st, c.g, c.m->d, qkv, dA, dqkv, max_keys, ao);
      CHECK_LAUNCH();
      return ALINE_OK;
    }
  }
  if (smem > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&attention_bwd_kernel<HD, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL((attention_bwd_kernel<HD, false>), dim3((unsigned)c.g.B), dim3(nthr), smem, c.st, c.g, c.m->d, qkv,
                     dA, dqkv, max_keys, ao);
  CHECK_LAUNCH();
  return ALINE_OK;
}

// token-local tail of layer l (tail_bwd.h): forward recompute (dY == nullptr: Y = layer output) or forward + backward
int launch_tail(const BCtx &c, int l, const float *X, const float *A, float *Y, const float *dY, float *dA, float *dU,
                const aline_grads *gr, long M, unsigned **da_max = nullptr, const unsigned *dy_scale = nullptr) {
  if (da_max) *da_max = nullptr;
  const aline_model &m = *c.m;
  tailbwd::Args a{};
  a.X = X; a.A = A; a.dY = dY; a.Y = Y; a.dA = dA; a.dU = dU; a.M = M;
  a.wo = m.out_proj_w[l]; a.bo = m.out_proj_b[l]; a.w1 = m.lin1_w[l]; a.b1 = m.lin1_b[l]; a.w2 = m.lin2_w[l]; a.b2 = m.lin2_b[l];
  a.g1 = m.norm1_w[l]; a.e1 = m.norm1_b[l]; a.g2 = m.norm2_w[l]; a.e2 = m.norm2_b[l];
  const size_t smem = (dY ? tailbwd::LDS_FLOATS : tailbwd::LDS_FLOATS_FWD) * sizeof(float);
  const long groups = ((M + 15) / 16 + tailbwd::WAVES - 1) / tailbwd::WAVES;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tailbwd::tail_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, tailbwd::LDS_FLOATS * (int)sizeof(float));
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tailbwd::tail_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, tailbwd::LDS_FLOATS_FWD * (int)sizeof(float));
  if (!dY) {
    hipLaunchKernelGGL(tailbwd::tail_kernel<false>, dim3((unsigned)std::min<long>(groups, 256 * 3)), dim3(tailbwd::THREADS), smem, c.st, a);
  } else {
    a.dwo = gr->out_proj_w[l]; a.dbo = gr->out_proj_b[l]; a.dw1 = gr->lin1_w[l]; a.db1 = gr->lin1_b[l];
    a.dw2 = gr->lin2_w[l]; a.db2 = gr->lin2_b[l]; a.dg1 = gr->norm1_w[l]; a.de1 = gr->norm1_b[l];
    a.dg2 = gr->norm2_w[l]; a.de2 = gr->norm2_b[l];
    if (m.precision == ALINE_PREC_F16X3 && !dbg(ALINE_DBG_BWD_GRAD_F32)) {
      // the tile program on the f16 matrix pipe (tail_bwd.h: tail16_kernel), every gradient scaled by the power of two of max |dY|
      a.dy_max_bits = dy_scale ? dy_scale : grad_absmax(c, dY, M, tailbwd::D, tailbwd::D);      // (the layer above's attention block left max |dX|)
      if (da_max) { a.da_absmax = new_scale_word(c); *da_max = a.da_absmax; }      // max |dA|: the scale of the attention block's f16 kernel
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tailbwd::tail16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, tailbwd::LDS_FLOATS * (int)sizeof(float));
      hipLaunchKernelGGL(tailbwd::tail16_kernel, dim3((unsigned)std::min<long>(groups, 256)), dim3(tailbwd::THREADS), smem, c.st, a);
      CHECK_LAUNCH();
      return ALINE_OK;
    }
    hipLaunchKernelGGL(tailbwd::tail_kernel<true>, dim3((unsigned)std::min<long>(groups, 256)), dim3(tailbwd::THREADS), smem, c.st, a);   // 434 registers: one wave per SIMD
  }
  CHECK_LAUNCH();
  return ALINE_OK;
}

}  // namespace

extern "C" size_t aline_rollout_backward_workspace_bytes(const aline_model *m, const aline_rollout *r,
                                                         int t_chunk) {
  if (!m || !r || validate_model(*m, 0) != 0 || t_chunk < 1) return 0;
  return make_bwd_plan(*m, r->B, r->P, r->n_target_data, std::min(t_chunk, (int)r->T)).total * sizeof(float);
}

extern "C" int aline_rollout_backward(const aline_model *m, const aline_rollout *r, const float *g_logp,
                                      const float *g_ll, const aline_grads *gr, int t_chunk, void *ws,
                                      size_t ws_bytes, void *stream) {
  if (!g_ll) return ALINE_EINVAL;
  return aline_rollout_backward_ex(m, r, g_logp, g_ll, nullptr, nullptr, nullptr, gr, t_chunk, ws, ws_bytes, stream);
}

// Which part of the chain a backward call covers, and the tensors at its cut points (token rows in the reference order
// ctx | query | target data | theta tokens, [B * N, d]; stage calls are single steps, T = 1):
//   x_in   input of the encoder (ST_ENC without ST_EMBED)        z_in   encoder output (ST_HEAD without ST_ENC)
//   d_in   upstream gradient wrt the stage's output (ST_ENC without ST_HEAD: dz; ST_EMBED alone: dx)
//   d_out  gradient wrt the stage's input (ST_HEAD alone: dz; ST_ENC without ST_EMBED: dx)
// per-instance attention kernels of the training backward (<= 48 keys: two or three key tiles).  Instantiations for 4 .. 10 key tiles (up
// to 160 keys: cfg3) were built and measured in round 3 and are NOT kept: with all scores of a head and the dK / dV tiles of every key tile in
// registers they run at one wave per SIMD (8 tiles: 94, 10 tiles: 281 spills) -- attn_block_bwd_kernel<10> 7.3 ms per call against 6.5 ms of the
// fp32-VALU kernel it would replace, layer_fwd_kernel<10> 3.7 ms against 2.9 ms of the three kernels it fuses (profiles/r03_cfg3_train_*).
template <int KT>
static int launch_layer_fwd_kt(hipStream_t st, const lfwd::Args &fa, int I) {
  const size_t lds = (size_t)lfwd::lds_floats(KT) * sizeof(float);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lfwd::layer_fwd_kernel<KT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(lfwd::layer_fwd_kernel<KT>, dim3((unsigned)std::min(I, 512)), dim3(lfwd::THREADS), lds, st, fa);      // two persistent workgroups per CU
  CHECK_LAUNCH();
  return ALINE_OK;
}
static int launch_layer_fwd(hipStream_t st, const lfwd::Args &fa, int I, int max_keys) {
  return max_keys <= 32 ? launch_layer_fwd_kt<2>(st, fa, I) : max_keys <= 48 ? launch_layer_fwd_kt<3>(st, fa, I) : ALINE_EUNSUPPORTED;
}
template <int KT>
static int launch_attn_block_bwd_kt(hipStream_t st, const abwd::BlockArgs &ba, int I) {
  if (ba.da_max_bits) {      // the f16 matrix pipe (attn_bwd_mfma.h: attn_block_bwd16_kernel)
    const size_t lds16 = (size_t)abwd::block16_lds_floats(KT) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&abwd::attn_block_bwd16_kernel<KT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
    hipLaunchKernelGGL(abwd::attn_block_bwd16_kernel<KT>, dim3((unsigned)std::min(I, KT <= 2 ? 512 : 256)), dim3(abwd::THREADS), lds16, st, ba);
    CHECK_LAUNCH();
    return ALINE_OK;
  }
  const size_t lds = (size_t)abwd::block_lds_floats(KT) * sizeof(float);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&abwd::attn_block_bwd_kernel<KT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  // persistent workgroups: two per CU at <= 32 keys (242 registers), one otherwise
  hipLaunchKernelGGL(abwd::attn_block_bwd_kernel<KT>, dim3((unsigned)std::min(I, KT <= 2 ? 512 : 256)), dim3(abwd::THREADS), lds, st, ba);
  CHECK_LAUNCH();
  return ALINE_OK;
}
static int launch_attn_block_bwd(hipStream_t st, const abwd::BlockArgs &ba, int I, int max_keys) {
  return max_keys <= 32 ? launch_attn_block_bwd_kt<2>(st, ba, I) : max_keys <= 48 ? launch_attn_block_bwd_kt<3>(st, ba, I) : ALINE_EUNSUPPORTED;
}

struct StageIO { int stages; const float *x_in, *z_in, *d_in; float *d_out; };

static int backward_impl(const aline_model *m, const aline_rollout *r, const float *g_logp, const float *g_ll,
                         const float *g_pm, const float *g_ps, const float *g_pw, const aline_grads *gr, int t_chunk,
                         void *ws, size_t ws_bytes, void *stream, StageIO io) {
  const bool do_emb = io.stages & ST_EMBED, do_enc = io.stages & ST_ENC, do_head = io.stages & ST_HEAD;
  if (!m || !r || !gr || !ws || t_chunk < 1) return ALINE_EINVAL;
  if (do_head && (!g_logp || (!g_ll && !g_pm && !g_ps && !g_pw))) return ALINE_EINVAL;
  TRY(validate_model(*m, io.stages));
  // (the forward activations are recomputed here in exact fp32 whichever reference-precision mode rolled the episodes out)
  if (m->precision != ALINE_PREC_F32 && m->precision != ALINE_PREC_F16X3) return ALINE_EUNSUPPORTED;
  if (!r->role || (do_head && (!r->slot || !r->target_all)) || (do_emb && (!r->point_x || !r->point_y))) return ALINE_EINVAL;
  if (io.stages != ST_ALL && r->T != 1) return ALINE_EINVAL;
  if ((do_enc && !do_emb && !io.x_in) || (do_head && !do_enc && !io.z_in) || (!do_head && !io.d_in) ||
      (!do_emb && !io.d_out))
    return ALINE_EINVAL;
  const int tc = std::min(t_chunk, (int)r->T);
  BCtx c;
  c.m = m;
  c.pl = make_bwd_plan(*m, r->B, r->P, r->n_target_data, tc);
  if (ws_bytes < c.pl.total * sizeof(float)) return ALINE_EWORKSPACE;
  c.ws = static_cast<float *>(ws);
  c.st = static_cast<hipStream_t>(stream);
  const bool ft = bwd_fused_tail(*m);
  const int B = r->B, P = r->P, n_td = r->n_target_data, n_th = m->n_theta, n_t = n_td + n_th;
  const int N = P + n_t, d = m->d, F = m->F, L = m->L, C = m->C, hd = m->H > 0 ? d / m->H : 0;
  const int rows_x = B * (P + n_td), rows_y = B * P;

  // ---- step-invariant point embeddings, keeping the hidden activations -----------------------------
  float *Ex = c.at(c.pl.Ex), *Ey = c.at(c.pl.Ey), *EHx = c.at(c.pl.EHx), *EHy = c.at(c.pl.EHy);
  Src3 xs{{r->point_x, r->target_x, nullptr}, {P, n_td, 0}};
  Src3 ys{{r->point_y, nullptr, nullptr}, {P, 0, 0}};
  if (do_emb) {
  hipLaunchKernelGGL(embed_hidden_kernel, grid1d((size_t)rows_x * F), dim3(256), 0, c.st, xs, P + n_td, B,
                     m->dim_x, F, m->x_w1, m->x_b1, EHx);
  CHECK_LAUNCH();
  TRY(launch_gemm(recompute_prec(*m), gemm_args(EHx, F, m->x_w2, m->x_b2, F, Ex, d, rows_x, d, F, false), 1, c.st));
  hipLaunchKernelGGL(embed_hidden_kernel, grid1d((size_t)rows_y * F), dim3(256), 0, c.st, ys, P, B, m->dim_y,
                     F, m->y_w1, m->y_b1, EHy);
  CHECK_LAUNCH();
  TRY(launch_gemm(recompute_prec(*m), gemm_args(EHy, F, m->y_w2, m->y_b2, F, Ey, d, rows_y, d, F, false), 1, c.st));
  CHECK_LAUNCH();
  (void)hipMemsetAsync(c.at(c.pl.dEx), 0, (size_t)rows_x * d * sizeof(float), c.st);
  (void)hipMemsetAsync(c.at(c.pl.dEy), 0, (size_t)rows_y * d * sizeof(float), c.st);
  }

  const bool img_rc = c.pl.rc_width && do_emb && do_enc && !ft;
  if (img_rc) {
    if (c.pl.rc_width == x3::D) TRY(x3::pack_weights(*m, reinterpret_cast<unsigned *>(c.at(c.pl.rc3.img)), nullptr, c.st));
    else TRY(x5::pack_weights(*m, reinterpret_cast<unsigned *>(c.at(c.pl.rc5.img)), nullptr, c.st));
  }
  for (int tA = 0; tA < r->T; tA += tc) {
    const int nt_steps = std::min(tc, r->T - tA);
    const int I = B * nt_steps;
    const long M = (long)I * N;
    Geo &g = c.g;
    g.B = I; g.P = P; g.n_td = n_td; g.n_th = n_th; g.N = N; g.n_ctx = 0; g.role = r->role;
    g.tmask = r->target_mask; g.inst_B = B; g.inst_t0 = tA; g.n_ctx0 = r->n_ctx0;
    const int max_keys = r->n_ctx0 + tA + nt_steps - 1 + n_t;
    // layer inputs / attention outputs of this chunk: recomputed into the workspace, or the rows the s3 rollout saved (aline_rollout.saved_acts)
    const bool use_saved = r->saved_acts && do_emb && do_enc && do_head && saved_acts_usable(*m, *r) && !dbg(ALINE_DBG_NO_BWD_SAVED_ACTS);
    const size_t sv_rows = (size_t)r->T * B * N, sv_off = (size_t)tA * B * N * d;
    auto Xs = [&](int l) { return use_saved ? r->saved_acts + (size_t)l * sv_rows * d + sv_off : c.at(c.pl.Xs) + (size_t)l * M * d; };
    auto QKVl = [&](int l) { return c.at(c.pl.QKV) + (size_t)l * M * 3 * d; };
    auto Al = [&](int l) { return use_saved ? r->saved_acts + (size_t)(L + 1 + l) * sv_rows * d + sv_off : c.at(c.pl.A) + (size_t)l * M * d; };
    auto U1l = [&](int l) { return c.at(c.pl.U1) + (size_t)l * M * d; };
    auto X1l = [&](int l) { return c.at(c.pl.X1) + (size_t)l * M * d; };
    auto Hidl = [&](int l) { return c.at(c.pl.Hid) + (size_t)l * M * F; };
    auto U2l = [&](int l) { return c.at(c.pl.U2) + (size_t)l * M * d; };
    float *dTmp = c.at(c.pl.dTmp), *dQKV = c.at(c.pl.dQKV), *dHid = c.at(c.pl.dHid);

    // ---- forward recompute, saving what the backward needs --------------------------------------------
    if (use_saved) {
      // (X_0 is in the saved rows too)
    } else if (do_emb && do_enc) {
      if (d % 4 == 0) hipLaunchKernelGGL(assemble4_kernel, grid1d((size_t)M * d / 4), dim3(256), 0, c.st, g, d, Ex, Ey, P, m->theta_tokens, Xs(0));
      else
      hipLaunchKernelGGL(assemble_kernel, grid1d((size_t)M * d), dim3(256), 0, c.st, g, d, Ex, Ey, P,
                         m->theta_tokens, Xs(0));
      CHECK_LAUNCH();
    } else if (do_enc) {
      (void)hipMemcpyAsync(Xs(0), io.x_in, (size_t)M * d * sizeof(float), hipMemcpyDeviceToDevice, c.st);
    }
    // With the fused attention-block backward nothing downstream reads QKV: Q for every row, K / V for the key rows only
    // (key_list_kernel + row-gather GEMM, as the generic rollout pipeline does), one buffer for all layers.
    const bool ckv = do_enc && fused_attn_block(*m, max_keys) && hd == abwd::HD && max_keys < N;
    int *keyidx = reinterpret_cast<int *>(c.at(c.pl.KeyIdx)), *kcnt = reinterpret_cast<int *>(c.at(c.pl.Kcnt));
    // compact K / V of layer l [I * max_keys, 2 d] (kept for the backward), then dK | dV of the layer in flight, then Q
    const size_t kv_floats = (size_t)I * max_keys * 2 * d;
    auto KVl = [&](int l) { return QKVl(0) + (size_t)l * kv_floats; };
    float *dKVc = QKVl(0) + (size_t)L * kv_floats, *Qbuf = dKVc + kv_floats;
    if (ckv) {
      hipLaunchKernelGGL(key_list_kernel, dim3(I), dim3(256), 0, c.st, g, max_keys, keyidx, kcnt);
      CHECK_LAUNCH();
    }
    // (d = 256 / 512, <= 64 keys: everything but QKV from the rollout's layer kernel run over the instances)
    const bool rc_now = img_rc && !use_saved && !ckv && max_keys <= x3::WNK;
    // ... Q too when the matrix-pipe attention backward runs (it reads K / V of the key rows only: those come from a row-gather GEMM on
    // the key list, 16 % of the rows at the d = 256 headline shape, scattered into the [M, 3 d] buffer the kernel indexes by token row)
    const bool rc_q = rc_now && max_keys < N && max_keys <= 48 && (hd == 32 || hd == 64) && !dbg(ALINE_DBG_NO_BWD_ATTN_MFMA);
    // The same key list on the way back: dK / dV are zero outside the key rows, so the in-projection's gradient products split into a
    // dense Q part (a third of the columns) and a K | V part over the key rows only (gathered by the list; dX scattered back)
    const bool kv_sparse = do_enc && !ckv && bwd_grad_f16(*m) && !dbg(ALINE_DBG_NO_BWD_ATTN_MFMA) && !dbg(ALINE_DBG_NO_BWD_KV_SPARSE) &&
                           (hd == 32 || hd == 64) && max_keys <= 48 && max_keys < N && d % 256 == 0;
    if (rc_q || kv_sparse) {
      hipLaunchKernelGGL(key_list_kernel, dim3(I), dim3(256), 0, c.st, g, max_keys, keyidx, kcnt);
      CHECK_LAUNCH();
    }
    if (rc_now) {
      if (c.pl.rc_width == x3::D) {
        x3::RecomputeRows rr{};
        rr.q_ld = 3 * d;
        for (int l = 0; l < L; ++l) { rr.Q[l] = rc_q ? QKVl(l) : c.at(c.pl.dQKV); rr.A[l] = Al(l); rr.U1[l] = U1l(l); rr.X1[l] = X1l(l); rr.Hid[l] = Hidl(l); rr.U2[l] = U2l(l); rr.Y[l] = Xs(l + 1); }
        TRY(x3::recompute(m, g, max_keys, Ex, Ey, P, c.ws, c.pl.rc3, rr, c.st));
      } else {
        x5::RecomputeRows rr{};
        rr.q_ld = 3 * d;
        for (int l = 0; l < L; ++l) { rr.Q[l] = rc_q ? QKVl(l) : c.at(c.pl.dQKV); rr.A[l] = Al(l); rr.U1[l] = U1l(l); rr.X1[l] = X1l(l); rr.Hid[l] = Hidl(l); rr.U2[l] = U2l(l); rr.Y[l] = Xs(l + 1); }
        TRY(x5::recompute(m, g, max_keys, Ex, Ey, P, c.ws, c.pl.rc5, rr, c.st));
      }
    }
    for (int l = 0; l < L && do_enc; ++l) {
      Ctx fc{}; fc.m = m; fc.g = g; fc.st = c.st; fc.ws = c.ws;
      if (rc_q) {      // K | V of the key rows, stored at their token rows
        GemmArgs ka = gemm_args(Xs(l), d, m->in_proj_w[l] + (size_t)d * d, m->in_proj_b[l] + d, d, QKVl(l) + d, 3 * d, I * max_keys, 2 * d, d, false);
        ka.row_index = keyidx; ka.out_index = keyidx;
        TRY(launch_gemm(recompute_prec(*m), ka, 1, c.st));
        continue;
      }
      if (rc_now) {      // the in-projection (K, V never leave the layer kernel's key image)
        TRY(launch_gemm(recompute_prec(*m), gemm_args(Xs(l), d, m->in_proj_w[l], m->in_proj_b[l], d, QKVl(l), 3 * d, (int)M, 3 * d, d, false), 1, c.st));
        continue;
      }
      if (ckv) {      // K / V of the key rows: row-gather GEMM on the key list
        GemmArgs ka = gemm_args(Xs(l), d, m->in_proj_w[l] + (size_t)d * d, m->in_proj_b[l] + d, d, KVl(l), 2 * d, I * max_keys, 2 * d, d, false);
        ka.row_index = keyidx;
        TRY(launch_gemm(recompute_prec(*m), ka, 1, c.st));
      }
      if (use_saved) {      // a and the layer's output are in the saved rows; without the fused attention block its backward reads QKV
        if (!ckv) TRY(launch_gemm(recompute_prec(*m), gemm_args(Xs(l), d, m->in_proj_w[l], m->in_proj_b[l], d, QKVl(l), 3 * d, (int)M, 3 * d, d, false), 1, c.st));
        continue;
      }
      if (ckv && ft && !dbg(ALINE_DBG_NO_BWD_LAYER_FWD)) {      // the rest of the layer in one kernel (layer_fwd.h): Xs(l) -> Al(l), Xs(l + 1)
        lfwd::Args fa{};
        fa.g = g; fa.X = Xs(l); fa.A = Al(l); fa.Y = Xs(l + 1); fa.win = m->in_proj_w[l]; fa.bin = m->in_proj_b[l];
        fa.kvc = KVl(l); fa.kcnt = kcnt; fa.max_keys = max_keys;
        fa.wo = m->out_proj_w[l]; fa.bo = m->out_proj_b[l]; fa.w1 = m->lin1_w[l]; fa.b1 = m->lin1_b[l];
        fa.w2 = m->lin2_w[l]; fa.b2 = m->lin2_b[l]; fa.g1 = m->norm1_w[l]; fa.e1 = m->norm1_b[l];
        fa.g2 = m->norm2_w[l]; fa.e2 = m->norm2_b[l];
        if (max_keys <= 48 && !dbg(ALINE_DBG_NO_BWD_LAYER_FWD_FLAT)) {      // (instance, tile) units from one flat list, K / V fragments from L2
          const long units = (long)I * ((N + 15) / 16);
          const unsigned fgrid = (unsigned)std::min<long>((units + lfwd::WAVES - 1) / lfwd::WAVES, 768);      // three workgroups per CU
          const size_t fl = lfwd::LDS_FLOATS_FLAT * sizeof(float);
          if (max_keys <= 32) hipLaunchKernelGGL(lfwd::layer_fwd_flat_kernel<2>, dim3(fgrid), dim3(lfwd::THREADS), fl, c.st, fa);
          else hipLaunchKernelGGL(lfwd::layer_fwd_flat_kernel<3>, dim3(fgrid), dim3(lfwd::THREADS), fl, c.st, fa);
          CHECK_LAUNCH();
          continue;
        }
        TRY(launch_layer_fwd(c.st, fa, I, max_keys));
        continue;
      }
      if (ckv) {
        TRY(launch_gemm(recompute_prec(*m), gemm_args(Xs(l), d, m->in_proj_w[l], m->in_proj_b[l], d, Qbuf, d, (int)M, d, d, false), 1, c.st));
        TRY(launch_attention<8>(fc, Qbuf, Al(l), max_keys, KVl(l), kcnt));
      } else {
      TRY(launch_gemm(recompute_prec(*m), gemm_args(Xs(l), d, m->in_proj_w[l], m->in_proj_b[l], d, QKVl(l), 3 * d, (int)M, 3 * d, d, false), 1, c.st));
      switch (hd) {
        case 4: TRY(launch_attention<4>(fc, QKVl(l), Al(l), max_keys)); break;
        case 8: TRY(launch_attention<8>(fc, QKVl(l), Al(l), max_keys)); break;
        case 16: TRY(launch_attention<16>(fc, QKVl(l), Al(l), max_keys)); break;
        case 32: TRY(launch_attention<32>(fc, QKVl(l), Al(l), max_keys)); break;
        case 64: TRY(launch_attention<64>(fc, QKVl(l), Al(l), max_keys)); break;
        default: return ALINE_EUNSUPPORTED;
      }
      }
      if (ft) { TRY(launch_tail(c, l, Xs(l), Al(l), Xs(l + 1), nullptr, nullptr, nullptr, nullptr, M)); continue; }
      TRY(launch_gemm(recompute_prec(*m), gemm_args(Al(l), d, m->out_proj_w[l], m->out_proj_b[l], d, dTmp, d, (int)M, d, d, false), 1, c.st));
      TRY(launch_add_layernorm(c.st, Xs(l), dTmp, m->norm1_w[l], m->norm1_b[l], X1l(l), M, d, U1l(l)));
      TRY(launch_gemm(recompute_prec(*m), gemm_args(X1l(l), d, m->lin1_w[l], m->lin1_b[l], d, Hidl(l), F, (int)M, F, d, true), 1, c.st));
      TRY(launch_gemm(recompute_prec(*m), gemm_args(Hidl(l), F, m->lin2_w[l], m->lin2_b[l], F, dTmp, d, (int)M, d, F, false), 1, c.st));
      TRY(launch_add_layernorm(c.st, X1l(l), dTmp, m->norm2_w[l], m->norm2_b[l], Xs(l + 1), M, d, U2l(l)));
    }
    const float *Z = do_enc ? Xs(L) : io.z_in;
    float *HidA = c.at(c.pl.HidA), *HidG = c.at(c.pl.HidG);
    float *tvec = c.at(c.pl.Tvec);                               // time token: t of every instance of this chunk
    const bool facq = do_head && fused_acq_head(*m);
    // GMM heads without the [rows, C F] hidden activations (acq_head_bwd.h, gmmb).  ALINE_BWD_GMM_FUSED=0: the per-op kernels.
    const bool fgmm = do_head && n_t > 0 && fused_gmm_heads(*m);
    if (do_head) {
      if (!facq) {
      const int ldw1 = m->time_token ? d + 1 : d;                // head.py:24-25: [F, d + 1] with a time token
      GemmArgs a = gemm_args(Z, d, m->acq_w1, m->acq_b1, ldw1, HidA, F, I * P, F, d, true);
      a.R_in = P; a.G_in = N; a.off_in = 0;
      if (m->time_token) {     // + t(instance) * W1[:, d]  (model/head.py:342-345; t / T per step: train_aline.py:80-82)
        const int TT = r->time_token_T > 0 ? r->time_token_T : r->time_token_T < 0 ? -r->time_token_T : r->T;
        hipLaunchKernelGGL(time_vec_kernel, grid1d((size_t)I), dim3(256), 0, c.st, tvec, I, B, tA, TT, r->time_token_T < 0 ? 1 : 0);
        CHECK_LAUNCH();
        a.tvec = tvec; a.tvec_div = P; a.tcol = m->acq_w1 + d; a.tcol_stride = d + 1;
      }
      TRY(launch_gemm(recompute_prec(*m), a, 1, c.st));
      }
      if (!fgmm) {
      GemmArgs ag = gemm_args(Z, d, nullptr, nullptr, d, HidG, C * F, I * n_t, F, d, true);
      ag.R_in = n_t; ag.G_in = N; ag.off_in = P; ag.col_per_group = F;
      for (int k = 0; k < C; ++k) { ag.W[k] = m->gmm_w1[k]; ag.bias[k] = m->gmm_b1[k]; }
      TRY(launch_gemm(recompute_prec(*m), ag, C, c.st));
      }
      CHECK_LAUNCH();
    }

    // ---- heads backward -> dZ --------------------------------------------------------------------------
    float *dX = c.at(c.pl.dXa), *dXn = c.at(c.pl.dXb);
    if (do_head && !facq) (void)hipMemsetAsync(dX, 0, (size_t)M * d * sizeof(float), c.st);
    else if (!do_head) (void)hipMemcpyAsync(dX, io.d_in, (size_t)M * d * sizeof(float), hipMemcpyDeviceToDevice, c.st);
    if (facq) {            // logits -> dLoss/dlogit -> dz of every row (zeros on the target rows) + parameter gradients
      acqb::Args a{};
      a.Z = Z; a.logit = dTmp; a.dZ = dX; a.M = M;
      a.w1 = m->acq_w1; a.b1 = m->acq_b1; a.w2 = m->acq_w2; a.b2 = m->acq_b2;
      a.dw1 = gr->acq_w1; a.db1 = gr->acq_b1; a.dw2 = gr->acq_w2;
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&acqb::bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, acqb::LDS_FLOATS * (int)sizeof(float));
      const long groups = ((M + 15) / 16 + acqb::WAVES - 1) / acqb::WAVES;
      if (m->precision == ALINE_PREC_F16X3 && !dbg(ALINE_DBG_BWD_GRAD_F32))
        hipLaunchKernelGGL(acqb::logit_kernel<true>, dim3((unsigned)std::min<long>(groups, 256 * 4)), dim3(acqb::THREADS), acqb::LDS_FLOATS_LOGIT * sizeof(float), c.st, a);
      else
        hipLaunchKernelGGL(acqb::logit_kernel<false>, dim3((unsigned)std::min<long>(groups, 256 * 4)), dim3(acqb::THREADS), acqb::LDS_FLOATS_LOGIT * sizeof(float), c.st, a);
      CHECK_LAUNCH();
      // F16X3 models: the products of the head kernels on the f16 matrix pipe, the gradient scaled by the power of two of its maximum
      // (tail_bwd.h: tail16_kernel; the producers -- dlogit_kernel, draw_kernel -- reduce what they write).  ALINE_DBG_BWD_GRAD_F32: exact fp32
      const bool head16 = m->precision == ALINE_PREC_F16X3 && !dbg(ALINE_DBG_BWD_GRAD_F32);
      acqb::DlArgs dl{};
      dl.g = g; dl.logit = dTmp; dl.g_logp = g_logp; dl.slot = r->slot; dl.T = r->T; dl.db2 = gr->acq_b2;
      dl.out_absmax = head16 ? new_scale_word(c) : nullptr;
      a.g_max_bits = dl.out_absmax;
      hipLaunchKernelGGL(acqb::dlogit_kernel, dim3((unsigned)std::min((I + 3) / 4, 2048)), dim3(256), 0, c.st, dl);
      CHECK_LAUNCH();
      if (head16) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&acqb::bwd16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, acqb::LDS_FLOATS16 * (int)sizeof(float));
        hipLaunchKernelGGL(acqb::bwd16_kernel, dim3((unsigned)std::min<long>(groups, 512)), dim3(acqb::THREADS), acqb::LDS_FLOATS16 * sizeof(float), c.st, a);
      } else
      hipLaunchKernelGGL(acqb::bwd_kernel, dim3((unsigned)std::min<long>(groups, 512)), dim3(acqb::THREADS), acqb::LDS_FLOATS * sizeof(float), c.st, a);
      CHECK_LAUNCH();
    } else if (do_head) {
      AcqBwdArgs a{};
      a.g = g; a.F = F; a.hid = HidA; a.w2 = m->acq_w2; a.b2 = m->acq_b2; a.g_logp = g_logp; a.slot = r->slot;
      a.T = r->T; a.dw2 = gr->acq_w2; a.db2 = gr->acq_b2;
      unsigned *sw_acq = bwd_grad_f16(*m) ? new_scale_word(c) : nullptr;      // (the kernel reduces what it stores: both products below read it)
      a.out_absmax = sw_acq;
      hipLaunchKernelGGL(acq_bwd_kernel, dim3(I), dim3(256), (size_t)(P + F) * sizeof(float), c.st, a);
      CHECK_LAUNCH();
      const int ldw1 = m->time_token ? d + 1 : d;
      TRY(gemm_dw(c, HidA, F, Z, d, gr->acq_w1, gr->acq_b1, (long)I * P, F, d, 1, 1, 0, P, N, 0, ldw1, sw_acq));
      if (m->time_token) {     // the time column: dW1[f, d] += sum_rows dHidA[row, f] t(row)
        const long rows = (long)I * P, rpb = 512;
        hipLaunchKernelGGL(time_col_grad_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, c.st, HidA, F, rows, P, tvec,
                           gr->acq_w1, d + 1, d, rpb);
        CHECK_LAUNCH();
      }
      // dZ[point rows] = dHidA . W1a  (the first d columns of W1)
      float *Wt = c.at(c.pl.Wt);
      TRY(transpose_to(c, m->acq_w1, F, d, Wt, ldw1));
      GemmArgs ga = gemm_args(HidA, F, Wt, nullptr, F, dX, d, I * P, d, F, false);
      ga.R_out = P; ga.G_out = N; ga.off_out = 0;
      TRY(launch_grad_gemm(c, ga, sw_acq));
      CHECK_LAUNCH();
    }
    if (fgmm) {
      gmmb::Args a{};
      a.Z = Z; a.n_t = n_t; a.N = N; a.P = P; a.rows = (long)I * n_t; a.C = C; a.std_min = m->std_min;
      for (int k = 0; k < C; ++k) {
        a.w1[k] = m->gmm_w1[k]; a.b1[k] = m->gmm_b1[k]; a.w2[k] = m->gmm_w2[k]; a.b2[k] = m->gmm_b2[k];
        a.dw1[k] = gr->gmm_w1[k]; a.db1[k] = gr->gmm_b1[k]; a.dw2[k] = gr->gmm_w2[k]; a.db2[k] = gr->gmm_b2[k];
      }
      a.raw = HidG; a.draw = a.raw + a.rows * C * 4; a.dzc = a.draw + a.rows * C * 4; a.dZ = dX;      // (40 of the C F floats per row)
      a.value = r->target_all; a.value_mod = (long)B * n_t;
      a.g_ll = g_ll ? g_ll + (size_t)tA * B * n_t : nullptr;
      a.g_mean = g_pm ? g_pm + (size_t)tA * B * n_t * C : nullptr;
      a.g_std = g_ps ? g_ps + (size_t)tA * B * n_t * C : nullptr;
      a.g_wgt = g_pw ? g_pw + (size_t)tA * B * n_t * C : nullptr;
      const long groups = ((a.rows + 15) / 16 + gmmb::WAVES - 1) / gmmb::WAVES;
      if (m->precision == ALINE_PREC_F16X3 && !dbg(ALINE_DBG_BWD_GRAD_F32))
        hipLaunchKernelGGL(gmmb::raw_kernel<true>, dim3((unsigned)std::min<long>(groups, std::max(1, 768 / C)), C), dim3(gmmb::THREADS), gmmb::LDS_FLOATS_RAW * sizeof(float), c.st, a);
      else
        hipLaunchKernelGGL(gmmb::raw_kernel<false>, dim3((unsigned)std::min<long>(groups, std::max(1, 768 / C)), C), dim3(gmmb::THREADS), gmmb::LDS_FLOATS_RAW * sizeof(float), c.st, a);
      CHECK_LAUNCH();
      const bool gmm16 = m->precision == ALINE_PREC_F16X3 && !dbg(ALINE_DBG_BWD_GRAD_F32);
      a.draw_absmax = gmm16 ? new_scale_word(c) : nullptr;
      hipLaunchKernelGGL(gmmb::draw_kernel, dim3((unsigned)((a.rows + 255) / 256)), dim3(256), 0, c.st, a);
      CHECK_LAUNCH();
      if (gmm16) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gmmb::bwd16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, gmmb::LDS_FLOATS16 * (int)sizeof(float));
        hipLaunchKernelGGL(gmmb::bwd16_kernel, dim3((unsigned)std::min<long>(groups, std::max(1, 256 / C)), C), dim3(gmmb::THREADS), gmmb::LDS_FLOATS16 * sizeof(float), c.st, a);
      } else
      hipLaunchKernelGGL(gmmb::bwd_kernel, dim3((unsigned)std::min<long>(groups, std::max(1, 256 / C)), C), dim3(gmmb::THREADS), gmmb::LDS_FLOATS * sizeof(float), c.st, a);
      CHECK_LAUNCH();
      hipLaunchKernelGGL(gmmb::dzsum_kernel, grid1d((size_t)a.rows * 8), dim3(256), 0, c.st, a);
      CHECK_LAUNCH();
    } else if (do_head) {
      GmmBwdArgs a{};
      a.hid = HidG; a.rows = (long)I * n_t; a.C = C; a.F = F; a.std_min = m->std_min;
      for (int k = 0; k < C; ++k) { a.w2[k] = m->gmm_w2[k]; a.b2[k] = m->gmm_b2[k]; a.dw2[k] = gr->gmm_w2[k]; a.db2[k] = gr->gmm_b2[k]; }
      a.value = r->target_all; a.value_mod = (long)B * n_t;
      a.g_ll = g_ll ? g_ll + (size_t)tA * B * n_t : nullptr;
      a.g_mean = g_pm ? g_pm + (size_t)tA * B * n_t * C : nullptr;
      a.g_std = g_ps ? g_ps + (size_t)tA * B * n_t * C : nullptr;
      a.g_wgt = g_pw ? g_pw + (size_t)tA * B * n_t * C : nullptr;
      if (F == 128 && C <= 10 && !dbg(ALINE_DBG_NO_BWD_GMM128))
        hipLaunchKernelGGL((gmm_bwd128_kernel<10, 128>), dim3((unsigned)((a.rows + 127) / 128)), dim3(256), 0, c.st, a);
      else if (F == 128 && !dbg(ALINE_DBG_NO_BWD_GMM128))
        hipLaunchKernelGGL((gmm_bwd128_kernel<16, GMM_BWD_ROWS>), dim3((unsigned)((a.rows + GMM_BWD_ROWS - 1) / GMM_BWD_ROWS)), dim3(256), 0, c.st, a);
      else if (F > 128 && C <= 16 && !dbg(ALINE_DBG_NO_BWD_GMM_WIDE))
        hipLaunchKernelGGL(gmm_bwd_wide_kernel, dim3((unsigned)((a.rows + GMM_WIDE_ROWS - 1) / GMM_WIDE_ROWS)), dim3(256), 0, c.st, a);
      else
        hipLaunchKernelGGL(gmm_bwd_kernel, dim3((unsigned)((a.rows + GMM_BWD_ROWS - 1) / GMM_BWD_ROWS)), dim3(256), 0, c.st, a);
      CHECK_LAUNCH();
      float *Wt = c.at(c.pl.Wt);   // [d, C*F]: column block k = W1_k^T
      // one scale word for the hidden gradients of all C components (2 C products read them; f16's range below the maximum is 2^28)
      const unsigned *sw_gmm = bwd_grad_f16(*m) ? grad_absmax(c, HidG, (long)I * n_t, C * F, C * F) : nullptr;
      if (F == 128 && d % 32 == 0 && !dbg(ALINE_DBG_NO_BWD_GMM_BATCHED)) {
        // all C components in one launch each: dW1_k / db1_k (column block k of the hidden gradients), and dz as one K = C F product
        GemmTnArgs ta{};
        ta.dY = HidG; ta.ldy = C * F; ta.Ry = 1; ta.Gy = 1; ta.offy = 0;
        ta.X = Z; ta.ldx = d; ta.Rx = n_t; ta.Gx = N; ta.offx = P;
        ta.ldw = d; ta.M = (long)I * n_t; ta.N = C * F; ta.K = d; ta.grouped = 1;
        for (int k = 0; k < C; ++k) { ta.dWg[k] = gr->gmm_w1[k]; ta.dbg[k] = gr->gmm_b1[k]; }
        ta.mchunk = 4096;
        while (ta.mchunk > 256 && ((ta.M + ta.mchunk - 1) / ta.mchunk) * C * (d / 32) < 512) ta.mchunk /= 2;
        ta.gx = (int)((ta.M + ta.mchunk - 1) / ta.mchunk); ta.nby = C; ta.nbz = d / 32;
        hipLaunchKernelGGL(gemm_tn_block_kernel<8>, dim3((unsigned)((ta.gx + 7) / 8 * 8 * C * (d / 32))), dim3(256), 0, c.st, ta);
        CHECK_LAUNCH();
        PackW1Args pa{};
        for (int k = 0; k < C; ++k) pa.w1[k] = m->gmm_w1[k];
        pa.C = C; pa.F = F; pa.d = d; pa.out = Wt;
        hipLaunchKernelGGL(gmm_w1_pack_kernel, grid1d((size_t)C * F * d), dim3(256), 0, c.st, pa);
        CHECK_LAUNCH();
        GemmArgs ga = gemm_args(HidG, C * F, Wt, nullptr, C * F, dX, d, I * n_t, d, C * F, false);
        ga.R_out = n_t; ga.G_out = N; ga.off_out = P; ga.accum = 1;
        TRY(launch_grad_gemm(c, ga, sw_gmm));
      } else {
      for (int k = 0; k < C; ++k) {
        TRY(gemm_dw(c, HidG + (size_t)k * F, C * F, Z, d, gr->gmm_w1[k], gr->gmm_b1[k], (long)I * n_t, F, d, 1, 1, 0,
                    n_t, N, P, 0, sw_gmm));
        // Wt[kk, k*F + f] = W1_k[f, kk]
        hipLaunchKernelGGL(transpose_kernel, grid1d((size_t)F * d), dim3(256), 0, c.st, m->gmm_w1[k], F, d, d,
                           Wt + (size_t)k * F * d);   // temporarily packed [d, F] blocks, fixed below
        CHECK_LAUNCH();
      }
      // dZ[target rows] += sum_k dHidG_k . W1_k : one GEMM per component (K = F), accumulating
      for (int k = 0; k < C; ++k) {
        GemmArgs ga = gemm_args(HidG + (size_t)k * F, C * F, Wt + (size_t)k * F * d, nullptr, F, dX, d, I * n_t, d, F, false);
        ga.R_out = n_t; ga.G_out = N; ga.off_out = P; ga.accum = 1;
        TRY(launch_grad_gemm(c, ga, sw_gmm));
      }
      }
      CHECK_LAUNCH();
    }

    if (!do_enc && do_head) {       // head stage alone: dLoss / dz is the result
      (void)hipMemcpyAsync(io.d_out, dX, (size_t)M * d * sizeof(float), hipMemcpyDeviceToDevice, c.st);
      continue;
    }
    // ---- encoder layers backward ---------------------------------------------------------------------------
    unsigned *sw_dx = nullptr;       // (f16 fused kernels: max |dX_l| left by the attention block of layer l, the scale of layer l - 1's tail)
    for (int l = L - 1; l >= 0 && do_enc; --l) {
      unsigned *sw_da = nullptr;     // (fused tail on the f16 pipe: max |dA|, the scale of the attention block's f16 kernel)
      if (ft) {                      // dTmp = dA, dXn = dU1 (the residual branch), parameter gradients of the tail
        TRY(launch_tail(c, l, Xs(l), Al(l), nullptr, dX, dTmp, dXn, gr, M, &sw_da, sw_dx));
        sw_dx = nullptr;
      } else {
      // LN2
      // (sw_*: the scale words of the F16X3 gradient products -- the producer of a gradient tensor leaves max |.| for its readers)
      unsigned *sw_u2 = nullptr, *sw_u1 = nullptr;
      TRY(ln_bwd(c, dX, U2l(l), m->norm2_w[l], dTmp, gr->norm2_w[l], gr->norm2_b[l], M, &sw_u2));   // dTmp = dU2
      // FFN
      unsigned *sw_hid = bwd_grad_f16(*m) ? new_scale_word(c) : nullptr;
      TRY(gemm_dw(c, dTmp, d, Hidl(l), F, gr->lin2_w[l], gr->lin2_b[l], M, d, F, 1, 1, 0, 1, 1, 0, 0, sw_u2));
      TRY(gemm_dx(c, dTmp, d, m->lin2_w[l], d, F, dHid, F, (int)M, false, Hidl(l), sw_u2, sw_hid));
      TRY(gemm_dw(c, dHid, F, X1l(l), d, gr->lin1_w[l], gr->lin1_b[l], M, F, d, 1, 1, 0, 1, 1, 0, 0, sw_hid));
      TRY(gemm_dx(c, dHid, F, m->lin1_w[l], F, d, dTmp, d, (int)M, true, nullptr, sw_hid));     // dTmp = dX1
      // LN1
      TRY(ln_bwd(c, dTmp, U1l(l), m->norm1_w[l], dXn, gr->norm1_w[l], gr->norm1_b[l], M, &sw_u1));   // dXn = dU1
      // out-proj
      TRY(gemm_dw(c, dXn, d, Al(l), d, gr->out_proj_w[l], gr->out_proj_b[l], M, d, d, 1, 1, 0, 1, 1, 0, 0, sw_u1));
      if (bwd_grad_f16(*m)) sw_da = new_scale_word(c);      // (max |dA| for the attention backward's f16 twin)
      TRY(gemm_dx(c, dXn, d, m->out_proj_w[l], d, d, dTmp, d, (int)M, false, nullptr, sw_u1, sw_da));  // dTmp = dA
      }
      // attention block: in-projection + attention in one kernel (attn_bwd_mfma.h), dXn = dU1 -> dX_l
      if (ckv) {
        abwd::BlockArgs ba{};
        ba.g = g; ba.X = Xs(l); ba.dA = dTmp; ba.dX = dXn; ba.win = m->in_proj_w[l]; ba.bin = m->in_proj_b[l];
        ba.dwin = gr->in_proj_w[l]; ba.dbin = gr->in_proj_b[l];
        ba.kvc = KVl(l); ba.dkvc = dKVc; ba.keyidx = keyidx; ba.kcnt = kcnt; ba.max_keys = max_keys;
        ba.da_max_bits = sw_da;
        if (sw_da && l > 0) { sw_dx = new_scale_word(c); ba.dx_absmax = sw_dx; }
        TRY(launch_attn_block_bwd(c.st, ba, I, max_keys));
        // key rows: dx += Wk^T dK + Wv^T dV, Wk / Wv gradients (one wave per (instance, key tile))
        const long units = (long)I * ((max_keys + 15) / 16);
        hipLaunchKernelGGL(abwd::kv_bwd_kernel, dim3((unsigned)std::min<long>((units + abwd::WAVES - 1) / abwd::WAVES, 1024)), dim3(abwd::THREADS), 0, c.st, ba);
        CHECK_LAUNCH();
        std::swap(dX, dXn);
        continue;
      }
      // attention
      unsigned *sw_att = nullptr;
      switch (hd) {
        case 4: TRY(launch_attention_bwd<4>(c, QKVl(l), dTmp, dQKV, max_keys, Al(l))); break;
        case 8: TRY(launch_attention_bwd<8>(c, QKVl(l), dTmp, dQKV, max_keys, Al(l), nullptr, false, sw_da)); break;
        case 16: TRY(launch_attention_bwd<16>(c, QKVl(l), dTmp, dQKV, max_keys, Al(l))); break;
        case 32: TRY(launch_attention_bwd<32>(c, QKVl(l), dTmp, dQKV, max_keys, Al(l), &sw_att, kv_sparse, sw_da)); break;
        case 64: TRY(launch_attention_bwd<64>(c, QKVl(l), dTmp, dQKV, max_keys, Al(l), &sw_att, kv_sparse, sw_da)); break;
        default: return ALINE_EUNSUPPORTED;
      }
      // in-proj (the matrix-pipe attention backward leaves max |dQKV|; the VALU kernels do not: a reduction pass)
      if (kv_sparse) {
        if (!sw_att) return ALINE_EUNSUPPORTED;      // (kv_sparse implies the matrix-pipe kernel ran and left its scale word)
        const long MK = (long)I * max_keys;
        TRY(gemm_dw(c, dQKV, 3 * d, Xs(l), d, gr->in_proj_w[l], gr->in_proj_b[l], M, d, d, 1, 1, 0, 1, 1, 0, 0, sw_att));
        TRY(gemm_dw(c, dQKV + d, 3 * d, Xs(l), d, gr->in_proj_w[l] + (size_t)d * d, gr->in_proj_b[l] + d, MK, 2 * d, d, 1, 1, 0, 1, 1, 0, 0, sw_att, keyidx));
        TRY(gemm_dx(c, dQKV, 3 * d, m->in_proj_w[l], d, d, dXn, d, (int)M, true, nullptr, sw_att));      // dXn = dX_l: + dQ Wq
        float *Wt = c.at(c.pl.Wt);
        TRY(transpose_to(c, m->in_proj_w[l] + (size_t)d * d, 2 * d, d, Wt));
        GemmArgs ka = gemm_args(dQKV + d, 3 * d, Wt, nullptr, 2 * d, dXn, d, (int)MK, d, 2 * d, false);      // + dK Wk + dV Wv on the key rows
        ka.accum = 1; ka.row_index = keyidx; ka.out_index = keyidx;
        TRY(launch_grad_gemm(c, ka, sw_att));
        CHECK_LAUNCH();
        std::swap(dX, dXn);
        continue;
      }
      const unsigned *sw_qkv = !bwd_grad_f16(*m) ? nullptr : sw_att ? sw_att : grad_absmax(c, dQKV, M, 3 * d, 3 * d);
      TRY(gemm_dw(c, dQKV, 3 * d, Xs(l), d, gr->in_proj_w[l], gr->in_proj_b[l], M, 3 * d, d, 1, 1, 0, 1, 1, 0, 0, sw_qkv));
      TRY(gemm_dx(c, dQKV, 3 * d, m->in_proj_w[l], 3 * d, d, dXn, d, (int)M, true, nullptr, sw_qkv));   // dXn = dX_l
      std::swap(dX, dXn);
    }
    if (!do_emb) {                  // encoder (+ head) without the embedder: dLoss / dx is the result
      (void)hipMemcpyAsync(io.d_out, dX, (size_t)M * d * sizeof(float), hipMemcpyDeviceToDevice, c.st);
      continue;
    }
    // ---- embeddings: sum over the chunk's steps ------------------------------------------------------------
    if (d % 4 == 0) hipLaunchKernelGGL(assemble_bwd4_kernel, grid1d((size_t)B * N * d / 4), dim3(256), 0, c.st, g, d, nt_steps, dX,
                       c.at(c.pl.dEx), c.at(c.pl.dEy), P, gr->theta_tokens ? gr->theta_tokens : c.at(c.pl.dTmp));
    else
    hipLaunchKernelGGL(assemble_bwd_kernel, grid1d((size_t)B * N * d), dim3(256), 0, c.st, g, d, nt_steps, dX,
                       c.at(c.pl.dEx), c.at(c.pl.dEy), P, gr->theta_tokens ? gr->theta_tokens : c.at(c.pl.dTmp));
    CHECK_LAUNCH();
  }

  // ---- point embedders backward (model/embedder.py:47-57) -------------------------------------------------
  if (do_emb) {
    float *dEx = c.at(c.pl.dEx), *dEy = c.at(c.pl.dEy), *dH = c.at(c.pl.dHid);
    TRY(gemm_dw(c, dEx, d, EHx, F, gr->x_w2, gr->x_b2, rows_x, d, F));
    TRY(gemm_dx(c, dEx, d, m->x_w2, d, F, dH, F, rows_x, false, EHx));
    if (F <= 128 && 1024 % F == 0) hipLaunchKernelGGL(embed_first_bwd_groups_kernel, dim3((rows_x + 255) / 256), dim3(std::min(1024, 8 * F)), 0, c.st, xs, P + n_td, B, m->dim_x, F, dH, gr->x_w1, gr->x_b1, 256);
    else
    hipLaunchKernelGGL(embed_first_bwd_kernel, dim3((rows_x + 255) / 256), dim3(128), 0, c.st, xs, P + n_td, B,
                       m->dim_x, F, dH, gr->x_w1, gr->x_b1, 256);
    CHECK_LAUNCH();
    TRY(gemm_dw(c, dEy, d, EHy, F, gr->y_w2, gr->y_b2, rows_y, d, F));
    TRY(gemm_dx(c, dEy, d, m->y_w2, d, F, dH, F, rows_y, false, EHy));
    if (F <= 128 && 1024 % F == 0) hipLaunchKernelGGL(embed_first_bwd_groups_kernel, dim3((rows_y + 255) / 256), dim3(std::min(1024, 8 * F)), 0, c.st, ys, P, B, m->dim_y, F, dH, gr->y_w1, gr->y_b1, 256);
    else
    hipLaunchKernelGGL(embed_first_bwd_kernel, dim3((rows_y + 255) / 256), dim3(128), 0, c.st, ys, P, B, m->dim_y,
                       F, dH, gr->y_w1, gr->y_b1, 256);
    CHECK_LAUNCH();
  }
  return ALINE_OK;
}

extern "C" int aline_rollout_backward_ex(const aline_model *m, const aline_rollout *r, const float *g_logp,
                                         const float *g_ll, const float *g_pm, const float *g_ps,
                                         const float *g_pw, const aline_grads *gr, int t_chunk, void *ws,
                                         size_t ws_bytes, void *stream) {
  return backward_impl(m, r, g_logp, g_ll, g_pm, g_ps, g_pw, gr, t_chunk, ws, ws_bytes, stream, StageIO{ST_ALL, nullptr, nullptr, nullptr, nullptr});
}

// ---- stage backward entry points (one step, T = 1; include/aline_hip.h) ------------------------------------------
extern "C" int aline_head_backward(const aline_model *m, const aline_rollout *r, const float *z, const float *g_logp,
                                   const float *g_pm, const float *g_ps, const float *g_pw, const aline_grads *gr,
                                   float *dz, void *ws, size_t ws_bytes, void *stream) {
  return backward_impl(m, r, g_logp, nullptr, g_pm, g_ps, g_pw, gr, 1, ws, ws_bytes, stream, StageIO{ST_HEAD, nullptr, z, nullptr, dz});
}
extern "C" int aline_encoder_backward(const aline_model *m, const aline_rollout *r, const float *x_in, const float *dz,
                                      const aline_grads *gr, float *dx, void *ws, size_t ws_bytes, void *stream) {
  return backward_impl(m, r, nullptr, nullptr, nullptr, nullptr, nullptr, gr, 1, ws, ws_bytes, stream, StageIO{ST_ENC, x_in, nullptr, dz, dx});
}
extern "C" int aline_embed_backward(const aline_model *m, const aline_rollout *r, const float *dx, const aline_grads *gr,
                                    void *ws, size_t ws_bytes, void *stream) {
  return backward_impl(m, r, nullptr, nullptr, nullptr, nullptr, nullptr, gr, 1, ws, ws_bytes, stream, StageIO{ST_EMBED, nullptr, nullptr, dx, nullptr});
}

extern "C" int aline_cholesky_upper(float *A, int n, int batch, int32_t *info, void *stream) {
  if (!A || n < 1 || batch < 1 || n > 8192) return ALINE_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t smem = ((size_t)n * CHOL_NB + CHOL_NB + 4) * sizeof(float);
  if (n <= 256) hipLaunchKernelGGL(cholesky_upper_kernel<1>, dim3(batch), dim3(256), smem, st, A, n, info);
  else if (n <= 512) hipLaunchKernelGGL(cholesky_upper_kernel<2>, dim3(batch), dim3(256), smem, st, A, n, info);
  else if (n <= 1024) hipLaunchKernelGGL(cholesky_upper_kernel<4>, dim3(batch), dim3(256), smem, st, A, n, info);
  else hipLaunchKernelGGL(cholesky_upper_rowwise_kernel, dim3(batch), dim3(256), (size_t)n * sizeof(float), st, A, n, info);
  CHECK_LAUNCH();
  return ALINE_OK;
}
