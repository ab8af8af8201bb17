/* aline_hip.h -- C ABI of the MI355X-native ALINE hot path (libaline_hip.so).
 *
 * Drop-in boundary: the reference has no FFI; the seam is the three nn.Modules hydra instantiates
 * (config/embedder|encoder|head/NAME.yaml -> train_aline.py:246-249) and the calls
 * `model.forward(batch)` (train_aline.py:84, utils/eval.py:28), `compute_ll` (train_aline.py:92),
 * `Task.update_batch` (train_aline.py:88) and `EIGStepLoss` (utils/eval.py:56-74).  Every entry
 * point below names the reference function it replaces.  Plain pointers and sizes only:
 *   - all tensors are device pointers to contiguous row-major fp32 (int64 for indices,
 *     int32 for roles, uint8 for masks) unless stated otherwise;
 *   - weights are read in PyTorch's own layout ([out, in] row-major) -- the caller repacks nothing;
 *   - no allocation inside: the caller passes a workspace (size from *_workspace_bytes), so every
 *     call is legal inside hipGraph capture;
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work, never synchronise;
 *   - return 0 on success, a negative ALINE_E* code otherwise; nothing throws across the ABI;
 *   - re-entrant; the library never reads the environment.  The only mutable process-wide state is the diagnostic
 *     word of aline_debug_set_flags (0 = normal operation, which nothing but tests / A-B measurements changes) and its
 *     two integer knobs, plus a sticky f16-range status word that aline_f16_range_status reads and clears.
 */
#ifndef ALINE_HIP_H
#define ALINE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the entry points declared here are exported. */
#pragma GCC visibility push(default)

#define ALINE_ABI_VERSION 5
#define ALINE_MAX_LAYERS 8
#define ALINE_MAX_COMPONENTS 16
#define ALINE_MAX_POINTS 4096     /* P = n_ctx0 + n_query0 of a rollout / n_ctx + n_query of a step (README.md:45,50 evaluate at n_query = 2000) */

enum { ALINE_OK = 0, ALINE_EINVAL = -1, ALINE_EUNSUPPORTED = -2, ALINE_EWORKSPACE = -3,
       ALINE_ELAUNCH = -4,
       ALINE_ERANGE = -5 };  /* an F16X3 operand left f16's range (see aline_f16_range_status) */

/* embedding_type of model/embedder.py:24 */
enum { ALINE_EMB_DATA = 0, ALINE_EMB_THETA = 1, ALINE_EMB_MIX = 2 };

/* arithmetic of the matrix products (accumulation, LayerNorm, softmax, log-likelihoods are
 * always fp32):  F32 = exact fp32 MFMA (v_mfma_f32_16x16x4_f32), the reference-precision mode;
 * BF16 = one bf16 MFMA pass; BF16X3 = split-bf16 (hi*hi + hi*lo + lo*hi), ~2^-16 relative error at
 * 3 bf16 MFMA passes; F16X3 = the same 3-term split in f16 (11 + 11 significant bits, dropped term
 * <= 2^-24): reference-grade results (posterior log-likelihood within 1e-4 of the fp32 reference on
 * every fixture, like F32) at 3 passes of the f16/bf16 matrix pipe -- the parity mode of the wide
 * (d = 256) path and of the generic GEMMs.  Operands of F16X3 products must be below 65504 in magnitude (weights are
 * pre-scaled by 2^8: |w| < 255): every F16X3 kernel checks the operands it splits and raises a sticky device flag in the
 * workspace when one is non-finite or out of range -- aline_f16_range_status() reads it (a host-synchronising call made by
 * the caller after the work, never by the library); the Python mirror re-runs such a call in F32 and warns. */
enum { ALINE_PREC_F32 = 0, ALINE_PREC_BF16 = 1, ALINE_PREC_BF16X3 = 2, ALINE_PREC_F16X3 = 3 };

/* design selection of model/head.py:350-358 */
enum { ALINE_SELECT_ARGMAX = 0,   /* eval: max -> log            (head.py:355-358) */
       ALINE_SELECT_SAMPLE = 1,   /* train: inverse-CDF sample of Categorical(zt) from uniform[b] */
       ALINE_SELECT_FORCED = 2 }; /* teacher forcing: idx given, log_prob = Categorical.log_prob */

/* Model hyper-parameters + weight pointers, keyed exactly like the reference state_dict
 * (SURVEY.md 8-b.6).  Mirrors the constructor kwargs of Embedder (model/embedder.py:17-26),
 * Encoder (model/encoder.py:56-63) and OutputHead (model/head.py:275-287). */
typedef struct aline_model {
  int32_t dim_x, dim_y, d, F, H, L, C;
  int32_t n_theta;          /* n_target_theta: number of learnable theta tokens (0 in data mode) */
  int32_t embedding_type;   /* ALINE_EMB_* */
  int32_t time_token;       /* head.py:24-25: acquisition MLP input is d+1 wide */
  float std_min;            /* head.py:176 */
  int32_t precision;        /* ALINE_PREC_* */
  /* embedder.{x,y}_embedder.{0,2}.{weight,bias}, embedder.theta_tokens */
  const float *x_w1, *x_b1, *x_w2, *x_b2;
  const float *y_w1, *y_b1, *y_w2, *y_b2;
  const float *theta_tokens;                       /* [n_theta, d] or NULL */
  /* encoder.encoder.layers.{l}.* */
  const float *in_proj_w[ALINE_MAX_LAYERS], *in_proj_b[ALINE_MAX_LAYERS];     /* [3d,d],[3d] */
  const float *out_proj_w[ALINE_MAX_LAYERS], *out_proj_b[ALINE_MAX_LAYERS];   /* [d,d],[d]  */
  const float *lin1_w[ALINE_MAX_LAYERS], *lin1_b[ALINE_MAX_LAYERS];           /* [F,d],[F]  */
  const float *lin2_w[ALINE_MAX_LAYERS], *lin2_b[ALINE_MAX_LAYERS];           /* [d,F],[d]  */
  const float *norm1_w[ALINE_MAX_LAYERS], *norm1_b[ALINE_MAX_LAYERS];
  const float *norm2_w[ALINE_MAX_LAYERS], *norm2_b[ALINE_MAX_LAYERS];
  /* head.acquisition_head.predictor.{0,2}.* */
  const float *acq_w1, *acq_b1, *acq_w2, *acq_b2;  /* [F,d(+1)],[F],[1,F],[1] */
  /* head.target_head.heads.{c}.{0,2}.* */
  const float *gmm_w1[ALINE_MAX_COMPONENTS], *gmm_b1[ALINE_MAX_COMPONENTS];   /* [F,d],[F] */
  const float *gmm_w2[ALINE_MAX_COMPONENTS], *gmm_b2[ALINE_MAX_COMPONENTS];   /* [3,F],[3] */
} aline_model;

/* One design step in the reference's own (shape-changing) batch layout: the AttrDict batch of
 * SURVEY.md 8-b.4.  Token order is context | query | target data | theta (model/embedder.py). */
typedef struct aline_step {
  int32_t B, n_ctx, n_query, n_target_data;     /* n_t = n_target_data + model.n_theta */
  const float *context_x, *context_y;           /* [B,n_ctx,dx], [B,n_ctx,dy] */
  const float *query_x;                         /* [B,n_query,dx] */
  const float *target_x;                        /* [B,n_target_data,dx] or NULL */
  const float *target_all;                      /* [B,n_t] values for compute_ll, or NULL */
  const uint8_t *target_mask;                   /* [n_t] (encoder.py:110) or NULL = all */
  const float *time_t;                          /* device scalar batch.t, or NULL */
  int32_t select_mode;                          /* ALINE_SELECT_* */
  const float *uniform;                         /* [B] in [0,1) for SAMPLE */
  const int64_t *forced_idx;                    /* [B] for FORCED */
  /* outputs (any may be NULL to skip) -- the AttrDict of model/head.py:384-392 */
  int64_t *idx;                                 /* [B,1]  design_out.idx (index into query list) */
  float *log_prob;                              /* [B]    design_out.log_prob */
  float *zt;                                    /* [B,n_query] design_out.zt */
  float *post_mean, *post_std, *post_weight;    /* [B,n_t,C]   posterior_out.* */
  float *postq_mean, *postq_std, *postq_weight; /* [B,n_query,C] posterior_out_query.* */
  float *target_ll;                             /* [B,n_t] compute_ll (utils/eval.py:200-207) */
  float *embedding;                             /* [B,N,d] Embedder.forward output */
  float *encoding;                              /* [B,N,d] Encoder.forward output */
} aline_step;

/* Whole T-step acquisition loop (train_aline.py:80-110 / utils/eval.py:24-30) on a shape-static
 * layout: all P = n_ctx0 + n_query0 candidate points stay in fixed slots, `role[b,p]` is 0 for a
 * remaining query and k>0 for the k-th context point (order of entry), so one step is
 * hipGraph-capturable.  Equivalent to the reference up to summation order because there are no
 * positional encodings and the mask depends only on the token's role (encoder.py:83-126). */
typedef struct aline_rollout {
  int32_t B, P, n_ctx0, n_target_data, T;
  const float *point_x, *point_y;               /* [B,P,dx], [B,P,dy]: ctx0 first, then queries */
  int32_t *role;                                /* [B,P] in/out; NULL-initialised by aline_rollout_init */
  const float *target_x;                        /* [B,n_target_data,dx] or NULL */
  const float *target_all;                      /* [B,n_t] */
  const uint8_t *target_mask;                   /* [n_t] or NULL */
  int32_t select_mode;
  const float *uniform;                         /* [T,B] */
  const int64_t *forced_idx;                    /* [B,T] index into the compacted query list */
  int32_t time_token_T;                         /* model.time_token: step t feeds t/T (train_aline.py:82); NEGATIVE: (|T| - t)/|T|, the
                                                   schedule of the reference's eval loop (utils/eval.py:24); 0: T = this->T */
  /* outputs */
  int64_t *idx;                                 /* [B,T] compacted index (reference convention) */
  int32_t *slot;                                /* [B,T] chosen slot p */
  float *log_prob;                              /* [B,T] */
  float *target_ll;                             /* [T,B,n_t] */
  float *zt;                                    /* [T,B,P - n_ctx0] zero padded, or NULL */
  float *post_mean, *post_std, *post_weight;    /* [T,B,n_t,C] or NULL */
  /* optional hipEvent_t pair recorded on `stream` right before / after the dominant kernel of
   * aline_rollout_forward (bench.py times that kernel with them); NULL = not recorded */
  void *ev_kernel_start, *ev_kernel_stop;
  /* posterior_out_query (model/head.py:366) of every step, by slot: [T,B,P,C]; the slots that are context points at
   * a step hold unspecified values there.  NULL = not computed (no caller in train_aline.py / utils/eval.py reads
   * it).  Served by the s3, x3 / x5 and generic paths (the exact-fp32 fused kernel does not: a request routes its rollouts to the
   * generic pipeline). */
  float *postq_mean, *postq_std, *postq_weight;
  /* Training rollouts: [2 L + 1][T * B * N * d] floats -- the encoder layers' inputs X_0 .. X_L, then their attention outputs
   * A_0 .. A_{L-1}, row (t * B + b) * N + token row -- written by aline_rollout_forward when non-NULL (s3 path; other paths leave it
   * alone) and read by aline_rollout_backward[_ex] INSTEAD of recomputing the layers of the rollout it is handed (fused small-width
   * backward; the caller passes the struct the forward ran with).  aline_rollout_saved_acts_bytes sizes it (0: this (m, r) cannot
   * use it).  NULL: the backward recomputes. */
  float *saved_acts;
  /* Which launch of the dominant kernel the ev_kernel_* pair brackets: 0 = the one of the last step (t = T - 1; the default),
   * k > 0 = the one of step t = k - 1 (bench.py walks all T steps for the AVERAGE launch duration its roofline figure uses). */
  int32_t ev_kernel_step;
} aline_rollout;

/* ABI / build info. */
int aline_abi_version(void);
const char *aline_error_string(int code);

/* --- Aline.forward (model/base.py:32-50) and its three stages ------------------------------ */
size_t aline_step_workspace_bytes(const aline_model *m, const aline_step *s);
/* Embedder.forward (model/embedder.py:67-95): writes s->embedding. */
int aline_embed_forward(const aline_model *m, const aline_step *s, void *ws, size_t ws_bytes,
                        void *stream);
/* Encoder.forward (model/encoder.py:128-141): x_in [B,N,d] -> s->encoding. */
int aline_encoder_forward(const aline_model *m, const aline_step *s, const float *x_in, void *ws,
                          size_t ws_bytes, void *stream);
/* OutputHead.forward (model/head.py:319-393) on z [B,N,d]. */
int aline_head_forward(const aline_model *m, const aline_step *s, const float *z, void *ws,
                       size_t ws_bytes, void *stream);
/* embedder -> encoder -> head in one call. */
int aline_step_forward(const aline_model *m, const aline_step *s, void *ws, size_t ws_bytes,
                       void *stream);

/* --- T-step loop on the static-slot layout -------------------------------------------------- */
size_t aline_rollout_workspace_bytes(const aline_model *m, const aline_rollout *r);
/* role <- initial ctx/query; caches the step-invariant point embeddings (x- and y-embedder of
 * every slot, model/embedder.py:47-57) in the workspace: the same `ws` must be passed to the steps. */
int aline_rollout_init(const aline_model *m, const aline_rollout *r, void *ws, size_t ws_bytes,
                       void *stream);
int aline_rollout_step(const aline_model *m, const aline_rollout *r, int t, void *ws,
                       size_t ws_bytes, void *stream);             /* forward + select + update */
int aline_rollout_forward(const aline_model *m, const aline_rollout *r, void *ws, size_t ws_bytes,
                          void *stream);                            /* init + T steps */
/* Which implementation aline_rollout_forward runs for (m, r): the generic per-stage pipeline or one of the fused
 * paths.  Negative: error code. */
enum { ALINE_PATH_GENERIC = 0,   /* stage kernels + GEMMs, any configuration */
       ALINE_PATH_FUSED = 1,     /* fused_rollout.h: d = 32, theta mode, F32, whole rollout in one launch */
       /* 2: the bf16 `wide` path of rounds 1-3 (NLL error 2e-2; removed in ABI 5: BF16 models run the generic pipeline) */
       ALINE_PATH_X3 = 3,        /* x3.h: d = 256, F16X3 (reference precision on the f16 matrix pipe) */
       ALINE_PATH_S3 = 4,        /* s3.h: d = 32, F16X3, any embedding mode, one launch per design step */
       ALINE_PATH_X5 = 5 };      /* x3.h, namespace x5: d = 512 / 8 heads of 64, F16X3 (the psychometric configuration's width) */
int aline_rollout_path(const aline_model *m, const aline_rollout *r);
size_t aline_rollout_saved_acts_bytes(const aline_model *m, const aline_rollout *r);   /* see aline_rollout.saved_acts */
/* Name (as rocprofv3 prints it, template arguments included) of the dominant kernel of that path for (m, r) -- the launch
 * the ev_kernel_start / ev_kernel_stop pair brackets -- written to buf; returns the ALINE_PATH_* value or a negative code. */
int aline_rollout_kernel_name(const aline_model *m, const aline_rollout *r, char *buf, size_t n);
/* Task.update_batch equivalent for callers that want the reference layout back
 * (tasks/base_task.py:133-154): context_x/y [B,n_ctx0+T,*] in order of entry, query remainder. */
int aline_rollout_export(const aline_rollout *r, int n_ctx, float *context_x, float *context_y,
                         float *query_x, float *query_y, int dim_x, int dim_y, void *stream);

/* --- objectives ----------------------------------------------------------------------------- */
/* compute_ll (utils/eval.py:200-207): value [rows], means/stds/weights [rows,C] -> out [rows]. */
int aline_compute_ll(const float *value, const float *means, const float *stds,
                     const float *weights, int64_t rows, int C, float *out, void *stream);

/* EIGStepLoss.step (loss/eig.py:174-193) with HiddenLocation.log_likelihood
 * (tasks/location_finding.py:110-164):  S[l,b] += log N(y[b]; log(base + sum_k 1/(m+|xi_b-th_lbk|^2)), noise)
 * theta [L1,B,K,D], xi [B,D], y [B], S [L1,B] in/out. */
int aline_eig_location_step(const float *theta, const float *xi, const float *y, float *S,
                            int64_t L1, int B, int K, int D, float noise_scale, float base_signal,
                            float max_signal, void *stream);
/* same with CESTask.log_likelihood (tasks/ces.py:169-210) and CensoredSigmoidNormal.log_prob
 * (distributions/censored_sigmoid_normal.py:47-86): theta [L1,B,5], xi [B,6], y [B].
 * nan_flag (device int, may be NULL) is set instead of raising (csn.py:83-84). */
int aline_eig_ces_step(const float *theta, const float *xi, const float *y, float *S, int64_t L1,
                       int B, float noise_scale, float epsilon, int32_t *nan_flag, void *stream);
/* EIGStepLoss.forward (loss/eig.py:195-209) + the bounds of utils/eval.py:77-78:
 * pce[b] = log(L+1) - (LSE_{l>=0} S - S[0]),  nmc[b] = log L - (LSE_{l>=1} S - S[0]). */
size_t aline_eig_finalize_workspace_bytes(int64_t L1, int B);
int aline_eig_finalize(const float *S, int64_t L1, int B, float *pce, float *nmc, void *ws,
                       size_t ws_bytes, void *stream);
/* All T steps of a design history in ONE pass over theta (compute_EIG_from_history, utils/eval.py:42-80, stepwise: the criterion of
 * loss/eig.py:174-209 applied to y[:, t], x[:, t] for t = 0 .. T - 1 on the same contrastive draw).  theta [L1, B, K, D] with row 0 the
 * true parameter of each episode, xi [B, T, D] (unnormalised designs in order of acquisition), y [B, T]; pce / nmc [B, T] receive the
 * sPCE / sNMC bounds after every step (either may be NULL).  Location-finding likelihood (tasks/location_finding.py:110-164). */
size_t aline_eig_history_workspace_bytes(int64_t L1, int B, int T);
int aline_eig_location_history(const float *theta, const float *xi, const float *y, int64_t L1, int B, int T, int K, int D,
                               float noise_scale, float base_signal, float max_signal, float *pce, float *nmc, void *ws,
                               size_t ws_bytes, void *stream);
/* The same for the CES likelihood (tasks/ces.py:96-115, :169-210; CensoredSigmoidNormal as aline_eig_ces_step): theta [L1, B, 5],
 * xi [B, T, 6], y [B, T].  ALINE_EUNSUPPORTED when T > 16 or the per-(episode, step) table (96 B T bytes) exceeds 64 KB: use the step
 * entry points then.  nan_flag as in aline_eig_ces_step. */
int aline_eig_ces_history(const float *theta, const float *xi, const float *y, int64_t L1, int B, int T, float noise_scale,
                          float epsilon, float *pce, float *nmc, int32_t *nan_flag, void *ws, size_t ws_bytes, void *stream);

/* --- task samplers (input generators) ---------------------------------------------- */
/* Batched in-place Cholesky A = U^T U of `batch` symmetric positive-definite [n,n] matrices (upper
 * factor; strict lower triangle zeroed), for GPTask.generate_gp_data (tasks/gaussian_process.py:391-415:
 * one torch.linalg.cholesky per episode in a Python loop).  info (device int, may be NULL) is set to 1
 * if a pivot is not positive. */
int aline_cholesky_upper(float *A, int n, int batch, int32_t *info, void *stream);

/* --- training: backward of the T-step objective ------------------------------------------------- */
/* Gradient buffers, one per weight of aline_model (same shapes), ACCUMULATED into (+=): the caller
 * zeroes them (optimizer.zero_grad(), train_aline.py:56) and owns them (param.grad storage). */
typedef struct aline_grads {
  float *x_w1, *x_b1, *x_w2, *x_b2, *y_w1, *y_b1, *y_w2, *y_b2, *theta_tokens;
  float *in_proj_w[ALINE_MAX_LAYERS], *in_proj_b[ALINE_MAX_LAYERS];
  float *out_proj_w[ALINE_MAX_LAYERS], *out_proj_b[ALINE_MAX_LAYERS];
  float *lin1_w[ALINE_MAX_LAYERS], *lin1_b[ALINE_MAX_LAYERS];
  float *lin2_w[ALINE_MAX_LAYERS], *lin2_b[ALINE_MAX_LAYERS];
  float *norm1_w[ALINE_MAX_LAYERS], *norm1_b[ALINE_MAX_LAYERS];
  float *norm2_w[ALINE_MAX_LAYERS], *norm2_b[ALINE_MAX_LAYERS];
  float *acq_w1, *acq_b1, *acq_w2, *acq_b2;
  float *gmm_w1[ALINE_MAX_COMPONENTS], *gmm_b1[ALINE_MAX_COMPONENTS];
  float *gmm_w2[ALINE_MAX_COMPONENTS], *gmm_b2[ALINE_MAX_COMPONENTS];
} aline_grads;

/* loss.backward() of train_aline.py:124-132 for a finished rollout.  The T steps are independent
 * given the designs (selection is discrete; rewards are detached, train_aline.py:116), so the caller
 * passes the two upstream gradients
 *     g_logp [B,T]      = dLoss / d design_out.log_prob[b,t]   (= -alpha R[b,t] / (B (T-1)), 0 at t=T-1)
 *     g_ll   [T,B,n_t]  = dLoss / d target_ll[t,b,j]           (= -1 / (T B n_t) for the plain NLL mean)
 * and gets dLoss/dW accumulated into `grads`.  Needs r->role (final roles) and r->slot from the
 * forward rollout.  Steps are processed `t_chunk` at a time (workspace ~ t_chunk * B * N * (14 d + 4 F)
 * floats).  fp32 only. */
size_t aline_rollout_backward_workspace_bytes(const aline_model *m, const aline_rollout *r, int t_chunk);
int aline_rollout_backward(const aline_model *m, const aline_rollout *r, const float *g_logp,
                           const float *g_ll, const aline_grads *grads, int t_chunk, void *ws,
                           size_t ws_bytes, void *stream);
/* Same, with (optional) upstream gradients of the GMM parameters themselves, [T,B,n_t,C] each:
 * dLoss/d posterior_out.mixture_{means,stds,weights} as autograd delivers them when the caller computes
 * the log-likelihood with its own differentiable code (utils/eval.py:200-207).  g_ll may then be NULL.
 * This is what `loss.backward()` of the reference's own training loop needs from Aline.forward. */
int aline_rollout_backward_ex(const aline_model *m, const aline_rollout *r, const float *g_logp,
                              const float *g_ll, const float *g_post_mean, const float *g_post_std,
                              const float *g_post_weight, const aline_grads *grads, int t_chunk, void *ws,
                              size_t ws_bytes, void *stream);

/* Stage backward entry points: the backward of ONE stand-alone stage of ONE step, so that a `_target_`-only swap of
 * model.embedder.Embedder / model.encoder.Encoder / model.head.OutputHead trains under the reference's own composition
 * `Aline.forward = head(batch, encoder(batch, embedder(batch)))` (model/base.py:47-50, train_aline.py:246-249) and its
 * `loss.backward()` (train_aline.py:128).  The step is described as a T = 1 rollout (role[b, p] = 1..n_ctx for the context
 * points, 0 for the queries; slot[b] = chosen point; only the members of the stage's own model fields are read).  Token rows
 * [B * N, d] in the reference order ctx | query | target data | theta tokens.  Gradients are ACCUMULATED into `grads` (members
 * of other stages may be NULL).  Workspace: aline_rollout_backward_workspace_bytes(m, r, 1).  fp32 only.
 *   aline_head_backward     OutputHead.forward (model/head.py:319-393): z, g_logp [B], g_post_* [B, n_t, C] (NULL = zero)
 *                           -> head gradients, dz [B * N, d]
 *   aline_encoder_backward  Encoder.forward (model/encoder.py:128-141): x_in (the embeddings), dz -> encoder gradients, dx
 *   aline_embed_backward    Embedder.forward (model/embedder.py:67-214): dx -> embedder gradients                       */
int aline_head_backward(const aline_model *m, const aline_rollout *r, const float *z, const float *g_logp,
                        const float *g_post_mean, const float *g_post_std, const float *g_post_weight,
                        const aline_grads *grads, float *dz, void *ws, size_t ws_bytes, void *stream);
int aline_encoder_backward(const aline_model *m, const aline_rollout *r, const float *x_in, const float *dz,
                           const aline_grads *grads, float *dx, void *ws, size_t ws_bytes, void *stream);
int aline_embed_backward(const aline_model *m, const aline_rollout *r, const float *dx, const aline_grads *grads,
                         void *ws, size_t ws_bytes, void *stream);

/* --- F16X3 operand range guard ------------------------------------------------------------------ */
/* Byte offset, inside any step / rollout workspace of this library, of a 32-bit status word the F16X3 kernels OR into:
 * bit 0 = an activation operand was non-finite or >= 65504 in magnitude when it was split into f16 halves, bit 1 = a weight
 * (after the 2^8 pre-scale).  aline_*_forward clear it when they start.  aline_f16_range_status copies it to the host
 * (hipMemcpyAsync + stream synchronise: call it after the work, outside graph capture) and returns the word (>= 0) or a
 * negative error code. */
size_t aline_f16_range_offset(void);
int aline_f16_range_status(const void *ws, void *stream);

/* --- diagnostics ----------------------------------------------------------------------------- */
/* Process-wide diagnostic word, 0 in normal operation.  Tests and A/B measurements use it to force the path a fused kernel
 * replaces (every fused kernel is cross-checked against that path) or to opt into an experiment.  Returns the old word. */
enum { ALINE_DBG_DISABLE_FUSED = 1u << 0,      /* fused::rollout_f32_kernel off -> generic pipeline */
       ALINE_DBG_DISABLE_X3 = 1u << 2,         /* x3 path off -> generic pipeline on the F16X3 GEMM policy */
       ALINE_DBG_DISABLE_S3 = 1u << 3,         /* s3 path off */
       ALINE_DBG_NO_LAYER_TAIL = 1u << 5,      /* generic d = 32 pipeline on per-op kernels */
       ALINE_DBG_FULL_QKV = 1u << 6,           /* generic pipeline: K / V of every row (as the reference) */
       ALINE_DBG_VALU_ATTENTION = 1u << 7,     /* generic pipeline: fp32 VALU attention instead of attn3 */
       ALINE_DBG_S3_GENERIC_EMBED = 1u << 8,   /* s3: point embedders on the generic kernels */
       ALINE_DBG_CES_GENERIC = 1u << 9,        /* CES likelihood: powf formulation */
       ALINE_DBG_FUSED_STAMPS = 1u << 10,      /* in-kernel phase stamps (diagnostic instantiations) */
       ALINE_DBG_NO_BWD_IMAGE_RECOMPUTE = 1u << 11,  /* per-op backward at d = 256 / 512: forward recompute on the generic kernels instead of x3 / x5 layer_save_kernel */
       ALINE_DBG_NO_BWD_KV_SPARSE = 1u << 12,  /* per-op backward: the in-projection's gradient products over all rows x all 3 d columns (dK / dV of the non-key rows are zeros) */
       ALINE_DBG_SELECT_WORKGROUP = 1u << 13,  /* design selection: the workgroup-per-episode kernel also where one wave per episode would do */
       ALINE_DBG_S3_SELECT_KERNEL = 1u << 14,  /* s3 path: the design selection as a launch of its own (acq_select_wave_kernel) instead of inside the step kernel */
       /* backward: switch ONE fused kernel back to the per-op pipeline it replaces */
       ALINE_DBG_NO_BWD_TAIL = 1u << 16, ALINE_DBG_NO_BWD_ATTN_BLOCK = 1u << 17, ALINE_DBG_NO_BWD_ACQ = 1u << 18,
       ALINE_DBG_NO_BWD_LAYER_FWD = 1u << 19, ALINE_DBG_NO_BWD_LAYER_FWD_FLAT = 1u << 20,
       ALINE_DBG_NO_BWD_GMM_FUSED = 1u << 21, ALINE_DBG_NO_BWD_GMM128 = 1u << 22, ALINE_DBG_NO_BWD_GMM_BATCHED = 1u << 23,
       ALINE_DBG_NO_BWD_ATTN_MFMA = 1u << 24,
       ALINE_DBG_NO_BWD_DW_WALK = 1u << 25,    /* weight-gradient products: the masked loop with per-row divisions for every row */
       ALINE_DBG_NO_BWD_GMM_WIDE = 1u << 26,   /* GMM head backward at F > 128: the per-row-atomics kernel instead of gmm_bwd_wide_kernel */
       ALINE_DBG_NO_BWD_SAVED_ACTS = 1u << 27,  /* backward: recompute the layers even when aline_rollout.saved_acts is given */
       ALINE_DBG_BWD_RECOMPUTE_F32 = 1u << 28,  /* per-op backward of an F16X3 model: forward recompute GEMMs in exact fp32 too */
       ALINE_DBG_BWD_DW_TK2 = 1u << 29,         /* weight-gradient products: 32 (not 64) columns of the narrow operand per workgroup */
       ALINE_DBG_BWD_GRAD_F32 = 1u << 30 };     /* per-op backward of an F16X3 model: gradient products (dX, dW) in exact fp32 instead of the scaled f16 split */
uint32_t aline_debug_set_flags(uint32_t flags);
uint32_t aline_debug_get_flags(void);
/* Integer knobs of the same kind (0 = automatic): launch shape of the s3 step kernel, precision of the backward GEMMs. */
enum { ALINE_DBG_S3_WAVES = 0, ALINE_DBG_S3_EPW = 1, ALINE_DBG_BWD_PREC = 2, ALINE_DBG_NPARAMS = 3 };
int aline_debug_set_param(int key, int value);
int aline_debug_get_param(int key);      /* the knob's current value (0 for an unknown key) */
/* Byte offset, inside a rollout workspace, of the per-phase cycle stamps the fused rollout kernel
 * writes under ALINE_DBG_FUSED_STAMPS (diagnostic instantiation only). */
size_t aline_debug_stamps_offset(const aline_model *m, const aline_rollout *r);
/* Same for the buffer the stamped diagnostic builds of the x3 / s3 kernels (tools/x3_stamps.py, tools/s3_stamps.py) read back. */
size_t aline_debug_xraw_offset(const aline_model *m, const aline_rollout *r);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* ALINE_HIP_H */
