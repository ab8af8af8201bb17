// Host side of the split-f16 tile-image paths (x3_impl.h): eligibility and the per-rollout launch sequence.  Included by
// aline_hip.hip once per width (namespace X3_NS = x3: d = 256, x5: d = 512), after Ctx / Plan / the shared launch helpers.
namespace X3_NS {

template <int NOUT>
static int launch_head(const Ctx &c, HeadArgs a) {
  const size_t smem = (size_t)NBUF * CHUNK_BYTES + (size_t)head_lds_params(a.F) * 4;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&head_kernel<NOUT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  a.ngroups = (int)((a.ntiles + WAVES - 1) / WAVES);
  hipLaunchKernelGGL(head_kernel<NOUT>, dim3((unsigned)std::min(a.ngroups, device_cus())), dim3(THREADS), smem, c.st, a);
  CHECK_LAUNCH();
  return ALINE_OK;
}

// weights -> split-f16 fragment pairs
static int pack_weights(const aline_model &mm, unsigned *img, unsigned *range_flag, hipStream_t st) {
  const aline_model *m = &mm;
  PackArgs pa{};
  pa.L = m->L; pa.F = m->F; pa.C = m->C;
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
  pa.out = img; pa.range_flag = range_flag; pa.time_token = m->time_token ? 1 : 0;
  hipLaunchKernelGGL(pack_kernel, dim3(2048), dim3(256), 0, st, pa);
  CHECK_LAUNCH();
  return ALINE_OK;
}

// ---- forward recompute of the training backward (round 4) -----------------------------------------------------------------
// The per-op backward of a d = 256 / 512 model needs, per layer, the attention output, both LayerNorm inputs and outputs and the
// hidden units of every (step, episode) INSTANCE of a chunk.  It used to recompute them with the generic kernels (four GEMMs, the
// attention kernel and two LayerNorm kernels per layer: 15 % of the d = 256 step); this runs the rollout's own layer kernel over
// the instances instead (layer_save_kernel: the same arithmetic, plus fp32 stores of those rows).  The buffers are float offsets
// into the backward workspace.
struct RecomputePlan { size_t img, in, a, b, kv, keys, kcnt, kx, kpos; };
static bool recompute_model_ok(const aline_model &m) {
  return m.precision == ALINE_PREC_F16X3 && m.d == D && m.H == H && m.F % 32 == 0 &&
         (size_t)NBUF * CHUNK_BYTES + (size_t)layer_params(m.F) * 4 <= 160 * 1024;
}
template <class Take>
static RecomputePlan recompute_plan(const aline_model &m, long I, int N, Take take) {
  RecomputePlan p{};
  const long tpe = (N + 15) / 16;
  const size_t im = (size_t)img_pieces(I * tpe) * 4;
  p.img = take((size_t)image_words(m.L, m.F, m.C));
  p.in = take(im); p.a = take(im); p.b = take(im);
  p.kv = take((size_t)I * KV_EP * 4);
  p.keys = take((size_t)I * WNK); p.kcnt = take((size_t)I * 2);
  p.kx = take((size_t)img_pieces(I * (WNK / 16)) * 4);
  p.kpos = take((size_t)I * tpe * 16 / 2 + 1);
  return p;
}
struct RecomputeRows { float *Q[ALINE_MAX_LAYERS]; long q_ld; float *A[ALINE_MAX_LAYERS], *U1[ALINE_MAX_LAYERS], *X1[ALINE_MAX_LAYERS], *Hid[ALINE_MAX_LAYERS], *U2[ALINE_MAX_LAYERS], *Y[ALINE_MAX_LAYERS]; };
// g: the instance geometry of the chunk (Geo.inst_B > 0); max_keys <= WNK
static int recompute(const aline_model *m, const Geo &g, int max_keys, const float *Ex, const float *Ey, int ey_rows, float *ws, const RecomputePlan &pl,
                     const RecomputeRows &rows, hipStream_t st) {
  const int I = g.B, N = g.N, F = m->F, tpe = (N + 15) / 16;
  const long tiles = (long)I * tpe;
  unsigned *img = reinterpret_cast<unsigned *>(ws + pl.img);
  u32x4 *XIN = reinterpret_cast<u32x4 *>(ws + pl.in), *XA = reinterpret_cast<u32x4 *>(ws + pl.a), *XB = reinterpret_cast<u32x4 *>(ws + pl.b);
  u32x4 *KV = reinterpret_cast<u32x4 *>(ws + pl.kv), *KX = reinterpret_cast<u32x4 *>(ws + pl.kx);
  int *keyrow = reinterpret_cast<int *>(ws + pl.keys), *kcnt = reinterpret_cast<int *>(ws + pl.kcnt);
  short *keypos = reinterpret_cast<short *>(ws + pl.kpos);
  AsmArgs aa{};
  aa.g = g; aa.tpe = tpe; aa.Ex = Ex; aa.Ey = Ey; aa.ey_rows = ey_rows; aa.theta_tokens = m->theta_tokens; aa.X = XIN; aa.range_flag = nullptr;
  hipLaunchKernelGGL(assemble_kernel, grid1d((size_t)tiles * NKS * 64), dim3(256), 0, st, aa);
  CHECK_LAUNCH();
  hipLaunchKernelGGL(keys_kernel, dim3(I), dim3(256), 0, st, g, tpe, keyrow, kcnt, keypos, XIN, KX);
  CHECK_LAUNCH();
  const long lw = layer_words(F);
  const size_t smem_layer = (size_t)NBUF * CHUNK_BYTES + (size_t)layer_params(F) * 4;
  const size_t smem_kv = (size_t)NBUF * CHUNK_BYTES + 2 * D * 4;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&layer_save_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_layer);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kv_all_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_kv);
  const int cus = device_cus();
  const int nkt2 = 2 * ((std::min(max_keys, WNK) + 31) / 32);
  const u32x4 *xin = XIN;
  for (int l = 0; l < m->L; ++l) {
    u32x4 *xout = (l & 1) ? XB : XA;
    KvArgs ka{};
    ka.g = g; ka.tpe = tpe; ka.nkt2 = nkt2; ka.X = KX; ka.img = img + l * lw; ka.F = F; ka.keyrow = keyrow; ka.kcnt = kcnt; ka.KV = KV;
    ka.ngroups = (int)(((long)I * nkt2 + WAVES - 1) / WAVES);
    hipLaunchKernelGGL(kv_all_kernel, dim3((unsigned)std::min(ka.ngroups, cus)), dim3(THREADS), smem_kv, st, ka);
    CHECK_LAUNCH();
    LayerArgs la{};
    la.g = g; la.tpe = tpe; la.ngroups = (int)((tiles + WAVES - 1) / WAVES);
    la.XIN = xin; la.XOUT = xout; la.img = img + l * lw; la.F = F; la.KV = KV; la.kcnt = kcnt; la.range_flag = nullptr; la.keypos = keypos; la.KXout = KX;
    la.svQ = rows.Q[l]; la.svQ_ld = rows.q_ld; la.svA = rows.A[l]; la.svU1 = rows.U1[l]; la.svX1 = rows.X1[l]; la.svHid = rows.Hid[l]; la.svU2 = rows.U2[l]; la.svY = rows.Y[l];
    hipLaunchKernelGGL(layer_save_kernel, dim3((unsigned)std::min(la.ngroups, cus)), dim3(THREADS), smem_layer, st, la);
    CHECK_LAUNCH();
    xin = xout;
  }
  return ALINE_OK;
}

// Eligibility of the tile-image path of this width (x3: d = 256 / 8 heads of 32, x5: d = 512 / 8 heads of 64) at reference
// precision -- every product a 3-term f16 split on the matrix pipe.
static bool eligible(const aline_model &m, const aline_rollout &r) {
  if (dbg(ALINE_DBG_DISABLE_X3)) return false;
  if (m.precision != ALINE_PREC_F16X3 || m.d != D || m.H != H || m.F % 32) return false;
  if (r.n_ctx0 + r.T - 1 + r.n_target_data + m.n_theta > WNK) return false;
  if ((size_t)NBUF * CHUNK_BYTES + (size_t)std::max(layer_params(m.F), head_lds_params(m.F)) * 4 > 160 * 1024) return false;
  return true;
}

static int rollout(const aline_model *m, const aline_rollout *r, void *ws, size_t ws_bytes, void *stream) {
  Ctx c;
  TRY(rollout_ctx(m, r, ws, ws_bytes, stream, c));
  TRY(check_select(r->select_mode, r->uniform, r->forced_idx));
  TRY(c.clear_flag());
  const int n_t = c.g.n_td + c.g.n_th, N = c.g.N, F = m->F, tpe = (N + 15) / 16, NP = 16 * tpe;
  const long tiles = (long)r->B * tpe;

  hipLaunchKernelGGL(role_init_kernel, grid1d((size_t)r->B * r->P), dim3(256), 0, c.st, r->role, r->B, r->P, r->n_ctx0);
  CHECK_LAUNCH();
  unsigned *img = reinterpret_cast<unsigned *>(c.at(c.pl.xImg));
  TRY(pack_weights(*m, img, c.flag(), c.st));      // weights -> split-f16 fragment pairs (once per rollout)
  // step-invariant point embeddings (fp32 rows; the generic GEMM runs the same 3-term f16 split)
  {
    Src3 xs{{r->point_x, r->target_x, nullptr}, {r->P, r->n_target_data, 0}};
    TRY(do_embed_points(c, xs, r->point_y, r->P));
  }
  u32x4 *XIN = reinterpret_cast<u32x4 *>(c.at(c.pl.xIn)), *XA = reinterpret_cast<u32x4 *>(c.at(c.pl.xA)), *XB = reinterpret_cast<u32x4 *>(c.at(c.pl.xB));
  u32x4 *KV = reinterpret_cast<u32x4 *>(c.at(c.pl.xKV)), *Zimg = reinterpret_cast<u32x4 *>(c.at(c.pl.xZimg));
  int *keyrow = reinterpret_cast<int *>(c.at(c.pl.xKeys)), *kcnt = reinterpret_cast<int *>(c.at(c.pl.xKcnt));
  short *keypos = reinterpret_cast<short *>(c.at(c.pl.xKpos));
  u32x4 *KX = reinterpret_cast<u32x4 *>(c.at(c.pl.xKX));
  float *logits = c.at(c.pl.xLog);
  AsmArgs aa{};
  aa.g = c.g; aa.tpe = tpe; aa.Ex = c.at(c.pl.Ex); aa.Ey = c.at(c.pl.Ey); aa.ey_rows = r->P; aa.theta_tokens = m->theta_tokens; aa.X = XIN; aa.range_flag = c.flag();
  hipLaunchKernelGGL(assemble_kernel, grid1d((size_t)tiles * NKS * 64), dim3(256), 0, c.st, aa);
  CHECK_LAUNCH();
  const bool want_gmm = r->post_mean || r->post_std || r->post_weight || r->target_ll;
  const long lw = layer_words(F);
  const size_t smem_layer = (size_t)NBUF * CHUNK_BYTES + (size_t)layer_params(F) * 4;
  const size_t smem_kv = (size_t)NBUF * CHUNK_BYTES + 2 * D * 4;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&layer_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_layer);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&layer_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_layer);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kv_all_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_kv);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kv_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_kv);
  const int cus = device_cus();
  for (int t = 0; t < r->T; ++t) {
    c.g.n_ctx = r->n_ctx0 + t;
    hipLaunchKernelGGL(keys_kernel, dim3(r->B), dim3(256), 0, c.st, c.g, tpe, keyrow, kcnt, keypos, XIN, KX);
    CHECK_LAUNCH();
    const int nkeys = r->n_ctx0 + t + n_t;                      // upper bound of an episode's key count at this step
    const int nkt2 = 2 * ((std::min(nkeys, WNK) + 31) / 32);
    const u32x4 *xin = XIN;
    for (int l = 0; l < m->L; ++l) {
      u32x4 *xout = (l & 1) ? XB : XA;
      KvArgs ka{};
      ka.g = c.g; ka.tpe = tpe; ka.nkt2 = nkt2; ka.X = KX; ka.img = img + l * lw; ka.F = F; ka.keyrow = keyrow; ka.kcnt = kcnt; ka.KV = KV;
      ka.ngroups = (int)(((long)r->B * nkt2 + WAVES - 1) / WAVES);
      // (groups of key tiles that do not fill the chip: one job -- K | V x channel half -- per workgroup, so that every CU streams)
      constexpr int NJ = 2 * CPK;
      if ((ka.ngroups * NJ + cus - 1) / cus < NJ * ((ka.ngroups + cus - 1) / cus))      // fewer chunk passes per CU when split
        hipLaunchKernelGGL(kv_split_kernel, dim3((unsigned)(std::min(ka.ngroups, std::max(1, cus / NJ)) * NJ)), dim3(THREADS), smem_kv, c.st, ka);
      else
        hipLaunchKernelGGL(kv_all_kernel, dim3((unsigned)std::min(ka.ngroups, cus)), dim3(THREADS), smem_kv, c.st, ka);
      CHECK_LAUNCH();
      LayerArgs la{};
      la.g = c.g; la.tpe = tpe; la.ngroups = (int)((tiles + WAVES - 1) / WAVES);
      la.XIN = xin; la.XOUT = xout; la.img = img + l * lw; la.F = F; la.KV = KV; la.kcnt = kcnt; la.range_flag = c.flag(); la.keypos = keypos; la.KXout = KX;
      const bool last = l == m->L - 1;
      la.zimg = (last && want_gmm) ? Zimg : nullptr; la.zrow0 = (long)t * r->B * n_t;
#ifdef X3_STAMPS
      la.stamps = want_gmm ? nullptr : reinterpret_cast<unsigned long long *>(c.at(c.pl.xRaw));
#endif
      const bool timed = (t == (r->ev_kernel_step > 0 ? r->ev_kernel_step - 1 : r->T - 1) && last);      // bench.py times this launch of the dominant kernel
      if (timed && r->ev_kernel_start) (void)hipEventRecord(static_cast<hipEvent_t>(r->ev_kernel_start), c.st);
      if (last) hipLaunchKernelGGL(layer_kernel<true>, dim3((unsigned)std::min(la.ngroups, cus)), dim3(THREADS), smem_layer, c.st, la);
      else hipLaunchKernelGGL(layer_kernel<false>, dim3((unsigned)std::min(la.ngroups, cus)), dim3(THREADS), smem_layer, c.st, la);
      if (timed && r->ev_kernel_stop) (void)hipEventRecord(static_cast<hipEvent_t>(r->ev_kernel_stop), c.st);
      CHECK_LAUNCH();
      xin = xout;
    }
    {   // acquisition logits of every token row (model/head.py:27-33); the selection reads the candidate slots
      HeadArgs ha{};
      ha.X = xin; ha.ntiles = tiles; ha.M = tiles * 16; ha.img = img + (long)m->L * lw; ha.F = F;
      ha.out = logits; ha.out_stride = 1; ha.out_off = 0;
      ha.tau = m->time_token ? step_time_token(*r, t) : 0.f;
      TRY(launch_head<1>(c, ha));
    }
    if (wants_postq(*r)) {
      // posterior_out_query of this step (model/head.py:366), by slot: the C GMM heads over the WHOLE output image of the last layer
      // (every token row of every episode; the finish kernel keeps the P point rows), before the selection moves a role
      float *rawq = c.at(c.pl.xRawQ);
      for (int k = 0; k < m->C; ++k) {
        HeadArgs hq{};
        hq.X = xin; hq.ntiles = tiles; hq.M = tiles * 16; hq.img = img + (long)m->L * lw + (long)(1 + k) * head_words(F); hq.F = F;
        hq.out = rawq; hq.out_stride = kRawStride; hq.out_off = 3 * k;
        TRY(launch_head<3>(c, hq));
      }
      img::GmmRawArgs gq{};
      gq.raw = rawq; gq.raw_stride = kRawStride; gq.rows = tiles * 16; gq.C = m->C; gq.std_min = m->std_min;
      gq.mean = r->postq_mean; gq.sd = r->postq_std; gq.wgt = r->postq_weight;
      gq.map_np = NP; gq.map_p = r->P; gq.out_row0 = (long)t * r->B * r->P;
      hipLaunchKernelGGL(img::gmm_raw_finish_kernel, grid1d((size_t)tiles * 16), dim3(256), 0, c.st, gq);
      CHECK_LAUNCH();
    }
    SelectArgs sel{};
    sel.g = c.g; sel.F = F; sel.logits = logits; sel.logit_stride = NP;
    sel.mode = r->select_mode;
    sel.uniform = r->uniform ? r->uniform + (size_t)t * r->B : nullptr;
    sel.forced = r->forced_idx ? r->forced_idx + t : nullptr; sel.forced_stride = r->T;
    sel.idx = r->idx ? r->idx + t : nullptr; sel.idx_stride = r->T;
    sel.slot = r->slot ? r->slot + t : nullptr; sel.slot_stride = r->T;
    sel.log_prob = r->log_prob ? r->log_prob + t : nullptr; sel.lp_stride = r->T;
    const int zw = r->P - r->n_ctx0;
    sel.zt = r->zt ? r->zt + (size_t)t * r->B * zw : nullptr; sel.zt_stride = zw; sel.zt_width = zw;
    sel.role_out = r->role;
    TRY(launch_acq_select(c, sel));
    CHECK_LAUNCH();
    if (t + 1 < r->T) {   // the chosen point enters the context: its input row becomes Ex + Ey
      aa.g = c.g;
      hipLaunchKernelGGL(patch_row_kernel, dim3(r->B), dim3(64), 0, c.st, aa, r->n_ctx0 + t + 1);
      CHECK_LAUNCH();
    }
  }
  if (want_gmm) {   // GMM heads of all T * B * n_t target rows, then the parameter maps + mixture log-likelihood
    const long per_step = (long)r->B * n_t, total = per_step * r->T;
    float *raw = c.at(c.pl.xRaw);
    for (int k = 0; k < m->C; ++k) {
      HeadArgs ha{};
      ha.X = Zimg; ha.ntiles = (total + 15) / 16; ha.M = total; ha.img = img + (long)m->L * lw + (long)(1 + k) * head_words(F); ha.F = F;
      ha.out = raw; ha.out_stride = kRawStride; ha.out_off = 3 * k;
      TRY(launch_head<3>(c, ha));
    }
    img::GmmRawArgs ga{};
    ga.range_flag = c.flag();
    ga.raw = raw; ga.raw_stride = kRawStride; ga.rows = total; ga.C = m->C; ga.std_min = m->std_min;
    ga.mean = r->post_mean; ga.sd = r->post_std; ga.wgt = r->post_weight;
    ga.value = r->target_all; ga.value_mod = per_step; ga.ll = r->target_ll;
    hipLaunchKernelGGL(img::gmm_raw_finish_kernel, grid1d((size_t)total), dim3(256), 0, c.st, ga);
    CHECK_LAUNCH();
  }
  return ALINE_OK;
}

}  // namespace X3_NS
