// Pieces shared by the tile-image kernels (s3.h, x3_impl.h, attn3.h) and the GMM head epilogue.  (Until round 3 these lived in
// wide.h, the bf16 d = 256 path; that path -- NLL error 2e-2, off the bench line since round 2 -- was removed in round 4 and only
// what the reference-precision paths use of it is kept here.)
#pragma once
#include "common.h"
#include "kernels.h"

namespace img {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

__device__ __forceinline__ float group_sum4(float v) {   // over the 4 lane groups holding one token
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// GMM parameter maps + mixture log-likelihood from the raw head outputs raw[row][3 c + j] (model/head.py:152-186,
// 251-266; utils/eval.py:200-207): mean_c = raw[c][0], std_c = softplus(raw[c][1]) + std_min, weight = softmax_c(raw[c][2])
struct GmmRawArgs {
  const float *raw; int raw_stride; long rows; int C; float std_min;
  int nblk; long blk_stride;                     // raw = sum of nblk partial arrays blk_stride apart (0 / 1: a single one)
  float *mean, *sd, *wgt;                        // [rows, C] or null
  const float *value; long value_mod;            // value[(value_row0 + row) % value_mod] or null
  long value_row0;
  float *ll;                                     // [rows] or null
  unsigned *range_flag;                          // f16 range guard (common.h): raised when a log-likelihood is not finite; may be null
  // optional row map of the outputs: raw row i is token row i % map_np of episode i / map_np of a tile image; rows >= map_p (targets,
  // padding) are skipped, the others go to output row out_row0 + episode * map_p + token row (posterior_out_query by slot)
  int map_np, map_p; long out_row0;
};
__global__ void gmm_raw_finish_kernel(GmmRawArgs a) {
  const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= a.rows) return;
  long orow = row;
  if (a.map_np > 0) {
    const long ep = row / a.map_np;
    const int tr = (int)(row - ep * a.map_np);
    if (tr >= a.map_p) return;
    orow = a.out_row0 + ep * a.map_p + tr;
  }
  float r[48];
  for (int e = 0; e < 3 * a.C; ++e) {
    float s = a.raw[row * a.raw_stride + e];
    for (int k = 1; k < a.nblk; ++k) s += a.raw[k * a.blk_stride + row * a.raw_stride + e];
    r[e] = s;
  }
  float mxw = -INFINITY;
  for (int c = 0; c < a.C; ++c) mxw = fmaxf(mxw, r[3 * c + 2]);
  float sw = 0.f;
  for (int c = 0; c < a.C; ++c) sw += __expf(r[3 * c + 2] - mxw);
  const float v = (a.ll && a.value) ? a.value[(a.value_row0 + row) % a.value_mod] : 0.f;
  float mx2 = -INFINITY, lps[16];
  for (int c = 0; c < a.C; ++c) {
    const float mean = r[3 * c], sd = softplus_f(r[3 * c + 1]) + a.std_min, w = __expf(r[3 * c + 2] - mxw) / sw;
    if (a.mean) a.mean[orow * a.C + c] = mean;
    if (a.sd) a.sd[orow * a.C + c] = sd;
    if (a.wgt) a.wgt[orow * a.C + c] = w;
    const float zz = (v - mean) / sd;
    lps[c] = -0.5f * zz * zz - logf(sd) - 0.91893853320467274178f + logf(w);
    mx2 = fmaxf(mx2, lps[c]);
  }
  if (a.ll && a.value) {
    float se = 0.f;
    for (int c = 0; c < a.C; ++c) se += __expf(lps[c] - mx2);
    const float ll = mx2 + logf(se);
    a.ll[row] = ll;
    if (!(fabsf(ll) <= 3.4e38f)) range_raise(a.range_flag, ALINE_RANGE_ACT);
  }
}

}  // namespace img
