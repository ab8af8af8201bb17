// Shared device helpers for the gfx950 (CDNA4) kernels of libaline_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#define WAVE 64  // CDNA wavefront width (hard-coded: no wave-size macro on gfx950)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// fp32 -> bf16 round-to-nearest-even through the hardware conversion (v_cvt_pk_bf16_f32): a NaN stays a NaN -- the integer
// form (u + 0x7FFF + lsb) >> 16 turns some NaNs into 0 / inf (MI355X_MICROARCH.md, correctness boundaries), which would
// hide a failed step in the bf16 modes.
__device__ __forceinline__ unsigned short f2bf(float f) {
  return __builtin_bit_cast(unsigned short, (__bf16)f);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

// split a = hi + lo (+ O(2^-17 |a|)) with hi, lo bf16  -- operand form of ALINE_PREC_BF16X3
__device__ __forceinline__ void split_bf16(float a, unsigned short &hi, unsigned short &lo) {
  hi = f2bf(a);
  lo = f2bf(a - bf2f(hi));
}

// ---- f16 range guard (include/aline_hip.h: aline_f16_range_status) -------------------------------------------------------
// The F16X3 kernels split fp32 operands into f16 halves; an operand that is non-finite or >= 65504 in magnitude turns into
// inf / NaN there.  Such values reach the next LayerNorm (whose statistics become NaN), the softmax of the design
// selection or the mixture log-likelihood, so the guard costs the hot kernels one add per LayerNorm: a per-lane
// accumulator of the reciprocal standard deviations that is NaN iff something overflowed, checked once per wave; the
// kernels off the hot path (weight packing, input image assembly, selection, GMM finish) test their values directly.
#define ALINE_RANGE_ACT 1u      /* an activation operand */
#define ALINE_RANGE_WEIGHT 2u   /* a weight (after the 2^8 pre-scale) */
__device__ __forceinline__ bool f16_out_of_range(float v) { return !(fabsf(v) < 65504.f); }    // (true for NaN)
__device__ __forceinline__ void range_raise(unsigned *flag, unsigned bit) { if (flag) atomicOr(flag, bit); }
__device__ __forceinline__ void range_check_nan(unsigned *flag, float chk) { if (flag && chk != chk) atomicOr(flag, ALINE_RANGE_ACT); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
  return v;
}

// ---- whole-wave reductions and scan on DPP (row shifts + row broadcasts: ~8 cycles a step; the __shfl_xor forms above go through the LDS
// crossbar, ~120 cycles a step, and a design selection is a chain of ~36 of them) ----------------------------------------------------
// dpp_ctrl: row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143.  A lane whose source lies outside its row (or that the row /
// bank mask switches off) gets `idle`.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_f32(float idle, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(idle), __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false));
}
// inclusive prefix sum over the 64 lanes (lane 63: the total)
__device__ __forceinline__ float wave_scan_sum(float v) {
  v += dpp_f32<0x111, 0xf, 0xf>(0.f, v);
  v += dpp_f32<0x112, 0xf, 0xf>(0.f, v);
  v += dpp_f32<0x114, 0xf, 0xf>(0.f, v);
  v += dpp_f32<0x118, 0xf, 0xf>(0.f, v);
  v += dpp_f32<0x142, 0xa, 0xf>(0.f, v);      // rows 1, 3 += lane 15 of rows 0, 2
  v += dpp_f32<0x143, 0xc, 0xf>(0.f, v);      // rows 2, 3 += lane 31
  return v;
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_scan_sum(v)), 63));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  const float ninf = -INFINITY;
  v = fmaxf(v, dpp_f32<0x111, 0xf, 0xf>(ninf, v));
  v = fmaxf(v, dpp_f32<0x112, 0xf, 0xf>(ninf, v));
  v = fmaxf(v, dpp_f32<0x114, 0xf, 0xf>(ninf, v));
  v = fmaxf(v, dpp_f32<0x118, 0xf, 0xf>(ninf, v));
  v = fmaxf(v, dpp_f32<0x142, 0xa, 0xf>(ninf, v));
  v = fmaxf(v, dpp_f32<0x143, 0xc, 0xf>(ninf, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float softplus_f(float x) {
  // torch.nn.functional.softplus(beta=1, threshold=20)
  return x > 20.f ? x : log1pf(expf(x));
}

// ReLU.  NOT the one-instruction integer form max(bits, 0) (it would save the canonicalising v_max_f32 x, x, x that
// llvm.maxnum emits in IEEE mode): v_max_i32 applied directly to MFMA results made the bf16 block kernel of round 1 return
// different logits for a token tile in 1 of 3 runs on gfx950 (ROCm 7.2 hipcc) -- an MFMA -> integer-VALU hazard the
// compiler's wait states do not cover -- while the float form is reproducible (60 of 60 runs).
__device__ __forceinline__ float relu_nn(float a) { return fmaxf(a, 0.f); }
