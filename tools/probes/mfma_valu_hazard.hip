// Probe: how many wait states does gfx950 need between v_mfma_f32_16x16x32_bf16 writing D and a VALU instruction reading
// D -- v_max_i32, v_max_f32 and v_accvgpr_read_b32 -- and does hipcc's own padding (compiler-scheduled variant) suffice?
// One asm statement per case holds MFMA, pad and reader, so nothing is padded for us (cdna_hip_programming.md 5.7 item 2).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hz tools/probes/mfma_valu_hazard.hip && /tmp/hz
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NOP, int KIND>
__global__ void k(const float *in, float *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)in[(i * 8 + j) & 4095]; b[j] = (__bf16)in[(i * 5 + j * 3 + 1) & 4095]; }
  float r;
  // the accumulator tile lives in v[20:23] (a[20:23] for the accvgpr case), named literally and listed as clobbers
#define INIT_V "v_mov_b32 v20, -1.0\n\tv_mov_b32 v21, 2.0\n\tv_mov_b32 v22, -3.0\n\tv_mov_b32 v23, 4.0\n\ts_nop 7\n\ts_nop 7\n\t"
  if (KIND == 0)        // integer max on the float bits (= ReLU for finite values)
    asm volatile(INIT_V "v_mfma_f32_16x16x32_bf16 v[20:23], %1, %2, v[20:23]\n\ts_nop %3\n\tv_max_i32 %0, 0, v20\n\ts_nop 7\n\ts_nop 7"
                 : "=&v"(r) : "v"(a), "v"(b), "n"(NOP) : "v20", "v21", "v22", "v23");
  else if (KIND == 1)
    asm volatile(INIT_V "v_mfma_f32_16x16x32_bf16 v[20:23], %1, %2, v[20:23]\n\ts_nop %3\n\tv_max_f32_e64 %0, 0, v20\n\ts_nop 7\n\ts_nop 7"
                 : "=&v"(r) : "v"(a), "v"(b), "n"(NOP) : "v20", "v21", "v22", "v23");
  else if (KIND == 5 || KIND == 6) {
    // integer (5) / float (6) max writing the two halves of a register pair, v_pk_fma_f32 reading the pair NOP + 1 states
    // later, with an MFMA in flight (the shape hipcc's SLP vectoriser produced in wide_block_kernel<WB_ACQ>)
    float x0 = in[(i * 3 + 1) & 4095], x1 = in[(i * 11 + 5) & 4095], w = in[(i * 13 + 2) & 4095];
    if (KIND == 5)
      asm volatile(INIT_V "v_mov_b32 v24, %4\n\tv_mov_b32 v25, %5\n\tv_mov_b32 v28, %6\n\tv_mov_b32 v29, %6\n\tv_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\ts_nop 7\n\t"
                   "v_mfma_f32_16x16x32_bf16 v[20:23], %1, %2, v[20:23]\n\t"
                   "v_max_i32 v26, 0, v24\n\tv_max_i32 v27, 0, v25\n\ts_nop %3\n\t"
                   "v_pk_fma_f32 v[30:31], v[26:27], v[28:29], v[30:31] op_sel_hi:[1,0,1]\n\ts_nop 7\n\tv_mov_b32 %0, v30\n\ts_nop 15"
                   : "=&v"(r) : "v"(a), "v"(b), "n"(NOP), "v"(x0), "v"(x1), "v"(w)
                   : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
    else
      asm volatile(INIT_V "v_mov_b32 v24, %4\n\tv_mov_b32 v25, %5\n\tv_mov_b32 v28, %6\n\tv_mov_b32 v29, %6\n\tv_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\ts_nop 7\n\t"
                   "v_mfma_f32_16x16x32_bf16 v[20:23], %1, %2, v[20:23]\n\t"
                   "v_max_f32_e64 v26, 0, v24\n\tv_max_f32_e64 v27, 0, v25\n\ts_nop %3\n\t"
                   "v_pk_fma_f32 v[30:31], v[26:27], v[28:29], v[30:31] op_sel_hi:[1,0,1]\n\ts_nop 7\n\tv_mov_b32 %0, v30\n\ts_nop 15"
                   : "=&v"(r) : "v"(a), "v"(b), "n"(NOP), "v"(x0), "v"(x1), "v"(w)
                   : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
  } else if (KIND == 4) {
    // the shape found in wide_block_kernel<WB_ACQ>: MFMA A, an independent MFMA B right behind it, then the reader of A's D
    asm volatile(INIT_V "v_mov_b32 v24, -1.0\n\tv_mov_b32 v25, 2.0\n\tv_mov_b32 v26, -3.0\n\tv_mov_b32 v27, 4.0\n\ts_nop 7\n\ts_nop 7\n\t"
                 "v_mfma_f32_16x16x32_bf16 v[20:23], %1, %2, v[20:23]\n\t"
                 "v_mfma_f32_16x16x32_bf16 v[24:27], %2, %1, v[24:27]\n\t"
                 "s_nop %3\n\tv_max_i32 %0, 0, v20\n\ts_nop 15\n\ts_nop 15"
                 : "=&v"(r) : "v"(a), "v"(b), "n"(NOP) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
  } else if (KIND == 3) {
    // write-after-read on SrcA: two MFMAs back to back (the second queues behind the first), then a VALU instruction
    // overwrites the first register of the second MFMA's A operand NOP + 1 states later (NOP = 15: never -> reference)
    float junk = in[(i * 7 + 3) & 4095] * 1e3f;
    asm volatile(INIT_V "v_mov_b32 v24, -1.0\n\tv_mov_b32 v25, 2.0\n\tv_mov_b32 v26, -3.0\n\tv_mov_b32 v27, 4.0\n\t"
                 "v_mov_b32 v28, %5\n\tv_mov_b32 v29, %5\n\tv_mov_b32 v30, %5\n\tv_mov_b32 v31, %5\n\t"
                 "v_mov_b32 v32, %4\n\ts_nop 7\n\ts_nop 7\n\t"
                 "v_mfma_f32_16x16x32_bf16 v[20:23], %1, %2, v[20:23]\n\t"
                 "v_mfma_f32_16x16x32_bf16 v[24:27], v[28:31], %2, v[24:27]\n\t"
                 "s_nop %3\n\t"
                 "v_mov_b32 v28, v32\n\t"
                 "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\tv_max_f32_e64 %0, 0, v24\n\ts_nop 7"
                 : "=&v"(r) : "v"(a), "v"(b), "n"(NOP), "v"(junk), "v"(__builtin_bit_cast(f32x4, a)[0])
                 : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32");
  } else {
    asm volatile("v_accvgpr_write_b32 a20, -1.0\n\tv_accvgpr_write_b32 a21, 2.0\n\tv_accvgpr_write_b32 a22, -4.0\n\tv_accvgpr_write_b32 a23, 4.0\n\ts_nop 7\n\ts_nop 7\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[20:23], %1, %2, a[20:23]\n\ts_nop %3\n\tv_accvgpr_read_b32 %0, a20\n\ts_nop 7\n\ts_nop 7"
                 : "=&v"(r) : "v"(a), "v"(b), "n"(NOP) : "a20", "a21", "a22", "a23");
    r = fmaxf(r, 0.f);
  }
  out[i] = r;
}
// what hipcc itself emits for the same pattern (builtin MFMA, integer max right behind it)
__global__ void k_compiler(const float *in, float *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)in[(i * 8 + j) & 4095]; b[j] = (__bf16)in[(i * 5 + j * 3 + 1) & 4095]; }
  f32x4 acc = {-1.f, 2.f, -3.f, 4.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  out[i] = __int_as_float(max(__float_as_int(acc[0]), 0));
}
template <int NOP, int KIND>
long run(const float *din, float *dout, const std::vector<float> &ref, int n, int reps) {
  std::vector<float> h(n);
  long bad = 0;
  for (int r = 0; r < reps; ++r) {
    hipLaunchKernelGGL((k<NOP, KIND>), dim3(n / 256), dim3(256), 0, 0, din, dout);
    (void)hipMemcpy(h.data(), dout, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) bad += (h[i] != ref[i]);
  }
  return bad;
}
int main() {
  const int n = 256 * 2048, reps = 8;
  std::vector<float> hin(4096);
  for (int i = 0; i < 4096; ++i) hin[i] = ((i * 2654435761u) % 2001) / 1000.f - 1.f;
  float *din, *dout;
  (void)hipMalloc(&din, 4096 * 4); (void)hipMalloc(&dout, n * 4);
  (void)hipMemcpy(din, hin.data(), 4096 * 4, hipMemcpyHostToDevice);
  std::vector<float> ref(n);
  hipLaunchKernelGGL((k<15, 1>), dim3(n / 256), dim3(256), 0, 0, din, dout);      // 16 states: beyond every table value
  (void)hipMemcpy(ref.data(), dout, n * 4, hipMemcpyDeviceToHost);
  printf("wait states between v_mfma_f32_16x16x32_bf16 (D) and the reader: mismatching lanes of %d x %d\n", reps, n);
#define ROW(N) printf("  s_nop %2d (%2d states): v_max_i32 %8ld   v_max_f32 %8ld   v_accvgpr_read %8ld\n", N, N + 1, \
                      run<N, 0>(din, dout, ref, n, reps), run<N, 1>(din, dout, ref, n, reps), run<N, 2>(din, dout, ref, n, reps));
  ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12) ROW(13)
  std::vector<float> h(n);
  long bad = 0;
  for (int r = 0; r < reps; ++r) {
    hipLaunchKernelGGL(k_compiler, dim3(n / 256), dim3(256), 0, 0, din, dout);
    (void)hipMemcpy(h.data(), dout, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) bad += (h[i] != ref[i]);
  }
  printf("  hipcc-scheduled builtin MFMA + integer max: %ld mismatching lanes\n", bad);
  // write-after-read on SrcA of a queued MFMA
  {
    std::vector<float> ref3(n), h3(n);
    hipLaunchKernelGGL((k<15, 3>), dim3(n / 256), dim3(256), 0, 0, din, dout);
    (void)hipMemcpy(ref3.data(), dout, n * 4, hipMemcpyDeviceToHost);
    printf("VALU write of SrcA (first register) of an MFMA queued behind another MFMA, s_nop between them:\n");
#define ROW3(N) { long bad3 = 0; for (int r = 0; r < reps; ++r) { hipLaunchKernelGGL((k<N, 3>), dim3(n / 256), dim3(256), 0, 0, din, dout); \
      (void)hipMemcpy(h3.data(), dout, n * 4, hipMemcpyDeviceToHost); for (int i = 0; i < n; ++i) bad3 += (h3[i] != ref3[i]); } \
      printf("  s_nop %2d: %8ld mismatching lanes\n", N, bad3); }
    ROW3(0) ROW3(1) ROW3(2) ROW3(3) ROW3(4) ROW3(6) ROW3(8) ROW3(12)
  }
  {
    std::vector<float> ref4(n), h4(n);
    hipLaunchKernelGGL((k<15, 4>), dim3(n / 512), dim3(512), 0, 0, din, dout);
    (void)hipMemcpy(ref4.data(), dout, n * 4, hipMemcpyDeviceToHost);
    printf("MFMA A, independent MFMA B, s_nop N, v_max_i32 reading A's D (512-thread blocks: two waves per SIMD):\n");
#define ROW4(N) { long bad4 = 0; for (int r = 0; r < reps; ++r) { hipLaunchKernelGGL((k<N, 4>), dim3(n / 512), dim3(512), 0, 0, din, dout); \
      (void)hipMemcpy(h4.data(), dout, n * 4, hipMemcpyDeviceToHost); for (int i = 0; i < n; ++i) bad4 += (h4[i] != ref4[i]); } \
      printf("  s_nop %2d (%2d states behind B): %8ld mismatching lanes\n", N, N + 1, bad4); }
    ROW4(0) ROW4(1) ROW4(2) ROW4(3) ROW4(4) ROW4(5) ROW4(6) ROW4(7) ROW4(8) ROW4(10) ROW4(12)
  }
  {
    std::vector<float> ref5(n), h5(n);
    hipLaunchKernelGGL((k<15, 6>), dim3(n / 512), dim3(512), 0, 0, din, dout);
    (void)hipMemcpy(ref5.data(), dout, n * 4, hipMemcpyDeviceToHost);
    printf("v_max_{i32,f32} x2 -> s_nop N -> v_pk_fma_f32 (low half read back), an MFMA in flight, 512-thread blocks:\n");
#define ROW5(N) { long bi = 0, bf = 0; for (int r = 0; r < reps; ++r) { \
      hipLaunchKernelGGL((k<N, 5>), dim3(n / 512), dim3(512), 0, 0, din, dout); (void)hipMemcpy(h5.data(), dout, n * 4, hipMemcpyDeviceToHost); \
      for (int i = 0; i < n; ++i) bi += (h5[i] != ref5[i]); \
      hipLaunchKernelGGL((k<N, 6>), dim3(n / 512), dim3(512), 0, 0, din, dout); (void)hipMemcpy(h5.data(), dout, n * 4, hipMemcpyDeviceToHost); \
      for (int i = 0; i < n; ++i) bf += (h5[i] != ref5[i]); } \
      printf("  s_nop %2d: int producer %8ld   float producer %8ld mismatching lanes\n", N, bi, bf); }
    ROW5(0) ROW5(1) ROW5(2) ROW5(4) ROW5(8)
  }
  return 0;
}
