// VGPR operand-bank microbenchmark (development aid): the same independent VALU instruction stream with source
// registers in distinct banks (index mod 4) or in one bank.
//   hipcc --offload-arch=gfx950 -O2 tools/bank_ubench.hip -o tools/bank_ubench && tools/bank_ubench
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP8(x) x x x x x x x x
#define INIT                                                                                     \
  "v_mov_b32 v1, 1.0\n v_mov_b32 v2, 0.5\n v_mov_b32 v3, 0.25\n v_mov_b32 v5, 0.5\n"             \
  "v_mov_b32 v9, 0.25\n v_mov_b32 v6, 0.5\n v_mov_b32 v7, 0.25\n v_mov_b32 v13, 2.0\n"
#define CLOB "v1", "v2", "v3", "v5", "v6", "v7", "v9", "v13", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27"

template <int V>
__global__ void __launch_bounds__(256) k(float *out, int iters) {
  asm volatile(INIT ::: CLOB);
  for (int i = 0; i < iters; ++i) {
    if (V == 0)  // three sources, three banks
      asm volatile(REP8("v_fma_f32 v20, v1, v2, v3\n v_fma_f32 v21, v1, v2, v3\n v_fma_f32 v22, v1, v2, v3\n v_fma_f32 v23, v1, v2, v3\n") ::: CLOB);
    if (V == 1)  // three sources, one bank
      asm volatile(REP8("v_fma_f32 v20, v1, v5, v9\n v_fma_f32 v21, v1, v5, v9\n v_fma_f32 v22, v1, v5, v9\n v_fma_f32 v23, v1, v5, v9\n") ::: CLOB);
    if (V == 2)  // two of three in one bank
      asm volatile(REP8("v_fma_f32 v20, v1, v5, v3\n v_fma_f32 v21, v1, v5, v3\n v_fma_f32 v22, v1, v5, v3\n v_fma_f32 v23, v1, v5, v3\n") ::: CLOB);
    if (V == 3)  // two sources, two banks
      asm volatile(REP8("v_mul_f32 v20, v1, v2\n v_mul_f32 v21, v1, v2\n v_mul_f32 v22, v1, v2\n v_mul_f32 v23, v1, v2\n") ::: CLOB);
    if (V == 4)  // two sources, one bank
      asm volatile(REP8("v_mul_f32 v20, v1, v5\n v_mul_f32 v21, v1, v5\n v_mul_f32 v22, v1, v5\n v_mul_f32 v23, v1, v5\n") ::: CLOB);
    if (V == 5)  // destination in the bank of a source of the same instruction
      asm volatile(REP8("v_fma_f32 v21, v1, v2, v3\n v_fma_f32 v25, v1, v2, v3\n v_fma_f32 v21, v1, v2, v3\n v_fma_f32 v25, v1, v2, v3\n") ::: CLOB);
    if (V == 6)  // same register read twice + one other
      asm volatile(REP8("v_fma_f32 v20, v1, v1, v3\n v_fma_f32 v21, v1, v1, v3\n v_fma_f32 v22, v1, v1, v3\n v_fma_f32 v23, v1, v1, v3\n") ::: CLOB);
    if (V == 7)  // dependent chain through one register (latency-bound per wave)
      asm volatile(REP8("v_fma_f32 v20, v20, v2, v3\n v_fma_f32 v20, v20, v2, v3\n v_fma_f32 v20, v20, v2, v3\n v_fma_f32 v20, v20, v2, v3\n") ::: CLOB);
    if (V == 8)  // v_max3 three banks
      asm volatile(REP8("v_max3_f32 v20, v1, v2, v3\n v_max3_f32 v21, v1, v2, v3\n v_max3_f32 v22, v1, v2, v3\n v_max3_f32 v23, v1, v2, v3\n") ::: CLOB);
    if (V == 9)  // v_max3 one bank
      asm volatile(REP8("v_max3_f32 v20, v1, v5, v9\n v_max3_f32 v21, v1, v5, v9\n v_max3_f32 v22, v1, v5, v9\n v_max3_f32 v23, v1, v5, v9\n") ::: CLOB);
    if (V == 10)  // SGPR-free VOP2 with literal constant (v_fmamk)
      asm volatile(REP8("v_fmamk_f32 v20, v1, 0x3fb8aa3b, v2\n v_fmamk_f32 v21, v1, 0x3fb8aa3b, v2\n v_fmamk_f32 v22, v1, 0x3fb8aa3b, v2\n v_fmamk_f32 v23, v1, 0x3fb8aa3b, v2\n") ::: CLOB);
    if (V == 11)  // same, both sources in one bank
      asm volatile(REP8("v_fmamk_f32 v20, v1, 0x3fb8aa3b, v5\n v_fmamk_f32 v21, v1, 0x3fb8aa3b, v5\n v_fmamk_f32 v22, v1, 0x3fb8aa3b, v5\n v_fmamk_f32 v23, v1, 0x3fb8aa3b, v5\n") ::: CLOB);
  }
  float r;
  asm volatile("v_mov_b32 %0, v20" : "=v"(r)::CLOB);
  if (r == 123.456f) out[0] = r;
}

template <int V>
void run_occ(const char *name, float *d, int waves_per_simd) {
  // blocks of 4 waves (one per SIMD); waves_per_simd blocks per CU: issue interval of a wave vs latency hiding
  const int iters = 4000, blocks = 256 * waves_per_simd;
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_simd = (double)waves_per_simd * iters * 32;
  printf("%-30s %2d waves/SIMD: %.2f cycles per wave-instruction per SIMD (%.2f per wave)\n", name, waves_per_simd,
         ms * 1e-3 * 2.4e9 / insts_per_simd, ms * 1e-3 * 2.4e9 / (iters * 32.0));
}

template <int V>
void run(const char *name, float *d) {
  const int iters = 2000, blocks = 256 * 4 * 4;  // 4 blocks of 4 waves per CU: 4 waves per SIMD
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 16 waves (4 resident x 4 rounds... all resident at 16/SIMD? 16 blocks/CU x 4 waves = 16 waves/SIMD)
  const double insts_per_simd = 16.0 * iters * 32;
  printf("%-44s %.3f ms  %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
}

int main() {
  float *d;
  hipMalloc(&d, 4);
  run<0>("fma 3 src, 3 banks", d);
  run<1>("fma 3 src, 1 bank", d);
  run<2>("fma 3 src, 2 in one bank", d);
  run<3>("mul 2 src, 2 banks", d);
  run<4>("mul 2 src, 1 bank", d);
  run<5>("fma dst in a source's bank", d);
  run<6>("fma same register twice", d);
  run<7>("fma dependent chain", d);
  run<8>("max3 3 banks", d);
  run<9>("max3 1 bank", d);
  run<10>("fmamk 2 banks", d);
  run<11>("fmamk 1 bank", d);
  for (int w : {1, 2, 3, 4, 6, 8}) run_occ<0>("independent fma", d, w);
  for (int w : {1, 2, 3, 4, 6, 8}) run_occ<7>("dependent fma chain", d, w);
  for (int w : {1, 2, 4, 8}) run_occ<3>("independent mul (2 src)", d, w);
  return 0;
}
