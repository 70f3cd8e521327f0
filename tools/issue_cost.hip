// Issue cost of every instruction class the fused PT-RWM step loop is made of, on the machine it runs on (gfx950):
// ns of SIMD time per wave64 instruction with 4 wavefronts resident per SIMD (the headline kernel's residency), eight
// independent accumulators per lane, each instruction written in inline asm so that what is timed is exactly that opcode.
// tools/issue_model.py multiplies these by the opcode histogram of the shipped kernel's step loop: the
// "cost-weighted" issue fraction of bench.py's roofline object.
//   hipcc --offload-arch=gfx950 -O3 tools/issue_cost.hip -o tools/issue_cost && tools/issue_cost > profiles/r03_issue_costs.json
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define ITERS 2000
#define REP 8  // x 8 accumulators = 64 instructions per loop iteration

enum Op {
  ADD_F32, MUL_F32, FMA_F32, FMAMK_F32, MAX3_F32, MED3_F32, CNDMASK, XOR_B32, LSHRREV, CVT_F32_U32, AND_OR, ADD_U32, BITOP3,
  MAD_U64_U32, MUL_LO_U32, EXP_F32, LOG_F32, SQRT_F32, SIN_F32, COS_F32, RCP_F32, CVT_F64_F32, ADD_F64, FMA_F64, CMP_F32,
  READLANE, WRITELANE, MOV_DPP, DS_READ_B32, DS_WRITE_B32, DS_READ_B128, PK_FMA_F32, PK_MUL_F32, PK_ADD_F32,
  // mixes (per GROUP of four instructions, not per instruction): do instruction classes overlap on the SIMD?
  MIX_EXP_FMA3, MIX_MAD_BITOP_FMA2, MIX_EXP_MAD_BITOP_FMA, N_OPS
};
static const char *kNames[N_OPS] = {
  "v_add_f32", "v_mul_f32", "v_fma_f32", "v_fmamk_f32", "v_max3_f32", "v_med3_f32", "v_cndmask_b32", "v_xor_b32",
  "v_lshrrev_b32", "v_cvt_f32_u32", "v_and_or_b32", "v_add_u32", "v_bitop3_b32", "v_mad_u64_u32", "v_mul_lo_u32",
  "v_exp_f32", "v_log_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_f32", "v_cvt_f64_f32", "v_add_f64", "v_fma_f64",
  "v_cmp_lt_f32", "v_readlane_b32", "v_writelane_b32", "v_mov_b32_dpp", "ds_read_b32", "ds_write_b32", "ds_read_b128",
  "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32",  // (two fp32 results per lane each)
  "mix:v_exp_f32+3*v_fma_f32", "mix:v_mad_u64_u32+v_bitop3_b32+2*v_fma_f32", "mix:v_exp_f32+v_mad_u64_u32+v_bitop3_b32+v_fma_f32"};

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed) {
  __shared__ float lds[256 * 4 + 64];
  uint32_t a[8];
  float f[8];
  double d[8];
  uint64_t p[8];
  const float c1 = 1.0001f + seed * 1e-9f, c2 = 0.5f;
  const uint32_t u1 = seed | 0x9E3779B9u, u2 = seed ^ 0x5bd1e995u;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = seed + threadIdx.x * 8 + i;
    f[i] = 0.5f + (float)(a[i] & 1023) * 1e-4f;
    d[i] = f[i];
    p[i] = a[i];
  }
  lds[threadIdx.x] = f[0];
  __syncthreads();
  uint32_t sreg = seed;
  const uint64_t lane_mask = 0x5555555555555555ull ^ seed;
  typedef float vec4 __attribute__((ext_vector_type(4)));
  vec4 q4 = {0, 0, 0, 0};
  typedef float vec2 __attribute__((ext_vector_type(2)));
  vec2 g[8];
  const vec2 g1 = {c1, c1}, g2 = {c2, c2};
#pragma unroll
  for (int i = 0; i < 8; ++i) g[i] = vec2{f[i], f[i] + 1.0f};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int r = 0; r < REP; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c2));
        if (OP == MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c1));
        if (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(c1), "v"(c2));
        if (OP == FMAMK_F32) asm volatile("v_fmamk_f32 %0, %0, 0x3f800347, %1" : "+v"(f[i]) : "v"(c2));
        if (OP == MAX3_F32) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(c1), "v"(c2));
        if (OP == MED3_F32) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(c1), "v"(c2));
        if (OP == CNDMASK) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(f[i]) : "v"(c1), "s"(lane_mask));  // an SGPR-pair mask, as the kernels use
        if (OP == XOR_B32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(u1));
        if (OP == LSHRREV) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i]));
        if (OP == CVT_F32_U32) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[i]) : "v"(a[i]));
        if (OP == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(u1), "v"(u2));
        if (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(u1));
        if (OP == BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(u1), "s"(sreg));
        if (OP == MAD_U64_U32) {
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p[i]) : "s"(0xD2511F53u), "v"(a[i]) : "vcc");
          a[i] = (uint32_t)p[i];
        }
        if (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(u1));
        if (OP == EXP_F32) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
        if (OP == LOG_F32) asm volatile("v_log_f32 %0, %0" : "+v"(f[i]));
        if (OP == SQRT_F32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[i]));
        if (OP == SIN_F32) asm volatile("v_sin_f32 %0, %0" : "+v"(f[i]));
        if (OP == COS_F32) asm volatile("v_cos_f32 %0, %0" : "+v"(f[i]));
        if (OP == RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
        if (OP == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
        if (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"((double)c1));
        if (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"((double)c1), "v"((double)c2));
        if (OP == CMP_F32) asm volatile("v_cmp_lt_f32 s[40:41], %0, %1" : : "v"(f[i]), "v"(c1) : "s40", "s41");
        if (OP == READLANE) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(sreg) : "v"(a[i]));
        if (OP == WRITELANE) asm volatile("v_writelane_b32 %0, %1, 5" : "+v"(a[i]) : "s"(sreg));
        if (OP == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
        if (OP == DS_READ_B32) asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(f[i]) : "v"((uint32_t)(threadIdx.x * 4)));
        if (OP == DS_WRITE_B32) asm volatile("ds_write_b32 %0, %1" : : "v"((uint32_t)(threadIdx.x * 4)), "v"(f[i]) : "memory");
        if (OP == DS_READ_B128) asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q4) : "v"((uint32_t)(threadIdx.x * 16)));
        if (OP == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(g[i]) : "v"(g1), "v"(g2));
        if (OP == PK_MUL_F32) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(g[i]) : "v"(g1));
        if (OP == PK_ADD_F32) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(g[i]) : "v"(g2));
        // mixes: one GROUP of four independent instructions per slot (a quarter as many groups: i < 2 only)
        if (OP == MIX_EXP_FMA3 && i < 2) {
          asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i + 2]) : "v"(c1), "v"(c2));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i + 4]) : "v"(c1), "v"(c2));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i + 6]) : "v"(c1), "v"(c2));
        }
        if (OP == MIX_MAD_BITOP_FMA2 && i < 2) {
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p[i]) : "s"(0xD2511F53u), "v"(a[i]) : "vcc");
          asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i + 2]) : "v"(u1), "s"(sreg));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i + 4]) : "v"(c1), "v"(c2));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i + 6]) : "v"(c1), "v"(c2));
          a[i] = (uint32_t)p[i];
        }
        if (OP == MIX_EXP_MAD_BITOP_FMA && i < 2) {
          asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p[i]) : "s"(0xD2511F53u), "v"(a[i]) : "vcc");
          asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i + 2]) : "v"(u1), "s"(sreg));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i + 6]) : "v"(c1), "v"(c2));
          a[i] = (uint32_t)p[i];
        }
      }
    }
  }
  uint32_t s = sreg;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + (uint32_t)f[i] + (uint32_t)d[i] + (uint32_t)p[i];
  s += (uint32_t)q4.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += (uint32_t)g[i].x + (uint32_t)g[i].y;
  if (s == 0x12345678u) out[threadIdx.x] = s;
}

template <int OP>
static double run(uint32_t *out, int blocks_per_cu) {
  const int blocks = 256 * blocks_per_cu;  // blocks of 4 waves, one per SIMD -> blocks_per_cu waves per SIMD
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  k<OP><<<blocks, 256>>>(out, 1);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) k<OP><<<blocks, 256>>>(out, 1);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / 5 * 1e6 / ((double)ITERS * REP * 8 * blocks_per_cu);  // ns of SIMD time per wave-instruction
}

template <int OP>
static void all(uint32_t *out, double *res4, double *res2) {
  res4[OP] = run<OP>(out, 4);
  res2[OP] = run<OP>(out, 2);
  if constexpr (OP + 1 < N_OPS) all<OP + 1>(out, res4, res2);
}

int main() {
  uint32_t *out;
  hipMalloc(&out, 4096);
  double r4[N_OPS], r2[N_OPS];
  all<0>(out, r4, r2);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  printf("{\n  \"device\": \"%s\", \"compute_units\": %d, \"unit\": \"ns of SIMD time per wave64 instruction\",\n", prop.gcnArchName,
         prop.multiProcessorCount);
  printf("  \"method\": \"tools/issue_cost.hip: %d x 64 inline-asm instructions per lane, 8 independent accumulators, 256-thread "
         "blocks, N blocks per CU = N wavefronts per SIMD\",\n", ITERS);
  printf("  \"waves_per_simd_4\": {");
  for (int i = 0; i < N_OPS; ++i) printf("%s\"%s\": %.4f", i ? ", " : "", kNames[i], r4[i]);
  printf("},\n  \"waves_per_simd_2\": {");
  for (int i = 0; i < N_OPS; ++i) printf("%s\"%s\": %.4f", i ? ", " : "", kNames[i], r2[i]);
  printf("}\n}\n");
  hipFree(out);
  return 0;
}
