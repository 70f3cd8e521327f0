// Issue cost of every instruction class the fused PT-RWM step loop is made of, on the machine it runs on (gfx950):
// ns of SIMD time per wave64 instruction with 4 wavefronts resident per SIMD (the headline kernel's residency), eight
// independent accumulators per lane, each instruction written in inline asm so that what is timed is exactly that opcode.
// tools/issue_model.py multiplies these by the opcode histogram of the shipped kernel's step loop: the
// "cost-weighted" issue fraction of bench.py's roofline object.
// Round 4: every kernel also stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its loop, so each cost
// is reported in CYCLES of the clock the chip really held during that kernel (in-kernel clock = d memtime / d memrealtime
// x 100 MHz, MI355X_MICROARCH.md "DVFS give-back" item 6), not only in ns; operand-kind variants of the plain opcodes
// (second source a VGPR / an SGPR / an inline constant / the destination itself) separate "the opcode" from "its
// operand fetch"; and the sweep runs at 1, 2 and 4 waves per SIMD (one wave alone issues every 4 cycles by the guide).
//   hipcc --offload-arch=gfx950 -O3 tools/issue_cost.hip -o tools/issue_cost && tools/issue_cost > profiles/r04_issue_costs.json
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <vector>

#define ITERS 2000
#define REP 8  // x 8 accumulators = 64 instructions per loop iteration

enum Op {
  ADD_F32, MUL_F32, FMA_F32, FMAMK_F32, MAX3_F32, MED3_F32, CNDMASK, XOR_B32, LSHRREV, CVT_F32_U32, AND_OR, ADD_U32, BITOP3,
  MAD_U64_U32, MUL_LO_U32, EXP_F32, LOG_F32, SQRT_F32, SIN_F32, COS_F32, RCP_F32, CVT_F64_F32, ADD_F64, FMA_F64, CMP_F32,
  READLANE, WRITELANE, MOV_DPP, DS_READ_B32, DS_WRITE_B32, DS_READ_B128, PK_FMA_F32, PK_MUL_F32, PK_ADD_F32,
  // mixes (per GROUP of four instructions, not per instruction): do instruction classes overlap on the SIMD?
  MIX_EXP_FMA3, MIX_MAD_BITOP_FMA2, MIX_EXP_MAD_BITOP_FMA,
  // operand-kind variants (round 4)
  ADD_F32_S, ADD_F32_K, ADD_F32_SELF, ADD_F32_2V, FMA_F32_SS, FMA_F32_KK, FMA_F32_3V, MAX3_F32_KK, MAX3_F32_S, CNDMASK_VCC,
  CNDMASK_K, BITOP3_VK, BITOP3_3V, AND_OR_KK, MOV_B32, MUL_F32_K,
  MAD_U64_VV, MAD_U64_VK, BITOP3_VVV_ACC, MOV_B32_S, XOR_B32_S, XOR_B32_LIT, MUL_F32_S, MUL_HI_U32, AND_B32, FMAC_F32, N_OPS
};
static const char *kNames[N_OPS] = {
  "v_add_f32", "v_mul_f32", "v_fma_f32", "v_fmamk_f32", "v_max3_f32", "v_med3_f32", "v_cndmask_b32", "v_xor_b32",
  "v_lshrrev_b32", "v_cvt_f32_u32", "v_and_or_b32", "v_add_u32", "v_bitop3_b32", "v_mad_u64_u32", "v_mul_lo_u32",
  "v_exp_f32", "v_log_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_f32", "v_cvt_f64_f32", "v_add_f64", "v_fma_f64",
  "v_cmp_lt_f32", "v_readlane_b32", "v_writelane_b32", "v_mov_b32_dpp", "ds_read_b32", "ds_write_b32", "ds_read_b128",
  "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32",  // (two fp32 results per lane each)
  "mix:v_exp_f32+3*v_fma_f32", "mix:v_mad_u64_u32+v_bitop3_b32+2*v_fma_f32", "mix:v_exp_f32+v_mad_u64_u32+v_bitop3_b32+v_fma_f32",
  "v_add_f32 v,s,v", "v_add_f32 v,0.5,v", "v_add_f32 v,v,v(self)", "v_add_f32 v,v1,v2(two other VGPRs)", "v_fma_f32 v,v,s,s",
  "v_fma_f32 v,v,1.0,0.5", "v_fma_f32 v,v1,v2,v3(three other VGPRs)", "v_max3_f32 v,v,1.0,0.5", "v_max3_f32 v,v,s,v",
  "v_cndmask_b32_e32 v,v,v,vcc", "v_cndmask_b32_e64 v,v,1.0,s[2]", "v_bitop3_b32 v,v,v,63", "v_bitop3_b32 v,v1,v2,v3",
  "v_and_or_b32 v,v,0x3f,1", "v_mov_b32 v,v", "v_mul_f32 v,2.0,v",
  "v_mad_u64_u32 v,v(const),v,0", "v_mad_u64_u32 v,v,v,63", "v_bitop3_b32 v,v,v1,v2 (accumulate)", "v_mov_b32 v,s", "v_xor_b32 v,s,v",
  "v_xor_b32 v,literal,v", "v_mul_f32 v,s,v", "v_mul_hi_u32 v,v,v", "v_and_b32 v,v,v", "v_fmac_f32 v,v1,v2"};

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed, unsigned long long *stamps) {
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
  const float c3 = 0.25f + seed * 1e-9f;
  const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, c2)));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
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
        if (OP == ADD_F32_S) asm volatile("v_add_f32 %0, %1, %0" : "+v"(f[i]) : "s"(sc));
        if (OP == ADD_F32_K) asm volatile("v_add_f32 %0, 0.5, %0" : "+v"(f[i]));
        if (OP == ADD_F32_SELF) asm volatile("v_add_f32 %0, %0, %0" : "+v"(f[i]));
        if (OP == ADD_F32_2V) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f[i]) : "v"(c1), "v"(c2));
        if (OP == FMA_F32_SS) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "s"(sc));
        if (OP == FMA_F32_KK) asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(f[i]));
        if (OP == FMA_F32_3V) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f[i]) : "v"(c1), "v"(c2), "v"(c3));
        if (OP == MAX3_F32_KK) asm volatile("v_max3_f32 %0, %0, 1.0, 0.5" : "+v"(f[i]));
        if (OP == MAX3_F32_S) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "s"(sc), "v"(c1));
        if (OP == CNDMASK_VCC) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(c1));
        if (OP == CNDMASK_K) asm volatile("v_cndmask_b32_e64 %0, %0, 1.0, %1" : "+v"(f[i]) : "s"(lane_mask));
        if (OP == BITOP3_VK) asm volatile("v_bitop3_b32 %0, %0, %1, 63 bitop3:0x96" : "+v"(a[i]) : "v"(u1));
        if (OP == BITOP3_3V) asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(a[i]) : "v"(u1), "v"(u2), "v"(threadIdx.x));
        if (OP == AND_OR_KK) asm volatile("v_and_or_b32 %0, %0, 0x3f, 1" : "+v"(a[i]));
        if (OP == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(u1));
        if (OP == MUL_F32_K) asm volatile("v_mul_f32 %0, 2.0, %0" : "+v"(f[i]));
        if (OP == MAD_U64_VV) {
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p[i]) : "v"(u1), "v"(a[i]) : "vcc");
          a[i] = (uint32_t)p[i];
        }
        if (OP == MAD_U64_VK) {
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 63" : "=v"(p[i]) : "v"(u1), "v"(a[i]) : "vcc");
          a[i] = (uint32_t)p[i];
        }
        if (OP == BITOP3_VVV_ACC) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(u1), "v"(u2));
        if (OP == MOV_B32_S) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "s"(sreg));
        if (OP == XOR_B32_S) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "s"(sreg));
        if (OP == XOR_B32_LIT) asm volatile("v_xor_b32 %0, 0x9e3779b9, %0" : "+v"(a[i]));
        if (OP == MUL_F32_S) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(f[i]) : "s"(sc));
        if (OP == MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(u1));
        if (OP == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(u1));
        if (OP == FMAC_F32) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[i]) : "v"(c1), "v"(c2));
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
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {  // one stamp pair per wave
    const unsigned w = blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[2 * w] = t1 - t0;
    stamps[2 * w + 1] = r1 - r0;
  }
  uint32_t s = sreg;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + (uint32_t)f[i] + (uint32_t)d[i] + (uint32_t)p[i];
  s += (uint32_t)q4.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += (uint32_t)g[i].x + (uint32_t)g[i].y;
  if (s == 0x12345678u) out[threadIdx.x] = s;
}

struct Cost {
  double ns, cyc, ghz;  // ns and shader cycles of SIMD time per wave-instruction; in-kernel clock of that run
};

static unsigned long long *g_stamps;  // device, 2 per wave

template <int OP>
static Cost run(uint32_t *out, int blocks_per_cu, int cus) {
  const int blocks = cus * blocks_per_cu;  // blocks of 4 waves, one per SIMD -> blocks_per_cu waves per SIMD
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) k<OP><<<blocks, 256>>>(out, 1, g_stamps);  // (warm-up: lets the clock settle under this load)
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) k<OP><<<blocks, 256>>>(out, 1, g_stamps);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const int waves = blocks * 4;
  std::vector<unsigned long long> st(2 * waves);
  hipMemcpy(st.data(), g_stamps, st.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cyc(waves), clk(waves);
  for (int w = 0; w < waves; ++w) {
    cyc[w] = (double)st[2 * w];
    clk[w] = (double)st[2 * w] / (double)st[2 * w + 1] * 0.1;  // GHz: memrealtime ticks at 100 MHz
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(clk.begin(), clk.end());
  const double n_inst = (double)ITERS * REP * 8;  // wave-instructions (mixes: groups) per wave
  Cost c;
  c.ns = ms / 5 * 1e6 / (n_inst * blocks_per_cu);
  // SIMD cycles per wave-instruction = the launch's wall time per instruction and SIMD (HIP events) x the clock the waves
  // measured inside it.  (The waves' own s_memtime spans are NOT used for this: a SIMD's waves need not be alive all at
  // once - tools/residency_probe.hip: 3.4 of 4 and 6.1 of 8 on average - so a span divided by the nominal residency reads low.)
  c.cyc = c.ns * clk[waves / 2];
  c.ghz = clk[waves / 2];
  hipEventDestroy(a);
  hipEventDestroy(b);
  return c;
}

template <int OP>
static void all(uint32_t *out, Cost *res8, Cost *res4, Cost *res2, Cost *res1, int cus) {
  res8[OP] = run<OP>(out, 8, cus);
  res4[OP] = run<OP>(out, 4, cus);
  res2[OP] = run<OP>(out, 2, cus);
  res1[OP] = run<OP>(out, 1, cus);
  if constexpr (OP + 1 < N_OPS) all<OP + 1>(out, res8, res4, res2, res1, cus);
}

static void dump(const char *key, const Cost *r, const char *tail) {
  printf("  \"%s\": {", key);
  for (int i = 0; i < N_OPS; ++i)
    printf("%s\"%s\": {\"ns\": %.4f, \"cycles\": %.3f, \"clock_ghz\": %.3f}", i ? ", " : "", kNames[i], r[i].ns, r[i].cyc, r[i].ghz);
  printf("}%s\n", tail);
}

int main() {
  uint32_t *out;
  hipMalloc(&out, 4096);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  hipMalloc(&g_stamps, (size_t)cus * 8 * 4 * 2 * 8);
  static Cost r8[N_OPS], r4[N_OPS], r2[N_OPS], r1[N_OPS];
  all<0>(out, r8, r4, r2, r1, cus);
  printf("{\n  \"device\": \"%s\", \"compute_units\": %d, \"unit\": \"SIMD time per wave64 instruction: ns (HIP events over the launch / instructions per SIMD), the in-kernel "
         "shader clock of that launch (s_memtime / s_memrealtime x 100 MHz, median over waves) and cycles = ns x clock\",\n",
         prop.gcnArchName, cus);
  printf("  \"method\": \"tools/issue_cost.hip: %d x 64 inline-asm instructions per lane, 8 independent accumulators, 256-thread "
         "blocks, N blocks per CU = N wavefronts per SIMD\",\n", ITERS);
  dump("waves_per_simd_8", r8, ",");
  dump("waves_per_simd_4", r4, ",");
  dump("waves_per_simd_2", r2, ",");
  dump("waves_per_simd_1", r1, "");
  printf("}\n");
  hipFree(out);
  hipFree(g_stamps);
  return 0;
}
