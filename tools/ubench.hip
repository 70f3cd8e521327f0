// Micro-benchmark of the VALU instruction mix of the PT-RWM kernel on gfx950: relative issue cost of
// v_mad_u64_u32, v_mul_lo/hi_u32, transcendentals, packed fp32 and fp64 adds at 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define REP 64
#define ITERS 2000

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed) {
  uint32_t a[8];
  float f[8];
  double d[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = seed + threadIdx.x * 8 + i;
    f[i] = 0.5f + (float)(a[i] & 1023) * 1e-4f;
    d[i] = f[i];
  }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) f[i] = __builtin_fmaf(f[i], 1.0001f, 0.5f);
        if (OP == 1) { uint64_t p = (uint64_t)a[i] * 0xD2511F53u; a[i] = (uint32_t)p ^ (uint32_t)(p >> 32); }
        if (OP == 2) a[i] = a[i] * 0xD2511F53u + 1u;
        if (OP == 3) a[i] = __umulhi(a[i], 0xD2511F53u) + 1u;
        if (OP == 4) f[i] = __builtin_amdgcn_exp2f(f[i]) * 0.25f;
        if (OP == 5) f[i] = __builtin_amdgcn_logf(f[i]) + 2.0f;
        if (OP == 6) f[i] = __builtin_amdgcn_sinf(f[i]) + 0.7f;
        if (OP == 7) f[i] = __builtin_amdgcn_sqrtf(f[i]) + 0.1f;
        if (OP == 8) a[i] = a[i] ^ (a[i] >> 3) ^ seed;
        if (OP == 9) d[i] = d[i] + 1.000001;
        if (OP == 10) a[i] = __umul24(a[i] & 0xffff, 0x1F53) + a[i];
        if (OP == 11) f[i] = __builtin_amdgcn_rcpf(f[i]) + 0.3f;
        if (OP == 12) a[i] = __builtin_amdgcn_ds_bpermute((int)((threadIdx.x * 4 + 4) & 255), (int)a[i]);
        // mixes: does a transcendental overlap with full-rate work of the SAME wave / of other waves?
        if (OP == 13) { f[i] = __builtin_amdgcn_exp2f(f[i]); a[i] = (a[i] ^ seed) + 3u; a[i] = (a[i] >> 1) ^ a[i]; a[i] += a[i] >> 3; }
        if (OP == 14) { a[i] = (a[i] ^ seed) + 3u; a[i] = (a[i] >> 1) ^ a[i]; a[i] += a[i] >> 3; }
        if (OP == 15) { f[i] = __builtin_amdgcn_exp2f(f[i]); }
        if (OP == 16) { uint32_t h, l; asm("v_mul_hi_u32 %0, %1, %2" : "=v"(h) : "s"(0xD2511F53u), "v"(a[i])); asm("v_mul_lo_u32 %0, %1, %2" : "=v"(l) : "s"(0xD2511F53u), "v"(a[i])); a[i] = h ^ l; }
        if (OP == 17) { uint32_t h, l; asm("v_mul_hi_u32 %0, %1, %2" : "=v"(h) : "s"(0xD2511F53u), "v"(a[i])); asm("v_mul_lo_u32 %0, %1, %2" : "=v"(l) : "s"(0xD2511F53u), "v"(a[i])); a[i] = h ^ l; f[i] = __builtin_amdgcn_exp2f(f[i]); }
      }
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + (uint32_t)f[i] + (uint32_t)d[i];
  if (s == 0x12345678u) out[threadIdx.x] = s;
}

// packed fp32: 2 lanes of work per instruction
__global__ void __launch_bounds__(256) kpk(uint32_t *out, uint32_t seed) {
  typedef float float2_ __attribute__((ext_vector_type(2)));
  float2_ f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = float2_{0.5f + threadIdx.x * 1e-4f, 0.25f + i};
  const float2_ m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = __builtin_elementwise_fma(f[i], m, c);
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += f[i].x + f[i].y;
  if (s == 1234.5f) out[threadIdx.x] = (uint32_t)s;
}

template <class K>
static double run(K kern, const char *name, double base) {
  uint32_t *out;
  hipMalloc(&out, 4096);
  const int blocks = 256 * 4;  // 4 blocks of 4 waves per CU -> 4 waves per SIMD
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  kern<<<blocks, 256>>>(out, 1);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) kern<<<blocks, 256>>>(out, 1);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  ms /= 5;
  // wave-instructions per SIMD = ITERS*REP * 4 waves
  const double ns_per_inst = ms * 1e6 / ((double)ITERS * REP * 4);
  printf("%-28s %8.3f ms  %6.3f ns per wave-instr per SIMD  (x%.2f of v_fma_f32; %.2f cyc @2.4GHz)\n", name, ms,
         ns_per_inst, base > 0 ? ns_per_inst / base : 1.0, ns_per_inst * 2.4);
  hipFree(out);
  return ns_per_inst;
}

int main() {
  double base = run(k<0>, "v_fma_f32", 0);
  run(k<1>, "v_mad_u64_u32 (+xor)", base);
  run(k<2>, "v_mul_lo_u32 (+add)", base);
  run(k<3>, "v_mul_hi_u32 (+add)", base);
  run(k<4>, "v_exp_f32 (+mul)", base);
  run(k<5>, "v_log_f32 (+add)", base);
  run(k<6>, "v_sin_f32 (+add)", base);
  run(k<7>, "v_sqrt_f32 (+add)", base);
  run(k<8>, "v_xor x2 + shift", base);
  run(k<9>, "v_add_f64", base);
  run(k<10>, "v_mad_u32_u24 (+and)", base);
  run(k<11>, "v_rcp_f32 (+add)", base);
  run(k<12>, "ds_bpermute_b32", base);
  run(kpk, "v_pk_fma_f32 (2 fma)", base);
  printf("-- mixes (cost per loop body, not per instruction: multiply ns by instructions in the body)\n");
  run(k<14>, "A: 6 int ops", base);
  run(k<15>, "B: 1 exp", base);
  run(k<13>, "A+B in one wave", base);
  run(k<16>, "C: mul_hi+mul_lo+xor", base);
  run(k<17>, "C + 1 exp", base);
  return 0;
}
