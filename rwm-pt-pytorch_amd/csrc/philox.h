// Philox4x32-10 counter-based RNG and raw-bits -> variate transforms for gfx950.
//
// The reference draws its randoms with torch.randn / torch.rand ahead of the
// loop (algorithms/rwm_gpu_optimized.py:490-511, algorithms/pt_rwm_gpu_optimized.py:710-723);
// at 65 536 chains x 32 temperatures that is not storable, so each
// (chain, temperature, step) derives its own Philox block instead.
//
// Counter layout (shared with oracle/ptrwm_oracle.c, which restates it):
//   c0 = block index within the step | (step >> 32) << 16
//   c1 = step (low 32 bits, 0-based)
//   c2 = global chain id (low 32 bits)
//   c3 = temperature index | stream << 8 | (global chain id >> 32) << 12
//   key = (seed low, seed high)
// stream 0 = MH proposal + accept draws, stream 1 = swap uniforms.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptrwm {

struct u32x4 {
  uint32_t x, y, z, w;
};

constexpr uint32_t kPhiloxM0 = 0xD2511F53u;
constexpr uint32_t kPhiloxM1 = 0xCD9E8D57u;
constexpr uint32_t kPhiloxW0 = 0x9E3779B9u;
constexpr uint32_t kPhiloxW1 = 0xBB67AE85u;

constexpr uint32_t kStreamMH = 0u;
constexpr uint32_t kStreamSwap = 1u;

// One Philox4x32-10 block.  The key schedule is wave-uniform (seed only), so the
// ten round keys live in SGPRs; per round the VALU work is two 32x32->64
// multiplies (v_mad_u64_u32) and two three-way xors.
__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)kPhiloxM0 * c0;
    const uint64_t p1 = (uint64_t)kPhiloxM1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += kPhiloxW0;
    k1 += kPhiloxW1;
  }
  return u32x4{c0, c1, c2, c3};
}

// top 24 bits -> [0, 1): same lattice as torch.rand(float32)
__device__ __forceinline__ float u01(uint32_t r) { return (float)(r >> 8) * 0x1p-24f; }
// top 24 bits -> (0, 1]
__device__ __forceinline__ float u01_open0(uint32_t r) {
  return ((float)(r >> 8) + 1.0f) * 0x1p-24f;
}

constexpr float kLn2 = 0.69314718055994530942f;
constexpr float kLog2e = 1.44269504088896340736f;

// hardware transcendental wrappers (v_exp_f32 / v_log_f32 are base 2)
__device__ __forceinline__ float hw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float hw_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float hw_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float hw_exp(float x) { return hw_exp2(x * kLog2e); }
__device__ __forceinline__ float hw_ln(float x) { return hw_log2(x) * kLn2; }

// Box-Muller on two raw words.  v_sin_f32 / v_cos_f32 take their argument in
// revolutions, so sin(2*pi*u2) is one instruction.
__device__ __forceinline__ void box_muller(uint32_t ra, uint32_t rb, float &z0, float &z1) {
  const float u1 = u01_open0(ra);
  const float u2 = u01(rb);
  const float rad = hw_sqrt((-2.0f * kLn2) * hw_log2(u1));
  z0 = rad * __builtin_amdgcn_sinf(u2);
  z1 = rad * __builtin_amdgcn_cosf(u2);
}

// Scheduling fence.  Every loop over the dim-vector is fully unrolled; without fences hipcc's
// machine scheduler hoists all Philox blocks of a step (they depend only on counters) ahead of
// their consumers and the live set spills to scratch.  A fence every few dimensions keeps the live
// set at x[] + y[] + one chunk of temporaries; latency is hidden by the 3-6 resident waves per SIMD.
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// IEEE single ops that the compiler must not contract into an fma: the state
// update x + scale*z is then bit-identical to the reference's two torch ops.
// (HIP's __fmul_rn/__fadd_rn are plain operators and do get contracted under the default
// -ffp-contract=fast, hence the pragma.)
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
__device__ __forceinline__ float div_rn(float a, float b) {
#pragma clang fp contract(off)
  return a / b;
}

}  // namespace ptrwm
