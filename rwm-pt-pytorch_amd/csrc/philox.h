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

// 32x32 -> 64 multiply by a round constant: one v_mad_u64_u32 (both halves, ~7 cycles per wave at 4 waves/SIMD).
// Splitting it into v_mul_hi_u32 + v_mul_lo_u32 was measured SLOWER (v_mul_lo_u32 is quarter rate: 2.4 + 8.5
// cycles; whole kernel 7.33 ms vs 6.81 ms, profiles/r01_bench_variants.txt), so the fused form stays.
__device__ __forceinline__ void mul_hilo(uint32_t m, uint32_t x, uint32_t &hi, uint32_t &lo) {
  const uint64_t p = (uint64_t)m * x;
  hi = (uint32_t)(p >> 32);
  lo = (uint32_t)p;
}

// One Philox4x32-10 block.  The key schedule is wave-uniform (seed only), so the
// ten round keys live in SGPRs; per round the VALU work is two 32x32->64
// multiplies (v_mad_u64_u32) and two three-way xors (v_bitop3_b32).
__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0, lo0, hi1, lo1;
    // (c0 of the first round is the block index: wave-uniform in the step kernel, so the compiler turns that
    // product into two SALU multiplies)
    mul_hilo(kPhiloxM0, c0, hi0, lo0);
    mul_hilo(kPhiloxM1, c2, hi1, lo1);
    uint32_t n0, n2;
    if (r >= 2) {
      // one v_bitop3_b32 (0x96 = a ^ b ^ c) per three-way xor; hipcc does not form it by itself when one operand
      // is an SGPR (the round key) and emits two v_xor_b32: 160 extra VALU instructions per dim-30 step
      n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96);
      n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
    } else {
      // rounds 0 and 1 still see wave-uniform inputs in the step kernel (step, block index and their products):
      // plain xors let the compiler fold those parts on the scalar unit
      n0 = hi1 ^ c1 ^ k0;
      n2 = hi0 ^ c3 ^ k1;
    }
    c1 = lo1;
    c3 = lo0;
    c0 = n0;
    c2 = n2;
    k0 += kPhiloxW0;
    k1 += kPhiloxW1;
  }
  return u32x4{c0, c1, c2, c3};
}

// Value barriers.  PTRWM_VALUE_BARRIER("+s" / "+v", x) is an EMPTY inline asm that names x as read and written: no
// instruction, but the optimiser must treat x as a new value from here on - it can neither hoist what is computed from it
// out of the step loop nor share it with the code before.  Every use in this library is of that kind (kernel.h thread_index_now
// / fresh_dim / late_args, uniform_vec below, quad.h q_fresh, targets.h HybridRosenbrock) and exists to keep loop-invariant
// values out of registers - i.e. out of spill lanes - across the ~1 500-instruction step loop.  Building with
// -DPTRWM_NO_VALUE_BARRIERS removes all of them (analysis only: tools/barrier_diff.py compiles both ways and records what
// they do to registers, spills and scratch, profiles/r03_barrier_diff.txt).
#ifdef PTRWM_NO_VALUE_BARRIERS
#define PTRWM_VALUE_BARRIER(...) ((void)0)
#else
#define PTRWM_VALUE_BARRIER(...) asm volatile("" : __VA_ARGS__)
#endif

// A comment in the emitted assembly marking a block that an ordinary step does not enter (tools/issue_model.py leaves such
// blocks off the step path it prices); no instruction.
#define PTRWM_COLD_PATH() asm volatile("; ptrwm-cold-path")

// Parameter vectors (means, per-dimension scales) are read-only for the whole launch and indexed wave-uniformly.
// Read through the constant address space they become scalar loads (s_load_dwordx*, SGPR operands) instead of
// 64-lane vector loads of a single address.  (No kernel writes them, so the scalar cache cannot go stale.)
// The pointer is passed through an empty asm once per evaluation so the loads stay inside the step loop: hoisted
// out of it as loop invariants they would pin dozens of SGPRs for the whole launch (and spill).
typedef const __attribute__((address_space(4))) float *const_float_ptr;
__device__ __forceinline__ const_float_ptr uniform_vec(const float *p) {
  uintptr_t v = (uintptr_t)p;
  PTRWM_VALUE_BARRIER("+s"(v));
  return (const_float_ptr)v;
}

// the same pointer, re-materialised: scalar loads through it cannot be merged with, or hoisted above, loads issued before
// this point - used every few dimensions to bound how many parameter words are in flight (in SGPRs) at once
__device__ __forceinline__ const_float_ptr uniform_vec_again(const_float_ptr p) {
  uintptr_t v = (uintptr_t)p;
  PTRWM_VALUE_BARRIER("+s"(v));
  return (const_float_ptr)v;
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

// Box-Muller on two raw words (ra -> radius, rb -> angle).
//   radius: u1 = ((ra >> 8) + 1) * 2^-24 in (0, 1], r^2 = -2 ln u1.  (Folding the 2^-24 into an fma behind the
//           log saves a multiply but cancels near u1 = 1 and costs 1e-5 sigma of absolute accuracy: not done.)
//   angle:  v_sin_f32 / v_cos_f32 take their argument in revolutions and are periodic, so the low 23 bits of rb
//           dropped into the mantissa of a float in [1, 2) ARE the angle (one v_and_or_b32, no cvt, no scaling).
__device__ __forceinline__ float bm_radius_sq(uint32_t ra) { return (-2.0f * kLn2) * hw_log2(u01_open0(ra)); }
__device__ __forceinline__ float bm_turns(uint32_t rb) { return __uint_as_float((rb & 0x007FFFFFu) | 0x3F800000u); }

__device__ __forceinline__ void box_muller(uint32_t ra, uint32_t rb, float &z0, float &z1) {
  const float rad = hw_sqrt(bm_radius_sq(ra));
  const float ang = bm_turns(rb);
  z0 = rad * __builtin_amdgcn_sinf(ang);
  z1 = rad * __builtin_amdgcn_cosf(ang);
}

// Scheduling fence.  Every loop over the dim-vector is fully unrolled; without fences hipcc's
// machine scheduler hoists all Philox blocks of a step (they depend only on counters) ahead of
// their consumers and the live set spills to scratch.  A fence every few dimensions keeps the live
// set at x[] + y[] + one chunk of temporaries; latency is hidden by the 3-6 resident waves per SIMD.
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// fences outside the Philox blocks can be switched off for tuning experiments (PTRWM_FENCE_RNG_ONLY)
__device__ __forceinline__ void sched_fence_soft() {
#ifndef PTRWM_FENCE_RNG_ONLY
  __builtin_amdgcn_sched_barrier(0);
#endif
}

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

// ---- loops over the dimensions of a replica -----------------------------------------------------------------------
// PTRWM_DIM_LOOP(d, DP, D, body): `body` for every d < D, with d a compile-time constant after unrolling (static register
// indices).  When D is a compile-time constant (kernels with dim compiled in) everything folds to the plain unrolled
// loop.  When D is a run-time, wave-uniform value the dimensions are walked in blocks of four: a block entirely below D
// runs without per-dimension tests, only the one block that straddles D tests each dimension, blocks above are skipped
// - one scalar compare-and-branch per four dimensions instead of one per dimension (the generic-width kernels spent
// ~110 scalar branches per step on those).  The body is instantiated twice per block in that case.
#define PTRWM_DIM_LOOP(d, DP, D, ...)                                                          \
  _Pragma("unroll") for (int d##_blk = 0; d##_blk < (DP); d##_blk += 4) {                      \
    if (d##_blk + 4 <= (D)) {                                                                  \
      _Pragma("unroll") for (int d = d##_blk; d < d##_blk + 4; ++d)                            \
        if (d < (DP)) { __VA_ARGS__ }                                                          \
    } else if (d##_blk < (D)) {                                                                \
      _Pragma("unroll") for (int d = d##_blk; d < d##_blk + 4; ++d)                            \
        if (d < (DP) && d < (D)) { __VA_ARGS__ }                                               \
    }                                                                                          \
  }

// ---- canonical reduction order -------------------------------------------------------------------------------
// Every sum over the dimensions of a replica (log-density terms, squared jump, the UniformRadius norm) is taken in ONE
// order, whatever kernel evaluates it: the dimensions are cut into FOUR contiguous ranges of canon_width() entries,
// each range is a sequential chain starting from the identity, and the four partial results are combined pairwise,
// (P0 + P1) + (P2 + P3).  The one-thread-per-replica kernel walks the four ranges itself; the lane-split kernel
// (kernel_quad.h) gives one range to each of the four lanes of a replica and combines with two DPP quad permutes.
// Same operations on the same operands in the same order => bit-identical results, so the choice of kernel (made by
// the C ABI from the batch size) can never change a trajectory.  The width depends only on the register-width class,
// which is a function of dim: dim <= 32 -> 8, dim <= 64 -> 16, dim <= 80 -> 20, dim <= 96 -> 24, dim <= 112 -> 28.
// (Above dim 64 the lane-split kernel is the only fused form and its cost follows 4 W, not dim: with a single class of
// W = 28 every dim 65..99 paid for 112 dimensions - a 2x drop in throughput from dim 64 to dim 65,
// profiles/r02_form_sweep.txt - hence the finer classes there.)
constexpr int canon_width(int dp) { return dp <= 32 ? 8 : (dp <= 64 ? 16 : (dp <= 80 ? 20 : (dp <= 96 ? 24 : 28))); }

__device__ __forceinline__ float tree4_add(const float (&p)[4]) { return add_rn(add_rn(p[0], p[1]), add_rn(p[2], p[3])); }
__device__ __forceinline__ float tree4_neg_add(const float (&p)[4]) { return -tree4_add(p); }

}  // namespace ptrwm
