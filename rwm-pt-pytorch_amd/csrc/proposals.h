// Proposal increment samplers as device functors.
//
// Each functor fills y = x + increment for one (chain, temperature) replica and
// returns the accept uniform.  Raw randoms come either from the replica's
// Philox blocks for this step, or (fixture mode) from caller-provided arrays of
// the same raw quantities the reference draws with torch.randn / torch.rand.
//
// Raw-word index map inside one step (stream 0), shared with the oracle:
//   NORMAL          dims pairwise via Box-Muller on words (2p, 2p+1);
//                   accept uniform = word 2*ceil(D/2)
//   LAPLACE         dim d uses word d; accept uniform = word D
//   UNIFORM_RADIUS  dims as NORMAL; radius uniform = word 2*ceil(D/2); accept = the next word
// word w lives in Philox block w/4, lane w%4.
#pragma once
#include "philox.h"
#include "../../include/ptrwm.h"

namespace ptrwm {

struct RngCtx {
  uint32_t c0hi, c1, c2, c3, k0, k1;
};

// What a proposal functor says about the squared length of the increment it just made (the Metropolis step's squared jump
// when the proposal is accepted).  The Philox paths know it without looking at y - x: a Box-Muller pair contributes
// rad^2 (sin^2 + cos^2) = the very argument of its square root, and a UniformRadius increment has length r by
// construction.  kJumpNone: compute it from the states (external randoms, Laplace).  kJumpTotal: `jump` is the squared
// jump (sums over Box-Muller pairs are taken in the canonical four-range order, philox.h, and combined before the functor
// returns - one live register across the density evaluation, not four).  kJumpPartial (lane-split form only): `jump` is
// this lane's range, to be combined across the quad by the caller.
enum { kJumpNone = 0, kJumpPartial = 1, kJumpTotal = 2 };
// The reference's statistic is |x_t - x_{t-1}|^2 of the STORED states (rwm_gpu_optimized.py:513-534,
// pt_rwm_gpu_optimized.py:772-789).  The increment's own length stands in for it only where the float sum x + inc keeps the
// increment: a replica whose largest coordinate exceeds kJumpTrust typical increments per dimension (half an ulp of x is
// then more than 2^-16 of the increment: the two definitions part beyond ~3e-5 relative) takes its jump from the states,
// as before.  Decided per replica, from its own state and scale only (so neither the kernel form nor the sharding can
// change it), when a launch loads the state: `jump_trusted()` of the kernels.
constexpr float kJumpTrust = 256.0f;

struct PParams {
  const float *__restrict__ dim_scale;  // [D] Laplace, wave-uniform
  float inv_dim;
};

__device__ __forceinline__ uint32_t pick(const u32x4 &r, int lane4) {
  return lane4 == 0 ? r.x : (lane4 == 1 ? r.y : (lane4 == 2 ? r.z : r.w));
}

// Fills z[0..D) with standard normals (Box-Muller pairs) and returns the Philox
// words at raw indices `w_a` and `w_a + 1` (as [0,1) uniforms) for the caller.
template <int DP>
__device__ __forceinline__ void philox_normals(float (&z)[DP], int D, const RngCtx &rc, int w_a,
                                               float &ua, float &ub) {
  constexpr int NB = (2 * ((DP + 1) / 2)) / 4 + 1;  // blocks up to the one holding word 2*ceil(DP/2)
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    if (4 * c + 4 <= DP && 4 * c + 4 <= D) {
      // a block entirely below dim: no per-pair tests (see PTRWM_DIM_LOOP); w_a >= D lies in a later block
      const u32x4 r = philox4x32_10(rc.c0hi | (uint32_t)c, rc.c1, rc.c2, rc.c3, rc.k0, rc.k1);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int d = (4 * c + 2 * h + 1 < DP) ? 4 * c + 2 * h : 0;
        float z0, z1;
        box_muller(h ? r.z : r.x, h ? r.w : r.y, z0, z1);
        z[d] = z0;
        z[d + 1] = z1;
      }
      sched_fence();
    } else if (4 * c <= w_a + 1) {
      const u32x4 r = philox4x32_10(rc.c0hi | (uint32_t)c, rc.c1, rc.c2, rc.c3, rc.k0, rc.k1);
      if (4 * c < DP && 4 * c < D) {
        float z0, z1;
        box_muller(r.x, r.y, z0, z1);
        z[4 * c < DP ? 4 * c : 0] = z0;
        if (4 * c + 1 < DP && 4 * c + 1 < D) z[4 * c + 1 < DP ? 4 * c + 1 : 0] = z1;
      }
      if (4 * c + 2 < DP && 4 * c + 2 < D) {
        float z0, z1;
        box_muller(r.z, r.w, z0, z1);
        z[4 * c + 2 < DP ? 4 * c + 2 : 0] = z0;
        if (4 * c + 3 < DP && 4 * c + 3 < D) z[4 * c + 3 < DP ? 4 * c + 3 : 0] = z1;
      }
      // w_a is even by construction (2*ceil(D/2)), so it is lane 0 or 2 of its block
      if (4 * c == w_a) {
        ua = u01(r.x);
        ub = u01(r.y);
      }
      if (4 * c + 2 == w_a) {
        ua = u01(r.z);
        ub = u01(r.w);
      }
      sched_fence();
    }
  }
}

// Philox path of the Normal proposal: y = x + (scale * radius) * {sin, cos}: the per-temperature scale is folded
// into the Box-Muller radius (one multiply per pair instead of one per dimension) and the add is an fma.
// Differs from "z, then z * scale, then x + inc" by <= 1 ulp of y, below the hardware sin/cos error; the
// external-randoms path keeps the reference's exact two-op form.
// The squared jump comes with it: rad^2 of every pair both of whose dimensions exist - which is c_t log2(u), the argument of
// the pair's square root, with c_t = -2 ln 2 tscale^2 folded once per step - summed per canonical range (a pair never
// straddles two ranges: W is even); the last dimension of an odd dim contributes (rad sin)^2.  One add per pair instead of
// a subtraction and an fma per dimension; within 1e-6 relative of |y - x|^2 (the rounding of x + inc and of sin^2 + cos^2).
template <int DP>
__device__ __forceinline__ float philox_normal_step(float (&y)[DP], const float (&x)[DP], int D, float tscale,
                                                    const RngCtx &rc, float &jump_total) {
  constexpr int W = canon_width(DP);
  float jump[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  const float c_t = mul_rn(mul_rn(-2.0f * kLn2, tscale), tscale);
  constexpr int NB = (2 * ((DP + 1) / 2)) / 4 + 1;  // blocks up to the one holding word 2*ceil(DP/2)
#ifdef PTRWM_INJECT_ACCEPT_WORD_SLIP
  // FAULT INJECTION (tools/inject_slip_check.sh only, never in a shipped build): the accept uniform is taken one pair
  // early - the radius word of the last Box-Muller pair - so the Philox-mode parity tests can be shown to fail on it
  const int w_a = 2 * ((D + 1) >> 1) - 2;
#else
  const int w_a = 2 * ((D + 1) >> 1);
#endif
  float u_acc = 0.0f;
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    if (4 * c + 4 <= DP && 4 * c + 4 <= D) {
      // a block entirely below dim: no per-pair tests (run-time dim: one scalar branch per block, see PTRWM_DIM_LOOP)
      const u32x4 r = philox4x32_10(rc.c0hi | (uint32_t)c, rc.c1, rc.c2, rc.c3, rc.k0, rc.k1);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int d = (4 * c + 2 * h + 1 < DP) ? 4 * c + 2 * h : 0;
        const uint32_t ra = h ? r.z : r.x, rb = h ? r.w : r.y;
        const float arg = mul_rn(c_t, hw_log2(u01_open0(ra)));
        const float rad = hw_sqrt(arg);
        const float ang = bm_turns(rb);
        y[d] = fmaf(rad, __builtin_amdgcn_sinf(ang), x[d]);
        y[d + 1] = fmaf(rad, __builtin_amdgcn_cosf(ang), x[d + 1]);
        jump[d / W] = add_rn(jump[d / W], arg);
      }
      sched_fence();
    } else if (4 * c <= w_a) {
      const u32x4 r = philox4x32_10(rc.c0hi | (uint32_t)c, rc.c1, rc.c2, rc.c3, rc.k0, rc.k1);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int d = 4 * c + 2 * h;
        const uint32_t ra = h ? r.z : r.x, rb = h ? r.w : r.y;
        if (d < DP && d < D) {
          const float arg = mul_rn(c_t, hw_log2(u01_open0(ra)));
          const float rad = hw_sqrt(arg);
          const float ang = bm_turns(rb);
          const int d0 = d < DP ? d : 0, d1 = d + 1 < DP ? d + 1 : 0;
          const float sn = __builtin_amdgcn_sinf(ang);
          y[d0] = fmaf(rad, sn, x[d0]);
          if (d + 1 < DP && d + 1 < D) {
            y[d1] = fmaf(rad, __builtin_amdgcn_cosf(ang), x[d1]);
            jump[d0 / W] = add_rn(jump[d0 / W], arg);
          } else {
            const float i0 = mul_rn(rad, sn);  // the last dimension of an odd dim: half a pair
            jump[d0 / W] = fmaf(i0, i0, jump[d0 / W]);
          }
        }
        if (d == w_a) u_acc = u01(ra);
      }
      sched_fence();
    }
  }
  jump_total = tree4_add(jump);
  return u_acc;
}

// NormalProposal.sample, proposal_distributions/normal.py:33-36,46-55 (`randn * std`) and the
// diagonal-Cholesky bmm of the PT class, algorithms/pt_rwm_gpu_optimized.py:445-455,576-592.
template <int DP>
struct NormalProposal {
  static constexpr int kKind = PTRWM_PROPOSAL_NORMAL;
  static constexpr bool kKnowsJump = true;
  // the size of a typical per-dimension increment (for jump_trusted)
  __device__ __forceinline__ static float increment_scale(float tscale, const PParams &) { return tscale; }
  __device__ __forceinline__ static float propose(float (&y)[DP], const float (&x)[DP], int D,
                                                  float tscale, const PParams &, const RngCtx &rc,
                                                  const float *ext_raw, float ext_u, float &jump, int &jump_kind) {
    float u_acc = ext_u;
    if (ext_raw != nullptr) {
      jump_kind = kJumpNone;
#pragma unroll
      for (int d = 0; d < DP; ++d)
        if (d < D) y[d] = add_rn(x[d], mul_rn(ext_raw[d], tscale));
    } else {
      jump_kind = kJumpTotal;
      u_acc = philox_normal_step<DP>(y, x, D, tscale, rc, jump);
    }
    return u_acc;
  }
};

// LaplaceProposal.sample, proposal_distributions/laplace.py:24-37,46-69:
//   u = rand - 0.5;  x = -scale * sign(u) * log1p(clamp(-2|u|, min=-0.999999))
template <int DP>
struct LaplaceProposal {
  static constexpr int kKind = PTRWM_PROPOSAL_LAPLACE;
  static constexpr bool kKnowsJump = false;
  __device__ __forceinline__ static float increment_scale(float tscale, const PParams &) { return tscale; }
  __device__ __forceinline__ static float transform(float u01v, float scale) {
    const float u = u01v - 0.5f;
    const float au = __builtin_fabsf(u);
    const float arg = __builtin_fmaxf(-2.0f * au, -0.999999f);
    const float l1p = hw_log2(1.0f + arg) * kLn2;
    // (-scale * sign(u)) * l1p: the product with sign(u) is exact, so this is scale*l1p with its sign flipped when
    // u > 0 (and a zero when u == 0, where l1p == 0): one sign-bit operation instead of two compares, two selects
    // and a multiply.  (~u has its sign bit set exactly when u >= +0.)
    const float t = mul_rn(scale, l1p);
    // v_bitop3_b32 truth table for a ^ (~b & c) with a = 0xf0, b = 0xcc, c = 0xaa: 0xf0 ^ (0x33 & 0xaa) = 0xd2
    return __builtin_bit_cast(float, __builtin_amdgcn_bitop3_b32(__builtin_bit_cast(uint32_t, t),
                                                                 __builtin_bit_cast(uint32_t, u), 0x80000000u, 0xd2));
  }
  __device__ __forceinline__ static float propose(float (&y)[DP], const float (&x)[DP], int D,
                                                  float tscale, const PParams &pp, const RngCtx &rc,
                                                  const float *ext_raw, float ext_u, float &, int &jump_kind) {
    float u_acc = ext_u;
    jump_kind = kJumpNone;
    const const_float_ptr dsc = uniform_vec(pp.dim_scale);
    if (ext_raw != nullptr) {
#pragma unroll
      for (int d = 0; d < DP; ++d)
        if (d < D) y[d] = add_rn(x[d], transform(ext_raw[d], mul_rn(dsc[d], tscale)));
    } else {
      constexpr int NB = DP / 4 + 1;
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        if (4 * c + 4 <= DP && 4 * c + 4 <= D) {
          // a block entirely below dim: no per-dimension tests (see PTRWM_DIM_LOOP)
          const u32x4 r = philox4x32_10(rc.c0hi | (uint32_t)c, rc.c1, rc.c2, rc.c3, rc.k0, rc.k1);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int d = (4 * c + q < DP) ? 4 * c + q : 0;
            y[d] = add_rn(x[d], transform(u01(pick(r, q)), mul_rn(dsc[d], tscale)));
          }
          sched_fence();
        } else if (4 * c <= D) {
          const u32x4 r = philox4x32_10(rc.c0hi | (uint32_t)c, rc.c1, rc.c2, rc.c3, rc.k0, rc.k1);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int d = 4 * c + q;
            if (d < DP && d < D) {
              const int ds = d < DP ? d : 0;
              y[ds] = add_rn(x[ds], transform(u01(pick(r, q)), mul_rn(dsc[ds], tscale)));
            }
            if (d == D) u_acc = u01(pick(r, q));
          }
          sched_fence();
        }
      }
    }
    return u_acc;
  }
};

// UniformRadiusProposal.sample, proposal_distributions/uniform.py:27-37,47-73:
//   g = randn(D); n = |g| (1 if <= 1e-12); r = R_eff * U^(1/D); inc = g / n * r
template <int DP>
struct UniformRadiusProposal {
  static constexpr int kKind = PTRWM_PROPOSAL_UNIFORM_RADIUS;
  static constexpr bool kKnowsJump = true;
  __device__ __forceinline__ static float increment_scale(float tscale, const PParams &pp) { return tscale * hw_sqrt(pp.inv_dim); }
  __device__ __forceinline__ static float propose(float (&y)[DP], const float (&x)[DP], int D,
                                                  float tscale, const PParams &pp, const RngCtx &rc,
                                                  const float *ext_raw, float ext_u, float &jump, int &jump_kind) {
    float u_acc = ext_u, u_rad;
    if (ext_raw != nullptr) {
#pragma unroll
      for (int d = 0; d < DP; ++d)
        if (d < D) y[d] = ext_raw[d];
      u_rad = ext_raw[D];
    } else {
      float ua = 0.0f, ub = 0.0f;
      philox_normals<DP>(y, D, rc, 2 * ((D + 1) >> 1), ua, ub);
      u_rad = ua;
      u_acc = ub;
    }
    constexpr int W = canon_width(DP);
    float n2p[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // |g|^2 in the canonical four-range order (philox.h)
    PTRWM_DIM_LOOP(d, DP, D, { n2p[d / W] = fmaf(y[d], y[d], n2p[d / W]); })
    const float nrm_sq = tree4_add(n2p);
    const float nrm = hw_sqrt(nrm_sq);
    const float safe = nrm > 1e-12f ? nrm : 1.0f;
    const float rad = tscale * hw_exp2(pp.inv_dim * hw_log2(u_rad));
    // g / n as g * (1/n) with one IEEE reciprocal per step: <= 1.5 ulp from the reference's per-element division
    // (uniform.py:58), and ~10 VALU instructions per dimension cheaper
    const float inv = div_rn(1.0f, safe);
    // per-dimension tests here, not PTRWM_DIM_LOOP: with the two instances of this in-place update per block the
    // optimiser merges them into one with a selected INDEX, which moves y[] to scratch (tools/kernel_stats.py --check)
    if (ext_raw != nullptr) {
      // external randoms: the reference's own operation order, (g / n) * r, then x + increment (uniform.py:58-73)
      jump_kind = kJumpNone;
#pragma unroll
      for (int d = 0; d < DP; ++d)
        if (d < D) y[d] = add_rn(x[d], mul_rn(mul_rn(y[d], inv), rad));
    } else {
      // Philox path (as philox_normal_step folds the scale into the Box-Muller radius): y = x + g (r / n) as ONE fma per
      // dimension instead of two multiplies and an add; differs from the form above by <= 1 ulp of the increment, below
      // the hardware sin / cos error of g itself
      const float k = mul_rn(inv, rad);
#pragma unroll
      for (int d = 0; d < DP; ++d)
        if (d < D) y[d] = fmaf(y[d], k, x[d]);
      // |g k|^2 = |g|^2 k^2: the squared jump without another pass over the dimensions
      jump_kind = kJumpTotal;
      jump = mul_rn(mul_rn(nrm_sq, k), k);
    }
    return u_acc;
  }
};

}  // namespace ptrwm
