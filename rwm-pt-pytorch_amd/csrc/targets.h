// Target log-densities as device functors over a register-resident dim-vector.
//
// Each functor restates, in fp32, the `log_density` of one reference class
// (citations per functor).  `y` is the proposal held in VGPRs (DP = compiled
// register width >= dim); every loop is fully unrolled so all indexing is static
// and the per-dimension parameter vectors are wave-uniform scalar loads.
#pragma once
// Evaluation functions with an `a * b +- c` pattern turn implicit FMA contraction OFF (explicit fmaf calls stay); the
// two mixtures write their only such product, the optional per-dimension scaling, as an explicit fma instead: the same functor is compiled into the fused step kernel, the
// stand-alone log-density kernel and both width groups, and left to the optimiser `a - b * b` became an fma in one of
// them and two roundings in another (a 1-ulp difference between ptrwm_logdensity and the in-loop value for
// FullRosenbrock, found by tools/fuzz_split.py).  Two roundings are also what the reference's torch expressions do.
#include "philox.h"
#include "../../include/ptrwm.h"

namespace ptrwm {

struct TParams {
  float p[12];
  int ip[4];
  const float *__restrict__ vec0;
  const float *__restrict__ vec1;
  unsigned long long mask[2];  // HybridRosenbrock: bit i set => coordinate i starts a block
};

constexpr float kNegInf = -__builtin_huge_valf();

// RoughCarpetDistributionTorch.log_density, target_distributions/multimodal_torch.py:470-510:
//   sum_d logsumexp_k( -0.5 (s_d x_d - m_k)^2 - log sqrt(2 pi) + log w_k ) + sum_d log s_d.
// Per dimension: the three exponents are formed directly from (x - m_k) (no
// expansion of the square, which would cancel catastrophically at the +-15
// modes), sorted with max3/med3/min3 so only two v_exp_f32 are needed, and the
// log of the 3-term sum is deferred: sum_d log2(s_d) = log2(prod_d s_d) with
// s_d in [1,3], two v_log_f32 per evaluation instead of one per dimension.  All
// sums over dimensions run in the canonical four-range order (philox.h).
// One dimension of the rough carpet from its three centred coordinates, each already scaled by kRcScale = sqrt(log2(e) / 2)
// so that the exponent of component k in the log2 domain is w_k - d_k^2: ONE fma per component (the caller forms d_k as one
// fma too: kRcScale x - kRcScale m_k; round 2 spent a subtraction, a square and an fma per component - three VALU
// instructions per dimension more, 8 % of the BASELINE step).  mx = the largest exponent, s = 1 + 2^(md - mx)
// (+ 2^(mn - mx)): the sum of the three terms relative to the largest.  Shared by both kernels.
// Accuracy: the square is still taken of the CENTRED coordinate (nothing is expanded); the only new rounding is that of the
// product kRcScale m_k, 6e-8 relative - for the component the point is close to it changes the exponent by
// 2 d_k 7.6e-7 ~ 0, for a far component by < 4e-5 in an exponent that no longer matters there.
constexpr float kRcScale = 0.84932180028801904f;  // sqrt(0.5 * log2(e))
template <bool STRICT, bool TWO>
__device__ __forceinline__ void rc_dim_term(float d0, float d1, float d2, float w0, float w1, float w2, float &mx, float &s) {
  const float a0 = fmaf(-d0, d0, w0);
  const float a1 = fmaf(-d1, d1, w1);
  const float a2 = fmaf(-d2, d2, w2);
  mx = __builtin_fmaxf(__builtin_fmaxf(a0, a1), a2);
  const float md = __builtin_amdgcn_fmed3f(a0, a1, a2);
  const float sh = STRICT ? __builtin_fmaxf(mx, -3.0e38f) : mx;
  s = 1.0f + hw_exp2(md - sh);
  if constexpr (!TWO) {
    const float mn = __builtin_fminf(__builtin_fminf(a0, a1), a2);
    s += hw_exp2(mn - sh);
  }
}

// scheduling knob: a fence after every (MASK + 1) dimensions of the rough-carpet loop.  The max-ILP schedule of the
// ~1 500-instruction step loop moves by several per cent with the placement of the fences; candidates are A/B-timed on
// one box (tools/ab_bench.sh).  With the canonical four-range sums a fence every 4 dimensions cost 3.5 % on BASELINE
// configs[2] (113.4 against 109.5 ms per 2 000-step launch); 8, 16 and no fences measured the same (109.2-109.5).
// Round 4: with the squared-jump sum parked in LDS the headline kernel needs 127 VGPRs and no scratch, and then a fence
// every 16 dimensions (or none) is 2.5 % faster than every 8 - 93.0 against 95.5 ms, and the same whatever the cadence
// of the update loop's fences (profiles/r04_scratch_ab.txt).
#ifndef PTRWM_RC_FENCE_MASK
#define PTRWM_RC_FENCE_MASK 15
#endif

template <int DP, bool TWO_TERM>
struct RoughCarpetT {
  static constexpr int kKind = PTRWM_TARGET_ROUGH_CARPET;
  // STRICT: all three exponents can be -inf only for |x| > ~1e19; torch.logsumexp then returns -inf where the
  // plain max-shift gives inf - inf = NaN.  The MH loop rejects either value, so only the stand-alone
  // log-density kernel pays for the guarded shift (one v_max per dimension).
  // TWO (kernel variant RoughCarpet2, chosen by capi.hip): the host has proved (rough_carpet_two_term) that for
  // every x the smallest of the three terms is below 2^-26 of the largest: adding it to a sum >= 1 cannot change
  // the fp32 result, so it (one v_exp_f32, min3, a subtract and an add per dimension) is dropped with
  // bit-identical output.  True for the +-15 modes of the benchmark target, false e.g. for modes +-4.
  template <bool SCALED, bool STRICT, bool TWO>
  __device__ __forceinline__ static float logp_impl(const float (&y)[DP], int D, const TParams &tp) {
    // the modes on the scaled axis, negated (the addend of the fma that centres a coordinate)
    const float m0 = -(tp.p[0] * kRcScale), m1 = -(tp.p[1] * kRcScale), m2 = -(tp.p[2] * kRcScale);
    [[maybe_unused]] const const_float_ptr uv0 = SCALED ? uniform_vec(tp.vec0) : nullptr;
    // log2-domain log-weights
    const float w0 = tp.p[3] * kLog2e, w1 = tp.p[4] * kLog2e, w2 = tp.p[5] * kLog2e;
    constexpr int W = canon_width(DP);
    float sm[4] = {0.0f, 0.0f, 0.0f, 0.0f}, pr[4] = {1.0f, 1.0f, 1.0f, 1.0f};  // canonical four-range partials (philox.h)
    PTRWM_DIM_LOOP(d, DP, D, {
      float d0, d1, d2;
      // kRcScale (s x - m_k) as ONE explicit fma per component, the same in every kernel
      const float sc = SCALED ? uv0[d] * kRcScale : kRcScale;
      d0 = fmaf(y[d], sc, m0), d1 = fmaf(y[d], sc, m1), d2 = fmaf(y[d], sc, m2);
      float mx, s;
      rc_dim_term<STRICT, TWO>(d0, d1, d2, w0, w1, w2, mx, s);
#ifdef PTRWM_RC_PROD_FIRST
      pr[d / W] = mul_rn(pr[d / W], s);
      sm[d / W] = add_rn(sm[d / W], mx);
#else
      sm[d / W] = add_rn(sm[d / W], mx);
      pr[d / W] = mul_rn(pr[d / W], s);
#endif
      if ((d & PTRWM_RC_FENCE_MASK) == PTRWM_RC_FENCE_MASK) sched_fence_soft();
    })
    const float sum_mx = tree4_add(sm);
    // each factor is in [1, 3] and a range holds at most 28 of them: the product of two ranges (<= 3^56) cannot
    // overflow, the product of all four could (3^112), so the log is taken per pair of ranges
    const float lg = add_rn(hw_log2(mul_rn(pr[0], pr[1])), hw_log2(mul_rn(pr[2], pr[3])));
    // p[6] = log_jacobian, p[7] = -dim * log(sqrt(2 pi)) folded on the host
    return add_rn(fmaf(add_rn(sum_mx, lg), kLn2, tp.p[7]), tp.p[6]);
  }
  template <bool STRICT = false>
  __device__ __forceinline__ static float logp(const float (&y)[DP], int D, const TParams &tp) {
    // wave-uniform branches per evaluation instead of per dimension
    constexpr bool two = TWO_TERM && !STRICT;
    return tp.vec0 != nullptr ? logp_impl<true, STRICT, two>(y, D, tp) : logp_impl<false, STRICT, two>(y, D, tp);
  }
};
template <int DP>
using RoughCarpet = RoughCarpetT<DP, false>;
template <int DP>
using RoughCarpet2 = RoughCarpetT<DP, true>;

// ThreeMixtureDistributionTorch.log_density, multimodal_torch.py:173-242.  cov_invs
// is always the identity (:87-98) so the [B,D]x[D,D] matmul at :234 is skipped.
//   logsumexp_k( -0.5 |s*x - mu_k|^2 + c_k ),  c_k = log_norm_const_k + log w_k (+ log_jacobian)
template <int DP>
struct ThreeMixture {
  static constexpr int kKind = PTRWM_TARGET_THREE_MIXTURE;
  template <bool SCALED>
  __device__ __forceinline__ static float logp_impl(const float (&y)[DP], int D, const TParams &tp) {
    constexpr int W = canon_width(DP);
    float q0p[4] = {0.0f, 0.0f, 0.0f, 0.0f}, q1p[4] = {0.0f, 0.0f, 0.0f, 0.0f}, q2p[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const_float_ptr uv0 = uniform_vec(tp.vec0);
    [[maybe_unused]] const const_float_ptr uv1 = SCALED ? uniform_vec(tp.vec1) : nullptr;
    PTRWM_DIM_LOOP(d, DP, D, {
      // three mean vectors: 3 dim scalar words per evaluation.  Loaded in one go they overflow the SGPR file (81 spilled
      // SGPRs and spill-lane VGPRs in scratch at dim 30); re-materialising the pointer every 8 dimensions keeps <= 24 words
      // in flight (tools/kernel_stats.py)
      if (d > 0 && (d & 7) == 0) uv0 = uniform_vec_again(uv0);
      float e0, e1, e2;
      if constexpr (SCALED) {
        const float sc = uv1[d];  // s x - mu_k as one explicit fma each, the same in every kernel
        e0 = fmaf(y[d], sc, -uv0[d]), e1 = fmaf(y[d], sc, -uv0[D + d]), e2 = fmaf(y[d], sc, -uv0[2 * D + d]);
      } else {
        e0 = sub_rn(y[d], uv0[d]), e1 = sub_rn(y[d], uv0[D + d]), e2 = sub_rn(y[d], uv0[2 * D + d]);
      }
      q0p[d / W] = fmaf(e0, e0, q0p[d / W]);
      q1p[d / W] = fmaf(e1, e1, q1p[d / W]);
      q2p[d / W] = fmaf(e2, e2, q2p[d / W]);
      if ((d & 7) == 7) sched_fence_soft();
    })
    const float q0 = tree4_add(q0p), q1 = tree4_add(q1p), q2 = tree4_add(q2p);
    return finish(q0, q1, q2, tp);
  }
  // logsumexp of the three components from their squared distances (shared with the lane-split kernel)
  __device__ __forceinline__ static float finish(float q0, float q1, float q2, const TParams &tp) {
#pragma clang fp contract(off)
    const float a0 = fmaf(-0.5f, q0, tp.p[0]);
    const float a1 = fmaf(-0.5f, q1, tp.p[1]);
    const float a2 = fmaf(-0.5f, q2, tp.p[2]);
    const float mx = __builtin_fmaxf(__builtin_fmaxf(a0, a1), a2);
    const float sh = __builtin_fmaxf(mx, -3.0e38f);  // all components -inf -> -inf like torch.logsumexp, not NaN
    const float s = (hw_exp(a0 - sh) + hw_exp(a1 - sh)) + hw_exp(a2 - sh);
    return fmaf(hw_log2(s), kLn2, mx);  // ONE explicit fma in every kernel (left to the optimiser it differed)
  }
  template <bool STRICT = false>
  __device__ __forceinline__ static float logp(const float (&y)[DP], int D, const TParams &tp) {
    return tp.vec1 != nullptr ? logp_impl<true>(y, D, tp) : logp_impl<false>(y, D, tp);
  }
};

// The same density when the caller declares (ip[0] = 1, include/ptrwm.h) that the three mean vectors agree in every
// coordinate but the first - the reference class's default centres and its experiments' +-15 centres do.  Then
//   |s x - mu_k|^2 = C + (s_0 x_0 - mu_k0)^2,   C = sum_{d >= 1} (s_d x_d - mu_0d)^2
// and C is evaluated ONCE (canonical four-range order over d >= 1), the first-coordinate term added last: 2 VALU
// instructions and one scalar load per dimension instead of 6 and 3.  Same fp32 tolerance against the oracle (which always
// evaluates the three full sums); not the same bits as ThreeMixture, so the C ABI uses this functor for EVERY evaluation
// of such a target (fused step kernels of both forms, ptrwm_logdensity).
template <int DP>
struct ThreeMixture1 {
  static constexpr int kKind = PTRWM_TARGET_THREE_MIXTURE;
  template <bool SCALED>
  __device__ __forceinline__ static float logp_impl(const float (&y)[DP], int D, const TParams &tp) {
    constexpr int W = canon_width(DP);
    float cp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const const_float_ptr uv0 = uniform_vec(tp.vec0);
    [[maybe_unused]] const const_float_ptr uv1 = SCALED ? uniform_vec(tp.vec1) : nullptr;
    PTRWM_DIM_LOOP(d, DP, D, {
      if (d >= 1) {
        float e;
        if constexpr (SCALED) {
          e = fmaf(y[d], uv1[d], -uv0[d]);
        } else {
          e = sub_rn(y[d], uv0[d]);
        }
        cp[d / W] = fmaf(e, e, cp[d / W]);
      }
      if ((d & 7) == 7) sched_fence_soft();
    })
    const float c = tree4_add(cp);
    float e0, e1, e2;
    if constexpr (SCALED) {
      const float sc = uv1[0];
      e0 = fmaf(y[0], sc, -uv0[0]), e1 = fmaf(y[0], sc, -uv0[D]), e2 = fmaf(y[0], sc, -uv0[2 * D]);
    } else {
      e0 = sub_rn(y[0], uv0[0]), e1 = sub_rn(y[0], uv0[D]), e2 = sub_rn(y[0], uv0[2 * D]);
    }
    return ThreeMixture<DP>::finish(fmaf(e0, e0, c), fmaf(e1, e1, c), fmaf(e2, e2, c), tp);
  }
  template <bool STRICT = false>
  __device__ __forceinline__ static float logp(const float (&y)[DP], int D, const TParams &tp) {
    return tp.vec1 != nullptr ? logp_impl<true>(y, D, tp) : logp_impl<false>(y, D, tp);
  }
};

// FullRosenbrockTorch.log_density, rosenbrock_torch.py:67-84:
//   -sum_{i<D-1} [ b (x_{i+1} - x_i^2)^2 + a (x_i - mu_i)^2 ]
template <int DP>
struct FullRosenbrock {
  static constexpr int kKind = PTRWM_TARGET_FULL_ROSENBROCK;
  template <bool STRICT = false>
  __device__ __forceinline__ static float logp(const float (&y)[DP], int D, const TParams &tp) {
#pragma clang fp contract(off)
    const float a = tp.p[0], b = tp.p[1];
    constexpr int W = canon_width(DP);
    float s1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, s2[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // term i belongs to the range of dim i
    const const_float_ptr uv0 = uniform_vec(tp.vec0);
    PTRWM_DIM_LOOP(i, DP - 1, D - 1, {  // terms i = 0 .. D-2
      const float r = y[i + 1] - y[i] * y[i];
      const float c = y[i] - uv0[i];
      s1[i / W] = fmaf(b * r, r, s1[i / W]);
      s2[i / W] = fmaf(a * c, c, s2[i / W]);
      if ((i & 7) == 7) sched_fence_soft();
    })
    return -(tree4_add(s1) + tree4_add(s2));
  }
};

// EvenRosenbrockTorch.log_density, rosenbrock_torch.py:194-210:
//   -sum_{i<D/2} [ a (x_{2i} - mu_i)^2 + b (x_{2i+1} - x_{2i}^2)^2 ]
template <int DP>
struct EvenRosenbrock {
  static constexpr int kKind = PTRWM_TARGET_EVEN_ROSENBROCK;
  template <bool STRICT = false>
  __device__ __forceinline__ static float logp(const float (&y)[DP], int D, const TParams &tp) {
#pragma clang fp contract(off)
    const float a = tp.p[0], b = tp.p[1];
    constexpr int W = canon_width(DP);  // even: a pair (2i, 2i+1) never straddles two ranges
    float s1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, s2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const const_float_ptr uv0 = uniform_vec(tp.vec0);
    PTRWM_DIM_LOOP(i, DP / 2, D >> 1, {  // pairs i = 0 .. D/2 - 1
      const float c = y[2 * i] - uv0[i];
      const float r = y[2 * i + 1] - y[2 * i] * y[2 * i];
      s1[(2 * i) / W] = fmaf(a * c, c, s1[(2 * i) / W]);
      s2[(2 * i) / W] = fmaf(b * r, r, s2[(2 * i) / W]);
      if ((i & 3) == 3) sched_fence_soft();
    })
    return -(tree4_add(s1) + tree4_add(s2));
  }
};

// HybridRosenbrockTorch.log_density, rosenbrock_torch.py:312-351:
//   -a (x_0 - mu)^2 - b sum_j (x_{j,2} - x_0^2)^2 - b sum_j sum_{i>=3} (x_{j,i} - x_{j,i-1}^2)^2
// with x_{j,i} stored flat after x_0 in blocks of n1-1.  `mask` bit i marks the
// first coordinate of a block (its parent is x_0, otherwise the parent is x_{i-1}).
template <int DP>
struct HybridRosenbrock {
  static constexpr int kKind = PTRWM_TARGET_HYBRID_ROSENBROCK;
  template <bool STRICT = false>
  __device__ __forceinline__ static float logp(const float (&y)[DP], int D, const TParams &tp) {
#pragma clang fp contract(off)
    const float a = tp.p[0], b = tp.p[1], mu = tp.p[2];
    const float c0 = y[0] - mu;
    constexpr int W = canon_width(DP);
    float acc[4] = {a * c0 * c0, 0.0f, 0.0f, 0.0f};  // the x_0 term opens the chain of the first range
    // the block-head bits, re-materialised per evaluation: tested in place (s_bitcmp1_b64 + s_cselect_b64 next to the
    // v_cndmask that uses them).  Left loop-invariant, the compiler hoists one 64-bit lane mask PER DIMENSION out of the
    // step loop - up to 128 SGPRs that live for the whole launch, i.e. in spill lanes (110-163 spilled SGPRs in these
    // kernels before; tools/kernel_stats.py)
    unsigned long long mask[2] = {tp.mask[0], tp.mask[1]};
    PTRWM_VALUE_BARRIER("+s"(mask[0]), "+s"(mask[1]));
    PTRWM_DIM_LOOP(i, DP, D, {
      if (i >= 1) {
        const bool head = (mask[i >> 6] >> (i & 63)) & 1ull;
        // select between the two VALUES: left alone the optimiser selects the INDEX (head ? 0 : i - 1), which makes
        // y[] dynamically indexed and moves the whole vector to scratch memory (16 + 4 DP bytes per thread, found by
        // tools/kernel_stats.py --check); the empty asm pins the second candidate in a register first
        float prev = y[i - 1];
        PTRWM_VALUE_BARRIER("+v"(prev));
        const float parent = head ? y[0] : prev;
        const float r = y[i] - parent * parent;
        acc[i / W] = fmaf(b * r, r, acc[i / W]);
      }
      if ((i & 7) == 7) sched_fence_soft();
    })
    return -tree4_add(acc);
  }
};

// IIDGammaTorch.log_density, iid_product_torch.py:52-91:
//   sum_d [ (k-1) log x_d - x_d / theta ] - dim (lgamma k + k log theta);  -inf if any x_d <= 0
template <int DP>
struct IIDGamma {
  static constexpr int kKind = PTRWM_TARGET_IID_GAMMA;
  template <bool STRICT = false>
  __device__ __forceinline__ static float logp(const float (&y)[DP], int D, const TParams &tp) {
#pragma clang fp contract(off)
    const float km1 = (tp.p[0] - 1.0f) * kLn2;
    const float inv_theta = 1.0f / tp.p[1];
    constexpr int W = canon_width(DP);
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    bool bad = false;
    PTRWM_DIM_LOOP(d, DP, D, {
      const float v = y[d];
      bad = bad || (v <= 0.0f);
      acc[d / W] += fmaf(km1, hw_log2(v), -(v * inv_theta));
      if ((d & 7) == 7) sched_fence_soft();
    })
    return bad ? kNegInf : tree4_add(acc) - tp.p[2];
  }
};

// IIDBetaTorch.log_density, iid_product_torch.py:188-229:
//   sum_d [ (alpha-1) log x_d + (beta-1) log(1-x_d) ] + dim log(1/B(alpha,beta));  -inf outside (0,1)
template <int DP>
struct IIDBeta {
  static constexpr int kKind = PTRWM_TARGET_IID_BETA;
  template <bool STRICT = false>
  __device__ __forceinline__ static float logp(const float (&y)[DP], int D, const TParams &tp) {
#pragma clang fp contract(off)
    const float am1 = (tp.p[0] - 1.0f) * kLn2;
    const float bm1 = (tp.p[1] - 1.0f) * kLn2;
    constexpr int W = canon_width(DP);
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    bool bad = false;
    PTRWM_DIM_LOOP(d, DP, D, {
      const float v = y[d];
      bad = bad || (v <= 0.0f) || (v >= 1.0f);
      acc[d / W] += fmaf(am1, hw_log2(v), bm1 * hw_log2(1.0f - v));
      if ((d & 7) == 7) sched_fence_soft();
    })
    return bad ? kNegInf : tree4_add(acc) + tp.p[2];
  }
};

// MultivariateNormalTorch.log_density with a diagonal covariance (multivariate_normal_torch.py:62-92; the dense
// [B,D]x[D,D] product is out of scope) and ScaledMultivariateNormalTorch.log_density (:199-224).
template <int DP>
struct DiagGaussian {
  static constexpr int kKind = PTRWM_TARGET_DIAG_GAUSSIAN;
  template <bool SCALED_FORM>
  __device__ __forceinline__ static float quad(const float (&y)[DP], int D, const TParams &tp) {
#pragma clang fp contract(off)
    constexpr int W = canon_width(DP);
    float q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const const_float_ptr uv0 = uniform_vec(tp.vec0);
    [[maybe_unused]] const const_float_ptr uv1 = SCALED_FORM ? nullptr : uniform_vec(tp.vec1);
    PTRWM_DIM_LOOP(d, DP, D, {
      if constexpr (SCALED_FORM) {
        const float sx = uv0[d] * y[d];
        q[d / W] = fmaf(sx, sx, q[d / W]);
      } else {
        const float c = y[d] - uv0[d];
        q[d / W] = fmaf(c * uv1[d], c, q[d / W]);
      }
      if ((d & 7) == 7) sched_fence_soft();
    })
    return tree4_add(q);
  }
  template <bool STRICT = false>
  __device__ __forceinline__ static float logp(const float (&y)[DP], int D, const TParams &tp) {
#pragma clang fp contract(off)
    const float q = tp.ip[0] != 0 ? quad<true>(y, D, tp) : quad<false>(y, D, tp);
    return fmaf(-0.5f, q, tp.p[0]);
  }
};

// HypercubeTorch.log_density, hypercube_torch.py:49-78: log(1/volume) inside [left, right]^D (closed), -inf outside.
template <int DP>
struct Hypercube {
  static constexpr int kKind = PTRWM_TARGET_HYPERCUBE;
  template <bool STRICT = false>
  __device__ __forceinline__ static float logp(const float (&y)[DP], int D, const TParams &tp) {
#pragma clang fp contract(off)
    const float lo = tp.p[0], hi = tp.p[1];
    bool inside = true;
    PTRWM_DIM_LOOP(d, DP, D, { inside = inside && (y[d] >= lo) && (y[d] <= hi); })
    return inside ? tp.p[2] : kNegInf;
  }
};

// NealFunnelTorch.log_density, funnel_torch.py:39-76:
//   -0.5 log 2pi - 0.5 log s2 - 0.5 (v - mu_v)^2 / s2 - 0.5 (D-1) log 2pi - 0.5 (D-1) v - 0.5 exp(-v) sum_k (z_k - mu_z)^2
template <int DP>
struct NealFunnel {
  static constexpr int kKind = PTRWM_TARGET_NEAL_FUNNEL;
  template <bool STRICT = false>
  __device__ __forceinline__ static float logp(const float (&y)[DP], int D, const TParams &tp) {
#pragma clang fp contract(off)
    const float mu_v = tp.p[0], s2 = tp.p[1], mu_z = tp.p[2];
    const float log_2pi = 1.8378770664093453f;
    const float v = y[0];
    const float dv = v - mu_v;
    const float prior = -0.5f * log_2pi - 0.5f * (hw_log2(s2) * kLn2) - 0.5f * (dv * dv) / s2;
    constexpr int W = canon_width(DP);
    float ssp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    PTRWM_DIM_LOOP(d, DP, D, {
      if (d >= 1) {
        const float c = y[d] - mu_z;
        ssp[d / W] = fmaf(c, c, ssp[d / W]);
      }
      if ((d & 7) == 7) sched_fence_soft();
    })
    const float ss = tree4_add(ssp);
    const float dm1 = (float)(D - 1);
    const float lik = -0.5f * dm1 * log_2pi - 0.5f * dm1 * v - 0.5f * hw_exp(-v) * ss;
    return D > 1 ? prior + lik : prior;
  }
};

}  // namespace ptrwm
