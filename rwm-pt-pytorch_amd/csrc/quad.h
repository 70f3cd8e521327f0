// The lane-split ("quad") form of the fused PT-RWM kernel: FOUR lanes per (chain, temperature) replica.
//
// Why: the one-thread-per-replica kernel (kernel.h) needs >= 2 wavefronts per SIMD to keep the VALU issuing
// (profiles/r02_pmc_cfg2.csv: BASELINE configs[1], 65 536 chains x 1 temperature = ONE wave per SIMD, issues at 0.41 of
// the peak against 0.59 at four waves), and above dim 64 its register arrays (x[], y[]: 2 x dim VGPRs) leave room for
// one wave per SIMD only.  Here lane q of a quad owns the dimensions [q W, (q+1) W) of its replica (W = canon_width:
// 8 / 16 / 20 / 24 / 28 for dim <= 32 / 64 / 80 / 96 / 112): a quarter of the Philox blocks, of the proposal transforms and of the
// log-density terms, 2 W registers for the state, four times the waves for the same batch.
//
// Bit-identical to kernel.h by construction: same Philox words (word w of a step is word w whoever computes it), same
// per-dimension arithmetic, and every sum over dimensions in the canonical four-range order of philox.h - each lane
// runs the chain of its own range, two DPP quad permutes combine them as (P0 + P1) + (P2 + P3).  All four lanes then
// hold the same log-density, make the same Metropolis and swap decisions and update their own quarter of the state.
// The C ABI picks the form from the batch size and dim (capi.hip); tools/check_all_variants.py checks every variant of
// this kernel against kernel.h on the same Philox stream, bit for bit.
//
// Thread map.  slot = tid / 4 (replica within the exchange group), q = tid % 4.  4 T <= 64 ("narrow", T <= 16): an
// exchange group is one wavefront holding 16 / T whole ladders; a workgroup is four independent wavefronts.
// T > 16 ("wide"): the exchange group is the workgroup - as many whole ladders as fill 256 threads best, or one ladder in
// 4 T threads rounded up to whole waves (<= 512: ladders of up to 128 temperatures); barriers for swaps.
#pragma once
#include "kernel.h"

namespace ptrwm {

constexpr int kQuad = 4;

// A lane runs two (dim 30) to seven Philox blocks per step: few enough to let the scheduler interleave them (at four
// waves per SIMD the dependent rounds of ONE block do not hide their own latency); PTRWM_QUAD_FENCE_RNG restores the
// per-block fences of the one-thread-per-replica kernel (needed there to bound the registers of eight-plus blocks).
__device__ __forceinline__ void quad_rng_fence() {
#ifdef PTRWM_QUAD_FENCE_RNG
  __builtin_amdgcn_sched_barrier(0);
#endif
}

// ---- DPP quad permutes (lane i of a quad reads lane SEL_i; full rate, no LDS) --------------------------------------
constexpr int quad_ctrl(int s0, int s1, int s2, int s3) { return s0 | (s1 << 2) | (s2 << 4) | (s3 << 6); }
constexpr int kDppSwapPair = quad_ctrl(1, 0, 3, 2);   // partner inside the pair
constexpr int kDppSwapHalf = quad_ctrl(2, 3, 0, 1);   // the other pair
constexpr int kDppNext = quad_ctrl(1, 2, 3, 3);       // lane q reads lane q + 1 (lane 3: itself)
constexpr int kDppPrev = quad_ctrl(0, 0, 1, 2);       // lane q reads lane q - 1 (lane 0: itself)

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
template <int LANE>
__device__ __forceinline__ float quad_bcast(float v) {
  return dpp_f<quad_ctrl(LANE, LANE, LANE, LANE)>(v);
}
// (P0 + P1) + (P2 + P3) on every lane: the canonical combination of philox.h (fp add is commutative bit for bit)
__device__ __forceinline__ float quad_tree_add(float p) {
  const float t = add_rn(p, dpp_f<kDppSwapPair>(p));
  return add_rn(t, dpp_f<kDppSwapHalf>(t));
}
__device__ __forceinline__ bool quad_any(bool b) {
  int v = b ? 1 : 0;
  v |= dpp_i<kDppSwapPair>(v);
  v |= dpp_i<kDppSwapHalf>(v);
  return v != 0;
}

// What a lane knows about its share of the replica
struct QLane {
  int q;      // lane within the quad
  int n_own;  // owned dimensions: clamp(D - q W, 0, W)
  int d0;     // first owned dimension, q W
};

// A lane-dependent value behind an empty asm: compares against it are redone where they are used (one v_cmp each)
// instead of being hoisted out of the step loop as 64-bit lane masks - in the kernels with a run-time dim that was
// dozens of SGPR pairs, spilled to VGPR lanes and fetched back with v_readlane (157 SGPR spills at width 28; dims
// 65..104 other than 100 gained 3-9 %, dims 16..64 in this form 2-3 %).
__device__ __forceinline__ int q_fresh(int v) {
  PTRWM_VALUE_BARRIER("+v"(v));
  return v;
}
// valid(j): does local slot j hold a dimension?  MIN_OWN (a compile-time lower bound of n_own over the four lanes, known
// when dim is compiled in) lets the compiler drop the test for the slots every lane owns; -1 = dim at run time.
// (Tried and measured slower, profiles/r02_bench_variants.txt: one divergent region per block of four slots instead of
// per slot; computing every slot with selects on the accumulators; building each slot's lane mask on the scalar unit
// from two wave masks and inverse_ballot - 14 % fewer VALU instructions, but 2-6 % slower.)
template <int MIN_OWN>
__device__ __forceinline__ bool q_valid(const QLane &l, int j) {
  if constexpr (MIN_OWN >= 0) return j < MIN_OWN || j < l.n_own;  // dim compiled in: a handful of distinct masks
  return j < q_fresh(l.n_own);
}

// ---- proposals ---------------------------------------------------------------------------------------------------------
// Same raw-word map as proposals.h: local Philox block b of lane q is global block q W/4 + b, i.e. dims 4c .. 4c+3.
// Every lane of the wave executes every block anyway, so a block is computed when ANY lane needs it (a wave-uniform
// test); the accept (and radius) words are picked up by the lane whose block holds them and shared through the quad.
template <int W, int MIN_OWN>
struct QNormal {
  static constexpr int kKind = PTRWM_PROPOSAL_NORMAL;
  static constexpr bool kKnowsJump = true;
  __device__ __forceinline__ static float increment_scale(float tscale, const PParams &) { return tscale; }
  // jump / jump_kind: proposals.h (this lane's range of the squared jump: kJumpPartial sums are combined by the caller)
  __device__ __forceinline__ static float propose(float (&y)[W], const float (&x)[W], const QLane &l, int D, float tscale,
                                                  const PParams &, const RngCtx &rc, const float *ext_rep, float ext_u,
                                                  float &jump, int &jump_kind) {
    if (ext_rep != nullptr) {
      jump_kind = kJumpNone;
      const float *er = ext_rep + l.d0;
#pragma unroll
      for (int j = 0; j < W; ++j)
        if (q_valid<MIN_OWN>(l, j)) y[j] = add_rn(x[j], mul_rn(er[j], tscale));
      return ext_u;
    }
    jump_kind = kJumpPartial;
    const float c_t = mul_rn(mul_rn(-2.0f * kLn2, tscale), tscale);
    const int w_a = 2 * ((D + 1) >> 1);  // accept word
    const int c_a = w_a >> 2;            // its block (uniform)
    const uint32_t cb0 = (uint32_t)(l.q * (W / 4));
    float u_loc = 0.0f;
#pragma unroll
    for (int b = 0; b < W / 4; ++b) {
      if (4 * b < D || (c_a < W && (c_a % (W / 4)) == b)) {  // wave-uniform: lane 0 has dims there, or the accept word is
        const uint32_t cb = cb0 + (uint32_t)b;
        const u32x4 r = philox4x32_10(rc.c0hi | cb, rc.c1, rc.c2, rc.c3, rc.k0, rc.k1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = 4 * b + 2 * h;
          const uint32_t ra = h ? r.z : r.x, rb = h ? r.w : r.y;
          if (q_valid<MIN_OWN>(l, j)) {
            const float arg = mul_rn(c_t, hw_log2(u01_open0(ra)));
            const float rad = hw_sqrt(arg);
            const float ang = bm_turns(rb);
            const float sn = __builtin_amdgcn_sinf(ang);
            y[j] = fmaf(rad, sn, x[j]);
            if (q_valid<MIN_OWN>(l, j + 1)) {
              y[j + 1] = fmaf(rad, __builtin_amdgcn_cosf(ang), x[j + 1]);
              jump = add_rn(jump, arg);
            } else {
              const float i0 = mul_rn(rad, sn);  // the last dimension of an odd dim: half a pair
              jump = fmaf(i0, i0, jump);
            }
          }
          if ((int)(4 * cb) + 2 * h == w_a) u_loc = u01(ra);
        }
        quad_rng_fence();
      }
    }
    if (c_a >= W) {  // dim = 4 W or 4 W - 1: the accept word opens block W, which no lane owns: all compute it
      const u32x4 r = philox4x32_10(rc.c0hi | (uint32_t)W, rc.c1, rc.c2, rc.c3, rc.k0, rc.k1);
      return u01(r.x);
    }
    return quad_tree_add(u_loc);  // exact: three of the four terms are zero
  }
};

template <int W, int MIN_OWN>
struct QLaplace {
  static constexpr int kKind = PTRWM_PROPOSAL_LAPLACE;
  static constexpr bool kKnowsJump = false;
  __device__ __forceinline__ static float increment_scale(float tscale, const PParams &) { return tscale; }
  __device__ __forceinline__ static float propose(float (&y)[W], const float (&x)[W], const QLane &l, int D, float tscale,
                                                  const PParams &pp, const RngCtx &rc, const float *ext_rep, float ext_u,
                                                  float &, int &jump_kind) {
    jump_kind = kJumpNone;
    const float *dsc = pp.dim_scale + l.d0;  // this lane's per-dimension scales (L1-resident; lane-dependent address)
    if (ext_rep != nullptr) {
      const float *er = ext_rep + l.d0;
#pragma unroll
      for (int j = 0; j < W; ++j)
        if (q_valid<MIN_OWN>(l, j)) y[j] = add_rn(x[j], LaplaceProposal<W>::transform(er[j], mul_rn(dsc[j], tscale)));
      return ext_u;
    }
    const int c_a = D >> 2;  // block of the accept word (word D)
    const uint32_t cb0 = (uint32_t)(l.q * (W / 4));
    float u_loc = 0.0f;
#pragma unroll
    for (int b = 0; b < W / 4; ++b) {
      if (4 * b < D || (c_a < W && (c_a % (W / 4)) == b)) {
        const uint32_t cb = cb0 + (uint32_t)b;
        const u32x4 r = philox4x32_10(rc.c0hi | cb, rc.c1, rc.c2, rc.c3, rc.k0, rc.k1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int j = 4 * b + k;
          if (q_valid<MIN_OWN>(l, j)) y[j] = add_rn(x[j], LaplaceProposal<W>::transform(u01(pick(r, k)), mul_rn(dsc[j], tscale)));
          // the accept word (word D) sits at position D & 3 of block D >> 2: the position is wave-uniform (a constant
          // when dim is compiled in), only the block is lane-dependent - one compare per block, not one per word
          if (k == (D & 3)) u_loc = ((int)cb == c_a) ? u01(pick(r, k)) : u_loc;
        }
        quad_rng_fence();
      }
    }
    if (c_a >= W) {
      const u32x4 r = philox4x32_10(rc.c0hi | (uint32_t)W, rc.c1, rc.c2, rc.c3, rc.k0, rc.k1);
      return u01(r.x);
    }
    return quad_tree_add(u_loc);
  }
};

template <int W, int MIN_OWN>
struct QUniformRadius {
  static constexpr int kKind = PTRWM_PROPOSAL_UNIFORM_RADIUS;
  static constexpr bool kKnowsJump = true;
  __device__ __forceinline__ static float increment_scale(float tscale, const PParams &pp) { return tscale * hw_sqrt(pp.inv_dim); }
  __device__ __forceinline__ static float propose(float (&y)[W], const float (&x)[W], const QLane &l, int D, float tscale,
                                                  const PParams &pp, const RngCtx &rc, const float *ext_rep, float ext_u,
                                                  float &jump, int &jump_kind) {
    float u_acc = ext_u, u_rad;
    if (ext_rep != nullptr) {
      const float *er = ext_rep + l.d0;
#pragma unroll
      for (int j = 0; j < W; ++j)
        if (q_valid<MIN_OWN>(l, j)) y[j] = er[j];
      u_rad = ext_rep[D];
    } else {
      const int w_a = 2 * ((D + 1) >> 1);  // radius word; the accept word is the next one (same block: w_a is even)
      const int c_a = w_a >> 2;
      const uint32_t cb0 = (uint32_t)(l.q * (W / 4));
      float ur_loc = 0.0f, ua_loc = 0.0f;
#pragma unroll
      for (int b = 0; b < W / 4; ++b) {
        if (4 * b < D || (c_a < W && (c_a % (W / 4)) == b)) {
          const uint32_t cb = cb0 + (uint32_t)b;
          const u32x4 r = philox4x32_10(rc.c0hi | cb, rc.c1, rc.c2, rc.c3, rc.k0, rc.k1);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int j = 4 * b + 2 * h;
            const uint32_t ra = h ? r.z : r.x, rb = h ? r.w : r.y;
            if (q_valid<MIN_OWN>(l, j)) {
              float z0, z1;
              box_muller(ra, rb, z0, z1);
              y[j] = z0;
              if (q_valid<MIN_OWN>(l, j + 1)) y[j + 1] = z1;
            }
            if ((int)(4 * cb) + 2 * h == w_a) {
              ur_loc = u01(ra);
              ua_loc = u01(rb);
            }
          }
          quad_rng_fence();
        }
      }
      if (c_a >= W) {
        const u32x4 r = philox4x32_10(rc.c0hi | (uint32_t)W, rc.c1, rc.c2, rc.c3, rc.k0, rc.k1);
        u_rad = u01(r.x);
        u_acc = u01(r.y);
      } else {
        u_rad = quad_tree_add(ur_loc);
        u_acc = quad_tree_add(ua_loc);
      }
    }
    float n2 = 0.0f;  // this lane's range of |g|^2, then the canonical combination
#pragma unroll
    for (int j = 0; j < W; ++j)
      if (q_valid<MIN_OWN>(l, j)) n2 = fmaf(y[j], y[j], n2);
    const float nrm_sq = quad_tree_add(n2);
    const float nrm = hw_sqrt(nrm_sq);
    const float safe = nrm > 1e-12f ? nrm : 1.0f;
    const float rad = tscale * hw_exp2(pp.inv_dim * hw_log2(u_rad));
    const float inv = div_rn(1.0f, safe);
    if (ext_rep != nullptr) {
      jump_kind = kJumpNone;
#pragma unroll
      for (int j = 0; j < W; ++j)
        if (q_valid<MIN_OWN>(l, j)) y[j] = add_rn(x[j], mul_rn(mul_rn(y[j], inv), rad));
    } else {  // Philox path: one fma per dimension (see proposals.h)
      const float k = mul_rn(inv, rad);
#pragma unroll
      for (int j = 0; j < W; ++j)
        if (q_valid<MIN_OWN>(l, j)) y[j] = fmaf(y[j], k, x[j]);
      jump_kind = kJumpTotal;  // |g k|^2 = |g|^2 k^2, the same bits on all four lanes
      jump = mul_rn(mul_rn(nrm_sq, k), k);
    }
    return u_acc;
  }
};

// ---- targets ---------------------------------------------------------------------------------------------------------
// logp(y_local, lane, D, params) -> the replica's log-density, the same bits on all four lanes.  Each functor restates
// the per-dimension arithmetic of its twin in targets.h on the lane's own range and combines canonically.
template <int W, int MIN_OWN, bool TWO_TERM>
struct QRoughCarpetT {
  static constexpr int kKind = PTRWM_TARGET_ROUGH_CARPET;
  template <bool SCALED>
  __device__ __forceinline__ static float impl(const float (&y)[W], const QLane &l, const TParams &tp) {
    const float m0 = -(tp.p[0] * kRcScale), m1 = -(tp.p[1] * kRcScale), m2 = -(tp.p[2] * kRcScale);  // (targets.h)
    [[maybe_unused]] const float *sc_v = SCALED ? tp.vec0 + l.d0 : nullptr;
    const float w0 = tp.p[3] * kLog2e, w1 = tp.p[4] * kLog2e, w2 = tp.p[5] * kLog2e;
    float sm = 0.0f, pr = 1.0f;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      if (q_valid<MIN_OWN>(l, j)) {
        const float sc = SCALED ? sc_v[j] * kRcScale : kRcScale;
        const float d0 = fmaf(y[j], sc, m0), d1 = fmaf(y[j], sc, m1), d2 = fmaf(y[j], sc, m2);
        float mx, s;
        rc_dim_term<false, TWO_TERM>(d0, d1, d2, w0, w1, w2, mx, s);
        sm = add_rn(sm, mx);
        pr = mul_rn(pr, s);
      }
      if ((j & 3) == 3) sched_fence_soft();
    }
    const float sum_mx = quad_tree_add(sm);
    const float lgp = hw_log2(mul_rn(pr, dpp_f<kDppSwapPair>(pr)));  // log2 of the pair's product (see targets.h)
    const float lg = add_rn(lgp, dpp_f<kDppSwapHalf>(lgp));
    return add_rn(fmaf(add_rn(sum_mx, lg), kLn2, tp.p[7]), tp.p[6]);
  }
  __device__ __forceinline__ static float logp(const float (&y)[W], const QLane &l, int, const TParams &tp) {
    return tp.vec0 != nullptr ? impl<true>(y, l, tp) : impl<false>(y, l, tp);
  }
};
template <int W, int MIN_OWN>
using QRoughCarpet = QRoughCarpetT<W, MIN_OWN, false>;
template <int W, int MIN_OWN>
using QRoughCarpet2 = QRoughCarpetT<W, MIN_OWN, true>;

template <int W, int MIN_OWN>
struct QThreeMixture {
  static constexpr int kKind = PTRWM_TARGET_THREE_MIXTURE;
  template <bool SCALED>
  __device__ __forceinline__ static float impl(const float (&y)[W], const QLane &l, int D, const TParams &tp) {
    const float *mu = tp.vec0 + l.d0;
    [[maybe_unused]] const float *sc_v = SCALED ? tp.vec1 + l.d0 : nullptr;
    float q0 = 0.0f, q1 = 0.0f, q2 = 0.0f;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      if (q_valid<MIN_OWN>(l, j)) {
        float e0, e1, e2;
        if constexpr (SCALED) {
          const float sc = sc_v[j];
          e0 = fmaf(y[j], sc, -mu[j]), e1 = fmaf(y[j], sc, -mu[D + j]), e2 = fmaf(y[j], sc, -mu[2 * D + j]);
        } else {
          e0 = sub_rn(y[j], mu[j]), e1 = sub_rn(y[j], mu[D + j]), e2 = sub_rn(y[j], mu[2 * D + j]);
        }
        q0 = fmaf(e0, e0, q0);
        q1 = fmaf(e1, e1, q1);
        q2 = fmaf(e2, e2, q2);
      }
      if ((j & 7) == 7) sched_fence_soft();
    }
    return ThreeMixture<W>::finish(quad_tree_add(q0), quad_tree_add(q1), quad_tree_add(q2), tp);
  }
  __device__ __forceinline__ static float logp(const float (&y)[W], const QLane &l, int D, const TParams &tp) {
    return tp.vec1 != nullptr ? impl<true>(y, l, D, tp) : impl<false>(y, l, D, tp);
  }
};

// targets.h ThreeMixture1 (means equal in every coordinate but the first): the shared part over d >= 1 on the lane's own
// range, combined canonically; the three first-coordinate terms from lane 0's y[0], broadcast through the quad.
template <int W, int MIN_OWN>
struct QThreeMixture1 {
  static constexpr int kKind = PTRWM_TARGET_THREE_MIXTURE;
  template <bool SCALED>
  __device__ __forceinline__ static float impl(const float (&y)[W], const QLane &l, int D, const TParams &tp) {
    const float *mu = tp.vec0 + l.d0;
    [[maybe_unused]] const float *sc_v = SCALED ? tp.vec1 + l.d0 : nullptr;
    float cl = 0.0f;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      if ((j >= 1 || l.q != 0) && q_valid<MIN_OWN>(l, j)) {  // every dimension but the first (dimension 0 = slot 0 of lane 0)
        float e;
        if constexpr (SCALED) {
          e = fmaf(y[j], sc_v[j], -mu[j]);
        } else {
          e = sub_rn(y[j], mu[j]);
        }
        cl = fmaf(e, e, cl);
      }
      if ((j & 7) == 7) sched_fence_soft();
    }
    const float c = quad_tree_add(cl);
    const float y0 = quad_bcast<0>(y[0]);
    const float *m0 = tp.vec0;  // wave-uniform addresses
    float e0, e1, e2;
    if constexpr (SCALED) {
      const float sc = tp.vec1[0];
      e0 = fmaf(y0, sc, -m0[0]), e1 = fmaf(y0, sc, -m0[D]), e2 = fmaf(y0, sc, -m0[2 * D]);
    } else {
      e0 = sub_rn(y0, m0[0]), e1 = sub_rn(y0, m0[D]), e2 = sub_rn(y0, m0[2 * D]);
    }
    return ThreeMixture<W>::finish(fmaf(e0, e0, c), fmaf(e1, e1, c), fmaf(e2, e2, c), tp);
  }
  __device__ __forceinline__ static float logp(const float (&y)[W], const QLane &l, int D, const TParams &tp) {
    return tp.vec1 != nullptr ? impl<true>(y, l, D, tp) : impl<false>(y, l, D, tp);
  }
};

template <int W, int MIN_OWN>
struct QFullRosenbrock {
  static constexpr int kKind = PTRWM_TARGET_FULL_ROSENBROCK;
  __device__ __forceinline__ static float logp(const float (&y)[W], const QLane &l, int D, const TParams &tp) {
#pragma clang fp contract(off)
    const float a = tp.p[0], b = tp.p[1];
    const float *mu = tp.vec0 + l.d0;
    const float halo = dpp_f<kDppNext>(y[0]);  // x_{i+1} of this lane's last term lives in the next lane
    float s1 = 0.0f, s2 = 0.0f;
    const int d0f = q_fresh(l.d0);  // (q_fresh: the per-slot lane masks are rebuilt here, not hoisted out of the step loop)
#pragma unroll
    for (int j = 0; j < W; ++j) {
      if (d0f + j + 1 < D) {  // term i = d0 + j exists (then slot j is owned)
        const float nxt = (j + 1 < W) ? y[j + 1 < W ? j + 1 : 0] : halo;
        const float r = nxt - y[j] * y[j];
        const float c = y[j] - mu[j];
        s1 = fmaf(b * r, r, s1);
        s2 = fmaf(a * c, c, s2);
      }
      if ((j & 7) == 7) sched_fence_soft();
    }
    return -(quad_tree_add(s1) + quad_tree_add(s2));
  }
};

template <int W, int MIN_OWN>
struct QEvenRosenbrock {
  static constexpr int kKind = PTRWM_TARGET_EVEN_ROSENBROCK;
  __device__ __forceinline__ static float logp(const float (&y)[W], const QLane &l, int D, const TParams &tp) {
#pragma clang fp contract(off)
    const float a = tp.p[0], b = tp.p[1];
    const float *mu = tp.vec0 + (l.d0 >> 1);  // one mu per pair; W is a multiple of 4, so pairs never straddle lanes
    float s1 = 0.0f, s2 = 0.0f;
    const int d0f = q_fresh(l.d0);
#pragma unroll
    for (int i = 0; 2 * i + 1 < W; ++i) {
      if (d0f + 2 * i + 1 < D) {
        const float c = y[2 * i] - mu[i];
        const float r = y[2 * i + 1] - y[2 * i] * y[2 * i];
        s1 = fmaf(a * c, c, s1);
        s2 = fmaf(b * r, r, s2);
      }
      if ((i & 3) == 3) sched_fence_soft();
    }
    return -(quad_tree_add(s1) + quad_tree_add(s2));
  }
};

template <int W, int MIN_OWN>
struct QHybridRosenbrock {
  static constexpr int kKind = PTRWM_TARGET_HYBRID_ROSENBROCK;
  __device__ __forceinline__ static float logp(const float (&y)[W], const QLane &l, int D, const TParams &tp) {
#pragma clang fp contract(off)
    const float a = tp.p[0], b = tp.p[1], mu = tp.p[2];
    const float y0 = quad_bcast<0>(y[0]);          // x_0, the parent of every block head
    const float halo = dpp_f<kDppPrev>(y[W - 1]);  // x_{i-1} of this lane's first coordinate lives in the previous lane
    const float c0 = y0 - mu;
    float acc = (l.q == 0) ? a * c0 * c0 : 0.0f;  // the x_0 term opens the chain of the first range
    const int d0f = q_fresh(l.d0);
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const int i = d0f + j;
      if (i >= 1 && i < D) {
        const bool head = (tp.mask[i >> 6] >> (i & 63)) & 1ull;
        const float prev = (j > 0) ? y[j > 0 ? j - 1 : 0] : halo;
        const float parent = head ? y0 : prev;
        const float r = y[j] - parent * parent;
        acc = fmaf(b * r, r, acc);
      }
      if ((j & 7) == 7) sched_fence_soft();
    }
    return -quad_tree_add(acc);
  }
};

template <int W, int MIN_OWN>
struct QIIDGamma {
  static constexpr int kKind = PTRWM_TARGET_IID_GAMMA;
  __device__ __forceinline__ static float logp(const float (&y)[W], const QLane &l, int, const TParams &tp) {
#pragma clang fp contract(off)
    const float km1 = (tp.p[0] - 1.0f) * kLn2;
    const float inv_theta = 1.0f / tp.p[1];
    float acc = 0.0f;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      if (q_valid<MIN_OWN>(l, j)) {
        const float v = y[j];
        bad = bad || (v <= 0.0f);
        acc += fmaf(km1, hw_log2(v), -(v * inv_theta));
      }
      if ((j & 7) == 7) sched_fence_soft();
    }
    const float tot = quad_tree_add(acc);
    return quad_any(bad) ? kNegInf : tot - tp.p[2];
  }
};

template <int W, int MIN_OWN>
struct QIIDBeta {
  static constexpr int kKind = PTRWM_TARGET_IID_BETA;
  __device__ __forceinline__ static float logp(const float (&y)[W], const QLane &l, int, const TParams &tp) {
#pragma clang fp contract(off)
    const float am1 = (tp.p[0] - 1.0f) * kLn2;
    const float bm1 = (tp.p[1] - 1.0f) * kLn2;
    float acc = 0.0f;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      if (q_valid<MIN_OWN>(l, j)) {
        const float v = y[j];
        bad = bad || (v <= 0.0f) || (v >= 1.0f);
        acc += fmaf(am1, hw_log2(v), bm1 * hw_log2(1.0f - v));
      }
      if ((j & 7) == 7) sched_fence_soft();
    }
    const float tot = quad_tree_add(acc);
    return quad_any(bad) ? kNegInf : tot + tp.p[2];
  }
};

template <int W, int MIN_OWN>
struct QDiagGaussian {
  static constexpr int kKind = PTRWM_TARGET_DIAG_GAUSSIAN;
  template <bool SCALED_FORM>
  __device__ __forceinline__ static float quad(const float (&y)[W], const QLane &l, const TParams &tp) {
#pragma clang fp contract(off)
    const float *v0 = tp.vec0 + l.d0;
    [[maybe_unused]] const float *v1 = SCALED_FORM ? nullptr : tp.vec1 + l.d0;
    float qd = 0.0f;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      if (q_valid<MIN_OWN>(l, j)) {
        if constexpr (SCALED_FORM) {
          const float sx = v0[j] * y[j];
          qd = fmaf(sx, sx, qd);
        } else {
          const float c = y[j] - v0[j];
          qd = fmaf(c * v1[j], c, qd);
        }
      }
      if ((j & 7) == 7) sched_fence_soft();
    }
    return quad_tree_add(qd);
  }
  __device__ __forceinline__ static float logp(const float (&y)[W], const QLane &l, int, const TParams &tp) {
#pragma clang fp contract(off)
    const float qd = tp.ip[0] != 0 ? quad<true>(y, l, tp) : quad<false>(y, l, tp);
    return fmaf(-0.5f, qd, tp.p[0]);
  }
};

template <int W, int MIN_OWN>
struct QHypercube {
  static constexpr int kKind = PTRWM_TARGET_HYPERCUBE;
  __device__ __forceinline__ static float logp(const float (&y)[W], const QLane &l, int, const TParams &tp) {
#pragma clang fp contract(off)
    const float lo = tp.p[0], hi = tp.p[1];
    bool outside = false;
#pragma unroll
    for (int j = 0; j < W; ++j)
      if (q_valid<MIN_OWN>(l, j)) outside = outside || !((y[j] >= lo) && (y[j] <= hi));
    return quad_any(outside) ? kNegInf : tp.p[2];
  }
};

template <int W, int MIN_OWN>
struct QNealFunnel {
  static constexpr int kKind = PTRWM_TARGET_NEAL_FUNNEL;
  __device__ __forceinline__ static float logp(const float (&y)[W], const QLane &l, int D, const TParams &tp) {
#pragma clang fp contract(off)
    const float mu_v = tp.p[0], s2 = tp.p[1], mu_z = tp.p[2];
    const float log_2pi = 1.8378770664093453f;
    const float v = quad_bcast<0>(y[0]);
    const float dv = v - mu_v;
    const float prior = -0.5f * log_2pi - 0.5f * (hw_log2(s2) * kLn2) - 0.5f * (dv * dv) / s2;
    float ssl = 0.0f;
    const int d0f = q_fresh(l.d0);
#pragma unroll
    for (int j = 0; j < W; ++j) {
      if (d0f + j >= 1 && q_valid<MIN_OWN>(l, j)) {
        const float c = y[j] - mu_z;
        ssl = fmaf(c, c, ssl);
      }
      if ((j & 7) == 7) sched_fence_soft();
    }
    const float ss = quad_tree_add(ssl);
    const float dm1 = (float)(D - 1);
    const float lik = -0.5f * dm1 * log_2pi - 0.5f * dm1 * v - 0.5f * hw_exp(-v) * ss;
    return D > 1 ? prior + lik : prior;
  }
};

// ---- the kernel ------------------------------------------------------------------------------------------------------
// dynamic LDS bytes of a workgroup: per replica slot a row of up to 4 W state elements (float, or double in the F64 form)
// plus the words of the swap machinery (per slot: log-density, swap uniform, outcome, the ladder's swap-form objection)
// -> (W * words-per-element + 2) floats per thread
constexpr unsigned quad_kernel_lds_bytes(int threads, int w, bool f64 = false) {
  return (unsigned)(threads * (w * (f64 ? 2 : 1) + 2)) * 4u;
}
// Widest workgroup = one ladder.  Every variant is compiled for workgroups of up to 512 threads (ladders of <= 128
// temperatures: up to 256 VGPRs).  (Until round 4 the W >= 20 classes also existed for 1024 threads: 128 VGPRs, 40 of
// them spilled inside the step loop - retired, variants.h.)
constexpr int kQuadThreads = 512;

// double-precision pieces of the F64 form: IEEE ops the compiler must not contract (the state update x + scale * z is then
// bit-identical to the reference's two float64 torch ops), and the canonical quad combination on 64-bit values
__device__ __forceinline__ double dmul_rn(double a, double b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ double dadd_rn(double a, double b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ double dsub_rn(double a, double b) {
#pragma clang fp contract(off)
  return a - b;
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(dpp_i<CTRL>(hi), dpp_i<CTRL>(lo));
}
__device__ __forceinline__ double quad_tree_add(double p) {
  const double t = dadd_rn(p, dpp_d<kDppSwapPair>(p));
  return dadd_rn(t, dpp_d<kDppSwapHalf>(t));
}
template <bool F64>
struct quad_state {
  typedef float type;
};
template <>
struct quad_state<true> {
  typedef double type;
};

// W      lane register width = canonical range width (8 / 16 / 20 / 24 / 28)
// DEXACT dim compiled in (0: run-time dim, any value the width class covers)
// MAXT   largest workgroup the variant may be launched with (kQuadThreads)
// F64    state_f64 of include/ptrwm.h - the reference's dtype=torch.float64 (pt_rwm_gpu_optimized.py:134,431-449): the state,
//        the proposal x + scale * z and the squared jump are carried in double (a.state / trace / ext_prop are double arrays);
//        the increment itself comes from the float proposal functor (Philox) or, with external randoms, is the reference's
//        double product of the float scale and a double normal; the log-density is evaluated on the proposal rounded to float
// Register budget (second __launch_bounds__ argument: waves per SIMD the allocation must allow).  The float kernels of
// the widest class need 173 - 176 VGPRs unconstrained - a handful over the 168 that let a third wave share the SIMD -
// and fit 168 with at most 24 B of scratch (compiled-in dim 100: none); everything else is left to the allocator
// (W <= 24: already <= 168; double state: 256; the FULL twins trace to HBM anyway).
constexpr int quad_min_waves(int w, bool f64, bool full) { return (w >= 28 && !f64 && !full) ? 3 : 1; }

template <class Target, class Proposal, int W, int DEXACT, int MAXT, bool FULL, bool F64 = false>
__global__ void __launch_bounds__(MAXT, quad_min_waves(W, F64, FULL)) ptrwm_quad_step_kernel(const KArgs a) {
  typedef typename quad_state<F64>::type state_t;
  constexpr int SW = F64 ? 2 : 1;  // 32-bit words per state element
  const int T = a.n_temps;
  const int D = DEXACT ? DEXACT : a.dim;
  const int cpw = a.chains_per_wave;  // ladders per exchange group (narrow: 16 / T per wave; wide: per workgroup)
  const bool wide = 4 * T > 64;       // grid-uniform: the exchange group is the whole workgroup
  const int tid = wide ? (int)threadIdx.x : (int)(threadIdx.x & 63);
  const int gthreads = wide ? (int)blockDim.x : 64;
  const int nslots = gthreads >> 2;
  const long long chain0 =
      (wide ? (long long)blockIdx.x : (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * cpw;
  if (chain0 >= a.n_chains) return;  // narrow only (wave-uniform; narrow waves never meet at a workgroup barrier)
  const int slot_raw = tid >> 2;
  QLane l;
  l.q = tid & 3;
  l.d0 = l.q * W;
  {
    const int left = D - l.d0;
    l.n_own = left < 0 ? 0 : (left > W ? W : left);
  }
  const int cw_raw = slot_raw / T;
  const int t_raw = slot_raw - cw_raw * T;
  const bool live = (cw_raw < cpw) && (chain0 + cw_raw < a.n_chains);
  // idle quads shadow replica (chain0, 0): they compute but never store and are never an exchange source
  const int slot = live ? slot_raw : 0;
  const int cw = live ? cw_raw : 0;
  const int t = live ? t_raw : 0;
  const long long chain = chain0 + cw;
  const long long rep = chain0 * T + slot;

  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  float *const rows_f = s_dyn + (wide ? 0 : (int)(threadIdx.x >> 6) * (64 * (W * SW + 2)));
  state_t *const rows = reinterpret_cast<state_t *>(rows_f);
  float *const s_l = rows_f + nslots * (4 * W * SW);
  float *const s_u = s_l + nslots;
  int *const s_landed = reinterpret_cast<int *>(s_u + nslots);
  auto sync_group = [&]() {
    if (wide) {
      __syncthreads();
    } else {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  };
  constexpr int MIN_OWN = DEXACT ? (DEXACT - 3 * W > 0 ? (DEXACT - 3 * W > W ? W : DEXACT - 3 * W) : 0) : -1;

  // ---- state load: the group's live replicas are one contiguous run of elements: coalesced copy into the slab (as 32-bit
  // words: two per element in the F64 form, whose runs start 8-byte aligned, so the slab offset stage_head is even),
  // then every lane picks its quarter of its replica's row (row stride = dim)
  state_t x[W];
  float y[W];
  [[maybe_unused]] double yd[F64 ? W : 1];
  {
    const long long live_chains = (a.n_chains - chain0 < cpw) ? (a.n_chains - chain0) : cpw;
    const int stage_total = (int)live_chains * T * D * SW;
    float *__restrict__ gs = a.state + chain0 * T * (long long)D * SW;
    stage_copy<true>(rows_f, gs, stage_total, tid, gthreads);
    s_landed[nslots + slot_raw] = 0;  // the ladders' swap-form objections (wide groups: the vote of a swap event, below)
    sync_group();
    const state_t *seg = reinterpret_cast<const state_t *>(rows_f + stage_head(gs)) + slot * D + l.d0;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      x[j] = q_valid<MIN_OWN>(l, j) ? seg[j] : (state_t)0;
      y[j] = 0.0f;
      if constexpr (F64) yd[j] = 0.0;
    }
  }
  float lp = a.logp[rep];
  const float beta_t = a.beta[t];
  const float tscale = a.temp_scale[t];
  // may the proposal's own squared increment stand for |y - x|^2 for this replica?  (proposals.h kJumpTrust: the replica's
  // largest coordinate, over all four lanes, against the increment scale - the same verdict as the thread form's)
  bool jump_trusted = false;
  if constexpr (Proposal::kKnowsJump && !F64) {
    float xmax = 0.0f;
#pragma unroll
    for (int j = 0; j < W; ++j) xmax = __builtin_fmaxf(xmax, __builtin_fabsf((float)x[j]));  // (slots not owned hold 0)
    xmax = __builtin_fmaxf(xmax, dpp_f<kDppSwapPair>(xmax));
    xmax = __builtin_fmaxf(xmax, dpp_f<kDppSwapHalf>(xmax));
    jump_trusted = xmax <= kJumpTrust * Proposal::increment_scale(tscale, a.pp);
  }

  const unsigned long long gchain = (unsigned long long)(a.chain_offset + chain);
  RngCtx rc;
  rc.c2 = (uint32_t)gchain;
  rc.k0 = a.k0;
  rc.k1 = a.k1;
  const uint32_t c3_base = (uint32_t)t | ((uint32_t)(gchain >> 32) << 12);

  unsigned n_acc = 0, n_swap_acc = 0;
  int last_event = -1;
  double sq = 0.0;

  const bool ext = FULL && a.full.ext_prop != nullptr;
  const bool trace_on =
      FULL && live && a.full.trace != nullptr && (chain < a.full.trace_chains) && (t < a.full.trace_temps);
  int to_swap = a.steps_to_swap;
  int to_trace = FULL ? a.full.steps_to_trace : 0;
  int trace_rows = 0;
  int swap_in_call = 0;
  const int ev_par0 = (int)(a.first_swap_event & 1);
  unsigned long long s = (unsigned long long)a.step0;

  for (int i = 0; i < a.n_steps; ++i, ++s) {
    const bool count_on = i >= a.burn_left;
    --to_swap;
    const bool multiple = (to_swap == 0);
    if (multiple) to_swap = a.swap_every;
    const bool swap_due = multiple && count_on && (T > 1);

    rc.c0hi = (uint32_t)(s >> 32) << 16;
    rc.c1 = (uint32_t)s;
    rc.c3 = c3_base | (kStreamMH << 8);

    long long srep = 0;
    const float *ext_rep = nullptr;
    float ext_u = 0.0f;
    if constexpr (FULL) {
      srep = ((long long)i * a.n_chains + chain) * T + t;
      if (ext) {
        ext_rep = a.full.ext_prop + srep * a.full.n_raw_ext * SW;
        ext_u = a.full.ext_u[srep];
      }
    }

    float u_acc;
    float jump = 0.0f;  // this lane's range of the squared jump, if the proposal provides it (proposals.h)
    int jump_kind = kJumpNone;
    if constexpr (!F64) {
      u_acc = Proposal::propose(y, x, l, D, tscale, a.pp, rc, ext_rep, ext_u, jump, jump_kind);
      if (jump_kind == kJumpPartial) jump = quad_tree_add(jump);  // the canonical combination, now: one live register
    } else {
      if (ext_rep != nullptr) {
        // the reference's float64 path: increments = bmm(diag(scale).double(), randn(float64)); proposals = states +
        // increments (pt_rwm_gpu_optimized.py:445-455,546-547,576-592): one double product, one double sum
        const double *er = reinterpret_cast<const double *>(ext_rep) + l.d0;
        const double ts = (double)tscale;
#pragma unroll
        for (int j = 0; j < W; ++j)
          if (q_valid<MIN_OWN>(l, j)) yd[j] = dadd_rn(x[j], dmul_rn(er[j], ts));
        u_acc = ext_u;
      } else {
        // Philox: the float proposal functor run from the origin gives the increment; the sum with the state is double
        float zero[W];
#pragma unroll
        for (int j = 0; j < W; ++j) zero[j] = 0.0f;
        float jump64 = 0.0f;  // (double states take their jump from the states themselves)
        int kind64;
        u_acc = Proposal::propose(y, zero, l, D, tscale, a.pp, rc, nullptr, 0.0f, jump64, kind64);
#pragma unroll
        for (int j = 0; j < W; ++j)
          if (q_valid<MIN_OWN>(l, j)) yd[j] = dadd_rn(x[j], (double)y[j]);
      }
#pragma unroll
      for (int j = 0; j < W; ++j) y[j] = q_valid<MIN_OWN>(l, j) ? (float)yd[j] : 0.0f;  // what the density is evaluated on
    }
    const float lp_new = Target::logp(y, l, D, a.tp);

    const bool acc = mh_accept(beta_t, lp_new, lp, u_acc);
    const float lp_mh = acc ? lp_new : lp;
    if constexpr (FULL) {
      if (a.full.accept_flags != nullptr && live && l.q == 0) a.full.accept_flags[srep] = acc ? 1 : 0;
    }

    state_t j2l = 0;  // this lane's range of the squared jump
    state_t j2;
    if (!swap_due && jump_kind != kJumpNone) {
      // (float states only) the proposal knows the length of its own increment: the move is one select per dimension;
      // replicas that may not trust it (kernel.h) take the jump from the states
      float from_states = 0.0f;
      if (!jump_trusted) {
        PTRWM_COLD_PATH();
        float p = 0.0f;
#pragma unroll
        for (int j = 0; j < W; ++j)
          if (q_valid<MIN_OWN>(l, j)) {
            const float dl = sub_rn(y[j], (float)x[j]);
            p = fmaf(dl, dl, p);
          }
        from_states = quad_tree_add(p);
      }
#pragma unroll
      for (int j = 0; j < W; ++j)
        if (q_valid<MIN_OWN>(l, j)) x[j] = acc ? (state_t)y[j] : x[j];
      j2 = acc ? (state_t)(jump_trusted ? jump : from_states) : (state_t)0;
      lp = lp_mh;
    } else if (!swap_due) {
#pragma unroll
      for (int j = 0; j < W; ++j) {
        if (q_valid<MIN_OWN>(l, j)) {
          if constexpr (F64) {
            const double dl = dsub_rn(yd[j], x[j]);
            j2l = __builtin_fma(dl, dl, j2l);
            x[j] = acc ? yd[j] : x[j];
          } else {
            const float dl = sub_rn(y[j], x[j]);
            j2l = fmaf(dl, dl, j2l);
            x[j] = acc ? y[j] : x[j];
          }
        }
      }
      j2 = quad_tree_add(j2l);
      if (!acc) j2 = 0;
      lp = lp_mh;
    } else {
      // ---- temperature swaps on the post-MH log-densities: kernel.h's swap_decide over replica slots ----------
      const int base = live ? slot - t : 0;
      int src = slot;
      float my_l = lp_mh;
      bool pair_acc = false;
      float us;
      if (ext) {
        us = (t < T - 1) ? a.full.ext_swap_u[((long long)swap_in_call * a.n_chains + chain) * (T - 1) + t] : 2.0f;
      } else {
        const u32x4 r = philox4x32_10(rc.c0hi, rc.c1, rc.c2, c3_base | (kStreamSwap << 8), rc.k0, rc.k1);
        us = u01(r.x);
      }
      // (the previous event's reads of s_l / the outcome slots are separated from these writes by its two row-exchange
      // synchronisations)
      s_l[slot_raw] = my_l;  // the four lanes of a quad write the same value
      s_u[slot_raw] = us;
      // may this ladder's sequential sweep take the threshold form in THIS event?  A verdict of the ladder alone, from the
      // values it enters the event with (kernel.h swap_pair_plain): narrow groups vote over the ladder's 4 T lanes of the
      // wavefront; a workgroup holding several whole ladders votes through one flag per ladder behind the outcome slots
      // (`base` is the ladder's first slot, unique to it)
      const bool pair_plain = swap_pair_plain(T, t, sub_rn(beta_t, a.beta[t < T - 1 ? t + 1 : t]), my_l, us);
      bool swap_plain;
      if (!wide) {
        swap_plain = ladder_votes_plain(pair_plain, 4 * base, 4 * T);
        sync_group();
      } else {
        // An objection is the event's own stamp (1 + its index in the call) in the ladder's word, written together with
        // the published values - the one barrier below orders both, the vote has no barrier of its own - and nothing has
        // to be armed again: the stamp of an earlier event is not this event's.  (The word is read for the last time
        // before this event's row-exchange barrier, the next objection is written after it.)
        if (!pair_plain) s_landed[nslots + base] = swap_in_call + 1;
        sync_group();
        swap_plain = s_landed[nslots + base] != swap_in_call + 1;
      }
      if (wide && a.swap_order == PTRWM_ORDER_SEQUENTIAL && a.swap_mode == PTRWM_SWAP_EXCHANGE) {
        // A wide ladder spans several wavefronts; the sequential sweep (kernel.h swap_decide: a scan over the ladder's
        // published values that every thread replays) would be replayed by every one of them - four times the work per
        // ladder of the one-thread-per-replica kernel.  Here ONE lane per ladder runs the scan and publishes, for every
        // position, the slot whose vector lands there; a barrier later everyone picks up its own position.  Same
        // decisions, same values: the threshold form of swap_decide (every quad turns its pair's uniform into a threshold
        // on the carried log-density first) or, for ladders it does not cover, the literal scan.
        if ((int)threadIdx.x < cpw) {  // lane i of the first wavefront scans ladder i of the workgroup
          const int b0 = (int)threadIdx.x * T;
          float car_l = s_l[b0];
          int car_i = b0;
          if (s_landed[nslots + b0] != swap_in_call + 1) {  // the verdict of ladder i (voted above), not of this lane's own ladder
            // (the scanning lane builds each pair's threshold itself - they do not depend on the carried state, so the
            // logs and reciprocals of successive pairs overlap - instead of a third barrier to have them published)
#pragma unroll 4
            for (int j = 0; j < T - 1; ++j) {
              const float lk = s_l[b0 + j + 1];
              const float thr = fmaf(-hw_ln(s_u[b0 + j]), __builtin_amdgcn_rcpf(sub_rn(a.beta[j], a.beta[j + 1])), lk);
              const bool ok = car_l < thr;
              s_landed[b0 + j] = ok ? b0 + j + 1 : car_i;
              car_l = ok ? car_l : lk;
              car_i = ok ? car_i : b0 + j + 1;
            }
          } else {
#pragma unroll 2
            for (int j = 0; j < T - 1; ++j) {
              const float lk = s_l[b0 + j + 1];
              const float u = s_u[b0 + j];
              const bool ok = swap_accept_test(u, swap_log_prob(a.beta[j], a.beta[j + 1], car_l, lk));
              s_landed[b0 + j] = ok ? b0 + j + 1 : car_i;
              car_l = ok ? car_l : lk;
              car_i = ok ? car_i : b0 + j + 1;
            }
          }
          s_landed[b0 + T - 1] = car_i;
        }
        __syncthreads();
        src = s_landed[base + t];
        my_l = s_l[src];
        pair_acc = (t < T - 1) && (src == base + t + 1);
      } else {
        swap_decide(T, t, base, slot, a.swap_mode, a.swap_order, (ev_par0 + swap_in_call) & 1, a.beta, beta_t, us, s_l, s_u,
                    s_landed, my_l, src, pair_acc, sync_group, swap_plain);
      }
      if (pair_acc) {
        n_swap_acc += 1;
        last_event = swap_in_call;
      }
      {
        state_t *my_seg = rows + slot_raw * D + l.d0;
#pragma unroll
        for (int j = 0; j < W; ++j)
          if (q_valid<MIN_OWN>(l, j)) {
            if constexpr (F64) my_seg[j] = acc ? yd[j] : x[j];
            else my_seg[j] = acc ? y[j] : x[j];
          }
        sync_group();
        const state_t *src_seg = rows + src * D + l.d0;
#pragma unroll
        for (int j = 0; j < W; ++j) {
          if (q_valid<MIN_OWN>(l, j)) {
            const state_t w = src_seg[j];
            if constexpr (F64) {
              const double dl = dsub_rn(w, x[j]);
              j2l = __builtin_fma(dl, dl, j2l);
            } else {
              const float dl = sub_rn(w, x[j]);
              j2l = fmaf(dl, dl, j2l);
            }
            x[j] = w;
          }
        }
        j2 = quad_tree_add(j2l);
      }
      lp = my_l;
      ++swap_in_call;
    }

    if (count_on) {
      n_acc += acc ? 1u : 0u;
      sq += (double)j2;
    }
    if constexpr (FULL) {
      bool trace_now = false;
      if (a.full.trace != nullptr) {
        --to_trace;
        trace_now = (to_trace == 0);
        if (trace_now) to_trace = a.full.trace_every;
      }
      if (trace_now && trace_on) {
        const long long row = ((a.full.trace_row0 + trace_rows) * a.full.trace_chains + chain) * a.full.trace_temps + t;
        state_t *__restrict__ tr = reinterpret_cast<state_t *>(a.full.trace) + row * D + l.d0;
#pragma unroll
        for (int j = 0; j < W; ++j)
          if (q_valid<MIN_OWN>(l, j)) tr[j] = x[j];
        if (a.full.trace_logp != nullptr && l.q == 0) a.full.trace_logp[row] = lp;
      }
      trace_rows += trace_now ? 1 : 0;
    }
  }

  // ---- state store: quarters -> slab rows -> coalesced HBM writes ---------------------------------------------
  const kargs_ptr ae = late_args();  // (kernel.h) the epilogue's arguments are loaded here, not held across the step loop
  {
    const long long n_chains2 = ae->n_chains;
    const long long live_chains = (n_chains2 - chain0 < cpw) ? (n_chains2 - chain0) : cpw;
    const int stage_total = (int)live_chains * T * D * SW;
    float *__restrict__ gs = ae->state + chain0 * T * (long long)D * SW;
    sync_group();  // the last swap's row reads are done before the rows are overwritten
    if (live) {
      state_t *seg = reinterpret_cast<state_t *>(rows_f + stage_head(gs)) + slot * D + l.d0;
#pragma unroll
      for (int j = 0; j < W; ++j)
        if (q_valid<MIN_OWN>(l, j)) seg[j] = x[j];
    }
    sync_group();
    stage_copy<false>(rows_f, gs, stage_total, tid, gthreads);
  }
  if (live && l.q == 0) {
    ae->logp[rep] = lp;
    if (ae->n_accept != nullptr && n_acc != 0u) count_add(&ae->n_accept[rep], (long long)n_acc);  // (kernel.h: only where there is a delta)
    if (ae->sq_jump != nullptr && sq != 0.0) ae->sq_jump[rep] += sq;
    if (ae->swap_accept != nullptr && n_swap_acc != 0u) count_add(&ae->swap_accept[rep], (long long)n_swap_acc);
    if (ae->last_swap_ordinal != nullptr && last_event >= 0) {
      const long long ev = ae->first_swap_event + last_event;
      const long long ord = (ae->swap_order == PTRWM_ORDER_SEQUENTIAL) ? ev * (T - 1) + t + 1 : ev + 1;
      if (ord > ae->last_swap_ordinal[rep]) ae->last_swap_ordinal[rep] = ord;
    }
  }
}

}  // namespace ptrwm
