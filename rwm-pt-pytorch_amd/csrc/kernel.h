// The fused PT-RWM kernel: one (chain, temperature) replica per thread, the
// temperatures of one chain contiguous inside one 64-lane wavefront.
//
// Replaces, per launch, n_steps iterations of
//   algorithms/rwm_gpu_optimized.py:289-336    (_single_step_ultra_fused, T = 1)
//   algorithms/pt_rwm_gpu_optimized.py:541-574 (step) + :594-633 (_attempt_all_swaps)
// with the replica's dim-vector, log-density and counters held in registers for
// the whole launch: HBM is touched once to load and once to store the state.
//
// Thread map.  T <= 64 ("narrow"): an exchange group is one wavefront, lane = cw * T + t (cw = chain slot within
// the wave, t = temperature), chains_per_wave = 64 / T; lanes >= chains_per_wave*T idle (only when T does not divide
// 64); a workgroup is four independent groups (measured 2 % faster than one-wave workgroups).  64 < T <= 256
// ("wide"): one ladder per workgroup of ceil(T / 64) waves, t = threadIdx.x.
// Either way a ladder lives inside one workgroup and swaps go through LDS (broadcast reads of the ladder's
// log-densities, row exchange through the staging rows), never HBM.
#pragma once
#include "philox.h"
#include "proposals.h"
#include "targets.h"

namespace ptrwm {

#ifndef PTRWM_BLOCK_THREADS
#define PTRWM_BLOCK_THREADS 256
#endif
constexpr int kBlockThreads = PTRWM_BLOCK_THREADS;
constexpr int kWavesPerBlock = kBlockThreads / 64;
// dynamic LDS bytes of a step-kernel workgroup of `threads` threads with register width dp: one row of dp floats
// per thread plus its log-density, swap-uniform and swap-outcome slots, three words that live for the whole
// launch but are touched only in swap events and the epilogue (parked in LDS to keep them out of the VGPR budget
// of the MH part: without that the compiler spilled five VGPRs to scratch, 42 MB of HBM traffic per launch) and the
// launch's sum of squared jumps (a double: two words, added to by ds_add_f64 - profiles/r04_scratch_ab.txt)
constexpr int kLdsExtraPerThread = 8;  // s_l, s_u, landed (swap-event scratch); c3, swap count, last event, the squared-jump sum (a double) (whole launch)
// (the streaming form, STREAM below, keeps TWO slabs of rows per wave: the one its current group lives in and the one the
// next group's state is landing in)
// and two small landing zones for the next group's log-densities (one float per thread), squared-jump sums (one double) and
// acceptance counts (one 64-bit integer)
constexpr int kLdsStreamStatPerThread = 5;  // floats per thread and landing zone
constexpr int kStreamSlabs = 2;
constexpr int lds_floats_per_thread(int dp, bool stream) {
  return (stream ? kStreamSlabs * (dp + kLdsStreamStatPerThread) : dp) + kLdsExtraPerThread;
}
constexpr unsigned step_kernel_lds_bytes(int threads, int dp, bool stream = false) {
  return (unsigned)(threads * lds_floats_per_thread(dp, stream)) * 4u;
}
// a wide group (one ladder = the workgroup, n_temps > 64) has one more word behind all of that: the ladder's objection to
// the threshold form of a swap event (ptrwm_step_kernel, the swap section)
constexpr unsigned kWideVoteBytes = 16u;

// Arguments only the fixture / trace variant of the kernel (FULL = true) reads.  Keeping them out
// of the production variant keeps its wave-uniform state inside the 100-odd SGPRs of a wave.
struct FullArgs {
  const float *__restrict__ ext_prop;
  const float *__restrict__ ext_u;
  const float *__restrict__ ext_swap_u;
  float *__restrict__ trace;
  float *__restrict__ trace_logp;
  unsigned char *__restrict__ accept_flags;
  long long trace_chains, trace_row0;
  int trace_temps, n_raw_ext;
  int trace_every;     // thinning period (>= 1)
  int steps_to_trace;  // steps until the first traced step of this launch (1 = the first step)
};

struct KArgs {
  float *__restrict__ state;
  float *__restrict__ logp;
  const float *__restrict__ beta;
  const float *__restrict__ temp_scale;
  long long *__restrict__ n_accept;
  double *__restrict__ sq_jump;
  long long *__restrict__ swap_accept;
  long long *__restrict__ last_swap_ordinal;
  long long n_chains, chain_offset, step0;
  long long first_swap_event;  // 0-based index (since the start of the run) of the first swap event in this call
  int n_steps;                 // steps in this launch (the C ABI splits longer requests)
  int burn_left;               // steps of this launch that still belong to burn-in (step_counter <= burn_in)
  int n_temps, dim, swap_every, swap_mode, swap_order, chains_per_wave;
  int steps_to_swap;  // steps until the next step whose step_counter is a multiple of swap_every (1 = the first step)
  unsigned k0, k1;
  TParams tp;
  PParams pp;
  FullArgs full;
};

// log swap probability exactly as fused_swap_probability_calculation evaluates it
// (algorithms/pt_rwm_gpu_optimized.py:37-48): four products summed left to right.
__device__ __forceinline__ float swap_log_prob(float bj, float bk, float lj, float lk) {
  return sub_rn(sub_rn(add_rn(mul_rn(bj, lk), mul_rn(bk, lj)), mul_rn(bj, lj)), mul_rn(bk, lk));
}

// swap_random < min(1, exp(log_prob))  (:617-621).  torch.min propagates NaN and NaN compares
// false; v_min_f32 would drop the NaN, so the clamp is a select of the threshold (one compare on the result:
// selecting between two finished compares costs the compiler four more instructions per pair).
__device__ __forceinline__ bool swap_accept_test(float u, float log_prob) {
  const float threshold = (log_prob >= 0.0f) ? 1.0f : hw_exp(log_prob);  // NaN log_prob -> NaN threshold -> false
  return u < threshold;
}

// ultra_fused_mcmc_step_basic (rwm_gpu_optimized.py:9-32) / ultra_fused_parallel_mcmc_step
// (pt_rwm_gpu_optimized.py:62-84):  r = beta (l' - l);  accept = (r > 0) | (u < exp r).  Shared by the fused step
// kernel and the split-step accept kernel.
__device__ __forceinline__ bool mh_accept(float beta_t, float lp_new, float lp, float u_acc) {
  const float ratio = mul_rn(beta_t, sub_rn(lp_new, lp));
  return (ratio > 0.0f) || (u_acc < hw_exp(ratio));
}

// The decision part of one swap event (pt_rwm_gpu_optimized.py:594-633), shared by the fused step kernel and the
// stand-alone sweep kernel (capi.hip).  In: this thread's temperature t (0 for idle threads), base = slot of
// temperature 0 of its ladder, slot = base + t, us = its swap uniform, my_l = its log-density; s_l / s_u = the
// ladder's published log-densities and uniforms (already synchronised), landed = one int of LDS scratch per slot;
// par = parity of the event (even/odd order); sync = the group's barrier (called by every thread of the group, or by
// none); plain = this thread's ladder takes the threshold form of the sequential sweep in this event (the same value on
// every thread of the ladder, below; ladders of one wavefront may differ).
// Out: my_l = the log-density that ends up at temperature t, src = the slot whose vector does, pair_acc = pair
// (t, t+1) accepted (recorded on the thread of temperature t).
//
// Sequential exchange sweep, threshold form.  The sweep j = 0..T-2 carries one state upward: at pair j the state now at
// position j (log-density c) meets the untouched state of position j+1 (log-density l_k), and the reference accepts when
//   u_j < min(1, exp((b_j - b_k)(l_k - c)))          (fused_swap_probability_calculation, :37-48, :617-621).
// For b_j > b_k (a ladder ordered cold to hot) that is   c < l_k - ln(u_j) / (b_j - b_k) =: thr_j,   and thr_j does not
// depend on the carried state: every thread computes the threshold of its own pair ONCE (one v_log_f32, one v_rcp_f32),
// publishes it in place of its uniform, and the scan that every thread of the ladder replays shrinks from four products,
// three sums, an exponential and two compares per pair to ONE compare per pair (plus the selects that carry the state):
// ~22 -> ~6 VALU instructions per pair, 4 % of BASELINE configs[2]'s instructions.  Same decisions up to rounding of the
// threshold (both forms are within a few ulp of the exact boundary; tests prove every decision that differs from the
// oracle's literal evaluation).  The reference's corner cases keep its literal rule: a LADDER in which any pair has
// b_j <= b_k, any replica enters the event with a log-density of -inf or NaN (the reference's four-product sum is then NaN
// and the swap is refused), or any pair's uniform is exactly 0 or >= 1 (ln u = -inf: the threshold would accept where
// u < exp(..) with an underflowed exponential refuses; u = 1 can never accept) takes the literal scan below - `plain`, a
// verdict of the ladder alone, taken at every event from the values it enters the event with (ladder_votes_plain): it
// cannot depend on which other ladders share the exchange group (kernel form, sharding) nor on where launches are cut.
// this thread's part of its ladder's verdict: its pair's temperatures are ordered (db = b_t - b_{t+1} > 0), its log-density
// is finite and its pair's uniform lies strictly inside (0, 1) (the last temperature has no pair: neither db nor the
// uniform is looked at).  ONE compare: u (1 - u) > 0 exactly for 0 < u < 1, lp * 0 is 0 for a finite log-density and NaN
// otherwise - three lane masks less in the SGPR file than the four compares spelt out (the run-time-dim lane-split kernels
// of the dim > 64 class live at the spilled-SGPR ceiling of tools/kernel_stats.py).
__device__ __forceinline__ bool swap_pair_plain(int T, int t, float db, float lp, float us) {
  const bool last = t >= T - 1;
  const float u = last ? 0.5f : us;
  const float d = last ? 1.0f : db;
  const float inside = mul_rn(u, sub_rn(1.0f, u));
  return add_rn(__builtin_fminf(inside, d), mul_rn(lp, 0.0f)) > 0.0f;  // (a NaN anywhere compares false)
}
// the AND of `mine` over the `lanes_per_ladder` consecutive lanes of a wavefront that start at `first_lane` (the ladder
// this thread belongs to), the same value on every one of them
__device__ __forceinline__ bool ladder_votes_plain(bool mine, int first_lane, int lanes_per_ladder) {
  const unsigned long long against = __builtin_amdgcn_ballot_w64(!mine);
  const unsigned long long ladder = (lanes_per_ladder >= 64 ? ~0ull : ((1ull << lanes_per_ladder) - 1ull)) << first_lane;
  return (against & ladder) == 0ull;
}

template <class Sync>
__device__ __forceinline__ void swap_decide(int T, int t, int base, int slot, int swap_mode, int swap_order, int par,
                                            const float *__restrict__ beta, float beta_t, float us, const float *s_l,
                                            float *s_u, int *landed, float &my_l, int &src, bool &pair_acc, Sync sync,
                                            bool plain) {
  if (swap_order == PTRWM_ORDER_SEQUENTIAL) {
    if (swap_mode == PTRWM_SWAP_EXCHANGE) {
      const bool has_k = t < T - 1;
      const float db = has_k ? sub_rn(beta_t, beta[t + 1]) : 1.0f;
      float car_l = s_l[base];
      int car_i = base;
      if (plain) {
        // own pair's threshold in place of its uniform (u = 0: ln = -inf, threshold +inf: accepted, as u < exp(..) is)
        const float lk = s_l[has_k ? slot + 1 : slot];
        s_u[slot] = fmaf(-hw_ln(us), __builtin_amdgcn_rcpf(db), lk);
        sync();
#pragma unroll 2
        for (int j = 0; j < T - 1; ++j) {
          const int ik = base + j + 1;
          const float lkj = s_l[ik];
          const bool ok = car_l < s_u[base + j];
          landed[base + j] = ok ? ik : car_i;
          car_l = ok ? car_l : lkj;
          car_i = ok ? car_i : ik;
        }
      } else {
        // the literal scan: every thread of the ladder replays the sweep from the published original values (uniform
        // addresses: LDS broadcasts, no dependent cross-lane traffic) and keeps what lands on its own position
#pragma unroll 2
        for (int j = 0; j < T - 1; ++j) {
          const float lk = s_l[base + j + 1];
          const float u = s_u[base + j];
          const float bj = beta[j], bk = beta[j + 1];
          const bool ok = swap_accept_test(u, swap_log_prob(bj, bk, car_l, lk));
          const int ik = base + j + 1;
          // Every thread of the ladder computes the same values, so the outcome for position j (which slot's vector
          // lands there) is written to LDS by all of them, identically, and each thread picks up its own position
          // after the loop: one ds_write per pair instead of a compare and three selects on t == j.
          landed[base + j] = ok ? ik : car_i;
          car_l = ok ? car_l : lk;
          car_i = ok ? car_i : ik;
        }
      }
      landed[base + T - 1] = car_i;
      // (a wave's LDS operations complete in program order, and every wave of a wide ladder writes all positions
      // itself before reading its own: no barrier needed)
      src = landed[base + t];
      my_l = s_l[src];  // log-densities travel with the states: s_l still holds the originals
      pair_acc = (t < T - 1) && (src == base + t + 1);
    } else {
      // reference_copy: row j <- row k, row k untouched, so every pair compares the original rows j and j+1:
      // no carried state, fully parallel.
      if (t < T - 1) {
        const float lk = s_l[slot + 1];
        const bool ok = swap_accept_test(us, swap_log_prob(beta_t, beta[t + 1], my_l, lk));
        if (ok) {
          my_l = lk;
          src = slot + 1;
        }
        pair_acc = ok;
      }
    }
  } else {
    // even/odd: event n attempts the disjoint pairs (j, j+1) with j == n (mod 2)
    const bool lower = ((t & 1) == par);          // this thread is the lower index j of its pair
    const int partner_t = lower ? t + 1 : t - 1;
    const bool valid = partner_t >= 0 && partner_t < T;
    if (valid) {
      const int partner = base + partner_t;
      const float l_other = s_l[partner];
      const float u_low = lower ? us : s_u[partner];
      const int tj = lower ? t : partner_t;
      const float lj = lower ? my_l : l_other;
      const float lk = lower ? l_other : my_l;
      const bool ok = swap_accept_test(u_low, swap_log_prob(beta[tj], beta[tj + 1], lj, lk));
      if (ok && (lower || swap_mode == PTRWM_SWAP_EXCHANGE)) {
        my_l = l_other;
        src = partner;
      }
      pair_acc = ok && lower;
    }
  }
}

// Staging between a group's contiguous run of `total` floats in HBM (`g`, any 4-byte alignment) and its LDS slab, 16 bytes
// per lane per instruction (global_load_dwordx4 / ds_write_b128 and the reverse).  Element i of the run lives at
// lds[stage_head(g) + i], stage_head = the run's first float index modulo 4: an aligned 16-byte vector of HBM is then an
// aligned 16-byte vector of the slab (slabs start 256-byte aligned), so whole vectors move as vectors; the <= 3 elements
// before the first and after the last whole vector are moved one by one.  The slab needs room for total + 3 floats: the
// rows fill it exactly when every slot is live and dim equals the register width, and the overhang then lands in the
// swap scratch behind the rows, which is dead while a copy runs.
// Four independent vectors in flight per lane: full chunks unpredicated, then one predicated chunk.  (A plain loop waits
// for every load before issuing the next one when the stride is a run-time value: 1.7x slower for one step per launch.)
__device__ __forceinline__ int stage_head(const float *g) { return (int)((reinterpret_cast<uintptr_t>(g) >> 2) & 3u); }

template <bool LOAD>  // LOAD: HBM -> LDS, else LDS -> HBM
__device__ __forceinline__ void stage_copy(float *__restrict__ lds, float *__restrict__ g, int total, int tid, int nthr) {
  typedef float vec4 __attribute__((ext_vector_type(4)));
  const int head = stage_head(g);
  vec4 *__restrict__ gv = reinterpret_cast<vec4 *>(g - head);  // the aligned frame; vector 0 is touched only if head == 0
  vec4 *__restrict__ lv = reinterpret_cast<vec4 *>(lds);
  const int v_lo = head ? 1 : 0;          // first whole vector
  const int v_hi = (head + total) >> 2;   // one past the last whole vector
  constexpr int kDepth = 4;
  int v0 = v_lo + tid;
  for (; v0 + (kDepth - 1) * nthr < v_hi; v0 += kDepth * nthr) {
    vec4 t[kDepth];
#pragma unroll
    for (int k = 0; k < kDepth; ++k) t[k] = LOAD ? gv[v0 + k * nthr] : lv[v0 + k * nthr];
#pragma unroll
    for (int k = 0; k < kDepth; ++k) (LOAD ? lv : gv)[v0 + k * nthr] = t[k];
  }
  if (v0 < v_hi) {
    vec4 t[kDepth - 1];
#pragma unroll
    for (int k = 0; k < kDepth - 1; ++k) {
      const int v = v0 + k * nthr;
      if (v < v_hi) t[k] = LOAD ? gv[v] : lv[v];
    }
#pragma unroll
    for (int k = 0; k < kDepth - 1; ++k) {
      const int v = v0 + k * nthr;
      if (v < v_hi) (LOAD ? lv : gv)[v] = t[k];
    }
  }
  // the ragged ends: elements [0, e_lo) before the first whole vector and [e_hi, total) after the last one
  const int e_lo = (4 * v_lo - head < total) ? 4 * v_lo - head : total;
  const int e_hi = (4 * v_hi - head > e_lo) ? 4 * v_hi - head : e_lo;
  if (tid < 6) {
    const int i = tid < 3 ? tid : e_hi + (tid - 3);
    const bool mine = tid < 3 ? (i < e_lo) : (i < total);
    if (mine) {
      if (LOAD) lds[head + i] = g[i];
      else g[i] = lds[head + i];
    }
  }
}

// The thread's index in its (one-dimensional) workgroup, re-derived from the hardware where it is needed - the lane from
// v_mbcnt, the wave's index in the workgroup from an SGPR set once at kernel entry (lanes fill the waves of a 1-D workgroup
// in order) - so that neither threadIdx.x nor anything computed from it has to live in a VGPR, i.e. in scratch, across the
// step loop: at the 128-VGPR cap the allocator spilled it and four words derived from it (28 B of scratch per thread,
// 13 % of a launch's HBM traffic at 2 000 steps per launch).  Used by the swap path and the epilogue alike: with the
// epilogue alone re-deriving it the headline kernel ran 2 % slower (95.1 against 93.1 ms per 2 000-step launch, same box,
// profiles/r03_scratch_ab.txt) - register allocation at the cap is that sensitive; candidates are A/B-timed.
__device__ __forceinline__ int wave_in_block() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ int thread_index_now(int wave) {
  unsigned all = ~0u;
  PTRWM_VALUE_BARRIER("+s"(all));  // (v_mbcnt is pure: without this it is computed once, before the loop, and kept)
  return wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(all, __builtin_amdgcn_mbcnt_lo(all, 0u));
}

// `*p += v` for an integer statistic that this thread alone updates, as a no-return atomic executed at the L2
// (global_atomic_add_x2): the same sum without a load, a dependent add and a store at the very end of a wave's life - the
// tail that keeps the wave's slot occupied in short launches (one step per launch: 0.130 -> 0.125 ms; long launches: no
// change).  The double sum of squared jumps stays a read-modify-write: a hardware fp64 atomic would do the same for another
// 1 %, but is not guaranteed on every kind of device allocation a caller of the C ABI may hand in.
__device__ __forceinline__ void count_add(long long *p, long long v) {
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool EXACT>
__device__ __forceinline__ int fresh_dim(int d0) {
  if constexpr (!EXACT) PTRWM_VALUE_BARRIER("+s"(d0));
  return d0;
}

// The kernel's arguments as they sit in the kernarg segment (KArgs is the kernel's only parameter: offset 0), through a
// pointer the optimiser cannot relate to `a`: what is read through it is loaded where it is used - one s_load in the
// epilogue - instead of being loaded at kernel entry and held in SGPRs, i.e. in spill lanes, across the whole step loop
// (the epilogue alone needs eight pointers and three scalars: ~20 SGPRs of the 102 a wave has).
typedef const __attribute__((address_space(4))) KArgs *kargs_ptr;
__device__ __forceinline__ kargs_ptr late_args() {
  uintptr_t p = (uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  PTRWM_VALUE_BARRIER("+s"(p));
  return (kargs_ptr)p;
}

// Register budget: the replica's x[DP] and y[DP] plus ~30 temporaries must stay in VGPRs.  Without a
// bound hipcc hoists every Philox block of a step ahead of its consumers and lands far above that.
#ifndef PTRWM_J2_FENCE_MASK  // fence cadence of the update / squared-jump loop (A/B-timed, tools/ab_bench.sh)
#define PTRWM_J2_FENCE_MASK 7
#endif
#ifndef PTRWM_WAVES_SMALL  // tuning knobs (profiles/r01_bench_variants.txt: 4 beats 3, 5 and 6 at dim 30)
#define PTRWM_WAVES_SMALL 4
#endif
#ifndef PTRWM_WAVES_40
#define PTRWM_WAVES_40 3
#endif
#ifndef PTRWM_WAVES_MID
#define PTRWM_WAVES_MID 2
#endif
// (Width 40 at 4 waves/SIMD spilled ~75 VGPRs; 3 waves/SIMD = 168 VGPRs holds it.  The generic width 32 sat at 125 of the
// 128 VGPRs that 4 waves/SIMD allow and spilled two of them to scratch once the swap event grew its threshold form: it
// is compiled for 3 waves/SIMD as well - dims 31 and 32 run at ~0.97 of the 4-wave rate (profiles/r03_form_sweep_dense.txt,
// A(3)), everything up to the exact width 30 keeps 4.)
constexpr int min_waves_per_simd(int dp) {
  return dp <= 30 ? PTRWM_WAVES_SMALL : (dp <= 44 ? PTRWM_WAVES_40 : (dp <= 64 ? PTRWM_WAVES_MID : 1));
}
// the widths whose step loop fills the 128 VGPRs of four waves per SIMD (in the shipped table: the compiled-in dim 30, and 29
// for HybridRosenbrock): what is loop-invariant there is recomputed rather than kept (ptrwm_step_kernel, the chain word)
constexpr bool step_loop_at_register_cap(int dp, bool stream) { return !stream && min_waves_per_simd(dp) >= 4 && dp > 24; }

// The streaming form (STREAM, below) keeps two slabs of rows per wave in LDS: fewer waves per SIMD fit (and each gets the
// registers of that residency).
constexpr int stream_waves_per_simd(int dp) {  // what two slabs per wave leave room for in 160 KB of LDS per CU, at most 4
  return dp <= 10 ? 4 : (dp <= 20 ? 2 : (dp <= 30 ? 2 : 1));
}

// (the register budget it is compiled for: never that of ONE wave per SIMD - 512 registers invite AGPR copies, the regime
// tools/kernel_stats.py --check keeps every kernel out of - even where LDS admits no second wave)
constexpr int stream_register_waves(int dp) { return stream_waves_per_simd(dp) < 2 ? 2 : stream_waves_per_simd(dp); }

// DP    compile-time width of the per-thread register arrays (>= dim)
// EXACT dim == DP is known at compile time: every per-dimension predicate folds away.  Otherwise
//       dim is a wave-uniform run-time value, re-read (opaquely) every step so the compiler tests
//       `d < dim` with one scalar compare in place instead of hoisting DP booleans into SGPRs.
// FULL  fixture/trace variant: external randoms, per-step trace and accept-flag outputs.
template <class Target, class Proposal, int DP, bool EXACT, bool FULL, bool STREAM = false>
__global__ void __launch_bounds__(kBlockThreads, STREAM ? stream_register_waves(DP) : min_waves_per_simd(DP)) ptrwm_step_kernel(const KArgs a) {
  static_assert(!(STREAM && FULL), "the streaming form is a production kernel: no fixture / trace arguments");
  const int T = a.n_temps;
  const int D0 = EXACT ? DP : a.dim;
  const int cpw = a.chains_per_wave;
  // An exchange group is one wavefront holding cpw = 64 / T whole ladders (T <= 64, "narrow": a workgroup is four
  // independent groups), or ceil(T / 64) wavefronts = the whole workgroup holding one ladder ("wide", cpw = 1).
  // Thread tid of its group is (cw, t) with tid = cw * T + t either way.
  const bool wide = !STREAM && T > 64;  // grid-uniform (the streaming form serves narrow ladders only)
  const int wave = wave_in_block();  // (an SGPR for the whole launch: thread_index_now)
  const int tid = wide ? (int)threadIdx.x : (int)(threadIdx.x & 63);
  // The group this wave (narrow) / workgroup (wide) works on.  Classic form: one group per wave, fixed by the grid.
  // Streaming form: the grid is sized to the device (capi.hip) and every wave walks the groups gw, gw + stride, ... with
  // the NEXT group's state, log-densities and squared-jump sums already on their way from HBM while it computes (below).
  // (the classic form derives it from threadIdx.x, as it always did: its register allocation at the 128-VGPR cap follows
  // every such detail; the streaming form carries it across groups in scalar registers)
  long long group = STREAM ? (long long)blockIdx.x * kWavesPerBlock + wave
                           : (wide ? (long long)blockIdx.x : (long long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
  [[maybe_unused]] const long long n_groups = STREAM ? a.n_chains / cpw : 0;  // (streaming: every group is whole, capi.hip)
  [[maybe_unused]] const long long group_stride = STREAM ? (long long)gridDim.x * kWavesPerBlock : 0;
  if (STREAM ? (group >= n_groups) : (group * cpw >= a.n_chains)) return;  // narrow only (wave-uniform; narrow waves never meet at a workgroup barrier)
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  // ---- streaming form: LDS-DMA double buffering ---------------------------------------------------------------------
  // A group's state is one run of cpw * T * dim floats, 16-byte aligned and a whole number of 16-byte vectors (capi.hip
  // takes the classic kernel otherwise).  Each wave owns TWO slabs: while the group in slab `cur` is being stepped, the
  // next group's run is on its way from HBM straight into the other slab (global_load_lds_dwordx4: 1 KiB per wave
  // instruction, no VGPR destination - the step keeps every register it has in the classic kernel), issued right after
  // the current group's rows have been picked up and waited for (a counted vmcnt placed by the compiler: the builtin is
  // a tracked memory operation) only when the next group's rows are read - the current group's stores, issued later,
  // stay in flight behind it.  HBM reads, the Metropolis step and HBM writes of different groups overlap inside ONE
  // wave, not only across waves.  The next group's log-densities and squared-jump sums travel the same way into two
  // small landing zones: an ordinary load to a VGPR anywhere in this loop would make the compiler drain every
  // outstanding DMA and store (vmcnt(0)) at its first use.
  // LDS of a wave, in floats: [slab 0: 64 DP][slab 1: 64 DP][swap scratch + parked words: 6 x 64][zone 0: 5 x 64][zone 1]
  // (a zone: 64 log-densities, 64 squared-jump sums, 64 acceptance counts)
  typedef float pf_vec4 __attribute__((ext_vector_type(4)));
  constexpr int NV = STREAM ? (DP + 3) / 4 : 1;  // 16-byte vectors per lane and group
  constexpr int kWaveFloats = 64 * lds_floats_per_thread(DP, STREAM);
  constexpr int kExtra0 = 64 * kStreamSlabs * DP;  // streaming form: swap scratch and parked words behind the slab(s)
  constexpr int kZone0 = kExtra0 + 64 * kLdsExtraPerThread, kZoneFloats = 64 * kLdsStreamStatPerThread;
  [[maybe_unused]] int cur = 0;  // which slab / landing zone holds the current group (streaming form; wave-uniform)
  typedef const __attribute__((address_space(1))) void *dma_src;
  typedef __attribute__((address_space(3))) void *dma_dst;
  [[maybe_unused]] auto prefetch = [&](long long g, int slab) {
    const int tid = thread_index_now(wave) & 63;  // (re-derived: no per-lane address parts kept, or spilled, across the step)
    const kargs_ptr ap = late_args();             // (and the array pointers are loaded here, not held in SGPRs across it)
    const int n_vec = (cpw * T * D0) >> 2;
    const int n_live = cpw * T;
    float *const wbase = s_dyn + wave * kWaveFloats;
    // log-densities: one dword per lane (idle lanes re-read replica 0 of the group)
    const long long r0 = g * n_live;
    float *const zone = wbase + kZone0 + slab * kZoneFloats;
    __builtin_amdgcn_global_load_lds((dma_src)(ap->logp + r0 + (tid < n_live ? tid : 0)), (dma_dst)(uintptr_t)zone, 4, 0, 0);
    // squared-jump sums: 16 bytes = two doubles per lane (n_live is even, capi.hip)
    if (ap->sq_jump != nullptr && 2 * tid < n_live)
      __builtin_amdgcn_global_load_lds((dma_src)(ap->sq_jump + r0 + 2 * tid), (dma_dst)(uintptr_t)(zone + 64), 16, 0, 0);
    // acceptance counts likewise: read ahead and stored back as old + delta.  (The classic kernel adds them by no-return
    // atomics to keep a load off the end of a wave's life; here the old value is in LDS before the step begins, and plain
    // 8-byte stores stream at several times the rate the memory-side atomics do.)
    if (ap->n_accept != nullptr && 2 * tid < n_live)
      __builtin_amdgcn_global_load_lds((dma_src)(ap->n_accept + r0 + 2 * tid), (dma_dst)(uintptr_t)(zone + 192), 16, 0, 0);
    const pf_vec4 *__restrict__ gv = reinterpret_cast<const pf_vec4 *>(ap->state + g * cpw * T * (long long)D0);
    float *const dst = wbase + slab * (64 * DP);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int v = tid + 64 * k;
      if (v < n_vec) __builtin_amdgcn_global_load_lds((dma_src)(gv + v), (dma_dst)(uintptr_t)(dst + 256 * k), 16, 0, 0);
    }
  };
  // (streaming form: what depends on the thread's temperature only is loaded ONCE, before the first DMA is issued - any
  // ordinary global load behind an LDS-DMA makes the compiler drain the DMA at the load's first use, which would put the
  // next group's HBM latency back in front of the current group's step)
  [[maybe_unused]] float beta_t_s = 0.0f, tscale_s = 0.0f;
  [[maybe_unused]] float db_s = 1.0f;  // b_t - b_{t+1} of this thread's pair (swap_pair_plain)
  // Outgoing results of the group just finished wait in registers until the NEXT group's rows have been picked up, and
  // are stored then (flush_pending): at the top of an iteration the only vector-memory operations that can still be
  // outstanding are the DMA of the group about to be stepped and the stores issued a whole step earlier - so the wait for
  // the DMA is a plain vmcnt(0) that never waits for a store just issued.
  [[maybe_unused]] pf_vec4 pend_o[NV];
  [[maybe_unused]] float pend_lp = 0.0f;
  [[maybe_unused]] unsigned pend_n_swap = 0u;
  [[maybe_unused]] long long pend_acc = 0;
  [[maybe_unused]] bool pend_acc_on = false;
  [[maybe_unused]] double pend_sq = 0.0;
  [[maybe_unused]] long long pend_ord = 0, pend_group = 0;
  [[maybe_unused]] bool pend_sq_on = false, pend_ord_on = false, have_pending = false;
  [[maybe_unused]] auto flush_pending = [&]() {
    const kargs_ptr ap = late_args();
    const int tid = thread_index_now(wave) & 63;
    const int n_live = cpw * T;
    const int n_vec = (n_live * D0) >> 2;
    pf_vec4 *__restrict__ gv = reinterpret_cast<pf_vec4 *>(ap->state + pend_group * n_live * (long long)D0);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int v = tid + 64 * k;
      if (v < n_vec) gv[v] = pend_o[k];
    }
    if (tid < n_live) {
      const long long r = pend_group * n_live + tid;
      ap->logp[r] = pend_lp;
      // statistics: only where the launch has something to add (kernel epilogue below)
      if (pend_acc_on) ap->n_accept[r] = pend_acc;
      if (pend_sq_on) ap->sq_jump[r] = pend_sq;
      if (ap->swap_accept != nullptr && pend_n_swap != 0u) count_add(&ap->swap_accept[r], (long long)pend_n_swap);
      // (the same maximum as the classic kernel's compare-and-store, as a no-return atomic at the L2: no load)
      if (pend_ord_on) (void)__hip_atomic_fetch_max(&ap->last_swap_ordinal[r], pend_ord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  if constexpr (STREAM) {
    const int cw_s = tid / T;
    const int t_s = cw_s < cpw ? tid - cw_s * T : 0;
    beta_t_s = a.beta[t_s];
    tscale_s = a.temp_scale[t_s];
    db_s = sub_rn(beta_t_s, a.beta[t_s < T - 1 ? t_s + 1 : t_s]);
    prefetch(group, 0);
  }
  do {
  const long long chain0 = group * cpw;
  const int cw_raw = tid / T;
  const int t_raw = tid - cw_raw * T;
  const bool live = (cw_raw < cpw) && (chain0 + cw_raw < a.n_chains);
  // idle threads shadow replica (chain0, 0): they compute but never store and are never an exchange source
  const int cw = live ? cw_raw : 0;
  const int t = live ? t_raw : 0;
  const long long chain = chain0 + cw;
  const long long rep = chain0 * T + (live ? tid : 0);  // cw * T + t == tid for a live thread

  // ---- LDS (dynamic: group threads * (DP + kLdsExtraPerThread) floats per group, sized by the launch: step_kernel_lds_bytes) --
  // s_stage: one row of up to DP floats per thread; the group packs its live replicas' rows back to back (row
  // stride = dim) for the coalesced state load / store and exchanges rows through it in a swap.
  // s_l / s_u / landed (behind the rows): per-thread log-density, swap uniform and swap outcome of a swap sweep.
  // (streaming form: the slab of the current group; the swap scratch and the parked words sit behind BOTH slabs)
  float *const s_stage = STREAM ? s_dyn + wave * kWaveFloats + cur * (64 * DP)
                                : s_dyn + (wide ? 0 : (int)(threadIdx.x >> 6) * (64 * (DP + kLdsExtraPerThread)));
  [[maybe_unused]] float *const s_extra = s_dyn + wave * kWaveFloats + kExtra0;  // streaming form only
  // group-wide ordering of LDS accesses: the group is one wave (narrow) or the workgroup (wide)
  auto sync_group = [&]() {
    if (wide) {
      __syncthreads();
    } else {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  };

  // ---- state load: coalesced HBM reads staged through LDS ------------------------------------------------
  // The group's live replicas are one contiguous run of n_live * dim floats in `state`.  The group copies that
  // run with fully coalesced dword loads (thread i takes elements i, i+n, ...) into its slab and each thread then
  // reads its own row (stride dim words: at most a 2-way bank conflict for even dim).  A direct per-thread row
  // read would touch 64 different cache lines per instruction.
  float x[DP], y[DP];
  {
    const long long live_chains = (a.n_chains - chain0 < cpw) ? (a.n_chains - chain0) : cpw;
    const int stage_total = (int)live_chains * T * D0;  // floats of this group's run
    float *__restrict__ gs = a.state + chain0 * T * (long long)D0;
    const int nthr = wide ? ((T + 63) & ~63) : 64;  // threads of the group = of the workgroup
    int row_head = 0;
    if constexpr (!STREAM) {
      stage_copy<true>(s_stage, gs, stage_total, tid, nthr);
      row_head = stage_head(gs);
      if (wide) reinterpret_cast<int *>(s_dyn)[nthr * (DP + kLdsExtraPerThread)] = 0;  // no objection yet (kWideVoteBytes)
    }
    // (streaming form: the run is landing in slab `cur` by LDS-DMA.  The compiler does not order LDS reads behind it:
    // this wait does - every DMA of this wave, and nothing younger than a whole step, see flush_pending)
    if constexpr (STREAM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    sync_group();
    const float *row = s_stage + row_head + (live ? tid : 0) * D0;  // idle threads shadow row 0 = replica (chain0, 0)
#pragma unroll
    for (int d = 0; d < DP; ++d) x[d] = (d < D0) ? row[d] : 0.0f;
  }
  float lp;
  [[maybe_unused]] double sq_old = 0.0;
  [[maybe_unused]] long long acc_old = 0;
  if constexpr (STREAM) {
    {
      const float *zone = s_dyn + wave * kWaveFloats + kZone0 + cur * kZoneFloats;
      lp = zone[tid];
      if (a.sq_jump != nullptr) sq_old = reinterpret_cast<const double *>(zone + 64)[tid < cpw * T ? tid : 0];
      if (a.n_accept != nullptr) acc_old = reinterpret_cast<const long long *>(zone + 192)[tid < cpw * T ? tid : 0];
    }
    // the previous group's results leave now, and behind them the next group's DMA is issued: the other slab is free,
    // the previous group's outgoing rows were read back into pend_o before this group began
    if (have_pending) flush_pending();
    if (group + group_stride < n_groups) prefetch(group + group_stride, cur ^ 1);
  } else {
    lp = a.logp[rep];
  }
  const float beta_t = STREAM ? beta_t_s : a.beta[t];
  const float tscale = STREAM ? tscale_s : a.temp_scale[t];
  // may the proposal's own squared increment stand for |y - x|^2 for this replica?  (proposals.h kJumpTrust)
  bool jump_trusted = false;
  if constexpr (Proposal::kKnowsJump) {
    float xmax = 0.0f;
#pragma unroll
    for (int d = 0; d < DP; ++d) xmax = __builtin_fmaxf(xmax, __builtin_fabsf(x[d]));  // (slots >= dim hold 0)
    jump_trusted = xmax <= kJumpTrust * Proposal::increment_scale(tscale, a.pp);  // (NaN: false)
  }

  const unsigned long long gchain = (unsigned long long)(a.chain_offset + chain);
  RngCtx rc;
  rc.c2 = (uint32_t)gchain;
  rc.k0 = a.k0;
  rc.k1 = a.k1;
  const uint32_t c3_base = (uint32_t)t | ((uint32_t)(gchain >> 32) << 12);

  unsigned n_acc = 0;
  {
    // parked in LDS (see step_kernel_lds_bytes): the Philox word with the temperature index, the number of accepted
    // swaps of pair (t, t+1) and the index within this launch of the last swap event in which it accepted
    const int gt = wide ? ((T + 63) & ~63) : 64;
    int *const park = reinterpret_cast<int *>(STREAM ? s_extra + 3 * gt : s_stage + gt * (DP + 3)) + tid;
    park[0] = (int)c3_base;
    park[gt] = 0;
    park[2 * gt] = -1;
    // ... and this launch's sum of squared jumps, a double behind them: added to by a no-return ds_add_f64 per counted
    // step - the same IEEE additions in the same order as a register would see, without the two VGPRs of a double that
    // lives across the whole step loop (the headline kernel sits at the 128-VGPR cap of four waves per SIMD)
    reinterpret_cast<double *>(park - tid + 3 * gt)[tid] = 0.0;
  }
#ifdef PTRWM_NO_SQ_LDS
  double sq_reg = 0.0;
#endif

  const bool ext = FULL && a.full.ext_prop != nullptr;
  const bool trace_on =
      FULL && live && a.full.trace != nullptr && (chain < a.full.trace_chains) && (t < a.full.trace_temps);
  int to_swap = a.steps_to_swap;
  int to_trace = FULL ? a.full.steps_to_trace : 0;
  int trace_rows = 0;    // rows of the trace written by this launch
  int swap_in_call = 0;  // swap events already done in this launch
  const int ev_par0 = (int)(a.first_swap_event & 1);
  unsigned long long s = (unsigned long long)a.step0;  // 0-based global step index

  for (int i = 0; i < a.n_steps; ++i, ++s) {
    const bool count_on = i >= a.burn_left;
    --to_swap;
    const bool multiple = (to_swap == 0);
    if (multiple) to_swap = a.swap_every;
    const bool swap_due = multiple && count_on && (T > 1);

    rc.c0hi = (uint32_t)(s >> 32) << 16;
    rc.c1 = (uint32_t)s;
    rc.c3 = c3_base | (kStreamMH << 8);
    // (at the register cap the chain word is made opaque per step: its product with the Philox multiplier is otherwise
    // hoisted out of the step loop as a 64-bit pair, which THERE is spilled and fetched back from scratch at the top of
    // every step - one v_mad_u64_u32 per step instead.  Only there: every kernel with registers to spare keeps the hoisted
    // product and runs 2-5 % faster for it - dims 24 / 48 / 50 in profiles/r04_scratch_ab.txt, table 6)
#ifndef PTRWM_NO_C2_OPAQUE
    if constexpr (step_loop_at_register_cap(DP, STREAM)) asm volatile("" : "+v"(rc.c2));
#endif

    long long srep = 0;
    const float *ext_raw = nullptr;
    float ext_u = 0.0f;
    if constexpr (FULL) {
      srep = ((long long)i * a.n_chains + chain) * T + t;
      if (ext) {
        ext_raw = a.full.ext_prop + srep * a.full.n_raw_ext;
        ext_u = a.full.ext_u[srep];
      }
    }

    constexpr int W = canon_width(DP);
    float jump = 0.0f;  // the squared length of the increment, if the proposal knows it (proposals.h)
    int jump_kind;      // (a constant in production kernels)
    const float u_acc = Proposal::propose(y, x, fresh_dim<EXACT>(D0), tscale, a.pp, rc, ext_raw, ext_u, jump, jump_kind);
    const float lp_new = Target::template logp<false>(y, fresh_dim<EXACT>(D0), a.tp);
    const int D = fresh_dim<EXACT>(D0);

    // ultra_fused_mcmc_step_basic / ultra_fused_parallel_mcmc_step:
    //   r = beta (l' - l);  accept = (r > 0) | (u < exp r)
    const bool acc = mh_accept(beta_t, lp_new, lp, u_acc);
    const float lp_mh = acc ? lp_new : lp;
    if constexpr (FULL) {
      if (a.full.accept_flags != nullptr && live) a.full.accept_flags[srep] = acc ? 1 : 0;
    }

    float j2;
    if (!swap_due && jump_kind != kJumpNone) {
      // the proposal knows the length of its own increment: the move itself is one select per dimension.  Replicas whose
      // state is too large for that to equal |y - x|^2 (jump_trusted, decided at the start of the launch: none, normally -
      // the branch is skipped) take it from the states
      float from_states = 0.0f;
      if (!jump_trusted) {
        PTRWM_COLD_PATH();
        float j2p[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        PTRWM_DIM_LOOP(d, DP, D, {
          const float dl = sub_rn(y[d], x[d]);
          j2p[d / W] = fmaf(dl, dl, j2p[d / W]);
        })
        from_states = tree4_add(j2p);
      }
      PTRWM_DIM_LOOP(d, DP, D, {
        x[d] = acc ? y[d] : x[d];
        if ((d & PTRWM_J2_FENCE_MASK) == PTRWM_J2_FENCE_MASK) sched_fence_soft();
      })
      j2 = acc ? (jump_trusted ? jump : from_states) : 0.0f;
      lp = lp_mh;
    } else if (!swap_due) {
      float j2p[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // squared jump in the canonical four-range order (philox.h)
#ifdef PTRWM_J2_SEPARATE
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        if (d < D) {
          const float dl = sub_rn(y[d], x[d]);
          j2p[d / W] = fmaf(dl, dl, j2p[d / W]);
        }
        if ((d & PTRWM_J2_FENCE_MASK) == PTRWM_J2_FENCE_MASK) sched_fence_soft();
      }
#pragma unroll
      for (int d = 0; d < DP; ++d)
        if (d < D) x[d] = acc ? y[d] : x[d];
#else
      PTRWM_DIM_LOOP(d, DP, D, {
        const float dl = sub_rn(y[d], x[d]);
        j2p[d / W] = fmaf(dl, dl, j2p[d / W]);
        x[d] = acc ? y[d] : x[d];
        if ((d & PTRWM_J2_FENCE_MASK) == PTRWM_J2_FENCE_MASK) sched_fence_soft();
      })
#endif
      j2 = tree4_add(j2p);
      if (!acc) j2 = 0.0f;
      lp = lp_mh;
    } else {
      // ---- temperature swaps on the post-MH log-densities (pt_rwm_gpu_optimized.py:594-633) ----
      // The exchange indices are rebuilt here from an opaque copy of threadIdx.x, so that none of them occupies a
      // register (or a scratch slot) across the MH part of the step.
      const int tid_s = thread_index_now(wave);
      const int slot = wide ? tid_s : (tid_s & 63);  // this thread's slot in s_l / s_u and its row in s_stage
      const int group_threads = wide ? ((T + 63) & ~63) : 64;
      float *const rows = STREAM ? s_dyn + wave * kWaveFloats + cur * (64 * DP)
                                 : s_dyn + (wide ? 0 : (tid_s >> 6) * (64 * (DP + kLdsExtraPerThread)));
      float *const s_l = STREAM ? s_dyn + wave * kWaveFloats + kExtra0 : rows + group_threads * DP;
      float *const s_u = s_l + group_threads;
      int *const park = reinterpret_cast<int *>(s_l + 3 * group_threads) + slot;
      const uint32_t c3_s = (uint32_t)park[0];
      const int t = (int)(c3_s & 0xffu);  // the temperature index, as the Philox counter holds it
      const int base = live ? slot - t : 0;            // slot of temperature 0 of this thread's ladder
      // src = slot whose post-MH vector ends up at this thread's temperature
      int src = slot;
      float my_l = lp_mh;
      bool pair_acc = false;  // did pair (t, t+1) accept (recorded on the thread of temperature t)
      float us;
      if (ext) {
        us = (t < T - 1) ? a.full.ext_swap_u[((long long)swap_in_call * a.n_chains + chain) * (T - 1) + t] : 2.0f;
      } else {
        const u32x4 r = philox4x32_10(rc.c0hi, rc.c1, rc.c2, c3_s | (kStreamSwap << 8), rc.k0, rc.k1);
        us = u01(r.x);
      }
      // publish this thread's log-density and swap uniform; the sweep reads them back with broadcast ds_reads
      s_l[slot] = my_l;
      s_u[slot] = us;
      // may this ladder's sequential sweep take the threshold form in THIS event?  (swap_decide: a verdict of the ladder)
      const bool pair_plain = swap_pair_plain(T, t, STREAM ? db_s : sub_rn(beta_t, a.beta[t < T - 1 ? t + 1 : t]), my_l, us);
      // (narrow groups: a ballot over the ladder's lanes.  Wide: an objection is the event's stamp in the ladder's word
      // behind the parked words, written together with the published values - the one barrier orders both - and read
      // for the last time before this event's row-exchange barrier, after which the next event may stamp it again)
      bool swap_plain;
      if (wide) {
        int *const objection = reinterpret_cast<int *>(s_l + kLdsExtraPerThread * group_threads);
        if (!pair_plain) *objection = swap_in_call + 1;
        sync_group();
        swap_plain = *objection != swap_in_call + 1;
      } else {
        swap_plain = ladder_votes_plain(pair_plain, base, T);
        sync_group();
      }
      swap_decide(T, t, base, slot, a.swap_mode, a.swap_order, (ev_par0 + swap_in_call) & 1, a.beta, beta_t, us, s_l, s_u,
                  reinterpret_cast<int *>(s_u + group_threads), my_l, src, pair_acc, sync_group, swap_plain);
      if (pair_acc) {
        park[group_threads] += 1;
        park[2 * group_threads] = swap_in_call;
      }
      // commit MH move and swap in one pass: every thread publishes its post-MH vector as its slab row, then
      // fetches the row of slot `src` (rows exchanged through LDS: 2 LDS ops per dimension, no HBM)
      {
        float j2p[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // (a swap step's jump is taken from the rows, in the canonical order)
        float *my_row = rows + slot * D;
        PTRWM_DIM_LOOP(d, DP, D, { my_row[d] = acc ? y[d] : x[d]; })
        sync_group();
        const int Dr = fresh_dim<EXACT>(D0);  // generic widths: fresh d < dim compares instead of 2*DP live masks
        const float *src_row = rows + src * Dr;
        PTRWM_DIM_LOOP(d, DP, Dr, {
          const float w = src_row[d];
          const float dl = sub_rn(w, x[d]);
          j2p[d / W] = fmaf(dl, dl, j2p[d / W]);
          x[d] = w;
          if ((d & 7) == 7) sched_fence_soft();
        })
        j2 = tree4_add(j2p);
      }
      lp = my_l;
      ++swap_in_call;
    }

    if (count_on) {
      n_acc += acc ? 1u : 0u;
#ifdef PTRWM_NO_SQ_LDS
      sq_reg += (double)j2;
#else
      const int tid_q = thread_index_now(wave);  // (the slot's address is rebuilt here, not carried across the step)
      const int gt_q = wide ? ((T + 63) & ~63) : 64;
      double *const sq_slot = reinterpret_cast<double *>(
                                  STREAM ? s_dyn + wave * kWaveFloats + kExtra0 + 6 * 64
                                         : s_dyn + (wide ? 0 : wave * (64 * (DP + kLdsExtraPerThread))) + gt_q * (DP + 6)) +
                              (wide ? tid_q : (tid_q & 63));
      (void)__hip_atomic_fetch_add(sq_slot, (double)j2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#endif
    }
    if constexpr (FULL) {
      bool trace_now = false;
      if (a.full.trace != nullptr) {  // wave-uniform thinning countdown
        --to_trace;
        trace_now = (to_trace == 0);
        if (trace_now) to_trace = a.full.trace_every;
      }
      if (trace_now && trace_on) {
        const long long row = ((a.full.trace_row0 + trace_rows) * a.full.trace_chains + chain) * a.full.trace_temps + t;
        float *__restrict__ tr = a.full.trace + row * D;
        PTRWM_DIM_LOOP(d, DP, D, { tr[d] = x[d]; })
        if (a.full.trace_logp != nullptr) a.full.trace_logp[row] = lp;
      }
      trace_rows += trace_now ? 1 : 0;
    }
  }

  // ---- state store: rows -> LDS slab -> coalesced HBM writes -----------------------------------------------
  const kargs_ptr ae = late_args();  // the epilogue's arguments are loaded here, not kept in SGPRs across the step loop
  long long c0_out;
  {
    // everything is recomputed from opaque copies so that nothing of the prologue stays live across the step loop
    const int T2 = fresh_dim<false>(T), D2 = EXACT ? DP : fresh_dim<false>(D0), cpw2 = fresh_dim<false>(cpw);
    const bool wide2 = T2 > 64;
    const int tx2 = thread_index_now(wave);
    const int tid2 = wide2 ? tx2 : (tx2 & 63);
    float *const rows2 = STREAM ? s_dyn + wave * kWaveFloats + cur * (64 * DP)
                                : s_dyn + (wide2 ? 0 : wave * (64 * (DP + kLdsExtraPerThread)));
    long long c0;
    if constexpr (STREAM) {
      c0 = group * cpw2;
    } else {
      const long long bid = (long long)fresh_dim<false>((int)blockIdx.x);  // re-read here, not carried in a VGPR
      c0 = (wide2 ? bid : bid * kWavesPerBlock + wave) * cpw2;
    }
    const long long n_chains2 = ae->n_chains;
    const long long live_chains = (n_chains2 - c0 < cpw2) ? (n_chains2 - c0) : cpw2;
    const int stage_total = (int)live_chains * T2 * D2;
    const long long stage_g0 = c0 * T2 * (long long)D2;
    c0_out = c0;
    float *__restrict__ gs = ae->state + stage_g0;
    sync_group();  // the last swap's row reads are done before the rows are overwritten
    if (live) {
      float *row = rows2 + stage_head(gs) + tid2 * D2;
#pragma unroll
      for (int d = 0; d < DP; ++d)
        if (d < D2) row[d] = x[d];
    }
    sync_group();
    const int nthr = wide2 ? ((T2 + 63) & ~63) : 64;
    if constexpr (STREAM) {
      // whole aligned vectors only (capi.hip): slab -> registers now, registers -> HBM in flush_pending
      const pf_vec4 *__restrict__ lv = reinterpret_cast<const pf_vec4 *>(rows2);
      const int n_vec = stage_total >> 2;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int v = tid2 + 64 * k;
        if (v < n_vec) pend_o[k] = lv[v];
      }
    } else {
      stage_copy<false>(rows2, gs, stage_total, tid2, nthr);
    }
  }
  if (live) {
    const int tid_o = thread_index_now(wave);
    const int T_o = fresh_dim<false>(T);
    const long long rep = c0_out * T_o + (T_o > 64 ? tid_o : (tid_o & 63));  // live: replica index in group == tid
    const int gt_o = T_o > 64 ? ((T_o + 63) & ~63) : 64;
    const int *const park = reinterpret_cast<const int *>(STREAM ? s_dyn + wave * kWaveFloats + kExtra0 + 3 * 64
                                                                 : s_dyn + (T_o > 64 ? 0 : (tid_o >> 6) * (64 * (DP + kLdsExtraPerThread)))
                                                                       + gt_o * (DP + 3)) + (T_o > 64 ? tid_o : (tid_o & 63));
    const int t = park[0] & 0xff;
    const unsigned n_swap_acc = (unsigned)park[gt_o];
    const int last_event = park[2 * gt_o];
    const int tid_g = T_o > 64 ? tid_o : (tid_o & 63);
#ifdef PTRWM_NO_SQ_LDS
    const double sq = sq_reg;
#else
    const double sq = reinterpret_cast<const double *>(park - tid_g + 3 * gt_o)[tid_g];
#endif
    if constexpr (STREAM) {
      // handed to flush_pending (the old squared-jump sum came in with the prefetch: the same double addition as the classic
      // kernel's read-modify-write, without a load)
      pend_lp = lp;
      pend_acc_on = ae->n_accept != nullptr && n_acc != 0u;
      pend_acc = acc_old + (long long)n_acc;
      pend_n_swap = n_swap_acc;
      pend_sq_on = ae->sq_jump != nullptr && sq != 0.0;
      pend_sq = sq_old + sq;
      pend_ord_on = ae->last_swap_ordinal != nullptr && last_event >= 0;
      const long long ev = ae->first_swap_event + last_event;
      pend_ord = (ae->swap_order == PTRWM_ORDER_SEQUENTIAL) ? ev * (T_o - 1) + t + 1 : ev + 1;
    } else {
    ae->logp[rep] = lp;
    // statistics: read-modify-write only where this launch has something to add (a launch without a swap event - nine in
    // ten at one step per launch - then leaves the swap counters' cache lines alone)
    if (ae->n_accept != nullptr && n_acc != 0u) count_add(&ae->n_accept[rep], (long long)n_acc);
    if (ae->sq_jump != nullptr && sq != 0.0) ae->sq_jump[rep] += sq;
    if (ae->swap_accept != nullptr && n_swap_acc != 0u) count_add(&ae->swap_accept[rep], (long long)n_swap_acc);
    if (ae->last_swap_ordinal != nullptr && last_event >= 0) {
      // 1-based attempt ordinal counted from the start of the run.  Sequential order: T-1 attempts
      // per event; even/odd events have a varying pair count, so the event number is recorded.
      const long long ev = ae->first_swap_event + last_event;
      const long long ord = (ae->swap_order == PTRWM_ORDER_SEQUENTIAL) ? ev * (T_o - 1) + t + 1 : ev + 1;
      if (ord > ae->last_swap_ordinal[rep]) ae->last_swap_ordinal[rep] = ord;
    }
    }
  }
  if constexpr (!STREAM) break;
  pend_group = group;
  have_pending = true;
  group += group_stride;
  cur ^= 1;
  } while (group < n_groups);
  if constexpr (STREAM) flush_pending();
}

// ---- standalone log-density kernel (unit parity of the targets; initial log-density) ----
template <class Target, int DP>
__global__ void __launch_bounds__(kBlockThreads) ptrwm_logdensity_kernel(const float *__restrict__ x,
                                                                          float *__restrict__ out,
                                                                          long long n, int D, TParams tp) {
  const long long i = (long long)blockIdx.x * kBlockThreads + threadIdx.x;
  if (i >= n) return;
  float y[DP];
  const float *__restrict__ xp = x + i * D;
#pragma unroll
  for (int d = 0; d < DP; ++d) y[d] = (d < D) ? xp[d] : 0.0f;
  out[i] = Target::template logp<true>(y, D, tp);
}

// The two stand-alone proposal kernels below hold x[DP] and y[DP] in registers: above width 64 they are compiled for two
// waves per SIMD, i.e. a budget of 256 registers, so that what does not fit goes to scratch (speed is irrelevant here)
// and not into AGPRs - the build gate (tools/kernel_stats.py --check) admits no kernel with AGPRs or > 256 VGPRs.
constexpr int standalone_min_waves(int dp) { return dp > 64 ? 2 : 1; }

// ---- standalone proposal kernel (unit parity of the three samplers) ----
template <class Proposal, int DP>
__global__ void __launch_bounds__(kBlockThreads, standalone_min_waves(DP)) ptrwm_propose_kernel(
    float *__restrict__ out, long long n, int D, int T, const float *__restrict__ temp_scale, PParams pp,
    const float *__restrict__ ext_raw, int n_raw_ext, unsigned k0, unsigned k1) {
  const long long i = (long long)blockIdx.x * kBlockThreads + threadIdx.x;
  if (i >= n * T) return;
  const long long row = i / T;
  const int t = (int)(i - row * T);
  float x[DP], y[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) x[d] = 0.0f;
  RngCtx rc;
  rc.c0hi = (uint32_t)((unsigned long long)row >> 32) << 16;
  rc.c1 = (uint32_t)row;  // "step" = row, chain 0
  rc.c2 = 0;
  rc.c3 = (uint32_t)t | (kStreamMH << 8);
  rc.k0 = k0;
  rc.k1 = k1;
  const float *er = ext_raw != nullptr ? ext_raw + i * n_raw_ext : nullptr;
  float jump = 0.0f;
  int jump_kind;
  Proposal::propose(y, x, D, temp_scale[t], pp, rc, er, 0.0f, jump, jump_kind);
  float *__restrict__ op = out + i * D;
#pragma unroll
  for (int d = 0; d < DP; ++d)
    if (d < D) op[d] = y[d];
}

// ---- split step, first half: proposals for one step written to HBM (targets evaluated by the caller) ----
// Same Philox words and the same arithmetic as the fused kernel's Proposal::propose call, so a split step driven
// with the library's own log-density reproduces ptrwm_run bit for bit.
// One wavefront per 64 replicas (64-thread workgroups): the tile's run of 64 x dim floats goes through a slab of LDS in both
// directions (stage_copy: coalesced 16-byte transfers whatever the run's alignment), each lane working on its own row -
// round 3's version read and wrote its row straight from HBM, 64 cache lines per instruction: 0.36 ms per step at
// 65 536 x 32 x dim 30, where moving the bytes takes 0.1.
constexpr unsigned split_tile_lds_bytes(int dim) { return (unsigned)(64 * dim + 4) * 4u; }

template <class Proposal, int DP>
__global__ void __launch_bounds__(64, standalone_min_waves(DP)) ptrwm_split_propose_kernel(
    const float *__restrict__ state, float *__restrict__ proposals, float *__restrict__ accept_u, long long n_chains,
    long long chain_offset, unsigned long long step, int D, int T, const float *__restrict__ temp_scale, PParams pp,
    const float *__restrict__ ext_raw, const float *__restrict__ ext_u, int n_raw_ext, unsigned k0, unsigned k1,
    const long long *__restrict__ device_step) {
  extern __shared__ __attribute__((aligned(16))) float s_tile[];
  const int lane = (int)threadIdx.x;
  const long long n_reps = n_chains * T;
  const long long first = (long long)blockIdx.x * 64;
  if (first >= n_reps) return;
  if (device_step != nullptr) step += (unsigned long long)*device_step;  // (include/ptrwm.h: the counter plus this call's offset)
  const int n_rows = (n_reps - first < 64) ? (int)(n_reps - first) : 64;
  const bool live = lane < n_rows;
  const long long i = first + (live ? lane : 0);
  const long long chain = i / T;
  const int t = (int)(i - chain * T);
  float x[DP], y[DP];
  float *__restrict__ gx = const_cast<float *>(state) + first * D;
  stage_copy<true>(s_tile, gx, n_rows * D, lane, 64);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  {
    const float *row = s_tile + stage_head(gx) + (live ? lane : 0) * D;
#pragma unroll
    for (int d = 0; d < DP; ++d) x[d] = (d < D) ? row[d] : 0.0f;
  }
  const unsigned long long gchain = (unsigned long long)(chain_offset + chain);
  RngCtx rc;
  rc.c0hi = (uint32_t)(step >> 32) << 16;
  rc.c1 = (uint32_t)step;
  rc.c2 = (uint32_t)gchain;
  rc.c3 = (uint32_t)t | ((uint32_t)(gchain >> 32) << 12) | (kStreamMH << 8);
  rc.k0 = k0;
  rc.k1 = k1;
  const float *er = ext_raw != nullptr ? ext_raw + i * n_raw_ext : nullptr;
  float jump = 0.0f;
  int jump_kind;
  const float u = Proposal::propose(y, x, D, temp_scale[t], pp, rc, er, ext_u != nullptr ? ext_u[i] : 0.0f, jump, jump_kind);
  float *__restrict__ gy = proposals + first * D;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();  // every lane has its row in registers: the slab takes the proposals
  if (live) {
    float *row = s_tile + stage_head(gy) + lane * D;
#pragma unroll
    for (int d = 0; d < DP; ++d)
      if (d < D) row[d] = y[d];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  stage_copy<false>(s_tile, gy, n_rows * D, lane, 64);
  if (!live) return;
  accept_u[i] = u;
  // second plane of the scratch array: the proposal's own squared jump, exactly as the fused kernel counts it, or -1 when
  // it is to be taken from the states (external randoms, Laplace): split_accept_kernel then reproduces ptrwm_run's sums
  float xmax = 0.0f;
#pragma unroll
  for (int d = 0; d < DP; ++d) xmax = __builtin_fmaxf(xmax, __builtin_fabsf(x[d]));
  const bool trusted = Proposal::kKnowsJump && xmax <= kJumpTrust * Proposal::increment_scale(temp_scale[t], pp);  // (proposals.h)
  accept_u[n_reps + i] = (jump_kind == kJumpTotal && trusted) ? jump : -1.0f;
}

}  // namespace ptrwm
