// C ABI of the PT-RWM engine (include/ptrwm.h): argument validation, variant
// dispatch and launch.  No allocation, no synchronisation, no retained pointers.
#include "../../include/ptrwm.h"
#include "variants.h"

namespace ptrwm {

struct VariantPair {
  const TargetVariants *narrow, *wide;  // each holds null entries for the other group's widths (variants.h)
  RunLaunchFn run(int proposal, int dpi) const {
    const RunLaunchFn f = narrow->run[proposal][dpi];
    return f != nullptr ? f : wide->run[proposal][dpi];
  }
  LogpLaunchFn logp(int dpi) const {
    const LogpLaunchFn f = narrow->logp[dpi];
    return f != nullptr ? f : wide->logp[dpi];
  }
};

// ---- which form of the step kernel runs (quad.h) ------------------------------------------------------------------
// The two forms are bit-identical on the same Philox stream, so this is a speed decision only; ptrwm_set_kernel_form()
// pins it for tests and tuning.  AUTO compares a model of both forms' throughput at the launch's size
// (tools/form_fit.py, fitted to profiles/r04_form_sweep_dense.txt - both forms timed over waves per SIMD, ladder lengths
// and dims on one MI355X - and checked on a held-out sweep, profiles/r04_form_sweep_heldout.txt).  With w = wavefronts
// per SIMD the one-thread-per-replica form would launch:
//   thread form   rate = A(k) w / k, k = ceil(w): a launch lasts as long as its fullest SIMDs, so at w = 1.25 the form
//                 runs at 0.625 of its two-waves rate, not at its one-wave rate (the dips of profiles/r02_form_sweep.txt)
//   lane-split    four times the waves with a quarter of the work each: the same saw-tooth on a four times finer scale,
//                 rate = Q(kq / 4) 4 w / kq, kq = ceil(4 w), Q read off the measured curve
//   dim < 16      never lane-split (a lane would own <= 3 dims: profiles/r02_single_ladder.txt)
//   dim > 64      always: it is the only form there (the one-thread-per-replica kernel needed 340-420 VGPRs and sat in
//                 the register regime in which hipcc miscompiled it twice; see variants.h)
//   w > 4         never (the thread form is saturated; the lane-split form repeats the per-replica scalar work)
#include "form_table.inc"

static void form_lerp(const float *a, const float *b, float t, float *out, int n) {
  for (int i = 0; i < n; ++i) out[i] = a[i] + (b[i] - a[i]) * t;
}

// interpolated model parameters at (dim, n_temps): dims within the lane-width class of `dim`, ladder lengths in log2
static void form_params(int dim, int n_temps, float *thread4, float *quad) {
  // A dim with kernels of its own (dim compiled in) has its own row; every other dim runs the run-time-dim kernels: last
  // generic grid dim <= dim and first generic grid dim >= dim within dim's lane-width class
  int d0 = -1, d1 = -1;
  for (int i = 0; i < kFormND; ++i)
    if (kFormDimExact[i] && kFormDims[i] == dim) d0 = d1 = i;
  const bool own_row = d0 >= 0;
  for (int i = 0; i < kFormND && !own_row; ++i) {
    if (kFormDimExact[i] || (kFormDims[i] <= 32) != (dim <= 32)) continue;
    if (kFormDims[i] <= dim) d0 = i;
    if (kFormDims[i] >= dim && d1 < 0) d1 = i;
  }
  if (d0 < 0) d0 = d1;  // below the class's first grid dim: clamp
  if (d1 < 0) d1 = d0;  // above its last
  const float td = d0 == d1 ? 0.0f : (float)(dim - kFormDims[d0]) / (float)(kFormDims[d1] - kFormDims[d0]);
  int t0 = 0, t1 = kFormNT - 1;
  for (int i = 0; i < kFormNT; ++i) {
    if (kFormTemps[i] <= n_temps) t0 = i;
    if (kFormTemps[i] >= n_temps) { t1 = i; break; }
  }
  if (t1 < t0) t1 = t0;
  const float tt = t0 == t1 ? 0.0f
                            : (float)((__builtin_log2((double)n_temps) - __builtin_log2((double)kFormTemps[t0])) /
                                      (__builtin_log2((double)kFormTemps[t1]) - __builtin_log2((double)kFormTemps[t0])));
  float lo[kFormNW], hi[kFormNW];
  form_lerp(kFormThread[d0][t0], kFormThread[d1][t0], td, lo, 4);
  form_lerp(kFormThread[d0][t1], kFormThread[d1][t1], td, hi, 4);
  form_lerp(lo, hi, tt, thread4, 4);
  form_lerp(kFormQuad[d0][t0], kFormQuad[d1][t0], td, lo, kFormNW);
  form_lerp(kFormQuad[d0][t1], kFormQuad[d1][t1], td, hi, kFormNW);
  form_lerp(lo, hi, tt, quad, kFormNW);
}

static bool lane_split_is_faster(int dim, int n_temps, double w) {
  if (dim < 16 || w > kFormW[kFormNW - 1]) return false;
  float a[4], q[kFormNW];
  form_params(dim, n_temps, a, q);
  int k = (int)__builtin_ceil(w - 1e-9);
  if (k < 1) k = 1;
  const double thread = a[k - 1] * w / k;
  // the lane-split form: the same saw-tooth on its four times finer scale - the measured rate at the next whole number of
  // lane-split waves per SIMD, times the fill of that last wave slot
  // (ladders of more than 16 temperatures: a whole workgroup of 2-8 waves per ladder is the unit and the dispatcher spreads
  // them over the CUs: no saw-tooth of its own, the measured curve is interpolated as it is)
  double kq = 4.0 * w;
  if (n_temps <= 16) {
    kq = __builtin_ceil(4.0 * w - 1e-9);
    if (kq < 1.0) kq = 1.0;
  }
  double wu = kq / 4.0, b;
  if (wu <= kFormW[0]) {
    b = q[0];
  } else {
    if (wu > kFormW[kFormNW - 1]) wu = kFormW[kFormNW - 1];
    int i = 0;
    while (i + 2 < kFormNW && kFormW[i + 1] <= wu + 1e-9) ++i;
    b = q[i] + (q[i + 1] - q[i]) * (wu - kFormW[i]) / (kFormW[i + 1] - kFormW[i]);
  }
  return b * (4.0 * w / kq) > thread;
}

static int g_kernel_form = PTRWM_FORM_AUTO;  // read / written with __atomic builtins (ptrwm_set_kernel_form may race with a launch)
static int g_stream_mode = PTRWM_STREAM_AUTO;  // likewise (ptrwm_set_stream_mode)

// ---- short launches: the streaming form of the one-thread-per-replica kernel (kernel.h STREAM) ---------------------------
// A launch of ONE Metropolis step over a large batch - the reference's step()-at-a-time loops
// (rwm_gpu_optimized.py:456-457, pt_rwm_gpu_optimized.py:736-737), the harness's benchmark_performance - is bound by
// memory traffic, not by instruction issue: the state is read, stepped once and written back.  The classic kernel gives
// every wave ONE group: load, compute, store, exit - the phases of a wave do not overlap and a SIMD's resident waves run
// them nearly in lock-step.  The streaming form keeps as many waves as the device holds resident and lets each walk many
// groups with the next group's state already in flight (LDS-DMA) and the previous group's stores still draining.  Same
// Philox words, same arithmetic, same canonical order: bit-identical to the classic kernel
// (tests/test_gpu_engine_parity.py test_streaming_form_*).
// Where it pays (profiles/r04_stream_variants.txt, BASELINE configs[2]'s shape over batch sizes, four boxes): two slabs of
// rows per wave in LDS leave room for half the classic kernel's waves, so it wins where overlap is worth more than
// residency - launches of one step whose arrays total about the size of the 256 MiB Infinity Cache (+12-13 % at
// 65 536 ladders x 32 x dim 30, 280 MiB: reads partly served on-die, the classic kernel's lock-step the bottleneck); it
// ties (+-2 %) on batches that fit the cache and trails by 1-6 % where everything streams from HBM (both forms then move
// 0.86-0.9 of what a plain copy moves); from two steps per launch on the classic kernel's residency wins.  AUTO takes it
//   - for launches of one step (kStreamMaxSteps),
//   - whose arrays total kStreamMinBytes..kStreamMaxBytes (0.75x .. 1.75x the Infinity Cache),
//   - where the variant has a streaming twin (dim compiled in, n_temps <= 64) and the layout allows whole aligned 16-byte
//     vectors per group (every group full: n_chains a multiple of the ladders per wave; cpw * n_temps * dim a multiple of
//     4 and cpw * n_temps even; state, sq_jump and n_accept 16-byte aligned) - otherwise the classic kernel's general staging.
constexpr int kStreamMaxSteps = 1;
constexpr long long kStreamMinBytes = 192ll << 20, kStreamMaxBytes = 448ll << 20;

// what the calling thread's most recent ptrwm_run launched (ptrwm_last_launch_kind: tests and the benchmark's record)
static thread_local int t_last_launch_kind = 0;

static bool stream_layout_ok(const ptrwm_run_args *args, int dim, int cpw) {
  return args->n_temps <= 64 && args->n_chains % cpw == 0 && ((long long)cpw * args->n_temps * dim) % 4 == 0 &&
         (cpw * args->n_temps) % 2 == 0 && (reinterpret_cast<uintptr_t>(args->state) & 15u) == 0 &&
         (reinterpret_cast<uintptr_t>(args->sq_jump) & 15u) == 0 && (reinterpret_cast<uintptr_t>(args->n_accept) & 15u) == 0;
}

// SIMDs (compute units x 4) of the device that owns `stream` - the device the launch will run on - or, for the NULL
// stream, of the calling thread's current device; asked once per device; 0 if the runtime cannot say.  The form rule is
// stated in wavefronts per SIMD, so a partitioned or CU-masked device (or another CDNA part) gets the rule scaled to what
// it really has - and where the count is unknown AUTO keeps the one-thread-per-replica form (never wrong, only a speed).
static long long device_simds(hipStream_t stream) {
  constexpr int kMaxDevices = 64;
  static int cached[kMaxDevices];  // 0 = not asked yet (a race merely asks twice)
  int dev = -1;
  hipDevice_t owner;
  if (stream != nullptr && hipStreamGetDevice(stream, &owner) == hipSuccess) dev = (int)owner;
  else if (hipGetDevice(&dev) != hipSuccess) dev = -1;
  if (dev < 0 || dev >= kMaxDevices) return 0;
  int n = __atomic_load_n(&cached[dev], __ATOMIC_RELAXED);
  if (n == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 0;
    n = 4 * cus;
    __atomic_store_n(&cached[dev], n, __ATOMIC_RELAXED);
  }
  return n;
}

// AUTO's choice for a launch of n_chains ladders on a device of n_simds SIMDs where both forms exist
static bool auto_prefers_lane_split(int dim, int n_temps, long long n_chains, long long n_simds) {
  if (n_simds <= 0) return false;
  const long long cpw1 = n_temps > 64 ? 1 : 64 / n_temps;
  const long long waves1 = n_temps > 64 ? n_chains * ((n_temps + 63) / 64) : (n_chains + cpw1 - 1) / cpw1;
  return lane_split_is_faster(dim, n_temps, (double)waves1 / (double)n_simds);
}

#ifndef PTRWM_SOURCE_HASH
#define PTRWM_SOURCE_HASH "unknown (built without csrc/Makefile)"
#endif


// alt: the specialised functor of the kind - RoughCarpet2 (the host proved the third mixture term negligible,
// rough_carpet_two_term) or ThreeMixture1 (the caller declared means that differ in the first coordinate only, ip[0] = 1)
static const QuadVariants &quad_variants(int kind, bool alt) {
  switch (kind) {
    case PTRWM_TARGET_ROUGH_CARPET: return alt ? rough_carpet2_variants_quad() : rough_carpet_variants_quad();
    case PTRWM_TARGET_THREE_MIXTURE: return alt ? three_mixture1_variants_quad() : three_mixture_variants_quad();
    case PTRWM_TARGET_FULL_ROSENBROCK: return full_rosenbrock_variants_quad();
    case PTRWM_TARGET_EVEN_ROSENBROCK: return even_rosenbrock_variants_quad();
    case PTRWM_TARGET_HYBRID_ROSENBROCK: return hybrid_rosenbrock_variants_quad();
    case PTRWM_TARGET_IID_GAMMA: return iid_gamma_variants_quad();
    case PTRWM_TARGET_IID_BETA: return iid_beta_variants_quad();
    case PTRWM_TARGET_DIAG_GAUSSIAN: return diag_gaussian_variants_quad();
    case PTRWM_TARGET_HYPERCUBE: return hypercube_variants_quad();
    default: return neal_funnel_variants_quad();
  }
}

static VariantPair target_variants(int kind, bool alt = false) {
#define PTRWM_PAIR(SYMBOL) VariantPair{&SYMBOL##_narrow(), &SYMBOL##_wide()}
  switch (kind) {
    case PTRWM_TARGET_ROUGH_CARPET: return alt ? PTRWM_PAIR(rough_carpet2_variants) : PTRWM_PAIR(rough_carpet_variants);
    case PTRWM_TARGET_THREE_MIXTURE: return alt ? PTRWM_PAIR(three_mixture1_variants) : PTRWM_PAIR(three_mixture_variants);
    case PTRWM_TARGET_FULL_ROSENBROCK: return PTRWM_PAIR(full_rosenbrock_variants);
    case PTRWM_TARGET_EVEN_ROSENBROCK: return PTRWM_PAIR(even_rosenbrock_variants);
    case PTRWM_TARGET_HYBRID_ROSENBROCK: return PTRWM_PAIR(hybrid_rosenbrock_variants);
    case PTRWM_TARGET_IID_GAMMA: return PTRWM_PAIR(iid_gamma_variants);
    case PTRWM_TARGET_IID_BETA: return PTRWM_PAIR(iid_beta_variants);
    case PTRWM_TARGET_DIAG_GAUSSIAN: return PTRWM_PAIR(diag_gaussian_variants);
    case PTRWM_TARGET_HYPERCUBE: return PTRWM_PAIR(hypercube_variants);
    default: return PTRWM_PAIR(neal_funnel_variants);
  }
#undef PTRWM_PAIR
}

static int check_target(const ptrwm_target_desc *t) {
  if (t == nullptr) return PTRWM_E_NULL;
  if (t->kind < 0 || t->kind >= PTRWM_TARGET_COUNT) return PTRWM_E_KIND;
  if (t->dim < 1 || t->dim > PTRWM_MAX_DIM) return PTRWM_E_DIM;
  switch (t->kind) {
    case PTRWM_TARGET_THREE_MIXTURE:
      if (t->vec0 == nullptr) return PTRWM_E_NULL;
      if (t->ip[0] != 0 && t->ip[0] != 1) return PTRWM_E_ARG;
      break;
    case PTRWM_TARGET_FULL_ROSENBROCK:
      if (t->dim < 2) return PTRWM_E_DIM;
      if (t->vec0 == nullptr) return PTRWM_E_NULL;
      break;
    case PTRWM_TARGET_EVEN_ROSENBROCK:
      if (t->dim < 2 || (t->dim & 1)) return PTRWM_E_DIM;
      if (t->vec0 == nullptr) return PTRWM_E_NULL;
      break;
    case PTRWM_TARGET_DIAG_GAUSSIAN:
      if (t->vec0 == nullptr || (t->ip[0] == 0 && t->vec1 == nullptr)) return PTRWM_E_NULL;
      break;
    case PTRWM_TARGET_HYBRID_ROSENBROCK:
      if (t->ip[0] < 2 || t->ip[1] < 1) return PTRWM_E_ARG;
      if (t->dim != 1 + t->ip[1] * (t->ip[0] - 1)) return PTRWM_E_DIM;
      break;
    default:
      break;
  }
  return PTRWM_OK;
}

// RoughCarpet: is the smallest of the three per-dimension mixture terms always < 2^-27 of the largest?
// In log2 units a_k(x) = -0.5 log2(e) (x - m_k)^2 + log2 w_k; the three parabolas share their curvature, so every
// difference a_j - a_k is LINEAR in x and g(x) = max_k a_k - min_k a_k is the maximum of six lines: convex and
// piecewise linear.  Its minimum over the real line is therefore attained where two of the lines cross (or g is
// constant), so checking the (at most 15) crossings is exact.  A few dozen flops: this runs on every ptrwm_run.
static bool rough_carpet_two_term(const float *p) {
  const double l2e = 1.4426950408889634;
  double sl[6], ic[6];  // line i: sl[i] * x + ic[i]
  int n = 0;
  for (int j = 0; j < 3; ++j)
    for (int k = 0; k < 3; ++k) {
      if (j == k) continue;
      const double mj = p[j], mk = p[k];
      sl[n] = l2e * (mj - mk);
      ic[n] = l2e * (-0.5 * (mj * mj - mk * mk) + ((double)p[3 + j] - (double)p[3 + k]));
      if (!(sl[n] == sl[n]) || !(ic[n] == ic[n])) return false;  // NaN parameters
      ++n;
    }
  auto g = [&](double x) {
    double v = -1e300;
    for (int i = 0; i < 6; ++i) {
      const double y = sl[i] * x + ic[i];
      v = y > v ? y : v;
    }
    return v;
  };
  double gmin = g(0.0);  // covers the all-slopes-equal (constant) case
  for (int i = 0; i < 6; ++i)
    for (int j = i + 1; j < 6; ++j) {
      if (sl[i] == sl[j]) continue;
      const double x = (ic[j] - ic[i]) / (sl[i] - sl[j]);
      if (!(x == x) || x > 1e30 || x < -1e30) continue;
      const double v = g(x);
      gmin = v < gmin ? v : gmin;
    }
  return gmin > 27.0;
}

static TParams make_tparams(const ptrwm_target_desc *t) {
  TParams tp;
  for (int i = 0; i < 12; ++i) tp.p[i] = t->p[i];
  for (int i = 0; i < 4; ++i) tp.ip[i] = t->ip[i];
  tp.vec0 = t->vec0;
  tp.vec1 = t->vec1;
  tp.mask[0] = tp.mask[1] = 0;
  if (t->kind == PTRWM_TARGET_ROUGH_CARPET) {
    // fold the per-dimension -log(sqrt(2 pi)) of multimodal_torch.py:500 into one constant
    tp.p[7] = -(float)t->dim * 0.91893853320467274178f;
  }
  if (t->kind == PTRWM_TARGET_HYBRID_ROSENBROCK) {
    const int blk = t->ip[0] - 1;
    for (int i = 1; i < t->dim; ++i)
      if ((i - 1) % blk == 0) tp.mask[i >> 6] |= (1ull << (i & 63));
  }
  return tp;
}

__global__ void split_advance_kernel(long long *device_step, long long n) { *device_step += n; }

__global__ void philox_raw_kernel(uint32_t *__restrict__ out, long long n, uint32_t c0, uint32_t c1, uint32_t c2,
                                  uint32_t c3, uint32_t k0, uint32_t k1) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32x4 r = philox4x32_10(c0 + (uint32_t)i, c1, c2, c3, k0, k1);
  out[4 * i + 0] = r.x;
  out[4 * i + 1] = r.y;
  out[4 * i + 2] = r.z;
  out[4 * i + 3] = r.w;
}

// Stand-alone swap event: one workgroup per ladder, thread t = temperature t.  The decision is swap_decide(), the
// code the fused kernel runs; the row permutation goes through LDS in column chunks (a permutation of rows can be
// applied to every block of columns independently), so any (n_temps, dim) fits the 32 KB static buffer.
struct SweepArgs {
  float *state, *logp;
  const float *beta, *ext_swap_u;
  long long *swap_accept, *last_swap_ordinal;
  const float *prev;  // split step: states before the step's MH move, or NULL
  double *sq_jump;    // split step: += |final - prev|^2 per replica (the fused kernel's swap-step jump), or NULL
  long long chain_offset, event_index;
  unsigned long long step;
  int n_temps, dim, swap_mode, swap_order, rng_stream, chunk;
  unsigned k0, k1;
  // device-step mode (include/ptrwm.h device_step): the step index, whether an event is due and its number come from here
  const long long *device_step;
  long long burn_in, swap_every, event_offset;
};
constexpr int kSweepLdsBytes = 32768;  // at most: rows (and, in a split step, the pre-step rows) of one column chunk

// state_t: float, or double for the reference's dtype=torch.float64 states (ptrwm_swap_sweep with state_f64 = 1: the
// permutation only - the squared-jump bookkeeping belongs to split steps, which carry float states).
// LDS is sized by the launch to what the ladder needs (rows of one column chunk, twice that in a split step): round 3's
// fixed 32 KB left four workgroups - four wavefronts - per CU and the kernel at a tenth of the memory rate.
template <class state_t>
__global__ void __launch_bounds__(256) swap_sweep_kernel(SweepArgs a) {
  if (a.device_step != nullptr) {
    // the swap event of step *device_step, if that step has one (ptrwm_split_accept's host-side rule, on the device)
    const long long s0 = *a.device_step + (long long)a.step, sc = s0 + 1;  // (a.step: this call's offset, include/ptrwm.h)
    if (!(sc > a.burn_in && sc % a.swap_every == 0)) return;  // (grid-uniform)
    a.step = (unsigned long long)s0;
    a.event_index = sc / a.swap_every - a.burn_in / a.swap_every - 1 + a.event_offset;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char s_sweep[];
  __shared__ float s_l[256], s_u[256];
  __shared__ int s_src[256];
  const int T = a.n_temps, D = a.dim, tid = threadIdx.x, nthr = blockDim.x;
  state_t *const s_rows = reinterpret_cast<state_t *>(s_sweep);
  state_t *const s_prev = s_rows + T * a.chunk;  // (split steps only)
  const long long chain = blockIdx.x;
  const bool live = tid < T;
  const int t = live ? tid : 0;
  float my_l = a.logp[chain * T + t];
  float us;
  if (a.ext_swap_u != nullptr) {
    us = (t < T - 1) ? a.ext_swap_u[chain * (T - 1) + t] : 2.0f;
  } else {
    // the counter layout of the fused kernel's swap stream (kernel.h): block 0 | step_hi, step, chain, t | stream | chain_hi
    const unsigned long long gchain = (unsigned long long)(a.chain_offset + chain);
    const u32x4 r = philox4x32_10((uint32_t)(a.step >> 32) << 16, (uint32_t)a.step, (uint32_t)gchain,
                                  (uint32_t)t | ((uint32_t)a.rng_stream << 8) | ((uint32_t)(gchain >> 32) << 12), a.k0,
                                  a.k1);
    us = u01(r.x);
  }
  if (live) {
    s_l[tid] = my_l;
    s_u[tid] = us;
  }
  // the ladder's verdict on the form of its sequential sweep, from the values it enters the event with (kernel.h)
  const bool plain = __syncthreads_and(swap_pair_plain(T, t, sub_rn(a.beta[t], a.beta[t < T - 1 ? t + 1 : t]), my_l, us) ? 1 : 0) != 0;
  int src = t;
  bool pair_acc = false;
  __shared__ int s_landed[256];
  swap_decide(T, t, 0, t, a.swap_mode, a.swap_order, (int)(a.event_index & 1), a.beta, a.beta[t], us, s_l, s_u, s_landed,
              my_l, src, pair_acc, [] { __syncthreads(); }, plain);
  if (live) s_src[tid] = src;
  __syncthreads();
  state_t *gs = reinterpret_cast<state_t *>(a.state) + chain * T * (long long)D;
  const float *prev = a.prev != nullptr ? a.prev + chain * T * (long long)D : nullptr;
  // |final - prev|^2 of this thread's replica in the canonical four-range order of the fused kernel (philox.h)
  const int W = canon_width(D);
  float j2p0 = 0.0f, j2p1 = 0.0f, j2p2 = 0.0f, j2p3 = 0.0f;
  for (int c0 = 0; c0 < D; c0 += a.chunk) {
    const int w = (D - c0 < a.chunk) ? D - c0 : a.chunk;
    const bool whole = w == D;  // one chunk holds whole rows: element i of the tile is element i of the ladder's run
    for (int i = tid; i < T * w; i += nthr) {
      const int tt = whole ? 0 : i / w, dd = whole ? i : i - tt * w;
      s_rows[i] = gs[tt * D + c0 + dd];
      if (prev != nullptr) s_prev[i] = static_cast<state_t>(prev[tt * D + c0 + dd]);
    }
    __syncthreads();
    for (int i = tid; i < T * w; i += nthr) {
      const int tt = i / w, dd = i - tt * w;
      gs[tt * D + c0 + dd] = s_rows[s_src[tt] * w + dd];
    }
    if (live && prev != nullptr) {
      for (int dd = 0; dd < w; ++dd) {
        const float dl = sub_rn(static_cast<float>(s_rows[src * w + dd]), static_cast<float>(s_prev[t * w + dd]));
        const int qi = (c0 + dd) / W;
        if (qi == 0) j2p0 = fmaf(dl, dl, j2p0);
        else if (qi == 1) j2p1 = fmaf(dl, dl, j2p1);
        else if (qi == 2) j2p2 = fmaf(dl, dl, j2p2);
        else j2p3 = fmaf(dl, dl, j2p3);
      }
    }
    __syncthreads();
  }
  if (live) {
    const long long rep = chain * T + t;
    a.logp[rep] = my_l;
    if (a.sq_jump != nullptr) a.sq_jump[rep] += (double)add_rn(add_rn(j2p0, j2p1), add_rn(j2p2, j2p3));
    if (pair_acc) {
      if (a.swap_accept != nullptr) a.swap_accept[rep] += 1;
      if (a.last_swap_ordinal != nullptr) {
        const long long ord =
            (a.swap_order == PTRWM_ORDER_SEQUENTIAL) ? a.event_index * (T - 1) + t + 1 : a.event_index + 1;
        if (ord > a.last_swap_ordinal[rep]) a.last_swap_ordinal[rep] = ord;
      }
    }
  }
}

// Split step, second half: Metropolis rule on caller-evaluated log-densities, one thread per replica.  `proposals`
// comes back holding the pre-step states (the swap event of this step, if any, needs them for the jump distance).
struct SplitAcceptArgs {
  float *state, *logp, *proposals;
  const float *beta, *accept_u, *logp_new;
  long long *n_accept;
  double *sq_jump;
  unsigned char *accept_flags;
  long long n_reps;
  int n_temps, dim, count_on, swap_due;
  const long long *device_step;  // device-step mode: count_on / swap_due are derived from *device_step + step_offset
  long long step_offset;
  long long burn_in, swap_every;
};

// One wavefront per tile of 64 replicas (64-thread workgroups): the tile's rows of `state` and of `proposals` go through two
// slabs of LDS in both directions (kernel.h stage_copy: coalesced 16-byte transfers), every lane works on its own row in
// LDS.  (Round 3's version walked its rows in HBM, 64 cache lines per instruction: 2.6 ms per step at 65 536 x 32 x
// dim 30 - 70 % of a split step - where moving the bytes takes 0.2.)
__global__ void __launch_bounds__(64) split_accept_kernel(SplitAcceptArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_tiles[];
  const int lane = (int)threadIdx.x;
  const long long first = (long long)blockIdx.x * 64;
  if (first >= a.n_reps) return;
  if (a.device_step != nullptr) {
    const long long sc = *a.device_step + a.step_offset + 1;  // step_counter of this step
    a.count_on = sc > a.burn_in;
    a.swap_due = a.n_temps > 1 && a.count_on && (sc % a.swap_every == 0);
  }
  const int D = a.dim;
  const int n_rows = (a.n_reps - first < 64) ? (int)(a.n_reps - first) : 64;
  float *const xs = s_tiles, *const ys = s_tiles + (64 * D + 4);
  float *__restrict__ gx = a.state + first * D;
  float *__restrict__ gy = a.proposals + first * D;
  stage_copy<true>(xs, gx, n_rows * D, lane, 64);
  stage_copy<true>(ys, gy, n_rows * D, lane, 64);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const bool live = lane < n_rows;
  const long long i = first + (live ? lane : 0);
  bool acc = false;
  float lp_new = 0.0f, j2 = 0.0f;
  if (live) {
    const int t = (int)(i % a.n_temps);
    const float lp = a.logp[i];
    lp_new = a.logp_new[i];
    acc = mh_accept(a.beta[t], lp_new, lp, a.accept_u[i]);
    float *__restrict__ x = xs + stage_head(gx) + lane * D;
    float *__restrict__ y = ys + stage_head(gy) + lane * D;
    // squared jump in the canonical four-range order of the fused kernel (philox.h)
    const int W = canon_width(D);
    float j2p[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int d1 = (q + 1) * W < D ? (q + 1) * W : D;
      for (int d = q * W; d < d1; ++d) {
        const float xo = x[d], yn = y[d];
        const float dl = sub_rn(yn, xo);
        j2p[q] = fmaf(dl, dl, j2p[q]);
        if (acc) x[d] = yn;
        y[d] = xo;
      }
    }
    // (the proposal's own squared jump when ptrwm_split_propose recorded one: second plane of accept_u, kernel.h)
    const float given = a.accept_u[a.n_reps + i];
    j2 = given >= 0.0f ? given : tree4_add(j2p);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  stage_copy<false>(xs, gx, n_rows * D, lane, 64);
  // the pre-step states go back in place of the proposals only where somebody reads them: the swap kernel of a swap step
  // (its squared jump is |final - pre-step|^2).  A quarter of this kernel's HBM traffic on the other nine steps in ten.
  if (a.swap_due) stage_copy<false>(ys, gy, n_rows * D, lane, 64);
  if (!live) return;
  if (acc) a.logp[i] = lp_new;
  if (a.accept_flags != nullptr) a.accept_flags[i] = acc ? 1 : 0;
  if (a.count_on) {
    if (a.n_accept != nullptr) a.n_accept[i] += acc ? 1 : 0;
    // a swap step's jump (MH move and swap together) is added by the sweep that follows
    if (a.sq_jump != nullptr && !a.swap_due && acc) a.sq_jump[i] += (double)j2;
  }
}

template <template <int> class Proposal>
static hipError_t launch_split_propose(int wi, const float *state, float *proposals, float *accept_u, long long n_chains,
                                       long long chain_offset, unsigned long long step, int D, int T, const float *ts,
                                       const PParams &pp, const float *ext_raw, const float *ext_u, int n_raw,
                                       unsigned k0, unsigned k1, const long long *dstep, hipStream_t st) {
  const long long tot = n_chains * T;
  const dim3 grid((unsigned)((tot + 63) / 64)), block(64);  // one wavefront per tile of 64 replicas (kernel.h)
  const unsigned lds = split_tile_lds_bytes(D);
  int idx = 0;
#define PTRWM_X_SPLIT(W, E)                                                                                       \
  if (wi == idx++)                                                                                                \
    hipLaunchKernelGGL((ptrwm_split_propose_kernel<Proposal<W>, W>), grid, block, lds, st, state, proposals,     \
                       accept_u, n_chains, chain_offset, step, D, T, ts, pp, ext_raw, ext_u, n_raw, k0, k1, dstep);
  PTRWM_WIDTHS(PTRWM_X_SPLIT)
#undef PTRWM_X_SPLIT
  return hipGetLastError();
}

template <template <int> class Proposal>
static hipError_t launch_propose(int wi, float *out, long long n, int D, int T, const float *ts, const PParams &pp,
                                 const float *ext_raw, int n_raw, unsigned k0, unsigned k1, hipStream_t st) {
  const long long tot = n * T;
  const dim3 grid((unsigned)((tot + kBlockThreads - 1) / kBlockThreads)), block(kBlockThreads);
  int idx = 0;
#define PTRWM_X_PROPOSE(W, E)                                                                              \
  if (wi == idx++)                                                                                         \
    hipLaunchKernelGGL((ptrwm_propose_kernel<Proposal<W>, W>), grid, block, 0, st, out, n, D, T, ts, pp,   \
                       ext_raw, n_raw, k0, k1);
  PTRWM_WIDTHS(PTRWM_X_PROPOSE)
#undef PTRWM_X_PROPOSE
  return hipGetLastError();
}

// One swap event over the current states (ptrwm_swap_sweep, and the swap step of ptrwm_split_accept).
static int32_t launch_sweep(const ptrwm_run_args *args, int32_t dim, int64_t event_index, int32_t rng_stream,
                            const float *prev, double *sq_jump, hipStream_t stream) {
  SweepArgs a;
  a.state = args->state;
  a.logp = args->logp;
  a.beta = args->beta;
  a.ext_swap_u = args->ext_swap_u;
  a.swap_accept = (long long *)args->swap_accept;
  a.last_swap_ordinal = (long long *)args->last_swap_ordinal;
  a.prev = prev;
  a.sq_jump = sq_jump;
  a.chain_offset = args->chain_offset;
  a.event_index = event_index;
  a.step = (unsigned long long)args->step0;
  a.n_temps = args->n_temps;
  a.dim = dim;
  a.swap_mode = args->swap_mode;
  a.swap_order = args->swap_order;
  a.rng_stream = rng_stream;
  // columns per pass: whole rows where they fit the LDS budget (a split step stages the pre-step rows beside them)
  const bool f64 = args->state_f64 == 1;
  const int per_elem = (f64 ? 8 : 4) * (prev != nullptr ? 2 : 1);
  int chunk = kSweepLdsBytes / (per_elem * args->n_temps);  // >= 16 columns
  if (chunk > dim) chunk = dim;
  a.chunk = chunk;
  const unsigned lds = (unsigned)(chunk * args->n_temps * per_elem);
  a.k0 = (unsigned)(args->seed & 0xffffffffull);
  a.k1 = (unsigned)(args->seed >> 32);
  a.device_step = prev != nullptr ? (const long long *)args->device_step : nullptr;  // (split steps only)
  a.burn_in = args->burn_in;
  a.swap_every = args->swap_every;
  a.event_offset = args->swap_event_offset;
  const unsigned block = (unsigned)((args->n_temps + 63) / 64 * 64);
  if (f64)
    hipLaunchKernelGGL(swap_sweep_kernel<double>, dim3((unsigned)args->n_chains), dim3(block), lds, stream, a);
  else
    hipLaunchKernelGGL(swap_sweep_kernel<float>, dim3((unsigned)args->n_chains), dim3(block), lds, stream, a);
  return hipGetLastError() == hipSuccess ? PTRWM_OK : PTRWM_E_LAUNCH;
}

}  // namespace ptrwm

using namespace ptrwm;

extern "C" {

int32_t ptrwm_abi_version(void) { return PTRWM_ABI_VERSION; }

const char *ptrwm_strerror(int32_t code) {
  switch (code) {
    case PTRWM_OK: return "ok";
    case PTRWM_E_NULL: return "required pointer is NULL";
    case PTRWM_E_DIM: return "dim out of range or invalid for this target";
    case PTRWM_E_TEMPS: return "n_temps out of range (1..256)";
    case PTRWM_E_KIND: return "unknown target or proposal kind";
    case PTRWM_E_ARG: return "invalid argument";
    case PTRWM_E_STRUCT: return "struct_size mismatch (ABI version)";
    case PTRWM_E_LAUNCH: return "HIP launch error";
    case PTRWM_E_NOVARIANT: return "variant not compiled";
    default: return "unknown error";
  }
}

int32_t ptrwm_set_kernel_form(int32_t form) {
  if (form != PTRWM_FORM_AUTO && form != PTRWM_FORM_THREAD && form != PTRWM_FORM_QUAD) return PTRWM_E_ARG;
  return __atomic_exchange_n(&g_kernel_form, form, __ATOMIC_RELAXED);
}

int32_t ptrwm_set_stream_mode(int32_t mode) {
  if (mode != PTRWM_STREAM_AUTO && mode != PTRWM_STREAM_OFF && mode != PTRWM_STREAM_ON) return PTRWM_E_ARG;
  return __atomic_exchange_n(&g_stream_mode, mode, __ATOMIC_RELAXED);
}

int32_t ptrwm_last_launch_kind(void) { return t_last_launch_kind; }

int32_t ptrwm_has_stream_variant(int32_t target_kind, int32_t proposal_kind, int32_t dim) {
  if (ptrwm_has_thread_variant(target_kind, proposal_kind, dim) == 0) return 0;
  const int dpi = width_index_for_dim(dim, target_kind);
  return has_stream_variant(kWidths[dpi].dp, kWidths[dpi].exact) ? 1 : 0;
}

int32_t ptrwm_has_quad_variant(int32_t target_kind, int32_t proposal_kind, int32_t dim, int32_t n_temps) {
  if (target_kind < 0 || target_kind >= PTRWM_TARGET_COUNT) return 0;
  if (proposal_kind < 0 || proposal_kind >= PTRWM_PROPOSAL_COUNT) return 0;
  const int qi = quad_index_for(dim, n_temps, target_kind);
  return qi >= 0 && quad_variants(target_kind, false).run[proposal_kind][qi] != nullptr ? 1 : 0;
}

int32_t ptrwm_device_simds(void *stream) {
  const long long n = device_simds((hipStream_t)stream);
  return n > 0 ? (int32_t)n : PTRWM_E_LAUNCH;
}

int32_t ptrwm_auto_form_for(int32_t target_kind, int32_t proposal_kind, int32_t dim, int32_t n_temps, int64_t n_chains,
                            int32_t n_simds) {
  if (n_temps < 1 || n_temps > PTRWM_MAX_TEMPS || n_chains < 1 || n_simds < 1) return PTRWM_E_ARG;
  const bool th = ptrwm_has_thread_variant(target_kind, proposal_kind, dim) != 0;
  const bool qu = ptrwm_has_quad_variant(target_kind, proposal_kind, dim, n_temps) != 0;
  if (!th && !qu) return PTRWM_E_NOVARIANT;
  if (!th) return PTRWM_FORM_QUAD;
  if (!qu) return PTRWM_FORM_THREAD;
  return auto_prefers_lane_split(dim, n_temps, n_chains, n_simds) ? PTRWM_FORM_QUAD : PTRWM_FORM_THREAD;
}

int32_t ptrwm_auto_form(int32_t target_kind, int32_t proposal_kind, int32_t dim, int32_t n_temps, int64_t n_chains) {
  const int32_t n = ptrwm_device_simds(nullptr);
  return n < 0 ? n : ptrwm_auto_form_for(target_kind, proposal_kind, dim, n_temps, n_chains, n);
}

const char *ptrwm_source_hash(void) { return PTRWM_SOURCE_HASH; }
const char *ptrwm_form_table_source_hash(void) { return kFormTableSourceHash; }

int32_t ptrwm_has_thread_variant(int32_t target_kind, int32_t proposal_kind, int32_t dim) {
  if (target_kind < 0 || target_kind >= PTRWM_TARGET_COUNT) return 0;
  if (proposal_kind < 0 || proposal_kind >= PTRWM_PROPOSAL_COUNT) return 0;
  const int dpi = width_index_for_dim(dim, target_kind);
  return dim >= 1 && dpi >= 0 && target_variants(target_kind).run(proposal_kind, dpi) != nullptr ? 1 : 0;
}

int32_t ptrwm_ext_raw_per_step(int32_t proposal_kind, int32_t dim) {
  switch (proposal_kind) {
    case PTRWM_PROPOSAL_NORMAL: return dim;
    case PTRWM_PROPOSAL_LAPLACE: return dim;
    case PTRWM_PROPOSAL_UNIFORM_RADIUS: return dim + 1;
    default: return PTRWM_E_KIND;
  }
}

int32_t ptrwm_has_variant(int32_t target_kind, int32_t proposal_kind, int32_t dim) {
  if (target_kind < 0 || target_kind >= PTRWM_TARGET_COUNT) return 0;
  if (proposal_kind < 0 || proposal_kind >= PTRWM_PROPOSAL_COUNT) return 0;
  const int dpi = width_index_for_dim(dim, target_kind);
  if (dim < 1 || dpi < 0) return 0;
  if (target_variants(target_kind).run(proposal_kind, dpi) != nullptr) return 1;
  const int qi = quad_index_for(dim, 1, target_kind);  // above width 64 the lane-split kernel is the fused kernel
  return qi >= 0 && quad_variants(target_kind, false).run[proposal_kind][qi] != nullptr ? 1 : 0;
}

int32_t ptrwm_run(const ptrwm_target_desc *target, const ptrwm_proposal_desc *proposal, const ptrwm_run_args *args,
                  void *hip_stream) {
  if (proposal == nullptr || args == nullptr) return PTRWM_E_NULL;
  if (int rc = check_target(target)) return rc;
  if (args->struct_size != sizeof(ptrwm_run_args)) return PTRWM_E_STRUCT;
  if (proposal->kind < 0 || proposal->kind >= PTRWM_PROPOSAL_COUNT) return PTRWM_E_KIND;
  if (args->n_temps < 1 || args->n_temps > PTRWM_MAX_TEMPS) return PTRWM_E_TEMPS;
  if (args->n_chains < 0 || args->n_steps < 0 || args->step0 < 0 || args->burn_in < 0 || args->swap_every < 1)
    return PTRWM_E_ARG;
  if (args->swap_mode != PTRWM_SWAP_EXCHANGE && args->swap_mode != PTRWM_SWAP_REFERENCE_COPY) return PTRWM_E_ARG;
  if (args->swap_order != PTRWM_ORDER_SEQUENTIAL && args->swap_order != PTRWM_ORDER_EVEN_ODD) return PTRWM_E_ARG;
  if ((args->state_f64 != 0 && args->state_f64 != 1) || args->split_flags != 0 || args->device_step != nullptr) return PTRWM_E_ARG;
  if (args->n_chains == 0 || args->n_steps == 0) return PTRWM_OK;  // empty batch: nothing to touch
  if (args->state == nullptr || args->logp == nullptr || args->beta == nullptr || proposal->temp_scale == nullptr)
    return PTRWM_E_NULL;
  if (proposal->kind == PTRWM_PROPOSAL_LAPLACE && proposal->dim_scale == nullptr) return PTRWM_E_NULL;
  const bool ext = args->ext_prop != nullptr;
  const bool f64 = args->state_f64 == 1;  // double state / trace / ext_prop (include/ptrwm.h): lane-split form only
  if (f64 && ext && proposal->kind != PTRWM_PROPOSAL_NORMAL) return PTRWM_E_ARG;
  if (ext && args->ext_u == nullptr) return PTRWM_E_NULL;
  if (args->trace != nullptr && (args->trace_chains < 1 || args->trace_temps < 1 || args->trace_row0 < 0 ||
                                 args->trace_temps > args->n_temps || args->trace_chains > args->n_chains ||
                                 args->trace_every < 0))
    return PTRWM_E_ARG;

  const int dpi = width_index_for_dim(target->dim, target->kind);
  if (dpi < 0) return PTRWM_E_DIM;
  const bool two_term = (target->kind == PTRWM_TARGET_ROUGH_CARPET && rough_carpet_two_term(target->p)) ||
                        (target->kind == PTRWM_TARGET_THREE_MIXTURE && target->ip[0] == 1);  // the kind's specialised functor
  RunLaunchFn fn = target_variants(target->kind, two_term).run(proposal->kind, dpi);  // null above width 64
  // lane-split form? (bit-identical results: a speed decision, see g_kernel_form - except above dim 64, where it is the
  // only form)
  bool quad = false;
  {
    const int qi = quad_index_for(target->dim, args->n_temps, target->kind);
    const QuadVariants &qv = quad_variants(target->kind, two_term);
    const RunLaunchFn qfn = qi >= 0 ? (f64 ? qv.run_f64 : qv.run)[proposal->kind][qi] : nullptr;
    const int form = __atomic_load_n(&g_kernel_form, __ATOMIC_RELAXED);
    if (f64) {
      fn = qfn;  // the only form with double state registers (null: ladder too long for a 512-thread workgroup)
      quad = true;
    } else if (qfn != nullptr && (fn == nullptr || form != PTRWM_FORM_THREAD)) {
      quad = fn == nullptr || form == PTRWM_FORM_QUAD ||
             auto_prefers_lane_split(target->dim, args->n_temps, args->n_chains, device_simds((hipStream_t)hip_stream));
      if (quad) fn = qfn;
    }
  }
  if (fn == nullptr) return PTRWM_E_NOVARIANT;
  const long long se = args->swap_every;
  const bool full = ext || args->trace != nullptr || args->accept_flags != nullptr;
  // swap events before step_counter sc = multiples m*se with burn_in < m*se <= sc
  auto events_upto = [&](long long sc) -> long long {
    const long long e = sc / se - args->burn_in / se;
    return e > 0 ? e : 0;
  };
  if (ext && args->n_temps > 1 && events_upto(args->step0 + args->n_steps) > events_upto(args->step0) &&
      args->ext_swap_u == nullptr)
    return PTRWM_E_NULL;

  KArgs k;
  k.state = args->state;
  k.logp = args->logp;
  k.beta = args->beta;
  k.temp_scale = proposal->temp_scale;
  k.n_accept = (long long *)args->n_accept;
  k.sq_jump = args->sq_jump;
  k.swap_accept = (long long *)args->swap_accept;
  k.last_swap_ordinal = (long long *)args->last_swap_ordinal;
  k.n_chains = args->n_chains;
  k.chain_offset = args->chain_offset;
  k.n_temps = args->n_temps;
  k.dim = target->dim;
  k.swap_every = args->swap_every;
  k.swap_mode = args->swap_mode;
  k.swap_order = args->swap_order;
  // exchange groups: one wavefront holding whole ladders, or ("wide") one workgroup per ladder
  const int lanes_per_replica = quad ? kQuad : 1;
  const bool wide = args->n_temps * lanes_per_replica > 64;
  k.chains_per_wave = quad ? quad_ladders_per_group(args->n_temps) : (wide ? 1 : 64 / args->n_temps);
  k.k0 = (unsigned)(args->seed & 0xffffffffull);
  k.k1 = (unsigned)(args->seed >> 32);
  k.tp = make_tparams(target);
  k.pp.dim_scale = proposal->dim_scale;
  k.pp.inv_dim = proposal->inv_dim;
  k.full.trace = args->trace;
  k.full.trace_logp = args->trace_logp;
  k.full.trace_chains = args->trace != nullptr ? args->trace_chains : 0;
  k.full.trace_temps = args->trace_temps;
  k.full.n_raw_ext = ptrwm_ext_raw_per_step(proposal->kind, target->dim);

  // the streaming form for short launches of the one-thread-per-replica kernel (see kStreamMaxSteps above)
  bool stream = false;
  if (!quad && !full && has_stream_variant(kWidths[dpi].dp, kWidths[dpi].exact) && stream_layout_ok(args, target->dim, k.chains_per_wave)) {
    const int mode = __atomic_load_n(&g_stream_mode, __ATOMIC_RELAXED);
    if (mode == PTRWM_STREAM_ON) {
      stream = true;
    } else if (mode == PTRWM_STREAM_AUTO && args->n_steps <= kStreamMaxSteps) {
      // state + log-density + the two statistics every launch touches (acceptance count, squared-jump sum)
      const long long bytes = args->n_chains * (long long)args->n_temps * (4ll * target->dim + 20ll);
      stream = bytes >= kStreamMinBytes && bytes <= kStreamMaxBytes;
    }
  }

  const long long n_waves = (args->n_chains + k.chains_per_wave - 1) / k.chains_per_wave;
  const long long n_blocks = wide ? n_waves : (n_waves + kWavesPerBlock - 1) / kWavesPerBlock;  // wide: one group per block
  if (n_blocks > 0x7fffffffll) return PTRWM_E_ARG;

  // One launch covers a bounded amount of work (32-bit in-kernel counters; no multi-second kernels on a shared
  // GPU): at most 2^16 steps and about 2^33 chain-steps (~0.2 s at 4e10/s).  Longer requests become back-to-back
  // launches on the same stream; step0 carries the swap schedule and the RNG position, so the split is invisible.
  // 2^16 steps also bound how stale a launch-start decision can get: the verdict whether a replica's squared jumps may be
  // taken from the proposal (proposals.h kJumpTrust) is re-taken at least that often - a coordinate cannot drift by more
  // than a few hundred typical increments in between, which keeps the two definitions of the jump within ~1e-4 relative.
  const long long kMaxUnitsPerLaunch = 1ll << 33;
  long long kMaxStepsPerLaunch = kMaxUnitsPerLaunch / (args->n_chains * (long long)args->n_temps);
  if (kMaxStepsPerLaunch < 1) kMaxStepsPerLaunch = 1;
  if (kMaxStepsPerLaunch > (1 << 16)) kMaxStepsPerLaunch = 1 << 16;
  t_last_launch_kind = quad ? PTRWM_LAUNCH_QUAD : (stream ? PTRWM_LAUNCH_STREAM : PTRWM_LAUNCH_THREAD);
  const long long te = args->trace_every > 1 ? args->trace_every : 1;
  k.full.trace_every = (int)te;
  const long long reps = args->n_chains * args->n_temps;
  const long long raw = k.full.n_raw_ext;
  long long done = 0;
  while (done < args->n_steps) {
    const long long step0 = args->step0 + done;
    const long long n = (args->n_steps - done < kMaxStepsPerLaunch) ? args->n_steps - done : kMaxStepsPerLaunch;
    const long long ev0 = events_upto(step0);
    k.step0 = step0;
    k.n_steps = (int)n;
    const long long burn_left = args->burn_in - step0;
    k.burn_left = burn_left <= 0 ? 0 : (burn_left > n ? (int)n : (int)burn_left);
    k.first_swap_event = ev0 + args->swap_event_offset;
    k.steps_to_swap = (int)(se - step0 % se);
    k.full.ext_prop = ext ? args->ext_prop + done * reps * raw * (f64 ? 2 : 1) : nullptr;  // (f64: a double array)
    k.full.ext_u = ext ? args->ext_u + done * reps : nullptr;
    k.full.ext_swap_u = (ext && args->ext_swap_u != nullptr)
                            ? args->ext_swap_u + (ev0 - events_upto(args->step0)) * args->n_chains * (args->n_temps - 1)
                            : nullptr;
    k.full.accept_flags = args->accept_flags != nullptr ? args->accept_flags + done * reps : nullptr;
    // traced steps are those whose step_counter is a multiple of trace_every: rows before this launch
    k.full.trace_row0 = args->trace_row0 + (step0 / te - args->step0 / te);
    k.full.steps_to_trace = (int)(te - step0 % te);
    const hipError_t err = fn(k, (unsigned)n_blocks, full ? kRunFull : (stream ? kRunStream : kRunProd), (hipStream_t)hip_stream);
    if (err != hipSuccess) return PTRWM_E_LAUNCH;
    done += n;
  }
  return PTRWM_OK;
}

int32_t ptrwm_swap_sweep(const ptrwm_run_args *args, int32_t dim, int64_t event_index, int32_t rng_stream,
                         void *stream) {
  if (args == nullptr) return PTRWM_E_NULL;
  if (args->struct_size != sizeof(ptrwm_run_args)) return PTRWM_E_STRUCT;
  if ((args->state_f64 != 0 && args->state_f64 != 1) || args->split_flags != 0 || args->device_step != nullptr) return PTRWM_E_ARG;
  if (dim < 1 || dim > PTRWM_MAX_DIM) return PTRWM_E_DIM;
  if (args->n_temps < 1 || args->n_temps > PTRWM_MAX_TEMPS) return PTRWM_E_TEMPS;
  if (args->n_chains < 0 || args->n_chains > 0x7fffffffll || args->step0 < 0 || event_index < 0 || rng_stream < 1 ||
      rng_stream > 15)
    return PTRWM_E_ARG;
  if (args->swap_mode != PTRWM_SWAP_EXCHANGE && args->swap_mode != PTRWM_SWAP_REFERENCE_COPY) return PTRWM_E_ARG;
  if (args->swap_order != PTRWM_ORDER_SEQUENTIAL && args->swap_order != PTRWM_ORDER_EVEN_ODD) return PTRWM_E_ARG;
  if (args->n_chains == 0 || args->n_temps == 1) return PTRWM_OK;  // nothing to exchange
  if (args->state == nullptr || args->logp == nullptr || args->beta == nullptr) return PTRWM_E_NULL;
  return launch_sweep(args, dim, event_index, rng_stream, nullptr, nullptr, (hipStream_t)stream);
}

static int32_t split_common_checks(const ptrwm_run_args *args, int32_t dim) {
  if (args == nullptr) return PTRWM_E_NULL;
  if (args->struct_size != sizeof(ptrwm_run_args)) return PTRWM_E_STRUCT;
  if (args->state_f64 != 0) return PTRWM_E_ARG;  // float states only
  if ((args->split_flags & ~PTRWM_SPLIT_NO_SWEEP) != 0 || (args->split_flags != 0 && args->device_step == nullptr)) return PTRWM_E_ARG;
  if (dim < 1 || dim > PTRWM_MAX_DIM) return PTRWM_E_DIM;
  if (args->n_temps < 1 || args->n_temps > PTRWM_MAX_TEMPS) return PTRWM_E_TEMPS;
  if (args->n_chains < 0 || args->n_chains > 0x7fffffffll || args->step0 < 0 || args->burn_in < 0 ||
      args->swap_every < 1)
    return PTRWM_E_ARG;
  if (args->swap_mode != PTRWM_SWAP_EXCHANGE && args->swap_mode != PTRWM_SWAP_REFERENCE_COPY) return PTRWM_E_ARG;
  if (args->swap_order != PTRWM_ORDER_SEQUENTIAL && args->swap_order != PTRWM_ORDER_EVEN_ODD) return PTRWM_E_ARG;
  if (args->device_step != nullptr && (args->ext_prop != nullptr || args->ext_u != nullptr || args->ext_swap_u != nullptr))
    return PTRWM_E_ARG;  // device-step mode draws from Philox only
  return PTRWM_OK;
}

int32_t ptrwm_split_propose(const ptrwm_proposal_desc *proposal, const ptrwm_run_args *args, int32_t dim,
                            float *proposals, float *accept_u, void *stream) {
  if (proposal == nullptr) return PTRWM_E_NULL;
  if (int rc = split_common_checks(args, dim)) return rc;
  if (proposal->kind < 0 || proposal->kind >= PTRWM_PROPOSAL_COUNT) return PTRWM_E_KIND;
  if (args->n_chains == 0) return PTRWM_OK;
  if (args->state == nullptr || proposals == nullptr || accept_u == nullptr || proposal->temp_scale == nullptr)
    return PTRWM_E_NULL;
  if (proposal->kind == PTRWM_PROPOSAL_LAPLACE && proposal->dim_scale == nullptr) return PTRWM_E_NULL;
  if (args->ext_prop != nullptr && args->ext_u == nullptr) return PTRWM_E_NULL;
  const int dpi = width_index_for_dim(dim);
  if (dpi < 0) return PTRWM_E_DIM;
  PParams pp;
  pp.dim_scale = proposal->dim_scale;
  pp.inv_dim = proposal->inv_dim;
  const unsigned k0 = (unsigned)(args->seed & 0xffffffffull), k1 = (unsigned)(args->seed >> 32);
  const int n_raw = ptrwm_ext_raw_per_step(proposal->kind, dim);
  hipError_t err;
#define PTRWM_SPLIT_CALL(P)                                                                                          \
  launch_split_propose<P>(dpi, args->state, proposals, accept_u, args->n_chains, args->chain_offset,                 \
                          (unsigned long long)args->step0, dim, args->n_temps, proposal->temp_scale, pp,             \
                          args->ext_prop, args->ext_prop != nullptr ? args->ext_u : nullptr, n_raw, k0, k1,         \
                          (const long long *)args->device_step, (hipStream_t)stream)
  switch (proposal->kind) {
    case PTRWM_PROPOSAL_NORMAL: err = PTRWM_SPLIT_CALL(NormalProposal); break;
    case PTRWM_PROPOSAL_LAPLACE: err = PTRWM_SPLIT_CALL(LaplaceProposal); break;
    default: err = PTRWM_SPLIT_CALL(UniformRadiusProposal); break;
  }
#undef PTRWM_SPLIT_CALL
  return err == hipSuccess ? PTRWM_OK : PTRWM_E_LAUNCH;
}

int32_t ptrwm_split_accept(const ptrwm_run_args *args, int32_t dim, float *proposals, const float *accept_u,
                           const float *logp_proposed, void *stream) {
  if (int rc = split_common_checks(args, dim)) return rc;
  if (args->n_chains == 0) return PTRWM_OK;
  if (args->state == nullptr || args->logp == nullptr || args->beta == nullptr || proposals == nullptr ||
      accept_u == nullptr || logp_proposed == nullptr)
    return PTRWM_E_NULL;
  const long long sc = args->step0 + 1;  // step_counter of this step
  const bool count_on = sc > args->burn_in;
  const bool swap_due = args->n_temps > 1 && count_on && (sc % args->swap_every == 0);
  const bool ext = args->ext_prop != nullptr;
  if (swap_due && ext && args->ext_swap_u == nullptr) return PTRWM_E_NULL;
  SplitAcceptArgs a;
  a.state = args->state;
  a.logp = args->logp;
  a.proposals = proposals;
  a.beta = args->beta;
  a.accept_u = accept_u;
  a.logp_new = logp_proposed;
  a.n_accept = (long long *)args->n_accept;
  a.sq_jump = args->sq_jump;
  a.accept_flags = args->accept_flags;
  a.n_reps = args->n_chains * (long long)args->n_temps;
  a.n_temps = args->n_temps;
  a.dim = dim;
  a.count_on = count_on ? 1 : 0;
  a.swap_due = swap_due ? 1 : 0;
  a.device_step = (const long long *)args->device_step;
  a.step_offset = args->step0;
  a.burn_in = args->burn_in;
  a.swap_every = args->swap_every;
  {
    const unsigned lds = 2u * split_tile_lds_bytes(dim);
    if (lds > 48u * 1024u) {  // (dim > 95: above the default dynamic-LDS allowance; raised once per device)
      static unsigned long long raised_mask = 0;
      if (raise_dynamic_lds((const void *)split_accept_kernel, (const void *)split_accept_kernel,
                            (int)(2u * split_tile_lds_bytes(PTRWM_MAX_DIM)), raised_mask) != hipSuccess)
        return PTRWM_E_LAUNCH;
    }
    hipLaunchKernelGGL(split_accept_kernel, dim3((unsigned)((a.n_reps + 63) / 64)), dim3(64), lds, (hipStream_t)stream, a);
  }
  if (hipGetLastError() != hipSuccess) return PTRWM_E_LAUNCH;
  if (args->device_step != nullptr) {
    // device-step mode: the sweep is enqueued with every step and decides on the device whether its event is due - unless
    // the caller vouches that this step has none (PTRWM_SPLIT_NO_SWEEP)
    if (args->n_temps < 2 || (args->split_flags & PTRWM_SPLIT_NO_SWEEP) != 0) return PTRWM_OK;
    return launch_sweep(args, dim, 0, (int)kStreamSwap, proposals, args->sq_jump, (hipStream_t)stream);
  }
  if (!swap_due) return PTRWM_OK;
  // the swap event of this step: event number as ptrwm_run counts them, swap uniforms from the fused kernel's stream
  const long long ev = sc / args->swap_every - args->burn_in / args->swap_every - 1 + args->swap_event_offset;
  return launch_sweep(args, dim, ev, (int)kStreamSwap, proposals, args->sq_jump, (hipStream_t)stream);
}

int32_t ptrwm_split_advance(const ptrwm_run_args *args, void *stream) {
  if (args == nullptr) return PTRWM_E_NULL;
  if (args->struct_size != sizeof(ptrwm_run_args)) return PTRWM_E_STRUCT;
  if (args->device_step == nullptr) return PTRWM_E_NULL;
  hipLaunchKernelGGL(split_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (long long *)args->device_step,
                     args->n_steps > 0 ? (long long)args->n_steps : 1ll);
  return hipGetLastError() == hipSuccess ? PTRWM_OK : PTRWM_E_LAUNCH;
}

int32_t ptrwm_logdensity(const ptrwm_target_desc *target, const float *x, float *out, int64_t n, void *stream) {
  if (int rc = check_target(target)) return rc;
  if (n < 0) return PTRWM_E_ARG;
  if (n == 0) return PTRWM_OK;
  if (x == nullptr || out == nullptr) return PTRWM_E_NULL;
  const int dpi = width_index_for_dim(target->dim, target->kind);
  if (dpi < 0) return PTRWM_E_DIM;
  // (RoughCarpet: the three-term functor always - the two-term one has the same bits where it applies; ThreeMixture1: a
  // different summation order, so a target declared that way is evaluated that way everywhere)
  const LogpLaunchFn fn = target_variants(target->kind, target->kind == PTRWM_TARGET_THREE_MIXTURE && target->ip[0] == 1).logp(dpi);
  if (fn == nullptr) return PTRWM_E_NOVARIANT;
  const hipError_t err = fn(x, out, n, target->dim, make_tparams(target), (hipStream_t)stream);
  return err == hipSuccess ? PTRWM_OK : PTRWM_E_LAUNCH;
}

int32_t ptrwm_propose(const ptrwm_proposal_desc *proposal, int32_t dim, int32_t n_temps, int64_t n,
                      const float *ext_raw, uint64_t seed, float *out, void *stream) {
  if (proposal == nullptr) return PTRWM_E_NULL;
  if (proposal->kind < 0 || proposal->kind >= PTRWM_PROPOSAL_COUNT) return PTRWM_E_KIND;
  if (dim < 1 || dim > PTRWM_MAX_DIM) return PTRWM_E_DIM;
  if (n_temps < 1 || n_temps > PTRWM_MAX_TEMPS) return PTRWM_E_TEMPS;
  if (n < 0) return PTRWM_E_ARG;
  if (n == 0) return PTRWM_OK;
  if (out == nullptr || proposal->temp_scale == nullptr) return PTRWM_E_NULL;
  if (proposal->kind == PTRWM_PROPOSAL_LAPLACE && proposal->dim_scale == nullptr) return PTRWM_E_NULL;
  const int dpi = width_index_for_dim(dim);
  if (dpi < 0) return PTRWM_E_DIM;
  PParams pp;
  pp.dim_scale = proposal->dim_scale;
  pp.inv_dim = proposal->inv_dim;
  const int n_raw = ptrwm_ext_raw_per_step(proposal->kind, dim);
  const unsigned k0 = (unsigned)(seed & 0xffffffffull), k1 = (unsigned)(seed >> 32);
  hipError_t err;
  switch (proposal->kind) {
    case PTRWM_PROPOSAL_NORMAL:
      err = launch_propose<NormalProposal>(dpi, out, n, dim, n_temps, proposal->temp_scale, pp, ext_raw, n_raw, k0, k1, (hipStream_t)stream);
      break;
    case PTRWM_PROPOSAL_LAPLACE:
      err = launch_propose<LaplaceProposal>(dpi, out, n, dim, n_temps, proposal->temp_scale, pp, ext_raw, n_raw, k0, k1, (hipStream_t)stream);
      break;
    default:
      err = launch_propose<UniformRadiusProposal>(dpi, out, n, dim, n_temps, proposal->temp_scale, pp, ext_raw, n_raw, k0, k1, (hipStream_t)stream);
      break;
  }
  return err == hipSuccess ? PTRWM_OK : PTRWM_E_LAUNCH;
}

int32_t ptrwm_philox_raw(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, int64_t n, uint32_t *out,
                         void *stream) {
  if (n < 0) return PTRWM_E_ARG;
  if (n == 0) return PTRWM_OK;
  if (out == nullptr) return PTRWM_E_NULL;
  const unsigned grid = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(philox_raw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, (long long)n, c0, c1, c2,
                     c3, (unsigned)(seed & 0xffffffffull), (unsigned)(seed >> 32));
  return hipGetLastError() == hipSuccess ? PTRWM_OK : PTRWM_E_LAUNCH;
}

}  // extern "C"
