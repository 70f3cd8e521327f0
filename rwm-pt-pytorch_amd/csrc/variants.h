// Compiled-variant registry: (target, proposal, register width DP).
//
// dim is a run-time (wave-uniform) value; DP is the compile-time width of the
// per-thread register arrays, the smallest entry of kDP that is >= dim.
#pragma once
#include "kernel.h"
#include "quad.h"

namespace ptrwm {

// Exact widths: dims the reference's experiments actually run (data/*dim{2,3,4,5,10,20,30,50,100}*),
// compiled with dim as a constant.  Generic widths serve every other dim <= 104 with run-time
// predicates: at widths <= 32 they measure within a few per cent of an exact kernel (dim 29/31/32 vs 30), at
// widths 40-64 (2 waves/SIMD) about 25 % slower per dimension (dim 41/49 vs 50); a dim that matters can be added
// to the exact list at build time (profiles/r01_bench_variants.txt).  X(width, exact)
// Two groups, compiled into separate objects from the same source (csrc/Makefile): the NARROW widths (<= 64, at most
// 232 VGPRs) are built with the max-ILP scheduling strategy, the WIDE ones (80, 100, 104: more than 256 VGPRs, i.e.
// AGPR spilling next to the SGPR-spill lanes) with the compiler's default - under max-ILP hipcc miscompiled wide
// fixture kernels (garbage SGPR reloads: a wild ext_u address in RoughCarpet2/UniformRadius width 80, found by
// tools/fuzz_vs_oracle.py and pinned down with rocgdb).
#define PTRWM_WIDTHS_NARROW(X) \
  X(2, true) X(3, true) X(4, true) X(5, true) X(10, true) X(20, true) X(30, true) X(50, true) \
  X(8, false) X(16, false) X(24, false) X(32, false) X(40, false) X(48, false) X(56, false) X(64, false)
// (one generic width per canonical class above 64 - 80 / 96 / 104 for W = 20 / 24 / 28, philox.h canon_width - so that the
// register width a dim maps to always has the dim's own canonical range width)
#define PTRWM_WIDTHS_WIDE(X) X(100, true) X(80, false) X(96, false) X(104, false)
#define PTRWM_WIDTHS(X) PTRWM_WIDTHS_NARROW(X) PTRWM_WIDTHS_WIDE(X)
// Dims compiled in for ONE target only (round 4): the reference's HybridRosenbrock data uses dims 9, 19 and 29
// (data/HybridRosenbrock_*_dim{9,19,29}_*), which no other family runs.  They sit BEHIND the common widths in every table
// (so an index that was found without naming the target never reaches them), only the HybridRosenbrock translation units
// instantiate them (PTRWM_TU_EXTRA_DIMS, null entries everywhere else), and a lookup finds them only for that target.
#define PTRWM_WIDTHS_EXTRA(X) X(9, true) X(19, true) X(29, true)
constexpr int kExtraDimsTarget = PTRWM_TARGET_HYBRID_ROSENBROCK;

// The max-ILP group must stay at register widths <= 64 (at most 232 VGPRs measured): a wider entry belongs in
// PTRWM_WIDTHS_WIDE.  (tools/kernel_stats.py --check, run by the Makefile, checks the compiled VGPR / AGPR counts too.)
#define PTRWM_X_NARROW_OK(W, E) static_assert(W <= 64, "register widths above 64 belong in PTRWM_WIDTHS_WIDE (default scheduler)");
PTRWM_WIDTHS_NARROW(PTRWM_X_NARROW_OK)
#undef PTRWM_X_NARROW_OK
#define PTRWM_X_WIDE_OK(W, E) static_assert(W > 64 && W <= PTRWM_MAX_DIM && (E || W == 4 * canon_width(W) || W == PTRWM_MAX_DIM), \
                                            "PTRWM_WIDTHS_WIDE: widths 65..PTRWM_MAX_DIM, generic ones at the top of a canonical class");
PTRWM_WIDTHS_WIDE(PTRWM_X_WIDE_OK)
#undef PTRWM_X_WIDE_OK

struct WidthInfo {
  int dp;
  bool exact;
  int only_target;  // -1: every target; else the one target that has this entry (PTRWM_WIDTHS_EXTRA)
};
#define PTRWM_X_INFO(W, E) {W, E, -1},
#define PTRWM_X_INFO_EXTRA(W, E) {W, E, kExtraDimsTarget},
constexpr WidthInfo kWidths[] = {PTRWM_WIDTHS(PTRWM_X_INFO) PTRWM_WIDTHS_EXTRA(PTRWM_X_INFO_EXTRA)};
#undef PTRWM_X_INFO
#undef PTRWM_X_INFO_EXTRA
constexpr int kNumWidths = (int)(sizeof(kWidths) / sizeof(kWidths[0]));
#define PTRWM_X_EXTRA_OK(W, E) static_assert(E && W <= 64, "PTRWM_WIDTHS_EXTRA: dims compiled in, of the max-ILP group");
PTRWM_WIDTHS_EXTRA(PTRWM_X_EXTRA_OK)
#undef PTRWM_X_EXTRA_OK

// exact width if one matches (for this target: target_kind < 0 sees the common entries only), else the narrowest generic
// width >= dim; -1 if none
inline int width_index_for_dim(int dim, int target_kind = -1) {
  int best = -1;
  for (int i = 0; i < kNumWidths; ++i) {
    if (kWidths[i].only_target >= 0 && kWidths[i].only_target != target_kind) continue;
    if (kWidths[i].exact) {
      if (kWidths[i].dp == dim) return i;
    } else if (kWidths[i].dp >= dim && (best < 0 || kWidths[i].dp < kWidths[best].dp)) {
      best = i;
    }
  }
  return best;
}

// mode: which twin of the variant runs - the production kernel, the fixture / trace kernel (FULL), or the streaming form of
// the production kernel (kernel.h STREAM: persistent waves that prefetch their next group; sizes its own grid, `grid` is
// ignored; hipErrorNotSupported where the variant has no streaming twin)
enum { kRunProd = 0, kRunFull = 1, kRunStream = 2 };
using RunLaunchFn = hipError_t (*)(const KArgs &, unsigned grid, int mode, hipStream_t);
// Streaming twins exist for the one-thread-per-replica kernels with dim compiled in (the dims the reference's experiments
// run); every other shape takes the classic kernel whatever the launch length.
constexpr bool has_stream_variant(int dp, bool exact) { return exact && dp <= 64; }
using LogpLaunchFn = hipError_t (*)(const float *, float *, long long, int, const TParams &, hipStream_t);

struct TargetVariants {
  RunLaunchFn run[PTRWM_PROPOSAL_COUNT][kNumWidths];
  LogpLaunchFn logp[kNumWidths];
};

// Raises the dynamic-LDS allowance of a kernel pair (fixture + production variant) to `cap` bytes: once per kernel AND
// per device (the attribute belongs to the function object of the device that is current when it is set; a process
// that samples on cuda:0 and later on cuda:1 must raise it on both).  `raised_mask` is the caller's per-kernel static.
inline hipError_t raise_dynamic_lds(const void *kfull, const void *kprod, int cap, unsigned long long &raised_mask) {
  constexpr int kMaxDevices = 64;  // bit d: raised on device d (a race merely sets the attribute twice)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
  const bool known = dev >= 0 && dev < kMaxDevices;
  if (known && ((__atomic_load_n(&raised_mask, __ATOMIC_ACQUIRE) >> dev) & 1ull)) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(kfull, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  if (e == hipSuccess) e = hipFuncSetAttribute(kprod, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  if (e != hipSuccess) return e;
  if (known) __atomic_fetch_or(&raised_mask, 1ull << dev, __ATOMIC_RELEASE);
  return hipSuccess;
}

// compute units of the current device, asked once per device (0 on error)
inline int device_cus() {
  constexpr int kMaxDevices = 64;
  static int cached[kMaxDevices];  // 0 = not asked yet (a race merely asks twice)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 0;
  int n = __atomic_load_n(&cached[dev], __ATOMIC_RELAXED);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 0;
    __atomic_store_n(&cached[dev], n, __ATOMIC_RELAXED);
  }
  return n;
}

// The streaming twin: a grid sized to the device - as many workgroups as are resident at once (occupancy of THIS kernel on
// THIS device, asked once), trimmed so that every wave walks the same number of groups (+-1).
template <class Target, class Proposal, int DP, bool EXACT>
hipError_t launch_run_stream(const KArgs &a, hipStream_t stream) {
  if constexpr (!has_stream_variant(DP, EXACT)) {
    return hipErrorNotSupported;
  } else {
    if (a.n_temps > 64) return hipErrorNotSupported;
    auto ks = ptrwm_step_kernel<Target, Proposal, DP, EXACT, false, true>;
    const unsigned lds = step_kernel_lds_bytes(kBlockThreads, DP, true);
    constexpr int kMaxDevices = 64;
    static int wg_per_cu[kMaxDevices];  // 0 = not asked yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
    int per_cu = __atomic_load_n(&wg_per_cu[dev], __ATOMIC_RELAXED);
    if (per_cu == 0) {
      if (lds > 48u * 1024u) {
        const hipError_t e = hipFuncSetAttribute((const void *)ks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
      }
      const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ks, kBlockThreads, lds);
      if (e != hipSuccess) return e;
      if (per_cu < 1) return hipErrorInvalidConfiguration;
      __atomic_store_n(&wg_per_cu[dev], per_cu, __ATOMIC_RELAXED);
    }
    const int cus = device_cus();
    if (cus <= 0) return hipErrorInvalidDevice;
    const long long n_groups = a.n_chains / a.chains_per_wave;
    const long long max_waves = (long long)cus * per_cu * kWavesPerBlock;
    const long long rounds = (n_groups + max_waves - 1) / max_waves;
    const long long waves = (n_groups + rounds - 1) / rounds;
    const unsigned grid = (unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(ks, dim3(grid), dim3(kBlockThreads), lds, stream, a);
    return hipGetLastError();
  }
}

template <class Target, class Proposal, int DP, bool EXACT>
hipError_t launch_run(const KArgs &a, unsigned grid, int mode, hipStream_t stream) {
  if (mode == kRunStream) return launch_run_stream<Target, Proposal, DP, EXACT>(a, stream);
  const bool full = mode == kRunFull;
  // narrow ladders: four independent one-wave groups per workgroup; wide ones (n_temps > 64): as many waves as
  // the ladder needs
  const unsigned block = a.n_temps > 64 ? (unsigned)((a.n_temps + 63) & ~63) : (unsigned)kBlockThreads;
  const unsigned lds = step_kernel_lds_bytes((int)block, DP) + (a.n_temps > 64 ? kWideVoteBytes : 0u);
  auto kfull = ptrwm_step_kernel<Target, Proposal, DP, EXACT, true>;
  auto kprod = ptrwm_step_kernel<Target, Proposal, DP, EXACT, false>;
  if (lds > 48u * 1024u) {  // above the default dynamic-LDS allowance (wide ladders at large dim)
    static unsigned long long raised_mask = 0;
    const hipError_t e = raise_dynamic_lds((const void *)kfull, (const void *)kprod,
                                           (int)(step_kernel_lds_bytes(kBlockThreads, DP) + kWideVoteBytes), raised_mask);
    if (e != hipSuccess) return e;
  }
  if (full)
    hipLaunchKernelGGL(kfull, dim3(grid), dim3(block), lds, stream, a);
  else
    hipLaunchKernelGGL(kprod, dim3(grid), dim3(block), lds, stream, a);
  return hipGetLastError();
}

template <class Target, int DP>
hipError_t launch_logp(const float *x, float *out, long long n, int D, const TParams &tp, hipStream_t stream) {
  const unsigned grid = (unsigned)((n + kBlockThreads - 1) / kBlockThreads);
  hipLaunchKernelGGL((ptrwm_logdensity_kernel<Target, DP>), dim3(grid), dim3(kBlockThreads), 0, stream, x, out,
                     n, D, tp);
  return hipGetLastError();
}

// ---- lane-split ("quad") variants (quad.h): X(lane width W, dim compiled in or 0 for a run-time dim, max threads) ----
// W is the canonical range width of the dim class (8 / 16 / 20 / 24 / 28 for dim <= 32 / 64 / 80 / 96 / 112); the BASELINE
// dims get a kernel with dim compiled in, every other dim runs the generic kernel of its class.  The classes above dim 64
// (W = 20, 24, 28) are the only form of the fused kernel there (see PTRWM_WIDTHS_WIDE above).
// (Round 4: the 1024-thread class of the dim > 64 kernels - ladders of 129..256 temperatures there - is retired: at the
// 128 VGPRs a 1024-thread workgroup allows, the W = 28 kernels spilled 41-42 VGPRs into scratch inside the step loop and
// a dozen more sat above the spilled-SGPR ceiling; a ladder that long at dim > 64 now gets PTRWM_E_NOVARIANT instead of a
// kernel nobody measured fast.  288 kernels less to build.)
#define PTRWM_QUAD_WIDTHS(X)                                                                              \
  X(8, 0, kQuadThreads) X(8, 20, kQuadThreads) X(8, 30, kQuadThreads) X(16, 0, kQuadThreads) X(16, 50, kQuadThreads) \
  X(20, 0, kQuadThreads) X(24, 0, kQuadThreads) X(28, 0, kQuadThreads) X(28, 100, kQuadThreads)
// (the lane-split twins of PTRWM_WIDTHS_EXTRA: one target only, behind the common entries)
#define PTRWM_QUAD_WIDTHS_EXTRA(X) X(8, 9, kQuadThreads) X(8, 19, kQuadThreads) X(8, 29, kQuadThreads)
struct QuadWidthInfo {
  int w, dexact, max_threads;
  int only_target;  // -1: every target
};
#define PTRWM_X_QINFO(W, E, M) {W, E, M, -1},
#define PTRWM_X_QINFO_EXTRA(W, E, M) {W, E, M, kExtraDimsTarget},
constexpr QuadWidthInfo kQuadWidths[] = {PTRWM_QUAD_WIDTHS(PTRWM_X_QINFO) PTRWM_QUAD_WIDTHS_EXTRA(PTRWM_X_QINFO_EXTRA)};
#undef PTRWM_X_QINFO
#undef PTRWM_X_QINFO_EXTRA
constexpr int kNumQuadWidths = (int)(sizeof(kQuadWidths) / sizeof(kQuadWidths[0]));
#define PTRWM_X_QOK(W, E, M) static_assert(W == canon_width(4 * W) && (E == 0 || (E <= 4 * W && canon_width(E) == W)), "quad width table");
PTRWM_QUAD_WIDTHS(PTRWM_X_QOK)
PTRWM_QUAD_WIDTHS_EXTRA(PTRWM_X_QOK)
#undef PTRWM_X_QOK

// Exchange groups of the lane-split form.  4 T <= 64: one wavefront holds 16 / T whole ladders, four such groups per
// 256-thread workgroup.  Longer ladders: the group is the workgroup; it holds as many whole ladders as make the best use
// of its lanes within 256 threads (T = 17: three ladders in 204 of 256 lanes instead of one in 68 of 128), one ladder
// in 4 T threads rounded up to whole waves when even one does not fit.
inline int quad_ladders_per_group(int n_temps) {
  const int need = 4 * n_temps;
  if (need <= 64) return 64 / need;
  int best_k = 1;
  double best_use = 0.0;
  for (int k = 1; k * need <= kBlockThreads; ++k) {
    const int b = (k * need + 63) & ~63;
    const double use = (double)(k * need) / b;
    if (use > best_use + 1e-9) {
      best_use = use;
      best_k = k;
    }
  }
  return best_k;
}
inline int quad_block_threads(int n_temps) {
  const int need = 4 * n_temps;
  return need <= 64 ? kBlockThreads : ((quad_ladders_per_group(n_temps) * need + 63) & ~63);
}

// the kernel with this dim compiled in if there is one, else the generic kernel of the dim's class, in the smallest
// workgroup class that holds the ladder; -1 if none
inline int quad_index_for(int dim, int n_temps, int target_kind = -1) {
  if (dim < 1 || dim > PTRWM_MAX_DIM || n_temps < 1) return -1;
  const int threads = quad_block_threads(n_temps);
  int best = -1;
  for (int i = 0; i < kNumQuadWidths; ++i) {
    const QuadWidthInfo &q = kQuadWidths[i];
    if (q.only_target >= 0 && q.only_target != target_kind) continue;
    if (q.w != canon_width(dim) || q.max_threads < threads || (q.dexact != 0 && q.dexact != dim)) continue;
    if (best < 0) {
      best = i;
      continue;
    }
    const QuadWidthInfo &b = kQuadWidths[best];
    // prefer the smaller workgroup class, then the kernel with dim compiled in
    if (q.max_threads < b.max_threads || (q.max_threads == b.max_threads && q.dexact != 0 && b.dexact == 0)) best = i;
  }
  return best;
}

struct QuadVariants {
  RunLaunchFn run[PTRWM_PROPOSAL_COUNT][kNumQuadWidths];
  RunLaunchFn run_f64[PTRWM_PROPOSAL_COUNT][kNumQuadWidths];  // state_f64 twins
};

// MIN_OWN of quad.h: the number of dimensions the LAST lane owns when dim is compiled in (every lane owns at least
// that); -1 marks the kernels that take dim at run time
constexpr int quad_min_own(int w, int dexact) {
  return dexact == 0 ? -1 : (dexact - 3 * w <= 0 ? 0 : (dexact - 3 * w > w ? w : dexact - 3 * w));
}

template <class Target, class Proposal, int W, int DEXACT, int MAXT, bool F64 = false>
hipError_t launch_run_quad(const KArgs &a, unsigned grid, int mode, hipStream_t stream) {
  if (mode == kRunStream) return hipErrorNotSupported;  // the lane-split form has no streaming twin
  const bool full = mode == kRunFull;
  // narrow ladders (4 T <= 64): four independent one-wave groups per workgroup; wide ones: one ladder per workgroup
  const unsigned block = (unsigned)quad_block_threads(a.n_temps);
  if ((int)block > MAXT) return hipErrorInvalidConfiguration;
  const unsigned lds = quad_kernel_lds_bytes((int)block, W, F64);
  auto kfull = ptrwm_quad_step_kernel<Target, Proposal, W, DEXACT, MAXT, true, F64>;
  auto kprod = ptrwm_quad_step_kernel<Target, Proposal, W, DEXACT, MAXT, false, F64>;
  if (lds > 48u * 1024u) {
    static unsigned long long raised_mask = 0;
    const hipError_t e = raise_dynamic_lds((const void *)kfull, (const void *)kprod,
                                           (int)quad_kernel_lds_bytes(MAXT, W, F64), raised_mask);
    if (e != hipSuccess) return e;
  }
  if (full)
    hipLaunchKernelGGL(kfull, dim3(grid), dim3(block), lds, stream, a);
  else
    hipLaunchKernelGGL(kprod, dim3(grid), dim3(block), lds, stream, a);
  return hipGetLastError();
}

#define PTRWM_X_QRUN_N(W, E, M) launch_run_quad<QTGT<W, quad_min_own(W, E)>, QNormal<W, quad_min_own(W, E)>, W, E, M>,
#define PTRWM_X_QRUN_L(W, E, M) launch_run_quad<QTGT<W, quad_min_own(W, E)>, QLaplace<W, quad_min_own(W, E)>, W, E, M>,
#define PTRWM_X_QRUN_U(W, E, M) launch_run_quad<QTGT<W, quad_min_own(W, E)>, QUniformRadius<W, quad_min_own(W, E)>, W, E, M>,
// the state_f64 twin of a variant (include/ptrwm.h): workgroups of up to 512 threads only - twice the state registers
// do not fit the 128-VGPR budget of a 1024-thread workgroup
template <class Target, class Proposal, int W, int DEXACT, int MAXT>
constexpr RunLaunchFn quad_f64_entry() {
  if constexpr (MAXT > kQuadThreads) return nullptr;
  else return launch_run_quad<Target, Proposal, W, DEXACT, MAXT, true>;
}
#define PTRWM_X_QRUN64_N(W, E, M) quad_f64_entry<QTGT<W, quad_min_own(W, E)>, QNormal<W, quad_min_own(W, E)>, W, E, M>(),
#define PTRWM_X_QRUN64_L(W, E, M) quad_f64_entry<QTGT<W, quad_min_own(W, E)>, QLaplace<W, quad_min_own(W, E)>, W, E, M>(),
#define PTRWM_X_QRUN64_U(W, E, M) quad_f64_entry<QTGT<W, quad_min_own(W, E)>, QUniformRadius<W, quad_min_own(W, E)>, W, E, M>(),
// (the one-target entries: real kernels in the translation unit that defines PTRWM_TU_EXTRA_DIMS, null in every other)
#define PTRWM_X_QNULL(W, E, M) nullptr,
#ifdef PTRWM_TU_EXTRA_DIMS
#define PTRWM_QUAD_EXTRA_ROW(X) PTRWM_QUAD_WIDTHS_EXTRA(X)
#define PTRWM_EXTRA_ROW(X) PTRWM_WIDTHS_EXTRA(X)
#else
#define PTRWM_QUAD_EXTRA_ROW(X) PTRWM_QUAD_WIDTHS_EXTRA(PTRWM_X_QNULL)
#define PTRWM_EXTRA_ROW(X) PTRWM_WIDTHS_EXTRA(PTRWM_X_NULL)
#endif
// One translation unit per target (csrc/quad_<target>.hip) defines its table with this macro.
#define PTRWM_DEFINE_QUAD_VARIANTS(SYMBOL, QTARGET)                        \
  template <int W, int M>                                                  \
  using QTGT = QTARGET<W, M>;                                              \
  const QuadVariants &SYMBOL##_quad() {                                    \
    static const QuadVariants v = {{{PTRWM_QUAD_WIDTHS(PTRWM_X_QRUN_N) PTRWM_QUAD_EXTRA_ROW(PTRWM_X_QRUN_N)},     \
                                    {PTRWM_QUAD_WIDTHS(PTRWM_X_QRUN_L) PTRWM_QUAD_EXTRA_ROW(PTRWM_X_QRUN_L)},     \
                                    {PTRWM_QUAD_WIDTHS(PTRWM_X_QRUN_U) PTRWM_QUAD_EXTRA_ROW(PTRWM_X_QRUN_U)}},    \
                                   {{PTRWM_QUAD_WIDTHS(PTRWM_X_QRUN64_N) PTRWM_QUAD_EXTRA_ROW(PTRWM_X_QRUN64_N)}, \
                                    {PTRWM_QUAD_WIDTHS(PTRWM_X_QRUN64_L) PTRWM_QUAD_EXTRA_ROW(PTRWM_X_QRUN64_L)}, \
                                    {PTRWM_QUAD_WIDTHS(PTRWM_X_QRUN64_U) PTRWM_QUAD_EXTRA_ROW(PTRWM_X_QRUN64_U)}}}; \
    return v;                                                              \
  }

// One translation unit per target (compiled in parallel) defines its table with this macro.
// The table lives inside a host function so the device pass does not try to emit it.
#define PTRWM_X_RUN_N(W, E) launch_run<TGT<W>, NormalProposal<W>, W, E>,
#define PTRWM_X_RUN_L(W, E) launch_run<TGT<W>, LaplaceProposal<W>, W, E>,
#define PTRWM_X_RUN_U(W, E) launch_run<TGT<W>, UniformRadiusProposal<W>, W, E>,
#define PTRWM_X_LOGP(W, E) launch_logp<TGT<W>, W>,
#define PTRWM_X_NULL(W, E) nullptr,
#ifdef PTRWM_PART_WIDE
#define PTRWM_PART_SUFFIX(SYMBOL) SYMBOL##_wide
#define PTRWM_PART_ROW(X) PTRWM_WIDTHS_NARROW(PTRWM_X_NULL) PTRWM_WIDTHS_WIDE(X) PTRWM_WIDTHS_EXTRA(PTRWM_X_NULL)
// No one-thread-per-replica STEP kernels above width 64 (they needed 340-420 VGPRs, i.e. AGPR copies next to ~150 spilled
// SGPRs: the regime in which hipcc produced wrong code twice - round 1 under max-ILP scheduling, round 2 under the default
// scheduler after a fence moved; profiles/r02_miscompile_width80.txt): dim > 64 always runs the lane-split kernel
// (<= 256 VGPRs, no AGPRs).  The wide objects keep the stand-alone log-density kernels (121 VGPRs).
#undef PTRWM_X_RUN_N
#undef PTRWM_X_RUN_L
#undef PTRWM_X_RUN_U
#define PTRWM_X_RUN_N(W, E) nullptr,
#define PTRWM_X_RUN_L(W, E) nullptr,
#define PTRWM_X_RUN_U(W, E) nullptr,
#else
#define PTRWM_PART_SUFFIX(SYMBOL) SYMBOL##_narrow
#define PTRWM_PART_ROW(X) PTRWM_WIDTHS_NARROW(X) PTRWM_WIDTHS_WIDE(PTRWM_X_NULL) PTRWM_EXTRA_ROW(X)
#endif
// Each object defines SYMBOL_narrow() or SYMBOL_wide(): a full-size table with null entries for the other group.
#define PTRWM_DEFINE_TARGET_VARIANTS(SYMBOL, TARGET)                                   \
  template <int W>                                                                     \
  using TGT = TARGET<W>;                                                               \
  const TargetVariants &PTRWM_PART_SUFFIX(SYMBOL)() {                                  \
    static const TargetVariants v = {{{PTRWM_PART_ROW(PTRWM_X_RUN_N)},                 \
                                      {PTRWM_PART_ROW(PTRWM_X_RUN_L)},                 \
                                      {PTRWM_PART_ROW(PTRWM_X_RUN_U)}},                \
                                     {PTRWM_PART_ROW(PTRWM_X_LOGP)}};                  \
    return v;                                                                          \
  }

#define PTRWM_DECLARE_TARGET_VARIANTS(SYMBOL) \
  const TargetVariants &SYMBOL##_narrow();    \
  const TargetVariants &SYMBOL##_wide();      \
  const QuadVariants &SYMBOL##_quad();
PTRWM_DECLARE_TARGET_VARIANTS(rough_carpet_variants)
PTRWM_DECLARE_TARGET_VARIANTS(rough_carpet2_variants)  // two-term specialisation, see targets.h
PTRWM_DECLARE_TARGET_VARIANTS(three_mixture_variants)
PTRWM_DECLARE_TARGET_VARIANTS(three_mixture1_variants)  // means differing in the first coordinate only, see targets.h
PTRWM_DECLARE_TARGET_VARIANTS(full_rosenbrock_variants)
PTRWM_DECLARE_TARGET_VARIANTS(even_rosenbrock_variants)
PTRWM_DECLARE_TARGET_VARIANTS(hybrid_rosenbrock_variants)
PTRWM_DECLARE_TARGET_VARIANTS(iid_gamma_variants)
PTRWM_DECLARE_TARGET_VARIANTS(iid_beta_variants)
PTRWM_DECLARE_TARGET_VARIANTS(diag_gaussian_variants)
PTRWM_DECLARE_TARGET_VARIANTS(hypercube_variants)
PTRWM_DECLARE_TARGET_VARIANTS(neal_funnel_variants)

}  // namespace ptrwm
