/*
 * ptrwm.h -- C ABI of the MI355X (gfx950) PT-RWM sampling engine.
 *
 * This is the drop-in boundary for the per-step hot loop of the reference
 * (aidanmrli/rwm-pt-pytorch):
 *
 *   algorithms/rwm_gpu_optimized.py:289-336   _single_step_ultra_fused
 *   algorithms/rwm_gpu_optimized.py:402-488   generate_samples (the `for i in range(total_steps)` loop)
 *   algorithms/pt_rwm_gpu_optimized.py:541-574 step
 *   algorithms/pt_rwm_gpu_optimized.py:594-633 _attempt_all_swaps
 *   algorithms/pt_rwm_gpu_optimized.py:694-770 generate_samples
 *
 * The reference has no FFI of its own (it is pure Python on torch tensors); the
 * binding a maintainer adds is the ctypes stub shown in INTEGRATION.md.  All
 * pointers named "device" are raw HIP device pointers (e.g. torch
 * `tensor.data_ptr()` on a ROCm build); `stream` is a `hipStream_t` passed as
 * `void*` (torch: `torch.cuda.current_stream().cuda_stream`).  Nothing here
 * retains a pointer past the call, allocates device memory, or synchronises:
 * every entry point only enqueues kernels on `stream`.
 *
 * Device contract (the usual HIP one): the device that owns `stream` and every
 * pointer must be the CURRENT device of the calling thread (hipSetDevice); the
 * library never switches devices.  The Python binding does this around every
 * call (ptrwm_hip.on_device), so `device="cuda:1"` works without set_device.
 *
 * All entry points return 0 on success or a negative PTRWM_E_* code; no C++
 * exception crosses this boundary.
 */
#ifndef PTRWM_H
#define PTRWM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTRWM_ABI_VERSION 3
#define PTRWM_SPLIT_NO_SWEEP 1 /* ptrwm_run_args.split_flags, device-step mode: do not enqueue the swap kernel for this step */
#define PTRWM_MAX_DIM 104  /* dim-vector lives in VGPRs; widest compiled variant */
#define PTRWM_MAX_TEMPS 256 /* one ladder lives in one wavefront (<= 64 temps) or one workgroup; above dim 64 (lane-split
                              kernel only, 512-thread workgroups): at most 128, longer ladders get PTRWM_E_NOVARIANT */

/* ---- status codes ------------------------------------------------------ */
enum {
  PTRWM_OK = 0,
  PTRWM_E_NULL = -1,        /* required pointer is NULL */
  PTRWM_E_DIM = -2,         /* dim outside [1, PTRWM_MAX_DIM] or invalid for the target */
  PTRWM_E_TEMPS = -3,       /* n_temps outside [1, PTRWM_MAX_TEMPS] */
  PTRWM_E_KIND = -4,        /* unknown target / proposal kind */
  PTRWM_E_ARG = -5,         /* other invalid argument (negative counts, swap_every < 1 ...) */
  PTRWM_E_STRUCT = -6,      /* struct_size does not match this ABI version */
  PTRWM_E_LAUNCH = -7,      /* HIP reported a launch error */
  PTRWM_E_NOVARIANT = -8    /* (target, proposal, dim) variant not compiled into this build */
};

/* ---- target densities ---------------------------------------------------
 * Each kind restates the fp32 `log_density` of one reference class. */
enum {
  /* target_distributions/multimodal_torch.py:470-510 RoughCarpetDistributionTorch
   *   p[0..2] = modes, p[3..5] = log weights, p[6] = log_jacobian (0 if unscaled)
   *   vec0 = scaling_factors[dim] or NULL */
  PTRWM_TARGET_ROUGH_CARPET = 0,
  /* multimodal_torch.py:173-242 ThreeMixtureDistributionTorch (cov = I)
   *   p[0..2] = log_norm_const_k + log_mixing_weight_k (+ log_jacobian if scaled)
   *   vec0 = means[3*dim] (row-major [3][dim]); vec1 = scaling_factors[dim] or NULL
   *   ip[0] = 1: the caller DECLARES that the three mean vectors are equal in every coordinate but the first (the
   *   class's default centres (-5,0,..), (0,..), (5,0,..) and the +-15 centres of experiment_pt_GPU.py:48-52 are): the
   *   kernel then evaluates the part of |s x - mu_k|^2 the components share once instead of three times (a third of the
   *   work, same tolerance).  ip[0] = 0: no assumption.  A false declaration gives a wrong density. */
  PTRWM_TARGET_THREE_MIXTURE = 1,
  /* rosenbrock_torch.py:67-84 FullRosenbrockTorch: p[0]=a, p[1]=b, vec0 = mu[dim-1] */
  PTRWM_TARGET_FULL_ROSENBROCK = 2,
  /* rosenbrock_torch.py:194-210 EvenRosenbrockTorch: p[0]=a, p[1]=b, vec0 = mu[dim/2] */
  PTRWM_TARGET_EVEN_ROSENBROCK = 3,
  /* rosenbrock_torch.py:312-351 HybridRosenbrockTorch: p[0]=a, p[1]=b, p[2]=mu,
   *   ip[0]=n1, ip[1]=n2, dim = 1 + n2*(n1-1) */
  PTRWM_TARGET_HYBRID_ROSENBROCK = 4,
  /* iid_product_torch.py:52-91 IIDGammaTorch: p[0]=shape, p[1]=scale, p[2]=log_norm_const (dim * 1d) */
  PTRWM_TARGET_IID_GAMMA = 5,
  /* iid_product_torch.py:188-229 IIDBetaTorch: p[0]=alpha, p[1]=beta, p[2]=log_norm_const (dim * 1d) */
  PTRWM_TARGET_IID_BETA = 6,
  /* Gaussians with diagonal structure:
   *   ip[0] = 0: multivariate_normal_torch.py:62-92 MultivariateNormalTorch with a DIAGONAL covariance:
   *              -0.5 sum_d vec1[d] (x_d - vec0[d])^2 + p[0];  vec0 = mean, vec1 = diag(cov_inv), p[0] = log_norm_const
   *   ip[0] = 1: multivariate_normal_torch.py:199-224 ScaledMultivariateNormalTorch:
   *              p[0] - 0.5 sum_d (vec0[d] x_d)^2;  vec0 = scaling_factors, p[0] = log_norm_const */
  PTRWM_TARGET_DIAG_GAUSSIAN = 7,
  /* hypercube_torch.py:49-78 HypercubeTorch: p[0] = left, p[1] = right, p[2] = log uniform density; -inf outside */
  PTRWM_TARGET_HYPERCUBE = 8,
  /* funnel_torch.py:39-76 NealFunnelTorch: p[0] = mu_v, p[1] = sigma_v^2, p[2] = mu_z; x[0] = v, x[1..] = z */
  PTRWM_TARGET_NEAL_FUNNEL = 9,
  PTRWM_TARGET_COUNT = 10
};

typedef struct ptrwm_target_desc {
  int32_t kind;
  int32_t dim;
  float p[12];
  int32_t ip[4];
  const float *vec0; /* device, see kind */
  const float *vec1; /* device, see kind */
} ptrwm_target_desc;

/* ---- proposal increments -------------------------------------------------
 * increment[d] for the replica at temperature t. */
enum {
  /* proposal_distributions/normal.py:33-36,46-55 and pt_rwm_gpu_optimized.py:445-455,576-592
   *   inc_d = temp_scale[t] * z_d,  z ~ N(0,1) */
  PTRWM_PROPOSAL_NORMAL = 0,
  /* proposal_distributions/laplace.py:24-37,46-69
   *   u = U[0,1) - 0.5;  inc_d = -(dim_scale[d]*temp_scale[t]) * sign(u) * log1p(max(-2|u|, -0.999999)) */
  PTRWM_PROPOSAL_LAPLACE = 1,
  /* proposal_distributions/uniform.py:27-37,47-73
   *   g = N(0,I_dim); n = |g| (1 if <= 1e-12); inc = g/n * temp_scale[t] * U^{inv_dim} */
  PTRWM_PROPOSAL_UNIFORM_RADIUS = 2,
  PTRWM_PROPOSAL_COUNT = 3
};

typedef struct ptrwm_proposal_desc {
  int32_t kind;
  float inv_dim;           /* UNIFORM_RADIUS: 1/dim as the reference stores it */
  const float *temp_scale; /* device [n_temps] */
  const float *dim_scale;  /* device [dim], LAPLACE only (NULL otherwise) */
} ptrwm_proposal_desc;

/* ---- swap semantics (pt_rwm_gpu_optimized.py:594-633) ---- */
enum {
  PTRWM_SWAP_EXCHANGE = 0,       /* rows j and k trade places (algorithms/pt_rwm.py:141-150) */
  PTRWM_SWAP_REFERENCE_COPY = 1  /* row j <- row k, row k unchanged: what
                                    fused_swap_execution_no_clone (pt_rwm_gpu_optimized.py:51-59) does */
};
enum {
  PTRWM_ORDER_SEQUENTIAL = 0, /* j = 0..T-2 in order, each sees the previous outcome (reference) */
  PTRWM_ORDER_EVEN_ODD = 1    /* n-th swap event (0-based) attempts the disjoint pairs j == n (mod 2) */
};

/* ---- form of the fused step kernel ---------------------------------------
 * ptrwm_run has two bit-identical implementations of the same loop (same Philox words, same per-dimension arithmetic,
 * every sum over dimensions in one canonical order): one thread per (chain, temperature) replica, and a lane-split
 * form with four lanes per replica for launches that would otherwise under-fill the GPU (fewer than two wavefronts
 * per SIMD); above dim 64 only the lane-split form exists.  AUTO picks by batch size and dim; the choice cannot change
 * a result, only the speed.  ptrwm_set_kernel_form pins it process-wide (tests, tuning) and returns the previous
 * value, or PTRWM_E_ARG. */
enum {
  PTRWM_FORM_AUTO = 0,
  PTRWM_FORM_THREAD = 1, /* one thread per replica wherever that variant exists (dim <= 64) */
  PTRWM_FORM_QUAD = 2    /* lane-split wherever that variant exists (dim <= 64: n_temps <= 128; dim > 64: all) */
};
int32_t ptrwm_set_kernel_form(int32_t form);
/* 1 if ptrwm_run has a lane-split variant for (target, proposal, dim, n_temps), else 0. */
int32_t ptrwm_has_quad_variant(int32_t target_kind, int32_t proposal_kind, int32_t dim, int32_t n_temps);
/* 1 if ptrwm_run has a one-thread-per-replica variant for (target, proposal, dim), else 0 (never above dim 64).  Where it
 * returns 0 PTRWM_FORM_THREAD runs the lane-split kernel: a comparison of the two forms is vacuous there. */
int32_t ptrwm_has_thread_variant(int32_t target_kind, int32_t proposal_kind, int32_t dim);
/* The form PTRWM_FORM_AUTO runs for a float-state launch of this shape on the current device (PTRWM_FORM_THREAD or
 * PTRWM_FORM_QUAD), or a negative status.  Introspection only: the forms give the same bits. */
int32_t ptrwm_auto_form(int32_t target_kind, int32_t proposal_kind, int32_t dim, int32_t n_temps, int64_t n_chains);
/* The same rule for a device of n_simds SIMDs (no HIP call: a pure function of its arguments and of the fitted table). */
int32_t ptrwm_auto_form_for(int32_t target_kind, int32_t proposal_kind, int32_t dim, int32_t n_temps, int64_t n_chains,
                            int32_t n_simds);
/* SIMDs (compute units x 4) of the device that owns `stream` (NULL: the calling thread's current device): what the AUTO
 * rule of ptrwm_run scales by; PTRWM_E_LAUNCH if the runtime cannot say (AUTO then keeps the thread form). */
int32_t ptrwm_device_simds(void *stream);
/* Identity of the build as far as kernel speed goes: sha256 over the kernel sources (tools/source_hash.py), and the hash
 * of the sources the AUTO form table (csrc/form_table.inc) was fitted on.  Different strings: the table is stale - the
 * forms still give the same bits, AUTO may just not pick the faster one - re-fit with tools/form_sweep.py + form_fit.py. */
const char *ptrwm_source_hash(void);
const char *ptrwm_form_table_source_hash(void);

/* ---- short launches ------------------------------------------------------
 * A launch of one step over a large batch (the reference's step()-at-a-time loops, rwm_gpu_optimized.py:456-457,
 * pt_rwm_gpu_optimized.py:736-737) is bound by memory traffic; for it ptrwm_run has a STREAMING form of the
 * one-thread-per-replica kernel: persistent wavefronts that walk the batch with the next group's state already in flight
 * while the current one is stepped and the previous one's results drain.  Same Philox words and arithmetic: the same bits
 * as the classic kernel.  AUTO takes it for one-step launches whose arrays total about the size of the Infinity Cache
 * (192-448 MiB: where it measured faster, csrc/capi.hip), where the variant has a streaming twin (dim compiled in,
 * n_temps <= 64) and the batch's layout allows whole aligned 16-byte vectors per group; OFF never; ON wherever twin and
 * layout allow (tests, tuning).  Process-wide; returns the previous value, or PTRWM_E_ARG. */
enum {
  PTRWM_STREAM_AUTO = 0,
  PTRWM_STREAM_OFF = 1,
  PTRWM_STREAM_ON = 2
};
int32_t ptrwm_set_stream_mode(int32_t mode);
/* 1 if the one-thread-per-replica variant for (target, proposal, dim) has a streaming twin, else 0. */
int32_t ptrwm_has_stream_variant(int32_t target_kind, int32_t proposal_kind, int32_t dim);

/* Which kernel the calling thread's most recent successful ptrwm_run enqueued (introspection for tests and benchmark
 * records; 0 before the first call). */
enum {
  PTRWM_LAUNCH_THREAD = 1, /* one thread per replica, classic form */
  PTRWM_LAUNCH_QUAD = 2,   /* lane-split form */
  PTRWM_LAUNCH_STREAM = 3  /* one thread per replica, streaming form */
};
int32_t ptrwm_last_launch_kind(void);

/* Number of raw random numbers one MH proposal consumes from `ext_prop`
 * (NORMAL: dim normals; LAPLACE: dim uniforms in [0,1); UNIFORM_RADIUS: dim
 * normals then one uniform). */
int32_t ptrwm_ext_raw_per_step(int32_t proposal_kind, int32_t dim);

typedef struct ptrwm_run_args {
  uint32_t struct_size; /* sizeof(ptrwm_run_args) */
  int32_t n_temps;
  int64_t n_chains;     /* independent ladders (RWM: independent chains) on this device */
  int64_t chain_offset; /* global id of local chain 0: the Philox subsequence, so results do not
                           depend on how chains are sharded over devices */
  /* state, device, updated in place */
  float *state; /* [n_chains, n_temps, dim] */
  float *logp;  /* [n_chains, n_temps] log-density of `state` */
  const float *beta; /* [n_temps] inverse temperatures, beta[0] is the cold chain */
  /* statistics, device, accumulated (+=); any may be NULL */
  int64_t *n_accept;    /* [n_chains, n_temps] MH acceptances at steps with step_counter > burn_in */
  double *sq_jump;      /* [n_chains, n_temps] sum of |x_t - x_{t-1}|^2 over those steps (swap moves included).  For an
                         * accepted Metropolis move of the Normal / UniformRadius proposals in Philox mode this is the
                         * squared length of the increment itself, which equals that of the float sum x + inc to ~3e-5
                         * relative or better wherever it is used: a replica whose largest |coordinate| exceeds 256
                         * typical increments when a launch loads it takes the jump from the states instead.
                         * That choice is made once per LAUNCH, from the state the launch starts with: state, logp and
                         * every counter are independent of how a run is cut into launches, sq_jump is too unless a
                         * replica crosses that bound in the middle of a launch - then the two cuts differ in the last
                         * bits of that replica's sum (both within the ~3e-5 above; the oracle follows the same rule
                         * per launch, tests/test_gpu_engine_parity.py test_squared_jump_across_the_trust_boundary). */
  int64_t *swap_accept; /* [n_chains, n_temps] accepted swaps of pair (t, t+1); column n_temps-1 unused */
  int64_t *last_swap_ordinal; /* [n_chains, n_temps] max 1-based attempt ordinal at which pair t accepted */
  /* schedule: this call performs steps step0 .. step0+n_steps-1 (0-based); step i has
   * step_counter = i+1.  Swaps happen after the MH move of a step when
   * step_counter % swap_every == 0 and step_counter > burn_in (pt_rwm_gpu_optimized.py:544,570). */
  int64_t step0;
  int64_t n_steps;
  int64_t burn_in;
  int32_t swap_every;
  int32_t swap_mode;
  int32_t swap_order;
  int32_t swap_event_offset; /* swap events performed outside ptrwm_run (ptrwm_swap_sweep) before this call: added to
                                the event numbers this call derives from step0 (attempt ordinals, even/odd parity) */
  uint64_t seed; /* Philox4x32-10 key */
  /* external randoms (test / fixture mode); all NULL => in-kernel Philox */
  const float *ext_prop;   /* [n_steps, n_chains, n_temps, ptrwm_ext_raw_per_step()] */
  const float *ext_u;      /* [n_steps, n_chains, n_temps] accept uniforms */
  const float *ext_swap_u; /* [n_swap_events_in_call, n_chains, n_temps-1] */
  /* optional per-step outputs */
  float *trace;        /* [trace_rows, trace_chains, trace_temps, dim] state after each traced step of this call */
  float *trace_logp;   /* [trace_rows, trace_chains, trace_temps] */
  int64_t trace_chains; /* first trace_chains local chains are traced */
  int32_t trace_temps;  /* first trace_temps temperatures are traced (1 = cold chain only) */
  int32_t trace_every;  /* thinning: a step is traced when step_counter % trace_every == 0 (0 or 1 = every step) */
  int64_t trace_row0;   /* row written by the first traced step of this call */
  uint8_t *accept_flags; /* [n_steps, n_chains, n_temps] MH accept decision of every step, or NULL */
  /* 0: as declared above.  1 (ptrwm_run only): `state`, `trace` and `ext_prop` point to DOUBLE arrays of the same shapes -
   * the reference's dtype=torch.float64 (pt_rwm_gpu_optimized.py:134,431-449; experiment_pt_GPU.py:236
   * --use_double_precision): states, the proposals x + scale * z and the squared-jump sums are carried in double, so a
   * state far from the origin keeps the low bits of its increments; log-densities (evaluated on the proposal rounded to
   * float), uniforms, temperatures and proposal scales stay float.  ext_prop in this mode: NORMAL proposal only (the
   * reference's PT class is Gaussian only).  Always the lane-split form of the kernel; ladders of <= 128 temperatures. */
  int32_t state_f64;
  int32_t split_flags; /* device-step mode of the split steps only (below); must be 0 everywhere else */
  /* Split steps only (ptrwm_split_propose / ptrwm_split_accept / ptrwm_split_advance; must be NULL for ptrwm_run and
   * ptrwm_swap_sweep): device pointer to a step counter.  When set, the step a call performs is *device_step + step0 -
   * the kernels read the counter from device memory, `step0` is an OFFSET baked into the call - and burn-in and swap
   * schedule are derived from that on the device: the argument list of the k-th step of a block no longer depends on
   * where the run stands, so a block of split steps - the caller's density evaluation included - can be captured ONCE in
   * a HIP graph (torch.cuda.CUDAGraph) with step0 = 0, 1, ..., n - 1 and ONE ptrwm_split_advance (n_steps = n) as its last
   * node, and replayed.  ptrwm_split_accept enqueues the swap kernel with every step (it returns at once when no event is
   * due) unless split_flags has PTRWM_SPLIT_NO_SWEEP set: the caller's assertion that the step is NOT a swap step - its
   * to keep (a caller that replays a block only from counters that are multiples of swap_every knows which offsets are
   * swap steps; a wrong assertion loses the event silently).  External randoms (ext_prop / ext_u / ext_swap_u) are not
   * available in this mode. */
  const int64_t *device_step;
} ptrwm_run_args;

/* Advance every (chain, temperature) replica by n_steps Metropolis steps (with
 * swaps) in one fused kernel launch on `stream`. */
int32_t ptrwm_run(const ptrwm_target_desc *target, const ptrwm_proposal_desc *proposal,
                  const ptrwm_run_args *args, void *stream);

/* One stand-alone swap event over the current states: what the reference's
 * ParallelTemperingRWM_GPU_Optimized._attempt_all_swaps() does when called on its own
 * (pt_rwm_gpu_optimized.py:594-633; tests/debug_pt_performance.py:156).  Exactly the swap part of a ptrwm_run step:
 * same decision rule, modes and orders.  Reads from `args`: n_temps, n_chains, chain_offset, state, logp, beta,
 * swap_accept, last_swap_ordinal (both may be NULL), swap_mode, swap_order, seed, ext_swap_u (device
 * [n_chains, n_temps-1] or NULL), and step0 = the Philox step index the swap uniforms are drawn at.
 * `event_index` = 0-based number of this event in the run (attempt ordinals, even/odd parity); `rng_stream` in
 * 1..15 selects the Philox stream (1 = the stream ptrwm_run's own swap events use, so a sweep with step0 = s,
 * rng_stream = 1 reproduces the swap ptrwm_run would perform at step s).  sq_jump is not touched. */
int32_t ptrwm_swap_sweep(const ptrwm_run_args *args, int32_t dim, int64_t event_index, int32_t rng_stream,
                         void *stream);

/* Split step, for target densities the library has no kernel for (a user-defined
 * TorchTargetDistribution.log_density: SURVEY 8b "anything unrecognised"; the reference calls target.log_density on
 * the proposals inside step(), pt_rwm_gpu_optimized.py:551, rwm_gpu_optimized.py:289-336).  One step = three calls:
 *   1. ptrwm_split_propose: for step args->step0, proposals[c,t,:] = state[c,t,:] + increment and
 *      accept_u[c,t] = the step's accept uniform - the same Philox words (or ext_prop [n_chains, n_temps, raw] /
 *      ext_u [n_chains, n_temps] of THIS step) and the same arithmetic as ptrwm_run;
 *   2. the caller evaluates logp_proposed[c,t] = log_density(proposals[c,t,:]) on the device, any way it likes;
 *   3. ptrwm_split_accept: Metropolis rule, state / logp / statistics update and, when this step's step_counter is a
 *      swap step, the swap event (ext_swap_u [n_chains, n_temps-1] of THIS event in external-randoms mode) - exactly
 *      what ptrwm_run does for the step.  Reads from `args`: everything ptrwm_run reads except n_steps and the
 *      trace fields (accept_flags, if set, is [n_chains, n_temps] for this step).
 * `proposals` [n_chains, n_temps, dim] and `accept_u` [2, n_chains, n_temps] are device scratch owned by the caller
 * (accept_u plane 0: the accept uniforms; plane 1: the squared length of the increment as the fused kernel counts it -
 * the Philox paths of the Normal and UniformRadius proposals know it without a pass over the dimensions - or -1);
 * after ptrwm_split_accept of a SWAP step `proposals` holds the states from before the step (the swap kernel reads them);
 * after any other step its contents are unspecified.  Driven with ptrwm_logdensity as the
 * density, a split step reproduces ptrwm_run bit for bit (tests/test_gpu_engine_parity.py) - state, log-densities and
 * counters always; sq_jump as ptrwm_run called one step at a time does: the split step takes the trust verdict of
 * ptrwm_run_args.sq_jump's comment from the state every step, ptrwm_run once per launch, so the two differ (within that
 * comment's ~3e-5) only for a replica that crosses the bound in the middle of a longer launch. */
int32_t ptrwm_split_propose(const ptrwm_proposal_desc *proposal, const ptrwm_run_args *args, int32_t dim,
                            float *proposals, float *accept_u, void *stream);
int32_t ptrwm_split_accept(const ptrwm_run_args *args, int32_t dim, float *proposals, const float *accept_u,
                           const float *logp_proposed, void *stream);
/* *args->device_step += max(1, args->n_steps) on `stream` (one thread): the last node of a block of split steps in
 * device-step mode. */
int32_t ptrwm_split_advance(const ptrwm_run_args *args, void *stream);

/* out[i] = log_density(x[i, :]) for i < n; x is device [n, dim], out device [n]. */
int32_t ptrwm_logdensity(const ptrwm_target_desc *target, const float *x, float *out, int64_t n,
                         void *stream);

/* Proposal increments only (unit parity of the three samplers):
 * out[i, t, :] for i < n.  If ext_raw != NULL it is device [n, n_temps, raw_per_step]
 * and the transform is applied to it; otherwise Philox(seed) with the same
 * counter layout as ptrwm_run at step index i, chain id 0.. */
int32_t ptrwm_propose(const ptrwm_proposal_desc *proposal, int32_t dim, int32_t n_temps, int64_t n,
                      const float *ext_raw, uint64_t seed, float *out, void *stream);

/* Raw Philox4x32-10 blocks: out[i*4 .. i*4+3] = philox(counter = (c0 + i, c1, c2, c3), key = seed).
 * out is device [n, 4] uint32.  Known-answer tested against the Random123 vectors. */
int32_t ptrwm_philox_raw(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                         int64_t n, uint32_t *out, void *stream);

/* 1 if the (target, proposal, dim) variant is compiled in, else 0. */
int32_t ptrwm_has_variant(int32_t target_kind, int32_t proposal_kind, int32_t dim);

int32_t ptrwm_abi_version(void);
const char *ptrwm_strerror(int32_t code);

#ifdef __cplusplus
}
#endif
#endif /* PTRWM_H */
