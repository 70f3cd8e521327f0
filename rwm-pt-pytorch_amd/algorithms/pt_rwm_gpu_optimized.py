"""Parallel-tempering Random-Walk Metropolis on the fused HIP kernel.

Drop-in for the reference's `ParallelTemperingRWM_GPU_Optimized`
(algorithms/pt_rwm_gpu_optimized.py:101-841).  One thread owns one (replica, temperature) pair and
the temperatures of a ladder sit in one wavefront, so the reference's per-step Python work -- the
batched MH move (:541-567), the sequential swap sweep with a host sync per pair (:594-633) and the
per-temperature chain writes (:635-653) -- is one kernel launch for the whole run.

Kept semantics: `num_chains` is the number of temperatures; the schedule (all temperatures move,
then swaps when step_counter % swap_every == 0 and step_counter > burn_in); swap statistics that
refresh on accepted swaps only.  Made explicit instead of silently inherited:

  swap_mode  "exchange" (default; the two rows trade places) or "reference_copy" (what the
             reference's tuple assignment on tensor views actually does: row j <- row k)
  swap_order "sequential" (default; the reference's j = 0..T-2 sweep) or "even_odd"

Extensions: `num_replicas` independent ladders advanced together (default 1); any of the three
proposal families via `proposal_distribution` (the reference PT class is Gaussian-only).
"""
from __future__ import annotations

import time
import warnings
from typing import Optional

import numpy as np
import torch

import ptrwm_hip
from interfaces import MHAlgorithm, TargetDistribution, TorchTargetDistribution
from proposal_distributions import LaplaceProposal, NormalProposal, ProposalDistribution, UniformRadiusProposal

from ._engine_core import EngineRun, resolve_device


def geometric_beta_ladder(n_temps: int, beta_min: float = 0.01) -> list:
    """beta_t = beta_min^(t/(n_temps-1)), t = 0..n_temps-1: the explicit n-point geometric ladder the
    reference lacks (its own geometric ladder always has 8 rungs, see `_construct_geometric_ladder`)."""
    if n_temps < 1:
        raise ValueError("n_temps must be >= 1")
    if n_temps == 1:
        return [1.0]
    return [float(beta_min ** (t / (n_temps - 1))) for t in range(n_temps)]


class ParallelTemperingRWM_GPU_Optimized(MHAlgorithm):
    def __init__(self, dim: int, var: float, target_dist=None, symmetric: bool = True, beta_ladder: list = None,
                 iterative_temp_spacing: bool = False, geom_temp_spacing: bool = False,
                 swap_acceptance_rate: float = 0.234, beta_min_iterative: float = 0.01, N_samples_swap_est: int = 3000,
                 iterative_tolerance: float = 0.005, iterative_initial_pn: float = 0.5,
                 iterative_pn_update_power: float = -0.25, iterative_max_pn_steps: int = 100,
                 iterative_pn_clamp_min: float = -10.0, iterative_pn_clamp_max: float = 10.0,
                 iterative_fail_tol_factor: float = 3.0, swap_every: int = 100, burn_in: int = 0, device: str = None,
                 pre_allocate_steps: int = None, dtype: torch.dtype = torch.float32, *,
                 num_replicas: int = 1, proposal_distribution: Optional[ProposalDistribution] = None,
                 swap_mode: str = "exchange", swap_order: str = "sequential", seed: Optional[int] = None,
                 chain_offset: int = 0, trace: str = "all", thin: int = 1):
        super().__init__(dim, var, target_dist, symmetric)
        self.device = resolve_device(device)
        # dtype=torch.float64 (experiment_pt_GPU.py:236 --use_double_precision; pt_rwm_gpu_optimized.py:134,431-449): states,
        # proposals x + scale * z, stored chains and jump distances in double (the engine's state_f64 mode: lane-split
        # kernel with double state registers); log-densities, temperatures and proposal scales stay float32 as the
        # reference allocates them (:436-442,:453-455).  Any other dtype is an error, as torch would raise further down.
        if dtype not in (torch.float32, torch.float64):
            raise TypeError(f"dtype must be torch.float32 or torch.float64, got {dtype}")
        self.dtype = dtype
        self.burn_in = max(0, burn_in)
        self.swap_every = swap_every
        self.ideal_swap_acceptance_rate = swap_acceptance_rate
        self.name = "PT_RWM_GPU_ULTRA_FUSED_ITERATIVE_LADDER" if iterative_temp_spacing else "PT_RWM_GPU_ULTRA_FUSED"
        if trace not in ("all", "cold", "none"):
            raise ValueError("trace must be 'all', 'cold' or 'none'")
        self._trace_mode = trace
        if thin < 1:
            raise ValueError("thin must be >= 1")
        self.thin = int(thin)  # store every thin-th state (1 = every state, the reference behaviour)

        self.use_torch_target = isinstance(self.target_dist, TorchTargetDistribution)
        if not self.use_torch_target:
            raise TypeError(
                "ParallelTemperingRWM_GPU_Optimized needs a TorchTargetDistribution the fused kernel implements; "
                f"got {type(target_dist).__name__} (legacy NumPy targets have no GPU path)."
            )
        self.target_dist.to(self.device)

        if beta_ladder is not None:
            self.beta_ladder = list(beta_ladder)
        elif iterative_temp_spacing:
            self.beta_ladder = self._construct_iterative_ladder(
                target_swap_acceptance_rate=swap_acceptance_rate, beta_min=beta_min_iterative,
                N_samples_for_swap_estimation=N_samples_swap_est, tolerance=iterative_tolerance,
                initial_pn=iterative_initial_pn, pn_update_power=iterative_pn_update_power,
                max_pn_adjustment_steps=iterative_max_pn_steps,
                pn_clamping_range=(iterative_pn_clamp_min, iterative_pn_clamp_max),
                convergence_failure_tolerance_factor=iterative_fail_tol_factor)
        else:
            self.beta_ladder = self._construct_geometric_ladder()
            if not geom_temp_spacing:
                warnings.warn("No specific ladder construction method chosen. Using geometric spacing as default.")

        self.num_chains = len(self.beta_ladder)  # = number of temperatures, as in the reference
        # Double states need the lane-split kernel with double state registers (ladders of <= 128 temperatures) and a target
        # with a fused kernel (split steps carry float32 states).  What it cannot serve runs in float32 and says so - the
        # reference's scripts pass --use_double_precision to every target (experiment_pt_GPU.py:236) and must keep running.
        if self.dtype == torch.float64:
            has_kernel = True
            try:
                self.target_dist.engine_target()
            except (NotImplementedError, AttributeError):
                has_kernel = False
            if not has_kernel or self.num_chains > 128:
                warnings.warn("dtype=torch.float64 is served by the fused kernel for ladders of <= 128 temperatures only: "
                              + (f"{type(self.target_dist).__name__} has no fused kernel (split steps)" if not has_kernel
                                 else f"{self.num_chains} temperatures") + "; states are kept in float32")
                self.dtype = torch.float32
        self.num_replicas = int(num_replicas)
        self.beta_tensor = torch.tensor(self.beta_ladder, device=self.device, dtype=torch.float32)

        if proposal_distribution is None:
            if var is None or var <= 0:
                raise ValueError("var must be a positive proposal variance")
            proposal_distribution = NormalProposal(dim, var, 1.0, self.device, torch.float32, None)
        elif not isinstance(proposal_distribution, (NormalProposal, LaplaceProposal, UniformRadiusProposal)):
            raise TypeError(f"{type(proposal_distribution).__name__} is not implemented by the fused kernel")
        self.proposal_dist = proposal_distribution
        self._swap_mode, self._swap_order = swap_mode, swap_order
        if swap_mode not in ptrwm_hip.SWAP_MODES or swap_order not in ptrwm_hip.SWAP_ORDERS:
            raise ValueError(f"swap_mode in {sorted(ptrwm_hip.SWAP_MODES)}, swap_order in {sorted(ptrwm_hip.SWAP_ORDERS)}")
        self._seed, self._chain_offset = seed, chain_offset

        self.pre_allocate_steps = pre_allocate_steps
        self._alloc_trace((self.burn_in + pre_allocate_steps) // self.thin + 1 if pre_allocate_steps else 0)
        self.step_counter = 0
        self._run: Optional[EngineRun] = None
        self._chain_cache = None
        self._initial_state = np.asarray(MHAlgorithm.get_curr_state(self))
        self.current_states = None
        self.current_log_densities = None

    # ---- ladders (host-side setup; reference :245-257 and :283-426) ---------------------------------
    def _construct_geometric_ladder(self):
        """1, 1/2, 1/4, ... while > 0.01, then 0.01 (always 8 rungs)."""
        ladder, b = [], 1.0
        while b > 1e-2:
            ladder.append(b)
            b *= 0.5
        ladder.append(1e-2)
        return ladder

    def _get_typical_samples_at_beta(self, beta_val: float, N_samples: int) -> torch.Tensor:
        if not hasattr(self.target_dist, "draw_samples_torch"):
            raise NotImplementedError("The target distribution must implement 'draw_samples_torch(n_samples, beta)' "
                                      "for iterative temperature ladder construction.")
        return self.target_dist.draw_samples_torch(N_samples, beta=beta_val).to(self.device)

    def _estimate_swap_rate(self, beta_hi: float, beta_lo: float, n: int) -> float:
        """Mean of min(1, exp((beta_hi - beta_lo)(l(x_lo) - l(x_hi)))) over typical samples at the two
        temperatures (reference :356-369); both log-density batches go through the engine."""
        x_lo = self._get_typical_samples_at_beta(beta_lo, n)
        x_hi = self._get_typical_samples_at_beta(beta_hi, n)
        log_r = (beta_hi - beta_lo) * (self.target_dist.log_density(x_lo) - self.target_dist.log_density(x_hi))
        return torch.exp(torch.clamp_max(log_r, 0.0)).mean().item()

    def _construct_iterative_ladder(self, target_swap_acceptance_rate, beta_min, N_samples_for_swap_estimation,
                                    tolerance, initial_pn, pn_update_power, max_pn_adjustment_steps, pn_clamping_range,
                                    convergence_failure_tolerance_factor) -> list:
        """Robbins-Monro ladder: from beta = 1 downwards, the next rung is beta / (1 + e^rho) with rho
        adjusted by n^power (a - a_target) until the estimated swap rate a is within tolerance."""
        ladder, beta_curr = [1.0], 1.0
        while beta_curr > beta_min + 1e-6:
            rho, n_upd, found = initial_pn, 1, False
            cand, cand_rate, hit_floor = -1.0, -1.0, False
            for it in range(1, max_pn_adjustment_steps + 1):
                cand = beta_curr / (1.0 + float(np.exp(np.clip(rho, *pn_clamping_range))))
                if cand < beta_min:
                    hit_floor = True
                    break
                cand_rate = self._estimate_swap_rate(beta_curr, cand, N_samples_for_swap_estimation)
                if abs(cand_rate - target_swap_acceptance_rate) <= tolerance:
                    found = True
                    break
                rho += (n_upd ** pn_update_power) * (cand_rate - target_swap_acceptance_rate)
                n_upd += 1
            if not found:
                within_wide = abs(cand_rate - target_swap_acceptance_rate) <= tolerance * convergence_failure_tolerance_factor
                if hit_floor or not within_wide:
                    break
            ladder.append(cand)
            beta_curr = cand
        if ladder[-1] > beta_min + 1e-5:
            ladder.append(beta_min)
        print(f"[Ladder Construction] beta_ladder (length {len(ladder)}): [{', '.join(f'{b:.6f}' for b in ladder)}]")
        return ladder

    # ---- storage ---------------------------------------------------------------------------------------
    def _alloc_trace(self, rows: int):
        T = len(self.beta_ladder)
        nt = {"all": T, "cold": 1, "none": 0}[self._trace_mode]
        self._trace_rows = rows
        if rows and nt:
            self._trace = torch.zeros((rows, 1, nt, self.dim), device=self.device, dtype=self.dtype)
            self._trace_logp = torch.zeros((rows, 1, nt), device=self.device, dtype=torch.float32)
            # reference layout [temperature, step, dim] as a view
            self.pre_allocated_chains = self._trace[:, 0].permute(1, 0, 2)
            self.pre_allocated_log_densities = self._trace_logp[:, 0].permute(1, 0)
        else:
            self._trace = self._trace_logp = None
            self.pre_allocated_chains = self.pre_allocated_log_densities = None
        self._rows_used = 0

    @property
    def chain_indices(self):
        """The reference's per-temperature write positions (pt_rwm_gpu_optimized.py:471): every temperature has stored
        the same number of states, `_rows_used`; built on demand (updating a tensor every step() costs microseconds)."""
        if getattr(self, "_trace", None) is None:
            return None
        return torch.full((len(self.beta_ladder),), self._rows_used, dtype=torch.long)

    def _ensure_started(self):
        if self._run is not None:
            return
        self._run = EngineRun(
            target_dist=self.target_dist, proposal=self.proposal_dist.engine_proposal(self.beta_ladder),
            beta_ladder=self.beta_ladder, dim=self.dim, device=self.device, n_replicas=self.num_replicas,
            initial_state=self._initial_state, burn_in=self.burn_in, swap_every=self.swap_every,
            swap_mode=self._swap_mode, swap_order=self._swap_order, seed=self._seed, chain_offset=self._chain_offset,
            dtype=self.dtype)
        # reference shapes for one ladder: [T, dim] / [T]; with replicas: [R, T, dim] / [R, T]
        self.current_states = self._run.state[0] if self.num_replicas == 1 else self._run.state
        self.current_log_densities = self._run.logp[0] if self.num_replicas == 1 else self._run.logp
        if self._trace is not None and self._rows_used == 0:
            nt = self._trace.shape[2]
            self._trace[0, 0] = self._run.state[0, :nt]
            self._trace_logp[0, 0] = self._run.logp[0, :nt]
            self._rows_used = 1

    def _advance(self, n_steps: int):
        self._ensure_started()
        if self._trace_mode != "none":
            rows = self._run.traced_rows(n_steps, self.thin)
            if self._trace is None or self._rows_used + rows > self._trace_rows:
                # no (or too small a) pre-allocation: grow the device-side chain storage GEOMETRICALLY (at least
                # doubling), so that a step()-by-step() run copies O(N) rows in total, not O(N^2); `_rows_used` /
                # `chain_indices` stay the logical length
                old, old_lp, used = self._trace, self._trace_logp, self._rows_used
                self._alloc_trace(max(max(used, 1) + rows, 2 * self._trace_rows, 64))
                if old is not None and used:
                    self._trace[:used] = old[:used]
                    self._trace_logp[:used] = old_lp[:used]
                    self._rows_used = used
                else:
                    nt = self._trace.shape[2]
                    self._trace[0, 0] = self._run.state[0, :nt]
                    self._trace_logp[0, 0] = self._run.logp[0, :nt]
                    self._rows_used = 1
            self._run.advance(n_steps, trace=self._trace, trace_logp=self._trace_logp, trace_row0=self._rows_used,
                              trace_every=self.thin)
            self._rows_used += rows
        else:
            self._run.advance(n_steps)
        self.step_counter += n_steps
        self._chain_cache = None

    # ---- public stepping API ---------------------------------------------------------------------------------
    def get_name(self):
        return self.name

    def reset(self):
        self._run = None
        self.step_counter = 0
        self._chain_cache = None
        self.current_states = self.current_log_densities = None
        self._rows_used = 0

    def step(self, step_index: int = None):
        """All temperatures take one MH step; swaps follow when due (one fused launch, n_steps = 1)."""
        self._advance(1)

    def generate_samples(self, num_samples: int):
        """Run burn_in + num_samples PT steps; return the cold chain's post-burn-in states (num_samples, dim)."""
        total_steps = self.burn_in + num_samples
        t0 = time.time()
        self._advance(total_steps)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        dt = max(time.time() - t0, 1e-12)
        print(f"Generated {num_samples} samples (+ {self.burn_in} burn-in), {self.num_chains} temperatures x "
              f"{self.num_replicas} replica(s), swaps every {self.swap_every}, in {dt:.3f}s "
              f"({total_steps * self.num_chains * self.num_replicas / dt:.3e} chain-steps/s, "
              f"swap accept {self.swap_acceptance_rate:.3f})")
        if self._trace_mode == "none":
            return torch.empty(0, self.dim, device=self.device)
        return self.get_cold_chain_gpu()[1 + self.burn_in // self.thin:]

    # ---- chains ----------------------------------------------------------------------------------------
    def get_all_chains_gpu(self):
        """Per-temperature chains of replica 0 (initial state and burn-in included)."""
        if self._trace is None:
            return []
        return [self._trace[:self._rows_used, 0, t] for t in range(self._trace.shape[2])]

    def get_cold_chain_gpu(self):
        chains = self.get_all_chains_gpu()
        return chains[0] if chains else torch.empty(0, self.dim, device=self.device)

    def _get_cold_chain_cpu(self):
        return self.get_cold_chain_gpu().detach().cpu().numpy().tolist()

    @property
    def chain(self):
        """Cold chain as a Python list, transferred lazily."""
        if getattr(self, "_chain_cache", None) is None:
            if getattr(self, "_trace", None) is None or self._rows_used == 0:
                return self._base_chain
            self._chain_cache = self._get_cold_chain_cpu()
        return self._chain_cache

    @chain.setter
    def chain(self, value):
        if not hasattr(self, "_base_chain"):
            self._base_chain = value  # the one-element list the base class creates
        else:
            self._chain_cache = value

    # ---- swap statistics (reference :619-633: refreshed only when a swap is accepted) -----------------------
    @property
    def num_swap_attempts(self) -> int:
        return 0 if self._run is None else self._run.swap_attempts_per_replica() * self.num_replicas

    @property
    def num_swap_acceptances(self) -> int:
        return 0 if self._run is None else int(self._run.swap_accept.sum().item())

    def _sq_beta_jumps(self) -> float:
        b = torch.tensor(self.beta_ladder, dtype=torch.float64)
        acc = self._run.swap_accept.sum(0).cpu().double()
        return float((acc[:-1] * (b[:-1] - b[1:]) ** 2).sum()) if len(b) > 1 else 0.0

    @property
    def squared_jump_distances(self) -> float:
        return 0.0 if self._run is None else self._sq_beta_jumps()

    def _stat_denominator(self) -> int:
        """The reference refreshes `swap_acceptance_rate` / `pt_esjd` only when a swap is accepted (:627-633), so what
        it reports is accepted / (attempt count AT THE LAST ACCEPTED SWAP).  With the sequential sweep every ladder
        records that ordinal; the denominator is their sum over the ladders - for one ladder exactly the reference's
        number, for many the same statistic pooled (no jump when `num_replicas` goes from 1 to 2).  Even/odd events have
        a varying pair count per event: all attempts are used there."""
        if self._run.swap_order == ptrwm_hip.ORDER_SEQUENTIAL:
            return int(self._run.last_ord.max(dim=1).values.sum().item())
        return self.num_swap_attempts

    @property
    def swap_acceptance_rate(self) -> float:
        if self._run is None:
            return 0.0
        den = self._stat_denominator()
        return self.num_swap_acceptances / den if den else 0.0

    @property
    def pt_esjd(self) -> float:
        if self._run is None:
            return 0.0
        den = self._stat_denominator()
        return self._sq_beta_jumps() / den if den else 0.0

    def mh_acceptance_rates(self) -> torch.Tensor:
        """Per-temperature MH acceptance rate (an extra: the reference PT class tracks none)."""
        self._ensure_started()
        n = max(1, self._run.post_burn_steps * self.num_replicas)
        return self._run.n_accept.sum(0).double().cpu() / n

    def expected_squared_jump_distance_gpu(self):
        """Cold-chain ESJD over post-burn-in steps, swap moves included (reference :772-789); online fp64
        accumulation in the kernel, averaged over replicas."""
        if self._run is None or self.step_counter <= self.burn_in:
            raise ValueError("Insufficient post-burn-in samples")
        return float(self._run.sq_jump[:, 0].sum().item()) / (self._run.post_burn_steps * self.num_replicas)

    # ---- helpers the reference's debug script pokes at ----------------------------------------------------
    def _generate_all_increments(self):
        """One proposal increment per temperature, [T, dim] (reference :576-592)."""
        self._ensure_started()
        return ptrwm_hip.propose(self._run.proposal, self.dim, 1, seed=self._run.seed ^ 0x5DEECE66D)[0]

    def _compute_log_densities_for_proposals(self, proposals):
        return self.target_dist.log_density(proposals)

    def _attempt_all_swaps(self):
        """One swap sweep over the current states, outside the step schedule (reference :594-633; called on its own
        by tests/debug_pt_performance.py:156).  Counted in num_swap_attempts / num_swap_acceptances; the stored
        chain is not extended (the reference appends states in step(), not here)."""
        self._ensure_started()
        self._run.swap_sweep()
        self._chain_cache = None

    def get_diagnostic_info(self):
        return {
            "device": str(self.device),
            "dtype": str(self.dtype),
            "algorithm": self.name,
            "num_chains": self.num_chains,
            "num_replicas": self.num_replicas,
            "beta_ladder": self.beta_ladder,
            "swap_every": self.swap_every,
            "swap_mode": self._swap_mode,
            "swap_order": self._swap_order,
            "step_counter": self.step_counter,
            "swap_acceptance_rate": self.swap_acceptance_rate,
            "pt_esjd": self.pt_esjd,
            "optimization_level": "HIP_FUSED_PERSISTENT_LADDER_PER_WAVEFRONT",
            "parallel_processing": f"{self.num_chains} temperatures x {self.num_replicas} replicas, one thread each",
            "batch_matrix_multiply": "diagonal Cholesky folded into a per-temperature scale (no bmm)",
            "precomputed_randoms": "none: Philox4x32-10 drawn in-kernel",
            "clone_free_swaps": "swaps are LDS exchanges inside one wavefront / workgroup (no HBM traffic)",
            "kernel_fusion": "proposal, log-density, accept, update, swaps and statistics in one HIP kernel",
            "memory_allocated_mb": torch.cuda.memory_allocated() / 1e6 if self.device.type == "cuda" else 0,
        }

    def performance_summary(self):
        info = self.get_diagnostic_info()
        print("=" * 70)
        print(f"{self.name} on {info['device']}: {info['parallel_processing']}")
        print(f"  ladder: {[f'{b:.3f}' for b in self.beta_ladder]}")
        print(f"  swap acceptance {info['swap_acceptance_rate']:.3f}, PT-ESJD {info['pt_esjd']:.6f}, "
              f"steps {self.step_counter}")
        print("=" * 70)
