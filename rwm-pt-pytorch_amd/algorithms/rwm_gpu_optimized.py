"""Random-Walk Metropolis on the fused HIP kernel.

Drop-in for the reference's `RandomWalkMH_GPU_Optimized` (algorithms/rwm_gpu_optimized.py:79-579):
same constructor, methods and attributes.  Where the reference runs one Python iteration, ten-odd
tiny kernels and a blocking `.item()` per step (:289-336), this class enqueues ONE launch of the
fused kernel for the whole run: Philox draw -> proposal -> log-density -> accept -> update ->
acceptance/ESJD accumulation all stay in registers (csrc/kernel.h).

Extension: `num_chains` independent chains are advanced together (default 1 = reference behaviour);
the returned / stored chain is always chain 0, the statistics cover all chains.
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


def ultra_fused_mcmc_step_basic(current_state, current_log_density, increment, random_val, beta, log_density_proposed):
    """The accept/select rule of one MH step on tensors (reference: rwm_gpu_optimized.py:9-32).

    Kept for callers/tests that import it; the samplers never call it -- the rule is the epilogue of
    the fused kernel (csrc/kernel.h, `ratio`/`acc`).  Returns (new_state, new_log_density, accepted).
    """
    log_accept_ratio = beta * (log_density_proposed - current_log_density)
    accepted = (log_accept_ratio > 0.0) | (random_val < torch.exp(log_accept_ratio))
    new_state = torch.where(accepted, current_state + increment, current_state)
    new_log_density = torch.where(accepted, log_density_proposed, current_log_density)
    return new_state, new_log_density, accepted


def _rebuild_proposal(p: ProposalDistribution, dim, beta, device, dtype, rng):
    """Same proposal family and base scale on another device/dtype (reference :166-200 reads the
    scale back from `std_dev` / `scale_vector` / `effective_radius`)."""
    if isinstance(p, NormalProposal):
        return NormalProposal(dim, float(p.std_dev**2 * p.beta), beta, device, dtype, rng)
    if isinstance(p, LaplaceProposal):
        return LaplaceProposal(dim, (p.scale_vector**2 * 2.0 * p.beta).to("cpu"), beta, device, dtype, rng)
    if isinstance(p, UniformRadiusProposal):
        return UniformRadiusProposal(dim, float(p.effective_radius) * float(p.beta) ** 0.5, beta, device, dtype, rng)
    raise TypeError(
        f"{type(p).__name__} is not a proposal the fused kernel implements (Normal, Laplace, UniformRadius)"
    )


class RandomWalkMH_GPU_Optimized(MHAlgorithm):
    def __init__(self, dim: int, var: float = None, target_dist=None, symmetric: bool = True, beta: float = 1.0,
                 burn_in: int = 0, device: str = None, pre_allocate_steps: int = None, use_efficient_rng: bool = True,
                 compile_mode: str = None, proposal_distribution: ProposalDistribution = None, *,
                 num_chains: int = 1, seed: Optional[int] = None, chain_offset: int = 0, thin: int = 1):
        if proposal_distribution is None and var is None:
            raise ValueError("Either var (backward compatibility) or proposal_distribution must be provided")
        super().__init__(dim, 1.0 if proposal_distribution is not None else var, target_dist, symmetric)
        if not isinstance(target_dist, TorchTargetDistribution):
            raise TypeError(
                "RandomWalkMH_GPU_Optimized needs a TorchTargetDistribution the fused kernel implements; "
                f"got {type(target_dist).__name__} (legacy NumPy targets have no GPU path)."
            )
        self.device = resolve_device(device)
        self.dtype = torch.float32
        self.beta = float(beta)
        self.beta_tensor = torch.tensor(beta, device=self.device, dtype=torch.float32)
        self.use_efficient_rng = use_efficient_rng
        self.rng_generator = None  # randoms are drawn in-kernel (Philox); no torch generator involved
        self.compile_mode = compile_mode
        self.compiled_log_density = None
        self.use_torch_target = True
        self.target_dist.to(self.device)

        if proposal_distribution is None:
            proposal_distribution = NormalProposal(dim, var, beta, self.device, self.dtype, None)
        elif proposal_distribution.device != self.device or proposal_distribution.dtype != self.dtype:
            proposal_distribution = _rebuild_proposal(proposal_distribution, dim, beta, self.device, self.dtype, None)
        elif not isinstance(proposal_distribution, (NormalProposal, LaplaceProposal, UniformRadiusProposal)):
            raise TypeError(f"{type(proposal_distribution).__name__} is not implemented by the fused kernel")
        self.proposal_dist = proposal_distribution
        self.name = f"RWM_GPU_FUSED_{self.proposal_dist.get_name()}"

        self.burn_in = max(0, burn_in)
        self.num_chains = int(num_chains)
        self._seed, self._chain_offset = seed, chain_offset
        self.total_steps = 0
        if thin < 1:
            raise ValueError("thin must be >= 1")
        self.thin = int(thin)  # store every thin-th state of chain 0 (1 = every state, the reference behaviour)
        self.pre_allocate_steps = pre_allocate_steps
        if pre_allocate_steps:
            rows = (self.burn_in + pre_allocate_steps) // self.thin + 1  # + initial state
            self._trace = torch.zeros((rows, 1, 1, dim), device=self.device, dtype=self.dtype)
            self._trace_logp = torch.zeros((rows, 1, 1), device=self.device, dtype=torch.float32)
            self.pre_allocated_chain = self._trace.view(rows, dim)
            self.pre_allocated_log_densities = self._trace_logp.view(rows)
            self.chain_index = 0
        else:
            self._trace = self._trace_logp = None
            self.pre_allocated_chain = self.pre_allocated_log_densities = None
            self.chain_index = None
        self._run: Optional[EngineRun] = None
        self.current_state = None
        self.log_target_density_current = None

    # counters live on the device; reading them synchronises
    @property
    def num_acceptances(self) -> int:
        return 0 if getattr(self, "_run", None) is None else int(self._run.n_accept.sum().item())

    @num_acceptances.setter
    def num_acceptances(self, value):  # the base class initialises it to 0
        pass

    @property
    def acceptance_rate(self) -> float:
        run = getattr(self, "_run", None)
        if run is None or run.post_burn_steps == 0:
            return 0.0
        return self.num_acceptances / (run.post_burn_steps * run.n_replicas)

    @acceptance_rate.setter
    def acceptance_rate(self, value):
        pass

    def get_name(self):
        return self.name

    def reset(self):
        super().reset()
        self._run = None
        self.total_steps = 0
        self.current_state = None
        self.log_target_density_current = None
        if self.pre_allocated_chain is not None:
            self.chain_index = 0

    # ---- engine plumbing ------------------------------------------------------------------------
    def _ensure_started(self):
        if self._run is not None:
            return
        self._run = EngineRun(
            target_dist=self.target_dist, proposal=self.proposal_dist.engine_proposal(), beta_ladder=[self.beta],
            dim=self.dim, device=self.device, n_replicas=self.num_chains, initial_state=self.chain[-1],
            burn_in=self.burn_in, swap_every=1, swap_mode="exchange", swap_order="sequential", seed=self._seed,
            chain_offset=self._chain_offset,
        )
        self.current_state = self._run.state[0, 0]  # views: always the live values
        self.log_target_density_current = self._run.logp[0, 0]
        if self.pre_allocated_chain is not None and self.chain_index == 0:
            self.pre_allocated_chain[0] = self.current_state
            self.pre_allocated_log_densities[0] = self.log_target_density_current
            self.chain_index = 1

    def _advance(self, n_steps: int):
        """n_steps fused MH steps; chain 0's states go to the pre-allocated chain or the Python list."""
        self._ensure_started()
        rows = self._run.traced_rows(n_steps, self.thin)
        if self.pre_allocated_chain is not None and self.chain_index + rows > self.pre_allocated_chain.shape[0]:
            warnings.warn("Pre-allocated chain full, switching to dynamic allocation")
            kept = self.pre_allocated_chain[1:self.chain_index].cpu().numpy()
            self.chain.extend(list(kept))
            self.pre_allocated_chain = self.pre_allocated_log_densities = None
            self._trace = self._trace_logp = None
        if self.pre_allocated_chain is not None:
            self._run.advance(n_steps, trace=self._trace, trace_logp=self._trace_logp, trace_row0=self.chain_index,
                              trace_every=self.thin)
            self.chain_index += rows
        else:
            tr = torch.empty((max(rows, 1), 1, 1, self.dim), device=self.device, dtype=self.dtype)
            self._run.advance(n_steps, trace=tr, trace_every=self.thin)
            self.chain.extend(list(tr[:rows].view(rows, self.dim).cpu().numpy()))
        self.total_steps += n_steps

    def step(self):
        """One MH step (one launch of the fused kernel with n_steps = 1)."""
        self._advance(1)

    def generate_samples(self, num_samples: int):
        """Run burn_in + num_samples steps; return chain 0's post-burn-in states, shape (num_samples, dim)."""
        total_steps = self.burn_in + num_samples
        t0 = time.time()
        self._advance(total_steps)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        dt = max(time.time() - t0, 1e-12)
        print(f"Generated {num_samples} samples (+ {self.burn_in} burn-in) x {self.num_chains} chain(s) with the fused "
              f"HIP kernel in {dt:.3f}s ({total_steps * self.num_chains / dt:.3e} chain-steps/s, "
              f"accept {self.acceptance_rate:.3f})")
        return self.get_chain_gpu()[1 + self.burn_in // self.thin:]

    # ---- read-outs ---------------------------------------------------------------------------------
    def get_chain_gpu(self):
        """Chain 0 including the initial state and burn-in, as a device tensor."""
        if self.pre_allocated_chain is not None:
            return self.pre_allocated_chain[:self.chain_index]
        return torch.tensor(np.array(self.chain), device=self.device, dtype=self.dtype)

    def get_log_densities_gpu(self):
        if self.pre_allocated_log_densities is not None:
            return self.pre_allocated_log_densities[:self.chain_index]
        return None

    def expected_squared_jump_distance_gpu(self):
        """Mean squared jump over the post-burn-in steps (reference :513-534), accumulated online by the
        kernel in fp64 instead of being recomputed from the stored chain; averaged over all chains."""
        run = self._run
        if run is None or run.post_burn_steps < 1:
            raise ValueError(f"Insufficient post-burn-in samples: total_steps={self.total_steps}, burn_in={self.burn_in}. "
                             f"Need at least {self.burn_in + 2} total samples.")
        return float(run.sq_jump.sum().item()) / (run.post_burn_steps * run.n_replicas)

    def per_chain_acceptance(self) -> torch.Tensor:
        """Acceptance rate of each chain (device tensor [num_chains])."""
        self._ensure_started()
        return self._run.n_accept[:, 0].double() / max(1, self._run.post_burn_steps)

    def get_diagnostic_info(self):
        return {
            "device": str(self.device),
            "dtype": str(self.dtype),
            "optimization_level": "HIP_FUSED_PERSISTENT",
            "use_efficient_rng": self.use_efficient_rng,
            "compiled_target": True,
            "total_steps": self.total_steps,
            "acceptance_rate": self.acceptance_rate,
            "num_chains": self.num_chains,
            "kernel_fusion": "Philox + proposal + log-density + accept + update + statistics in one HIP kernel",
            "memory_allocated_mb": torch.cuda.memory_allocated() / 1e6 if self.device.type == "cuda" else 0,
            "memory_efficiency": "state in registers for the whole launch; HBM touched at launch start/end",
            "random_generation": "Philox4x32-10 in-kernel (no precomputed random tensors)",
        }

    def performance_comparison_summary(self):
        info = self.get_diagnostic_info()
        print("=" * 60)
        print(f"{self.name} on {info['device']}: {info['kernel_fusion']}")
        print(f"  chains: {self.num_chains}, steps: {self.total_steps}, acceptance: {info['acceptance_rate']:.4f}")
        print(f"  RNG: {info['random_generation']}")
        print("=" * 60)
