"""Simulation harness in front of the GPU samplers.

Drop-in for the non-plotting half of the reference's `MCMCSimulation_GPU`
(interfaces/simulation_gpu.py:13-311, :380-438): builds the proposal from `proposal_config`,
instantiates the sampler (dispatching on the 'GPU' / 'ParallelTempering' substrings of the class
name, :81-83), seeds torch/numpy AFTER construction (:144-148), runs `generate_samples` and exposes
acceptance rate / ESJD / PT-ESJD.  Plotting (traceplot, histograms, benchmark sweeps) is out of scope.
"""
import time
from typing import Optional

import numpy as np
import torch

from proposal_distributions import LaplaceProposal, NormalProposal, ProposalDistribution, UniformRadiusProposal


class MCMCSimulation_GPU:
    def __init__(self, dim: int, sigma: float = None, proposal_config: dict = None, num_iterations: int = 1000,
                 algorithm=None, target_dist=None, symmetric: bool = True, seed: Optional[int] = None,
                 beta_ladder: Optional[list] = None, swap_acceptance_rate: Optional[float] = None,
                 device: Optional[str] = None, pre_allocate: bool = True, burn_in: int = 0, **kwargs):
        if proposal_config is None:
            if sigma is None:
                raise ValueError("Either sigma (backward compatibility) or proposal_config must be provided")
            proposal_config = {"name": "Normal", "params": {"base_variance_scalar": sigma}}
        self.num_iterations = num_iterations
        self.burn_in = max(0, burn_in)
        self.target_dist = target_dist
        self.proposal_config = proposal_config
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.device = device
        self.pre_allocate = pre_allocate
        steps = num_iterations if pre_allocate else None

        algo_name = getattr(algorithm, "__name__", "")
        if "GPU" in algo_name and "ParallelTempering" in algo_name:
            pt_kwargs = dict(kwargs)
            if swap_acceptance_rate is not None:  # None means "keep the sampler's default"
                pt_kwargs["swap_acceptance_rate"] = swap_acceptance_rate
            if proposal_config.get("name") != "Normal" or sigma is None:
                pt_kwargs.setdefault("proposal_distribution", self._create_proposal_distribution(
                    dim, 1.0, proposal_config, torch.device(device), torch.float32))
            self.algorithm = algorithm(dim, sigma, target_dist, symmetric, device=device, pre_allocate_steps=steps,
                                       beta_ladder=beta_ladder, burn_in=self.burn_in, **pt_kwargs)
        elif "GPU" in algo_name:
            algo_beta = beta_ladder[0] if beta_ladder else 1.0
            proposal = self._create_proposal_distribution(dim, algo_beta, proposal_config, torch.device(device),
                                                          torch.float32, kwargs.get("use_efficient_rng", True))
            self.algorithm = algorithm(dim=dim, proposal_distribution=proposal, target_dist=target_dist,
                                       symmetric=symmetric, beta=algo_beta, device=device, pre_allocate_steps=steps,
                                       burn_in=self.burn_in, **kwargs)
        else:
            self.algorithm = algorithm(dim, sigma, target_dist, symmetric, beta_ladder=beta_ladder,
                                       swap_acceptance_rate=swap_acceptance_rate, burn_in=self.burn_in, **kwargs)

        # seeds are set after the sampler exists, so the sampler's initial point (drawn from the global
        # NumPy RNG in its constructor) is NOT covered by `seed`; the run itself is
        if seed is not None:
            torch.manual_seed(seed)
            np.random.seed(seed)
            if torch.cuda.is_available():
                torch.cuda.manual_seed(seed)

    def reset(self):
        self.algorithm.reset()

    def has_run(self):
        """Has the sampler taken a step?  Decided from host-side counters wherever they exist (the reference reads
        `pre_allocated_chain` / `chain_index`, simulation_gpu.py:160-161); `len(chain)` - for the PT class a lazy
        property that copies the whole cold chain to the host and turns it into a Python list - only as the last
        resort for samplers that have nothing else."""
        alg = self.algorithm
        for counter in ("step_counter", "total_steps"):  # PT class / RWM class: plain Python ints
            if isinstance(getattr(alg, counter, None), int):
                return getattr(alg, counter) > 0
        if getattr(alg, "pre_allocated_chain", None) is not None:
            return alg.chain_index > 1
        return len(alg.chain) > 1

    def generate_samples(self, progress_bar=True, as_list=True):
        """Run the sampler; returns the post-burn-in chain (list of lists like the reference, or the
        device tensor itself with `as_list=False`, which avoids an O(N*dim) host copy)."""
        if self.has_run():
            raise ValueError("Please reset the algorithm before running it again.")
        t0 = time.time()
        cls = type(self.algorithm).__name__
        if hasattr(self.algorithm, "generate_samples") and ("ParallelTempering" in cls or "GPU" in cls):
            chain = self.algorithm.generate_samples(self.num_iterations)
            if as_list and hasattr(chain, "cpu"):
                chain = chain.cpu().numpy().tolist()
        else:
            for _ in range(self.num_iterations + self.burn_in):
                self.algorithm.step()
            chain = self.algorithm.chain
        dt = max(time.time() - t0, 1e-12)
        print(f"Drew {self.num_iterations} samples in {dt:.2f} seconds ({self.num_iterations / dt:.0f} samples/s)")
        print(f"Final acceptance rate: {self.acceptance_rate():.3f}")
        return chain

    def _require_run(self):
        if not self.has_run():
            raise ValueError("The algorithm has not been run yet.")

    def acceptance_rate(self):
        self._require_run()
        return self.algorithm.acceptance_rate

    def expected_squared_jump_distance(self):
        self._require_run()
        if hasattr(self.algorithm, "expected_squared_jump_distance_gpu"):
            return self.algorithm.expected_squared_jump_distance_gpu()
        chain = np.array(self.algorithm.chain)
        if self.burn_in > 0 and len(chain) <= self.burn_in + 1:
            return 0.0
        post = chain[self.burn_in:]
        return np.mean(np.sum((post[1:] - post[:-1]) ** 2, axis=1))

    def pt_expected_squared_jump_distance(self):
        self._require_run()
        return self.algorithm.pt_esjd

    def benchmark_performance(self, num_samples_list=(1000, 5000, 10000, 50000), compare_cpu=False):
        """Wall-clock of `generate_samples` at several run lengths (reference simulation_gpu.py:252-311; same result
        keys).  There is no CPU sampler behind this harness, so the cpu_* / speedup entries stay None."""
        if compare_cpu:
            import warnings

            warnings.warn("compare_cpu=True ignored: this engine has no CPU path to compare against")
        sizes = list(num_samples_list)
        results = {"sample_sizes": sizes, "gpu_times": [], "gpu_samples_per_sec": [], "cpu_times": None,
                   "cpu_samples_per_sec": None, "speedup": None}
        original = self.num_iterations
        try:
            for n in sizes:
                self.reset()
                self.num_iterations = n
                t0 = time.time()
                self.generate_samples(progress_bar=False, as_list=False)
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                dt = max(time.time() - t0, 1e-12)
                results["gpu_times"].append(dt)
                results["gpu_samples_per_sec"].append(n / dt)
        finally:
            self.num_iterations = original
        return results

    def _create_proposal_distribution(self, dim: int, beta: float, proposal_config: dict, device: torch.device,
                                      dtype: torch.dtype, use_efficient_rng: bool = True) -> ProposalDistribution:
        """{'name': 'Normal'|'Laplace'|'UniformRadius', 'params': {...}} -> proposal object."""
        name = proposal_config.get("name")
        params = proposal_config.get("params", {})
        if name == "Normal":
            var = params.get("base_variance_scalar")
            if var is None:
                raise ValueError("Normal proposal requires 'base_variance_scalar' parameter")
            return NormalProposal(dim, var, beta, device, dtype, None)
        if name == "Laplace":
            vec = params.get("base_variance_vector")
            if vec is None:
                raise ValueError("Laplace proposal requires 'base_variance_vector' parameter")
            if isinstance(vec, (list, tuple)):
                vec = torch.tensor(vec, dtype=dtype)
            elif isinstance(vec, (int, float)):
                vec = torch.full((dim,), float(vec), dtype=dtype)
            elif isinstance(vec, torch.Tensor):
                vec = vec.to(dtype=dtype)
            else:
                raise ValueError(f"Invalid base_variance_vector type: {type(vec)}")
            return LaplaceProposal(dim, vec, beta, device, dtype, None)
        if name == "UniformRadius":
            radius = params.get("base_radius")
            if radius is None:
                raise ValueError("UniformRadius proposal requires 'base_radius' parameter")
            return UniformRadiusProposal(dim, radius, beta, device, dtype, None)
        raise ValueError(f"Unknown proposal distribution name: {name}")
