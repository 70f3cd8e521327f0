"""BASELINE.json configs[1] and configs[2] at their literal sizes through the drop-in classes (needs a GPU):
configs[1] RWM, RoughCarpet dim 30, Normal proposal, 65 536 chains, 1 000 000 iterations (burn-in 1 000, seed 42);
configs[2] PT-RWM, 32 geometric temperatures, swap_every 10, 65 536 ladders, 100 000 iterations."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rwm-pt-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from algorithms import ParallelTemperingRWM_GPU_Optimized, RandomWalkMH_GPU_Optimized, geometric_beta_ladder  # noqa: E402
from target_distributions import RoughCarpetDistributionTorch  # noqa: E402

dev = torch.device("cuda:0")
target = RoughCarpetDistributionTorch(30, device=dev, mode_centers=[-15.0, 0.0, 15.0])
alg = RandomWalkMH_GPU_Optimized(30, 2.38**2 / 30, target, burn_in=1000, device=dev, num_chains=65536, seed=42)
alg._ensure_started()
torch.cuda.synchronize()
t0 = time.perf_counter()
alg._run.advance(1000 + 1_000_000)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"configs[1]: 65536 chains x 1001000 steps in {dt:.2f} s = {65536 * 1001000 / dt:.3e} chain-MH-steps/s; "
      f"acceptance {alg.acceptance_rate:.6f}, ESJD {alg.expected_squared_jump_distance_gpu():.5f}", flush=True)
pt = ParallelTemperingRWM_GPU_Optimized(30, 2.38**2 / 30, target, beta_ladder=geometric_beta_ladder(32), swap_every=10,
                                        burn_in=1000, device=dev, num_replicas=65536, seed=42, trace="none")
pt._ensure_started()
torch.cuda.synchronize()
t0 = time.perf_counter()
pt._advance(1000 + 100_000)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"configs[2]: 65536 ladders x 32 temps x 101000 steps in {dt:.2f} s = {65536 * 32 * 101000 / dt:.3e} chain-MH-steps/s; "
      f"swap acceptance {pt.num_swap_acceptances / pt.num_swap_attempts:.6f}, cold ESJD {pt.expected_squared_jump_distance_gpu():.4f}, "
      f"cold MH acceptance {float(pt.mh_acceptance_rates()[0]):.6f}", flush=True)
