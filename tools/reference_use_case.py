"""The reference's own calling pattern, end to end (development aid; needs a GPU): MCMCSimulation_GPU with ONE chain / ONE
ladder, as experiment_RWM_GPU.py and experiment_pt_GPU.py drive it - construct (iterative ladder for PT), generate_samples
(list-of-lists return, as the reference's harness does), acceptance / ESJD queries - wall clock of the whole thing.
The reference's GPU classes run one Python iteration per step (~3e3 steps/s measured in SURVEY section 6)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rwm-pt-pytorch_amd"))
import torch  # noqa: E402

from algorithms import ParallelTemperingRWM_GPU_Optimized, RandomWalkMH_GPU_Optimized  # noqa: E402
from interfaces import MCMCSimulation_GPU  # noqa: E402
from target_distributions import RoughCarpetDistributionTorch, ThreeMixtureDistributionTorch  # noqa: E402

dev = "cuda"
for cls in (RoughCarpetDistributionTorch, ThreeMixtureDistributionTorch):
    for dim in (20, 30):
        for n in (100_000, 1_000_000):
            for algo, kw in ((RandomWalkMH_GPU_Optimized, {}),
                             (ParallelTemperingRWM_GPU_Optimized, dict(iterative_temp_spacing=True, swap_acceptance_rate=0.234))):
                target = cls(dim, device=torch.device(dev))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                sim = MCMCSimulation_GPU(dim=dim, sigma=2.38**2 / dim, num_iterations=n, algorithm=algo, target_dist=target,
                                         symmetric=True, seed=1, burn_in=1000, device=dev, pre_allocate=True, **kw)
                t1 = time.perf_counter()
                old = sys.stdout
                sys.stdout = open(os.devnull, "w")
                try:
                    chain = sim.generate_samples(progress_bar=False)
                finally:
                    sys.stdout = old
                t2 = time.perf_counter()
                acc, esjd = sim.acceptance_rate(), sim.expected_squared_jump_distance()
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                temps = len(getattr(sim.algorithm, "beta_ladder", [1.0]))
                print(f"{cls.__name__[:12]:12s} dim {dim} {algo.__name__[:12]:12s} temps {temps:2d} N {n:7d}: construct {t1 - t0:6.3f} s, "
                      f"generate_samples (list) {t2 - t1:6.3f} s, statistics {t3 - t2:6.3f} s -> {n / (t3 - t0):9.3e} iterations/s end to end; "
                      f"acceptance {acc:.3f}", flush=True)
