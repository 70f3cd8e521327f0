"""Wall time of the iterative temperature-ladder construction (pt_rwm_gpu_optimized.py:283-426 in the reference) through the
drop-in class, for the sample counts the reference's drivers use (development aid; needs a GPU).  Round 2, one MI355X: 9-130 ms."""
import os, sys, time
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "rwm-pt-pytorch_amd"))
import torch
from algorithms import ParallelTemperingRWM_GPU_Optimized
from target_distributions import RoughCarpetDistributionTorch, ThreeMixtureDistributionTorch
dev = torch.device("cuda:0")
for cls, dim in ((RoughCarpetDistributionTorch, 30), (ThreeMixtureDistributionTorch, 30), (RoughCarpetDistributionTorch, 10)):
    for n in (3000, 100000, 1000000):
        target = cls(dim, device=dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pt = ParallelTemperingRWM_GPU_Optimized(dim, 2.38**2 / dim, target, iterative_temp_spacing=True, swap_acceptance_rate=0.234,
                                                N_samples_swap_est=n, device=dev, seed=1, trace="none")
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{cls.__name__[:12]} dim {dim} N_samples_swap_est {n}: ladder of {len(pt.beta_ladder)} temps in {dt:.3f} s")
