"""The reference's own calling pattern with chain storage: one ladder, every state of every temperature kept
(generate_samples with the default trace='all'), and the step()-at-a-time loop (development aid; needs a GPU)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rwm-pt-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from algorithms import ParallelTemperingRWM_GPU_Optimized, RandomWalkMH_GPU_Optimized  # noqa: E402
from target_distributions import RoughCarpetDistributionTorch  # noqa: E402

dev = torch.device("cuda:0")
for dim in (20, 30):
    target = RoughCarpetDistributionTorch(dim, device=dev, mode_centers=[-15.0, 0.0, 15.0])
    n = 100000
    for trace in ("all", "cold", "none"):
        alg = ParallelTemperingRWM_GPU_Optimized(dim, 2.38**2 / dim, target, geom_temp_spacing=True, swap_every=10, burn_in=1000,
                                                 device=dev, pre_allocate_steps=n, trace=trace, seed=1)
        t0 = time.perf_counter()
        alg.generate_samples(n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"PT dim {dim} 8 temps trace={trace}: {n} samples in {dt:.3f} s = {dt / (n + 1000) * 1e6:.2f} us per PT step", flush=True)
    alg = RandomWalkMH_GPU_Optimized(dim, 2.38**2 / dim, target, burn_in=1000, device=dev, pre_allocate_steps=n)
    t0 = time.perf_counter()
    alg.generate_samples(n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"RWM dim {dim} one chain, chain stored: {n} samples in {dt:.3f} s = {dt / (n + 1000) * 1e6:.2f} us per step", flush=True)
    alg = ParallelTemperingRWM_GPU_Optimized(dim, 2.38**2 / dim, target, geom_temp_spacing=True, swap_every=10, device=dev,
                                             pre_allocate_steps=20000, trace="all", seed=1)
    t0 = time.perf_counter()
    for _ in range(20000):
        alg.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"PT dim {dim} step() x 20000: {dt / 20000 * 1e6:.2f} us per step (host-bound)", flush=True)
