"""Host cost of one-step launches (development aid): wall-clock per `advance(1)` with and without HIP events
between the launches.  python tools/host_overhead.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rwm-pt-pytorch_amd"))
import torch  # noqa: E402

from algorithms import ParallelTemperingRWM_GPU_Optimized, geometric_beta_ladder  # noqa: E402
from target_distributions import RoughCarpetDistributionTorch  # noqa: E402

dev = torch.device("cuda:0")
for C, T in ((1, 8), (2048, 32), (65536, 32)):
    tgt = RoughCarpetDistributionTorch(30, device=dev, mode_centers=[-15.0, 0.0, 15.0])
    alg = ParallelTemperingRWM_GPU_Optimized(30, 2.38**2 / 30, tgt, beta_ladder=geometric_beta_ladder(T), swap_every=10,
                                             device=dev, num_replicas=C, seed=1, trace="none")
    alg._ensure_started()
    run = alg._run
    for _ in range(50):
        run.advance(1)
    torch.cuda.synchronize()
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n):
        run.advance(1)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        run.advance(1)
        b.record()
    torch.cuda.synchronize()
    t_ev = time.perf_counter() - t0
    kms = sum(a.elapsed_time(b) for a, b in ev) / n
    t0 = time.perf_counter()
    for _ in range(n):
        alg.step()
    torch.cuda.synchronize()
    t_step = time.perf_counter() - t0
    print(f"C={C} T={T}: issue {t_issue / n * 1e6:.1f} us/launch, end-to-end {t_all / n * 1e6:.1f} us/launch, "
          f"with event pairs {t_ev / n * 1e6:.1f} us/launch (event-measured {kms * 1e3:.1f} us), "
          f"alg.step() {t_step / n * 1e6:.1f} us")
