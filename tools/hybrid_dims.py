"""HybridRosenbrock at the reference's data dims (9, 19, 29: kernels with the dim compiled in for this target alone, variants.h
PTRWM_WIDTHS_EXTRA) beside their neighbours, which run the run-time-dim kernels of the same register width class
(development aid; needs a GPU):  python tools/hybrid_dims.py > profiles/r04_hybrid_dims.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rwm-pt-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import ptrwm_hip as E  # noqa: E402
from algorithms import ParallelTemperingRWM_GPU_Optimized, geometric_beta_ladder  # noqa: E402
from target_distributions import HybridRosenbrockTorch  # noqa: E402

dev = torch.device("cuda:0")


def rate(n1, n2, T, C, form):
    target = HybridRosenbrockTorch(n1, n2, device=dev)
    dim = target.dim
    alg = ParallelTemperingRWM_GPU_Optimized(dim, 0.02, target, beta_ladder=geometric_beta_ladder(T) if T > 1 else [1.0], swap_every=10,
                                             burn_in=0, device=dev, num_replicas=C, seed=1, trace="none")
    alg._ensure_started()
    inner = int(max(50, min(20000, 2e9 / (C * T * dim / 30.0))))
    with E.kernel_form(form):
        alg._run.advance(inner)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            alg._run.advance(inner)
        e1.record()
        torch.cuda.synchronize()
    return dim, C * T * inner * 3 / (e0.elapsed_time(e1) * 1e-3)


print(f"{'dim':>4} {'(n1,n2)':>8} {'compiled in':>11} {'T':>3} {'ladders':>8} {'thread':>10} {'per dim':>10} {'lane-split':>10} {'per dim':>10}")
# dim = 1 + n2 (n1 - 1)
for n1, n2 in ((2, 7), (3, 4), (2, 10), (3, 8), (4, 6), (3, 10), (4, 9), (5, 7), (3, 15)):
    for T, C in ((32, 16384), (8, 65536)):
        d, a = rate(n1, n2, T, C, E.FORM_THREAD)
        _, b = rate(n1, n2, T, C, E.FORM_QUAD)
        own = bool(E.has_stream_variant(E.TARGET_HYBRID_ROSENBROCK, E.PROPOSAL_NORMAL, d))
        print(f"{d:4d} {f'({n1},{n2})':>8} {str(own):>11} {T:3d} {C:8d} {a:10.3e} {a * d:10.3e} {b:10.3e} {b * d:10.3e}", flush=True)
