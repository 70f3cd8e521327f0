"""Throughput of the fused kernel for EVERY target family x proposal at a full batch (development aid, needs a GPU):

    python tools/family_sweep.py [dims ...]       # default dims 30 50 24 48 100

16 384 ladders x 32 temperatures (sequential exchange swaps every 10 steps), Philox mode, form AUTO; dims 30 / 50 are
compiled in for the thread kernel, 24 / 48 run its generic widths, 100 runs the lane-split kernel.  Prints
chain-MH-steps/s and dim-steps/s; the point is to see outliers among the families, not to tune one of them."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rwm-pt-pytorch_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import helpers as H  # noqa: E402
import ptrwm_hip as E  # noqa: E402
from check_all_variants import make_spec  # noqa: E402
from oracle import oracle as O  # noqa: E402

f32 = np.float32
NAMES = {0: "RoughCarpet (3-term)", 10: "RoughCarpet (2-term)", 1: "ThreeMixture", 2: "FullRosenbrock", 3: "EvenRosenbrock",
         4: "HybridRosenbrock", 5: "IIDGamma", 6: "IIDBeta", 7: "DiagGaussian", 8: "Hypercube", 9: "NealFunnel"}


def rate(kind, pk, dim, T=32, C=16384, target_ms=60.0):
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7)
    spec, x0 = make_spec(H, kind, dim, rng)
    beta = (0.01 ** (np.arange(T) / (T - 1))).astype(f32)
    scale = 2.38**2 / dim * (0.05 if kind in (2, 3, 4, 6) else 1.0)
    if pk == "Normal":
        prop = H.proposal_spec(pk, dim, beta, base_variance_scalar=scale)
    elif pk == "Laplace":
        prop = H.proposal_spec(pk, dim, beta, base_variance_vector=np.full(dim, scale, f32))
    else:
        prop = H.proposal_spec(pk, dim, beta, base_radius=float(np.sqrt(scale * dim)))
    lp0 = f32(O.logdensity(spec.oracle(), x0[None].astype(f32))[0])
    state = torch.tensor(x0, dtype=torch.float32, device=dev).expand(C, T, dim).contiguous()
    logp = torch.full((C, T), float(lp0), device=dev)
    plan = E.RunPlan(spec.engine(dev), prop.engine(dev), state=state, logp=logp, beta=torch.tensor(beta, device=dev), burn_in=0,
                     swap_every=10, swap_mode=E.SWAP_EXCHANGE, swap_order=E.ORDER_SEQUENTIAL, seed=3, chain_offset=0,
                     n_accept=torch.zeros(C, T, dtype=torch.int64, device=dev), sq_jump=torch.zeros(C, T, dtype=torch.float64, device=dev),
                     swap_accept=torch.zeros(C, T, dtype=torch.int64, device=dev),
                     last_swap_ordinal=torch.zeros(C, T, dtype=torch.int64, device=dev))
    inner = 100
    plan.launch(0, inner)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    plan.launch(inner, inner)
    e1.record()
    torch.cuda.synchronize()
    inner2 = int(max(100, min(20000, inner * target_ms / e0.elapsed_time(e1)))) // 10 * 10
    step = 2 * inner
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        plan.launch(step, inner2)
        step += inner2
    e1.record()
    torch.cuda.synchronize()
    return C * T * inner2 * 3 / (e0.elapsed_time(e1) * 1e-3)


def main():
    dims = [int(a) for a in sys.argv[1:]] or [30, 50, 24, 48, 100]
    print(f"{'target':22s} {'proposal':14s} " + " ".join(f"{'dim ' + str(d):>10s}" for d in dims) + "   (chain-MH-steps/s)")
    for kind in (10, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9):
        for pk in ("Normal", "Laplace", "UniformRadius"):
            row = []
            for dim in dims:
                d = dim
                if kind == 3 and d % 2:
                    d += 1
                row.append(rate(kind, pk, d))
            print(f"{NAMES[kind]:22s} {pk:14s} " + " ".join(f"{r:10.3e}" for r in row), flush=True)


if __name__ == "__main__":
    main()
