"""Classic against streaming form of the step kernel through the C ABI (development aid; ptrwm_set_stream_mode).

    python tools/stream_ab.py [--chains 65536[,131072,...]] [--temps 32] [--dim 30] [--steps 1,2,4,8,16,32] [--launches 300]

For each launch length: both forms from the same initial state and seed - bitwise comparison of everything a launch
writes - then ms per launch of each (HIP events around a train of launches), the SURVEY 8(d) fraction of the HBM peak
((8 dim + 24) B per chain-step) for one-step launches, and chain-steps/s.  Output: one JSON object per line.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rwm-pt-pytorch_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", default="65536", help="ladders (comma-separated list: a sweep)")
    ap.add_argument("--temps", type=int, default=32)
    ap.add_argument("--dim", type=int, default=30)
    ap.add_argument("--steps", default="1,2,4,8,16,32")
    ap.add_argument("--launches", type=int, default=300)
    ap.add_argument("--target", default="rough_carpet", choices=["rough_carpet", "even_rosenbrock", "three_mixture"])
    args = ap.parse_args()

    import numpy as np
    import torch

    import ptrwm_hip as P
    from algorithms import ParallelTemperingRWM_GPU_Optimized, RandomWalkMH_GPU_Optimized, geometric_beta_ladder
    from proposal_distributions import LaplaceProposal, UniformRadiusProposal
    from target_distributions import EvenRosenbrockTorch, RoughCarpetDistributionTorch, ThreeMixtureDistributionTorch

    dev = torch.device("cuda:0")
    T, D = args.temps, args.dim

    def make(C):
        np.random.seed(7)
        prop = None
        if args.target == "rough_carpet":
            tgt = RoughCarpetDistributionTorch(D, device=dev, mode_centers=[-15.0, 0.0, 15.0])
        elif args.target == "even_rosenbrock":
            tgt = EvenRosenbrockTorch(D, device=dev)
            prop = LaplaceProposal(D, torch.full((D,), 0.004), 1.0, dev, torch.float32)
        else:
            tgt = ThreeMixtureDistributionTorch(D, device=dev)
            prop = UniformRadiusProposal(D, 2.4, 1.0, dev, torch.float32)
        if T == 1:
            alg = RandomWalkMH_GPU_Optimized(D, None if prop is not None else 2.38**2 / D, tgt, burn_in=0, device=dev,
                                             num_chains=C, seed=42, proposal_distribution=prop)
        else:
            alg = ParallelTemperingRWM_GPU_Optimized(D, 2.38**2 / D, tgt, beta_ladder=geometric_beta_ladder(T), swap_every=10,
                                                     burn_in=0, device=dev, num_replicas=C, seed=42, trace="none",
                                                     proposal_distribution=prop)
        alg._ensure_started()
        return alg

    def timed(run, n, inner):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            run.advance(inner)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    P.set_kernel_form(P.FORM_THREAD)
    for C, inner in [(int(c), int(x)) for c in args.chains.split(",") for x in args.steps.split(",")]:
        a, b = make(C), make(C)
        with P.stream_mode(P.STREAM_OFF):
            for _ in range(23):
                a._run.advance(inner)
        with P.stream_mode(P.STREAM_ON):
            for _ in range(23):
                b._run.advance(inner)
        torch.cuda.synchronize()
        ra, rb = a._run, b._run
        mism = {k: int((getattr(ra, k) != getattr(rb, k)).sum().item())
                for k in ("state", "logp", "n_accept", "sq_jump", "swap_accept", "last_ord")}
        n = max(20, args.launches // inner)
        res = {"chains": C, "temps": T, "dim": D, "target": args.target, "steps_per_launch": inner, "mismatch": mism,
               "accepted": int(ra.n_accept.sum().item())}
        for name, mode, r in (("classic", P.STREAM_OFF, ra), ("stream", P.STREAM_ON, rb)):
            with P.stream_mode(mode):
                timed(r, 10, inner)
                ms = sorted(timed(r, n, inner) for _ in range(3))
            res[name + "_ms"] = ms
            res[name + "_chain_steps_per_s"] = C * T * inner / (ms[1] * 1e-3)
            if inner == 1:
                res[name + "_frac_8d24"] = (8 * D + 24) * C * T / (ms[1] * 1e-3) / 8e12
        print(json.dumps(res), flush=True)
        del a, b


if __name__ == "__main__":
    main()
