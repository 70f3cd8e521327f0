"""Iterative temperature ladders built by the REAL reference (aidanmrli/rwm-pt-pytorch at /root/reference,
algorithms/pt_rwm_gpu_optimized.py:283-426), run on CPU in the build container: for every (target, target swap
acceptance) the ladders of N_SEEDS independent constructions, so a GPU test can compare the ladders the engine builds
(`_construct_iterative_ladder`, log-densities through the HIP kernel) rung by rung against the reference's own
distribution of ladders.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_ladders.py [n_seeds]

Writes tests/golden/reference_ladders.json (data only: the beta values of every ladder, the constructor arguments
that produced them, per-rung mean / sd)."""
import contextlib
import io
import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

with contextlib.redirect_stdout(io.StringIO()):
    import algorithms as ref_alg  # noqa: E402
    import target_distributions as ref_tgt  # noqa: E402

C15 = [[-15.0] + [0.0] * 29, [0.0] * 30, [15.0] + [0.0] * 29]
# key -> (class name, constructor kwargs, dim): the PT driver's targets (experiment_pt_GPU.py:36,48-52)
TARGETS = {
    "rc15_d10": ("RoughCarpetDistributionTorch", dict(mode_centers=[-15.0, 0.0, 15.0]), 10),
    "rc15_d30": ("RoughCarpetDistributionTorch", dict(mode_centers=[-15.0, 0.0, 15.0]), 30),
    "tm15_d30": ("ThreeMixtureDistributionTorch", dict(mode_centers=C15), 30),
}
SWAP_TARGETS = [0.15, 0.234, 0.35]
N_EST = 3000  # the class default N_samples_swap_est (the Slurm script of the reference uses more)


def build(tkey, swap_target, seed):
    cls, kw, dim = TARGETS[tkey]
    with contextlib.redirect_stdout(io.StringIO()):
        target = getattr(ref_tgt, cls)(dim, device="cpu", **kw)
        np.random.seed(seed)
        torch.manual_seed(seed)
        alg = ref_alg.ParallelTemperingRWM_GPU_Optimized(
            dim, 2.38**2 / dim, target, True, iterative_temp_spacing=True, swap_acceptance_rate=swap_target,
            N_samples_swap_est=N_EST, swap_every=10, burn_in=0, device="cpu", pre_allocate_steps=4)
    return [float(b) for b in alg.beta_ladder]


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    out = {"n_samples_swap_est": N_EST, "var_rule": "2.38^2/dim", "targets": {k: {"class": v[0], "kwargs": v[1], "dim": v[2]}
                                                                              for k, v in TARGETS.items()},
           "cases": {}}
    for tkey in TARGETS:
        for a in SWAP_TARGETS:
            ladders = [build(tkey, a, 9000 + s) for s in range(n_seeds)]
            lens = np.array([len(l) for l in ladders])
            n_common = int(lens.min())
            arr = np.array([l[:n_common] for l in ladders])
            case = {
                "swap_target": a, "seeds": [9000 + s for s in range(n_seeds)], "ladders": ladders,
                "length_mean": float(lens.mean()), "length_sd": float(lens.std(ddof=1)),
                "rung_mean": arr.mean(0).tolist(), "rung_sd": arr.std(0, ddof=1).tolist(),
            }
            out["cases"][f"{tkey}@{a}"] = case
            print(f"{tkey} a={a}: length {lens.mean():.2f} +- {lens.std(ddof=1):.2f}, first rungs "
                  f"{np.round(arr.mean(0)[:5], 4).tolist()}", flush=True)
    with open(os.path.join(OUT, "reference_ladders.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
