"""Statistical anchors from the REAL reference (aidanmrli/rwm-pt-pytorch at /root/reference), run on CPU in the build
container: long-run acceptance rate and ESJD of its torch samplers on the BASELINE target, with standard errors
over independent runs, so a GPU test can check the north-star parity bound (acceptance / ESJD within 1e-3 relative)
against the reference itself rather than against the restatement.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_anchors.py [n_procs] [rwm_steps] [pt_steps]

Writes tests/golden/reference_anchors.json (data only)."""
import contextlib
import io
import json
import multiprocessing as mp
import os
import sys
import time

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
DIM = 30
VAR = 2.38**2 / DIM
GEO8 = [1.0, 0.5, 0.25, 0.125, 0.0625, 0.03125, 0.015625, 0.01]


def _run(job):
    kind, seed, n, burn = job
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import numpy as np
    import torch

    torch.set_num_threads(1)
    with contextlib.redirect_stdout(io.StringIO()):
        import algorithms as ref_alg
        import target_distributions as ref_tgt

        target = ref_tgt.RoughCarpetDistributionTorch(DIM, device="cpu", mode_centers=[-15.0, 0.0, 15.0])
        np.random.seed(seed)
        torch.manual_seed(seed)
        t0 = time.time()
        if kind == "rwm":
            alg = ref_alg.RandomWalkMH_GPU_Optimized(DIM, VAR, target, burn_in=burn, device="cpu", pre_allocate_steps=n)
            alg.generate_samples(n)
            out = {"acceptance_rate": float(alg.acceptance_rate), "esjd": float(alg.expected_squared_jump_distance_gpu())}
        else:
            alg = ref_alg.ParallelTemperingRWM_GPU_Optimized(DIM, VAR, target, True, beta_ladder=GEO8, swap_every=10,
                                                             burn_in=burn, device="cpu", pre_allocate_steps=n)
            alg.generate_samples(n)
            out = {"swap_accept_fraction": alg.num_swap_acceptances / max(1, alg.num_swap_attempts),
                   "num_swap_attempts": int(alg.num_swap_attempts),
                   "cold_esjd": float(alg.expected_squared_jump_distance_gpu())}
    out["seconds"] = time.time() - t0
    return kind, seed, out


def main():
    n_procs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rwm_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400_000
    pt_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
    burn = 1000
    jobs = [("rwm", 1000 + i, rwm_steps, burn) for i in range(n_procs * 2)] + \
           [("pt", 2000 + i, pt_steps, burn) for i in range(n_procs * 2)]
    import numpy as np

    with mp.get_context("spawn").Pool(n_procs) as pool:
        res = pool.map(_run, jobs, chunksize=1)
    summary = {"target": "RoughCarpetDistributionTorch(30, mode_centers=[-15,0,15])", "var": VAR, "burn_in": burn,
               "rwm": {"steps_per_run": rwm_steps, "runs": []}, "pt": {"steps_per_run": pt_steps, "runs": [],
                                                                        "beta_ladder": GEO8, "swap_every": 10}}
    for kind, seed, out in res:
        summary[kind]["runs"].append({"seed": seed, **out})
    for kind, keys in (("rwm", ("acceptance_rate", "esjd")), ("pt", ("swap_accept_fraction", "cold_esjd"))):
        for k in keys:
            v = np.array([r[k] for r in summary[kind]["runs"]])
            summary[kind][k] = {"mean": float(v.mean()), "stderr": float(v.std(ddof=1) / np.sqrt(len(v))), "n_runs": len(v)}
    with open(os.path.join(OUT, "reference_anchors.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps({k: summary[k] for k in ("rwm", "pt")}, default=str)[:600])


if __name__ == "__main__":
    main()
