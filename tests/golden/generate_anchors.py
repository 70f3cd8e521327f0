"""Statistical anchors from the REAL reference (aidanmrli/rwm-pt-pytorch at /root/reference), run on CPU in the build
container: long-run acceptance rate and ESJD of its torch samplers on the BASELINE target, with standard errors
over independent runs, so a GPU test can check the north-star parity bound (acceptance / ESJD within 1e-3 relative)
against the reference itself rather than against the restatement.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_anchors.py [n_procs] [rwm_steps] [pt_steps]
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_anchors.py extend [n_procs] [runs_per_family]

Writes tests/golden/reference_anchors.json (data only).  `extend` (round 2) ADDS to that file: RWM anchors for the
families of BASELINE configs[3] / [4] - EvenRosenbrock d=30 with the Laplace proposal, ThreeMixture d=50 with the
UniformRadius proposal (rwm_gpu_optimized.py:402-488 with `proposal_distribution=`) - and more PT runs of the
BASELINE target (new seeds, appended to the existing ones)."""
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


FAMILIES = {  # extra RWM anchor families: (target class, dim, target kwargs, proposal class, proposal scale argument)
    "rwm_even_laplace": ("EvenRosenbrockTorch", 30, {}, "LaplaceProposal", 0.004),
    "rwm_tm_uniform": ("ThreeMixtureDistributionTorch", 50, {}, "UniformRadiusProposal", 2.4),
}


def _run_family(job):
    fam, seed, n, burn = job
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import numpy as np
    import torch

    torch.set_num_threads(1)
    cls, dim, tkw, pcls, scale = FAMILIES[fam]
    with contextlib.redirect_stdout(io.StringIO()):
        import algorithms as ref_alg
        import proposal_distributions as ref_prop
        import target_distributions as ref_tgt

        target = getattr(ref_tgt, cls)(dim, device="cpu", **tkw)
        arg = torch.full((dim,), scale) if pcls == "LaplaceProposal" else scale
        prop = getattr(ref_prop, pcls)(dim, arg, 1.0, torch.device("cpu"), torch.float32)
        np.random.seed(seed)  # the initial state (1e-8 N(0,1) for the Rosenbrock family) comes from the global NumPy RNG
        torch.manual_seed(seed)
        t0 = time.time()
        alg = ref_alg.RandomWalkMH_GPU_Optimized(dim=dim, target_dist=target, burn_in=burn, device="cpu",
                                                 pre_allocate_steps=n, proposal_distribution=prop)
        alg.generate_samples(n)
        out = {"acceptance_rate": float(alg.acceptance_rate), "esjd": float(alg.expected_squared_jump_distance_gpu())}
    out["seconds"] = time.time() - t0
    return fam, seed, out


def _stats(runs, key):
    import numpy as np

    v = np.array([r[key] for r in runs])
    return {"mean": float(v.mean()), "stderr": float(v.std(ddof=1) / np.sqrt(len(v))), "n_runs": len(v)}


def extend():
    n_procs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    n_runs = int(sys.argv[3]) if len(sys.argv) > 3 else 14
    path = os.path.join(OUT, "reference_anchors.json")
    with open(path) as f:
        summary = json.load(f)
    burn = summary["burn_in"]
    fam_steps = 1_000_000
    jobs = [(fam, 3000 + 100 * i + j, fam_steps, burn) for i, fam in enumerate(FAMILIES) for j in range(n_runs)]
    have = {r["seed"] for r in summary["pt"]["runs"]}
    pt_jobs = [("pt", s, summary["pt"]["steps_per_run"], burn) for s in range(2100, 2100 + 2 * n_runs) if s not in have]
    with mp.get_context("spawn").Pool(n_procs) as pool:
        fam_res = pool.map(_run_family, jobs, chunksize=1)
        pt_res = pool.map(_run, pt_jobs, chunksize=1)
    for fam, (cls, dim, tkw, pcls, scale) in FAMILIES.items():
        runs = [{"seed": seed, **out} for f, seed, out in fam_res if f == fam]
        summary[fam] = {"target": f"{cls}({dim})", "proposal": f"{pcls}(scale argument {scale}, beta 1)", "dim": dim,
                        "proposal_scale": scale, "steps_per_run": fam_steps, "runs": runs,
                        "acceptance_rate": _stats(runs, "acceptance_rate"), "esjd": _stats(runs, "esjd")}
        print(fam, summary[fam]["acceptance_rate"], summary[fam]["esjd"])
    summary["pt"]["runs"] += [{"seed": seed, **out} for _, seed, out in pt_res]
    for k in ("swap_accept_fraction", "cold_esjd"):
        summary["pt"][k] = _stats(summary["pt"]["runs"], k)
    print("pt", summary["pt"]["swap_accept_fraction"], summary["pt"]["cold_esjd"])
    with open(path, "w") as f:
        json.dump(summary, f, indent=1)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "extend":
        return extend()
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
