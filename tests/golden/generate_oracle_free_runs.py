"""Free-running (never restarted) Philox-mode runs of the C oracle for the BASELINE families, as anchors for the engine's
PRODUCTION path over a long horizon (VERDICT r03 #5b: the segment-restarted proofs of helpers.check_parity_philox show every
decision, not what a production run looks like after 10^4 steps on its own).

    python tests/golden/generate_oracle_free_runs.py [n_procs]      ->  tests/golden/oracle_free_runs.json  (~10 CPU-minutes)

Per family: 11 000 steps (burn-in 1 000), the engine's Philox counter layout (oracle/ptrwm_oracle.c restates it), seed and
ladder ids as the test uses them; stored: mean and standard error over ladders of the per-temperature Metropolis
acceptance rate, the per-temperature mean squared jump and the swap-acceptance fraction.  TEST INFRASTRUCTURE (the oracle is
the checker; nothing on the product path reads this)."""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
BURN, STEPS, SE, SEED = 1000, 10_000, 10, 20261005

# name -> (golden target key, proposal, proposal kwargs, temperatures, ladders)   [BASELINE.json configs[1..4]]
FAMILIES = {
    "configs1_rwm_rc15_normal": ("rc15_d30", "Normal", dict(base_variance_scalar=2.38**2 / 30), 1, 8192),
    "configs2_pt_rc15_normal": ("rc15_d30", "Normal", dict(base_variance_scalar=2.38**2 / 30), 32, 256),
    "configs3_pt_even_laplace": ("even_d30", "Laplace", dict(base_variance_vector=np.full(30, 0.004)), 32, 256),
    "configs4_pt_tm50_uniform": ("tm_d50", "UniformRadius", dict(base_radius=2.4), 64, 128),
}


def ladder(T):
    return (0.01 ** (np.arange(T) / (T - 1))).astype(np.float32) if T > 1 else np.ones(1, np.float32)


def start(spec, Cn, T):
    """Every replica from the family's start (zeros; 1e-8 N(0,1) for the Rosenbrock family), as the samplers do."""
    from oracle import oracle as O

    x0 = 1e-8 * np.random.default_rng(7).standard_normal(spec.dim) if "Rosenbrock" in spec.cls else np.zeros(spec.dim)
    st = np.broadcast_to(x0.astype(np.float32), (Cn, T, spec.dim)).copy()
    lp = np.broadcast_to(O.logdensity(spec.oracle(), x0[None].astype(np.float32)).astype(np.float32), (Cn, T)).copy()
    return st, lp


def _work(job):
    name, first, count = job
    import helpers as H
    from oracle import oracle as O

    tkey, pkind, pkw, T, _ = FAMILIES[name]
    spec = H.target_spec(tkey)
    beta = ladder(T)
    prop = H.proposal_spec(pkind, spec.dim, beta, **pkw) if T > 1 else H.proposal_spec(pkind, spec.dim, [1.0], single=True, **pkw)
    st, lp = start(spec, count, T)
    r = O.run(spec.oracle(), prop.oracle(), state=st, logp=lp, beta=beta, step0=0, n_steps=BURN + STEPS, burn_in=BURN,
              swap_every=SE, seed=SEED, chain_offset=first)
    return name, r["n_accept"] / STEPS, r["sq_jump"] / STEPS, r["swap_accept"]


def main():
    n_procs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    t0 = time.time()
    jobs = []
    for name, (_, _, _, T, Cn) in FAMILIES.items():
        per = max(1, Cn // (4 * n_procs))
        jobs += [(name, f, min(per, Cn - f)) for f in range(0, Cn, per)]
    acc, sq, sw = {n: [] for n in FAMILIES}, {n: [] for n in FAMILIES}, {n: [] for n in FAMILIES}
    with mp.get_context("spawn").Pool(n_procs) as pool:
        for name, a, q, s in pool.imap(_work, jobs, chunksize=1):
            acc[name].append(a)
            sq[name].append(q)
            sw[name].append(s)
    out = {"burn_in": BURN, "steps": STEPS, "swap_every": SE, "seed": SEED, "ladder": "beta_t = 0.01^(t/(T-1))",
           "engine": "oracle/ptrwm_oracle.c (fp32 build), Philox mode, one free run per ladder", "families": {}}
    for name, (tkey, pkind, pkw, T, Cn) in FAMILIES.items():
        a, q, s = np.concatenate(acc[name]), np.concatenate(sq[name]), np.concatenate(sw[name])
        events = (BURN + STEPS) // SE - BURN // SE
        frac = s[:, :max(T - 1, 1)].sum(1) / (events * max(T - 1, 1))
        n = a.shape[0]
        out["families"][name] = {
            "target": tkey, "proposal": pkind, "temps": T, "ladders": n,
            "acceptance": {"mean": a.mean(0).tolist(), "stderr": (a.std(0, ddof=1) / np.sqrt(n)).tolist()},
            "mean_sq_jump": {"mean": q.mean(0).tolist(), "stderr": (q.std(0, ddof=1) / np.sqrt(n)).tolist()},
            "swap_fraction": {"mean": float(frac.mean()), "stderr": float(frac.std(ddof=1) / np.sqrt(n))},
        }
        print(name, n, "cold acceptance %.5f +- %.5f, cold mean sq jump %.5f +- %.5f, swap fraction %.5f" % (
            a[:, 0].mean(), a[:, 0].std(ddof=1) / np.sqrt(n), q[:, 0].mean(), q[:, 0].std(ddof=1) / np.sqrt(n), frac.mean()), flush=True)
    out["cpu_seconds"] = time.time() - t0
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_free_runs.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
