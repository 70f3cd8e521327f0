"""A TIGHT statistical anchor for the one statistic the reference itself cannot pin within the round: the PT cold-chain
ESJD of the BASELINE target (reference_anchors.json: +-0.4 % after 42 reference runs; 3e-4 would take ~200x more
reference CPU time).  The C oracle (oracle/ptrwm_oracle.c) reproduces the reference's trajectories decision for decision
on shared randoms (tests/test_oracle_golden.py), so run in Philox mode it samples the same Markov chain; it is ~50x
faster than the reference's Python loop.  This script runs it on all host cores:

    python tests/golden/generate_oracle_anchor.py [n_procs] [ladders_per_proc] [steps]

and writes tests/golden/oracle_anchor_pt.json: mean and standard error (over independent ladders) of the cold-chain ESJD
and of the swap-acceptance fraction, same schedule as the reference anchor (RoughCarpet d=30 modes +-15, var 2.38^2/30,
8 geometric temperatures, swap_every 10, burn-in 1000, the reference's row-copy swap).  TEST INFRASTRUCTURE."""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
GEO8 = [1.0, 0.5, 0.25, 0.125, 0.0625, 0.03125, 0.015625, 0.01]
DIM, BURN, SE, SEED = 30, 1000, 10, 424242


def _work(job):
    proc, n_ladders, steps = job
    from oracle import oracle as O

    lw = np.log(np.array([0.5, 0.3, 0.2], dtype=np.float32))
    tgt = O.Target(O.TARGET_ROUGH_CARPET, DIM, p=[-15.0, 0.0, 15.0, *lw, 0.0])
    beta = np.asarray(GEO8, np.float32)
    prop = O.Proposal(O.PROPOSAL_NORMAL, np.sqrt((2.38**2 / DIM / beta.astype(np.float64)).astype(np.float32)))
    st = np.zeros((n_ladders, 8, DIM), np.float32)
    lp = np.tile(O.logdensity(tgt, np.zeros((1, DIM), np.float32)).astype(np.float32), (n_ladders, 8))
    r = O.run(tgt, prop, state=st, logp=lp, beta=beta, step0=0, n_steps=BURN + steps, burn_in=BURN, swap_every=SE,
              swap_mode=O.SWAP_REFERENCE_COPY, seed=SEED, chain_offset=proc * 1_000_000)
    events = (BURN + steps) // SE - BURN // SE
    return (r["sq_jump"][:, 0] / steps).tolist(), (r["swap_accept"].sum(1) / (events * 7)).tolist(), \
        (r["n_accept"][:, 0] / steps).tolist()


def main():
    n_procs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400_000
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    t0 = time.time()
    esjd, frac, acc = [], [], []
    with mp.get_context("spawn").Pool(n_procs) as pool:
        for k in range(rounds):
            for e, f, a in pool.map(_work, [(k * n_procs + p, per, steps) for p in range(n_procs)], chunksize=1):
                esjd += e
                frac += f
                acc += a
            n = len(esjd)
            out = {"target": "RoughCarpet d=30 modes +-15", "beta_ladder": GEO8, "swap_every": SE, "burn_in": BURN,
                   "steps_per_ladder": steps, "n_ladders": n, "swap_mode": "reference_copy", "seed": SEED,
                   "engine": "oracle/ptrwm_oracle.c (fp32 build), Philox mode", "cpu_seconds": time.time() - t0}
            for name, v in (("cold_esjd", esjd), ("swap_accept_fraction", frac), ("cold_acceptance_rate", acc)):
                v = np.asarray(v)
                out[name] = {"mean": float(v.mean()), "stderr": float(v.std(ddof=1) / np.sqrt(n)), "n": n}
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_anchor_pt.json"), "w") as f:
                json.dump(out, f, indent=1)
            print(k, n, out["cold_esjd"], f"{time.time() - t0:.0f}s", flush=True)


if __name__ == "__main__":
    main()
