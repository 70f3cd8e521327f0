"""Development aid (needs a GPU): how far do the HIP kernel's Philox-mode states drift from the oracle's on the same
stream while every decision still agrees?  Prints, per family, the largest |state difference| relative to the state
tolerance of tests/helpers.check_parity (1e-4 |x| + 2e-5) after 10 / 30 / 100 / 300 agreeing steps - the basis of
PHILOX_SEGMENT in tests/test_gpu_engine_parity.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rwm-pt-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import helpers as H  # noqa: E402
from oracle import oracle as O  # noqa: E402
from test_gpu_engine_parity import SWEEP, start_state  # noqa: E402

dev = torch.device("cuda:0")
N = 300
for tkey, pkind, T, Cn, pkw in SWEEP:
    spec = H.target_spec(tkey)
    beta = (0.05 ** (np.arange(T) / max(1, T - 1))).astype(np.float32)
    prop = H.proposal_spec(pkind, spec.dim, beta, **pkw)
    st, lp = start_state(spec, Cn, T, np.random.default_rng(11))
    kw = dict(state=st, logp=lp, beta=beta, step0=0, n_steps=N, burn_in=0, swap_every=5, seed=12345, chain_offset=7)
    got = H.gpu_runner(spec, prop, dev)(**kw)
    want = O.run(spec.oracle(), prop.oracle(), trace_chains=Cn, trace_temps=T, want_flags=True, **kw)
    same = np.all(got["accept_flags"] == want["accept_flags"], axis=2)  # [N, Cn]
    alive = np.cumprod(same, axis=0).astype(bool)
    tolr = 1e-4 * np.abs(want["trace"]) + 2e-5
    rel = (np.abs(got["trace"] - want["trace"]) / tolr).reshape(N, Cn, -1).max(axis=2)  # [N, Cn] in tolerance units
    # a swap flip shows as a state difference without a flag difference: ladders stay "alive" only while within 1000 tol
    alive &= np.cumprod(rel < 1000, axis=0).astype(bool)
    out = []
    for n in (10, 30, 100, 300):
        m = alive[n - 1]
        out.append(f"{n}: {rel[:n][:, m].max() if m.any() else float('nan'):7.3f} ({int(m.sum())}/{Cn})")
    print(f"{tkey:10s} {pkind:14s} T={T:3d}  max drift / tolerance after n agreeing steps  " + "  ".join(out), flush=True)
