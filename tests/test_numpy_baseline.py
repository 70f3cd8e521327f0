"""The NumPy restatement of the reference's CPU samplers (oracle/numpy_baseline.py) reproduces numbers obtained
by running the real reference (tests/golden/generate_golden.py): same seeds, same RNG stream, exact equality."""
import json
import os

import numpy as np

import helpers as H
from oracle import numpy_baseline as NB


def test_rwm_short_chain_is_identical():
    z = H.load("numpy_rwm_d30.npz")
    alg = NB.run_rwm(30, 2.38**2 / 30, 3000, seed=7)
    assert np.array_equal(np.array(alg.chain), z["chain"])
    assert alg.acceptance_rate == float(z["acceptance_rate"])
    assert NB.esjd(alg.chain) == float(z["esjd"])


def test_pt_short_run_is_identical():
    z = H.load("numpy_pt_d30.npz")
    alg = NB.run_pt(30, 2.38**2 / 30, list(z["beta_ladder"]), 1500, seed=11)
    assert np.array_equal(np.array(alg.chain), z["chain"])
    assert np.array_equal(np.array([c.chain[-1] for c in alg.chains]), z["final_states"])
    assert alg.num_swap_attempts == int(z["num_swap_attempts"])
    assert alg.num_acceptances == int(z["num_swap_acceptances"])
    assert alg.acceptance_rate == float(z["swap_acceptance_rate"])
    assert alg.pt_esjd == float(z["pt_esjd"])


def test_baseline_config1_scalars():
    """BASELINE.json configs[0]: RWM NumPy, RoughCarpet dim=20, 1 chain, 100 000 iterations, seed 42."""
    with open(os.path.join(H.GOLDEN, "numpy_baseline.json")) as f:
        g = json.load(f)["config1"]
    assert (g["num_acceptances"], g["acceptance_rate"], g["esjd"]) == (24841, 0.24840751592484075, 1.2806414154613754)
    alg = NB.run_rwm(20, 2.38**2 / 20, 100000, seed=42)
    assert alg.num_acceptances == 24841
    assert alg.acceptance_rate == 0.24840751592484075
    assert NB.esjd(alg.chain) == 1.2806414154613754
