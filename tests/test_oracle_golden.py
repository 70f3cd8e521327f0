"""Pins the CPU oracle (oracle/ptrwm_oracle.c) to the golden vectors captured from the real reference
(tests/golden/generate_golden.py).  CPU only."""
import numpy as np
import pytest

import helpers as H
from oracle import oracle as O

TARGET_KEYS = ["rc15_d30", "rc5_d30", "rc4_d20", "rc15s_d10", "tm_d50", "tm15_d30", "tms_d10", "full_d30", "full_d10",
               "even_d30", "hyb_3_5", "hyb_5_4", "gamma_d50", "gamma_d5", "beta_d50", "beta_d5",
               "mvn_d50", "mvnd_d8", "smvn_d20", "cube_d5", "cube2_d3", "funnel_d10", "funnel_d1"]


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    assert O.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert O.philox4x32_10([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert O.philox4x32_10([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


@pytest.mark.parametrize("key", TARGET_KEYS)
def test_logdensity_matches_reference(key):
    spec, x, ref, _ = H.golden_targets()[key]
    fin = np.isfinite(ref)
    for prec, rtol in (("f32", 3e-6), ("f64", 3e-6)):
        got = O.logdensity(spec.oracle(), x, prec)
        # support masks (-inf) must agree exactly
        assert np.array_equal(np.isneginf(got), np.isneginf(ref)), key
        scale = np.maximum(1.0, np.abs(ref[fin]))
        # tolerance stated: 3e-6 relative to max(1, |log p|) (the reference itself is fp32); Beta needs 1e-5
        tol = 1e-5 if key.startswith("beta") else rtol
        assert np.max(np.abs(got[fin] - ref[fin]) / scale) < tol, (key, prec)


def test_logdensity_survey_anchors():
    """SURVEY section 8c anchor values (probed from the reference during the survey)."""
    spec, x, ref, _ = H.golden_targets()["rc15_d30"]
    np.testing.assert_allclose(ref[:4], [-85.06665802, -317.99981689, -63.68734741, -67.43735504], rtol=2e-7)
    np.testing.assert_allclose(O.logdensity(spec.oracle(), x[:4], "f64"),
                               [-85.06665037414523, -317.9998048390914, -63.68734012591826, -67.43734012591827],
                               rtol=1e-6)  # NumPy RoughCarpet log(density + 1e-300), fp64


@pytest.mark.parametrize("tag", ["d30_b1.0", "d7_b0.37", "d50_b0.01"])
def test_proposal_transforms_match_reference(tag):
    z = H.load("proposals.npz")
    D = int(tag[1:tag.index("_")])
    raw, inc = z[f"normal_{tag}__raw"], z[f"normal_{tag}__inc"]
    p = H.ProposalSpec(O.PROPOSAL_NORMAL, np.array([z[f"normal_{tag}__std"]]))
    got = O.propose(p.oracle(), D, raw.shape[0], ext_raw=raw[:, None, :])[:, 0]
    assert np.array_equal(got.astype(np.float32), inc)  # one IEEE multiply: bit exact

    raw, inc = z[f"laplace_{tag}__raw"], z[f"laplace_{tag}__inc"]
    p = H.ProposalSpec(O.PROPOSAL_LAPLACE, np.ones(1, np.float32), z[f"laplace_{tag}__scale"])
    got = O.propose(p.oracle(), D, raw.shape[0], ext_raw=raw[:, None, :])[:, 0]
    np.testing.assert_allclose(got, inc, rtol=3e-7, atol=1e-9)  # log1pf vs torch.log1p: <= 2 ulp

    raw, inc = z[f"uniform_{tag}__raw"], z[f"uniform_{tag}__inc"]
    p = H.ProposalSpec(O.PROPOSAL_UNIFORM_RADIUS, np.array([z[f"uniform_{tag}__radius"]]), None, 1.0 / D)
    got = O.propose(p.oracle(), D, raw.shape[0], ext_raw=raw[:, None, :])[:, 0]
    np.testing.assert_allclose(got, inc, rtol=2e-6, atol=1e-9)  # norm accumulation order + powf
    assert np.all(np.linalg.norm(got, axis=1) <= float(z[f"uniform_{tag}__radius"]) * (1 + 1e-6))


RWM_CASES = ["rwm_rc15_normal", "rwm_rc4_normal_beta", "rwm_even_laplace", "rwm_tm_uniform", "rwm_full_normal",
             "rwm_hyb_laplace", "rwm_gamma_normal", "rwm_beta_uniform", "rwm_rc15s_normal", "rwm_tms_normal",
             "rwm_mvn_laplace", "rwm_smvn_normal", "rwm_cube_uniform", "rwm_funnel_normal"]


def rwm_case(name):
    z = H.load(name + ".npz")
    tkey, kind = str(z["target_key"]), str(z["proposal_kind"])
    spec = H.target_spec(tkey)
    beta = float(z["beta"])
    kw = {k[3:]: z[k] for k in z.files if k.startswith("pp_")}
    if "base_variance_scalar" in kw:
        kw["base_variance_scalar"] = float(kw["base_variance_scalar"])
    if "base_radius" in kw:
        kw["base_radius"] = float(kw["base_radius"])
    prop = H.proposal_spec(kind, spec.dim, [beta], single=True, **kw)
    return z, spec, prop, beta


@pytest.mark.parametrize("name", RWM_CASES)
def test_rwm_trajectory_matches_reference(name):
    """Free-running oracle fed the reference's own random tensors reproduces the reference chain."""
    z, spec, prop, beta = rwm_case(name)
    chain, N, burn = z["chain"], int(z["n_samples"]), int(z["burn_in"])
    total = N + burn
    x0 = chain[0][None, None, :]
    lp0 = O.logdensity(spec.oracle(), chain[0][None, :]).astype(np.float32).reshape(1, 1)
    np.testing.assert_allclose(lp0[0, 0], z["logp_chain"][0], rtol=3e-6)
    res = O.run(spec.oracle(), prop.oracle(), state=x0, logp=lp0, beta=[beta], step0=0, n_steps=total, burn_in=burn,
                ext_prop=z["raw"][:, None, None, :], ext_u=z["u"][:, None, None], trace_chains=1, trace_temps=1,
                want_flags=True)
    got = res["trace"][:, 0, 0]
    # accept flags of the reference, recovered from its stored chain (a rejected step repeats the row)
    ref_moved = np.any(chain[1:] != chain[:-1], axis=1)
    first = H.first_mismatch(res["accept_flags"][:, 0, 0].astype(bool), ref_moved)
    assert first is None, f"{name}: accept decision differs from the reference at step {first}"
    # states: the update x + inc is IEEE-exact in both; Laplace/UniformRadius increments differ by <= 2 ulp
    if str(z["proposal_kind"]) == "Normal":
        assert np.array_equal(got, chain[1:])
    else:
        np.testing.assert_allclose(got, chain[1:], rtol=2e-5, atol=2e-6)
    # log-densities are sums of O(10)-sized fp32 terms that cancel (Beta): absolute tolerance 1e-4
    np.testing.assert_allclose(res["trace_logp"][:, 0, 0], z["logp_chain"][1:], rtol=2e-5, atol=1e-4)
    assert int(res["n_accept"][0, 0]) == int(z["num_acceptances"])
    assert res["n_accept"][0, 0] / N == pytest.approx(float(z["acceptance_rate"]), rel=1e-12)
    assert res["sq_jump"][0, 0] / N == pytest.approx(float(z["esjd"]), rel=2e-5)


PT_CASES = ["pt_rc15_geo8", "pt_rc5_fine12", "pt_tm15_t32", "pt_even_t5", "pt_hyb_t4"]


def pt_case(name):
    z = H.load(name + ".npz")
    spec = H.target_spec(str(z["target_key"]))
    ladder = z["beta_ladder"]
    prop = H.proposal_spec("Normal", spec.dim, ladder, base_variance_scalar=float(z["var"]))
    return z, spec, prop, ladder.astype(np.float32)


@pytest.mark.parametrize("name", PT_CASES)
def test_pt_trajectory_matches_reference(name):
    """Oracle in the reference's own swap semantics (sequential order, Q1 row copy) on the reference's random
    tensors reproduces every temperature's chain and the swap statistics."""
    z, spec, prop, beta = pt_case(name)
    chains = z["chains"]  # [T, total+1, D]
    T, rows, D = chains.shape
    total, burn, se = rows - 1, int(z["burn_in"]), int(z["swap_every"])
    x0 = np.ascontiguousarray(chains[:, 0, :])[None]
    lp0 = z["logp_chains"][:, 0][None].astype(np.float32)
    np.testing.assert_allclose(O.logdensity(spec.oracle(), x0[0]), lp0[0], rtol=3e-6)
    res = O.run(spec.oracle(), prop.oracle(), state=x0, logp=lp0, beta=beta, step0=0, n_steps=total, burn_in=burn,
                swap_every=se, swap_mode=O.SWAP_REFERENCE_COPY, swap_order=O.ORDER_SEQUENTIAL,
                ext_prop=z["ext_prop"][:, None], ext_u=z["ext_u"][:, None], ext_swap_u=z["ext_swap_u"][:, None],
                trace_chains=1, trace_temps=T)
    got = res["trace"][:, 0].transpose(1, 0, 2)  # [T, total, D]
    first = H.first_mismatch(got.transpose(1, 0, 2), chains[:, 1:].transpose(1, 0, 2))
    assert first is None, f"{name}: trajectory leaves the reference at step {first}"
    np.testing.assert_allclose(res["trace_logp"][:, 0].T, z["logp_chains"][:, 1:], rtol=2e-5, atol=2e-5)
    n_events = total // se - burn // se
    assert n_events * (T - 1) == int(z["num_swap_attempts"])
    assert int(res["swap_accept"].sum()) == int(z["num_swap_acceptances"])
    last = int(res["last_swap_ordinal"].max())
    # the reference refreshes both statistics only when a swap is accepted (pt_rwm_gpu_optimized.py:627-633)
    assert res["swap_accept"].sum() / last == pytest.approx(float(z["swap_acceptance_rate"]), rel=1e-12)
    b = z["beta_ladder"]
    sq = float((res["swap_accept"][0, :-1] * (b[:-1] - b[1:]) ** 2).sum())
    assert sq / last == pytest.approx(float(z["pt_esjd"]), rel=1e-9)
    N = total - burn
    assert res["sq_jump"][0, 0] / N == pytest.approx(float(z["esjd"]), rel=2e-5)


def test_pt_exchange_differs_from_reference_copy():
    """Q1 is a real behavioural difference: with the same randoms the two swap modes part ways."""
    z, spec, prop, beta = pt_case("pt_rc5_fine12")
    chains = z["chains"]
    T, rows, D = chains.shape
    kw = dict(state=np.ascontiguousarray(chains[:, 0, :])[None], logp=z["logp_chains"][:, 0][None], beta=beta, step0=0,
              n_steps=rows - 1, burn_in=int(z["burn_in"]), swap_every=int(z["swap_every"]),
              ext_prop=z["ext_prop"][:, None], ext_u=z["ext_u"][:, None], ext_swap_u=z["ext_swap_u"][:, None])
    a = O.run(spec.oracle(), prop.oracle(), swap_mode=O.SWAP_EXCHANGE, **kw)
    b = O.run(spec.oracle(), prop.oracle(), swap_mode=O.SWAP_REFERENCE_COPY, **kw)
    assert int(b["swap_accept"].sum()) > 0
    assert not np.array_equal(a["state"], b["state"])
    # exchange conserves the multiset of rows at a swap; copy duplicates rows
    assert len({tuple(r) for r in a["state"][0]}) == T


def test_standalone_swap_sweep_matches_reference():
    """`_attempt_all_swaps()` called on its own (pt_rwm_gpu_optimized.py:594-633): the oracle's sweep reproduces the
    reference's states, log-densities and counters on 48 random ladders, bit for bit (rows are only copied)."""
    f = H.load("pt_sweep.npz")
    beta = f["beta_ladder"].astype(np.float32)
    T = len(beta)
    for prec in ("f32", "f64"):
        got = O.swap_sweep(state=f["state_in"], logp=f["logp_in"], beta=beta, event_index=0,
                           swap_mode=O.SWAP_REFERENCE_COPY, swap_order=O.ORDER_SEQUENTIAL, ext_swap_u=f["swap_u"],
                           precision=prec)
        assert np.array_equal(got["state"], f["state_out"])
        assert np.array_equal(got["logp"], f["logp_out"])
        acc = got["swap_accept"].sum(1)
        assert np.array_equal(acc, f["num_swap_acceptances"])
        assert np.all(f["num_swap_attempts"] == T - 1)
        # the reference refreshes its rate only when a swap is accepted (:627-633): acc / ordinal of the last accept
        last = got["last_swap_ordinal"].max(1)
        assert np.allclose(acc / np.maximum(last, 1), f["swap_acceptance_rate"], rtol=0, atol=1e-15)
    # exchange mode differs from the reference's row copy exactly where a swap was accepted: it is a permutation
    ex = O.swap_sweep(state=f["state_in"], logp=f["logp_in"], beta=beta, event_index=0, ext_swap_u=f["swap_u"])
    assert np.array_equal(np.sort(ex["logp"], axis=1), np.sort(f["logp_in"], axis=1))
    assert not np.array_equal(ex["state"], f["state_out"])
