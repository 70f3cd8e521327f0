"""Parity of the HIP engine, called through its C ABI (include/ptrwm.h), against the reference's golden
vectors and against the CPU oracle on seeded inputs.  Needs an MI355X: run with `-m gpu`.

Tolerances (stated once):
  * integer / index work (Philox words, accept counts given equal decisions, swap bookkeeping): exact
  * state updates x + scale*z (Normal proposal): bit-exact IEEE single
  * log-densities: |gpu - fp64 oracle| <= 4e-6 * max(1, |log p|) + 1e-4  (fp32 evaluation; v_exp/v_log 1 ulp)
  * accept / swap decisions: can only differ where |u - exp(r)| is inside that error.  Every run against the oracle
    goes through helpers.check_parity: the FULL horizon is compared; at the first differing decision of a ladder the
    step's beta (l' - l) is recomputed in fp64 from the last agreed state and |u - exp(r)| <= exp(r) beta 2 (4e-6
    max(1, |l|) + 1e-4) is ASSERTED, then both engines restart from the oracle's state and the comparison continues
    (tests/test_parity_checker.py shows that a wrong decision at step 15 fails).  Golden trajectories are also
    replayed one step at a time ("teacher forced"), agreement >= 99.9 % required.
"""
import os
import zlib

import numpy as np
import pytest
import torch

import helpers as H
import ptrwm_hip as E
from oracle import oracle as O
from test_oracle_golden import PT_CASES, RWM_CASES, TARGET_KEYS, pt_case, rwm_case

pytestmark = pytest.mark.gpu


dev_t, gpu_run, gpu_runner = H.dev_t, H.gpu_run, H.gpu_runner  # the engine through its C ABI (tests/helpers.py)


# Philox-mode comparisons restart both engines from the oracle's own trajectory every PHILOX_SEGMENT steps: the two
# agree on each proposal to a few ulp only (hardware sin / cos / log against libm), and the state tolerance of
# helpers.check_parity (1e-4 relative + 2e-5) is meant per step, not for the drift of a long horizon
PHILOX_SEGMENT = 30


def logp_close(got, want, extra_abs=1e-4):
    want = np.asarray(want, dtype=np.float64)
    fin = np.isfinite(want)
    assert np.array_equal(np.isneginf(got), np.isneginf(want))
    assert np.array_equal(np.isnan(got), np.isnan(want))
    err = np.abs(np.asarray(got, np.float64)[fin] - want[fin])
    assert np.all(err <= 4e-6 * np.maximum(1, np.abs(want[fin])) + extra_abs), float(err.max())


# ---------------------------------------------------------------------------------------------------------
def test_philox_known_answers(device):
    for ctr, key, want in (
        ((0, 0, 0, 0), 0, (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
        ((0xFFFFFFFF,) * 4, 0xFFFFFFFFFFFFFFFF, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
        ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0x299F31D0 << 32) | 0xA4093822,
         (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
    ):
        got = E.philox_raw(key, *ctr, 1, device).cpu().numpy()[0]
        assert tuple(int(v) for v in got) == want


def test_philox_blocks_match_oracle(device):
    got = E.philox_raw(0x0123456789ABCDEF, 5, 77, 123456, 9, 4096, device).cpu().numpy()
    for i in (0, 1, 63, 64, 4095):
        assert list(got[i]) == O.philox4x32_10([5 + i, 77, 123456, 9], [0x89ABCDEF, 0x01234567])


@pytest.mark.parametrize("key", TARGET_KEYS)
def test_logdensity_matches_reference_and_oracle(device, key):
    spec, x, ref, _ = H.golden_targets()[key]
    got = E.logdensity(spec.engine(device), dev_t(x, device)).cpu().numpy()
    logp_close(got, ref)                                          # the reference's own fp32 numbers
    logp_close(got, O.logdensity(spec.oracle(), x, "f64"))        # fp64 truth


@pytest.mark.parametrize("key", ["rc15_d30", "tm_d50", "even_d30", "hyb_5_4", "gamma_d50", "beta_d50"])
def test_logdensity_large_batch_vs_oracle(device, key):
    """100 000 random states (the survey's tolerance experiment, section 8c)."""
    spec = H.target_spec(key)
    rng = np.random.default_rng(5)
    n = 100_000
    if key.startswith("beta"):
        x = rng.uniform(0.001, 0.999, (n, spec.dim))
    elif key.startswith("gamma"):
        x = rng.gamma(2.0, 3.0, (n, spec.dim))
    elif key.startswith(("even", "hyb")):
        x = rng.normal(0.8, 0.6, (n, spec.dim))
    else:
        x = rng.choice([-15.0, -5.0, 0.0, 5.0, 15.0], (n, spec.dim)) + rng.normal(0, 1.5, (n, spec.dim))
    x = x.astype(np.float32)
    got = E.logdensity(spec.engine(device), dev_t(x, device)).cpu().numpy()
    logp_close(got, O.logdensity(spec.oracle(), x, "f64"))


def test_logdensity_edge_values(device):
    spec = H.target_spec("rc15_d30")
    x = np.zeros((4, 30), np.float32)
    x[1, 3] = np.inf
    x[2, 0] = np.nan
    x[3] = 1e30
    got = E.logdensity(spec.engine(device), dev_t(x, device)).cpu().numpy()
    # torch.logsumexp semantics: every component -inf -> -inf (not NaN); NaN input -> NaN
    assert np.isfinite(got[0]) and got[1] == -np.inf and np.isnan(got[2]) and got[3] == -np.inf
    assert E.logdensity(spec.engine(device), torch.zeros(0, 30, device=device)).shape == (0,)


@pytest.mark.parametrize("tag", ["d30_b1.0", "d7_b0.37", "d50_b0.01"])
def test_proposal_transforms_match_reference(device, tag):
    z = H.load("proposals.npz")
    D = int(tag[1:tag.index("_")])
    raw, inc = z[f"normal_{tag}__raw"], z[f"normal_{tag}__inc"]
    p = H.ProposalSpec(O.PROPOSAL_NORMAL, np.array([z[f"normal_{tag}__std"]]))
    got = E.propose(p.engine(device), D, raw.shape[0], ext_raw=dev_t(raw[:, None, :], device)).cpu().numpy()[:, 0]
    assert np.array_equal(got, inc)  # bit exact

    raw, inc = z[f"laplace_{tag}__raw"], z[f"laplace_{tag}__inc"]
    p = H.ProposalSpec(O.PROPOSAL_LAPLACE, np.ones(1, np.float32), z[f"laplace_{tag}__scale"])
    got = E.propose(p.engine(device), D, raw.shape[0], ext_raw=dev_t(raw[:, None, :], device)).cpu().numpy()[:, 0]
    np.testing.assert_allclose(got, inc, rtol=2e-6, atol=1e-8)  # v_log_f32 in place of log1p

    raw, inc = z[f"uniform_{tag}__raw"], z[f"uniform_{tag}__inc"]
    p = H.ProposalSpec(O.PROPOSAL_UNIFORM_RADIUS, np.array([z[f"uniform_{tag}__radius"]]), None, 1.0 / D)
    got = E.propose(p.engine(device), D, raw.shape[0], ext_raw=dev_t(raw[:, None, :], device)).cpu().numpy()[:, 0]
    np.testing.assert_allclose(got, inc, rtol=5e-6, atol=1e-8)


@pytest.mark.parametrize("kind,dim", [("Normal", 30), ("Normal", 7), ("Laplace", 30), ("Laplace", 13),
                                      ("UniformRadius", 50), ("UniformRadius", 5)])
def test_philox_proposals_match_oracle_and_moments(device, kind, dim):
    """In-kernel Philox draws: same words, same transforms as the oracle restatement; and the moments the
    reference's tests/test_proposals.py:160-210 check (variance = base/beta, radius <= R)."""
    betas = [1.0, 0.3, 0.05]
    kw = dict(base_variance_scalar=0.4, base_variance_vector=np.linspace(0.1, 0.9, dim), base_radius=1.3)
    kw = {k: v for k, v in kw.items() if (kind == "Normal" and k == "base_variance_scalar")
          or (kind == "Laplace" and k == "base_variance_vector") or (kind == "UniformRadius" and k == "base_radius")}
    p = H.proposal_spec(kind, dim, betas, **kw)
    n = 20000
    got = E.propose(p.engine(device), dim, n, seed=99).cpu().numpy()
    want = O.propose(p.oracle(), dim, 512, seed=99, precision="f64")
    # hardware sin/cos/log vs libm in fp64: absolute 2e-6 of the per-temperature scale
    scale = np.abs(want).max(axis=(0, 2), keepdims=True)
    assert np.max(np.abs(got[:512] - want) / scale) < 5e-6
    for t, b in enumerate(betas):
        if kind == "Normal":
            assert got[:, t].var() == pytest.approx(0.4 / b, rel=0.02)
        elif kind == "Laplace":
            np.testing.assert_allclose(got[:, t].var(axis=0), np.linspace(0.1, 0.9, dim) / b, rtol=0.08)
        else:
            r = np.linalg.norm(got[:, t], axis=1)
            R = 1.3 / np.sqrt(b)
            assert r.max() <= R * (1 + 1e-5)
            assert np.mean((r.astype(np.float64) / R) ** dim) == pytest.approx(0.5, abs=0.02)  # (r/R)^d is U(0,1)
        assert abs(got[:, t].mean()) < 4 * np.sqrt(got[:, t].var() / (n * dim)) + 1e-3


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", RWM_CASES)
def test_rwm_trajectory_matches_reference(device, name):
    """The reference's stored chain, its random tensors fed to the kernel.  Every golden step is replayed as an
    independent one-step problem (state_i, randoms_i) -> state_{i+1}: decisions agree >= 99.9 %; where the
    decision agrees the new state is the reference's."""
    z, spec, prop, beta = rwm_case(name)
    chain, lchain = z["chain"], z["logp_chain"]
    total = chain.shape[0] - 1
    res = gpu_run(spec, prop, device, state=chain[:-1][:, None, :], logp=lchain[:-1][:, None], beta=[beta], step0=0,
                  n_steps=1, ext_prop=z["raw"][None, :, None, :], ext_u=z["u"][None, :, None], want_flags=True)
    moved = np.any(chain[1:] != chain[:-1], axis=1)
    agree = res["accept_flags"][0, :, 0].astype(bool) == moved
    assert agree.mean() >= 0.999, f"{name}: {np.sum(~agree)} of {total} decisions differ"
    exact = str(z["proposal_kind"]) == "Normal"
    got, want = res["state"][agree, 0], chain[1:][agree]
    if exact:
        assert np.array_equal(got, want)
    else:
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6)
    logp_close(res["logp"][agree, 0], lchain[1:][agree])

    # free run from the initial state over the whole golden horizon: the oracle reproduces the reference's chain
    # exactly (tests/test_oracle_golden.py); the kernel must follow it, any decision flip proven (check_parity)
    lp0 = E.logdensity(spec.engine(device), dev_t(chain[:1], device)).cpu().numpy().reshape(1, 1)
    burn = int(z["burn_in"])
    kw = dict(state=chain[:1][None], logp=lp0, beta=[beta], n_steps=total, burn_in=burn, swap_every=1,
              ext_prop=np.ascontiguousarray(z["raw"][:, None, None, :]), ext_u=np.ascontiguousarray(z["u"][:, None, None]))
    flips = H.check_parity(gpu_runner(spec, prop, device), H.oracle_runner(spec, prop), spec, prop, exact_states=exact, **kw)
    free = gpu_run(spec, prop, device, step0=0, trace_temps=1, want_flags=True, **kw)
    if not flips:
        N = int(z["n_samples"])
        if exact:
            assert np.array_equal(free["trace"][:, 0, 0], chain[1:])
        else:
            np.testing.assert_allclose(free["trace"][:, 0, 0], chain[1:], rtol=1e-4, atol=1e-5)
        assert int(free["n_accept"][0, 0]) == int(z["num_acceptances"])
        assert free["sq_jump"][0, 0] / N == pytest.approx(float(z["esjd"]), rel=1e-4)
    else:
        first = flips[0][0]
        assert np.allclose(free["trace"][:first, 0, 0], chain[1:first + 1], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", PT_CASES)
def test_pt_trajectory_matches_reference(device, name):
    """Reference PT run (sequential sweep, Q1 row copy) replayed on the GPU from the reference's random tensors:
    teacher-forced per swap period, then free-running."""
    z, spec, prop, beta = pt_case(name)
    chains, lchains = z["chains"], z["logp_chains"]  # [T, total+1, D], [T, total+1]
    T, rows, D = chains.shape
    total, burn, se = rows - 1, int(z["burn_in"]), int(z["swap_every"])
    kw = dict(beta=beta, burn_in=burn, swap_every=se, swap_mode=E.SWAP_REFERENCE_COPY, swap_order=E.ORDER_SEQUENTIAL)

    run_kw = dict(state=np.ascontiguousarray(chains[:, 0])[None], logp=lchains[:, 0][None], n_steps=total,
                  ext_prop=np.ascontiguousarray(z["ext_prop"][:, None]), ext_u=np.ascontiguousarray(z["ext_u"][:, None]),
                  ext_swap_u=np.ascontiguousarray(z["ext_swap_u"][:, None]), **kw)
    # the whole golden horizon against the oracle (which reproduces the reference exactly), flips proven
    flips = H.check_parity(gpu_runner(spec, prop, device), H.oracle_runner(spec, prop), spec, prop, exact_states=True,
                           **run_kw)
    free = gpu_run(spec, prop, device, step0=0, trace_temps=T, **run_kw)
    got = free["trace"][:, 0]  # [total, T, D]
    want = chains[:, 1:].transpose(1, 0, 2)
    if not flips:
        assert np.array_equal(got, want)
        assert int(free["swap_accept"].sum()) == int(z["num_swap_acceptances"])
        last = int(free["last_swap_ordinal"].max())
        assert free["swap_accept"].sum() / last == pytest.approx(float(z["swap_acceptance_rate"]), rel=1e-12)
        b = z["beta_ladder"]
        sq = float((free["swap_accept"][0, :-1] * (b[:-1] - b[1:]) ** 2).sum())
        assert sq / last == pytest.approx(float(z["pt_esjd"]), rel=1e-9)
        assert free["sq_jump"][0, 0] / (total - burn) == pytest.approx(float(z["esjd"]), rel=1e-4)
        logp_close(free["trace_logp"][:, 0], lchains[:, 1:].T)
    else:
        first = flips[0][0]
        assert np.array_equal(got[:first], want[:first])

    # teacher forced: restart from the reference state at every step i (as replica i), one step each, the
    # step's place in the swap schedule kept through step0 -- done per residue class so step0 is shared
    n_bad = n_all = 0
    ev_of_step = {}
    e = 0
    for i in range(total):
        if (i + 1) % se == 0 and (i + 1) > burn:
            ev_of_step[i] = e
            e += 1
    for i in list(range(0, total, max(1, total // 40))) + sorted(ev_of_step)[:25]:
        us = None
        if i in ev_of_step:
            us = z["ext_swap_u"][ev_of_step[i]][None, None]
        one = gpu_run(spec, prop, device, state=np.ascontiguousarray(chains[:, i])[None], logp=lchains[:, i][None],
                      step0=i, n_steps=1, ext_prop=z["ext_prop"][i][None, None], ext_u=z["ext_u"][i][None, None],
                      ext_swap_u=us, **kw)
        same = np.all(one["state"][0] == chains[:, i + 1], axis=1)
        n_bad += int(np.sum(~same))
        n_all += T
    assert n_bad / n_all <= 1e-3, f"{name}: {n_bad}/{n_all} teacher-forced replica-steps differ"


# ---------------------------------------------------------------------------------------------------------
SWEEP = [
    ("rc15_d30", "Normal", 8, 7, dict(base_variance_scalar=2.38**2 / 30)),
    ("even_d30", "Laplace", 32, 4, dict(base_variance_vector=np.full(30, 0.02))),
    ("tm_d50", "UniformRadius", 64, 2, dict(base_radius=2.5)),
    ("hyb_3_5", "Normal", 5, 13, dict(base_variance_scalar=0.03)),      # 5 does not divide 64: idle lanes
    ("gamma_d5", "Laplace", 3, 30, dict(base_variance_vector=np.full(5, 1.0))),
    ("beta_d5", "UniformRadius", 1, 70, dict(base_radius=0.3)),          # T = 1: plain RWM, two waves
    ("full_d10", "Normal", 16, 9, dict(base_variance_scalar=0.02)),
    ("rc15s_d10", "Normal", 11, 6, dict(base_variance_scalar=0.5)),
    ("tms_d10", "Laplace", 7, 10, dict(base_variance_vector=np.full(10, 0.5))),
]


def start_state(spec, Cn, T, rng):
    if spec.cls == "IIDBetaTorch":
        x0 = rng.uniform(0.2, 0.8, spec.dim)
    elif spec.cls == "IIDGammaTorch":
        x0 = 5 + 0.01 * rng.standard_normal(spec.dim)
    elif "Rosenbrock" in spec.cls:
        x0 = 1e-8 * rng.standard_normal(spec.dim)
    else:
        x0 = np.zeros(spec.dim)
    st = np.broadcast_to(x0.astype(np.float32), (Cn, T, spec.dim)).copy()
    lp = np.broadcast_to(O.logdensity(spec.oracle(), x0[None].astype(np.float32)).astype(np.float32), (Cn, T)).copy()
    return st, lp


@pytest.mark.parametrize("tkey,pkind,T,Cn,pkw", SWEEP, ids=[f"{s[0]}-{s[1]}-T{s[2]}" for s in SWEEP])
@pytest.mark.parametrize("mode", ["exchange", "reference_copy"])
@pytest.mark.parametrize("order", ["sequential", "even_odd"])
def test_external_randoms_vs_oracle(device, tkey, pkind, T, Cn, pkw, mode, order):
    """Every (target, proposal) family, every swap semantics: kernel and oracle consume identical random arrays.
    Counters must match exactly whenever the decisions match; decisions may differ only by fp32-level flips."""
    spec = H.target_spec(tkey)
    rng = np.random.default_rng(zlib.crc32(f"{tkey}-{pkind}-{T}".encode()))
    beta = (0.05 ** (np.arange(T) / max(1, T - 1))).astype(np.float32)
    prop = H.proposal_spec(pkind, spec.dim, beta, **pkw)
    N, burn, se = 120, 17, 6
    raw = O.ext_raw_per_step(prop.kind, spec.dim)
    ext = rng.standard_normal((N, Cn, T, raw)).astype(np.float32)
    if pkind == "Laplace":
        ext = rng.random((N, Cn, T, raw)).astype(np.float32)
    elif pkind == "UniformRadius":
        ext[..., -1] = rng.random((N, Cn, T)).astype(np.float32)
    u = rng.random((N, Cn, T)).astype(np.float32)
    n_ev = N // se - burn // se
    us = rng.random((n_ev, Cn, max(T - 1, 1))).astype(np.float32)[:, :, :T - 1]
    st, lp = start_state(spec, Cn, T, rng)
    kw = dict(state=st, logp=lp, beta=beta, n_steps=N, burn_in=burn, swap_every=se, ext_prop=ext, ext_u=u,
              ext_swap_u=us if T > 1 else None)
    H.check_parity(gpu_runner(spec, prop, device), H.oracle_runner(spec, prop), spec, prop, exact_states=pkind == "Normal",
                   swap_mode=E.SWAP_MODES[mode], swap_order=E.SWAP_ORDERS[order], **kw)
    # carried log-densities are those of the carried states (checked at the kernel's own states, so that the
    # <= 2 ulp increment differences of Laplace / UniformRadius cannot leak into this tolerance)
    got = gpu_run(spec, prop, device, trace_temps=T, step0=0, swap_mode=E.SWAP_MODES[mode],
                  swap_order=E.SWAP_ORDERS[order], **kw)
    own = O.logdensity(spec.oracle(), got["trace"].reshape(-1, spec.dim), "f64").reshape(got["trace_logp"].shape)
    logp_close(got["trace_logp"], own, extra_abs=3e-4)


@pytest.mark.parametrize("tkey,pkind,T,Cn,pkw", SWEEP, ids=[f"{s[0]}-{s[1]}-T{s[2]}" for s in SWEEP])
@pytest.mark.parametrize("mode,order", [("exchange", "sequential"), ("reference_copy", "even_odd")])
def test_lane_split_kernel_vs_oracle(device, tkey, pkind, T, Cn, pkw, mode, order):
    """The lane-split (quad) form of the fused kernel (csrc/quad.h: four lanes per replica, chosen by the C ABI for
    under-filled launches and large dims) directly against the oracle over the full horizon, every family, with the
    form pinned - so its correctness does not rest on its bit-identity with the one-thread-per-replica kernel alone."""
    spec = H.target_spec(tkey)
    if not E.has_quad_variant(spec.kind, H.PROPOSAL_KIND[pkind], spec.dim, T):
        pytest.skip("no lane-split variant for this shape")
    rng = np.random.default_rng(zlib.crc32(f"quad-{tkey}-{pkind}-{T}".encode()))
    beta = (0.05 ** (np.arange(T) / max(1, T - 1))).astype(np.float32)
    prop = H.proposal_spec(pkind, spec.dim, beta, **pkw)
    N, burn, se = 60, 7, 4
    raw = O.ext_raw_per_step(prop.kind, spec.dim)
    ext = _ext_arrays(rng, pkind, N, Cn, T, raw)
    us = rng.random((N // se - burn // se, Cn, max(T - 1, 1))).astype(np.float32)[:, :, :T - 1]
    st, lp = start_state(spec, Cn, T, rng)
    with E.kernel_form(E.FORM_QUAD):
        H.check_parity(gpu_runner(spec, prop, device), H.oracle_runner(spec, prop), spec, prop, exact_states=pkind == "Normal",
                       state=st, logp=lp, beta=beta, n_steps=N, burn_in=burn, swap_every=se, ext_prop=ext,
                       ext_u=rng.random((N, Cn, T)).astype(np.float32), ext_swap_u=us if T > 1 else None,
                       swap_mode=E.SWAP_MODES[mode], swap_order=E.SWAP_ORDERS[order])


def test_form_selection_never_changes_a_result(device):
    """The C ABI picks the kernel form from the batch size (csrc/capi.hip, AUTO): a whole batch large enough for the
    one-thread-per-replica kernel and its two halves small enough for the lane-split kernel must give the SAME bits
    (the multi-GPU sharding rule would otherwise depend on the shard size), and so must the pinned forms."""
    spec = H.target_spec("rc15_d30")
    T, Cn, N = 1, 160000, 40  # 2 500 waves (2.4 per SIMD) -> thread form; halves: 1 250 waves -> lane-split form
    beta = np.ones(1, np.float32)
    prop = H.proposal_spec("Normal", 30, [1.0], base_variance_scalar=2.38**2 / 30, single=True)
    st, lp = start_state(spec, Cn, T, np.random.default_rng(3))
    kw = dict(beta=beta, step0=0, n_steps=N, burn_in=5, swap_every=1, seed=99)
    whole = gpu_run(spec, prop, device, state=st, logp=lp, chain_offset=1000, **kw)
    lo = gpu_run(spec, prop, device, state=st[:Cn // 2], logp=lp[:Cn // 2], chain_offset=1000, **kw)
    hi = gpu_run(spec, prop, device, state=st[Cn // 2:], logp=lp[Cn // 2:], chain_offset=1000 + Cn // 2, **kw)
    for k in ("state", "logp", "n_accept", "sq_jump"):
        assert np.array_equal(whole[k], np.concatenate([lo[k], hi[k]])), k
    for form in (E.FORM_THREAD, E.FORM_QUAD):
        with E.kernel_form(form):
            pinned = gpu_run(spec, prop, device, state=st[:4096], logp=lp[:4096], chain_offset=1000, **kw)
        for k in ("state", "logp", "n_accept", "sq_jump"):
            assert np.array_equal(pinned[k], whole[k][:4096]), (k, form)
    assert E.set_kernel_form(E.FORM_AUTO) == E.FORM_AUTO  # the context managers restored it
    with pytest.raises(E.PTRWMError):
        E.set_kernel_form(7)


def test_lane_split_kernel_is_bit_identical_to_the_thread_kernel(device):
    """tools/check_quad.py: every target kernel x proposal x lane width (generic and dim-compiled-in) on a fixed grid plus
    60 random configurations; production and fixture variants, Philox and external randoms, narrow and wide ladders:
    states, log-densities, all statistics (fp64 bits), traces and accept flags identical between the two forms."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_quad.py"), "60", "5"], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1000:])
    assert "kernels bit-identical" in r.stdout and "MISMATCH" not in r.stdout


@pytest.mark.parametrize("tkey,pkind,T,Cn,pkw", SWEEP, ids=[f"{s[0]}-{s[1]}-T{s[2]}" for s in SWEEP])
@pytest.mark.parametrize("form", ["thread", "quad"])
def test_philox_mode_vs_oracle(device, tkey, pkind, T, Cn, pkw, form):
    """The PRODUCTION path - in-kernel Philox, hardware Box-Muller with the scale folded into the radius - against the
    oracle's restatement of the same counter layout (same seed, same chain ids), both forms of the kernel, every family:
    the FULL horizon, every differing Metropolis / swap decision proven in fp64 from the numbers the oracle drew
    (helpers.check_parity_philox), bookkeeping identical on agreeing segments.  No agreement-rate threshold: a Philox
    word taken from the wrong place fails (tests/test_parity_checker.py::test_a_slipped_philox_word_fails)."""
    spec = H.target_spec(tkey)
    kform = {"thread": E.FORM_THREAD, "quad": E.FORM_QUAD}[form]
    if form == "quad" and not E.has_quad_variant(spec.kind, H.PROPOSAL_KIND[pkind], spec.dim, T):
        pytest.skip("no lane-split variant for this shape")
    beta = (0.05 ** (np.arange(T) / max(1, T - 1))).astype(np.float32)
    prop = H.proposal_spec(pkind, spec.dim, beta, **pkw)
    st, lp = start_state(spec, Cn, T, np.random.default_rng(11))
    with E.kernel_form(kform):
        for order, mode in (("sequential", "exchange"), ("even_odd", "reference_copy")):
            H.check_parity_philox(gpu_runner(spec, prop, device), spec, prop, state=st, logp=lp, beta=beta, step0=3,
                                  n_steps=90, burn_in=9, swap_every=4, seed=0xC0FFEE1234, chain_offset=(1 << 32) + 1000003,
                                  swap_mode=E.SWAP_MODES[mode], swap_order=E.SWAP_ORDERS[order], segment=PHILOX_SEGMENT)


def test_launch_split_and_resume_are_invisible(device):
    """n steps in one call == the same steps in several calls (step0 carries the schedule and the RNG position)."""
    spec = H.target_spec("rc15_d30")
    T, Cn = 8, 9
    beta = (0.01 ** (np.arange(T) / (T - 1))).astype(np.float32)
    prop = H.proposal_spec("Normal", 30, beta, base_variance_scalar=2.38**2 / 30)
    st, lp = start_state(spec, Cn, T, np.random.default_rng(0))
    kw = dict(beta=beta, burn_in=13, swap_every=5, seed=77, chain_offset=5)
    one = gpu_run(spec, prop, device, state=st, logp=lp, step0=0, n_steps=90, **kw)
    a = gpu_run(spec, prop, device, state=st, logp=lp, step0=0, n_steps=31, **kw)
    b = gpu_run(spec, prop, device, state=a["state"], logp=a["logp"], step0=31, n_steps=59, **kw)
    assert np.array_equal(one["state"], b["state"]) and np.array_equal(one["logp"], b["logp"])
    for k in ("n_accept", "swap_accept"):
        assert np.array_equal(one[k], a[k] + b[k])
    assert np.array_equal(one["last_swap_ordinal"], np.maximum(a["last_swap_ordinal"], b["last_swap_ordinal"]))
    # a request longer than one launch may be (capi.hip: 2^16 steps) is cut by the library itself: the same again
    spec2 = H.target_spec("rc15_d30")
    prop2 = H.proposal_spec("Normal", 30, beta[:2], base_variance_scalar=2.38**2 / 30)
    st2, lp2 = start_state(spec2, 2, 2, np.random.default_rng(1))
    kw2 = dict(beta=beta[:2], burn_in=100, swap_every=7, seed=5, chain_offset=9)
    long_run = gpu_run(spec2, prop2, device, state=st2, logp=lp2, step0=0, n_steps=70000, **kw2)
    a2 = gpu_run(spec2, prop2, device, state=st2, logp=lp2, step0=0, n_steps=65536, **kw2)
    b2 = gpu_run(spec2, prop2, device, state=a2["state"], logp=a2["logp"], step0=65536, n_steps=70000 - 65536, **kw2)
    assert np.array_equal(long_run["state"], b2["state"]) and np.array_equal(long_run["logp"], b2["logp"])
    for k in ("n_accept", "swap_accept"):
        assert np.array_equal(long_run[k], a2[k] + b2[k])
    assert np.allclose(long_run["sq_jump"], a2["sq_jump"] + b2["sq_jump"], rtol=1e-12)


# (target, proposal, temperatures, ladders, proposal arguments, does ptrwm_run's streaming form serve this shape?)
STREAM_CASES = [
    ("rc15_d30", "Normal", 32, 2 * 37, dict(base_variance_scalar=2.38**2 / 30), True),   # BASELINE configs[2]'s shape
    ("even_d30", "Laplace", 32, 2 * 20, dict(base_variance_vector=np.full(30, 0.02)), True),
    ("tm_d50", "UniformRadius", 64, 23, dict(base_radius=2.5), True),                    # one ladder per wavefront
    ("tm15_d30", "UniformRadius", 8, 8 * 5, dict(base_radius=2.5), True),
    ("rc15_d30", "Normal", 1, 64 * 5, dict(base_variance_scalar=2.38**2 / 30), True),    # plain RWM: 64 chains per wavefront
    ("rc15_d30", "Normal", 5, 12 * 7, dict(base_variance_scalar=2.38**2 / 30), True),    # 5 does not divide 64: idle lanes
    ("full_d10", "Normal", 16, 4 * 9, dict(base_variance_scalar=0.02), True),
    ("beta_d5", "UniformRadius", 4, 16 * 6, dict(base_radius=0.3), True),
    ("gamma_d5", "Laplace", 3, 21 * 4, dict(base_variance_vector=np.full(5, 1.0)), False),  # 63 live lanes: no whole 16-byte pairs
    ("rc15_d30", "Normal", 32, 2 * 37 + 1, dict(base_variance_scalar=2.38**2 / 30), False),  # a ragged last group
    ("hyb_3_5", "Normal", 4, 16 * 3, dict(base_variance_scalar=0.03), False),             # dim 11: no kernel with dim compiled in
]


@pytest.mark.parametrize("tkey,pkind,T,Cn,pkw,streams", STREAM_CASES,
                         ids=[f"{c[0]}-{c[1]}-T{c[2]}-C{c[3]}" for c in STREAM_CASES])
def test_streaming_form_never_changes_a_result(device, tkey, pkind, T, Cn, pkw, streams):
    """Short launches of large batches run a streaming form of the step kernel (kernel.h STREAM: persistent wavefronts,
    the next group's state arriving by LDS-DMA while the current one is stepped; capi.hip picks it by launch length and
    batch size).  Same Philox words, same arithmetic, same canonical order: launches of 1, 2 and 3 steps chained over two
    and a half swap periods give, launch by launch, the bits of the classic kernel - states, log-densities, acceptance
    and swap counts, last-swap ordinals and the fp64 squared-jump sums - and those of ONE classic launch over the whole
    horizon.  Shapes the streaming form does not serve (ragged last group, no whole 16-byte vectors per group, no kernel
    with dim compiled in) silently take the classic kernel under STREAM_ON."""
    spec = H.target_spec(tkey)
    beta = (0.05 ** (np.arange(T) / max(1, T - 1))).astype(np.float32)
    prop = H.proposal_spec(pkind, spec.dim, beta, **pkw)
    st0, lp0 = start_state(spec, Cn, T, np.random.default_rng(T))
    kw = dict(beta=beta, burn_in=4, swap_every=5, seed=91, chain_offset=3)
    keys = ("state", "logp", "n_accept", "sq_jump", "swap_accept", "last_swap_ordinal")
    assert bool(E.has_stream_variant(spec.kind, prop.kind, spec.dim)) == (tkey != "hyb_3_5")
    with E.kernel_form(E.FORM_THREAD):
        for n in (1, 2, 3):
            launches = 26 // n
            st_c, lp_c, st_s, lp_s = st0, lp0, st0, lp0
            tot = {k: 0 for k in keys[2:]}
            for i in range(launches):
                with E.stream_mode(E.STREAM_OFF):
                    c = gpu_run(spec, prop, device, state=st_c, logp=lp_c, step0=i * n, n_steps=n, **kw)
                    assert E.last_launch_kind() == E.LAUNCH_THREAD
                with E.stream_mode(E.STREAM_ON):
                    s = gpu_run(spec, prop, device, state=st_s, logp=lp_s, step0=i * n, n_steps=n, **kw)
                    assert E.last_launch_kind() == (E.LAUNCH_STREAM if streams else E.LAUNCH_THREAD)
                for k in keys:
                    assert np.array_equal(c[k], s[k]), (k, n, i)
                st_c, lp_c, st_s, lp_s = c["state"], c["logp"], s["state"], s["logp"]
                for k in ("n_accept", "swap_accept"):
                    tot[k] = tot[k] + s[k]
                tot["last_swap_ordinal"] = np.maximum(tot["last_swap_ordinal"], s["last_swap_ordinal"])
            with E.stream_mode(E.STREAM_OFF):
                one = gpu_run(spec, prop, device, state=st0, logp=lp0, step0=0, n_steps=launches * n, **kw)
            assert np.array_equal(one["state"], st_s) and np.array_equal(one["logp"], lp_s)
            for k in ("n_accept", "swap_accept", "last_swap_ordinal"):
                assert np.array_equal(one[k], tot[k]), (k, n)
            assert one["n_accept"].sum() > 0 and (T == 1 or one["swap_accept"].sum() > 0)


def test_streaming_form_accumulates_statistics_in_place(device):
    """The streaming form reads the squared-jump sums of the next group ahead of time and writes old + delta back; the
    integer statistics are added by atomics: non-zero starting values are carried exactly as the classic kernel's
    read-modify-write carries them (the drop-in classes accumulate over many launches)."""
    spec = H.target_spec("rc15_d30")
    T, Cn = 32, 2 * 40
    beta = (0.01 ** (np.arange(T) / (T - 1))).astype(np.float32)
    prop = H.proposal_spec("Normal", 30, beta, base_variance_scalar=2.38**2 / 30)
    st0, lp0 = start_state(spec, Cn, T, np.random.default_rng(3))
    out = {}
    for mode in (E.STREAM_OFF, E.STREAM_ON):
        st, lp = dev_t(st0, device), dev_t(lp0, device)
        stats = dict(n_accept=torch.full((Cn, T), 7, dtype=torch.int64, device=device),
                     sq_jump=torch.full((Cn, T), 0.1, dtype=torch.float64, device=device),
                     swap_accept=torch.full((Cn, T), 5, dtype=torch.int64, device=device),
                     last_swap_ordinal=torch.full((Cn, T), 40, dtype=torch.int64, device=device))
        plan = E.RunPlan(spec.engine(device), prop.engine(device), state=st, logp=lp, beta=dev_t(beta, device), burn_in=0,
                         swap_every=3, seed=12, **stats)
        with E.kernel_form(E.FORM_THREAD), E.stream_mode(mode):
            for i in range(12):
                plan.launch(i, 1)
        torch.cuda.synchronize()
        out[mode] = {k: v.cpu().numpy() for k, v in stats.items()}
        out[mode]["state"] = st.cpu().numpy()
    for k in out[E.STREAM_OFF]:
        assert np.array_equal(out[E.STREAM_OFF][k], out[E.STREAM_ON][k]), k
    assert (out[E.STREAM_ON]["sq_jump"] > 0.1).any() and (out[E.STREAM_ON]["last_swap_ordinal"] > 40).any()


def test_chain_offset_makes_sharding_invisible(device):
    """Chains [0, 2n) in one call == chains [0, n) and [n, 2n) in two calls with chain_offset (multi-GPU rule)."""
    spec = H.target_spec("even_d30")
    T, Cn = 4, 40
    beta = np.array([1, 0.6, 0.3, 0.1], np.float32)
    prop = H.proposal_spec("Laplace", 30, beta, base_variance_vector=np.full(30, 0.02))
    st, lp = start_state(spec, Cn, T, np.random.default_rng(1))
    kw = dict(beta=beta, step0=0, n_steps=40, burn_in=0, swap_every=3, seed=5)
    whole = gpu_run(spec, prop, device, state=st, logp=lp, chain_offset=100, **kw)
    lo = gpu_run(spec, prop, device, state=st[:17], logp=lp[:17], chain_offset=100, **kw)
    hi = gpu_run(spec, prop, device, state=st[17:], logp=lp[17:], chain_offset=117, **kw)
    assert np.array_equal(whole["state"], np.concatenate([lo["state"], hi["state"]]))
    assert np.array_equal(whole["n_accept"], np.concatenate([lo["n_accept"], hi["n_accept"]]))


@pytest.mark.parametrize("form", ["thread", "quad"])
@pytest.mark.parametrize("dim,T,Cn", [(7, 3, 100), (30, 32, 9), (5, 1, 333), (13, 70, 3), (81, 4, 21)])
def test_state_alignment_is_invisible(device, dim, T, Cn, form):
    """The kernels stage their group's run of the state through LDS in 16-byte vectors whatever the run's alignment
    (kernel.h stage_copy: the vectors of the aligned frame, the ragged ends element by element): a state array that
    starts 4, 8 or 12 bytes past a 16-byte boundary, with group runs of every alignment inside it, gives the bits of the
    aligned one, and nothing outside the array is touched."""
    params = {"modes": np.float32([-4, 0, 4]), "weights": np.float32([0.2, 0.5, 0.3])}
    spec = H.spec_from_params("RoughCarpetDistributionTorch", dim, params)
    if form == "thread" and not E.has_thread_variant(spec.kind, 0, dim):
        pytest.skip("one form only above dim 64")
    beta = (0.1 ** (np.arange(T) / max(1, T - 1))).astype(np.float32)
    prop = H.proposal_spec("Normal", dim, beta, base_variance_scalar=2.38**2 / dim)
    rng = np.random.default_rng(dim * T)
    st = rng.normal(0, 3, (Cn, T, dim)).astype(np.float32)
    lp = O.logdensity(spec.oracle(), st.reshape(-1, dim)).astype(np.float32).reshape(Cn, T)
    kw = dict(beta=dev_t(beta, device), step0=0, n_steps=25, burn_in=2, swap_every=3, seed=5, chain_offset=11)
    n = st.size
    outs = []
    with E.kernel_form({"thread": E.FORM_THREAD, "quad": E.FORM_QUAD}[form]):
        for shift in (0, 1, 2, 3):
            buf = torch.full((n + 8,), 7.25, device=device)
            view = buf[shift:shift + n].view(Cn, T, dim)
            view.copy_(dev_t(st, device))
            lpd = dev_t(lp, device)
            E.run(spec.engine(device), prop.engine(device), state=view, logp=lpd, **kw)
            torch.cuda.synchronize()
            assert bool((buf[:shift] == 7.25).all()) and bool((buf[shift + n:] == 7.25).all())
            outs.append((view.cpu().numpy().copy(), lpd.cpu().numpy()))
    for s_, l_ in outs[1:]:
        assert np.array_equal(s_, outs[0][0]) and np.array_equal(l_, outs[0][1])
    assert not np.array_equal(outs[0][0], st)


def test_argument_validation_through_the_abi(device):
    spec = H.target_spec("rc15_d30")
    prop = H.proposal_spec("Normal", 30, [1.0], base_variance_scalar=0.1)
    st = torch.zeros(2, 1, 30, device=device)
    lp = torch.zeros(2, 1, device=device)
    b = torch.ones(1, device=device)
    with pytest.raises(E.PTRWMError) as ei:
        E.run(spec.engine(device), prop.engine(device), state=st, logp=lp, beta=b, step0=0, n_steps=1, swap_every=0)
    assert ei.value.code == -5
    with pytest.raises(E.PTRWMError):
        E.run(spec.engine(device), prop.engine(device), state=st, logp=lp, beta=b, step0=-1, n_steps=1)
    # zero work is a no-op, not an error
    E.run(spec.engine(device), prop.engine(device), state=st, logp=lp, beta=b, step0=0, n_steps=0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        E.logdensity(spec.engine(device), torch.zeros(3, 30))


# ---------------------------------------------------------------------------------------------------------
def _ext_arrays(rng, pkind, N, Cn, T, raw):
    ext = rng.standard_normal((N, Cn, T, raw)).astype(np.float32)
    if pkind == "Laplace":
        ext = rng.random((N, Cn, T, raw)).astype(np.float32)
    elif pkind == "UniformRadius":
        ext[..., -1] = rng.random((N, Cn, T)).astype(np.float32)
    return ext


ALL_DIMS = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 15, 16, 17, 20, 23, 24, 25, 29, 30, 31, 32, 33, 40, 41, 48, 49, 50, 51, 56, 57,
            63, 64, 65, 80, 81, 99, 100, 101, 104]


@pytest.mark.parametrize("dim", ALL_DIMS)
def test_every_register_width_vs_oracle(device, dim):
    """Exact-width kernels (dims the reference's experiments use) and generic-width kernels (run-time dim with
    predicated tails) for every boundary case of the width table, three-term RoughCarpet (modes +-4)."""
    params = {"modes": np.float32([-4, 0, 4]), "weights": np.float32([0.2, 0.5, 0.3])}
    spec = H.spec_from_params("RoughCarpetDistributionTorch", dim, params)
    T, Cn, N = 3, 23, 24
    beta = np.float32([1.0, 0.4, 0.1])
    prop = H.proposal_spec("Normal", dim, beta, base_variance_scalar=2.38**2 / dim)
    rng = np.random.default_rng(dim)
    st = rng.normal(0, 3, (Cn, T, dim)).astype(np.float32)
    lp = O.logdensity(spec.oracle(), st.reshape(-1, dim)).astype(np.float32).reshape(Cn, T)
    ext = _ext_arrays(rng, "Normal", N, Cn, T, dim)
    u = rng.random((N, Cn, T)).astype(np.float32)
    us = rng.random((N // 4, Cn, T - 1)).astype(np.float32)
    kw = dict(state=st, logp=lp, beta=beta, n_steps=N, burn_in=3, swap_every=4, ext_prop=ext, ext_u=u,
              ext_swap_u=us)
    H.check_parity(gpu_runner(spec, prop, device), H.oracle_runner(spec, prop), spec, prop, exact_states=True, **kw)
    # Philox mode (the production path) at the same width, every proposal's word map: every decision that differs from
    # the oracle's restated stream proven, to the end of the horizon
    for pkind in ("Normal", "Laplace", "UniformRadius"):
        pkw = {"Normal": dict(base_variance_scalar=2.38**2 / dim), "Laplace": dict(base_variance_vector=np.full(dim, 2.38**2 / dim)),
               "UniformRadius": dict(base_radius=2.4)}[pkind]
        pp = H.proposal_spec(pkind, dim, beta, **pkw)
        H.check_parity_philox(gpu_runner(spec, pp, device), spec, pp, state=st, logp=lp, beta=beta, step0=5, n_steps=N,
                              burn_in=3, swap_every=4, seed=dim * 7919, chain_offset=3, segment=PHILOX_SEGMENT)
    g2 = gpu_run(spec, prop, device, state=st, logp=lp, beta=beta, step0=5, n_steps=N, burn_in=3, swap_every=4,
                 seed=dim * 7919, chain_offset=3)
    own = O.logdensity(spec.oracle(), g2["state"].reshape(-1, dim), "f64").reshape(Cn, T)
    logp_close(g2["logp"], own, extra_abs=3e-4)


GENERIC_TARGETS = [
    ("ThreeMixtureDistributionTorch", 12, {"means": np.float32(np.linspace(-3, 3, 36).reshape(3, 12)),
                                            "mixing_weights": np.float32([0.3, 0.3, 0.4])}),
    ("ThreeMixtureDistributionTorch", 12, {"means": np.float32(np.linspace(-3, 3, 36).reshape(3, 12)),
                                            "mixing_weights": np.float32([0.3, 0.3, 0.4]),
                                            "scaling_factors": np.float32(np.linspace(0.3, 1.7, 12))}),
    ("RoughCarpetDistributionTorch", 13, {"modes": np.float32([-15, 0, 15]), "weights": np.float32([0.5, 0.3, 0.2]),
                                           "scaling_factors": np.float32(np.linspace(0.5, 1.5, 13))}),
    ("FullRosenbrockTorch", 9, {"a_coeff": np.float32(0.05), "b_coeff": np.float32(5), "mu": np.float32(np.linspace(0.8, 1.2, 8))}),
    ("EvenRosenbrockTorch", 12, {"a_coeff": np.float32(0.05), "b_coeff": np.float32(5), "mu": np.float32(np.ones(6))}),
    ("HybridRosenbrockTorch", 13, {"a_coeff": np.float32(0.05), "b_coeff": np.float32(5), "mu": np.float32(1), "n1": 4, "n2": 4}),
    ("HybridRosenbrockTorch", 73, {"a_coeff": np.float32(0.05), "b_coeff": np.float32(5), "mu": np.float32(1), "n1": 7, "n2": 12}),
    ("IIDGammaTorch", 7, {"shape": np.float32(2.0), "scale": np.float32(3.0)}),
    ("IIDBetaTorch", 11, {"alpha": np.float32(2.0), "beta": np.float32(3.0)}),
]


@pytest.mark.parametrize("cls,dim,params", GENERIC_TARGETS, ids=[f"{c[0][:12]}-{c[1]}-{i}" for i, c in enumerate(GENERIC_TARGETS)])
@pytest.mark.parametrize("pkind", ["Normal", "Laplace", "UniformRadius"])
def test_generic_width_targets_vs_oracle(device, cls, dim, params, pkind):
    """Every target functor through a run-time-dim (generic width) kernel, with every proposal."""
    spec = H.spec_from_params(cls, dim, params)
    T, Cn, N = 4, 17, 30
    beta = np.float32([1.0, 0.5, 0.2, 0.05])
    scale = 0.02 if "Rosenbrock" in cls or "Beta" in cls else 0.5
    pkw = {"Normal": dict(base_variance_scalar=scale), "Laplace": dict(base_variance_vector=np.full(dim, scale)),
           "UniformRadius": dict(base_radius=float(np.sqrt(scale * (dim + 2))))}[pkind]
    prop = H.proposal_spec(pkind, dim, beta, **pkw)
    rng = np.random.default_rng(zlib.crc32(f"{cls}{dim}{pkind}".encode()))
    st, lp = start_state(spec, Cn, T, rng)
    raw = O.ext_raw_per_step(prop.kind, dim)
    kw = dict(state=st, logp=lp, beta=beta, n_steps=N, burn_in=4, swap_every=5,
              ext_prop=_ext_arrays(rng, pkind, N, Cn, T, raw), ext_u=rng.random((N, Cn, T)).astype(np.float32),
              ext_swap_u=rng.random((N // 5, Cn, T - 1)).astype(np.float32))
    H.check_parity(gpu_runner(spec, prop, device), H.oracle_runner(spec, prop), spec, prop, exact_states=pkind == "Normal",
                   **kw)
    got = gpu_run(spec, prop, device, trace_temps=T, step0=0, **kw)
    own = O.logdensity(spec.oracle(), got["trace"].reshape(-1, dim), "f64").reshape(got["trace_logp"].shape)
    logp_close(got["trace_logp"], own, extra_abs=3e-4)


# dim = 1 + n2 (n1 - 1): the dims of the reference's HybridRosenbrock data (data/HybridRosenbrock_*_dim{9,19,29}_*)
HYBRID_DATA_DIMS = [(9, 3, 4), (19, 4, 6), (29, 5, 7)]


@pytest.mark.parametrize("dim,n1,n2", HYBRID_DATA_DIMS)
@pytest.mark.parametrize("pkind", ["Normal", "Laplace", "UniformRadius"])
def test_hybrid_rosenbrock_data_dims_have_kernels_of_their_own(device, dim, n1, n2, pkind):
    """Dims 9, 19 and 29 are compiled in for HybridRosenbrock ALONE (variants.h PTRWM_WIDTHS_EXTRA: entries behind the
    common width table that only a lookup naming that target finds).  Visible through the ABI: the target has a streaming
    twin there (only kernels with dim compiled in do), other targets at the same dim do not; and the kernels are the same
    arithmetic - thread form and lane-split form against the oracle on shared randoms and on the production Philox path,
    the two forms and the streaming form against each other bit for bit."""
    params = {"a_coeff": np.float32(0.05), "b_coeff": np.float32(5), "mu": np.float32(1), "n1": n1, "n2": n2}
    spec = H.spec_from_params("HybridRosenbrockTorch", dim, params)
    T, Cn, N = 4, 16 * 3, 30
    beta = np.float32([1.0, 0.5, 0.2, 0.05])
    pkw = {"Normal": dict(base_variance_scalar=0.02), "Laplace": dict(base_variance_vector=np.full(dim, 0.02)),
           "UniformRadius": dict(base_radius=float(np.sqrt(0.02 * (dim + 2))))}[pkind]
    prop = H.proposal_spec(pkind, dim, beta, **pkw)
    assert E.has_stream_variant(spec.kind, prop.kind, dim) == 1
    other = H.spec_from_params("FullRosenbrockTorch", dim, {"a_coeff": np.float32(0.05), "b_coeff": np.float32(5),
                                                           "mu": np.float32(np.ones(dim - 1))})
    assert E.has_stream_variant(other.kind, prop.kind, dim) == 0 and E.has_variant(other.kind, prop.kind, dim) == 1
    rng = np.random.default_rng(zlib.crc32(f"hybrid{dim}{pkind}".encode()))
    st, lp = start_state(spec, Cn, T, rng)
    raw = O.ext_raw_per_step(prop.kind, dim)
    kw = dict(state=st, logp=lp, beta=beta, n_steps=N, burn_in=4, swap_every=5,
              ext_prop=_ext_arrays(rng, pkind, N, Cn, T, raw), ext_u=rng.random((N, Cn, T)).astype(np.float32),
              ext_swap_u=rng.random((N // 5, Cn, T - 1)).astype(np.float32))
    keys = ("state", "logp", "n_accept", "sq_jump", "swap_accept", "last_swap_ordinal")
    runs = {}
    for form in (E.FORM_THREAD, E.FORM_QUAD):
        with E.kernel_form(form):
            H.check_parity(gpu_runner(spec, prop, device), H.oracle_runner(spec, prop), spec, prop,
                           exact_states=pkind == "Normal", **kw)
            H.check_parity_philox(gpu_runner(spec, prop, device), spec, prop, state=st, logp=lp, beta=beta, step0=5, n_steps=N,
                                  burn_in=3, swap_every=4, seed=dim * 104729, chain_offset=3, segment=PHILOX_SEGMENT)
            runs[form] = gpu_run(spec, prop, device, state=st, logp=lp, beta=beta, step0=5, n_steps=N, burn_in=3, swap_every=4,
                                 seed=dim * 104729, chain_offset=3)
            assert E.last_launch_kind() == (E.LAUNCH_THREAD if form == E.FORM_THREAD else E.LAUNCH_QUAD)
    for k in keys:
        assert np.array_equal(runs[E.FORM_THREAD][k], runs[E.FORM_QUAD][k]), k
    # one-step launches through the streaming twin: the same bits as the classic kernel's
    with E.kernel_form(E.FORM_THREAD):
        a = b = dict(state=st, logp=lp)
        for i in range(7):
            with E.stream_mode(E.STREAM_OFF):
                a = gpu_run(spec, prop, device, state=a["state"], logp=a["logp"], beta=beta, step0=i, n_steps=1, burn_in=2,
                            swap_every=3, seed=11, chain_offset=0)
            with E.stream_mode(E.STREAM_ON):
                b = gpu_run(spec, prop, device, state=b["state"], logp=b["logp"], beta=beta, step0=i, n_steps=1, burn_in=2,
                            swap_every=3, seed=11, chain_offset=0)
                assert E.last_launch_kind() == E.LAUNCH_STREAM
            for k in keys:
                assert np.array_equal(a[k], b[k]), (k, i)


@pytest.mark.parametrize("T,Cn", [(64, 3), (33, 2), (63, 5), (21, 4), (2, 100), (1, 1), (32, 1),
                                  (65, 2), (100, 3), (128, 1), (200, 2), (256, 2)])  # > 64: one ladder per workgroup
def test_ladder_shapes_vs_oracle(device, T, Cn):
    """Ladders that fill a wavefront exactly, leave idle lanes (T not dividing 64), span several waves, or are
    wider than a wavefront (one ladder per 256-thread workgroup, exchange through LDS with workgroup barriers)."""
    spec = H.target_spec("rc15_d30")
    beta = (0.02 ** (np.arange(T) / max(1, T - 1))).astype(np.float32)
    prop = H.proposal_spec("Normal", 30, beta, base_variance_scalar=2.38**2 / 30)
    rng = np.random.default_rng(T * 1000 + Cn)
    st, lp = start_state(spec, Cn, T, rng)
    N = 40
    for order, mode in (("sequential", "exchange"), ("even_odd", "exchange"), ("sequential", "reference_copy"),
                        ("even_odd", "reference_copy")):
        kw = dict(state=st, logp=lp, beta=beta, n_steps=N, burn_in=2, swap_every=3,
                  ext_prop=rng.standard_normal((N, Cn, T, 30)).astype(np.float32),
                  ext_u=rng.random((N, Cn, T)).astype(np.float32),
                  ext_swap_u=rng.random((N // 3, Cn, T - 1)).astype(np.float32) if T > 1 else None,
                  swap_order=E.SWAP_ORDERS[order], swap_mode=E.SWAP_MODES[mode])
        H.check_parity(gpu_runner(spec, prop, device), H.oracle_runner(spec, prop), spec, prop, exact_states=True, **kw)


def test_zero_chains_and_support_edges(device):
    spec = H.target_spec("beta_d5")
    prop = H.proposal_spec("Normal", 5, [1.0], base_variance_scalar=0.5, single=True)
    # empty batch: nothing is launched, nothing is touched
    E.run(spec.engine(device), prop.engine(device), state=torch.zeros(0, 1, 5, device=device),
          logp=torch.zeros(0, 1, device=device), beta=torch.ones(1, device=device), step0=0, n_steps=10)
    # a chain started OUTSIDE the support (log p = -inf) accepts its first in-support proposal and never leaves
    st = np.full((64, 1, 5), 1.05, np.float32)
    lp = np.full((64, 1), -np.inf, np.float32)
    got = gpu_run(spec, prop, device, state=st, logp=lp, beta=[1.0], step0=0, n_steps=400, seed=3, want_flags=True)
    inside = np.all((got["state"] > 0) & (got["state"] < 1), axis=2)
    assert np.array_equal(np.isfinite(got["logp"]), inside)
    assert inside.mean() > 0.5
    # ... decision for decision as the oracle on the same Philox stream (a flip only on the edge of the support or inside
    # the fp32 band of the threshold, proven)
    H.check_parity_philox(gpu_runner(spec, prop, device), spec, prop, state=st, logp=lp, beta=np.float32([1.0]), n_steps=400,
                          burn_in=0, swap_every=1, seed=3, segment=PHILOX_SEGMENT)


def gpu_sweep(device, *, state, logp, beta, event_index, dim, **kw):
    """Mirror of oracle.swap_sweep for the engine (ptrwm_swap_sweep through a RunPlan)."""
    Cn, T, D = state.shape
    st, lp = dev_t(state, device), dev_t(logp, device).reshape(Cn, T).contiguous()
    sw = torch.zeros(Cn, T, dtype=torch.int64, device=device)
    lo = torch.zeros(Cn, T, dtype=torch.int64, device=device)
    spec = H.target_spec("rc15_d30") if dim == 30 else None
    tgt = spec.engine(device) if spec is not None else E.Target(E.TARGET_HYPERCUBE, dim, p=(0.0, 1.0))
    prop = E.Proposal(E.PROPOSAL_NORMAL, torch.ones(T, device=device))
    ext = kw.pop("ext_swap_u", None)
    plan = E.RunPlan(tgt, prop, state=st, logp=lp, beta=dev_t(beta, device), swap_accept=sw, last_swap_ordinal=lo,
                     swap_mode=kw.pop("swap_mode", E.SWAP_EXCHANGE), swap_order=kw.pop("swap_order", E.ORDER_SEQUENTIAL),
                     seed=kw.pop("seed", 0), chain_offset=kw.pop("chain_offset", 0))
    plan.swap_sweep(kw.pop("rng_step", 0), event_index, kw.pop("rng_stream", 2),
                    ext_swap_u=None if ext is None else dev_t(ext, device))
    assert not kw
    torch.cuda.synchronize()
    return {"state": st.cpu().numpy(), "logp": lp.cpu().numpy(), "swap_accept": sw.cpu().numpy(),
            "last_swap_ordinal": lo.cpu().numpy()}


def test_standalone_swap_sweep_matches_reference(device):
    """ptrwm_swap_sweep against the reference's own `_attempt_all_swaps()` outputs (tests/golden/pt_sweep.npz)."""
    f = H.load("pt_sweep.npz")
    beta = f["beta_ladder"].astype(np.float32)
    got = gpu_sweep(device, state=f["state_in"], logp=f["logp_in"], beta=beta, event_index=0, dim=30,
                    swap_mode=E.SWAP_REFERENCE_COPY, ext_swap_u=f["swap_u"])
    assert np.array_equal(got["state"], f["state_out"])
    assert np.array_equal(got["logp"], f["logp_out"])
    assert np.array_equal(got["swap_accept"].sum(1), f["num_swap_acceptances"])


@pytest.mark.parametrize("T,D,Cn", [(2, 3, 70), (5, 30, 33), (12, 30, 20), (64, 7, 5), (100, 30, 4), (256, 104, 3),
                                     (200, 33, 2)])
def test_standalone_swap_sweep_vs_oracle(device, T, D, Cn):
    """All four mode / order combinations, Philox uniforms (stream 2) and external ones, ladders up to 256
    temperatures, dims up to 104 (several column chunks through the 32 KB exchange buffer): bit-exact."""
    rng = np.random.default_rng(T * 131 + D)
    st = rng.normal(0, 3, (Cn, T, D)).astype(np.float32)
    lp = rng.normal(-50, 30, (Cn, T)).astype(np.float32)
    lp[0, 0] = -np.inf  # a replica outside the support takes part in swaps like any other
    beta = (0.01 ** (np.arange(T) / (T - 1))).astype(np.float32)
    accepted = 0
    for order in ("sequential", "even_odd"):
        for mode in ("exchange", "reference_copy"):
            for ev in (0, 7):
                for ext in (None, rng.random((Cn, T - 1)).astype(np.float32)):
                    kw = dict(state=st, logp=lp, beta=beta, event_index=ev, swap_mode=E.SWAP_MODES[mode],
                              swap_order=E.SWAP_ORDERS[order], seed=99, chain_offset=3, rng_step=5, ext_swap_u=ext)
                    want = O.swap_sweep(**kw)
                    got = gpu_sweep(device, dim=D, **kw)
                    for k in ("state", "logp", "swap_accept", "last_swap_ordinal"):
                        assert np.array_equal(got[k], want[k]), (k, order, mode, ev, ext is None)
                        accepted += int(want["swap_accept"].sum())
    assert accepted > 0


def test_standalone_sweep_equals_the_fused_kernels_swap(device):
    """MH steps without a swap followed by ptrwm_swap_sweep on Philox stream 1 at the last step's index == the same
    steps with the swap fused into the last one (same states, log-densities and swap counters)."""
    spec = H.target_spec("rc15_d30")
    T, Cn, N = 16, 40, 10
    beta = (0.01 ** (np.arange(T) / (T - 1))).astype(np.float32)
    prop = H.proposal_spec("Normal", 30, beta, base_variance_scalar=2.38**2 / 30)
    st, lp = start_state(spec, Cn, T, np.random.default_rng(1))
    for order in ("sequential", "even_odd"):
        kw = dict(beta=beta, seed=31, chain_offset=2, swap_order=E.SWAP_ORDERS[order])
        fused = gpu_run(spec, prop, device, state=st, logp=lp, step0=0, n_steps=N, swap_every=N, **kw)
        plain = gpu_run(spec, prop, device, state=st, logp=lp, step0=0, n_steps=N, swap_every=10 * N, **kw)
        assert plain["swap_accept"].sum() == 0 and fused["swap_accept"].sum() > 0
        got = gpu_sweep(device, state=plain["state"], logp=plain["logp"], beta=beta, event_index=0, dim=30,
                        rng_step=N - 1, rng_stream=1, **{k: v for k, v in kw.items() if k != "beta"})
        assert np.array_equal(got["state"], fused["state"]) and np.array_equal(got["logp"], fused["logp"])
        assert np.array_equal(got["swap_accept"], fused["swap_accept"])
        assert np.array_equal(got["last_swap_ordinal"], fused["last_swap_ordinal"])



@pytest.mark.parametrize("tkey,pkind,T,pkw", [
    ("rc15_d30", "Normal", 8, dict(base_variance_scalar=2.38**2 / 30)),
    ("rc5_d30", "Normal", 12, dict(base_variance_scalar=2.38**2 / 30)),
    ("even_d30", "Laplace", 5, dict(base_variance_vector=np.full(30, 0.004))),
    ("tm_d50", "UniformRadius", 16, dict(base_radius=2.4)),
    ("hyb_5_4", "Normal", 1, dict(base_variance_scalar=0.02)),
])
def test_split_step_reproduces_the_fused_kernel(device, tkey, pkind, T, pkw):
    """ptrwm_split_propose -> (log-density by ptrwm_logdensity) -> ptrwm_split_accept, step by step, equals
    ptrwm_run bit for bit: states, log-densities and all four statistics, with swaps in every mode and order, from
    Philox and from external randoms.  (This is the path a user-defined density takes.)"""
    spec = H.target_spec(tkey)
    D = spec.dim
    beta = (0.02 ** (np.arange(T) / max(1, T - 1))).astype(np.float32) if T > 1 else np.ones(1, np.float32)
    prop = H.proposal_spec(pkind, D, beta, **pkw) if T > 1 else H.proposal_spec(pkind, D, [1.0], single=True, **pkw)
    Cn, N, burn, se = 9, 36, 5, 4
    rng = np.random.default_rng(zlib.crc32(f"{tkey}{pkind}".encode()))
    st0, lp0 = start_state(spec, Cn, T, rng)
    raw = E.ext_raw_per_step(H.PROPOSAL_KIND[pkind], D)
    for order, mode, ext in (("sequential", "exchange", False), ("even_odd", "exchange", True),
                             ("sequential", "reference_copy", True), ("even_odd", "reference_copy", False)):
        kw = dict(beta=beta, step0=0, burn_in=burn, swap_every=se, seed=123, chain_offset=4,
                  swap_order=E.SWAP_ORDERS[order], swap_mode=E.SWAP_MODES[mode])
        ep = eu = es = None
        if ext:
            ep = _ext_arrays(rng, pkind, N, Cn, T, raw)
            eu = rng.random((N, Cn, T)).astype(np.float32)
            es = rng.random((N // se, Cn, max(T - 1, 1))).astype(np.float32)[:, :, :T - 1] if T > 1 else None
        fused = gpu_run(spec, prop, device, state=st0, logp=lp0, n_steps=N, ext_prop=ep, ext_u=eu, ext_swap_u=es,
                        want_flags=True, **kw)
        # the same run, one split step at a time
        st, lp = dev_t(st0, device), dev_t(lp0, device).reshape(Cn, T).contiguous()
        stats = {k: torch.zeros(Cn, T, dtype=(torch.float64 if k == "sq_jump" else torch.int64), device=device)
                 for k in ("n_accept", "sq_jump", "swap_accept", "last_swap_ordinal")}
        plan = E.RunPlan(None, prop.engine(device), state=st, logp=lp, beta=dev_t(beta, device), burn_in=burn,
                         swap_every=se, swap_mode=E.SWAP_MODES[mode], swap_order=E.SWAP_ORDERS[order], seed=123,
                         chain_offset=4, **stats)
        tgt = spec.engine(device)
        flags = torch.zeros(N, Cn, T, dtype=torch.uint8, device=device)
        ev = 0
        for s in range(N):
            props = plan.split_propose(s, ext_prop=None if ep is None else dev_t(ep[s], device),
                                       ext_u=None if eu is None else dev_t(eu[s], device))
            lp_new = E.logdensity(tgt, props.view(-1, D)).view(Cn, T)
            due = T > 1 and (s + 1) > burn and (s + 1) % se == 0
            plan.split_accept(s, lp_new, accept_flags=flags[s],
                              ext_swap_u=dev_t(es[ev], device) if (due and es is not None) else None)
            ev += 1 if due else 0
        torch.cuda.synchronize()
        assert np.array_equal(flags.cpu().numpy(), fused["accept_flags"]), (order, mode, ext)
        assert np.array_equal(st.cpu().numpy(), fused["state"]), (order, mode, ext)
        assert np.array_equal(lp.cpu().numpy(), fused["logp"])
        for k, v in stats.items():
            assert np.array_equal(v.cpu().numpy(), fused[k]), (k, order, mode, ext)
        if T > 1:
            assert fused["swap_accept"].sum() > 0


@pytest.mark.parametrize("tkey,pkind,T,pkw,order,mode", [
    ("rc15_d30", "Normal", 8, dict(base_variance_scalar=2.38**2 / 30), "sequential", "exchange"),
    ("even_d30", "Laplace", 5, dict(base_variance_vector=np.full(30, 0.004)), "even_odd", "exchange"),
    ("tm_d50", "UniformRadius", 16, dict(base_radius=2.4), "sequential", "reference_copy"),
    ("hyb_5_4", "Normal", 1, dict(base_variance_scalar=0.02), "sequential", "exchange"),
])
def test_split_steps_from_a_device_step_counter_and_a_captured_graph(device, tkey, pkind, T, pkw, order, mode):
    """include/ptrwm.h `device_step`: with a step counter in device memory the argument list of the k-th step of a block
    does not depend on where the run stands (`step0` is an offset added to the counter), so a block of steps - proposal
    kernel, the density's kernels, Metropolis kernel, swap kernel - is captured ONCE in a HIP graph with one counter
    increment at its end and replayed.  Eager calls in device-step mode and graph replays both reproduce ptrwm_run bit for
    bit (states, log-densities, all four statistics; burn-in and the swap schedule are derived from the counter on the
    device), starting from a step index that is not zero.  The block starts at a multiple of swap_every, so the caller
    knows which offsets are swap steps and asserts PTRWM_SPLIT_NO_SWEEP for the others: the swap kernel is not even
    enqueued there."""
    spec = H.target_spec(tkey)
    D = spec.dim
    beta = (0.02 ** (np.arange(T) / max(1, T - 1))).astype(np.float32) if T > 1 else np.ones(1, np.float32)
    prop = H.proposal_spec(pkind, D, beta, **pkw) if T > 1 else H.proposal_spec(pkind, D, [1.0], single=True, **pkw)
    Cn, burn, se, s0 = 9, 7, 4, 3
    K, replays, tail = 8, 3, 5  # steps per captured block (a multiple of swap_every), replays, eager steps behind them
    N = 1 + K * replays + tail  # (s0 + 1 = 4: the block starts at a multiple of swap_every)
    st0, lp0 = start_state(spec, Cn, T, np.random.default_rng(zlib.crc32(tkey.encode())))
    kw = dict(beta=beta, burn_in=burn, swap_every=se, seed=321, chain_offset=2, swap_order=E.SWAP_ORDERS[order],
              swap_mode=E.SWAP_MODES[mode])
    fused = gpu_run(spec, prop, device, state=st0, logp=lp0, step0=s0, n_steps=N, **kw)
    st, lp = dev_t(st0, device), dev_t(lp0, device).reshape(Cn, T).contiguous()
    stats = {k: torch.zeros(Cn, T, dtype=(torch.float64 if k == "sq_jump" else torch.int64), device=device)
             for k in ("n_accept", "sq_jump", "swap_accept", "last_swap_ordinal")}
    plan = E.RunPlan(None, prop.engine(device), state=st, logp=lp, beta=dev_t(beta, device), burn_in=burn, swap_every=se,
                     swap_mode=E.SWAP_MODES[mode], swap_order=E.SWAP_ORDERS[order], seed=321, chain_offset=2, **stats)
    tgt = spec.engine(device)
    counter = torch.full((1,), s0, dtype=torch.int64, device=device)
    plan.set_device_step(counter)

    def step(offset=0, no_sweep=False, advance=1):
        props = plan.split_propose(offset)
        plan.split_accept(offset, E.logdensity(tgt, props.view(-1, D)).view(Cn, T), no_sweep=no_sweep)
        if advance:
            plan.split_advance(advance)

    side = torch.cuda.Stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        step()  # one step outside capture
    torch.cuda.current_stream(device).wait_stream(side)
    assert (s0 + 1) % se == 0
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for j in range(K):  # step counter + j: a swap step iff (j + 1) % swap_every == 0
            step(offset=j, no_sweep=(j + 1) % se != 0, advance=K if j == K - 1 else 0)
    for _ in range(replays):
        g.replay()
    for _ in range(tail):
        step()
    torch.cuda.synchronize()
    assert int(counter.item()) == s0 + N
    assert np.array_equal(st.cpu().numpy(), fused["state"]) and np.array_equal(lp.cpu().numpy(), fused["logp"])
    for k, v in stats.items():
        assert np.array_equal(v.cpu().numpy(), fused[k]), k
    assert fused["n_accept"].sum() > 0 and (T == 1 or fused["swap_accept"].sum() > 0)
    # the flag belongs to device-step mode: refused without a counter
    plan.set_device_step(None)
    with pytest.raises(E.PTRWMError):
        plan.split_accept(0, lp, no_sweep=True)
    # the fused kernel and the stand-alone sweep refuse the field; external randoms are refused with it
    plan.set_device_step(counter)
    with pytest.raises(RuntimeError, match="device-step"):
        plan.swap_sweep(0, 0)
    with pytest.raises(E.PTRWMError):
        plan.split_propose(0, ext_prop=torch.zeros(Cn, T, E.ext_raw_per_step(prop.kind, D), device=device),
                           ext_u=torch.zeros(Cn, T, device=device))


def test_ladders_beyond_the_compiled_workgroup_are_refused(device):
    """dim > 64 runs the lane-split kernel only, compiled for workgroups of up to 512 threads = ladders of up to 128
    temperatures (round 4 retired its 1024-thread class: kernels that spilled 40 VGPRs inside the step loop): a longer
    ladder there is refused with PTRWM_E_NOVARIANT, through the C ABI and - with a sentence - through the classes."""
    spec = H.spec_from_params("RoughCarpetDistributionTorch", 100, {"modes": np.float32([-4, 0, 4]), "weights": np.float32([0.2, 0.5, 0.3])})
    T = 200
    beta = (0.05 ** (np.arange(T) / (T - 1))).astype(np.float32)
    prop = H.proposal_spec("Normal", 100, beta, base_variance_scalar=2.38**2 / 100)
    st, lp = start_state(spec, 2, T, np.random.default_rng(0))
    with pytest.raises(E.PTRWMError) as ei:
        gpu_run(spec, prop, device, state=st, logp=lp, beta=beta, step0=0, n_steps=2, seed=1)
    assert ei.value.code == -8
    assert not E.has_quad_variant(spec.kind, prop.kind, 100, 129) and E.has_quad_variant(spec.kind, prop.kind, 100, 128)
    from algorithms import ParallelTemperingRWM_GPU_Optimized
    from target_distributions import RoughCarpetDistributionTorch

    pt = ParallelTemperingRWM_GPU_Optimized(100, 2.38**2 / 100, RoughCarpetDistributionTorch(100, device=device),
                                            beta_ladder=[float(b) for b in beta], device=device, num_replicas=2, seed=1, trace="none")
    with pytest.raises(ValueError, match="128 temperatures"):
        pt._ensure_started()


@pytest.mark.parametrize("T,dim", [(128, 100), (100, 104), (130, 64), (65, 50)])
def test_wide_ladder_with_large_dim_vs_oracle(device, T, dim):
    """Wide ladders (one workgroup of ceil(T/64) waves, or 4 T lanes in the lane-split form above dim 64) at large dims: more
    than 48 KB of dynamic LDS per workgroup, which needs the raised dynamic-LDS allowance."""
    rng = np.random.default_rng(T + dim)
    beta = (0.05 ** (np.arange(T) / (T - 1))).astype(np.float32)
    prop = H.proposal_spec("Normal", dim, beta, base_variance_scalar=2.38**2 / dim)
    Cn, N = 2, 12
    # a diagonal Gaussian of this dimension
    mean = np.linspace(-1.0, 1.0, dim).astype(np.float32)
    prec = np.linspace(0.5, 2.0, dim).astype(np.float32)
    cst = float(-0.5 * dim * np.log(2 * np.pi) + 0.5 * np.log(prec.astype(np.float64)).sum())
    spec = H.TargetSpec(O.TARGET_DIAG_GAUSSIAN, dim, (cst,), (0,), mean, prec, cls="MultivariateNormalTorch")
    st, lp = start_state(spec, Cn, T, rng)
    kw = dict(state=st, logp=lp, beta=beta, n_steps=N, burn_in=1, swap_every=2,
              ext_prop=rng.standard_normal((N, Cn, T, dim)).astype(np.float32),
              ext_u=rng.random((N, Cn, T)).astype(np.float32),
              ext_swap_u=rng.random((N // 2, Cn, T - 1)).astype(np.float32))
    H.check_parity(gpu_runner(spec, prop, device), H.oracle_runner(spec, prop), spec, prop, exact_states=True, **kw)
    # production (non-trace) variant at the same shape: runs and keeps the log-densities consistent with the states
    got2 = gpu_run(spec, prop, device, state=st, logp=lp, beta=beta, step0=0, n_steps=N, burn_in=1, swap_every=2, seed=5)
    chk = O.logdensity(spec.oracle(), got2["state"].reshape(-1, dim), "f64").reshape(Cn, T)
    assert np.allclose(got2["logp"], chk, rtol=1e-5, atol=1e-3)


def test_every_compiled_variant_against_the_oracle(device):
    """tools/check_all_variants.py: all 11 target kernels x 3 proposals x 19 register widths x {fixture, production}
    (1 254 kernels), one child process per (target, proposal) so that a GPU fault is reported instead of ending the
    test run.  This is the guard that would have caught the width-80 miscompile under the max-ILP scheduler."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_all_variants.py")], capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "FAILED pairs: none" in r.stdout


def test_randomised_configurations_against_the_oracle(device):
    """tools/fuzz_vs_oracle.py, 80 random (target, dim, ladder, proposal, swap mode/order/period, burn-in) cases."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_vs_oracle.py"), "80", "7"], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-1000:])
    assert "80 cases agree" in r.stdout


def test_randomised_split_steps_against_the_fused_kernel(device):
    """tools/fuzz_split.py, 60 random cases: ptrwm_split_propose / ptrwm_logdensity / ptrwm_split_accept reproduce
    ptrwm_run bit for bit (states, log-densities, all statistics) for every target family, proposal and ladder shape."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_split.py"), "60", "3"], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-1000:])
    assert "60 cases: split steps reproduce the fused kernel bit for bit" in r.stdout


def test_philox_stream_quality_on_the_run_layout(device):
    """The in-kernel randoms are this engine's own design (the reference draws torch.randn / torch.rand ahead of its
    loop, rwm_gpu_optimized.py:490-511), so their quality is checked on the layout ptrwm_run really uses - one Philox
    subsequence per (chain, temperature, step): the standardised Normal increments of 4 096 chains x 4 temperatures x
    dim 30 over 4 steps are N(0, 1) (Kolmogorov-Smirnov, kurtosis), the accept uniforms are U(0, 1), and neighbouring
    chains, temperatures, steps and dimensions are uncorrelated."""
    from scipy import stats

    Cn, T, D, steps = 4096, 4, 30, 4
    f32 = np.float32
    beta = np.array([1.0, 0.5, 0.2, 0.05], f32)
    prop = H.proposal_spec("Normal", D, beta, base_variance_scalar=0.37)
    zeros = torch.zeros(Cn, T, D, device=device)
    plan = E.RunPlan(None, prop.engine(device), state=zeros, logp=torch.zeros(Cn, T, device=device), beta=dev_t(beta, device),
                     seed=20240607, chain_offset=123456789)
    z, u, j = [], [], []
    for s in range(steps):
        props = plan.split_propose(s)  # state is zero: the proposals are the increments
        z.append((props / dev_t(prop.temp_scale, device)[None, :, None]).cpu().numpy().astype(np.float64))
        # (the accept scratch has two planes: the uniforms, and the squared length of the increment as the proposal itself
        # counts it - the sum of its Box-Muller pairs' squared radii, never |y - x|^2 recomputed)
        acc_u, jump = plan._split_buffers()[1]
        u.append(acc_u.cpu().numpy().astype(np.float64))
        j.append((jump / dev_t(prop.temp_scale, device)[None, :] ** 2).cpu().numpy().astype(np.float64))
    z, u, j = np.stack(z), np.stack(u), np.stack(j)  # [steps, C, T, D], [steps, C, T], [steps, C, T]
    np.testing.assert_allclose(j, (z ** 2).sum(-1), rtol=2e-5)  # ... and it IS the squared length of the increment
    n = z.size
    assert stats.kstest(z.ravel(), "norm").pvalue > 1e-4
    assert abs(z.mean()) < 5 / np.sqrt(n) and abs(z.var() - 1.0) < 5 * np.sqrt(2.0 / n)
    assert abs(stats.kurtosis(z.ravel())) < 5 * np.sqrt(24.0 / n)
    assert stats.kstest(u.ravel(), "uniform").pvalue > 1e-4
    assert u.min() >= 0.0 and u.max() < 1.0

    def corr(a, b):
        return abs(np.corrcoef(a.ravel(), b.ravel())[0, 1])

    lim = lambda m: 5.0 / np.sqrt(m)  # noqa: E731  (five standard errors of a sample correlation)
    assert corr(z[:, :-1], z[:, 1:]) < lim(z[:, 1:].size)              # neighbouring chains
    assert corr(z[:, :, :-1], z[:, :, 1:]) < lim(z[:, :, 1:].size)      # neighbouring temperatures
    assert corr(z[:-1], z[1:]) < lim(z[1:].size)                        # consecutive steps
    assert corr(z[..., :-1], z[..., 1:]) < lim(z[..., 1:].size)         # neighbouring dimensions (incl. Box-Muller pairs)
    assert corr(z[..., 0::2] ** 2, z[..., 1::2] ** 2) < lim(z[..., 0::2].size)  # the two outputs of a pair, second moments
    assert corr(u, z[..., 0]) < lim(u.size) and corr(u, z[..., -1]) < lim(u.size)  # accept uniform vs its step's normals
    assert corr(u[:, :-1], u[:, 1:]) < lim(u[:, 1:].size)


# ---------------------------------------------------------------------------------------------------------
# state_f64 (include/ptrwm.h): the reference's dtype=torch.float64 (pt_rwm_gpu_optimized.py:134,431-449)
# ---------------------------------------------------------------------------------------------------------
def test_float64_golden_trajectory(device):
    """tests/golden/pt_rc15_geo8_f64.npz: the reference's PT sampler run with dtype=torch.float64 (RoughCarpet d=30, its
    geometric 8-rung ladder, 450 steps), with the double normals and float uniforms it consumed.  The kernel's F64 form,
    fed the same arrays, must reproduce the reference's double states BIT FOR BIT at every temperature and step up to
    a Metropolis / swap decision that differs - and such a decision must be a proven fp32-level flip (the reference
    evaluates the density of its double states in double, the kernel on the state rounded to float)."""
    f = H.load("pt_rc15_geo8_f64.npz")
    spec = H.target_spec(str(f["target_key"]))
    ladder = f["beta_ladder"]
    T, D = len(ladder), spec.dim
    burn, N, se = int(f["burn_in"]), int(f["n_samples"]), int(f["swap_every"])
    total = burn + N
    assert f["chains"].dtype == np.float64 and f["ext_prop"].dtype == np.float64
    prop = H.proposal_spec("Normal", D, ladder, base_variance_scalar=float(f["var"]))
    x0 = np.broadcast_to(f["x0"], (1, T, D)).astype(np.float64).copy()
    lp0 = O.logdensity(spec.oracle(), x0.reshape(-1, D), "f64").astype(np.float32).reshape(1, T)
    kw = dict(state=x0, logp=lp0, beta=ladder.astype(np.float32), n_steps=total, burn_in=burn, swap_every=se,
              ext_prop=np.ascontiguousarray(f["ext_prop"][:, None]), ext_u=np.ascontiguousarray(f["ext_u"][:, None]),
              ext_swap_u=np.ascontiguousarray(f["ext_swap_u"][:, None]), swap_mode=E.SWAP_REFERENCE_COPY)
    chains = f["chains"].transpose(1, 0, 2)  # [step, T, D], row 0 = the initial state
    # the oracle's double instantiation is the reference's trajectory, bit for bit ...
    want = H.oracle_runner(spec, prop)(step0=0, **kw)
    assert want["trace"].dtype == np.float64 and np.array_equal(want["trace"][:, 0], chains[1:])
    assert int(want["swap_accept"].sum()) == int(f["num_swap_acceptances"])
    # ... and the kernel follows it: double states identical, any differing decision proven
    flips = H.check_parity(gpu_runner(spec, prop, device), H.oracle_runner(spec, prop), spec, prop, exact_states=True, **kw)
    got = gpu_runner(spec, prop, device)(step0=0, **kw)
    assert got["trace"].dtype == np.float64
    if not flips:
        assert np.array_equal(got["trace"][:, 0], chains[1:])
        assert int(got["swap_accept"].sum()) == int(f["num_swap_acceptances"])
        esjd = got["sq_jump"][0, 0] / N
        assert esjd == pytest.approx(float(f["esjd"]), rel=1e-12)
    logp_close(got["trace_logp"][:, 0], f["logp_chains"].T[1:], extra_abs=3e-4)


F64_CASES = [("rc15_d30", 8, 5), ("tm_d50", 20, 3), ("even_d30", 3, 11), ("gamma_d5", 1, 40), ("hyb_5_4", 70, 2)]


@pytest.mark.parametrize("tkey,T,Cn", F64_CASES, ids=[f"{c[0]}-T{c[1]}" for c in F64_CASES])
def test_float64_states_vs_oracle(device, tkey, T, Cn):
    """The F64 form on other targets, ladder shapes (one temperature, a wavefront's worth, wider than a wavefront) and
    every swap semantics: external double normals (states bit-identical to the oracle's double path between proven flips)
    and the in-kernel Philox stream (every differing decision proven)."""
    spec = H.target_spec(tkey)
    rng = np.random.default_rng(zlib.crc32(f"f64-{tkey}-{T}".encode()))
    beta = (0.05 ** (np.arange(T) / max(1, T - 1))).astype(np.float32)
    scale = 0.01 if "Rosenbrock" in spec.cls else (0.3 if "Gamma" in spec.cls else 2.38**2 / spec.dim)
    prop = H.proposal_spec("Normal", spec.dim, beta, base_variance_scalar=scale)
    st32, lp = start_state(spec, Cn, T, rng)
    st = st32.astype(np.float64) + 1e-9 * rng.standard_normal(st32.shape)  # genuinely double starting points
    lp = O.logdensity(spec.oracle(), st.reshape(-1, spec.dim), "f64").astype(np.float32).reshape(Cn, T)
    N, burn, se = 60, 7, 4
    n_ev = H.events_upto(N, se, burn)
    for mode, order in (("exchange", "sequential"), ("reference_copy", "even_odd")):
        kw = dict(state=st, logp=lp, beta=beta, n_steps=N, burn_in=burn, swap_every=se, swap_mode=E.SWAP_MODES[mode],
                  swap_order=E.SWAP_ORDERS[order])
        H.check_parity(gpu_runner(spec, prop, device), H.oracle_runner(spec, prop), spec, prop, exact_states=True,
                       ext_prop=rng.standard_normal((N, Cn, T, spec.dim)), ext_u=rng.random((N, Cn, T)).astype(np.float32),
                       ext_swap_u=rng.random((n_ev, Cn, T - 1)).astype(np.float32) if T > 1 else None, **kw)
        H.check_parity_philox(gpu_runner(spec, prop, device), spec, prop, seed=77, chain_offset=5, segment=PHILOX_SEGMENT, **kw)


def test_float64_states_keep_increments_a_float_state_loses(device):
    """What dtype=float64 is for: far from the origin a float state cannot represent x + increment (half an ulp of
    3e4 is 1e-3, the increments here are ~1e-4): the float run accepts moves that leave the state where it was, the double
    run carries them.  Same seed, same Philox stream, same target (a unit Gaussian centred at 3e4)."""
    dim, Cn, N = 8, 64, 200
    mean = np.full(dim, 3.0e4, np.float32)
    spec = H.TargetSpec(O.TARGET_DIAG_GAUSSIAN, dim, (np.float32(-0.5 * dim * np.log(2 * np.pi)),), (0,), mean, np.ones(dim, np.float32))
    prop = H.proposal_spec("Normal", dim, [1.0], base_variance_scalar=1e-8, single=True)
    st = np.broadcast_to(mean.astype(np.float64), (Cn, 1, dim)).copy()
    lp = O.logdensity(spec.oracle(), st.reshape(-1, dim), "f64").astype(np.float32).reshape(Cn, 1)
    kw = dict(logp=lp, beta=np.float32([1.0]), step0=0, n_steps=N, burn_in=0, swap_every=1, seed=9)
    g32 = gpu_run(spec, prop, device, state=st.astype(np.float32), **kw)
    g64 = gpu_run(spec, prop, device, state=st, **kw)
    assert g32["state"].dtype == np.float32 and g64["state"].dtype == np.float64
    assert g32["n_accept"].sum() > 0.9 * Cn * N and g64["n_accept"].sum() > 0.9 * Cn * N  # tiny moves: nearly all accepted
    assert np.array_equal(g32["state"], st.astype(np.float32)) and g32["sq_jump"].sum() == 0.0  # ... and all of them lost
    moved = np.abs(g64["state"] - st)
    assert moved.min() > 0 and 1e-4 < np.sqrt((moved**2).mean()) < 1e-2  # a random walk of 200 steps of 1e-4
    assert g64["sq_jump"].sum() / (Cn * N) == pytest.approx(dim * 1e-8, rel=0.1)


@pytest.mark.parametrize("prop_name", ["Normal", "UniformRadius"])
def test_squared_jump_is_that_of_the_stored_states(device, prop_name):
    """rwm_gpu_optimized.py:513-534 / pt_rwm_gpu_optimized.py:772-789 define the statistic on the STORED states.  The
    production kernels take the squared length from the proposal itself where the two agree (proposals.h kJumpTrust) and
    from the states where they do not.  A batch mixing both kinds of replica - half near the origin, half 3e4 away where a
    float state swallows most of each increment - must report, per replica, the squared jumps of its own stored trace, and
    the same bits in both kernel forms."""
    dim, Cn, N = 30, 64, 60
    far = np.zeros((Cn, 1, 1), np.float32)
    far[Cn // 2:] = 3.0e4
    mean = np.zeros(dim, np.float32)
    spec = H.TargetSpec(O.TARGET_DIAG_GAUSSIAN, dim, (np.float32(-0.5 * dim * np.log(2 * np.pi)),), (0,), mean, np.full(dim, 1e-10, np.float32))
    # (a nearly flat target: every move accepted wherever the replica sits)
    prop = H.proposal_spec(prop_name, dim, [1.0], base_variance_scalar=1e-4, base_radius=0.05, single=True)
    rng = np.random.default_rng(3)
    st = (far + 0.1 * rng.normal(size=(Cn, 1, dim)).astype(np.float32)).astype(np.float32)  # near: |x| < 0.5, trusted
    lp = O.logdensity(spec.oracle(), st.reshape(-1, dim)).reshape(Cn, 1)
    kw = dict(state=st, logp=lp, beta=np.float32([1.0]), step0=0, n_steps=N, burn_in=0, swap_every=1, seed=5, trace_temps=1)
    runs = {}
    for form in (E.FORM_THREAD, E.FORM_QUAD):
        with E.kernel_form(form):
            runs[form] = gpu_run(spec, prop, device, **kw)
    g = runs[E.FORM_THREAD]
    for k in ("state", "logp", "n_accept", "sq_jump", "trace"):
        assert np.array_equal(g[k], runs[E.FORM_QUAD][k]), k
    assert g["n_accept"].min() > 0.8 * N
    path = np.concatenate([st[None], g["trace"]], axis=0).astype(np.float64)  # [N+1, Cn, 1, dim]
    want = (np.diff(path, axis=0) ** 2).sum(axis=(0, 3))
    near_err = np.abs(g["sq_jump"][: Cn // 2] / want[: Cn // 2] - 1).max()
    far_err = np.abs(g["sq_jump"][Cn // 2:] / want[Cn // 2:] - 1).max()
    assert near_err < 1e-5, near_err   # the proposal's own length: the float sum x + inc keeps the increment here
    assert far_err < 1e-6, far_err     # |y - x|^2 of the states themselves (fp32 accumulation order only)
    # ... and the two definitions really differ out there (an ulp of 3e4 is 2e-3, the increments are ~1e-2)
    intended = dim * 1e-4 * N if prop_name == "Normal" else None
    if intended is not None:
        assert np.abs(want[Cn // 2:] / intended - 1).mean() > 5e-3


@pytest.mark.parametrize("prop_name", ["Normal", "UniformRadius"])
def test_squared_jump_across_the_trust_boundary(device, prop_name):
    """The verdict whether a replica's squared jumps may be taken from the proposal (largest coordinate within kJumpTrust =
    256 typical increments) is taken when a launch loads the state and kept for the launch (include/ptrwm.h sq_jump).  A
    replica that starts INSIDE the trusted range (200-255 increments out) and drifts across the boundary within ONE launch
    - a Gaussian target centred 600 increments out pulls it there - keeps the verdict it started with; one that starts just
    outside (257-300) is measured on its states.  Either way the reported sum is the squared distance of its own stored
    states to 1e-4 relative (the north star's bound is 1e-3), in both kernel forms, bit-identical between them."""
    dim, Cn, N = 30, 96, 1500
    scale = 1e-2  # typical increment per dimension
    var = scale**2
    prop = H.proposal_spec(prop_name, dim, [1.0], base_variance_scalar=var, base_radius=scale * np.sqrt(dim), single=True)
    mean = np.zeros(dim, np.float32)
    mean[0] = 600 * scale
    prec = np.full(dim, 1.0 / (20 * scale) ** 2, np.float32)  # a well 20 increments wide: a steady pull towards the centre
    cst = float(-0.5 * dim * np.log(2 * np.pi) + 0.5 * np.log(prec.astype(np.float64)).sum())
    spec = H.TargetSpec(O.TARGET_DIAG_GAUSSIAN, dim, (cst,), (0,), mean, prec, cls="MultivariateNormalTorch")
    rng = np.random.default_rng(11)
    start = np.concatenate([rng.uniform(200, 255, Cn // 2), rng.uniform(257, 300, Cn // 2)]) * scale
    st = (0.1 * scale * rng.normal(size=(Cn, 1, dim))).astype(np.float32)
    st[:, 0, 0] = start
    lp = O.logdensity(spec.oracle(), st.reshape(-1, dim)).reshape(Cn, 1)
    kw = dict(state=st, logp=lp, beta=np.float32([1.0]), step0=0, n_steps=N, burn_in=0, swap_every=1, seed=9, trace_temps=1)
    runs = {}
    for form in (E.FORM_THREAD, E.FORM_QUAD):
        with E.kernel_form(form):
            runs[form] = gpu_run(spec, prop, device, **kw)
    g = runs[E.FORM_THREAD]
    for k in ("state", "logp", "n_accept", "sq_jump", "trace"):
        assert np.array_equal(g[k], runs[E.FORM_QUAD][k]), k
    final = np.abs(g["state"][:, 0, 0]) / scale
    crossed = (start[: Cn // 2] / scale <= 256) & (final[: Cn // 2] > 256)
    assert crossed.sum() >= Cn // 4, (crossed.sum(), final[: Cn // 2].min())  # most trusted replicas walk out of the range
    assert g["n_accept"].min() > 0.1 * N
    path = np.concatenate([st[None], g["trace"]], axis=0).astype(np.float64)
    want = (np.diff(path, axis=0) ** 2).sum(axis=(0, 3))[:, 0]
    err = np.abs(g["sq_jump"][:, 0] / want - 1)
    assert err[: Cn // 2].max() < 1e-4, err[: Cn // 2].max()   # the proposal's own length, verdict older than the crossing
    assert err[Cn // 2:].max() < 1e-5, err[Cn // 2:].max()     # measured on the states from the start


def test_a_ladder_outside_the_support_changes_nothing_for_its_neighbours(device):
    """Which scan a ladder's sequential swap sweep takes - the threshold form or the reference's literal rule - is a verdict
    of the ladder alone, taken at every event from the values it enters the event with (kernel.h swap_pair_plain /
    ladder_votes_plain): a ladder that starts outside a bounded support (log-density -inf: the literal rule refuses its
    swaps) must not change the arithmetic of the ladders it shares a wavefront or workgroup with.  Same batch in the
    one-thread-per-replica form (16 ladders per wavefront), the lane-split form (4 per wavefront), cut into one-step
    launches, through split steps and through stand-alone sweeps: the same bits everywhere; and every ladder that starts
    inside the support gets the bits it gets when ALL ladders start inside."""
    spec = H.target_spec("gamma_d5")
    D, T, Cn, N, se = 5, 4, 48, 24, 2
    beta = np.float32([1.0, 0.6, 0.3, 0.1])
    prop = H.proposal_spec("Laplace", D, beta, base_variance_vector=np.full(D, 0.5))
    st_in, lp_in = start_state(spec, Cn, T, np.random.default_rng(2))
    outside = [3, 17, 18, 40]
    st0, lp0 = st_in.copy(), lp_in.copy()
    for c in outside:
        st0[c, 1:, 0] = -1.0 - 0.1 * np.arange(T - 1)  # all but the cold replica outside (0, inf)
    lp0 = O.logdensity(spec.oracle(), st0.reshape(-1, D)).astype(np.float32).reshape(Cn, T)
    assert np.isneginf(lp0[outside, 1:]).all() and np.isfinite(lp0[outside, 0]).all()
    kw = dict(beta=beta, burn_in=0, swap_every=se, seed=77, chain_offset=1)
    keys = ("state", "logp", "n_accept", "sq_jump", "swap_accept", "last_swap_ordinal")
    with E.kernel_form(E.FORM_THREAD):
        ref = gpu_run(spec, prop, device, state=st0, logp=lp0, step0=0, n_steps=N, **kw)
        clean = gpu_run(spec, prop, device, state=st_in, logp=lp_in, step0=0, n_steps=N, **kw)
        st, lp, tot = st0, lp0, None
        for i in range(N):  # one-step launches
            r = gpu_run(spec, prop, device, state=st, logp=lp, step0=i, n_steps=1, **kw)
            st, lp = r["state"], r["logp"]
            tot = r if tot is None else {k: (np.maximum(tot[k], r[k]) if k == "last_swap_ordinal" else tot[k] + r[k]) for k in keys[2:]}
        assert np.array_equal(st, ref["state"]) and np.array_equal(lp, ref["logp"])
        for k in ("n_accept", "swap_accept", "last_swap_ordinal"):
            assert np.array_equal(tot[k], ref[k]), k
    with E.kernel_form(E.FORM_QUAD):
        quad = gpu_run(spec, prop, device, state=st0, logp=lp0, step0=0, n_steps=N, **kw)
    for k in keys:
        assert np.array_equal(ref[k], quad[k]), k
    inside = np.setdiff1d(np.arange(Cn), outside)
    for k in keys:
        assert np.array_equal(ref[k][inside], clean[k][inside]), k
    assert ref["swap_accept"][inside].sum() > 0 and ref["n_accept"][outside].sum() > 0
    # split steps (Metropolis kernel + stand-alone sweep kernel, one workgroup per ladder)
    std, lpd = dev_t(st0, device), dev_t(lp0, device)
    stats = {k: torch.zeros(Cn, T, dtype=(torch.float64 if k == "sq_jump" else torch.int64), device=device)
             for k in ("n_accept", "sq_jump", "swap_accept", "last_swap_ordinal")}
    plan = E.RunPlan(None, prop.engine(device), state=std, logp=lpd, beta=dev_t(beta, device), burn_in=0, swap_every=se,
                     seed=77, chain_offset=1, **stats)
    tgt = spec.engine(device)
    for i in range(N):
        props = plan.split_propose(i)
        plan.split_accept(i, E.logdensity(tgt, props.view(-1, D)).view(Cn, T))
    torch.cuda.synchronize()
    assert np.array_equal(std.cpu().numpy(), ref["state"]) and np.array_equal(lpd.cpu().numpy(), ref["logp"])
    for k, v in stats.items():
        assert np.array_equal(v.cpu().numpy(), ref[k]), k


def test_swap_uniforms_on_the_edges_of_the_unit_interval(device):
    """A swap uniform of exactly 0 or exactly 1 (external randoms through the C ABI; u = 0 also comes out of the 24-bit
    lattice once in 1.7e7 draws): the reference's `u < min(1, exp(..))` never accepts u = 1 and refuses u = 0 where the
    exponential underflows to 0; a ladder holding such a uniform takes the literal scan (kernel.h swap_pair_plain), so the
    decisions are the oracle's, decision for decision - with pairs whose log-densities differ by thousands among them."""
    spec = H.target_spec("rc15_d30")
    D, T, Cn, N = 30, 6, 40, 8
    beta = (0.05 ** (np.arange(T) / (T - 1))).astype(np.float32)
    prop = H.proposal_spec("Normal", D, beta, base_variance_scalar=2.38**2 / D)
    rng = np.random.default_rng(5)
    st = rng.normal(0, 1.0, (Cn, T, D)).astype(np.float32)
    st[:, ::2] *= 30.0  # every other rung far out: log-densities thousands apart, exp underflows
    lp = O.logdensity(spec.oracle(), st.reshape(-1, D)).astype(np.float32).reshape(Cn, T)
    es = rng.random((N, Cn, T - 1)).astype(np.float32)
    es[:, ::3, 0] = 0.0
    es[:, 1::3, 2] = 1.0
    es[:, 2::3, 4] = 0.0
    kw = dict(state=st, logp=lp, beta=beta, step0=0, n_steps=N, burn_in=0, swap_every=1, seed=3,
              ext_prop=np.zeros((N, Cn, T, D), np.float32), ext_u=np.full((N, Cn, T), 2.0, np.float32), ext_swap_u=es)
    # (zero increments and accept uniforms of 2: the Metropolis step keeps every state - the run is swap events only)
    want = O.run(spec.oracle(), prop.oracle(), state=st.copy(), logp=lp.copy(), beta=beta, step0=0, n_steps=N, burn_in=0,
                 swap_every=1, seed=3, ext_prop=kw["ext_prop"], ext_u=kw["ext_u"], ext_swap_u=es)
    for form in (E.FORM_THREAD, E.FORM_QUAD):
        with E.kernel_form(form):
            got = gpu_run(spec, prop, device, **kw)
        assert np.array_equal(got["swap_accept"], want["swap_accept"]), form
        assert np.array_equal(got["state"], want["state"]) and np.array_equal(got["last_swap_ordinal"], want["last_swap_ordinal"])
    assert want["swap_accept"].sum() > 0


@pytest.mark.parametrize("name", ["configs1_rwm_rc15_normal", "configs2_pt_rc15_normal", "configs3_pt_even_laplace",
                                  "configs4_pt_tm50_uniform"])
def test_long_free_running_philox_runs_match_the_oracle(device, name):
    """The production path on its own over a long horizon: 11 000 steps (burn-in 1 000) of in-kernel Philox, never
    restarted, for each BASELINE family, against the C oracle's free run on the same Philox stream
    (tests/golden/oracle_free_runs.json, tests/golden/generate_oracle_free_runs.py: 8 192 replicas per family).  The two
    follow the same trajectories until an fp32-level flip (proven decision by decision in check_parity_philox) and sample
    the same chain afterwards: per-temperature Metropolis acceptance, per-temperature mean squared jump and the swap
    fraction agree at 1e-3 relative + 4 combined standard errors (errors over ladders)."""
    import importlib.util
    import json

    here = os.path.dirname(os.path.abspath(__file__))
    spec_ = importlib.util.spec_from_file_location("free_runs", os.path.join(here, "golden", "generate_oracle_free_runs.py"))
    G = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(G)
    with open(os.path.join(here, "golden", "oracle_free_runs.json")) as f:
        anchors = json.load(f)
    fam = anchors["families"][name]
    tkey, pkind, pkw, T, Cn = G.FAMILIES[name]
    assert (anchors["burn_in"], anchors["steps"], anchors["swap_every"], anchors["seed"]) == (G.BURN, G.STEPS, G.SE, G.SEED)
    assert fam["ladders"] == Cn and fam["temps"] == T
    spec = H.target_spec(tkey)
    beta = G.ladder(T)
    prop = H.proposal_spec(pkind, spec.dim, beta, **pkw) if T > 1 else H.proposal_spec(pkind, spec.dim, [1.0], single=True, **pkw)
    st, lp = G.start(spec, Cn, T)
    r = gpu_run(spec, prop, device, state=st, logp=lp, beta=beta, step0=0, n_steps=G.BURN + G.STEPS, burn_in=G.BURN,
                swap_every=G.SE, seed=G.SEED, chain_offset=0)

    def close(got, want, se_want, what):
        got = np.asarray(got, np.float64)
        m, se = got.mean(0), got.std(0, ddof=1) / np.sqrt(got.shape[0])
        want, se_want = np.asarray(want), np.asarray(se_want)
        tol = 1e-3 * np.abs(want) + 4 * np.sqrt(se**2 + se_want**2)
        assert np.all(np.abs(m - want) <= tol), (what, float(np.max(np.abs(m - want) / tol)))

    close(r["n_accept"] / G.STEPS, fam["acceptance"]["mean"], fam["acceptance"]["stderr"], "acceptance")
    close(r["sq_jump"] / G.STEPS, fam["mean_sq_jump"]["mean"], fam["mean_sq_jump"]["stderr"], "mean squared jump")
    if T > 1:
        events = (G.BURN + G.STEPS) // G.SE - G.BURN // G.SE
        frac = r["swap_accept"][:, :T - 1].sum(1) / (events * (T - 1))
        close(frac[:, None], [fam["swap_fraction"]["mean"]], [fam["swap_fraction"]["stderr"]], "swap fraction")
        assert 0.02 < fam["swap_fraction"]["mean"] < 0.98
    assert 0.01 < fam["acceptance"]["mean"][0] < 0.9


def test_float64_argument_validation(device):
    spec = H.target_spec("rc15_d30")
    st = torch.zeros(2, 1, 30, device=device, dtype=torch.float64)
    lp = torch.zeros(2, 1, device=device)
    b = torch.ones(1, device=device)
    lap = H.proposal_spec("Laplace", 30, [1.0], base_variance_vector=np.full(30, 0.1), single=True)
    with pytest.raises(E.PTRWMError) as ei:  # external randoms in double: the Normal proposal only
        E.run(spec.engine(device), lap.engine(device), state=st, logp=lp, beta=b, step0=0, n_steps=1,
              ext_prop=torch.zeros(1, 2, 1, 30, device=device, dtype=torch.float64), ext_u=torch.zeros(1, 2, 1, device=device))
    assert ei.value.code == -5
    with pytest.raises(TypeError):  # a float trace for double states
        E.run(spec.engine(device), lap.engine(device), state=st, logp=lp, beta=b, step0=0, n_steps=1,
              trace=torch.zeros(1, 2, 1, 30, device=device))
    plan = E.RunPlan(spec.engine(device), lap.engine(device), state=st, logp=lp, beta=b)
    with pytest.raises(TypeError, match="float32 states"):  # split steps carry float states
        plan.split_propose(0)
    with pytest.raises(TypeError, match="float32 or torch.float64"):
        E.RunPlan(spec.engine(device), lap.engine(device), state=st.half(), logp=lp, beta=b)
    E.run(spec.engine(device), lap.engine(device), state=st, logp=lp, beta=b, step0=0, n_steps=3, seed=1)  # Philox: any proposal
    torch.cuda.synchronize()
    assert st.dtype == torch.float64 and bool(torch.isfinite(st).all())
