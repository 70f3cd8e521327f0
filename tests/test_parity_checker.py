"""The parity comparator itself (tests/helpers.check_parity), exercised on the CPU with the oracle standing in for
BOTH engines: one side is fed deliberately altered randoms, so that it makes (a) no different decision, (b) a
legitimate fp32-level flip (uniform one lattice step across the threshold), (c) a WRONG Metropolis decision at step 15,
(d) a WRONG swap outcome.  (a) and (b) must pass - (b) with the flip proven and the run resynchronised to the end of
the horizon - and (c), (d) must fail.  This is the guard that the GPU parity tests cover their full horizon."""
import numpy as np
import pytest

import helpers as H
from oracle import oracle as O

f32 = np.float32


def _case(pkind="Normal", T=6, Cn=4, N=40, burn=5, se=4, seed=0):
    spec = H.target_spec("rc15_d30")
    beta = (0.05 ** (np.arange(T) / (T - 1))).astype(f32)
    prop = H.proposal_spec(pkind, 30, beta, base_variance_scalar=2.38**2 / 30) if pkind == "Normal" else \
        H.proposal_spec(pkind, 30, beta, base_variance_vector=np.full(30, 0.2, f32))
    rng = np.random.default_rng(seed)
    st = np.zeros((Cn, T, 30), f32)
    lp = np.broadcast_to(O.logdensity(spec.oracle(), np.zeros((1, 30), f32)).astype(f32), (Cn, T)).copy()
    ext = (rng.standard_normal((N, Cn, T, 30)) if pkind == "Normal" else rng.random((N, Cn, T, 30))).astype(f32)
    kw = dict(state=st, logp=lp, beta=beta, n_steps=N, burn_in=burn, swap_every=se, ext_prop=ext,
              ext_u=rng.random((N, Cn, T)).astype(f32),
              ext_swap_u=rng.random((N // se - burn // se, Cn, T - 1)).astype(f32), exact_states=pkind == "Normal")
    return spec, prop, kw


def _tamper(run, step0_of_tamper, edit):
    """An 'engine' that is the oracle fed edited copies of the random arrays (edits addressed by GLOBAL step)."""
    def run_a(**kw):
        kw = dict(kw)
        kw["ext_u"] = kw["ext_u"].copy()
        kw["ext_swap_u"] = None if kw["ext_swap_u"] is None else kw["ext_swap_u"].copy()
        edit(kw)
        return run(**kw)
    return run_a


def test_identical_engines_have_no_flips():
    spec, prop, kw = _case()
    run = H.oracle_runner(spec, prop)
    assert H.check_parity(run, run, spec, prop, **kw) == []
    # fp64 arithmetic against fp32 arithmetic on the same randoms: any flip must be provable (usually none)
    flips = H.check_parity(H.oracle_runner(spec, prop, "f64"), run, spec, prop, **{**kw, "exact_states": False})
    assert len(flips) <= 2


@pytest.mark.parametrize("pkind", ["Normal", "Laplace"])
def test_a_legitimate_flip_is_proven_and_the_run_resynchronised(pkind):
    spec, prop, kw = _case(pkind)
    run = H.oracle_runner(spec, prop)
    s0, c0, t0 = 15, 2, 3
    # place the shared uniform of (s0, c0, t0) just ABOVE exp(r): beyond the fp32 evaluation error of exp(r) (~1e-5
    # relative) but well inside the stated band (~3e-4 beta relative), so the oracle rejects ...
    base = run(state=kw["state"], logp=kw["logp"], beta=kw["beta"], step0=0, n_steps=s0, burn_in=kw["burn_in"],
               swap_every=kw["swap_every"], swap_mode=0, swap_order=0, ext_prop=kw["ext_prop"][:s0], ext_u=kw["ext_u"][:s0],
               ext_swap_u=kw["ext_swap_u"])
    r, *_ = H.step_log_ratios(spec, prop, base["trace"][s0 - 1, c0], kw["ext_prop"][s0, c0], kw["beta"])
    for t in range(len(r)):  # pick a temperature whose ratio is comfortably negative
        if -3.0 < r[t] < -0.05 and kw["beta"][t] > 0.2:
            t0 = t
            break
    thr = np.exp(r[t0])
    eps = 1e-4 * float(kw["beta"][t0])
    kw["ext_u"][s0, c0, t0] = f32(thr * (1 + eps))
    assert kw["ext_u"][s0, c0, t0] > thr

    def edit(k):  # ... and give engine A a uniform just BELOW it (A accepts)
        i = s0 - k["step0"]
        if 0 <= i < k["n_steps"] and k["state"].shape[0] == kw["state"].shape[0]:
            k["ext_u"][i, c0, t0] = f32(thr * (1 - eps))

    flips = H.check_parity(_tamper(run, 0, edit), run, spec, prop, **kw)
    assert len(flips) == 1 and flips[0][:3] == (s0, c0, "mh") and flips[0][3] <= 1.0


def test_a_wrong_metropolis_decision_at_step_15_fails():
    spec, prop, kw = _case()
    run = H.oracle_runner(spec, prop)
    want = run(state=kw["state"], logp=kw["logp"], beta=kw["beta"], step0=0, n_steps=kw["n_steps"],
               burn_in=kw["burn_in"], swap_every=kw["swap_every"], swap_mode=0, swap_order=0, ext_prop=kw["ext_prop"],
               ext_u=kw["ext_u"], ext_swap_u=kw["ext_swap_u"])
    s0 = 15
    rej = np.argwhere((want["accept_flags"][s0] == 0) & (kw["ext_u"][s0] > 0.3))  # a clear rejection at step 15
    c0, t0 = rej[0]

    def edit(k):
        i = s0 - k["step0"]
        if 0 <= i < k["n_steps"] and k["state"].shape[0] == kw["state"].shape[0]:
            k["ext_u"][i, c0, t0] = 0.0  # engine A accepts where it must not

    with pytest.raises(AssertionError, match="WRONG Metropolis decision"):
        H.check_parity(_tamper(run, 0, edit), run, spec, prop, **kw)


@pytest.mark.parametrize("order", [O.ORDER_SEQUENTIAL, O.ORDER_EVEN_ODD])
@pytest.mark.parametrize("mode", [O.SWAP_EXCHANGE, O.SWAP_REFERENCE_COPY])
def test_a_wrong_swap_outcome_fails(order, mode):
    spec, prop, kw = _case(seed=3)
    kw.update(swap_mode=mode, swap_order=order)
    run = H.oracle_runner(spec, prop)
    assert H.check_parity(run, run, spec, prop, **kw) == []
    base = run(step0=0, **{k: v for k, v in kw.items() if k != "exact_states"})

    def make_edit(ev0, c0):
        def edit(k):
            if k["state"].shape[0] != kw["state"].shape[0] or k["step0"] != 0:
                return
            row = k["ext_swap_u"][ev0, c0]
            # invert clear decisions of this event: uniforms near 0 where they were large and vice versa
            k["ext_swap_u"][ev0, c0] = np.where(row > 0.5, 1e-6, 0.999999).astype(f32)
        return edit

    # an (event, ladder) whose inverted uniforms really change the outcome
    for ev0, c0 in [(e, c) for e in range(2, kw["ext_swap_u"].shape[0]) for c in range(kw["state"].shape[0])]:
        tampered = _tamper(run, 0, make_edit(ev0, c0))
        out = tampered(step0=0, **{k: v for k, v in kw.items() if k != "exact_states"})
        if not np.array_equal(out["trace"], base["trace"]):
            break
    else:
        pytest.fail("no tampering changed the trajectory")
    with pytest.raises(AssertionError, match="WRONG swap outcome"):
        H.check_parity(tampered, run, spec, prop, **kw)


def test_stat_mismatch_without_decision_difference_fails():
    spec, prop, kw = _case()
    run = H.oracle_runner(spec, prop)

    def run_a(**k):
        out = run(**k)
        out["n_accept"] = out["n_accept"].copy()
        out["n_accept"][0, 0] += 1  # bookkeeping bug
        return out

    with pytest.raises(AssertionError, match="n_accept"):
        H.check_parity(run_a, run, spec, prop, **kw)


def test_an_invisible_swap_flip_must_be_provable():
    """A swap between replicas holding the same state (the reference's row copy produces such pairs) changes only the
    counters.  The comparator accepts a counter difference on pair j only if some swap event of the segment has pair
    j's uniform inside the fp32 band of its threshold; otherwise it is a bookkeeping error."""
    spec, prop, kw = _case(seed=5)
    kw.update(swap_mode=O.SWAP_REFERENCE_COPY)
    run = H.oracle_runner(spec, prop)
    c0, j0 = 2, 1

    def run_a(**k):
        out = run(**k)
        if k["state"].shape[0] == kw["state"].shape[0]:
            out["swap_accept"] = out["swap_accept"].copy()
            out["swap_accept"][c0, j0] += 1
        return out

    with pytest.raises(AssertionError, match="swap bookkeeping"):
        H.check_parity(run_a, run, spec, prop, **kw)
    # now put pair j0's uniform of one event of ladder c0 on its threshold: the same difference becomes provable
    base = run(step0=0, **{k: v for k, v in kw.items() if k != "exact_states"})
    ev, d = 3, None
    steps = [s for s in range(kw["n_steps"]) if (s + 1) > kw["burn_in"] and (s + 1) % kw["swap_every"] == 0]
    d = steps[ev]
    pre = base["trace"][d - 1, c0]
    _, l_x, l_y, _ = H.step_log_ratios(spec, prop, pre, kw["ext_prop"][d, c0], kw["beta"])
    lm = np.where(base["accept_flags"][d, c0].astype(bool), l_y, l_x)
    # reference_copy, sequential: pair j compares the rows as the sweep left them; pair j0 = 1 sees row 1 (untouched so far)
    lm_seq = lm.copy()
    for j in range(j0):  # replay the earlier pairs to know what sits at j0
        lpr = (kw["beta"][j] - kw["beta"][j + 1]) * (lm_seq[j + 1] - lm_seq[j])
        if kw["ext_swap_u"][ev, c0, j] < min(1.0, np.exp(lpr)):
            lm_seq[j] = lm_seq[j + 1]
    lpr = float((kw["beta"][j0] - kw["beta"][j0 + 1]) * (lm_seq[j0 + 1] - lm_seq[j0]))
    thr = min(1.0, float(np.exp(lpr)))
    kw["ext_swap_u"][ev, c0, j0] = f32(min(thr * (1 - 2e-6), 1 - 2e-7)) if thr > 0.01 else f32(thr)
    out2 = run(step0=0, **{k: v for k, v in kw.items() if k != "exact_states"})

    def run_b(**k):  # engine A: the oracle's outputs with ONE extra count on that pair (as an invisible flip would leave)
        out = run(**k)
        if k["state"].shape[0] == kw["state"].shape[0]:
            out["swap_accept"] = out["swap_accept"].copy()
            out["swap_accept"][c0, j0] += 1
        return out

    flips = H.check_parity(run_b, run, spec, prop, **kw)
    assert ("swap-invisible" in [f[2] for f in flips]) and out2["swap_accept"].shape == base["swap_accept"].shape


def _edge_case():
    """IIDGamma (shape 1: the density is finite right up to the edge x = 0) with a Laplace proposal, whose states are
    compared with a tolerance.  Returns the case and the two neighbouring lattice uniforms of (step 3, ladder 0,
    temperature 0, coordinate 0) between which the proposal's first coordinate changes sign."""
    D, T, Cn, N, s0 = 3, 2, 2, 8, 3
    spec = H.spec_from_params("IIDGammaTorch", D, {"shape": f32(1.0), "scale": f32(1.0)})
    beta = np.array([1.0, 0.5], f32)
    prop = H.proposal_spec("Laplace", D, beta, base_variance_vector=np.full(D, 2.0, f32))
    rng = np.random.default_rng(5)
    st = np.full((Cn, T, D), 1.0, f32)
    lp = np.broadcast_to(O.logdensity(spec.oracle(), st[0, :1]).astype(f32), (Cn, T)).copy()
    ext = (0.5 + 0.02 * (rng.random((N, Cn, T, D)) - 0.5)).astype(f32)  # small moves: the chain stays near 1
    kw = dict(state=st, logp=lp, beta=beta, n_steps=N, burn_in=0, swap_every=4, ext_prop=ext,
              ext_u=rng.random((N, Cn, T)).astype(f32), ext_swap_u=rng.random((N // 4, Cn, T - 1)).astype(f32),
              exact_states=False, state_rtol=2e-5, state_atol=2e-5)
    run = H.oracle_runner(spec, prop)
    base = run(state=st, logp=lp, beta=beta, step0=0, n_steps=s0, burn_in=0, swap_every=4, swap_mode=0, swap_order=0,
               ext_prop=ext[:s0], ext_u=kw["ext_u"][:s0], ext_swap_u=kw["ext_swap_u"])
    x = base["trace"][s0 - 1, 0]  # [T, D] state before step s0
    scale = float(prop.dim_scale[0]) * float(prop.temp_scale[0])
    u_star = 0.5 - (1.0 - np.exp(-float(x[0, 0]) / scale)) / 2.0  # increment = -x: the proposal lands on 0
    cands = (np.round(u_star * 2**24) + np.arange(-16, 17)) * 2.0**-24
    rows = np.repeat(ext[s0, 0][None], len(cands), axis=0)
    rows[:, 0, 0] = cands.astype(f32)
    y0 = np.array([H.step_log_ratios(spec, prop, x, r, beta)[3][0, 0] for r in rows])
    k = int(np.nonzero((y0[:-1] <= 0) != (y0[1:] <= 0))[0][0])
    u_out, u_in = (cands[k], cands[k + 1]) if y0[k] <= 0 else (cands[k + 1], cands[k])
    kw["ext_u"][s0, 0, 0] = f32(1e-6)  # whoever sees a finite log-density accepts
    return spec, prop, kw, run, s0, f32(u_in), f32(u_out)


def test_a_proposal_on_the_edge_of_the_support_is_a_provable_flip():
    """Found by the randomised runs (IIDGamma / UniformRadius): two engines whose proposals agree to the stated tolerance
    can land on different sides of x = 0 - one sees -inf and rejects, the other accepts.  Legitimate only when the
    proposal is within that tolerance of the edge; check_parity proves it and resynchronises."""
    spec, prop, kw, run, s0, u_in, u_out = _edge_case()
    kw["ext_prop"][s0, 0, 0, 0] = u_out  # the reference side: just outside, -inf, rejected

    def run_a(**k):  # the other engine: one lattice step away, just inside, accepted
        k = dict(k)
        i = s0 - k["step0"]
        if 0 <= i < k["n_steps"] and k["state"].shape[0] == kw["state"].shape[0]:
            k["ext_prop"] = k["ext_prop"].copy()
            k["ext_prop"][i, 0, 0, 0] = u_in
        return run(**k)

    flips = H.check_parity(run_a, run, spec, prop, **kw)
    assert [f[:3] for f in flips] == [(s0, 0, "mh-edge")]
    # with bit-identical proposals (exact_states) the edge cannot separate two engines: the same difference must fail
    with pytest.raises(AssertionError, match="not an fp32-level flip"):
        H.check_parity(run_a, run, spec, prop, **{**kw, "exact_states": True})


def test_a_decision_flip_far_from_the_edge_of_the_support_fails():
    spec, prop, kw, run, s0, u_in, u_out = _edge_case()
    kw["ext_prop"][s0, 0, 0, 0] = f32(0.02)  # the reference side: far outside the support (x_0 < -1), rejected

    def run_a(**k):  # the other engine ignores that and moves somewhere legal: a wrong decision
        k = dict(k)
        i = s0 - k["step0"]
        if 0 <= i < k["n_steps"] and k["state"].shape[0] == kw["state"].shape[0]:
            k["ext_prop"] = k["ext_prop"].copy()
            k["ext_prop"][i, 0, 0, 0] = f32(0.5)
        return run(**k)

    with pytest.raises(AssertionError, match="WRONG Metropolis decision"):
        H.check_parity(run_a, run, spec, prop, **kw)


# ---------------------------------------------------------------------------------------------------------
# Philox mode (the production path): the oracle exports the numbers it draws, check_parity_philox proves every flip
# ---------------------------------------------------------------------------------------------------------
def _philox_case(pkind, T=5, Cn=3, dim=30):
    spec = H.target_spec("rc15_d30")
    beta = (0.05 ** (np.arange(T) / max(1, T - 1))).astype(f32)
    pkw = {"Normal": dict(base_variance_scalar=2.38**2 / dim), "Laplace": dict(base_variance_vector=np.full(dim, 0.2, f32)),
           "UniformRadius": dict(base_radius=2.5)}[pkind]
    prop = H.proposal_spec(pkind, dim, beta, **pkw)
    st = np.zeros((Cn, T, dim), f32)
    lp = np.broadcast_to(O.logdensity(spec.oracle(), np.zeros((1, dim), f32)).astype(f32), (Cn, T)).copy()
    return spec, prop, dict(state=st, logp=lp, beta=beta, n_steps=45, burn_in=7, swap_every=4, seed=0xABCDEF0123,
                            chain_offset=(1 << 33) + 5, step0=11)


def _oracle_philox_engine(spec, prop, precision="f32"):
    def run(**kw):
        Cn, T = kw["state"].shape[:2]
        return O.run(spec.oracle(), prop.oracle(), trace_chains=Cn, trace_temps=T, want_flags=True, precision=precision, **kw)
    return run


@pytest.mark.parametrize("pkind", ["Normal", "Laplace", "UniformRadius"])
@pytest.mark.parametrize("order", [O.ORDER_SEQUENTIAL, O.ORDER_EVEN_ODD])
def test_the_oracle_in_philox_mode_equals_the_oracle_on_its_exported_randoms(pkind, order):
    """oracle_philox_randoms writes out exactly the numbers the oracle's Philox mode consumes: both runs are the same
    float operations, so everything - states, traces, flags, fp64 jump sums, swap bookkeeping - is bit-identical."""
    spec, prop, kw = _philox_case(pkind)
    Cn, T, D = kw["state"].shape
    ep, eu, es = O.philox_randoms(prop.kind, D, T, Cn, seed=kw["seed"], step0=kw["step0"], n_steps=kw["n_steps"],
                                  burn_in=kw["burn_in"], swap_every=kw["swap_every"], chain_offset=kw["chain_offset"])
    assert es.shape[0] == H.events_upto(kw["step0"] + kw["n_steps"], 4, 7) - H.events_upto(kw["step0"], 4, 7)
    run = _oracle_philox_engine(spec, prop)
    a = run(swap_order=order, **kw)
    b = run(swap_order=order, ext_prop=ep, ext_u=eu, ext_swap_u=es, **{k: v for k, v in kw.items() if k != "seed"})
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f"), k
    assert a["n_accept"].sum() > 0 and a["swap_accept"].sum() > 0


@pytest.mark.parametrize("pkind", ["Normal", "Laplace", "UniformRadius"])
def test_philox_parity_of_the_oracle_with_itself_and_with_its_fp64_twin(pkind):
    spec, prop, kw = _philox_case(pkind)
    assert H.check_parity_philox(_oracle_philox_engine(spec, prop), spec, prop, **kw) == []
    assert H.check_parity_philox(_oracle_philox_engine(spec, prop), spec, prop, segment=16, **kw) == []
    # an engine with different arithmetic on the same stream (the fp64 oracle: Box-Muller, transforms and densities in
    # double): whatever differs must be a proven flip, to the end of the horizon
    flips = H.check_parity_philox(_oracle_philox_engine(spec, prop, "f64"), spec, prop, segment=16, **kw)
    assert all(f[2] in ("mh", "swap", "swap-invisible", "mh-edge") for f in flips)


@pytest.mark.parametrize("slip", ["accept_word", "normal_pair", "swap_word", "chain_id"])
def test_a_slipped_philox_word_fails(slip):
    """What the production-path check exists for: an engine whose accept uniform comes from the
    NEXT Philox word, whose normals are taken one pair late, whose swap uniform is another word of its block, or which
    numbers its chains from the wrong offset, follows the oracle statistically (same acceptance rate) but not decision
    for decision - every one of them must FAIL the Philox-mode comparison."""
    spec, prop, kw = _philox_case("Normal")
    Cn, T, D = kw["state"].shape
    args = dict(seed=kw["seed"], step0=kw["step0"], n_steps=kw["n_steps"], burn_in=kw["burn_in"],
                swap_every=kw["swap_every"])
    run = _oracle_philox_engine(spec, prop)

    def engine(**k):
        off, s0, n = k.pop("chain_offset"), k["step0"], k["n_steps"]
        k.pop("seed")
        c = k["state"].shape[0]
        a = dict(args, step0=s0, n_steps=n)
        ep, eu, es = O.philox_randoms(prop.kind, D, T, c, chain_offset=off + (1 if slip == "chain_id" else 0), **a)
        if slip == "accept_word":  # word 2 ceil(D/2) + 1 instead of 2 ceil(D/2): the UniformRadius map's accept word
            eu = O.philox_randoms(O.PROPOSAL_UNIFORM_RADIUS, D, T, c, chain_offset=off, **a)[1]
        elif slip == "normal_pair":
            ep = np.roll(ep, 2, axis=-1)
        elif slip == "swap_word":  # temperature t + 1's block instead of temperature t's
            es = np.roll(es, 1, axis=-1)
        return run(ext_prop=ep, ext_u=eu, ext_swap_u=es, chain_offset=off, **k)

    with pytest.raises(AssertionError):
        H.check_parity_philox(engine, spec, prop, **kw)
    with pytest.raises(AssertionError):
        H.check_parity_philox(engine, spec, prop, segment=16, **kw)
