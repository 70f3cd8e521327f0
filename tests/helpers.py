"""Shared test plumbing: golden fixtures -> target/proposal descriptions for the oracle and the engine."""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

KIND = {
    "RoughCarpetDistributionTorch": O.TARGET_ROUGH_CARPET,
    "ThreeMixtureDistributionTorch": O.TARGET_THREE_MIXTURE,
    "FullRosenbrockTorch": O.TARGET_FULL_ROSENBROCK,
    "EvenRosenbrockTorch": O.TARGET_EVEN_ROSENBROCK,
    "HybridRosenbrockTorch": O.TARGET_HYBRID_ROSENBROCK,
    "IIDGammaTorch": O.TARGET_IID_GAMMA,
    "IIDBetaTorch": O.TARGET_IID_BETA,
    "MultivariateNormalTorch": O.TARGET_DIAG_GAUSSIAN,
    "ScaledMultivariateNormalTorch": O.TARGET_DIAG_GAUSSIAN,
    "HypercubeTorch": O.TARGET_HYPERCUBE,
    "NealFunnelTorch": O.TARGET_NEAL_FUNNEL,
}
PROPOSAL_KIND = {"Normal": O.PROPOSAL_NORMAL, "Laplace": O.PROPOSAL_LAPLACE, "UniformRadius": O.PROPOSAL_UNIFORM_RADIUS}

f32 = np.float32


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@dataclass
class TargetSpec:
    """Backend-neutral target description: the fields of ptrwm_target_desc with host arrays."""

    kind: int
    dim: int
    p: tuple = ()
    ip: tuple = ()
    vec0: Optional[np.ndarray] = None
    vec1: Optional[np.ndarray] = None
    cls: str = ""
    params: dict = field(default_factory=dict)

    def oracle(self):
        return O.Target(self.kind, self.dim, self.p, self.ip, self.vec0, self.vec1)

    def engine(self, device):
        import torch

        import ptrwm_hip as E

        def dev(a):
            return None if a is None else torch.tensor(np.asarray(a, dtype=f32), device=device)

        return E.Target(self.kind, self.dim, tuple(self.p), tuple(self.ip), dev(self.vec0), dev(self.vec1))


def _lgamma32(v):
    import math

    return f32(math.lgamma(float(v)))


def spec_from_params(cls: str, dim: int, params: dict) -> TargetSpec:
    """Folds class-level parameters (as the reference classes store them, fp32) into descriptor fields the way
    rwm-pt-pytorch_amd/target_distributions/*.engine_target does."""
    kind = KIND[cls]
    g = lambda k: np.asarray(params[k], dtype=f32)  # noqa: E731
    if kind == O.TARGET_ROUGH_CARPET:
        lw = np.log(g("weights"))
        sf = g("scaling_factors") if "scaling_factors" in params else None
        lj = f32(np.sum(np.log(sf), dtype=f32)) if sf is not None else f32(0)
        return TargetSpec(kind, dim, (*g("modes"), *lw, lj), (), sf, None, cls, params)
    if kind == O.TARGET_THREE_MIXTURE:
        lw = np.log(g("mixing_weights"))
        log_2pi = f32(np.log(f32(2 * np.pi)))
        lnc = f32(-0.5) * (f32(dim) * log_2pi + f32(0))
        sf = g("scaling_factors") if "scaling_factors" in params else None
        if sf is not None:
            c = (lnc + f32(np.sum(np.log(sf), dtype=f32))) + lw
        else:
            c = lnc + lw
        # ip[0] = 1 (include/ptrwm.h): means that differ in the first coordinate only -> the engine's ThreeMixture1 functor
        m = g("means").reshape(3, dim)
        first_only = bool(np.array_equal(m[0, 1:], m[1, 1:]) and np.array_equal(m[1, 1:], m[2, 1:]))
        return TargetSpec(kind, dim, tuple(c.astype(f32)), (1 if first_only else 0,), m.reshape(-1), sf, cls, params)
    if kind in (O.TARGET_FULL_ROSENBROCK, O.TARGET_EVEN_ROSENBROCK):
        return TargetSpec(kind, dim, (g("a_coeff"), g("b_coeff")), (), g("mu"), None, cls, params)
    if kind == O.TARGET_HYBRID_ROSENBROCK:
        return TargetSpec(kind, dim, (g("a_coeff"), g("b_coeff"), g("mu")), (params["n1"], params["n2"]), None, None,
                          cls, params)
    if kind == O.TARGET_IID_GAMMA:
        k, th = g("shape"), g("scale")
        lnc = f32(dim) * (_lgamma32(k) + k * np.log(th))
        return TargetSpec(kind, dim, (k, th, lnc), (), None, None, cls, params)
    if kind == O.TARGET_IID_BETA:
        a, b = g("alpha"), g("beta")
        lnc = f32(dim) * (_lgamma32(a + b) - _lgamma32(a) - _lgamma32(b))
        return TargetSpec(kind, dim, (a, b, lnc), (), None, None, cls, params)
    if cls == "MultivariateNormalTorch":
        cov = np.asarray(params["cov"], dtype=np.float64)
        assert np.count_nonzero(cov - np.diag(np.diag(cov))) == 0
        return TargetSpec(kind, dim, (g("log_norm_const"),), (0,), g("mean"), (1.0 / np.diag(cov)).astype(f32), cls,
                          params)
    if cls == "ScaledMultivariateNormalTorch":
        return TargetSpec(kind, dim, (g("log_norm_const"),), (1,), g("scaling_factors"), None, cls, params)
    if kind == O.TARGET_HYPERCUBE:
        return TargetSpec(kind, dim, (g("left_boundary"), g("right_boundary"), g("log_uniform_density")), (), None, None,
                          cls, params)
    if kind == O.TARGET_NEAL_FUNNEL:
        return TargetSpec(kind, dim, (g("mu_v"), g("sigma_v_sq"), g("mu_z")), (), None, None, cls, params)
    raise KeyError(cls)


_LOGD = None


def golden_targets():
    """{key: (TargetSpec, x[n, D], reference logp[n])} from logdensity.npz."""
    global _LOGD
    if _LOGD is None:
        z = load("logdensity.npz")
        meta = json.loads(bytes(z["meta_json"]).decode())
        out = {}
        for key, m in meta.items():
            params = {k[len(key) + 4:]: z[k] for k in z.files if k.startswith(key + "__p_")}
            params.update({k: v for k, v in m.items() if k in ("n1", "n2")})
            out[key] = (spec_from_params(m["class"], m["dim"], params), z[f"{key}__x"], z[f"{key}__logp"], m)
        _LOGD = out
    return _LOGD


def target_spec(key) -> TargetSpec:
    return golden_targets()[key][0]


@dataclass
class ProposalSpec:
    kind: int
    temp_scale: np.ndarray
    dim_scale: Optional[np.ndarray] = None
    inv_dim: float = 0.0

    def oracle(self):
        return O.Proposal(self.kind, self.temp_scale, self.dim_scale, self.inv_dim)

    def engine(self, device):
        import torch

        import ptrwm_hip as E

        ds = None if self.dim_scale is None else torch.tensor(np.asarray(self.dim_scale, f32), device=device)
        return E.Proposal(self.kind, torch.tensor(np.asarray(self.temp_scale, f32), device=device), ds, self.inv_dim)


def proposal_spec(kind_name: str, dim: int, betas, *, base_variance_scalar=None, base_variance_vector=None,
                  base_radius=None, single=False) -> ProposalSpec:
    """Kernel-side proposal parameters by the rules of proposal_distributions/*.engine_proposal.
    single=True: one temperature, scale tempered by betas[0] the way the proposal's constructor does."""
    betas = np.asarray(betas, dtype=np.float64)
    kind = PROPOSAL_KIND[kind_name]
    if kind == O.PROPOSAL_NORMAL:
        ts = np.sqrt((float(base_variance_scalar) / betas).astype(f32))
        return ProposalSpec(kind, ts.astype(f32))
    if kind == O.PROPOSAL_LAPLACE:
        bv = np.asarray(base_variance_vector, dtype=f32)
        if single:
            return ProposalSpec(kind, np.ones(1, f32), np.sqrt((bv / f32(betas[0])) / f32(2)).astype(f32))
        return ProposalSpec(kind, (f32(1) / np.sqrt(betas.astype(f32))).astype(f32), np.sqrt(bv / f32(2)).astype(f32))
    r = (f32(base_radius) / np.sqrt(betas.astype(f32))).astype(f32)
    return ProposalSpec(kind, r, None, 1.0 / dim)


def first_mismatch(a, b):
    """Index of the first step at which two [steps, ...] arrays differ, or None."""
    neq = np.any((a != b).reshape(a.shape[0], -1), axis=1)
    idx = np.nonzero(neq)[0]
    return None if idx.size == 0 else int(idx[0])


# (fixture, seed) tables of tests/golden/generate_golden.py (gen_rwm / gen_pt): the value `np.random.seed` received
# just before the reference sampler was constructed (the initial state is drawn from the global NumPy RNG)
RWM_FIXTURE_SEEDS = {
    "rwm_rc15_normal": 42, "rwm_rc4_normal_beta": 43, "rwm_even_laplace": 44, "rwm_tm_uniform": 45,
    "rwm_full_normal": 46, "rwm_hyb_laplace": 47, "rwm_gamma_normal": 48, "rwm_beta_uniform": 49,
    "rwm_rc15s_normal": 50, "rwm_tms_normal": 51, "rwm_mvn_laplace": 52, "rwm_smvn_normal": 53,
    "rwm_cube_uniform": 54, "rwm_funnel_normal": 55,
}
PT_FIXTURE_SEEDS = {"pt_rc15_geo8": 142, "pt_rc5_fine12": 143, "pt_tm15_t32": 144, "pt_even_t5": 145, "pt_hyb_t4": 146}


def build_target_class(key, device):
    """The drop-in target class for a golden target key, from the constructor arguments generate_golden.make_targets
    gave the reference class (scaled variants: the reference's drawn scaling factors are installed afterwards)."""
    import torch

    import target_distributions as TD

    c15 = [[-15.0] + [0.0] * 29, [0.0] * 30, [15.0] + [0.0] * 29]
    cg = [list(np.linspace(-3, 1, 10)), list(np.linspace(0.5, -0.5, 10)), list(np.linspace(2, 4, 10))]
    G = golden_targets()
    make = {
        "rc15_d30": lambda: TD.RoughCarpetDistributionTorch(30, device=device, mode_centers=[-15.0, 0.0, 15.0]),
        "rc5_d30": lambda: TD.RoughCarpetDistributionTorch(30, device=device),
        "rc4_d20": lambda: TD.RoughCarpetDistributionTorch(20, device=device, mode_centers=[-4.0, 0.0, 4.0],
                                                           mode_weights=[0.2, 0.5, 0.3]),
        "rc15s_d10": lambda: TD.RoughCarpetDistributionTorch(10, scaling=True, device=device,
                                                             mode_centers=[-15.0, 0.0, 15.0]),
        "tm_d50": lambda: TD.ThreeMixtureDistributionTorch(50, device=device),
        "tm15_d30": lambda: TD.ThreeMixtureDistributionTorch(30, device=device, mode_centers=c15),
        "tms_d10": lambda: TD.ThreeMixtureDistributionTorch(10, scaling=True, device=device, mode_centers=cg,
                                                            mode_weights=[0.2, 0.3, 0.5]),
        "full_d30": lambda: TD.FullRosenbrockTorch(30, device=device),
        "full_d10": lambda: TD.FullRosenbrockTorch(10, a_coeff=0.1, b_coeff=2.0, mu=torch.linspace(0.5, 1.5, 9),
                                                   device=device),
        "even_d30": lambda: TD.EvenRosenbrockTorch(30, device=device),
        "hyb_3_5": lambda: TD.HybridRosenbrockTorch(3, 5, device=device),
        "hyb_5_4": lambda: TD.HybridRosenbrockTorch(5, 4, device=device),
        "gamma_d50": lambda: TD.IIDGammaTorch(50, device=device),
        "gamma_d5": lambda: TD.IIDGammaTorch(5, shape=3.5, scale=0.7, device=device),
        "beta_d50": lambda: TD.IIDBetaTorch(50, device=device),
        "beta_d5": lambda: TD.IIDBetaTorch(5, alpha=1.5, beta=4.0, device=device),
        "mvn_d50": lambda: TD.MultivariateNormalTorch(50, device=device),
        "smvn_d20": lambda: TD.ScaledMultivariateNormalTorch(
            20, scaling_factors=G["smvn_d20"][0].params["scaling_factors"], device=device),
        "cube_d5": lambda: TD.HypercubeTorch(5, device=device),
        "funnel_d10": lambda: TD.NealFunnelTorch(10, device=device),
    }
    return make[key]()


# ---------------------------------------------------------------------------------------------------------
# Full-horizon parity with PROVEN flips (VERDICT r01 "next" #2).
#
# Two engines (the HIP kernel and the oracle, or - in the CPU self-test - two oracle runs) consume identical random
# arrays.  They may legitimately disagree on a Metropolis or swap decision only where the uniform sits inside the
# fp32 error band of its threshold.  `check_parity` walks every ladder to the END of the horizon: at the first step
# where a ladder's decisions or states differ it recomputes that step's log-ratio in fp64 from the last agreed
# state, ASSERTS the uniform is within the stated log-density tolerance of exp(ratio), then restarts both engines
# from the oracle's state of that step and keeps comparing.  A wrong decision anywhere in the horizon fails.
# ---------------------------------------------------------------------------------------------------------
def logp_tol(l):
    """Stated fp32 log-density tolerance (tests/test_gpu_engine_parity.py header): 4e-6 max(1, |l|) + 1e-4."""
    return 4e-6 * np.maximum(1.0, np.abs(np.asarray(l, dtype=np.float64))) + 1e-4


def events_upto(sc, swap_every, burn_in):
    """Swap events with step_counter <= sc (multiples m * swap_every with burn_in < m * swap_every <= sc)."""
    return max(0, sc // swap_every - burn_in // swap_every)


def step_log_ratios(spec, prop, pre_state, ext_row, beta):
    """fp64 log accept ratios beta_t (l(y_t) - l(x_t)) of ONE ladder for one step, and the pieces a proof needs.
    pre_state [T, D] float32 (the last agreed state), ext_row [T, raw] the step's raw proposal randoms."""
    T, D = pre_state.shape
    if np.asarray(pre_state).dtype == np.float64:
        # state_f64 mode (Normal proposal): y = x + scale * z, one double product and one double sum, as the reference's
        # dtype=torch.float64 path and the kernel's F64 form compute it
        assert prop.kind == O.PROPOSAL_NORMAL
        x = np.ascontiguousarray(pre_state, dtype=np.float64)
        y = x + np.asarray(ext_row, np.float64) * np.asarray(prop.temp_scale, f32).astype(np.float64)[:, None]
    else:
        inc = O.propose(prop.oracle(), D, 1, ext_raw=np.ascontiguousarray(ext_row, dtype=f32)[None])[0]  # fp32 arithmetic
        x = np.ascontiguousarray(pre_state, dtype=f32)
        y = (x + inc.astype(f32)).astype(f32)
    l_x = O.logdensity(spec.oracle(), x, "f64")
    l_y = O.logdensity(spec.oracle(), y, "f64")
    with np.errstate(invalid="ignore"):
        r = np.asarray(beta, dtype=np.float64) * (l_y - l_x)
    return r, l_x, l_y, y


def _prove_mh_flip(r, l_x, l_y, beta_t, u, slack, where):
    assert np.isfinite(l_x) and np.isfinite(l_y), f"{where}: decisions differ although a log-density is not finite " \
                                                  f"(l {l_x}, l' {l_y}): not an fp32-level flip"
    tol_r = float(beta_t) * float(logp_tol(l_x) + logp_tol(l_y)) * slack
    thr = float(np.exp(min(r, 0.0)))
    gap = abs(float(u) - thr) if r < 0 else (0.0 if r <= tol_r else np.inf)
    # |u - exp(r)| <= exp(r) * (beta * 2 * (4e-6 max(1, |l|) + 1e-4))  [+ one lattice step of u]
    assert r <= tol_r and gap <= thr * tol_r + 6e-8, \
        f"{where}: WRONG Metropolis decision: u = {float(u):.9g}, exp(r) = {np.exp(min(r, 50.0)):.9g} (r = {r:.6g}), " \
        f"allowed |u - exp(r)| <= {thr * tol_r + 6e-8:.3g}"
    return gap / max(thr * tol_r + 6e-8, 1e-300)


def _prove_edge_flip(spec, x_t, y_t, l_x, l_y, rtol, where):
    """The other legitimate kind of Metropolis flip, possible only where the proposals are compared with a tolerance
    (Laplace / UniformRadius, or any proposal in Philox mode; never with exact_states): the proposal lands within that
    tolerance of the EDGE of the target's support (IIDGamma x > 0, IIDBeta 0 < x < 1, Hypercube), so one engine's
    proposal is inside (finite log-density) and the other's outside (-inf, rejected).  Proof: moving the oracle's
    proposal by delta = rtol max(|x|, |y|) per coordinate, towards either side, changes whether its log-density is
    finite.  A proposal that is clearly inside or clearly outside the support proves nothing and fails."""
    # (a current state outside the support, l = -inf, is allowed: the rule then accepts exactly the proposals with a
    # finite log-density - r = +inf - so the decision still turns on which side of the edge the proposal falls)
    assert np.isfinite(l_x) or l_x == -np.inf, f"{where}: the current state has log-density {l_x}"
    x64, y64 = np.asarray(x_t, np.float64), np.asarray(y_t, np.float64)
    delta = rtol * np.maximum(np.abs(x64), np.abs(y64))
    near = O.logdensity(spec.oracle(), np.stack([y64 + delta, y64 - delta]).astype(f32), "f64")
    assert bool(np.any(np.isfinite(near) != np.isfinite(l_y))), \
        f"{where}: WRONG Metropolis decision: the proposal is not within {rtol:g} (relative) of the edge of the support " \
        f"(l' = {l_y}, at y +- delta: {near.tolist()})"
    return 0.0


def _prove_swap_flip(lm, beta, us, swap_mode, swap_order, ev_number, slack, where, in_band=None):
    """Replays one swap event along the oracle's path in fp64 and asserts that at least one attempted pair has its
    uniform inside the tolerance band of its threshold (so a different outcome is an fp32-level flip).
    in_band: a set - the pairs whose uniform is inside the band are added to it and nothing is asserted."""
    T = len(lm)
    lm = np.array(lm, dtype=np.float64)
    b = np.asarray(beta, dtype=np.float64)
    seq = swap_order == O.ORDER_SEQUENTIAL
    best = np.inf
    for j in range(0 if seq else int(ev_number & 1), T - 1, 1 if seq else 2):
        k = j + 1
        with np.errstate(invalid="ignore", over="ignore"):
            lpr = (b[j] - b[k]) * (lm[k] - lm[j])
            thr = 1.0 if lpr >= 0 else float(np.exp(lpr))
            mag = abs(b[j] * lm[k]) + abs(b[k] * lm[j]) + abs(b[j] * lm[j]) + abs(b[k] * lm[k])
        accepted = (us[j] < thr) if np.isfinite(lpr) or lpr == -np.inf else False
        if np.isfinite(lpr) and np.isfinite(mag):
            tol = (abs(b[j] - b[k]) * float(logp_tol(lm[j]) + logp_tol(lm[k])) + 2.0 ** -22 * mag) * slack
            if lpr < 0 or lpr <= tol:
                bands = abs(float(us[j]) - thr) / (thr * tol + 6e-8)
                best = min(best, bands)
                if in_band is not None and bands <= 1.0:
                    in_band.add(j)
        if accepted:
            if swap_mode == O.SWAP_EXCHANGE:
                lm[j], lm[k] = lm[k], lm[j]
            else:
                lm[j] = lm[k]
    if in_band is None:
        assert best <= 1.0, f"{where}: WRONG swap outcome: no attempted pair has its uniform within the fp32 band of " \
                            f"its threshold (closest is {best:.3g} bands away)"
    return best


def check_parity(run_a, run_b, spec, prop, *, state, logp, beta, n_steps, burn_in, swap_every, swap_mode=O.SWAP_EXCHANGE,
                 swap_order=O.ORDER_SEQUENTIAL, ext_prop, ext_u, ext_swap_u=None, exact_states, step0=0, slack=1.0,
                 state_rtol=1e-4, state_atol=2e-5, max_flip_rate=1e-3, chain_offset=0, _depth=0, _flips=None):
    """run_a / run_b: callables(**kw) -> dict with trace [n, C, T, D], trace_logp [n, C, T], accept_flags [n, C, T],
    n_accept, sq_jump, swap_accept, last_swap_ordinal (engine A = the one under test, B = the oracle).
    chain_offset: global id of ladder 0 (the Philox subsequence): passed to both engines and advanced when a single
    ladder is restarted, so an engine that draws its randoms from (seed, step, chain) stays on its stream.
    Returns the list of proven flips [(global step, ladder, kind, margin in tolerance bands)]."""
    flips = [] if _flips is None else _flips
    state = np.ascontiguousarray(state, dtype=np.float64 if np.asarray(state).dtype == np.float64 else f32)  # f64: state_f64 mode
    Cn, T, D = state.shape
    logp = np.ascontiguousarray(logp, dtype=f32).reshape(Cn, T)
    beta = np.asarray(beta, dtype=f32)
    if not exact_states:
        slack = slack * 2.0  # the increments themselves differ by <= 2 ulp between the engines
    kw = dict(state=state, logp=logp, beta=beta, step0=step0, n_steps=n_steps, burn_in=burn_in, swap_every=swap_every,
              swap_mode=swap_mode, swap_order=swap_order, ext_prop=ext_prop, ext_u=ext_u,
              ext_swap_u=ext_swap_u if T > 1 else None, chain_offset=chain_offset)
    got, want = run_a(**kw), run_b(**kw)
    ev0 = events_upto(step0, swap_every, burn_in)
    for c in range(Cn):
        fl = np.any(got["accept_flags"][:, c] != want["accept_flags"][:, c], axis=1)
        g, w = got["trace"][:, c].reshape(n_steps, -1), want["trace"][:, c].reshape(n_steps, -1)
        if exact_states:
            same = np.all((g == w) | (np.isnan(g) & np.isnan(w)), axis=1)
        else:
            same = np.all(np.isclose(g, w, rtol=state_rtol, atol=state_atol, equal_nan=True), axis=1)
        bad = np.nonzero(fl | ~same)[0]
        if bad.size == 0:
            # the whole segment agrees: integer bookkeeping must be identical, the jump sums equal to fp32 rounding
            assert np.array_equal(got["n_accept"][c], want["n_accept"][c]), \
                f"n_accept of ladder {c} differs although every decision agrees"
            diff = np.nonzero((got["swap_accept"][c] != want["swap_accept"][c]) |
                              (got["last_swap_ordinal"][c] != want["last_swap_ordinal"][c]))[0]
            if diff.size:
                # A swap decision can flip WITHOUT a visible trace: the reference's row copy (and any swap between two
                # replicas that hold the same state, e.g. before either has moved) exchanges identical rows, so only the
                # counters show it.  Legitimate only if, in some swap event of this segment, the uniform of exactly that
                # pair sits inside the fp32 band of its threshold (evaluated on the oracle's states): prove it.
                in_band = set()
                for d in range(n_steps):
                    sc = step0 + d + 1
                    if not (T > 1 and sc > burn_in and sc % swap_every == 0):
                        continue
                    pre_x = want["trace"][d - 1, c] if d > 0 else state[c]
                    _, l_x, l_y, _ = step_log_ratios(spec, prop, pre_x, ext_prop[d, c], beta)
                    lm = np.where(want["accept_flags"][d, c].astype(bool), l_y, l_x)
                    ev_rel = events_upto(sc, swap_every, burn_in) - 1 - ev0
                    _prove_swap_flip(lm, beta, ext_swap_u[ev_rel, c], swap_mode, swap_order, ev0 + ev_rel, slack, "",
                                     in_band=in_band)
                assert set(diff.tolist()) <= in_band, \
                    f"swap bookkeeping of ladder {c} differs at pairs {diff.tolist()} although every decision agrees and " \
                    f"no swap event has those pairs' uniforms inside the fp32 band of their thresholds (in band: {sorted(in_band)})"
                assert np.abs(got["swap_accept"][c] - want["swap_accept"][c]).max() <= 2
                flips.append((step0, c, "swap-invisible", 0.0))
            if exact_states:
                np.testing.assert_allclose(got["sq_jump"][c], want["sq_jump"][c], rtol=1e-4, atol=1e-9)
            else:
                # the two engines' states agree to the state tolerance only, and a squared jump amplifies that by
                # |x| / |increment| (a move of 1e-2 from x ~ 1: an ulp of x is 1e-5 of the move).  Bound the difference
                # by what the two traces themselves allow: | |a|^2 - |b|^2 | <= 2 |b| |a - b| + |a - b|^2 per step.
                pg = np.concatenate([state[c][None], got["trace"][:, c]]).astype(np.float64)
                pw = np.concatenate([state[c][None], want["trace"][:, c]]).astype(np.float64)
                dg, dw = np.diff(pg, axis=0), np.diff(pw, axis=0)
                gap2 = ((dg - dw) ** 2).sum(-1)
                room = (2.0 * np.sqrt((dw ** 2).sum(-1) * gap2) + gap2).sum(0)  # [T]
                err = np.abs(got["sq_jump"][c] - want["sq_jump"][c])
                assert np.all(err <= 1e-4 * np.abs(want["sq_jump"][c]) + 1e-9 + 1.01 * room), \
                    f"sq_jump of ladder {c} differs beyond what the two traces allow: {err.max():.3g}"
            continue
        d = int(bad[0])
        s_glob = step0 + d
        where = f"ladder {c}, global step {s_glob}"
        pre_x = want["trace"][d - 1, c] if d > 0 else state[c]
        r, l_x, l_y, y = step_log_ratios(spec, prop, pre_x, ext_prop[d, c], beta)
        sc = s_glob + 1
        swap_due = T > 1 and sc > burn_in and sc % swap_every == 0
        if fl[d]:
            for t in np.nonzero(got["accept_flags"][d, c] != want["accept_flags"][d, c])[0]:
                wt = f"{where}, temperature {t}"
                try:
                    m = _prove_mh_flip(float(r[t]), float(l_x[t]), float(l_y[t]), beta[t], ext_u[d, c, t], slack, wt)
                    kind = "mh"
                except AssertionError as band_error:
                    if exact_states:  # proposals are bit-identical: the edge of the support cannot separate them
                        raise
                    try:
                        m = _prove_edge_flip(spec, pre_x[t], y[t], float(l_x[t]), float(l_y[t]), state_rtol, wt)
                    except AssertionError as edge_error:
                        raise AssertionError(f"{band_error}; and {edge_error}") from None
                    kind = "mh-edge"
                flips.append((s_glob, c, kind, m))
        else:
            assert swap_due, f"{where}: states differ although every decision agrees and no swap is due"
            acc = want["accept_flags"][d, c].astype(bool)
            lm = np.where(acc, l_y, l_x)
            ev_rel = events_upto(sc, swap_every, burn_in) - 1 - ev0
            m = _prove_swap_flip(lm, beta, ext_swap_u[ev_rel, c], swap_mode, swap_order,
                                 ev0 + ev_rel, slack, where)
            flips.append((s_glob, c, "swap", m))
        # resynchronise from the oracle's state of step d and compare the rest of the horizon
        if d + 1 < n_steps:
            assert _depth < 12, f"{where}: more than 12 decision flips in one ladder"
            ev_next = events_upto(sc, swap_every, burn_in) - ev0
            check_parity(run_a, run_b, spec, prop, state=want["trace"][d, c][None], logp=want["trace_logp"][d, c][None],
                         beta=beta, n_steps=n_steps - d - 1, burn_in=burn_in, swap_every=swap_every, swap_mode=swap_mode,
                         swap_order=swap_order, ext_prop=np.ascontiguousarray(ext_prop[d + 1:, c:c + 1]),
                         ext_u=np.ascontiguousarray(ext_u[d + 1:, c:c + 1]),
                         ext_swap_u=None if ext_swap_u is None else np.ascontiguousarray(ext_swap_u[ev_next:, c:c + 1]),
                         exact_states=exact_states, step0=sc, slack=slack / (1.0 if exact_states else 2.0),
                         state_rtol=state_rtol, state_atol=state_atol, max_flip_rate=max_flip_rate,
                         chain_offset=chain_offset + c, _depth=_depth + 1, _flips=flips)
    if _depth == 0:
        budget = 3 + max_flip_rate * n_steps * Cn * T
        assert len(flips) <= budget, f"{len(flips)} decision flips in {n_steps * Cn * T} decisions: too many for fp32 " \
                                     f"rounding (budget {budget:.1f})"
    return flips


def oracle_runner(spec, prop, precision="f32"):
    """run_b for check_parity: the C oracle with per-step trace and accept flags."""
    def run(**kw):
        Cn, T = kw["state"].shape[:2]
        return O.run(spec.oracle(), prop.oracle(), trace_chains=Cn, trace_temps=T, want_flags=True, precision=precision, **kw)
    return run


# ---------------------------------------------------------------------------------------------------------
# The HIP engine as a runner (GPU tests, tools/, __graft_entry__.smoke)
# ---------------------------------------------------------------------------------------------------------
def dev_t(a, device, dtype=None):
    import torch

    dtype = torch.float32 if dtype is None else dtype
    return torch.tensor(np.ascontiguousarray(a), device=device, dtype=dtype)


def gpu_run(spec, prop, device, *, state, logp, beta, n_steps, trace_temps=0, want_flags=False, ext_prop=None,
            ext_u=None, ext_swap_u=None, **kw):
    """Mirror of oracle.run for the HIP engine, through its C ABI (ptrwm_hip.run): returns numpy results."""
    import torch

    import ptrwm_hip as E

    Cn, T, D = state.shape
    sdt = torch.float64 if np.asarray(state).dtype == np.float64 else torch.float32  # float64: the engine's state_f64 mode
    st, lp = dev_t(state, device, sdt), dev_t(logp, device).reshape(Cn, T).contiguous()
    res = {
        "n_accept": torch.zeros(Cn, T, dtype=torch.int64, device=device),
        "sq_jump": torch.zeros(Cn, T, dtype=torch.float64, device=device),
        "swap_accept": torch.zeros(Cn, T, dtype=torch.int64, device=device),
        "last_swap_ordinal": torch.zeros(Cn, T, dtype=torch.int64, device=device),
    }
    trace = trace_logp = flags = None
    if trace_temps:
        trace = torch.zeros(n_steps, Cn, trace_temps, D, device=device, dtype=sdt)
        trace_logp = torch.zeros(n_steps, Cn, trace_temps, device=device)
    if want_flags:
        flags = torch.zeros(n_steps, Cn, T, dtype=torch.uint8, device=device)
    E.run(spec.engine(device), prop.engine(device), state=st, logp=lp, beta=dev_t(beta, device), n_steps=n_steps,
          n_accept=res["n_accept"], sq_jump=res["sq_jump"], swap_accept=res["swap_accept"],
          last_swap_ordinal=res["last_swap_ordinal"], trace=trace, trace_logp=trace_logp, accept_flags=flags,
          ext_prop=None if ext_prop is None else dev_t(ext_prop, device, sdt),
          ext_u=None if ext_u is None else dev_t(ext_u, device),
          ext_swap_u=None if ext_swap_u is None else dev_t(ext_swap_u, device), **kw)
    torch.cuda.synchronize()
    out = {k: v.cpu().numpy() for k, v in res.items()}
    out["state"], out["logp"] = st.cpu().numpy(), lp.cpu().numpy()
    if trace is not None:
        out["trace"], out["trace_logp"] = trace.cpu().numpy(), trace_logp.cpu().numpy()
    if flags is not None:
        out["accept_flags"] = flags.cpu().numpy()
    return out


def gpu_runner(spec, prop, device):
    """run_a for helpers.check_parity: the HIP engine through the C ABI, per-step trace and accept flags on."""
    def run(**kw):
        return gpu_run(spec, prop, device, trace_temps=kw["state"].shape[1], want_flags=True, **kw)
    return run


# ---------------------------------------------------------------------------------------------------------
# The same comparison for the PRODUCTION path: the kernel draws its randoms itself (Philox, in-kernel Box-Muller with the
# hardware sin / cos, the per-temperature scale folded into the radius), the oracle restates the same counter layout
# and exports the numbers it drew (oracle_philox_randoms), so every differing decision can be proven exactly as above.
# ---------------------------------------------------------------------------------------------------------
def philox_runner(run, seed):
    """run_a for check_parity from an engine runner: the external-randoms arrays are dropped, the engine draws from
    Philox(seed, step, chain_offset + chain) itself."""
    def run_a(**kw):
        kw = {k: v for k, v in kw.items() if k not in ("ext_prop", "ext_u", "ext_swap_u")}
        return run(seed=seed, **kw)
    return run_a


def check_parity_philox(run, spec, prop, *, state, logp, beta, n_steps, burn_in, swap_every, seed, chain_offset=0, step0=0,
                        swap_mode=O.SWAP_EXCHANGE, swap_order=O.ORDER_SEQUENTIAL, segment=None, **tol):
    """Full-horizon parity of an engine in Philox mode (`run(seed=, chain_offset=, step0=, ...)`) with the oracle on the
    same stream: every differing Metropolis / swap decision PROVEN (check_parity), bookkeeping identical on agreeing
    segments, states equal to the stated tolerance at every step.  The two engines agree on the proposals only to a few
    ulp (hardware sin / cos / log against libm), so states are compared with a tolerance (`exact_states=False`) and, if
    `segment` is given, both engines restart from the oracle's own trajectory every `segment` steps so that rounding
    drift cannot build up over a long horizon.  Returns the proven flips."""
    f64 = np.asarray(state).dtype == np.float64  # the engine's state_f64 mode
    state = np.ascontiguousarray(state, dtype=np.float64 if f64 else f32)
    Cn, T, D = state.shape
    logp = np.ascontiguousarray(logp, dtype=f32).reshape(Cn, T)
    ext_prop, ext_u, ext_swap_u = O.philox_randoms(prop.kind, D, T, Cn, seed=seed, step0=step0, n_steps=n_steps,
                                                   burn_in=burn_in, swap_every=swap_every, chain_offset=chain_offset)
    if f64:
        # the kernel adds its float increment to the double state; the oracle's double path multiplies the same float
        # normal by the float scale in double: equal to ~1e-8 of the increment, compared with the usual state tolerance
        assert prop.kind == O.PROPOSAL_NORMAL, "state_f64 comparisons against the oracle: Normal proposal"
        ext_prop = ext_prop.astype(np.float64)
    run_a, run_b = philox_runner(run, seed), oracle_runner(spec, prop)
    common = dict(beta=beta, burn_in=burn_in, swap_every=swap_every, swap_mode=swap_mode, swap_order=swap_order,
                  chain_offset=chain_offset, exact_states=False, **tol)
    if segment is None or segment >= n_steps:
        return check_parity(run_a, run_b, spec, prop, state=state, logp=logp, n_steps=n_steps, step0=step0,
                            ext_prop=ext_prop, ext_u=ext_u, ext_swap_u=ext_swap_u, **common)
    path = run_b(state=state, logp=logp, beta=np.asarray(beta, f32), step0=step0, n_steps=n_steps, burn_in=burn_in,
                 swap_every=swap_every, swap_mode=swap_mode, swap_order=swap_order, ext_prop=ext_prop, ext_u=ext_u,
                 ext_swap_u=ext_swap_u, chain_offset=chain_offset)  # the oracle's own trajectory
    flips = []
    ev0 = events_upto(step0, swap_every, burn_in)
    for a in range(0, n_steps, segment):
        n = min(segment, n_steps - a)
        e0 = events_upto(step0 + a, swap_every, burn_in) - ev0
        flips += check_parity(run_a, run_b, spec, prop, state=state if a == 0 else path["trace"][a - 1],
                              logp=logp if a == 0 else path["trace_logp"][a - 1], n_steps=n, step0=step0 + a,
                              ext_prop=np.ascontiguousarray(ext_prop[a:a + n]), ext_u=np.ascontiguousarray(ext_u[a:a + n]),
                              ext_swap_u=None if ext_swap_u is None else np.ascontiguousarray(ext_swap_u[e0:]), **common)
    return flips


def spec_from_engine(target) -> TargetSpec:
    """The TargetSpec of a ptrwm_hip.Target (device tensors copied to the host): the oracle then sees exactly the parameter
    bits the kernel was given (a sampler class may fold its parameters with torch ops that differ from numpy's by an ulp)."""
    host = lambda t: None if t is None else t.detach().cpu().numpy().astype(f32)  # noqa: E731
    return TargetSpec(target.kind, target.dim, tuple(float(v) for v in target.p), tuple(int(v) for v in target.ip),
                      host(target.vec0), host(target.vec1))


def prop_from_engine(proposal) -> ProposalSpec:
    host = lambda t: None if t is None else t.detach().cpu().numpy().astype(f32)  # noqa: E731
    return ProposalSpec(proposal.kind, host(proposal.temp_scale), host(proposal.dim_scale), float(proposal.inv_dim))


def check_production_run(run, x0, n_cmp, segment=50):
    """The first `n_cmp` ladders of a finished PRODUCTION run of a sampler class (`run`: its algorithms._engine_core
    EngineRun - the non-fixture kernel variant, no per-step trace, whatever kernel form the batch size selected) tied to
    the oracle without any agreement-rate threshold, in two links:
      1. the fixture variant of the same kernel, run on the same Philox stream over just these ladders with a per-step
         trace and accept flags attached, reproduces the production run's final states, log-densities and all four
         statistics BIT FOR BIT;
      2. that traced run follows the oracle decision for decision over the full horizon, every differing decision
         proven (check_parity_philox).
    Target and proposal parameters are read back from the run itself; x0 [dim]: the common starting point (every replica
    and temperature starts there); the starting log-densities are the engine's own (ptrwm_logdensity, the call the classes
    make), so both runs start from the same bits."""
    import ptrwm_hip as E

    assert run.density_fn is None and run.manual_sweeps == 0
    device = run.device
    spec, prop = spec_from_engine(run.target), prop_from_engine(run.proposal)
    T, D = run.n_temps, run.dim
    import torch

    sdt = np.float64 if getattr(run, "dtype", torch.float32) == torch.float64 else f32
    state = np.broadcast_to(np.asarray(x0, sdt), (n_cmp, T, D)).copy()
    logp = E.logdensity(run.target, dev_t(state.reshape(-1, D), device)).cpu().numpy().reshape(n_cmp, T)
    kw = dict(state=state, logp=logp, beta=run.beta.cpu().numpy(), n_steps=run.steps_done, burn_in=run.burn_in,
              swap_every=run.swap_every, swap_mode=run.swap_mode, swap_order=run.swap_order, chain_offset=run.chain_offset)
    full = gpu_runner(spec, prop, device)(step0=0, seed=run.seed, **kw)
    produced = {"state": run.state, "logp": run.logp, "n_accept": run.n_accept, "sq_jump": run.sq_jump,
                "swap_accept": run.swap_accept, "last_swap_ordinal": run.last_ord}
    for k, v in produced.items():
        v = v[:n_cmp].cpu().numpy()
        assert v.shape == full[k].shape and np.array_equal(v.view(np.uint8), np.ascontiguousarray(full[k]).view(np.uint8)), \
            f"production run and its traced fixture twin differ in `{k}`"
    return check_parity_philox(gpu_runner(spec, prop, device), spec, prop, seed=run.seed, segment=segment, **kw)
