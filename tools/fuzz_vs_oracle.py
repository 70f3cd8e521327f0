"""Randomised differential test of the HIP engine against the CPU oracle (development aid, needs a GPU):

    python tools/fuzz_vs_oracle.py [n_cases] [seed]

Each case draws a target family with random parameters, a dimension in 1..104, a ladder of 1..256 temperatures, a
proposal, swap mode / order / period, burn-in and a chain offset, runs both engines on the same external randoms - and
then once more in Philox mode, the production arithmetic (helpers.check_parity_philox) - and
compares them over the FULL horizon with tests/helpers.check_parity: per-step traces identical (Normal: bit for bit),
every differing Metropolis or swap decision PROVEN to sit inside the fp32 band of its threshold, both engines then
restarted from the oracle's state; all four statistics exact on every agreeing segment."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rwm-pt-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import helpers as H  # noqa: E402
import ptrwm_hip as E  # noqa: E402
from oracle import oracle as O  # noqa: E402

f32 = np.float32


def random_target(rng, dim):
    fam = rng.choice(["rc", "rc_scaled", "tm", "tm_first", "full", "even", "hyb", "gamma", "beta", "mvn", "smvn", "cube", "funnel"])
    if fam in ("rc", "rc_scaled"):
        m = np.sort(rng.uniform(-12, 12, 3)).astype(f32)
        w = rng.dirichlet([2, 2, 2]).astype(f32)
        params = {"modes": m, "weights": w}
        if fam == "rc_scaled":
            params["scaling_factors"] = rng.uniform(0.5, 1.5, dim).astype(f32)
        return H.spec_from_params("RoughCarpetDistributionTorch", dim, params), np.zeros(dim)
    if fam in ("tm", "tm_first"):
        means = rng.normal(0, 3, (3, dim)).astype(f32)
        if fam == "tm_first":  # the three means differ in the first coordinate only: the ThreeMixture1 kernels (ip[0] = 1)
            means[1:, 1:] = means[0, 1:]
        if rng.random() < 0.3:
            return H.spec_from_params("ThreeMixtureDistributionTorch", dim,
                                      {"means": means, "mixing_weights": rng.dirichlet([2, 2, 2]).astype(f32),
                                       "scaling_factors": rng.uniform(0.5, 1.5, dim).astype(f32)}), np.zeros(dim)
        return H.spec_from_params("ThreeMixtureDistributionTorch", dim,
                                  {"means": means, "mixing_weights": rng.dirichlet([2, 2, 2]).astype(f32)}), np.zeros(dim)
    if fam in ("full", "even"):
        if dim < 2:
            dim = 2
        if fam == "even" and dim % 2:
            dim += 1 if dim < 104 else -1
        n_mu = dim - 1 if fam == "full" else dim // 2
        cls = "FullRosenbrockTorch" if fam == "full" else "EvenRosenbrockTorch"
        return H.spec_from_params(cls, dim, {"a_coeff": f32(rng.uniform(0.02, 0.2)), "b_coeff": f32(rng.uniform(1, 6)),
                                            "mu": rng.uniform(0.5, 1.5, n_mu).astype(f32)}), 1e-3 * rng.standard_normal(dim)
    if fam == "hyb":
        n1, n2 = int(rng.integers(2, 6)), int(rng.integers(1, 8))
        dim = 1 + n2 * (n1 - 1)
        return H.spec_from_params("HybridRosenbrockTorch", dim, {"a_coeff": f32(0.05), "b_coeff": f32(5.0), "mu": f32(1.0),
                                                                "n1": n1, "n2": n2}), 1e-3 * rng.standard_normal(dim)
    if fam == "gamma":
        return H.spec_from_params("IIDGammaTorch", dim, {"shape": f32(rng.uniform(1.5, 4)), "scale": f32(rng.uniform(0.5, 3))}), \
            5 + 0.01 * rng.standard_normal(dim)
    if fam == "beta":
        return H.spec_from_params("IIDBetaTorch", dim, {"alpha": f32(rng.uniform(1.2, 4)), "beta": f32(rng.uniform(1.2, 4))}), \
            rng.uniform(0.3, 0.7, dim)
    if fam == "mvn":
        var = rng.uniform(0.3, 3, dim)
        lnc = -0.5 * (dim * np.log(2 * np.pi) + np.log(var).sum())
        return H.spec_from_params("MultivariateNormalTorch", dim, {"cov": np.diag(var), "mean": rng.normal(0, 1, dim).astype(f32),
                                                                  "log_norm_const": f32(lnc)}), np.zeros(dim)
    if fam == "smvn":
        c = rng.uniform(0.3, 1.8, dim).astype(f32)
        return H.spec_from_params("ScaledMultivariateNormalTorch", dim,
                                  {"scaling_factors": c, "log_norm_const": f32(np.log(c).sum() - 0.5 * dim * np.log(2 * np.pi))}), \
            np.zeros(dim)
    if fam == "cube":
        lo, hi = f32(rng.uniform(-2, 0)), f32(rng.uniform(0.5, 2))
        return H.spec_from_params("HypercubeTorch", dim, {"left_boundary": lo, "right_boundary": hi,
                                                         "log_uniform_density": f32(-dim * np.log(hi - lo))}), \
            np.full(dim, 0.5 * (lo + hi))
    return H.spec_from_params("NealFunnelTorch", dim, {"mu_v": f32(0.0), "sigma_v_sq": f32(rng.uniform(1, 9)), "mu_z": f32(0.0)}), \
        0.1 * rng.standard_normal(dim)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    first_case = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # replay: skip the engine for earlier cases
    dev = torch.device("cuda:0")
    flips = 0
    for case in range(n_cases):
        dim = int(rng.choice([1, 2, 3, 5, 7, 10, 13, 20, 24, 30, 33, 41, 50, 57, 64, 77, 100, 104]))
        spec, x0 = random_target(rng, dim)
        dim = spec.dim
        T = int(rng.choice([1, 2, 3, 5, 8, 21, 32, 33, 64, 65, 100, 130, 200, 256]))
        if dim > 64 and T > 128:
            T = 128  # above dim 64 (lane-split kernel only, 512-thread workgroups) a ladder holds at most 128 temperatures
        Cn = int(rng.integers(1, 6)) if T > 64 else int(rng.integers(1, 40))
        beta = (0.03 ** (np.arange(T) / max(1, T - 1))).astype(f32)
        pk = str(rng.choice(["Normal", "Laplace", "UniformRadius"]))
        scale = float(rng.uniform(0.2, 1.5)) * 2.38**2 / dim * (0.05 if "Rosenbrock" in spec.cls or "Beta" in spec.cls else 1.0)
        if pk == "Normal":
            prop = H.proposal_spec(pk, dim, beta, base_variance_scalar=scale)
        elif pk == "Laplace":
            prop = H.proposal_spec(pk, dim, beta, base_variance_vector=np.full(dim, scale, f32))
        else:
            prop = H.proposal_spec(pk, dim, beta, base_radius=float(np.sqrt(scale * dim)))
        N, se, burn = int(rng.integers(6, 30)), int(rng.integers(1, 6)), int(rng.integers(0, 6))
        order, mode = str(rng.choice(["sequential", "even_odd"])), str(rng.choice(["exchange", "reference_copy"]))
        st = np.broadcast_to(x0.astype(f32), (Cn, T, dim)).copy()
        lp = np.broadcast_to(O.logdensity(spec.oracle(), x0[None].astype(f32)).astype(f32), (Cn, T)).copy()
        raw = E.ext_raw_per_step(prop.kind, dim)
        ep = rng.standard_normal((N, Cn, T, raw)).astype(f32)
        if pk == "Laplace":
            ep = rng.random((N, Cn, T, raw)).astype(f32)
        elif pk == "UniformRadius":
            ep[..., -1] = rng.random((N, Cn, T)).astype(f32)
        n_ev = max(0, N // se - burn // se)
        kw = dict(beta=beta, step0=0, burn_in=burn, swap_every=se, swap_order=E.SWAP_ORDERS[order],
                  swap_mode=E.SWAP_MODES[mode], chain_offset=int(rng.integers(0, 1000)))
        ext = dict(ext_prop=ep, ext_u=rng.random((N, Cn, T)).astype(f32),
                   ext_swap_u=rng.random((max(n_ev, 1), Cn, max(T - 1, 1))).astype(f32)[:n_ev, :, :T - 1] if T > 1 else None)
        if case < first_case:
            continue
        print(f"case {case}: {spec.cls} dim {dim} T {T} C {Cn} {pk} {order} {mode} N {N} se {se} burn {burn}", flush=True)
        dt = lambda a, d=torch.float32: torch.tensor(np.ascontiguousarray(a), device=dev, dtype=d)  # noqa: E731

        def engine(**k):
            """the engine through the C ABI with per-step trace and accept flags (run_a of check_parity)"""
            c, t = k["state"].shape[:2]
            n = k["n_steps"]
            s_d, l_d = dt(k["state"]), dt(k["logp"])
            out = {a: torch.zeros(c, t, dtype=(torch.float64 if a == "sq_jump" else torch.int64), device=dev)
                   for a in ("n_accept", "sq_jump", "swap_accept", "last_swap_ordinal")}
            trace = torch.zeros(n, c, t, dim, device=dev)
            tlp = torch.zeros(n, c, t, device=dev)
            flags = torch.zeros(n, c, t, dtype=torch.uint8, device=dev)
            es = k["ext_swap_u"]
            E.run(spec.engine(dev), prop.engine(dev), state=s_d, logp=l_d, beta=dt(k["beta"]), step0=k["step0"], n_steps=n,
                  burn_in=k["burn_in"], swap_every=k["swap_every"], swap_mode=k["swap_mode"], swap_order=k["swap_order"],
                  chain_offset=kw["chain_offset"], trace=trace, trace_logp=tlp, accept_flags=flags,
                  ext_prop=dt(k["ext_prop"]), ext_u=dt(k["ext_u"]),
                  ext_swap_u=None if es is None or es.shape[0] == 0 else dt(es), **out)
            torch.cuda.synchronize()
            res = {a: b.cpu().numpy() for a, b in out.items()}
            res.update(trace=trace.cpu().numpy(), trace_logp=tlp.cpu().numpy(), accept_flags=flags.cpu().numpy())
            return res

        # the FULL horizon: every differing decision must be a proven fp32-level flip, after which both engines restart
        # from the oracle's state (tests/helpers.check_parity); bookkeeping exact on every agreeing segment
        es = ext["ext_swap_u"]
        if T > 1 and es is not None and es.shape[0] == 0:
            es = np.zeros((1, Cn, T - 1), f32)
        try:
            fl = H.check_parity(engine, H.oracle_runner(spec, prop), spec, prop, state=st, logp=lp, beta=beta, n_steps=N,
                                burn_in=burn, swap_every=se, swap_mode=kw["swap_mode"], swap_order=kw["swap_order"],
                                ext_prop=ep, ext_u=ext["ext_u"], ext_swap_u=es, exact_states=pk == "Normal",
                                state_rtol=2e-5, state_atol=2e-5)
        except AssertionError as e:
            print(f"case {case:3d}: MISMATCH {e}")
            sys.exit(1)
        # the same case on the PRODUCTION arithmetic: the kernel draws from Philox itself (hardware Box-Muller, scale folded
        # into the radius, squared jumps from the proposal), the oracle restates the stream and exports what it drew
        try:
            fl += H.check_parity_philox(H.gpu_runner(spec, prop, dev), spec, prop, state=st, logp=lp, beta=beta, n_steps=N,
                                        burn_in=burn, swap_every=se, seed=int(rng.integers(1, 1 << 40)),
                                        chain_offset=kw["chain_offset"], step0=int(rng.integers(0, 50)),
                                        swap_mode=kw["swap_mode"], swap_order=kw["swap_order"], segment=30)
        except AssertionError as e:
            print(f"case {case:3d}: MISMATCH (Philox mode) {e}")
            sys.exit(1)
        flips += len(fl)
        tag = f"case {case:3d}: {spec.cls:30s} dim {dim:3d} T {T:3d} C {Cn:2d} {pk:13s} {order:10s} {mode:14s} N {N:2d} se {se} burn {burn}"
        print(tag, "ok", "" if not fl else f"({len(fl)} proven flip(s): {[(f[0], f[2]) for f in fl]})")
    print(f"{n_cases} cases agree; {flips} proven fp32-level decision flip(s) (MH or swap), every case compared to its end")
    assert flips <= max(3, n_cases // 4), "too many decision flips for fp32-level differences"


if __name__ == "__main__":
    main()
