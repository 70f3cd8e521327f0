"""Every compiled kernel variant against the oracle (needs a GPU):  target kind x proposal x register width x
{fixture (external randoms, traced), production (Philox)}.

    python tools/check_all_variants.py            # all (target, proposal) pairs, one child process each
    python tools/check_all_variants.py 0 2        # one pair in this process (target kind 0, proposal kind 2)

Fixture kernels: the full horizon against the oracle on shared external randoms, for a narrow ladder (T = 3, one
wavefront) and a ladder wider than a wavefront (T = 66: workgroup barriers, > 48 KB of dynamic LDS at the large widths);
every differing decision must be a PROVEN fp32-level flip (tests/helpers.check_parity), there are no retries.
Production kernels: Philox mode against the oracle's restated stream, and bit-identity with their fixture twin.

A miscompiled variant shows up as a mismatch or as a GPU fault of the child (this is how a scheduler-flag miscompile of
the width-80 UniformRadius fixture kernels was found)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rwm-pt-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
f32 = np.float32
WIDTH_DIMS = [2, 3, 4, 5, 10, 20, 30, 50, 100, 9, 19, 29, 7, 15, 23, 31, 39, 47, 55, 63, 79, 103]  # exact widths (9 / 19 / 29: compiled in for HybridRosenbrock alone, generic kernels for the others), then W-1 per generic


def make_spec(H, kind, dim, rng):
    if kind == 0:  # three-term kernels (modes too close for the two-term specialisation)
        return H.spec_from_params("RoughCarpetDistributionTorch", dim, {"modes": f32([-4, 0, 4]), "weights": f32([0.2, 0.5, 0.3])}), np.zeros(dim)
    if kind == 10:  # RoughCarpet2 kernels (well-separated modes)
        return H.spec_from_params("RoughCarpetDistributionTorch", dim, {"modes": f32([-15, 0, 15]), "weights": f32([0.5, 0.3, 0.2])}), np.zeros(dim)
    if kind == 11:  # ThreeMixture1 kernels (means differing in the first coordinate only: spec_from_params sets ip[0] = 1)
        means = np.tile(rng.normal(0, 2, (1, dim)).astype(f32), (3, 1))
        means[:, 0] = f32([-4.0, 0.5, 5.0])
        spec = H.spec_from_params("ThreeMixtureDistributionTorch", dim, {"means": means, "mixing_weights": f32([0.3, 0.3, 0.4])})
        assert spec.ip[0] == 1
        return spec, np.zeros(dim)
    if kind == 1:
        return H.spec_from_params("ThreeMixtureDistributionTorch", dim, {"means": rng.normal(0, 2, (3, dim)).astype(f32),
                                                                          "mixing_weights": f32([0.3, 0.3, 0.4])}), np.zeros(dim)
    if kind == 2:
        return H.spec_from_params("FullRosenbrockTorch", dim, {"a_coeff": f32(0.05), "b_coeff": f32(5), "mu": np.ones(dim - 1, f32)}), 1e-3 * rng.standard_normal(dim)
    if kind == 3:
        return H.spec_from_params("EvenRosenbrockTorch", dim, {"a_coeff": f32(0.05), "b_coeff": f32(5), "mu": np.ones(dim // 2, f32)}), 1e-3 * rng.standard_normal(dim)
    if kind == 4:
        return H.spec_from_params("HybridRosenbrockTorch", dim, {"a_coeff": f32(0.05), "b_coeff": f32(5), "mu": f32(1), "n1": 2, "n2": dim - 1}), 1e-3 * rng.standard_normal(dim)
    if kind == 5:
        return H.spec_from_params("IIDGammaTorch", dim, {"shape": f32(2), "scale": f32(3)}), 5 + 0.01 * rng.standard_normal(dim)
    if kind == 6:
        return H.spec_from_params("IIDBetaTorch", dim, {"alpha": f32(2), "beta": f32(3)}), rng.uniform(0.3, 0.7, dim)
    if kind == 7:
        var = rng.uniform(0.5, 2, dim)
        return H.spec_from_params("MultivariateNormalTorch", dim, {"cov": np.diag(var), "mean": np.zeros(dim, f32),
                                                                  "log_norm_const": f32(-0.5 * (dim * np.log(2 * np.pi) + np.log(var).sum()))}), np.zeros(dim)
    if kind == 8:
        return H.spec_from_params("HypercubeTorch", dim, {"left_boundary": f32(-1), "right_boundary": f32(1), "log_uniform_density": f32(-dim * np.log(2.0))}), np.zeros(dim)
    return H.spec_from_params("NealFunnelTorch", dim, {"mu_v": f32(0), "sigma_v_sq": f32(9), "mu_z": f32(0)}), 0.1 * rng.standard_normal(dim)


def one_pair(tk, pk):
    import torch

    import helpers as H
    import ptrwm_hip as E
    from oracle import oracle as O

    dev = torch.device("cuda:0")
    pname = ["Normal", "Laplace", "UniformRadius"][pk]
    T, Cn, N, se, burn = 3, 4, 6, 2, 1
    beta = f32([1.0, 0.4, 0.1])
    bad = 0
    for dim in WIDTH_DIMS:
        if tk == 2 and dim < 2:
            continue
        if tk == 3 and dim % 2:
            dim -= 1
        rng = np.random.default_rng(1000 * tk + 10 * pk + dim)
        spec, x0 = make_spec(H, tk, dim, rng)
        small = tk in (2, 3, 4, 6)
        sc = (0.02 if small else 1.0) * 2.38**2 / dim
        def make_prop(b):
            if pk == 0:
                return H.proposal_spec(pname, dim, b, base_variance_scalar=sc)
            if pk == 1:
                return H.proposal_spec(pname, dim, b, base_variance_vector=np.full(dim, sc, f32))
            return H.proposal_spec(pname, dim, b, base_radius=float(np.sqrt(sc * dim)))

        prop = make_prop(beta)
        st = np.broadcast_to(x0.astype(f32), (Cn, T, dim)).copy()
        lp = np.broadcast_to(O.logdensity(spec.oracle(), x0[None].astype(f32)).astype(f32), (Cn, T)).copy()
        raw = E.ext_raw_per_step(prop.kind, dim)
        dt = lambda a, d=torch.float32: torch.tensor(np.ascontiguousarray(a), device=dev, dtype=d)  # noqa: E731

        def ext_arrays(n, c, t):
            ep = rng.standard_normal((n, c, t, raw)).astype(f32)
            if pk == 1:
                ep = rng.random((n, c, t, raw)).astype(f32)
            elif pk == 2:
                ep[..., -1] = rng.random((n, c, t)).astype(f32)
            return ep, rng.random((n, c, t)).astype(f32), rng.random((max(1, n // se), c, t - 1)).astype(f32)

        cur = {"prop": prop}  # the proposal (one scale per temperature) of the ladder being checked

        def engine(**k):
            """the fixture variant through the C ABI: per-step trace and accept flags (run_a of check_parity)"""
            c, t = k["state"].shape[:2]
            n = k["n_steps"]
            s_d, l_d = dt(k["state"]), dt(k["logp"])
            out = {"n_accept": torch.zeros(c, t, dtype=torch.int64, device=dev),
                   "sq_jump": torch.zeros(c, t, dtype=torch.float64, device=dev),
                   "swap_accept": torch.zeros(c, t, dtype=torch.int64, device=dev),
                   "last_swap_ordinal": torch.zeros(c, t, dtype=torch.int64, device=dev)}
            trace = torch.zeros(n, c, t, dim, device=dev)
            tlp = torch.zeros(n, c, t, device=dev)
            flags = torch.zeros(n, c, t, dtype=torch.uint8, device=dev)
            E.run(spec.engine(dev), cur["prop"].engine(dev), state=s_d, logp=l_d, beta=dt(k["beta"]), step0=k["step0"], n_steps=n,
                  burn_in=k["burn_in"], swap_every=k["swap_every"], swap_mode=k["swap_mode"], swap_order=k["swap_order"],
                  trace=trace, trace_logp=tlp, accept_flags=flags, ext_prop=dt(k["ext_prop"]), ext_u=dt(k["ext_u"]),
                  ext_swap_u=None if k["ext_swap_u"] is None or k["ext_swap_u"].shape[0] == 0 else dt(k["ext_swap_u"]), **out)
            torch.cuda.synchronize()
            res = {a: b.cpu().numpy() for a, b in out.items()}
            res.update(trace=trace.cpu().numpy(), trace_logp=tlp.cpu().numpy(), accept_flags=flags.cpu().numpy())
            return res

        # fixture variant, narrow ladder (one wavefront) and a ladder wider than a wavefront (workgroup barriers, more than
        # 48 KB of dynamic LDS at the large widths): the FULL horizon against the oracle, every decision flip proven
        # (tests/helpers.check_parity) - no retries
        ok_full, first = True, None
        for (Tn, Cw, Nw) in ((T, Cn, N), (66, 2, 4)):
            bw = beta if Tn == T else (0.05 ** (np.arange(Tn) / (Tn - 1))).astype(f32)
            pw = prop if Tn == T else make_prop(bw)
            stw = np.broadcast_to(x0.astype(f32), (Cw, Tn, dim)).copy()
            lpw = np.broadcast_to(lp[0, 0], (Cw, Tn)).copy()
            ep, eu, es = ext_arrays(Nw, Cw, Tn)
            cur["prop"] = pw
            try:
                H.check_parity(engine, H.oracle_runner(spec, pw), spec, pw, state=stw, logp=lpw, beta=bw, n_steps=Nw,
                               burn_in=burn, swap_every=se, ext_prop=ep, ext_u=eu, ext_swap_u=es, exact_states=pk == 0,
                               state_rtol=3e-5, state_atol=3e-5)
            except AssertionError as e:
                ok_full, first = False, f"T={Tn}: {str(e)[:200]}"
        # production path (in-kernel Philox): the fixture variant on the Philox stream against the oracle's restatement of
        # it - full horizon, every differing decision proven (helpers.check_parity_philox) ...
        ok_prod, first_p = True, None
        try:
            H.check_parity_philox(H.gpu_runner(spec, prop, dev), spec, prop, state=st, logp=lp, beta=beta, step0=3, n_steps=40,
                                  burn_in=0, swap_every=se, seed=77 + dim, chain_offset=2, segment=20)
        except AssertionError as e:
            ok_prod, first_p = False, str(e)[:200]
        # ... the production variant's final log-densities belong to its final states ...
        s_d, l_d = dt(st), dt(lp)
        nacc = torch.zeros(Cn, T, dtype=torch.int64, device=dev)
        E.run(spec.engine(dev), prop.engine(dev), state=s_d, logp=l_d, beta=dt(beta), step0=3, n_steps=40, burn_in=0,
              swap_every=se, seed=77 + dim, chain_offset=2, n_accept=nacc)
        torch.cuda.synchronize()
        own = O.logdensity(spec.oracle(), s_d.cpu().numpy().reshape(-1, dim), "f64").reshape(Cn, T)
        got_lp = l_d.cpu().numpy()
        fin = np.isfinite(own)
        ok_prod &= np.array_equal(np.isfinite(got_lp), fin) and np.allclose(got_lp[fin], own[fin], rtol=2e-5, atol=2e-3)
        # ... and it is reproduced bit for bit by the fixture variant run on the same Philox stream with a trace attached
        # (the two are separate compilations of one template): production == fixture twin -> oracle, no tolerance on counts
        s_f, l_f = dt(st), dt(lp)
        nacc_f = torch.zeros(Cn, T, dtype=torch.int64, device=dev)
        tr = torch.zeros(40, Cn, T, dim, device=dev)
        E.run(spec.engine(dev), prop.engine(dev), state=s_f, logp=l_f, beta=dt(beta), step0=3, n_steps=40, burn_in=0,
              swap_every=se, seed=77 + dim, chain_offset=2, n_accept=nacc_f, trace=tr)
        torch.cuda.synchronize()
        ok_prod &= torch.equal(s_f, s_d) and torch.equal(l_f, l_d) and torch.equal(nacc_f, nacc) and torch.equal(tr[-1], s_d)
        if not (ok_full and ok_prod):
            bad += 1
            print(f"  MISMATCH target {tk} proposal {pname} dim {dim}: fixture ok={ok_full} ({first}) production ok={ok_prod} ({first_p})", flush=True)
    print(f"pair target {tk} proposal {pname}: {len(WIDTH_DIMS)} widths x (fixture narrow + wide ladder, production), {bad} bad", flush=True)
    return bad


if __name__ == "__main__":
    if len(sys.argv) == 3:
        sys.exit(1 if one_pair(int(sys.argv[1]), int(sys.argv[2])) else 0)
    failed = []
    for tk in range(12):
        for pk in range(3):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), str(tk), str(pk)], capture_output=True, text=True, timeout=600)
            tail = [l for l in r.stdout.splitlines() if l.strip()][-3:]
            print("\n".join(tail), flush=True)
            if r.returncode != 0:
                failed.append((tk, pk, r.returncode))
                err = [l for l in r.stderr.splitlines() if "fault" in l.lower() or "error" in l.lower()][:2]
                print(f"  child for target {tk} proposal {pk} exited {r.returncode} {err}", flush=True)
    print("FAILED pairs:", failed if failed else "none")
    sys.exit(1 if failed else 0)
