"""Every compiled kernel variant against the oracle (needs a GPU):  target kind x proposal x register width x
{fixture (external randoms, traced), production (Philox)}.

    python tools/check_all_variants.py            # all (target, proposal) pairs, one child process each
    python tools/check_all_variants.py 0 2        # one pair in this process (target kind 0, proposal kind 2)

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
WIDTH_DIMS = [2, 3, 4, 5, 10, 20, 30, 50, 100, 7, 15, 23, 31, 39, 47, 55, 63, 79, 103]  # exact widths, then W-1 per generic


def make_spec(H, kind, dim, rng):
    if kind == 0:  # three-term kernels (modes too close for the two-term specialisation)
        return H.spec_from_params("RoughCarpetDistributionTorch", dim, {"modes": f32([-4, 0, 4]), "weights": f32([0.2, 0.5, 0.3])}), np.zeros(dim)
    if kind == 10:  # RoughCarpet2 kernels (well-separated modes)
        return H.spec_from_params("RoughCarpetDistributionTorch", dim, {"modes": f32([-15, 0, 15]), "weights": f32([0.5, 0.3, 0.2])}), np.zeros(dim)
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
        if pk == 0:
            prop = H.proposal_spec(pname, dim, beta, base_variance_scalar=sc)
        elif pk == 1:
            prop = H.proposal_spec(pname, dim, beta, base_variance_vector=np.full(dim, sc, f32))
        else:
            prop = H.proposal_spec(pname, dim, beta, base_radius=float(np.sqrt(sc * dim)))
        st = np.broadcast_to(x0.astype(f32), (Cn, T, dim)).copy()
        lp = np.broadcast_to(O.logdensity(spec.oracle(), x0[None].astype(f32)).astype(f32), (Cn, T)).copy()
        raw = E.ext_raw_per_step(prop.kind, dim)
        kw = dict(beta=beta, step0=0, n_steps=N, burn_in=burn, swap_every=se)
        dt = lambda a, d=torch.float32: torch.tensor(np.ascontiguousarray(a), device=dev, dtype=d)  # noqa: E731
        ok_full, first = False, None
        for attempt in range(3):  # an fp32-level decision flip in the first two steps: try other randoms
            ep = rng.standard_normal((N, Cn, T, raw)).astype(f32)
            if pk == 1:
                ep = rng.random((N, Cn, T, raw)).astype(f32)
            elif pk == 2:
                ep[..., -1] = rng.random((N, Cn, T)).astype(f32)
            eu, es = rng.random((N, Cn, T)).astype(f32), rng.random((N // se, Cn, T - 1)).astype(f32)
            want = O.run(spec.oracle(), prop.oracle(), state=st, logp=lp, ext_prop=ep, ext_u=eu, ext_swap_u=es,
                         trace_chains=Cn, trace_temps=T, want_flags=True, **kw)
            s_d, l_d = dt(st), dt(lp)
            trace = torch.zeros(N, Cn, T, dim, device=dev)
            flags = torch.zeros(N, Cn, T, dtype=torch.uint8, device=dev)
            nacc = torch.zeros(Cn, T, dtype=torch.int64, device=dev)
            E.run(spec.engine(dev), prop.engine(dev), state=s_d, logp=l_d, beta=dt(beta), step0=0, n_steps=N, burn_in=burn,
                  swap_every=se, trace=trace, accept_flags=flags, ext_prop=dt(ep), ext_u=dt(eu), ext_swap_u=dt(es),
                  n_accept=nacc)
            torch.cuda.synchronize()
            first = H.first_mismatch(flags.cpu().numpy(), want["accept_flags"])
            upto = N if first is None else first
            same = np.allclose(trace.cpu().numpy()[:upto], want["trace"][:upto], rtol=3e-5, atol=3e-5, equal_nan=True)
            if not same:
                break
            if upto >= 2:
                ok_full = True
                break
        # production variant, Philox: the final log-densities must belong to the final states, counts must be close to
        # the oracle's on the same stream
        s_d, l_d = dt(st), dt(lp)
        nacc = torch.zeros(Cn, T, dtype=torch.int64, device=dev)
        E.run(spec.engine(dev), prop.engine(dev), state=s_d, logp=l_d, beta=dt(beta), step0=3, n_steps=40, burn_in=0,
              swap_every=se, seed=77 + dim, chain_offset=2, n_accept=nacc)
        torch.cuda.synchronize()
        w2 = O.run(spec.oracle(), prop.oracle(), state=st, logp=lp, beta=beta, step0=3, n_steps=40, burn_in=0, swap_every=se,
                   seed=77 + dim, chain_offset=2)
        own = O.logdensity(spec.oracle(), s_d.cpu().numpy().reshape(-1, dim), "f64").reshape(Cn, T)
        got_lp = l_d.cpu().numpy()
        fin = np.isfinite(own)
        ok_prod = np.array_equal(np.isfinite(got_lp), fin) and np.allclose(got_lp[fin], own[fin], rtol=2e-5, atol=2e-3)
        ok_prod &= abs(int(nacc.sum()) - int(w2["n_accept"].sum())) <= max(4, 0.1 * int(w2["n_accept"].sum()))
        # ... and the fixture variant of the same kernel, run on the same Philox stream with a trace attached, must
        # reproduce the production variant bit for bit (the two are separate compilations of one template)
        s_f, l_f = dt(st), dt(lp)
        nacc_f = torch.zeros(Cn, T, dtype=torch.int64, device=dev)
        tr = torch.zeros(40, Cn, T, dim, device=dev)
        E.run(spec.engine(dev), prop.engine(dev), state=s_f, logp=l_f, beta=dt(beta), step0=3, n_steps=40, burn_in=0,
              swap_every=se, seed=77 + dim, chain_offset=2, n_accept=nacc_f, trace=tr)
        torch.cuda.synchronize()
        ok_prod &= torch.equal(s_f, s_d) and torch.equal(l_f, l_d) and torch.equal(nacc_f, nacc) and torch.equal(tr[-1], s_d)
        if not (ok_full and ok_prod):
            bad += 1
            print(f"  MISMATCH target {tk} proposal {pname} dim {dim}: fixture ok={ok_full} (first flip {first}) production ok={ok_prod}", flush=True)
    print(f"pair target {tk} proposal {pname}: {len(WIDTH_DIMS)} widths x 2 variants, {bad} bad", flush=True)
    return bad


if __name__ == "__main__":
    if len(sys.argv) == 3:
        sys.exit(1 if one_pair(int(sys.argv[1]), int(sys.argv[2])) else 0)
    failed = []
    for tk in range(11):
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
