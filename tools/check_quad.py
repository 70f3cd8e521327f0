"""The lane-split (quad) form of the fused kernel against the one-thread-per-replica form, bit for bit (needs a GPU):

    python tools/check_quad.py [n_random_cases] [seed]

First a fixed grid - every target kernel x proposal x a dim of every quad width (generic and dim-compiled-in), narrow and
wide ladders - then random configurations (the generator of tools/fuzz_vs_oracle.py).  Each case runs the SAME Philox
stream (and, for the fixture variants, the same external randoms) through ptrwm_run twice, with the form pinned to
THREAD and to QUAD: states, log-densities, acceptance counts, squared-jump sums (fp64 bits), swap counts, last-swap
ordinals, per-step traces and accept flags must be IDENTICAL.

Above dim 64 there is no one-thread-per-replica step kernel (ptrwm_has_thread_variant == 0: FORM_THREAD would run the
lane-split kernel too, and comparing it with itself proves nothing): those cases check the lane-split kernel against the
ORACLE instead - external randoms and the Philox stream, full horizon, every differing decision proven
(tests/helpers.check_parity / check_parity_philox) - and are counted separately."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rwm-pt-pytorch_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import helpers as H  # noqa: E402
import ptrwm_hip as E  # noqa: E402
from oracle import oracle as O  # noqa: E402

f32 = np.float32
dev = torch.device("cuda:0")


def dt(a, d=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), device=dev, dtype=d)


def run_form(form, spec, prop, st, lp, beta, N, *, burn, se, mode, order, seed, offset, ext=None, full=False):
    Cn, T, D = st.shape
    s_d, l_d = dt(st), dt(lp)
    out = {a: torch.zeros(Cn, T, dtype=(torch.float64 if a == "sq_jump" else torch.int64), device=dev)
           for a in ("n_accept", "sq_jump", "swap_accept", "last_swap_ordinal")}
    kw = {}
    if full or ext is not None:
        kw["trace"] = torch.zeros(N, Cn, T, D, device=dev)
        kw["trace_logp"] = torch.zeros(N, Cn, T, device=dev)
        kw["accept_flags"] = torch.zeros(N, Cn, T, dtype=torch.uint8, device=dev)
    if ext is not None:
        kw.update(ext_prop=dt(ext[0]), ext_u=dt(ext[1]), ext_swap_u=None if ext[2] is None or ext[2].shape[0] == 0 else dt(ext[2]))
    with E.kernel_form(form):
        E.run(spec.engine(dev), prop.engine(dev), state=s_d, logp=l_d, beta=dt(beta), step0=3, n_steps=N, burn_in=burn,
              swap_every=se, swap_mode=mode, swap_order=order, seed=seed, chain_offset=offset, **out, **kw)
    torch.cuda.synchronize()
    res = {k: v.cpu().numpy() for k, v in out.items()}
    res.update(state=s_d.cpu().numpy(), logp=l_d.cpu().numpy())
    for k in ("trace", "trace_logp", "accept_flags"):
        if k in kw:
            res[k] = kw[k].cpu().numpy()
    return res


def same(a, b):
    bad = []
    for k in a:
        x, y = a[k], b[k]
        if x.dtype.kind == "f":
            ok = np.array_equal(x.view(np.uint32 if x.dtype == np.float32 else np.uint64),
                                y.view(np.uint32 if y.dtype == np.float32 else np.uint64))
        else:
            ok = np.array_equal(x, y)
        if not ok:
            bad.append(k)
    return bad


def one_case(tag, spec, x0, pk, T, Cn, N, rng, se=3, burn=2, mode=E.SWAP_EXCHANGE, order=E.ORDER_SEQUENTIAL):
    dim = spec.dim
    if not E.has_quad_variant(spec.kind, H.PROPOSAL_KIND[pk], dim, T):
        return None
    beta = (0.03 ** (np.arange(T) / max(1, T - 1))).astype(f32)
    scale = 2.38**2 / dim * (0.05 if "Rosenbrock" in spec.cls or "Beta" in spec.cls else 1.0)
    if pk == "Normal":
        prop = H.proposal_spec(pk, dim, beta, base_variance_scalar=scale)
    elif pk == "Laplace":
        prop = H.proposal_spec(pk, dim, beta, base_variance_vector=np.full(dim, scale, f32))
    else:
        prop = H.proposal_spec(pk, dim, beta, base_radius=float(np.sqrt(scale * dim)))
    st = np.broadcast_to(x0.astype(f32), (Cn, T, dim)).copy()
    st += (1e-3 * rng.standard_normal(st.shape)).astype(f32) * (0 if "Beta" in spec.cls else 1)
    lp = O.logdensity(spec.oracle(), st.reshape(-1, dim)).astype(f32).reshape(Cn, T)
    kw = dict(burn=burn, se=se, mode=mode, order=order, seed=int(rng.integers(1, 2**40)), offset=int(rng.integers(0, 10**6)))
    problems = []
    if not E.has_thread_variant(spec.kind, H.PROPOSAL_KIND[pk], dim):
        # only one form exists: the oracle is the second opinion
        raw = E.ext_raw_per_step(prop.kind, dim)
        ep = rng.standard_normal((N, Cn, T, raw)).astype(f32)
        if pk == "Laplace":
            ep = rng.random((N, Cn, T, raw)).astype(f32)
        elif pk == "UniformRadius":
            ep[..., -1] = rng.random((N, Cn, T)).astype(f32)
        n_ev = H.events_upto(3 + N, se, burn) - H.events_upto(3, se, burn)
        common = dict(state=st, logp=lp, beta=beta, step0=3, n_steps=N, burn_in=burn, swap_every=se, swap_mode=mode,
                      swap_order=order, chain_offset=kw["offset"])
        try:
            H.check_parity(H.gpu_runner(spec, prop, dev), H.oracle_runner(spec, prop), spec, prop, ext_prop=ep,
                           ext_u=rng.random((N, Cn, T)).astype(f32),
                           ext_swap_u=rng.random((n_ev, Cn, T - 1)).astype(f32) if T > 1 else None,
                           exact_states=pk == "Normal", **common)
            H.check_parity_philox(H.gpu_runner(spec, prop, dev), spec, prop, seed=kw["seed"], segment=30, **common)
        except AssertionError as e:
            print(f"  MISMATCH {tag} (vs oracle): {spec.cls} dim {dim} T {T} C {Cn} {pk} mode {mode} order {order}: {str(e)[:300]}", flush=True)
            return False, 0.0, "oracle"
        return True, 0.0, "oracle"
    # production variants (Philox), fixture variants on the same Philox stream with trace/flags, fixture on external randoms
    a, b = run_form(E.FORM_THREAD, spec, prop, st, lp, beta, N, **kw), run_form(E.FORM_QUAD, spec, prop, st, lp, beta, N, **kw)
    problems += [f"production:{k}" for k in same(a, b)]
    fa = run_form(E.FORM_THREAD, spec, prop, st, lp, beta, N, full=True, **kw)
    fb = run_form(E.FORM_QUAD, spec, prop, st, lp, beta, N, full=True, **kw)
    problems += [f"fixture-philox:{k}" for k in same(fa, fb)]
    problems += [f"fixture-vs-production:{k}" for k in same(a, {k: fb[k] for k in a})]
    raw = E.ext_raw_per_step(prop.kind, dim)
    ep = rng.standard_normal((N, Cn, T, raw)).astype(f32)
    if pk == "Laplace":
        ep = rng.random((N, Cn, T, raw)).astype(f32)
    elif pk == "UniformRadius":
        ep[..., -1] = rng.random((N, Cn, T)).astype(f32)
    ext = (ep, rng.random((N, Cn, T)).astype(f32), rng.random((N // se + 1, Cn, max(T - 1, 1))).astype(f32)[:, :, :T - 1] if T > 1 else None)
    ea = run_form(E.FORM_THREAD, spec, prop, st, lp, beta, N, ext=ext, **kw)
    eb = run_form(E.FORM_QUAD, spec, prop, st, lp, beta, N, ext=ext, **kw)
    problems += [f"fixture-external:{k}" for k in same(ea, eb)]
    moved = float(np.mean(a["n_accept"] > 0))
    if problems:
        print(f"  MISMATCH {tag}: {spec.cls} dim {dim} T {T} C {Cn} {pk} mode {mode} order {order}: {problems}", flush=True)
    return (not problems), moved, "forms"


def main():
    from check_all_variants import make_spec
    from fuzz_vs_oracle import random_target

    n_random = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    n = bad = n_oracle = 0
    # fixed grid: 12 target kernels x 3 proposals x dims of every quad width (30 / 50 / 100 have dim compiled in)
    for tk in range(12):
        for pk in ("Normal", "Laplace", "UniformRadius"):
            for dim, T, Cn in ((30, 1, 70), (7, 5, 9), (32, 16, 3), (50, 8, 5), (41, 17, 2), (100, 4, 6), (77, 40, 2)):
                if tk == 2 and dim < 2:
                    continue
                if tk == 3 and dim % 2:
                    dim -= 1
                spec, x0 = make_spec(H, tk, dim, rng)
                r = one_case("grid", spec, x0, pk, T, Cn, 24, rng, order=E.ORDER_EVEN_ODD if (tk + dim) % 2 else E.ORDER_SEQUENTIAL,
                             mode=E.SWAP_REFERENCE_COPY if tk % 3 == 0 else E.SWAP_EXCHANGE)
                if r is not None:
                    n += 1
                    bad += 0 if r[0] else 1
                    n_oracle += r[2] == "oracle"
    print(f"grid: {n} cases ({n - n_oracle} comparing two different kernels, {n_oracle} above dim 64 against the oracle), "
          f"{bad} mismatching", flush=True)
    for case in range(n_random):
        dim = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 13, 20, 24, 29, 30, 31, 32, 33, 41, 48, 50, 57, 63, 64, 65, 77, 96, 97, 100, 104]))
        spec, x0 = random_target(rng, dim)
        T = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 21, 32, 33, 64, 65, 100, 128]))
        Cn = int(rng.integers(1, 6)) if T > 16 else int(rng.integers(1, 80))
        pk = str(rng.choice(["Normal", "Laplace", "UniformRadius"]))
        r = one_case(f"random {case}", spec, x0, pk, T, Cn, int(rng.integers(5, 40)), rng, se=int(rng.integers(1, 6)),
                     burn=int(rng.integers(0, 6)), mode=int(rng.integers(0, 2)), order=int(rng.integers(0, 2)))
        if r is not None:
            n += 1
            bad += 0 if r[0] else 1
            n_oracle += r[2] == "oracle"
    print(f"{n - n_oracle} cases: lane-split and one-thread-per-replica kernels bit-identical; {n_oracle} cases above dim 64 (one "
          f"form only): lane-split kernel follows the oracle" if bad == 0 else f"{bad} of {n} cases MISMATCH")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
