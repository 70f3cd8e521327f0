"""Randomised check of the split-step entry points (ptrwm_split_propose / ptrwm_split_accept, with the library's own
log-density between them) against the fused kernel: bit-exact states, log-densities and statistics (needs a GPU).

    python tools/fuzz_split.py [n_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rwm-pt-pytorch_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import helpers as H  # noqa: E402
import ptrwm_hip as E  # noqa: E402
from fuzz_vs_oracle import random_target  # noqa: E402
from oracle import oracle as O  # noqa: E402

f32 = np.float32


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda:0")
    dt = lambda a, d=torch.float32: torch.tensor(np.ascontiguousarray(a), device=dev, dtype=d)  # noqa: E731
    for case in range(n_cases):
        dim = int(rng.choice([1, 2, 3, 5, 7, 10, 13, 20, 24, 30, 33, 41, 50, 57, 64, 77, 100, 104]))
        spec, x0 = random_target(rng, dim)
        dim = spec.dim
        T = int(rng.choice([1, 2, 3, 5, 8, 21, 32, 33, 64, 65, 100, 130, 200, 256]))
        if dim > 64 and T > 128:
            T = 128  # above dim 64 (lane-split kernel only, 512-thread workgroups) a ladder holds at most 128 temperatures
        Cn = int(rng.integers(1, 5)) if T > 64 else int(rng.integers(1, 30))
        beta = (0.03 ** (np.arange(T) / max(1, T - 1))).astype(f32)
        pk = str(rng.choice(["Normal", "Laplace", "UniformRadius"]))
        scale = float(rng.uniform(0.2, 1.5)) * 2.38**2 / dim * (0.05 if "Rosenbrock" in spec.cls or "Beta" in spec.cls else 1.0)
        if pk == "Normal":
            prop = H.proposal_spec(pk, dim, beta, base_variance_scalar=scale)
        elif pk == "Laplace":
            prop = H.proposal_spec(pk, dim, beta, base_variance_vector=np.full(dim, scale, f32))
        else:
            prop = H.proposal_spec(pk, dim, beta, base_radius=float(np.sqrt(scale * dim)))
        N, se, burn = int(rng.integers(4, 20)), int(rng.integers(1, 6)), int(rng.integers(0, 6))
        order, mode = str(rng.choice(["sequential", "even_odd"])), str(rng.choice(["exchange", "reference_copy"]))
        seed, off = int(rng.integers(0, 2**40)), int(rng.integers(0, 1000))
        st = np.broadcast_to(x0.astype(f32), (Cn, T, dim)).copy()
        lp = np.broadcast_to(O.logdensity(spec.oracle(), x0[None].astype(f32)).astype(f32), (Cn, T)).copy()

        def fresh():
            return dt(st), dt(lp), {k: torch.zeros(Cn, T, dtype=(torch.float64 if k == "sq_jump" else torch.int64), device=dev)
                                   for k in ("n_accept", "sq_jump", "swap_accept", "last_swap_ordinal")}

        tgt, pr = spec.engine(dev), prop.engine(dev)
        common = dict(burn_in=burn, swap_every=se, swap_mode=E.SWAP_MODES[mode], swap_order=E.SWAP_ORDERS[order], seed=seed,
                      chain_offset=off)
        s1, l1, st1 = fresh()
        E.run(tgt, pr, state=s1, logp=l1, beta=dt(beta), step0=0, n_steps=N, **common, **st1)
        s2, l2, st2 = fresh()
        plan = E.RunPlan(None, pr, state=s2, logp=l2, beta=dt(beta), **common, **st2)
        for s in range(N):
            props = plan.split_propose(s)
            plan.split_accept(s, E.logdensity(tgt, props.view(-1, dim)).view(Cn, T))
        torch.cuda.synchronize()
        ok = torch.equal(s1, s2) and torch.equal(l1, l2) and all(torch.equal(st1[k], st2[k]) for k in st1)
        print(f"case {case:3d}: {spec.cls:30s} dim {dim:3d} T {T:3d} C {Cn:2d} {pk:13s} {order:10s} {mode:14s} N {N:2d} se {se} "
              f"burn {burn}", "ok" if ok else "MISMATCH", flush=True)
        if not ok:
            for k in st1:
                if not torch.equal(st1[k], st2[k]):
                    print("   ", k, "differs", (st1[k] != st2[k]).sum().item(), "entries")
            print("    state differs in", (s1 != s2).sum().item(), "logp in", (l1 != l2).sum().item())
            sys.exit(1)
    print(f"{n_cases} cases: split steps reproduce the fused kernel bit for bit")


if __name__ == "__main__":
    main()
