import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rwm-pt-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import helpers as H
import ptrwm_hip as E
from algorithms import geometric_beta_ladder
dev = torch.device("cuda:0")
dim, T = 30, 32
spec = H.target_spec("rc15_d30")
ladder = geometric_beta_ladder(T)
prop = H.proposal_spec("Normal", dim, ladder, base_variance_scalar=2.38**2 / dim)
def run(C, form, full, n=100):
    st = np.zeros((C, T, dim), np.float32)
    lp = E.logdensity(spec.engine(dev), H.dev_t(st.reshape(-1, dim), dev)).cpu().numpy().reshape(C, T)
    with E.kernel_form(form):
        return H.gpu_run(spec, prop, dev, state=st, logp=lp, beta=np.float32(ladder), step0=0, n_steps=n, burn_in=20, swap_every=10,
                         seed=4242, trace_temps=T if full else 0, want_flags=full)
base = run(65536, E.FORM_AUTO, False)
for C in (64, 4096):
    for form, fname in ((E.FORM_THREAD, "thread"), (E.FORM_QUAD, "quad")):
        for full in (False, True):
            r = run(C, form, full)
            out = []
            for k in ("state", "logp", "n_accept", "sq_jump", "swap_accept", "last_swap_ordinal"):
                a, b = base[k][:C], r[k]
                ne = (a != b)
                out.append(f"{k}:{int(ne.sum())}")
                if k == "logp" and ne.any():
                    idx = np.argwhere(ne)[:5]
                    out.append(str([(tuple(i), float(a[tuple(i)]), float(b[tuple(i)])) for i in idx]))
            print(C, fname, "full" if full else "prod", " ".join(out), flush=True)
