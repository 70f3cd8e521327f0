"""Both kernel forms over the dims of the 33..64 class (generic register widths of the thread kernel) at a full batch
(development aid; needs a GPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rwm-pt-pytorch_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import ptrwm_hip as E  # noqa: E402
from form_sweep import rate  # noqa: E402

print(f"{'dim':>4} {'T':>4} {'chains':>7} {'thread':>10} {'quad':>10} {'quad/thread':>11} {'thread per dim':>14}")
for dim in (16, 17, 24, 25, 29, 31, 32, 33, 40, 41, 48, 49, 50, 56, 57, 60, 63, 64):
    for T, C in ((32, 16384), (1, 262144)):
        a, b = rate(dim, T, C, E.FORM_THREAD), rate(dim, T, C, E.FORM_QUAD)
        print(f"{dim:4d} {T:4d} {C:7d} {a:10.3e} {b:10.3e} {b / a:11.2f} {a * dim:14.3e}", flush=True)
