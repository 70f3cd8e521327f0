"""Throughput over the dims around and above 64 at BASELINE configs[2]'s batch (65 536 ladders x 32 temperatures) - the
record for the canonical classes W = 20 / 24 / 28 of dim 65..80 / 81..96 / 97..112 (philox.h canon_width), which replaced
the single W = 28 class whose cost made dim 65 run at half the rate of dim 64 (profiles/r02_form_sweep.txt).
Development aid; needs a GPU:  python tools/cliff_sweep.py > profiles/r03_dim_cliff.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rwm-pt-pytorch_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import ptrwm_hip as E  # noqa: E402
from form_sweep import rate  # noqa: E402

C, T = 65536, 32
print(f"{'dim':>4} {'T':>3} {'ladders':>7} {'form':>6} {'chain-steps/s':>14} {'dim-steps/s':>12} {'vs dim 64 per dim':>18}")
base = None
for dim in (56, 60, 64, 65, 68, 72, 76, 80, 81, 84, 88, 92, 96, 97, 100, 104):
    r = rate(dim, T, C, E.FORM_AUTO, target_steps=4e9)
    if dim == 64:
        base = r * dim
    rel = f"{r * dim / base:18.2f}" if base else " " * 18
    print(f"{dim:4d} {T:3d} {C:7d} {'thread' if E.has_thread_variant(0, 0, dim) else 'quad':>6} {r:14.3e} {r * dim:12.3e} {rel}", flush=True)
