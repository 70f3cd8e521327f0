"""Accuracy of the variate transforms where the hardware transcendentals are weakest (development aid, needs a GPU):
  Laplace:    log1p(-2|u|) is evaluated as log2(1 + arg) ln 2; on the 24-bit uniform lattice 1 + arg is exact, so the only
              error is v_log_f32's own near 1.  Checked on every lattice point within 4096 steps of u = 0.5 (both sides) and
              on a random sample, against the float64 formula of proposal_distributions/laplace.py:24-37.
  Normal:     Box-Muller radius sqrt(-2 ln u1) for the smallest and largest u1 of the lattice (tails and near-zero radii),
              against float64.
Prints the worst absolute and relative errors of the increments."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rwm-pt-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import helpers as H  # noqa: E402
import ptrwm_hip as E  # noqa: E402

dev = torch.device("cuda:0")
f32 = np.float32
dim = 4
beta = np.ones(1, f32)

# ---- Laplace -----------------------------------------------------------------------------------------------------------
scale = f32(0.7)
prop = H.proposal_spec("Laplace", dim, beta, base_variance_vector=np.full(dim, 2 * scale * scale, f32))
k = np.arange(-4096, 4097)
near = (0.5 + k * 2.0**-24).astype(f32)  # lattice points around u = 0.5 (increment ~ 0)
rng = np.random.default_rng(0)
rand = (rng.integers(0, 2**24, 200000) * 2.0**-24).astype(f32)
u = np.concatenate([near, rand, f32([0.0, 2.0**-24, 1 - 2.0**-24])])
pad = (-len(u)) % dim
u = np.concatenate([u, np.full(pad, 0.25, f32)]).reshape(-1, 1, dim)
got = E.propose(prop.engine(dev), dim, u.shape[0], ext_raw=torch.tensor(u, device=dev)).cpu().numpy().astype(np.float64).ravel()
uu = u.astype(np.float64).ravel() - 0.5
s_eff = float(np.sqrt(f32(2 * scale * scale) / f32(2)))
want = -s_eff * np.sign(uu) * np.log1p(np.maximum(-2 * np.abs(uu), float(f32(-0.999999))))  # the clamp constant as fp32 holds it
err = np.abs(got - want)
rel = err / np.maximum(np.abs(want), 1e-300)
nz = want != 0
print(f"Laplace, scale {s_eff:.3f}: {len(want)} points; worst abs error {err.max():.3e}; worst rel error {rel[nz].max():.3e} "
      f"(at |increment| = {np.abs(want[nz][np.argmax(rel[nz])]):.3e}); rel error over |increment| > 1e-3: "
      f"{rel[np.abs(want) > 1e-3].max():.3e}; exact zeros reproduced: {bool(np.all(got[~nz] == 0))}")

# ---- Normal: radius tails ------------------------------------------------------------------------------------------------
# ext_raw for the Normal proposal are the normals themselves, so the Philox path is exercised through ptrwm_philox_raw-free
# arithmetic here: the radius transform alone, evaluated with torch on the same hardware ops the kernel uses
# (v_log_f32, v_sqrt_f32 via torch.log2 / torch.sqrt in float32) against float64.
n = torch.cat([torch.arange(0, 4096), torch.arange(2**24 - 4096, 2**24)]).to(dev)
u1 = ((n.to(torch.float32) + 1.0) * 2.0**-24)
rad = torch.sqrt((-2.0 * 0.6931471805599453) * torch.log2(u1)).cpu().numpy().astype(np.float64)
want = np.sqrt(-2.0 * np.log((n.cpu().numpy().astype(np.float64) + 1.0) * 2.0**-24))
err = np.abs(rad - want)
print(f"Box-Muller radius on the {len(want)} extreme lattice points: worst abs error {err.max():.3e} "
      f"(radius up to {want.max():.3f}); worst rel error over radius > 1e-2: {(err / np.maximum(want, 1e-300))[want > 1e-2].max():.3e}")
