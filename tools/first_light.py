"""First-light check of the HIP engine on a real MI355X: KAT, log-densities, one ext-random
trajectory and a timing of BASELINE config 3.  Development aid; the real checks live in tests/."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rwm-pt-pytorch_amd"))
sys.path.insert(0, ROOT)

import ptrwm_hip as E  # noqa: E402
from oracle import oracle as O  # noqa: E402

dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0))

# 1. Philox KAT
r = E.philox_raw(0, 0, 0, 0, 0, 2, dev).cpu().numpy()
print("philox", [hex(int(v)) for v in r[0]], "expect 6627e8d5 e169c58d bc57ac4c 9b00dbd8")

# 2. log-density
D = 30
lw = np.log(np.array([0.5, 0.3, 0.2], dtype=np.float32))
ot = O.Target(O.TARGET_ROUGH_CARPET, D, p=[-15, 0, 15, *lw, 0.0])
et = E.Target(E.TARGET_ROUGH_CARPET, D, p=(-15, 0, 15, *lw, 0.0))
rng = np.random.default_rng(0)
pts = (rng.standard_normal((4096, D)) * 8).astype(np.float32)
g = E.logdensity(et, torch.tensor(pts, device=dev)).cpu().numpy()
o32 = O.logdensity(ot, pts, "f32")
o64 = O.logdensity(ot, pts, "f64")
print("logp max|gpu-f64|", np.abs(g - o64).max(), "max|o32-f64|", np.abs(o32 - o64).max())

# 3. ext-random trajectory, PT T=8
T, Cn, N = 8, 5, 400
beta = (0.01 ** (np.arange(T) / (T - 1))).astype(np.float32)
ts = np.sqrt(np.float32(2.38**2 / D) / beta).astype(np.float32)
op = O.Proposal(O.PROPOSAL_NORMAL, ts)
ep = E.Proposal(E.PROPOSAL_NORMAL, torch.tensor(ts, device=dev))
z = rng.standard_normal((N, Cn, T, D)).astype(np.float32)
u = rng.random((N, Cn, T)).astype(np.float32)
us = rng.random((N // 10, Cn, T - 1)).astype(np.float32)
st0 = np.zeros((Cn, T, D), np.float32)
lp0 = np.tile(O.logdensity(ot, np.zeros((1, D))).astype(np.float32), (Cn, T))
for mode in (0, 1):
    for order in (0, 1):
        ro = O.run(ot, op, state=st0, logp=lp0, beta=beta, step0=0, n_steps=N, burn_in=20, swap_every=10,
                   swap_mode=mode, swap_order=order, ext_prop=z, ext_u=u, ext_swap_u=us, want_flags=True)
        st = torch.tensor(st0, device=dev)
        lp = torch.tensor(lp0, device=dev)
        na = torch.zeros(Cn, T, dtype=torch.int64, device=dev)
        sq = torch.zeros(Cn, T, dtype=torch.float64, device=dev)
        sa = torch.zeros(Cn, T, dtype=torch.int64, device=dev)
        lo = torch.zeros(Cn, T, dtype=torch.int64, device=dev)
        fl = torch.zeros(N, Cn, T, dtype=torch.uint8, device=dev)
        E.run(et, ep, state=st, logp=lp, beta=torch.tensor(beta, device=dev), step0=0, n_steps=N, burn_in=20,
              swap_every=10, swap_mode=mode, swap_order=order, n_accept=na, sq_jump=sq, swap_accept=sa,
              last_swap_ordinal=lo, ext_prop=torch.tensor(z, device=dev), ext_u=torch.tensor(u, device=dev),
              ext_swap_u=torch.tensor(us, device=dev), accept_flags=fl)
        torch.cuda.synchronize()
        print(f"mode {mode} order {order}: flags equal {np.array_equal(fl.cpu().numpy(), ro['accept_flags'])}",
              "state max diff", np.abs(st.cpu().numpy() - ro["state"]).max(),
              "n_acc eq", np.array_equal(na.cpu().numpy(), ro["n_accept"]),
              "swap eq", np.array_equal(sa.cpu().numpy(), ro["swap_accept"]), int(sa.sum()),
              "ord eq", np.array_equal(lo.cpu().numpy(), ro["last_swap_ordinal"]),
              "sq rel", float(np.abs(sq.cpu().numpy() - ro["sq_jump"]).max() / ro["sq_jump"].max()))

# 4. Philox mode vs oracle
ro = O.run(ot, op, state=st0, logp=lp0, beta=beta, step0=0, n_steps=200, burn_in=20, swap_every=10, seed=1234,
           chain_offset=7, want_flags=True)
st = torch.tensor(st0, device=dev)
lp = torch.tensor(lp0, device=dev)
fl = torch.zeros(200, Cn, T, dtype=torch.uint8, device=dev)
E.run(et, ep, state=st, logp=lp, beta=torch.tensor(beta, device=dev), step0=0, n_steps=200, burn_in=20,
      swap_every=10, seed=1234, chain_offset=7, accept_flags=fl)
torch.cuda.synchronize()
f = fl.cpu().numpy()
print("philox mode: flag agreement", (f == ro["accept_flags"]).mean(), "state max diff",
      np.abs(st.cpu().numpy() - ro["state"]).max())

# 5. timing config 3
Cn, T = 65536, 32
beta = (0.01 ** (np.arange(T) / (T - 1))).astype(np.float32)
ts = np.sqrt(np.float32(2.38**2 / D) / beta).astype(np.float32)
ep = E.Proposal(E.PROPOSAL_NORMAL, torch.tensor(ts, device=dev))
bt = torch.tensor(beta, device=dev)
st = torch.zeros(Cn, T, D, device=dev)
lp = E.logdensity(et, st.view(-1, D)).view(Cn, T).contiguous()
na = torch.zeros(Cn, T, dtype=torch.int64, device=dev)
sq = torch.zeros(Cn, T, dtype=torch.float64, device=dev)
sa = torch.zeros(Cn, T, dtype=torch.int64, device=dev)
for K in (1, 10, 100, 100, 100):
    torch.cuda.synchronize()
    t0 = time.time()
    E.run(et, ep, state=st, logp=lp, beta=bt, step0=0, n_steps=K, burn_in=0, swap_every=10, seed=1, n_accept=na,
          sq_jump=sq, swap_accept=sa)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"cfg3 K={K}: {dt*1e3:.2f} ms -> {Cn*T*K/dt:.3e} chain-steps/s  ({Cn*T*K/dt*264/1e9:.1f} GB/s algorithmic)")
print("acc rate per temp", (na.sum(0).double() / (Cn * 311)).cpu().numpy().round(3))
print("swap acc per pair", (sa.sum(0).double() / (Cn * 31)).cpu().numpy().round(3))
