"""ctypes/numpy front-end of the CPU oracle (oracle/ptrwm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Never imported by the product package (rwm-pt-pytorch_amd/).

All arrays are host numpy arrays; the struct layouts restate include/ptrwm.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

TARGET_ROUGH_CARPET, TARGET_THREE_MIXTURE, TARGET_FULL_ROSENBROCK, TARGET_EVEN_ROSENBROCK = 0, 1, 2, 3
TARGET_HYBRID_ROSENBROCK, TARGET_IID_GAMMA, TARGET_IID_BETA = 4, 5, 6
TARGET_DIAG_GAUSSIAN, TARGET_HYPERCUBE, TARGET_NEAL_FUNNEL = 7, 8, 9
PROPOSAL_NORMAL, PROPOSAL_LAPLACE, PROPOSAL_UNIFORM_RADIUS = 0, 1, 2
SWAP_EXCHANGE, SWAP_REFERENCE_COPY = 0, 1
ORDER_SEQUENTIAL, ORDER_EVEN_ODD = 0, 1


class TargetDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dim", C.c_int32), ("p", C.c_float * 12), ("ip", C.c_int32 * 4),
                ("vec0", C.c_void_p), ("vec1", C.c_void_p)]


class ProposalDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("inv_dim", C.c_float), ("temp_scale", C.c_void_p), ("dim_scale", C.c_void_p)]


class RunArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("n_temps", C.c_int32), ("n_chains", C.c_int64), ("chain_offset", C.c_int64),
        ("state", C.c_void_p), ("logp", C.c_void_p), ("beta", C.c_void_p), ("n_accept", C.c_void_p),
        ("sq_jump", C.c_void_p), ("swap_accept", C.c_void_p), ("last_swap_ordinal", C.c_void_p),
        ("step0", C.c_int64), ("n_steps", C.c_int64), ("burn_in", C.c_int64), ("swap_every", C.c_int32),
        ("swap_mode", C.c_int32), ("swap_order", C.c_int32), ("swap_event_offset", C.c_int32), ("seed", C.c_uint64),
        ("ext_prop", C.c_void_p), ("ext_u", C.c_void_p), ("ext_swap_u", C.c_void_p), ("trace", C.c_void_p),
        ("trace_logp", C.c_void_p), ("trace_chains", C.c_int64), ("trace_temps", C.c_int32),
        ("trace_every", C.c_int32), ("trace_row0", C.c_int64), ("accept_flags", C.c_void_p),
        ("state_f64", C.c_int32), ("split_flags", C.c_int32), ("device_step", C.c_void_p),
    ]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        for sfx in ("f32", "f64"):
            getattr(_lib, f"oracle_logdensity_{sfx}").restype = C.c_int32
            getattr(_lib, f"oracle_propose_{sfx}").restype = C.c_int32
            getattr(_lib, f"oracle_run_{sfx}").restype = C.c_int32
            sweep = getattr(_lib, f"oracle_swap_sweep_{sfx}")
            sweep.restype, sweep.argtypes = C.c_int32, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32]
    return _lib


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return None if a is None else a.ctypes.data


class Target:
    """Host description of a target: kind, dim, scalar params p/ip, optional per-dim vectors."""

    def __init__(self, kind, dim, p=(), ip=(), vec0=None, vec1=None):
        self.kind, self.dim, self.p, self.ip = kind, dim, tuple(p), tuple(ip)
        self.vec0, self.vec1 = _f32(vec0), _f32(vec1)

    def desc(self):
        d = TargetDesc()
        d.kind, d.dim = self.kind, self.dim
        for i, v in enumerate(self.p):
            d.p[i] = float(v)
        for i, v in enumerate(self.ip):
            d.ip[i] = int(v)
        d.vec0, d.vec1 = _ptr(self.vec0), _ptr(self.vec1)
        return d


class Proposal:
    def __init__(self, kind, temp_scale, dim_scale=None, inv_dim=0.0):
        self.kind = kind
        self.temp_scale = _f32(np.atleast_1d(temp_scale))
        self.dim_scale = _f32(dim_scale)
        self.inv_dim = float(inv_dim)

    def desc(self):
        d = ProposalDesc()
        d.kind, d.inv_dim = self.kind, self.inv_dim
        d.temp_scale, d.dim_scale = _ptr(self.temp_scale), _ptr(self.dim_scale)
        return d


def ext_raw_per_step(kind, dim):
    return {PROPOSAL_NORMAL: dim, PROPOSAL_LAPLACE: dim, PROPOSAL_UNIFORM_RADIUS: dim + 1}[kind]


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*[int(v) & 0xFFFFFFFF for v in ctr])
    k = (C.c_uint32 * 2)(*[int(v) & 0xFFFFFFFF for v in key])
    o = (C.c_uint32 * 4)()
    lib().oracle_philox4x32_10(c, k, o)
    return [int(v) for v in o]


def logdensity(target: Target, x, precision="f32"):
    """log-density of every row of x.  float64 input (the states of a state_f64 run) is evaluated as given, in double."""
    if np.asarray(x).dtype == np.float64 and precision == "f64":
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, target.dim)
        out = np.empty(x.shape[0], dtype=np.float64)
        d = target.desc()
        fn = lib().oracle_logdensity_d_f64
        fn.restype = C.c_int32
        rc = fn(C.byref(d), C.c_void_p(x.ctypes.data), C.c_void_p(out.ctypes.data), C.c_int64(x.shape[0]))
        assert rc == 0, rc
        return out
    x = _f32(x).reshape(-1, target.dim)
    out = np.empty(x.shape[0], dtype=np.float64)
    d = target.desc()
    rc = getattr(lib(), f"oracle_logdensity_{precision}")(C.byref(d), C.c_void_p(x.ctypes.data),
                                                         C.c_void_p(out.ctypes.data), C.c_int64(x.shape[0]))
    assert rc == 0, rc
    return out


def propose(proposal: Proposal, dim, n, seed=0, ext_raw=None, precision="f32"):
    T = proposal.temp_scale.size
    ext_raw = _f32(ext_raw)
    out = np.empty((n, T, dim), dtype=np.float64)
    d = proposal.desc()
    rc = getattr(lib(), f"oracle_propose_{precision}")(C.byref(d), C.c_int32(dim), C.c_int32(T), C.c_int64(n),
                                                       C.c_void_p(_ptr(ext_raw)), C.c_uint64(seed),
                                                       C.c_void_p(out.ctypes.data))
    assert rc == 0, rc
    return out


def philox_randoms(proposal_kind, dim, n_temps, n_chains, *, seed, step0, n_steps, burn_in=0, swap_every=1,
                   chain_offset=0):
    """The raw randoms the oracle's Philox mode consumes for this step range, as external-randoms arrays
    (oracle_philox_randoms): ext_prop [n, C, T, raw], ext_u [n, C, T], ext_swap_u [events, C, T-1] (None for T == 1).
    `run(..., ext_prop=, ext_u=, ext_swap_u=)` with them equals `run(..., seed=)` bit for bit."""
    raw = ext_raw_per_step(proposal_kind, dim)
    ext_prop = np.zeros((n_steps, n_chains, n_temps, raw), dtype=np.float32)
    ext_u = np.zeros((n_steps, n_chains, n_temps), dtype=np.float32)
    n_ev = max(0, (step0 + n_steps) // swap_every - burn_in // swap_every) - max(0, step0 // swap_every - burn_in // swap_every)
    ext_swap_u = np.zeros((n_ev, n_chains, n_temps - 1), dtype=np.float32) if n_temps > 1 else None
    fn = lib().oracle_philox_randoms
    fn.restype = C.c_int32
    rc = fn(C.c_int32(proposal_kind), C.c_int32(dim), C.c_int32(n_temps), C.c_int64(n_chains), C.c_int64(chain_offset),
            C.c_uint64(seed & (2**64 - 1)), C.c_int64(step0), C.c_int64(n_steps), C.c_int64(burn_in), C.c_int32(swap_every),
            C.c_void_p(ext_prop.ctypes.data), C.c_void_p(ext_u.ctypes.data), C.c_void_p(_ptr(ext_swap_u)))
    assert rc == 0, rc
    return ext_prop, ext_u, ext_swap_u


def run(target: Target, proposal: Proposal, *, state, logp, beta, step0, n_steps, burn_in=0, swap_every=1,
        swap_mode=SWAP_EXCHANGE, swap_order=ORDER_SEQUENTIAL, seed=0, chain_offset=0, ext_prop=None, ext_u=None,
        ext_swap_u=None, trace_chains=0, trace_temps=0, want_flags=False, precision="f32", swap_event_offset=0):
    """Runs the oracle.  Returns a dict with the updated state/logp (copies) and statistics.  A float64 `state` selects the
    state_f64 mode of include/ptrwm.h (the reference's dtype=torch.float64): state, trace and ext_prop in double, and the
    double-precision instantiation of the oracle (precision is forced to "f64")."""
    sf = np.asarray(state).dtype == np.float64
    sdt = np.float64 if sf else np.float32
    if sf:
        precision = "f64"
    state = np.array(state, dtype=sdt, order="C", copy=True)
    Cn, T, D = state.shape
    logp = np.array(logp, dtype=np.float32, order="C", copy=True).reshape(Cn, T)
    beta = _f32(beta)
    res = {
        "n_accept": np.zeros((Cn, T), dtype=np.int64),
        "sq_jump": np.zeros((Cn, T), dtype=np.float64),
        "swap_accept": np.zeros((Cn, T), dtype=np.int64),
        "last_swap_ordinal": np.zeros((Cn, T), dtype=np.int64),
    }
    a = RunArgs()
    a.struct_size = C.sizeof(RunArgs)
    a.n_temps, a.n_chains, a.chain_offset = T, Cn, chain_offset
    a.state, a.logp, a.beta = state.ctypes.data, logp.ctypes.data, beta.ctypes.data
    for k, v in res.items():
        setattr(a, k, v.ctypes.data)
    a.step0, a.n_steps, a.burn_in = step0, n_steps, burn_in
    a.swap_every, a.swap_mode, a.swap_order, a.seed = swap_every, swap_mode, swap_order, seed
    a.swap_event_offset = swap_event_offset
    a.state_f64 = 1 if sf else 0
    ext_prop = (None if ext_prop is None else np.ascontiguousarray(ext_prop, dtype=np.float64)) if sf else _f32(ext_prop)
    ext_u, ext_swap_u = _f32(ext_u), _f32(ext_swap_u)
    a.ext_prop, a.ext_u, a.ext_swap_u = _ptr(ext_prop), _ptr(ext_u), _ptr(ext_swap_u)
    if trace_chains:
        res["trace"] = np.zeros((n_steps, trace_chains, trace_temps, D), dtype=sdt)
        res["trace_logp"] = np.zeros((n_steps, trace_chains, trace_temps), dtype=np.float32)
        a.trace, a.trace_logp = res["trace"].ctypes.data, res["trace_logp"].ctypes.data
        a.trace_chains, a.trace_temps = trace_chains, trace_temps
    if want_flags:
        res["accept_flags"] = np.zeros((n_steps, Cn, T), dtype=np.uint8)
        a.accept_flags = res["accept_flags"].ctypes.data
    td, pd = target.desc(), proposal.desc()
    rc = getattr(lib(), f"oracle_run_{precision}")(C.byref(td), C.byref(pd), C.byref(a))
    assert rc == 0, rc
    res["state"], res["logp"] = state, logp
    return res


def swap_sweep(*, state, logp, beta, event_index, rng_step=0, rng_stream=2, swap_mode=SWAP_EXCHANGE,
               swap_order=ORDER_SEQUENTIAL, seed=0, chain_offset=0, ext_swap_u=None, precision="f32"):
    """One stand-alone swap event (oracle_swap_sweep: the reference's _attempt_all_swaps, pt_rwm_gpu_optimized.py:
    594-633).  Returns the updated state / logp and the swap statistics of this one event."""
    state = np.array(state, dtype=np.float32, order="C", copy=True)
    Cn, T, D = state.shape
    logp = np.array(logp, dtype=np.float32, order="C", copy=True).reshape(Cn, T)
    beta = _f32(beta)
    res = {"swap_accept": np.zeros((Cn, T), dtype=np.int64), "last_swap_ordinal": np.zeros((Cn, T), dtype=np.int64)}
    a = RunArgs()
    a.struct_size = C.sizeof(RunArgs)
    a.n_temps, a.n_chains, a.chain_offset = T, Cn, chain_offset
    a.state, a.logp, a.beta = state.ctypes.data, logp.ctypes.data, beta.ctypes.data
    a.swap_accept, a.last_swap_ordinal = res["swap_accept"].ctypes.data, res["last_swap_ordinal"].ctypes.data
    a.step0, a.swap_mode, a.swap_order, a.seed = rng_step, swap_mode, swap_order, seed
    ext_swap_u = _f32(ext_swap_u)
    a.ext_swap_u = _ptr(ext_swap_u)
    rc = getattr(lib(), f"oracle_swap_sweep_{precision}")(C.byref(a), D, event_index, rng_stream)
    assert rc == 0, rc
    res["state"], res["logp"] = state, logp
    return res
