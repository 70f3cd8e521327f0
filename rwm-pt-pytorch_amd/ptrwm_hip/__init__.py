"""ctypes binding of the C-ABI engine library (include/ptrwm.h -> lib/libptrwm_hip.so).

This is the only place Python touches the HIP engine.  torch is used for device memory and streams
only: every call passes raw ``tensor.data_ptr()`` device pointers and the current HIP stream.

There is no CPU or eager-torch fallback: if the library is missing, or a tensor is not on a ROCm
device, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import torch  # noqa: F401  (must be imported first: loads the HIP runtime the library binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
# PTRWM_LIB: alternative build of the same library (A/B tuning experiments)
LIB_PATH = os.environ.get("PTRWM_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libptrwm_hip.so")

ABI_VERSION = 3
MAX_DIM = 104
MAX_TEMPS = 256

# target kinds (include/ptrwm.h)
TARGET_ROUGH_CARPET = 0
TARGET_THREE_MIXTURE = 1
TARGET_FULL_ROSENBROCK = 2
TARGET_EVEN_ROSENBROCK = 3
TARGET_HYBRID_ROSENBROCK = 4
TARGET_IID_GAMMA = 5
TARGET_IID_BETA = 6
TARGET_DIAG_GAUSSIAN = 7
TARGET_HYPERCUBE = 8
TARGET_NEAL_FUNNEL = 9
# proposal kinds
PROPOSAL_NORMAL = 0
PROPOSAL_LAPLACE = 1
PROPOSAL_UNIFORM_RADIUS = 2
# swap semantics
SWAP_EXCHANGE = 0
SWAP_REFERENCE_COPY = 1
ORDER_SEQUENTIAL = 0
ORDER_EVEN_ODD = 1

# form of the fused step kernel (ptrwm_set_kernel_form): a speed choice only, results are bit-identical
FORM_AUTO, FORM_THREAD, FORM_QUAD = 0, 1, 2

# short launches (ptrwm_set_stream_mode): the streaming form of the one-thread-per-replica kernel; same bits either way
STREAM_AUTO, STREAM_OFF, STREAM_ON = 0, 1, 2

SWAP_MODES = {"exchange": SWAP_EXCHANGE, "reference_copy": SWAP_REFERENCE_COPY}
SWAP_ORDERS = {"sequential": ORDER_SEQUENTIAL, "even_odd": ORDER_EVEN_ODD}


class PTRWMError(RuntimeError):
    """A C-ABI call returned a negative status."""

    def __init__(self, code: int, where: str):
        self.code = code
        super().__init__(f"{where}: {strerror(code)} (status {code})")


class TargetDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("dim", C.c_int32),
        ("p", C.c_float * 12),
        ("ip", C.c_int32 * 4),
        ("vec0", C.c_void_p),
        ("vec1", C.c_void_p),
    ]


class ProposalDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("inv_dim", C.c_float),
        ("temp_scale", C.c_void_p),
        ("dim_scale", C.c_void_p),
    ]


class RunArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("n_temps", C.c_int32),
        ("n_chains", C.c_int64),
        ("chain_offset", C.c_int64),
        ("state", C.c_void_p),
        ("logp", C.c_void_p),
        ("beta", C.c_void_p),
        ("n_accept", C.c_void_p),
        ("sq_jump", C.c_void_p),
        ("swap_accept", C.c_void_p),
        ("last_swap_ordinal", C.c_void_p),
        ("step0", C.c_int64),
        ("n_steps", C.c_int64),
        ("burn_in", C.c_int64),
        ("swap_every", C.c_int32),
        ("swap_mode", C.c_int32),
        ("swap_order", C.c_int32),
        ("swap_event_offset", C.c_int32),
        ("seed", C.c_uint64),
        ("ext_prop", C.c_void_p),
        ("ext_u", C.c_void_p),
        ("ext_swap_u", C.c_void_p),
        ("trace", C.c_void_p),
        ("trace_logp", C.c_void_p),
        ("trace_chains", C.c_int64),
        ("trace_temps", C.c_int32),
        ("trace_every", C.c_int32),
        ("trace_row0", C.c_int64),
        ("accept_flags", C.c_void_p),
        ("state_f64", C.c_int32),
        ("split_flags", C.c_int32),
        ("device_step", C.c_void_p),
    ]


# every symbol include/ptrwm.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "ptrwm_abi_version": (C.c_int32, []),
    "ptrwm_strerror": (C.c_char_p, [C.c_int32]),
    "ptrwm_ext_raw_per_step": (C.c_int32, [C.c_int32, C.c_int32]),
    "ptrwm_has_variant": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    "ptrwm_set_kernel_form": (C.c_int32, [C.c_int32]),
    "ptrwm_set_stream_mode": (C.c_int32, [C.c_int32]),
    "ptrwm_has_stream_variant": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    "ptrwm_last_launch_kind": (C.c_int32, []),
    "ptrwm_has_quad_variant": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "ptrwm_has_thread_variant": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    "ptrwm_auto_form": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64]),
    "ptrwm_auto_form_for": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32]),
    "ptrwm_device_simds": (C.c_int32, [C.c_void_p]),
    "ptrwm_source_hash": (C.c_char_p, []),
    "ptrwm_form_table_source_hash": (C.c_char_p, []),
    "ptrwm_run": (C.c_int32, [C.POINTER(TargetDesc), C.POINTER(ProposalDesc), C.POINTER(RunArgs), C.c_void_p]),
    "ptrwm_swap_sweep": (C.c_int32, [C.POINTER(RunArgs), C.c_int32, C.c_int64, C.c_int32, C.c_void_p]),
    "ptrwm_split_propose": (
        C.c_int32, [C.POINTER(ProposalDesc), C.POINTER(RunArgs), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ptrwm_split_accept": (
        C.c_int32, [C.POINTER(RunArgs), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ptrwm_split_advance": (C.c_int32, [C.POINTER(RunArgs), C.c_void_p]),
    "ptrwm_logdensity": (C.c_int32, [C.POINTER(TargetDesc), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ptrwm_propose": (
        C.c_int32,
        [C.POINTER(ProposalDesc), C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p],
    ),
    "ptrwm_philox_raw": (
        C.c_int32,
        [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int64, C.c_void_p, C.c_void_p],
    ),
}

_lib = None


def load_library(path: Optional[str] = None):
    """Load (once) and type the engine library.  Raises if it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            f"HIP engine library not found at {p}. Build it first: "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C rwm-pt-pytorch_amd/csrc`. "
            "There is no CPU fallback."
        )
    lib = C.CDLL(p)
    # the version first: a library of another ABI version may lack symbols this binding types below
    lib.ptrwm_abi_version.restype, lib.ptrwm_abi_version.argtypes = SYMBOLS["ptrwm_abi_version"]
    if lib.ptrwm_abi_version() != ABI_VERSION:
        raise RuntimeError(f"ABI mismatch: library {p} is version {lib.ptrwm_abi_version()}, this binding {ABI_VERSION} "
                           "(rebuild: make -C rwm-pt-pytorch_amd/csrc)")
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    form = os.environ.get("PTRWM_KERNEL_FORM")  # tuning aid: auto | thread | quad (results are identical)
    if form:
        lib.ptrwm_set_kernel_form({"auto": FORM_AUTO, "thread": FORM_THREAD, "quad": FORM_QUAD}[form.lower()])
    smode = os.environ.get("PTRWM_STREAM")  # tuning aid: auto | off | on (results are identical)
    if smode:
        lib.ptrwm_set_stream_mode({"auto": STREAM_AUTO, "off": STREAM_OFF, "on": STREAM_ON}[smode.lower()])
    if path is None:
        _lib = lib
    return lib


def strerror(code: int) -> str:
    return load_library().ptrwm_strerror(code).decode()


def _require_device(t: torch.Tensor, name: str, dtype: torch.dtype) -> int:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on {t.device}: the PT-RWM engine only runs on a ROCm GPU (no CPU fallback)."
        )
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t.data_ptr()


def _opt(t: Optional[torch.Tensor], name: str, dtype: torch.dtype) -> Optional[int]:
    return None if t is None else _require_device(t, name, dtype)


def _stream(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class on_device:
    """Makes `device` the current HIP device around a C-ABI call and restores the previous one afterwards.

    The C ABI launches on whatever device is current (the usual HIP contract, stated in include/ptrwm.h), while the
    class API accepts `device='cuda:1'` without the caller ever calling `torch.cuda.set_device`: without this guard
    the kernel would be launched on device 0 with device-1 pointers.  A no-op (two integer compares) when the device
    is already current, so the per-step launch path stays at a few microseconds."""

    __slots__ = ("idx", "prev")

    def __init__(self, device: torch.device):
        self.idx = device.index if device.index is not None else torch.cuda.current_device()
        self.prev = self.idx

    def __enter__(self):
        self.prev = torch.cuda.current_device()
        if self.prev != self.idx:
            torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev != self.idx:
            torch.cuda.set_device(self.prev)
        return False


@dataclass
class Target:
    """Host-side description of a target density; tensors are kept alive by the owner."""

    kind: int
    dim: int
    p: tuple = ()
    ip: tuple = ()
    vec0: Optional[torch.Tensor] = None
    vec1: Optional[torch.Tensor] = None

    def desc(self) -> TargetDesc:
        d = TargetDesc()
        d.kind = self.kind
        d.dim = self.dim
        for i, v in enumerate(self.p):
            d.p[i] = float(v)
        for i, v in enumerate(self.ip):
            d.ip[i] = int(v)
        d.vec0 = _opt(self.vec0, "target.vec0", torch.float32)
        d.vec1 = _opt(self.vec1, "target.vec1", torch.float32)
        return d


@dataclass
class Proposal:
    kind: int
    temp_scale: torch.Tensor  # [T] float32 device
    dim_scale: Optional[torch.Tensor] = None  # [D] float32 device (Laplace)
    inv_dim: float = 0.0

    def desc(self) -> ProposalDesc:
        d = ProposalDesc()
        d.kind = self.kind
        d.inv_dim = float(self.inv_dim)
        d.temp_scale = _require_device(self.temp_scale, "proposal.temp_scale", torch.float32)
        d.dim_scale = _opt(self.dim_scale, "proposal.dim_scale", torch.float32)
        return d


def ext_raw_per_step(proposal_kind: int, dim: int) -> int:
    n = load_library().ptrwm_ext_raw_per_step(proposal_kind, dim)
    if n < 0:
        raise PTRWMError(n, "ptrwm_ext_raw_per_step")
    return n


def has_variant(target_kind: int, proposal_kind: int, dim: int) -> bool:
    return bool(load_library().ptrwm_has_variant(target_kind, proposal_kind, dim))


def has_quad_variant(target_kind: int, proposal_kind: int, dim: int, n_temps: int) -> bool:
    return bool(load_library().ptrwm_has_quad_variant(target_kind, proposal_kind, dim, n_temps))


def has_thread_variant(target_kind: int, proposal_kind: int, dim: int) -> bool:
    """Is there a one-thread-per-replica step kernel for this shape (never above dim 64)?"""
    return bool(load_library().ptrwm_has_thread_variant(target_kind, proposal_kind, dim))


def auto_form(target_kind: int, proposal_kind: int, dim: int, n_temps: int, n_chains: int) -> int:
    """The form FORM_AUTO runs for a launch of this shape on the current device (FORM_THREAD or FORM_QUAD)."""
    rc = load_library().ptrwm_auto_form(target_kind, proposal_kind, dim, n_temps, n_chains)
    if rc < 0:
        raise PTRWMError(rc, "ptrwm_auto_form")
    return rc


def auto_form_for(target_kind: int, proposal_kind: int, dim: int, n_temps: int, n_chains: int, n_simds: int) -> int:
    """The AUTO rule for a device of ``n_simds`` SIMDs (a pure function of its arguments and of the fitted table)."""
    rc = load_library().ptrwm_auto_form_for(target_kind, proposal_kind, dim, n_temps, n_chains, n_simds)
    if rc < 0:
        raise PTRWMError(rc, "ptrwm_auto_form_for")
    return rc


def device_simds(device) -> int:
    """SIMDs (compute units x 4) of ``device`` as the library sees them through the device's current stream."""
    device = torch.device(device)
    with on_device(device):
        rc = load_library().ptrwm_device_simds(_stream(device))
    if rc < 0:
        raise PTRWMError(rc, "ptrwm_device_simds")
    return rc


def source_hash() -> str:
    """sha256 of the kernel sources this library was built from (tools/source_hash.py)."""
    return load_library().ptrwm_source_hash().decode()


def form_table_source_hash() -> str:
    """Hash of the kernel sources the AUTO form table (csrc/form_table.inc) was fitted on."""
    return load_library().ptrwm_form_table_source_hash().decode()


def set_kernel_form(form: int) -> int:
    """Pin the form of the fused step kernel (FORM_AUTO / FORM_THREAD / FORM_QUAD); returns the previous setting.
    The forms are bit-identical on the same Philox stream: this changes speed only (tests and tuning)."""
    prev = load_library().ptrwm_set_kernel_form(form)
    if prev < 0:
        raise PTRWMError(prev, "ptrwm_set_kernel_form")
    return prev


LAUNCH_THREAD, LAUNCH_QUAD, LAUNCH_STREAM = 1, 2, 3


def last_launch_kind() -> int:
    """Which kernel this thread's most recent ptrwm_run enqueued (LAUNCH_THREAD / LAUNCH_QUAD / LAUNCH_STREAM)."""
    return load_library().ptrwm_last_launch_kind()


def has_stream_variant(target_kind: int, proposal_kind: int, dim: int) -> bool:
    """Does the one-thread-per-replica kernel of this shape have a streaming twin for short launches?"""
    return bool(load_library().ptrwm_has_stream_variant(target_kind, proposal_kind, dim))


def set_stream_mode(mode: int) -> int:
    """When ptrwm_run takes the streaming form of the step kernel for short launches (STREAM_AUTO / STREAM_OFF /
    STREAM_ON); returns the previous setting.  The forms give the same bits: this changes speed only."""
    prev = load_library().ptrwm_set_stream_mode(mode)
    if prev < 0:
        raise PTRWMError(prev, "ptrwm_set_stream_mode")
    return prev


class stream_mode:
    """Context manager around set_stream_mode."""

    def __init__(self, mode: int):
        self.mode = mode

    def __enter__(self):
        self.prev = set_stream_mode(self.mode)
        return self

    def __exit__(self, *exc):
        set_stream_mode(self.prev)
        return False


class kernel_form:
    """Context manager around set_kernel_form."""

    def __init__(self, form: int):
        self.form = form

    def __enter__(self):
        self.prev = set_kernel_form(self.form)
        return self

    def __exit__(self, *exc):
        set_kernel_form(self.prev)
        return False


def logdensity(target: Target, x: torch.Tensor) -> torch.Tensor:
    """log-density of every row of ``x`` ([n, dim] float32 on the GPU) -> [n] float32."""
    lib = load_library()
    if x.dim() != 2 or x.shape[1] != target.dim:
        raise ValueError(f"x must have shape [n, {target.dim}], got {tuple(x.shape)}")
    xp = _require_device(x, "x", torch.float32)
    out = torch.empty(x.shape[0], device=x.device, dtype=torch.float32)
    desc = target.desc()
    with on_device(x.device):
        rc = lib.ptrwm_logdensity(C.byref(desc), xp, out.data_ptr(), x.shape[0], _stream(x.device))
    if rc != 0:
        raise PTRWMError(rc, "ptrwm_logdensity")
    return out


def propose(proposal: Proposal, dim: int, n: int, seed: int = 0, ext_raw: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Proposal increments [n, T, dim] from Philox(seed) or from external raw randoms [n, T, raw]."""
    lib = load_library()
    T = proposal.temp_scale.numel()
    dev = proposal.temp_scale.device
    desc = proposal.desc()
    if ext_raw is not None:
        if tuple(ext_raw.shape) != (n, T, ext_raw_per_step(proposal.kind, dim)):
            raise ValueError(f"ext_raw has shape {tuple(ext_raw.shape)}")
    out = torch.empty(n, T, dim, device=dev, dtype=torch.float32)
    with on_device(dev):
        rc = lib.ptrwm_propose(C.byref(desc), dim, T, n, _opt(ext_raw, "ext_raw", torch.float32), seed & (2**64 - 1),
                               out.data_ptr(), _stream(dev))
    if rc != 0:
        raise PTRWMError(rc, "ptrwm_propose")
    return out


def philox_raw(seed: int, c0: int, c1: int, c2: int, c3: int, n: int, device) -> torch.Tensor:
    """n consecutive Philox4x32-10 blocks (counter c0+i) as an int64 tensor [n, 4] of uint32 values."""
    lib = load_library()
    out = torch.empty(n, 4, device=device, dtype=torch.int32)
    if not out.is_cuda:
        raise RuntimeError("philox_raw needs a ROCm device")
    with on_device(out.device):
        rc = lib.ptrwm_philox_raw(seed, c0, c1, c2, c3, n, out.data_ptr(), _stream(out.device))
    if rc != 0:
        raise PTRWMError(rc, "ptrwm_philox_raw")
    return out.to(torch.int64) & 0xFFFFFFFF


class RunPlan:
    """The arguments of ``ptrwm_run`` that do not change between launches of one sampler run, validated and
    marshalled once.  ``launch`` only fills in the step range, the optional trace / fixture buffers and the
    current stream, so a one-step launch costs a few microseconds of host time (the reference's per-step
    ``step()`` calling pattern stays launch-bound on the kernel, not on Python).  The plan holds references to
    its tensors, so the device pointers stay valid for as long as the plan lives."""

    def __init__(
        self,
        target: Target,
        proposal: Proposal,
        *,
        state: torch.Tensor,  # [C, T, D] float32, or float64 (state_f64 of include/ptrwm.h: the reference's dtype=float64)
        logp: torch.Tensor,  # [C, T] float32
        beta: torch.Tensor,  # [T] float32
        burn_in: int = 0,
        swap_every: int = 1,
        swap_mode: int = SWAP_EXCHANGE,
        swap_order: int = ORDER_SEQUENTIAL,
        seed: int = 0,
        chain_offset: int = 0,
        n_accept: Optional[torch.Tensor] = None,  # [C, T] int64
        sq_jump: Optional[torch.Tensor] = None,  # [C, T] float64
        swap_accept: Optional[torch.Tensor] = None,  # [C, T] int64
        last_swap_ordinal: Optional[torch.Tensor] = None,  # [C, T] int64
    ):
        self._lib = load_library()
        if isinstance(state, torch.Tensor) and state.dtype not in (torch.float32, torch.float64):
            raise TypeError(f"state must be torch.float32 or torch.float64, got {state.dtype}")
        self.state_dtype = state.dtype if isinstance(state, torch.Tensor) else torch.float32
        if state.dim() != 3:
            raise ValueError("state must be [n_chains, n_temps, dim]")
        Cn, T, D = state.shape
        if target is not None and D != target.dim:
            raise ValueError(f"state dim {D} != target dim {target.dim}")
        if tuple(logp.shape) != (Cn, T) or beta.numel() != T or proposal.temp_scale.numel() != T:
            raise ValueError("logp/beta/temp_scale shapes do not match state")
        self.shape = (Cn, T, D)
        self.proposal_kind = proposal.kind
        self.device = state.device
        for name, t in (("logp", logp), ("beta", beta), ("temp_scale", proposal.temp_scale)):
            if t.device != self.device:
                raise ValueError(f"{name} is on {t.device}, state on {self.device}: all tensors of a run live on one GPU")
        a = RunArgs()
        a.struct_size = C.sizeof(RunArgs)
        a.n_temps = T
        a.n_chains = Cn
        a.chain_offset = chain_offset
        a.state = _require_device(state, "state", self.state_dtype)
        a.state_f64 = 1 if self.state_dtype == torch.float64 else 0
        a.logp = _require_device(logp, "logp", torch.float32)
        a.beta = _require_device(beta, "beta", torch.float32)
        for name, t, dt in (
            ("n_accept", n_accept, torch.int64),
            ("sq_jump", sq_jump, torch.float64),
            ("swap_accept", swap_accept, torch.int64),
            ("last_swap_ordinal", last_swap_ordinal, torch.int64),
        ):
            if t is not None and tuple(t.shape) != (Cn, T):
                raise ValueError(f"{name} must have shape [{Cn}, {T}]")
            setattr(a, name, _opt(t, name, dt))
        a.burn_in = burn_in
        a.swap_every = swap_every
        a.swap_mode = swap_mode
        a.swap_order = swap_order
        a.seed = seed & (2**64 - 1)
        self._a = a
        # target None: a plan for split steps only (the caller evaluates the density, see split_propose)
        self._t = target.desc() if target is not None else None
        self._p = proposal.desc()
        self._refs = (self._t, self._p, C.byref(self._t) if target is not None else None, C.byref(self._p), C.byref(a))
        self._keep = (target, proposal, state, logp, beta, n_accept, sq_jump, swap_accept, last_swap_ordinal)
        self._plain = True  # no per-launch buffers set in _a
        self._last_trace = (None, None, None)  # (trace, trace_logp, trace_every) marshalled into _a by the last launch
        self._guard = on_device(self.device)  # (after the checks above: they reject CPU tensors first)

    def launch(
        self,
        step0: int,
        n_steps: int,
        *,
        ext_prop: Optional[torch.Tensor] = None,
        ext_u: Optional[torch.Tensor] = None,
        ext_swap_u: Optional[torch.Tensor] = None,
        trace: Optional[torch.Tensor] = None,  # [rows, trace_chains, trace_temps, D]
        trace_logp: Optional[torch.Tensor] = None,
        trace_row0: int = 0,
        trace_every: int = 1,
        accept_flags: Optional[torch.Tensor] = None,  # [n_steps, C, T] uint8
        swap_event_offset: int = 0,  # stand-alone sweeps (swap_sweep) performed before this launch
    ) -> None:
        """Enqueue ``n_steps`` fused MH(+swap) steps, starting at global step ``step0``, on the current stream."""
        if self._t is None:
            raise RuntimeError("this plan has no target description: use split_propose / split_accept")
        a = self._a
        if a.device_step:
            raise RuntimeError("device-step mode is for split steps only: set_device_step(None) before launch()")
        a.step0 = step0
        a.n_steps = n_steps
        a.swap_event_offset = swap_event_offset
        plain = (ext_prop is None and ext_u is None and ext_swap_u is None and trace is None and trace_logp is None
                 and accept_flags is None)
        # trace-only launches into the SAME buffers as the previous launch (the reference's step()-at-a-time pattern with
        # chain storage): only the row offset moves - skip re-marshalling (a dozen ctypes field stores, ~4 us)
        same_trace = (trace is not None and trace is self._last_trace[0] and trace_logp is self._last_trace[1]
                      and trace_every == self._last_trace[2] and ext_prop is None and ext_u is None and ext_swap_u is None
                      and accept_flags is None)
        if same_trace:
            te = a.trace_every
            if trace.shape[0] < trace_row0 + ((step0 + n_steps) // te - step0 // te):
                raise ValueError("trace must be [rows >= trace_row0 + traced steps, trace_chains, trace_temps, dim]")
            a.trace_row0 = trace_row0
        elif not (plain and self._plain):
            self._last_trace = (None, None, None)
            Cn, T, D = self.shape
            a.ext_prop = _opt(ext_prop, "ext_prop", self.state_dtype)  # (the state's dtype: double normals for double states)
            a.ext_u = _opt(ext_u, "ext_u", torch.float32)
            a.ext_swap_u = _opt(ext_swap_u, "ext_swap_u", torch.float32)
            if ext_prop is not None:
                raw = ext_raw_per_step(self.proposal_kind, D)
                if (tuple(ext_prop.shape) != (n_steps, Cn, T, raw) or ext_u is None
                        or tuple(ext_u.shape) != (n_steps, Cn, T)):
                    raise ValueError("ext_prop/ext_u shapes do not match [n_steps, n_chains, n_temps, raw]")
            a.trace = _opt(trace, "trace", self.state_dtype)
            a.trace_logp = _opt(trace_logp, "trace_logp", torch.float32)
            a.trace_every, a.trace_chains, a.trace_temps = 0, 0, 0
            if trace is not None:
                te = max(1, int(trace_every))
                rows = (step0 + n_steps) // te - step0 // te  # steps of this call whose step_counter is a multiple of te
                if trace.dim() != 4 or trace.shape[3] != D or trace.shape[0] < trace_row0 + rows:
                    raise ValueError("trace must be [rows >= trace_row0 + traced steps, trace_chains, trace_temps, dim]")
                a.trace_every = te
                a.trace_chains = trace.shape[1]
                a.trace_temps = trace.shape[2]
            a.trace_row0 = trace_row0
            if accept_flags is not None and tuple(accept_flags.shape) != (n_steps, Cn, T):
                raise ValueError("accept_flags must be [n_steps, n_chains, n_temps]")
            a.accept_flags = _opt(accept_flags, "accept_flags", torch.uint8)
            self._plain = plain
            if trace is not None and ext_prop is None and ext_u is None and ext_swap_u is None and accept_flags is None:
                self._last_trace = (trace, trace_logp, trace_every)
        with self._guard:
            rc = self._lib.ptrwm_run(self._refs[2], self._refs[3], self._refs[4], _stream(self.device))
        if rc != 0:
            raise PTRWMError(rc, "ptrwm_run")

    def _split_buffers(self):
        if getattr(self, "_split", None) is None:
            Cn, T, D = self.shape
            # proposals, and the two planes of the accept scratch (uniforms; the proposal's own squared jump or -1)
            self._split = (torch.empty(Cn, T, D, device=self.device, dtype=torch.float32),
                           torch.empty(2, Cn, T, device=self.device, dtype=torch.float32))
        return self._split

    def set_device_step(self, counter: Optional[torch.Tensor]) -> None:
        """Device-step mode of the split-step calls (include/ptrwm.h `device_step`): ``counter`` is a one-element int64
        device tensor; split_propose / split_accept then perform step ``counter + step`` - their ``step`` argument is an
        offset baked into the call - so the argument lists of a block of steps do not depend on where the run stands and
        the block can be captured in a HIP graph (``split_advance(n)`` adds n to the counter as its last node).  ``None``
        switches it off."""
        if counter is not None:
            if counter.numel() != 1 or counter.device != self.device:
                raise ValueError("device_step must be a one-element int64 tensor on the run's device")
            self._a.device_step = _require_device(counter, "device_step", torch.int64)
        else:
            self._a.device_step = None
        self._device_step = counter  # (kept alive)

    def split_advance(self, n: int = 1) -> None:
        """*device_step += n on the current stream (ptrwm_split_advance)."""
        if n < 1:
            raise ValueError("split_advance: n >= 1")
        self._a.n_steps = n
        with self._guard:
            rc = self._lib.ptrwm_split_advance(self._refs[4], _stream(self.device))
        if rc != 0:
            raise PTRWMError(rc, "ptrwm_split_advance")

    def split_propose(self, step: int, ext_prop: Optional[torch.Tensor] = None,
                      ext_u: Optional[torch.Tensor] = None) -> torch.Tensor:
        """First half of a split step (ptrwm_split_propose): returns the proposals [C, T, D] of global step ``step``
        (a buffer owned by the plan, overwritten by the next call).  The caller evaluates its log-density on them and
        passes the result to ``split_accept``."""
        a = self._a
        Cn, T, D = self.shape
        if self.state_dtype != torch.float32:
            raise TypeError("split steps take float32 states (float64 states: the fused kernel only)")
        props, acc_u = self._split_buffers()
        a.step0 = step
        self._last_trace = (None, None, None)
        a.ext_prop = _opt(ext_prop, "ext_prop", torch.float32)
        a.ext_u = _opt(ext_u, "ext_u", torch.float32)
        if ext_prop is not None:
            raw = ext_raw_per_step(self.proposal_kind, D)
            if tuple(ext_prop.shape) != (Cn, T, raw) or ext_u is None or tuple(ext_u.shape) != (Cn, T):
                raise ValueError("ext_prop / ext_u of one step must be [n_chains, n_temps, raw] / [n_chains, n_temps]")
        self._plain = False
        with self._guard:
            rc = self._lib.ptrwm_split_propose(self._refs[3], self._refs[4], D, props.data_ptr(), acc_u.data_ptr(),
                                               _stream(self.device))
        if rc != 0:
            raise PTRWMError(rc, "ptrwm_split_propose")
        return props

    def split_accept(self, step: int, logp_proposed: torch.Tensor, ext_swap_u: Optional[torch.Tensor] = None,
                     accept_flags: Optional[torch.Tensor] = None, swap_event_offset: int = 0, no_sweep: bool = False) -> None:
        """Second half (ptrwm_split_accept): Metropolis rule on ``logp_proposed`` [C, T], updates, and the swap event
        when ``step`` is a swap step.  ``no_sweep`` (device-step mode only, PTRWM_SPLIT_NO_SWEEP): the caller's assertion
        that this step is not a swap step - the swap kernel is then not enqueued at all."""
        a = self._a
        Cn, T, D = self.shape
        props, acc_u = self._split_buffers()
        if tuple(logp_proposed.shape) != (Cn, T):
            raise ValueError(f"logp_proposed must be [{Cn}, {T}]")
        if ext_swap_u is not None and tuple(ext_swap_u.shape) != (Cn, T - 1):
            raise ValueError(f"ext_swap_u of one event must be [{Cn}, {T - 1}]")
        if accept_flags is not None and tuple(accept_flags.shape) != (Cn, T):
            raise ValueError(f"accept_flags of one step must be [{Cn}, {T}]")
        a.step0 = step
        a.swap_event_offset = swap_event_offset
        a.split_flags = 1 if no_sweep else 0
        self._last_trace = (None, None, None)
        a.ext_swap_u = _opt(ext_swap_u, "ext_swap_u", torch.float32)
        a.accept_flags = _opt(accept_flags, "accept_flags", torch.uint8)
        self._plain = False
        try:
            with self._guard:
                rc = self._lib.ptrwm_split_accept(self._refs[4], D, props.data_ptr(), acc_u.data_ptr(),
                                                  _require_device(logp_proposed, "logp_proposed", torch.float32),
                                                  _stream(self.device))
        finally:
            a.split_flags = 0  # (every other entry point requires 0)
        if rc != 0:
            raise PTRWMError(rc, "ptrwm_split_accept")

    def swap_sweep(self, rng_step: int, event_index: int, rng_stream: int = 2,
                   ext_swap_u: Optional[torch.Tensor] = None) -> None:
        """One stand-alone swap event over the current states (the reference's ``_attempt_all_swaps()`` called on
        its own).  Swap uniforms: ``ext_swap_u`` [C, T-1], else Philox stream ``rng_stream`` at step ``rng_step``
        (stream 1 = the stream the fused kernel's own swap events use)."""
        a = self._a
        Cn, T, D = self.shape
        if ext_swap_u is not None and tuple(ext_swap_u.shape) != (Cn, T - 1):
            raise ValueError(f"ext_swap_u must be [{Cn}, {T - 1}]")
        if a.device_step:
            raise RuntimeError("device-step mode is for split steps only: set_device_step(None) before swap_sweep()")
        a.step0 = rng_step
        self._last_trace = (None, None, None)
        a.ext_swap_u = _opt(ext_swap_u, "ext_swap_u", torch.float32)
        self._plain = False  # per-launch fields of _a were touched: the next launch() rewrites them
        with self._guard:
            rc = self._lib.ptrwm_swap_sweep(self._refs[4], D, event_index, rng_stream, _stream(self.device))
        if rc != 0:
            raise PTRWMError(rc, "ptrwm_swap_sweep")


def run(
    target: Target,
    proposal: Proposal,
    *,
    state: torch.Tensor,  # [C, T, D] float32 (or float64: states, trace and ext_prop in double, include/ptrwm.h state_f64)
    logp: torch.Tensor,  # [C, T] float32
    beta: torch.Tensor,  # [T] float32
    step0: int,
    n_steps: int,
    burn_in: int = 0,
    swap_every: int = 1,
    swap_mode: int = SWAP_EXCHANGE,
    swap_order: int = ORDER_SEQUENTIAL,
    seed: int = 0,
    chain_offset: int = 0,
    n_accept: Optional[torch.Tensor] = None,  # [C, T] int64
    sq_jump: Optional[torch.Tensor] = None,  # [C, T] float64
    swap_accept: Optional[torch.Tensor] = None,  # [C, T] int64
    last_swap_ordinal: Optional[torch.Tensor] = None,  # [C, T] int64
    **per_launch,
) -> None:
    """One-shot form: enqueue ``n_steps`` fused MH(+swap) steps for every (chain, temperature) replica.
    ``per_launch``: ext_prop, ext_u, ext_swap_u, trace, trace_logp, trace_row0, trace_every, accept_flags."""
    RunPlan(target, proposal, state=state, logp=logp, beta=beta, burn_in=burn_in, swap_every=swap_every,
            swap_mode=swap_mode, swap_order=swap_order, seed=seed, chain_offset=chain_offset, n_accept=n_accept,
            sq_jump=sq_jump, swap_accept=swap_accept, last_swap_ordinal=last_swap_ordinal).launch(
        step0, n_steps, **per_launch)
