"""Device-side state of a sampler run and the calls into the fused HIP kernel.

Shared by `RandomWalkMH_GPU_Optimized` (one temperature) and `ParallelTemperingRWM_GPU_Optimized`
(a ladder).  Layout in HBM (all contiguous, row-major):

    state        float32 [n_replicas, n_temps, dim]   the replicas' current points
    logp         float32 [n_replicas, n_temps]        their log-densities
    n_accept     int64   [n_replicas, n_temps]        MH acceptances after burn-in
    sq_jump      float64 [n_replicas, n_temps]        sum of squared jump distances after burn-in
    swap_accept  int64   [n_replicas, n_temps]        accepted swaps of pair (t, t+1)
    last_ord     int64   [n_replicas, n_temps]        attempt ordinal of the pair's last accepted swap

`n_replicas` is the axis the reference does not have: independent copies of the whole chain /
ladder, one Philox subsequence each (global replica id = chain_offset + local index), so a run is
invariant to how replicas are sharded over GPUs.
"""
from __future__ import annotations

import warnings
from typing import Optional, Sequence

import numpy as np
import torch

import ptrwm_hip


def resolve_device(device) -> torch.device:
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    return torch.device(device)


def draw_seed() -> int:
    """A Philox key from torch's global CPU generator: `torch.manual_seed(s)` (which the harness calls
    after constructing the sampler, interfaces/simulation_gpu.py) therefore fixes the whole run."""
    return int(torch.randint(0, 2**62, (1,)).item())


class EngineRun:
    def __init__(self, *, target_dist, proposal: "ptrwm_hip.Proposal", beta_ladder: Sequence[float], dim: int,
                 device: torch.device, n_replicas: int, initial_state: np.ndarray, burn_in: int, swap_every: int,
                 swap_mode: str, swap_order: str, seed: Optional[int], chain_offset: int = 0,
                 dtype: torch.dtype = torch.float32):
        if swap_mode not in ptrwm_hip.SWAP_MODES:
            raise ValueError(f"swap_mode must be one of {sorted(ptrwm_hip.SWAP_MODES)}, got {swap_mode!r}")
        if swap_order not in ptrwm_hip.SWAP_ORDERS:
            raise ValueError(f"swap_order must be one of {sorted(ptrwm_hip.SWAP_ORDERS)}, got {swap_order!r}")
        if n_replicas < 1:
            raise ValueError("number of replicas must be >= 1")
        n_temps = len(beta_ladder)
        if not 1 <= n_temps <= ptrwm_hip.MAX_TEMPS:
            raise ValueError(f"the fused kernel keeps one ladder inside one workgroup: 1..{ptrwm_hip.MAX_TEMPS} "
                             f"temperatures, got {n_temps}")
        if not 1 <= dim <= ptrwm_hip.MAX_DIM:
            raise ValueError(f"dim must be in 1..{ptrwm_hip.MAX_DIM} for the fused kernel, got {dim}")
        if device.type != "cuda":
            raise RuntimeError(
                "The PT-RWM engine runs only on a ROCm GPU (device='cuda'); there is no CPU fallback. "
                f"Requested device: {device}"
            )
        # Targets the fused kernel knows describe themselves (engine_target).  Any other TorchTargetDistribution - a
        # user-defined density, a dense-covariance Gaussian - runs in split steps: HIP kernels for proposal, accept and
        # swap (same Philox streams, same arithmetic) around one device-side `log_density` call per step
        # (the reference calls target.log_density on the proposals the same way, pt_rwm_gpu_optimized.py:551).
        self.density_fn = None
        try:
            self.target = target_dist.engine_target()
        except (NotImplementedError, AttributeError):
            if not callable(getattr(target_dist, "log_density", None)):
                raise TypeError(f"{type(target_dist).__name__} has neither engine_target() nor log_density()")
            self.target = None
            self.density_fn = target_dist.log_density
            warnings.warn(f"{type(target_dist).__name__} has no fused kernel: running split steps (HIP proposal / "
                          "accept / swap kernels around its log_density, three launches per step).")
        if self.target is not None and not (ptrwm_hip.has_thread_variant(self.target.kind, proposal.kind, dim)
                                            or ptrwm_hip.has_quad_variant(self.target.kind, proposal.kind, dim, n_temps)):
            raise ValueError(f"no fused kernel for dim {dim} with a ladder of {n_temps} temperatures: above dim 64 a ladder "
                             "holds at most 128 temperatures (one 512-thread workgroup of the lane-split kernel)")
        self.proposal = proposal
        self.dim, self.n_temps, self.n_replicas = dim, n_temps, n_replicas
        self.device = device
        self.burn_in, self.swap_every = int(burn_in), int(swap_every)
        self.swap_mode = ptrwm_hip.SWAP_MODES[swap_mode]
        self.swap_order = ptrwm_hip.SWAP_ORDERS[swap_order]
        self.seed = draw_seed() if seed is None else int(seed)
        self.chain_offset = int(chain_offset)
        self.steps_done = 0
        self.manual_sweeps = 0  # stand-alone swap events (swap_sweep) performed so far
        self.use_graph = True   # split steps: replay a captured HIP graph where no per-step trace is asked for
        self._graph = None
        self.beta = torch.tensor(list(beta_ladder), device=device, dtype=torch.float32)
        if dtype not in (torch.float32, torch.float64):
            raise TypeError(f"state dtype must be torch.float32 or torch.float64, got {dtype}")
        if dtype == torch.float64 and self.density_fn is not None:
            raise NotImplementedError("dtype=torch.float64 needs a target with a fused kernel (split steps carry float32 "
                                      "states)")
        self.dtype = dtype  # float64: the engine's state_f64 mode (the reference's dtype=torch.float64)
        x0 = torch.as_tensor(np.asarray(initial_state), dtype=dtype).to(device)
        # every temperature (and replica) starts from the same point (pt_rwm_gpu_optimized.py:478-484)
        self.state = x0.expand(n_replicas, n_temps, dim).contiguous()
        # (log-densities are float32 in either mode: the density kernels evaluate the state rounded to float)
        self.logp = self._density(self.state.view(-1, dim).to(torch.float32)).view(n_replicas, n_temps).contiguous()
        shape = (n_replicas, n_temps)
        self.n_accept = torch.zeros(shape, device=device, dtype=torch.int64)
        self.sq_jump = torch.zeros(shape, device=device, dtype=torch.float64)
        self.swap_accept = torch.zeros(shape, device=device, dtype=torch.int64)
        self.last_ord = torch.zeros(shape, device=device, dtype=torch.int64)
        # everything ptrwm_run needs that stays fixed for this run, marshalled once (the tensors above are never
        # re-allocated: the kernel updates them in place)
        self._plan = ptrwm_hip.RunPlan(
            self.target, self.proposal, state=self.state, logp=self.logp, beta=self.beta, burn_in=self.burn_in,
            swap_every=self.swap_every, swap_mode=self.swap_mode, swap_order=self.swap_order, seed=self.seed,
            chain_offset=self.chain_offset, n_accept=self.n_accept, sq_jump=self.sq_jump,
            swap_accept=self.swap_accept, last_swap_ordinal=self.last_ord)

    def _density(self, rows: torch.Tensor) -> torch.Tensor:
        """log-density of every row [n, dim] -> float32 [n] on the device."""
        if self.density_fn is None:
            return ptrwm_hip.logdensity(self.target, rows)
        out = self.density_fn(rows)
        if not torch.is_tensor(out) or out.shape != (rows.shape[0],):
            raise ValueError("log_density must map a [n, dim] device tensor to a [n] tensor")
        if not out.is_cuda:
            raise RuntimeError("log_density returned a CPU tensor: a split-step target must evaluate on the GPU "
                               "(there is no CPU path)")
        return out.to(torch.float32).contiguous()

    # ---- stepping ---------------------------------------------------------------------------
    def traced_rows(self, n_steps: int, every: int) -> int:
        """Rows a trace with thinning period `every` receives from the next n_steps steps."""
        return (self.steps_done + n_steps) // every - self.steps_done // every

    def advance(self, n_steps: int, trace: Optional[torch.Tensor] = None, trace_logp: Optional[torch.Tensor] = None,
                trace_row0: int = 0, trace_every: int = 1) -> None:
        """Enqueue n_steps fused steps (no host synchronisation)."""
        if n_steps <= 0:
            return
        if self.density_fn is not None:
            self._advance_split(n_steps, trace, trace_logp, trace_row0, max(1, int(trace_every)))
            return
        if trace is None and trace_logp is None:
            self._plan.launch(self.steps_done, n_steps, swap_event_offset=self.manual_sweeps)
        else:
            self._plan.launch(self.steps_done, n_steps, trace=trace, trace_logp=trace_logp, trace_row0=trace_row0,
                              trace_every=trace_every, swap_event_offset=self.manual_sweeps)
        self.steps_done += n_steps

    # ---- split steps through a captured HIP graph ------------------------------------------------------------
    # A split step is four launches of this library (proposal, Metropolis rule, swap event, step counter) around the
    # caller's density evaluation.  Issued one by one from Python they cost two ctypes calls plus the density's own torch
    # dispatch per step - tens of microseconds of host time against a few microseconds of kernels for a small batch.  In
    # device-step mode (include/ptrwm.h `device_step`) no argument of those launches depends on the step, so GRAPH_STEPS
    # steps are captured ONCE with torch.cuda.CUDAGraph - the density's kernels included - and replayed.  Same kernels,
    # same Philox words: bit-identical to the eager loop and (with the library's own density) to ptrwm_run.
    GRAPH_STEPS = 16

    def _split_step_on_device_counter(self, offset: int = 0, no_sweep: bool = False, advance: int = 1) -> None:
        """One split step at device step counter + offset; `advance` > 0 adds that much to the counter afterwards."""
        C, T, D = self.n_replicas, self.n_temps, self.dim
        props = self._plan.split_propose(offset)
        lp_new = self._density(props.view(-1, D)).view(C, T)
        self._plan.split_accept(offset, lp_new, swap_event_offset=self.manual_sweeps, no_sweep=no_sweep)
        if advance:
            self._plan.split_advance(advance)

    def _graph_block(self) -> int:
        """Steps per captured block: a multiple of swap_every (>= GRAPH_STEPS) where that is short enough, so that a block
        replayed from a counter that is a multiple of swap_every has its swap steps at FIXED offsets and the swap kernel
        is enqueued only there; GRAPH_STEPS otherwise (the swap kernel then rides with every step and decides on the
        device)."""
        se = int(self.swap_every)
        if self.n_temps < 2 or se > 4 * self.GRAPH_STEPS:
            return self.GRAPH_STEPS
        return se * -(-self.GRAPH_STEPS // se)

    def _advance_split_graph(self, n_steps: int) -> bool:
        """n_steps split steps by graph replay; False if the density cannot be captured (the caller falls back)."""
        if getattr(self, "_graph_failed", False):
            return False
        if getattr(self, "_dstep", None) is None:
            self._dstep = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._dstep.fill_(self.steps_done)
        self._plan.set_device_step(self._dstep)
        block = self._graph_block()
        aligned = self.n_temps >= 2 and block % int(self.swap_every) == 0  # swap steps sit at fixed offsets of a block
        se = int(self.swap_every)
        try:
            done = 0

            def eager(k):  # k steps, one at a time, on the device counter
                for _ in range(k):
                    self._split_step_on_device_counter()

            key = (self.manual_sweeps, block)  # (baked into the captured launches)
            if getattr(self, "_graph_key", None) != key:
                self._graph = None
            if self._graph is None:
                # one step outside capture first, on a side stream (lazy initialisations of whatever the density calls
                # must not happen while capturing); it is a real step of the run
                cur = torch.cuda.current_stream(self.device)
                side = torch.cuda.Stream(self.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    self._split_step_on_device_counter()
                cur.wait_stream(side)
                done = 1
            if aligned:  # replay only from counters that are multiples of swap_every
                k = min(n_steps - done, (-(self.steps_done + done)) % se)
                eager(k)
                done += k
            if self._graph is None and n_steps - done >= block:
                g = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(g):
                        for j in range(block):
                            # step counter + j has step_counter = counter + j + 1: a swap step iff (j + 1) % swap_every == 0
                            self._split_step_on_device_counter(offset=j, no_sweep=aligned and (j + 1) % se != 0,
                                                               advance=block if j == block - 1 else 0)
                except Exception as e:  # a density that synchronises, allocates outside the pool, ...
                    self._graph_failed = True
                    torch.cuda.synchronize(self.device)
                    warnings.warn(f"split steps: the density could not be captured in a HIP graph ({type(e).__name__}: {e}); "
                                  "running step by step")
                    self._dstep.fill_(self.steps_done + done)
                    eager(n_steps - done)
                    self.steps_done += n_steps
                    return True
                self._graph, self._graph_key = g, key
            while self._graph is not None and n_steps - done >= block:
                self._graph.replay()
                done += block
            eager(n_steps - done)
            self.steps_done += n_steps
            return True
        finally:
            self._plan.set_device_step(None)

    def _advance_split(self, n_steps, trace, trace_logp, trace_row0, trace_every) -> None:
        """n_steps split steps; traced steps (step_counter a multiple of trace_every) are copied with torch."""
        if trace is None and n_steps >= 2 and self.use_graph and self._advance_split_graph(n_steps):
            return
        C, T, D = self.n_replicas, self.n_temps, self.dim
        row = trace_row0
        for _ in range(n_steps):
            s = self.steps_done
            props = self._plan.split_propose(s)
            lp_new = self._density(props.view(-1, D)).view(C, T)
            self._plan.split_accept(s, lp_new, swap_event_offset=self.manual_sweeps)
            self.steps_done += 1
            if trace is not None and self.steps_done % trace_every == 0:
                tc, tt = trace.shape[1], trace.shape[2]
                trace[row] = self.state[:tc, :tt]
                if trace_logp is not None:
                    trace_logp[row] = self.logp[:tc, :tt]
                row += 1

    def swap_sweep(self) -> None:
        """One stand-alone swap event over the current states (`_attempt_all_swaps()` called on its own,
        pt_rwm_gpu_optimized.py:594-633; no host synchronisation).  Its uniforms come from Philox stream 2 at
        counter = number of stand-alone sweeps so far, so they never coincide with the fused kernel's."""
        if self.n_temps < 2:
            return
        self._plan.swap_sweep(rng_step=self.manual_sweeps, event_index=self.swap_events(), rng_stream=2)
        self.manual_sweeps += 1

    # ---- summaries (each read synchronises) ---------------------------------------------------
    @property
    def post_burn_steps(self) -> int:
        return max(0, self.steps_done - self.burn_in)

    def swap_events(self) -> int:
        e = self.steps_done // self.swap_every - self.burn_in // self.swap_every
        return max(0, e) + self.manual_sweeps if self.n_temps > 1 else 0

    def swap_attempts_per_replica(self) -> int:
        """Swap attempts one ladder has made so far (deterministic, needs no device read)."""
        ev, T = self.swap_events(), self.n_temps
        if self.swap_order == ptrwm_hip.ORDER_SEQUENTIAL:
            return ev * (T - 1)
        n_even, n_odd = (T - 1 + 1) // 2, (T - 1) // 2  # pairs with even / odd lower index
        # events are numbered from 0: even-numbered events take the even pairs
        return ((ev + 1) // 2) * n_even + (ev // 2) * n_odd

    def summary(self) -> dict:
        """Whole-shard sums, as plain Python numbers / CPU tensors (one device sync)."""
        acc = self.n_accept.sum(0).cpu()
        sq = self.sq_jump.sum(0).cpu()
        sw = self.swap_accept.sum(0).cpu()
        return {
            "n_replicas": self.n_replicas,
            "post_burn_steps": self.post_burn_steps,
            "accept_count": acc,            # [T] int64
            "sq_jump_sum": sq,              # [T] float64
            "swap_accept_count": sw,        # [T] int64 (last entry unused)
            "swap_attempts": self.swap_attempts_per_replica() * self.n_replicas,
        }
