"""Multi-GPU layout of a run: independent replicas sharded over ranks, one process per GPU.

The reference has no distributed path at all.  Replicas (chains / whole ladders) are
independent, so the sampling itself needs NO collective: rank r owns a contiguous block of global replica
ids and passes its first id as `chain_offset`, which is the Philox subsequence -- results do not depend on
the number of GPUs.  The only exchange is one tiny all-reduce(SUM) of the summary counters at the end
(RCCL over xGMI with backend "nccl"; latency-bound, a few hundred bytes).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world_size: int) -> Tuple[int, int]:
    """(offset, count) of rank's contiguous block when n_total replicas are split as evenly as possible
    (the first n_total % world_size ranks get one extra)."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    if n_total < 0:
        raise ValueError("n_total must be >= 0")
    base, extra = divmod(n_total, world_size)
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def pack_summary(summary: dict, device) -> torch.Tensor:
    """EngineRun.summary() -> one float64 vector [4T + 3] (exact for counts < 2^53)."""
    parts = [
        summary["accept_count"].double(),
        summary["sq_jump_sum"].double(),
        summary["swap_accept_count"].double(),
        torch.tensor([summary["n_replicas"], summary["swap_attempts"], summary["post_burn_steps"]], dtype=torch.float64),
    ]
    return torch.cat(parts).to(device)


def unpack_summary(vec: torch.Tensor, n_temps: int, world_size: int) -> dict:
    v = vec.detach().cpu()
    T = n_temps
    post = int(round(v[3 * T + 2].item() / world_size))  # identical on every rank
    n_rep = int(round(v[3 * T].item()))
    acc, sq, sw = v[:T], v[T:2 * T], v[2 * T:3 * T]
    denom = max(1, n_rep * post)
    attempts = int(round(v[3 * T + 1].item()))
    return {
        "n_replicas": n_rep,
        "post_burn_steps": post,
        "accept_count": acc.round().long(),
        "acceptance_rate": acc / denom,                 # per temperature
        "esjd": sq / denom,                             # per temperature
        "swap_accept_count": sw.round().long(),
        "swap_attempts": attempts,
        "swap_acceptance_rate": float(sw.sum() / attempts) if attempts else 0.0,
    }


def allreduce_summary(summary: dict, device, group: Optional[dist.ProcessGroup] = None) -> dict:
    """Whole-job acceptance / ESJD / swap statistics from every rank's shard (one all-reduce)."""
    n_temps = summary["accept_count"].numel()
    vec = pack_summary(summary, device)
    world = 1
    if dist.is_available() and dist.is_initialized():
        world = dist.get_world_size(group)
        dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)
    return unpack_summary(vec, n_temps, world)
