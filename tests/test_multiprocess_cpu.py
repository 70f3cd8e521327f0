"""The N > 1 path on CPU: two processes over gloo own disjoint blocks of global chain ids, run their shards
independently (oracle in place of the GPU kernel) and all-reduce the summary -- the same code path bench.py
takes with backend nccl (RCCL).  Result must equal the single-process run over all chains."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H
from algorithms.sharding import allreduce_summary, shard_range
from oracle import oracle as O

N_CHAINS, T, DIM, STEPS, BURN, SE, SEED = 10, 4, 10, 60, 10, 5, 99


def _shard_summary(offset, count):
    spec = H.target_spec("rc15s_d10")
    beta = np.array([1.0, 0.5, 0.2, 0.05], np.float32)
    prop = H.proposal_spec("Normal", DIM, beta, base_variance_scalar=0.5)
    st = np.zeros((count, T, DIM), np.float32)
    lp = np.tile(O.logdensity(spec.oracle(), np.zeros((1, DIM), np.float32)).astype(np.float32), (count, T))
    r = O.run(spec.oracle(), prop.oracle(), state=st, logp=lp, beta=beta, step0=0, n_steps=STEPS, burn_in=BURN,
              swap_every=SE, seed=SEED, chain_offset=offset)
    events = STEPS // SE - BURN // SE
    return {"n_replicas": count, "post_burn_steps": STEPS - BURN, "swap_attempts": events * (T - 1) * count,
            "accept_count": torch.tensor(r["n_accept"].sum(0)), "sq_jump_sum": torch.tensor(r["sq_jump"].sum(0)),
            "swap_accept_count": torch.tensor(r["swap_accept"].sum(0))}, r["state"]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off, cnt = shard_range(N_CHAINS, rank, world)
    summary, state = _shard_summary(off, cnt)
    total = allreduce_summary(summary, torch.device("cpu"))
    q.put((rank, off, state, {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in total.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    single, state = _shard_summary(0, N_CHAINS)
    want = allreduce_summary(single, torch.device("cpu"))  # no process group: identity
    assert np.array_equal(np.concatenate([g[2] for g in got]), state)  # sharding is invisible in the states
    for _, _, _, total in got:  # every rank holds the whole-job summary
        assert total["n_replicas"] == N_CHAINS and total["post_burn_steps"] == STEPS - BURN
        assert total["swap_attempts"] == want["swap_attempts"]
        assert np.array_equal(total["accept_count"], want["accept_count"].numpy())
        np.testing.assert_allclose(total["esjd"], want["esjd"].numpy(), rtol=1e-12)
        assert total["swap_acceptance_rate"] == want["swap_acceptance_rate"]
