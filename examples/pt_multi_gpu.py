"""Whole-node PT-RWM run: independent ladders sharded over the GPUs of one node, one process per GPU.

    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/pt_multi_gpu.py --ladders 1048576

BASELINE.json configs[4] by default: ThreeMixture dim 50, UniformRadius proposal, 64 geometric temperatures,
1 048 576 ladders over the node, RCCL all-reduce of the acceptance / ESJD summaries at the end.  There is no
collective on the sampling path: rank r owns the global ladder ids [offset, offset + count) and passes `offset` as
the Philox subsequence base, so the result does not depend on the number of GPUs.
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rwm-pt-pytorch_amd"))

from algorithms import ParallelTemperingRWM_GPU_Optimized, geometric_beta_ladder  # noqa: E402
from algorithms.sharding import allreduce_summary, shard_range  # noqa: E402
from proposal_distributions import UniformRadiusProposal  # noqa: E402
from target_distributions import ThreeMixtureDistributionTorch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ladders", type=int, default=1048576, help="ladders over the whole job")
    ap.add_argument("--temps", type=int, default=64)
    ap.add_argument("--dim", type=int, default=50)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--burn-in", type=int, default=500)
    ap.add_argument("--seed", type=int, default=42)
    args = ap.parse_args()

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    offset, count = shard_range(args.ladders, rank, world)
    target = ThreeMixtureDistributionTorch(args.dim, device=dev)
    proposal = UniformRadiusProposal(args.dim, 2.4, 1.0, dev, torch.float32)
    pt = ParallelTemperingRWM_GPU_Optimized(
        args.dim, 2.38**2 / args.dim, target, beta_ladder=geometric_beta_ladder(args.temps), swap_every=10,
        burn_in=args.burn_in, device=dev, num_replicas=count, chain_offset=offset, seed=args.seed,
        proposal_distribution=proposal, trace="none")
    t0 = time.time()
    pt.generate_samples(args.steps)
    total = allreduce_summary(pt._run.summary(), dev)  # the only collective of the job
    dt = time.time() - t0
    if rank == 0:
        n = (args.steps + args.burn_in) * args.temps * args.ladders
        print(f"{world} GPU(s), {args.ladders} ladders x {args.temps} temps x {args.steps + args.burn_in} steps in {dt:.2f}s "
              f"= {n / dt:.3e} chain-MH-steps/s")
        print("MH acceptance per temperature:", [round(float(v), 3) for v in total["acceptance_rate"]][:8], "...")
        print(f"cold-chain ESJD {float(total['esjd'][0]):.4f}, swap acceptance {total['swap_acceptance_rate']:.4f}")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
