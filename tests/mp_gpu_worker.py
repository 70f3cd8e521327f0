"""Child process of tests/test_gpu_api.py::test_two_gpu_processes_equal_one (NOT a test module): one rank of a
world_size-2 job whose ranks share cuda:0 and talk over gloo - the same code path bench.py takes over RCCL with one
rank per GPU.  Each rank owns the global ladder ids [rank * C, (rank + 1) * C) (chain_offset = Philox subsequence),
runs its shard through the drop-in class, all-reduces the summary and writes state + summary to an .npz."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rwm-pt-pytorch_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def build(device, n_ladders, offset):
    from algorithms import ParallelTemperingRWM_GPU_Optimized, geometric_beta_ladder
    from target_distributions import RoughCarpetDistributionTorch

    target = RoughCarpetDistributionTorch(30, device=device, mode_centers=[-15.0, 0.0, 15.0])
    return ParallelTemperingRWM_GPU_Optimized(30, 2.38**2 / 30, target, beta_ladder=geometric_beta_ladder(8), swap_every=5,
                                              burn_in=7, device=device, num_replicas=n_ladders, seed=20260,
                                              chain_offset=offset, trace="none")


def main():
    rank, world, port, C, steps, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    backend = sys.argv[7] if len(sys.argv) > 7 else "gloo"  # "nccl" = RCCL (one rank per GPU; here: world size 1)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    from algorithms.sharding import allreduce_summary

    dev = torch.device("cuda:0")  # NO torch.cuda.set_device: the binding's device guard must do it
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    alg = build(dev, C, rank * C)
    alg._advance(steps)
    torch.cuda.synchronize()
    total = allreduce_summary(alg._run.summary(), dev if backend == "nccl" else torch.device("cpu"))
    np.savez(out, state=alg._run.state.cpu().numpy(), logp=alg._run.logp.cpu().numpy(),
             n_accept=alg._run.n_accept.cpu().numpy(),
             **{f"sum_{k}": (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in total.items()})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
