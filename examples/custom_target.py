"""A user-defined target density with the drop-in samplers.

    python examples/custom_target.py

The fused HIP kernel knows the reference's densities; anything else - here a ring-shaped density written in plain torch -
runs in split steps: the engine's HIP kernels draw the proposals (Philox), apply the Metropolis rule, update the
statistics and perform the temperature swaps, and call this class's `log_density` on the GPU once per step in between.
Nothing runs on the CPU."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rwm-pt-pytorch_amd"))
from algorithms import ParallelTemperingRWM_GPU_Optimized, geometric_beta_ladder  # noqa: E402
from interfaces import TorchTargetDistribution  # noqa: E402


class Ring(TorchTargetDistribution):
    """log p(x) = -(|x| - R)^2 / (2 w^2): mass on a thin sphere of radius R."""

    def __init__(self, dim, radius=4.0, width=0.25, device=None):
        super().__init__(dim, device)
        self.radius, self.width = radius, width

    def get_name(self):
        return "Ring"

    def log_density(self, x):
        x = torch.as_tensor(x, device=self.device, dtype=torch.float32)
        r = x.norm(dim=-1)
        return -0.5 * ((r - self.radius) / self.width) ** 2

    def density(self, x):
        return torch.exp(self.log_density(x))


if __name__ == "__main__":
    dev = torch.device("cuda")
    target = Ring(4, device=dev)
    pt = ParallelTemperingRWM_GPU_Optimized(4, 0.3, target, beta_ladder=geometric_beta_ladder(6, 0.05), swap_every=5,
                                            burn_in=500, device=dev, num_replicas=8192, seed=7, trace="cold")
    pt.generate_samples(1500)
    x = pt._run.state[:, 0]  # the cold replica of every ladder after 2000 steps
    print(f"mean radius {float(x.norm(dim=1).mean()):.3f} (target 4.0 +- 0.25), mean position "
          f"{[round(v, 2) for v in x.mean(0).tolist()]} (target 0), swap acceptance {pt.swap_acceptance_rate:.3f}, "
          f"MH acceptance per temperature {[round(v, 2) for v in pt.mh_acceptance_rates().tolist()]}")
