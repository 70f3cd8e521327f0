"""Uniform density on a hypercube (reference: target_distributions/hypercube_torch.py:5-113); evaluated by the HIP
engine (csrc/targets.h Hypercube)."""
import numpy as np
import torch

import ptrwm_hip
from interfaces.target_torch import TorchTargetDistribution


class HypercubeTorch(TorchTargetDistribution):
    """1 / (right - left)^dim on [left, right]^dim (boundaries included), 0 outside."""

    def __init__(self, dim, left_boundary=0.0, right_boundary=1.0, device=None):
        super().__init__(dim, device)
        self.name = "HypercubeTorch"
        self.left_boundary = torch.tensor(left_boundary, device=self.device, dtype=torch.float32)
        self.right_boundary = torch.tensor(right_boundary, device=self.device, dtype=torch.float32)
        volume = (right_boundary - left_boundary) ** dim
        self.uniform_density = torch.tensor(1.0 / volume, device=self.device, dtype=torch.float32)
        self.log_uniform_density = torch.log(self.uniform_density)

    def get_name(self):
        return self.name

    def engine_target(self):
        return ptrwm_hip.Target(ptrwm_hip.TARGET_HYPERCUBE, self.dim,
                                p=(float(self.left_boundary), float(self.right_boundary), float(self.log_uniform_density)))

    def log_density(self, x):
        return self._engine_log_density(x)

    def density(self, x):
        return torch.exp(self.log_density(x))

    def draw_sample(self, beta=1.0):
        return np.random.uniform(float(self.left_boundary), float(self.right_boundary), self.dim)

    def draw_samples_torch(self, n_samples, beta=1.0):
        u = torch.rand(n_samples, self.dim, device=self.device, dtype=torch.float32)
        return u * (self.right_boundary - self.left_boundary) + self.left_boundary

    def to(self, device):
        super().to(device)
        for attr in ("left_boundary", "right_boundary", "uniform_density", "log_uniform_density"):
            setattr(self, attr, getattr(self, attr).to(device))
        return self
