"""Neal's funnel (reference: target_distributions/funnel_torch.py:6-110 NealFunnelTorch); evaluated by the HIP engine
(csrc/targets.h NealFunnel).  SuperFunnelTorch (per-group data tensors, einsum) is out of scope."""
import math

import torch

import ptrwm_hip
from interfaces.target_torch import TorchTargetDistribution


class NealFunnelTorch(TorchTargetDistribution):
    """log p(v, z) = log N(v | mu_v, sigma_v^2) + sum_k log N(z_k | mu_z, exp(v)); x = (v, z_1 .. z_{D-1})."""

    def __init__(self, dim, mu_v=0.0, sigma_v_sq=9.0, mu_z=0.0, device=None):
        super().__init__(dim, device)
        if dim < 1:
            raise ValueError("dim must be at least 1 for Neal's Funnel (v variable).")
        if sigma_v_sq <= 0:
            raise ValueError("sigma_v_sq must be positive.")
        self.mu_v = torch.tensor(mu_v, device=self.device, dtype=torch.float32)
        self.sigma_v_sq = torch.tensor(sigma_v_sq, device=self.device, dtype=torch.float32)
        self.mu_z = torch.tensor(mu_z, device=self.device, dtype=torch.float32)
        self.log_sigma_v_sq = torch.log(self.sigma_v_sq)
        self.log_2_pi = torch.tensor(2.0 * math.pi, device=self.device, dtype=torch.float32).log()

    def get_name(self):
        return f"NealFunnelTorch_D{self.dim}"

    def engine_target(self):
        return ptrwm_hip.Target(ptrwm_hip.TARGET_NEAL_FUNNEL, self.dim,
                                p=(float(self.mu_v), float(self.sigma_v_sq), float(self.mu_z)))

    def log_density(self, x_tensor):
        return self._engine_log_density(x_tensor)

    def density(self, x):
        return torch.exp(self.log_density(x))

    def draw_sample(self, beta=1.0):
        raise NotImplementedError("NealFunnelTorch.draw_sample is not implemented.")

    def to(self, device):
        super().to(device)
        for attr in ("mu_v", "sigma_v_sq", "mu_z", "log_sigma_v_sq", "log_2_pi"):
            setattr(self, attr, getattr(self, attr).to(device))
        return self
