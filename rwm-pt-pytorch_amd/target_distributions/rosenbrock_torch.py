"""n-dimensional Rosenbrock targets (Pagani et al. 2022): Full, Even and Hybrid variants.

Parameter holders with the reference's constructors and names
(target_distributions/rosenbrock_torch.py:13-130, :133-256, :259-410); densities are evaluated by
the HIP engine (csrc/targets.h FullRosenbrock / EvenRosenbrock / HybridRosenbrock).
"""
from typing import Union

import numpy as np
import torch

import ptrwm_hip
from interfaces.target_torch import TorchTargetDistribution

DEFAULT_A_COEFF = 1.0 / 20.0
DEFAULT_B_COEFF = 100.0 / 20.0
DEFAULT_MU = 1.0
_F32 = torch.float32


def _mu_vector(mu, n, device):
    if isinstance(mu, (int, float)):
        return torch.full((n,), mu, device=device, dtype=_F32)
    if isinstance(mu, torch.Tensor):
        if mu.ndim == 0:
            return torch.full((n,), mu.item(), device=device, dtype=_F32)
        if mu.shape == (n,):
            return mu.to(device=device, dtype=_F32)
        raise ValueError(f"mu tensor must be scalar or have shape ({n},)")
    raise TypeError("mu must be float, int, or torch.Tensor")


class _RosenbrockBase(TorchTargetDistribution):
    def _init_coeffs(self, a_coeff, b_coeff):
        self.a_coeff = torch.tensor(a_coeff, device=self.device, dtype=_F32)
        self.b_coeff = torch.tensor(b_coeff, device=self.device, dtype=_F32)

    def log_density(self, x):
        return self._engine_log_density(x)

    def density(self, x):
        return torch.exp(self.log_density(x))

    def get_name(self):
        return self._name

    def draw_sample(self, beta=1.0):
        return self.draw_samples_torch(1, beta)[0].cpu().numpy()

    def to(self, device):
        super().to(device)
        self.a_coeff = self.a_coeff.to(device)
        self.b_coeff = self.b_coeff.to(device)
        self.mu = self.mu.to(device)
        return self

    def _std(self, coeff, beta):
        # x ~ N(m, 1 / (2 * coeff * beta)); unit variance when the effective coefficient is not positive
        eff = float(coeff) * beta
        return (1.0 / (2.0 * eff)) ** 0.5 if eff > 0 else 1.0


class FullRosenbrockTorch(_RosenbrockBase):
    """log p(x) = -sum_{i<n-1} [ b (x_{i+1} - x_i^2)^2 + a (x_i - mu_i)^2 ]"""

    def __init__(self, dim, a_coeff=DEFAULT_A_COEFF, b_coeff=DEFAULT_B_COEFF, mu: Union[float, torch.Tensor] = DEFAULT_MU,
                 device=None):
        if dim < 2:
            raise ValueError("Dimension for FullRosenbrockTorch must be at least 2.")
        super().__init__(dim, device)
        self._init_coeffs(a_coeff, b_coeff)
        self.mu = _mu_vector(mu, dim - 1, self.device)
        self._name = "FullRosenbrockTorch"

    def engine_target(self):
        return ptrwm_hip.Target(ptrwm_hip.TARGET_FULL_ROSENBROCK, self.dim,
                                p=(float(self.a_coeff), float(self.b_coeff)), vec0=self.mu.contiguous())

    def draw_samples_torch(self, n_samples, beta=1.0):
        raise NotImplementedError("Draw samples for FullRosenbrockTorch is not implemented yet.")


class EvenRosenbrockTorch(_RosenbrockBase):
    """log p(x) = -sum_{i<n/2} [ a (x_{2i} - mu_i)^2 + b (x_{2i+1} - x_{2i}^2)^2 ],  n even"""

    def __init__(self, dim, a_coeff=DEFAULT_A_COEFF, b_coeff=DEFAULT_B_COEFF, mu: Union[float, torch.Tensor] = DEFAULT_MU,
                 device=None):
        if dim < 2 or dim % 2 != 0:
            raise ValueError("Dimension for EvenRosenbrockTorch must be at least 2 and even.")
        super().__init__(dim, device)
        self._init_coeffs(a_coeff, b_coeff)
        self.mu = _mu_vector(mu, dim // 2, self.device)
        self._name = "EvenRosenbrockTorch"

    def engine_target(self):
        return ptrwm_hip.Target(ptrwm_hip.TARGET_EVEN_ROSENBROCK, self.dim,
                                p=(float(self.a_coeff), float(self.b_coeff)), vec0=self.mu.contiguous())

    def draw_samples_torch(self, n_samples, beta=1.0):
        """Exact sampler: x_{2i} ~ N(mu_i, 1/(2 a beta)), x_{2i+1} | x_{2i} ~ N(x_{2i}^2, 1/(2 b beta))."""
        half = self.dim // 2
        first = self.mu + torch.randn(n_samples, half, device=self.device, dtype=_F32) * self._std(self.a_coeff, beta)
        second = first**2 + torch.randn(n_samples, half, device=self.device, dtype=_F32) * self._std(self.b_coeff, beta)
        out = torch.empty(n_samples, self.dim, device=self.device, dtype=_F32)
        out[:, 0::2] = first
        out[:, 1::2] = second
        return out


class HybridRosenbrockTorch(_RosenbrockBase):
    """log p(x) = -a (x_0 - mu)^2 - b sum_j (x_{j,2} - x_0^2)^2 - b sum_j sum_{i=3..n1} (x_{j,i} - x_{j,i-1}^2)^2,
    dimension 1 + n2 (n1 - 1); blocks are stored one after another behind x_0."""

    def __init__(self, n1, n2, a_coeff=DEFAULT_A_COEFF, b_coeff=DEFAULT_B_COEFF, mu: float = DEFAULT_MU, device=None):
        if n1 < 2:
            raise ValueError("n1 (block length parameter) must be at least 2.")
        if n2 < 1:
            raise ValueError("n2 (number of blocks) must be at least 1.")
        super().__init__(1 + n2 * (n1 - 1), device)
        self.n1, self.n2 = n1, n2
        self._init_coeffs(a_coeff, b_coeff)
        self.mu = torch.tensor(mu, device=self.device, dtype=_F32)
        self._name = f"HybridRosenbrockTorch(n1={n1}, n2={n2}, a={a_coeff:.2f}, b={b_coeff:.2f}, mu={mu:.2f})"

    def engine_target(self):
        return ptrwm_hip.Target(ptrwm_hip.TARGET_HYBRID_ROSENBROCK, self.dim,
                                p=(float(self.a_coeff), float(self.b_coeff), float(self.mu)), ip=(self.n1, self.n2))

    def draw_samples_torch(self, n_samples, beta=1.0):
        """Algorithm 1 of the paper: x_0 first, then every block coordinate given its parent."""
        sb = self._std(self.b_coeff, beta)
        out = torch.empty(n_samples, self.dim, device=self.device, dtype=_F32)
        out[:, 0] = self.mu + torch.randn(n_samples, device=self.device, dtype=_F32) * self._std(self.a_coeff, beta)
        blk = self.n1 - 1
        for i in range(1, self.dim):
            parent = out[:, 0] if (i - 1) % blk == 0 else out[:, i - 1]
            out[:, i] = parent**2 + torch.randn(n_samples, device=self.device, dtype=_F32) * sb
        return out
