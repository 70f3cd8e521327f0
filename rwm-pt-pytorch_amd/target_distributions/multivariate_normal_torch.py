"""Gaussian targets with diagonal structure.

Parameter holders with the reference's constructors and names
(target_distributions/multivariate_normal_torch.py:5-131 MultivariateNormalTorch, :134-295
ScaledMultivariateNormalTorch); densities are evaluated by the HIP engine (csrc/targets.h DiagGaussian).
The fused kernel implements the diagonal-covariance case; a dense covariance is a [D,D] contraction per proposal
(a library GEMM) and runs through the engine's split steps like any user-defined density.
"""
import numpy as np
import torch

import ptrwm_hip
from interfaces.target_torch import TorchTargetDistribution

_F32 = torch.float32


class MultivariateNormalTorch(TorchTargetDistribution):
    """N(mean, cov); the kernel requires cov to be diagonal (the default is the identity)."""

    def __init__(self, dim, mean=None, cov=None, device=None):
        super().__init__(dim, device)
        self.name = "MultivariateNormalTorch"
        self.mean = (torch.zeros(dim, device=self.device, dtype=_F32) if mean is None
                     else torch.as_tensor(mean, dtype=_F32).to(self.device))
        self.cov = (torch.eye(dim, device=self.device, dtype=_F32) if cov is None
                    else torch.as_tensor(cov, dtype=_F32).to(self.device))
        if self.mean.shape != (dim,) or self.cov.shape != (dim, dim):
            raise ValueError(f"mean must have shape ({dim},) and cov ({dim}, {dim})")
        self.cov_inv = torch.linalg.inv(self.cov)
        self.cov_det = torch.linalg.det(self.cov)
        log_2pi = torch.log(torch.tensor(2.0 * torch.pi, device=self.device, dtype=_F32))
        self.log_norm_const = -0.5 * (dim * log_2pi + torch.log(self.cov_det))

    def get_name(self):
        return self.name

    def engine_target(self):
        if not self._is_diagonal():
            raise NotImplementedError(
                "MultivariateNormalTorch with a non-diagonal covariance has no fused-kernel implementation "
                "(a dense [D,D] contraction per proposal); the samplers run it with split steps."
            )
        return ptrwm_hip.Target(ptrwm_hip.TARGET_DIAG_GAUSSIAN, self.dim, p=(float(self.log_norm_const),), ip=(0,),
                                vec0=self.mean.contiguous(), vec1=torch.diagonal(self.cov_inv).contiguous())

    def _is_diagonal(self):
        # decided once per covariance (a device -> host read): log_density is called once per step by the split-step path,
        # which must neither synchronise nor break a HIP-graph capture
        key = (self.cov.data_ptr(), self.cov._version)
        if getattr(self, "_diag_key", None) != key:
            off_diag = self.cov - torch.diag(torch.diagonal(self.cov))
            self._diag_cached = not bool((off_diag != 0).any())
            self._diag_key = key
        return self._diag_cached

    def log_density(self, x):
        if self._is_diagonal():
            return self._engine_log_density(x)
        # dense covariance (multivariate_normal_torch.py:58-93 in the reference): one [B, D] x [D, D] library GEMM on
        # the device; the samplers reach it through split steps (ptrwm_split_propose / ptrwm_split_accept)
        if not torch.is_tensor(x):
            x = torch.as_tensor(x)
        x = x.to(device=self.device, dtype=_F32)
        single = x.dim() == 1
        c = (x.unsqueeze(0) if single else x) - self.mean
        out = -0.5 * torch.sum((c @ self.cov_inv) * c, dim=1) + self.log_norm_const
        return out[0] if single else out

    def density(self, x):
        return torch.exp(self.log_density(x))

    def draw_sample(self, beta=1.0):
        return np.random.multivariate_normal(self.mean.cpu().numpy(), self.cov.cpu().numpy() / beta)

    def draw_samples_torch(self, n_samples, beta=1.0):
        z = torch.randn(n_samples, self.dim, device=self.device, dtype=_F32)
        return self.mean + z @ torch.linalg.cholesky(self.cov / beta).T

    def to(self, device):
        super().to(device)
        for attr in ("mean", "cov", "cov_inv", "cov_det", "log_norm_const"):
            setattr(self, attr, getattr(self, attr).to(device))
        return self


class ScaledMultivariateNormalTorch(TorchTargetDistribution):
    """pi(x) = prod_i c_i N(c_i x_i | 0, 1): independent coordinates with standard deviations 1/c_i."""

    def __init__(self, dim, scaling_factors=None, scaling_range=(0.02, 1.98), device=None, seed=None):
        super().__init__(dim, device)
        self.name = "ScaledMultivariateNormalTorch"
        if seed is not None:
            torch.manual_seed(seed)
        if scaling_factors is not None:
            self.scaling_factors = torch.as_tensor(scaling_factors, dtype=_F32).clone().detach().to(self.device)
        else:
            lo, hi = scaling_range
            self.scaling_factors = torch.rand(dim, device=self.device, dtype=_F32) * (hi - lo) + lo
        assert self.scaling_factors.shape == (dim,), f"Scaling factors must have shape ({dim},), got {self.scaling_factors.shape}"
        log_2pi = torch.log(torch.tensor(2.0 * torch.pi, device=self.device, dtype=_F32))
        self.log_norm_const = torch.sum(torch.log(self.scaling_factors)) - 0.5 * self.dim * log_2pi

    def get_name(self):
        return self.name

    def engine_target(self):
        return ptrwm_hip.Target(ptrwm_hip.TARGET_DIAG_GAUSSIAN, self.dim, p=(float(self.log_norm_const),), ip=(1,),
                                vec0=self.scaling_factors.contiguous())

    def log_density(self, x):
        return self._engine_log_density(x)

    def density(self, x):
        return torch.exp(self.log_density(x))

    def draw_samples_torch(self, n_samples, beta=1.0):
        z = torch.randn(n_samples, self.dim, device=self.device, dtype=_F32)
        return z / (self.scaling_factors * float(beta) ** 0.5)

    def draw_sample(self, beta=1.0):
        c = self.scaling_factors.cpu().numpy()
        return np.array([np.random.normal(0.0, 1.0 / (c[i] * np.sqrt(beta))) for i in range(self.dim)])

    def get_scaling_factors(self):
        return self.scaling_factors.clone()

    def get_variances(self):
        return 1.0 / (self.scaling_factors ** 2)

    def get_diagonal_covariance_matrix(self):
        return torch.diag(self.get_variances())

    def to(self, device):
        super().to(device)
        self.scaling_factors = self.scaling_factors.to(device)
        self.log_norm_const = self.log_norm_const.to(device)
        return self

    def __repr__(self):
        return (f"ScaledMultivariateNormalTorch(dim={self.dim}, scaling_range=({self.scaling_factors.min().item():.4f}, "
                f"{self.scaling_factors.max().item():.4f}), device={self.device})")
