"""Products of IID Gamma / Beta marginals.

Parameter holders with the reference's constructors and names
(target_distributions/iid_product_torch.py:5-131 IIDGammaTorch, :134-274 IIDBetaTorch); densities
(support mask included) are evaluated by the HIP engine (csrc/targets.h IIDGamma / IIDBeta).
"""
import numpy as np
import torch

import ptrwm_hip
from interfaces.target_torch import TorchTargetDistribution

_F32 = torch.float32


class IIDGammaTorch(TorchTargetDistribution):
    """prod_d Gamma(x_d | shape, scale)."""

    def __init__(self, dim, shape=2.0, scale=3.0, device=None):
        super().__init__(dim, device)
        self.name = "IIDGammaTorch"
        self.shape = torch.tensor(shape, device=self.device, dtype=_F32)
        self.scale = torch.tensor(scale, device=self.device, dtype=_F32)
        self.log_gamma_shape = torch.lgamma(self.shape)
        self.log_norm_const_1d = self.log_gamma_shape + self.shape * torch.log(self.scale)
        self.log_norm_const = dim * self.log_norm_const_1d

    def get_name(self):
        return self.name

    def engine_target(self):
        return ptrwm_hip.Target(ptrwm_hip.TARGET_IID_GAMMA, self.dim,
                                p=(float(self.shape), float(self.scale), float(self.log_norm_const)))

    def log_density(self, x):
        return self._engine_log_density(x)

    def density(self, x):
        return torch.exp(self.log_density(x))

    def draw_sample(self, beta=1.0):
        return np.random.gamma(float(self.shape) * beta, float(self.scale), self.dim)

    def draw_samples_torch(self, n_samples, beta=1.0):
        dist = torch.distributions.Gamma(self.shape * beta, 1.0 / self.scale)
        return dist.sample((n_samples, self.dim)).to(self.device)

    def to(self, device):
        super().to(device)
        for attr in ("shape", "scale", "log_gamma_shape", "log_norm_const_1d", "log_norm_const"):
            setattr(self, attr, getattr(self, attr).to(device))
        return self


class IIDBetaTorch(TorchTargetDistribution):
    """prod_d Beta(x_d | alpha, beta)."""

    def __init__(self, dim, alpha=2.0, beta=3.0, device=None):
        super().__init__(dim, device)
        self.name = "IIDBetaTorch"
        self.alpha = torch.tensor(alpha, device=self.device, dtype=_F32)
        self.beta = torch.tensor(beta, device=self.device, dtype=_F32)
        self.log_gamma_alpha = torch.lgamma(self.alpha)
        self.log_gamma_beta = torch.lgamma(self.beta)
        self.log_gamma_alpha_beta = torch.lgamma(self.alpha + self.beta)
        self.log_norm_const_1d = self.log_gamma_alpha_beta - self.log_gamma_alpha - self.log_gamma_beta
        self.log_norm_const = dim * self.log_norm_const_1d

    def get_name(self):
        return self.name

    def engine_target(self):
        return ptrwm_hip.Target(ptrwm_hip.TARGET_IID_BETA, self.dim,
                                p=(float(self.alpha), float(self.beta), float(self.log_norm_const)))

    def log_density(self, x):
        return self._engine_log_density(x)

    def density(self, x):
        return torch.exp(self.log_density(x))

    def draw_sample(self, beta_temp=1.0):
        return np.random.beta(float(self.alpha) * beta_temp, float(self.beta) * beta_temp, self.dim)

    def draw_samples_torch(self, n_samples, beta_temp=1.0):
        dist = torch.distributions.Beta(self.alpha * beta_temp, self.beta * beta_temp)
        return dist.sample((n_samples, self.dim)).to(self.device)

    def to(self, device):
        super().to(device)
        for attr in ("alpha", "beta", "log_gamma_alpha", "log_gamma_beta", "log_gamma_alpha_beta",
                     "log_norm_const_1d", "log_norm_const"):
            setattr(self, attr, getattr(self, attr).to(device))
        return self
