"""Multimodal targets: three-component Gaussian mixture and the "rough carpet" product density.

Same constructors, attributes and names as the reference classes
(target_distributions/multimodal_torch.py:4-334 ThreeMixtureDistributionTorch, :337-575
RoughCarpetDistributionTorch); the density itself is evaluated by the HIP engine
(csrc/targets.h RoughCarpet / ThreeMixture), these classes hold the parameters.
"""
import math

import numpy as np
import torch

import ptrwm_hip
from interfaces.target_torch import TorchTargetDistribution

_F32 = torch.float32


def _check_weights(mode_weights):
    if len(mode_weights) != 3:
        raise ValueError(f"mode_weights must contain exactly 3 weights, got {len(mode_weights)}")
    w = torch.tensor(mode_weights, dtype=_F32)
    if not torch.all(w > 0):
        raise ValueError("All mode_weights must be positive")
    if not torch.allclose(torch.sum(w), torch.tensor(1.0), rtol=1e-6):
        raise ValueError(f"mode_weights must sum to 1.0, got sum = {torch.sum(w).item()}")


def _suffix(is_default, scaling):
    return ("" if is_default else "Custom") + ("Scaled" if scaling else "")


def _random_scaling(dim, device):
    # uniform on [0.02, 1.98]: expectation 1 (multimodal_torch.py:82, :382)
    return torch.rand(dim, device=device, dtype=_F32) * (1.98 - 0.02) + 0.02


class ThreeMixtureDistributionTorch(TorchTargetDistribution):
    """p(x) = sum_k w_k N(x | mu_k, I); with scaling=True, p(x) = prod_j s_j * sum_k w_k N(x*s | mu_k, I)."""

    def __init__(self, dim, scaling=False, device=None, mode_centers=None, mode_weights=None):
        super().__init__(dim, device)
        default_centers = [[-5.0] + [0.0] * (dim - 1), [0.0] * dim, [5.0] + [0.0] * (dim - 1)]
        default_weights = [1 / 3, 1 / 3, 1 / 3]
        if mode_centers is None:
            mode_centers = default_centers
        if mode_weights is None:
            mode_weights = default_weights
        if len(mode_centers) != 3:
            raise ValueError(f"mode_centers must contain exactly 3 modes, got {len(mode_centers)}")
        for i, center in enumerate(mode_centers):
            if len(center) != dim:
                raise ValueError(f"Mode {i} has dimension {len(center)}, expected {dim}")
        _check_weights(mode_weights)

        self.means = torch.tensor(mode_centers, device=self.device, dtype=_F32)
        self.mixing_weights = torch.tensor(mode_weights, device=self.device, dtype=_F32)
        self.log_mixing_weights = torch.log(self.mixing_weights)
        log_2pi = torch.log(torch.tensor(2.0 * torch.pi, device=self.device, dtype=_F32))
        eye = torch.eye(dim, device=self.device, dtype=_F32)
        self.covs = eye.unsqueeze(0).repeat(3, 1, 1)
        self.cov_invs = self.covs.clone()
        self.cov_dets = torch.ones(3, device=self.device, dtype=_F32)
        self.log_norm_consts = -0.5 * (dim * log_2pi + torch.log(self.cov_dets))
        self.scaling_arg_from_constructor = bool(scaling)
        if scaling:
            self.scaling_factors = _random_scaling(dim, self.device)
            self.log_jacobian = torch.sum(torch.log(self.scaling_factors))
            self.base_log_norm_const_for_scaled = -0.5 * self.dim * log_2pi
        is_default = torch.allclose(
            torch.tensor(mode_centers, dtype=_F32), torch.tensor(default_centers, dtype=_F32), rtol=1e-6
        ) and torch.allclose(torch.tensor(mode_weights, dtype=_F32), torch.tensor(default_weights, dtype=_F32), rtol=1e-6)
        self.name = "ThreeMixtureTorch" + _suffix(is_default, scaling)

    def get_name(self):
        return self.name

    def engine_target(self):
        if self.scaling_arg_from_constructor:
            c = (self.base_log_norm_const_for_scaled + self.log_jacobian) + self.log_mixing_weights
            vec1 = self.scaling_factors.contiguous()
        else:
            c = self.log_norm_consts + self.log_mixing_weights
            vec1 = None
        # means that differ in the first coordinate only (the class default, and the +-15 centres of the reference's
        # experiments): declared to the engine (ip[0] = 1, include/ptrwm.h), which then evaluates the part of the
        # squared distances the three components share once
        first_only = self.dim == 1 or bool(torch.equal(self.means[0, 1:], self.means[1, 1:])
                                           and torch.equal(self.means[1, 1:], self.means[2, 1:]))
        return ptrwm_hip.Target(
            ptrwm_hip.TARGET_THREE_MIXTURE, self.dim, p=tuple(c.tolist()), ip=(1 if first_only else 0,),
            vec0=self.means.contiguous().view(-1), vec1=vec1
        )

    def log_density(self, x):
        return self._engine_log_density(x)

    def density(self, x):
        return torch.exp(self.log_density(x))

    def _sample(self, n, beta):
        comp = torch.multinomial(self.mixing_weights, n, replacement=True)
        noise = torch.randn(n, self.dim, device=self.device, dtype=_F32) / math.sqrt(beta)
        y = self.means[comp] + noise  # y ~ N(mu_k, I / beta)
        return y / self.scaling_factors if self.scaling_arg_from_constructor else y

    def draw_sample(self, beta=1.0):
        with torch.no_grad():
            if self.scaling_arg_from_constructor:
                return self._sample(1, beta)[0].cpu().numpy()
            # unscaled single draw picks the component uniformly (multimodal_torch.py:261)
            k = torch.randint(0, 3, (1,), device=self.device).item()
            z = torch.randn(self.dim, device=self.device, dtype=_F32) / math.sqrt(beta)
            return (self.means[k] + z).cpu().numpy()

    def draw_samples_torch(self, n_samples, beta=1.0):
        return self._sample(n_samples, beta)

    def to(self, device):
        super().to(device)
        for attr in ("means", "covs", "cov_invs", "cov_dets", "log_norm_consts", "mixing_weights",
                     "log_mixing_weights", "scaling_factors", "log_jacobian", "base_log_norm_const_for_scaled"):
            if hasattr(self, attr):
                setattr(self, attr, getattr(self, attr).to(device))
        return self


class RoughCarpetDistributionTorch(TorchTargetDistribution):
    """Product over coordinates of a 1-D three-mode Gaussian mixture (optionally coordinate-scaled)."""

    def __init__(self, dim, scaling=False, device=None, mode_centers=None, mode_weights=None):
        super().__init__(dim, device)
        default_centers, default_weights = [-5.0, 0.0, 5.0], [0.5, 0.3, 0.2]
        if mode_centers is None:
            mode_centers = default_centers
        if mode_weights is None:
            mode_weights = default_weights
        if len(mode_centers) != 3:
            raise ValueError(f"mode_centers must contain exactly 3 modes, got {len(mode_centers)}")
        for i, center in enumerate(mode_centers):
            if not isinstance(center, (int, float)):
                raise ValueError(f"Mode center {i} must be a scalar, got {type(center)}")
        _check_weights(mode_weights)
        is_default = torch.allclose(
            torch.tensor(mode_centers, dtype=_F32), torch.tensor(default_centers, dtype=_F32), rtol=1e-6
        ) and torch.allclose(torch.tensor(mode_weights, dtype=_F32), torch.tensor(default_weights, dtype=_F32), rtol=1e-6)
        self.name = "RoughCarpetTorch" + _suffix(is_default, scaling)
        self.modes = torch.tensor(mode_centers, device=self.device, dtype=_F32)
        self.weights = torch.tensor(mode_weights, device=self.device, dtype=_F32)
        self.log_weights = torch.log(self.weights)
        self.log_sqrt_2pi = torch.log(torch.sqrt(torch.tensor(2.0 * torch.pi, device=self.device, dtype=_F32)))
        if scaling:
            self.scaling_factors = _random_scaling(dim, self.device)

    def get_name(self):
        return self.name

    def engine_target(self):
        scaled = hasattr(self, "scaling_factors")
        log_jac = float(torch.sum(torch.log(self.scaling_factors))) if scaled else 0.0
        p = tuple(self.modes.tolist()) + tuple(self.log_weights.tolist()) + (log_jac,)
        return ptrwm_hip.Target(
            ptrwm_hip.TARGET_ROUGH_CARPET, self.dim, p=p, vec0=self.scaling_factors.contiguous() if scaled else None
        )

    def log_density(self, x):
        return self._engine_log_density(x)

    def density(self, x):
        return torch.exp(self.log_density(x))

    def density_1d(self, x):
        """Density of the 1-D three-mode factor at scalar(s) x (plotting helper of the reference, :436-456):
        the dim-1 rough carpet, evaluated by the same engine kernel."""
        x = torch.as_tensor(x, device=self.device, dtype=_F32)
        one = ptrwm_hip.Target(ptrwm_hip.TARGET_ROUGH_CARPET, 1,
                               p=tuple(self.modes.tolist()) + tuple(self.log_weights.tolist()) + (0.0,))
        return torch.exp(ptrwm_hip.logdensity(one, x.reshape(-1, 1).contiguous())).reshape(x.shape)

    def draw_samples_torch(self, n_samples, beta=1.0):
        idx = torch.multinomial(self.weights, n_samples * self.dim, replacement=True).view(n_samples, self.dim)
        noise = torch.randn(n_samples, self.dim, device=self.device, dtype=_F32) / math.sqrt(beta)
        y = self.modes[idx] + noise
        return y / self.scaling_factors if hasattr(self, "scaling_factors") else y

    def draw_sample(self, beta=1.0):
        with torch.no_grad():
            return self.draw_samples_torch(1, beta)[0].cpu().numpy()

    def to(self, device):
        super().to(device)
        for attr in ("modes", "weights", "log_weights", "log_sqrt_2pi", "scaling_factors"):
            if hasattr(self, attr):
                setattr(self, attr, getattr(self, attr).to(device))
        return self
