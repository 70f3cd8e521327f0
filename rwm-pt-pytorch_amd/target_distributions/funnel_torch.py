"""Neal's funnel (reference: target_distributions/funnel_torch.py:6-110 NealFunnelTorch); evaluated by the HIP engine
(csrc/targets.h NealFunnel).  SuperFunnelTorch (:112-345, hierarchical logistic regression on per-group data) has no
fused kernel: its density is a batched device-side torch expression and the samplers run it in split steps."""
import math

import torch

import torch.nn.functional as F

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


class SuperFunnelTorch(TorchTargetDistribution):
    """Hierarchical logistic regression "super-funnel" (reference funnel_torch.py:112-297).

    theta = (alpha_1..J, beta_11..JK, mu_alpha, mu_beta_1..K, tau_alpha, tau_beta), dim = J + J K + K + 3;
    alpha_j ~ N(mu_alpha, tau_alpha^2), beta_jk ~ N(mu_beta_k, tau_beta^2), hyper-means ~ N(0, s^2), taus ~
    HalfCauchy(scale), y_ij ~ Bernoulli(sigmoid(alpha_j + x_ij . beta_j)); log density -inf unless both taus > 1e-9.

    The ragged per-group data is flattened once into one [N, K] design matrix with a group index per observation, so
    the likelihood of a batch of states is two gathers, one fused multiply-sum and a softplus over [B, N] - no Python
    loop over groups.  No fused kernel: the samplers evaluate it between ptrwm_split_propose / ptrwm_split_accept."""

    def __init__(self, J, K, X_data, Y_data, prior_hypermean_std=10.0, prior_tau_scale=2.5, device=None):
        self.J, self.K = int(J), int(K)
        super().__init__(self.J + self.J * self.K + 1 + self.K + 2, device)
        if not (isinstance(X_data, list) and len(X_data) == J):
            raise ValueError(f"X_data must be a list of J={J} tensors.")
        if not (isinstance(Y_data, list) and len(Y_data) == J):
            raise ValueError(f"Y_data must be a list of J={J} tensors.")
        for j in range(self.J):
            if not isinstance(X_data[j], torch.Tensor) or not isinstance(Y_data[j], torch.Tensor):
                raise ValueError(f"X_data[{j}] and Y_data[{j}] must be PyTorch tensors.")
            if X_data[j].ndim != 2 or X_data[j].shape[1] != K:
                raise ValueError(f"X_data[{j}] must have shape (n_j, K={K}). Got {X_data[j].shape}")
            if Y_data[j].ndim != 1 or Y_data[j].shape[0] != X_data[j].shape[0]:
                raise ValueError(f"Y_data[{j}] must have shape (n_j,). Got {Y_data[j].shape}, X_data had "
                                 f"{X_data[j].shape[0]} samples.")
        self.X_data = [x.to(self.device, torch.float32) for x in X_data]
        self.Y_data = [y.to(self.device, torch.float32) for y in Y_data]
        self.n_j_array = torch.tensor([y.shape[0] for y in Y_data], device=self.device, dtype=torch.long)
        self._x = torch.cat(self.X_data, 0) if self.J else torch.zeros(0, K, device=self.device)   # [N, K]
        self._y = torch.cat(self.Y_data, 0) if self.J else torch.zeros(0, device=self.device)      # [N]
        self._g = torch.repeat_interleave(torch.arange(self.J, device=self.device), self.n_j_array)  # [N] group of each row
        self.prior_hypermean_std = torch.tensor(float(prior_hypermean_std), device=self.device)
        self.prior_tau_scale = torch.tensor(float(prior_tau_scale), device=self.device)
        s2, k = float(prior_hypermean_std) ** 2, float(self.K)
        # every additive constant of the priors, once
        self._const = (-0.5 * self.J * math.log(2 * math.pi) - 0.5 * self.J * k * math.log(2 * math.pi)
                       - 0.5 * (1 + k) * (math.log(2 * math.pi) + math.log(s2))
                       + 2.0 * (math.log(2.0) - math.log(math.pi) - math.log(float(prior_tau_scale))))
        self._inv_s2 = 1.0 / s2

    def get_name(self):
        return f"SuperFunnelTorch_J{self.J}_K{self.K}"

    def log_density(self, theta):
        if not torch.is_tensor(theta):
            theta = torch.as_tensor(theta)
        theta = theta.to(self.device, torch.float32)
        single = theta.ndim == 1
        th = theta.unsqueeze(0) if single else theta
        J, K = self.J, self.K
        alpha = th[:, :J]                                        # [B, J]
        beta = th[:, J:J + J * K].reshape(-1, J, K)               # [B, J, K]
        mu_a = th[:, J + J * K]
        mu_b = th[:, J + J * K + 1:J + J * K + 1 + K]            # [B, K]
        tau_a, tau_b = th[:, -2], th[:, -1]
        valid = (tau_a > 1e-9) & (tau_b > 1e-9)
        ta = torch.where(valid, tau_a, torch.ones_like(tau_a))
        tb = torch.where(valid, tau_b, torch.ones_like(tau_b))
        # likelihood: y log sigmoid(eta) + (1 - y) log sigmoid(-eta) = y eta - softplus(eta)
        eta = alpha[:, self._g] + (beta[:, self._g, :] * self._x).sum(-1)   # [B, N]
        ll = (self._y * eta - F.softplus(eta)).sum(1)
        prior = (-J * torch.log(ta) - 0.5 * ((alpha - mu_a[:, None]) ** 2).sum(1) / ta ** 2
                 - J * K * torch.log(tb) - 0.5 * ((beta - mu_b[:, None, :]) ** 2).sum((1, 2)) / tb ** 2
                 - 0.5 * self._inv_s2 * (mu_a ** 2 + (mu_b ** 2).sum(1))
                 - torch.log1p((ta / self.prior_tau_scale) ** 2) - torch.log1p((tb / self.prior_tau_scale) ** 2))
        out = torch.where(valid, ll + prior + self._const, torch.full_like(ll, -torch.inf))
        return out[0] if single else out

    def density(self, x):
        return torch.exp(self.log_density(x))

    def draw_sample(self, beta=1.0):
        raise NotImplementedError("SuperFunnelTorch.draw_sample is not implemented.")
