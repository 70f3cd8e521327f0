"""Uniform increments in a dim-ball (reference: proposal_distributions/uniform.py:6-45)."""
from typing import Optional

import torch

import ptrwm_hip
from .base import ProposalDistribution


class UniformRadiusProposal(ProposalDistribution):
    """inc uniform in the ball of radius effective_radius = base_radius / sqrt(beta)."""

    def __init__(self, dim: int, base_radius: float, beta: float, device: torch.device, dtype: torch.dtype,
                 rng_generator: Optional[torch.Generator] = None):
        super().__init__(dim, beta, device, dtype, rng_generator)
        self.name = "UniformRadius"
        if base_radius <= 0:
            raise ValueError("base_radius must be positive")
        self.base_radius = base_radius
        self.effective_radius = base_radius / torch.sqrt(torch.tensor(self.beta, device=self.device, dtype=self.dtype))
        self.inv_dim = 1.0 / self.dim

    def get_name(self) -> str:
        return self.name

    def engine_proposal(self, beta_ladder=None):
        if beta_ladder is None:
            radius = self.effective_radius.reshape(1).to(torch.float32)
        else:
            radius = self.base_radius / torch.sqrt(torch.tensor(list(beta_ladder), device=self.device, dtype=torch.float32))
        return ptrwm_hip.Proposal(ptrwm_hip.PROPOSAL_UNIFORM_RADIUS, temp_scale=radius.contiguous(), inv_dim=self.inv_dim)
