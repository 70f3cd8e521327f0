"""Independent Laplace increments per coordinate (reference: proposal_distributions/laplace.py:5-44)."""
from typing import Optional

import torch

import ptrwm_hip
from .base import ProposalDistribution


class LaplaceProposal(ProposalDistribution):
    """inc_d = Laplace(0, scale_d), Var = 2 scale_d^2 = base_variance_vector[d] / beta."""

    def __init__(self, dim: int, base_variance_vector: torch.Tensor, beta: float, device: torch.device,
                 dtype: torch.dtype, rng_generator: Optional[torch.Generator] = None):
        super().__init__(dim, beta, device, dtype, rng_generator)
        self.name = "Laplace"
        if base_variance_vector.shape != (dim,):
            raise ValueError(f"base_variance_vector must have shape ({dim},), got {base_variance_vector.shape}")
        if not (base_variance_vector > 0).all():
            raise ValueError("All elements of base_variance_vector must be positive")
        self.base_variance_vector = base_variance_vector.to(device=self.device, dtype=self.dtype)
        self.scale_vector = torch.sqrt((self.base_variance_vector / self.beta) / 2.0)

    def get_name(self) -> str:
        return self.name

    def engine_proposal(self, beta_ladder=None):
        if beta_ladder is None:
            dim_scale = self.scale_vector.to(torch.float32)
            temp = torch.ones(1, device=self.device, dtype=torch.float32)
        else:
            # scale_{t,d} = sqrt(base_var_d / 2) * beta_t^{-1/2}
            dim_scale = torch.sqrt(self.base_variance_vector.to(torch.float32) / 2.0)
            temp = torch.rsqrt(torch.tensor(list(beta_ladder), device=self.device, dtype=torch.float32))
        return ptrwm_hip.Proposal(ptrwm_hip.PROPOSAL_LAPLACE, temp_scale=temp.contiguous(), dim_scale=dim_scale.contiguous())
