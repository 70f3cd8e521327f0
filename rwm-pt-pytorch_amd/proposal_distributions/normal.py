"""Isotropic Gaussian increments (reference: proposal_distributions/normal.py:5-42)."""
from typing import Optional

import torch

import ptrwm_hip
from .base import ProposalDistribution


class NormalProposal(ProposalDistribution):
    """inc = std_dev * z, z ~ N(0, I), std_dev = sqrt(base_variance_scalar / beta)."""

    def __init__(self, dim: int, base_variance_scalar: float, beta: float, device: torch.device, dtype: torch.dtype,
                 rng_generator: Optional[torch.Generator] = None):
        super().__init__(dim, beta, device, dtype, rng_generator)
        self.name = "Normal"
        if base_variance_scalar <= 0:
            raise ValueError("base_variance_scalar must be positive")
        self.base_variance_scalar = base_variance_scalar
        self.std_dev = torch.sqrt(torch.tensor(base_variance_scalar / self.beta, device=self.device, dtype=self.dtype))

    def get_name(self) -> str:
        return self.name

    def engine_proposal(self, beta_ladder=None):
        if beta_ladder is None:
            scale = self.std_dev.reshape(1).to(torch.float32)
        else:
            # per temperature sqrt(var / beta_t) in fp32, the diagonal of the reference PT class's
            # Cholesky factor (algorithms/pt_rwm_gpu_optimized.py:453-455)
            scale = torch.sqrt(torch.tensor([self.base_variance_scalar / b for b in beta_ladder],
                                            device=self.device, dtype=torch.float32))
        return ptrwm_hip.Proposal(ptrwm_hip.PROPOSAL_NORMAL, temp_scale=scale.contiguous())
