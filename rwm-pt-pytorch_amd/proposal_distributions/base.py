"""Proposal plug-in interface (reference: proposal_distributions/base.py:7-56).

A proposal object is configuration: the samplers read its parameters (`std_dev`, `scale_vector`,
`effective_radius`, `inv_dim`) and hand them to the fused kernel through `engine_proposal`, which
draws the increments in-kernel from Philox.  `sample(n)` is kept for callers that want increments
directly; it runs the same device code through `ptrwm_propose`.
"""
from abc import ABC, abstractmethod
from typing import Optional, Sequence

import torch

import ptrwm_hip


class ProposalDistribution(ABC):
    def __init__(self, dim: int, beta: float, device: torch.device, dtype: torch.dtype,
                 rng_generator: Optional[torch.Generator] = None):
        self.dim = dim
        self.beta = beta
        self.device = device
        self.dtype = dtype
        self.rng_generator = rng_generator
        self._draws = 0

    @abstractmethod
    def get_name(self) -> str:
        ...

    @abstractmethod
    def engine_proposal(self, beta_ladder: Optional[Sequence[float]] = None) -> "ptrwm_hip.Proposal":
        """Kernel-side description.  With `beta_ladder=None` the proposal's own beta is used (one
        temperature); with a ladder, the base scale is re-tempered per temperature by the same rule
        the constructor applies to `beta`."""

    def _next_seed(self) -> int:
        """Philox key for one `sample` call: from the generator's seed if one was given, otherwise
        from torch's global CPU generator (so torch.manual_seed makes `sample` reproducible)."""
        self._draws += 1
        if self.rng_generator is not None:
            return (self.rng_generator.initial_seed() * 0x9E3779B97F4A7C15 + self._draws) & (2**63 - 1)
        return int(torch.randint(0, 2**62, (1,)).item())

    def sample(self, n_samples: int) -> torch.Tensor:
        """`n_samples` increments, shape (n_samples, dim), drawn on the device by the engine."""
        inc = ptrwm_hip.propose(self.engine_proposal(), self.dim, n_samples, seed=self._next_seed())
        return inc[:, 0, :].to(self.dtype)

    def sample_into(self, n_samples: int, output_tensor: torch.Tensor) -> None:
        output_tensor.copy_(self.sample(n_samples))
