from .base import ProposalDistribution
from .normal import NormalProposal
from .laplace import LaplaceProposal
from .uniform import UniformRadiusProposal

__all__ = ["ProposalDistribution", "NormalProposal", "LaplaceProposal", "UniformRadiusProposal"]
