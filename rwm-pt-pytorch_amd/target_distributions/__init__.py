"""Target densities the fused HIP kernel evaluates (the in-scope subset)."""
from .multimodal_torch import RoughCarpetDistributionTorch, ThreeMixtureDistributionTorch
from .rosenbrock_torch import EvenRosenbrockTorch, FullRosenbrockTorch, HybridRosenbrockTorch
from .iid_product_torch import IIDBetaTorch, IIDGammaTorch

__all__ = [
    "RoughCarpetDistributionTorch",
    "ThreeMixtureDistributionTorch",
    "FullRosenbrockTorch",
    "EvenRosenbrockTorch",
    "HybridRosenbrockTorch",
    "IIDGammaTorch",
    "IIDBetaTorch",
]
