"""Target densities the fused HIP kernel evaluates: the in-scope subset plus the cheap
"next" targets of section 8f-3 (diagonal MVN, ScaledMVN, Hypercube, NealFunnel); dense-covariance MVN and SuperFunnel
evaluate on the device with torch and run through the engine's split steps."""
from .multimodal_torch import RoughCarpetDistributionTorch, ThreeMixtureDistributionTorch
from .rosenbrock_torch import EvenRosenbrockTorch, FullRosenbrockTorch, HybridRosenbrockTorch
from .iid_product_torch import IIDBetaTorch, IIDGammaTorch
from .multivariate_normal_torch import MultivariateNormalTorch, ScaledMultivariateNormalTorch
from .hypercube_torch import HypercubeTorch
from .funnel_torch import NealFunnelTorch, SuperFunnelTorch

__all__ = [
    "RoughCarpetDistributionTorch",
    "ThreeMixtureDistributionTorch",
    "FullRosenbrockTorch",
    "EvenRosenbrockTorch",
    "HybridRosenbrockTorch",
    "IIDGammaTorch",
    "IIDBetaTorch",
    "MultivariateNormalTorch",
    "ScaledMultivariateNormalTorch",
    "HypercubeTorch",
    "NealFunnelTorch",
    "SuperFunnelTorch",
]
