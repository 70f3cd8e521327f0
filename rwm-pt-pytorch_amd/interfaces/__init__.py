from .metropolis import MHAlgorithm, initial_state_for
from .target import TargetDistribution
from .target_torch import TorchTargetDistribution
from .simulation_gpu import MCMCSimulation_GPU

__all__ = ["MHAlgorithm", "initial_state_for", "TargetDistribution", "TorchTargetDistribution", "MCMCSimulation_GPU"]
