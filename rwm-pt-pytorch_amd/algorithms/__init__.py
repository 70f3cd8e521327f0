"""GPU samplers behind the reference's class names (the NumPy CPU samplers of the reference,
algorithms/rwm.py and pt_rwm.py, are not part of the product; their restatement lives in oracle/)."""
from .rwm_gpu_optimized import RandomWalkMH_GPU_Optimized, ultra_fused_mcmc_step_basic
from .pt_rwm_gpu_optimized import ParallelTemperingRWM_GPU_Optimized, geometric_beta_ladder

# BASELINE.json's short name for the RWM class
RWM_GPU_Optimized = RandomWalkMH_GPU_Optimized

__all__ = [
    "RandomWalkMH_GPU_Optimized",
    "RWM_GPU_Optimized",
    "ParallelTemperingRWM_GPU_Optimized",
    "geometric_beta_ladder",
    "ultra_fused_mcmc_step_basic",
]
