"""Target-distribution plug-in interface of the GPU samplers.

Same contract as the reference's `TorchTargetDistribution` (interfaces/target_torch.py:5-67):
`density`, `log_density`, `get_name`, `draw_sample(beta)`, `to(device)` and an optional
`draw_samples_torch(n, beta)`.  Targets the fused HIP kernel can evaluate additionally implement
`engine_target()`, which describes the density to the C ABI (include/ptrwm.h ptrwm_target_desc);
their `log_density` is itself served by the engine (`ptrwm_logdensity`), so there is exactly one
implementation of each density on the product path.
"""
from abc import ABC, abstractmethod

import torch


class TorchTargetDistribution(ABC):
    def __init__(self, dimension, device=None):
        self.dim = dimension
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.device = torch.device(device)

    @abstractmethod
    def density(self, x):
        raise NotImplementedError("Subclasses must implement the density method.")

    @abstractmethod
    def log_density(self, x):
        raise NotImplementedError("Subclasses must implement the log_density method.")

    @abstractmethod
    def get_name(self):
        raise NotImplementedError("Subclasses must implement the get_name method.")

    def draw_sample(self, beta=1.0):
        raise NotImplementedError("Subclasses should implement draw_sample for compatibility.")

    def to(self, device):
        self.device = torch.device(device)
        return self

    # ---- HIP engine hook -------------------------------------------------------------------
    def engine_target(self):
        """Return a `ptrwm_hip.Target` if the fused kernel knows this density, else raise NotImplementedError: the
        samplers then run the target in split steps (HIP proposal / accept / swap kernels around `log_density`)."""
        raise NotImplementedError(f"{type(self).__name__} has no fused-kernel implementation (engine_target).")

    def _engine_log_density(self, x):
        """Shared `log_density` body: evaluate rows of `x` with the engine's log-density kernel."""
        import ptrwm_hip

        if not torch.is_tensor(x):
            x = torch.as_tensor(x)
        if x.device != self.device:
            x = x.to(self.device)
        single = x.dim() == 1
        if x.dim() not in (1, 2) or x.shape[-1] != self.dim:
            raise ValueError(f"Expected tensor of shape ({self.dim},) or (batch_size, {self.dim}), got {tuple(x.shape)}")
        rows = x.reshape(-1, self.dim).to(torch.float32).contiguous()
        out = ptrwm_hip.logdensity(self.engine_target(), rows)
        return out[0] if single else out
