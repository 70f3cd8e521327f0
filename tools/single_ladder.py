"""One ladder (the reference's own calling pattern: num_replicas = 1) through generate_samples: PT steps per second for
the dims the reference's experiments use, both kernel forms (development aid; needs a GPU)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rwm-pt-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import ptrwm_hip as E  # noqa: E402
from algorithms import ParallelTemperingRWM_GPU_Optimized  # noqa: E402
from target_distributions import RoughCarpetDistributionTorch, ThreeMixtureDistributionTorch  # noqa: E402

dev = torch.device("cuda:0")
print(f"{'target':>14} {'dim':>4} {'temps':>5} {'form':>7} {'PT steps/s':>11} {'us/step':>8}")
only_rc = os.environ.get("PTRWM_SL_ONLY_RC")  # experimental libraries hold the RoughCarpet kernels only
dims = [int(d) for d in os.environ.get("PTRWM_SL_DIMS", "2,5,10,20,30,50,100").split(",")]
forms_wanted = os.environ.get("PTRWM_SL_FORMS", "thread,quad,auto").split(",")
for cls in (RoughCarpetDistributionTorch,) if only_rc else (RoughCarpetDistributionTorch, ThreeMixtureDistributionTorch):
    for dim in dims:
        target = cls(dim, device=dev)
        for form, name in ((E.FORM_THREAD, "thread"), (E.FORM_QUAD, "quad"), (E.FORM_AUTO, "auto")):
            if name not in forms_wanted:
                continue
            with E.kernel_form(form):
                alg = ParallelTemperingRWM_GPU_Optimized(dim, 2.38**2 / dim, target, geom_temp_spacing=True, swap_every=10,
                                                         device=dev, trace="none", seed=1)
                alg._advance(2000)
                torch.cuda.synchronize()
                n = 200000
                t0 = time.perf_counter()
                alg._advance(n)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            print(f"{cls.__name__[:14]:>14} {dim:4d} {alg.num_chains:5d} {name:>7} {n / dt:11.3e} {dt / n * 1e6:8.3f}", flush=True)
