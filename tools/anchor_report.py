"""Engine against the statistical anchors, printed (development aid; the assertions live in tests/test_gpu_api.py)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rwm-pt-pytorch_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from algorithms import ParallelTemperingRWM_GPU_Optimized  # noqa: E402
from target_distributions import RoughCarpetDistributionTorch  # noqa: E402

dev = torch.device("cuda:0")
G = os.path.join(ROOT, "tests", "golden")
ora = json.load(open(os.path.join(G, "oracle_anchor_pt.json")))
ref = json.load(open(os.path.join(G, "reference_anchors.json")))
target = RoughCarpetDistributionTorch(30, device=dev, mode_centers=[-15.0, 0.0, 15.0])
n, burn, lad = ora["steps_per_ladder"], ora["burn_in"], 16384
pt = ParallelTemperingRWM_GPU_Optimized(30, 2.38**2 / 30, target, beta_ladder=ora["beta_ladder"], swap_every=ora["swap_every"],
                                        burn_in=burn, device=dev, num_replicas=lad, seed=31337, swap_mode="reference_copy",
                                        trace="none")
pt._ensure_started()
pt._run.advance(burn + n)
att = pt._run.swap_attempts_per_replica()
got = {"cold_esjd": (pt._run.sq_jump[:, 0] / n).cpu().numpy(),
       "swap_accept_fraction": (pt._run.swap_accept.sum(1).double() / att).cpu().numpy(),
       "cold_acceptance_rate": (pt._run.n_accept[:, 0].double() / n).cpu().numpy()}
for k, v in got.items():
    a = ora[k]
    se = v.std(ddof=1) / np.sqrt(lad)
    line = f"{k:22s} engine {v.mean():.6f} +- {se:.6f}   oracle {a['mean']:.6f} +- {a['stderr']:.6f}   rel diff {(v.mean() - a['mean']) / a['mean']:+.2e}"
    r = ref["pt"].get(k)
    if r:
        line += f"   reference {r['mean']:.6f} +- {r['stderr']:.6f}"
    print(line)
