"""Split steps of a density without a fused kernel (dense-covariance MultivariateNormalTorch, dim 30) at BASELINE configs[2]'s
batch size, step by step (no graph), for `rocprofv3 --kernel-trace --stats`: which kernels a split step spends its time in.
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_split -- python3 tools/split_profile.py [ladders] [temps] [steps]"""
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rwm-pt-pytorch_amd"))
import torch  # noqa: E402

from algorithms import ParallelTemperingRWM_GPU_Optimized, geometric_beta_ladder  # noqa: E402
from target_distributions import MultivariateNormalTorch  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dim, dev = 30, torch.device("cuda:0")
idx = torch.arange(dim, dtype=torch.float32)
cov = 0.5 ** (idx[:, None] - idx[None, :]).abs()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    pt = ParallelTemperingRWM_GPU_Optimized(dim, 2.38**2 / dim, MultivariateNormalTorch(dim, cov=cov, device=dev),
                                            beta_ladder=geometric_beta_ladder(T), swap_every=10, burn_in=0, device=dev,
                                            num_replicas=C, seed=42, trace="none")
    pt._ensure_started()
pt._run.use_graph = False
pt._run.advance(N)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
pt._run.advance(N)
e1.record()
torch.cuda.synchronize()
print(f"{C} x {T} x dim {dim}: {e0.elapsed_time(e1) / N * 1e3:.1f} us per split step")
