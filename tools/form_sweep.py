"""Speed of the two forms of the fused kernel (one thread per replica / lane-split) over batch sizes, ladder lengths and
dims (development aid for the AUTO rule in csrc/capi.hip; needs a GPU):  python tools/form_sweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rwm-pt-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import ptrwm_hip as E  # noqa: E402
from algorithms import ParallelTemperingRWM_GPU_Optimized, geometric_beta_ladder  # noqa: E402
from target_distributions import RoughCarpetDistributionTorch  # noqa: E402

dev = torch.device("cuda:0")


def rate(dim, T, C, form, target_steps=2e9):
    target = RoughCarpetDistributionTorch(dim, device=dev, mode_centers=[-15.0, 0.0, 15.0])
    alg = ParallelTemperingRWM_GPU_Optimized(dim, 2.38**2 / dim, target, beta_ladder=geometric_beta_ladder(T) if T > 1 else [1.0],
                                             swap_every=10, burn_in=0, device=dev, num_replicas=C, seed=1, trace="none")
    alg._ensure_started()
    inner = int(max(50, min(20000, target_steps / (C * T * dim / 30.0))))
    with E.kernel_form(form):
        alg._run.advance(inner)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            alg._run.advance(inner)
        e1.record()
        torch.cuda.synchronize()
    return C * T * inner * 3 / (e0.elapsed_time(e1) * 1e-3)


def dense(held_out=False, only_dims=None):
    """The table the AUTO rule of csrc/capi.hip is fitted on (and, with held_out, the one it is checked against): both
    pinned forms and AUTO over waves-per-SIMD of the thread form, ladder lengths and dims; one line per case."""
    ws = (0.6, 0.9, 1.1, 1.4, 1.6, 1.9, 2.2, 2.75, 3.5) if held_out else (0.25, 0.5, 0.75, 1.0, 1.25, 1.5, 1.75, 2.0, 2.5, 3.0, 4.0)
    # the fit grid: the dims with a kernel of their own (dim compiled in: 20, 30, 50) and, for all the others, dims across the
    # generic register widths; held out: other generic dims, and the compiled-in ones at other batch sizes
    dims = (18, 22, 26, 30, 38, 46, 50, 54, 62) if held_out else (16, 20, 24, 28, 30, 32, 36, 40, 44, 48, 50, 52, 56, 60, 64)
    if only_dims:  # (a sweep can be cut into several GPU calls: python tools/form_sweep.py dense 16,20,24 ...; concatenate)
        dims = tuple(d for d in dims if d in only_dims)
    print(f"# source_hash {E.source_hash()}")  # the kernels this sweep measures (tools/form_fit.py stamps the table with it)
    print(f"{'dim':>4} {'T':>4} {'chains':>7} {'w':>5} {'thread':>10} {'quad':>10} {'auto':>10} {'auto/best':>9}")
    worst = 1.0
    for dim in dims:
        for T in (1, 8, 32, 64):
            cpw = 1 if T > 64 else 64 // T
            for w in ws:
                C = max(1, int(round(w * 1024 * cpw)))
                a, b, c = rate(dim, T, C, E.FORM_THREAD, 1e9), rate(dim, T, C, E.FORM_QUAD, 1e9), rate(dim, T, C, E.FORM_AUTO, 1e9)
                worst = min(worst, c / max(a, b))
                print(f"{dim:4d} {T:4d} {C:7d} {w:5.2f} {a:10.3e} {b:10.3e} {c:10.3e} {c / max(a, b):9.3f}", flush=True)
    print(f"# worst AUTO / best pinned form: {worst:.3f}")


def main():
    if len(sys.argv) > 1 and sys.argv[1] in ("dense", "heldout"):
        return dense(sys.argv[1] == "heldout", [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else None)
    cases = [(30, 1, c) for c in (1024, 8192, 32768, 65536, 81920, 98304, 131072, 262144)] + \
            [(30, 8, c) for c in (1, 64, 1024, 4096, 8192, 16384, 32768)] + \
            [(30, 32, c) for c in (1, 64, 1024, 2048, 3072, 4096, 8192)] + \
            [(30, 64, c) for c in (256, 1024, 2048, 4096)] + [(30, 128, c) for c in (128, 1024, 4096)] + \
            [(50, 1, 65536), (50, 8, 8192), (50, 32, 1024), (50, 32, 4096), (50, 64, 2048)] + \
            [(100, 1, 65536), (100, 1, 262144), (100, 8, 8192), (100, 8, 65536), (100, 32, 1024), (100, 32, 65536), (100, 64, 1024),
             (100, 64, 16384), (100, 128, 8192), (80, 32, 16384), (65, 32, 16384), (64, 32, 16384)] + \
            [(d, 32, c) for d in (41, 48, 57, 64) for c in (1024, 1536, 2048, 3072, 4096)] + \
            [(d, 1, c) for d in (48, 64) for c in (32768, 65536, 98304, 131072)]
    print(f"{'dim':>4} {'T':>4} {'chains':>7} {'thread-waves/SIMD':>18} {'thread':>10} {'quad':>10} {'quad/thread':>11}")
    for dim, T, C in cases:
        if not E.has_quad_variant(0, 0, dim, T):
            continue
        a, b = rate(dim, T, C, E.FORM_THREAD), rate(dim, T, C, E.FORM_QUAD)
        w1 = (C * ((T + 63) // 64) if T > 64 else -(-C // (64 // T))) / 1024
        print(f"{dim:4d} {T:4d} {C:7d} {w1:18.2f} {a:10.3e} {b:10.3e} {b / a:11.2f}", flush=True)


if __name__ == "__main__":
    main()
