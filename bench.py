"""Benchmark of the PT-RWM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A bench "step" is ONE launch of the fused kernel over the rank's whole batch: `--inner` Metropolis steps
(default 2000 = 200 swap periods, ~0.11 s per launch, so that the driver's 20-launch timed region is > 2 s of steady
state) for every (ladder, temperature) replica.  Workload = BASELINE.json configs[2], the
configuration the headline metric is quoted on: PT-RWM, RoughCarpet dim 30 (modes +-15), Normal proposal
2.38^2/30, 32 geometric temperatures 1 -> 0.01, swap_every 10, 65 536 ladders PER GPU (weak scaling: ranks own
disjoint blocks of global ladder ids, no collective on the data path; one summary all-reduce at the end).

metric = chain-MH-steps/s: one unit = one (ladder, temperature) replica advancing one Metropolis step.
Inputs are resident in HBM before the timed region (state is generated on the device).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     dominant kernel (ptrwm_step_kernel).  The kernel keeps the state in registers for the whole launch, so
               the BINDING resource is VALU issue, not HBM: bound = "valu_issue", achieved = wave-level VALU
               instructions per second (SQ_INSTS_VALU per launch, from the rocprofv3 PMC pass of this same command
               kept in profiles/traffic.json together with the sha256 of the library it was measured on, divided by the
               launch duration measured HERE with HIP events on the launch stream), peak = 1024 SIMDs x shader clock /
               2 cycles per wave64 instruction (MI355X_MICROARCH.md), frac = achieved / peak <= 1.  `traffic` = HBM
               bytes per launch from the FETCH_SIZE / WRITE_SIZE passes.  The SURVEY 8(d) streaming-bytes accounting
               ((8*dim+24) B per chain-MH-step) is reported in `hbm_streaming_accounting`: it is an accounting
               figure - those bytes are never moved - and is NOT the roofline fraction.
               `hbm_stream_inner1` is the north star's literal formulation measured in the same run: ONE MH step per
               launch, where HBM streaming IS the bound (real bytes per launch / launch duration / 8 TB/s).
  cpu_baseline the NumPy port of the reference's CPU sampler (algorithms/pt_rwm.py) timed on this host, rank 0, N=1.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rwm-pt-pytorch_amd"))

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--inner", type=int, default=2000, help="Metropolis steps per launch")
    ap.add_argument("--chains", type=int, default=65536, help="ladders per GPU")
    ap.add_argument("--temps", type=int, default=32)
    ap.add_argument("--dim", type=int, default=30)
    ap.add_argument("--swap-every", type=int, default=10)
    ap.add_argument("--swap-order", default="sequential", choices=["sequential", "even_odd"])
    ap.add_argument("--workload", default="cfg3", choices=["cfg2", "cfg3", "cfg4", "cfg5", "pt", "rwm"],
                    help="BASELINE.json configs[1..4] (cfg3 = configs[2], the headline); pt = cfg3, rwm = cfg2")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra single-GPU readings (configs[1], even/odd)")
    return ap.parse_args()


def _cpu_worker(job):
    """One ladder of the NumPy port for `seconds`; returns (MH steps taken, iterations, elapsed)."""
    dim, temps, seconds, seed = job
    import numpy as np

    from oracle import numpy_baseline as NB

    ladder = [float(0.01 ** (t / (temps - 1))) for t in range(temps)] if temps > 1 else [1.0]
    np.random.seed(seed)
    if temps > 1:
        alg = NB.ParallelTemperingNumpy(dim, 2.38**2 / dim, NB.RoughCarpetNumpy(dim), ladder)
    else:
        alg = NB.RandomWalkMHNumpy(dim, 2.38**2 / dim, NB.RoughCarpetNumpy(dim))
    t0 = time.perf_counter()
    iters = mh_steps = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            alg.step()
            iters += 1
            # pt_rwm.py:175-181: on every 20th iteration only the hottest chain moves, the rest attempt swaps
            mh_steps += 1 if (temps > 1 and iters % NB.ParallelTemperingNumpy.swap_every == 0) else temps
    return mh_steps, iters, time.perf_counter() - t0


def cpu_baseline(dim, temps, seconds):
    """The reference's NumPy CPU path (restated in oracle/numpy_baseline.py, pinned to the reference by
    tests/test_numpy_baseline.py) on a bounded sample of the same workload.  `value` is the reference's native
    behaviour: one ladder on one core.  Because ladders are independent, the fairest CPU figure is one ladder per
    host core at the same time: reported as `all_cores`."""
    import multiprocessing as mp

    steps, iters, dt = _cpu_worker((dim, temps, seconds, 1))
    out = {
        "value": steps / dt, "unit": "chain-MH-steps/s", "cores": 1, "kind": "port",
        "sample": f"NumPy port of algorithms/{'pt_rwm' if temps > 1 else 'rwm'}.py, RoughCarpet dim {dim}, "
                  f"{temps} temperature(s), 1 ladder, {iters} iterations in {dt:.1f} s on one host core",
    }
    # worker pool sized to the host share of one GPU (16 cores on the benchmark boxes), never above the affinity mask
    n = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16))
    if n > 1:
        os.environ.setdefault("OMP_NUM_THREADS", "1")
        with mp.get_context("spawn").Pool(n) as pool:
            res = pool.map(_cpu_worker, [(dim, temps, min(seconds, 8.0), 100 + i) for i in range(n)])
        out["all_cores"] = {"value": sum(r[0] / r[2] for r in res), "cores": n,
                            "sample": f"{n} independent ladders, one process per host core, {min(seconds, 8.0):.0f} s each"}
    return out


def c_oracle_rate(dim, temps, swap_every):
    """Extra, informative: the compiled C oracle (fp32, one core) on 8 ladders x 200 steps."""
    import numpy as np

    from oracle import oracle as O

    lw = np.log(np.array([0.5, 0.3, 0.2], dtype=np.float32))
    ot = O.Target(O.TARGET_ROUGH_CARPET, dim, p=[-15.0, 0.0, 15.0, *lw, 0.0])
    beta = np.array([0.01 ** (t / max(1, temps - 1)) for t in range(temps)], dtype=np.float32)
    op = O.Proposal(O.PROPOSAL_NORMAL, np.sqrt((2.38**2 / dim / beta.astype(np.float64)).astype(np.float32)))
    C, N = 8, 200
    st = np.zeros((C, temps, dim), np.float32)
    lp = np.tile(O.logdensity(ot, np.zeros((1, dim), np.float32)).astype(np.float32), (C, temps))
    t0 = time.perf_counter()
    O.run(ot, op, state=st, logp=lp, beta=beta, step0=0, n_steps=N, swap_every=swap_every, seed=3)
    return C * temps * N / (time.perf_counter() - t0)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # convenience: re-launch under torch.distributed.run as a child (nothing has touched the GPU yet)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29531"),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    # CPU baseline first: it spawns worker processes, which must happen before this process touches the GPU
    cpu = None
    if world == 1 and args.cpu_seconds > 0:
        wl0 = {"pt": "cfg3", "rwm": "cfg2"}.get(args.workload, args.workload)
        cpu = cpu_baseline(args.dim, 1 if wl0 == "cfg2" else args.temps, args.cpu_seconds)
        cpu["c_oracle_chain_steps_per_s_1core"] = c_oracle_rate(args.dim, 1 if wl0 == "cfg2" else args.temps,
                                                                args.swap_every)

    import torch
    import torch.distributed as dist

    from algorithms import ParallelTemperingRWM_GPU_Optimized, RandomWalkMH_GPU_Optimized, geometric_beta_ladder
    from algorithms.sharding import allreduce_summary
    from target_distributions import RoughCarpetDistributionTorch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the product path has no CPU fallback")
    # one rank per GPU over RCCL (backend "nccl").  PTRWM_BENCH_BACKEND=gloo rehearses the multi-process path on a
    # box with fewer GPUs than ranks (ranks then share devices; collectives go through host tensors).
    backend = os.environ.get("PTRWM_BENCH_BACKEND", "nccl")
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    # under torch.distributed.run (RANK / WORLD_SIZE in the environment) the process group is created whatever the world
    # size: a one-rank launch then exercises the same init / barrier / all-reduce / destroy sequence as the 8-GPU one
    distributed = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from proposal_distributions import LaplaceProposal, UniformRadiusProposal
    from target_distributions import EvenRosenbrockTorch, ThreeMixtureDistributionTorch

    wl = {"pt": "cfg3", "rwm": "cfg2"}.get(args.workload, args.workload)
    dim, T, C = args.dim, args.temps, args.chains
    burn = 0
    proposal = None
    np_seed = 1234  # EvenRosenbrock starts at 1e-8 N(0,1) drawn from the global NumPy RNG: fix it
    import numpy as np

    np.random.seed(np_seed)
    if wl == "cfg2":
        T = 1
        target = RoughCarpetDistributionTorch(dim, device=dev, mode_centers=[-15.0, 0.0, 15.0])
        desc = "BASELINE configs[1]: RWM HIP, RoughCarpet dim=30 modes[-15,0,15], Normal proposal 2.38^2/dim, 65536 chains per GPU"
    elif wl == "cfg3":
        target = RoughCarpetDistributionTorch(dim, device=dev, mode_centers=[-15.0, 0.0, 15.0])
        desc = ("BASELINE configs[2]: PT-RWM HIP, RoughCarpet dim=30 modes[-15,0,15], Normal proposal 2.38^2/dim, "
                "32 geometric temps 1->0.01, swap_every=10, 65536 ladders per GPU")
    elif wl == "cfg4":
        target = EvenRosenbrockTorch(dim, device=dev)
        proposal = LaplaceProposal(dim, torch.full((dim,), 0.004), 1.0, dev, torch.float32)
        desc = ("BASELINE configs[3]: PT-RWM HIP, EvenRosenbrock dim=30, Laplace proposal base variance 0.004 per dim, 32 geometric "
                "temps, swap_every=10, 65536 ladders per GPU (524288 over 8), no collectives")
    else:
        dim, T = 50, 64
        C = args.chains * 2 if args.chains == 65536 else args.chains  # 131072 ladders per GPU (1048576 over 8)
        target = ThreeMixtureDistributionTorch(dim, device=dev)
        proposal = UniformRadiusProposal(dim, 2.4, 1.0, dev, torch.float32)
        desc = ("BASELINE configs[4]: PT-RWM HIP, ThreeMixture dim=50 (class defaults), UniformRadius base_radius 2.4, "
                "64 geometric temps, swap_every=10, 131072 ladders per GPU (1048576 over 8), summary all-reduce")
    offset = rank * C  # weak scaling: global ladder ids [rank*C, (rank+1)*C)
    if wl == "cfg2":
        alg = RandomWalkMH_GPU_Optimized(dim, 2.38**2 / dim, target, burn_in=burn, device=dev, num_chains=C, seed=42,
                                         chain_offset=offset)
    else:
        alg = ParallelTemperingRWM_GPU_Optimized(dim, 2.38**2 / dim, target, beta_ladder=geometric_beta_ladder(T),
                                                 swap_every=args.swap_every, burn_in=burn, device=dev, num_replicas=C,
                                                 seed=42, chain_offset=offset, trace="none", swap_order=args.swap_order,
                                                 proposal_distribution=proposal)
    alg._ensure_started()
    run = alg._run

    def barrier():
        if distributed:
            dist.barrier()

    for _ in range(args.warmup):
        run.advance(args.inner)
    torch.cuda.synchronize()
    barrier()
    # timed region: exactly K launches; one HIP event pair per launch on the launch stream (the current stream)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        run.advance(args.inner)
        b.record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    local_elapsed = elapsed
    if distributed:
        t = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    launch_ms = sorted(a.elapsed_time(b) for a, b in ev)
    kernel_ms = sum(launch_ms) / len(launch_ms)
    # Per-rank record for the N > 1 line (gathered below): which physical device each rank ran on and its own launch times,
    # so that a straggler GPU - or two ranks on one device - shows in the line itself
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "local_rank": local_rank, "device_index": local_dev, "device_name": props.name,
          "pci": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", -1) & 0xff,
                                     getattr(props, "pci_device_id", -1) & 0xff),
          "uuid": str(getattr(props, "uuid", "")), "kernel_ms_min": launch_ms[0], "kernel_ms_median": launch_ms[len(launch_ms) // 2],
          "kernel_ms_max": launch_ms[-1], "kernel_ms_mean": kernel_ms, "wall_s": local_elapsed}
    ranks = [me]
    if distributed:
        ranks = [None] * dist.get_world_size()
        dist.all_gather_object(ranks, me)
    distinct = len({(r["uuid"], r["pci"]) for r in ranks})

    def informative(r):  # a runtime that reports placeholders must not fail a healthy run: abort on real identifiers only
        return r["uuid"].replace("0", "").replace("-", "") != "" and not r["pci"].endswith(":ff:ff")

    if distributed and backend == "nccl" and distinct != len(ranks) and all(informative(r) for r in ranks):
        raise SystemExit(f"bench.py: {len(ranks)} ranks but only {distinct} distinct devices: {[(r['rank'], r['pci']) for r in ranks]}")

    def units_per_launch_fn(inner):
        return C * T * inner

    # The other single-GPU readings of the same metric, outside the timed region (N = 1 only): a sustained reading
    # (median of 3 repeats of >= 1 s each), the north star's one-step-per-launch formulation (HBM-streaming
    # bound), BASELINE configs[1] (RWM, 65 536 chains, one temperature) and configs[2] with even/odd swaps.
    others, inner1 = None, None
    if world == 1 and wl == "cfg3" and not args.no_extras:
        def timed_launches(r, n, inner):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(n):
                r.advance(inner)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n  # ms per launch

        def quick(make, inner, n=6):
            a2 = make()
            a2._ensure_started()
            a2._run.advance(inner)
            ms = timed_launches(a2._run, n, inner)
            return {"value": a2._run.n_replicas * a2._run.n_temps * inner / (ms * 1e-3), "kernel_ms": ms,
                    "mh_steps_per_launch": inner}

        # Split steps: the path a density WITHOUT a fused kernel takes (any user-defined TorchTargetDistribution, the
        # dense-covariance Gaussian, SuperFunnel): HIP proposal / Metropolis / swap kernels around the class's own torch
        # log_density, per step.  Dense-covariance MultivariateNormalTorch (multivariate_normal_torch.py:62-92: a [B, D] x
        # [D, D] product per evaluation), dim 30, a full batch and the reference's own single ladder, each replayed from a
        # captured HIP graph (16 steps per graph, algorithms/_engine_core.py) and issued step by step from Python.
        import warnings

        from target_distributions import MultivariateNormalTorch

        def split_reading(n_rep, temps, steps):
            idx = torch.arange(dim, dtype=torch.float32)
            cov = 0.5 ** (idx[:, None] - idx[None, :]).abs()  # a dense SPD covariance
            out = {}
            for name, use_graph in (("graph_replay", True), ("step_by_step", False)):
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    pt = ParallelTemperingRWM_GPU_Optimized(
                        dim, 2.38**2 / dim, MultivariateNormalTorch(dim, cov=cov, device=dev),
                        beta_ladder=geometric_beta_ladder(temps), swap_every=args.swap_every, burn_in=0, device=dev,
                        num_replicas=n_rep, seed=42, trace="none")
                    pt._ensure_started()
                pt._run.use_graph = use_graph
                pt._run.advance(1 + 2 * pt._run.GRAPH_STEPS)  # warm-up: capture included
                ms = timed_launches(pt._run, 3, steps) / steps  # ms per Metropolis step
                out[name] = {"value": n_rep * temps / (ms * 1e-3), "us_per_step": ms * 1e3}
                del pt
            # bytes one chain-step moves through HBM, by the arrays each kernel reads and writes (an estimate: the density's
            # intermediates are torch's): proposal kernel 8 D + 8, density 16 D + 4, Metropolis kernel 12 D + 56 (state and
            # proposals read, state written; the pre-step states are written back only on swap steps: + 0.4 D at swap_every 10)
            per = 36.4 * dim + 68
            g = out["graph_replay"]
            g["bytes_per_chain_step_estimate"] = per
            g["GBps_estimate"] = g["value"] * per / 1e9
            g["frac_of_hbm_peak_estimate"] = g["GBps_estimate"] / HBM_PEAK_GBPS
            out["value"] = g["value"]
            out["graph_over_step_by_step"] = g["value"] / out["step_by_step"]["value"]
            return out

        per_launch_s = kernel_ms * 1e-3
        n_rep = max(3, int(1.0 / per_launch_s) + 1)  # launches per repeat: >= 1 s
        reps = sorted(units_per_launch_fn(args.inner) / (timed_launches(run, n_rep, args.inner) * 1e-3) for _ in range(3))
        others = {
            "configs[2] sustained: median of 3 repeats": {
                "value": reps[1], "min": reps[0], "max": reps[2], "launches_per_repeat": n_rep,
                "seconds_per_repeat": n_rep * per_launch_s, "mh_steps_per_launch": args.inner},
            "configs[1]: RWM HIP, RoughCarpet dim 30, Normal proposal, 65536 chains x 1 temperature (lane-split form)": quick(
                lambda: RandomWalkMH_GPU_Optimized(dim, 2.38**2 / dim, target, burn_in=0, device=dev, num_chains=C,
                                                   seed=42), args.inner),
            "one ladder (the reference's own use: 1 replica x 8 geometric temps), lane-split form": quick(
                lambda: ParallelTemperingRWM_GPU_Optimized(dim, 2.38**2 / dim, target, geom_temp_spacing=True,
                                                           swap_every=args.swap_every, burn_in=0, device=dev,
                                                           num_replicas=1, seed=42, trace="none"), 20000, n=3),
            "configs[2] at dim 100 (lane-split form: the thread form holds one wave per SIMD there)": quick(
                lambda: ParallelTemperingRWM_GPU_Optimized(
                    100, 2.38**2 / 100, RoughCarpetDistributionTorch(100, device=dev, mode_centers=[-15.0, 0.0, 15.0]),
                    beta_ladder=geometric_beta_ladder(T), swap_every=args.swap_every, burn_in=0, device=dev,
                    num_replicas=C, seed=42, trace="none"), max(50, args.inner // 10), n=3),
            "configs[3] per-GPU shard: EvenRosenbrock dim 30, Laplace proposal, 32 temps, 65536 ladders": quick(
                lambda: ParallelTemperingRWM_GPU_Optimized(
                    30, 2.38**2 / 30, EvenRosenbrockTorch(30, device=dev), beta_ladder=geometric_beta_ladder(32),
                    swap_every=args.swap_every, burn_in=0, device=dev, num_replicas=65536, seed=42, trace="none",
                    proposal_distribution=LaplaceProposal(30, torch.full((30,), 0.004), 1.0, dev, torch.float32)),
                max(50, args.inner // 4), n=3),
            "configs[4] per-GPU shard: ThreeMixture dim 50, UniformRadius proposal, 64 temps, 131072 ladders": quick(
                lambda: ParallelTemperingRWM_GPU_Optimized(
                    50, 2.38**2 / 50, ThreeMixtureDistributionTorch(50, device=dev), beta_ladder=geometric_beta_ladder(64),
                    swap_every=args.swap_every, burn_in=0, device=dev, num_replicas=131072, seed=42, trace="none",
                    proposal_distribution=UniformRadiusProposal(50, 2.4, 1.0, dev, torch.float32)),
                max(50, args.inner // 10), n=3),
            "split steps (density without a fused kernel): dense-covariance MVN dim 30, 65536 ladders x 32 temps":
                split_reading(C, T, 48),
            "split steps: dense-covariance MVN dim 30, one ladder x 8 temps (the reference's own use)":
                split_reading(1, 8, 1600),
            "configs[2] with swap_order=even_odd": quick(
                lambda: ParallelTemperingRWM_GPU_Optimized(dim, 2.38**2 / dim, target,
                                                           beta_ladder=geometric_beta_ladder(T),
                                                           swap_every=args.swap_every, burn_in=0, device=dev,
                                                           num_replicas=C, seed=42, trace="none",
                                                           swap_order="even_odd"), args.inner),
        }
        # ONE Metropolis step per launch (the north star's literal formulation): every launch reads and writes the whole
        # state, log-densities and the four statistics arrays once - HBM streaming is the bound.  Bytes are the arrays'
        # sizes (the PMC passes of `--inner 1` in profiles/traffic.json agree); one HIP event pair per launch.
        import ptrwm_hip as P

        n1 = 300

        def one_step_launches(mode):
            with P.stream_mode(mode):
                for _ in range(20):
                    run.advance(1)
                kind = P.last_launch_kind()
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n1)]
                torch.cuda.synchronize()
                for a, b in evs:
                    a.record()
                    run.advance(1)
                    b.record()
                torch.cuda.synchronize()
            return sorted(a.elapsed_time(b) for a, b in evs), kind

        # what ptrwm_run picks by itself (AUTO: the streaming form of the kernel for this shape) and, for the record, the
        # classic kernel pinned, in the same process on the same state
        ms1, kind1 = one_step_launches(P.STREAM_AUTO)
        ms1_classic, _ = one_step_launches(P.STREAM_OFF)
        ms1_mean = sum(ms1) / n1
        reps_ct = C * T
        # bytes a launch really touches: state and log p read and written, the acceptance counts and the squared-jump sums
        # read-modify-written (some replica of nearly every cache line accepts), the swap counts and last-swap ordinals only
        # in the one launch in swap_every that has a swap event (the kernel leaves zero deltas alone)
        bytes1 = 2 * (reps_ct * dim * 4 + reps_ct * 4 + 2 * reps_ct * 8) + 2 * (2 * reps_ct * 8) / args.swap_every
        alg1 = (8 * dim + 24) * reps_ct  # algorithmic bytes of one Metropolis step over the batch
        inner1 = {"bound": "hbm", "algorithmic_bytes_per_launch": alg1, "bytes_per_launch": bytes1, "kernel_ms_mean": ms1_mean,
                  "kernel_ms_median": ms1[n1 // 2], "launches": n1,
                  "kernel": {P.LAUNCH_THREAD: "classic", P.LAUNCH_QUAD: "lane-split", P.LAUNCH_STREAM: "streaming"}.get(kind1),
                  # frac: by the SURVEY 8(d) accounting ((8 dim + 24) B per chain-step), the figure the north star's 0.60 is
                  # stated in; achieved_touched / frac_touched: by the bytes the kernel really touches (int64 / fp64 statistics)
                  "achieved": alg1 / (ms1_mean * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                  "frac": alg1 / (ms1_mean * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                  "achieved_touched": bytes1 / (ms1_mean * 1e-3) / 1e9,
                  "frac_touched": bytes1 / (ms1_mean * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                  "classic_kernel_pinned": {"kernel_ms_mean": sum(ms1_classic) / n1, "kernel_ms_median": ms1_classic[n1 // 2],
                                            "frac": alg1 / (sum(ms1_classic) / n1 * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                  "chain_mh_steps_per_s": reps_ct / (ms1_mean * 1e-3),
                  "note": "one MH step per launch.  bytes_per_launch = what the kernel touches: state [C,T,D] f32 and log p read "
                          "and written, accept counts (i64) and squared-jump sums (f64) read-modify-written, swap counts and "
                          "last-swap ordinals only in the 1 launch in swap_every with a swap event; `traffic` = the same from "
                          "the FETCH_SIZE / WRITE_SIZE counters of this command, and frac uses the counters when they are of "
                          "this build"}

    # HBM copy probe (the measured copy bandwidth as a second denominator): 1 GiB device-to-device
    copy_gbps = None
    if rank == 0:
        src = torch.empty(1 << 28, device=dev, dtype=torch.float32).normal_()
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbps = 10 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst

    summary = allreduce_summary(run.summary(), coll_dev)  # the only collective: whole-job acceptance / ESJD
    units_per_launch = C * T * args.inner            # per GPU
    value = world * units_per_launch * args.steps / elapsed
    alg_bytes = (8 * dim + 24) * units_per_launch
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9

    if rank == 0:
        # Counters of the SAME command from the committed rocprofv3 PMC passes (profiles/traffic.json: FETCH_SIZE /
        # WRITE_SIZE / GRBM_GUI_ACTIVE / SQ_* collected in separate passes, FETCH doubled per the gfx950 note).  They are
        # properties of the code object: the record carries the sha256 of the library it was measured on and the line
        # says whether that is the library being timed now.
        import hashlib

        import ptrwm_hip

        with open(ptrwm_hip.LIB_PATH, "rb") as f:
            lib_sha = hashlib.sha256(f.read()).hexdigest()
        traffic, rec = None, {}
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            with open(tfile) as f:
                tj = json.load(f)
            key = f"{'rwm' if T == 1 else 'pt'}_d{dim}_T{T}_C{C}" if wl in ("cfg2", "cfg3") else wl
            rec = tj.get(key, {})
            traffic = rec.get("hbm_bytes_per_launch")  # state in / state out: does not depend on the steps per launch
            rec1 = tj.get(key + "_inner1", {})
            if inner1 is not None and rec1:
                inner1["traffic"] = rec1.get("hbm_bytes_per_launch")
                inner1["traffic_source"] = rec1.get("source")
                inner1["profiled_kernel_ms"] = rec1.get("profiled_kernel_ms")
                inner1["traffic_of_this_build"] = rec1.get("lib_sha256") == lib_sha
                if inner1["traffic_of_this_build"] and inner1["traffic"]:
                    inner1["achieved_counters"] = inner1["traffic"] / (inner1["kernel_ms_mean"] * 1e-3) / 1e9
                    inner1["frac_counters"] = inner1["achieved_counters"] / HBM_PEAK_GBPS
        clock_ghz = rec.get("shader_clock_ghz", 2.4)
        valu_peak = 1024 * clock_ghz * 1e9 / 2  # 1024 SIMDs, one wave64 VALU instruction per 2 cycles each
        if "valu_insts_per_launch" in rec:
            # the instruction count of a launch is proportional to its Metropolis steps (prologue and epilogue are
            # < 0.1 % at >= 100 steps); the record states the steps it was collected at
            valu_launch = rec["valu_insts_per_launch"] * args.inner / rec.get("mh_steps_per_launch", args.inner)
            valu_rate = valu_launch / (kernel_ms * 1e-3)
            roof = {"bound": "valu_issue", "achieved": valu_rate / 1e9, "peak": valu_peak / 1e9, "unit": "Gwave-instr/s",
                    "frac": valu_rate / valu_peak, "traffic": traffic,
                    "valu_wave_insts_per_launch": valu_launch,
                    "valu_wave_insts_per_wave_step": valu_launch * 64 / units_per_launch,
                    "counters_collected_at_mh_steps_per_launch": rec.get("mh_steps_per_launch"),
                    "shader_clock_ghz": clock_ghz, "counters_source": rec.get("source"),
                    "counters_lib_sha256": rec.get("lib_sha256"), "counters_match_this_build": rec.get("lib_sha256") == lib_sha}
        else:  # no PMC record for this configuration: the fraction cannot be stated
            roof = {"bound": "valu_issue", "achieved": None, "peak": valu_peak / 1e9, "unit": "Gwave-instr/s", "frac": None,
                    "traffic": traffic, "note_counters": f"no PMC record {key!r} in profiles/traffic.json"}
        # The ceiling of THIS instruction mix, measured two ways (profiles/r04_issue_model.json):
        #  (a) tools/mix_probe.py: the step path of the shipped kernel - the same instructions with the same operand forms -
        #      replayed with every dependency removed, at the kernel's own residency (four waves per SIMD, 8 192 workgroups):
        #      the rate at which this hardware issues this mix when nothing but issue limits it.  peak_measured_mix is that
        #      rate (the better of two orders), frac_of_measured_mix = achieved / that.
        #  (b) tools/issue_model.py: the sum over the path of each opcode's stand-alone issue cost (tools/issue_cost.hip, cycles
        #      of the clock measured inside each microbenchmark launch; an SGPR source operand halves a full-rate opcode's
        #      rate): sum_of_standalone_costs.
        # Both say the nominal 2-cycle peak (frac) is out of reach for this mix whatever its schedule: only 47 % of the
        # step's VALU instructions are full-rate all-VGPR ones.
        imf = os.path.join(ROOT, "profiles", "r04_issue_model.json")
        if wl == "cfg3" and os.path.exists(imf) and roof.get("achieved"):
            with open(imf) as f:
                im = json.load(f)
            mm = im.get("measured_mix") or {}
            per_inst = im.get("floor_cycles_per_valu_instruction")
            if mm.get("ns_of_simd_time_per_wave_step") and mm.get("valu_per_iteration"):
                peak_mix = 1024 * mm["valu_per_iteration"] / mm["ns_of_simd_time_per_wave_step"]  # Gwave-instr/s
                roof["peak_measured_mix"] = peak_mix
                roof["frac_of_measured_mix"] = roof["achieved"] / peak_mix
                roof["measured_mix"] = {
                    "replay_ns_of_simd_time_per_wave_step": mm["ns_of_simd_time_per_wave_step"],
                    "replay_program_order_ns": mm.get("program_order_ns"), "replay_spread_evenly_ns": mm.get("spread_evenly_ns"),
                    "kernel_ns_of_simd_time_per_wave_step_without_swap_events": im.get("ns_per_wave_step"),
                    "valu_on_step_path": im.get("valu_on_path"), "class_counts": im.get("class_counts"),
                    "sum_of_standalone_costs_cycles_per_wave_step": im.get("floor_cycles_per_wave_step"),
                    "sum_of_standalone_costs_cycles_per_valu_instruction": per_inst,
                    "kernel_cycles_per_wave_step_without_swap_events": im.get("measured_cycles_per_wave_step"),
                    "pmc_check": im.get("pmc_check"), "model_of_this_build": im.get("lib_sha256") == lib_sha,
                    "source": "profiles/r04_issue_model.txt / .json (tools/issue_model.py), profiles/r04_mix_probe.json "
                              "(tools/mix_probe.py), profiles/r04_issue_costs.json (tools/issue_cost.hip), "
                              "profiles/r04_residency.json (tools/residency_probe.hip)",
                    "note": "frac (nominal) prices every VALU instruction at 2 cycles.  The replay of this kernel's own step "
                            "path without dependencies takes 3 480-3 530 cycles of SIMD time per wave-step, the sum of the "
                            "stand-alone opcode costs is 3 610, the kernel itself takes 3 360 (a launch without swap "
                            "events): the kernel issues its mix 3-4 % FASTER than the dependency-free replay does, so "
                            "frac_of_measured_mix is slightly above one.  Read: the step is at the issue ceiling of its "
                            "instruction mix; no schedule of these instructions is faster on this hardware; only a "
                            "different mix (fewer or cheaper instructions) is."}
        roof.update({
            "kernel": rec.get("kernel") or f"fused step kernel <{type(target).__name__}, {alg.proposal_dist.get_name()}, dim {dim}, "
                                           "production>, form chosen by the C ABI",
            "kernel_ms": kernel_ms, "lib_sha256": lib_sha,
            "hbm_counter_traffic": None if traffic is None else {
                "bytes_per_launch": traffic, "GBps": traffic / (kernel_ms * 1e-3) / 1e9,
                "frac_of_hbm_peak": traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
            "hbm_streaming_accounting": {
                "algorithmic_bytes_per_launch": alg_bytes, "GBps": achieved, "ratio_to_hbm_peak": achieved / HBM_PEAK_GBPS,
                "measured_copy_GBps": copy_gbps,
                "note": "SURVEY 8(d) accounting: (8*dim+24) B per chain-MH-step as if every step streamed its state through "
                        "HBM.  The fused kernel never moves those bytes (state stays in registers for the whole launch), "
                        "so a ratio above 1 is possible and is NOT a roofline fraction."},
            "hbm_stream_inner1": inner1,
        })
        if inner1 is not None and copy_gbps:
            # the same box's plain device-to-device copy (torch's copy_ of 1 GiB) as a second denominator: the
            # bytes really moved against the bytes a copy moves in the same time
            inner1["frac_of_measured_copy"] = inner1["achieved_touched"] / copy_gbps
            inner1["frac_of_measured_copy_algorithmic"] = inner1["achieved"] / copy_gbps
        out = {
            "metric": "chain-MH-steps/sec", "value": value, "unit": "chain-MH-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": desc,
                "dim": dim, "temps": T, "ladders_per_gpu": C, "mh_steps_per_launch": args.inner,
                "swap_every": args.swap_every, "swap_mode": "exchange", "swap_order": args.swap_order,
                "rng": "Philox4x32-10 in-kernel", "sharding": f"{world} x {C} independent ladders, no data-path collective",
                "collective_backend": (backend if distributed else None),
                "world_size": (dist.get_world_size() if distributed else 1),
            },
            # one entry per rank: its device and its own launch times (value uses the MAX wall clock over ranks)
            "ranks": ranks, "distinct_devices": distinct,
            "roofline": roof,
            "summary": {
                "acceptance_rate_cold": float(summary["acceptance_rate"][0]),
                "acceptance_rate_hot": float(summary["acceptance_rate"][-1]),
                "esjd_cold": float(summary["esjd"][0]),
                "swap_acceptance_rate": summary["swap_acceptance_rate"],
                "replicas": summary["n_replicas"],
            },
        }
        if others is not None:
            out["other_single_gpu_readings"] = others
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
