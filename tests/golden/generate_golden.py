"""Generates the golden fixtures under tests/golden/ by importing and RUNNING the real reference
(aidanmrli/rwm-pt-pytorch mounted read-only at /root/reference) on CPU in the build container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_golden.py

Fixtures hold data only (inputs and the reference's outputs); no reference source is copied.  The
reference is not available on the GPU box, so tests read the committed .npz / .json files.

What is captured:
  logdensity.npz   fp32 `log_density` of every in-scope target at fixed and random points
  proposals.npz    raw torch randoms and the increments `proposal.sample()` makes from them
  rwm_*.npz        full RandomWalkMH_GPU_Optimized trajectories + the random tensors they consumed
  pt_*.npz         full ParallelTemperingRWM_GPU_Optimized trajectories + their random tensors
  pt_sweep.npz     `_attempt_all_swaps()` called on its own on a batch of ladders
  superfunnel.npz  SuperFunnelTorch.log_density on ragged synthetic data
  numpy_baseline.json / numpy_*.npz   the NumPy CPU samplers (algorithms/rwm.py, pt_rwm.py)
"""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

with contextlib.redirect_stdout(io.StringIO()):
    import algorithms as ref_alg  # noqa: E402
    import interfaces as ref_if  # noqa: E402
    import proposal_distributions as ref_prop  # noqa: E402
    import target_distributions as ref_tgt  # noqa: E402

CPU = torch.device("cpu")


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}: {os.path.getsize(path) / 1024:.0f} KiB")


# ------------------------------------------------------------------------------------------------
# targets: (key, constructor, dim)
# ------------------------------------------------------------------------------------------------
def make_targets():
    torch.manual_seed(1234)  # scaled variants draw their scaling factors from the global torch RNG
    T = {}
    T["rc15_d30"] = ref_tgt.RoughCarpetDistributionTorch(30, device="cpu", mode_centers=[-15.0, 0.0, 15.0])
    T["rc5_d30"] = ref_tgt.RoughCarpetDistributionTorch(30, device="cpu")
    T["rc4_d20"] = ref_tgt.RoughCarpetDistributionTorch(20, device="cpu", mode_centers=[-4.0, 0.0, 4.0],
                                                        mode_weights=[0.2, 0.5, 0.3])
    T["rc15s_d10"] = ref_tgt.RoughCarpetDistributionTorch(10, scaling=True, device="cpu",
                                                          mode_centers=[-15.0, 0.0, 15.0])
    T["tm_d50"] = ref_tgt.ThreeMixtureDistributionTorch(50, device="cpu")
    c15 = [[-15.0] + [0.0] * 29, [0.0] * 30, [15.0] + [0.0] * 29]
    T["tm15_d30"] = ref_tgt.ThreeMixtureDistributionTorch(30, device="cpu", mode_centers=c15)
    cg = [list(np.linspace(-3, 1, 10)), list(np.linspace(0.5, -0.5, 10)), list(np.linspace(2, 4, 10))]
    T["tms_d10"] = ref_tgt.ThreeMixtureDistributionTorch(10, scaling=True, device="cpu", mode_centers=cg,
                                                         mode_weights=[0.2, 0.3, 0.5])
    T["full_d30"] = ref_tgt.FullRosenbrockTorch(30, device="cpu")
    T["full_d10"] = ref_tgt.FullRosenbrockTorch(10, a_coeff=0.1, b_coeff=2.0, mu=torch.linspace(0.5, 1.5, 9), device="cpu")
    T["even_d30"] = ref_tgt.EvenRosenbrockTorch(30, device="cpu")
    T["hyb_3_5"] = ref_tgt.HybridRosenbrockTorch(3, 5, device="cpu")
    T["hyb_5_4"] = ref_tgt.HybridRosenbrockTorch(5, 4, device="cpu")
    T["gamma_d50"] = ref_tgt.IIDGammaTorch(50, device="cpu")
    T["gamma_d5"] = ref_tgt.IIDGammaTorch(5, shape=3.5, scale=0.7, device="cpu")
    T["beta_d50"] = ref_tgt.IIDBetaTorch(50, device="cpu")
    T["beta_d5"] = ref_tgt.IIDBetaTorch(5, alpha=1.5, beta=4.0, device="cpu")
    # section 8f-3 targets
    T["mvn_d50"] = ref_tgt.MultivariateNormalTorch(50, device="cpu")
    T["mvnd_d8"] = ref_tgt.MultivariateNormalTorch(8, mean=list(np.linspace(-1, 1, 8)),
                                                   cov=np.diag(np.linspace(0.3, 2.5, 8)).tolist(), device="cpu")
    T["smvn_d20"] = ref_tgt.ScaledMultivariateNormalTorch(20, device="cpu", seed=77)
    T["cube_d5"] = ref_tgt.HypercubeTorch(5, device="cpu")
    T["cube2_d3"] = ref_tgt.HypercubeTorch(3, left_boundary=-2.0, right_boundary=1.5, device="cpu")
    T["funnel_d10"] = ref_tgt.NealFunnelTorch(10, device="cpu")
    T["funnel_d1"] = ref_tgt.NealFunnelTorch(1, mu_v=0.5, sigma_v_sq=4.0, device="cpu")
    return T


def target_params(key, t):
    """The constructor-level parameters a test needs to rebuild the same target."""
    p = {"class": type(t).__name__, "dim": int(t.dim), "name": t.get_name()}
    for attr in ("modes", "weights", "means", "mixing_weights", "scaling_factors", "mu", "a_coeff", "b_coeff",
                 "shape", "scale", "alpha", "beta", "mean", "cov", "log_norm_const", "left_boundary", "right_boundary",
                 "log_uniform_density", "mu_v", "sigma_v_sq", "mu_z"):
        if hasattr(t, attr):
            p[attr] = np.asarray(getattr(t, attr).detach().cpu().numpy(), dtype=np.float32)
    for attr in ("n1", "n2"):
        if hasattr(t, attr):
            p[attr] = int(getattr(t, attr))
    return p


def points_for(key, t, rng):
    D = t.dim
    anchors = np.stack([np.linspace(-2, 2, D), np.linspace(-16, 16, D), np.zeros(D), np.full(D, 0.5)])
    if key.startswith("cube"):
        lo, hi = float(t.left_boundary), float(t.right_boundary)
        rnd = rng.uniform(lo - 0.2 * (hi - lo), hi + 0.2 * (hi - lo), size=(60, D))
        rnd[:20] = rng.uniform(lo, hi, size=(20, D))
        rnd[0, 0], rnd[1, 0] = lo, hi  # the closed boundary belongs to the support
    elif key.startswith(("mvn", "smvn", "funnel")):
        rnd = rng.normal(0.0, 2.0, size=(60, D))
    elif key.startswith("beta"):
        rnd = rng.uniform(0.01, 0.99, size=(60, D))
        rnd[:6, 0] = [-0.1, 0.0, 1.0, 1.2, 0.5, 1e-6]  # outside / on the boundary of the support
    elif key.startswith("gamma"):
        rnd = rng.gamma(2.0, 3.0, size=(60, D))
        rnd[:4, -1] = [-1.0, 0.0, 1e-6, 50.0]
    elif key.startswith(("full", "even", "hyb")):
        rnd = rng.normal(0.8, 0.7, size=(60, D))
    else:
        centers = rng.choice([-15.0, -5.0, 0.0, 5.0, 15.0], size=(60, D))
        rnd = centers + rng.normal(0, 1.3, size=(60, D))
    return np.concatenate([anchors, rnd]).astype(np.float32)


def gen_logdensity(T):
    print("log-density fixtures")
    rng = np.random.default_rng(2024)
    arrays, meta = {}, {}
    for key, t in T.items():
        x = points_for(key, t, rng)
        with torch.no_grad():
            lp = t.log_density(torch.tensor(x)).numpy().astype(np.float32)
            lp1 = np.array([float(t.log_density(torch.tensor(x[i]))) for i in range(4)], dtype=np.float32)
        assert np.allclose(lp[:4], lp1, rtol=1e-5, atol=1e-5, equal_nan=True)
        arrays[f"{key}__x"] = x
        arrays[f"{key}__logp"] = lp
        for k, v in target_params(key, t).items():
            if isinstance(v, np.ndarray):
                arrays[f"{key}__p_{k}"] = v
            else:
                meta.setdefault(key, {})[k] = v
    arrays["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save("logdensity.npz", **arrays)


# ------------------------------------------------------------------------------------------------
# proposals
# ------------------------------------------------------------------------------------------------
def gen_proposals():
    print("proposal fixtures")
    arrays = {}
    n = 64
    for D, beta in ((30, 1.0), (7, 0.37), (50, 0.01)):
        tag = f"d{D}_b{beta}"
        # Normal: randn(n, D) * std
        p = ref_prop.NormalProposal(D, 2.38**2 / D, beta, CPU, torch.float32)
        torch.manual_seed(7)
        z = torch.randn((n, D))
        torch.manual_seed(7)
        inc = p.sample(n)
        arrays[f"normal_{tag}__raw"], arrays[f"normal_{tag}__inc"] = z.numpy(), inc.numpy()
        arrays[f"normal_{tag}__std"] = np.float32(p.std_dev.item())
        # Laplace: rand(n, D) - 0.5 -> inverse CDF
        bv = torch.linspace(0.05, 0.6, D)
        p = ref_prop.LaplaceProposal(D, bv, beta, CPU, torch.float32)
        torch.manual_seed(8)
        u = torch.rand((n, D))
        torch.manual_seed(8)
        inc = p.sample(n)
        arrays[f"laplace_{tag}__raw"], arrays[f"laplace_{tag}__inc"] = u.numpy(), inc.numpy()
        arrays[f"laplace_{tag}__base_var"], arrays[f"laplace_{tag}__scale"] = bv.numpy(), p.scale_vector.numpy()
        # UniformRadius: randn(n, D) then rand(n, 1)
        p = ref_prop.UniformRadiusProposal(D, 1.7, beta, CPU, torch.float32)
        torch.manual_seed(9)
        g = torch.randn((n, D))
        ur = torch.rand((n, 1))
        torch.manual_seed(9)
        inc = p.sample(n)
        arrays[f"uniform_{tag}__raw"] = torch.cat([g, ur], dim=1).numpy()
        arrays[f"uniform_{tag}__inc"] = inc.numpy()
        arrays[f"uniform_{tag}__radius"] = np.float32(p.effective_radius.item())
    save("proposals.npz", **arrays)


# ------------------------------------------------------------------------------------------------
# RWM trajectories (torch class on CPU)
# ------------------------------------------------------------------------------------------------
def make_proposal(kind, D, beta, scale):
    if kind == "Normal":
        return ref_prop.NormalProposal(D, scale, beta, CPU, torch.float32), {"base_variance_scalar": scale}
    if kind == "Laplace":
        bv = torch.full((D,), scale)
        return ref_prop.LaplaceProposal(D, bv, beta, CPU, torch.float32), {"base_variance_vector": bv.numpy()}
    return ref_prop.UniformRadiusProposal(D, scale, beta, CPU, torch.float32), {"base_radius": scale}


def raw_for(kind, total, D):
    """The raw torch draws `proposal.sample(total)` makes, in its order."""
    if kind == "Normal":
        return torch.randn((total, D))
    if kind == "Laplace":
        return torch.rand((total, D))
    g = torch.randn((total, D))
    return torch.cat([g, torch.rand((total, 1))], dim=1)


def gen_rwm(T):
    print("RWM trajectory fixtures")
    cases = [  # (fixture, target key, proposal kind, scale, beta, N, burn_in, seed)
        ("rwm_rc15_normal", "rc15_d30", "Normal", 2.38**2 / 30, 1.0, 1500, 200, 42),
        ("rwm_rc4_normal_beta", "rc4_d20", "Normal", 2.38**2 / 20, 0.5, 800, 0, 43),
        ("rwm_even_laplace", "even_d30", "Laplace", 0.02, 1.0, 1500, 200, 44),
        ("rwm_tm_uniform", "tm_d50", "UniformRadius", 2.2, 1.0, 1000, 100, 45),
        ("rwm_full_normal", "full_d10", "Normal", 0.05, 1.0, 1000, 100, 46),
        ("rwm_hyb_laplace", "hyb_3_5", "Laplace", 0.05, 1.0, 1000, 100, 47),
        ("rwm_gamma_normal", "gamma_d5", "Normal", 1.5, 1.0, 1000, 100, 48),
        ("rwm_beta_uniform", "beta_d5", "UniformRadius", 0.35, 1.0, 1000, 100, 49),
        ("rwm_rc15s_normal", "rc15s_d10", "Normal", 2.38**2 / 10, 1.0, 1000, 100, 50),
        ("rwm_tms_normal", "tms_d10", "Normal", 2.38**2 / 10, 1.0, 1000, 100, 51),
        ("rwm_mvn_laplace", "mvn_d50", "Laplace", 2.38**2 / 50, 1.0, 800, 100, 52),
        ("rwm_smvn_normal", "smvn_d20", "Normal", 0.3, 1.0, 800, 100, 53),
        ("rwm_cube_uniform", "cube_d5", "UniformRadius", 0.4, 1.0, 800, 100, 54),
        ("rwm_funnel_normal", "funnel_d10", "Normal", 0.2, 1.0, 800, 100, 55),
    ]
    for name, tkey, kind, scale, beta, N, burn, seed in cases:
        t = T[tkey]
        D = t.dim
        total = burn + N
        prop, pparams = make_proposal(kind, D, beta, scale)
        np.random.seed(seed)  # initial state comes from the global NumPy RNG
        alg = quiet(ref_alg.RandomWalkMH_GPU_Optimized, dim=D, target_dist=t, beta=beta, burn_in=burn, device="cpu",
                    pre_allocate_steps=N, proposal_distribution=prop)
        x0 = np.asarray(alg.chain[0], dtype=np.float64)
        torch.manual_seed(seed)
        raw = raw_for(kind, total, D)
        u = torch.rand(total)
        torch.manual_seed(seed)
        samples = quiet(alg.generate_samples, N)
        chain = alg.get_chain_gpu().numpy()
        assert chain.shape == (total + 1, D) and samples.shape == (N, D)
        arrays = dict(
            x0=x0, raw=raw.numpy(), u=u.numpy(), chain=chain, logp_chain=alg.get_log_densities_gpu().numpy(),
            num_acceptances=np.int64(alg.num_acceptances), acceptance_rate=np.float64(alg.acceptance_rate),
            esjd=np.float64(alg.expected_squared_jump_distance_gpu()), beta=np.float64(beta), burn_in=np.int64(burn),
            n_samples=np.int64(N), proposal_kind=np.array(kind), target_key=np.array(tkey),
            alg_name=np.array(alg.get_name()),
        )
        for k, v in pparams.items():
            arrays[f"pp_{k}"] = np.asarray(v, dtype=np.float32)
        save(name + ".npz", **arrays)


# ------------------------------------------------------------------------------------------------
# PT trajectories (torch class on CPU; the reference's swap is the Q1 row copy, sequential order)
# ------------------------------------------------------------------------------------------------
def gen_pt(T, only=None):
    print("PT trajectory fixtures")
    geo8 = [1.0, 0.5, 0.25, 0.125, 0.0625, 0.03125, 0.015625, 0.01]
    cases = [  # (fixture, target key, var, ladder, swap_every, N, burn_in, seed[, dtype])
        ("pt_rc15_geo8", "rc15_d30", 2.38**2 / 30, geo8, 10, 400, 50, 142),
        # the same run with dtype=torch.float64 (experiment_pt_GPU.py:236 --use_double_precision): states, increments and
        # the Cholesky factors in double; the log-densities come out double too (type promotion in log_density)
        ("pt_rc15_geo8_f64", "rc15_d30", 2.38**2 / 30, geo8, 10, 400, 50, 147, torch.float64),
        ("pt_rc5_fine12", "rc5_d30", 2.38**2 / 30, [float(0.05 ** (i / 11)) for i in range(12)], 5, 300, 20, 143),
        ("pt_tm15_t32", "tm15_d30", 2.38**2 / 30, [float(0.01 ** (i / 31)) for i in range(32)], 10, 150, 0, 144),
        ("pt_even_t5", "even_d30", 0.01, [1.0, 0.7, 0.45, 0.3, 0.2], 3, 300, 31, 145),
        ("pt_hyb_t4", "hyb_5_4", 0.02, [1.0, 0.6, 0.35, 0.2], 4, 300, 10, 146),
    ]
    for name, tkey, var, ladder, se, N, burn, seed, *rest in cases:
        dtype = rest[0] if rest else torch.float32
        if only is not None and name not in only:
            continue
        t = T[tkey]
        D, Tn = t.dim, len(ladder)
        total = burn + N
        np.random.seed(seed)
        alg = quiet(ref_alg.ParallelTemperingRWM_GPU_Optimized, D, var, t, True, beta_ladder=ladder, swap_every=se,
                    burn_in=burn, device="cpu", pre_allocate_steps=N, dtype=dtype)
        x0 = alg.current_states[0].numpy().copy()
        # the random tensors generate_samples draws, in its order (pt_rwm_gpu_optimized.py:710-723)
        torch.manual_seed(seed)
        max_swaps = total // se * (Tn - 1) + 100
        swap_r = torch.rand(max_swaps + 100)
        mh_u = torch.rand(total + 10, Tn)
        z = torch.randn(total + 10, Tn, D, dtype=dtype)
        torch.manual_seed(seed)
        cold = quiet(alg.generate_samples, N)
        assert alg.current_states.dtype == dtype and cold.dtype == dtype
        chains = alg.pre_allocated_chains.numpy()  # [T, total+1, D]
        logps = alg.pre_allocated_log_densities.numpy()  # [T, total+1]
        assert cold.shape == (N, D)
        n_events = alg.num_swap_attempts // (Tn - 1)
        arrays = dict(
            x0=x0, beta_ladder=np.asarray(ladder, dtype=np.float64), var=np.float64(var),
            # engine layout: step i uses increments row i+1 (1-based, Q3) and accept uniforms row i
            ext_prop=z[1:total + 1].numpy(), ext_u=mh_u[:total].numpy(),
            ext_swap_u=swap_r[:n_events * (Tn - 1)].reshape(n_events, Tn - 1).numpy(),
            chains=chains, logp_chains=logps,
            num_swap_attempts=np.int64(alg.num_swap_attempts), num_swap_acceptances=np.int64(alg.num_swap_acceptances),
            swap_acceptance_rate=np.float64(alg.swap_acceptance_rate), pt_esjd=np.float64(alg.pt_esjd),
            esjd=np.float64(alg.expected_squared_jump_distance_gpu()), swap_every=np.int64(se), burn_in=np.int64(burn),
            n_samples=np.int64(N), target_key=np.array(tkey), alg_name=np.array(alg.get_name()),
        )
        save(name + ".npz", **arrays)


def gen_sweep(T):
    """`_attempt_all_swaps()` called on its own (pt_rwm_gpu_optimized.py:594-633, as tests/debug_pt_performance.py:156
    does): a batch of independent ladders, each given random states, the matching log-densities and its own swap
    uniforms; the reference's states / log-densities / counters after ONE sweep are recorded."""
    print("stand-alone swap sweep fixture")
    t = T["rc15_d30"]
    D = t.dim
    ladder = [float(0.01 ** (i / 11)) for i in range(12)]
    Tn, n_lad = len(ladder), 48
    rng = np.random.default_rng(777)
    st0 = rng.normal(0.0, 6.0, size=(n_lad, Tn, D)).astype(np.float32)
    us = rng.random(size=(n_lad, Tn - 1)).astype(np.float32)
    out_state, out_logp, in_logp, acc, att, rate, esjd = [], [], [], [], [], [], []
    for c in range(n_lad):
        np.random.seed(5)
        alg = quiet(ref_alg.ParallelTemperingRWM_GPU_Optimized, D, 2.38**2 / D, t, True, beta_ladder=ladder,
                    swap_every=10, burn_in=0, device="cpu", pre_allocate_steps=4)
        alg.current_states = torch.from_numpy(st0[c].copy())
        alg.current_log_densities = t.log_density(alg.current_states).to(torch.float32)
        in_logp.append(alg.current_log_densities.numpy().copy())
        alg.precomputed_swap_randoms = torch.from_numpy(us[c].copy())
        alg.swap_random_index = 0
        alg._attempt_all_swaps()
        out_state.append(alg.current_states.numpy().copy())
        out_logp.append(alg.current_log_densities.numpy().copy())
        acc.append(alg.num_swap_acceptances)
        att.append(alg.num_swap_attempts)
        rate.append(alg.swap_acceptance_rate)
        esjd.append(alg.pt_esjd)
    save("pt_sweep.npz", state_in=st0, logp_in=np.asarray(in_logp, np.float32), swap_u=us,
         beta_ladder=np.asarray(ladder, np.float64), state_out=np.asarray(out_state, np.float32),
         logp_out=np.asarray(out_logp, np.float32), num_swap_acceptances=np.asarray(acc, np.int64),
         num_swap_attempts=np.asarray(att, np.int64), swap_acceptance_rate=np.asarray(rate, np.float64),
         pt_esjd=np.asarray(esjd, np.float64), target_key=np.array("rc15_d30"))


def gen_superfunnel():
    """SuperFunnelTorch.log_density (funnel_torch.py:193-291) on ragged synthetic data at random states, some with a
    non-positive tau (log density -inf)."""
    print("SuperFunnel log-density fixture")
    J, K, n_j = 3, 2, [5, 7, 4]
    rng = np.random.default_rng(99)
    X = [rng.normal(size=(n, K)).astype(np.float32) for n in n_j]
    Y = [(rng.random(n) < 0.5).astype(np.float32) for n in n_j]
    t = quiet(ref_tgt.SuperFunnelTorch, J, K, [torch.from_numpy(x) for x in X], [torch.from_numpy(y) for y in Y],
              device="cpu")
    dim = J + J * K + 1 + K + 2
    th = rng.normal(0.0, 1.5, size=(40, dim)).astype(np.float32)
    th[:, -2:] = np.abs(th[:, -2:]) + 0.05
    th[3, -1] = -0.3
    th[7, -2] = 0.0
    th[11, -2:] = 1e-10
    ref = t.log_density(torch.from_numpy(th)).numpy()
    save("superfunnel.npz", J=np.int64(J), K=np.int64(K), n_j=np.asarray(n_j, np.int64), X=np.concatenate(X, 0),
         Y=np.concatenate(Y, 0), theta=th, log_density=ref, target_name=np.array(t.get_name()), dim=np.int64(t.dim))


# ------------------------------------------------------------------------------------------------
# NumPy CPU samplers (BASELINE config 1 and a small PT run)
# ------------------------------------------------------------------------------------------------
def gen_numpy():
    print("NumPy baseline fixtures")
    out = {}
    # config 1, exactly as experiment.py drives it
    t0 = time.time()
    sim = quiet(ref_if.MCMCSimulation, dim=20, sigma=2.38**2 / 20, num_iterations=100000,
                algorithm=ref_alg.RandomWalkMH, target_dist=ref_tgt.RoughCarpetDistribution(20), seed=42)
    import tqdm

    class _NoBar:
        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def update(self, n):
            pass

    ref_if.simulation.tqdm.tqdm = _NoBar
    quiet(sim.generate_samples)
    out["config1"] = {
        "dim": 20, "sigma": 2.38**2 / 20, "num_iterations": 100000, "seed": 42,
        "num_acceptances": int(sim.algorithm.num_acceptances), "acceptance_rate": float(sim.acceptance_rate()),
        "esjd": float(sim.expected_squared_jump_distance()), "seconds": time.time() - t0,
    }
    print("   config1:", out["config1"])
    # short RWM run with the chain kept
    sim = quiet(ref_if.MCMCSimulation, dim=30, sigma=2.38**2 / 30, num_iterations=3000, algorithm=ref_alg.RandomWalkMH,
                target_dist=ref_tgt.RoughCarpetDistribution(30), seed=7)
    quiet(sim.generate_samples)
    save("numpy_rwm_d30.npz", chain=np.array(sim.algorithm.chain), acceptance_rate=np.float64(sim.acceptance_rate()),
         esjd=np.float64(sim.expected_squared_jump_distance()))
    # short PT run (swap_every = 20 hard-coded in the reference)
    ladder = [1.0, 0.5, 0.25, 0.125, 0.0625, 0.03125, 0.015625, 0.01]
    sim = quiet(ref_if.MCMCSimulation, dim=30, sigma=2.38**2 / 30, num_iterations=1500,
                algorithm=ref_alg.ParallelTemperingRWM, target_dist=ref_tgt.RoughCarpetDistribution(30), seed=11,
                beta_ladder=ladder)
    quiet(sim.generate_samples)
    a = sim.algorithm
    save("numpy_pt_d30.npz", chain=np.array(a.chain), beta_ladder=np.array(ladder),
         swap_acceptance_rate=np.float64(a.acceptance_rate), pt_esjd=np.float64(a.pt_esjd),
         num_swap_attempts=np.int64(a.num_swap_attempts), num_swap_acceptances=np.int64(a.num_acceptances),
         esjd=np.float64(sim.expected_squared_jump_distance()),
         final_states=np.array([c.chain[-1] for c in a.chains]))
    with open(os.path.join(OUT, "numpy_baseline.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    which = sys.argv[1:] or ["logdensity", "proposals", "rwm", "pt", "sweep", "superfunnel", "numpy"]
    T = make_targets()
    if "logdensity" in which:
        gen_logdensity(T)
    if "proposals" in which:
        gen_proposals()
    if "rwm" in which:
        gen_rwm(T)
    if "pt" in which:
        gen_pt(T)
    if "pt_f64" in which:  # only the float64 trajectory (the other fixtures are left untouched)
        gen_pt(T, only={"pt_rc15_geo8_f64"})
    if "sweep" in which:
        gen_sweep(T)
    if "superfunnel" in which:
        gen_superfunnel()
    if "numpy" in which:
        gen_numpy()
