"""Host-side logic of the drop-in classes: constructor contracts, names, ladders, initial states, proposal
parameters, harness dispatch.  Runs without a GPU (objects are built on device='cpu'; any attempt to sample
must raise, never fall back)."""
import numpy as np
import pytest
import torch

import helpers as H
import ptrwm_hip
from algorithms import (ParallelTemperingRWM_GPU_Optimized, RandomWalkMH_GPU_Optimized, RWM_GPU_Optimized,
                        geometric_beta_ladder)
from algorithms.sharding import pack_summary, shard_range, unpack_summary
from interfaces import MCMCSimulation_GPU, MHAlgorithm, TargetDistribution, TorchTargetDistribution, initial_state_for
from proposal_distributions import LaplaceProposal, NormalProposal, ProposalDistribution, UniformRadiusProposal
from target_distributions import (EvenRosenbrockTorch, FullRosenbrockTorch, HybridRosenbrockTorch, IIDBetaTorch,
                                  IIDGammaTorch, RoughCarpetDistributionTorch, ThreeMixtureDistributionTorch)

CPU = torch.device("cpu")


def test_target_names_and_validation():
    assert RoughCarpetDistributionTorch(4, device="cpu").get_name() == "RoughCarpetTorch"
    assert RoughCarpetDistributionTorch(4, device="cpu", mode_centers=[-15.0, 0.0, 15.0]).get_name() == "RoughCarpetTorchCustom"
    assert RoughCarpetDistributionTorch(4, scaling=True, device="cpu").get_name() == "RoughCarpetTorchScaled"
    assert ThreeMixtureDistributionTorch(4, device="cpu").get_name() == "ThreeMixtureTorch"
    assert ThreeMixtureDistributionTorch(3, device="cpu", mode_weights=[0.2, 0.3, 0.5], scaling=True).get_name() == \
        "ThreeMixtureTorchCustomScaled"
    assert FullRosenbrockTorch(4, device="cpu").get_name() == "FullRosenbrockTorch"
    assert EvenRosenbrockTorch(4, device="cpu").get_name() == "EvenRosenbrockTorch"
    assert HybridRosenbrockTorch(3, 5, device="cpu").get_name() == "HybridRosenbrockTorch(n1=3, n2=5, a=0.05, b=5.00, mu=1.00)"
    assert HybridRosenbrockTorch(3, 5, device="cpu").dim == 11
    assert IIDGammaTorch(5, device="cpu").get_name() == "IIDGammaTorch" and IIDBetaTorch(5, device="cpu").get_name() == "IIDBetaTorch"
    for bad in (dict(mode_centers=[0.0, 1.0]), dict(mode_weights=[0.5, 0.5]), dict(mode_weights=[0.5, 0.6, 0.2]),
                dict(mode_weights=[1.2, -0.1, -0.1]), dict(mode_centers=[[0.0], 1.0, 2.0])):
        with pytest.raises(ValueError):
            RoughCarpetDistributionTorch(4, device="cpu", **bad)
    with pytest.raises(ValueError):
        ThreeMixtureDistributionTorch(4, device="cpu", mode_centers=[[0.0] * 3] * 3)
    with pytest.raises(ValueError):
        EvenRosenbrockTorch(5, device="cpu")
    with pytest.raises(ValueError):
        FullRosenbrockTorch(1, device="cpu")
    with pytest.raises(ValueError):
        HybridRosenbrockTorch(1, 2, device="cpu")
    with pytest.raises(ValueError):
        FullRosenbrockTorch(4, mu=torch.zeros(7), device="cpu")
    with pytest.raises(NotImplementedError):
        FullRosenbrockTorch(4, device="cpu").draw_samples_torch(3)


def test_engine_descriptors_fold_parameters_like_the_reference():
    """engine_target(): exactly the constants the reference classes precompute (fp32)."""
    t = RoughCarpetDistributionTorch(30, device="cpu", mode_centers=[-15.0, 0.0, 15.0])
    d = t.engine_target()
    assert d.kind == ptrwm_hip.TARGET_ROUGH_CARPET and d.dim == 30 and d.vec0 is None
    np.testing.assert_allclose(d.p[:3], [-15, 0, 15])
    np.testing.assert_allclose(d.p[3:6], np.log(np.float32([0.5, 0.3, 0.2])), rtol=1e-7)
    tm = ThreeMixtureDistributionTorch(50, device="cpu")
    d = tm.engine_target()
    want = np.float32(-0.5 * 50 * np.log(2 * np.pi)) + np.log(np.float32(1 / 3))
    np.testing.assert_allclose(d.p, [want] * 3, rtol=1e-6)
    assert d.vec0.shape == (150,) and d.vec1 is None
    g = IIDGammaTorch(50, device="cpu").engine_target()
    from math import lgamma, log
    assert g.p[2] == pytest.approx(50 * (lgamma(2.0) + 2.0 * log(3.0)), rel=1e-6)
    b = IIDBetaTorch(50, device="cpu").engine_target()
    assert b.p[2] == pytest.approx(50 * (lgamma(5.0) - lgamma(2.0) - lgamma(3.0)), rel=1e-6)
    h = HybridRosenbrockTorch(3, 5, device="cpu").engine_target()
    assert h.ip == (3, 5) and h.p == pytest.approx((0.05, 5.0, 1.0))

    class Mine(TorchTargetDistribution):
        def density(self, x):
            return x

        def log_density(self, x):
            return x

        def get_name(self):
            return "mine"

    with pytest.raises(NotImplementedError, match="fused-kernel"):
        Mine(3, device="cpu").engine_target()


def test_initial_state_rule():
    """interfaces/metropolis.py:21-64 of the reference: name-dependent start, global NumPy RNG."""
    np.random.seed(0)
    a = initial_state_for(IIDBetaTorch(6, device="cpu"), 6)
    np.random.seed(0)
    assert np.array_equal(a, np.random.uniform(0.2, 0.8, size=6).astype(np.float32)) and a.dtype == np.float32
    np.random.seed(1)
    a = initial_state_for(IIDGammaTorch(6, device="cpu"), 6)
    np.random.seed(1)
    assert np.array_equal(a, 5 + 0.01 * np.random.randn(6))
    assert np.array_equal(initial_state_for(RoughCarpetDistributionTorch(6, device="cpu"), 6), np.zeros(6))
    assert np.array_equal(initial_state_for(ThreeMixtureDistributionTorch(6, device="cpu"), 6), np.zeros(6))
    np.random.seed(2)
    a = initial_state_for(EvenRosenbrockTorch(6, device="cpu"), 6)
    np.random.seed(2)
    assert np.array_equal(a, 0.00000001 * np.random.randn(6))
    alg = MHAlgorithm(6, 1.0, RoughCarpetDistributionTorch(6, device="cpu"))
    assert len(alg.chain) == 1 and alg.get_curr_state().shape == (6,)
    with pytest.raises(NotImplementedError):
        alg.step()


@pytest.mark.parametrize("fixture", sorted(H.RWM_FIXTURE_SEEDS) + sorted(H.PT_FIXTURE_SEEDS))
def test_initial_state_equals_the_references(fixture):
    """a18 pinned to the reference itself: every golden trajectory stores the `x0` the reference sampler started
    from (interfaces/metropolis.py:21-64 after `np.random.seed(seed)`; PT: broadcast to fp32 rows,
    pt_rwm_gpu_optimized.py:478-484).  The drop-in's rule, given the drop-in's own target class (its NAME decides
    the branch), must reproduce it bit for bit - and so must the sampler classes built on top of it."""
    z = H.load(fixture + ".npz")
    key = str(z["target_key"])
    target = H.build_target_class(key, "cpu")
    is_pt = fixture.startswith("pt_")
    seed = (H.PT_FIXTURE_SEEDS if is_pt else H.RWM_FIXTURE_SEEDS)[fixture]
    np.random.seed(seed)
    got = np.asarray(initial_state_for(target, target.dim))
    want = z["x0"]
    if is_pt:  # the reference PT class stores the start as a float32 row
        assert want.dtype == np.float32
        assert np.array_equal(torch.as_tensor(got, dtype=torch.float32).numpy(), want)
    else:
        assert want.dtype == np.float64 and np.array_equal(got.astype(np.float64), want)
        assert got.dtype == (np.float32 if "Beta" in target.get_name() else np.float64)
    # the same through the sampler classes (what a user gets): the constructor consumes the global NumPy RNG once
    np.random.seed(seed)
    if is_pt:
        alg = ParallelTemperingRWM_GPU_Optimized(target.dim, float(z["var"]), target, beta_ladder=list(z["beta_ladder"]),
                                                 device="cpu")
        assert np.array_equal(np.asarray(alg._initial_state, dtype=np.float32), want)
    else:
        alg = RandomWalkMH_GPU_Optimized(target.dim, 0.1, target, device="cpu")
        assert np.array_equal(np.asarray(alg.chain[0], dtype=np.float64), want)
    # ... and it is the first row of the reference's stored chain
    first = z["chains"][:, 0] if is_pt else z["chain"][:1]
    assert np.array_equal(first, np.broadcast_to(np.asarray(want, np.float32), first.shape))


def test_proposal_parameters_and_validation():
    """tests/test_proposals.py:118-140 (ValueError on bad args) and the beta scaling table (:420-456)."""
    for beta in (1.0, 0.5, 0.01):
        n = NormalProposal(5, 0.6, beta, CPU, torch.float32)
        assert float(n.std_dev) == pytest.approx(np.sqrt(0.6 / beta), rel=1e-6)
        bv = torch.tensor([0.1, 0.2, 0.3, 0.4, 0.5])
        lap = LaplaceProposal(5, bv, beta, CPU, torch.float32)
        np.testing.assert_allclose(lap.scale_vector.numpy(), np.sqrt(bv.numpy() / beta / 2), rtol=1e-6)
        u = UniformRadiusProposal(5, 1.5, beta, CPU, torch.float32)
        assert float(u.effective_radius) == pytest.approx(1.5 / np.sqrt(beta), rel=1e-6) and u.inv_dim == 0.2
    assert (n.get_name(), lap.get_name(), u.get_name()) == ("Normal", "Laplace", "UniformRadius")
    with pytest.raises(ValueError):
        NormalProposal(5, -1.0, 1.0, CPU, torch.float32)
    with pytest.raises(ValueError):
        LaplaceProposal(5, torch.ones(4), 1.0, CPU, torch.float32)
    with pytest.raises(ValueError):
        LaplaceProposal(5, torch.tensor([1.0, 1.0, -1.0, 1.0, 1.0]), 1.0, CPU, torch.float32)
    with pytest.raises(ValueError):
        UniformRadiusProposal(5, 0.0, 1.0, CPU, torch.float32)
    # kernel-side description: one temperature keeps the object's own tempered scale ...
    e = NormalProposal(5, 0.6, 0.5, CPU, torch.float32).engine_proposal()
    assert e.kind == ptrwm_hip.PROPOSAL_NORMAL and e.temp_scale.tolist() == [pytest.approx(np.sqrt(1.2), rel=1e-6)]
    # ... a ladder re-tempers the base scale per temperature (pt_rwm_gpu_optimized.py:453-455 of the reference)
    lad = [1.0, 0.25, 0.01]
    e = NormalProposal(5, 0.6, 1.0, CPU, torch.float32).engine_proposal(lad)
    np.testing.assert_allclose(e.temp_scale.numpy(), np.sqrt(np.float32(0.6) / np.float32(lad)), rtol=1e-6)
    e = lap.engine_proposal(lad)
    np.testing.assert_allclose(e.dim_scale.numpy(), np.sqrt(bv.numpy() / 2), rtol=1e-6)
    np.testing.assert_allclose(e.temp_scale.numpy(), [1.0, 2.0, 10.0], rtol=1e-6)
    e = u.engine_proposal(lad)
    np.testing.assert_allclose(e.temp_scale.numpy(), [1.5, 3.0, 15.0], rtol=1e-6)
    assert e.inv_dim == 0.2
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        n.sample(3)


def test_rwm_constructor_contract():
    t = RoughCarpetDistributionTorch(6, device="cpu")
    assert RWM_GPU_Optimized is RandomWalkMH_GPU_Optimized
    with pytest.raises(ValueError, match="Either var"):
        RandomWalkMH_GPU_Optimized(6, target_dist=t, device="cpu")
    a = RandomWalkMH_GPU_Optimized(6, 0.3, t, device="cpu", burn_in=-5, pre_allocate_steps=10)
    assert a.get_name() == "RWM_GPU_FUSED_Normal" and a.burn_in == 0 and a.dtype == torch.float32
    assert a.pre_allocated_chain.shape == (11, 6) and a.chain_index == 0 and a.total_steps == 0
    assert a.acceptance_rate == 0.0 and a.num_acceptances == 0 and a.current_state is None
    assert isinstance(a.proposal_dist, NormalProposal) and float(a.proposal_dist.std_dev) == pytest.approx(np.sqrt(0.3))
    lap = LaplaceProposal(6, torch.full((6,), 0.2), 0.5, CPU, torch.float32)
    b = RandomWalkMH_GPU_Optimized(6, target_dist=t, beta=0.5, device="cpu", proposal_distribution=lap)
    assert b.get_name() == "RWM_GPU_FUSED_Laplace" and b.pre_allocated_chain is None and b.chain_index is None
    # a proposal built for another dtype is rebuilt from its read-back base scale (rwm_gpu_optimized.py:166-200)
    u64 = UniformRadiusProposal(6, 1.5, 0.5, CPU, torch.float64)
    c = RandomWalkMH_GPU_Optimized(6, target_dist=t, beta=0.5, device="cpu", proposal_distribution=u64)
    assert c.proposal_dist.dtype == torch.float32 and float(c.proposal_dist.effective_radius) == pytest.approx(1.5 / np.sqrt(0.5), rel=1e-6)

    class Legacy(TargetDistribution):
        pass

    with pytest.raises(TypeError, match="no GPU path"):
        RandomWalkMH_GPU_Optimized(6, 0.3, Legacy(6), device="cpu")
    # no silent CPU path: sampling on a CPU device raises
    for call in (a.step, lambda: a.generate_samples(5)):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            call()
    with pytest.raises(ValueError, match="Insufficient"):
        a.expected_squared_jump_distance_gpu()


def test_pt_constructor_contract_and_ladders():
    t = RoughCarpetDistributionTorch(6, device="cpu")
    with pytest.warns(UserWarning, match="geometric spacing"):
        a = ParallelTemperingRWM_GPU_Optimized(6, 0.3, t, device="cpu")
    assert a.beta_ladder == [1.0, 0.5, 0.25, 0.125, 0.0625, 0.03125, 0.015625, 0.01] and a.num_chains == 8
    assert a.get_name() == "PT_RWM_GPU_ULTRA_FUSED" and a.swap_every == 100 and a.step_counter == 0
    assert a.beta_tensor.dtype == torch.float32 and a.beta_tensor.shape == (8,)
    assert (a.num_swap_attempts, a.num_swap_acceptances, a.swap_acceptance_rate, a.pt_esjd) == (0, 0, 0.0, 0.0)
    assert len(a.chain) == 1 and np.array_equal(a.chain[0], np.zeros(6))
    # positional call exactly as interfaces/simulation_gpu.py:86-94 makes it
    b = ParallelTemperingRWM_GPU_Optimized(6, 0.3, t, True, device="cpu", pre_allocate_steps=20, beta_ladder=[1.0, 0.3],
                                           swap_acceptance_rate=0.3, burn_in=7, swap_every=5)
    assert b.num_chains == 2 and b.burn_in == 7 and b.ideal_swap_acceptance_rate == 0.3
    assert b.pre_allocated_chains.shape == (2, 28, 6) and b.pre_allocated_log_densities.shape == (2, 28)
    lad = geometric_beta_ladder(32)
    assert len(lad) == 32 and lad[0] == 1.0 and lad[-1] == pytest.approx(0.01) and lad[1] / lad[0] == pytest.approx(lad[5] / lad[4])
    assert geometric_beta_ladder(1) == [1.0]
    with pytest.raises(ValueError):
        ParallelTemperingRWM_GPU_Optimized(6, 0.3, t, beta_ladder=[1.0, 0.5], device="cpu", swap_mode="swap")
    with pytest.raises(ValueError):
        ParallelTemperingRWM_GPU_Optimized(6, 0.3, t, beta_ladder=[1.0, 0.5], device="cpu", swap_order="random")
    with pytest.raises(TypeError):
        ParallelTemperingRWM_GPU_Optimized(6, 0.3, TargetDistribution(6), beta_ladder=[1.0], device="cpu")
    with pytest.raises(NotImplementedError):  # FullRosenbrock has no sampler (as in the reference)
        ParallelTemperingRWM_GPU_Optimized(6, 0.3, FullRosenbrockTorch(6, device="cpu"), iterative_temp_spacing=True,
                                           device="cpu")

    class NoSampler(TorchTargetDistribution):
        density = log_density = lambda self, x: x

        def get_name(self):
            return "nosampler"

    with pytest.raises(NotImplementedError, match="draw_samples_torch"):
        ParallelTemperingRWM_GPU_Optimized(6, 0.3, NoSampler(6, device="cpu"), iterative_temp_spacing=True, device="cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        b.step()
    # more temperatures than a wavefront holds is refused, not silently truncated
    c = ParallelTemperingRWM_GPU_Optimized(6, 0.3, t, beta_ladder=[0.99**i for i in range(257)], device="cpu")
    with pytest.raises(ValueError, match="one workgroup"):
        c.step()
    # dtype=torch.float64 is honoured (states, chains, increments in double: the engine's state_f64 mode), no warning;
    # anything else is refused
    import warnings as _w
    with _w.catch_warnings():
        _w.simplefilter("error")
        d = ParallelTemperingRWM_GPU_Optimized(6, 0.3, t, beta_ladder=[1.0, 0.5], device="cpu", dtype=torch.float64,
                                               pre_allocate_steps=4)
    assert d.dtype == torch.float64 and d.pre_allocated_chains.dtype == torch.float64
    assert d.pre_allocated_log_densities.dtype == torch.float32 and d.beta_tensor.dtype == torch.float32
    with pytest.raises(TypeError, match="float32 or torch.float64"):
        ParallelTemperingRWM_GPU_Optimized(6, 0.3, t, beta_ladder=[1.0, 0.5], device="cpu", dtype=torch.float16)


def test_harness_dispatch_and_proposal_factory():
    t = ThreeMixtureDistributionTorch(5, device="cpu")
    with pytest.raises(ValueError, match="Either sigma"):
        MCMCSimulation_GPU(5, algorithm=RandomWalkMH_GPU_Optimized, target_dist=t, device="cpu")
    s = MCMCSimulation_GPU(5, sigma=0.4, num_iterations=50, algorithm=RandomWalkMH_GPU_Optimized, target_dist=t,
                           device="cpu", burn_in=5, beta_ladder=[0.5], seed=1)
    assert isinstance(s.algorithm, RandomWalkMH_GPU_Optimized) and s.algorithm.beta == 0.5
    assert float(s.algorithm.proposal_dist.std_dev) == pytest.approx(np.sqrt(0.8))
    assert s.algorithm.pre_allocated_chain.shape == (56, 5) and not s.has_run()
    for call in (s.acceptance_rate, s.expected_squared_jump_distance, s.pt_expected_squared_jump_distance):
        with pytest.raises(ValueError, match="not been run"):
            call()
    p = MCMCSimulation_GPU(5, sigma=0.4, num_iterations=50, algorithm=ParallelTemperingRWM_GPU_Optimized, target_dist=t,
                           device="cpu", beta_ladder=[1.0, 0.5, 0.1], swap_every=7, pre_allocate=False)
    assert isinstance(p.algorithm, ParallelTemperingRWM_GPU_Optimized) and p.algorithm.swap_every == 7
    assert p.algorithm.pre_allocated_chains is None and p.algorithm.ideal_swap_acceptance_rate == 0.234
    f = s._create_proposal_distribution
    assert isinstance(f(5, 1.0, {"name": "Laplace", "params": {"base_variance_vector": 0.2}}, CPU, torch.float32), LaplaceProposal)
    assert isinstance(f(5, 1.0, {"name": "Laplace", "params": {"base_variance_vector": [0.2] * 5}}, CPU, torch.float32), LaplaceProposal)
    assert isinstance(f(5, 1.0, {"name": "UniformRadius", "params": {"base_radius": 1.0}}, CPU, torch.float32), UniformRadiusProposal)
    for bad in ({"name": "Normal", "params": {}}, {"name": "Laplace", "params": {}}, {"name": "UniformRadius"},
                {"name": "Cauchy", "params": {}}, {"name": "Laplace", "params": {"base_variance_vector": "x"}}):
        with pytest.raises(ValueError):
            f(5, 1.0, bad, CPU, torch.float32)
    # seeding happens after the sampler exists (simulation_gpu.py:144-148): same seed -> same torch stream
    MCMCSimulation_GPU(5, sigma=0.4, algorithm=RandomWalkMH_GPU_Optimized, target_dist=t, device="cpu", seed=77)
    a = torch.rand(3)
    MCMCSimulation_GPU(5, sigma=0.4, algorithm=RandomWalkMH_GPU_Optimized, target_dist=t, device="cpu", seed=77)
    assert torch.equal(a, torch.rand(3))


def test_shard_ranges_partition_the_chains():
    for n, w in ((524288, 8), (1048576, 8), (10, 4), (3, 8), (0, 2), (65536, 1)):
        blocks = [shard_range(n, r, w) for r in range(w)]
        assert blocks[0][0] == 0 and sum(c for _, c in blocks) == n
        for (o1, c1), (o2, _) in zip(blocks, blocks[1:]):
            assert o1 + c1 == o2
        assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
    assert shard_range(524288, 3, 8) == (196608, 65536)
    with pytest.raises(ValueError):
        shard_range(10, 4, 4)


def test_summary_pack_roundtrip():
    T = 5
    s = {"n_replicas": 12, "post_burn_steps": 40, "swap_attempts": 96,
         "accept_count": torch.arange(T, dtype=torch.int64) * 7 + 100,
         "sq_jump_sum": torch.linspace(1.5, 9.5, T, dtype=torch.float64),
         "swap_accept_count": torch.tensor([5, 6, 7, 8, 0], dtype=torch.int64)}
    out = unpack_summary(pack_summary(s, CPU), T, world_size=1)
    assert out["n_replicas"] == 12 and out["post_burn_steps"] == 40 and out["swap_attempts"] == 96
    assert torch.equal(out["accept_count"], s["accept_count"])
    np.testing.assert_allclose(out["acceptance_rate"].numpy(), s["accept_count"].numpy() / 480)
    np.testing.assert_allclose(out["esjd"].numpy(), s["sq_jump_sum"].numpy() / 480)
    assert out["swap_acceptance_rate"] == pytest.approx(26 / 96)


def test_superfunnel_density_matches_the_reference():
    """SuperFunnelTorch (hierarchical logistic regression, ragged data): the batched formulation reproduces the
    reference's log_density (tests/golden/superfunnel.npz, funnel_torch.py:193-291), -inf cases included.  It has no
    fused kernel (engine_target raises): the samplers run it in split steps."""
    import numpy as np
    import torch

    import helpers as H
    from target_distributions import SuperFunnelTorch

    f = H.load("superfunnel.npz")
    cuts = np.cumsum(f["n_j"])[:-1]
    X = [torch.from_numpy(x) for x in np.split(f["X"], cuts)]
    Y = [torch.from_numpy(y) for y in np.split(f["Y"], cuts)]
    t = SuperFunnelTorch(int(f["J"]), int(f["K"]), X, Y, device="cpu")
    assert t.get_name() == str(f["target_name"]) and t.dim == int(f["dim"])
    got = t.log_density(torch.from_numpy(f["theta"])).numpy()
    ref = f["log_density"]
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), fin) and (~fin).sum() == 3
    assert np.allclose(got[fin], ref[fin], rtol=2e-6, atol=1e-4)
    assert float(t.log_density(torch.from_numpy(f["theta"][0]))) == pytest.approx(float(ref[0]), rel=2e-6)
    with pytest.raises(NotImplementedError):
        t.engine_target()
    with pytest.raises(ValueError):
        SuperFunnelTorch(2, 2, X, Y)


def test_driver_target_factory_calls_construct():
    """Every constructor call the reference's GPU experiment drivers make (experiment_pt_GPU.py:20-140,
    experiment_RWM_GPU.py: get_target_distribution, use_torch=True) works against this package with the same keyword
    arguments - a driver switched to this package by PYTHONPATH finds every torch target it asks for."""
    import torch

    import target_distributions as TD

    dim, dev = 10, torch.device("cpu")
    centers = [[-15.0] + [0.0] * (dim - 1), [0.0] * dim, [15.0] + [0.0] * (dim - 1)]
    made = [
        TD.MultivariateNormalTorch(dim, device=dev),
        TD.ScaledMultivariateNormalTorch(dim, device=dev),
        TD.RoughCarpetDistributionTorch(dim, scaling=False, device=dev, mode_centers=[-15.0, 0.0, 15.0],
                                        mode_weights=[0.5, 0.3, 0.2]),
        TD.RoughCarpetDistributionTorch(dim, scaling=True, device=dev, mode_centers=[-15.0, 0.0, 15.0],
                                        mode_weights=[0.5, 0.3, 0.2]),
        TD.ThreeMixtureDistributionTorch(dim, scaling=False, device=dev, mode_centers=centers, mode_weights=[1 / 3] * 3),
        TD.ThreeMixtureDistributionTorch(dim, scaling=True, device=dev, mode_centers=centers, mode_weights=[1 / 3] * 3),
        TD.HypercubeTorch(dim, left_boundary=-1, right_boundary=1, device=dev),
        TD.IIDGammaTorch(dim, shape=2, scale=3, device=dev),
        TD.IIDBetaTorch(dim, alpha=2, beta=3, device=dev),
        TD.FullRosenbrockTorch(dim, a_coeff=1.0 / 20.0, b_coeff=100.0 / 20.0, mu=1.0, device=dev),
        TD.EvenRosenbrockTorch(dim, a_coeff=1.0 / 20.0, b_coeff=100.0 / 20.0, mu=1.0, device=dev),
        TD.HybridRosenbrockTorch(n1=3, n2=5, a_coeff=1.0 / 20.0, b_coeff=100.0 / 20.0, mu=1.0, device=dev),
        TD.NealFunnelTorch(dim, mu_v=0.0, sigma_v_sq=9.0, mu_z=0.0, device=dev),
    ]
    J, K = 5, 3
    torch.manual_seed(42)
    X = [torch.randn(20, K) for _ in range(J)]
    Y = [(torch.rand(20) < 0.5).float() for _ in range(J)]
    made.append(TD.SuperFunnelTorch(J, K, X, Y, prior_hypermean_std=10.0, prior_tau_scale=2.5, device=dev))
    assert made[-1].dim == J + J * K + 1 + K + 1 + 1 and made[11].dim == 1 + 5 * (3 - 1)
    for t in made:
        assert isinstance(t.get_name(), str) and t.dim >= 1
        # everything but the dense Gaussian / SuperFunnel describes itself to the fused kernel
    fused = [t for t in made if t is not made[-1]]
    for t in fused:
        assert t.engine_target().dim == t.dim


def test_abi_calls_run_with_the_tensors_device_current(monkeypatch):
    """ptrwm_hip.on_device: every C-ABI call is made with the tensors' device current (HIP launches on the current
    device whatever the pointers say) and the previous device is restored, also when the call raises.  Checked
    without a GPU by recording what the guard does to torch.cuda's current device."""
    cur = {"dev": 0, "log": []}
    monkeypatch.setattr(torch.cuda, "current_device", lambda: cur["dev"])

    def set_device(i):
        cur["dev"] = int(i)
        cur["log"].append(int(i))

    monkeypatch.setattr(torch.cuda, "set_device", set_device)
    with ptrwm_hip.on_device(torch.device("cuda", 1)):
        assert cur["dev"] == 1
    assert cur["dev"] == 0 and cur["log"] == [1, 0]
    cur["log"].clear()
    with ptrwm_hip.on_device(torch.device("cuda", 0)):  # already current: nothing to do
        assert cur["dev"] == 0
    assert cur["log"] == []
    with pytest.raises(KeyError):
        with ptrwm_hip.on_device(torch.device("cuda", 3)):
            assert cur["dev"] == 3
            raise KeyError("boom")
    assert cur["dev"] == 0
    # a bare "cuda" device means the current one
    cur["dev"] = 2
    g = ptrwm_hip.on_device(torch.device("cuda"))
    assert g.idx == 2
    # every entry point that reaches the library goes through the guard
    import inspect

    src = inspect.getsource(ptrwm_hip)
    host_only = ("ptrwm_strerror", "ptrwm_abi_version", "ptrwm_ext_raw_per_step", "ptrwm_has_variant",
                 "ptrwm_has_quad_variant", "ptrwm_has_thread_variant", "ptrwm_auto_form", "ptrwm_set_kernel_form",
                 "ptrwm_set_stream_mode", "ptrwm_has_stream_variant", "ptrwm_last_launch_kind", "ptrwm_auto_form_for",
                 "ptrwm_source_hash", "ptrwm_form_table_source_hash")  # no launch behind these
    calls = [ln for ln in src.splitlines() if "lib.ptrwm_" in ln and "(" in ln and not any(h in ln for h in host_only)]
    assert len(calls) >= 7
    lines = src.splitlines()
    for ln in calls:
        i = lines.index(ln)
        assert lines[i - 1].strip().startswith("with "), f"unguarded C-ABI call: {ln.strip()}"


def test_every_tool_at_least_parses():
    import os
    """tools/ is tuning and evidence apparatus that the suites do not run (it needs a GPU and minutes per script); what can be
    held here: every Python tool compiles, every shell tool passes `bash -n`, and every tool opens with a description."""
    import glob
    import py_compile
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tools = sorted(glob.glob(os.path.join(root, "tools", "*.py")) + glob.glob(os.path.join(root, "tools", "*.sh")))
    assert len(tools) > 30
    for path in tools:
        if os.path.basename(path).startswith("_"):
            continue  # scratch of a GPU call, not tracked
        with open(path) as f:
            head = f.read(400)
        if path.endswith(".py"):
            py_compile.compile(path, doraise=True)
            assert head.lstrip().startswith(('"""', "'''", "#")), path
        else:
            assert subprocess.run(["bash", "-n", path]).returncode == 0, path
            assert head.startswith("#!/bin/bash") and "\n#" in head, path
