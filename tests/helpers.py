"""Shared test plumbing: golden fixtures -> target/proposal descriptions for the oracle and the engine."""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

KIND = {
    "RoughCarpetDistributionTorch": O.TARGET_ROUGH_CARPET,
    "ThreeMixtureDistributionTorch": O.TARGET_THREE_MIXTURE,
    "FullRosenbrockTorch": O.TARGET_FULL_ROSENBROCK,
    "EvenRosenbrockTorch": O.TARGET_EVEN_ROSENBROCK,
    "HybridRosenbrockTorch": O.TARGET_HYBRID_ROSENBROCK,
    "IIDGammaTorch": O.TARGET_IID_GAMMA,
    "IIDBetaTorch": O.TARGET_IID_BETA,
    "MultivariateNormalTorch": O.TARGET_DIAG_GAUSSIAN,
    "ScaledMultivariateNormalTorch": O.TARGET_DIAG_GAUSSIAN,
    "HypercubeTorch": O.TARGET_HYPERCUBE,
    "NealFunnelTorch": O.TARGET_NEAL_FUNNEL,
}
PROPOSAL_KIND = {"Normal": O.PROPOSAL_NORMAL, "Laplace": O.PROPOSAL_LAPLACE, "UniformRadius": O.PROPOSAL_UNIFORM_RADIUS}

f32 = np.float32


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@dataclass
class TargetSpec:
    """Backend-neutral target description: the fields of ptrwm_target_desc with host arrays."""

    kind: int
    dim: int
    p: tuple = ()
    ip: tuple = ()
    vec0: Optional[np.ndarray] = None
    vec1: Optional[np.ndarray] = None
    cls: str = ""
    params: dict = field(default_factory=dict)

    def oracle(self):
        return O.Target(self.kind, self.dim, self.p, self.ip, self.vec0, self.vec1)

    def engine(self, device):
        import torch

        import ptrwm_hip as E

        def dev(a):
            return None if a is None else torch.tensor(np.asarray(a, dtype=f32), device=device)

        return E.Target(self.kind, self.dim, tuple(self.p), tuple(self.ip), dev(self.vec0), dev(self.vec1))


def _lgamma32(v):
    import math

    return f32(math.lgamma(float(v)))


def spec_from_params(cls: str, dim: int, params: dict) -> TargetSpec:
    """Folds class-level parameters (as the reference classes store them, fp32) into descriptor fields the way
    rwm-pt-pytorch_amd/target_distributions/*.engine_target does."""
    kind = KIND[cls]
    g = lambda k: np.asarray(params[k], dtype=f32)  # noqa: E731
    if kind == O.TARGET_ROUGH_CARPET:
        lw = np.log(g("weights"))
        sf = g("scaling_factors") if "scaling_factors" in params else None
        lj = f32(np.sum(np.log(sf), dtype=f32)) if sf is not None else f32(0)
        return TargetSpec(kind, dim, (*g("modes"), *lw, lj), (), sf, None, cls, params)
    if kind == O.TARGET_THREE_MIXTURE:
        lw = np.log(g("mixing_weights"))
        log_2pi = f32(np.log(f32(2 * np.pi)))
        lnc = f32(-0.5) * (f32(dim) * log_2pi + f32(0))
        sf = g("scaling_factors") if "scaling_factors" in params else None
        if sf is not None:
            c = (lnc + f32(np.sum(np.log(sf), dtype=f32))) + lw
        else:
            c = lnc + lw
        return TargetSpec(kind, dim, tuple(c.astype(f32)), (), g("means").reshape(-1), sf, cls, params)
    if kind in (O.TARGET_FULL_ROSENBROCK, O.TARGET_EVEN_ROSENBROCK):
        return TargetSpec(kind, dim, (g("a_coeff"), g("b_coeff")), (), g("mu"), None, cls, params)
    if kind == O.TARGET_HYBRID_ROSENBROCK:
        return TargetSpec(kind, dim, (g("a_coeff"), g("b_coeff"), g("mu")), (params["n1"], params["n2"]), None, None,
                          cls, params)
    if kind == O.TARGET_IID_GAMMA:
        k, th = g("shape"), g("scale")
        lnc = f32(dim) * (_lgamma32(k) + k * np.log(th))
        return TargetSpec(kind, dim, (k, th, lnc), (), None, None, cls, params)
    if kind == O.TARGET_IID_BETA:
        a, b = g("alpha"), g("beta")
        lnc = f32(dim) * (_lgamma32(a + b) - _lgamma32(a) - _lgamma32(b))
        return TargetSpec(kind, dim, (a, b, lnc), (), None, None, cls, params)
    if cls == "MultivariateNormalTorch":
        cov = np.asarray(params["cov"], dtype=np.float64)
        assert np.count_nonzero(cov - np.diag(np.diag(cov))) == 0
        return TargetSpec(kind, dim, (g("log_norm_const"),), (0,), g("mean"), (1.0 / np.diag(cov)).astype(f32), cls,
                          params)
    if cls == "ScaledMultivariateNormalTorch":
        return TargetSpec(kind, dim, (g("log_norm_const"),), (1,), g("scaling_factors"), None, cls, params)
    if kind == O.TARGET_HYPERCUBE:
        return TargetSpec(kind, dim, (g("left_boundary"), g("right_boundary"), g("log_uniform_density")), (), None, None,
                          cls, params)
    if kind == O.TARGET_NEAL_FUNNEL:
        return TargetSpec(kind, dim, (g("mu_v"), g("sigma_v_sq"), g("mu_z")), (), None, None, cls, params)
    raise KeyError(cls)


_LOGD = None


def golden_targets():
    """{key: (TargetSpec, x[n, D], reference logp[n])} from logdensity.npz."""
    global _LOGD
    if _LOGD is None:
        z = load("logdensity.npz")
        meta = json.loads(bytes(z["meta_json"]).decode())
        out = {}
        for key, m in meta.items():
            params = {k[len(key) + 4:]: z[k] for k in z.files if k.startswith(key + "__p_")}
            params.update({k: v for k, v in m.items() if k in ("n1", "n2")})
            out[key] = (spec_from_params(m["class"], m["dim"], params), z[f"{key}__x"], z[f"{key}__logp"], m)
        _LOGD = out
    return _LOGD


def target_spec(key) -> TargetSpec:
    return golden_targets()[key][0]


@dataclass
class ProposalSpec:
    kind: int
    temp_scale: np.ndarray
    dim_scale: Optional[np.ndarray] = None
    inv_dim: float = 0.0

    def oracle(self):
        return O.Proposal(self.kind, self.temp_scale, self.dim_scale, self.inv_dim)

    def engine(self, device):
        import torch

        import ptrwm_hip as E

        ds = None if self.dim_scale is None else torch.tensor(np.asarray(self.dim_scale, f32), device=device)
        return E.Proposal(self.kind, torch.tensor(np.asarray(self.temp_scale, f32), device=device), ds, self.inv_dim)


def proposal_spec(kind_name: str, dim: int, betas, *, base_variance_scalar=None, base_variance_vector=None,
                  base_radius=None, single=False) -> ProposalSpec:
    """Kernel-side proposal parameters by the rules of proposal_distributions/*.engine_proposal.
    single=True: one temperature, scale tempered by betas[0] the way the proposal's constructor does."""
    betas = np.asarray(betas, dtype=np.float64)
    kind = PROPOSAL_KIND[kind_name]
    if kind == O.PROPOSAL_NORMAL:
        ts = np.sqrt((float(base_variance_scalar) / betas).astype(f32))
        return ProposalSpec(kind, ts.astype(f32))
    if kind == O.PROPOSAL_LAPLACE:
        bv = np.asarray(base_variance_vector, dtype=f32)
        if single:
            return ProposalSpec(kind, np.ones(1, f32), np.sqrt((bv / f32(betas[0])) / f32(2)).astype(f32))
        return ProposalSpec(kind, (f32(1) / np.sqrt(betas.astype(f32))).astype(f32), np.sqrt(bv / f32(2)).astype(f32))
    r = (f32(base_radius) / np.sqrt(betas.astype(f32))).astype(f32)
    return ProposalSpec(kind, r, None, 1.0 / dim)


def first_mismatch(a, b):
    """Index of the first step at which two [steps, ...] arrays differ, or None."""
    neq = np.any((a != b).reshape(a.shape[0], -1), axis=1)
    idx = np.nonzero(neq)[0]
    return None if idx.size == 0 else int(idx[0])
