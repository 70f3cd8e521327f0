"""Base class of the samplers: holds the chain list and picks the initial state.

The starting point is part of parity with the reference (interfaces/metropolis.py:16-64): it depends
on the *name* of the target, and it consumes the global NumPy RNG in the same way.
"""
import numpy as np


def _target_name(target_dist):
    """All name strings the reference consults, in its order: the `name` attribute, then get_name()."""
    names = []
    attr = getattr(target_dist, "name", None)
    if isinstance(attr, str):
        names.append(attr)
    getter = getattr(target_dist, "get_name", None)
    if callable(getter):
        try:
            got = getter()
        except NotImplementedError:
            got = None
        if isinstance(got, str):
            names.append(got)
    return names


def initial_state_for(target_dist, dim):
    """Name-dependent start (reference rules, first match wins):
    "Beta" -> U(0.2, 0.8) float32; "Gamma" -> 5 + 0.01 N(0,1); "RoughCarpet" / "ThreeMixture" -> 0;
    anything else -> 1e-8 N(0,1)."""
    names = _target_name(target_dist)

    def has(word):
        return any(word in n for n in names)

    if has("Beta"):
        return np.random.uniform(0.2, 0.8, size=dim).astype(np.float32)
    if has("Gamma"):
        return 5 + 0.01 * np.random.randn(dim)
    if has("RoughCarpet") or has("ThreeMixture"):
        return np.zeros(dim)
    return 0.00000001 * np.random.randn(dim)


class MHAlgorithm:
    """General Metropolis-Hastings sampler interface: `chain` is the list of visited states and the
    current state is its last element; subclasses implement `step` and `get_name`."""

    def __init__(self, dim, var, target_dist=None, symmetric=True):
        self.dim = dim
        self.var = var
        self.target_dist = target_dist
        self.chain = [initial_state_for(target_dist, dim)]
        self.symmetric = symmetric
        self.num_acceptances = 0
        self.acceptance_rate = 0
        self.target_density = getattr(target_dist, "density", None) if target_dist is not None else None

    def reset(self):
        self.chain = [self.chain[0]]

    def step(self):
        raise NotImplementedError("Step method must be implemented in subclass")

    def get_curr_state(self):
        return self.chain[-1]

    def set_curr_state(self, state):
        self.chain[-1] = state

    def get_name(self):
        raise NotImplementedError("Subclasses must implement the get_name method.")
