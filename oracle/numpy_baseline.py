"""NumPy restatement of the reference's CPU samplers -- the CPU baseline timed beside the HIP path.

TEST INFRASTRUCTURE ONLY (tests/, bench.py cpu_baseline leg).  The reference's Python files cannot
travel to the GPU box, so its NumPy path is restated here with the same RNG call order, so that
seeded runs reproduce the reference's numbers exactly (pinned by tests/golden/numpy_*.npz and
numpy_baseline.json, which were produced by running the real reference):

  RandomWalkMH            algorithms/rwm.py:6-66
  ParallelTemperingRWM    algorithms/pt_rwm.py:7-184 (explicit beta_ladder path only)
  RoughCarpetDistribution target_distributions/multimodal.py:66-115 (unscaled)
  seeding / ESJD          interfaces/simulation.py:23-34, :62-81; interfaces/metropolis.py:16-64
"""
import numpy as np


class RoughCarpetNumpy:
    """density(x) = prod_d sum_k w_k N(x_d | m_k, 1), modes [-15, 0, 15], weights [.5, .3, .2]."""

    name = "RoughCarpet"
    modes = (-15.0, 0.0, 15.0)
    weights = (0.5, 0.3, 0.2)

    def __init__(self, dim):
        self.dim = dim

    def density_1d(self, x):
        c = np.sqrt(2 * np.pi)
        m0 = np.exp(-0.5 * (x - self.modes[0]) ** 2) / c
        m1 = np.exp(-0.5 * (x - self.modes[1]) ** 2) / c
        m2 = np.exp(-0.5 * (x - self.modes[2]) ** 2) / c
        return self.weights[0] * m0 + self.weights[1] * m1 + self.weights[2] * m2

    def density(self, x):
        # per-coordinate python loop + np.prod, as the reference evaluates it (multimodal.py:103)
        return np.prod([self.density_1d(x[i]) for i in range(self.dim)])


class RandomWalkMHNumpy:
    """One chain; log pi = log(density + 1e-300), -inf if the density is exactly 0 (rwm.py:52-55); the accept
    uniform is drawn only when log_ratio <= 0 (short-circuit `or`, rwm.py:32); the acceptance rate divides by
    len(chain) = steps + 1 (rwm.py:36,39)."""

    def __init__(self, dim, var, target, beta=1.0):
        self.dim, self.var, self.target, self.beta = dim, var, target, beta
        self.chain = [np.zeros(dim)]  # RoughCarpet starts at 0 (metropolis.py:42-47)
        self.log_pi = -np.inf
        self.num_acceptances = 0
        self.acceptance_rate = 0

    def step(self):
        # np.random.multivariate_normal(x, c I) consumes the stream exactly like x + sqrt(c) * standard_normal(dim)
        # for a diagonal covariance; keep the real call so the stream is identical by construction
        prop = np.random.multivariate_normal(self.chain[-1], (self.var / self.beta) * np.eye(self.dim))
        dens = self.target.density(prop)
        log_pi_prop = -np.inf if dens == 0 else np.log(dens + 1e-300)
        log_ratio = self.beta * (log_pi_prop - self.log_pi)
        if log_ratio > 0 or np.random.random() < np.exp(log_ratio):
            self.chain.append(prop)
            self.log_pi = log_pi_prop
            self.num_acceptances += 1
        else:
            self.chain.append(self.chain[-1])
        self.acceptance_rate = self.num_acceptances / len(self.chain)


class ParallelTemperingNumpy:
    """List of RWM chains; every 20th step the chains 0..T-2 attempt a swap with their upper neighbour INSTEAD of
    moving and only the last chain moves (pt_rwm.py:169-181); swaps exchange the two current states."""

    swap_every = 20

    def __init__(self, dim, var, target, beta_ladder):
        self.beta_ladder = list(beta_ladder)
        self.chains = [RandomWalkMHNumpy(dim, var, target, b) for b in self.beta_ladder]
        self.step_counter = 0
        self.num_swap_attempts = self.num_acceptances = 0
        self.acceptance_rate = 0
        self.squared_jump_distances = 0
        self.pt_esjd = 0

    @property
    def chain(self):
        return self.chains[0].chain

    def attempt_swap(self, j, k):
        b, c = self.beta_ladder, self.chains
        log_p = b[j] * c[k].log_pi + b[k] * c[j].log_pi - b[j] * c[j].log_pi - b[k] * c[k].log_pi
        self.num_swap_attempts += 1
        if np.random.random() < min(1, np.exp(log_p)):
            c[j].chain[-1], c[k].chain[-1] = c[k].chain[-1].copy(), c[j].chain[-1].copy()
            c[j].log_pi, c[k].log_pi = c[k].log_pi, c[j].log_pi
            self.num_acceptances += 1
            self.acceptance_rate = self.num_acceptances / self.num_swap_attempts
            self.squared_jump_distances += (b[j] - b[k]) ** 2
            self.pt_esjd = self.squared_jump_distances / self.num_swap_attempts

    def step(self):
        self.step_counter += 1
        swap = self.step_counter % self.swap_every == 0
        for i, ch in enumerate(self.chains):
            if swap and i < len(self.chains) - 1:
                self.attempt_swap(i, i + 1)
            else:
                ch.step()


def esjd(chain, burn_in=0):
    c = np.asarray(chain)[burn_in:]
    return float(np.mean(np.sum((c[1:] - c[:-1]) ** 2, axis=1)))


def run_rwm(dim, var, n_iter, seed):
    """MCMCSimulation(dim, sigma=var, num_iterations=n_iter, algorithm=RandomWalkMH,
    target_dist=RoughCarpetDistribution(dim), seed=seed): the sampler is built first, then seeded (`if seed:`)."""
    alg = RandomWalkMHNumpy(dim, var, RoughCarpetNumpy(dim))
    if seed:
        np.random.seed(seed)
    for _ in range(n_iter):
        alg.step()
    return alg


def run_pt(dim, var, beta_ladder, n_iter, seed):
    alg = ParallelTemperingNumpy(dim, var, RoughCarpetNumpy(dim), beta_ladder)
    if seed:
        np.random.seed(seed)
    for _ in range(n_iter):
        alg.step()
    return alg
