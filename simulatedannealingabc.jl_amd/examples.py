"""Host-side example models, written the way a user of the reference writes `f_dist` (plain Python / NumPy, no device
code): used by `bench.py --config host` and the tests of the host-callback path."""
from __future__ import annotations

import numpy as np


def gaussian_mean_batched(obs_mean, n_obs=100, seed=0):
    """BASELINE config 2's simulator as a vectorised NumPy callable: for every proposal theta the distance
    |mean(x) - mean(y_obs)| with x_1..n_obs ~ N(theta, 1), drawn through its sufficient statistic
    (mean(x) ~ N(theta, 1 / n_obs)).  `fn(Theta)` takes the m proposals at once (HostDistance(batched=True))."""
    rng = np.random.default_rng(seed)
    sd = 1.0 / np.sqrt(n_obs)

    def fn(theta):
        theta = np.asarray(theta, dtype=np.float64)
        return np.abs(theta + sd * rng.standard_normal(theta.shape[0]) - obs_mean)

    return fn


def sir_gillespie(seed=11, N=100, i0=5, t_max=30.0, n_grid=16):
    """The reference's documentation example (docs/src/example.md:75-173): a stochastic SIR epidemic simulated with
    Gillespie's algorithm, observed on a time grid.  Returns (simulate(beta, gamma) -> infected on the grid,
    f_dist(theta, data) -> root-mean-square distance)."""
    rng = np.random.default_rng(seed)
    grid = np.linspace(0.0, t_max, n_grid)

    def simulate(beta, gamma):
        s, i, t, k = N - i0, i0, 0.0, 0
        out = np.zeros(len(grid))
        while k < len(grid):
            rate_inf, rate_rec = beta * s * i / N, gamma * i
            total = rate_inf + rate_rec
            t_next = t + rng.exponential(1 / total) if total > 0 else np.inf
            while k < len(grid) and grid[k] < t_next:
                out[k] = i
                k += 1
            if not np.isfinite(t_next):
                break
            t = t_next
            if rng.random() < rate_inf / total:
                s, i = s - 1, i + 1
            else:
                i -= 1
        return out

    def f_dist(theta, data):
        return float(np.sqrt(np.mean((simulate(theta[0], theta[1]) - data) ** 2)))

    return simulate, f_dist
