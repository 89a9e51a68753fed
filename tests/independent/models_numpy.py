"""Vectorised simulators / priors for tests/independent/sabc_numpy.py: the same models as DESIGN.md "Simulators", written
from their definitions (not from the oracle or the kernels).  TEST INFRASTRUCTURE."""
import numpy as np

SEED = 20241220


def y_obs_mean():
    return float(np.random.default_rng(SEED).normal(1.5, 1.0, 100).mean())


def cfg2():
    """1-D Gaussian mean: theta ~ N(0, 2^2); x_1..100 ~ N(theta, 1); rho = |mean(x) - mean(y_obs)|.
    mean(x) is drawn directly: it is exactly N(theta, 1/100)."""
    yb = y_obs_mean()
    sim = lambda th, rng: np.abs(th[:, 0] + rng.standard_normal(len(th)) / 10.0 - yb)[:, None]
    sample = lambda n, rng: rng.normal(0.0, 2.0, (n, 1))
    logpdf = lambda th: -0.5 * (th[:, 0] / 2.0) ** 2 - np.log(2.0) - 0.5 * np.log(2 * np.pi)
    post_var = 1.0 / (1.0 / 4.0 + 100.0)
    return dict(sim=sim, sample=sample, logpdf=logpdf, post_mean=post_var * 100.0 * yb, post_var=post_var)


def cfg3(n_obs=50, r=0.6, obs_mean=(1.2, -0.7), obs_varsum=2.1, obs_cov=0.55):
    """2-D correlated Gaussian, 3 statistics: theta ~ N(0, 3^2)^2; x_i ~ N(theta, [[1, r], [r, 1]]), i <= n_obs;
    rho = (||mean - obs||_2, |var_1 + var_2 - obs|, |cov_12 - obs|) with n - 1 denominators."""
    L = np.linalg.cholesky(np.array([[1.0, r], [r, 1.0]]))
    obs = np.array(obs_mean)

    def sim(th, rng):
        # sufficient statistics drawn directly instead of the n_obs points: the sample mean is N(theta, Sigma / n_obs), the
        # scatter matrix is Wishart(n_obs - 1, Sigma) and independent of it (Bartlett decomposition: S = L A A' L')
        m = len(th)
        mean = th + (rng.standard_normal((m, 2)) @ L.T) / np.sqrt(n_obs)
        a11 = np.sqrt(rng.chisquare(n_obs - 1, m))
        a22 = np.sqrt(rng.chisquare(n_obs - 2, m))
        a21 = rng.standard_normal(m)
        A = np.zeros((m, 2, 2))
        A[:, 0, 0], A[:, 1, 0], A[:, 1, 1] = a11, a21, a22
        LA = L @ A
        Sc = LA @ np.transpose(LA, (0, 2, 1)) / (n_obs - 1)
        return np.stack([np.linalg.norm(mean - obs, axis=1), np.abs(Sc[:, 0, 0] + Sc[:, 1, 1] - obs_varsum),
                         np.abs(Sc[:, 0, 1] - obs_cov)], axis=1)

    def sim_pointwise(th, rng):                       # the definition itself, for the cross-check in the tests
        x = th[:, None, :] + rng.standard_normal((len(th), n_obs, 2)) @ L.T
        m = x.mean(1)
        c = x - m[:, None, :]
        var = (c ** 2).sum(1) / (n_obs - 1)
        cov = (c[:, :, 0] * c[:, :, 1]).sum(1) / (n_obs - 1)
        return np.stack([np.linalg.norm(m - obs, axis=1), np.abs(var.sum(1) - obs_varsum), np.abs(cov - obs_cov)], axis=1)

    sample = lambda n, rng: rng.normal(0.0, 3.0, (n, 2))
    logpdf = lambda th: (-0.5 * (th / 3.0) ** 2 - np.log(3.0) - 0.5 * np.log(2 * np.pi)).sum(1)
    Sig = np.array([[1.0, r], [r, 1.0]])
    Lam = np.eye(2) / 9.0 + n_obs * np.linalg.inv(Sig)
    C = np.linalg.inv(Lam)
    return dict(sim=sim, sim_pointwise=sim_pointwise, sample=sample, logpdf=logpdf,
                post_mean=C @ (n_obs * np.linalg.inv(Sig) @ obs), post_cov=C)
