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


def cfg4(n_draws=128, c=0.8, ranks=(16, 48, 80, 112), obs=(1.9, 2.7, 3.6, 6.4)):
    """g-and-k: theta = (A, B, g, k) ~ U(0, 10)^4; x = A + B (1 + c tanh(g z / 2)) (1 + z^2)^k z for n_draws standard
    normals z; rho_j = |x_(rank_j) - obs_j| (1-based order statistics); a non-finite distance counts as 1e30."""
    idx = np.array(ranks) - 1
    ob = np.array(obs)

    def sim(th, rng):
        z = rng.standard_normal((len(th), n_draws))
        A, B, g, k = (th[:, j][:, None] for j in range(4))
        with np.errstate(over="ignore", invalid="ignore"):
            x = A + B * (1.0 + c * np.tanh(g * z / 2.0)) * (1.0 + z * z) ** k * z
            x.sort(axis=1)
            r = np.abs(x[:, idx] - ob)
        r[~np.isfinite(r)] = 1e30
        return r

    sample = lambda n, rng: rng.uniform(0.0, 10.0, (n, 4))

    def logpdf(th):
        inside = np.all((th >= 0.0) & (th <= 10.0), axis=1)
        return np.where(inside, -4.0 * np.log(10.0), -np.inf)
    return dict(sim=sim, sample=sample, logpdf=logpdf)


def cfg5(n_steps=256, dt=0.05, sigma=0.1, x0=50.0, y0=50.0, obs=(18.0, 17.0, 14.0, 12.0)):
    """Stochastic Lotka-Volterra by Euler-Maruyama: dX = (aX - bXY) dt + sigma X dW1, dY = (bXY - cY) dt + sigma Y dW2, both
    from the old state, clamped at 0; theta = (a, b, c) ~ U(0, 2) x U(0, 0.1) x U(0, 2); rho = |mean X, sd X, mean Y, sd Y
    of the n_steps states after each step - obs| (sd with n - 1); a non-finite distance counts as 1e30."""
    ob = np.array(obs)
    hi = np.array([2.0, 0.1, 2.0])

    def sim(th, rng):
        m = len(th)
        a, b, c = th[:, 0], th[:, 1], th[:, 2]
        X, Y = np.full(m, x0), np.full(m, y0)
        sx = np.zeros(m); qx = np.zeros(m); sy = np.zeros(m); qy = np.zeros(m)
        sq = np.sqrt(dt)
        with np.errstate(over="ignore", invalid="ignore"):
            for _ in range(n_steps):
                dW1, dW2 = sq * rng.standard_normal(m), sq * rng.standard_normal(m)
                nX = X + (a * X - b * X * Y) * dt + sigma * X * dW1
                nY = Y + (b * X * Y - c * Y) * dt + sigma * Y * dW2
                X, Y = np.maximum(nX, 0.0), np.maximum(nY, 0.0)
                sx += X; qx += X * X; sy += Y; qy += Y * Y
            mx, my = sx / n_steps, sy / n_steps
            vx, vy = (qx - sx * mx) / (n_steps - 1), (qy - sy * my) / (n_steps - 1)
            st = np.stack([mx, np.sqrt(np.maximum(vx, 0.0)), my, np.sqrt(np.maximum(vy, 0.0))], axis=1)
            r = np.abs(st - ob)
        r[~np.isfinite(r)] = 1e30
        return r

    sample = lambda n, rng: rng.uniform(0.0, 1.0, (n, 3)) * hi

    def logpdf(th):
        inside = np.all((th >= 0.0) & (th <= hi), axis=1)
        return np.where(inside, -np.log(hi).sum(), -np.inf)
    return dict(sim=sim, sample=sample, logpdf=logpdf)
