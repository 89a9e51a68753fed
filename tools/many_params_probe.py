"""What the RandomWalk proposal's covariance + Cholesky factor cost per population update on the control lane at d = 16
parameters (host-callable simulator, n = 2000, s = 2): RandomWalk against DifferentialEvolution (no covariance), library time."""
import time

import numpy as np

import sabc_amd as S

d, s, n, k = 16, 2, 2000, 300
rng = np.random.default_rng(1)
truth = np.linspace(-1, 1, d)


def f(theta):
    r = theta - truth + 0.05 * rng.standard_normal(theta.shape)
    return np.stack([np.abs(r[:, :8]).mean(1), np.abs(r[:, 8:]).mean(1)], axis=1)


prior = S.product_distribution([S.Normal(0.0, 2.0)] * d)
for name, prop in (("DifferentialEvolution", S.DifferentialEvolution(n_para=d)), ("RandomWalk", S.RandomWalk(n_para=d))):
    h = S.SabcHandle(n_particles=n, model=S.HostDistance(f, n_stats=s, n_para=d, univariate=False, batched=True), prior=prior, seed=5)
    h.initialize(n)
    h.update(n_simulation=20 * n, proposal=prop)
    cb0, t0 = h.host_callback_seconds, time.perf_counter()
    h.update(n_simulation=k * n, proposal=prop)
    dt, cb = time.perf_counter() - t0, h.host_callback_seconds - cb0
    print(f"{name}: {dt / k * 1e6:.1f} us per update, {cb / k * 1e6:.1f} in the callback, library {(dt - cb) / k * 1e6:.1f}", flush=True)
    h.close()
