"""What the multi-epsilon schedule costs per population update when f_dist returns dozens of distances (a time series of 48
points as 48 statistics, host-callable simulator, n = 2000): single-eps against multi-eps, library time = wall - callback."""
import time

import numpy as np

import sabc_amd as S

d, s, n, k = 3, 48, 2000, 300
t = np.linspace(0.0, 4.0, s)
obs = 2.0 * np.exp(-0.6 * t) + 0.3
rng = np.random.default_rng(1)


def f(theta):
    return np.abs(theta[:, :1] * np.exp(-theta[:, 1:2] * t) + theta[:, 2:3] + 0.05 * rng.standard_normal((len(theta), s)) - obs)


prior = S.product_distribution([S.Uniform(0.5, 4.0), S.Uniform(0.05, 2.0), S.Normal(0.0, 1.0)])
for alg in ("single_eps", "multi_eps"):
    h = S.SabcHandle(n_particles=n, model=S.HostDistance(f, n_stats=s, n_para=d, univariate=False, batched=True), prior=prior, seed=5,
                     algorithm=S._lib.ALG_MULTI_EPS if alg == "multi_eps" else S._lib.ALG_SINGLE_EPS)
    h.initialize(n)
    h.update(n_simulation=20 * n, proposal=S.RandomWalk(n_para=d))
    cb0, t0 = h.host_callback_seconds, time.perf_counter()
    h.update(n_simulation=k * n, proposal=S.RandomWalk(n_para=d))
    dt, cb = time.perf_counter() - t0, h.host_callback_seconds - cb0
    print(f"{alg}: {dt / k * 1e6:.1f} us per update, {cb / k * 1e6:.1f} in the callback, library {(dt - cb) / k * 1e6:.1f}; eps[:3] {h.eps[:3]}", flush=True)
    h.close()
