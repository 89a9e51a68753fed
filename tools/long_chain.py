"""A long chain in ONE sabc_update call (GPU box): 200 000 population updates of 10 000 particles -- the mailbox ring, the
history buffer's growth, the queue-ahead protocol and the resample test over hundreds of thousands of steps."""
import sys
import time

import numpy as np

import sabc_amd as S

n, k = 10_000, int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
for prop in (S.RandomWalk(n_para=1), S.DifferentialEvolution(n_para=1)):
    h = S.SabcHandle(n_particles=n, model=S.GaussianIID(n_obs=100, sd=1.0, obs_mean=1.6), prior=S.Normal(0, 2), seed=3)
    h.initialize((k + 1) * n)
    t0 = time.perf_counter()
    h.update(n_simulation=k * n, proposal=prop, checkpoint_history=100)
    dt = time.perf_counter() - t0
    c, th = h.counters, h.get_population()[0][0]
    e, _, _ = h.history
    assert c["n_population_updates"] == k and c["n_simulation"] == (k + 1) * n, c
    assert np.isfinite(th).all() and abs(th.mean() - 1.6) < 0.02 and 0.05 < th.std() < 0.2, (th.mean(), th.std())
    assert len(e) == k // 100 + 1 and np.all(np.diff(e[:, 0]) <= 1e-12), len(e)          # epsilon never rises
    print(f"{type(prop).__name__}: {k} updates in {dt:.1f} s ({dt / k * 1e6:.1f} us each), n_accept {c['n_accept']}, n_resampling {c['n_resampling']}, "
          f"eps {h.eps[0]:.3e}, mean {th.mean():.4f}, sd {th.std():.4f}", flush=True)
    h.close()
