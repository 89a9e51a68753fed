"""Wall time of whole sabc() calls at the sizes the reference's documentation works with (docs/src/usage.md: n_particles = 1000,
n_simulation = 1e6): handle creation, initialisation, the updates, the result.  usage (GPU box): python tools/small_run_wall.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sabc_amd as S

model = S.GaussianIID(n_obs=100, sd=1.0, obs_mean=1.4)
prior = S.Normal(0.0, 2.0)
for n, nsim in ((1000, 1_000_000), (1000, 10_000_000), (5000, 5_000_000)):
    ts = []
    for rep in range(6):
        t0 = time.perf_counter()
        res = S.sabc(model, prior, n_particles=n, n_simulation=nsim, seed=rep + 1)
        ts.append(time.perf_counter() - t0)
    upd = nsim // n - 1
    print(f"n_particles {n} n_simulation {nsim}: sabc() wall first {ts[0]*1e3:.1f} ms, then {np.median(ts[1:])*1e3:.2f} ms "
          f"({upd} updates: {np.median(ts[1:]) / upd * 1e6:.2f} us per update all in; {res.state.n_resampling} resamples), posterior mean {res.population.mean():.4f}", flush=True)
# the same with the progress bar a terminal user gets (50 steps; calls end where 0.1 s of work have passed: api.next_stop)
for n, nsim in ((1000, 1_000_000),):
    ts = []
    for rep in range(4):
        t0 = time.perf_counter()
        res = S.sabc(model, prior, n_particles=n, n_simulation=nsim, seed=rep + 1, show_progressbar=True, show_checkpoint=float("inf"))
        ts.append(time.perf_counter() - t0)
    print(f"n_particles {n} n_simulation {nsim} with the progress bar: {np.median(ts[1:])*1e3:.2f} ms", flush=True)
# where a call's time goes: the pieces, timed apart
n, nsim = 1000, 1_000_000
t0 = time.perf_counter(); h = S.SabcHandle(n_particles=n, model=model, prior=prior, seed=3); t1 = time.perf_counter()
h.initialize(nsim); t2 = time.perf_counter()
h.update(n_simulation=nsim - n, proposal=S.RandomWalk(n_para=1)); t3 = time.perf_counter()
nres, launches, syncs = h.counters["n_resampling"], h.kernel_launches, h.host_syncs
pop = h.get_population(); t4 = time.perf_counter()
h.close(); t5 = time.perf_counter()
c = dict(h.counters) if False else None
print(f"pieces at n = {n}: create {1e3*(t1-t0):.2f} ms, initialize {1e3*(t2-t1):.2f}, update ({nsim//n - 1} updates) {1e3*(t3-t2):.2f}, get_population {1e3*(t4-t3):.2f}, close {1e3*(t5-t4):.2f}; {nres} resamples, {launches} kernel launches, {syncs} host syncs")
