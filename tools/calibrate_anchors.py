#!/usr/bin/env python3
"""How fast the device run approaches the analytic posteriors (used to size tests/test_gpu_anchors.py).
Prints one JSON line per (config, proposal, checkpoint)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sabc_amd as S  # noqa: E402
from scipy import stats  # noqa: E402

SEED = 20241220


def cfg2(seed, prop, n, marks, v=1.0):
    yb = float(np.random.default_rng(SEED).normal(1.5, 1.0, 100).mean())
    pv = 1.0 / (1.0 / 4.0 + 100.0)
    pm = pv * 100.0 * yb
    model, prior = S.GaussianIID(n_obs=100, sd=1.0, obs_mean=yb), S.Normal(0.0, 2.0)
    h = S.SabcHandle(n_particles=n, model=model, prior=prior, seed=seed)
    h.initialize(n)
    proposal = {"rw": S.RandomWalk(n_para=1), "de": S.DifferentialEvolution(n_para=1), "stretch": S.StretchMove()}[prop]
    done = 0
    for m in marks:
        t0 = time.time()
        h.update(n_simulation=(m - done) * n, proposal=proposal, v=v, checkpoint_history=100)
        done = m
        th = h.get_population(u=False, rho=False)[0][0]
        ks = stats.kstest(th[:: max(n // 200000, 1)], "norm", args=(pm, np.sqrt(pv))).statistic
        print(json.dumps(dict(cfg="cfg2", prop=prop, seed=seed, n=n, v=v, updates=m, mean_rel=abs(th.mean() / pm - 1), var_rel=abs(th.var() / pv - 1),
                              var_signed=th.var() / pv - 1, ks=ks, eps=h.eps.tolist(), acc=h.counters["n_accept"], res=h.counters["n_resampling"],
                              dt=time.time() - t0)), flush=True)
    h.close()


def cfg3(seed, prop, n, marks, alg="single_eps", v=1.0):
    obs = np.array([1.2, -0.7])
    Sig = np.array([[1.0, 0.6], [0.6, 1.0]])
    nobs = 50
    Lam = np.eye(2) / 9.0 + nobs * np.linalg.inv(Sig)
    C = np.linalg.inv(Lam)
    pm = C @ (nobs * np.linalg.inv(Sig) @ obs)
    model = S.Gaussian2D(n_obs=nobs, r=0.6, obs_mean=tuple(obs), obs_varsum=2.1, obs_cov=0.55)
    prior = S.product_distribution([S.Normal(0, 3), S.Normal(0, 3)])
    a = S._lib.ALG_MULTI_EPS if alg == "multi_eps" else S._lib.ALG_SINGLE_EPS
    h = S.SabcHandle(n_particles=n, model=model, prior=prior, seed=seed, algorithm=a)
    h.initialize(n)
    proposal = {"rw": S.RandomWalk(n_para=2), "de": S.DifferentialEvolution(n_para=2), "stretch": S.StretchMove()}[prop]
    done = 0
    for m in marks:
        t0 = time.time()
        h.update(n_simulation=(m - done) * n, proposal=proposal, v=v, checkpoint_history=100)
        done = m
        th = h.get_population(u=False, rho=False)[0]
        cov = np.cov(th)
        print(json.dumps(dict(cfg="cfg3", prop=prop, alg=alg, seed=seed, n=n, v=v, updates=m, mean_err=(th.mean(1) - pm).tolist(), mean_rel=float(np.linalg.norm(th.mean(1) - pm) / np.linalg.norm(pm)),
                              cov_rel=float(np.linalg.norm(cov - C) / np.linalg.norm(C)), cov=cov.tolist(), C=C.tolist(), eps=h.eps.tolist(),
                              acc=h.counters["n_accept"], res=h.counters["n_resampling"], dt=time.time() - t0)), flush=True)
    h.close()


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "cfg2"):
        for prop in ("rw", "de"):
            cfg2(SEED, prop, 1_000_000, [100, 250, 500, 1000, 2000])
        cfg2(7, "rw", 1_000_000, [1000, 2000])
    if which in ("all", "cfg3"):
        for prop in ("rw", "de"):
            cfg3(SEED, prop, 1_000_000, [100, 250, 500, 1000, 2000])
        cfg3(SEED, "rw", 1_000_000, [500, 1000, 2000], alg="multi_eps")
