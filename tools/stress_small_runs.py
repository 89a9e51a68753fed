"""The reference's integration tests (test/runtests.jl:56-267) with MANY seeds: n_particles = 100, 1000 + 1000 + 50 simulations,
every model of those tests x both epsilon schedules x all proposals.  Prints every seed whose run raises or violates what
the tests assert.  usage (GPU box): python tools/stress_small_runs.py [n_seeds]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import sabc_amd as S
from tests.cases import MODELS, hip_model_prior, hip_proposal

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for name in ["gauss1_uniform", "gauss2_meansd", "gauss1_2stats", "gauss2_2stats"]:
    model, prior = hip_model_prior(S, name)
    for alg in ["multi_eps", "single_eps"]:
        for prop in ["de", "rw", "stretch"]:
            fails = []
            for seed in range(1, n_seeds + 1):
                try:
                    p = hip_proposal(S, prop, len(prior))
                    res = S.sabc(model, prior, n_particles=100, n_simulation=1000, algorithm=alg, proposal=p, seed=seed)
                    ok = res.state.n_simulation <= 1000 and res.state.n_population_updates == 9 and len(res.population) == 100
                    if MODELS[name]["s"] > 1:
                        ok = ok and bool(np.all(np.asarray(res.state.ϵ) < 1))
                    S.update_population_(res, model, prior, n_simulation=1000, proposal=p)
                    ok = ok and res.state.n_simulation <= 2000 and res.state.n_population_updates == 19
                    ok = ok and bool(np.isfinite(res.population).all())
                    if not ok:
                        fails.append((seed, "assert", [float(e) for e in np.atleast_1d(res.state.ϵ)], res.state.n_population_updates))
                except Exception as e:     # noqa: BLE001
                    fails.append((seed, type(e).__name__, str(e)[:120]))
            bad += len(fails)
            print(f"{name:16s} {alg:10s} {prop:8s} failures {len(fails)}/{n_seeds}", fails[:4], flush=True)
print("total failures", bad)
