import os, sys, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import sabc_amd as S
model = S.GaussianIID(n_obs=100, sd=1.0, obs_mean=1.4); prior = S.Normal(0.0, 2.0)
S.sabc(model, prior, n_particles=1000, n_simulation=1_000_000, seed=1)
pr = cProfile.Profile(); pr.enable()
res = S.sabc(model, prior, n_particles=1000, n_simulation=10_000_000, seed=2)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
