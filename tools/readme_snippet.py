"""The Python usage snippet of INTEGRATION.md, verbatim (run on the GPU box to check that the docs do not rot)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sabc_amd as S                      # repo-root shim for the directory `simulatedannealingabc.jl_amd`
model = S.GaussianIID(n_obs=100, sd=1.0, obs_mean=1.62)
res = S.sabc(model, S.Normal(0, 2), n_particles=1_000_000, n_simulation=51_000_000,
             proposal=S.RandomWalk(n_para=1), seed=20241220)
S.update_population_(res, model, S.Normal(0, 2), n_simulation=10_000_000, proposal=S.RandomWalk(n_para=1))
print(res.population.shape, res.u.shape, res.ρ.shape, res.state.ϵ, len(res.state.ϵ_history), res.state.n_accept)
print(res)
