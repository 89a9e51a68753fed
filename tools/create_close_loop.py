import os, sys, time
sys.path.insert(0, os.getcwd())
import sabc_amd as S
from tests.cases import hip_model_prior, hip_proposal
model, prior = hip_model_prior(S, "gauss1_cfg2")
for rep in range(20):
    h = S.SabcHandle(n_particles=1000, model=model, prior=prior, seed=7); h.initialize(1000)
    h.update(n_simulation=100 * 1000, proposal=hip_proposal(S, "rw", 1))
    h.close()
