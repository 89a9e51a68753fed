"""Golden fixtures (tests/golden/*.json; self-derived, see make_golden.py).
CPU: the oracle still reproduces them.  GPU: the HIP engine reproduces them through the C-ABI
without the oracle being present at all."""
import glob
import json
import os

import numpy as np
import pytest

from tests.cases import MODELS, hip_model_prior, hip_proposal, oracle_run

FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.json")))


def load(path):
    with open(path) as f:
        return json.load(f)


def test_fixtures_exist():
    assert len(FILES) >= 6


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[:-5] for p in FILES])
def test_oracle_reproduces_golden(O, path):
    g = load(path)
    c = g["case"]
    run = oracle_run(O, c["model"], c["n_particles"], c["n_simulation"], algorithm=c["algorithm"], prop=c["proposal"],
                     seed=c["seed"], resample=c["resample"])
    assert run.counters == g["counters"]
    np.testing.assert_allclose(run.theta, g["theta"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(run.u, g["u"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(run.rho, g["rho"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(run.eps, g["eps"], rtol=1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[:-5] for p in FILES])
def test_hip_reproduces_golden(S, gpu, path):
    g = load(path)
    c = g["case"]
    model, prior = hip_model_prior(S, c["model"])
    d = len(MODELS[c["model"]]["prior"])
    res = S.sabc(model, prior, n_particles=c["n_particles"], n_simulation=c["n_simulation"], algorithm=c["algorithm"],
                 proposal=hip_proposal(S, c["proposal"], d), resample=c["resample"], seed=c["seed"])
    st = res.state
    got = dict(n_simulation=st.n_simulation, n_accept=st.n_accept, n_resampling=st.n_resampling,
               n_population_updates=st.n_population_updates)
    assert got == g["counters"]                                  # integer work: exact
    tol = dict(rtol=1e-9, atol=1e-12)                            # f64 work: libm differences only
    pop = res.population.reshape(-1, 1) if d == 1 else res.population
    np.testing.assert_allclose(pop.T, g["theta"], **tol)
    np.testing.assert_allclose(res.u.T, g["u"], rtol=1e-9, atol=1e-9)   # ECDF slope amplifies rho differences
    np.testing.assert_allclose(res.ρ.T, g["rho"], **tol)
    np.testing.assert_allclose(st.ϵ, g["eps"], rtol=1e-9)
    np.testing.assert_allclose(np.array(st.ϵ_history), g["eps_history"], rtol=1e-9)
    np.testing.assert_allclose(np.array(st.u_history), g["u_history"], rtol=1e-9)
    np.testing.assert_allclose(np.array(st.ρ_history), g["rho_history"], rtol=1e-9)
    for j, (ln, head) in enumerate(zip(g["cdf_len"], g["cdf_knots_head"])):
        kn = st.cdfs_dist_prior.knots(j)
        assert len(kn) == ln
        np.testing.assert_allclose(kn[:8], head, rtol=1e-9, atol=1e-12)   # distances are differences: cancellation
