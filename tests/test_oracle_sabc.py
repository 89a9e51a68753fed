"""The oracle's whole-algorithm path against the reference's integration tests
(test/runtests.jl:31-269: counters, eps < 1, every proposal runs) and against the analytic
conjugate posterior, which the reference's tests do not check.  CPU only."""
import numpy as np
import pytest

from tests.cases import MODELS, SEED, oracle_config, oracle_proposal, oracle_run, y_obs_mean

ALGS = ["multi_eps", "single_eps"]


def test_n_simulation_too_small(O):
    run = O.OracleRun(oracle_config(O, "gauss1_uniform", 100))
    with pytest.raises(O.OracleError) as e:                      # runtests.jl:39-40
        run.initialize(10)
    assert e.value.code == -1


@pytest.mark.parametrize("field,val,code", [("v", -0.1, -3), ("delta", -0.1, -4)])
def test_illegal_tuning_parameters(O, field, val, code):
    run = O.OracleRun(oracle_config(O, "gauss1_uniform", 100))
    run.initialize(1000)
    a = O.make_update_args(n_simulation=900, n_particles=100, **{field: val})
    with pytest.raises(O.OracleError) as e:                      # SimulatedAnnealingABC.jl:261-262
        run.update(a)
    assert e.value.code == code


@pytest.mark.parametrize("beta", [-0.1, 1.1])
def test_illegal_beta(O, beta):
    run = O.OracleRun(oracle_config(O, "gauss1_uniform", 100))
    run.initialize(1000)
    with pytest.raises(O.OracleError) as e:                      # proposals.jl:30
        run.update(O.make_update_args(n_simulation=900, n_particles=100, proposal=(O.PROP_RANDOMWALK, beta, 0)))
    assert e.value.code == -6


def test_bad_algorithm(O):
    cfg = oracle_config(O, "gauss1_uniform", 100)
    cfg.algorithm = 7
    with pytest.raises(O.OracleError) as e:                      # :462-464
        O.OracleRun(cfg)
    assert e.value.code == -5


@pytest.mark.parametrize("alg", ALGS)
@pytest.mark.parametrize("name", ["gauss1_uniform", "gauss2_meansd", "gauss1_2stats", "gauss2_2stats"])
def test_reference_integration_counters(O, name, alg):
    """runtests.jl:56-79,95-116,133-156,172-196 (default proposal = DifferentialEvolution)."""
    n = 100
    d = len(MODELS[name]["prior"])
    run = oracle_run(O, name, n, 1000, algorithm=alg, prop="de")
    c = run.counters
    assert c["n_simulation"] <= 1000 and c["n_population_updates"] == 9 and run.theta.shape == (d, n)
    if MODELS[name]["s"] > 1:
        assert np.all(run.eps < 1)                               # runtests.jl:140,179
    upd = lambda ns: run.update(O.make_update_args(n_simulation=ns, n_para=d, n_particles=n,
                                                   proposal=oracle_proposal(O, "de", d)))
    upd(1000)
    c = run.counters
    assert c["n_simulation"] <= 2000 and c["n_population_updates"] == 19
    upd(50)                                                      # too few simulations: no update
    assert run.counters["n_simulation"] == c["n_simulation"] and run.counters["n_population_updates"] == 19
    # state invariants of SimulatedAnnealingABC.jl:223,334,342
    assert run.counters["n_resampling"] >= 1 and 0 <= run.counters["n_accept"] <= 1900
    assert np.all((run.u >= 0) & (run.u <= 1)) and np.all(run.rho >= 0)


def test_state_after_initialization(O):
    run = O.OracleRun(oracle_config(O, "gauss1_uniform", 100))
    run.initialize(1000)
    assert run.counters == dict(n_simulation=100, n_accept=0, n_resampling=1, n_population_updates=0)   # :213,223
    e, u, r = run.history
    assert e.shape == (1, 1) and u.shape == (1, 1) and r.shape == (1, 1)                                # :180,207-208
    assert r[0, 0] == pytest.approx(run.rho.mean())
    # rho is NOT permuted by the initial resample (:131-132,225): u is, so cdf(rho) != u row by row
    kn = run.cdf_knots(0)
    assert len(kn) == 102 and kn[0] == 0 and kn[-1] == pytest.approx(1.5 * run.rho.max())
    u_of_rho = O.cdf_apply(kn, run.rho[0])
    assert not np.allclose(u_of_rho, run.u[0]) and set(np.round(run.u[0], 12)) <= set(np.round(u_of_rho, 12))


@pytest.mark.parametrize("prop", ["de", "stretch", "rw"])
@pytest.mark.parametrize("name", ["gauss1_uniform", "gauss2_meansd"])
def test_every_proposal_runs(O, name, prop):
    """runtests.jl:211-267."""
    run = oracle_run(O, name, 100, 1000, prop=prop)
    assert run.counters["n_simulation"] <= 1000
    d = len(MODELS[name]["prior"])
    run.update(O.make_update_args(n_simulation=1000, n_para=d, n_particles=100, proposal=oracle_proposal(O, prop, d)))
    assert run.counters["n_simulation"] <= 2000 and run.counters["n_accept"] > 0


def test_history_cadence(O):
    """:367-382: a row every checkpoint_history updates, plus a final row if the last one was skipped."""
    run = O.OracleRun(oracle_config(O, "gauss1_uniform", 100))
    run.initialize(1000)
    run.update(O.make_update_args(n_simulation=1000, n_particles=100, checkpoint_history=4))   # 10 updates: 4, 8, +final
    assert run.history[0].shape[0] == 1 + 3
    run.update(O.make_update_args(n_simulation=800, n_particles=100, checkpoint_history=4))    # 8 updates: 4, 8
    assert run.history[0].shape[0] == 1 + 3 + 2


def test_random_walk_sigma_is_beta_times_cov(O):
    """proposals.jl:47 (n-D, +1e-8 I) and :59 (1-D, no jitter)."""
    run = oracle_run(O, "gauss2_meansd", 200, 2000, prop="rw")
    np.testing.assert_allclose(run.sigma, 0.8 * (np.cov(run.theta) + 1e-8 * np.eye(2)), rtol=1e-10)
    run = oracle_run(O, "gauss1_uniform", 200, 2000, prop="rw")
    assert run.sigma[0, 0] == pytest.approx(0.8 * run.theta[0].var(ddof=1), rel=1e-10)


def test_eps_satisfies_its_equation_after_update(O):
    run = oracle_run(O, "gauss1_uniform", 200, 4000, prop="rw")
    ubar, e = run.u.mean(), run.eps[0]
    assert abs(e * e + e ** 1.5 - ubar * ubar) < 1e-12 * ubar * ubar          # :93 with v = 1


@pytest.mark.parametrize("prop", ["rw", "de", "stretch"])
def test_conjugate_gaussian_posterior(O, prop):
    """BASELINE config 1: theta ~ N(0, 2^2), x_1..100 ~ N(theta, 1), distance on the sufficient
    statistic -> the ABC posterior tends to N(post_mean, post_var) as eps -> 0."""
    n = 4000
    O.set_threads(8)
    try:
        run = oracle_run(O, "gauss1_cfg2", n, 60 * n, prop=prop)
    finally:
        O.set_threads(1)
    post_var = 1 / (1 / 4 + 100)
    post_mean = post_var * 100 * y_obs_mean()
    th = run.theta[0]
    assert run.eps[0] < 0.01
    # finite eps and a finite, resampled population: agreement to a fraction of the posterior sd (0.0999)
    assert abs(th.mean() - post_mean) < 0.02
    assert 0.7 < th.var() / post_var < 1.35
