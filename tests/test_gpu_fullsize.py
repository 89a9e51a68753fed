"""BASELINE.json full sizes (n_particles = 1e6) on the device: size-independent properties
the domain offers, plus moment parity with the oracle run at the same size and seed."""
import numpy as np
import pytest

from tests.cases import SEED, hip_model_prior, oracle_run, y_obs_mean

pytestmark = pytest.mark.gpu
N = 1_000_000
UPDATES = 55


@pytest.fixture(scope="module")
def cfg2(S, gpu):
    model, prior = hip_model_prior(S, "gauss1_cfg2")
    res = S.sabc(model, prior, n_particles=N, n_simulation=(UPDATES + 1) * N, proposal=S.RandomWalk(n_para=1), seed=SEED)
    return res


def test_cfg2_counters_and_invariants(cfg2):
    st = cfg2.state
    assert st.n_population_updates == UPDATES and st.n_simulation == (UPDATES + 1) * N
    assert 0 < st.n_accept <= UPDATES * N and st.n_resampling >= 2
    assert len(st.ϵ_history) == UPDATES + 1 and np.all(np.diff(np.array(st.ϵ_history)[:, 0]) < 0)     # annealing: eps falls
    assert np.all((cfg2.u >= 0) & (cfg2.u <= 1)) and np.all(cfg2.ρ >= 0) and np.all(np.isfinite(cfg2.population))
    ubar, e = cfg2.u.mean(), st.ϵ[0]
    assert abs(e * e + e ** 1.5 - ubar * ubar) < 1e-9 * ubar * ubar                          # :93, v = 1
    assert st.u_history[-1][0] == pytest.approx(ubar, rel=1e-12)
    assert st.ρ_history[-1][0] == pytest.approx(cfg2.ρ.mean(), rel=1e-10)      # a running sum: += the change per update


def test_cfg2_ecdf_table(cfg2):
    kn = cfg2.state.cdfs_dist_prior.knots(0)
    assert len(kn) == N + 2 and kn[0] == 0.0 and np.all(np.diff(kn) >= 0) and kn[-1] == 1.5 * kn[-2]
    q = np.sort(np.random.default_rng(0).random(1000) * kn[-1])
    u = np.array([cfg2.state.cdfs_dist_prior([x])[0] for x in q[::50]])
    assert np.all(np.diff(u) >= 0) and u[0] >= 0 and u[-1] <= 1


def test_cfg2_posterior_mean_matches_analytic_within_1_percent(cfg2):
    """The population mean is within 1 % of the conjugate posterior mean.  The VARIANCE is deliberately not held to the
    analytic value: the reference algorithm's population falls through it around update 60-90 and settles ~20 % below
    (RandomWalk; tests/test_independent_numpy.py and tests/test_gpu_anchors.py show the same trajectory in an independent
    NumPy restatement) -- UPDATES = 55 happens to sit near the crossing, which is why a 5 % band used to pass here."""
    post_var = 1 / (1 / 4 + 100)
    post_mean = post_var * 100 * y_obs_mean()
    th = cfg2.population
    assert abs(th.mean() / post_mean - 1) < 0.01
    assert 0.7 < th.var() / post_var < 1.6              # on its way through the analytic value, see above


def test_cfg2_moments_match_cpu_run_within_1_percent(S, O, cfg2):
    """North-star criterion: posterior mean and variance within 1 % of the CPU run, same seed.
    (They agree to ~1e-12 because the RNG streams are shared; 1 % is the stated bound.)"""
    O.set_threads(16)
    run = oracle_run(O, "gauss1_cfg2", N, (UPDATES + 1) * N, prop="rw")
    O.set_threads(1)
    th = run.theta[0]
    assert cfg2.state.n_accept == run.counters["n_accept"]
    assert abs(cfg2.population.mean() / th.mean() - 1) < 1e-9
    assert abs(cfg2.population.var() / th.var() - 1) < 1e-9
    np.testing.assert_allclose(cfg2.population, th, rtol=1e-8, atol=1e-11)


def test_cfg3_multistat_fullsize(S, O, gpu):
    """BASELINE config 3: 2-D correlated Gaussian, 3 statistics, both eps schedules, pop-cov
    RandomWalk; device moments against the CPU run at the same size and seed."""
    model, prior = hip_model_prior(S, "gauss2d_cfg3")
    for alg in ("single_eps", "multi_eps"):
        res = S.sabc(model, prior, n_particles=N, n_simulation=16 * N, algorithm=alg,
                     proposal=S.RandomWalk(n_para=2), seed=SEED)
        assert res.state.n_population_updates == 15 and np.all(res.state.ϵ < 1)
        assert np.all((res.u >= 0) & (res.u <= 1)) and np.all(res.ρ >= 0)
        # two of the three statistics carry no information on theta, so annealing is slow; the
        # population must nevertheless have moved from the prior mean (0,0) towards the data
        m = res.population.mean(0)
        assert m[0] > 0.2 and m[1] < -0.1
        sg = res._handle.proposal_sigma
        np.testing.assert_allclose(sg, 0.8 * (np.cov(res.population.T) + 1e-8 * np.eye(2)), rtol=1e-6)
        O.set_threads(16)
        run = oracle_run(O, "gauss2d_cfg3", N, 16 * N, algorithm=alg, prop="rw")
        O.set_threads(1)
        assert res.state.n_accept == run.counters["n_accept"] and res.state.n_resampling == run.counters["n_resampling"]
        np.testing.assert_allclose(m, run.theta.mean(1), rtol=1e-9)
        np.testing.assert_allclose(res.population.var(0), run.theta.var(1), rtol=1e-9)
        np.testing.assert_allclose(res.state.ϵ, run.eps, rtol=1e-9)


@pytest.mark.parametrize("prop,updates", [("rw", 300), ("de", 40)])
def test_long_run_soak(S, O, gpu, prop, updates):
    """Hundreds of population updates with many in-loop resamples through the queued-ahead loop (device-side
    resample test, halt flag, aborted launches, mailbox ring): accept and resample counts stay identical to
    the oracle's and the particles stay on the same trajectory."""
    from tests.cases import hip_proposal
    n, name = 100_000, "gauss1_2stats"
    model, prior = hip_model_prior(S, name)
    res = S.sabc(model, prior, n_particles=n, n_simulation=(updates + 1) * n, algorithm="multi_eps",
                 proposal=hip_proposal(S, prop, 1), resample=n // 3, seed=SEED, checkpoint_history=7)
    O.set_threads(16)
    run = oracle_run(O, name, n, (updates + 1) * n, algorithm="multi_eps", prop=prop, resample=n // 3, checkpoint_history=7)
    O.set_threads(1)
    c = run.counters
    assert (res.state.n_accept, res.state.n_resampling, res.state.n_population_updates) == \
        (c["n_accept"], c["n_resampling"], updates)
    assert res.state.n_resampling > (12 if prop == "rw" else 5)
    assert len(res.state.ϵ_history) == run.history[0].shape[0] == 1 + updates // 7 + (1 if updates % 7 else 0)
    tol = 1e-7 if prop == "rw" else 1e-4        # DE compounds differences by ~(1 + 2 gamma) per update
    np.testing.assert_allclose(res.state.ϵ, run.eps, rtol=tol)
    np.testing.assert_allclose(np.array(res.state.ϵ_history), run.history[0], rtol=tol)
    np.testing.assert_allclose(res.population, run.theta[0], rtol=tol, atol=tol)


@pytest.mark.parametrize("name,n,alg,prop,updates", [
    ("gk_cfg4", 4_000_000, "multi_eps", "de", 4),       # BASELINE config 4 at its full size, the reference's default proposal
    ("lv_cfg5", 1_000_000, "single_eps", "rw", 4),      # BASELINE config 5
])
def test_cfg4_cfg5_fullsize_against_cpu_run(S, O, gpu, name, n, alg, prop, updates):
    """g-and-k (wave-per-particle kernel, 128-value bitonic sort) and Lotka-Volterra (256 Euler-Maruyama steps)
    at BASELINE.json's particle counts: counters identical to the CPU run, state within the proposal's tolerance,
    and the size-independent invariants."""
    from tests.cases import hip_proposal, MODELS
    d = len(MODELS[name]["prior"])
    model, prior = hip_model_prior(S, name)
    res = S.sabc(model, prior, n_particles=n, n_simulation=(updates + 1) * n, algorithm=alg,
                 proposal=hip_proposal(S, prop, d), seed=SEED)
    st = res.state
    assert st.n_population_updates == updates and st.n_simulation == (updates + 1) * n
    assert res.population.shape == (n, d) and np.all(np.isfinite(res.population))
    assert np.all((res.u >= 0) & (res.u <= 1)) and np.all(res.ρ >= 0) and np.all(np.isfinite(res.ρ))
    for j in range(res.u.shape[1]):
        kn = st.cdfs_dist_prior.knots(j)
        assert kn[0] == 0.0 and np.all(np.diff(kn) >= 0) and kn[-1] == 1.5 * kn[-2]
    O.set_threads(16)
    run = oracle_run(O, name, n, (updates + 1) * n, algorithm=alg, prop=prop)
    O.set_threads(1)
    assert (st.n_accept, st.n_resampling) == (run.counters["n_accept"], run.counters["n_resampling"])
    tol = 1e-9 if prop == "rw" else 1e-6
    np.testing.assert_allclose(st.ϵ, run.eps, rtol=tol)
    # rho = |summary - observed|: the subtraction cancels, so the bound is absolute (summaries are O(10))
    np.testing.assert_allclose(res.ρ, run.rho.T, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(res.population.mean(0), run.theta.mean(1), rtol=tol)
    np.testing.assert_allclose(res.population.var(0), run.theta.var(1), rtol=tol)


def test_resample_beyond_the_lds_index(S, O, gpu):
    """n > 4096 * 1024 particles: the chunk offsets of the weight scan no longer fit the gather kernel's LDS copy and are
    searched in global memory instead; same draws as the oracle (the initial resample and one in-loop resample)."""
    from tests.cases import hip_proposal
    n, updates = 5_000_000, 3
    model, prior = hip_model_prior(S, "gauss1_cfg2")
    res = S.sabc(model, prior, n_particles=n, n_simulation=(updates + 1) * n, proposal=hip_proposal(S, "rw", 1), resample=n // 4, seed=SEED)
    O.set_threads(16)
    run = oracle_run(O, "gauss1_cfg2", n, (updates + 1) * n, prop="rw", resample=n // 4)
    O.set_threads(1)
    c = run.counters
    assert (res.state.n_accept, res.state.n_resampling) == (c["n_accept"], c["n_resampling"]) and c["n_resampling"] >= 2
    np.testing.assert_allclose(res.population, run.theta[0], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(res.u[:, 0], run.u[0], rtol=1e-7, atol=1e-12)
