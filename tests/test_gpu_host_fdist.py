"""an arbitrary host callable as `f_dist` -- proposal, prior gate, ECDF, acceptance,
reductions and resampling on the device, only the simulator on the host.  This is what makes the
package a drop-in for every user model of the reference, not only the device-coded ones."""
import numpy as np
import pytest

from tests.cases import SEED, hip_model_prior, oracle_proposal

pytestmark = pytest.mark.gpu


def gauss_iid_keyed(O, n_obs, obs_mean):
    """f_dist(θ, particle_id, iter): the Gaussian i.i.d. simulator on the host, drawing from the same
    Philox stream the device-coded simulator uses -- so all three implementations must agree."""
    def f(θ, pid, it):
        z = np.array([O.normal_pair(SEED, pid, O.PURPOSE_SIM, it, b) for b in range((n_obs + 1) // 2)]).ravel()[:n_obs]
        x = float(np.atleast_1d(θ)[0]) + z
        s = 0.0
        for v in x:              # same summation order as the oracle / the kernel
            s += v
        return abs(obs_mean - s / n_obs)
    return f


@pytest.mark.parametrize("prop", ["rw", "de", "stretch"])
def test_host_simulator_matches_oracle_and_device_coded_model(S, O, gpu, prop):
    n, k, n_obs, ybar = 300, 6, 20, 1.4
    prior = S.Normal(0.0, 2.0)
    f = gauss_iid_keyed(O, n_obs, ybar)
    from tests.cases import hip_proposal
    hd = S.HostDistance(f, n_stats=1, n_para=1, univariate=True, with_ids=True)
    res = S.sabc(hd, prior, n_particles=n, n_simulation=(k + 1) * n, proposal=hip_proposal(S, prop, 1), resample=n // 4, seed=SEED)
    # the oracle with the very same Python callable as its f_dist
    cb = O.host_simulator(f, 1, 1)
    cfg = O.make_config(n_particles=n, n_para=1, n_stats=1, model_id=O.MODEL_HOST, model_params=[], seed=SEED,
                        prior=[(O.PRIOR_NORMAL, 0.0, 2.0)], host_fn=cb)
    run = O.OracleRun(cfg)
    run.initialize((k + 1) * n)
    run.update(O.make_update_args(n_simulation=k * n, proposal=oracle_proposal(O, prop, 1), n_para=1, n_particles=n, resample=n // 4))
    c = run.counters
    assert (res.state.n_accept, res.state.n_resampling, res.state.n_population_updates) == \
        (c["n_accept"], c["n_resampling"], c["n_population_updates"])
    tol = {"rw": 1e-9, "stretch": 1e-7, "de": 1e-6}[prop]
    np.testing.assert_allclose(res.population, run.theta[0], rtol=tol, atol=tol * 1e-2)
    np.testing.assert_allclose(res.ρ.T, run.rho, rtol=tol, atol=tol * 1e-2)
    np.testing.assert_allclose(res.state.ϵ, run.eps, rtol=tol)
    # ... and the device-coded simulator, which draws the same normals on the GPU
    dev = S.sabc(S.GaussianIID(n_obs=n_obs, sd=1.0, obs_mean=ybar), prior, n_particles=n, n_simulation=(k + 1) * n,
                 proposal=hip_proposal(S, prop, 1), resample=n // 4, seed=SEED)
    assert dev.state.n_accept == res.state.n_accept
    np.testing.assert_allclose(dev.population, res.population, rtol=tol, atol=tol * 1e-2)


@pytest.mark.parametrize("alg", ["multi_eps", "single_eps"])
def test_reference_closures_run_unchanged(S, gpu, alg):
    """test/runtests.jl:35-79 and :125-156 with the f_dist written as a plain closure, like in the reference."""
    rng = np.random.default_rng(1)
    f_dist = lambda θ: abs(0.0 - np.mean(rng.normal(θ, 1.0, 100)))                      # runtests.jl:35
    prior = S.Uniform(-10, 10)
    res = S.sabc(f_dist, prior, n_particles=100, n_simulation=1000, algorithm=alg, seed=SEED)   # (seeded: see test_gpu_parity.py)
    assert res.state.n_simulation <= 1000 and res.state.n_population_updates == 9 and len(res.population) == 100
    S.update_population_(res, f_dist, prior, n_simulation=1000)
    assert res.state.n_simulation <= 2000 and res.state.n_population_updates == 19
    n_sim = res.state.n_simulation
    S.update_population_(res, f_dist, prior, n_simulation=50)
    assert res.state.n_simulation == n_sim

    def two_stats(θ):                                                                    # runtests.jl:128-131
        y = rng.normal(θ[0], θ[1], 10)
        return (abs(0 - np.mean(y)), abs(1 - np.mean(y ** 2)))
    prior2 = S.product_distribution([S.Normal(0, 1), S.Uniform(0, 2)])
    res = S.sabc(two_stats, prior2, n_particles=100, n_simulation=1000, algorithm=alg, seed=SEED)
    assert np.all(res.state.ϵ < 1) and res.u.shape == (100, 2) and res.population.shape == (100, 2)


def test_any_dimension_and_statistics(S, gpu):
    """d = 5 parameters and s = 3 statistics: no device-coded simulator has this shape."""
    rng = np.random.default_rng(2)
    truth = np.array([1.0, -2.0, 0.5, 3.0, 0.0])
    def f_dist(θ, scale, offset=0.0):
        y = θ + scale * rng.standard_normal(5) + offset
        return np.array([np.abs(y - truth).mean(), abs(y.sum() - truth.sum()), np.abs(y - truth).max()])
    prior = S.product_distribution([S.Normal(0, 3)] * 5)
    res = S.sabc(f_dist, prior, 0.05, n_particles=400, n_simulation=400 * 40, proposal=S.RandomWalk(n_para=5), offset=0.0,
                 seed=SEED)                                  # args / kwargs are forwarded to f_dist (:315)
    assert res.population.shape == (400, 5) and res.u.shape == (400, 3) and res.state.n_population_updates == 39
    assert np.abs(res.population.mean(0) - truth).max() < 0.8           # the prior sd is 3
    sg = res._handle.proposal_sigma
    np.testing.assert_allclose(sg, 0.8 * (np.cov(res.population.T) + 1e-8 * np.eye(5)), rtol=1e-6)


def test_batched_host_simulator(S, gpu):
    rng = np.random.default_rng(3)
    def f_batch(Θ):                                                    # Θ: length-m vector (univariate prior)
        return np.abs(1.5 - (Θ[:, None] + rng.standard_normal((len(Θ), 100))).mean(1))
    hd = S.HostDistance(f_batch, n_stats=1, n_para=1, univariate=True, batched=True)
    res = S.sabc(hd, S.Normal(0, 2), n_particles=5000, n_simulation=5000 * 30, proposal=S.RandomWalk(n_para=1), seed=SEED)
    post_var = 1 / (1 / 4 + 100)
    assert abs(res.population.mean() - post_var * 100 * 1.5) < 0.03 and 0.5 < res.population.var() / post_var < 5.0   # 30 updates: not yet converged


def test_exception_in_f_dist_propagates(S, gpu):
    calls = {"n": 0}
    def f_dist(θ):
        calls["n"] += 1
        if calls["n"] > 150:
            raise ZeroDivisionError("simulator blew up")
        return abs(θ)
    with pytest.raises(ZeroDivisionError, match="blew up"):
        S.sabc(f_dist, S.Normal(0, 1), n_particles=100, n_simulation=1000)


def test_failed_update_leaves_the_state_untouched(S, gpu):
    """The reference works on copies and leaves `population_state` untouched when `update_population!` throws
    (SimulatedAnnealingABC.jl:264-267, :387-397).  Here the particles are updated in place on the device, so after a
    failure inside the loop the library puts counters, eps and histories back, and refuses further updates on the raw
    handle until the particles have been restored -- which `update_population_` does from the result's own arrays."""
    armed = {"on": False, "calls": 0}
    rng = np.random.default_rng(3)

    def f_dist(θ):
        armed["calls"] += 1
        if armed["on"] and armed["calls"] > 250:
            raise ZeroDivisionError("simulator blew up")
        return abs(0.3 - np.mean(rng.normal(θ, 1.0, 20)))
    prior = S.Normal(0, 1)
    res = S.sabc(f_dist, prior, n_particles=100, n_simulation=400, proposal=S.RandomWalk(n_para=1), seed=5)
    h = res._handle
    before = (dict(h.counters), h.eps.copy(), [a.copy() for a in h.history], res.population.copy(), res.u.copy())
    armed.update(on=True, calls=0)
    with pytest.raises(ZeroDivisionError, match="blew up"):
        S.update_population_(res, f_dist, prior, n_simulation=600, proposal=S.RandomWalk(n_para=1))
    assert dict(h.counters) == before[0]
    np.testing.assert_array_equal(h.eps, before[1])
    for a, b in zip(h.history, before[2]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(res.population, before[3])          # the result object was not refreshed
    assert res.state.n_population_updates == before[0]["n_population_updates"]
    with pytest.raises(S.SABCError, match="half-updated"):              # the raw handle: SABC_ERR_STATE until restored
        h.update(n_simulation=100, proposal=S.RandomWalk(n_para=1))
    armed.update(on=False)
    S.update_population_(res, f_dist, prior, n_simulation=200, proposal=S.RandomWalk(n_para=1))   # restores, then updates
    assert res.state.n_population_updates == before[0]["n_population_updates"] + 2
    assert res.state.n_simulation == before[0]["n_simulation"] + 200


def test_negative_distance_from_host_simulator(S, gpu):
    with pytest.raises(S.SABCError, match="Negative distances"):       # SimulatedAnnealingABC.jl:185
        S.sabc(lambda θ: θ, S.Normal(0, 1), n_particles=100, n_simulation=1000)


def test_docs_sir_example_as_host_f_dist(S, gpu):
    """The reference's documentation example (docs/src/example.md:75-198: stochastic SIR by Gillespie's
    algorithm, prior product_distribution([Uniform(0.1, 1), Uniform(0.05, 0.5)]), f_dist(θ, data_obs)
    returning one or two distances), restated in Python and run through the device loop as a host f_dist
    with `data_obs` as a positional argument, exactly like `sabc(f_dist, prior, data_obs; ...)` there."""
    rng = np.random.default_rng(11)

    def sir(β, γ, N=100, i0=5, t_max=30.0, grid=np.linspace(0, 30, 16)):
        s, i, t, k = N - i0, i0, 0.0, 0
        out = np.zeros(len(grid))
        while k < len(grid):
            rate_inf, rate_rec = β * s * i / N, γ * i
            total = rate_inf + rate_rec
            t_next = t + rng.exponential(1 / total) if total > 0 else np.inf
            while k < len(grid) and grid[k] < t_next:
                out[k] = i
                k += 1
            if not np.isfinite(t_next):
                break
            t = t_next
            if rng.random() < rate_inf / total:
                s, i = s - 1, i + 1
            else:
                i -= 1
        return out

    data_obs = sir(0.6, 0.15)

    def f_dist_single_stat(θ, data):
        return float(np.sqrt(np.mean((sir(θ[0], θ[1]) - data) ** 2)))

    def f_dist_multi_stats(θ, data):
        sim = sir(θ[0], θ[1])
        return (abs(sim.max() - data.max()), abs(sim.argmax() - data.argmax()) + 0.5 * abs(sim[-1] - data[-1]))

    prior = S.product_distribution([S.Uniform(0.1, 1), S.Uniform(0.05, 0.5)])
    res_1 = S.sabc(f_dist_single_stat, prior, data_obs, n_particles=200, n_simulation=200 * 25)
    res_2 = S.sabc(f_dist_multi_stats, prior, data_obs, n_particles=200, n_simulation=200 * 25, algorithm="multi_eps")
    for res, s in ((res_1, 1), (res_2, 2)):
        assert res.population.shape == (200, 2) and res.u.shape == (200, s) and res.state.n_population_updates == 24
        assert np.all((res.population[:, 0] >= 0.1) & (res.population[:, 0] <= 1) & (res.population[:, 1] >= 0.05) & (res.population[:, 1] <= 0.5))
    # the posterior has moved from the prior mean (0.55, 0.275) towards the truth (0.6, 0.15): gamma is well identified
    assert res_1.population[:, 1].mean() < 0.24
    S.update_population_(res_1, f_dist_single_stat, prior, data_obs, n_simulation=200 * 5, proposal=S.StretchMove())
    assert res_1.state.n_population_updates == 29
