"""Edge cases of the path on the device: the smallest populations the reference's loop admits, odd sizes,
budgets that are not multiples of n, priors that gate almost every proposal, the maximum shapes of
the host-simulator mode, and simulators that return zeros."""
import numpy as np
import pytest

from tests.cases import MODELS, SEED, hip_model_prior, hip_proposal, oracle_proposal, oracle_run

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [2, 3, 4, 5, 63, 64, 65, 257])
def test_tiny_and_ragged_populations(S, O, gpu, n):
    """RandomWalk works from n = 2 (cov needs two particles); DE needs two particles per half batch (n >= 4)."""
    name, d = "gauss1_cfg2", 1
    model, prior = hip_model_prior(S, name)
    for prop in (("rw", "stretch") if n < 4 else ("rw", "stretch", "de")):
        res = S.sabc(model, prior, n_particles=n, n_simulation=n * 8 + (n - 1), proposal=hip_proposal(S, prop, d), seed=SEED)
        run = oracle_run(O, name, n, n * 8 + (n - 1), prop=prop)
        c = run.counters
        assert (res.state.n_simulation, res.state.n_accept, res.state.n_resampling, res.state.n_population_updates) == \
            (c["n_simulation"], c["n_accept"], c["n_resampling"], c["n_population_updates"])
        assert res.state.n_population_updates == 7
        np.testing.assert_allclose(res.population, run.theta[0], rtol=1e-6, atol=1e-9)


def test_too_few_particles_for_differential_evolution(S, gpu):
    model, prior = hip_model_prior(S, "gauss1_cfg2")
    with pytest.raises(S.SABCError) as e:       # the reference would spin forever in proposals.jl:104-107
        S.sabc(model, prior, n_particles=3, n_simulation=30, proposal=S.DifferentialEvolution(n_para=1), seed=SEED)
    assert e.value.code == -8


def test_prior_gate_rejects_most_proposals(S, O, gpu):
    """A narrow Uniform prior: most proposals fall outside, log_accept = -Inf (:320-322), the simulator is
    skipped for them, and n_simulation still counts them (:276,391)."""
    n = 2000
    model = S.GaussianIID(n_obs=100, sd=1.0, obs_mean=0.0)
    prior = S.Uniform(-0.01, 0.01)
    res = S.sabc(model, prior, n_particles=n, n_simulation=n * 11, proposal=S.StretchMove(), seed=SEED)
    assert res.state.n_simulation == n * 11 and np.all(np.abs(res.population) <= 0.01)
    cfg = O.make_config(n_particles=n, n_para=1, n_stats=1, model_id=O.MODEL_GAUSS_IID, model_params=model.params,
                        prior=[(O.PRIOR_UNIFORM, -0.01, 0.01)], seed=SEED)
    run = O.OracleRun(cfg)
    run.initialize(n * 11)
    run.update(O.make_update_args(n_simulation=n * 10, proposal=oracle_proposal(O, "stretch", 1), n_particles=n))
    assert res.state.n_accept == run.counters["n_accept"]
    np.testing.assert_allclose(res.population, run.theta[0], rtol=1e-7, atol=1e-12)


def test_zero_distances_are_dropped_from_the_ecdf(S, gpu):
    """cdf_estimators.jl:29: zeros are filtered before the knots are built; a statistic that is zero for some
    particles still gives a valid table (u = 0 for them)."""
    def f_dist(θ):
        return (abs(θ), 0.0 if θ < 0 else abs(θ) + 0.5)
    res = S.sabc(f_dist, S.Normal(0, 1), n_particles=400, n_simulation=400 * 6, seed=SEED)
    k1, k2 = res.state.cdfs_dist_prior.knots(0), res.state.cdfs_dist_prior.knots(1)
    assert len(k1) == 402 and 100 < len(k2) < 402 and k2[0] == 0.0 and k2[1] > 0.0
    assert np.all((res.u >= 0) & (res.u <= 1))
    with pytest.raises(S.SABCError) as e:       # every distance zero: maximum(x) of an empty collection (:33)
        S.sabc(lambda θ: 0.0, S.Normal(0, 1), n_particles=100, n_simulation=1000)
    assert e.value.code == -10


def test_maximum_shapes_in_host_mode(S, O, gpu):
    """d = 8 parameters, s = 8 statistics (the maxima of rounds 1-2), both eps schedules, every proposal,
    against the oracle driven by the same (deterministic, id-keyed) host simulator.  With 8 statistics the
    multi-eps schedule of :100-117 can leave the (0, 1/2) branch of its beta equation (mean u > 1/2 gives a
    negative beta): the device-side control step must follow the oracle there too."""
    truth = np.linspace(-1, 1, 8)
    def f_dist(θ, pid, it):
        z = np.array([O.normal_pair(SEED, pid, O.PURPOSE_SIM, it, b) for b in range(4)]).ravel()
        return np.abs(θ + 0.1 * z - truth)
    pri = [("N", 0.0, 2.0)] * 4 + [("U", -3.0, 3.0)] * 4
    prior = S.product_distribution([S.Normal(a, b) if k == "N" else S.Uniform(a, b) for k, a, b in pri])
    opri = [(O.PRIOR_NORMAL if k == "N" else O.PRIOR_UNIFORM, a, b) for k, a, b in pri]
    n, k = 200, 8
    for alg in ("single_eps", "multi_eps"):
        for prop in ("rw", "de", "stretch"):
            hd = S.HostDistance(f_dist, n_stats=8, n_para=8, univariate=False, with_ids=True)
            res = S.sabc(hd, prior, n_particles=n, n_simulation=n * (k + 1), algorithm=alg, proposal=hip_proposal(S, prop, 8),
                         resample=n // 2, seed=SEED)
            cb = O.host_simulator(f_dist, 8, 8)
            cfg = O.make_config(n_particles=n, n_para=8, n_stats=8, model_id=O.MODEL_HOST, model_params=[], prior=opri, seed=SEED,
                                algorithm=O.ALG_MULTI_EPS if alg == "multi_eps" else O.ALG_SINGLE_EPS, host_fn=cb)
            run = O.OracleRun(cfg)
            run.initialize(n * (k + 1))
            run.update(O.make_update_args(n_simulation=n * k, proposal=oracle_proposal(O, prop, 8), n_para=8, n_particles=n, resample=n // 2))
            assert res.population.shape == (n, 8) and res.u.shape == (n, 8) and len(res.state.ϵ) == (8 if alg == "multi_eps" else 1)
            assert res.state.n_accept == run.counters["n_accept"] and res.state.n_resampling == run.counters["n_resampling"]
            tol = {"rw": 1e-8, "stretch": 1e-6, "de": 1e-5}[prop]
            np.testing.assert_allclose(res.state.ϵ, run.eps, rtol=tol)
            np.testing.assert_allclose(res.population.T, run.theta, rtol=tol, atol=tol)
    with pytest.raises(S.SABCError):
        S.sabc(lambda θ: np.zeros(65) + 1, prior, n_particles=100, n_simulation=1000)     # s = 65 > SABC_MAX_STATS


def test_beyond_eight_dimensions_in_host_mode(S, O, gpu):
    """d = 12 parameters, s = 10 statistics (the reference takes any length(prior) and any number of distances,
    SimulatedAnnealingABC.jl:163-167,181; here up to 16 x 16 for host-callback and source-compiled simulators): a 12 x 12
    RandomWalk covariance and its Cholesky factor in the control step, 145 sum columns per particle, rows of d + s = 22
    doubles in the resample (no packed line).  Against the oracle driven by the same id-keyed host simulator."""
    d, s = 12, 10
    truth = np.linspace(-1, 1, d)
    def f_dist(θ, pid, it):
        z = np.array([O.normal_pair(SEED, pid, O.PURPOSE_SIM, it, b) for b in range(5)]).ravel()
        return np.abs(θ[:s] + 0.5 * θ[(np.arange(s) + 3) % d] + 0.1 * z - truth[:s])
    prior = S.product_distribution([S.Normal(0.0, 2.0)] * 6 + [S.Uniform(-3.0, 3.0)] * 6)
    opri = [(O.PRIOR_NORMAL, 0.0, 2.0)] * 6 + [(O.PRIOR_UNIFORM, -3.0, 3.0)] * 6
    n, k = 300, 8
    for alg, prop in (("single_eps", "rw"), ("multi_eps", "de"), ("single_eps", "stretch")):
        hd = S.HostDistance(f_dist, n_stats=s, n_para=d, univariate=False, with_ids=True)
        res = S.sabc(hd, prior, n_particles=n, n_simulation=n * (k + 1), algorithm=alg, proposal=hip_proposal(S, prop, d),
                     resample=40, seed=SEED)
        cfg = O.make_config(n_particles=n, n_para=d, n_stats=s, model_id=O.MODEL_HOST, model_params=[], prior=opri, seed=SEED,
                            algorithm=O.ALG_MULTI_EPS if alg == "multi_eps" else O.ALG_SINGLE_EPS, host_fn=O.host_simulator(f_dist, d, s))
        run = O.OracleRun(cfg)
        run.initialize(n * (k + 1))
        run.update(O.make_update_args(n_simulation=n * k, proposal=oracle_proposal(O, prop, d), n_para=d, n_particles=n, resample=40))
        assert res.population.shape == (n, d) and res.u.shape == (n, s) and len(res.state.ϵ) == (s if alg == "multi_eps" else 1)
        assert res.state.n_accept == run.counters["n_accept"] and res.state.n_resampling == run.counters["n_resampling"] >= 2
        tol = {"rw": 1e-8, "stretch": 1e-6, "de": 1e-5}[prop]
        np.testing.assert_allclose(res.state.ϵ, run.eps, rtol=tol)
        np.testing.assert_allclose(res.population.T, run.theta, rtol=tol, atol=tol)
        np.testing.assert_allclose(res.ρ.T, run.rho, rtol=tol, atol=tol)


def test_dozens_of_statistics_in_host_mode(S, O, gpu):
    """A time series as summary statistics: d = 3 parameters, s = 48 distances (one per time point; SABC_MAX_STATS = 64 for
    host-callback simulators -- the reference takes any number, SimulatedAnnealingABC.jl:164-167,181): 48 ECDF tables, a
    48-vector of epsilons under :multi_eps (its q^(s/2) terms and c_n at s = 48), rows of 51 doubles in the resample, 103
    sum columns.  Against the oracle driven by the same id-keyed host simulator."""
    d, s = 3, 48
    t = np.linspace(0.0, 4.0, s)
    obs = 2.0 * np.exp(-0.6 * t) + 0.3

    def f_dist(θ, pid, it):
        z = np.array([O.normal_pair(SEED, pid, O.PURPOSE_SIM, it, b) for b in range(s // 2)]).ravel()
        return np.abs(θ[0] * np.exp(-θ[1] * t) + θ[2] + 0.05 * z - obs)
    prior = S.product_distribution([S.Uniform(0.5, 4.0), S.Uniform(0.05, 2.0), S.Normal(0.0, 1.0)])
    opri = [(O.PRIOR_UNIFORM, 0.5, 4.0), (O.PRIOR_UNIFORM, 0.05, 2.0), (O.PRIOR_NORMAL, 0.0, 1.0)]
    n, k = 300, 8
    for alg, prop in (("multi_eps", "rw"), ("single_eps", "de"), ("multi_eps", "stretch")):
        hd = S.HostDistance(f_dist, n_stats=s, n_para=d, univariate=False, with_ids=True)
        res = S.sabc(hd, prior, n_particles=n, n_simulation=n * (k + 1), algorithm=alg, proposal=hip_proposal(S, prop, d),
                     resample=60, seed=SEED)
        cfg = O.make_config(n_particles=n, n_para=d, n_stats=s, model_id=O.MODEL_HOST, model_params=[], prior=opri, seed=SEED,
                            algorithm=O.ALG_MULTI_EPS if alg == "multi_eps" else O.ALG_SINGLE_EPS, host_fn=O.host_simulator(f_dist, d, s))
        run = O.OracleRun(cfg)
        run.initialize(n * (k + 1))
        run.update(O.make_update_args(n_simulation=n * k, proposal=oracle_proposal(O, prop, d), n_para=d, n_particles=n, resample=60))
        assert res.population.shape == (n, d) and res.u.shape == (n, s) and len(res.state.ϵ) == (s if alg == "multi_eps" else 1)
        assert res.state.n_accept == run.counters["n_accept"] and res.state.n_resampling == run.counters["n_resampling"] >= 2
        tol = {"rw": 1e-8, "stretch": 1e-6, "de": 1e-5}[prop]
        np.testing.assert_allclose(res.state.ϵ, run.eps, rtol=tol)
        np.testing.assert_allclose(res.population.T, run.theta, rtol=tol, atol=tol)
        np.testing.assert_allclose(res.ρ.T, run.rho, rtol=tol, atol=tol)
        np.testing.assert_allclose(np.array(res.state.ϵ_history), run.history[0], rtol=tol)


def test_the_largest_shapes_in_host_mode(S, O, gpu):
    """d = 16 parameters and s = 64 statistics at once (SABC_MAX_PARA, SABC_MAX_STATS): rows of 281 fused sums (more than a
    256-thread workgroup has lanes), a 16 x 16 covariance and Cholesky factor on the control lane, 64 epsilons under
    :multi_eps.  Against the oracle driven by the same id-keyed host simulator."""
    d, s = 16, 64
    truth = np.linspace(-1, 1, d)

    def f_dist(θ, pid, it):
        z = np.array([O.normal_pair(SEED, pid, O.PURPOSE_SIM, it, b) for b in range(s // 2)]).ravel()
        return np.abs(np.tile(θ - truth, 4) + 0.2 * z) + 0.01
    prior = S.product_distribution([S.Normal(0.0, 2.0)] * 8 + [S.Uniform(-3.0, 3.0)] * 8)
    opri = [(O.PRIOR_NORMAL, 0.0, 2.0)] * 8 + [(O.PRIOR_UNIFORM, -3.0, 3.0)] * 8
    n, k = 200, 8
    for alg, prop in (("multi_eps", "rw"), ("single_eps", "de")):
        hd = S.HostDistance(f_dist, n_stats=s, n_para=d, univariate=False, with_ids=True)
        res = S.sabc(hd, prior, n_particles=n, n_simulation=n * (k + 1), algorithm=alg, proposal=hip_proposal(S, prop, d),
                     resample=30, seed=SEED)
        cfg = O.make_config(n_particles=n, n_para=d, n_stats=s, model_id=O.MODEL_HOST, model_params=[], prior=opri, seed=SEED,
                            algorithm=O.ALG_MULTI_EPS if alg == "multi_eps" else O.ALG_SINGLE_EPS, host_fn=O.host_simulator(f_dist, d, s))
        run = O.OracleRun(cfg)
        run.initialize(n * (k + 1))
        run.update(O.make_update_args(n_simulation=n * k, proposal=oracle_proposal(O, prop, d), n_para=d, n_particles=n, resample=30))
        assert res.population.shape == (n, d) and res.u.shape == (n, s) and len(res.state.ϵ) == (s if alg == "multi_eps" else 1)
        assert res.state.n_accept == run.counters["n_accept"] and res.state.n_resampling == run.counters["n_resampling"] >= 1
        tol = {"rw": 1e-8, "de": 1e-5}[prop]
        np.testing.assert_allclose(res.state.ϵ, run.eps, rtol=tol)
        np.testing.assert_allclose(res.population.T, run.theta, rtol=tol, atol=tol)
        np.testing.assert_allclose(res.ρ.T, run.rho, rtol=tol, atol=tol)


def test_resample_disabled_and_every_update(S, O, gpu):
    """`resample` is any positive real (:255): inf never resamples, a tiny value resamples after every update."""
    n, name = 600, "gauss1_2stats"
    model, prior = hip_model_prior(S, name)
    for resample, expect in ((float("inf"), 1), (1e-9, 1 + 9)):
        res = S.sabc(model, prior, n_particles=n, n_simulation=10 * n, proposal=S.RandomWalk(n_para=1), resample=resample, seed=SEED)
        run = oracle_run(O, name, n, 10 * n, prop="rw", resample=resample)
        assert res.state.n_resampling == run.counters["n_resampling"] == expect
        assert res.state.n_accept == run.counters["n_accept"]
        np.testing.assert_allclose(res.population, run.theta[0], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("name", ["gauss1_2stats", "gauss2d_cfg3", "lv_cfg5", "gk_cfg4"])
def test_resample_draws_at_ragged_sizes(S, O, gpu, name):
    """The one-shard resample searches packed 128-byte lines (4, 2 or 1 particles per line, 16 lines per group, 1024 particles
    per chunk): population sizes that end in the middle of a line, a group and a chunk, one short of and one past a chunk,
    against the oracle's draws (the initialization ends with a resample, :124-137)."""
    model, prior = hip_model_prior(S, name)
    d = len(MODELS[name]["prior"])
    for n in (70, 1023, 1025, 2 * 1024 + 16 * 4 + 3, 33_003):
        res = S.sabc(model, prior, n_particles=n, n_simulation=n, proposal=hip_proposal(S, "rw", d), seed=SEED)
        run = oracle_run(O, name, n, n, prop="rw")
        assert res.state.n_resampling == run.counters["n_resampling"] == 1
        θ = res.population.T if d > 1 else res.population[None, :]
        # a draw that picked a neighbouring particle would differ in the first digits
        np.testing.assert_allclose(θ, run.theta, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(np.atleast_2d(res.u.T), run.u, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("name", ["gauss1_2stats", "gauss2d_cfg3"])
def test_resample_with_all_but_weightless_particles(S, O, gpu, name):
    """δ = 400 in w = exp(-δ Σ u/ū) (:126-127) leaves a few percent of the particles with all of the weight: most packed lines
    then add nothing to the running sum, whole runs of them share one bucket of the guide table, and a draw's walk from
    its guide entry falls back to bisecting the chunk (kernels.hip: packed_search).  A resample after every update."""
    from tests.cases import oracle_config
    n, k, delta = 20_011, 5, 400.0
    d = len(MODELS[name]["prior"])
    model, prior = hip_model_prior(S, name)
    res = S.sabc(model, prior, n_particles=n, n_simulation=(k + 1) * n, proposal=hip_proposal(S, "rw", d), resample=1e-9, δ=delta,
                 seed=SEED)
    run = O.OracleRun(oracle_config(O, name, n, delta=delta))
    run.initialize((k + 1) * n)
    run.update(O.make_update_args(n_simulation=k * n, proposal=oracle_proposal(O, "rw", d), n_para=d, n_particles=n, resample=1e-9,
                                  delta=delta))
    assert res.state.n_resampling == run.counters["n_resampling"] >= 1          # (1 + k under the default seed; the population
    assert res.state.n_accept == run.counters["n_accept"]                       #  can also collapse so far that nothing is accepted)
    θ = res.population.T if d > 1 else res.population[None, :]
    # Under some seeds (777: found by tools/seed_sweep.sh) the population collapses onto ONE particle: the RandomWalk's
    # Σ = β var(population) is then rounding noise of the one-pass variance (~1e-18), in which device and oracle agree
    # to a digit or two only, and the next proposals -- steps of ~1e-9 -- differ by that much.  Same draws, same lineage.
    collapsed = len(np.unique(θ[0])) < 5
    np.testing.assert_allclose(θ, run.theta, rtol=1e-6 if collapsed else 1e-9, atol=1e-12)
    assert len(np.unique(θ[0])) < 0.2 * n              # the weights really were that uneven


@pytest.mark.parametrize("name,prop", [("gauss1_cfg2", "rw"), ("gauss2_2stats", "de")])
def test_a_handle_can_be_initialised_again(S, gpu, name, prop):
    """sabc_initialize on a handle that has already run (bench.py repeats its timed region this way): the same seed gives the
    very same trajectory as the first time -- counters, epsilon, histories and particles."""
    from tests.cases import MODELS, SEED, hip_model_prior, hip_proposal
    n, k = 20_000, 8
    d = len(MODELS[name]["prior"])
    model, prior = hip_model_prior(S, name)
    h = S.SabcHandle(n_particles=n, model=model, prior=prior, seed=SEED)
    runs = []
    for _ in range(3):
        h.initialize((k + 1) * n)
        h.update(n_simulation=k * n, proposal=hip_proposal(S, prop, d), resample=n // 2)
        runs.append((dict(h.counters), h.eps.copy(), [a.copy() for a in h.history], [a.copy() for a in h.get_population()]))
    h.close()
    for r in runs[1:]:
        assert r[0] == runs[0][0] and r[0]["n_resampling"] >= 3
        np.testing.assert_array_equal(r[1], runs[0][1])
        for a, b in zip(r[2] + r[3], runs[0][2] + runs[0][3]):
            np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("prop", ["rw", "de"])
def test_accept_counter_beyond_two_to_the_32(S, gpu, prop):
    """A long-running chain: n_accept and the resample threshold (n_resampling + 1) * resample far beyond 2^32 (resumed from
    stored counters).  The device-side resample test, the two mailbox words the host polls (n_accept split at bit 32) and the
    history must behave exactly as for the same run counted from zero."""
    from tests.cases import SEED, hip_model_prior, hip_proposal
    n, k, resample = 4000, 12, 2000
    offset = resample * 5_000_000                                   # 1e10 > 2^33, a multiple of `resample`: the same firing pattern
    runs = []
    for start in (0, offset):
        model, prior = hip_model_prior(S, "gauss1_cfg2")
        h = S.SabcHandle(n_particles=n, model=model, prior=prior, seed=SEED)
        h.initialize((k + 1) * n)
        c = h.counters
        h.set_counters(c["n_simulation"], c["n_accept"] + start, c["n_resampling"] + start // resample, c["n_population_updates"])
        h.update(n_simulation=k * n, proposal=hip_proposal(S, prop, 1), resample=resample)
        runs.append((h.counters, h.get_population(), h.eps, h.history))
        h.close()
    (c0, p0, e0, h0), (c1, p1, e1, h1) = runs
    assert c1["n_accept"] == c0["n_accept"] + offset and c1["n_accept"] > 2 ** 33
    assert c1["n_resampling"] == c0["n_resampling"] + offset // resample and c0["n_resampling"] >= 3
    for a, b in zip(p0, p1):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(e0, e1)
    for a, b in zip(h0, h1):
        np.testing.assert_array_equal(a, b)
