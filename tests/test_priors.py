"""Prior families beyond the reference's tests (the reference takes any Distributions.Distribution,
SimulatedAnnealingABC.jl:163,174,314,318; the device path needs rand and logpdf as data): Gamma, Beta and truncated Normal
next to Normal / Uniform / Exponential / LogNormal.  Sampler and log density are checked against scipy.stats -- an
implementation that shares nothing with the oracle or the kernels -- on both sides (oracle: CPU; device: through the
C-ABI operator sabc_op_prior), and the device draws against the oracle's on the same Philox blocks."""
import numpy as np
import pytest
from scipy import stats

from tests.cases import SEED

FAMILIES = {
    # name: (oracle descriptor, scipy frozen distribution)
    "normal": (("N", 0.7, 1.3), stats.norm(0.7, 1.3)),
    "uniform": (("U", -2.0, 5.0), stats.uniform(-2.0, 7.0)),
    "exponential": (("E", 0.7, 0.0), stats.expon(scale=0.7)),
    "lognormal": (("L", -0.5, 0.5), stats.lognorm(0.5, scale=np.exp(-0.5))),
    "gamma": (("G", 2.5, 0.8), stats.gamma(2.5, scale=0.8)),
    "gamma_shape_below_one": (("G", 0.6, 2.0), stats.gamma(0.6, scale=2.0)),
    "beta": (("B", 2.0, 3.5), stats.beta(2.0, 3.5)),
    "beta_u_shaped": (("B", 0.5, 0.7), stats.beta(0.5, 0.7)),
    "truncnormal": (("T", 1.0, 2.0, -0.5, 2.5), stats.truncnorm((-0.5 - 1.0) / 2.0, (2.5 - 1.0) / 2.0, loc=1.0, scale=2.0)),
    "truncnormal_upper_tail": (("T", 0.0, 1.0, 2.0, 6.0), stats.truncnorm(2.0, 6.0)),
    # Phi(8) rounds to 1 - 6e-16: drawn in the mirrored lower tail, else every draw lands on a bound
    "truncnormal_far_tail": (("T", 0.0, 1.0, 8.0, 9.0), stats.truncnorm(8.0, 9.0)),
    "truncnormal_half_line": (("T", 1.0, 0.5, 5.5, np.inf), stats.truncnorm(9.0, np.inf, loc=1.0, scale=0.5)),
}
M = 40_000


def oracle_cfg(O, desc):
    kinds = {"N": O.PRIOR_NORMAL, "U": O.PRIOR_UNIFORM, "E": O.PRIOR_EXPONENTIAL, "L": O.PRIOR_LOGNORMAL, "G": O.PRIOR_GAMMA,
             "B": O.PRIOR_BETA, "T": O.PRIOR_TRUNCNORMAL}
    return O.make_config(n_particles=100, n_para=1, n_stats=1, model_id=O.MODEL_GAUSS_IID, model_params=[10, 1.0, 0.0, 0.0],
                         prior=[(kinds[desc[0]],) + tuple(desc[1:])], seed=SEED)


def hip_prior(S, desc):
    k, a = desc[0], desc[1:]
    return {"N": lambda: S.Normal(*a), "U": lambda: S.Uniform(*a), "E": lambda: S.Exponential(a[0]), "L": lambda: S.LogNormal(*a),
            "G": lambda: S.Gamma(*a), "B": lambda: S.Beta(*a), "T": lambda: S.truncated(S.Normal(a[0], a[1]), a[2], a[3])}[k]()


@pytest.mark.parametrize("name", sorted(FAMILIES))
def test_oracle_prior_against_scipy(O, name):
    desc, dist = FAMILIES[name]
    cfg = oracle_cfg(O, desc)
    x = np.array([O.prior_sample(cfg, pid)[0] for pid in range(M)])
    assert stats.kstest(x, dist.cdf).pvalue > 1e-3, stats.kstest(x, dist.cdf)
    lp = np.array([O.prior_logpdf(cfg, [v]) for v in x[:2000]])
    np.testing.assert_allclose(lp, dist.logpdf(x[:2000]), rtol=1e-11, atol=1e-11)
    lo, hi = dist.support()
    for outside in ([lo - 1.0] if np.isfinite(lo) else []) + ([hi + 1.0] if np.isfinite(hi) else []):
        assert O.prior_logpdf(cfg, [outside]) == -np.inf


def test_oracle_normal_quantile(O):
    L = O.lib()
    import ctypes as C
    L.orc_norm_quantile.argtypes, L.orc_norm_quantile.restype = [C.c_double], C.c_double
    p = np.concatenate([np.logspace(-300, -1, 200), np.linspace(0.1, 0.9, 101), 1 - np.logspace(-16, -1, 100)])
    got = np.array([L.orc_norm_quantile(float(v)) for v in p])
    np.testing.assert_allclose(got, stats.norm.ppf(p), rtol=2e-15 * 50, atol=1e-15)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(FAMILIES))
def test_device_prior_against_scipy_and_oracle(S, O, gpu, name):
    desc, dist = FAMILIES[name]
    h = S.SabcHandle(n_particles=256, model=S.GaussianIID(n_obs=10), prior=hip_prior(S, desc), seed=SEED)
    th, lp = h.prior(0, M)
    h.close()
    x = th[0]
    assert stats.kstest(x, dist.cdf).pvalue > 1e-3, stats.kstest(x, dist.cdf)
    np.testing.assert_allclose(lp, dist.logpdf(x), rtol=1e-10, atol=1e-10)
    cfg = oracle_cfg(O, desc)
    ref = np.array([O.prior_sample(cfg, pid)[0] for pid in range(4000)])
    # same Philox blocks, same algorithm: the draws agree (a rejection step decided differently in the last bit would show)
    np.testing.assert_allclose(x[:4000], ref, rtol=1e-9, atol=1e-300)


MU3 = np.array([0.5, -1.0, 2.0])
A3 = np.array([[1.5, 0.0, 0.0], [-0.7, 0.9, 0.0], [0.3, 0.4, 0.5]])
SIGMA3 = A3 @ A3.T


def mv_oracle_cfg(O, mu, sigma):
    d = len(mu)
    return O.make_config(n_particles=100, n_para=d, n_stats=1, model_id=O.MODEL_GAUSS_IID, model_params=[10, 1.0, 0.0, 0.0],
                         prior=[(O.PRIOR_NORMAL, float(mu[k]), float(np.sqrt(sigma[k, k]))) for k in range(d)], seed=SEED,
                         prior_chol=np.linalg.cholesky(sigma))


def check_mvnormal_draws(x, lp, mu, sigma):
    """x: d x M draws, lp their log densities: scipy's multivariate_normal for the density, the whitened draws
    L^-1 (x - mu) against N(0, 1) per coordinate and their cross moments for the sampler."""
    dist = stats.multivariate_normal(mu, sigma)
    np.testing.assert_allclose(lp, dist.logpdf(x.T), rtol=1e-11, atol=1e-11)
    z = np.linalg.solve(np.linalg.cholesky(sigma), x - mu[:, None])
    for k in range(len(mu)):
        assert stats.kstest(z[k], stats.norm.cdf).pvalue > 1e-3
    m = x.shape[1]
    np.testing.assert_allclose(np.cov(x), sigma, atol=6 * np.abs(sigma).max() * np.sqrt(2.0 / m))
    np.testing.assert_allclose(x.mean(axis=1), mu, atol=5 * np.sqrt(np.diag(sigma) / m).max())


def test_oracle_mvnormal_against_scipy(O):
    cfg = mv_oracle_cfg(O, MU3, SIGMA3)
    x = np.array([O.prior_sample(cfg, pid) for pid in range(M)]).T
    lp = np.array([O.prior_logpdf(cfg, x[:, i]) for i in range(M)])
    check_mvnormal_draws(x, lp, MU3, SIGMA3)
    # a diagonal Sigma is the product of Normals, draw for draw
    diag = np.diag([1.3 ** 2, 0.4 ** 2, 2.0 ** 2])
    cj = mv_oracle_cfg(O, MU3, diag)
    cp = O.make_config(n_particles=100, n_para=3, n_stats=1, model_id=O.MODEL_GAUSS_IID, model_params=[10, 1.0, 0.0, 0.0],
                       prior=[(O.PRIOR_NORMAL, MU3[0], 1.3), (O.PRIOR_NORMAL, MU3[1], 0.4), (O.PRIOR_NORMAL, MU3[2], 2.0)], seed=SEED)
    for pid in range(200):
        a, b = O.prior_sample(cj, pid), O.prior_sample(cp, pid)
        np.testing.assert_allclose(a, b, rtol=1e-15)
        assert O.prior_logpdf(cj, a) == pytest.approx(O.prior_logpdf(cp, b), rel=1e-13)


def test_mvnormal_rejects_bad_covariances(S):
    with pytest.raises(ValueError):
        S.MvNormal([0.0, 0.0], [[1.0, 2.0], [2.0, 1.0]])          # not positive definite
    with pytest.raises(ValueError):
        S.MvNormal([0.0, 0.0], [[1.0, 0.5], [0.1, 1.0]])          # not symmetric
    with pytest.raises(ValueError):
        S.MvNormal([0.0, 0.0, 0.0], [[1.0, 0.0], [0.0, 1.0]])     # wrong shape


@pytest.mark.gpu
def test_device_mvnormal_against_scipy_and_oracle(S, O, gpu):
    h = S.SabcHandle(n_particles=256, model=S.LotkaVolterra(n_steps=8), prior=S.MvNormal(MU3, SIGMA3), seed=SEED)
    th, lp = h.prior(0, M)
    h.close()
    check_mvnormal_draws(th, lp, MU3, SIGMA3)
    cfg = mv_oracle_cfg(O, MU3, SIGMA3)
    ref = np.array([O.prior_sample(cfg, pid) for pid in range(4000)]).T
    np.testing.assert_allclose(th[:, :4000], ref, rtol=1e-9, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("case,alg,prop", [("gauss2_gamma_sd", "single_eps", "rw"), ("gauss2_truncnormal_beta", "multi_eps", "de"),
                                           ("gauss2d_mvnormal", "multi_eps", "rw"), ("gauss2d_mvnormal", "single_eps", "stretch")])
def test_trajectory_parity_with_the_new_priors(S, O, gpu, case, alg, prop):
    from tests.cases import MODELS, hip_model_prior, hip_proposal, oracle_run
    n, k = 3000, 8
    d = len(MODELS[case]["prior"])
    model, prior = hip_model_prior(S, case)
    res = S.sabc(model, prior, n_particles=n, n_simulation=(k + 1) * n, proposal=hip_proposal(S, prop, d), algorithm=alg, seed=SEED,
                 resample=n // 2)
    run = oracle_run(O, case, n, (k + 1) * n, alg, prop, resample=n // 2)
    c = run.counters
    assert (res.state.n_accept, res.state.n_resampling) == (c["n_accept"], c["n_resampling"])
    tol = {"rw": 1e-9, "de": 1e-6, "stretch": 1e-6}[prop]
    np.testing.assert_allclose(res.population.T, run.theta, rtol=tol, atol=tol * 1e-2)
    np.testing.assert_allclose(res.state.ϵ, run.eps, rtol=tol)
