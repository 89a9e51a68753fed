"""The oracle against everything that pins it: the reference's own property tests
(test/runtests.jl:9-29), the equations written in the reference source, hand-derived vectors
 and published known answers (Random123 Philox KAT).  CPU only."""
import math

import numpy as np
import pytest
from scipy import stats

SQRT_EPS = math.sqrt(np.finfo(float).eps)


def test_philox_known_answers(O):
    # Random123 kat_vectors, philox4x32-10
    assert O.philox([0, 0], [0, 0, 0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert O.philox([0xFFFFFFFF] * 2, [0xFFFFFFFF] * 4) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert O.philox([0xA4093822, 0x299F31D0], [0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_uniform_open_interval_and_exact(O):
    assert O.lib().orc_u52(0, 0) == 2.0 ** -53
    assert O.lib().orc_u52(0xFFFFFFFF, 0xFFFFFFFF) == 1.0 - 2.0 ** -53
    assert O.lib().orc_u52(0x00080000, 0) == 0.5 + 2.0 ** -53        # mantissa = low 20 bits of hi : lo
    assert O.lib().orc_u52(0xFFF00000, 0) == 2.0 ** -53              # the top 12 bits of hi are not used


def test_normal_stream_moments(O):
    z = np.array([O.normal_pair(11, pid, O.PURPOSE_SIM, 3, k) for pid in range(200) for k in range(50)]).ravel()
    assert abs(z.mean()) < 4 / math.sqrt(z.size)
    assert abs(z.var() - 1) < 0.03
    assert stats.kstest(z, "norm").pvalue > 1e-3


# ---- test/runtests.jl:9-29 "cdf estimator" ----
@pytest.mark.parametrize("data", ["random", "repeats", "zeros"])
def test_cdf_reference_properties(O, data):
    rng = np.random.default_rng(1)
    x = {"random": rng.random(100) * 4, "repeats": [1, 2, 2, 3, 3, 3], "zeros": [1, 0, 2, 0, 3]}[data]
    kn = O.build_cdf(x)
    assert O.cdf_apply(kn, 0.0) <= SQRT_EPS                       # runtests.jl:13
    assert O.cdf_apply(kn, math.inf) == pytest.approx(1.0)        # runtests.jl:14
    q = np.sort(rng.random(100) * 3)
    assert np.all(np.diff(O.cdf_apply(kn, q)) >= 0)               # runtests.jl:15


def test_cdf_hand_derived_vectors(O):
    # knots and ordinates follow from cdf_estimators.jl:29-36 alone
    kn = O.build_cdf([1, 2, 2, 3, 3, 3])
    np.testing.assert_array_equal(kn, [0, 1, 2, 2, 3, 3, 3, 4.5])
    np.testing.assert_allclose(O.cdf_apply(kn, [0, 1, 3, 4.5]), [0, 1 / 7, 4 / 7, 1.0], rtol=0, atol=1e-16)
    assert O.cdf_apply(kn, 2.0) == pytest.approx(2 / 7)           # first of the duplicated knots
    assert O.cdf_apply(kn, 2.5) == pytest.approx(3 / 7 + 0.5 / 7) # interval of the LAST duplicate
    kn = O.build_cdf([1, 0, 2, 0, 3])
    np.testing.assert_array_equal(kn, [0, 1, 2, 3, 4.5])
    np.testing.assert_allclose(O.cdf_apply(kn, [0, 1, 2, 3, 4.5, 0.5, 3.75]), [0, .25, .5, .75, 1, .125, .875])
    assert O.cdf_apply(kn, -1.0) == 0.0 and O.cdf_apply(kn, 1e300) == 1.0


def test_cdf_all_zero_is_an_error(O):
    with pytest.raises(O.OracleError):
        O.build_cdf([0.0, 0.0])


# ---- epsilon schedules: residuals of the equations at SimulatedAnnealingABC.jl:93,113-114 ----
@pytest.mark.parametrize("ubar", [0.5, 0.3, 0.05, 1e-3, 1e-8])
@pytest.mark.parametrize("v", [0.3, 1.0, 5.0])
def test_eps_single_residual(O, ubar, v):
    e = O.eps_single(ubar, v)
    assert 0 < e < ubar
    assert abs(e * e + v * e ** 1.5 - ubar * ubar) < 1e-12 * ubar * ubar


def test_eps_single_zero_mean(O):
    assert O.eps_single(0.0, 1.0) == 0.0 and O.eps_single(1e-17, 1.0) == 0.0   # :93 ubar <= eps()


@pytest.mark.parametrize("ub", [0.01, 0.2, 0.4999, 0.5, 0.5001, 0.8])
def test_multi_eps_beta_residual(O, ub):
    b = O.lib().orc_multi_eps_beta(ub)
    if b == 0:
        assert ub == 0.5
        return
    f = 1 / b - 1 / math.expm1(b)        # == (1-e^-b(1+b))/(b(1-e^-b)), :113
    lit = (1 - math.exp(-b) * (1 + b)) / (b * (1 - math.exp(-b)))
    assert abs(f - ub) < 1e-12 and abs(lit - ub) < 1e-9


def test_eps_multi_closed_form(O):
    for s, catalan in [(1, 2), (2, 5), (3, 14), (4, 42)]:     # c_n = Catalan(s+1), :103
        ubar = np.linspace(0.2, 0.4, s)
        eps = O.eps_multi(ubar, 1.0)
        for i in range(s):
            q = ubar / ubar[i]
            num = 1 + np.sum(q ** (s / 2))
            den = catalan * (s + 1) * ubar[i] ** (1 + s / 2) * np.prod(q)
            beta = O.lib().orc_multi_eps_beta(float(ubar[i]))
            assert eps[i] == pytest.approx(1 / (beta + num / den), rel=1e-13)
    with pytest.raises(O.OracleError):
        O.eps_multi([0.3, 0.0], 1.0)                          # :107-109


def test_prior_logpdf_against_scipy(O):
    from tests.cases import oracle_config
    cfg = oracle_config(O, "gauss2_meansd", 10)
    assert O.prior_logpdf(cfg, [0.3, 0.5]) == pytest.approx(stats.norm.logpdf(0.3) + 0.0)
    assert O.prior_logpdf(cfg, [0.3, 1.5]) == -math.inf
    assert O.prior_logpdf(cfg, [0.3, 1.0]) == pytest.approx(stats.norm.logpdf(0.3))   # closed support
    th = np.array([O.prior_sample(cfg, i) for i in range(4000)])
    assert stats.kstest(th[:, 0], "norm").pvalue > 1e-3
    assert stats.kstest(th[:, 1], "uniform").pvalue > 1e-3


def test_simulators_are_what_design_md_says(O):
    """Recompute each simulator in numpy from the documented definition and the same normals."""
    from tests.cases import MODELS, oracle_config
    seed, pid, it = 20241220, 5, 3
    def normals(k):
        return np.array([O.normal_pair(seed, pid, O.PURPOSE_SIM, it, b) for b in range((k + 1) // 2)]).ravel()[:k]
    # Gaussian iid, two stats, d = 2
    cfg = oracle_config(O, "gauss2_2stats", 10)
    th = [0.3, 1.7]
    y = th[0] + th[1] * normals(10)
    np.testing.assert_allclose(O.simulate(cfg, th, pid, it), [abs(0 - y.mean()), abs(1 - (y * y).mean())], rtol=1e-13)
    # 2-D Gaussian
    cfg = oracle_config(O, "gauss2d_cfg3", 10)
    kw = MODELS["gauss2d_cfg3"]["model"][1]
    z = normals(100).reshape(50, 2)
    x = np.stack([1.0 + z[:, 0], -0.5 + 0.6 * z[:, 0] + 0.8 * z[:, 1]], 1)
    c = np.cov(x.T)
    want = [np.hypot(*(x.mean(0) - kw["obs_mean"])), abs(c[0, 0] + c[1, 1] - kw["obs_varsum"]), abs(c[0, 1] - kw["obs_cov"])]
    np.testing.assert_allclose(O.simulate(cfg, [1.0, -0.5], pid, it), want, rtol=1e-11)
    # g-and-k
    cfg = oracle_config(O, "gk_cfg4", 10)
    kw = MODELS["gk_cfg4"]["model"][1]
    A, B, g, k = 3.0, 1.0, 2.0, 0.5
    zz = normals(128)
    xs = np.sort(A + B * (1 + 0.8 * np.tanh(g * zz / 2)) * (1 + zz * zz) ** k * zz)
    want = [abs(xs[r - 1] - o) for r, o in zip(kw["ranks"], kw["obs"])]
    np.testing.assert_allclose(O.simulate(cfg, [A, B, g, k], pid, it), want, rtol=1e-12)
    # Lotka-Volterra
    cfg = oracle_config(O, "lv_cfg5", 10)
    kw = MODELS["lv_cfg5"]["model"][1]
    a, b, c_ = 1.0, 0.02, 0.8
    z = normals(512).reshape(256, 2)
    X, Y, xs, ys = 50.0, 50.0, [], []
    for t in range(256):
        dX = (a * X - b * X * Y) * 0.05 + 0.1 * X * math.sqrt(0.05) * z[t, 0]
        dY = (b * X * Y - c_ * Y) * 0.05 + 0.1 * Y * math.sqrt(0.05) * z[t, 1]
        X, Y = max(X + dX, 0.0), max(Y + dY, 0.0)
        xs.append(X); ys.append(Y)
    st = [np.mean(xs), np.std(xs, ddof=1), np.mean(ys), np.std(ys, ddof=1)]
    np.testing.assert_allclose(O.simulate(cfg, [a, b, c_], pid, it), np.abs(np.array(st) - kw["obs"]), rtol=1e-10)


def test_exponential_and_lognormal_priors_against_scipy(O):
    """Distributions.jl parametrisation: Exponential(theta = scale), LogNormal(mu, sigma)."""
    from tests.cases import oracle_config
    cfg = oracle_config(O, "gauss2_lognormal_sd", 10)
    for x in (0.05, 0.6, 3.0):
        assert O.prior_logpdf(cfg, [0.2, x]) == pytest.approx(stats.norm.logpdf(0.2) + stats.lognorm.logpdf(x, s=0.5, scale=math.exp(-0.5)), rel=1e-13)
    assert O.prior_logpdf(cfg, [0.2, 0.0]) == -math.inf and O.prior_logpdf(cfg, [0.2, -1.0]) == -math.inf
    th = np.array([O.prior_sample(cfg, i) for i in range(4000)])
    assert stats.kstest(th[:, 1], "lognorm", args=(0.5, 0, math.exp(-0.5))).pvalue > 1e-3
    cfg = oracle_config(O, "gauss2_exponential_sd", 10)
    for x in (0.0, 0.3, 5.0):
        assert O.prior_logpdf(cfg, [1.0, x]) == pytest.approx(math.log(0.25) + stats.expon.logpdf(x, scale=0.7), rel=1e-13)
    assert O.prior_logpdf(cfg, [1.0, -1e-9]) == -math.inf and O.prior_logpdf(cfg, [2.5, 1.0]) == -math.inf
    th = np.array([O.prior_sample(cfg, i) for i in range(4000)])
    assert stats.kstest(th[:, 1], "expon", args=(0, 0.7)).pvalue > 1e-3 and th[:, 1].min() > 0


@pytest.mark.parametrize("n", [1, 5, 1023, 1024, 1025, 4099, 300_000])
def test_weight_scan_and_resample_index(O, n):
    """The running weight sum of the resample (:129) in its specified blocked order: equal to the exact prefix sums
    up to rounding, total = last chunk offset + last chunk, and the draw on it is the inverse CDF (first k with
    cum[k] > t), also across chunk boundaries and for zero weights."""
    rng = np.random.default_rng(n)
    w = np.exp(-rng.random(n) * 3.0)
    if n > 4:
        w[rng.integers(0, n, size=max(n // 50, 1))] = 0.0          # exp(-a) underflows to 0 for hopeless particles
    cum, bs, tot = O.weight_scan(w)
    exact = np.cumsum(w.astype(np.longdouble))
    assert np.max(np.abs(cum - exact) / exact.clip(1e-300)) < 1e-13
    assert tot[0] == pytest.approx(float(exact[-1]), rel=1e-14) and tot[1] == pytest.approx(float(np.sum(w * w)), rel=1e-13)
    assert bs[0] == 0.0 and np.all(np.diff(bs) >= 0) and len(bs) == (n + 1023) // 1024
    ts = np.concatenate([rng.random(200) * tot[0], cum[:: max(n // 37, 1)], bs, [0.0, np.nextafter(tot[0], 0)]])
    for t in ts[ts < tot[0]]:
        k = O.resample_index(cum, bs, t)
        assert 0 <= k < n and cum[k] > t and (k == 0 or cum[k - 1] <= t or (k % 1024 == 0))    # chunk starts may sit one ulp off
        assert w[k] > 0.0 or k == n - 1


def test_cdf_equals_piecewise_linear_interpolation_property(O):
    """Property test (hypothesis): for arbitrary non-negative samples -- ties, zeros, wide dynamic range -- the oracle's
    estimator is the piecewise-linear interpolant through (knot_k, k / (len - 1)) with flat extrapolation
    (cdf_estimators.jl:29-42), it is monotone, and at a run of equal knots it takes the FIRST duplicate's ordinate."""
    from hypothesis import given, settings, strategies as st

    vals = st.one_of(st.floats(min_value=1e-290, max_value=1e6, allow_nan=False), st.sampled_from([0.0, 1.0, 2.5, 1e-12]))   # no denormals

    @settings(max_examples=150, deadline=None, derandomize=True)
    @given(st.lists(vals, min_size=1, max_size=60), st.lists(st.floats(min_value=-1.0, max_value=2e6, allow_nan=False), min_size=1, max_size=25))
    def check(xs, qs):
        xs = np.array(xs)
        if not np.any(xs > 0):
            with pytest.raises(O.OracleError):
                O.build_cdf(xs)
            return
        kn = O.build_cdf(xs)
        pos = np.sort(xs[xs > 0])
        np.testing.assert_array_equal(kn, np.concatenate([[0.0], pos, [1.5 * pos[-1]]]))
        y = np.arange(len(kn)) / (len(kn) - 1)
        q = np.sort(np.array(qs))
        got = np.atleast_1d(O.cdf_apply(kn, q))
        assert np.all(np.diff(got) >= -1e-15) and got[0] >= 0.0 and got[-1] <= 1.0
        for x, g in zip(q, got):
            if x <= 0:
                assert g == 0.0
            elif x >= kn[-1]:
                assert g == 1.0
            else:
                i0 = int(np.searchsorted(kn, x, side="left")) - 1           # last knot < x
                want = y[i0] + (y[i0 + 1] - y[i0]) * ((x - kn[i0]) / (kn[i0 + 1] - kn[i0]))
                assert g == pytest.approx(want, rel=1e-12, abs=1e-15)

    check()


def test_epsilon_schedules_property(O):
    """Property test (hypothesis) of both schedules over their whole input range: the single-eps root lies in (0, ubar) and
    satisfies eps^2 + v eps^1.5 = ubar^2 (:93); the multi-eps beta satisfies its equation (:113) for every mean in (0, 1);
    both are monotone in ubar (a population closer to the data gets a smaller tolerance)."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=300, deadline=None, derandomize=True)
    @given(st.floats(min_value=1e-12, max_value=1.0), st.floats(min_value=1e-3, max_value=1e3))
    def single(ubar, v):
        e = O.eps_single(ubar, v)
        assert 0.0 < e < ubar
        assert abs(e * e + v * e ** 1.5 - ubar * ubar) <= 1e-11 * ubar * ubar
        assert O.eps_single(ubar * 1.01, v) > e

    @settings(max_examples=300, deadline=None, derandomize=True)
    @given(st.floats(min_value=1e-6, max_value=1.0 - 1e-6))
    def beta(ub):
        b = O.lib().orc_multi_eps_beta(ub)
        if abs(b) < 1e-8:
            assert abs(ub - 0.5) < 1e-8
            return
        tail = 0.0 if b > 700 else (-1.0 if b < -700 else 1.0 / math.expm1(b))       # 1 / (e^b - 1) without overflow
        mean = 1.0 / b - tail if abs(b) > 1e-2 else 0.5 - b / 12 + b ** 3 / 720
        assert abs(mean - ub) < 1e-10
        assert (b > 0) == (ub < 0.5)

    single()
    beta()


@pytest.mark.parametrize("case,alg,prop", [("lv_cfg5", "single_eps", "rw"), ("gauss2d_cfg3", "multi_eps", "de"),
                                           ("gauss1_cfg2", "single_eps", "stretch")])
def test_rewritten_arithmetic_against_the_literal_expressions(O, case, alg, prop):
    """The factored Lotka-Volterra step and the weight-form ECDF interpolant were introduced in the oracle and the
    device code together; a frozen literal form of both (orc_set_literal) must give the same run up to rounding."""
    from tests.cases import oracle_run
    n, budget = 600, 600 * 9
    a = oracle_run(O, case, n, budget, alg, prop)
    O.set_literal(True)
    try:
        b = oracle_run(O, case, n, budget, alg, prop)
    finally:
        O.set_literal(False)
    assert a.counters == b.counters
    np.testing.assert_allclose(a.theta, b.theta, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(a.rho, b.rho, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(a.u, b.u, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(a.eps, b.eps, rtol=1e-9)
    assert not np.array_equal(a.rho, b.rho) or case != "lv_cfg5"      # the two forms do round differently
