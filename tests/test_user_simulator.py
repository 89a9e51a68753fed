"""f_dist as HIP source (SABC_MODEL_USER, include/sabc_hip.h: sabc_register_device_simulator): the user's simulator is
compiled at run time with hipRTC into the same fused update kernel as the built-in ones.  Closes the gap between the
reference's "any closure" (SimulatedAnnealingABC.jl:164,175,315) and a device path that needs the simulator as code.

CPU: the compiler stage alone (hipRTC needs no device).  GPU: the Gaussian i.i.d. simulator registered from source must
reproduce the compiled-in one BIT FOR BIT (same kernel template, same Philox streams, same reductions), and a simulator
that exists nowhere else in the repository must agree with the oracle driving the same arithmetic from Python."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from tests.cases import SEED, hip_proposal, oracle_proposal

GAUSS_IID_SRC = r"""
// Sim<SABC_MODEL_GAUSS_IID, 1, 1>::run of csrc/device_models.hpp, written as a user would
__device__ void sabc_user_simulate(const double *theta, const double *p, sabc::NormalStream &rng, double *rho) {
  const int n_obs = (int)p[0];
  const double mu = theta[0], sd = p[1];
  double sz = 0.0;
  for (int k = 0; k < (n_obs >> 1); ++k) {
    double z0, z1;
    rng.pair(z0, z1);
    sz += z0;
    sz += z1;
  }
  if (n_obs & 1) { double z0, z1; rng.pair(z0, z1); sz += z0; }
  rho[0] = fabs(p[2] - (mu + sd * sz / (double)n_obs));
}
"""

# the same simulator drawing its bulk through NormalStream::for_pairs -- the loop the built-in simulators use: the same stream,
# and where a quad of lanes runs the particle (small shards) sixteen pairs at a time
GAUSS_IID_FOR_PAIRS_SRC = r"""
__device__ void sabc_user_simulate(const double *theta, const double *p, sabc::NormalStream &rng, double *rho) {
  const int n_obs = (int)p[0];
  const double mu = theta[0], sd = p[1];
  double sz = 0.0;
  rng.for_pairs(n_obs >> 1, [&](const double z0, const double z1) { sz += z0; sz += z1; });
  if (n_obs & 1) { double z0, z1; rng.pair(z0, z1); sz += z0; }
  rho[0] = fabs(p[2] - (mu + sd * sz / (double)n_obs));
}
"""

# a model nothing else in the repository knows: exponential decay observed with noise at 8 times, 2 parameters
# (amplitude, rate), 2 statistics (mean absolute residual, absolute error of the last point); uses normals AND uniforms
DECAY_SRC = r"""
__device__ void sabc_user_simulate(const double *theta, const double *p, sabc::NormalStream &rng, double *rho) {
  const double amp = theta[0], rate = theta[1], noise = p[0];
  double acc = 0.0, last = 0.0;
  for (int k = 0; k < 4; ++k) {
    double z0, z1;
    rng.pair(z0, z1);
    const double t0 = 0.5 * (2 * k), t1 = 0.5 * (2 * k + 1);
    const double y0 = amp * exp(-rate * t0) + noise * z0, y1 = amp * exp(-rate * t1) + noise * z1;
    acc += fabs(y0 - p[1 + 2 * k]);
    acc += fabs(y1 - p[2 + 2 * k]);
    last = y1;
  }
  double u0, u1;
  rng.uniform_pair(u0, u1);                       // a multiplicative jitter on the second statistic
  rho[0] = acc / 8.0;
  rho[1] = fabs(last - p[8]) * (0.9 + 0.2 * u0) + 1e-3 * u1;
}
"""
DECAY_OBS = [3.0 * np.exp(-0.7 * 0.5 * k) for k in range(8)]


def test_source_compiles_without_a_device(S):
    assert S.DeviceSource(GAUSS_IID_SRC, 1, 1, [100, 1.0, 1.5, 0.0]).compile_check()
    assert S.DeviceSource(GAUSS_IID_FOR_PAIRS_SRC, 1, 1, [100, 1.0, 1.5, 0.0]).compile_check()
    assert S.DeviceSource(DECAY_SRC, 2, 2, [0.1] + DECAY_OBS).compile_check()


def test_compiler_errors_are_reported(S):
    with pytest.raises(S.SABCError) as e:
        S.DeviceSource("__device__ void sabc_user_simulate(int x) { }", 1, 1).compile_check()
    assert "sabc_user_simulate" in str(e.value) and "error" in str(e.value)
    with pytest.raises(S.SABCError):
        S.DeviceSource("this is not HIP", 1, 1).compile_check()
    with pytest.raises(S.SABCError):                       # no definition at all: the wrapper cannot call it
        S.DeviceSource("// nothing here", 2, 3).compile_check()


@pytest.mark.gpu
@pytest.mark.parametrize("persistent", ["0", "1"])
@pytest.mark.parametrize("prop", ["rw", "de", "stretch"])
@pytest.mark.parametrize("src,n,n_obs", [("pair", 20_000, 100), ("for_pairs", 3000, 100), ("for_pairs", 3000, 75), ("pair", 3000, 75)])
def test_gauss_iid_from_source_is_bit_identical_to_the_built_in(S, gpu, monkeypatch, prop, persistent, src, n, n_obs):
    """The same kernel templates, compiled at build time for the built-in simulator and at run time for the one from source:
    the launch chain (k_update) and the one-launch form of small shards (k_update_persistent) alike -- at n = 3000 with a quad
    of lanes per particle, where a simulator that draws through for_pairs takes sixteen pairs at a time (37 pairs: two groups
    of sixteen, a group of four, one pair of a last group, and the odd draw through pair()) and one that loops over pair()
    a group of four at a time: the same stream either way."""
    monkeypatch.setenv("SABC_PERSISTENT", persistent)
    k, ybar = 12, 1.4
    prior = S.Normal(0.0, 2.0)
    runs = []
    source = GAUSS_IID_SRC if src == "pair" else GAUSS_IID_FOR_PAIRS_SRC
    for model in (S.GaussianIID(n_obs=n_obs, sd=1.0, obs_mean=ybar), S.DeviceSource(source, 1, 1, [n_obs, 1.0, ybar, 0.0])):
        res = S.sabc(model, prior, n_particles=n, n_simulation=(k + 1) * n, proposal=hip_proposal(S, prop, 1), resample=n // 2, seed=SEED)
        runs.append(res)
    a, b = runs
    assert (a.state.n_accept, a.state.n_resampling) == (b.state.n_accept, b.state.n_resampling) and a.state.n_resampling >= 3
    np.testing.assert_array_equal(a.population, b.population)
    np.testing.assert_array_equal(a.u, b.u)
    np.testing.assert_array_equal(a.ρ, b.ρ)
    np.testing.assert_array_equal(a.state.ϵ, b.state.ϵ)
    np.testing.assert_array_equal(np.array(a.state.ϵ_history), np.array(b.state.ϵ_history))


@pytest.mark.gpu
def test_a_new_model_from_source_against_the_oracle(S, O, gpu):
    """The decay model exists only as the HIP source above; the oracle runs it through its host-callback model with a
    Python transcription that draws the same Philox blocks (O.normal_pair / O.stream_block)."""
    n, k = 2000, 8
    params = [0.1] + DECAY_OBS

    def f(θ, pid, it):
        amp, rate = θ
        acc, last = 0.0, 0.0
        for b in range(4):
            z0, z1 = O.normal_pair(SEED, pid, O.PURPOSE_SIM, it, b)
            t0, t1 = 0.5 * (2 * b), 0.5 * (2 * b + 1)
            y0, y1 = amp * np.exp(-rate * t0) + params[0] * z0, amp * np.exp(-rate * t1) + params[0] * z1
            acc += abs(y0 - params[1 + 2 * b])
            acc += abs(y1 - params[2 + 2 * b])
            last = y1
        w = O.stream_block(SEED, pid, O.PURPOSE_SIM, it, 4)
        u0, u1 = O.u52(w[0], w[1]), O.u52(w[2], w[3])
        return acc / 8.0, abs(last - params[8]) * (0.9 + 0.2 * u0) + 1e-3 * u1

    prior = S.product_distribution([S.Uniform(0.5, 6.0), S.Uniform(0.05, 2.0)])
    res = S.sabc(S.DeviceSource(DECAY_SRC, 2, 2, params), prior, n_particles=n, n_simulation=(k + 1) * n,
                 proposal=S.RandomWalk(n_para=2), resample=n // 2, algorithm="multi_eps", seed=SEED)
    cb = O.host_simulator(f, 2, 2)
    cfg = O.make_config(n_particles=n, n_para=2, n_stats=2, model_id=O.MODEL_HOST, model_params=[], seed=SEED,
                        prior=[(O.PRIOR_UNIFORM, 0.5, 6.0), (O.PRIOR_UNIFORM, 0.05, 2.0)], host_fn=cb, algorithm=O.ALG_MULTI_EPS)
    run = O.OracleRun(cfg)
    run.initialize((k + 1) * n)
    run.update(O.make_update_args(n_simulation=k * n, proposal=oracle_proposal(O, "rw", 2), n_para=2, n_particles=n, resample=n // 2))
    c = run.counters
    assert (res.state.n_accept, res.state.n_resampling) == (c["n_accept"], c["n_resampling"])
    np.testing.assert_allclose(res.population.T, run.theta, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res.ρ.T, run.rho, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(res.state.ϵ, run.eps, rtol=1e-9)
    # and the run moved towards the truth (amp 3, rate 0.7)
    assert abs(res.population[:, 0].mean() - 3.0) < 1.0 and abs(res.population[:, 1].mean() - 0.7) < 0.4


# distances on a grid of 1/8: thousands of tied ECDF knots (count data do this) -- and both statistics tie differently
TIES_SRC = r"""
__device__ void sabc_user_simulate(const double *theta, const double *p, sabc::NormalStream &rng, double *rho) {
  double z0, z1;
  rng.pair(z0, z1);
  rho[0] = floor(fabs(theta[0] + z0 - p[0]) * 8.0) / 8.0 + 0.125;
  rho[1] = floor(fabs(theta[0] + 0.5 * z1 - p[0]) * 2.0) / 2.0 + 0.5;
}
"""


@pytest.mark.gpu
@pytest.mark.parametrize("alg", ["single_eps", "multi_eps"])
def test_tied_distances_against_the_oracle(S, O, gpu, alg):
    """Discrete-valued distances give an ECDF table of a few dozen distinct values repeated thousands of times: the
    LDS-indexed three-level lookup of the fused update kernel (knots equal to the query on every level), the sort and the
    knot construction against the oracle's plain interpolation (cdf_estimators.jl:29-42)."""
    n, k, center = 6000, 8, 0.7

    def f(θ, pid, it):
        z0, z1 = O.normal_pair(SEED, pid, O.PURPOSE_SIM, it, 0)
        th = float(np.atleast_1d(θ)[0])
        return (float(np.floor(abs(th + z0 - center) * 8.0) / 8.0 + 0.125), float(np.floor(abs(th + 0.5 * z1 - center) * 2.0) / 2.0 + 0.5))
    prior = S.Normal(0.0, 2.0)
    res = S.sabc(S.DeviceSource(TIES_SRC, 1, 2, [center]), prior, n_particles=n, n_simulation=(k + 1) * n,
                 proposal=S.RandomWalk(n_para=1), resample=n // 2, algorithm=alg, seed=SEED)
    cfg = O.make_config(n_particles=n, n_para=1, n_stats=2, model_id=O.MODEL_HOST, model_params=[], seed=SEED,
                        prior=[(O.PRIOR_NORMAL, 0.0, 2.0)], host_fn=O.host_simulator(f, 1, 2),
                        algorithm=O.ALG_MULTI_EPS if alg == "multi_eps" else O.ALG_SINGLE_EPS)
    run = O.OracleRun(cfg)
    run.initialize((k + 1) * n)
    run.update(O.make_update_args(n_simulation=k * n, proposal=oracle_proposal(O, "rw", 1), n_para=1, n_particles=n, resample=n // 2))
    c = run.counters
    assert len(np.unique(res.ρ[:, 0])) < 100 and len(np.unique(res.ρ[:, 1])) < 30         # the ties are there
    assert (res.state.n_accept, res.state.n_resampling) == (c["n_accept"], c["n_resampling"])
    np.testing.assert_allclose(res.population, run.theta[0], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res.u.T, run.u, rtol=1e-9, atol=1e-12)
    np.testing.assert_array_equal(res.ρ.T, run.rho)
    np.testing.assert_allclose(res.state.ϵ, run.eps, rtol=1e-9)


@pytest.mark.gpu
def test_unregistered_or_broken_source_fails_loudly(S, gpu):
    with pytest.raises(S.SABCError) as e:
        S.sabc(S.DeviceSource("__device__ void sabc_user_simulate() {}", 1, 1), S.Normal(0, 1), n_particles=256, n_simulation=512)
    assert "compiling the device simulator failed" in str(e.value)


# ---- the shapes the path advertises (any d, s within the maxima): a generic simulator parametrised by (d, s) ----
SHAPE_SRC = r"""
__device__ void sabc_user_simulate(const double *theta, const double *p, sabc::NormalStream &rng, double *rho) {
  const int d = (int)p[0], s = (int)p[1];
  for (int j = 0; j < s; ++j) {
    const double z = rng.next();
    rho[j] = fabs(theta[j % d] + 0.5 * theta[(j + 1) % d] + p[2] * z - p[3 + j]);
  }
}
"""


def shape_params(d, s):
    return [d, s, 0.4] + [0.3 * ((j % 5) - 2) for j in range(s)]


def test_source_compiles_at_the_corner_shapes(S):
    """The compiler stage for the corners of (d, s) in 1..8 x 1..8 (the full grid: test below, opt-in -- ~4 s per shape)
    and for the maxima (16, 16): 128 KB of LDS index, 185 sum columns per particle."""
    for d, s in ((1, 1), (8, 8), (1, 8), (8, 1), (16, 16)):
        assert S.DeviceSource(SHAPE_SRC, d, s, shape_params(d, s)).compile_check(), (d, s)


@pytest.mark.skipif(not __import__("os").environ.get("SABC_RTC_FULL_GRID"), reason="64 compilations, ~4 min: set SABC_RTC_FULL_GRID=1")
def test_source_compiles_for_every_shape(S):
    from concurrent.futures import ThreadPoolExecutor

    def one(ds):
        try:
            return S.DeviceSource(SHAPE_SRC, ds[0], ds[1], shape_params(*ds)).compile_check()
        except S.SABCError as e:
            return str(e)
    shapes = [(d, s) for d in range(1, 9) for s in range(1, 9)]
    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(one, shapes))
    assert all(r is True for r in res), [(sh, r) for sh, r in zip(shapes, res) if r is not True]


@pytest.mark.gpu
@pytest.mark.parametrize("d,s,alg", [(8, 8, "multi_eps"), (4, 8, "single_eps"), (8, 1, "single_eps"), (5, 3, "multi_eps"),
                                     (12, 10, "single_eps")])
def test_source_simulator_at_the_advertised_shapes(S, O, gpu, d, s, alg):
    """(d, s) up to (8, 8): 128 KB of LDS index, 61 sum columns per particle, an 8 x 8 Cholesky in the control step.  Against
    the oracle's host-callback model driving the same arithmetic from Python with the same Philox blocks."""
    n, k = 1500, 5
    params = shape_params(d, s)

    def f(θ, pid, it):
        th = np.atleast_1d(θ)
        out = []
        for j in range(s):
            z = O.normal_pair(SEED, pid, O.PURPOSE_SIM, it, j // 2)[j % 2]
            out.append(abs(th[j % d] + 0.5 * th[(j + 1) % d] + params[2] * z - params[3 + j]))
        return tuple(out)

    prior = S.product_distribution([S.Normal(0.0, 1.0)] * d) if d > 1 else S.Normal(0.0, 1.0)
    res = S.sabc(S.DeviceSource(SHAPE_SRC, d, s, params), prior, n_particles=n, n_simulation=(k + 1) * n,
                 proposal=S.RandomWalk(n_para=d), resample=n // 2, algorithm=alg, seed=SEED)
    cfg = O.make_config(n_particles=n, n_para=d, n_stats=s, model_id=O.MODEL_HOST, model_params=[], seed=SEED,
                        prior=[(O.PRIOR_NORMAL, 0.0, 1.0)] * d, host_fn=O.host_simulator(f, d, s),
                        algorithm=O.ALG_MULTI_EPS if alg == "multi_eps" else O.ALG_SINGLE_EPS)
    run = O.OracleRun(cfg)
    run.initialize((k + 1) * n)
    run.update(O.make_update_args(n_simulation=k * n, proposal=oracle_proposal(O, "rw", d), n_para=d, n_particles=n, resample=n // 2))
    c = run.counters
    assert (res.state.n_accept, res.state.n_resampling) == (c["n_accept"], c["n_resampling"])
    pop = res.population.reshape(n, -1)
    np.testing.assert_allclose(pop.T, run.theta, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(res.ρ.T, run.rho, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(res.state.ϵ, run.eps, rtol=1e-8)


@pytest.mark.gpu
def test_philox_round_without_the_three_input_xor_is_bit_identical(S, gpu, monkeypatch):
    """csrc/device_rng.hpp uses v_bitop3_b32 (new on gfx950) where the compiler knows the builtin and two v_xor_b32 otherwise.
    The run-time compiled unit built WITHOUT it (-DSABC_NO_BITOP3) must reproduce the compiled-in kernels, which use it,
    bit for bit: same Philox words, hence the same run."""
    n, k, ybar = 20_000, 6, 1.4
    monkeypatch.setenv("SABC_RTC_EXTRA_FLAGS", "-DSABC_NO_BITOP3")
    prior = S.Normal(0.0, 2.0)
    a = S.sabc(S.GaussianIID(n_obs=100, sd=1.0, obs_mean=ybar), prior, n_particles=n, n_simulation=(k + 1) * n,
               proposal=hip_proposal(S, "de", 1), resample=n // 2, seed=SEED)
    b = S.sabc(S.DeviceSource(GAUSS_IID_SRC, 1, 1, [100, 1.0, ybar, 0.0]), prior, n_particles=n, n_simulation=(k + 1) * n,
               proposal=hip_proposal(S, "de", 1), resample=n // 2, seed=SEED)
    assert a.state.n_accept == b.state.n_accept
    np.testing.assert_array_equal(a.population, b.population)
    np.testing.assert_array_equal(a.ρ, b.ρ)


# ---- the prior, too, as device code in the same source (sabc_config::prior_joint = 3; Python: SourcePrior) ----
W_MIX = 0.35
MIX_SRC = r"""
// simulator: y_1..10 ~ N(theta_0, theta_1), distance of the sample mean to p[0]
__device__ void sabc_user_simulate(const double *theta, const double *p, sabc::NormalStream &rng, double *rho) {
  double sz = 0.0;
  for (int k = 0; k < 5; ++k) { double z0, z1; rng.pair(z0, z1); sz += z0; sz += z1; }
  rho[0] = fabs(p[0] - (theta[0] + theta[1] * sz / 10.0));
}
// prior: theta_0 ~ w Gamma(2, scale 0.5) + (1 - w) Gamma(3, scale 1) (integer shapes: sums of exponentials), theta_1 ~ Uniform(0.5, 2)
__device__ void sabc_user_prior_sample(const double *p, sabc::NormalStream &rng, double *th) {
  double u0, u1, u2, u3, u4, u5;
  rng.uniform_pair(u0, u1);
  rng.uniform_pair(u2, u3);
  rng.uniform_pair(u4, u5);
  th[0] = u0 < p[1] ? -0.5 * (log(u1) + log(u2)) : -(log(u1) + log(u2) + log(u3));
  th[1] = 0.5 + 1.5 * u4;
}
__device__ double sabc_user_prior_logpdf(const double *th, const double *p) {
  const double x = th[0], y = th[1];
  if (!(x > 0.0) || !(y >= 0.5 && y <= 2.0)) return -INFINITY;
  const double g2 = x * exp(-x / 0.5) / 0.25, g3 = x * x * exp(-x) / 2.0;
  return log(p[1] * g2 + (1.0 - p[1]) * g3) - log(1.5);
}
"""


def test_source_with_its_prior_compiles_without_a_device(S):
    assert S.DeviceSource(MIX_SRC, 2, 1, [1.2, W_MIX]).compile_check(with_prior=True)
    with pytest.raises(S.SABCError):                       # the prior functions are missing: the link of the unit fails
        S.DeviceSource(GAUSS_IID_SRC, 1, 1, [100, 1.0, 1.5, 0.0]).compile_check(with_prior=True)
    with pytest.raises(TypeError):
        S.sabc(S.GaussianIID(n_obs=10), S.SourcePrior(1), n_particles=100, n_simulation=200)


@pytest.mark.gpu
@pytest.mark.parametrize("prop", ["rw", "de"])
def test_prior_from_source_equals_the_same_prior_as_host_callbacks(S, O, gpu, prop):
    """A Gamma-mixture x Uniform prior that is none of the built-in families, (A) as device code next to the simulator
    (SourcePrior: rand and logpdf run inside the fused kernel) and (B) as host callbacks next to the same simulator as a
    host callable (HostPrior + HostDistance), both drawing the same Philox blocks: particle for particle the same run."""
    from scipy import stats
    n, k, center = 3000, 8, 1.2
    params = [center, W_MIX]
    res_a = S.sabc(S.DeviceSource(MIX_SRC, 2, 1, params), S.SourcePrior(2), n_particles=n, n_simulation=(k + 1) * n,
                   proposal=hip_proposal(S, prop, 2), resample=n // 2, seed=SEED)

    def f(θ, pid, it):
        sz = 0.0
        for b in range(5):
            z0, z1 = O.normal_pair(SEED, pid, O.PURPOSE_SIM, it, b)
            sz += z0
            sz += z1
        return abs(center - (θ[0] + θ[1] * sz / 10.0))

    def sample(ids):
        out = np.empty((len(ids), 2))
        for r, pid in enumerate(ids):
            w = [O.stream_block(SEED, int(pid), O.PURPOSE_PRIOR, 0, b) for b in range(3)]
            u = [O.u52(w[b][0], w[b][1]) for b in range(3)] + [O.u52(w[b][2], w[b][3]) for b in range(3)]
            u0, u2, u4, u1, u3, _ = u
            out[r, 0] = -0.5 * (np.log(u1) + np.log(u2)) if u0 < W_MIX else -(np.log(u1) + np.log(u2) + np.log(u3))
            out[r, 1] = 0.5 + 1.5 * u4
        return out

    def logpdf(th):
        x, y = th[:, 0], th[:, 1]
        with np.errstate(divide="ignore", invalid="ignore"):
            g2, g3 = x * np.exp(-x / 0.5) / 0.25, x * x * np.exp(-x) / 2.0
            lp = np.log(W_MIX * g2 + (1.0 - W_MIX) * g3) - np.log(1.5)
        return np.where((x > 0) & (y >= 0.5) & (y <= 2.0), lp, -np.inf)

    hd = S.HostDistance(f, n_stats=1, n_para=2, univariate=False, with_ids=True)
    res_b = S.sabc(hd, S.HostPrior(sample, logpdf, 2), n_particles=n, n_simulation=(k + 1) * n,
                   proposal=hip_proposal(S, prop, 2), resample=n // 2, seed=SEED)
    assert (res_a.state.n_accept, res_a.state.n_resampling) == (res_b.state.n_accept, res_b.state.n_resampling)
    tol = {"rw": 1e-9, "de": 1e-6}[prop]
    np.testing.assert_allclose(res_a.population, res_b.population, rtol=tol, atol=tol * 1e-2)
    np.testing.assert_allclose(res_a.ρ, res_b.ρ, rtol=tol, atol=tol * 1e-2)
    np.testing.assert_allclose(res_a.state.ϵ, res_b.state.ϵ, rtol=tol)
    # the device draws of the prior follow the mixture (sabc_op_prior through the run-time compiled unit)
    th, lp = res_a._handle.prior(0, 20_000)
    mix_cdf = lambda x: W_MIX * stats.gamma(2, scale=0.5).cdf(x) + (1 - W_MIX) * stats.gamma(3, scale=1.0).cdf(x)
    assert stats.kstest(th[0], mix_cdf).pvalue > 1e-3 and stats.kstest(th[1], stats.uniform(0.5, 1.5).cdf).pvalue > 1e-3
    np.testing.assert_allclose(lp, logpdf(th.T), rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("prop", ["rw", "de"])
def test_host_prior_next_to_a_simulator_from_source(S, gpu, prop):
    """A prior in HOST callbacks next to a simulator compiled from source (the third way to give a DeviceSource a prior, after
    data and SourcePrior): the run-time compiled k_simulate_batch runs between the proposal and the accept kernel, over the
    proposals the host's gate bytes let through.  Same run as with the prior as data."""
    from scipy import stats
    n, k = 1500, 6
    params = [0.1] + DECAY_OBS
    prior = S.product_distribution([S.Uniform(0.5, 6.0), S.Uniform(0.05, 2.0)])
    kw = dict(n_particles=n, n_simulation=(k + 1) * n, proposal=hip_proposal(S, prop, 2), resample=n // 3, algorithm="multi_eps", seed=SEED)
    ref = S.sabc(S.DeviceSource(DECAY_SRC, 2, 2, params), prior, **kw)
    helper = S.SabcHandle(n_particles=256, model=S.GaussianIID(n_obs=10, sd=1.0, obs_mean=0.0), prior=prior, seed=SEED)
    outside = []

    def logpdf(th):
        lp = stats.uniform(0.5, 5.5).logpdf(th[:, 0]) + stats.uniform(0.05, 1.95).logpdf(th[:, 1])
        outside.append(int(np.isneginf(lp).sum()))
        return lp
    hp = S.HostPrior(lambda ids: helper.prior(int(ids[0]), len(ids))[0].T, logpdf, 2, univariate=False)
    res = S.sabc(S.DeviceSource(DECAY_SRC, 2, 2, params), hp, **kw)
    helper.close()
    assert (res.state.n_accept, res.state.n_resampling) == (ref.state.n_accept, ref.state.n_resampling) and ref.state.n_resampling >= 1
    np.testing.assert_allclose(res.population, ref.population, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res.ρ, ref.ρ, rtol=1e-9, atol=1e-12)
    assert sum(outside) > 0                              # the gate was exercised: some proposals left the support


@pytest.mark.gpu
def test_compiled_simulators_are_kept_on_disk(S, gpu, tmp_path):
    """Compiling a simulator costs seconds, and every sabc() of every process paid them again.  The code object is kept under
    SABC_RTC_CACHE_DIR, keyed by everything that goes into the compilation (shape, flags, the user's source, the text of the
    headers, the ABI): a second PROCESS creates its handle in a fraction of the time and runs to the same numbers; a file that
    does not parse is ignored and replaced; SABC_RTC_CACHE=0 compiles every time."""
    import json
    import subprocess
    import sys
    code = r'''
import json, sys, time
sys.path.insert(0, %r)
import numpy as np
import sabc_amd as S
from tests.test_user_simulator import GAUSS_IID_SRC
t0 = time.perf_counter()
model = S.DeviceSource(GAUSS_IID_SRC + "\n// cache test %s\n", 1, 1, [100, 1.0, 1.4, 0.0])
h = S.SabcHandle(n_particles=3000, model=model, prior=S.Normal(0.0, 2.0), seed=7)
create = time.perf_counter() - t0
h.initialize(4 * 3000)
h.update(n_simulation=3 * 3000, proposal=S.DifferentialEvolution(n_para=1))
print(json.dumps(dict(create=create, n_accept=h.counters["n_accept"], mean=float(h.get_population()[0].mean()))))
''' % (ROOT, tmp_path.name)
    cache = tmp_path / "rtc_cache"

    def run(**env):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, SABC_RTC_CACHE_DIR=str(cache), **env))
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])

    first = run()
    files = sorted(cache.glob("*.sabcrtc"))
    assert len(files) == 1 and files[0].stat().st_size > 10_000
    second = run()
    assert (second["n_accept"], second["mean"]) == (first["n_accept"], first["mean"])
    assert second["create"] < 0.5 * first["create"], (first["create"], second["create"])     # (seconds of compilation against a file read)
    files[0].write_bytes(files[0].read_bytes()[:1000])                  # a truncated file: ignored, compiled afresh, replaced
    third = run()
    assert (third["n_accept"], third["mean"]) == (first["n_accept"], first["mean"]) and third["create"] > 2 * second["create"]
    assert sorted(cache.glob("*.sabcrtc"))[0].stat().st_size > 10_000
    off = run(SABC_RTC_CACHE="0")
    assert off["create"] > 2 * second["create"] and off["n_accept"] == first["n_accept"]
