"""f_dist as HIP source (SABC_MODEL_USER, include/sabc_hip.h: sabc_register_device_simulator): the user's simulator is
compiled at run time with hipRTC into the same fused update kernel as the built-in ones.  Closes the gap between the
reference's "any closure" (SimulatedAnnealingABC.jl:164,175,315) and a device path that needs the simulator as code.

CPU: the compiler stage alone (hipRTC needs no device).  GPU: the Gaussian i.i.d. simulator registered from source must
reproduce the compiled-in one BIT FOR BIT (same kernel template, same Philox streams, same reductions), and a simulator
that exists nowhere else in the repository must agree with the oracle driving the same arithmetic from Python."""
import numpy as np
import pytest

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
@pytest.mark.parametrize("prop", ["rw", "de", "stretch"])
def test_gauss_iid_from_source_is_bit_identical_to_the_built_in(S, gpu, prop):
    n, k, ybar = 20_000, 12, 1.4
    prior = S.Normal(0.0, 2.0)
    runs = []
    for model in (S.GaussianIID(n_obs=100, sd=1.0, obs_mean=ybar), S.DeviceSource(GAUSS_IID_SRC, 1, 1, [100, 1.0, ybar, 0.0])):
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
