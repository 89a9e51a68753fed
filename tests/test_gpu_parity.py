"""Parity of the HIP path against the oracle, through the C-ABI, on a real MI355X.

Integer / index work (Philox words, accept counts, resample counts, ECDF knot counts, sorted
knots) must be bit-exact.  Floating-point work is f64 on both sides; the only differences are
the device libm (log, sqrt, sincospi, exp, tanh, pow) and FMA contraction, so the tolerance
is rtol = 1e-9 on particle values after tens of population updates (observed ~1e-13)."""
import math

import numpy as np
import pytest

from tests.cases import MODELS, SEED, hip_model_prior, hip_proposal, oracle_config, oracle_proposal, oracle_run

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-9, atol=1e-12)
SQRT_EPS = math.sqrt(np.finfo(float).eps)


def test_philox_words_bit_exact(S, O, gpu):
    for seed, pid, purpose, it, k in [(0, 0, 0, 0, 0), (SEED, 123456, 1, 77, 49), (2**63 + 5, 2**33 + 9, 5, 2**32 + 3, 7)]:
        w, z = S.op_philox(seed, pid, purpose, it, k)
        assert w == O.stream_block(seed, pid, purpose, it, k)
        np.testing.assert_allclose(z, O.normal_pair(seed, pid, purpose, it, k), rtol=0, atol=4e-15)
    assert S.op_philox(0, 0, 0, 0, 0)[0] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]   # Random123 KAT


def test_normal_pairs_accuracy(S, O, gpu):
    """The device Box-Muller (table-driven log / sincos, range-specialised sqrt in csrc/device_rng.hpp) against the
    oracle's glibc normals on 200k Philox blocks: same words, so only the f64 math differs.  (Against exact
    arithmetic the device is within 3 ulp: tools/check_normals.py --exact; most of the 4e-15 here is the oracle's.)"""
    m = 200_000
    got = S.op_normal_pairs(SEED, 10**9, m, purpose=1, it=3, k=7)
    want = np.array([O.normal_pair(SEED, 10**9 + i, 1, 3, 7) for i in range(m)])
    err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    assert err.max() < 4e-15                      # a few ulp; the oracle's own cos(2 pi u) carries ~7e-16
    z = got.ravel()
    assert abs(z.mean()) < 4 / np.sqrt(z.size) and abs(z.var() - 1) < 0.01 and np.abs(z).max() < 8.6


@pytest.mark.parametrize("name", list(MODELS))
def test_device_simulators_match_oracle(S, O, gpu, name):
    model, prior = hip_model_prior(S, name)
    h = S.SabcHandle(n_particles=64, model=model, prior=prior, seed=SEED)
    cfg = oracle_config(O, name, 64)
    m = 300
    th = np.array([O.prior_sample(cfg, 1000 + i) for i in range(m)]).T            # [d][m]
    got = h.simulate(th, pid0=17, it=5)
    want = np.array([O.simulate(cfg, th[:, i], 17 + i, 5) for i in range(m)]).T
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-12)
    h.close()


# ---- test/runtests.jl:9-29 on the device operators ----
@pytest.mark.parametrize("data", ["random", "repeats", "zeros", "big"])
def test_cdf_estimator_on_device(S, O, gpu, data):
    rng = np.random.default_rng(3)
    x = {"random": rng.random(100) * 4, "repeats": np.array([1, 2, 2, 3, 3, 3.0]), "zeros": np.array([1, 0, 2, 0, 3.0]),
         "big": np.abs(rng.standard_normal(200_003))}[data]
    kn = S.op_build_cdf(x)
    np.testing.assert_array_equal(kn, O.build_cdf(x))                             # sorted knots: bit-exact
    assert S.op_cdf_eval(kn, 0.0) <= SQRT_EPS                                     # runtests.jl:13
    assert S.op_cdf_eval(kn, math.inf) == pytest.approx(1.0)                      # runtests.jl:14
    q = np.sort(rng.random(100) * 3)
    got = S.op_cdf_eval(kn, q)
    assert np.all(np.diff(got) >= 0)                                              # runtests.jl:15
    qq = np.concatenate([q, kn[:50], [-1.0, 0.0, kn[-1], kn[-1] * 2]])
    np.testing.assert_allclose(S.op_cdf_eval(kn, qq), O.cdf_apply(kn, qq), rtol=1e-14, atol=1e-16)


@pytest.mark.parametrize("n", [1, 2, 63, 255, 256, 257, 1023, 1024, 1025, 4097, 1_000_003])
def test_radix_sort_on_device(S, gpu, n):
    """The hand-written LSD radix sort behind build_cdf (csrc/sort.hip) on arbitrary doubles: negatives, signed zeros,
    infinities, denormals, heavy ties, every tile-boundary size."""
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) * 10.0 ** rng.integers(-300, 300, n)
    k = max(n // 7, 1)
    x[rng.integers(0, n, k)] = rng.choice([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, 5e-324, -5e-324, 1.5, 1.5, 1.5], k)
    got = S.op_sort(x)
    want = np.sort(x)
    np.testing.assert_array_equal(got, want)                       # (-0.0 == 0.0 compares equal: their mutual order is free)
    assert np.all(np.signbit(got[got < 0])) and not np.any(np.signbit(got[got > 0]))
    # signed zeros: all -0.0 before all +0.0 (the key order), nothing lost
    z = got[got == 0]
    assert np.array_equal(np.signbit(z), np.sort(np.signbit(z))[::-1]) and len(z) == np.sum(x == 0)
    # heavy ties and an already sorted / reversed input
    t = rng.integers(0, 3, n).astype(float)
    np.testing.assert_array_equal(S.op_sort(t), np.sort(t))
    np.testing.assert_array_equal(S.op_sort(want[::-1].copy()), want)


def test_cdf_estimator_errors(S, gpu):
    with pytest.raises(S.SABCError) as e:
        S.op_build_cdf([0.0, 0.0, 0.0])
    assert e.value.code == -10
    with pytest.raises(S.SABCError) as e:
        S.op_build_cdf([1.0, -0.5, 2.0])
    assert e.value.code == -2


# RandomWalk adds independent noise to a particle's own value: differences stay at libm level.
# StretchMove / DifferentialEvolution build theta' from 2-3 other particles, so a 1e-16 difference
# compounds by ~(1 + 2 gamma) per population update (3.4^12 ~ 2e6 for d = 1): looser bound, same counts.
PROP_TOL = {"rw": 1e-9, "stretch": 1e-7, "de": 1e-6}


def compare(res, run, d, prop="rw"):
    TOL = dict(rtol=PROP_TOL[prop], atol=PROP_TOL[prop] * 1e-2)
    st, c = res.state, run.counters
    assert (st.n_simulation, st.n_accept, st.n_resampling, st.n_population_updates) == \
        (c["n_simulation"], c["n_accept"], c["n_resampling"], c["n_population_updates"])
    pop = res.population.reshape(-1, 1) if d == 1 else res.population
    np.testing.assert_allclose(pop.T, run.theta, **TOL)
    # u = ECDF(rho): a 1e-15 relative difference in rho is amplified by rho / (knot spacing)
    np.testing.assert_allclose(res.u.T, run.u, rtol=TOL['rtol'], atol=max(1e-9, TOL['rtol']))
    np.testing.assert_allclose(res.ρ.T, run.rho, **TOL)
    np.testing.assert_allclose(st.ϵ, run.eps, rtol=TOL["rtol"])
    e, u, r = run.history
    np.testing.assert_allclose(np.array(st.ϵ_history), e, rtol=TOL["rtol"])
    np.testing.assert_allclose(np.array(st.u_history), u, rtol=TOL["rtol"])
    np.testing.assert_allclose(np.array(st.ρ_history), r, rtol=TOL["rtol"])


CASES = [(name, alg, prop) for name in MODELS for alg in ("single_eps", "multi_eps") for prop in ("rw", "de", "stretch")]


@pytest.mark.parametrize("name,alg,prop", CASES, ids=["-".join(c) for c in CASES])
def test_trajectory_parity(S, O, gpu, name, alg, prop):
    """sabc() on the device == sabc() on the oracle, same seed: identical accept / resample
    counts and particle values to 1e-9 after 12 population updates."""
    heavy = name.startswith("gk_") or name == "lv_cfg5"
    n, k = (301, 8) if heavy else (1001, 12)          # odd n: the two half batches differ in size (:300-301)
    d = len(MODELS[name]["prior"])
    O.set_threads(8)
    run = oracle_run(O, name, n, (k + 1) * n, algorithm=alg, prop=prop, resample=n // 4)
    O.set_threads(1)
    model, prior = hip_model_prior(S, name)
    res = S.sabc(model, prior, n_particles=n, n_simulation=(k + 1) * n, algorithm=alg,
                 proposal=hip_proposal(S, prop, d), resample=n // 4, seed=SEED)
    assert res.state.n_resampling >= 2               # the resample path ran inside the loop
    compare(res, run, d, prop)
    for j in range(MODELS[name]["s"]):
        kn = res.state.cdfs_dist_prior.knots(j)          # sorted device-simulated distances
        assert len(kn) == len(run.cdf_knots(j))
        np.testing.assert_allclose(kn, run.cdf_knots(j), rtol=1e-11, atol=1e-14)
    if prop == "rw":
        got = np.atleast_2d(res._handle.proposal_sigma)
        np.testing.assert_allclose(got, run.sigma, rtol=1e-8)


def test_resume_matches_oracle(S, O, gpu):
    """update_population! twice (docs/src/usage.md:43-45) == the oracle doing the same; a call
    with fewer simulations than particles changes nothing (runtests.jl:73-78)."""
    n = 500
    name, d = "gauss2_meansd", 2
    model, prior = hip_model_prior(S, name)
    res = S.sabc(model, prior, n_particles=n, n_simulation=6 * n, proposal=S.RandomWalk(n_para=d), seed=SEED)
    run = oracle_run(O, name, n, 6 * n, prop="rw")
    for prop, budget in (("de", 4 * n + 17), ("stretch", 3 * n), ("rw", n - 1)):
        S.update_population_(res, model, prior, n_simulation=budget, proposal=hip_proposal(S, prop, d), v=0.7, δ=0.2,
                             checkpoint_history=3)
        run.update(O.make_update_args(n_simulation=budget, n_para=d, n_particles=n, v=0.7, delta=0.2,
                                      proposal=oracle_proposal(O, prop, d), checkpoint_history=3))
        compare(res, run, d, "de")
    assert res.state.n_population_updates == 5 + 4 + 3


def test_cdfs_dist_prior_callable(S, O, gpu):
    res = S.sabc(*hip_model_prior(S, "gauss2_2stats"), n_particles=200, n_simulation=600, seed=SEED)
    run = oracle_run(O, "gauss2_2stats", 200, 600)
    f = res.state.cdfs_dist_prior
    for rho in ([0.1, 0.2], [0.0, 5.0], [1e9, 1e-9]):
        want = [O.cdf_apply(f.knots(j), rho[j]) for j in range(2)]     # same knots: only the lookup differs
        np.testing.assert_allclose(f(rho), want, rtol=1e-14, atol=1e-16)
        want = [O.cdf_apply(run.cdf_knots(j), rho[j]) for j in range(2)]
        np.testing.assert_allclose(f(rho), want, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("alg", ["multi_eps", "single_eps"])
@pytest.mark.parametrize("name", ["gauss1_uniform", "gauss2_meansd", "gauss1_2stats", "gauss2_2stats"])
def test_reference_integration_tests_on_device(S, gpu, name, alg):
    """test/runtests.jl:56-79,95-116,133-156,172-196 restated on the device path.  Seeded, unlike the reference: `all(ϵ .< 1)`
    after 1000 simulations of 100 particles is a property of most runs, not of all -- tools/stress_small_runs.py finds 1 seed
    in 200 (multi-eps, two statistics) where the device AND the oracle end the first stage with ϵ = (2.2, -1.6), the
    schedule's negative-β branch; an unseeded test would turn red once in a few dozen suite runs."""
    model, prior = hip_model_prior(S, name)
    res = S.sabc(model, prior, n_particles=100, n_simulation=1000, algorithm=alg, seed=SEED)
    assert res.state.n_simulation <= 1000 and res.state.n_population_updates == 9 and len(res.population) == 100
    if MODELS[name]["s"] > 1:
        assert np.all(res.state.ϵ < 1)
    S.update_population_(res, model, prior, n_simulation=1000)
    assert res.state.n_simulation <= 2000 and res.state.n_population_updates == 19
    n_sim = res.state.n_simulation
    S.update_population_(res, model, prior, n_simulation=50)
    assert res.state.n_simulation == n_sim


@pytest.mark.parametrize("prop", ["de", "stretch", "rw"])
@pytest.mark.parametrize("name", ["gauss1_uniform", "gauss2_meansd"])
def test_reference_proposal_tests_on_device(S, gpu, name, prop):
    """test/runtests.jl:211-267."""
    model, prior = hip_model_prior(S, name)
    p = hip_proposal(S, prop, len(prior))
    res = S.sabc(model, prior, proposal=p, n_particles=100, n_simulation=1000, seed=SEED)
    assert res.state.n_simulation <= 1000 and len(res.population) == 100
    S.update_population_(res, model, prior, proposal=p, n_simulation=1000)
    assert res.state.n_simulation <= 2000


def test_negative_distance_and_device_errors(S, gpu):
    with pytest.raises(S.SABCError) as e:
        S.sabc(S.GaussianIID(), S.Uniform(-1, 1), n_particles=100, n_simulation=1000, device=99)
    assert e.value.code == -20


def test_save_load_resume(S, O, gpu, tmp_path):
    """a stored result (ECDF knots included) resumes in a fresh handle exactly where the
    original would have continued (the RNG streams are keyed by the global update index)."""
    n, name, d = 777, "gauss2_2stats", 2
    model, prior = hip_model_prior(S, name)
    p = S.RandomWalk(n_para=d)
    a = S.sabc(model, prior, n_particles=n, n_simulation=7 * n, algorithm="multi_eps", proposal=p, seed=SEED)
    S.save_result(str(tmp_path / "res"), a)
    b = S.load_result(str(tmp_path / "res"), model, prior)
    assert b.state.n_population_updates == 6 and len(b.state.ϵ_history) == 7
    np.testing.assert_array_equal(b.population, a.population)
    for r in (a, b):
        S.update_population_(r, model, prior, n_simulation=6 * n, proposal=S.RandomWalk(n_para=d))
    assert (a.state.n_accept, a.state.n_resampling, a.state.n_population_updates) == \
        (b.state.n_accept, b.state.n_resampling, b.state.n_population_updates)
    np.testing.assert_allclose(b.population, a.population, rtol=1e-12)
    np.testing.assert_allclose(b.state.ϵ, a.state.ϵ, rtol=1e-12)
    assert len(b.state.ϵ_history) == len(a.state.ϵ_history) == 13
    run = oracle_run(O, name, n, 7 * n, algorithm="multi_eps", prop="rw")
    run.update(O.make_update_args(n_simulation=6 * n, n_para=d, n_particles=n, proposal=oracle_proposal(O, "rw", d)))
    compare(b, run, d, "rw")


def test_show_checkpoint_chunking_keeps_the_history(S, gpu, caplog):
    """`show_checkpoint` (SimulatedAnnealingABC.jl:359-364) only splits the call into chunks; counters,
    histories and particles are those of the unsplit call."""
    import logging
    model, prior = hip_model_prior(S, "gauss1_cfg2")
    kw = dict(n_particles=500, n_simulation=500 * 13, proposal=S.RandomWalk(n_para=1), seed=SEED, checkpoint_history=2)
    a = S.sabc(model, prior, **kw)
    with caplog.at_level(logging.INFO, logger="SimulatedAnnealingABC"):
        b = S.sabc(model, prior, show_checkpoint=4, **kw)
    assert any("Update 4 of 12" in r.message for r in caplog.records)
    assert len(a.state.ϵ_history) == len(b.state.ϵ_history) == 1 + 6
    np.testing.assert_allclose(b.population, a.population, rtol=1e-12)
    assert a.state.n_accept == b.state.n_accept
    c = S.sabc(model, prior, show_progressbar=True, **kw)            # ProgressMeter bar (:290-292,374): chunks only
    assert len(c.state.ϵ_history) == 1 + 6 and c.state.n_accept == a.state.n_accept
    np.testing.assert_allclose(c.population, a.population, rtol=1e-12)


def test_effective_sample_size(S, O, gpu):
    """ess = (sum w)^2 / sum w^2 of the last resample (:134), returned by the reference and unused by its callers."""
    n = 2000
    model, prior = hip_model_prior(S, "gauss1_cfg2")
    h = S.SabcHandle(n_particles=n, model=model, prior=prior, seed=SEED)
    h.initialize(n)
    from tests.cases import oracle_config
    run = O.OracleRun(oracle_config(O, "gauss1_cfg2", n))
    run.initialize(n)
    assert 0 < h.ess <= n and h.ess == pytest.approx(run.ess, rel=1e-9)
    h.close()
