"""Small shards run the population updates of a call in ONE launch (kernels.hip: k_update_persistent): every workgroup keeps
the control block, the ECDF index and the generator tables in LDS and loops over the updates itself -- body (a lane, a quad or
a row of 16 lanes per particle), the workgroups' rows exchanged as tagged words, their sums, the control step on its own copy
(its second part on a wave of its own beside the next update's drafts) -- until the resample test fires.  Same Philox streams,
same per-particle arithmetic, same control step as the launch chain (k_update -> k_reduce_control per update): the same
accept and resample counts, particles and epsilon to rounding (the rows are summed in another order); and both equal the
oracle (tests/test_gpu_parity.py runs at sizes that take the persistent form by default)."""
import numpy as np
import pytest

from tests.cases import MODELS, SEED, hip_model_prior, hip_proposal

pytestmark = pytest.mark.gpu


def run(S, name, alg, prop, n, k, monkeypatch, persistent, calls=1, resample=None, lanes=None):
    monkeypatch.setenv("SABC_PERSISTENT", "1" if persistent else "0")
    if lanes is None:
        monkeypatch.delenv("SABC_PERSISTENT_LANES", raising=False)
    else:
        monkeypatch.setenv("SABC_PERSISTENT_LANES", str(lanes))      # 1: a lane per particle | 4, 16: a quad, a row of lanes per particle
    monkeypatch.setenv("SABC_PERSISTENT_MAX", "65536")
    model, prior = hip_model_prior(S, name)
    d = len(MODELS[name]["prior"])
    h = S.SabcHandle(n_particles=n, model=model, prior=prior, seed=SEED,
                     algorithm=S._lib.ALG_MULTI_EPS if alg == "multi_eps" else S._lib.ALG_SINGLE_EPS)
    h.initialize((calls * k + 1) * n)
    l0 = h.kernel_launches
    for _ in range(calls):
        h.update(n_simulation=k * n, proposal=hip_proposal(S, prop, d), resample=resample or n // 3, checkpoint_history=3)
    out = dict(zip(("theta", "u", "rho"), h.get_population()), eps=h.eps, counters=dict(h.counters), hist=h.history,
               launches=h.kernel_launches - l0, sigma=h.proposal_sigma, lanes=h.persistent_lanes)
    h.close()
    return out


@pytest.mark.parametrize("name,alg,prop,n", [("gauss1_cfg2", "single_eps", "rw", 1000), ("gauss1_cfg2", "single_eps", "de", 1001),
                                             ("gauss2_2stats", "multi_eps", "stretch", 5000), ("gauss2d_cfg3", "multi_eps", "rw", 3000),
                                             ("gauss2d_cfg3", "single_eps", "de", 20_001), ("lv_cfg5", "single_eps", "rw", 700),
                                             ("gauss1_uniform", "single_eps", "de", 130), ("gauss2_meansd", "multi_eps", "rw", 40_000)])
@pytest.mark.parametrize("lanes", [1, 4, 16])
def test_one_launch_equals_the_launch_chain(S, gpu, monkeypatch, name, alg, prop, n, lanes):
    """lanes = 4: the four lanes of a quad run one particle and share its generator work (device_rng.hpp: NormalStream, coop) --
    the same streams bit for bit, so the same run; where four times the workgroups do not fit the launch (n = 40 000, 20 001
    with 512-thread workgroups) the request falls back to a lane per particle."""
    k = 14
    a = run(S, name, alg, prop, n, k, monkeypatch, persistent=False, calls=2)
    b = run(S, name, alg, prop, n, k, monkeypatch, persistent=True, calls=2, lanes=lanes)
    assert a["counters"] == b["counters"] and a["counters"]["n_resampling"] >= 3
    per_launch = n if prop == "rw" else n - n // 2
    block = 256 if b["rho"].shape[0] == 1 else 512          # threads of a workgroup (one statistic | more: kernels.hpp)
    thin = min(block - 64, 256)                             # a wave per SIMD at most, one wave of the block left to the control step
    fits = {w: -(-w * per_launch // thin) <= 256 or -(-w * per_launch // block) <= 256 for w in (4, 16)}
    expect = 16 if lanes == 16 and fits[16] else 4 if lanes >= 4 and fits[4] else 1     # (a team that does not fit: the next smaller)
    assert a["lanes"] == 0 and b["lanes"] == expect, (b["lanes"], per_launch)
    tol = 1e-10 if prop == "rw" else 1e-6                 # (DE / Stretch compound an ulp by ~(1 + 2 gamma) per update)
    for key in ("theta", "u", "rho", "eps", "sigma"):
        np.testing.assert_allclose(b[key], a[key], rtol=tol, atol=tol * 1e-2)
    for x, y in zip(a["hist"], b["hist"]):
        assert x.shape == y.shape
        np.testing.assert_allclose(y, x, rtol=tol)
    # the chain launches >= 2 kernels per update; the persistent form one per stretch between two resamples
    resamples = a["counters"]["n_resampling"] - 1
    assert a["launches"] >= 2 * 2 * k
    assert b["launches"] <= 2 * (resamples + 1) + 8 * resamples + 12, (b["launches"], resamples)


@pytest.mark.parametrize("name,prop,n", [("gauss1_cfg2", "rw", 1000), ("gauss2d_cfg3", "de", 3001), ("lv_cfg5", "stretch", 700)])
@pytest.mark.parametrize("team", [4, 16])
def test_a_quad_per_particle_is_the_same_run_as_a_lane_per_particle(S, gpu, monkeypatch, name, prop, n, team):
    """One update, no epsilon feedback in between: the quad-cooperative generator hands every simulator the stream the one-lane
    generator does, so the two forms make the same proposals, simulate the same distances and take the same decisions -- to
    the last bits only because they are two instantiations of the same source (the compiler is free to contract a*b + c
    differently in each: a 1-ulp difference in 2 % of the particles, as between k_update and either of them)."""
    a = run(S, name, "single_eps", prop, n, 1, monkeypatch, persistent=True, lanes=1, resample=10 ** 9)
    b = run(S, name, "single_eps", prop, n, 1, monkeypatch, persistent=True, lanes=team, resample=10 ** 9)
    assert a["counters"] == b["counters"] and a["counters"]["n_accept"] > 0 and (a["lanes"], b["lanes"]) == (1, team)
    for key in ("theta", "u", "rho"):
        np.testing.assert_allclose(b[key], a[key], rtol=1e-13, atol=1e-15)


def test_one_launch_stops_where_the_host_has_to_act(S, gpu, monkeypatch):
    """A resample threshold that fires after EVERY update: each launch does exactly one update, the host resamples, the next
    launch continues -- the same run as the chain; and a threshold that never fires: the whole call is one launch."""
    name, n, k = "gauss1_cfg2", 2000, 9
    a = run(S, name, "single_eps", "rw", n, k, monkeypatch, persistent=False, resample=1)
    b = run(S, name, "single_eps", "rw", n, k, monkeypatch, persistent=True, resample=1)
    assert a["counters"] == b["counters"] and b["counters"]["n_resampling"] == 1 + k
    np.testing.assert_allclose(b["theta"], a["theta"], rtol=1e-10)
    c = run(S, name, "single_eps", "rw", n, k, monkeypatch, persistent=True, resample=10 ** 9)
    assert c["counters"]["n_resampling"] == 1 and c["launches"] <= 6     # entry statistics + ONE update launch + the last history row


def test_large_shards_keep_the_launch_chain(S, gpu, monkeypatch):
    monkeypatch.delenv("SABC_PERSISTENT", raising=False)
    monkeypatch.delenv("SABC_PERSISTENT_MAX", raising=False)
    model, prior = hip_model_prior(S, "gauss1_cfg2")
    h = S.SabcHandle(n_particles=300_000, model=model, prior=prior, seed=SEED)
    h.initialize(300_000)
    l0 = h.kernel_launches
    h.update(n_simulation=5 * 300_000, proposal=hip_proposal(S, "rw", 1))
    assert h.kernel_launches - l0 >= 10
    h.close()


def test_a_simulator_from_source_takes_the_one_launch_form_too(S, O, gpu, monkeypatch):
    """k_update_persistent is compiled with the user's HIP source like k_update is (persistent_kernel.hpp through hipRTC): a
    model that exists only as source runs its small populations in one launch per stretch, and equals its launch-chain run."""
    from tests.test_user_simulator import GAUSS_IID_SRC
    n, k = 3000, 10
    outs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("SABC_PERSISTENT", mode)
        model = S.DeviceSource(GAUSS_IID_SRC, 1, 1, [100, 1.0, 1.4, 0.0])
        h = S.SabcHandle(n_particles=n, model=model, prior=S.Normal(0.0, 2.0), seed=SEED)
        h.initialize((k + 1) * n)
        l0 = h.kernel_launches
        h.update(n_simulation=k * n, proposal=hip_proposal(S, "de", 1), resample=n // 2)
        mid = h.kernel_launches
        h.update(n_simulation=k * n, proposal=hip_proposal(S, "de", 1), resample=10 ** 9)      # no resample: the call is ONE update launch
        outs.append(dict(theta=h.get_population()[0], counters=dict(h.counters), eps=h.eps, launches=mid - l0, quiet=h.kernel_launches - mid))
        h.close()
    a, b = outs
    assert a["counters"] == b["counters"] and a["counters"]["n_resampling"] >= 2
    np.testing.assert_allclose(b["theta"], a["theta"], rtol=1e-6, atol=1e-9)
    assert b["launches"] < a["launches"] and a["quiet"] >= 3 * k and b["quiet"] <= 6, (a["launches"], b["launches"], a["quiet"], b["quiet"])


def test_a_device_too_full_for_the_launch_hands_the_call_to_the_launch_chain(S, gpu, monkeypatch):
    """All workgroups of a one-launch update have to be resident at once; on a device shared with other handles' or processes'
    kernels they may not be.  Before anything is touched they meet at a rendezvous with a short bound; test hook: workgroup 1
    never shows up.  The others decide ABORT (one word, by compare-and-swap), the launch reports that it has touched nothing,
    and the engine runs the call as the launch chain: no error, the run of SABC_PERSISTENT=0 bit for bit."""
    import time
    name, n, k = "gauss1_cfg2", 2000, 6
    model, prior = hip_model_prior(S, name)

    def go(absent):
        if absent:
            monkeypatch.setenv("SABC_PERSISTENT_TEST_ABSENT_WG", "-2")
        else:
            monkeypatch.delenv("SABC_PERSISTENT_TEST_ABSENT_WG", raising=False)
        h = S.SabcHandle(n_particles=n, model=model, prior=prior, seed=SEED)
        h.initialize((2 * k + 1) * n)
        t0 = time.perf_counter()
        for _ in range(2):
            h.update(n_simulation=k * n, proposal=hip_proposal(S, "de", 1), resample=n // 3)
        out = (dict(h.counters), h.eps.copy(), [a.copy() for a in h.get_population()], h.persistent_launches, h.persistent_fallbacks,
               time.perf_counter() - t0)
        h.close()
        return out

    monkeypatch.setenv("SABC_PERSISTENT", "0")
    chain = go(False)
    monkeypatch.setenv("SABC_PERSISTENT", "1")
    full = go(True)
    # (one attempt, 20 ms; the second call comes within the 200 ms pause that follows a fallback and goes straight to the chain)
    assert full[3] == 0 and full[4] == 1 and full[5] < 2.0, full[3:]
    assert full[0] == chain[0]
    np.testing.assert_array_equal(full[1], chain[1])
    for a, b in zip(full[2], chain[2]):
        np.testing.assert_array_equal(a, b)
    ok = go(False)
    assert ok[3] > 0 and ok[4] == 0 and ok[0] == chain[0]


def test_a_workgroup_that_is_lost_fails_the_call_within_the_bound(S, gpu, monkeypatch):
    """The waits inside the launch are bounded like every wait of this library.  Test hook: workgroup 1 of the launch leaves right
    after the rendezvous (every workgroup was resident; one is lost).  The others run into the bound
    (50 ms here), raise the abort flag, and the call returns SABC_ERR_HIP -- nothing hangs; the error contract of sabc_update
    holds (counters, epsilon as at entry; the handle refuses updates until the particles are restored), and after restoring
    them the same handle repeats the call to exactly the run of a fresh handle."""
    import time
    name, n, k = "gauss1_cfg2", 2000, 6                  # 8 workgroups
    model, prior = hip_model_prior(S, name)
    h = S.SabcHandle(n_particles=n, model=model, prior=prior, seed=SEED)
    h.initialize((k + 1) * n)
    before = (dict(h.counters), h.eps.copy(), [a.copy() for a in h.get_population()])
    monkeypatch.setenv("SABC_PERSISTENT_TIMEOUT_MS", "50")
    monkeypatch.setenv("SABC_PERSISTENT_TEST_ABSENT_WG", "2")
    t0 = time.perf_counter()
    with pytest.raises(S.SABCError, match="grid barrier") as ei:
        h.update(n_simulation=k * n, proposal=hip_proposal(S, "rw", 1))
    assert ei.value.code == -21 and time.perf_counter() - t0 < 5.0
    assert dict(h.counters) == before[0]
    np.testing.assert_array_equal(h.eps, before[1])
    with pytest.raises(S.SABCError, match="half-updated"):
        h.update(n_simulation=n, proposal=hip_proposal(S, "rw", 1))
    monkeypatch.delenv("SABC_PERSISTENT_TEST_ABSENT_WG")
    h.set_population(*before[2])
    h.update(n_simulation=k * n, proposal=hip_proposal(S, "rw", 1))
    ref = S.SabcHandle(n_particles=n, model=model, prior=prior, seed=SEED)
    ref.initialize((k + 1) * n)
    ref.update(n_simulation=k * n, proposal=hip_proposal(S, "rw", 1))
    assert dict(h.counters) == dict(ref.counters)
    np.testing.assert_array_equal(h.get_population()[0], ref.get_population()[0])
    h.close(); ref.close()
