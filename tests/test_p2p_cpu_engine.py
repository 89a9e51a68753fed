"""The host engine's peer-to-peer paths WITHOUT a GPU (engine.cpp: the exchange inside the control step, the barrier
between DE / Stretch half batches, the ECDF build and the resample over the owners' memory, the end-of-call status
exchange, abort, and the fall-back to the collectives underneath).

tests/cpu_engine/ref_backend.cpp emulates csrc/p2p.hpp between shards that live in ONE process (a host thread each: the
peers' memory is the pointer itself, the slots are atomics, every wait is bounded) -- the same protocol HipBackend's
kernels speak on the device, where tests/test_p2p.py runs it (two shards in one process, two and four processes over
hipIpc).  The reference is the same engine over gloo collectives, one process per rank."""
import ctypes as C
import threading
import time

import numpy as np
import pytest

from tests.test_distributed import launch


class ThreadCollectives:
    """allreduce / allgather between the shards' host threads (host pointers): what sits underneath the peer-to-peer
    transport in the fall-back tests."""

    def __init__(self, world):
        self.world, self.barrier, self.slots = world, threading.Barrier(world), [None] * world

    def hooks(self, rank):
        W = self.world

        def allreduce(ctx, buf, count, stream):
            a = np.ctypeslib.as_array((C.c_double * count).from_address(buf))
            self.slots[rank] = a.copy()
            self.barrier.wait()
            tot = self.slots[0].copy()
            for r in range(1, W):
                tot += self.slots[r]
            self.barrier.wait()
            a[:] = tot
            return 0

        def allgather(ctx, send, recv, count, stream):
            self.slots[rank] = np.ctypeslib.as_array((C.c_double * count).from_address(send)).copy()
            self.barrier.wait()
            out = np.ctypeslib.as_array((C.c_double * (count * W)).from_address(recv))
            for r in range(W):
                out[r * count:(r + 1) * count] = self.slots[r]
            self.barrier.wait()
            return 0

        return allreduce, allgather


def run_threads(S, case, alg, prop, n, k, resample, world, calls=1, timeout_ms=None, silence=None, underneath=None):
    """silence = (rank, call, posts_before): that shard lets `posts_before` posts go out in that call, then skips one."""
    from tests import cpu_engine
    from tests.cases import MODELS, SEED, hip_model_prior, hip_proposal
    Handle = cpu_engine.handle_class()
    d = len(MODELS[case]["prior"])
    descs, out, err = [None] * world, [None] * world, [None] * world
    barrier = threading.Barrier(world)

    def shard(rank):
        try:
            model, prior = hip_model_prior(S, case)
            h = Handle(n_particles=n, model=model, prior=prior, seed=SEED, rank=rank, world=world,
                       algorithm=S._lib.ALG_MULTI_EPS if alg == "multi_eps" else S._lib.ALG_SINGLE_EPS)
            if underneath is not None:
                ar, ag = underneath.hooks(rank)
                h.set_collectives(ar, ag, False)

            def setup():
                descs[rank] = h.p2p_descriptor()
                barrier.wait()
                if timeout_ms:
                    h.p2p_set_timeout(timeout_ms)
                h.p2p_init(list(descs))
                barrier.wait()
                assert h.p2p_active

            setup()
            h.initialize((calls * k + 1) * n)
            res = dict(errors=[], seconds=[])
            for call in range(calls):
                before = (dict(h.counters), h.eps.copy(), [a.copy() for a in h.history], [a.copy() for a in h.get_population()])
                if silence and silence[0] == rank and silence[1] == call:
                    if len(silence) > 3 and silence[3] == "loss":      # the post reaches the shard's own slots only
                        h.p2p_inject_loss(silence[2])
                    else:
                        h.p2p_inject_silence(-silence[2])
                t0 = time.perf_counter()
                try:
                    h.update(n_simulation=k * n, proposal=hip_proposal(S, prop, d), resample=resample)
                    res["errors"].append(None)
                except S.SABCError as e:
                    res["errors"].append(e)
                    res["seconds"].append(time.perf_counter() - t0)
                    assert dict(h.counters) == before[0] and not h.p2p_active
                    np.testing.assert_array_equal(h.eps, before[1])
                    for a, b in zip(h.history, before[2]):
                        np.testing.assert_array_equal(a, b)
                    with pytest.raises(S.SABCError, match="half-updated"):
                        h.update(n_simulation=n, proposal=hip_proposal(S, prop, d))
                    barrier.wait()
                    h.set_population(*before[3])
                    setup()
                    h.update(n_simulation=k * n, proposal=hip_proposal(S, prop, d), resample=resample)
            th, u, rho = h.get_population()
            res.update(theta=th, u=u, rho=rho, eps=h.eps, counters=h.counters, collective_calls=h.collective_calls, hist=h.history,
                       fallbacks=h.p2p_fallbacks, active=h.p2p_active, comm=h.comm_bytes)
            out[rank] = res
            barrier.wait()
            h.close()
        except BaseException as e:
            err[rank] = e
            barrier.abort()
            if underneath is not None:
                underneath.barrier.abort()

    ts = [threading.Thread(target=shard, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    assert all(e is None for e in err), err
    return out


def check(out, ref, tol=1e-10):
    theta = np.concatenate([o["theta"] for o in out], 1)
    c = out[0]["counters"]
    assert [c[q] for q in ("n_simulation", "n_accept", "n_resampling", "n_population_updates")] == list(ref["counters"])
    assert all(o["counters"] == c for o in out)
    np.testing.assert_allclose(theta, ref["theta"], rtol=tol, atol=tol * 1e-2)
    np.testing.assert_allclose(np.concatenate([o["rho"] for o in out], 1), ref["rho"], rtol=tol, atol=tol * 1e-2)
    np.testing.assert_allclose(out[0]["eps"], ref["eps"], rtol=tol)
    np.testing.assert_allclose(out[0]["hist"][0], ref["eps_hist"], rtol=tol)
    for o in out[1:]:
        np.testing.assert_array_equal(o["eps"], out[0]["eps"])          # rank-order sums: the same control step on every shard


@pytest.mark.parametrize("world,case,alg,prop,n", [(2, "gauss1_cfg2", "single_eps", "rw", 1001), (3, "gauss2_2stats", "multi_eps", "de", 1000),
                                                   (2, "gauss2d_cfg3", "single_eps", "stretch", 777), (3, "gauss2_meansd", "single_eps", "rw", 1003),
                                                   (8, "gauss1_cfg2", "single_eps", "de", 1003)])     # a whole node's geometry, ragged last shard
def test_engine_over_the_peer_to_peer_paths_equals_the_collectives(S, tmp_path, world, case, alg, prop, n):
    k = 10
    out = run_threads(S, case, alg, prop, n, k, resample=n // 4, world=world)
    ref = launch(world, str(tmp_path / "ref.npz"), engine="cpu", backend="gloo", case=case, alg=alg, prop=prop, n=n, updates=k,
                 resample=n // 4)
    check(out, ref)
    assert out[0]["counters"]["n_resampling"] >= 2 and all(o["collective_calls"] == 0 and o["active"] for o in out)
    if prop != "rw":
        # partners are READ where they live (rows actually read, a share (W - 1) / W of them remote), not gathered twice per update
        assert out[0]["comm"] < int(ref["comm_bytes"][0]) + int(ref["comm_bytes"][1])


@pytest.mark.parametrize("prop,silent_call", [("rw", 0), ("de", 1)])
def test_engine_a_silent_shard_fails_the_call_and_the_state_is_restored(S, tmp_path, prop, silent_call):
    """No collectives underneath: both shards' calls return SABC_ERR_COMM within a few bounds, the error contract holds, a
    fresh set-up repeats the call to exactly the uninterrupted run."""
    case, n, k = "gauss1_cfg2", 800, 6
    out = run_threads(S, case, "single_eps", prop, n, k, resample=n // 4, world=2, calls=2, timeout_ms=200.0, silence=(1, silent_call, 4))
    for o in out:
        failed = [e for e in o["errors"] if e is not None]
        assert len(failed) == 1 and failed[0].code == -22 and o["errors"][silent_call] is not None
        assert o["seconds"][0] < 10.0
    ref = launch(2, str(tmp_path / "ref.npz"), engine="cpu", backend="gloo", case=case, alg="single_eps", prop=prop, n=n, updates=2 * k,
                 resample=n // 4)
    theta = np.concatenate([o["theta"] for o in out], 1)
    assert [out[0]["counters"][q] for q in ("n_simulation", "n_accept", "n_resampling", "n_population_updates")] == list(ref["counters"])
    np.testing.assert_allclose(theta, ref["theta"], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("prop,where", [("rw", "update"), ("de", "update"), ("rw", "initialize")])
def test_engine_falls_back_to_the_collectives_underneath(S, tmp_path, prop, where):
    """With collectives installed underneath the caller never sees the failure: the engine puts the particles back (or, in
    sabc_initialize, starts over) and finishes the call over them."""
    case, n, k = "gauss2_2stats", 900, 8
    under = ThreadCollectives(2)
    silence = (1, 0, 5) if where == "update" else None
    if where == "initialize":
        # sabc_initialize's third post: armed before the handle initialises (run_threads arms `silence` per update call only)
        from tests import cpu_engine
        orig = cpu_engine.handle_class

        def patched():
            H = orig()

            class Armed(H):
                def initialize(self, n_simulation):
                    if self.cfg.rank == 1:
                        self.p2p_inject_silence(-2)
                    return super().initialize(n_simulation)
            return Armed
        cpu_engine.handle_class = patched
        try:
            out = run_threads(S, case, "multi_eps", prop, n, k, resample=n // 4, world=2, timeout_ms=200.0, underneath=under)
        finally:
            cpu_engine.handle_class = orig
    else:
        out = run_threads(S, case, "multi_eps", prop, n, k, resample=n // 4, world=2, timeout_ms=200.0, silence=silence, underneath=under)
    assert all(o["errors"] == [None] and o["fallbacks"] == 1 and not o["active"] and o["collective_calls"] > 0 for o in out), \
        [(o["errors"], o["fallbacks"], o["active"]) for o in out]
    ref = launch(2, str(tmp_path / "ref.npz"), engine="cpu", backend="gloo", case=case, alg="multi_eps", prop=prop, n=n, updates=k,
                 resample=n // 4)
    check(out, ref)


# ------------------------------------------------------------------------------------------------------------------
# The transport's life cycle (csrc/p2p.hpp "LEAVES"; include/sabc_hip.h "LEAVING"): a shard that goes away -- destroyed,
# or switched back to the collectives -- while its peer is inside a call or between two; set-up generations and buffer
# parities after a failed call; the set-up sequence with the shards agreeing inside the library.
# ------------------------------------------------------------------------------------------------------------------
def make_handle(S, case, alg, n, rank, world, underneath=None):
    from tests import cpu_engine
    from tests.cases import SEED, hip_model_prior
    model, prior = hip_model_prior(S, case)
    h = cpu_engine.handle_class()(n_particles=n, model=model, prior=prior, seed=SEED, rank=rank, world=world,
                                  algorithm=S._lib.ALG_MULTI_EPS if alg == "multi_eps" else S._lib.ALG_SINGLE_EPS)
    if underneath is not None:
        ar, ag = underneath.hooks(rank)
        h.set_collectives(ar, ag, False)
    return h


def run_shard_threads(world, body):
    """body(rank, barrier) on one host thread per shard; returns what each returned."""
    out, err = [None] * world, [None] * world
    barrier = threading.Barrier(world)

    def shard(rank):
        try:
            out[rank] = body(rank, barrier)
        except BaseException as e:
            err[rank] = e
            barrier.abort()

    ts = [threading.Thread(target=shard, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert all(e is None for e in err), err
    return out


def manual_setup(h, descs, rank, barrier, timeout_ms):
    descs[rank] = h.p2p_descriptor()
    barrier.wait()
    h.p2p_set_timeout(timeout_ms)
    h.p2p_init(list(descs))
    barrier.wait()
    h.p2p_selftest()
    assert h.p2p_active


@pytest.mark.parametrize("prop", ["rw", "de"])
def test_engine_a_shard_destroyed_while_its_peer_is_inside_a_call(S, prop):
    """Shard 1 is destroyed while shard 0 is in sabc_update, waiting for its row.  Shard 0's wait ends at once (the leave
    word, not the 20 s bound) with SABC_ERR_COMM 'has left the group', its state is as the error contract says, it leaves the
    group itself -- which is what lets shard 1's sabc_destroy return with everything freed, nothing parked --, and it can go
    on: the particles restored, the same handle works again once it is alone... as a fresh single-shard handle does."""
    from tests.cases import MODELS, hip_proposal
    case, n, k, bound_ms = "gauss1_cfg2", 600, 6, 20_000.0
    d = len(MODELS[case]["prior"])
    descs = [None, None]

    def body(rank, barrier):
        h = make_handle(S, case, "single_eps", n, rank, 2)
        manual_setup(h, descs, rank, barrier, bound_ms)
        h.initialize((k + 1) * n)
        parked0 = h.p2p_parked_bytes()
        barrier.wait()
        if rank == 1:
            time.sleep(0.3)                                  # shard 0 is inside its call by now
            t0 = time.perf_counter()
            h.close()                                        # leaves, waits for shard 0's `released`, frees
            return dict(close_seconds=time.perf_counter() - t0, parked=h.p2p_parked_bytes() - parked0)
        before = (dict(h.counters), h.eps.copy(), [a.copy() for a in h.get_population()])
        t0 = time.perf_counter()
        with pytest.raises(S.SABCError, match="has left the group") as ei:
            h.update(n_simulation=k * n, proposal=hip_proposal(S, prop, d), resample=n // 4)
        seconds = time.perf_counter() - t0
        assert ei.value.code == -22 and not h.p2p_active
        assert dict(h.counters) == before[0]
        np.testing.assert_array_equal(h.eps, before[1])
        with pytest.raises(S.SABCError, match="half-updated"):
            h.update(n_simulation=n, proposal=hip_proposal(S, prop, d))
        h.set_population(*before[2])
        h.close()
        return dict(seconds=seconds)

    out = run_shard_threads(2, body)
    assert out[0]["seconds"] < 5.0, out                      # not the bound
    assert out[1]["close_seconds"] < 5.0 and out[1]["parked"] == 0, out


def test_engine_a_destroyed_shard_whose_peer_never_answers_is_parked(S):
    """Shard 0 sits idle (no call in which it could notice): shard 1's sabc_destroy waits its bound, then PARKS what shard 0
    could still read instead of freeing it; with a wait of 0 -- what a finalizer asks for -- it does not wait at all.  Shard 0
    finds out at the entry of its next call, before it touches anything."""
    from tests.cases import MODELS, hip_proposal
    case, n, k = "gauss1_cfg2", 400, 3
    d = len(MODELS[case]["prior"])
    descs = [None, None]

    def body(rank, barrier):
        h = make_handle(S, case, "single_eps", n, rank, 2)
        manual_setup(h, descs, rank, barrier, 400.0)
        h.initialize((k + 1) * n)
        parked0 = h.p2p_parked_bytes()
        barrier.wait()
        if rank == 1:
            h.p2p_set_destroy_wait(0.0)
            t0 = time.perf_counter()
            h.close()
            res = dict(close_seconds=time.perf_counter() - t0, parked=h.p2p_parked_bytes() - parked0)
            barrier.wait()
            return res
        barrier.wait()                                       # shard 1 is gone
        t0 = time.perf_counter()
        with pytest.raises(S.SABCError, match="has left the peer-to-peer group"):
            h.update(n_simulation=k * n, proposal=hip_proposal(S, "de", d), resample=n // 4)
        seconds = time.perf_counter() - t0
        assert not h.p2p_active
        h.update(n_simulation=0, proposal=hip_proposal(S, "de", d))     # the handle was left untouched: nothing had been launched
        h.close()
        return dict(seconds=seconds)

    out = run_shard_threads(2, body)
    assert out[1]["close_seconds"] < 0.3 and out[1]["parked"] == 1, out
    assert out[0]["seconds"] < 0.3, out


@pytest.mark.parametrize("when", ["between_calls", "mid_call"])
def test_engine_a_shard_that_switches_back_takes_its_peer_along(S, tmp_path, when):
    """sabc_comm_p2p_disable on ONE shard, collectives installed underneath.  Between two calls: the peer sees it at the entry
    of its next call and both run over the collectives (no failed attempt).  While the peer is already inside a call: the
    peer's wait ends at the leave word, the engine puts its particles back and repeats the call over the collectives.  Either
    way the caller sees successful calls and the run is the reference run."""
    from tests.cases import MODELS, hip_proposal
    case, alg, prop, n, k = "gauss2_2stats", "multi_eps", "de", 900, 5
    d = len(MODELS[case]["prior"])
    under = ThreadCollectives(2)
    descs = [None, None]

    def body(rank, barrier):
        try:
            h = make_handle(S, case, alg, n, rank, 2, underneath=under)
            manual_setup(h, descs, rank, barrier, 20_000.0)
            h.initialize((2 * k + 1) * n)
            h.update(n_simulation=k * n, proposal=hip_proposal(S, prop, d), resample=n // 4)
            assert h.p2p_active and h.collective_calls == 0
            barrier.wait()
            if rank == 1:
                if when == "mid_call":
                    time.sleep(0.3)
                h.p2p_disable()
            if when == "between_calls":
                barrier.wait()
            t0 = time.perf_counter()
            h.update(n_simulation=k * n, proposal=hip_proposal(S, prop, d), resample=n // 4)
            th, u, rho = h.get_population()
            res = dict(theta=th, u=u, rho=rho, eps=h.eps, counters=h.counters, hist=h.history, fallbacks=h.p2p_fallbacks,
                       active=h.p2p_active, collective_calls=h.collective_calls, seconds=time.perf_counter() - t0)
            barrier.wait()
            h.close()
            return res
        except BaseException:
            under.barrier.abort()
            raise

    out = run_shard_threads(2, body)
    assert all(not o["active"] and o["collective_calls"] > 0 and o["seconds"] < 10.0 for o in out), out
    assert [o["fallbacks"] for o in out] == ([0, 0] if when == "between_calls" else [1, 0]), [o["fallbacks"] for o in out]
    ref = launch(2, str(tmp_path / "ref.npz"), engine="cpu", backend="gloo", case=case, alg=alg, prop=prop, n=n, updates=2 * k,
                 resample=n // 4)
    check(out, ref)


@pytest.mark.parametrize("prop,posts_before", [(p, q) for p in ("rw", "de") for q in (1, 2, 4, 5, 8)])
def test_engine_set_up_again_after_a_failed_call_wherever_it_failed(S, tmp_path, prop, posts_before):
    """A shard whose post is lost on the wire (test hook sabc_comm_p2p_inject_loss) still RECEIVES its peer's row: when that step fires the resample it goes on to flip its
    population buffers while the peer, which timed out one exchange earlier, does not -- after the failed call the shards
    stand on different buffer parities.  The next set-up's descriptors carry each owner's parity and its generation, so the
    repeated call reads the right buffers (partners, resampled rows) and matches no word of the old generation: the run is the
    uninterrupted run wherever the silence falls."""
    case, n, k = "gauss1_cfg2", 800, 6
    out = run_threads(S, case, "single_eps", prop, n, k, resample=n // 4, world=2, calls=2, timeout_ms=150.0,
                      silence=(1, 0, posts_before, "loss"))
    for o in out:
        assert [e is not None for e in o["errors"]] == [True, False]
    ref = launch(2, str(tmp_path / "ref.npz"), engine="cpu", backend="gloo", case=case, alg="single_eps", prop=prop, n=n, updates=2 * k,
                 resample=n // 4)
    check(out, ref, tol=1e-9)


@pytest.mark.parametrize("trouble", ["none", "export", "map", "stale", "silent"])
def test_engine_set_up_in_one_call_with_the_shards_agreeing(S, tmp_path, trouble):
    """sabc_comm_p2p_setup (csrc/p2p_setup.hpp -- the product's own sequence, here over the CPU backend): descriptors over the
    collectives -> map -> agreement -> self-test (slots, then patterns read through the mappings) -> agreement.  One shard in
    trouble -- it cannot export, cannot map, reads a stale line, or never posts -- leaves EVERY shard on the collectives with
    the same answer; only the silent shard costs its peers a bound (inside the self-test, where they wait for it), nobody
    finds out in its first exchange.  The run that follows is the reference run."""
    from tests.cases import MODELS, hip_proposal
    case, alg, prop, n, k = "gauss2_2stats", "multi_eps", "de", 700, 5
    d = len(MODELS[case]["prior"])
    under = ThreadCollectives(2)
    bound_ms = 300.0 if trouble == "silent" else 20_000.0

    def body(rank, barrier):
        try:
            h = make_handle(S, case, alg, n, rank, 2, underneath=under)
            h.p2p_set_timeout(bound_ms)
            if rank == 1 and trouble in ("export", "map"):
                h._L.sabc_test_p2p_inject_setup_failure(h._h, 1 if trouble == "export" else 2)
            if rank == 1 and trouble == "stale":
                h.p2p_inject_stale(1)
            if rank == 1 and trouble == "silent":
                h.p2p_inject_silence(1)
            t0 = time.perf_counter()
            ok = h.p2p_setup()
            seconds = time.perf_counter() - t0
            assert ok == h.p2p_active
            calls0 = h.collective_calls
            h.initialize((k + 1) * n)
            h.update(n_simulation=k * n, proposal=hip_proposal(S, prop, d), resample=n // 4)
            th, u, rho = h.get_population()
            res = dict(ok=ok, note=h.p2p_setup_note, seconds=seconds, theta=th, u=u, rho=rho, eps=h.eps, counters=h.counters,
                       hist=h.history, collective_calls=h.collective_calls - calls0, fallbacks=h.p2p_fallbacks)
            barrier.wait()
            parked0 = h.p2p_parked_bytes()
            t0 = time.perf_counter()
            h.close()
            # a set-up that ended with every shard on the collectives leaves nothing to acknowledge: no wait, nothing parked
            res.update(close_seconds=time.perf_counter() - t0, parked=h.p2p_parked_bytes() - parked0)
            return res
        except BaseException:
            under.barrier.abort()
            raise

    out = run_shard_threads(2, body)
    assert [o["ok"] for o in out] == [trouble == "none"] * 2, [(o["ok"], o["note"]) for o in out]
    assert all(o["fallbacks"] == 0 for o in out)
    if trouble == "none":
        assert all(o["collective_calls"] == 0 for o in out)
    else:
        assert all(o["collective_calls"] > k and "stays on the collectives" in o["note"] for o in out), [o["note"] for o in out]
        assert all(o["close_seconds"] < 1.0 for o in out), [o["close_seconds"] for o in out]
        assert sum(o["parked"] for o in out) == 0, [o["parked"] for o in out]
        assert all(o["seconds"] < (5.0 if trouble != "silent" else 10 * bound_ms * 1e-3 + 2.0) for o in out), [o["seconds"] for o in out]
    ref = launch(2, str(tmp_path / "ref.npz"), engine="cpu", backend="gloo", case=case, alg=alg, prop=prop, n=n, updates=k, resample=n // 4)
    check(out, ref)
