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
