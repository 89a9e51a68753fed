#!/usr/bin/env python3
"""One rank of a sharded SABC run (launched by tests/test_distributed.py, one process per rank).

  python tests/dist_worker.py --engine cpu|hip --backend gloo|nccl --case NAME --alg A --prop P
                              --n N --updates K --out FILE

engine=cpu : the product's host engine over the oracle-backed Backend (tests/cpu_engine) -- no GPU.
engine=hip : libsabc_hip.so; with --backend gloo all ranks may share one GPU (host-staged hooks).
Rank 0 writes the gathered global population, counters, eps and histories to FILE (npz)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--engine", default="cpu")
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--case", default="gauss1_cfg2")
    ap.add_argument("--alg", default="single_eps")
    ap.add_argument("--prop", default="rw")
    ap.add_argument("--n", type=int, default=1001)
    ap.add_argument("--updates", type=int, default=10)
    ap.add_argument("--resample", type=float, default=0.0)
    ap.add_argument("--alltoallv", type=int, default=1, help="0: no personalised exchange (the resample allgathers the population)")
    ap.add_argument("--p2p", type=int, default=0, help="1: the peer-to-peer transport on top of the collectives (hip engine only)")
    ap.add_argument("--silence", type=int, default=0, help="p2p test hook: rank 1 lets this many posts go out, then skips one")
    ap.add_argument("--silence-init", type=int, default=0, help="the same inside sabc_initialize")
    ap.add_argument("--silence-selftest", type=int, default=0, help="rank 1 posts nothing in the transport's self-test (first contact fails)")
    ap.add_argument("--stale-selftest", type=int, default=0, help="rank 1's self-test reads its peers' memory as if a stale line had been served")
    ap.add_argument("--scenario", default="", help="destroy-mid-call: rank 1 destroys its handle while rank 0 is inside sabc_update")
    ap.add_argument("--p2p-timeout-ms", type=float, default=0.0)
    ap.add_argument("--host-fdist", type=int, default=0, help="hip engine, case gauss1_small: f_dist as a HOST callable that draws the device simulator's Philox blocks")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    import sabc_amd as S
    from sabc_amd.dist import install_collectives
    from tests.cases import MODELS, SEED, hip_model_prior, hip_proposal

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = 0
    if a.engine == "hip":
        ndev = torch.cuda.device_count()
        device = local % max(ndev, 1)
        torch.cuda.set_device(device)
    dist.init_process_group(a.backend, rank=rank, world_size=world)

    if a.engine == "cpu":
        from tests import cpu_engine
        Handle = cpu_engine.handle_class()
    else:
        Handle = S.SabcHandle
    model, prior = hip_model_prior(S, a.case)
    if a.host_fdist:
        assert a.engine == "hip" and a.case == "gauss1_small"
        from oracle import oracle as O
        O.build()
        from tests.test_gpu_host_fdist import gauss_iid_keyed
        kw = MODELS[a.case]["model"][1]
        model = S.HostDistance(gauss_iid_keyed(O, kw["n_obs"], kw["obs_mean"]), n_stats=1, n_para=1, univariate=True, with_ids=True)
    d = len(MODELS[a.case]["prior"])
    alg = S._lib.ALG_MULTI_EPS if a.alg == "multi_eps" else S._lib.ALG_SINGLE_EPS
    h = Handle(n_particles=a.n, model=model, prior=prior, algorithm=alg, seed=SEED, device=device, rank=rank, world=world)
    transport = "none"
    if a.silence_selftest and rank == 1:
        h.p2p_inject_silence(a.silence_selftest)       # that many of rank 1's next posts (row, barrier flag) are skipped
    if a.stale_selftest and rank == 1:
        h.p2p_inject_stale(a.stale_selftest)
    if a.p2p_timeout_ms > 0 and a.silence_selftest:
        os.environ["SABC_P2P_TIMEOUT_MS"] = str(a.p2p_timeout_ms)
    import time
    t_setup = time.perf_counter()
    if world > 1:
        transport = install_collectives(h, device, alltoallv=bool(a.alltoallv), p2p=bool(a.p2p) if a.engine == "hip" else False)
        assert transport in ("p2p", "rccl", "hooks-nccl", "hooks-gloo"), transport
    setup_seconds = time.perf_counter() - t_setup
    calls0 = h.collective_calls if a.engine == "hip" else 0       # (the self-test of the base transport used some)
    if a.p2p_timeout_ms > 0 and transport == "p2p":
        h.p2p_set_timeout(a.p2p_timeout_ms)
    if a.silence_init and rank == 1 and transport == "p2p":
        h.p2p_inject_silence(-a.silence_init)
    h.initialize((a.updates + 1) * a.n)
    if a.scenario == "destroy-mid-call":
        return destroy_mid_call(a, h, S, dist, rank, d, transport)
    bytes_init = h.comm_bytes
    if a.silence and rank == 1 and transport == "p2p":
        h.p2p_inject_silence(-a.silence)
    h.update(n_simulation=a.updates * a.n, proposal=hip_proposal(S, a.prop, d),
             resample=a.resample if a.resample > 0 else None)
    th, u, rho = h.get_population()
    parts = [None] * world
    dist.all_gather_object(parts, (h.local_offset, th, u, rho))
    if rank == 0:
        parts.sort(key=lambda p: p[0])
        e, uh, rh = h.history
        np.savez(a.out, theta=np.concatenate([p[1] for p in parts], 1), u=np.concatenate([p[2] for p in parts], 1),
                 rho=np.concatenate([p[3] for p in parts], 1), eps=h.eps, eps_hist=e, u_hist=uh, rho_hist=rh,
                 counters=np.array([h.counters[k] for k in ("n_simulation", "n_accept", "n_resampling", "n_population_updates")]),
                 sigma=h.proposal_sigma, offsets=np.array([p[0] for p in parts]),
                 comm_bytes=np.array([bytes_init, h.comm_bytes - bytes_init]), transport=np.array(transport),
                 collective_calls=np.array((h.collective_calls - calls0) if a.engine == "hip" else -1),
                 p2p_fallbacks=np.array(h.p2p_fallbacks if a.engine == "hip" else 0),
                 p2p_active_at_end=np.array(bool(h.p2p_active) if a.engine == "hip" else False),
                 setup_seconds=np.array(setup_seconds), setup_note=np.array(getattr(h, "p2p_setup_note", "")))
    dist.barrier()
    h.close()
    dist.destroy_process_group()


def destroy_mid_call(a, h, S, dist, rank, d, transport):
    """Rank 1 destroys its handle while rank 0 is inside sabc_update (tests/test_p2p.py)."""
    import time
    from tests.cases import SEED, hip_model_prior, hip_proposal
    parked0 = h.p2p_parked_bytes()
    if rank == 0:
        # whoever takes rank 1 away takes the collectives underneath away with it: hooks that fail at once (with the gloo hooks
        # left in place the engine's fallback would sit in an allreduce rank 1 never joins)
        h.set_collectives(lambda ctx, buf, count, stream: -1, lambda ctx, send, recv, count, stream: -1, False)
    dist.barrier()
    res = {}
    if rank == 1:
        time.sleep(0.3)                                   # rank 0 is inside its call by now
        t0 = time.perf_counter()
        h.close()
        res = dict(close_seconds=time.perf_counter() - t0, parked=h.p2p_parked_bytes() - parked0)
    else:
        t0 = time.perf_counter()
        try:
            h.update(n_simulation=a.updates * a.n, proposal=hip_proposal(S, a.prop, d), resample=a.resample if a.resample > 0 else None)
            res = dict(error_code=0, error_text="")
        except S.SABCError as e:
            res = dict(error_code=e.code, error_text=str(e))
        res.update(seconds=time.perf_counter() - t0, p2p_active_at_end=bool(h.p2p_active))
        h.close()
        model, prior = hip_model_prior(S, a.case)         # rank 0 carries on alone on the same device
        alone = S.SabcHandle(n_particles=4096, model=model, prior=prior, seed=SEED, device=0)
        alone.initialize(3 * 4096)
        alone.update(n_simulation=2 * 4096, proposal=hip_proposal(S, "rw", d))
        res["alone_updates"] = alone.counters["n_population_updates"]
        alone.close()
    parts = [None] * 2
    dist.all_gather_object(parts, res)
    if rank == 0:
        merged = dict(parts[1], **parts[0])
        np.savez(a.out, transport=np.array(transport), **{k: np.array(v) for k, v in merged.items()})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
