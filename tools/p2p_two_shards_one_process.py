#!/usr/bin/env python3
"""Two (or more) shards of one population in ONE process over the peer-to-peer transport -- a host thread and a stream per
shard, no collectives installed -- timed like bench.py's step; prints one JSON line (particle-simulations/s, us per update,
kernel launches per update).

NOT under rocprofv3: its kernel tracing serialises the dispatches of the process' queues, and an exchange kernel that waits for
a peer whose kernel cannot start until the waiter has finished runs into its bound (tried: shard 1 gave up at exchange 4,
shard 0 -- which then found shard 1's row -- at exchange 5; SABC_ERR_COMM on both, nothing hung).  One process per GPU, the
deployment form, has no such coupling.  The per-launch times of the exchange come from HIP events instead
(`bench.py --gpus N`: `exchange.reduce_control_us`)."""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--n-particles", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--proposal", default="randomwalk", choices=["randomwalk", "de", "stretch"])
    a = ap.parse_args()
    import numpy as np
    import torch
    import sabc_amd as S
    n, W, K = a.n_particles, a.world, a.steps
    ybar = float(np.random.default_rng(20241220).normal(1.5, 1.0, 100).mean())
    proposal = {"randomwalk": S.RandomWalk(n_para=1), "de": S.DifferentialEvolution(n_para=1), "stretch": S.StretchMove()}[a.proposal]
    descs, out, err = [None] * W, [None] * W, [None] * W
    barrier = threading.Barrier(W)

    def shard(rank):
        try:
            torch.cuda.set_device(0)
            h = S.SabcHandle(n_particles=n, model=S.GaussianIID(n_obs=100, sd=1.0, obs_mean=ybar), prior=S.Normal(0.0, 2.0),
                             seed=20241220, rank=rank, world=W)
            descs[rank] = h.p2p_descriptor()
            barrier.wait()
            h.p2p_init(list(descs))
            barrier.wait()
            h.p2p_selftest()
            h.initialize(n)
            if a.warmup:
                h.update(n_simulation=a.warmup * n, proposal=proposal)
            torch.cuda.synchronize()
            barrier.wait()
            l0, t0 = h.kernel_launches, time.perf_counter()
            h.update(n_simulation=K * n, proposal=proposal)
            torch.cuda.synchronize()
            barrier.wait()
            dt = time.perf_counter() - t0
            out[rank] = dict(dt=dt, launches=(h.kernel_launches - l0) / K, collective_calls=h.collective_calls, counters=h.counters,
                             eps=h.eps.tolist())
            barrier.wait()
            h.close()
        except BaseException as e:
            err[rank] = e
            barrier.abort()

    ts = [threading.Thread(target=shard, args=(r,)) for r in range(W)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if any(err):
        raise SystemExit(f"a shard failed: {err}")
    dt = max(o["dt"] for o in out)
    print(json.dumps({"metric": "particle-simulations/sec, %d shards in one process over the peer-to-peer transport" % W, "value": K * n / dt,
                      "us_per_update": dt / K * 1e6, "n_particles": n, "proposal": a.proposal, "steps": K, "kernel_launches_per_update": out[0]["launches"],
                      "collective_calls": out[0]["collective_calls"], "n_accept": out[0]["counters"]["n_accept"],
                      "n_resampling": out[0]["counters"]["n_resampling"], "eps": out[0]["eps"]}), flush=True)


if __name__ == "__main__":
    main()
