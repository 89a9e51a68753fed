"""What ONE sabc_update call costs beyond its population updates (a wrapper's progress output cuts a run into 50 calls or more):
the same 2000 updates as one call, as calls of 20, and as calls of 1.  usage (GPU box): python tools/call_overhead.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sabc_amd as S
from tests.cases import hip_model_prior, hip_proposal

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
model, prior = hip_model_prior(S, "gauss1_cfg2")
for per_call in (2000, 400, 100, 20, 5, 1):
    h = S.SabcHandle(n_particles=n, model=model, prior=prior, seed=7); h.initialize(n)
    h.update(n_simulation=50 * n, proposal=hip_proposal(S, "rw", 1), resample=10 ** 12)
    calls = 2000 // per_call
    l0, s0 = h.kernel_launches, h.host_syncs
    t0 = time.perf_counter()
    done = 50
    for c in range(calls):
        h.update(n_simulation=per_call * n, proposal=hip_proposal(S, "rw", 1), resample=10 ** 12, history_phase=done, more_chunks_follow=c + 1 < calls)
        done += per_call
    dt = time.perf_counter() - t0
    print(f"n {n}: {calls} calls of {per_call} updates: {dt * 1e3:.2f} ms, {dt / 2000 * 1e6:.2f} us per update, {(h.kernel_launches - l0) / calls:.1f} launches and "
          f"{(h.host_syncs - s0) / calls:.1f} host waits per call", flush=True)
    h.close()
