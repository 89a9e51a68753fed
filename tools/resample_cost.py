"""What a resample costs at small n: a threshold that fires after EVERY update against one that never fires.
usage (GPU box): python tools/resample_cost.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sabc_amd as S
from tests.cases import hip_model_prior, hip_proposal

model, prior = hip_model_prior(S, "gauss1_cfg2")
for n in [int(x) for x in sys.argv[1:]] or [1000, 10000, 100000]:
    out = {}
    for label, thr in (("never", 10 ** 12), ("every update", 1)):
        h = S.SabcHandle(n_particles=n, model=model, prior=prior, seed=7); h.initialize(n)
        h.update(n_simulation=20 * n, proposal=hip_proposal(S, "rw", 1), resample=thr)
        l0, s0 = h.kernel_launches, h.host_syncs
        t0 = time.perf_counter(); h.update(n_simulation=300 * n, proposal=hip_proposal(S, "rw", 1), resample=thr); dt = time.perf_counter() - t0
        out[label] = (dt / 300 * 1e6, (h.kernel_launches - l0) / 300, (h.host_syncs - s0) / 300, h.counters["n_resampling"])
        h.close()
    a, b = out["never"], out["every update"]
    print(f"n {n}: update {a[0]:.1f} us; update + resample {b[0]:.1f} us ({b[1]:.1f} launches, {b[2]:.1f} host syncs per update) -> a resample costs {b[0] - a[0]:.1f} us", flush=True)
