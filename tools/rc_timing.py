"""Where a reduce-and-control launch spends its time (library variant built with -DSABC_RC_TIMING: tools/build_variants.sh
rctiming="-DSABC_RC_TIMING"; copied over simulatedannealingabc.jl_amd/libsabc_hip.so for the run: tools/rc_timing.sh)."""
import ctypes as C
import sys

import numpy as np

import sabc_amd as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
k = 200
dbg = S._lib.lib().sabc_debug_rc_ticks
out = (C.c_ulonglong * 16)()
h = S.SabcHandle(n_particles=n, model=S.GaussianIID(n_obs=100, sd=1.0, obs_mean=1.6), prior=S.Normal(0, 2), seed=7)
h.initialize((k + 1) * n)
assert dbg(out, 1) == 0
h.update(n_simulation=k * n, proposal=S.RandomWalk(n_para=1))
assert dbg(out, 0) == 0
t = np.array(list(out), dtype=float)
order = [(7, "thread 0's control word arrived"), (8, "thread 0's partial rows arrived"), (1, "barrier: all waves' loads"), (2, "column sums"),
         (3, "exchange"), (9, "control: sums taken over"), (10, "control: proposal"), (11, "control: epsilon"), (12, "control: history"),
         (4, "control: pivot + barrier"), (5, "write back issued"), (6, "mailbox")]
print(f"n = {n}: {int(t[0])} launches")
for i, name in order:
    print(f"  {name:34s} {t[i] / t[0] * 10:8.1f} ns per launch")
print(f"  {'sum':34s} {t[1:].sum() / t[0] * 10:8.1f} ns")
