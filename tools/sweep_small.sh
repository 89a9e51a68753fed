#!/bin/bash
# step time at the population sizes the reference's documentation works with, launch chain (SABC_PERSISTENT=0) against the
# updates of a call in one launch (kernels.hip: k_update_persistent); usage: tools/sweep_small.sh [proposal]
prop=${1:-randomwalk}
mkdir -p gpurun_out
for n in ${SWEEP_N:-100 1000 5000 10000 62500}; do
  for mode in ${SWEEP_MODES:-0 1 4 16}; do      # 0: launch chain | 1: one launch, a lane per particle | 4: one launch, a quad per particle where it fits
    SABC_PERSISTENT=$((mode > 0)) SABC_PERSISTENT_LANES=$mode SABC_PERSISTENT_MAX=65536 python bench.py --n-particles $n --proposal $prop --steps 500 --warmup 20 --no-cpu-baseline --no-kernel-events --repeats 3 2>/dev/null > gpurun_out/small_${prop}_${n}_$mode.json
    python - <<PY
import json; d=json.loads(open("gpurun_out/small_${prop}_${n}_$mode.json").read().strip().splitlines()[-1]); print("$prop n", $n, {0: "chain        ", 1: "persistent x1 ", 4: "persistent x4 ", 16: "persistent x16"}[$mode], "%.2f us/update" % (d["ms_per_step"]*1e3), "%.3e sims/s" % d["value"], "launches/update %.2f" % d["launches_per_update"], "resamples", d["resamples_in_timed_region"], "n_accept", d["state"]["n_accept"])
PY
  done
done
