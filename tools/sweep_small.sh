#!/bin/bash
# step time at the population sizes the reference's documentation works with, launch chain (SABC_PERSISTENT=0) against the
# updates of a call in one launch (kernels.hip: k_update_persistent); usage: tools/sweep_small.sh [proposal]
# SWEEP_MODES: 0 launch chain | 1, 4, 16 one launch with that many lanes per particle (where they fit) | a: one launch, the default choice
prop=${1:-randomwalk}
mkdir -p gpurun_out
for n in ${SWEEP_N:-100 1000 5000 10000 62500}; do
  for mode in ${SWEEP_MODES:-0 1 4 16 a}; do
    if [ $mode = a ]; then lanes_env=; persist=1; else lanes_env=$mode; persist=$((mode > 0)); fi
    SABC_PERSISTENT=$persist SABC_PERSISTENT_LANES=$lanes_env SABC_PERSISTENT_MAX=65536 python bench.py --n-particles $n --proposal $prop --steps 500 --warmup 20 --no-cpu-baseline --no-kernel-events --repeats 3 2>/dev/null > gpurun_out/small_${prop}_${n}_$mode.json
    python - <<PY
import json; d=json.loads(open("gpurun_out/small_${prop}_${n}_$mode.json").read().strip().splitlines()[-1]); print("$prop n", $n, {"0": "chain         ", "1": "one launch x1 ", "4": "one launch x4 ", "16": "one launch x16", "a": "one launch    "}["$mode"], "%.2f us/update" % (d["ms_per_step"]*1e3), "%.3e sims/s" % d["value"], "launches/update %.2f" % d["launches_per_update"], "lanes", d.get("persistent_lanes"), "resamples", d["resamples_in_timed_region"], "n_accept", d["state"]["n_accept"])
PY
  done
done
