#!/bin/bash
# per-call fixed cost of sabc_update() versus per-update cost: same workload, more updates per call
for k in 10 50 200; do
  python bench.py --steps $k --warmup 5 --no-cpu-baseline 2>/dev/null > gpurun_out/sweeps_$k.json
  python - <<PY
import json; d=json.load(open("gpurun_out/sweeps_$k.json")); print("steps", $k, "kernel %.1f us" % d["roofline"]["avg_launch_us"], "%.1f us/step" % (d["ms_per_step"]*1e3), "%.3e sims/s" % d["value"], "n_resampling", d["state"]["n_resampling"])
PY
done
