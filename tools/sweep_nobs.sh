#!/bin/bash
# kernel time of cfg2 as a function of the draws per simulation: splits k_update into the
# simulation loop and everything else (proposal, ECDF search, accept, reductions)
mkdir -p gpurun_out
for k in 2 50 100 200; do
  python bench.py --n-obs $k --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null > gpurun_out/sweep_$k.json
  python - <<PY
import json; d=json.load(open("gpurun_out/sweep_$k.json")); print("n_obs", $k, "kernel %.1f us" % d["roofline"]["avg_launch_us"], "%.3f ms/step" % d["ms_per_step"])
PY
done
