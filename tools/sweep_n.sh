#!/bin/bash
# step time versus particles on one GPU: the small sizes are a proxy for one shard at N = 8
mkdir -p gpurun_out
for n in 62500 125000 250000 500000 1000000 4000000; do
  python bench.py --n-particles $n --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null > gpurun_out/sweepn_$n.json
  python - <<PY
import json; d=json.load(open("gpurun_out/sweepn_$n.json")); print("n", $n, "kernel %.1f us" % d["roofline"]["avg_launch_us"], "%.1f us/step" % (d["ms_per_step"]*1e3), "%.3e sims/s" % d["value"], "syncs", d["host_syncs_in_timed_region"], "resamples", d["resamples_in_timed_region"])
PY
done
