#!/bin/bash
# k_update time around whole numbers of waves per SIMD (n = w * 65536 fills every SIMD of an MI355X with exactly w waves)
for n in 917504 983040 1000000 1048576 1114112 1966080 2097152; do
  python bench.py --n-particles $n --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/q_$n.json
  python - <<PY
import json; d=json.load(open("gpurun_out/q_$n.json")); print("n", $n, "waves/SIMD %.2f" % ($n/65536), "kernel %.1f us" % d["roofline"]["avg_launch_us"], "-> %.1f us per 1e6" % (d["roofline"]["avg_launch_us"]*1e6/$n))
PY
done
