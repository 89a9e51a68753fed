#!/bin/bash
# full GPU suite + the four benches (run after every kernel change that is kept)
set -x
out=gpurun_out/r02_full; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=5 > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $out/pytest.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
for cfg in cfg3 cfg4 cfg5; do
timeout -k 10 200 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_$cfg.json 2> $out/bench_$cfg.err; echo "bench rc=$?"
done
python - <<'PY'
import json
for f in ["default", "cfg3", "cfg4", "cfg5"]:
    try:
        j = json.loads(open(f"gpurun_out/r02_full/bench_{f}.json").read().strip().splitlines()[-1])
        print(f, "%.3e" % j["value"], "%.1f" % (j["ms_per_step"] * 1e3), "%.1f" % j["roofline"]["avg_launch_us"])
    except Exception as e:
        print(f, e)
PY
