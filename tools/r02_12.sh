#!/bin/bash
set -x
out=gpurun_out/r02l; mkdir -p $out
for n in 62500 125000 250000 500000 1000000 4000000; do
  timeout -k 10 120 python bench.py --n-particles $n --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/sweep_$n.json
  timeout -k 10 120 python bench.py --n-particles $n --steps 50 --warmup 5 --no-cpu-baseline --proposal de 2>/dev/null | tail -1 > $out/sweep_de_$n.json
done
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_fullsize.py -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -4 $out/pytest.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02l/sweep_*.json"), key=lambda p: (("de" in p), int(p.split("_")[-1][:-5]))):
    j = json.load(open(f))
    print(f.split("/")[-1], "ms/step %.4f kernel %.1f value %.3e" % (j["ms_per_step"], j["roofline"]["avg_launch_us"], j["value"]))
PY
