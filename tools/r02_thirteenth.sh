#!/bin/bash
# round 2, call 13: g-and-k with 64 particles per wave -- full suite, benches
set -x
out=gpurun_out/r02m; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=5 > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $out/pytest.log
timeout -k 10 200 python bench.py --config cfg4 --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_cfg4.json 2> $out/bench_cfg4.err; echo "bench rc=$?"
timeout -k 10 200 python bench.py --config cfg4 --steps 30 --warmup 5 --no-cpu-baseline --proposal de > $out/bench_cfg4_de.json 2> $out/bench_cfg4_de.err; echo "bench rc=$?"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
