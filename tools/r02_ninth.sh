#!/bin/bash
set -x
out=gpurun_out/r02i; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=6 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -14 $out/pytest.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
timeout -k 10 900 bash tools/profile_round.sh r02 > $out/profile_round.log 2>&1; echo "profile rc=$?"
