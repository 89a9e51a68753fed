#!/bin/bash
set -x
out=gpurun_out/r02d; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=8 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -15 $out/pytest.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
timeout -k 10 900 bash tools/profile_round.sh r02 > $out/profile_round.log 2>&1; echo "profile rc=$?"
timeout -k 10 300 bash tools/pmc_sq.sh cfg4 > $out/pmc_cfg4.log 2>&1; echo "pmc rc=$?"
timeout -k 10 300 bash tools/pmc_sq.sh cfg2 > $out/pmc_cfg2.log 2>&1; echo "pmc rc=$?"
