#!/bin/bash
# round 2, call 8: MvNormal prior on the device + full suite + bench
set -x
out=gpurun_out/r02h; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_priors.py -m gpu -q > $out/pytest_priors.log 2>&1; echo "rc=$?"; tail -15 $out/pytest_priors.log
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=6 --deselect tests/test_priors.py > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $out/pytest.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
timeout -k 10 200 python bench.py --config cfg3 --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_cfg3.json 2> $out/bench_cfg3.err; echo "bench rc=$?"
