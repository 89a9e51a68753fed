#!/bin/bash
# round 2, call 10: line-wise resample search -- parity, then bench
set -x
out=gpurun_out/r02j; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_edges.py tests/test_gpu_parity.py tests/test_gpu_host_fdist.py tests/test_user_simulator.py -m gpu -q -x > $out/pytest_a.log 2>&1; rc=$?; echo "rc=$rc"; tail -15 $out/pytest_a.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x > $out/pytest_b.log 2>&1; rc=$?; echo "rc=$rc"; tail -8 $out/pytest_b.log
[ $rc -eq 0 ] || exit 1
for i in 1 2; do
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --all-kernel-events > $out/bench_all_$i.json 2> $out/bench_all_$i.err; echo "bench rc=$?"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_default_$i.json 2> $out/bench_default_$i.err; echo "bench rc=$?"
done
timeout -k 10 200 python bench.py --config cfg4 --steps 30 --warmup 5 --no-cpu-baseline --all-kernel-events > $out/bench_cfg4.json 2> $out/bench_cfg4.err; echo "bench rc=$?"
