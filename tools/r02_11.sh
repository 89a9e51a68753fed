#!/bin/bash
set -x
out=gpurun_out/r02k; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_edges.py -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -4 $out/pytest.log
timeout -k 10 600 bash tools/exp_ab2.sh "cfg2" 4 > $out/ab.log 2>&1; tail -6 $out/ab.log
for i in 1 2; do timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_default_$i.json 2> $out/bench_default_$i.err; echo "bench rc=$?"; done
