#!/bin/bash
# round 2, call 11: g-and-k sorts the normals -- parity, then bench
set -x
out=gpurun_out/r02k; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -q -x -k "gk or simulator or ragged" > $out/pytest_a.log 2>&1; rc=$?; echo "rc=$rc"; tail -15 $out/pytest_a.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=5 > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $out/pytest.log
for i in 1 2; do
timeout -k 10 200 python bench.py --config cfg4 --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_cfg4_$i.json 2> $out/bench_cfg4_$i.err; echo "bench rc=$?"
done
timeout -k 10 200 python bench.py --config cfg4 --steps 30 --warmup 5 --no-cpu-baseline --proposal de > $out/bench_cfg4_de.json 2> $out/bench_cfg4_de.err; echo "bench rc=$?"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
