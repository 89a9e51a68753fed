#!/bin/bash
set -x
out=gpurun_out/r02g; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_distributed.py -m gpu -q -k "in_flight" > $out/pytest_inflight.log 2>&1; echo "rc=$?" >> $out/pytest_inflight.log
tail -30 $out/pytest_inflight.log
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=6 --deselect tests/test_distributed.py::test_device_collectives_in_flight_two_shards_one_gpu > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -12 $out/pytest.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
timeout -k 10 200 python bench.py --config cfg4 --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_cfg4.json 2> $out/bench_cfg4.err; echo "bench rc=$?"
