#!/bin/bash
set -x
out=gpurun_out/r02e; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_golden.py tests/test_gpu_edges.py tests/test_user_simulator.py tests/test_gpu_host_fdist.py -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -6 $out/pytest.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
root=$(pwd); export TMPDIR=/tmp
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_cfg2 -- python3 $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $root/$out/prof_cfg2.json 2> $root/$out/prof_cfg2.err); echo "prof rc=$?"
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_cfg4 -- python3 $root/bench.py --config cfg4 --steps 30 --warmup 5 --no-cpu-baseline > $root/$out/prof_cfg4.json 2> $root/$out/prof_cfg4.err); echo "prof rc=$?"
