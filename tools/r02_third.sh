#!/bin/bash
set -x
out=gpurun_out/r02c; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=12 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -30 $out/pytest.log
timeout -k 10 900 bash tools/exp_ab2.sh "cfg3 cfg4 cfg2" 2 > $out/ab.log 2>&1; tail -30 $out/ab.log
root=$(pwd); export TMPDIR=/tmp
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_cfg2 -- python3 $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $root/$out/prof_cfg2.json 2> $root/$out/prof_cfg2.err); echo "prof rc=$?"
