#!/bin/bash
set -x
out=gpurun_out/r02j; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=5 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -12 $out/pytest.log
for i in 1 2 3; do timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_default_$i.json 2> $out/bench_default_$i.err; echo "bench rc=$?"; done
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err; echo "bench rc=$?"
root=$(pwd); export TMPDIR=/tmp
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_cfg2 -- python3 $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $root/$out/prof_cfg2.json 2> $root/$out/prof_cfg2.err); echo "prof rc=$?"
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"
