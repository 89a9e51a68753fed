#!/bin/bash
# second GPU call of round 2: full GPU suite (anchors, user simulators, launcher), profiles and SQ counters per config
set -x
out=gpurun_out/r02b; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=15 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -25 $out/pytest.log
for cfg in cfg2 cfg3 cfg4 cfg5; do
  timeout -k 10 200 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_$cfg.json 2> $out/bench_$cfg.err; echo "bench $cfg rc=$?"
done
timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --proposal de > $out/bench_cfg2_de.json 2> $out/bench_cfg2_de.err
root=$(pwd); export TMPDIR=/tmp
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_cfg2 -- python3 $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $root/$out/prof_cfg2.json 2> $root/$out/prof_cfg2.err); echo "prof rc=$?"
for cfg in cfg2 cfg3 cfg4 cfg5; do
  timeout -k 10 400 bash tools/pmc_sq.sh $cfg > $out/pmc_$cfg.log 2>&1; echo "pmc $cfg rc=$?"
done
ls gpurun_out | head -50
