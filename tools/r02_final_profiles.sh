#!/bin/bash
# round 2, final evidence pass: kernel stats for cfg2-5, FETCH/WRITE_SIZE, SQ counters for cfg2-5
set -x
timeout -k 10 900 bash tools/profile_round.sh r02 > gpurun_out/profile_round_r02.log 2>&1; echo "profile rc=$?"
for cfg in cfg2 cfg3 cfg4 cfg5; do
  timeout -k 10 300 bash tools/pmc_sq.sh $cfg > gpurun_out/pmc_sq_$cfg.log 2>&1; echo "pmc $cfg rc=$?"
done
