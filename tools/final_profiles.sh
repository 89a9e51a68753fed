#!/bin/bash
# The evidence pass of a round (GPU box, repo root): kernel stats for cfg2-5, FETCH/WRITE_SIZE for cfg2, SQ counters for
# cfg2-5.  usage: tools/final_profiles.sh r03   -> gpurun_out/...; tools/summarize_profiles.py r03 copies into profiles/.
tag=${1:-r04}
set -x
timeout -k 10 900 bash tools/profile_round.sh "$tag" > "gpurun_out/profile_round_$tag.log" 2>&1; echo "profile rc=$?"
for cfg in cfg2 cfg3 cfg4 cfg5; do
  timeout -k 10 300 bash tools/pmc_sq.sh $cfg > gpurun_out/pmc_sq_$cfg.log 2>&1; echo "pmc $cfg rc=$?"
done
