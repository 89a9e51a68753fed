#!/bin/bash
# Collects the rocprofv3 evidence bench.py's numbers are checked against (run on the GPU box from the repo root):
#   tools/profile_round.sh r01          -> gpurun_out/prof_r01_*; then tools/summarize_profiles.py r01 copies the
#                                          summaries into profiles/ (tracked).
# Kernel-trace/stats and the PMC passes are separate runs (never combined with sys/hip traces).  The kernel-trace pass keeps
# bench.py's one-second pre-heat (the averages are those of the sustained state, like the bench line's); the counter passes
# run cold (--preheat-seconds 0: counters per dispatch, a few dozen dispatches are enough).
set -e
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
for cfg in cfg2 cfg3 cfg4 cfg5; do
  steps=30; [ $cfg = cfg2 ] && steps=50
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_${tag}_$cfg" -- \
    python3 "$root/bench.py" --config $cfg --steps $steps --warmup 5 --no-cpu-baseline --repeats 1 \
    > "$out/prof_${tag}_$cfg.json" 2> "$out/prof_${tag}_$cfg.err"
  echo "profiled $cfg"
done
for pmc in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d "$out/pmc_${tag}_$pmc" -- \
    python3 "$root/bench.py" --steps 50 --warmup 5 --no-cpu-baseline --repeats 1 --preheat-seconds 0 \
    > "$out/pmc_${tag}_$pmc.json" 2> "$out/pmc_${tag}_$pmc.err"
  echo "counted $pmc"
done
