#!/bin/bash
# one rocprofv3 --kernel-trace --stats pass of one config, top kernels printed: tools/prof_quick.sh <cfg> <steps> <tag>
cfg=${1:-cfg2}; steps=${2:-50}; tag=${3:-quick}
root=$(pwd); out=$root/gpurun_out; mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_${tag}_$cfg" -- \
  python3 "$root/bench.py" --config $cfg --steps $steps --warmup 5 --no-cpu-baseline > "$out/prof_${tag}_$cfg.json" 2> "$out/prof_${tag}_$cfg.err"
cd "$root"
f=$(find "$out/prof_${tag}_$cfg" -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("%-70s %5s %9.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
