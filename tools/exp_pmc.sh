#!/bin/bash
# FETCH_SIZE of k_update for each prebuilt variant in tools/exp_libs (one box session)
export TMPDIR=/tmp; root=$(pwd)
cp simulatedannealingabc.jl_amd/libsabc_hip.so /tmp/lib_orig.so
for f in tools/exp_libs/lib_*.so; do
  cp $f simulatedannealingabc.jl_amd/libsabc_hip.so
  tag=$(basename $f .so)
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $root/gpurun_out/pmcx_$tag -- python3 $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline --preheat-seconds 0 > $root/gpurun_out/pmcx_$tag.json 2>/dev/null)
done
cp /tmp/lib_orig.so simulatedannealingabc.jl_amd/libsabc_hip.so
python - <<'PY'
import csv, glob, os
for d in sorted(glob.glob("gpurun_out/pmcx_lib_*")):
    if not os.path.isdir(d): continue
    f = sorted(glob.glob(d + "/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)[-1]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_update<" in r["Kernel_Name"]]
    print(os.path.basename(d), "FETCH_SIZE avg KiB %.0f" % (sum(v) / len(v)), "min %.0f max %.0f" % (min(v), max(v)))
PY
