#!/bin/bash
# A/B of prebuilt library variants in ONE box session: tools/exp_ab.sh <bench args...>
# (variants are tools/exp_libs/lib_*.so, built locally with different -D flags; not tracked)
cp simulatedannealingabc.jl_amd/libsabc_hip.so /tmp/lib_orig.so
for rep in 1 2 3; do
  for f in tools/exp_libs/lib_*.so; do
    cp $f simulatedannealingabc.jl_amd/libsabc_hip.so
    python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 > gpurun_out/ab_$(basename $f .so)_$rep.json
  done
done
cp /tmp/lib_orig.so simulatedannealingabc.jl_amd/libsabc_hip.so
python - <<'PY'
import glob, json, collections
r = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/ab_lib_*_*.json")):
    j = json.load(open(f)); r[f.split("ab_")[1].rsplit("_", 1)[0]].append((j["roofline"]["avg_launch_us"], j["ms_per_step"] * 1e3))
for k, v in sorted(r.items()):
    print(k, "kernel", " ".join("%.1f" % a for a, _ in v), "| step", " ".join("%.1f" % b for _, b in v))
PY
