#!/bin/bash
# A/B of prebuilt library variants in ONE box session: tools/exp_ab2.sh "cfg3 cfg4" [reps]
# (variants are tools/exp_libs/lib_*.so from tools/build_variants.sh; not tracked)
cfgs=${1:-cfg2}; reps=${2:-2}
mkdir -p gpurun_out/ab
cp simulatedannealingabc.jl_amd/libsabc_hip.so /tmp/lib_orig.so
for rep in $(seq 1 $reps); do
  for f in tools/exp_libs/lib_*.so; do
    cp $f simulatedannealingabc.jl_amd/libsabc_hip.so
    for cfg in $cfgs; do
      timeout -k 10 120 python bench.py --no-cpu-baseline --config $cfg --steps 30 --warmup 5 2>/dev/null | tail -1 > gpurun_out/ab/$(basename $f .so)__${cfg}__$rep.json
    done
  done
done
cp /tmp/lib_orig.so simulatedannealingabc.jl_amd/libsabc_hip.so
python - <<'PY'
import glob, json, collections
r = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/ab/lib_*__*__*.json")):
    try:
        j = json.load(open(f))
    except Exception:
        continue
    lib, cfg, _ = f.split("/")[-1][:-5].split("__")
    r[(cfg, lib)].append((j["roofline"]["avg_launch_us"], j["ms_per_step"] * 1e3, j["value"]))
for k, v in sorted(r.items()):
    print("%-6s %-28s" % k, "kernel", " ".join("%.1f" % a for a, _, _ in v), "| step", " ".join("%.1f" % b for _, b, _ in v), "| %.3e" % max(c for _, _, c in v))
PY
