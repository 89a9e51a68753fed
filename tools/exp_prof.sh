#!/bin/bash
# per-kernel rocprofv3 averages for every library variant in tools/exp_libs: tools/exp_prof.sh <cfg> <kernel-substring>
cfg=${1:-cfg2}; pat=${2:-k_reduce_control}
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cp simulatedannealingabc.jl_amd/libsabc_hip.so /tmp/lib_orig.so
export TMPDIR=/tmp
for f in tools/exp_libs/lib_*.so; do
  cp $f simulatedannealingabc.jl_amd/libsabc_hip.so
  name=$(basename $f .so)
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out/expprof_$name" -- python3 "$root/bench.py" --config $cfg --steps 50 --warmup 5 --no-cpu-baseline > "$out/expprof_$name.json" 2> "$out/expprof_$name.err")
  k=$(find "$out/expprof_$name" -name '*kernel_stats.csv' | head -1)
  echo "$name: $(grep "$pat" "$k" | awk -F'","' '{printf "%s calls avg %.2f us; ", $2, $4/1000}')"
done
cp /tmp/lib_orig.so simulatedannealingabc.jl_amd/libsabc_hip.so
