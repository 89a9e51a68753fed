#!/bin/bash
# same-box A/B of library variants (tools/exp_libs/lib_*.so): wall time per update at shard sizes of an 8-, 4-, 2- and 1-GPU run
cp simulatedannealingabc.jl_amd/libsabc_hip.so /tmp/lib_orig.so
for rep in 1 2; do
for f in tools/exp_libs/lib_*.so; do
  cp $f simulatedannealingabc.jl_amd/libsabc_hip.so
  for n in 125000 1000000; do
    timeout -k 10 200 python bench.py --n-particles $n --steps 200 --warmup 5 --no-cpu-baseline --repeats 5 --no-kernel-events > /tmp/ab.json 2>/dev/null
    python3 -c "
import json
j=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); print('$(basename $f .so) n=$n', '%.4e sims/s  median %.2f us/update (best %.2f)' % (j['value'], j['ms_per_step']*1e3, $n/j['value_max']*1e6))"
  done
done
done
cp /tmp/lib_orig.so simulatedannealingabc.jl_amd/libsabc_hip.so
