#!/bin/bash
# What an 8-GPU strong-scaling shard of n = 1e6 looks like on ONE GPU: n = 125 000 (and 250 000 / 500 000: the 4- and 2-GPU
# shards), wall time per update against the kernels' own durations (rocprofv3 kernel trace of the same command).
out=gpurun_out/small; mkdir -p $out
export TMPDIR=/tmp
for n in 125000; do
  timeout -k 10 200 python bench.py --n-particles $n --steps 200 --warmup 5 --no-cpu-baseline --repeats 3 --no-kernel-events > $out/bench_$n.json 2> $out/bench_$n.err; echo "bench $n rc=$?"
  python3 -c "
import json
j=json.loads(open('$out/bench_$n.json').read().strip().splitlines()[-1]); print('n=$n', '%.3e sims/s  %.2f us/update  launches/update %.2f' % (j['value'], j['ms_per_step']*1e3, j['launches_per_update']), j['state']['n_resampling'])"
done
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/prof125 -- python3 /root/repo/bench.py --n-particles 125000 --steps 200 --warmup 5 --no-cpu-baseline --repeats 1 --no-kernel-events > /root/repo/$out/prof125.json 2> /root/repo/$out/prof125.err); echo "prof rc=$?"
find $out/prof125 -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_125k.csv \; && python3 -c "
import csv
for r in list(csv.DictReader(open('$out/kernel_stats_125k.csv')))[:12]: print(r['Name'][:50], r['Calls'], r['AverageNs'], r['Percentage'])"
