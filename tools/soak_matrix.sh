#!/bin/bash
# every (config, proposal, algorithm) at n = 1e6 for a few hundred updates + two long runs: completes, sane state
out=gpurun_out/soak; mkdir -p $out
for cfg in cfg2 cfg3 cfg4 cfg5; do
  for prop in randomwalk de stretch; do
    for alg in single_eps multi_eps; do
      steps=200; [ $cfg = cfg5 ] && steps=60; [ $cfg = cfg4 ] && steps=100
      timeout -k 10 280 python bench.py --config $cfg --proposal $prop --algorithm $alg --steps $steps --warmup 2 --no-cpu-baseline --repeats 1 \
        > $out/${cfg}_${prop}_${alg}.json 2> $out/${cfg}_${prop}_${alg}.err
      echo "$cfg $prop $alg rc=$? $(python3 -c "
import json,sys
try:
    j=json.loads(open('$out/${cfg}_${prop}_${alg}.json').read().strip().splitlines()[-1]); s=j['state']
    print('%.3e sims/s  %.1f us/update  n_accept=%d n_resampling=%d eps=%s' % (j['value'], j['ms_per_step']*1e3, s['n_accept'], s['n_resampling'], ['%.3g'%e for e in s['eps']]))
except Exception as e: print('PARSE FAIL', e)
")"
    done
  done
done
timeout -k 10 280 python bench.py --steps 3000 --warmup 2 --no-cpu-baseline --repeats 1 > $out/long_cfg2.json 2> $out/long_cfg2.err; echo "long cfg2 rc=$?"
python3 -c "
import json
j=json.loads(open('$out/long_cfg2.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['state'])"
