#!/bin/bash
# the device-vs-oracle suites under other seeds than the one the tests are written with (SABC_TEST_SEED): accept / resample
# counts are compared exactly, so a decision that flips between the device and the oracle on some stream would show
# usage (GPU box): tools/seed_sweep.sh 1 2 3 ...   (full logs: gpurun_out/seed_<seed>.log)
mkdir -p gpurun_out
for seed in "$@"; do
  SABC_TEST_SEED=$seed timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_priors.py tests/test_user_simulator.py tests/test_gpu_host_fdist.py tests/test_host_prior.py tests/test_gpu_edges.py tests/test_p2p.py tests/test_persistent.py tests/test_history_chunks.py \
    -m gpu -q -x -p no:cacheprovider > gpurun_out/seed_$seed.log 2>&1
  tail -2 gpurun_out/seed_$seed.log | sed "s/^/seed $seed: /"
  grep -E "^(FAILED|ERROR)" gpurun_out/seed_$seed.log | sed "s/^/seed $seed: /"
done
