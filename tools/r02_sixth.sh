#!/bin/bash
set -x
out=gpurun_out/r02f; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_golden.py -m gpu -q -k "gk or cfg4 or golden" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -4 $out/pytest.log
timeout -k 10 600 bash tools/exp_ab2.sh "cfg4" 3 > $out/ab.log 2>&1; tail -8 $out/ab.log
