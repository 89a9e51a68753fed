#!/bin/bash
set -x
out=gpurun_out/r02m; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_anchors.py -m gpu -q --durations=8 > $out/pytest_anchors.log 2>&1; echo "rc=$?" >> $out/pytest_anchors.log
tail -25 $out/pytest_anchors.log
