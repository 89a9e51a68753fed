#!/bin/bash
# first GPU call of round 2: full GPU suite, bench (1 rank + 2-rank gloo rehearsal), anchor calibration
set -x
mkdir -p gpurun_out/r02a
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a/pytest.log
tail -5 gpurun_out/r02a/pytest.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r02a/bench1.json 2> gpurun_out/r02a/bench1.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --gpus 2 --dist-backend gloo --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02a/bench2_gloo.json 2> gpurun_out/r02a/bench2_gloo.err; echo "bench2 rc=$?"
timeout -k 10 300 python bench.py --gpus 2 --dist-backend gloo --steps 20 --warmup 5 --no-cpu-baseline --proposal de > gpurun_out/r02a/bench2_gloo_de.json 2> gpurun_out/r02a/bench2_gloo_de.err; echo "bench2de rc=$?"
timeout -k 10 600 python tools/calibrate_anchors.py all > gpurun_out/r02a/anchors.jsonl 2> gpurun_out/r02a/anchors.err; echo "anchors rc=$?"
