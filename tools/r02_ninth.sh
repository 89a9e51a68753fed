#!/bin/bash
# round 2, call 9: A/B of the k_update workgroup size (grid tail)
set -x
bash tools/exp_ab2.sh "cfg2 cfg3" 2 > gpurun_out/ab_block.txt 2>&1
cat gpurun_out/ab_block.txt | tail -12
