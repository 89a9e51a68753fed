#!/bin/bash
set -x
bash tools/exp_ab2.sh "cfg4" 2 > gpurun_out/ab_pw.txt 2>&1
tail -6 gpurun_out/ab_pw.txt
