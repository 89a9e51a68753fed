#!/bin/bash
# SQ counters of the update kernel (two --pmc passes, kernel-trace only): where the wave cycles go.
# usage (GPU box, repo root): tools/pmc_sq.sh [bench args]   -> gpurun_out/pmc_sq_{a,b}/...
set -e
root=$(pwd); out=$root/gpurun_out; mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d "$out/pmc_sq_a" -- python3 "$root/bench.py" --steps 20 --warmup 3 --no-cpu-baseline "$@" > "$out/pmc_sq_a.json" 2> "$out/pmc_sq_a.err"
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU \
  --kernel-trace --output-format csv -d "$out/pmc_sq_b" -- python3 "$root/bench.py" --steps 20 --warmup 3 --no-cpu-baseline "$@" > "$out/pmc_sq_b.json" 2> "$out/pmc_sq_b.err"
echo done
