#!/bin/bash
# SQ counters of the update kernel (two --pmc passes, kernel-trace only): where the wave cycles go.
# usage (GPU box, repo root): tools/pmc_sq.sh cfg2|cfg3|cfg4|cfg5 [more bench args]   -> gpurun_out/pmc_sq_{a,b}_<cfg>/...
# (the program sits directly behind `--`: no env / bash -c hop under the profiler)
set -e
cfg=${1:-cfg2}; shift || true
root=$(pwd); out=$root/gpurun_out; mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d "$out/pmc_sq_a_$cfg" -- python3 "$root/bench.py" --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --repeats 1 --preheat-seconds 0 "$@" > "$out/pmc_sq_a_$cfg.json" 2> "$out/pmc_sq_a_$cfg.err"
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU \
  --kernel-trace --output-format csv -d "$out/pmc_sq_b_$cfg" -- python3 "$root/bench.py" --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --repeats 1 --preheat-seconds 0 "$@" > "$out/pmc_sq_b_$cfg.json" 2> "$out/pmc_sq_b_$cfg.err"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS \
  --kernel-trace --output-format csv -d "$out/pmc_sq_c_$cfg" -- python3 "$root/bench.py" --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --repeats 1 --preheat-seconds 0 "$@" > "$out/pmc_sq_c_$cfg.json" 2> "$out/pmc_sq_c_$cfg.err" || echo "pass c not available"
echo "done $cfg"
