#!/bin/bash
# GPU box: the instrumented variant in place of the library for three sizes, then the library back
cp simulatedannealingabc.jl_amd/libsabc_hip.so /tmp/lib_orig.so
cp tools/exp_libs/lib_rctiming.so simulatedannealingabc.jl_amd/libsabc_hip.so
for n in 125000 1000000; do PYTHONPATH=. timeout -k 10 120 python tools/rc_timing.py $n; done
cp /tmp/lib_orig.so simulatedannealingabc.jl_amd/libsabc_hip.so
