#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the product's HOST engine (engine.cpp, control.hpp, host_math.hpp) on
# the CPU: builds tests/cpu_engine/libsabc_cpu_engine.so instrumented, runs the CPU-engine tests (single shard, gloo world
# sizes 2/3/8, failed collectives, comm-byte accounting) under it, and restores the plain build.  GPU sanitizers are not
# available on the pool; the device side is covered by the parity tests.
set -e
cd "$(dirname "$0")/.."
lib=tests/cpu_engine/libsabc_cpu_engine.so
python -c "import tests.cpu_engine as e; e.build()" 2>/dev/null || true
cp $lib /tmp/libsabc_cpu_engine.plain.so
trap 'cp /tmp/libsabc_cpu_engine.plain.so '$lib'; touch '$lib EXIT
flags="-O1 -g -fPIC -fno-fast-math -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer"
gcc $flags -std=c11 -fopenmp -c oracle/sabc_oracle.c -o /tmp/sabc_oracle_asan.o
g++ $flags -std=c++17 -shared -fopenmp -pthread -o $lib tests/cpu_engine/ref_backend.cpp simulatedannealingabc.jl_amd/csrc/engine.cpp /tmp/sabc_oracle_asan.o -lm
touch $lib
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python -m pytest tests/test_host_api.py tests/test_distributed.py tests/test_p2p_cpu_engine.py -q -m "not gpu" -x
