#!/bin/bash
# builds library variants for an A/B session on the GPU box (tools/exp_ab2.sh): tools/exp_libs/lib_<name>.so
# usage: tools/build_variants.sh name1="-DFLAG=1 -DX=2" name2="..."
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/exp_libs
src=simulatedannealingabc.jl_amd/csrc
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  tmp=$(mktemp -d)
  for f in kernels sort hip_backend capi; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function $flags -c $src/$f.hip -o $tmp/$f.o &
  done
  hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -x c++ $flags -c $src/engine.cpp -o $tmp/engine.o &
  hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include $flags -c $src/rtc.cpp -o $tmp/rtc.o &
  wait
  for f in kernels sort hip_backend capi engine rtc; do [ -s $tmp/$f.o ] || { echo "lib_$name: $f did not compile"; exit 1; }; done
  hipcc --offload-arch=gfx950 -shared -Wl,-z,defs -o tools/exp_libs/lib_$name.so $tmp/*.o -ldl
  rm -rf $tmp
  echo "built lib_$name.so ($flags)"
done
