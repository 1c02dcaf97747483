#!/bin/bash
# Build an experimental variant of libalchemy_hip.so: tools/build_variant.sh NAME "-DFLAG=..." -> alchemy_amd/lib/variants/NAME.so
# Only one instantiation unit is rebuilt with the flags (UNIT=inst_32_15 by default: the n = 2^15 kernels; UNIT=inst_gen: the general-index
# kernels); every other object comes from the normal build.
set -e
cd "$(dirname "$0")/../alchemy_amd/csrc"
name=$1; shift
unit=${UNIT:-inst_32_15}
mkdir -p ../lib/variants build/var_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $unit.hip -o build/var_$name/$unit.o
objs=$(ls build/*.o | grep -v "/$unit.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/$name.so $objs build/var_$name/$unit.o
echo built ../lib/variants/$name.so
