#!/bin/bash
# Build an experimental variant of libalchemy_hip.so: tools/build_variant.sh NAME "-DFLAG=..." -> alchemy_amd/lib/variants/NAME.so
# Only inst_32_15.hip is rebuilt with the flags; every other object comes from the normal build.
set -e
cd "$(dirname "$0")/../alchemy_amd/csrc"
name=$1; shift
mkdir -p ../lib/variants build/var_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c inst_32_15.hip -o build/var_$name/inst_32_15.o
objs=$(ls build/*.o | grep -v inst_32_15.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/$name.so $objs build/var_$name/inst_32_15.o
echo built ../lib/variants/$name.so
