#!/bin/bash
# Builds the three compiled replays of the reference's examples against alchemy_amd/lib/libalchemy_hip.so.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
for ex in arithmetic_replay homomrlwr_replay tunnel_replay; do
    g++ -O2 -std=c++17 -o "$ROOT/examples/$ex" "$ROOT/examples/$ex.cpp" -L"$ROOT/alchemy_amd/lib" -lalchemy_hip \
        -Wl,-rpath,"$ROOT/alchemy_amd/lib" -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib
done
# the native multi-GPU driver also needs the HIP runtime (hipSetDevice) and the optional RCCL route
g++ -O2 -std=c++17 -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o "$ROOT/examples/ringround_multi" "$ROOT/examples/ringround_multi.cpp" \
    -L"$ROOT/alchemy_amd/lib" -lalchemy_rccl -lalchemy_hip -Wl,-rpath,"$ROOT/alchemy_amd/lib" -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
