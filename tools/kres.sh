#!/bin/bash
# Compact per-kernel resource table of one instantiation unit: tools/kres.sh inst_64_big [name-filter]
# (VGPRs, scratch bytes per lane, waves per SIMD) from hipcc's kernel-resource-usage remarks.
cd "$(dirname "$0")/../alchemy_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$1.hip" -Rpass-analysis=kernel-resource-usage -o /dev/null 2>&1 |
  sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/Function Name:/ {name=$NF} / VGPRs:/ {v=$NF} /ScratchSize/ {s=$NF} /Occupancy/ {o=$NF; sub(/_ZN4alch/,"",name); printf "%-100s vgpr %4s scratch %5s occ %s\n", substr(name,1,100), v, s, o}' |
  grep -E "${2:-.}"
