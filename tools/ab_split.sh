#!/bin/bash
# Same-box A/B of library variants on the split rings (tools/split_probe.py): tools/ab_split.sh splitnt
mkdir -p gpurun_out; out=gpurun_out/ab_split.txt; : > $out
for round in 1 2; do
for v in cur "$@"; do
  lib=""; [ "$v" != cur ] && lib=alchemy_amd/lib/variants/$v.so
  ALCH_LIB_PATH=$lib timeout -k 10 200 python3 tools/split_probe.py 2>/dev/null | grep "^{" | python3 -c "
import sys, json
print('$v', [json.loads(l)['by_split_fused']['2']['ops_per_s'] for l in sys.stdin])" >> $out || exit 1
done; done
cat $out
