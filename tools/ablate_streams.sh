#!/bin/bash
# Timing-only ablation (wrong results by design, -DALCH_ABLATE build): which of kernel B's memory streams costs what.
# ALCH_EXP_FLAGS bits: 1 tensor inputs, 2 digits, 4 outputs, 8 hint rows aliased to a cache-resident range.
mkdir -p gpurun_out
out=gpurun_out/ablate_streams.txt; : > $out
for f in 0 1 2 4 8 10 15 0; do
  ALCH_LIB_PATH=alchemy_amd/lib/variants/ablate.so ALCH_EXP_FLAGS=$f timeout -k 10 120 python3 tools/perf_probe.py 2>/dev/null | grep mul_relin | sed "s/^/flags=$f /" >> $out || exit 1
done
cat $out
