#!/bin/bash
# Same-box A/B of library variants on the general-index key switch (H5') and the HomomRLWR pipeline: tools/ab_general.sh gennt ...
mkdir -p gpurun_out; out=gpurun_out/ab_general.txt; : > $out
for round in 1 2; do
for v in cur "$@"; do
  lib=""; [ "$v" != cur ] && lib=alchemy_amd/lib/variants/$v.so
  g=$(ALCH_LIB_PATH=$lib timeout -k 10 200 python3 tools/bench_general.py 20475 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['mul_relin_ops_per_s']), round(d['mul_full_4_5_3_ops_per_s']))") || exit 1
  p=$(ALCH_LIB_PATH=$lib timeout -k 10 200 python3 tools/bench_homomrlwr.py 1024 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['pipelines_per_s']), d['out_checksum'])") || exit 1
  echo "$v $g $p" >> $out
done; done
cat $out
