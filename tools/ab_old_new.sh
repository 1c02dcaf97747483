#!/bin/bash
# Same-box A/B of two builds of the library on the headline: tools/ab_old_new.sh alchemy_amd/lib/variants/old.so
mkdir -p gpurun_out; out=gpurun_out/ab_old_new.txt; : > $out
F="--no-pow --no-full --no-general --no-pipeline --no-tunnel-hs --no-config2 --no-q30 --no-n16 --cpu-ops 0 --steps 10 --warmup 2"
for lib in "$1" "" "$1" "" "$1" ""; do
  v=$(ALCH_LIB_PATH=$lib timeout -k 10 200 python3 bench.py $F 2>/dev/null | tail -1 | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['value']))") || exit 1
  echo "${lib:-current} $v" >> $out
done
cat $out
