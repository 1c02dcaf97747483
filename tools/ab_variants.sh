#!/bin/bash
# AB_FLAGS="" adds the full mul_ and Pow-basis lines to every run (columns 3 and 4).
# Same-box comparison of library variants (tools/build_variant.sh) on the headline: tools/ab_variants.sh nt_in nt_out ...   ("cur" = current build)
mkdir -p gpurun_out; out=gpurun_out/ab_variants.txt; : > $out
F="${AB_FLAGS:---no-pow --no-full} --no-general --no-pipeline --no-tunnel-hs --no-config2 --no-q30 --no-n16 --cpu-ops 0 --steps 10 --warmup 2"
for round in 1 2; do
for v in cur "$@"; do
  lib=""; [ "$v" != cur ] && lib=alchemy_amd/lib/variants/$v.so
  r=$(ALCH_LIB_PATH=$lib timeout -k 10 200 python3 bench.py $F 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['batch_checksum']['ok'], round(d['full_mul']['ops_per_s']) if 'full_mul' in d else '', round(d.get('pow_basis_in_out_ops_per_s') or 0))") || exit 1
  echo "$v $r" >> $out
done; done
cat $out
