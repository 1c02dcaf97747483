#!/bin/bash
# Same-box A/B of two builds of the library on bench.py's headline, full_mul, Pow-in/out, < 2^30-moduli and n = 2^16 lines:
# tools/ab_lib_lines.sh LIB_A LIB_B [rounds]   ("product" = alchemy_amd/lib/libalchemy_hip.so)
cd "$(dirname "$0")/.."
A=$1; B=$2; R=${3:-2}
run() {
    lib=$1; [ "$lib" = product ] && lib=alchemy_amd/lib/libalchemy_hip.so
    ALCH_LIB_PATH=$lib timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --cpu-ops 0 --no-general --no-pipeline --no-tunnel-hs --no-config2 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('$1', 'headline', round(d['value']), 'full_mul', round(d['full_mul']['ops_per_s']), d['full_mul']['batch_checksum'].get('ok'), 'pow', round(d['pow_basis_in_out_ops_per_s']), 'q30', round(d['moduli_below_2_30']['ops_per_s']), 'n16', round(d['n16_six_limbs']['ops_per_s']))
"
}
for i in $(seq 1 $R); do run $A; run $B; done
