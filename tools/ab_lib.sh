#!/bin/bash
# Same-box A/B of two builds of the library on bench.py's headline: tools/ab_lib.sh LIB_A LIB_B [rounds]   (paths of .so files;
# "product" = alchemy_amd/lib/libalchemy_hip.so).  Alternates A, B, A, B ...; prints op/s of each run.
cd "$(dirname "$0")/.."
A=$1; B=$2; R=${3:-2}
run() {
    lib=$1; [ "$lib" = product ] && lib=alchemy_amd/lib/libalchemy_hip.so
    ALCH_LIB_PATH=$lib timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --cpu-ops 0 --no-full --no-pow --no-general --no-pipeline --no-tunnel-hs --no-config2 --no-q30 --no-n16 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', round(d['value']), d['batch_checksum'].get('ok'))
"
}
for i in $(seq 1 $R); do run $A; run $B; done
