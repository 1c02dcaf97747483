#!/bin/bash
# Same-box A/B of two builds of the library on the Tunnel.hs hops and the HomomRLWR pipeline: tools/ab_hops.sh LIB_A LIB_B [rounds]
# ("product" = alchemy_amd/lib/libalchemy_hip.so).  Alternates A, B, A, B ...
cd "$(dirname "$0")/.."
A=$1; B=$2; R=${3:-2}
run() {
    lib=$1; [ "$lib" = product ] && lib=alchemy_amd/lib/libalchemy_hip.so
    h=$(ALCH_LIB_PATH=$lib timeout -k 10 200 python3 tools/bench_tunnel.py 2048 2>/dev/null | python3 -c "
import sys, json
print([round(json.loads(l)['tunnels_per_s']) for l in sys.stdin if l.startswith('{')])")
    p=$(ALCH_LIB_PATH=$lib timeout -k 10 200 python3 tools/bench_homomrlwr.py 1024 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(round(d['pipelines_per_s']), d['checksum_at_positions'])")
    echo "$1 hops $h pipeline $p"
}
for i in $(seq 1 $R); do run $A; run $B; done
