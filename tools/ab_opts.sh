#!/bin/bash
# Same-box A/B of launch options on bench.py's headline / full_mul / Pow lines.   usage: tools/ab_opts.sh "nstreams=2" "nstreams=3" ...
# (each argument is a space-separated list of NAME=VALUE options for one run; runs are issued in the order given)
cd "$(dirname "$0")/.."
for o in "$@"; do
    args=""
    for kv in $o; do args="$args --opt $kv"; done
    timeout -k 10 200 python3 bench.py --no-config2 --no-q30 --no-n16 --no-tunnel-hs --no-general --no-pipeline --cpu-ops 0 --steps 4 $args 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print(json.dumps({'opts': '$o', 'headline': round(d['value']), 'full_mul': round(d['full_mul']['ops_per_s']), 'full_mul_ok': d['full_mul']['batch_checksum'].get('ok'), 'pow_in_out': round(d['pow_basis_in_out_ops_per_s'])}))
"
done
