#!/bin/bash
# usage: tools/sweep.sh "ENV1=.. ENV2=.." ...   -> one bench.py run (5 steps) per configuration, interleaved twice
run() { env $1 timeout -k 10 100 python bench.py --cpu-ops 0 --steps 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']))"; }
for rep in 1 2; do for cfg in "$@"; do echo "$cfg -> $(run "$cfg")"; done; done
