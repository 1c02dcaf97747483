#!/bin/bash
# Headline (alch_ct_mul_relin, n = 2^15, L = 4, B = 8192) under launch-option variants, one box, back to back.
# usage: tools/sweep_headline.sh "pipe=1" "pipe=1 chunk=512" ...   ("" = defaults); results -> gpurun_out/sweep_headline.jsonl
mkdir -p gpurun_out
out=gpurun_out/sweep_headline.jsonl
: > "$out"
F="--no-pow --no-full --no-general --no-pipeline --no-tunnel-hs --no-config2 --no-q30 --no-n16 --cpu-ops 0 --steps 10 --warmup 2"
for v in "$@"; do
    opts=""
    for kv in $v; do opts="$opts --opt $kv"; done
    line=$(timeout -k 10 300 python3 bench.py $F $opts 2>>gpurun_out/sweep_headline.err | tail -1) || { echo "variant '$v' failed" >> "$out"; exit 1; }
    python3 - "$v" "$line" >> "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(json.dumps({"variant": sys.argv[1], "value": d["value"], "ms_per_step": d["ms_per_step"], "frac": d["roofline"]["frac"], "checksum_ok": d.get("batch_checksum", {}).get("ok")}))
PY
done
cat "$out"
