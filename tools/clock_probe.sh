#!/bin/bash
# Run ON THE GPU BOX: what the shader clock and the board power are WHILE the headline kernels run (the instruction-count ceiling of
# DESIGN.md section 4 is priced at the 2.4 GHz peak clock; a power-limited clock moves that ceiling).  Samples rocm-smi once a second
# next to a long headline-only bench.py run, then once more when idle.   usage: tools/clock_probe.sh [steps]   -> stdout
steps=${1:-600}
root="$(cd "$(dirname "$0")/.." && pwd)"
cd "$root"
HEAD="--cpu-ops 0 --no-full --no-pow --no-general --no-pipeline --no-tunnel-hs --no-config2 --no-q30 --no-n16"
python3 bench.py --steps "$steps" --warmup 5 $HEAD > /tmp/clock_probe_bench.json 2> /tmp/clock_probe_bench.err &
pid=$!
sleep 6                                   # import torch, build the ring, generate the batch
i=0
while kill -0 $pid 2> /dev/null && [ $i -lt 40 ]; do
    echo "== busy sample $i"
    rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "sclk|mclk|fclk|Power|GPU use|busy" | head -12
    sleep 1
    i=$((i + 1))
done
wait $pid
echo "== bench line"
python3 -c "
import json; d = json.loads(open('/tmp/clock_probe_bench.json').read().strip().splitlines()[-1]); print(json.dumps({k: d[k] for k in ('value', 'ms_per_step', 'steps')}), d['roofline']['frac'])"
sleep 3
echo "== idle sample"
rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "sclk|mclk|fclk|Power|GPU use|busy" | head -12
