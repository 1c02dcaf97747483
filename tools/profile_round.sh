#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: tools/profile_round.sh r01
# Collects, with the same bench.py command line each time,
#   1. rocprofv3 --kernel-trace --stats            -> gpurun_out/prof_$tag/stats_kernel_stats.csv
#   2. rocprofv3 --kernel-trace --pmc FETCH_SIZE   -> gpurun_out/prof_$tag/fetch_counter_collection.csv
#   3. rocprofv3 --kernel-trace --pmc WRITE_SIZE   -> gpurun_out/prof_$tag/write_counter_collection.csv
# (counters in their own passes, never together with --stats or an API trace) and summarises them with
# tools/traffic_summary.py.  The program after `--` is python3 itself.
set -e
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $root/bench.py --steps 5 --warmup 1 --cpu-ops 0 --no-full --no-pow --no-general"
PMC="python3 $root/bench.py --steps 1 --warmup 0 --batch 2048 --cpu-ops 0 --no-full --no-pow --no-general"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o stats -- $CMD > "$out/stats.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out" -o fetch -- $PMC > "$out/fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out" -o write -- $PMC > "$out/write.log" 2>&1
cd "$root"
python3 tools/traffic_summary.py "$out" 2048 > "$out/traffic.json"
cat "$out/traffic.json"
