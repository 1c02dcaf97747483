#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: tools/profile_round.sh r02
# Collects, with bench.py command lines recorded next to each file,
#   1. rocprofv3 --kernel-trace --stats, default launch structure (two streams), headline only   -> ${tag}_kernel_stats.csv
#   2. the same with --opt one_stream=1 and every extra line (mul_, Pow-basis, general index)    -> ${tag}_kernel_stats_one_stream_all_lines.csv
#   3. rocprofv3 --kernel-trace --pmc FETCH_SIZE   (own pass, headline only, B = 2048)
#   4. rocprofv3 --kernel-trace --pmc WRITE_SIZE   (own pass)                                     -> ${tag}_traffic_pmc.json
#   5. tools/bench_general.py under --kernel-trace --stats                                        -> ${tag}_general_kernel_stats.csv
# Counters run in their own passes, never together with --stats or an API trace.  The program after `--` is python3 itself.
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
HEAD="--cpu-ops 0 --no-full --no-pow --no-general"
run() { name=$1; shift; echo "== $name: $*" >> "$out/commands.txt"; timeout -k 10 300 "$@" > "$out/$name.log" 2>&1 || echo "$name failed" >> "$out/commands.txt"; }
run stats   rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o stats -- python3 $root/bench.py --steps 5 --warmup 1 $HEAD
run stats1  rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o stats1 -- python3 $root/bench.py --steps 5 --warmup 1 --cpu-ops 0 --opt one_stream=1
run fetch   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out" -o fetch -- python3 $root/bench.py --steps 1 --warmup 0 --batch 2048 $HEAD
run write   rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out" -o write -- python3 $root/bench.py --steps 1 --warmup 0 --batch 2048 $HEAD
run general rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o general -- python3 $root/tools/bench_general.py 11648 20475
cd "$root"
python3 tools/traffic_summary.py "$out" 2048 > "$out/traffic.json" 2> "$out/traffic.err"
for f in stats stats1 general; do
  src=$(find "$out" -name "${f}_kernel_stats.csv" | head -1)
  [ -n "$src" ] && cp "$src" "$out/${tag}_${f}_kernel_stats.csv"
done
ls "$out" | head -40
cat "$out/traffic.json"
