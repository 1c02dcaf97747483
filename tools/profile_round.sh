#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: tools/profile_round.sh r02
# Collects, with bench.py command lines recorded next to each file,
#   1. rocprofv3 --kernel-trace --stats, default launch structure (two streams), headline only   -> ${tag}_kernel_stats.csv
#   2. the same with --opt one_stream=1 and every extra line (mul_, Pow-basis, general index)    -> ${tag}_kernel_stats_one_stream_all_lines.csv
#   3. rocprofv3 --kernel-trace --pmc FETCH_SIZE   (own pass, headline only, B = 2048)
#   4. rocprofv3 --kernel-trace --pmc WRITE_SIZE   (own pass)                                     -> ${tag}_traffic_pmc.json
#   5. tools/bench_general.py under --kernel-trace --stats                                        -> ${tag}_general_kernel_stats.csv
#   6. the HomomRLWR pipeline (config 4) under --kernel-trace --stats                             -> ${tag}_homomrlwr_kernel_stats.csv
#   7. the measurement tools themselves (no profiler): general indices, pipeline, Tunnel.hs hops, config 2, other paths -> ${tag}_*.jsonl
# Counters run in their own passes, never together with --stats or an API trace.  The program after `--` is python3 itself.
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
HEAD="--cpu-ops 0 --no-full --no-pow --no-general --no-pipeline --no-tunnel-hs --no-config2 --no-q30 --no-n16"
run() { name=$1; shift; echo "== $name: $*" >> "$out/commands.txt"; timeout -k 10 300 "$@" > "$out/$name.log" 2>&1 || echo "$name failed" >> "$out/commands.txt"; }
run stats   rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o stats -- python3 $root/bench.py --steps 5 --warmup 1 $HEAD
run stats1  rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o stats1 -- python3 $root/bench.py --steps 5 --warmup 1 --cpu-ops 0 --no-tunnel-hs --no-config2 --opt one_stream=1
run fetch   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out" -o fetch -- python3 $root/bench.py --steps 1 --warmup 0 --batch 2048 $HEAD
run write   rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out" -o write -- python3 $root/bench.py --steps 1 --warmup 0 --batch 2048 $HEAD
run general rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o general -- python3 $root/tools/bench_general.py 11648 20475
run homom   rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o homom -- python3 $root/tools/bench_homomrlwr.py 1024
cd "$root"
tool() { name=$1; shift; echo "== $name: $*" >> "$out/commands.txt"; timeout -k 10 300 "$@" 2> "$out/$name.err" | grep "^{" > "$out/${tag}_$name.jsonl" || echo "$name failed" >> "$out/commands.txt"; }
tool general_index       python3 tools/bench_general.py
tool homomrlwr_pipeline  python3 tools/bench_homomrlwr.py 4096
tool homomrlwr_pipeline_1024        python3 tools/bench_homomrlwr.py 1024
tool homomrlwr_pipeline_1024_1lane  python3 tools/bench_homomrlwr.py 1024 lanes=1
tool tunnel_base2        python3 tools/bench_tunnel.py
tool config2             python3 tools/bench_config2.py
tool crt_half            python3 tools/bench_crt_half.py
tool extra               python3 tools/bench_extra.py
python3 tools/traffic_summary.py "$out" 2048 > "$out/traffic.json" 2> "$out/traffic.err"
for f in stats stats1 general homom; do
  src=$(find "$out" -name "${f}_kernel_stats.csv" | head -1)
  [ -n "$src" ] && cp "$src" "$out/${tag}_${f}_kernel_stats.csv"
done
ls "$out" | head -40
cat "$out/traffic.json"
