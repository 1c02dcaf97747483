#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: tools/pmc_valu.sh r03
# SQ counters of the kernels the roofline statements rest on, each workload in its own rocprofv3 pass with --kernel-trace only
# (never together with --stats or an API trace; the program after `--` is python3 itself):
#   full_mul   bench.py with its full_mul line (B = 2048)        (k_ks_accum_half<UP>, k_rescale_out_lin)
#   tunnel_hs  tools/bench_tunnel.py 1024                         (the BaseBGad-2 hops of examples/Tunnel.hs)
#   headline   bench.py, config 3 only, B = 2048, one stream      (k_tensor_intt_split, k_ks_accum_half)
#   general    tools/bench_general.py 20475                        (k_gen_crt, k_gen_crt_digits, k_hint_mac_v, k_tensor_ew, ...)
#   homomrlwr  tools/bench_homomrlwr.py 1024                       (the whole ringRound pipeline)
# tools/pmc_valu_summary.py turns the CSVs into profiles/${tag}_pmc_valu.json.
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
CTRS="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run() { name=$1; shift; echo "== $name: rocprofv3 --kernel-trace --pmc $CTRS -- $*" >> "$out/commands.txt";
        timeout -k 10 400 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$out" -o "$name" -- "$@" > "$out/$name.log" 2>&1 || echo "$name failed" >> "$out/commands.txt"; }
run headline  python3 $root/bench.py --steps 1 --warmup 0 --batch 2048 --cpu-ops 0 --no-full --no-pow --no-general --no-pipeline --no-tunnel-hs --no-config2 --no-q30 --no-n16 --opt one_stream=1
# round 4: PT2CT's whole mul_ (k_ks_accum_half<UP>, k_rescale_out_lin) and the Tunnel.hs hops (k_gen_crt_base2_digits, k_tunnel_mac_e)
run full_mul  python3 $root/bench.py --steps 1 --warmup 0 --batch 2048 --cpu-ops 0 --no-pow --no-general --no-pipeline --no-tunnel-hs --no-config2 --no-q30 --no-n16 --opt one_stream=1
run tunnel_hs python3 $root/tools/bench_tunnel.py 1024
run general   python3 $root/tools/bench_general.py 20475
run homomrlwr python3 $root/tools/bench_homomrlwr.py 1024
cd "$root"
python3 tools/pmc_valu_summary.py "$out" > "$out/${tag}_pmc_valu.json" 2> "$out/summary.err"
tail -c 1500 "$out/${tag}_pmc_valu.json"
