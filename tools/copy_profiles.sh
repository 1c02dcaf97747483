#!/bin/bash
# Run HERE after tools/profile_round.sh r03, tools/pmc_valu.sh r03 and a bench.py run into gpurun_out/r3/bench_final.json on the GPU box:
# copies the summaries the judge reads from gpurun_out/ (scratch) into profiles/ (tracked).
cd /root/repo
d=gpurun_out/prof_r03
cp $d/r03_stats_kernel_stats.csv profiles/r03_kernel_stats.csv; cp $d/r03_stats1_kernel_stats.csv profiles/r03_kernel_stats_one_stream_all_lines.csv
cp $d/r03_general_kernel_stats.csv profiles/r03_general_kernel_stats.csv; cp $d/r03_homom_kernel_stats.csv profiles/r03_homomrlwr_kernel_stats.csv
for f in general_index homomrlwr_pipeline tunnel_base2 config2 crt_half extra; do cp $d/r03_$f.jsonl profiles/; done
cp $d/traffic.json profiles/r03_traffic_pmc.json; cp $d/traffic.json profiles/traffic_latest.json
cat $d/commands.txt gpurun_out/pmc_r03/commands.txt > profiles/r03_commands.txt
cp gpurun_out/pmc_r03/r03_pmc_valu.json profiles/r03_pmc_valu.json
cp gpurun_out/r3/bench_final.json profiles/r03_bench.json
python - <<'PY'
import json,sys
sys.path.insert(0,'/root/repo'); import bench
t=json.load(open('/root/repo/profiles/traffic_latest.json')); p=json.load(open('/root/repo/profiles/r03_pmc_valu.json'))
print("sha now", bench.kernel_src_sha16(), "traffic", t['kernel_src_sha16'], "pmc", p['kernel_src_sha16'])
d=json.load(open('/root/repo/profiles/r03_bench.json')); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'])
PY
