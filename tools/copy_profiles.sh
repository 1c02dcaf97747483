#!/bin/bash
# Run HERE after tools/profile_round.sh TAG, tools/pmc_valu.sh TAG and a bench.py run into gpurun_out/TAG_bench_final.json on the GPU
# box: copies the summaries the judge reads from gpurun_out/ (scratch) into profiles/ (tracked).   usage: tools/copy_profiles.sh r04
set -e
tag=${1:-r04}
root="$(cd "$(dirname "$0")/.." && pwd)"
cd "$root"
d=gpurun_out/prof_$tag
cp $d/${tag}_stats_kernel_stats.csv profiles/${tag}_kernel_stats.csv
cp $d/${tag}_stats1_kernel_stats.csv profiles/${tag}_kernel_stats_one_stream_all_lines.csv
cp $d/${tag}_general_kernel_stats.csv profiles/${tag}_general_kernel_stats.csv
cp $d/${tag}_homom_kernel_stats.csv profiles/${tag}_homomrlwr_kernel_stats.csv
for f in general_index homomrlwr_pipeline homomrlwr_pipeline_1024 homomrlwr_pipeline_1024_1lane tunnel_base2 config2 crt_half extra; do cp $d/${tag}_$f.jsonl profiles/; done
cp $d/traffic.json profiles/${tag}_traffic_pmc.json
cp $d/traffic.json profiles/traffic_latest.json
cat $d/commands.txt gpurun_out/pmc_$tag/commands.txt > profiles/${tag}_commands.txt
cp gpurun_out/pmc_$tag/${tag}_pmc_valu.json profiles/${tag}_pmc_valu.json
cp gpurun_out/${tag}_bench_final.json profiles/${tag}_bench.json
python3 - "$root" "$tag" <<'PY'
import json, os, sys
root, tag = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
import bench
t = json.load(open(os.path.join(root, "profiles", "traffic_latest.json")))
p = json.load(open(os.path.join(root, "profiles", f"{tag}_pmc_valu.json")))
print("sha now", bench.kernel_src_sha16(), "traffic", t["kernel_src_sha16"], "pmc", p["kernel_src_sha16"])
d = json.load(open(os.path.join(root, "profiles", f"{tag}_bench.json")))
print(d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_source"])
PY
