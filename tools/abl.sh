cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for g in 256 512 1024 4096 16384; do
ALCH_KS_GRID=$g ALCH_CHUNK=2048 ALCH_EXP_FLAGS=3584 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ablg$g -- python3 tools/kprobe.py > gpurun_out/ablg$g.log 2>&1
python3 - <<PY
import csv,glob
for p in glob.glob("gpurun_out/ablg$g/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(p)):
        if "ks_accum" in r["Name"]: print("grid $g", r["Calls"], r["AverageNs"])
PY
done
