cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for g in 0 -1; do
ALCH_TI_GRID=$g ALCH_CHUNK=2048 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ablA$g -- python3 tools/kprobe.py > gpurun_out/ablA$g.log 2>&1
tail -1 gpurun_out/ablA$g.log
python3 - <<PY
import csv,glob
for p in glob.glob("gpurun_out/ablA$g/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(p)):
        if "alch::k_" in r["Name"]: print("TI_GRID $g", r["Name"][11:40], r["Calls"], r["AverageNs"])
PY
done
