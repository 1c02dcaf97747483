#!/usr/bin/env python3
"""HBM-side bytes per op of the two kernels of alch_ct_mul_relin from rocprofv3 --pmc passes (FETCH_SIZE and
WRITE_SIZE collected separately, tools/profile_round.sh).  Units and the gfx950 correction follow
MI355X_MICROARCH.md: both counters are in KiB; FETCH_SIZE is doubled (it reports half of a 16 B/lane coalesced
read stream on gfx950 -- confirmed here on k_tensor_intt, which must read exactly 1 MiB per op).
usage: traffic_summary.py DIR BATCH"""
import csv, glob, json, os, sys, time
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_src_sha16

d, batch = sys.argv[1], int(sys.argv[2])

def per_kernel(prefix, counter):
    tot = defaultdict(float)
    for path in glob.glob(os.path.join(d, "**", f"{prefix}_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == counter:
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                tot[name] += float(row["Counter_Value"])
    return tot

fetch, write = per_kernel("fetch", "FETCH_SIZE"), per_kernel("write", "WRITE_SIZE")
kernels, total = {}, 0.0
for name in sorted(set(fetch) | set(write)):
    if "k_ks_accum" not in name and "k_tensor_intt" not in name:
        continue
    f = 2.0 * fetch.get(name, 0.0) * 1024 / batch
    w = write.get(name, 0.0) * 1024 / batch
    kernels[name] = {"fetch_bytes_per_op_corrected": f, "write_bytes_per_op": w}
    total += f + w
print(json.dumps({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py "
                             "--steps 1 --warmup 0 --batch %d --cpu-ops 0 --no-full --no-pow --no-general" % batch,
                  "batch": batch,
                  "unit_note": "FETCH_SIZE/WRITE_SIZE in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950)",
                  "kernel_src_sha16": kernel_src_sha16(), "collected": time.strftime("%Y-%m-%d"),
                  "kernels": kernels, "hbm_bytes_per_op": total}, indent=1))
