#!/usr/bin/env python3
"""Per-kernel SQ counter totals of tools/pmc_valu.sh (rocprofv3 --kernel-trace --pmc, one pass per workload) with the kernel
durations of the same pass, and the two ratios the roofline statements use:
  valu_issue_frac   SQ_INSTS_VALU / (duration x 1024 SIMDs x clock / 4): wave-instructions issued against the chip's integer issue
                    ceiling (one VALU instruction per SIMD every 4 cycles; clock 2.4 GHz nominal -- the kernels run at 2.2-2.4)
  lds_conflict_frac SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE: extra LDS cycles over all LDS-array cycles
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md).
usage: pmc_valu_summary.py DIR"""
import csv, glob, json, os, sys, time
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_src_sha16

d = sys.argv[1]
CLOCK_HZ, SIMDS = 2.4e9, 1024


def short(name):
    return name.split("(")[0].replace("void ", "").replace("alch::", "")


out = {"collected": time.strftime("%Y-%m-%d"), "kernel_src_sha16": kernel_src_sha16(),
       "commands": open(os.path.join(d, "commands.txt")).read().splitlines() if os.path.exists(os.path.join(d, "commands.txt")) else [],
       "note": "counter totals over all dispatches of the kernel in the pass; durations from the kernel trace of the same pass (counter "
               "collection serialises dispatches and slows them: use the ratios, not the rates); clock assumed 2.4 GHz", "workloads": {}}
for wl in ("headline", "full_mul", "general", "homomrlwr", "tunnel_hs"):
    ctr = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(int)
    for path in glob.glob(os.path.join(d, "**", f"{wl}_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = short(row["Kernel_Name"])
            ctr[k][row["Counter_Name"]] += float(row["Counter_Value"])
    dur = defaultdict(float)
    for path in glob.glob(os.path.join(d, "**", f"{wl}_kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = short(row["Kernel_Name"])
            dur[k] += (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-9
            disp[k] += 1
    ks = {}
    total = sum(dur.values()) or 1.0
    for k in sorted(ctr, key=lambda k: -dur.get(k, 0)):
        c = ctr[k]
        if dur.get(k, 0) / total < 0.01:
            continue
        rec = {"dispatches": disp[k], "duration_ms": dur[k] * 1e3, "share_of_gpu_time": dur[k] / total, **{n: v for n, v in c.items()}}
        if dur[k] > 0 and "SQ_INSTS_VALU" in c:
            rec["valu_issue_frac"] = c["SQ_INSTS_VALU"] / (dur[k] * SIMDS * CLOCK_HZ / 4)
        if c.get("SQ_WAVE_CYCLES"):
            rec["valu_active_per_wave_cycle"] = c.get("SQ_ACTIVE_INST_VALU", 0) / c["SQ_WAVE_CYCLES"]
            rec["wait_any_per_wave_cycle"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
            rec["wait_inst_any_per_wave_cycle"] = c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
        if c.get("SQ_LDS_IDX_ACTIVE"):
            rec["lds_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]
        ks[k] = rec
    out["workloads"][wl] = ks
print(json.dumps(out, indent=1))
