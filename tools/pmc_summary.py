#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

def main(paths):
    acc = defaultdict(lambda: defaultdict(list))
    for pat in paths:
        for path in glob.glob(pat):
            with open(path) as f:
                for row in csv.DictReader(f):
                    name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, ctrs in sorted(acc.items()):
        print(name)
        for c, vals in sorted(ctrs.items()):
            print(f"    {c:28s} mean {sum(vals)/len(vals):16.1f}  n={len(vals)}")

if __name__ == "__main__":
    main(sys.argv[1:])
