#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc passes (one directory per pass) as a small table.
usage: pmc_table.py <kernel substring> <dir> [<dir> ...]"""
import csv, glob, os, sys
ksub = sys.argv[1]
vals = {}
for d in sys.argv[2:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if ksub in row["Kernel_Name"]:
                    vals.setdefault((row["Kernel_Name"].split("(")[0][:70], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if ksub in row["Kernel_Name"]:
                    vals.setdefault((row["Kernel_Name"].split("(")[0][:70], "duration_us"), []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for (k, c), v in sorted(vals.items()):
    print(f"{k:72s} {c:28s} n={len(v):4d} mean {sum(v)/len(v):16.1f}")
