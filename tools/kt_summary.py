#!/usr/bin/env python3
"""Per (kernel, grid size) summary of a rocprofv3 --kernel-trace CSV: the bench launches the same
kernel template on several operators (C2 + the HBM-resident legs), which `--stats` averages together.
usage: tools/kt_summary.py <dir with *_kernel_trace.csv> <out.csv>"""
import csv
import glob
import os
import sys

rows = {}
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0]
            key = (name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))
            rows.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(sys.argv[2], "w") as out:
    out.write("kernel,workgroups,calls,avg_ns,median_ns,min_ns,max_ns\n")
    for (name, wgs), d in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        d.sort()
        out.write(f"\"{name}\",{wgs},{len(d)},{sum(d) / len(d):.1f},{d[len(d) // 2]},{d[0]},{d[-1]}\n")
print(open(sys.argv[2]).read())
