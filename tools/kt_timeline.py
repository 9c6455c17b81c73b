#!/usr/bin/env python3
"""Timeline of the LAST product in a rocprofv3 --kernel-trace CSV of tools/dist_trace.py: every kernel of the
last `n` launches with start / end relative to the first of them.  usage: kt_timeline.py <dir> <kernels per product>"""
import csv, glob, os, sys
rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Stream_Id", "?"), r.get("Queue_Id", "?")))
rows.sort()
n = int(sys.argv[2])
last = rows[-n:]
t0 = last[0][0]
for s, e, k, st, q in last:
    print(f"{(s - t0) / 1e3:9.2f} -> {(e - t0) / 1e3:9.2f} us  ({(e - s) / 1e3:7.2f})  stream {st} queue {q}  {k}")
print(f"product span {(max(e for _, e, *_ in last) - t0) / 1e3:.2f} us")
