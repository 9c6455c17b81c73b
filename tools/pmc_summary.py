#!/usr/bin/env python3
"""Condenses rocprofv3 --pmc passes into the JSON files bench.py reads (profiles/r02_c2_pmc.json,
profiles/r02_c4_mfma.json).  One counter per pass, --kernel-trace only (MI355X_MICROARCH.md, rocprofv3
PMC slots / HBM): FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts the 128-B
requests of 16-B-per-lane streaming reads as 64 B, so it is doubled before it is compared with a
byte count.

    tools/pmc_summary.py traffic <out.json> <kernel substring> <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> [alg_bytes]
    tools/pmc_summary.py mfma    <out.json> <kernel substring> <dir with the MFMA counter pass>
"""
import csv
import glob
import json
import os
import sys


def read_pass(directory, kernel_sub):
    """{counter: [value per dispatch]} for kernels whose name contains kernel_sub"""
    out = {}
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if kernel_sub in row["Kernel_Name"]:
                    out.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return out


def mean(v):
    return sum(v) / len(v) if v else None


def main():
    mode, out_path, ksub = sys.argv[1], sys.argv[2], sys.argv[3]
    if mode == "traffic":
        fetch = read_pass(sys.argv[4], ksub).get("FETCH_SIZE", [])
        write = read_pass(sys.argv[5], ksub).get("WRITE_SIZE", [])
        alg = int(sys.argv[6]) if len(sys.argv) > 6 else None
        f, w = mean(fetch), mean(write)
        traffic = int(f * 1024 * 2 + w * 1024)
        res = {"kernel": ksub, "fetch_size_kib_mean": round(f, 2), "write_size_kib_mean": round(w, 2),
               "dispatches": [len(fetch), len(write)],
               "traffic_bytes_per_launch": traffic,
               "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); "
                         "traffic = FETCH_SIZE KiB x 1024 x 2 (gfx950: 128-B requests tallied at 64 B) + WRITE_SIZE KiB x 1024"}
        if alg:
            res["alg_bytes_per_launch"] = alg
            res["traffic_over_algorithmic"] = round(traffic / alg, 4)
    else:
        vals = read_pass(sys.argv[4], ksub)
        res = {"kernel": ksub, "counters": {k: {"dispatches": len(v), "sum": sum(v)} for k, v in sorted(vals.items())},
               "note": "MFMA instruction / busy counters of the C4 product (128x128 fp32 blocks): expected 0 -- a single "
                       "right-hand side fills 1/16 of an MFMA tile and v_mfma_f32_*_f32 runs at the VALU FMA rate on "
                       "gfx950 (MI355X_MICROARCH.md, matrix cores), so the HBM-bound panel kernel uses v_fma only"}
    # the build of the kernels the counters were taken with (bench.py refuses a file of another build)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bsm_amd import _lib
    res["build"] = _lib.lib().bsm_version().decode().split("build ")[-1]
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1)
        fh.write("\n")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
