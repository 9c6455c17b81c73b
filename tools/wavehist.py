#!/usr/bin/env python3
"""Developer probe: distribution of the per-wave work of a packed image (bytes, iterations of the
inner loop = dependent memory round trips) -- the serial depth that bounds a single-round launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, bsm_amd as bsm
from _common import get_image
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
S = bsm.synthetic
p = {"c2": S.config2, "c2h": lambda: S.config2(n=50000, nblocks=2500),
     "c2u": lambda: S.config2(n=100000, lo=36, hi=36, nblocks=5000),
     "c2p": lambda: S.config2(n=100000, lo=32, hi=32, nblocks=6400),
     "c2w": lambda: S.config2(n=100000, lo=64, hi=64, nblocks=1650)}[which]()
A = S.build(p, device=-2)
w = get_image(A)[3]
f = w["first"]
m = w["m"].astype(np.int64); ns = f["nstrips"].astype(np.int64)
P = np.where(m <= 8, 8, np.where(m <= 16, 16, np.where(m <= 32, 32, 64)))
G = 64 // P
es = 16
iters = np.ceil(ns / (G * 8)).astype(int)
nbytes = ns * m * es
act = (w["work"] == 1) & (w["npieces"] > 0)
print("waves", len(w), "panel waves", int(act.sum()), "grp hist", np.bincount(w["grp"][act & (w["lead"] > 0)]))
print("bytes per wave: mean %.0f max %d  p50 %d p90 %d p99 %d" % (nbytes[act].mean(), nbytes[act].max(), *np.percentile(nbytes[act], [50, 90, 99])))
print("iterations per wave (hist from 0):", np.bincount(iters[act]))
print("bytes in waves by iteration count:", [int(nbytes[act & (iters == k)].sum() // 1000) for k in range(iters[act].max() + 1)], "KB")
print("mean m/P:", float((m[act] / P[act]).mean()))
