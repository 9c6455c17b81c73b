#!/usr/bin/env python3
"""Developer probe: wall time of the *_create calls (host analysis + packing + upload) per config.
usage: createtime.py [device]   (device -2 = host analysis only)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bsm_amd as bsm
S = bsm.synthetic
dev = int(sys.argv[1]) if len(sys.argv) > 1 else -1
for name, mk in (("c2", S.config2), ("c3", S.config3), ("c4s", lambda: S.config4(row_lo=0, row_hi=1953)),
                 ("c5s", lambda: S.config5(n=625000))):
    p = mk()
    for rep in range(2):
        t = time.perf_counter()
        A = S.build(p, device=dev)
        dt = time.perf_counter() - t
        st = A.stats()
        print(name, "create %.3f s  %.1f MB  -> %.2f GB/s" % (dt, st['alg_bytes'] / 1e6, st['alg_bytes'] / dt / 1e9), flush=True)
        del A
    del p
