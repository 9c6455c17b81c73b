#!/usr/bin/env python3
"""Developer probe: cost of mul! with HOST vectors (BSM_MEM_HOST: what a Julia caller with plain
Vector{T} gets) vs device-resident vectors."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bsm_amd as bsm
p = bsm.synthetic.config2()
A = bsm.synthetic.build(p)
x = p["x"]; y = np.zeros_like(x)
for _ in range(5):
    bsm.mul(y, A, x)
t0 = time.perf_counter()
for _ in range(200):
    bsm.mul(y, A, x)
t = (time.perf_counter() - t0) / 200
print(f"host-vector mul!: {t*1e6:.1f} us per call ({A.stats()['alg_bytes']/t/1e9:.0f} GB/s algorithmic)")
