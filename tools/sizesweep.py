#!/usr/bin/env python3
"""Developer probe: warm rate of the VBCRS forward product vs operator size (C2-shaped operators
from 14 MB to 1.7 GB): launch-latency-bound -> Infinity-Cache-resident -> HBM-streaming."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bsm_amd as bsm
S = bsm.synthetic
for k in (0.25, 0.5, 1, 2, 3, 4, 6, 8, 16, 32):
    p = S.config2(n=int(100000 * k), nblocks=int(5000 * k))
    A = S.build(p)
    st = A.stats()
    x = torch.from_numpy(p["x"]).cuda()
    y = torch.zeros_like(x)
    plan = bsm.MulPlan(y, A, x)
    for _ in range(20):
        plan()
    torch.cuda.synchronize()
    reps = 400 if k <= 4 else 100
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        plan()
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) * 1e-3 / reps
    print(f"{st['alg_bytes'] / 1e6:8.1f} MB  {t * 1e6:8.2f} us  {st['alg_bytes'] / t / 1e9:6.0f} GB/s  wgs {st['nworkgroups']}", flush=True)
    del A, p, plan
