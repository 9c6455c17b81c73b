#!/usr/bin/env python3
"""Developer probe: K right-hand sides through the TRANSPOSED product on the single image (op T: all atomics), with and without
the interleaved pass (BSM_MULTI_IL=0 / 2).  usage: multirhs_t.py [c2x20|c4s|c2 ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bsm_amd as bsm
S = bsm.synthetic


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / reps)
    return sorted(ts)[1]


CASES = {"c2": lambda: S.config2(on_device=True), "c2x20": lambda: S.config2(n=2_000_000, nblocks=100_000, on_device=True),
         "c4s": lambda: S.config4(on_device=True, row_lo=0, row_hi=1953)}
for name in sys.argv[1:] or list(CASES):
    prob = CASES[name]()
    A = bsm.transpose(S.build(prob))
    x = prob["x"]
    n = x.shape[0]
    y = torch.zeros_like(x)
    plan = bsm.MulPlan(y, A, x)
    reps = 100 if name == "c2" else 15
    t1 = timed(plan, reps)
    line = f"{name:6s} op T 1 rhs {t1:8.1f} us"
    for K in (8, 16):
        X = torch.empty((K, n), dtype=x.dtype, device="cuda").t()
        for k in range(K):
            X[:, k] = x * (k + 1)
        Y = torch.zeros((K, n), dtype=x.dtype, device="cuda").t()
        tk = timed(lambda: bsm.mul(Y, A, X), reps)
        line += f"   {K} rhs {tk:8.1f} us = {tk / t1:4.2f}"
    print(line, flush=True)
