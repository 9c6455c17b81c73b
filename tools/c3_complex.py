#!/usr/bin/env python3
"""Developer probe: the C3 structure (64 x 64 blocks, half-bandwidth 8) with COMPLEX entries, K right-hand sides with and without
the interleaved pass (BSM_MULTI_IL=0 / 2): the policy for complex types over tall panels.  usage: c3_complex.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bsm_amd as bsm
S = bsm.synthetic
def timed(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / reps)
    return sorted(ts)[1]
for dt in (np.complex128, np.complex64):
    p = S.config3(nseg=700, dtype=np.float64)
    rng = np.random.default_rng(1)
    for k in ("diagonals", "offdiagonals"):
        p[k] = [np.asfortranarray((b + 1j * rng.standard_normal(b.shape)).astype(dt)) for b in p[k]]
    p["diagonals"] = [np.asfortranarray((d + d.T) / 2) for d in p["diagonals"]]
    A = S.build(p)
    n = p["size"][0]
    x = torch.from_numpy((rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(dt)).cuda()
    y = torch.zeros_like(x)
    plan = bsm.MulPlan(y, A, x)
    t1 = timed(plan, 30)
    line = f"c3-like {np.dtype(dt).name} {A.stats()['alg_bytes']/1e6:.0f} MB: 1 rhs {t1:.1f} us"
    for K in (4, 8):
        X = torch.empty((K, n), dtype=x.dtype, device="cuda").t()
        for k in range(K): X[:, k] = x * (k + 1)
        Y = torch.zeros((K, n), dtype=x.dtype, device="cuda").t()
        tk = timed(lambda: bsm.mul(Y, A, X), 15)
        line += f"   {K} rhs {tk:.1f} us = {tk/t1:.2f}"
    print(line, flush=True)
