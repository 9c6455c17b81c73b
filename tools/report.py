#!/usr/bin/env python3
"""Fills BASELINE.md's result table: for every BASELINE.json config (C4/C5 as the slice one GPU
of eight owns) the CPU oracle rate (1 thread, scalar port of the reference loops), the GPU rate
through the C ABI (warm, back-to-back launches; HIP events) and the parity error.
usage: tools/report.py [out.md]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import bsm_amd as bsm
from oracle import load_oracle
from oracle.oracle import load_oracle_native
from _common import N, T, oracle_mul, relerr

S = bsm.synthetic
orc = load_oracle()               # parity checks: the -O2 -fno-fast-math build
orc_fast = load_oracle_native()   # CPU rate: -O3 -march=native, every argument marshalled ONCE
CONFIGS = [
    ("C1 BlockSparseMatrix 1000^2, 50x 32x32 fp64", lambda: S.config1(), 1),
    ("C2 VBCRS 100k^2, 5000 blocks 8-64 fp64", lambda: S.config2(), 1),
    ("C3 Symmetric 200k^2, 64x64 fp64, half-bandwidth 8", lambda: S.config3(), 1),
    ("C4 VBCRS 2M^2, 128x128 fp32: block rows 0..1952 (1/8)", lambda: S.config4(row_lo=0, row_hi=1953), 8),
    ("C5 Symmetric 5M^2, sizes 16-256 fp64: first 625k rows (1/8)", lambda: S.config5(n=625_000), 8),
]
if "--full" in sys.argv:  # the complete 8-GPU configs on ONE MI355X (16.4 GB / 28.7 GB of HBM)
    sys.argv.remove("--full")
    # (host generation: minutes per config and 45 GB of host memory.  `bench.py --gpus 1 --workload c5`
    # times the same two operators generated in HBM in seconds -- without the oracle comparison)
    CONFIGS = [
        ("C4 VBCRS 2M^2, 250000x 128x128 fp32, FULL on one GPU", lambda: S.config4(), 1),
        ("C5 Symmetric 5M^2, sizes 16-256 fp64, FULL on one GPU", lambda: S.config5(), 1),
    ]
lines = ["| config | CPU port 1 core GB/s (-O3 -march=native, pre-marshalled) | GPU N GB/s (% of 8 TB/s) | GPU T GB/s | rel-err N | rel-err T | alg MB | A*X, 4 / 8 / 16 columns (single products; worst column vs single products; 16 real columns = one matrix-pipe pass) |",
         "|---|---|---|---|---|---|---|---|"]
for name, make, share in CONFIGS:
    prob = make()
    A = S.build(prob)
    st = A.stats()
    dt = A.dtype
    nr, nc = prob["size"]
    x = prob["x"]
    y0 = np.zeros(nr, dtype=dt)
    # CPU oracle (one host core), bounded sample; the timed loop is a pre-marshalled C call
    ref = oracle_mul(orc, prob, N, x, y0)
    call = oracle_mul(orc_fast, prob, N, x, y0, prepare=True)
    call()
    reps, t0 = 0, time.perf_counter()
    while True:
        call()
        reps += 1
        el = time.perf_counter() - t0
        if el > 4.0 or reps >= 500:
            break
    cpu = st["alg_bytes"] * reps / el / 1e9
    out = {}
    for opname, Aop, op in (("N", A, N), ("T", bsm.transpose(A), T)):
        xd = torch.from_numpy(x).cuda()
        yd = torch.zeros(nr, dtype=xd.dtype, device="cuda")
        plan = bsm.MulPlan(yd, Aop, xd)
        for _ in range(10):
            plan()
        torch.cuda.synchronize()
        r = 300 if st["alg_bytes"] < 2e8 else 40
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(r):
            plan()
        b.record()
        torch.cuda.synchronize()
        tt = a.elapsed_time(b) * 1e-3 / r
        refop = ref if op == N else oracle_mul(orc, prob, T, x, y0)
        tol_ref = relerr(yd.cpu().numpy(), refop)
        out[opname] = (st["alg_bytes"] / tt / 1e9, tol_ref, tt)
    g, e, tt = out["N"]
    gt, et, _ = out["T"]
    # A * X (bsm_mul_multi): cost in single products, parity of every column against its own single product
    multi = []
    xd = torch.from_numpy(x).cuda()
    y1 = torch.zeros(nr, dtype=xd.dtype, device="cuda")
    for K in (4, 8, 16):
        X = torch.empty((K, nc), dtype=xd.dtype, device="cuda").t()
        for k in range(K):
            X[:, k] = xd * (k + 1) / K
        Y = torch.zeros((K, nr), dtype=xd.dtype, device="cuda").t()
        for _ in range(30):  # (the first tens of launches of a kernel on a fresh operator run 5-8 % slower)
            bsm.mul(Y, A, X)
        torch.cuda.synchronize()
        r = 100 if st["alg_bytes"] < 2e8 else 15
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(r):
            bsm.mul(Y, A, X)
        b.record()
        torch.cuda.synchronize()
        tk = a.elapsed_time(b) * 1e-3 / r
        worst = 0.0
        for k in range(K):
            bsm.mul(y1, A, X[:, k].contiguous())
            worst = max(worst, float((Y[:, k] - y1).abs().max() / y1.abs().max()))
        multi.append((tk / tt, worst))
        del X, Y
    lines.append(f"| {name} | {cpu:.2f} | {g:.0f} ({100 * g / 8000:.0f} %, {tt * 1e6:.1f} us) | {gt:.0f} | "
                 f"{e:.1e} | {et:.1e} | {st['alg_bytes'] / 1e6:.1f} | {multi[0][0]:.2f} / {multi[1][0]:.2f} / {multi[2][0]:.2f} "
                 f"({max(m[1] for m in multi):.0e}) |")
    print(lines[-1], flush=True)
    del A, prob
text = "\n".join(lines)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(text + "\n")
print(text)
