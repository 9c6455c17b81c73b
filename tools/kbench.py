#!/usr/bin/env python3
"""Kernel micro-benchmark (developer tool): times warm/cold mul! launches of one config under the
current BSM_* environment knobs.  usage: kbench.py [c2|c3|c3s|c4s|c5s] [reps] [T]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bsm_amd as bsm

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
S = bsm.synthetic
prob = {"c2": lambda: S.config2(), "c3": lambda: S.config3(), "c3s": lambda: S.config3(nseg=800),
        "c4s": lambda: S.config4(row_lo=0, row_hi=1953),
        "c2t": lambda: S.config2(n=6000, lo=1, hi=1, nblocks=14000),
        "c2q": lambda: S.config2(n=30000, nblocks=1500),
        "c2h": lambda: S.config2(n=50000, nblocks=2500),
        "c2u": lambda: S.config2(n=100000, lo=36, hi=36, nblocks=5000),
        "c2p": lambda: S.config2(n=100000, lo=32, hi=32, nblocks=6400),
        "c2w": lambda: S.config2(n=100000, lo=64, hi=64, nblocks=1650), "c5s": lambda: S.config5(n=600_000),
        "c2x20": lambda: S.config2(n=2_000_000, nblocks=100_000),
        "c5u64": lambda: S.config5(n=600_000, lo=64, hi=64),
        "c5u128": lambda: S.config5(n=600_000, lo=128, hi=128),
        "c5u40": lambda: S.config5(n=300_000, lo=40, hi=40),
        "c5u100": lambda: S.config5(n=600_000, lo=100, hi=100),
        "c5m": lambda: S.config5(n=600_000, lo=65, hi=127),
        "c5h": lambda: S.config5(n=2_500_000),
        "bem": lambda: S.config5(n=1_500_000, lo=8, hi=28, halfband=8)}[which]()
kw = {"accumulate": os.environ.get("KB_ACC", "auto")}
if os.environ.get("KB_TIMG") and prob["kind"] != "symmetric":
    kw["transpose_image"] = True
if os.environ.get("KB_COMPLEX"):
    # complex variant of the same structure (BEM operators are ComplexF64 in the reference's fixtures)
    cdt = np.complex64 if prob["x"].dtype == np.float32 else np.complex128
    for key in ("blocks", "diagonals", "offdiagonals"):
        if key in prob:
            prob[key] = [np.asfortranarray(b + 1j * b[::-1, ::-1]).astype(cdt) for b in prob[key]]
    if prob["kind"] == "symmetric":
        prob["diagonals"] = [np.asfortranarray((d + d.T) / 2) for d in prob["diagonals"]]
    prob["x"] = (prob["x"] + 1j * prob["x"][::-1]).astype(cdt)
if os.environ.get("KB_AS_BSM") and prob["kind"] == "symmetric":
    # same bytes as a forward-only BlockSparseMatrix (isolates the cost of the transposed half)
    prob = dict(kind="blocksparse", blocks=prob["diagonals"] + prob["offdiagonals"],
                rowindices=prob["diagonalindices"] + prob["rowindices"],
                colindices=prob["diagonalindices"] + prob["colindices"], size=prob["size"], x=prob["x"])
A = S.build(prob, **kw)
st = A.stats()
x = torch.from_numpy(prob["x"]).cuda()
ops = [("N", A)]
if len(sys.argv) > 3 and sys.argv[3] == "T":
    ops.append(("T", bsm.transpose(A)))
for name, Aop in ops:
    y = torch.zeros(bsm.size(Aop)[0], dtype=x.dtype, device="cuda")
    plan = bsm.MulPlan(y, Aop, x)
    for _ in range(20):
        plan()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        plan()
    b.record()
    torch.cuda.synchronize()
    warm = a.elapsed_time(b) * 1e-3 / reps
    flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    cold = []
    for _ in range(15):
        flush.fill_(1)
        a.record()
        plan()
        b.record()
        torch.cuda.synchronize()
        cold.append(a.elapsed_time(b) * 1e-3)
    cold.sort()
    del flush
    nrhs = int(os.environ.get("KB_NRHS", "0"))
    if nrhs > 1:
        n_in, n_out = bsm.size(Aop)[1], bsm.size(Aop)[0]
        X = torch.randn((nrhs, n_in), dtype=x.dtype, device="cuda").t()
        Y = torch.zeros((nrhs, n_out), dtype=x.dtype, device="cuda").t()
        for _ in range(5):
            bsm.mul(Y, Aop, X)
        torch.cuda.synchronize()
        a.record()
        for _ in range(max(reps // 4, 5)):
            bsm.mul(Y, Aop, X)
        b.record()
        torch.cuda.synchronize()
        tm = a.elapsed_time(b) * 1e-3 / max(reps // 4, 5)
        print(f"   multi-RHS k={nrhs}: {tm * 1e6:.2f}us per call = {tm / nrhs * 1e6:.2f}us per column "
              f"(single-RHS warm {warm * 1e6:.2f}us) -> {warm * nrhs / tm:.2f}x", flush=True)
    knobs = {k: v for k, v in os.environ.items() if k.startswith("BSM_")}
    print(f"{which} op={name} {knobs} wgs={st['nworkgroups']} tasks={st['ntasks']} excl={st['exclusive']} "
          f"warm={warm * 1e6:.2f}us ({st['alg_bytes'] / warm / 1e9:.0f} GB/s) "
          f"cold={cold[len(cold) // 2] * 1e6:.2f}us ({st['alg_bytes'] / cold[len(cold) // 2] / 1e9:.0f} GB/s)",
          flush=True)
