#!/usr/bin/env python3
"""Developer probe: does the warm mul! time of one operator depend on where hipMalloc places it?
Re-creates the same handle several times in one process and times each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bsm_amd as bsm
S = bsm.synthetic
which = sys.argv[1] if len(sys.argv) > 1 else "c5s"
prob = {"c5s": lambda: S.config5(n=600_000), "c3": lambda: S.config3(), "c4s": lambda: S.config4(row_lo=0, row_hi=1953)}[which]()
x = torch.from_numpy(prob["x"]).cuda()
keep = []
for trial in range(6):
    A = S.build(prob)
    y = torch.zeros(prob["size"][0], dtype=x.dtype, device="cuda")
    plan = bsm.MulPlan(y, A, x)
    for _ in range(5):
        plan()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for rep in range(3):
        a.record()
        for _ in range(20):
            plan()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / 20)
    print(f"{which} trial {trial}: " + " ".join(f"{t:.1f}us" for t in ts), flush=True)
    if trial % 2 == 0:
        keep.append(A)  # hold some handles so the next allocation lands elsewhere
        pad = torch.empty((trial + 1) * 37_000_000, dtype=torch.uint8, device="cuda")
        keep.append(pad)
