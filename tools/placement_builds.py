#!/usr/bin/env python3
"""Developer probe: is the 10 % process-to-process spread of long launches (same schedule: C5 slice 542 / 593 us,
1 GB VBCRS 167 / 184 us) a property of where the allocation landed?  Builds the same operator several times in ONE
process (earlier handles and some padding kept alive, so every build gets other addresses) and times each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bsm_amd as bsm
S = bsm.synthetic
which = sys.argv[1] if len(sys.argv) > 1 else "c2x20"
p = {"c2x20": lambda: S.config2(n=2_000_000, nblocks=100_000, on_device=True),
     "c5s": lambda: S.config5(n=600_000, on_device=True)}[which]()
keep = []
x = p["x"] if torch.is_tensor(p["x"]) else torch.from_numpy(p["x"]).cuda()
for trial in range(6):
    A = S.build(p)
    y = torch.zeros_like(x)
    plan = bsm.MulPlan(y, A, x)
    for _ in range(10):
        plan()
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            plan()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / 50)
    print(f"{which} build {trial}: " + " ".join(f"{t:.1f}" for t in ts) + " us", flush=True)
    keep.append((A, y, plan, torch.empty((trial + 1) * 37_000_000 + 4096 * trial, dtype=torch.uint8, device="cuda")))
