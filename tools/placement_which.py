#!/usr/bin/env python3
"""Developer probe: WHICH allocation carries the placement effect of a long fused launch (C5 slice 533 / 578 us with the
same schedule)?  One handle with several y / x allocations, then one y with several handles.
usage: placement_which.py [c5s|c2x20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bsm_amd as bsm
S = bsm.synthetic
which = sys.argv[1] if len(sys.argv) > 1 else "c5s"
p = {"c5s": lambda: S.config5(n=600_000, on_device=True),
     "c2x20": lambda: S.config2(n=2_000_000, nblocks=100_000, on_device=True)}[which]()
x0 = p["x"]


def t_of(plan):
    for _ in range(10):
        plan()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(60):
        plan()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / 60


keep = []
A = S.build(p)
for k in range(6):
    y = torch.zeros_like(x0)
    keep.append(torch.empty((k + 1) * 3_700_000 + 4096 * k, dtype=torch.uint8, device="cuda"))
    print(f"same handle, y #{k} at {y.data_ptr():#x}: {t_of(bsm.MulPlan(y, A, x0)):.1f} us", flush=True)
    keep.append(y)
y = keep[1]
for k in range(6):
    x = x0.clone()
    keep.append(torch.empty((k + 1) * 3_700_000 + 4096 * k, dtype=torch.uint8, device="cuda"))
    print(f"same handle, same y, x #{k} at {x.data_ptr():#x}: {t_of(bsm.MulPlan(y, A, x)):.1f} us", flush=True)
    keep.append(x)
for k in range(5):
    B = S.build(p)
    keep.append(torch.empty((k + 1) * 37_000_000 + 4096 * k, dtype=torch.uint8, device="cuda"))
    print(f"same y and x, handle #{k}: {t_of(bsm.MulPlan(y, B, x0)):.1f} us", flush=True)
    keep.append(B)
