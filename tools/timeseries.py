#!/usr/bin/env python3
"""Developer probe: the same plan timed in consecutive batches for a while -- is a bimodal time a matter of where the
allocations landed (constant for one handle) or of the clocks (drifts in time)?  usage: timeseries.py [c2x20|c3] [batches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bsm_amd as bsm
S = bsm.synthetic
which = sys.argv[1] if len(sys.argv) > 1 else "c2x20"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 40
p = {"c2x20": lambda: S.config2(n=2_000_000, nblocks=100_000, on_device=True), "c3": lambda: S.config3(on_device=True)}[which]()
A = S.build(p)
x = p["x"]
y = torch.zeros_like(x)
plan = bsm.MulPlan(y, A, x)
out = []
t0 = time.time()
for b in range(nb):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100):
        plan()
    e.record()
    torch.cuda.synchronize()
    out.append((time.time() - t0, a.elapsed_time(e) * 10))
    if b % 10 == 9:
        time.sleep(0.5)  # an idle gap: do the clocks come back different?
print(which, " ".join(f"{t:.1f}s:{us:.1f}" for t, us in out))
