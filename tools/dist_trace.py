#!/usr/bin/env python3
"""Developer probe for rocprofv3 --kernel-trace: N products through a multi-device handle over virtual parts.
usage: dist_trace.py <c3|c5s> <nparts> [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bsm_amd as bsm
S = bsm.synthetic
which, nparts = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
prob = {"c3": lambda: S.config3(on_device=True), "c5s": lambda: S.config5(n=625_000, on_device=True)}[which]()
A = S.build(prob, **({"devices": [0] * nparts} if nparts else {}))
x = prob["x"]
y = torch.zeros_like(x)
for _ in range(reps):
    bsm.mul(y, A, x)
torch.cuda.synchronize()
