#!/usr/bin/env python3
"""Calibration: what does a plain streaming read of N bytes cost on this GPU (warm / cold)?"""
import sys
import torch

def timeit(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / reps

for mb in (54, 216, 924):
    n = mb * 1000 * 1000 // 8
    v = torch.randn(n, dtype=torch.float64, device="cuda")
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    t = timeit(lambda: v.sum())
    print(f"sum   {mb} MB: {t*1e6:.2f} us  {mb/1e3/t:.0f} GB/s")
    t = timeit(lambda: torch.mul(v, 2.0, out=out))
    print(f"scale {mb} MB (r+w {2*mb} MB): {t*1e6:.2f} us  {2*mb/1e3/t:.0f} GB/s")
    vf = v.view(torch.float32)
    t = timeit(lambda: vf.sum())
    print(f"sumf32 {mb} MB: {t*1e6:.2f} us  {mb/1e3/t:.0f} GB/s")
