#!/usr/bin/env python3
"""Developer probe: does a BARE streaming read see the allocation-placement effect (the same product runs 533 or 578 us
depending on where its value stream landed)?  Allocates several multi-GB buffers and streams each (bsm_bench_stream)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bsm_amd import _lib
L = _lib.lib()
nbytes = int(float(sys.argv[1]) * 1e9) if len(sys.argv) > 1 else 3_450_000_000
nbytes = nbytes // 16 * 16
scratch = torch.zeros(8192 // 8 + 16, dtype=torch.float64, device="cuda")
bufs = []
for k in range(8):
    b = torch.empty(nbytes // 8, dtype=torch.float64, device="cuda")
    b.fill_(1.0)
    bufs.append(b)
    pad = torch.empty((k + 1) * 37_000_000, dtype=torch.uint8, device="cuda")
    bufs.append(pad)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for rnd in range(2):
    for k in range(0, len(bufs), 2):
        b = bufs[k]
        f = lambda: L.bsm_bench_stream(C.c_void_p(b.data_ptr()), nbytes, C.c_void_p(scratch.data_ptr()), scratch.numel() * 8, 0, st)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            f()
        e.record(); torch.cuda.synchronize()
        t = a.elapsed_time(e) * 1e3 / 10
        print(f"round {rnd} buffer {k // 2} at {b.data_ptr():#x}: {t:8.1f} us  {nbytes / t / 1e3:6.0f} GB/s", flush=True)
