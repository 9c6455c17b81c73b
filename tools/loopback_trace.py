#!/usr/bin/env python3
"""The N > 1 step of bench.py (C5 share of one rank of eight, overlapped exchange) as a ONE-rank RCCL loopback, for a
kernel-trace timeline:  rocprofv3 --kernel-trace -d gpurun_out/lb -o lb --output-format csv -- python3 tools/loopback_trace.py
then tools/kt_timeline.py gpurun_out/lb <kernels per step>.  Prints the event-timed step and the local launches alone."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import bsm_amd as bsm
from bsm_amd import distributed as D

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.125
backend = sys.argv[2] if len(sys.argv) > 2 else "nccl"
torch.cuda.set_device(0)
dist = bench.init_one_rank(backend, torch, 0)
S = bsm.synthetic
n = int(5_000_000 * scale)
start, sz = S.config5_segments(n=n)
nseg = len(sz)
prob = S.config5(n=n, on_device=True, seg_lo=0, seg_hi=nseg)
edge = 8
own = (int(start[edge]) + 1, int(start[nseg - edge]))
P = D.build_overlapped(prob, own, symmetric=True, xmode="halo", loopback=own)
x = prob["x"]
y = torch.full((n,), float("nan"), dtype=x.dtype, device="cuda")
reserve = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if reserve:
    # experiment: the caller's compute stream with a CU mask that leaves `reserve` CUs to the collective layer's kernels
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (ncu + 31) // 32
    bits = [1] * ncu
    stride = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    for i in range(reserve):
        bits[(i * stride) % ncu + (i * stride) // ncu] = 0
    mask = (C.c_uint32 * words)(*[sum(bits[w * 32 + b] << b for b in range(32) if w * 32 + b < ncu) for w in range(words)])
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(words), mask)
    assert rc == 0, rc
    ext = torch.cuda.ExternalStream(st.value)
    torch.cuda.synchronize()
    torch.cuda.set_stream(ext)
    print("compute stream with %d of %d CUs, reserved bits stride %d" % (ncu - reserve, ncu, stride), flush=True)
# per-step device time of the FIRST steps (events between the steps, no host sync): how many steps until steady state
evs = [torch.cuda.Event(enable_timing=True) for _ in range(31)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(30):
    evs[i].record()
    P.mul_overlapped(y, x)
evs[30].record()
torch.cuda.synchronize()
print("first 30 steps: wall %.1f us per step; device us per step: %s" % ((time.perf_counter() - t0) / 30 * 1e6,
      " ".join("%.0f" % (evs[i].elapsed_time(evs[i + 1]) * 1e3) for i in range(30))), flush=True)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
t0 = time.perf_counter()
a.record()
for _ in range(reps):
    P.mul_overlapped(y, x)
b.record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / reps
step = a.elapsed_time(b) * 1e-3 / reps
plans = [bsm.MulPlan(torch.zeros_like(y), h, x) for h in (P.interior, P.local) if h is not None]
def local():
    for pl in plans:
        pl()
for _ in range(5):
    local()
torch.cuda.synchronize()
k = bench.timed(local, reps, torch)
# host-side issue time of one step (no device wait)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    P.mul_overlapped(y, x)
issue = (time.perf_counter() - t0) / 20
torch.cuda.synchronize()
print("rows %d (scale %g), backend %s: step %.1f us (wall %.1f), local launches alone %.1f us, exchange %.1f us, host issue %.1f us"
      % (n, scale, backend, step * 1e6, wall * 1e6, k * 1e6, (step - k) * 1e6, issue * 1e6), flush=True)
# marker: three more steps, the LAST kernels of the trace
torch.cuda.synchronize()
for _ in range(3):
    P.mul_overlapped(y, x)
torch.cuda.synchronize()
dist.destroy_process_group()
