#!/usr/bin/env python3
"""Developer probe: cost of mul! with HOST vectors (BSM_MEM_HOST: what a Julia caller with plain
Vector{T} gets) -- pageable memory (library mirrors) and page-locked memory (bsm_host_register) --
vs device-resident vectors."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bsm_amd as bsm
p = bsm.synthetic.config2()
A = bsm.synthetic.build(p)
ab = A.stats()["alg_bytes"]
x = p["x"].copy(); y = np.zeros_like(x)


def run(label, *args):
    for _ in range(10):
        bsm.mul(y, A, x, *args)
    t0 = time.perf_counter()
    for _ in range(300):
        bsm.mul(y, A, x, *args)
    t = (time.perf_counter() - t0) / 300
    print(f"{label}: {t*1e6:.1f} us per call ({ab/t/1e9:.0f} GB/s algorithmic)", flush=True)


run("pageable host vectors, mul!(y, A, x)       ")
run("pageable host vectors, mul!(y, A, x, a, b) ", 0.5, 2.0)
bsm.host_register(x); bsm.host_register(y)
run("page-locked host vectors, mul!(y, A, x)      ")
run("page-locked host vectors, mul!(y, A, x, a, b)", 0.5, 2.0)
bsm.host_unregister(x); bsm.host_unregister(y)
xd, yd = torch.from_numpy(x).cuda(), torch.zeros(len(x), dtype=torch.float64, device="cuda")
plan = bsm.MulPlan(yd, A, xd)
for _ in range(10):
    plan()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    plan()
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / 300
print(f"device-resident vectors: {t*1e6:.1f} us per call ({ab/t/1e9:.0f} GB/s algorithmic)")
