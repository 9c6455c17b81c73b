#!/usr/bin/env python3
"""Developer probe: cost of the in-library multi-device fan-out (bsm_ctx_t, csrc/bsm_dist.cpp) on ONE GPU
listed several times (virtual devices): the same operator as one ordinary handle and spread over 2 / 4
parts, device-resident and host vectors.  On one GPU the parts share the device, so the kernel time is
the same; what shows is the fan-out itself (events, peer copies, halo adds, delivery)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bsm_amd as bsm
S = bsm.synthetic
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
prob = {"c3": lambda: S.config3(on_device=True), "c2": lambda: S.config2(on_device=True),
        "c5s": lambda: S.config5(n=625_000, on_device=True)}[which]()
x = prob["x"]
xh = x.cpu().numpy()
for devs in (None, [0], [0, 0], [0, 0, 0, 0]):
    A = S.build(prob, **({"devices": devs} if devs else {}))
    st = A.stats()
    y = torch.zeros_like(x)
    yh = np.zeros_like(xh)
    for _ in range(5):
        bsm.mul(y, A, x)
    torch.cuda.synchronize()
    reps = 100
    t0 = time.perf_counter()
    for _ in range(reps):
        bsm.mul(y, A, x)
    torch.cuda.synchronize()
    td = (time.perf_counter() - t0) / reps
    for _ in range(3):
        bsm.mul(yh, A, xh)
    t0 = time.perf_counter()
    for _ in range(30):
        bsm.mul(yh, A, xh)
    th = (time.perf_counter() - t0) / 30
    tp = float("nan")
    if devs:  # partitioned vectors (bsm_mul_parts): every part holds its x / y slice only
        parts = A.parts()
        xp = [x[p["cols"][0] - 1:p["cols"][1]].clone() for p in parts]
        yp = [torch.zeros(max(p["own"][1] - p["own"][0] + 1, 0), dtype=x.dtype, device="cuda") for p in parts]
        for _ in range(5):
            bsm.mul_parts(yp, A, xp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            bsm.mul_parts(yp, A, xp)
        torch.cuda.synchronize()
        tp = (time.perf_counter() - t0) / reps
    print(f"{which} devices={devs}: device vectors {td*1e6:8.1f} us ({st['alg_bytes']/td/1e9:6.0f} GB/s)   "
          f"partitioned vectors {tp*1e6:8.1f} us   host vectors {th*1e6:8.1f} us", flush=True)
    del A
