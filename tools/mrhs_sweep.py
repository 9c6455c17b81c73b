#!/usr/bin/env python3
"""Developer probe: cost of nrhs = 1..9 right-hand sides in single products (batches of 8 / 4, padded remainders).
usage: mrhs_sweep.py [c3|c5s|c4s|bem_f64 ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, bsm_amd as bsm
S = bsm.synthetic


def timed(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / reps)
    return sorted(ts)[1]


def bem(tiles, dtype, part):
    from _common import fixture_problem
    p = fixture_problem("cuboid", dtype, part)
    n0 = p["size"][0]
    tile = lambda lists: [l + k * n0 for k in range(tiles) for l in lists]
    prob = dict(kind="symmetric", diagonals=p["diagonals"] * tiles, diagonalindices=tile(p["diagonalindices"]),
                offdiagonals=p["offdiagonals"] * tiles, rowindices=tile(p["rowindices"]), colindices=tile(p["colindices"]),
                size=(n0 * tiles, n0 * tiles))
    xh = np.random.default_rng(0).standard_normal(n0 * tiles)
    if np.dtype(dtype).kind == "c":
        xh = xh + 1j * np.random.default_rng(1).standard_normal(n0 * tiles)
    prob["x"] = torch.from_numpy(xh.astype(dtype)).cuda()
    return prob


CASES = {"c3": lambda: S.config3(on_device=True), "c5s": lambda: S.config5(n=625_000, on_device=True),
         "c4s": lambda: S.config4(on_device=True, row_lo=0, row_hi=1953), "bem_f64": lambda: bem(400, np.float64, "real"),
         "bem_f32": lambda: bem(400, np.float32, "real"), "bem_c128": lambda: bem(400, np.complex128, "full"),
         "bem_c64": lambda: bem(400, np.complex64, "full")}
for name in sys.argv[1:] or ["c3"]:
    prob = CASES[name]()
    A = S.build(prob)
    x = prob["x"]
    n = x.shape[0]
    y = torch.zeros_like(x)
    plan = bsm.MulPlan(y, A, x)
    reps = 15
    for _ in range(30):
        plan()
    t1 = timed(plan, reps)
    line = f"{name:8s} 1: {t1:7.1f} us |"
    for K in range(2, 10):
        X = torch.randn((K, n), dtype=x.dtype, device="cuda").t()
        Y = torch.zeros((K, n), dtype=x.dtype, device="cuda").t()
        tk = timed(lambda: bsm.mul(Y, A, X), reps)
        line += f" {K}: {tk / t1:4.2f}"
        del X, Y
    print(line + "   (single products)", flush=True)
    del plan, A, prob
    torch.cuda.empty_cache()
