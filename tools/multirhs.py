#!/usr/bin/env python3
"""Developer probe: K right-hand sides through bsm_mul_multi against K single products (and one), on the
standard operator set.  usage: multirhs.py [name ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, bsm_amd as bsm
S = bsm.synthetic


def timed(fn, reps):
    for _ in range(3):
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


def bem(K, dtype, part):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _common import fixture_problem
    p = fixture_problem("cuboid", dtype, part)
    n0 = p["size"][0]
    tile = lambda lists: [l + k * n0 for k in range(K) for l in lists]
    prob = dict(kind="symmetric", diagonals=p["diagonals"] * K, diagonalindices=tile(p["diagonalindices"]),
                offdiagonals=p["offdiagonals"] * K, rowindices=tile(p["rowindices"]), colindices=tile(p["colindices"]),
                size=(n0 * K, n0 * K))
    rng = np.random.default_rng(0)
    xh = rng.standard_normal(n0 * K)
    if np.dtype(dtype).kind == "c":  # a FULL complex x: zero imaginary parts run 5-8 % faster (less switching, higher clock)
        xh = xh + 1j * rng.standard_normal(n0 * K)
    prob["x"] = torch.from_numpy(xh.astype(dtype)).cuda()
    return prob


CASES = {
    "c2": lambda: S.config2(on_device=True),
    "c2x20": lambda: S.config2(n=2_000_000, nblocks=100_000, on_device=True),
    "c3": lambda: S.config3(on_device=True),
    "c3_f32": lambda: S.config3(on_device=True, dtype=np.float32),
    "c4s": lambda: S.config4(on_device=True, row_lo=0, row_hi=1953),
    "c5s": lambda: S.config5(n=625_000, on_device=True),
    "c5small": lambda: S.config5(n=400_000, lo=8, hi=32, halfband=8, on_device=True),  # banded, panels of 8-32 rows
    "c5small_f32": lambda: S.config5(n=400_000, lo=8, hi=32, halfband=8, dtype=np.float32, on_device=True),
    "bem_c128": lambda: bem(400, np.complex128, "full"),
    "bem_c64": lambda: bem(400, np.complex64, "full"),
    "bem_f64": lambda: bem(400, np.float64, "real"),
    "bem_f32": lambda: bem(400, np.float32, "real"),
}
names = sys.argv[1:] or list(CASES)
for name in names:
    prob = CASES[name]()
    A = S.build(prob)
    st = A.stats()
    x = prob["x"]
    n = x.shape[0]
    y = torch.zeros_like(x)
    plan = bsm.MulPlan(y, A, x)
    reps = 100 if st["alg_bytes"] < 200e6 else 15
    t1 = timed(plan, reps)
    line = f"{name:9s} 1 rhs {t1:8.1f} us ({st['alg_bytes']/t1/1e3:5.0f} GB/s)"
    for K in (4, 8, 16):
        X = torch.empty((K, n), dtype=x.dtype, device="cuda").t()  # column-major n x K
        for k in range(K):
            X[:, k] = x * (k + 1)
        Y = torch.zeros((K, n), dtype=x.dtype, device="cuda").t()
        tk = timed(lambda: bsm.mul(Y, A, X), reps)
        line += f"   {K} rhs {tk:8.1f} us = {tk/t1:4.2f} x one product"
        del X, Y
    print(line, flush=True)
    del plan, A, prob
    torch.cuda.empty_cache()
