#!/usr/bin/env python3
"""Developer probe for profilers: N launches of the K-right-hand-side product of one operator.
usage: mrhs_one.py <case> <K> [launches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, bsm_amd as bsm
S = bsm.synthetic
name, K = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
def bem(tiles, dtype, part):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
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


prob = {"bem_c128": lambda: bem(400, np.complex128, "full"), "bem_c64": lambda: bem(400, np.complex64, "full"),
        "bem_f64": lambda: bem(400, np.float64, "real"), "bem_f32": lambda: bem(400, np.float32, "real"),
        "c2": lambda: S.config2(on_device=True), "c3": lambda: S.config3(on_device=True),
        "c2x20": lambda: S.config2(n=2_000_000, nblocks=100_000, on_device=True),
        "c4s": lambda: S.config4(on_device=True, row_lo=0, row_hi=1953),
        "c5s": lambda: S.config5(n=625_000, on_device=True)}[name]()
A = S.build(prob)
x = prob["x"]
n = x.shape[0]
if K == 1:
    y = torch.zeros_like(x)
    f = bsm.MulPlan(y, A, x)
else:
    X = torch.empty((K, n), dtype=x.dtype, device="cuda").t()
    for k in range(K):
        X[:, k] = x * (k + 1)
    Y = torch.zeros((K, n), dtype=x.dtype, device="cuda").t()
    f = lambda: bsm.mul(Y, A, X)
for _ in range(reps):
    f()
torch.cuda.synchronize()
print("done", name, K, A.stats()["alg_bytes"])
