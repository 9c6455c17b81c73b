#!/usr/bin/env python3
"""Developer probe: timing-only ablations of the pipelined multi-RHS kernel (experiment build, results WRONG
with any bit set).  usage: ablate_multi.py [c3|c5s|c2x20|c4s|bem_f64|bem_c128|bem_c64] [K]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("BSM_LIB", os.path.join(ROOT, "blocksparsematrices.jl_amd", "libbsmrocm_exp.so"))
sys.path.insert(0, ROOT)
import numpy as np, torch, bsm_amd as bsm
S = bsm.synthetic
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
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


prob = {"c3": lambda: S.config3(on_device=True), "c5s": lambda: S.config5(n=625_000, on_device=True),
        "c2x20": lambda: S.config2(n=2_000_000, nblocks=100_000, on_device=True),
        "c4s": lambda: S.config4(on_device=True, row_lo=0, row_hi=1953),
        "bem_f64": lambda: bem(400, np.float64, "real"), "bem_c128": lambda: bem(400, np.complex128, "full"),
        "bem_c64": lambda: bem(400, np.complex64, "full")}[name]()
A = S.build(prob)
x = prob["x"]; n = x.shape[0]
X = torch.empty((K, n), dtype=x.dtype, device="cuda").t()
for k in range(K):
    X[:, k] = x * (k + 1)
Y = torch.zeros((K, n), dtype=x.dtype, device="cuda").t()
VAR = [("full", 0), ("no matrix loads", 64), ("no atomics", 1), ("no forward half", 128), ("no transposed half", 256),
       ("no loads, no atomics", 65), ("loads + atomics only", 128 + 256), ("loads only", 128 + 256 + 1), ("nothing", 64 + 1 + 128 + 256),
       ("nothing, no x gather", 64 + 1 + 128 + 256 + 16), ("nothing, no forward output", 64 + 1 + 128 + 256 + 32),
       ("nothing, no x gather, no forward output", 64 + 1 + 128 + 256 + 16 + 32), ("no x gather", 16), ("no forward output", 32)]
for r in range(2):
    for nm, bits in VAR:
        os.environ["BSM_DEBUG_FLAGS"] = str(bits)
        for _ in range(3):
            bsm.mul(Y, A, X)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(15):
            bsm.mul(Y, A, X)
        b.record(); torch.cuda.synchronize()
        st = A.stats()
        print(f"{name} x{K} waves {st['ntasks']} {nm:28s} {a.elapsed_time(b)*1e3/15:8.1f} us", flush=True)
