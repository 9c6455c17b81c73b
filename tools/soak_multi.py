#!/usr/bin/env python3
"""Developer probe: soak of the tile-pipelined multi-RHS kernels -- the same K-column product many times over large
operators, every result compared (on the GPU) with K single products.  A tile read before its LDS-DMA has landed
would show as a rare wrong column.  usage: soak_multi.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, bsm_amd as bsm
S = bsm.synthetic
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200


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
         "bem_f64": lambda: bem(200, np.float64, "real"), "c3_f32": lambda: S.config3(on_device=True, dtype=np.float32),
         # the matrix-pipe kernels: 8 complex columns (and padded 3-7), 16 real ones (and padded 9-15)
         "bem_c128": lambda: bem(200, np.complex128, "full"), "bem_c64": lambda: bem(200, np.complex64, "full")}
only = [a for a in sys.argv[2:]]
if only:
    CASES = {k: v for k, v in CASES.items() if k in only}
bad = 0
for name, make in CASES.items():
    prob = make()
    A = S.build(prob)
    x = prob["x"]
    n = x.shape[0]
    tol = 1e-12 if x.dtype in (torch.float64, torch.complex128) else 2e-5
    for K in ((8, 5, 4, 3) if x.dtype.is_complex else (16, 11, 8, 5, 4, 3)):
        g = torch.Generator(device="cuda").manual_seed(K)
        X = torch.randn((K, n), dtype=x.dtype, device="cuda", generator=g).t()
        ref = torch.zeros((K, n), dtype=x.dtype, device="cuda").t()
        y = torch.zeros_like(x)
        for k in range(K):
            bsm.mul(y, A, X[:, k].contiguous())
            ref[:, k] = y
        scale = ref.abs().max()
        Y = torch.empty((K, n), dtype=x.dtype, device="cuda").t()
        worst = 0.0
        for r in range(rounds):
            Y.fill_(float("nan"))
            bsm.mul(Y, A, X)
            err = float(((Y - ref).abs().max() / scale).item())
            worst = max(worst, err if err == err else float("inf"))
        ok = worst < tol
        bad += not ok
        print(f"{name:8s} K={K}: {rounds} products, worst rel-err vs single products {worst:.2e} {'OK' if ok else 'FAILED'}", flush=True)
    del A, prob
    torch.cuda.empty_cache()
sys.exit(1 if bad else 0)
