#!/usr/bin/env python3
"""Developer probe: timing-only ablations of the fused symmetric kernel on the tiled BEM fixture (or a
synthetic config): which part of a wave's life costs what.  Needs the experiment build
(make -C blocksparsematrices.jl_amd/csrc exp; BSM_LIB=.../libbsmrocm_exp.so); results of a run with any
bit set are WRONG by construction, only the time means something.  usage: ablate.py [K] [c128|f64|c64|f32] [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("BSM_LIB", os.path.join(ROOT, "blocksparsematrices.jl_amd", "libbsmrocm_exp.so"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, bsm_amd as bsm
from _common import fixture_problem
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if len(sys.argv) > 1 and not sys.argv[1].isdigit():  # a synthetic configuration instead of the tiled fixture
    S = bsm.synthetic
    prob = {"c3": lambda: S.config3(on_device=True), "c5s": lambda: S.config5(n=625_000, on_device=True)}[sys.argv[1]]()
    K, dt = sys.argv[1], np.dtype(np.float64)
    A = S.build(prob)
    st = A.stats()
    x = prob["x"]
else:
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    tname = sys.argv[2] if len(sys.argv) > 2 else "c128"
    ftype = {"c128": np.complex128, "f64": np.float64, "c64": np.complex64, "f32": np.float32}[tname]
    real = tname in ("f64", "f32")
    p = fixture_problem("cuboid", ftype, "real" if real else "full")
    n0 = p["size"][0]
    tile = lambda lists: [l + k * n0 for k in range(K) for l in lists]
    prob = dict(kind="symmetric", diagonals=p["diagonals"] * K, diagonalindices=tile(p["diagonalindices"]),
                offdiagonals=p["offdiagonals"] * K, rowindices=tile(p["rowindices"]), colindices=tile(p["colindices"]),
                size=(n0 * K, n0 * K))
    dt = p["diagonals"][0].dtype
    xh = np.random.default_rng(0).standard_normal(n0 * K)
    if not real:
        xh = xh + 1j * np.random.default_rng(1).standard_normal(n0 * K)
    xh = xh.astype(dt)
    A = bsm.synthetic.build(prob)
    st = A.stats()
    x = torch.from_numpy(xh).cuda()
y = torch.zeros_like(x)
plan = bsm.MulPlan(y, A, x)
VARIANTS = [("full kernel", 0), ("no global atomics", 1), ("no window adds", 2), ("no atomics, no window adds", 3),
            ("no emission loop", 8), ("no butterfly", 4), ("no butterfly, no emission", 12),
            ("no x gather", 16), ("no forward output", 32), ("no x gather, no emission, no butterfly, no fwd out", 60)]
res = {n: [] for n, _ in VARIANTS}
for r in range(rounds):
    for name, bits in VARIANTS:
        os.environ["BSM_DEBUG_FLAGS"] = str(bits)
        for _ in range(3):
            plan()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            plan()
        b.record()
        torch.cuda.synchronize()
        res[name].append(a.elapsed_time(b) * 1e3 / 20)
print(f"operator {K} ({dt}), {st['alg_bytes']/1e6:.0f} MB algorithmic, {st['nworkgroups']} workgroups; us per launch (median of {rounds} interleaved rounds, min)")
for name, _ in VARIANTS:
    v = sorted(res[name])
    print(f"  {name:55s} {v[len(v)//2]:7.1f}  {v[0]:7.1f}   {st['alg_bytes']/v[len(v)//2]/1e3:6.0f} GB/s")
