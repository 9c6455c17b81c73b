#!/usr/bin/env python3
"""Developer probe: warm mul! time of the standard operator set under ONE library build (BSM_LIB) and the
current BSM_* knobs -- run it once per build, interleaved, to A/B a change over every kind of operator.
usage: abbench.py [name ...]   (default: all); ABB_ACC=gather|atomic|colored picks the accumulation mode"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, bsm_amd as bsm
from _common import fixture_problem
S = bsm.synthetic


def bem(K, dtype, part):
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


KB = int(os.environ.get("ABB_K", 400))  # tiles of the BEM fixture
CASES = {
    "c2": lambda: S.config2(on_device=True),
    "c2x20": lambda: S.config2(n=2_000_000, nblocks=100_000, on_device=True),
    "c3": lambda: S.config3(on_device=True),
    "c4s": lambda: S.config4(on_device=True, row_lo=0, row_hi=1953),
    "c5s": lambda: S.config5(n=625_000, on_device=True),
    "c5s_f32": lambda: S.config5(n=625_000, dtype=np.float32, on_device=True),
    "c3_f32": lambda: S.config3(dtype=np.float32, on_device=True),
    "bem_c128": lambda: bem(KB, np.complex128, "full"),
    "bem_f64": lambda: bem(KB, np.float64, "real"),
    "bem_c64": lambda: bem(KB, np.complex64, "full"),
    "bem_f32": lambda: bem(KB, np.float32, "real"),
}
names = sys.argv[1:] or list(CASES)
tag = os.path.basename(os.environ.get("BSM_LIB", "libbsmrocm.so")) + (":" + os.environ["ABB_ACC"] if os.environ.get("ABB_ACC") else "")
for name in names:
    prob = CASES[name]()
    A = S.build(prob, **({"accumulate": os.environ["ABB_ACC"]} if os.environ.get("ABB_ACC") else {}))
    st = A.stats()
    x = prob["x"]
    y = torch.zeros_like(x)
    plan = bsm.MulPlan(y, A, x)
    reps = int(os.environ.get("ABB_REPS", 0)) or (200 if st["alg_bytes"] < 200e6 else 30)
    for _ in range(10):
        plan()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            plan()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / reps)
    ts.sort()
    print(f"{tag:22s} {name + (f'x{KB}' if name.startswith('bem') and KB != 400 else ''):13s} median {ts[2]:8.2f} us  min {ts[0]:8.2f}  {st['alg_bytes']/ts[2]/1e3:6.0f} GB/s", flush=True)
    del plan, A, prob
    torch.cuda.empty_cache()
