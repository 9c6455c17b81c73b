#!/usr/bin/env python3
"""Developer probe: the reference's real BEM fixture structure (tests/golden/symmetric_cuboid.bin:
ComplexF64, 3-28 row leaves, wide non-contiguous near-field panels) tiled K times along the
diagonal -> a large operator with the true block shapes.  Times the fused symmetric product against
a forward-only sweep of the same bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, bsm_amd as bsm
from _common import fixture_problem
K = int(sys.argv[1]) if len(sys.argv) > 1 else 400
real = len(sys.argv) > 2 and sys.argv[2] == "real"
p = fixture_problem("cuboid", np.float64 if real else np.complex128, "real" if real else "full")
n0 = p["size"][0]
def tile(lists):
    return [l + k * n0 for k in range(K) for l in lists]
prob = dict(kind="symmetric", diagonals=p["diagonals"] * K, diagonalindices=tile(p["diagonalindices"]),
            offdiagonals=p["offdiagonals"] * K, rowindices=tile(p["rowindices"]), colindices=tile(p["colindices"]),
            size=(n0 * K, n0 * K))
dt = p["diagonals"][0].dtype
rng = np.random.default_rng(0)
xh = rng.standard_normal(n0 * K).astype(dt)
def run(problem, name, **kw):
    A = bsm.synthetic.build(problem, **kw)
    st = A.stats()
    x = torch.from_numpy(xh).cuda()
    y = torch.zeros_like(x)
    plan = bsm.MulPlan(y, A, x)
    for _ in range(5):
        plan()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        plan()
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) * 1e-3 / 20
    em = st.get("win_emissions", 0)
    atom = (em - st.get("win_inside", 0) + st.get("win_flushed", 0)) / em if em else float("nan")
    print(f"{name}: {t*1e6:.1f} us, {st['alg_bytes']/t/1e9:.0f} GB/s algorithmic, stored {st['stored_entries']*dt.itemsize/1e6:.0f} MB, "
          f"wgs {st['nworkgroups']}, excl {st['exclusive']}, y contributions leaving as atomics {atom:.2f}", flush=True)
run(prob, "fused symmetric")
if os.environ.get("BEM_GATHER"):
    run(prob, "fused symmetric, gather mode (no atomics)", accumulate="gather")
fwd = dict(kind="blocksparse", blocks=prob["diagonals"] + prob["offdiagonals"],
           rowindices=prob["diagonalindices"] + prob["rowindices"],
           colindices=prob["diagonalindices"] + prob["colindices"], size=prob["size"])
run(fwd, "forward-only, same bytes")
