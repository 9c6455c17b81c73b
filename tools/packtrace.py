#!/usr/bin/env python3
"""Developer probe: where a wave of the fused product of the tiled BEM fixture spends its life -- per-wave stamps of
the trace build (make -C csrc trace -> libbsmrocm_trace.so), medians of the intervals between them.
usage: BSM_LIB=.../libbsmrocm_trace.so tools/packtrace.py [c128|f64|c64|f32] [K]     (BSM_PACK=0: the ordinary kernel)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("BSM_LIB", os.path.join(ROOT, "blocksparsematrices.jl_amd", "libbsmrocm_trace.so"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, bsm_amd as bsm
from bsm_amd import _lib
from _common import fixture_problem
tname = sys.argv[1] if len(sys.argv) > 1 else "f32"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 400
ftype = {"c128": np.complex128, "f64": np.float64, "c64": np.complex64, "f32": np.float32}[tname]
real = tname in ("f64", "f32")
p = fixture_problem("cuboid", ftype, "real" if real else "full")
n0 = p["size"][0]
tile = lambda lists: [l + k * n0 for k in range(K) for l in lists]
prob = dict(kind="symmetric", diagonals=p["diagonals"] * K, diagonalindices=tile(p["diagonalindices"]),
            offdiagonals=p["offdiagonals"] * K, rowindices=tile(p["rowindices"]), colindices=tile(p["colindices"]),
            size=(n0 * K, n0 * K))
xh = np.random.default_rng(0).standard_normal(n0 * K)
if not real:
    xh = xh + 1j * np.random.default_rng(1).standard_normal(n0 * K)
A = bsm.synthetic.build(prob)
st = A.stats()
x = torch.from_numpy(xh.astype(ftype)).cuda()
y = torch.zeros_like(x)
plan = bsm.MulPlan(y, A, x)
for _ in range(20):
    plan()
torch.cuda.synchronize()
nw = (st["nworkgroups"] + 64) * 4 * 2
buf = torch.zeros(nw * 16, dtype=torch.int64, device="cuda")
L = _lib.lib()
L.bsm_debug_set_trace.argtypes = [C.c_void_p]
assert L.bsm_debug_set_trace(buf.data_ptr()) == 0
buf.zero_()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    plan()
e1.record(); torch.cuda.synchronize()
print(f"{tname} x{K}: {e0.elapsed_time(e1) * 1e3 / 20:.1f} us per product under the trace build ({st['alg_bytes']/1e6:.0f} MB)")
t = buf.cpu().numpy().reshape(nw, 16).astype(np.float64)
names = {0: "start", 1: "record", 2: "x staged", 3: "first bytes", 4: "streamed", 5: "emitted", 9: "fwd out", 10: "counted", 11: "flushed"}
for kind, sel, last in (("packed waves", t[:, 12] == 1, 11), ("ordinary waves", (t[:, 12] == 0) & (t[:, 6] > 0), 5)):
    w = t[sel & (t[:, 8] > t[:, 7]) & (t[:, last] > t[:, 0])]
    if len(w) == 0:
        continue
    tick_us = ((w[:, 8] - w[:, 7]).sum() / 100.0) / (w[:, last] - w[:, 0]).sum()
    life = (w[:, last] - w[:, 0]) * tick_us
    print(f"{kind}: {len(w)} (the last launch that wrote their slots); s_memtime tick {tick_us*1e3:.3f} ns; wave life p10 {np.percentile(life,10):.2f} "
          f"p50 {np.percentile(life,50):.2f} p90 {np.percentile(life,90):.2f} mean {life.mean():.2f} us")
    seq = [k for k in (0, 1, 2, 3, 4, 5, 9, 10, 11) if k <= last and np.median(w[:, k]) > 0]
    for a, b in zip(seq[:-1], seq[1:]):
        d = (w[:, b] - w[:, a]) * tick_us
        print(f"  {names[a]:>12s} -> {names[b]:12s} p10 {np.percentile(d,10):6.2f}  p50 {np.percentile(d,50):6.2f}  p90 {np.percentile(d,90):6.2f}  mean {d.mean():6.2f} us")
