#!/usr/bin/env python3
"""Developer probe: per-wave timeline of ONE interleaved multi-RHS launch (panel_kernel_il_*) from the trace build
(make -C csrc trace -> libbsmrocm_trace.so).  usage: BSM_LIB=.../libbsmrocm_trace.so tools/il_trace.py [bem_c128|bem_c64|bem_f64|bem_f32|c3|c3_f32|c5s] [K]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("BSM_LIB", os.path.join(ROOT, "blocksparsematrices.jl_amd", "libbsmrocm_trace.so"))
sys.path.insert(0, ROOT)
import numpy as np, torch, bsm_amd as bsm
from bsm_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else "bem_c128"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _common import fixture_problem
S = bsm.synthetic


def bem(dtype, part):
    p = fixture_problem("cuboid", dtype, part)
    n0, tiles = p["size"][0], 400
    tile = lambda lists: [l + k * n0 for k in range(tiles) for l in lists]
    prob = dict(kind="symmetric", diagonals=p["diagonals"] * tiles, diagonalindices=tile(p["diagonalindices"]),
                offdiagonals=p["offdiagonals"] * tiles, rowindices=tile(p["rowindices"]), colindices=tile(p["colindices"]),
                size=(n0 * tiles, n0 * tiles))
    xh = np.random.default_rng(0).standard_normal(n0 * tiles)
    if np.dtype(dtype).kind == "c":
        xh = xh + 1j * np.random.default_rng(1).standard_normal(n0 * tiles)
    prob["x"] = torch.from_numpy(xh.astype(dtype)).cuda()
    return prob


prob = {"bem_c128": lambda: bem(np.complex128, "full"), "bem_c64": lambda: bem(np.complex64, "full"),
        "bem_f64": lambda: bem(np.float64, "real"), "bem_f32": lambda: bem(np.float32, "real"),
        "c3": lambda: S.config3(on_device=True), "c3_f32": lambda: S.config3(on_device=True, dtype=np.float32),
        "c5s": lambda: S.config5(n=625_000, on_device=True)}[name]()
x = prob["x"]
A = bsm.synthetic.build(prob)
n = x.shape[0]
X = torch.empty((K, n), dtype=x.dtype, device="cuda").t()
for k in range(K):
    X[:, k] = x * (k + 1)
Y = torch.zeros((K, n), dtype=x.dtype, device="cuda").t()
f = lambda: bsm.mul(Y, A, X)
for _ in range(10):
    f()
torch.cuda.synchronize()
nw = 400_000  # (more than any of these launches has waves)
buf = torch.zeros(nw * 16, dtype=torch.int64, device="cuda")
L = _lib.lib()
L.bsm_debug_set_trace.argtypes = [C.c_void_p]
assert L.bsm_debug_set_trace(buf.data_ptr()) == 0
buf.zero_()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    f()
e1.record(); torch.cuda.synchronize()
print(f"event-timed {e0.elapsed_time(e1) * 1e3 / 10:.1f} us per product (pack + pass + finish)")
t = buf.cpu().numpy().reshape(nw, 16).astype(np.float64)
act = t[:, 6] > 0
w = t[act]
tick_us = ((w[:, 8] - w[:, 7]).sum() / 100.0) / (w[:, 5] - w[:, 1]).sum()
base = (w[:, 7] - w[:, 7].min()) / 100.0
us = base[:, None] + (w[:, :6] - w[:, 1:2]) * tick_us
us -= us[:, 0].min()
ncols = (w[:, 6] // 65536).astype(int); m = (w[:, 6] % 65536).astype(int)
names = ["start", "descriptor", "lists + x rows + first tile", "first x operands", "tiles done", "stored"]
print(f"{name} x {K}: {act.sum()} panel waves; traced span {us[:, 5].max():.1f} us (s_memtime tick {tick_us * 1e3:.3f} ns); m p50 {np.median(m)}, ncols p50 {np.median(ncols)}")
for a, b in ((0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (0, 5)):
    d = us[:, b] - us[:, a]
    print(f"  {names[a]:>28s} -> {names[b]:28s} p10 {np.percentile(d, 10):6.2f}  p50 {np.percentile(d, 50):6.2f}  p90 {np.percentile(d, 90):6.2f}  mean {d.mean():6.2f}")
steps = np.ceil(ncols / 16) * np.ceil(m / 16)
d = (us[:, 4] - us[:, 3]) / np.maximum(steps, 1)
print(f"  per step (tile x row block) in the loop: p10 {np.percentile(d, 10):5.2f}  p50 {np.percentile(d, 50):5.2f}  p90 {np.percentile(d, 90):5.2f} us; steps per wave p50 {np.median(steps)}")
# resident waves over time
ev = np.concatenate([np.stack([us[:, 0], np.ones(len(us))], 1), np.stack([us[:, 5], -np.ones(len(us))], 1)])
ev = ev[np.argsort(ev[:, 0])]
res = np.cumsum(ev[:, 1])
dur = np.diff(ev[:, 0], append=ev[-1, 0])
print(f"  resident waves (time-weighted mean) {np.sum(res * dur) / np.sum(dur):.0f}")
