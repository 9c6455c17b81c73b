#!/usr/bin/env python3
"""Developer probe: per-wave timeline of ONE C2 launch from the trace build (make -C csrc trace ->
libbsmrocm_trace.so): when waves start, when their descriptor / x slice / first matrix bytes arrive,
when they finish (s_memrealtime, 100 MHz).  usage: BSM_LIB=.../libbsmrocm_trace.so tools/wavetrace.py [c2|c2p|c2w]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bsm_amd as bsm
from bsm_amd import _lib
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
S = bsm.synthetic
p = {"c2": S.config2, "c2p": lambda: S.config2(n=100000, lo=32, hi=32, nblocks=6400),
     "c2w": lambda: S.config2(n=100000, lo=64, hi=64, nblocks=1650),
     "bem": lambda: S.config5(n=400_000, lo=8, hi=28, halfband=8),
     "c2x20": lambda: S.config2(n=2_000_000, nblocks=100_000),
     "c4s": lambda: S.config4(row_lo=0, row_hi=1953),
     "c3s": lambda: S.config3(nseg=800)}[which]()
A = S.build(p)
st = A.stats()
nw = st["nworkgroups"] * 4
x = torch.from_numpy(p["x"]).cuda(); y = torch.zeros_like(x)
plan = bsm.MulPlan(y, A, x)
for _ in range(20):
    plan()
torch.cuda.synchronize()
buf = torch.zeros(nw * 16, dtype=torch.int64, device="cuda")
L = _lib.lib()
L.bsm_debug_set_trace.argtypes = [C.c_void_p]
assert L.bsm_debug_set_trace(buf.data_ptr()) == 0
res = []
for rep in range(5):
    buf.zero_()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):  # back to back: the buffer keeps the LAST launch (clocks up, caches as in the bench)
        plan()
    e1.record(); torch.cuda.synchronize()
    per_launch = e0.elapsed_time(e1) * 1e3 / 50
    t = buf.cpu().numpy().reshape(nw, 16).astype(np.float64)
    act = t[:, 6] > 0
    # s_memtime runs per XCD (unsynchronised bases, shader-clock rate): every wave also takes the
    # chip-wide 100 MHz s_memrealtime at its descriptor stamp and at its end.  Rate from the sums,
    # placement of each wave from its own realtime stamp.
    w = t[act]
    tick_us = ((w[:, 8] - w[:, 7]).sum() / 100.0) / (w[:, 5] - w[:, 1]).sum()
    base = (w[:, 7] - w[:, 7].min()) / 100.0           # descriptor stamp on the common clock (10 ns steps)
    us = base[:, None] + (w[:, :6] - w[:, 1:2]) * tick_us
    us -= us[:, 0].min()
    res.append(us)
    nbytes = t[act, 6]
us = res[-1]
names = ["start", "descriptor", "x staged", "first bytes", "streamed", "stored"]
print(f"event-timed {per_launch:.2f} us per launch; traced span {us[:,5].max():.2f} us (s_memtime tick {tick_us*1e3:.3f} ns)")
print(f"{which}: {act.sum()} panel waves, {nbytes.sum()/1e6:.1f} MB; times in us since the first wave start (last of 5 launches)")
for k, nm in enumerate(names):
    v = us[:, k]
    print(f"  {nm:12s} min {v.min():6.2f}  p10 {np.percentile(v,10):6.2f}  p50 {np.percentile(v,50):6.2f}  p90 {np.percentile(v,90):6.2f}  max {v.max():6.2f}")
for a, b in ((0, 1), (1, 2), (2, 3), (3, 4), (4, 5)):
    d = us[:, b] - us[:, a]
    print(f"  {names[a]:>12s} -> {names[b]:12s} p10 {np.percentile(d,10):5.2f}  p50 {np.percentile(d,50):5.2f}  p90 {np.percentile(d,90):5.2f}  max {d.max():5.2f}")
# bytes delivered over time (by 'streamed' timestamps)
order = np.argsort(us[:, 4]); cum = np.cumsum(nbytes[order])
for q in (0.1, 0.25, 0.5, 0.75, 0.9, 1.0):
    k = np.searchsorted(cum, q * cum[-1]);  k = min(k, len(cum) - 1)
    print(f"  {int(q*100):3d} % of the bytes consumed by t = {us[order[k], 4]:.2f} us")
# piece sizes: the critical path of a short launch is the wave with the most 8 KB iterations
it = np.ceil(nbytes / 8192.0)
print("  bytes per wave: p50 %.0f  p90 %.0f  p99 %.0f  max %.0f;  iterations per wave: " % tuple(np.percentile(nbytes, [50, 90, 99, 100])) +
      "  ".join(f"{int(k)}:{int((it == k).sum())}" for k in np.unique(it)))
for k in np.unique(it):
    sel = it == k
    print(f"    {int(k)} iteration(s): {sel.sum():5d} waves, stored p50 {np.percentile(us[sel,5],50):5.2f}  max {us[sel,5].max():5.2f} us; first bytes p50 {np.percentile(us[sel,3],50):5.2f}")
