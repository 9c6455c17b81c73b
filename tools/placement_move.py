#!/usr/bin/env python3
"""Developer probe: which array of a device image carries the handle's placement level (1 GB VBCRS leg: 164 vs 178 us)?
One handle; its wave records / column pool / row pool / value stream are moved to fresh allocations one at a time
(bsm_debug_move_image_array) and the product is timed after every move.  usage: placement_move.py [c2x20|c3|c4s|c5s] [values-only]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bsm_amd as bsm
from bsm_amd import _lib as L
S = bsm.synthetic
which_op = sys.argv[1] if len(sys.argv) > 1 else "c2x20"
p = {"c2x20": lambda: S.config2(n=2_000_000, nblocks=100_000, on_device=True), "c3": lambda: S.config3(on_device=True),
     "c4s": lambda: S.config4(on_device=True, row_lo=0, row_hi=1953), "c5s": lambda: S.config5(n=625_000, on_device=True)}[which_op]()
only_values = len(sys.argv) > 2
A = S.build(p)
x = p["x"]
y = torch.zeros_like(x)
lib = L.lib()
lib.bsm_debug_move_image_array.argtypes = [C.c_void_p, C.c_int]
lib.bsm_debug_move_image_array.restype = C.c_int


def t_of():
    plan = bsm.MulPlan(y, A, x)  # (a plan caches nothing about the image's addresses: it calls bsm_mul)
    for _ in range(10):
        plan()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(60):
        plan()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / 60


keep = []
print(f"{which_op} as built: {t_of():.1f} us", flush=True)
for which, name in ((3, "wave records"), (2, "column pool"), (1, "row pool"), (0, "value stream")):
    if only_values and which != 0:
        continue
    for k in range(6 if only_values else 4):
        keep.append(torch.empty((k + 1) * 1_300_000 + 4096 * k, dtype=torch.uint8, device="cuda"))  # shift the allocator
        L.check(lib.bsm_debug_move_image_array(A._h.ptr, which))
        print(f"{name} moved ({k + 1}): {t_of():.1f} us", flush=True)
