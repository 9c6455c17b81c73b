#!/usr/bin/env python3
"""Offline model of a short (C2-sized) launch: every wave issues one 8 KB iteration at a time, the memory
system serves requests first-come-first-served at a fixed rate after a fixed latency.  Evaluates how the
wave-split policy (bytes -> waves per row group) moves the end of the launch.  Calibrated on
tools/wavetrace.py (span 8.3-9.0 us for the round-1 policy)."""
import heapq, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bsm_amd as bsm

def groups_c2(p):
    rows = np.asarray(p["rowstart"]); m = np.array([b.shape[0] for b in p["blocks"]]); n = np.array([b.shape[1] for b in p["blocks"]])
    out = {}
    for r, mm, nn in zip(rows, m, n):
        g = out.setdefault(int(r), [mm, 0]); g[1] += nn
    return [(mm, ((w + 1) // 2) * 2) for mm, w in out.values()]  # width padded to whole strips

def simulate(waves, rate=10.0e6, lat=1.2, t_req0=1.0, it_bytes=8192, order="large"):
    """waves: bytes per wave; rate in bytes/us; returns end time (us)"""
    # event simulation: each wave requests iteration k at time t; server FIFO
    if order == "large":
        waves = sorted(waves, reverse=True)
    ev = [(t_req0 + 0.3 * i / len(waves), i, b) for i, b in enumerate(waves)]  # start ramp 0.3 us
    heapq.heapify(ev)
    server_free = 0.0; end = 0.0
    while ev:
        t, i, left = heapq.heappop(ev)
        chunk = min(left, it_bytes)
        start = max(t, server_free)
        server_free = start + chunk / rate
        arrive = server_free + lat
        left -= chunk
        if left > 0:
            heapq.heappush(ev, (arrive + 0.05, i, left))
        else:
            end = max(end, arrive + 0.4)
    return end

def split(groups, policy, wpw=4):
    waves = []
    for m, w in groups:
        b = m * w * 8
        nw = policy(b)
        strips = w // 2; per = -(-strips // nw)
        for k in range(nw):
            s = max(0, min(strips, (k + 1) * per) - k * per)
            waves.append(s * m * 16)
    return waves

if __name__ == "__main__":
    p = bsm.synthetic.config2()
    G = groups_c2(p)
    gb = np.array([m * w * 8 for m, w in G])
    print(f"{len(G)} row groups, bytes p50 {np.percentile(gb,50):.0f} p90 {np.percentile(gb,90):.0f} max {gb.max()}, total {gb.sum()/1e6:.1f} MB")
    pols = {
        "round-1 (8K->2, 24K->4)": lambda b: 4 if b >= 24576 else (2 if b >= 8192 else 1),
        "8K->2, 16K->4": lambda b: 4 if b >= 16384 else (2 if b >= 8192 else 1),
        "8K->2, 16K->4, 40K->8": lambda b: 8 if b >= 40960 else (4 if b >= 16384 else (2 if b >= 8192 else 1)),
        "8K->2, 24K->4, 48K->8": lambda b: 8 if b >= 49152 else (4 if b >= 24576 else (2 if b >= 8192 else 1)),
        "12K->2, 24K->4, 48K->8": lambda b: 8 if b >= 49152 else (4 if b >= 24576 else (2 if b >= 12288 else 1)),
        "ceil(b/8K) pow2 <= 8": lambda b: min(8, 1 << max(0, int(np.ceil(np.log2(max(b, 1) / 8192.0))))),
        "ceil(b/8K) pow2 <= 16": lambda b: min(16, 1 << max(0, int(np.ceil(np.log2(max(b, 1) / 8192.0))))),
    }
    for name, pol in pols.items():
        w = split(G, pol)
        w = [x for x in w if x > 0]
        it = np.ceil(np.array(w) / 8192)
        print(f"{name:28s}: {len(w):5d} waves, max {max(w):6d} B, iterations " + " ".join(f"{int(k)}:{int((it==k).sum())}" for k in np.unique(it)) +
              f"  -> end {simulate(w):.2f} us")
