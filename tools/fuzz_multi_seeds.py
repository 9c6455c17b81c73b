#!/usr/bin/env python3
"""Developer probe: the multi-RHS part of tests/test_gpu_fuzz.py over OTHER seeds (the suite's own seeds are fixed):
random operators of the three types and four element types, every accumulation mode, ops N / T / C, 2-35 columns,
complex scalars for the complex types, every column against the oracle.  usage: fuzz_multi_seeds.py [first seed] [count]"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, bsm_amd as bsm
from oracle import load_oracle
from _common import Cc, N, T, oracle_mul, rand_vec
from _fuzz import GEN
orc = load_oracle()
TOL = {np.dtype(np.float64): 1e-12, np.dtype(np.complex128): 1e-12, np.dtype(np.float32): 1e-5, np.dtype(np.complex64): 1e-5}
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 6
bad = done = 0
for seed in range(first, first + count):
    for kind in ("blocksparse", "vbcrs", "symmetric"):
        for dtype in (np.float64, np.complex128, np.float32, np.complex64):
            dtype = np.dtype(dtype)
            rng = np.random.default_rng(seed * 1000 + zlib.crc32((kind + dtype.str).encode()) % 997)
            for case in range(8):
                p = GEN[kind](rng, dtype)
                modes = ["auto", "atomic", "gather"] + (["colored"] if kind != "vbcrs" else [])
                acc = modes[case % len(modes)]
                kw = {"accumulate": acc}
                if kind != "symmetric" and case % 3 == 0:
                    kw["transpose_image"] = True
                try:
                    A = bsm.synthetic.build(p, **kw)
                except RuntimeError as e:
                    if acc == "colored" and "repeat" in str(e):
                        continue
                    raise
                nr, nc = p["size"]
                for op in (N, T, Cc):
                    if op == Cc and dtype.kind != "c":
                        continue
                    xl, yl = (nc, nr) if op == N else (nr, nc)
                    Aop = A if op == N else (bsm.transpose(A) if op == T else bsm.adjoint(A))
                    k = int(rng.choice([2, 3, 4, 5, 7, 8, 9, 11, 15, 16, 17, 24, 35]))
                    X = np.asfortranarray(np.stack([rand_vec(rng, xl, dtype) for _ in range(k)], axis=1))
                    Y0 = np.asfortranarray(np.stack([rand_vec(rng, yl, dtype) for _ in range(k)], axis=1))
                    am, bm = (-0.5 + 0.75j, 1.25 - 0.5j) if dtype.kind == "c" else (-0.5, 1.25)
                    strong = bool(rng.integers(0, 2))
                    Yd = torch.from_numpy(Y0.T.copy()).cuda().T
                    bsm.mul(Yd, Aop, torch.from_numpy(X.T.copy()).cuda().T, am, False if strong else bm)
                    got = Yd.cpu().numpy()
                    for j in range(k):
                        ref = oracle_mul(orc, p, op, X[:, j].copy(), Y0[:, j].copy(), am, bm, strong)
                        err = np.max(np.abs(got[:, j] - ref)) / max(np.max(np.abs(ref)), 1e-30)
                        done += 1
                        if not err < TOL[dtype]:
                            bad += 1
                            print("MISMATCH", seed, kind, dtype, case, acc, op, k, j, err, flush=True)
    print(f"seed {seed}: {done} columns checked, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
