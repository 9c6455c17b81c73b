"""Seeded random operators for the differential tests (CPU: packed-image interpreter vs oracle;
GPU: HIP path vs oracle).  The generator aims at the corners of the packed layout rather than at
realism: row counts around the 8/16/32/64 lane-group boundaries and the 64-row chunk limit, single
rows and columns, empty blocks, repeated row sets (merged panels), overlapping row ranges, unsorted
and strided index lists, rectangular operators, more than three column runs per panel."""
import numpy as np

EDGE = [1, 2, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 129]


def _size(rng):
    return int(rng.choice(EDGE)) if rng.random() < 0.7 else int(rng.integers(1, 90))


def _block(rng, m, n, dtype):
    b = rng.standard_normal((m, n))
    if np.dtype(dtype).kind == "c":
        b = b + 1j * rng.standard_normal((m, n))
    return np.asfortranarray(b.astype(dtype))


def _index_list(rng, k, n):
    """k distinct 1-based indices: contiguous, strided, sorted-scattered or shuffled"""
    mode = rng.integers(0, 4)
    if mode == 0 or k == 1:
        s = int(rng.integers(1, n - k + 2))
        return np.arange(s, s + k, dtype=np.int64)
    if mode == 1 and 2 * k <= n:
        s = int(rng.integers(1, n - 2 * k + 2))
        return np.arange(s, s + 2 * k, 2, dtype=np.int64)
    idx = rng.choice(n, size=k, replace=False).astype(np.int64) + 1
    return np.sort(idx) if mode == 2 else idx


def random_blocksparse(rng, dtype):
    nr, nc = int(rng.integers(150, 700)), int(rng.integers(150, 700))
    blocks, rows, cols = [], [], []
    pool = []  # row sets that get reused: several blocks on the SAME rows merge into one panel
    for _ in range(int(rng.integers(1, 40))):
        if pool and rng.random() < 0.4:
            r = pool[int(rng.integers(0, len(pool)))]
        else:
            r = _index_list(rng, min(_size(rng), nr), nr)
            pool.append(r)
        n = 0 if rng.random() < 0.05 else min(_size(rng), nc)
        c = _index_list(rng, n, nc) if n else np.zeros(0, np.int64)
        if rng.random() < 0.05:
            r = np.zeros(0, np.int64)
        blocks.append(_block(rng, len(r), len(c), dtype))
        rows.append(r)
        cols.append(c)
    if rng.random() < 0.2:  # one very wide panel: many x chunks, a row group cut into several work items
        r = _index_list(rng, int(rng.choice([17, 40, 64])), nr)
        for _ in range(int(rng.integers(8, 30))):
            c = _index_list(rng, min(int(rng.integers(60, 130)), nc), nc)
            blocks.append(_block(rng, len(r), len(c), dtype))
            rows.append(r)
            cols.append(c)
    return dict(kind="blocksparse", blocks=blocks, rowindices=rows, colindices=cols, size=(nr, nc))


def random_vbcrs(rng, dtype):
    nr, nc = int(rng.integers(150, 900)), int(rng.integers(150, 900))
    blocks, r0, c0 = [], [], []
    starts = []
    for _ in range(int(rng.integers(1, 50))):
        m, n = min(_size(rng), nr), min(_size(rng), nc)
        if starts and rng.random() < 0.5:  # another block of an existing block row
            rs, m = starts[int(rng.integers(0, len(starts)))]
        else:
            rs = int(rng.integers(1, nr - m + 2))
            starts.append((rs, m))
        blocks.append(_block(rng, m, n, dtype))
        r0.append(rs)
        c0.append(int(rng.integers(1, nc - n + 2)))
    return dict(kind="vbcrs", blocks=blocks, rowstart=np.array(r0, np.int64), colstart=np.array(c0, np.int64),
                size=(nr, nc))


def random_symmetric(rng, dtype):
    n = int(rng.integers(200, 800))
    perm = rng.permutation(n) + 1 if rng.random() < 0.5 else np.arange(1, n + 1)
    sets, pos = [], 0
    while pos < n:
        k = min(_size(rng) if rng.random() < 0.5 else int(rng.integers(1, 40)), n - pos)
        s = perm[pos:pos + k].astype(np.int64)
        sets.append(np.sort(s) if rng.random() < 0.5 else s)
        pos += k
        if rng.random() < 0.1:
            pos += int(rng.integers(0, 20))  # rows no block covers
    diags, dsets = [], []
    for s in sets:
        if rng.random() < 0.85:
            d = _block(rng, len(s), len(s), dtype)
            diags.append(np.asfortranarray((d + d.T) / 2))
            dsets.append(s)
    offs, ri, ci = [], [], []
    for _ in range(int(rng.integers(0, 3 * len(sets)))):
        i, j = int(rng.integers(0, len(sets))), int(rng.integers(0, len(sets)))
        if i == j:
            continue
        offs.append(_block(rng, len(sets[i]), len(sets[j]), dtype))
        ri.append(sets[i])
        ci.append(sets[j])
    if rng.random() < 0.2 and len(sets) > 3:  # one row set coupled to (almost) every other: a very wide panel
        i = int(rng.integers(0, len(sets)))
        for j in range(len(sets)):
            if j != i and rng.random() < 0.9:
                offs.append(_block(rng, len(sets[i]), len(sets[j]), dtype))
                ri.append(sets[i])
                ci.append(sets[j])
    return dict(kind="symmetric", diagonals=diags, diagonalindices=dsets, offdiagonals=offs, rowindices=ri,
                colindices=ci, size=(n, n))


GEN = {"blocksparse": random_blocksparse, "vbcrs": random_vbcrs, "symmetric": random_symmetric}




def seed_of(kind, dtype):
    """deterministic per (type, element type); BSM_FUZZ_OFFSET=k explores other streams"""
    import os
    off = int(os.environ.get("BSM_FUZZ_OFFSET", "0"))
    return (sum(map(ord, kind)) * 1000003 + sum(map(ord, np.dtype(dtype).str)) + 7919 * off) % (2 ** 32)
