"""Synthetic block-sparse operators of BASELINE.json's configs (definitions: SURVEY.md 8d).

Portable counter-based RNG so that C++/Python/Julia produce bit-identical inputs:
    mix(z)      = SplitMix64 finaliser
    u(seed, k)  = mix(seed + GOLDEN * (k + 1))                       (k-th draw of stream `seed`)
    value       = (u >> 11) * 2^-53 * 2 - 1          in [-1, 1)
    size in lo..hi = lo + u mod (hi - lo + 1)
Structure draws use stream seed = 0xB5A0 + config; block b's values use stream
mix(seed ^ mix(b + 1)); x uses stream mix(seed ^ 0x5851F42D4C957F2D).
Every index list returned is 1-based (Julia convention), blocks are column-major.
"""
import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def mix(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def draws(seed, start, count):
    """u(seed, start .. start+count-1) as uint64."""
    k = np.arange(start + 1, start + count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return mix(np.uint64(seed) + GOLDEN * k)


def to_unit(u, dtype=np.float64):
    v = (u >> np.uint64(11)).astype(np.float64) * (2.0 ** -53) * 2.0 - 1.0
    return v.astype(dtype)


def block_values(seed, b, m, n, dtype=np.float64):
    """column-major m x n block b of config stream `seed`."""
    s = int(mix(np.uint64(seed) ^ mix(np.uint64(b + 1))))
    return to_unit(draws(s, 0, m * n), dtype).reshape((m, n), order="F")


def vector(seed, n, dtype=np.float64):
    s = int(mix(np.uint64(seed) ^ np.uint64(0x5851F42D4C957F2D)))
    return to_unit(draws(s, 0, n), dtype)


def _segments(seed, stream_off, total, lo, hi):
    """consecutive segments of size U{lo..hi} covering 1..total (last truncated) -> starts (0-based), sizes"""
    est = total // lo + 2
    sz = (lo + (draws(seed, stream_off, est) % np.uint64(hi - lo + 1))).astype(np.int64)
    ends = np.cumsum(sz)
    nseg = int(np.searchsorted(ends, total, side="left")) + 1
    sz = sz[:nseg].copy()
    starts = np.concatenate([[0], ends[:nseg - 1]])
    sz[-1] = total - starts[-1]
    return starts, sz


# ---- generation in HBM (include/bsm_synth.h): same streams, bit-identical values --------------------
def device_blocks(seed, ids, ms, ns, dtype=np.float64, symmetrise=None):
    """Blocks `ids` of config stream `seed`, generated ON the current GPU: one flat allocation, one
    kernel launch, returned as column-major torch CUDA views (m x n, stride (1, m)) -- what
    block_values() yields, without the host ever holding the operator.  symmetrise[k] != 0 stores
    (D + D^T) / 2 of the draw (the diagonal blocks of the symmetric configs)."""
    import ctypes as C
    import torch
    from . import _lib as L
    dtype = np.dtype(dtype)
    tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}[dtype]
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    ms = np.ascontiguousarray(ms, dtype=np.int64)
    ns = np.ascontiguousarray(ns, dtype=np.int64)
    cnt = ms * ns
    # every block starts on a 16-byte boundary
    al = 16 // dtype.itemsize
    padded = (cnt + al - 1) // al * al
    offs = np.concatenate([[0], np.cumsum(padded)])
    flat = torch.empty(int(offs[-1]), dtype=tdt, device="cuda")
    base = flat.data_ptr()
    ptrs = (C.c_void_p * max(len(ids), 1))()
    addr = base + offs[:-1] * dtype.itemsize
    for k in range(len(ids)):
        ptrs[k] = int(addr[k])
    sym = None
    if symmetrise is not None:
        sym = np.ascontiguousarray(symmetrise, dtype=np.int32)
    I = C.POINTER(C.c_int64)
    L.check(L.lib().bsm_synth_blocks(
        {np.dtype(np.float32): L.BSM_F32, np.dtype(np.float64): L.BSM_F64}[dtype], int(seed), len(ids),
        ids.ctypes.data_as(I), ms.ctypes.data_as(I), ns.ctypes.data_as(I),
        sym.ctypes.data_as(C.POINTER(C.c_int32)) if sym is not None else None, ptrs,
        C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    out = []
    for k in range(len(ids)):
        m, n = int(ms[k]), int(ns[k])
        out.append(flat[int(offs[k]):int(offs[k]) + m * n].view(n, m).t())
    return out


def device_vector(seed, n, dtype=np.float64, first=0):
    """x[first : first + n] of config `seed`, generated on the current GPU (== vector(seed, ...)[first:first+n])."""
    import ctypes as C
    import torch
    from . import _lib as L
    dtype = np.dtype(dtype)
    tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}[dtype]
    out = torch.empty(int(n), dtype=tdt, device="cuda")
    L.check(L.lib().bsm_synth_vector(
        {np.dtype(np.float32): L.BSM_F32, np.dtype(np.float64): L.BSM_F64}[dtype], int(seed), int(first), int(n),
        C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


def config1(n=1000, nblocks=50, bs=32, dtype=np.float64, seed=0xB5A1):
    """C1: BlockSparseMatrix n x n, nblocks blocks bs x bs; row and column lists are bs distinct
    indices drawn without replacement from 1:n, unsorted (partial Fisher-Yates on the stream)."""
    blocks, rows, cols = [], [], []
    ctr = 0
    for b in range(nblocks):
        lists = []
        for _ in range(2):
            perm = np.arange(1, n + 1, dtype=np.int64)
            u = draws(seed, ctr, bs)
            ctr += bs
            for k in range(bs):
                j = k + int(u[k] % np.uint64(n - k))
                perm[k], perm[j] = perm[j], perm[k]
            lists.append(perm[:bs].copy())
        rows.append(lists[0])
        cols.append(lists[1])
        blocks.append(block_values(seed, b, bs, bs, dtype))
    return dict(kind="blocksparse", blocks=blocks, rowindices=rows, colindices=cols, size=(n, n),
                x=vector(seed, n, dtype))


def config2(n=100_000, nblocks=5000, lo=8, hi=64, dtype=np.float64, seed=0xB5A2, part=None, on_device=False):
    """C2: VBCRS n x n; rows and cols each cut into consecutive segments of size U{lo..hi};
    nblocks distinct (row-seg, col-seg) pairs drawn uniformly; block = seg-height x seg-width.

    part=(rank, nparts): weak-scaling form.  The GLOBAL operator is (n*nparts)^2 with
    nblocks*nparts blocks; the returned problem holds only the blocks whose row segment starts in
    rank's 1/nparts slice of the rows (block rows are the independent units of the VBCRS product,
    reference src/vbcrs.jl:275-283), bit-identical to what a single process would generate, plus
    `own` = the 1-based inclusive row range this rank is responsible for."""
    rank, nparts = part if part is not None else (0, 1)
    n_per, n = n, n * nparts
    nblocks = nblocks * nparts
    rstart, rsz = _segments(seed, 0, n, lo, hi)
    cstart, csz = _segments(seed, 1 << 20, n, lo, hi)
    nr, nc = len(rsz), len(csz)
    nblocks = min(nblocks, nr * nc)
    # distinct pairs: draw until nblocks unique, in stream order
    pairs, seen, off = [], set(), 2 << 20
    while len(pairs) < nblocks:
        u = draws(seed, off, 2 * nblocks)
        off += 2 * nblocks
        for k in range(nblocks):
            p = (int(u[2 * k] % np.uint64(nr)), int(u[2 * k + 1] % np.uint64(nc)))
            if p not in seen:
                seen.add(p)
                pairs.append(p)
                if len(pairs) == nblocks:
                    break
    own = None
    keep = list(range(len(pairs)))
    if part is not None:
        seg_lo = int(np.searchsorted(rstart, rank * n_per, side="left"))
        seg_hi = int(np.searchsorted(rstart, (rank + 1) * n_per, side="left")) if rank + 1 < nparts else nr
        keep = [b for b, (i, _) in enumerate(pairs) if seg_lo <= i < seg_hi]
        own = (int(rstart[seg_lo]) + 1, int(rstart[seg_hi]) if seg_hi < nr else n)
    if on_device:  # blocks and x generated in HBM (torch CUDA tensors), bit-identical values
        blocks = device_blocks(seed, keep, [int(rsz[pairs[b][0]]) for b in keep],
                               [int(csz[pairs[b][1]]) for b in keep], dtype)
    else:
        blocks = [block_values(seed, b, int(rsz[pairs[b][0]]), int(csz[pairs[b][1]]), dtype) for b in keep]
    rowstart = np.array([rstart[pairs[b][0]] + 1 for b in keep], dtype=np.int64)
    colstart = np.array([cstart[pairs[b][1]] + 1 for b in keep], dtype=np.int64)
    out = dict(kind="vbcrs", blocks=blocks, rowstart=rowstart, colstart=colstart, size=(n, n),
               x=device_vector(seed, n, dtype) if on_device else vector(seed, n, dtype))
    if own is not None:
        out["own"] = own
    return out


def config3(nseg=3125, bs=64, halfband=8, dtype=np.float64, seed=0xB5A3, on_device=False):
    """C3: SymmetricBlockMatrix (nseg*bs)^2; nseg diagonal bs x bs blocks symmetrised (D + D^T)/2;
    off-diagonal (I, J) for J = I-1 .. I-halfband; contiguous index lists."""
    n = nseg * bs
    if on_device:
        return _banded_symmetric_on_device(seed, np.arange(nseg, dtype=np.int64) * bs,
                                           np.full(nseg, bs, dtype=np.int64), halfband, n, dtype)
    diag, didx, off, ridx, cidx = [], [], [], [], []
    b = 0
    for I in range(nseg):
        d = block_values(seed, b, bs, bs, dtype)
        b += 1
        diag.append(np.asfortranarray((d + d.T) / 2))
        didx.append(np.arange(I * bs + 1, (I + 1) * bs + 1, dtype=np.int64))
    for I in range(nseg):
        for k in range(1, halfband + 1):
            J = I - k
            if J < 0:
                continue
            off.append(block_values(seed, b, bs, bs, dtype))
            b += 1
            ridx.append(np.arange(I * bs + 1, (I + 1) * bs + 1, dtype=np.int64))
            cidx.append(np.arange(J * bs + 1, (J + 1) * bs + 1, dtype=np.int64))
    return dict(kind="symmetric", diagonals=diag, diagonalindices=didx, offdiagonals=off,
                rowindices=ridx, colindices=cidx, size=(n, n), x=vector(seed, n, dtype))


def _config4_cols(seed, I, ngrid, per_row):
    """block columns of block row I of C4: the diagonal one, then distinct uniform draws"""
    colsI = [I]
    u = draws(seed, I * 4 * per_row, 4 * per_row)
    k = 0
    while len(colsI) < min(per_row, ngrid):
        J = int(u[k % len(u)] % np.uint64(ngrid)) if k < len(u) else (colsI[-1] + 1) % ngrid
        k += 1
        if J not in colsI:
            colsI.append(J)
    return colsI


def config4_sample(block_rows=(), block_cols=(), ngrid=15625, bs=128, per_row=16, dtype=np.float32, seed=0xB5A4):
    """HOST sub-problem of C4 for a sampled-row check at full size: exactly the blocks of the given block rows (every
    contribution to those rows of A x) and the blocks of the given block columns (every contribution to those rows of
    A^T x), bit-identical to what config4(on_device=True) generates in HBM.  Returns (problem, row ranges, column
    ranges) with 1-based inclusive ranges of the sampled rows / columns; problem["x"] is the full x."""
    n = ngrid * bs
    want_r, want_c = set(int(i) for i in block_rows), set(int(j) for j in block_cols)
    blocks, rowstart, colstart, seen = [], [], [], set()
    for I in (range(ngrid) if want_c else sorted(want_r)):
        colsI = _config4_cols(seed, I, ngrid, per_row)
        for t, J in enumerate(colsI):
            if (I in want_r or J in want_c) and (I, t) not in seen:
                seen.add((I, t))
                blocks.append(block_values(seed, I * per_row + t, bs, bs, dtype))
                rowstart.append(I * bs + 1)
                colstart.append(J * bs + 1)
    prob = dict(kind="vbcrs", blocks=blocks, rowstart=np.array(rowstart, np.int64), colstart=np.array(colstart, np.int64),
                size=(n, n), x=vector(seed, n, dtype))
    return prob, [(I * bs + 1, (I + 1) * bs) for I in sorted(want_r)], [(J * bs + 1, (J + 1) * bs) for J in sorted(want_c)]


def config4(ngrid=15625, bs=128, per_row=16, dtype=np.float32, seed=0xB5A4, row_lo=0, row_hi=None,
            on_device=False):
    """C4: VBCRS (ngrid*bs)^2, bs x bs blocks, per_row distinct block columns per block row (one on
    the diagonal, the rest uniform).  row_lo/row_hi select a range of block rows (multi-GPU
    partition: every rank generates only its own rows, bit-identically)."""
    n = ngrid * bs
    row_hi = ngrid if row_hi is None else row_hi
    blocks, rowstart, colstart = [], [], []
    for I in range(row_lo, row_hi):
        colsI = _config4_cols(seed, I, ngrid, per_row)
        for t, J in enumerate(colsI):
            blocks.append(I * per_row + t if on_device else block_values(seed, I * per_row + t, bs, bs, dtype))
            rowstart.append(I * bs + 1)
            colstart.append(J * bs + 1)
    if on_device:
        blocks = device_blocks(seed, blocks, [bs] * len(blocks), [bs] * len(blocks), dtype)
    return dict(kind="vbcrs", blocks=blocks, rowstart=np.array(rowstart, np.int64),
                colstart=np.array(colstart, np.int64), size=(n, n),
                x=device_vector(seed, n, dtype) if on_device else vector(seed, n, dtype))


def _banded_symmetric_on_device(seed, start, sz, halfband, n, dtype, seg_lo=0, seg_hi=None):
    """Diagonal segments [seg_lo, seg_hi) of a banded symmetric config with their off-diagonal blocks
    (I, J), J = I-1 .. I-halfband, generated in HBM.  Block numbering as in config3 / config5: the
    diagonal blocks are b = 0 .. nseg-1, the off-diagonal ones follow in (I, k) order."""
    nseg = len(sz)
    seg_hi = nseg if seg_hi is None else seg_hi
    # number of off-diagonal blocks in front of segment I: sum_{i < I} min(i, halfband)
    I = np.arange(nseg + 1, dtype=np.int64)
    nbefore = np.where(I <= halfband, I * (I - 1) // 2, halfband * (halfband - 1) // 2 + (I - halfband) * halfband)
    ids, ms, ns, sym = [], [], [], []
    for i in range(seg_lo, seg_hi):
        ids.append(i)
        ms.append(int(sz[i]))
        ns.append(int(sz[i]))
        sym.append(1)
    didx = [np.arange(start[i] + 1, start[i] + sz[i] + 1, dtype=np.int64) for i in range(seg_lo, seg_hi)]
    ridx, cidx = [], []
    for i in range(seg_lo, seg_hi):
        for k in range(1, halfband + 1):
            j = i - k
            if j < 0:
                continue
            ids.append(int(nseg + nbefore[i] + (k - 1)))
            ms.append(int(sz[i]))
            ns.append(int(sz[j]))
            sym.append(0)
            ridx.append(didx[i - seg_lo])
            cidx.append(np.arange(start[j] + 1, start[j] + sz[j] + 1, dtype=np.int64))
    blocks = device_blocks(seed, ids, ms, ns, dtype, sym)
    nd = seg_hi - seg_lo
    return dict(kind="symmetric", diagonals=blocks[:nd], diagonalindices=didx, offdiagonals=blocks[nd:],
                rowindices=ridx, colindices=cidx, size=(n, n), x=device_vector(seed, n, dtype))


def config5_segments(n=5_000_000, lo=16, hi=256, seed=0xB5A5):
    """(start, size) of C5's diagonal segments (0-based starts)."""
    return _segments(seed, 0, n, lo, hi)


def config5(n=5_000_000, lo=16, hi=256, halfband=4, dtype=np.float64, seed=0xB5A5, on_device=False,
            seg_lo=0, seg_hi=None):
    """C5: SymmetricBlockMatrix n x n, segments U{lo..hi}, off-diagonal (I, J), J = I-1..I-halfband.
    on_device: blocks and x generated in HBM; seg_lo / seg_hi (on_device only) select a range of
    diagonal segments with their off-diagonal blocks (one rank's share of a row partition)."""
    start, sz = _segments(seed, 0, n, lo, hi)
    if on_device:
        return _banded_symmetric_on_device(seed, start, sz, halfband, n, dtype, seg_lo, seg_hi)
    nseg = len(sz)
    diag, didx, off, ridx, cidx = [], [], [], [], []
    b = 0
    for I in range(nseg):
        m = int(sz[I])
        d = block_values(seed, b, m, m, dtype)
        b += 1
        diag.append(np.asfortranarray((d + d.T) / 2))
        didx.append(np.arange(start[I] + 1, start[I] + m + 1, dtype=np.int64))
    for I in range(nseg):
        for k in range(1, halfband + 1):
            J = I - k
            if J < 0:
                continue
            off.append(block_values(seed, b, int(sz[I]), int(sz[J]), dtype))
            b += 1
            ridx.append(np.arange(start[I] + 1, start[I] + sz[I] + 1, dtype=np.int64))
            cidx.append(np.arange(start[J] + 1, start[J] + sz[J] + 1, dtype=np.int64))
    return dict(kind="symmetric", diagonals=diag, diagonalindices=didx, offdiagonals=off,
                rowindices=ridx, colindices=cidx, size=(n, n), x=vector(seed, n, dtype))


def config5_sample(segs, n=5_000_000, lo=16, hi=256, halfband=4, dtype=np.float64, seed=0xB5A5):
    """HOST sub-problem of C5 for a sampled-row check at full size: for every sampled diagonal segment I its diagonal
    block, its off-diagonal blocks (I, J), J = I-1 .. I-halfband (the forward sweep into rows I) and the blocks (I', I),
    I' = I+1 .. I+halfband (whose transposes reach rows I) -- every contribution to y[rows of I] and nothing else is
    complete.  Values are bit-identical to config5(on_device=True).  Returns (problem, row ranges 1-based inclusive)."""
    start, sz = _segments(seed, 0, n, lo, hi)
    nseg = len(sz)
    Iall = np.arange(nseg + 1, dtype=np.int64)
    nbefore = np.where(Iall <= halfband, Iall * (Iall - 1) // 2, halfband * (halfband - 1) // 2 + (Iall - halfband) * halfband)
    segs = sorted(set(int(i) for i in segs))
    diag, didx, off, ridx, cidx, pairs = [], [], [], [], [], set()
    rng_of = lambda i: np.arange(start[i] + 1, start[i] + sz[i] + 1, dtype=np.int64)
    for I in segs:
        m = int(sz[I])
        d = block_values(seed, I, m, m, dtype)
        diag.append(np.asfortranarray((d + d.T) / 2))
        didx.append(rng_of(I))
        for (i, k) in [(I, k) for k in range(1, halfband + 1)] + [(I + k, k) for k in range(1, halfband + 1)]:
            j = i - k
            if j < 0 or i >= nseg or (i, j) in pairs:
                continue
            pairs.add((i, j))
            off.append(block_values(seed, int(nseg + nbefore[i] + (k - 1)), int(sz[i]), int(sz[j]), dtype))
            ridx.append(rng_of(i))
            cidx.append(rng_of(j))
    prob = dict(kind="symmetric", diagonals=diag, diagonalindices=didx, offdiagonals=off, rowindices=ridx, colindices=cidx,
                size=(n, n), x=vector(seed, n, dtype))
    return prob, [(int(start[I]) + 1, int(start[I] + sz[I])) for I in segs]


def sample_ids(count, total, seed=12345, lo=0, hi=None):
    """`count` distinct ids of lo .. hi-1 (default 0 .. total-1): the two ends and a seeded spread between them"""
    hi = total if hi is None else hi
    if hi - lo <= count:
        return list(range(lo, hi))
    rng = np.random.default_rng(seed)
    ids = set([lo, hi - 1]) | set(int(v) for v in rng.choice(np.arange(lo, hi), size=count, replace=False)[:count - 2])
    return sorted(ids)


def build(problem, **kw):
    """Instantiate the matching matrix type from a config dict."""
    from . import matrices as M
    k = problem["kind"]
    if k == "blocksparse":
        return M.BlockSparseMatrix(problem["blocks"], problem["rowindices"], problem["colindices"],
                                   problem["size"], **kw)
    if k == "vbcrs":
        return M.VariableBlockCompressedRowStorage(problem["blocks"], problem["rowstart"],
                                                   problem["colstart"], problem["size"], **kw)
    if k == "symmetric":
        return M.SymmetricBlockMatrix(problem["diagonals"], problem["diagonalindices"],
                                      problem["offdiagonals"], problem["rowindices"],
                                      problem["colindices"], problem["size"], **kw)
    raise ValueError(k)
