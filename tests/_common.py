"""Shared test helpers: fixture reader, oracle drivers, and a numpy interpreter of the packed
device image (so the host analysis is checked on CPU, without a GPU).  Test code only."""
import ctypes as C
import os
import struct

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
N, T, Cc = 0, 1, 2  # ops


def read_fixture(key):
    """tests/golden/symmetric_<key>.bin (decoded from the reference's
    test/assets/symmetricblockexamples.jld2 by tests/golden/decode_jld2.c) ->
    (diagonalblocks, selfindices, offblocks, testindices, trialindices), ComplexF64, 1-based."""
    b = open(os.path.join(GOLDEN, f"symmetric_{key}.bin"), "rb").read()
    assert b[:8] == b"BSMFIX01"
    off = 8

    def i64():
        nonlocal off
        v = struct.unpack_from("<q", b, off)[0]
        off += 8
        return v

    fields = []
    for ismat in (1, 0, 1, 0, 0):
        items = []
        for _ in range(i64()):
            if ismat:
                m, n = i64(), i64()
                a = np.frombuffer(b, dtype="<c16", count=m * n, offset=off).reshape((m, n), order="F")
                off += 16 * m * n
            else:
                ln = i64()
                a = np.frombuffer(b, dtype="<i8", count=ln, offset=off)
                off += 8 * ln
            items.append(a.copy())
        fields.append(items)
    assert off == len(b)
    return fields


def fixture_problem(key, dtype=np.complex128, part="full"):
    d, si, o, ti, tr = read_fixture(key)
    if part == "real":
        d, o = [x.real for x in d], [x.real for x in o]
    elif part == "imag":
        d, o = [x.imag for x in d], [x.imag for x in o]
    d = [np.asfortranarray(x, dtype=dtype) for x in d]
    o = [np.asfortranarray(x, dtype=dtype) for x in o]
    n = max(max(int(i.max()) for i in ti), max(int(i.max()) for i in tr), max(int(i.max()) for i in si))
    return dict(kind="symmetric", diagonals=d, diagonalindices=si, offdiagonals=o, rowindices=ti,
                colindices=tr, size=(n, n))


def fixture_as_blocksparse(key, dtype=np.complex128, part="full"):
    """The fixture's off-diagonal panels as a plain BlockSparseMatrix (the missing
    blockexamples.jld2 has the same character: unsorted, non-contiguous lists)."""
    p = fixture_problem(key, dtype, part)
    return dict(kind="blocksparse", blocks=p["offdiagonals"], rowindices=p["rowindices"],
                colindices=p["colindices"], size=p["size"])


def rand_vec(rng, n, dtype):
    dtype = np.dtype(dtype)
    v = rng.standard_normal(n)
    if dtype.kind == "c":
        v = v + 1j * rng.standard_normal(n)
    return v.astype(dtype)


def relerr(a, b):
    """the reference's own norm: max|a-b| / max|b|  (test/test_vbcrs.jl:35)"""
    if len(b) == 0:
        return 0.0
    den = float(np.max(np.abs(b)))
    if den == 0:
        den = 1.0
    return float(np.max(np.abs(a - b)) / den)


def sampled_relerr(got, ref, ranges):
    """max |got - ref| / max |ref| over the sampled 1-based inclusive row ranges only (a sub-problem's oracle product
    is complete on exactly those rows: synthetic.config4_sample / config5_sample); `got` may be a torch tensor"""
    num = den = 0.0
    for a, b in ranges:
        g = got[a - 1:b]
        g = g.cpu().numpy() if hasattr(g, "cpu") else np.asarray(g)
        r = ref[a - 1:b]
        if not np.all(np.isfinite(g)):
            return float("inf")
        num = max(num, float(np.max(np.abs(g - r))))
        den = max(den, float(np.max(np.abs(r))))
    return num / (den if den > 0 else 1.0)


def single_color(n):
    return [list(range(1, n + 1))]


def oracle_mul(orc, problem, op, x, y0, alpha=1, beta=0, strong=True, colorsets=None, prepare=False):
    """y = alpha*op(A)*x + beta*y0 through the CPU oracle (reference loop structure).
    prepare=True: returns a zero-argument closure with every argument marshalled once (timing loops)."""
    y = np.array(y0, copy=True)
    k = problem["kind"]
    if k == "blocksparse":
        nb = len(problem["blocks"])
        cs = colorsets if colorsets is not None else (single_color(nb), single_color(nb))
        return orc.bsm_mul(op, problem["blocks"], problem["rowindices"], problem["colindices"],
                           cs[0] if op == N else cs[1], x, y, alpha, beta, strong, prepare=prepare)
    if k == "symmetric":
        nd, no = len(problem["diagonals"]), len(problem["offdiagonals"])
        cs = colorsets if colorsets is not None else (single_color(no), single_color(no), single_color(nd))
        return orc.sym_mul(op, problem["diagonals"], problem["diagonalindices"], problem["offdiagonals"],
                           problem["rowindices"], problem["colindices"], cs, x, y, alpha, beta, strong,
                           prepare=prepare)
    if k == "vbcrs":
        perm, rowptr, colind, rowind = orc.vbcrs_build(problem["rowstart"], problem["colstart"])
        blocks = [problem["blocks"][p - 1] for p in perm]
        return orc.vbcrs_mul(op, blocks, rowptr, colind, rowind, x, y, alpha, beta, strong, prepare=prepare)
    raise ValueError(k)


def coo_of(problem):
    """independent COO triples (1-based), the reference's test-oracle path (src/sparse.jl)."""
    rows, cols, vals = [], [], []

    def push(b, ri, ci):
        R, K = np.meshgrid(np.asarray(ri), np.asarray(ci), indexing="ij")
        rows.append(R.ravel())
        cols.append(K.ravel())
        vals.append(np.asarray(b).ravel())

    k = problem["kind"]
    if k == "blocksparse":
        for b, r, c in zip(problem["blocks"], problem["rowindices"], problem["colindices"]):
            push(b, r, c)
    elif k == "symmetric":
        for b, d in zip(problem["diagonals"], problem["diagonalindices"]):
            push(b, d, d)
        for b, r, c in zip(problem["offdiagonals"], problem["rowindices"], problem["colindices"]):
            push(b, r, c)
            push(b.T, c, r)
    else:
        for b, r0, c0 in zip(problem["blocks"], problem["rowstart"], problem["colstart"]):
            push(b, np.arange(r0, r0 + b.shape[0]), np.arange(c0, c0 + b.shape[1]))
    if not rows:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0)
    return np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)


def scipy_mul(problem, op, x, y0, alpha=1, beta=0, strong=True):
    """scipy.sparse as an independent checker (never the oracle, never shipped)."""
    import scipy.sparse as sp
    r, c, v = coo_of(problem)
    S = sp.coo_matrix((v, (r - 1, c - 1)), shape=problem["size"]).tocsr()
    if op == T:
        S = S.T
    elif op == Cc:
        S = S.conj().T
    base = np.zeros_like(y0) if strong else beta * y0
    return alpha * (S @ x) + base


# ---- packed-image interpreter -------------------------------------------------------------------
PIECE_DT = np.dtype([("val_off", "<u8"), ("xbase", "<i4"), ("col_off", "<i4"), ("nstrips", "<i4"),
                     ("ncols", "<i4"), ("kind", "<i4"), ("seg2_x", "<i4")])
WAVE_DT = np.dtype([("seg1_w", "<i4"), ("win_base", "<i4"), ("row_off", "<i4"), ("rbase", "<i4"),
                    ("m", "<u2"), ("work", "u1"), ("grp", "u1"), ("lead", "u1"), ("wg_sync", "u1"),
                    ("npieces", "u1"), ("win_span8", "u1"),
                    ("seg1_x", "<i4"), ("seg2_w", "<i4"), ("first", PIECE_DT)])
assert PIECE_DT.itemsize == 32 and WAVE_DT.itemsize == 64
WORK_NOP, WORK_PANEL, WORK_SCALE = 0, 1, 2
KIND_DIAG, KIND_OFF = 1, 2
KIND_HAS_OFF = 1 << 8
KIND_GROUP_HAS_OFF = 1 << 9


def get_image(A, timage=False, multi=False):
    """multi: the coarser wave records the multi-RHS kernels walk (bsm_get_image 8) in place of the ordinary ones
    (an empty list means the ordinary records serve both)"""
    from bsm_amd import _lib as L
    out = []
    for which, dt in ((0, np.uint8), (1, np.int32), (2, np.int32), (8 if multi else 3, WAVE_DT)):
        which += 16 if timage else 0
        n = C.c_int64(0)
        L.check(L.lib().bsm_get_image(A._h.ptr, which, None, C.byref(n)))
        buf = np.zeros(max(n.value, 1), dtype=np.uint8)
        L.check(L.lib().bsm_get_image(A._h.ptr, which, buf.ctypes.data, C.byref(n)))
        out.append(buf[:n.value].view(dt))
    return out


def get_inverted_index(A, timage=False):
    """gather handles: ((ptrN, idxN), (ptrT, idxT)) or None"""
    from bsm_amd import _lib as L
    out = []
    for which, dt in ((4, np.int64), (5, np.int32), (6, np.int64), (7, np.int32)):
        which += 16 if timage else 0
        n = C.c_int64(0)
        L.check(L.lib().bsm_get_image(A._h.ptr, which, None, C.byref(n)))
        buf = np.zeros(max(n.value, 1), dtype=np.uint8)
        L.check(L.lib().bsm_get_image(A._h.ptr, which, buf.ctypes.data, C.byref(n)))
        out.append(buf[:n.value].view(dt))
    if len(out[0]) == 0:
        return None
    return (out[0], out[1]), (out[2], out[3])


def _image_exclusive(waves, rows, ylen):
    """every y row produced by at most one lead wave group (the library's exclusivity proof)"""
    seen = np.zeros(ylen, dtype=np.int32)
    for W in waves[(waves["work"] == WORK_PANEL) & (waves["lead"] == 1)]:
        m = int(W["m"])
        r = (np.arange(W["rbase"], W["rbase"] + m) if W["rbase"] >= 0 else rows[W["row_off"]:W["row_off"] + m])
        np.add.at(seen, r, 1)
    return bool(np.all(seen <= 1))


def interpret_image(A, op, x, y0, alpha=1, beta=0, strong=True, timage=False, multi=False):
    """Executes the packed image the way the HIP kernel walks it (same descriptors and index
    arithmetic, numpy arithmetic) -- checks packing + schedule on CPU.
    timage: run op T / C as a FORWARD product on the handle's second (transposed) ordering.
    multi: walk the multi-RHS kernels' coarser wave records instead."""
    values, rows, cols, waves = get_image(A, timage, multi)
    dt = A.dtype
    E = 16 // dt.itemsize
    vals = values.view(dt)
    st = A.stats()
    conj = op == Cc
    if timage:
        assert op != N
        opT = False
        works = waves["work"]
        direct = bool(np.any(works == WORK_SCALE)) or _image_exclusive(waves, rows, len(y0))
    else:
        opT = op != N
        direct = (not opT) and st["exclusive"] == 1
    y = np.array(y0, copy=True)
    inv = get_inverted_index(A, timage)
    gather = inv is not None
    if gather:
        assert not direct
        ws = np.full(len(cols) + int(np.sum(waves["m"][(waves["work"] == WORK_PANEL) & (waves["lead"] == 1)])) + 8,
                     np.nan, dtype=dt)  # NaN: reading a slot nobody wrote must show
        fbase = len(cols) if np.any(waves["work"] == WORK_PANEL) else 0
    elif not direct:
        y[:] = 0 if strong else beta * y
    wpw = int(st["ntasks"]) // max(int(st["nworkgroups"]), 1) if len(waves) else 4  # waves per workgroup
    assert wpw in (4, 8) and len(waves) % wpw == 0
    pw = waves[waves["work"] == WORK_PANEL]
    has_off = bool(np.any(pw["first"]["kind"][pw["npieces"] > 0] & KIND_HAS_OFF))
    fwd_kernel = (not opT) or has_off  # the launcher's choice of the FWD template flag
    for wg in range(len(waves) // wpw):
        us = [None] * wpw
        for w in range(wpw):
            W = waves[wg * wpw + w]
            if W["work"] == WORK_SCALE:
                if direct:
                    r0, cnt = int(W["rbase"]), int(W["first"]["ncols"])
                    y[r0:r0 + cnt] = 0 if strong else beta * y[r0:r0 + cnt]
                continue
            if W["work"] != WORK_PANEL:
                continue
            m = int(W["m"])
            ridx = (np.arange(W["rbase"], W["rbase"] + m) if W["rbase"] >= 0
                    else rows[W["row_off"]:W["row_off"] + m])
            u = np.zeros(m, dtype=dt)
            assert int(W["npieces"]) in (0, 1)
            for pi in range(int(W["npieces"])):
                P = W["first"]
                ns, nc = int(P["nstrips"]), int(P["ncols"])
                base = int(P["val_off"]) * 16 // dt.itemsize
                full = vals[base:base + ns * m * E].reshape(ns, m, E).transpose(1, 0, 2).reshape(m, ns * E)
                B = full[:, :nc]
                assert not np.any(full[:, nc:]), "strip padding must be zero"
                if conj:
                    B = B.conj()
                kinds = int(P["kind"])
                wv = np.arange(nc)
                pool = cols[P["col_off"]:P["col_off"] + nc]
                if P["xbase"] >= 0:  # up to three inline contiguous runs, each of one kind
                    s1w, s2w = int(W["seg1_w"]), int(W["seg2_w"])
                    cidx = np.where(wv < s1w, int(P["xbase"]) + wv,
                                    np.where(wv < s2w, int(W["seg1_x"]) + wv - s1w, int(P["seg2_x"]) + wv - s2w))
                    ckind = np.where(wv < s1w, kinds & 3, np.where(wv < s2w, (kinds >> 2) & 3, (kinds >> 4) & 3))
                    assert np.array_equal(cidx, pool & 0x7fffffff)
                else:  # cols pool; a set sign bit flags a diagonal column of a mixed panel
                    cidx = pool & 0x7fffffff
                    ckind = np.where(pool < 0, KIND_DIAG, kinds & 3)
                assert bool(kinds & KIND_HAS_OFF) == bool(np.any(ckind == KIND_OFF))
                off = ckind == KIND_OFF
                fcols = np.ones(nc, bool) if not opT else off      # columns used by the forward product
                tcols = off if not opT else np.ones(nc, bool)      # columns used by the transposed product
                if np.any(fcols):
                    u += B[:, fcols] @ x[cidx[fcols]]
                if np.any(tcols):
                    v = B[:, tcols].T @ x[ridx]
                    if gather:  # one store per column sum, slot = position in the cols pool
                        ws[int(P["col_off"]) + np.nonzero(tcols)[0]] = v
                    else:
                        np.add.at(y, cidx[tcols], alpha * v)
            us[w] = (u, ridx, int(W["grp"]), int(W["lead"]))
        for w in range(wpw):
            if us[w] is None or not us[w][3] or not fwd_kernel:
                continue
            u, ridx, grp, _ = us[w]
            u = u.copy()
            for k in range(1, grp):
                assert us[w + k] is not None and not us[w + k][3]
                assert np.array_equal(us[w + k][1], ridx)
                u += us[w + k][0]
            if gather:
                lead = waves[wg * wpw + w]
                if (not opT) or (int(lead["first"]["kind"]) & KIND_GROUP_HAS_OFF):
                    ws[fbase + int(lead["win_base"]):fbase + int(lead["win_base"]) + len(ridx)] = u
            elif direct:
                y[ridx] = alpha * u if strong else beta * y[ridx] + alpha * u
            else:
                np.add.at(y, ridx, alpha * u)
    if gather:  # second launch: fixed-order sums per y entry
        ptr, idx = inv[1] if opT else inv[0]  # ELL per 64-row tile, -1 = no contribution
        assert len(ptr) == (len(y) + 63) // 64 + 1
        sums = np.zeros(len(y), dtype=dt)
        for j in range(len(y)):
            t, lane = divmod(j, 64)
            sl = idx[ptr[t] * 64 + lane:ptr[t + 1] * 64 + lane:64]
            sl = sl[sl >= 0]
            sums[j] = ws[sl].sum() if len(sl) else 0
        assert not np.any(np.isnan(sums)), "a contributing workspace slot was never written"
        y = alpha * sums if strong else beta * y + alpha * sums
    return y
