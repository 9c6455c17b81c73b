"""ctypes front-end of oracle/libbsm_oracle.so (TEST INFRASTRUCTURE, never shipped).

Each method is a thin marshalling layer over one C function of bsm_oracle.c /
bsm_oracle_impl.h, which carry the reference citations.  All index lists are 1-based
int64 numpy arrays, blocks are column-major (Fortran-order) 2-D numpy arrays, exactly the
data a Julia caller of the reference holds.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SFX = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64",
        np.dtype(np.complex64): "c64", np.dtype(np.complex128): "c128"}
_I64P = C.POINTER(C.c_int64)


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libbsm_oracle.so"])


def load_oracle(build=True):
    path = os.path.join(_HERE, "libbsm_oracle.so")
    if build and not os.path.exists(path):
        build_oracle()
    return Oracle(C.CDLL(path))


def load_oracle_native():
    """The -O3 -march=native build (cpu_baseline timing only; compiled on the machine it runs on)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "_native/libbsm_oracle_native.so"])
    return Oracle(C.CDLL(os.path.join(_HERE, "_native", "libbsm_oracle_native.so")))


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _ptr_array(arrs):
    """void*[] over a list of numpy arrays (kept alive by the caller)."""
    out = (C.c_void_p * max(len(arrs), 1))()
    for i, a in enumerate(arrs):
        out[i] = a.ctypes.data
    return out


def _scalar(dtype, v):
    """Pass a T by value: real -> c_float/c_double, complex -> struct of two."""
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return C.c_float(float(np.real(v)))
    if dtype == np.float64:
        return C.c_double(float(np.real(v)))
    base = C.c_float if dtype == np.complex64 else C.c_double

    class _Cx(C.Structure):
        _fields_ = [("re", base), ("im", base)]

    v = complex(v)
    return _Cx(v.real, v.imag)


def _fblocks(blocks, dtype):
    return [np.asfortranarray(b, dtype=dtype) for b in blocks]


def _colors_csr(colors):
    """list of 1-based id lists -> (ncolors, ptr, blk)"""
    ptr = np.zeros(len(colors) + 1, dtype=np.int64)
    for i, c in enumerate(colors):
        ptr[i + 1] = ptr[i] + len(c)
    blk = _i64(np.concatenate([np.asarray(c, dtype=np.int64) for c in colors])
               if len(colors) else np.zeros(0, np.int64))
    return len(colors), ptr, blk


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.orc_vbcrs_build.restype = C.c_int64
        lib.orc_color_dsatur.restype = C.c_int64
        lib.orc_color_check.restype = C.c_int64

    # -- bookkeeping -----------------------------------------------------------------
    def vbcrs_build(self, rowstart, colstart):
        rs, cs = _i64(rowstart), _i64(colstart)
        n = len(rs)
        perm = np.zeros(n, np.int64)
        rowptr = np.zeros(n + 1, np.int64)
        colind = np.zeros(n, np.int64)
        rowind = np.zeros(n, np.int64)
        nbr = self.lib.orc_vbcrs_build(C.c_int64(n), rs.ctypes.data_as(_I64P),
                                       cs.ctypes.data_as(_I64P), perm.ctypes.data_as(_I64P),
                                       rowptr.ctypes.data_as(_I64P),
                                       colind.ctypes.data_as(_I64P),
                                       rowind.ctypes.data_as(_I64P))
        return perm, rowptr[:nbr + 1].copy(), colind, rowind[:nbr].copy()

    def color_dsatur(self, lists):
        lists = [_i64(l) for l in lists]
        n = len(lists)
        if n == 0:
            return []
        lens = _i64([len(l) for l in lists])
        maxindex = int(max(int(l.max()) for l in lists if len(l)))
        col = np.zeros(n, np.int64)
        nc = self.lib.orc_color_dsatur(C.c_int64(n), _ptr_array(lists),
                                       lens.ctypes.data_as(_I64P), C.c_int64(maxindex),
                                       col.ctypes.data_as(_I64P))
        return [list(np.nonzero(col == c)[0] + 1) for c in range(nc)]

    def color_workstream(self, lists):
        """WorkstreamDSATUR per the published algorithm (orc_color_workstream) -> classes of 1-based ids."""
        lists = [_i64(l) for l in lists]
        n = len(lists)
        if n == 0:
            return []
        lens = _i64([len(l) for l in lists])
        maxindex = int(max([int(l.max()) for l in lists if len(l)] + [0]))
        col = np.zeros(n, np.int64)
        self.lib.orc_color_workstream.restype = C.c_int64
        nc = self.lib.orc_color_workstream(C.c_int64(n), _ptr_array(lists), lens.ctypes.data_as(_I64P),
                                           C.c_int64(maxindex), col.ctypes.data_as(_I64P))
        return [list(np.nonzero(col == c)[0] + 1) for c in range(nc)]

    def color_check(self, lists, colors):
        """colors: list of classes (1-based ids). True iff valid partition + conflict-free."""
        lists = [_i64(l) for l in lists]
        n = len(lists)
        col = np.full(n, -1, np.int64)
        seen = 0
        for c, cls in enumerate(colors):
            for b in cls:
                if col[b - 1] != -1:
                    return False
                col[b - 1] = c
                seen += 1
        if seen != n or (col < 0).any():
            return False
        if n == 0:
            return True
        lens = _i64([len(l) for l in lists])
        maxindex = int(max(int(l.max()) for l in lists if len(l)))
        bad = self.lib.orc_color_check(C.c_int64(n), _ptr_array(lists),
                                       lens.ctypes.data_as(_I64P), C.c_int64(maxindex),
                                       col.ctypes.data_as(_I64P))
        return bad == 0

    # -- products --------------------------------------------------------------------
    def bsm_mul(self, op, blocks, rowidx, colidx, colors, x, y, alpha=1, beta=0,
                strong_zero=True, prepare=False):
        dt = np.dtype(x.dtype)
        fb = _fblocks(blocks, dt)
        ri = [_i64(r) for r in rowidx]
        ci = [_i64(c) for c in colidx]
        m = _i64([b.shape[0] for b in fb])
        n = _i64([b.shape[1] for b in fb])
        ld = _i64([max(b.shape[0], 1) for b in fb])
        nc, cp, cb = _colors_csr(colors)
        x = np.ascontiguousarray(x, dt)
        assert y.dtype == dt and y.flags.c_contiguous
        fn = getattr(self.lib, "orc_bsm_mul_" + _SFX[dt])
        fn.restype = None
        args = [C.c_int(op), C.c_int64(len(y)), C.c_int64(len(fb)), _ptr_array(fb),
                m.ctypes.data_as(_I64P), n.ctypes.data_as(_I64P), ld.ctypes.data_as(_I64P),
                _ptr_array(ri), _ptr_array(ci), C.c_int64(nc), cp.ctypes.data_as(_I64P),
                cb.ctypes.data_as(_I64P), C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data),
                _scalar(dt, alpha), _scalar(dt, beta), C.c_int(1 if strong_zero else 0)]
        if prepare:  # everything marshalled once: the returned closure is one C call (timing loops)
            keep = (fb, ri, ci, m, n, ld, cp, cb, x, y)
            return lambda: fn(*args) or keep and None
        fn(*args)
        return y

    def sym_mul(self, op, diag, didx, off, rowidx, colidx, colorsets, x, y, alpha=1, beta=0,
                strong_zero=True, prepare=False):
        """colorsets = (offdiagonalcolors, transposeoffdiagonalcolors, diagonalcolors)."""
        dt = np.dtype(x.dtype)
        fd = _fblocks(diag, dt)
        fo = _fblocks(off, dt)
        di = [_i64(d) for d in didx]
        ri = [_i64(r) for r in rowidx]
        ci = [_i64(c) for c in colidx]
        ds = _i64([b.shape[0] for b in fd])
        dld = _i64([max(b.shape[0], 1) for b in fd])
        m = _i64([b.shape[0] for b in fo])
        n = _i64([b.shape[1] for b in fo])
        ld = _i64([max(b.shape[0], 1) for b in fo])
        csr = [_colors_csr(c) for c in colorsets]
        ncol = _i64([c[0] for c in csr])
        cptr = _ptr_array([c[1] for c in csr])
        cblk = _ptr_array([c[2] for c in csr])
        x = np.ascontiguousarray(x, dt)
        assert y.dtype == dt and y.flags.c_contiguous
        fn = getattr(self.lib, "orc_sym_mul_" + _SFX[dt])
        fn.restype = None
        args = [C.c_int(op), C.c_int64(len(y)), C.c_int64(len(fd)), _ptr_array(fd),
                ds.ctypes.data_as(_I64P), dld.ctypes.data_as(_I64P), _ptr_array(di),
                C.c_int64(len(fo)), _ptr_array(fo), m.ctypes.data_as(_I64P),
                n.ctypes.data_as(_I64P), ld.ctypes.data_as(_I64P), _ptr_array(ri), _ptr_array(ci),
                ncol.ctypes.data_as(_I64P), cptr, cblk, C.c_void_p(x.ctypes.data),
                C.c_void_p(y.ctypes.data), _scalar(dt, alpha), _scalar(dt, beta),
                C.c_int(1 if strong_zero else 0)]
        if prepare:
            keep = (fd, fo, di, ri, ci, ds, dld, m, n, ld, csr, ncol, x, y)
            return lambda: fn(*args) or keep and None
        fn(*args)
        return y

    def vbcrs_mul(self, op, blocks, rowptr, colindices, rowindices, x, y, alpha=1, beta=0,
                  strong_zero=True, parallel=False, prepare=False):
        """blocks already in VBCRS (sorted) order; rowptr/colindices/rowindices 1-based."""
        dt = np.dtype(x.dtype)
        fb = _fblocks(blocks, dt)
        m = _i64([b.shape[0] for b in fb])
        n = _i64([b.shape[1] for b in fb])
        ld = _i64([max(b.shape[0], 1) for b in fb])
        rp, ci, ri = _i64(rowptr), _i64(colindices), _i64(rowindices)
        x = np.ascontiguousarray(x, dt)
        assert y.dtype == dt and y.flags.c_contiguous
        args = [C.c_int64(len(y)), C.c_int64(len(rp) - 1), rp.ctypes.data_as(_I64P),
                ci.ctypes.data_as(_I64P), ri.ctypes.data_as(_I64P), _ptr_array(fb),
                m.ctypes.data_as(_I64P), n.ctypes.data_as(_I64P), ld.ctypes.data_as(_I64P),
                C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data), _scalar(dt, alpha),
                _scalar(dt, beta), C.c_int(1 if strong_zero else 0)]
        if op == 0:
            fn = getattr(self.lib, ("orc_vbcrs_mul_par_" if parallel else "orc_vbcrs_mul_") + _SFX[dt])
        else:
            fn = getattr(self.lib, "orc_vbcrs_mul_t_" + _SFX[dt])
            args = [C.c_int(op)] + args
        fn.restype = None
        if prepare:
            keep = (fb, m, n, ld, rp, ci, ri, x, y)
            return lambda: fn(*args) or keep and None
        fn(*args)
        return y

    def vbcrs_bench(self, blocks, rowptr, colindices, rowindices, x, y, seconds=10.0, parallel=False):
        """Timed forward products (fp64, 3-argument form) with everything marshalled ONCE: the loop
        and the clock live in C (orc_vbcrs_bench_f64).  Returns (reps, elapsed seconds)."""
        dt = np.dtype(np.float64)
        fb = _fblocks(blocks, dt)
        m = _i64([b.shape[0] for b in fb])
        n = _i64([b.shape[1] for b in fb])
        ld = _i64([max(b.shape[0], 1) for b in fb])
        rp, ci, ri = _i64(rowptr), _i64(colindices), _i64(rowindices)
        x = np.ascontiguousarray(x, dt)
        assert y.dtype == dt and y.flags.c_contiguous
        fn = self.lib.orc_vbcrs_bench_f64
        fn.restype = C.c_double
        ptrs = _ptr_array(fb)
        args = [C.c_int64(len(y)), C.c_int64(len(rp) - 1), rp.ctypes.data_as(_I64P), ci.ctypes.data_as(_I64P),
                ri.ctypes.data_as(_I64P), ptrs, m.ctypes.data_as(_I64P), n.ctypes.data_as(_I64P),
                ld.ctypes.data_as(_I64P), C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data)]
        par = C.c_int(1 if parallel else 0)
        t1 = fn(C.c_int64(3), par, *args) / 3  # calibrate
        reps = max(3, int(seconds / max(t1, 1e-6)))
        return reps, fn(C.c_int64(reps), par, *args)

    def sym_bench(self, diag, didx, off, rowidx, colidx, x, y, seconds=5.0):
        """Timed products S * x of a SymmetricBlockMatrix (3-argument form, the reference's three sweeps, serial colour
        sets) with everything marshalled ONCE: the loop and the clock live in C (orc_sym_bench_*).
        Returns (reps, elapsed seconds)."""
        dt = np.dtype(x.dtype)
        fd = _fblocks(diag, dt)
        fo = _fblocks(off, dt)
        di = [_i64(d) for d in didx]
        ri = [_i64(r) for r in rowidx]
        ci = [_i64(c) for c in colidx]
        ds = _i64([b.shape[0] for b in fd])
        dld = _i64([max(b.shape[0], 1) for b in fd])
        m = _i64([b.shape[0] for b in fo])
        n = _i64([b.shape[1] for b in fo])
        ld = _i64([max(b.shape[0], 1) for b in fo])
        single = lambda k: [list(range(1, k + 1))]
        csr = [_colors_csr(c) for c in (single(len(fo)), single(len(fo)), single(len(fd)))]
        ncol = _i64([c[0] for c in csr])
        cptr = _ptr_array([c[1] for c in csr])
        cblk = _ptr_array([c[2] for c in csr])
        x = np.ascontiguousarray(x, dt)
        assert y.dtype == dt and y.flags.c_contiguous
        fn = getattr(self.lib, "orc_sym_bench_" + _SFX[dt])
        fn.restype = C.c_double
        keep = (_ptr_array(fd), _ptr_array(di), _ptr_array(fo), _ptr_array(ri), _ptr_array(ci))
        args = [C.c_int64(len(y)), C.c_int64(len(fd)), keep[0], ds.ctypes.data_as(_I64P), dld.ctypes.data_as(_I64P), keep[1],
                C.c_int64(len(fo)), keep[2], m.ctypes.data_as(_I64P), n.ctypes.data_as(_I64P), ld.ctypes.data_as(_I64P),
                keep[3], keep[4], ncol.ctypes.data_as(_I64P), cptr, cblk, C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data)]
        t1 = fn(C.c_int64(2), *args) / 2  # calibrate
        reps = max(2, int(seconds / max(t1, 1e-6)))
        return reps, fn(C.c_int64(reps), *args)

    def coo_mul(self, rows, cols, vals, x, y, alpha=1, beta=0, strong_zero=True):
        dt = np.dtype(x.dtype)
        r, c = _i64(rows), _i64(cols)
        v = np.ascontiguousarray(vals, dt)
        x = np.ascontiguousarray(x, dt)
        assert y.dtype == dt and y.flags.c_contiguous
        fn = getattr(self.lib, "orc_coo_mul_" + _SFX[dt])
        fn.restype = None
        fn(C.c_int64(len(y)), C.c_int64(len(r)), r.ctypes.data_as(_I64P),
           c.ctypes.data_as(_I64P), C.c_void_p(v.ctypes.data), C.c_void_p(x.ctypes.data),
           C.c_void_p(y.ctypes.data), _scalar(dt, alpha), _scalar(dt, beta),
           C.c_int(1 if strong_zero else 0))
        return y
