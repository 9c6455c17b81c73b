"""Host-side mirror of the reference's operator surface for the mul! hot path.

Same names, argument meaning and error behaviour as BlockSparseMatrices.jl:

    BlockSparseMatrix(blocks, rowindices, colindices, size; scheduler, coloringalgorithm)
        -- reference src/blockmatrix.jl:26-109
    SymmetricBlockMatrix(diagonals, diagonalindices, offdiagonals, rowindices, colindices,
                         size; scheduler)            -- src/symmetricblockmatrix.jl:33-126
    VariableBlockCompressedRowStorage(matrices, rowindices, colindices, size; scheduler)
    VariableBlockCompressedRowStorage(bsm | sbm)     -- src/vbcrs.jl:36-199
    mul(y, A, x[, alpha, beta])  == LinearAlgebra.mul!   (LinearMaps._unsafe_mul!)
    A @ x / A * x, transpose(A) / A.T, adjoint(A) / A.H  (LinearMaps wrappers, field .lmap)
    nnz, size, eltype, block, eachblockindex, rowindices, colindices, colors, ...

Julia conventions are kept on purpose so the parity tests read like the reference's tests:
index lists are 1-BASED, blocks are column-major 2-D arrays.  Every product runs in
libbsmrocm.so (HIP, gfx950); this module only marshals arguments.  x / y may be numpy
arrays (host memory: the library stages them over PCIe) or torch CUDA tensors (device
memory, enqueued on torch's current stream).
"""
import ctypes as C

import numpy as np

from . import _lib as L

try:  # torch is plumbing only (device memory, streams)
    import torch
except Exception:  # pragma: no cover
    torch = None

__all__ = [
    "SerialScheduler", "DynamicScheduler", "isserial", "AbstractBlockMatrix", "BlockSparseMatrix",
    "SymmetricBlockMatrix", "VariableBlockCompressedRowStorage", "TransposeMap", "AdjointMap",
    "transpose", "adjoint", "mul", "mul_parts", "MulPlan", "nnz", "size", "eltype", "scheduler", "block", "eachblockindex",
    "rowindices", "colindices", "colors", "transposecolors", "diagonal", "offdiagonal",
    "eachdiagonalindex", "eachoffdiagonalindex", "diagonalindices", "diagonalcolors",
    "offdiagonalcolors", "transposeoffdiagonalcolors", "rowcolvals", "sparse", "ColorInfo", "conflicts",
    "color", "coloringalgorithm", "Context", "partition_rows", "host_register", "host_unregister", "rowcolvals_device", "sparse_device",
]

_DT = {np.dtype(np.float32): L.BSM_F32, np.dtype(np.float64): L.BSM_F64,
       np.dtype(np.complex64): L.BSM_C64, np.dtype(np.complex128): L.BSM_C128}


# ---- schedulers (OhMyThreads names; reference src/BlockSparseMatrices.jl:12-18) --------------
class SerialScheduler:
    def __repr__(self):
        return "SerialScheduler()"


class DynamicScheduler:
    def __repr__(self):
        return "DynamicScheduler()"


def isserial(s):
    return isinstance(s, SerialScheduler)


# ---- marshalling helpers -------------------------------------------------------------------------
def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _ptrs(arrs):
    out = (C.c_void_p * max(len(arrs), 1))()
    if len(arrs) and _is_dev(arrs):
        for i, a in enumerate(arrs):
            out[i] = a.data_ptr()
        return out
    for i, a in enumerate(arrs):
        out[i] = a.ctypes.data
    return out


_TORCH_DT = {}
if torch is not None:
    _TORCH_DT = {torch.float32: np.dtype(np.float32), torch.float64: np.dtype(np.float64),
                 torch.complex64: np.dtype(np.complex64), torch.complex128: np.dtype(np.complex128)}


def _is_dev(blocks):
    """Blocks given as torch CUDA tensors: device-resident operator data (bsm_options.blocks_memspace
    = BSM_MEM_DEVICE), repacked by a kernel -- the Python stand-in for a Julia caller's ROCArrays."""
    return torch is not None and len(blocks) > 0 and isinstance(blocks[0], torch.Tensor) and blocks[0].is_cuda


def _dev_blocks(*blocklists):
    """Checks device-resident blocks (all CUDA tensors of one dtype and device, column-major) -> numpy dtype."""
    dt = dev = None
    for bl in blocklists:
        for b in bl:
            if not (isinstance(b, torch.Tensor) and b.is_cuda and b.dim() == 2):
                raise TypeError("device-resident blocks must all be 2-D torch CUDA tensors")
            if b.dtype not in _TORCH_DT or (dt is not None and _TORCH_DT[b.dtype] != dt):
                raise TypeError("device-resident blocks must share one supported element type")
            if b.shape[0] > 1 and b.stride(0) != 1 or (b.shape[1] > 1 and b.stride(1) < max(b.shape[0], 1)):
                raise TypeError("device-resident blocks must be column-major (e.g. torch.empty(n, m).t())")
            if dev is not None and b.device != dev:
                raise ValueError("device-resident blocks must live on one device")
            dt, dev = _TORCH_DT[b.dtype], b.device
    return dt


def _lds(blocks):
    """leading dimensions (elements) of column-major blocks"""
    if _is_dev(blocks):
        return _i64([max(b.stride(1) if b.shape[1] > 1 else b.shape[0], b.shape[0], 1) for b in blocks])
    return _i64([max(b.shape[0], 1) for b in blocks])


def _host(b):
    """numpy view of a block wherever it lives (accessors used by sparse() / rowcolvals())"""
    return b.cpu().numpy() if (torch is not None and isinstance(b, torch.Tensor)) else _dense(b)


def _dense(b):
    """A block as a numpy array.  The reference takes any AbstractMatrix as a block and counts it as prod(size)
    (`_nnz`, src/abstractblockmatrix.jl:65-71) -- sparse blocks included; here such a block (anything with
    `.toarray()`, e.g. a scipy.sparse matrix) is densified ONCE, at construction: the packed image holds dense
    panels anyway."""
    if hasattr(b, "toarray") and not isinstance(b, np.ndarray):
        b = b.toarray()
    return np.asarray(b)


def _blocks_dtype(*blocklists):
    dt = None
    for bl in blocklists:
        for b in bl:
            d = _dense(b).dtype if hasattr(b, "toarray") and not isinstance(b, np.ndarray) else np.asarray(b).dtype
            dt = d if dt is None else np.promote_types(dt, d)
    if dt is None:
        dt = np.dtype(np.float64)
    if dt not in _DT:
        dt = np.promote_types(dt, np.float32) if dt.kind in "iub" else dt
    if np.dtype(dt) == np.float16:  # (Julia would promote Float16 blocks with Float32 scalars the same way)
        dt = np.dtype(np.float32)
    if np.dtype(dt) not in _DT:
        raise TypeError(f"unsupported block element type {dt}")
    return np.dtype(dt)


def _fblocks(blocks, dt):
    out = []
    for b in blocks:
        a = _dense(b)
        if a.ndim != 2:
            raise ValueError("every block must be a 2-D array")
        out.append(np.asfortranarray(a, dtype=dt))
    return out


class Context:
    """bsm_ctx_t: the GPUs of one node ONE handle is spread over (`devices=[0, 1, ...]` on any
    constructor).  The MI355X counterpart of the reference's `@tasks` fan-out over block rows /
    colour classes (src/vbcrs.jl:275-276, src/symmetricblockmatrix.jl:395-432).  The same ordinal may
    be listed several times (virtual devices)."""
    _cache = {}

    def __init__(self, devices):
        self.devices = tuple(int(d) for d in devices)
        ids = (C.c_int32 * len(self.devices))(*self.devices)
        h = C.c_void_p()
        L.check(L.lib().bsm_ctx_create(ids, len(self.devices), C.byref(h)))
        self.ptr = h

    @classmethod
    def get(cls, devices):
        """One context per device tuple for the life of the process (handles keep a pointer to it)."""
        key = tuple(int(d) for d in devices)
        if key not in cls._cache:
            cls._cache[key] = cls(key)
        return cls._cache[key]


def host_register(a):
    """bsm_host_register: page-locks a numpy vector used as x / y of host-memory products (DMA straight
    from / to it).  Keep `a` alive until host_unregister(a)."""
    L.check(L.lib().bsm_host_register(a.ctypes.data, a.nbytes))
    return a


def host_unregister(a):
    L.check(L.lib().bsm_host_unregister(a.ctypes.data))


def partition_rows(nrows, rowkeys, weights, nparts):
    """bsm_partition_rows: the row partition both multi-GPU layers use.  rowkeys[b] = smallest row
    index of block b (1-based), weights[b] = its stored entries.
    Returns (part_of_block, own) with own[p] = (lo, hi), 1-based inclusive (hi = lo - 1: empty)."""
    key, w = _i64(rowkeys), _i64(weights)
    nb = len(key)
    part = np.zeros(max(nb, 1), dtype=np.int32)
    lo, hi = np.zeros(nparts, dtype=np.int64), np.zeros(nparts, dtype=np.int64)
    I = C.POINTER(C.c_int64)
    L.check(L.lib().bsm_partition_rows(int(nrows), nb, key.ctypes.data_as(I), w.ctypes.data_as(I), int(nparts),
                                       part.ctypes.data_as(C.POINTER(C.c_int32)), lo.ctypes.data_as(I),
                                       hi.ctypes.data_as(I)))
    return part[:nb], [(int(a), int(b)) for a, b in zip(lo, hi)]


def _options(scheduler, device, accumulate, own=None, transpose_image=False, devices=None, dev_blocks=False,
             coloring=None):
    o = L.BsmOptions()
    L.lib().bsm_options_default(C.byref(o))
    o.scheduler = L.BSM_SCHED_SERIAL if isserial(scheduler) else L.BSM_SCHED_DYNAMIC
    o.device = device
    o.accumulate = {"auto": L.BSM_ACC_AUTO, "atomic": L.BSM_ACC_ATOMIC,
                    "colored": L.BSM_ACC_COLORED, "gather": L.BSM_ACC_GATHER,
                    "direct": L.BSM_ACC_DIRECT}[accumulate]
    if own is not None:
        o.own_lo, o.own_hi = int(own[0]), int(own[1])
    o.transpose_image = 2 if transpose_image == "auto" else (1 if transpose_image else 0)
    if devices is not None:
        if own is not None or transpose_image:
            raise ValueError("devices= does not combine with own= / transpose_image=")
        o.ctx = Context.get(devices).ptr
    o.blocks_memspace = L.BSM_MEM_DEVICE if dev_blocks else L.BSM_MEM_HOST
    o.coloring = _coloring_id(coloring)
    return o


def _default_device():
    """Current torch CUDA device when a GPU is visible, else analysis-only."""
    if torch is not None and torch.cuda.is_available():
        return torch.cuda.current_device()
    return L.BSM_DEVICE_NONE


def _scalar_buf(v, dt):
    return np.asarray([v], dtype=dt)


def _classes(flat):
    out, p = [], 1
    for _ in range(int(flat[0])):
        n = int(flat[p])
        out.append([int(v) for v in flat[p + 1:p + 1 + n]])
        p += 1 + n
    return out


class _Handle:
    """Owns a bsm_matrix_t; freed with the Python object (Julia side: a finalizer)."""

    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            if self.ptr:
                L.lib().bsm_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


# ---- LinearMaps-style base ---------------------------------------------------------------------------
class _LinearMap:
    """The slice of LinearMaps.LinearMap the reference relies on (`*`, mul!, adjoint, transpose)."""

    @property
    def shape(self):
        return size(self)

    def __matmul__(self, x):
        return _apply(self, x)

    def __mul__(self, x):
        return _apply(self, x)

    @property
    def T(self):
        return transpose(self)

    @property
    def H(self):
        return adjoint(self)

    def __getitem__(self, key):
        """A[:, :] -- LinearMaps materialises through products with unit vectors."""
        if key != (slice(None), slice(None)):
            raise IndexError("only A[:, :] is supported")
        m, n = size(self)
        dt = eltype(self)
        out = np.zeros((m, n), dtype=dt, order="F")
        step = 64  # unit vectors go through the multi-RHS entry (A streamed once per 8 of them)
        for j0 in range(0, n, step):
            k = min(step, n - j0)
            e = np.zeros((n, k), dtype=dt, order="F")
            e[np.arange(j0, j0 + k), np.arange(k)] = 1
            mul(out[:, j0:j0 + k], self, e)
        return out


class AbstractBlockMatrix(_LinearMap):
    """reference src/abstractblockmatrix.jl:13-62"""

    def _finish(self, handle, dt, sz, sched, device=None, devices=None):
        self._h = _Handle(handle)
        self.dtype = dt
        self.size = (int(sz[0]), int(sz[1]))
        self.scheduler = sched
        self.devices = None if devices is None else tuple(int(d) for d in devices)
        self.device = None if (devices is not None or device == L.BSM_DEVICE_NONE) else int(device)

    def parts(self):
        """Per-device view of a multi-device handle: list of dicts (device, own, touched, ...)."""
        if self.devices is None:
            raise ValueError("not a multi-device handle")
        out = []
        for p in range(len(self.devices)):
            pi = L.BsmPartInfo()
            L.check(L.lib().bsm_part_info(self._h.ptr, p, C.byref(pi)))
            out.append(dict(device=pi.device, own=(pi.own_lo, pi.own_hi), touched=(pi.touched_lo, pi.touched_hi),
                            cols=(pi.col_lo, pi.col_hi), device_bytes=pi.device_bytes, nblocks=pi.nblocks))
        return out

    def stats(self):
        st = L.BsmStats()
        L.check(L.lib().bsm_stats(self._h.ptr, C.byref(st)))
        return {k: getattr(st, k) for k, _ in L.BsmStats._fields_ if k != "reserved"}

    def _bookkeeping(self, which):
        n = C.c_int64(0)
        L.check(L.lib().bsm_get_bookkeeping(self._h.ptr, which, None, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.int64)
        L.check(L.lib().bsm_get_bookkeeping(self._h.ptr, which,
                                            out.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(n)))
        return out[:n.value]


class TransposeMap(_LinearMap):
    """LinearMaps.TransposeMap: lazy wrapper with field `.lmap`."""
    _op = L.BSM_OP_T

    def __init__(self, lmap):
        self.lmap = lmap


class AdjointMap(_LinearMap):
    """LinearMaps.AdjointMap: lazy wrapper with field `.lmap`."""
    _op = L.BSM_OP_C

    def __init__(self, lmap):
        self.lmap = lmap


def transpose(A):
    if isinstance(A, TransposeMap):
        return A.lmap
    if isinstance(A, AdjointMap):
        raise NotImplementedError("transpose(A') (= conj(A)) is not wrapped")
    return TransposeMap(A)


def adjoint(A):
    if isinstance(A, AdjointMap):
        return A.lmap
    if isinstance(A, TransposeMap):
        raise NotImplementedError("adjoint(transpose(A)) (= conj(A)) is not wrapped")
    return AdjointMap(A)


def _unwrap(A):
    if isinstance(A, (TransposeMap, AdjointMap)):
        return A.lmap, A._op
    return A, L.BSM_OP_N


# ---- colouring adapter (reference src/coloring.jl:15-61) ----------------------------------------------
# the reference's const (src/BlockSparseMatrices.jl:10); "DSATUR" selects plain DSATUR (GraphsColoring's names)
coloringalgorithm = "WorkstreamDSATUR"
_COLORING = {"WorkstreamDSATUR": L.BSM_COLOR_WORKSTREAM_DSATUR, "DSATUR": L.BSM_COLOR_DSATUR}


def _coloring_id(algorithm):
    name = coloringalgorithm if algorithm is None else algorithm
    if name not in _COLORING:
        raise ValueError(f"unknown coloring algorithm {name!r} (WorkstreamDSATUR, DSATUR)")
    return _COLORING[name]


class ColorInfo:
    """struct ColorInfo{R}: wraps the per-block conflict index lists (src/coloring.jl:15-17)."""

    def __init__(self, conflictindices):
        self.conflictindices = [_i64(c) for c in conflictindices]


class ConflictFunctor:
    def __init__(self, indices):
        self.indices = indices

    def __call__(self, i):
        return self.indices[i - 1]


def conflicts(blocks):
    """(eachindex(indices), ConflictFunctor(indices), Base.OneTo(maxconflict)) -- src/coloring.jl:45-61"""
    idx = blocks.conflictindices
    maxconflict = max(int(np.max(l)) for l in idx)
    return range(1, len(idx) + 1), ConflictFunctor(idx), range(1, maxconflict + 1)


def color(info, algorithm=None):
    """color(conflictgraph(info); algorithm).colors: classes of 1-based block ids that share no
    index (bsm_color; algorithm: "WorkstreamDSATUR" (default, like the reference) or "DSATUR")."""
    lists = info.conflictindices
    n = len(lists)
    lens = _i64([len(l) for l in lists])
    out = np.zeros(max(n, 1), dtype=np.int64)
    nc = C.c_int64(0)
    I = C.POINTER(C.c_int64)
    L.check(L.lib().bsm_color(n, _ptrs(lists), lens.ctypes.data_as(I), _coloring_id(algorithm),
                              out.ctypes.data_as(I), C.byref(nc)))
    return [[int(b) + 1 for b in np.nonzero(out[:n] == c)[0]] for c in range(nc.value)]


# ---- the three storage types ---------------------------------------------------------------------------
class BlockSparseMatrix(AbstractBlockMatrix):
    """reference src/blockmatrix.jl:26-109.  Fields: blocks, rowindices, colindices, size,
    colors, transposecolors, scheduler."""

    def __init__(self, blocks, rowindices, colindices, size, cols=None, *, scheduler=None,
                 coloringalgorithm=None, device=None, accumulate="auto", own=None,
                 transpose_image=False, devices=None):
        if cols is not None:  # (blocks, rowindices, colindices, rows, cols) form, :81-89
            size = (size, cols)
        scheduler = SerialScheduler() if scheduler is None else scheduler
        devb = _is_dev(blocks)
        dt = _dev_blocks(blocks) if devb else _blocks_dtype(blocks)
        self.blocks = list(blocks) if devb else _fblocks(blocks, dt)
        self.rowindices = [_i64(r) for r in rowindices]
        self.colindices = [_i64(c) for c in colindices]
        nb = len(self.blocks)
        if len(self.rowindices) != nb or len(self.colindices) != nb:
            raise ValueError("blocks, rowindices and colindices must have equal lengths")
        for b, r, c in zip(self.blocks, self.rowindices, self.colindices):
            if b.shape != (len(r), len(c)):
                raise ValueError("block shape does not match its index lists")
        m = _i64([b.shape[0] for b in self.blocks])
        n = _i64([b.shape[1] for b in self.blocks])
        ld = _lds(self.blocks)
        dev = _default_device() if device is None else device
        o = _options(scheduler, dev, accumulate, own, transpose_image, devices, devb, coloringalgorithm)
        h = C.c_void_p()
        I = C.POINTER(C.c_int64)
        L.check(L.lib().bsm_blocksparse_create(
            _DT[dt], int(size[0]), int(size[1]), nb, _ptrs(self.blocks), m.ctypes.data_as(I),
            n.ctypes.data_as(I), ld.ctypes.data_as(I), _ptrs(self.rowindices),
            _ptrs(self.colindices), C.byref(o), C.byref(h)))
        self._finish(h, dt, size, scheduler, dev, devices)
        self.colors = _classes(self._bookkeeping(L.BSM_BK_COLORS))
        self.transposecolors = _classes(self._bookkeeping(L.BSM_BK_TRANSPOSECOLORS))


class SymmetricBlockMatrix(AbstractBlockMatrix):
    """reference src/symmetricblockmatrix.jl:33-126.  The tuple-size constructor defaults to
    DynamicScheduler() (:80)."""

    def __init__(self, diagonals, diagonalindices, offdiagonals, rowindices, colindices, size,
                 cols=None, *, scheduler=None, device=None, accumulate="auto", own=None, devices=None):
        if cols is not None:  # rows, cols form defaults to SerialScheduler() (:102)
            size = (size, cols)
            scheduler = SerialScheduler() if scheduler is None else scheduler
        scheduler = DynamicScheduler() if scheduler is None else scheduler
        devb = _is_dev(diagonals) or _is_dev(offdiagonals)
        dt = _dev_blocks(diagonals, offdiagonals) if devb else _blocks_dtype(diagonals, offdiagonals)
        self.diagonals = list(diagonals) if devb else _fblocks(diagonals, dt)
        self.diagonalindices = [_i64(d) for d in diagonalindices]
        self.offdiagonals = list(offdiagonals) if devb else _fblocks(offdiagonals, dt)
        self.rowindices = [_i64(r) for r in rowindices]
        self.colindices = [_i64(c) for c in colindices]
        nd, no = len(self.diagonals), len(self.offdiagonals)
        if len(self.diagonalindices) != nd or len(self.rowindices) != no or len(self.colindices) != no:
            raise ValueError("block and index list counts differ")
        for b, d in zip(self.diagonals, self.diagonalindices):
            if b.shape != (len(d), len(d)):
                raise ValueError("diagonal block shape does not match its index list")
        for b, r, c in zip(self.offdiagonals, self.rowindices, self.colindices):
            if b.shape != (len(r), len(c)):
                raise ValueError("off-diagonal block shape does not match its index lists")
        ds = _i64([b.shape[0] for b in self.diagonals])
        dld = _lds(self.diagonals)
        m = _i64([b.shape[0] for b in self.offdiagonals])
        n = _i64([b.shape[1] for b in self.offdiagonals])
        ld = _lds(self.offdiagonals)
        dev = _default_device() if device is None else device
        o = _options(scheduler, dev, accumulate, own, False, devices, devb)
        h = C.c_void_p()
        I = C.POINTER(C.c_int64)
        L.check(L.lib().bsm_symmetric_create(
            _DT[dt], int(size[0]), int(size[1]), nd, _ptrs(self.diagonals), ds.ctypes.data_as(I),
            dld.ctypes.data_as(I), _ptrs(self.diagonalindices), no, _ptrs(self.offdiagonals),
            m.ctypes.data_as(I), n.ctypes.data_as(I), ld.ctypes.data_as(I),
            _ptrs(self.rowindices), _ptrs(self.colindices), C.byref(o), C.byref(h)))
        self._finish(h, dt, size, scheduler, dev, devices)
        self.offdiagonalcolors = _classes(self._bookkeeping(L.BSM_BK_COLORS))
        self.transposeoffdiagonalcolors = _classes(self._bookkeeping(L.BSM_BK_TRANSPOSECOLORS))
        self.diagonalcolors = _classes(self._bookkeeping(L.BSM_BK_DIAGONALCOLORS))


class VariableBlockCompressedRowStorage(AbstractBlockMatrix):
    """reference src/vbcrs.jl:36-199.  Fields: blocks (sorted), rowptr, colindices (per block),
    rowindices (per block row), size, scheduler -- all 1-based like the reference."""

    def __init__(self, matrices, rowindices=None, colindices=None, matrixsize=None, *,
                 scheduler=None, device=None, accumulate="auto", own=None, materialize=False,
                 transpose_image=False, devices=None):
        I = C.POINTER(C.c_int64)
        h = C.c_void_p()
        dev = _default_device() if device is None else device
        if isinstance(matrices, SymmetricBlockMatrix) and not materialize:  # src/vbcrs.jl:189-264
            # Same bookkeeping as the reference's expansion [diagonals..., offdiagonals...,
            # transpose(offdiagonals)...] (:222-241), but the transposes are NOT materialised:
            # the device image is the symmetric one and each off-diagonal block is streamed once.
            s = matrices
            scheduler = s.scheduler if scheduler is None else scheduler
            dt = s.dtype
            mats = list(s.diagonals) + list(s.offdiagonals) + [o.T for o in s.offdiagonals]  # views
            ds = _i64([b.shape[0] for b in s.diagonals])
            dld = _lds(s.diagonals)
            d0 = _i64([int(d[0]) for d in s.diagonalindices])  # first(...), :231-239
            m = _i64([b.shape[0] for b in s.offdiagonals])
            n = _i64([b.shape[1] for b in s.offdiagonals])
            ld = _lds(s.offdiagonals)
            r0 = _i64([int(r[0]) for r in s.rowindices])
            c0 = _i64([int(c[0]) for c in s.colindices])
            o = _options(scheduler, dev, accumulate, own, False, devices,
                         _is_dev(s.diagonals) or _is_dev(s.offdiagonals))
            L.check(L.lib().bsm_vbcrs_create_from_symmetric(
                _DT[dt], int(s.size[0]), int(s.size[1]), len(s.diagonals), _ptrs(s.diagonals),
                ds.ctypes.data_as(I), dld.ctypes.data_as(I), d0.ctypes.data_as(I), len(s.offdiagonals),
                _ptrs(s.offdiagonals), m.ctypes.data_as(I), n.ctypes.data_as(I), ld.ctypes.data_as(I),
                r0.ctypes.data_as(I), c0.ctypes.data_as(I), C.byref(o), C.byref(h)))
            matrixsize = s.size
            fb = mats
        else:
            if isinstance(matrices, BlockSparseMatrix):  # src/vbcrs.jl:150-160
                # bsm_vbcrs_create_from_blocksparse: first(rowindices(b, i)) / first(colindices(b, i))
                # are taken inside the library (src/vbcrs.jl:201-215)
                b = matrices
                scheduler = b.scheduler if scheduler is None else scheduler
                if len(b.blocks) < 1:
                    raise IndexError("VariableBlockCompressedRowStorage needs at least one block")  # :81
                dt = b.dtype
                fb = b.blocks
                m = _i64([k.shape[0] for k in fb])
                n = _i64([k.shape[1] for k in fb])
                ld = _lds(fb)
                o = _options(scheduler, dev, accumulate, own, transpose_image, devices, _is_dev(fb))
                L.check(L.lib().bsm_vbcrs_create_from_blocksparse(
                    _DT[dt], int(b.size[0]), int(b.size[1]), len(fb), _ptrs(fb), m.ctypes.data_as(I),
                    n.ctypes.data_as(I), ld.ctypes.data_as(I), _ptrs(b.rowindices), _ptrs(b.colindices),
                    C.byref(o), C.byref(h)))
                matrixsize = b.size
                mats = None
            elif isinstance(matrices, SymmetricBlockMatrix):  # reference behaviour: materialise
                s = matrices
                scheduler = s.scheduler if scheduler is None else scheduler
                mats = list(s.diagonals) + list(s.offdiagonals) + [o.T for o in s.offdiagonals]
                rowindices = ([int(d[0]) for d in s.diagonalindices] + [int(r[0]) for r in s.rowindices]
                              + [int(c[0]) for c in s.colindices])
                colindices = ([int(d[0]) for d in s.diagonalindices] + [int(c[0]) for c in s.colindices]
                              + [int(r[0]) for r in s.rowindices])
                matrixsize = s.size
            else:
                mats = matrices
            scheduler = SerialScheduler() if scheduler is None else scheduler
            if mats is not None:
                if len(mats) < 1:
                    raise IndexError("VariableBlockCompressedRowStorage needs at least one block")  # :81
                devb = _is_dev(mats)
                dt = _dev_blocks(mats) if devb else _blocks_dtype(mats)
                fb = list(mats) if devb else _fblocks(mats, dt)
                rs, cs = _i64(rowindices), _i64(colindices)
                if len(rs) != len(fb) or len(cs) != len(fb):
                    raise ValueError("matrices, rowindices and colindices must have equal lengths")
                m = _i64([b.shape[0] for b in fb])
                n = _i64([b.shape[1] for b in fb])
                ld = _lds(fb)
                o = _options(scheduler, dev, accumulate, own, transpose_image, devices, devb)
                L.check(L.lib().bsm_vbcrs_create(
                    _DT[dt], int(matrixsize[0]), int(matrixsize[1]), len(fb), _ptrs(fb),
                    m.ctypes.data_as(I), n.ctypes.data_as(I), ld.ctypes.data_as(I), rs.ctypes.data_as(I),
                    cs.ctypes.data_as(I), C.byref(o), C.byref(h)))
        self._finish(h, dt, matrixsize, scheduler, dev, devices)
        self.perm = self._bookkeeping(L.BSM_BK_VBCRS_PERM).copy()
        self.rowptr = self._bookkeeping(L.BSM_BK_VBCRS_ROWPTR).copy()
        self.colindices = self._bookkeeping(L.BSM_BK_VBCRS_COLINDICES).copy()
        self.rowindices = self._bookkeeping(L.BSM_BK_VBCRS_ROWINDICES).copy()
        self.blocks = [fb[p - 1] for p in self.perm]


# ---- mul! ------------------------------------------------------------------------------------------------
def _check_device(v, base, name):
    """A tensor on another GPU than the handle's would be dereferenced by kernels of the handle's
    device: a device memory fault that aborts the process.  Raise instead."""
    if base.devices is not None:  # multi-device handle: x / y may live on any device (peer copies)
        return
    if base.device is not None and v.device.index != base.device:
        raise ValueError(f"{name} lives on cuda:{v.device.index} but the matrix was created on cuda:{base.device}")


def _vec_info(v, dt, n, name, base=None):
    """-> (pointer, memspace, stream, keepalive)"""
    if torch is not None and isinstance(v, torch.Tensor) and v.is_cuda and base is not None:
        _check_device(v, base, name)
    if torch is not None and isinstance(v, torch.Tensor):
        if v.dim() != 1 or v.numel() != n:
            raise ValueError(f"DimensionMismatch: {name} has length {tuple(v.shape)}, expected {n}")
        tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
               np.dtype(np.complex64): torch.complex64, np.dtype(np.complex128): torch.complex128}[dt]
        if v.dtype != tdt or not v.is_contiguous():
            raise TypeError(f"{name} must be a contiguous {tdt} tensor")
        if v.is_cuda:
            return v.data_ptr(), L.BSM_MEM_DEVICE, torch.cuda.current_stream(v.device).cuda_stream, v
        a = v.numpy()
        return a.ctypes.data, L.BSM_MEM_HOST, None, a
    if not isinstance(v, np.ndarray):
        raise TypeError(f"{name} must be a numpy array or a torch tensor")
    if v.ndim != 1 or v.shape[0] != n:
        raise ValueError(f"DimensionMismatch: {name} has shape {v.shape}, expected ({n},)")
    if v.dtype != dt or not v.flags.c_contiguous:
        raise TypeError(f"{name} must be a contiguous {dt} array")
    return v.ctypes.data, L.BSM_MEM_HOST, None, v


def _mat_info(v, dt, n, name, base=None):
    """2-D column-major operand -> (pointer, ld, ncols, memspace, stream, keepalive)"""
    if torch is not None and isinstance(v, torch.Tensor) and v.is_cuda and base is not None:
        _check_device(v, base, name)
    if torch is not None and isinstance(v, torch.Tensor):
        tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
               np.dtype(np.complex64): torch.complex64, np.dtype(np.complex128): torch.complex128}[dt]
        if v.dim() != 2 or v.shape[0] != n:
            raise ValueError(f"DimensionMismatch: {name} has shape {tuple(v.shape)}, expected ({n}, k)")
        if v.dtype != tdt or v.stride(0) != 1 or (v.shape[1] > 1 and v.stride(1) < max(n, 1)):
            raise TypeError(f"{name} must be a column-major {tdt} tensor (e.g. torch.empty(k, n).t())")
        ld = v.stride(1) if v.shape[1] > 1 else max(n, 1)
        if v.is_cuda:
            return (v.data_ptr(), ld, v.shape[1], L.BSM_MEM_DEVICE,
                    torch.cuda.current_stream(v.device).cuda_stream, v)
        a = v.numpy()
        return a.ctypes.data, ld, v.shape[1], L.BSM_MEM_HOST, None, a
    if not isinstance(v, np.ndarray) or v.ndim != 2 or v.shape[0] != n:
        raise ValueError(f"DimensionMismatch: {name} must be a 2-D array with {n} rows")
    if v.dtype != dt or not v.flags.f_contiguous:
        raise TypeError(f"{name} must be a column-major (Fortran-order) {dt} array")
    return v.ctypes.data, max(n, 1), v.shape[1], L.BSM_MEM_HOST, None, v


def _mul_matrix(Y, A, X, alpha, beta):
    """mul!(Y, A, X, alpha, beta) with matrices: one bsm_mul_multi call (A streamed once per batch
    of up to 8 columns) instead of LinearMaps' column loop over _unsafe_mul!."""
    base, op = _unwrap(A)
    dt = base.dtype
    nr, nc = base.size
    ylen, xlen = (nr, nc) if op == L.BSM_OP_N else (nc, nr)
    xp, ldx, kx, xms, _, _kx = _mat_info(X, dt, xlen, "X", base)
    yp, ldy, ky, yms, yst, _ky = _mat_info(Y, dt, ylen, "Y", base)
    if kx != ky:
        raise ValueError("DimensionMismatch: X and Y have different numbers of columns")
    if xms != yms:
        raise ValueError("X and Y must live in the same memory space")
    strong = beta is False
    a = _scalar_buf(1 if alpha is True else alpha, dt)
    b = _scalar_buf(0 if strong else (1 if beta is True else beta), dt)
    L.check(L.lib().bsm_mul_multi(base._h.ptr, op, kx, xp, ldx, yp, ldy, a.ctypes.data, b.ctypes.data,
                                  1 if strong else 0, xms, yst))
    return Y


def mul(y, A, x, alpha=True, beta=False):
    """LinearAlgebra.mul!(y, A, x, alpha, beta): y = alpha*A*x + beta*y, returns y.
    x / y may also be matrices (column-major): the multi right-hand-side product.

    `beta is False` (the 3-argument form, reference src/abstractblockmatrix.jl:27-34) is Julia's
    strong zero: y is overwritten, NaN/Inf in the incoming y do not propagate.  A numeric 0.0
    multiplies.  Dimension checks mirror LinearMaps' check_dim_mul (DimensionMismatch)."""
    base, op = _unwrap(A)
    if not isinstance(base, AbstractBlockMatrix):
        raise TypeError("A must be a block matrix or its transpose/adjoint wrapper")
    dt = base.dtype
    nr, nc = base.size
    ylen, xlen = (nr, nc) if op == L.BSM_OP_N else (nc, nr)
    if getattr(x, "ndim", 1) == 2 or getattr(y, "ndim", 1) == 2:
        if dt.kind != "c" and (np.iscomplexobj(alpha) or np.iscomplexobj(beta)):
            raise TypeError("complex alpha/beta with a real matrix is not supported on the GPU path")
        return _mul_matrix(y, A, x, alpha, beta)
    if dt.kind != "c" and (np.iscomplexobj(alpha) or np.iscomplexobj(beta)):
        raise TypeError("complex alpha/beta with a real matrix is not supported on the GPU path")
    xp, xms, xst, _kx = _vec_info(x, dt, xlen, "x", base)
    yp, yms, yst, _ky = _vec_info(y, dt, ylen, "y", base)
    if xms != yms:
        raise ValueError("x and y must live in the same memory space")
    strong = beta is False
    a = _scalar_buf(1 if alpha is True else alpha, dt)
    b = _scalar_buf(0 if strong else (1 if beta is True else beta), dt)
    L.check(L.lib().bsm_mul(base._h.ptr, op, xp, yp, a.ctypes.data, b.ctypes.data,
                            1 if strong else 0, xms, yst if yst is not None else None))
    return y


def mul_parts(y_parts, A, x_parts, alpha=True, beta=False):
    """bsm_mul_parts: mul!(y, A, x, alpha, beta) on a multi-device handle with x and y PARTITIONED over its
    devices.  x_parts[p] / y_parts[p]: torch CUDA tensors on the device of part p holding, for A (op N), the
    x entries of the part's column range (`A.parts()[p]["cols"]`) and the y entries of its row range
    (`["own"]`); for transpose(A) / adjoint(A) the other way round.  Enqueued on the current torch stream
    of every part's device; nothing is synchronised."""
    base, op = _unwrap(A)
    if base.devices is None:
        raise ValueError("mul_parts needs a multi-device handle (devices=[...])")
    dt = base.dtype
    parts = base.parts()
    if len(x_parts) != len(parts) or len(y_parts) != len(parts):
        raise ValueError("one x part and one y part per device of the handle")
    P = len(parts)
    xp, yp, st = (C.c_void_p * P)(), (C.c_void_p * P)(), (C.c_void_p * P)()
    for p, info in enumerate(parts):
        xr, yr = (info["cols"], info["own"]) if op == L.BSM_OP_N else (info["own"], info["cols"])
        for v, (lo, hi), name, arr in ((x_parts[p], xr, "x", xp), (y_parts[p], yr, "y", yp)):
            n = max(hi - lo + 1, 0)
            if v is None and n == 0:
                arr[p] = None
                continue
            if not (isinstance(v, torch.Tensor) and v.is_cuda and v.dim() == 1 and v.is_contiguous()):
                raise TypeError(f"{name}_parts[{p}] must be a contiguous 1-D CUDA tensor")
            if v.numel() != n or _TORCH_DT.get(v.dtype) != dt:
                raise ValueError(f"DimensionMismatch: {name}_parts[{p}] has {v.numel()} entries of {v.dtype}, expected {n} of {dt}")
            if v.device.index != info["device"]:
                raise ValueError(f"{name}_parts[{p}] lives on cuda:{v.device.index}, part {p} on cuda:{info['device']}")
            arr[p] = v.data_ptr()
        st[p] = torch.cuda.current_stream(torch.device("cuda", info["device"])).cuda_stream
    strong = beta is False
    a = _scalar_buf(1 if alpha is True else alpha, dt)
    b = _scalar_buf(0 if strong else (1 if beta is True else beta), dt)
    L.check(L.lib().bsm_mul_parts(base._h.ptr, op, xp, yp, a.ctypes.data, b.ctypes.data, 1 if strong else 0, st))
    return y_parts


class MulPlan:
    """Pre-marshalled mul!(y, A, x, alpha, beta) for device-resident x / y: `plan()` is one
    ctypes call into bsm_mul (no per-call Python marshalling), enqueued on the CURRENT torch
    stream -- what a Julia caller gets from `ccall` directly.  Graph-capturable."""

    def __init__(self, y, A, x, alpha=True, beta=False):
        base, op = _unwrap(A)
        dt = base.dtype
        nr, nc = base.size
        ylen, xlen = (nr, nc) if op == L.BSM_OP_N else (nc, nr)
        xp, xms, _, self._kx = _vec_info(x, dt, xlen, "x", base)
        yp, yms, _, self._ky = _vec_info(y, dt, ylen, "y", base)
        if xms != L.BSM_MEM_DEVICE or yms != L.BSM_MEM_DEVICE:
            raise ValueError("MulPlan needs device-resident x and y")
        strong = beta is False
        self._a = _scalar_buf(1 if alpha is True else alpha, dt)
        self._b = _scalar_buf(0 if strong else (1 if beta is True else beta), dt)
        self._base = base
        self._fn = L.lib().bsm_mul
        self._dev = y.device
        self._args = [base._h.ptr, C.c_int(op), C.c_void_p(xp), C.c_void_p(yp),
                      C.c_void_p(self._a.ctypes.data), C.c_void_p(self._b.ctypes.data),
                      C.c_int(1 if strong else 0), C.c_int(L.BSM_MEM_DEVICE)]

    def __call__(self):
        rc = self._fn(*self._args, C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream))
        if rc:
            L.check(rc)


class SegmentAdd:
    """Pre-marshalled `y[a_s : a_s + len_s] += src_s` over DISJOINT segments of one device vector, ONE launch on the
    current torch stream (bsm_vec_add_segments): the delivery step of distributed.RowPartitioned -- the own rows of the
    boundary blocks' sums + every received partial-y segment.  dsts: views of y, srcs: tensors of the same lengths."""

    def __init__(self, y, dsts, srcs):
        dt = {torch.float32: 0, torch.float64: 1, torch.complex64: 2, torch.complex128: 3}[y.dtype]
        es = y.element_size()
        n = len(dsts)
        for d, s_ in zip(dsts, srcs):
            if d.dtype != y.dtype or s_.dtype != y.dtype or d.shape != s_.shape or d.dim() != 1 or \
                    not d.is_contiguous() or not s_.is_contiguous() or d.device != y.device or s_.device != y.device:
                raise ValueError("segments must be contiguous 1-D tensors of y's type on y's device")
        # (an empty view has no address of its own)
        self._off = (C.c_int64 * n)(*[(d.data_ptr() - y.data_ptr()) // es if d.numel() else 0 for d in dsts])
        self._len = (C.c_int64 * n)(*[d.shape[0] for d in dsts])
        self._src = (C.c_void_p * n)(*[s_.data_ptr() if s_.numel() else None for s_ in srcs])
        if any(o < 0 or o + l_ > y.numel() for o, l_ in zip(self._off, self._len)):
            raise ValueError("a segment lies outside y")
        self._keep = (y, dsts, srcs)
        self._fn = L.lib().bsm_vec_add_segments
        self._args = [C.c_int(dt), C.c_void_p(y.data_ptr()), C.c_int32(n), self._off, self._src, self._len]
        self._dev = y.device

    def matches(self, y, dsts, srcs):
        ky, kd, ks = self._keep
        return ky.data_ptr() == y.data_ptr() and len(kd) == len(dsts) and \
            all(a.data_ptr() == b.data_ptr() and a.shape == b.shape for a, b in zip(kd, dsts)) and \
            all(a.data_ptr() == b.data_ptr() for a, b in zip(ks, srcs))

    def __call__(self):
        rc = self._fn(*self._args, C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream))
        if rc:
            L.check(rc)


def _apply(A, x):
    """A * x: allocates y like LinearMaps does (similar(x, ...), uninitialised) then mul!."""
    m, _ = size(A)
    if torch is not None and isinstance(x, torch.Tensor):
        if x.dim() == 2:  # A * X: column-major result
            y = torch.empty((x.shape[1], m), dtype=x.dtype, device=x.device).t()
            if x.stride(0) != 1:
                x = x.t().contiguous().t()
        else:
            y = torch.empty(m, dtype=x.dtype, device=x.device)
    else:
        x = np.asarray(x, dtype=eltype(A))
        if x.ndim == 2:
            x = np.asfortranarray(x)
            y = np.empty((m, x.shape[1]), dtype=x.dtype, order="F")
        else:
            x = np.ascontiguousarray(x)
            y = np.empty(m, dtype=x.dtype)
    return mul(y, A, x)


# ---- accessors (reference names) ---------------------------------------------------------------------------
def size(A, dim=None):
    base, op = _unwrap(A)
    s = base.size if op == L.BSM_OP_N else (base.size[1], base.size[0])
    return s if dim is None else s[dim - 1]


def eltype(A):
    return _unwrap(A)[0].dtype


def scheduler(A):  # src/abstractblockmatrix.jl:50-62
    return _unwrap(A)[0].scheduler


def _wrapblock(b, op):
    if op == L.BSM_OP_T:
        return b.T
    if op == L.BSM_OP_C:
        return b.conj().T if not (torch is not None and isinstance(b, torch.Tensor)) else b.conj().t()
    return b


def eachblockindex(A):  # src/blockmatrix.jl:124-134
    return range(1, len(_unwrap(A)[0].blocks) + 1)


def block(A, i):  # src/blockmatrix.jl:150-160
    base, op = _unwrap(A)
    return _wrapblock(base.blocks[i - 1], op)


def rowindices(A, i):  # src/symmetricblockmatrix.jl:341-352
    base, op = _unwrap(A)
    return (base.rowindices if op == L.BSM_OP_N else base.colindices)[i - 1]


def colindices(A, i):  # src/symmetricblockmatrix.jl:354-365
    base, op = _unwrap(A)
    return (base.colindices if op == L.BSM_OP_N else base.rowindices)[i - 1]


def colors(A):  # src/blockmatrix.jl:177-206
    base, op = _unwrap(A)
    return base.colors if op == L.BSM_OP_N else base.transposecolors


def transposecolors(A):
    base, op = _unwrap(A)
    return base.transposecolors if op == L.BSM_OP_N else base.colors


def eachoffdiagonalindex(A):
    return range(1, len(_unwrap(A)[0].offdiagonals) + 1)


def eachdiagonalindex(A):
    return range(1, len(_unwrap(A)[0].diagonals) + 1)


def offdiagonal(A, i):  # src/symmetricblockmatrix.jl:199-237
    base, op = _unwrap(A)
    return _wrapblock(base.offdiagonals[i - 1], op)


def diagonal(A, i):
    base, op = _unwrap(A)
    return _wrapblock(base.diagonals[i - 1], op)


def diagonalindices(A, i):  # src/symmetricblockmatrix.jl:327-339
    return _unwrap(A)[0].diagonalindices[i - 1]


def diagonalcolors(A):
    return _unwrap(A)[0].diagonalcolors


def offdiagonalcolors(A):  # swaps for wrappers, src/symmetricblockmatrix.jl:307-325
    base, op = _unwrap(A)
    return base.offdiagonalcolors if op == L.BSM_OP_N else base.transposeoffdiagonalcolors


def transposeoffdiagonalcolors(A):
    base, op = _unwrap(A)
    return base.transposeoffdiagonalcolors if op == L.BSM_OP_N else base.offdiagonalcolors


def nnz(A):
    """SparseArrays.nnz -- src/blockmatrix.jl:208-223, src/symmetricblockmatrix.jl:367-384
    (off-diagonal blocks count twice), src/vbcrs.jl:290-296."""
    return int(_unwrap(A)[0].stats()["nnz"])


# ---- conversion used by the reference's tests as their oracle (host utility, not the hot path) ----
def rowcolvals(A):
    """(rows, cols, vals), 1-based -- reference src/sparse.jl:17-123."""
    base, op = _unwrap(A)
    rows, cols, vals = [], [], []

    def push(b, ri, ci):  # _pushblocktoarrays!, src/sparse.jl:131-139 (row-major enumeration)
        R, Cc = np.meshgrid(np.asarray(ri), np.asarray(ci), indexing="ij")
        rows.append(R.ravel())
        cols.append(Cc.ravel())
        vals.append(_host(b).ravel())

    if isinstance(base, BlockSparseMatrix):
        for col in colors(A):
            for bid in col:
                push(block(A, bid), rowindices(A, bid), colindices(A, bid))
    elif isinstance(base, SymmetricBlockMatrix):
        for col in offdiagonalcolors(A):
            for bid in col:
                push(offdiagonal(A, bid), rowindices(A, bid), colindices(A, bid))
        for col in transposeoffdiagonalcolors(A):
            for bid in col:
                push(offdiagonal(A, bid).T, colindices(A, bid), rowindices(A, bid))
        for col in diagonalcolors(A):
            for bid in col:
                push(diagonal(A, bid), diagonalindices(A, bid), diagonalindices(A, bid))
    elif isinstance(base, VariableBlockCompressedRowStorage):
        if op != L.BSM_OP_N:
            raise NotImplementedError("rowcolvals of a wrapped VBCRS (the reference has none either)")
        for br in range(len(base.rowptr) - 1):
            for bi in range(base.rowptr[br], base.rowptr[br + 1]):
                b = _host(base.blocks[bi - 1])
                r0, c0 = base.rowindices[br], base.colindices[bi - 1]
                R, Cc = np.meshgrid(np.arange(r0, r0 + b.shape[0]), np.arange(c0, c0 + b.shape[1]),
                                    indexing="ij")
                rows.append(R.ravel(order="F"))
                cols.append(Cc.ravel(order="F"))
                vals.append(b.ravel(order="F"))
    else:
        raise TypeError("not a block matrix")
    if not rows:
        z = np.zeros(0, np.int64)
        return z, z, np.zeros(0, base.dtype)
    return np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)


def rowcolvals_device(A, device=True):
    """bsm_rowcolvals: (rows, cols, vals), 1-based, written by a kernel from the packed DEVICE image
    (no block returns to the host).  device=True: torch CUDA tensors on the handle's device (multi-
    device handles: numpy arrays); False: numpy arrays.  Wrapped matrices: rows / cols swapped, values
    conjugated for the adjoint."""
    base, op = _unwrap(A)
    n = C.c_int64(0)
    L.check(L.lib().bsm_rowcolvals(base._h.ptr, None, None, None, C.byref(n), L.BSM_MEM_HOST, None))
    cnt = n.value
    dt = base.dtype
    if device and base.devices is None and torch is not None:
        tdt = {v: k for k, v in _TORCH_DT.items()}[dt]
        dev = torch.device("cuda", base.device)
        r = torch.empty(max(cnt, 1), dtype=torch.int64, device=dev)
        c = torch.empty(max(cnt, 1), dtype=torch.int64, device=dev)
        v = torch.empty(max(cnt, 1), dtype=tdt, device=dev)
        L.check(L.lib().bsm_rowcolvals(base._h.ptr, r.data_ptr(), c.data_ptr(), v.data_ptr(), C.byref(n),
                                       L.BSM_MEM_DEVICE, torch.cuda.current_stream(dev).cuda_stream))
        r, c, v = r[:cnt], c[:cnt], v[:cnt]
        if op != L.BSM_OP_N:
            r, c = c, r
            if op == L.BSM_OP_C:
                v = v.conj()
        return r, c, v
    r = np.zeros(max(cnt, 1), dtype=np.int64)
    c = np.zeros(max(cnt, 1), dtype=np.int64)
    v = np.zeros(max(cnt, 1), dtype=dt)
    L.check(L.lib().bsm_rowcolvals(base._h.ptr, r.ctypes.data, c.ctypes.data, v.ctypes.data, C.byref(n),
                                   L.BSM_MEM_HOST, None))
    r, c, v = r[:cnt], c[:cnt], v[:cnt]
    if op != L.BSM_OP_N:
        r, c = c, r
        if op == L.BSM_OP_C:
            v = v.conj()
    return r, c, v


def sparse_device(A):
    """sparse(A) assembled ON the GPU: COO triples from the packed image (bsm_rowcolvals), duplicates
    summed and rows compressed by torch -> torch.sparse_csr_tensor on the handle's device
    (direct VBCRS -> CSR)."""
    r, c, v = rowcolvals_device(A, device=True)
    coo = torch.sparse_coo_tensor(torch.stack([r - 1, c - 1]), v, size=size(A)).coalesce()
    return coo.to_sparse_csr()


def sparse(A):
    """SparseArrays.sparse(A) -> scipy.sparse.csc_matrix (duplicates summed) -- src/sparse.jl:127-129."""
    import scipy.sparse as sp
    r, c, v = rowcolvals(A)
    return sp.coo_matrix((v, (r - 1, c - 1)), shape=size(A)).tocsc()
