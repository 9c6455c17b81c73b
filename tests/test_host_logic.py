"""CPU suite, part 2: the C-ABI library loads, exports every declared symbol, and its HOST
logic (reference bookkeeping, colouring, packing, wave schedule) is right -- no GPU needed:
handles are created with BSM_DEVICE_NONE and the packed image is executed by a numpy
interpreter (tests/_common.py) and compared with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from _common import (Cc, N, T, fixture_as_blocksparse, fixture_problem, interpret_image, oracle_mul,
                     rand_vec, relerr)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODEV = -2
OPS = [N, T, Cc]


def test_library_exports_every_declared_symbol(bsm):
    from bsm_amd import _lib as L
    hdr = open(os.path.join(ROOT, "include", "bsm_rocm.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(bsm_[a-z_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    lib = L.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/bsm_rocm.h but not exported"
    assert sorted(L.EXPORTS) == declared
    # ... and the bench / test utility header
    syn = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "bsm_synth.h")).read(), flags=re.S)
    syn_declared = sorted(set(re.findall(r"\b(bsm_(?:synth|bench)_[a-z_]+)\s*\(", syn)))
    assert syn_declared == sorted(L.SYNTH_EXPORTS)
    for name in syn_declared:
        assert hasattr(lib, name), f"{name} declared in include/bsm_synth.h but not exported"
    assert b"gfx950" in lib.bsm_version()
    assert C.sizeof(L.BsmOptions) == 72 and C.sizeof(L.BsmStats) == 128


def test_mul_without_device_image_fails_loudly(bsm):
    A = bsm.BlockSparseMatrix([np.eye(2)], [[1, 2]], [[1, 2]], (2, 2), device=NODEV)
    with pytest.raises(RuntimeError, match="no device image"):
        bsm.mul(np.zeros(2), A, np.ones(2))


# ---- reference bookkeeping, bit-exact ----------------------------------------------------------
def test_vbcrs_bookkeeping_kat(bsm):
    blocks = [np.ones((2, 3)), np.ones((4, 2)), np.ones((4, 4)), np.ones((2, 1))]
    A = bsm.VariableBlockCompressedRowStorage(blocks, [5, 1, 1, 5], [1, 7, 3, 9], (6, 9), device=NODEV)
    assert A.perm.tolist() == [3, 2, 1, 4]
    assert A.rowptr.tolist() == [1, 3, 5]
    assert A.rowindices.tolist() == [1, 5]
    assert A.colindices.tolist() == [3, 7, 1, 9]
    assert bsm.nnz(A) == 6 + 8 + 16 + 2


def test_vbcrs_bookkeeping_matches_oracle_random(bsm, oracle):
    rng = np.random.default_rng(0)
    for _ in range(20):
        nb = int(rng.integers(1, 40))
        rs = rng.integers(1, 8, nb) * 4 - 3
        cs = rng.integers(1, 8, nb) * 4 - 3
        blocks = [np.ones((4, 4))] * nb
        A = bsm.VariableBlockCompressedRowStorage(blocks, rs, cs, (32, 32), device=NODEV)
        perm, rowptr, colind, rowind = oracle.vbcrs_build(rs, cs)
        assert np.array_equal(A.perm, perm) and np.array_equal(A.rowptr, rowptr)
        assert np.array_equal(A.colindices, colind) and np.array_equal(A.rowindices, rowind)


def test_serial_scheduler_gives_single_colour(bsm):
    p = fixture_as_blocksparse("sphere")
    A = bsm.BlockSparseMatrix(p["blocks"], p["rowindices"], p["colindices"], p["size"], device=NODEV)
    nb = len(p["blocks"])
    assert A.colors == [list(range(1, nb + 1))] and A.transposecolors == [list(range(1, nb + 1))]
    assert bsm.colors(bsm.transpose(A)) == A.transposecolors


@pytest.mark.parametrize("key", ["cuboid", "sphere"])
def test_colouring_equals_oracle_spec_and_is_valid(bsm, oracle, key):
    p = fixture_problem(key)
    S = bsm.SymmetricBlockMatrix(p["diagonals"], p["diagonalindices"], p["offdiagonals"],
                                 p["rowindices"], p["colindices"], p["size"], device=NODEV)
    for got, lists in ((S.offdiagonalcolors, p["rowindices"]), (S.transposeoffdiagonalcolors, p["colindices"]),
                       (S.diagonalcolors, p["diagonalindices"])):
        assert oracle.color_check(lists, got)
        assert got == oracle.color_workstream(lists)  # WorkstreamDSATUR (the reference's default), bit-exact vs the spec
    assert bsm.offdiagonalcolors(bsm.adjoint(S)) == S.transposeoffdiagonalcolors  # :307-325
    q = fixture_as_blocksparse(key)
    B = bsm.BlockSparseMatrix(q["blocks"], q["rowindices"], q["colindices"], q["size"],
                              scheduler=bsm.DynamicScheduler(), device=NODEV)
    assert B.colors == oracle.color_workstream(q["rowindices"])
    assert B.transposecolors == oracle.color_workstream(q["colindices"])
    D = bsm.BlockSparseMatrix(q["blocks"], q["rowindices"], q["colindices"], q["size"],
                              scheduler=bsm.DynamicScheduler(), coloringalgorithm="DSATUR", device=NODEV)
    assert D.colors == oracle.color_dsatur(q["rowindices"])  # the `coloringalgorithm` keyword, src/blockmatrix.jl:67
    assert D.transposecolors == oracle.color_dsatur(q["colindices"])


def test_colorinfo_adapter(bsm, oracle):
    # reference src/coloring.jl:45-61: conflicts() hands (1:n, functor, 1:maxindex) to the colouring
    lists = [[1, 2], [2, 3], [4], [3, 9]]
    info = bsm.ColorInfo(lists)
    ids, functor, rng = bsm.conflicts(info)
    assert list(ids) == [1, 2, 3, 4] and list(functor(2)) == [2, 3] and rng == range(1, 10)
    classes = bsm.color(info)
    assert classes == oracle.color_workstream(lists) and oracle.color_check(lists, classes)
    classes = bsm.color(info, "DSATUR")
    assert classes == oracle.color_dsatur(lists) and oracle.color_check(lists, classes)
    with pytest.raises(ValueError):
        bsm.color(info, "Greedy")


def test_colouring_random_lists(bsm, oracle):
    rng = np.random.default_rng(7)
    for _ in range(10):
        nb = int(rng.integers(2, 30))
        lists = [rng.choice(40, size=int(rng.integers(1, 6)), replace=False) + 1 for _ in range(nb)]
        blocks = [np.ones((len(l), 1)) for l in lists]
        A = bsm.BlockSparseMatrix(blocks, lists, [[1]] * nb, (40, 40), scheduler=bsm.DynamicScheduler(),
                                  device=NODEV)
        assert A.colors == oracle.color_workstream(lists)
        assert oracle.color_check(lists, A.colors)
        assert bsm.color(bsm.ColorInfo(lists), "DSATUR") == oracle.color_dsatur(lists)


def test_workstream_colouring_structure(bsm, oracle):
    """WorkstreamDSATUR as published (Turcksin, Kronbichler, Bangerth 2016, section 3.2) on graphs whose
    zones can be written down by hand."""
    # a path 1-2-3-4-5-6 (list k = {k, k+1}): zones {1},{2},...,{6}; one colour per zone; the even
    # zones (blocks 1,3,5) gather into one class, the odd ones (2,4,6) into the other
    path = [[k, k + 1] for k in range(1, 7)]
    assert bsm.color(bsm.ColorInfo(path)) == [[1, 3, 5], [2, 4, 6]] == oracle.color_workstream(path)
    # two components: a triangle {1,2,3} on index 1 and an isolated block 4.  Zones: {1}, {2,3}, {4}:
    # zone 1 needs two colours; zone 2 (the second seed) is even like zone 0 and joins its smallest class
    tri = [[1], [1], [1], [9]]
    got = bsm.color(bsm.ColorInfo(tri))
    assert got == oracle.color_workstream(tri) and oracle.color_check(tri, got)
    assert got == [[1, 4], [2], [3]]
    # star: centre block 1 touches every index; leaves are pairwise independent
    star = [[1, 2, 3, 4, 5]] + [[k] for k in range(1, 6)]
    assert bsm.color(bsm.ColorInfo(star)) == [[1], [2, 3, 4, 5, 6]]
    # larger random graphs: valid, deterministic, equal to the oracle's independent restatement
    rng = np.random.default_rng(11)
    for _ in range(20):
        nb = int(rng.integers(2, 60))
        lists = [rng.choice(50, size=int(rng.integers(1, 5)), replace=False) + 1 for _ in range(nb)]
        got = bsm.color(bsm.ColorInfo(lists))
        assert got == oracle.color_workstream(lists) and oracle.color_check(lists, got)
        assert sorted(b for c in got for b in c) == list(range(1, nb + 1))


# ---- nnz / size / accessors ---------------------------------------------------------------------
def test_nnz_and_accessors(bsm):
    p = fixture_problem("cuboid")
    S = bsm.SymmetricBlockMatrix(p["diagonals"], p["diagonalindices"], p["offdiagonals"],
                                 p["rowindices"], p["colindices"], p["size"], device=NODEV)
    assert bsm.nnz(S) == 2 * 93842 + 21264 == bsm.sparse(S).nnz
    assert bsm.nnz(bsm.adjoint(S)) == bsm.nnz(S)
    assert bsm.size(S) == (1344, 1344) and bsm.eltype(S) == np.complex128
    assert isinstance(bsm.scheduler(S), bsm.DynamicScheduler)  # tuple-size ctor default, :80
    assert np.array_equal(bsm.rowindices(bsm.transpose(S), 3), p["colindices"][2])
    assert np.array_equal(bsm.offdiagonal(bsm.adjoint(S), 1), p["offdiagonals"][0].conj().T)
    V = bsm.VariableBlockCompressedRowStorage([np.ones((2, 2))], [1], [1], (2, 2), device=NODEV)
    assert isinstance(bsm.scheduler(V), bsm.SerialScheduler)


def test_error_behaviour(bsm):
    with pytest.raises(IndexError):  # matrices[1] on an empty vector, src/vbcrs.jl:81
        bsm.VariableBlockCompressedRowStorage([], [], [], (4, 4), device=NODEV)
    with pytest.raises(RuntimeError, match="out of range"):
        bsm.BlockSparseMatrix([np.ones((1, 1))], [[5]], [[1]], (4, 4), device=NODEV)
    with pytest.raises(RuntimeError, match="outside matrix"):
        bsm.VariableBlockCompressedRowStorage([np.ones((3, 3))], [3], [1], (4, 4), device=NODEV)
    A = bsm.BlockSparseMatrix([np.ones((1, 1))], [[1]], [[1]], (4, 3), device=NODEV)
    with pytest.raises(ValueError, match="DimensionMismatch"):
        bsm.mul(np.zeros(4), A, np.zeros(4))
    with pytest.raises(ValueError, match="DimensionMismatch"):
        bsm.mul(np.zeros(4), bsm.transpose(A), np.zeros(4))


# ---- packed image == oracle ----------------------------------------------------------------------
def _check_image(bsm, oracle, problem, A, dtype, ops=OPS, tol=None):
    rng = np.random.default_rng(42)
    nr, nc = problem["size"]
    tol = tol or (1e-5 if np.dtype(dtype) in (np.float32, np.complex64) else 1e-13)
    for op in ops:
        if op == Cc and np.dtype(dtype).kind != "c":
            continue
        xl, yl = (nc, nr) if op == N else (nr, nc)
        x, y0 = rand_vec(rng, xl, dtype), rand_vec(rng, yl, dtype)
        for alpha, beta, strong in ((1, 0, True), (0.75, -1.5, False)):
            ref = oracle_mul(oracle, problem, op, x, y0, alpha, beta, strong)
            got = interpret_image(A, op, x, y0, alpha, beta, strong)
            assert relerr(got, ref) < tol, (op, alpha, beta, strong)


@pytest.mark.parametrize("key", ["cuboid", "sphere"])
@pytest.mark.parametrize("dtype,part", [(np.complex128, "full"), (np.float64, "real"), (np.float32, "imag")])
def test_image_symmetric_fixture(bsm, oracle, key, dtype, part):
    p = fixture_problem(key, dtype, part)
    A = bsm.synthetic.build(p, device=NODEV)
    assert A.stats()["exclusive"] == 0
    _check_image(bsm, oracle, p, A, dtype)


@pytest.mark.parametrize("wave_bytes", ["65536", "20000"])
def test_image_with_fat_waves_of_long_launches(bsm, oracle, monkeypatch, wave_bytes):
    # operators of hundreds of MB give every wave up to 64 KB (bsm_analysis.h Tunables::wave_bytes);
    # forced here on small ones: pieces of several 8 KB iterations and several x chunks per wave
    monkeypatch.setenv("BSM_WAVE_BYTES", wave_bytes)
    p = bsm.synthetic.config5(n=4000, lo=16, hi=160, halfband=3)
    A = bsm.synthetic.build(p, device=NODEV)
    monkeypatch.delenv("BSM_WAVE_BYTES")
    B = bsm.synthetic.build(p, device=NODEV)
    assert A.stats()["ntasks"] < B.stats()["ntasks"]
    _check_image(bsm, oracle, p, A, np.float64)
    monkeypatch.setenv("BSM_WAVE_BYTES", wave_bytes)
    q = bsm.synthetic.config2(n=6000, lo=20, hi=64, nblocks=400, dtype=np.float32)
    _check_image(bsm, oracle, q, bsm.synthetic.build(q, device=NODEV, transpose_image=True), np.float32)
    r = fixture_problem("cuboid")
    _check_image(bsm, oracle, r, bsm.synthetic.build(r, device=NODEV, accumulate="gather"), np.complex128)


def test_bytes_per_wave_follow_lane_fill_and_operator_size(bsm, monkeypatch):
    # long launches of row groups that do not fill their lanes get fat waves (Tunables::wave_bytes):
    # "long" is scaled down to test size through BSM_TARGET_WAVES
    low = bsm.synthetic.config5(n=6000, lo=8, hi=28, halfband=6)    # 8-28 rows in 8 / 16 / 32 lanes
    full = bsm.synthetic.config5(n=6000, lo=64, hi=64, halfband=2)   # 64 rows in 64 lanes
    def tasks(p):
        return bsm.synthetic.build(p, device=NODEV).stats()["ntasks"]
    thin_low, thin_full = tasks(low), tasks(full)
    monkeypatch.setenv("BSM_TARGET_WAVES", "64")
    assert tasks(low) < 0.7 * thin_low
    assert tasks(full) == thin_full
    monkeypatch.setenv("BSM_TARGET_WAVES", "1000000")  # too short to pay
    assert tasks(low) == thin_low


@pytest.mark.parametrize("key", ["cuboid", "sphere"])
def test_image_blocksparse_fixture(bsm, oracle, key):
    p = fixture_as_blocksparse(key)
    A = bsm.synthetic.build(p, device=NODEV)
    assert A.stats()["exclusive"] == 1  # the fixture's test lists are mutually disjoint
    _check_image(bsm, oracle, p, A, np.complex128)


def test_image_config1_blocksparse(bsm, oracle):
    p = bsm.synthetic.config1()
    A = bsm.synthetic.build(p, device=NODEV)
    assert A.stats()["exclusive"] == 0  # random row lists overlap
    assert bsm.nnz(A) == 50 * 32 * 32
    _check_image(bsm, oracle, p, A, np.float64)


def test_blocks_of_any_matrix_type_and_indices_of_any_integer_type(bsm, oracle):
    """The reference takes any AbstractMatrix as a block (counted as prod(size): src/abstractblockmatrix.jl:65-71)
    and any P <: Integer as an index.  The mirror densifies a sparse block once at construction, takes row-major
    arrays, views with strides, np.matrix, Float16 and integer blocks (promoted like Julia would with Float32 /
    Float64 scalars) and Int32 / UInt16 indices -- the packed image and `sparse(A)` are those of the dense blocks."""
    import scipy.sparse as sp
    rng = np.random.default_rng(12)
    dense = [rng.standard_normal((9, 7)), rng.standard_normal((5, 11)), rng.standard_normal((6, 6)), rng.standard_normal((4, 3))]
    big = rng.standard_normal((20, 30))
    dense[1] = big[2:7, 3:25:2].copy()
    mixed = [sp.csr_matrix(dense[0]), big[2:7, 3:25:2], np.matrix(dense[2]), sp.csc_matrix(dense[3])]
    rows = [np.arange(1, 10, dtype=np.int32), np.arange(20, 25, dtype=np.uint16), np.arange(30, 36), np.arange(40, 44, dtype=np.int32)]
    cols = [np.arange(3, 10, dtype=np.int32), np.arange(12, 23, dtype=np.int64), np.arange(30, 36, dtype=np.uint16), np.arange(1, 4)]
    ref = bsm.BlockSparseMatrix(dense, [r.astype(np.int64) for r in rows], [c.astype(np.int64) for c in cols], (50, 40), device=NODEV)
    A = bsm.BlockSparseMatrix(mixed, rows, cols, (50, 40), device=NODEV)
    assert A.dtype == np.float64 and bsm.nnz(A) == bsm.nnz(ref) == sum(d.size for d in dense)
    assert abs(bsm.sparse(A) - bsm.sparse(ref)).max() == 0
    p = dict(kind="blocksparse", blocks=[np.asfortranarray(d) for d in dense], rowindices=[r.astype(np.int64) for r in rows],
             colindices=[c.astype(np.int64) for c in cols], size=(50, 40))
    _check_image(bsm, oracle, p, A, np.float64)
    # element types: Float16 -> Float32, Int32 -> Float64, a complex block promotes the whole matrix
    h = bsm.BlockSparseMatrix([d.astype(np.float16) for d in dense], rows, cols, (50, 40), device=NODEV)
    assert h.dtype == np.float32
    i = bsm.VariableBlockCompressedRowStorage([np.arange(12, dtype=np.int32).reshape(3, 4)], [2], [5], (10, 10), device=NODEV)
    assert i.dtype == np.float64 and bsm.sparse(i)[3, 6] == 10  # row-major input: 0-based entry (2, 2) of the block = 2 * 4 + 2
    c = bsm.BlockSparseMatrix([dense[0], sp.csr_matrix(dense[1] * (1 + 2j))], rows[:2], cols[:2], (50, 40), device=NODEV)
    assert c.dtype == np.complex128


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_image_config2_vbcrs_small(bsm, oracle, dtype):
    p = bsm.synthetic.config2(n=4000, nblocks=400, dtype=dtype)
    A = bsm.synthetic.build(p, device=NODEV)
    st = A.stats()
    assert st["exclusive"] == 1 and st["nnz"] == sum(b.size for b in p["blocks"])
    _check_image(bsm, oracle, p, A, dtype)


def test_image_config3_symmetric_small(bsm, oracle):
    p = bsm.synthetic.config3(nseg=40, bs=64, halfband=8)
    A = bsm.synthetic.build(p, device=NODEV)
    _check_image(bsm, oracle, p, A, np.float64)


def test_coarser_wave_records_of_the_multi_rhs_kernels(bsm, oracle, monkeypatch):
    # bsm_mul_multi walks a second, coarser split of the same panels (waves of 64 KB instead of 8 KB: a wave's
    # fixed part is K times the single product's): same value stream, every strip exactly once -- the image
    # interpreter on those records must give the same products -- and clearly fewer waves; operators whose
    # ordinary split is as coarse already (small panels), exclusive and coloured ones carry no second list
    from _common import WORK_PANEL, get_image
    rng = np.random.default_rng(5)
    for p in (bsm.synthetic.config3(nseg=24, bs=64, halfband=8), bsm.synthetic.config5(n=9000, lo=16, hi=256, halfband=4)):
        A = bsm.synthetic.build(p, device=NODEV)
        main, multi = get_image(A)[3], get_image(A, multi=True)[3]
        nm, nk = int(np.sum(main["work"] == WORK_PANEL)), int(np.sum(multi["work"] == WORK_PANEL))
        assert 0 < nk <= 0.75 * nm and len(multi) % 4 == 0
        n = p["size"][0]
        x, y0 = rand_vec(rng, n, np.float64), rand_vec(rng, n, np.float64)
        for op in (N, T):
            ref = oracle_mul(oracle, p, op, x, y0, 0.5, 2.0, False)
            assert relerr(interpret_image(A, op, x, y0, 0.5, 2.0, False, multi=True), ref) < 1e-13
        # panel waves of one group split ITS strips: bytes per wave near the 64 KB target, never above 4 x
        by = multi["first"]["nstrips"].astype(np.int64) * multi["m"] * 16
        assert by.max() <= 4 * 65536 and by[multi["work"] == WORK_PANEL].mean() > 3 * (main["first"]["nstrips"].astype(np.int64) * main["m"] * 16)[main["work"] == WORK_PANEL].mean() / 2
    small = fixture_problem("cuboid", np.float64, "real")              # gathered columns, 3-28-row panels
    A = bsm.synthetic.build(small, device=NODEV)
    if len(get_image(A, multi=True)[3]):
        n = small["size"][0]
        x, y0 = rand_vec(rng, n, np.float64), rand_vec(rng, n, np.float64)
        assert relerr(interpret_image(A, N, x, y0, 1, 0, True, multi=True), oracle_mul(oracle, small, N, x, y0, 1, 0, True)) < 1e-13
    v = bsm.synthetic.config2(n=20000, nblocks=1000)                   # exclusive forward launch
    assert len(get_image(bsm.synthetic.build(v, device=NODEV), multi=True)[3]) == 0
    p = bsm.synthetic.config3(nseg=24, bs=64, halfband=8)
    assert len(get_image(bsm.synthetic.build(p, device=NODEV, accumulate="colored"), multi=True)[3]) == 0
    monkeypatch.setenv("BSM_MULTI_WAVE_BYTES", "0")
    assert len(get_image(bsm.synthetic.build(p, device=NODEV), multi=True)[3]) == 0


def test_image_config4_vbcrs_small_f32(bsm, oracle):
    p = bsm.synthetic.config4(ngrid=24, bs=128, per_row=6)
    A = bsm.synthetic.build(p, device=NODEV)
    assert A.stats()["exclusive"] == 1
    _check_image(bsm, oracle, p, A, np.float32)


def test_image_config5_symmetric_small(bsm, oracle):
    p = bsm.synthetic.config5(n=6000, lo=16, hi=256, halfband=4)
    A = bsm.synthetic.build(p, device=NODEV)
    _check_image(bsm, oracle, p, A, np.float64)


def test_image_edge_cases(bsm, oracle):
    rng = np.random.default_rng(3)
    # 1x1 blocks, empty blocks, m > 64 (row chunks), odd widths, duplicate index inside a list,
    # two blocks on the same rows (grouped), rows nobody covers
    blocks = [rng.standard_normal((1, 1)), np.zeros((0, 3)), np.zeros((2, 0)),
              rng.standard_normal((130, 7)), rng.standard_normal((5, 9)), rng.standard_normal((5, 3)),
              rng.standard_normal((3, 2))]
    rows = [[7], [], [1, 2], list(range(20, 150)), [1, 3, 5, 7, 9], [1, 3, 5, 7, 9], [200, 200, 201]]
    cols = [[9], [1, 2, 3], [], [4, 3, 2, 1, 10, 11, 12], list(range(50, 59)), [2, 4, 6], [1, 1]]
    p = dict(kind="blocksparse", blocks=[np.asfortranarray(b) for b in blocks], rowindices=rows,
             colindices=cols, size=(210, 60))
    A = bsm.synthetic.build(p, device=NODEV)
    assert A.stats()["exclusive"] == 0
    _check_image(bsm, oracle, p, A, np.float64)


def test_image_vbcrs_unequal_heights_and_overlap(bsm, oracle):
    # reference semantics (src/vbcrs.jl:277-283): heights are per block, nothing is checked.
    rng = np.random.default_rng(4)
    blocks = [rng.standard_normal((4, 4)), rng.standard_normal((6, 2)), rng.standard_normal((3, 5))]
    p = dict(kind="vbcrs", blocks=[np.asfortranarray(b) for b in blocks],
             rowstart=np.array([1, 1, 3]), colstart=np.array([1, 5, 2]), size=(8, 8))
    A = bsm.synthetic.build(p, device=NODEV)
    assert A.stats()["exclusive"] == 0  # overlapping block rows -> atomics
    _check_image(bsm, oracle, p, A, np.float64)


def test_image_vbcrs_from_symmetric_and_blocksparse(bsm, oracle):
    # converters, reference src/vbcrs.jl:150-264 (enumeration [diag..., off..., transpose(off)...])
    p = bsm.synthetic.config3(nseg=12, bs=16, halfband=3)
    S = bsm.synthetic.build(p, device=NODEV)
    V = bsm.VariableBlockCompressedRowStorage(S, device=NODEV)
    assert bsm.nnz(V) == bsm.nnz(S)
    rng = np.random.default_rng(8)
    n = p["size"][0]
    x, y0 = rand_vec(rng, n, np.float64), rand_vec(rng, n, np.float64)
    ref = oracle_mul(oracle, p, N, x, y0)
    assert relerr(interpret_image(V, N, x, y0), ref) < 1e-13
    assert relerr(interpret_image(V, T, x, y0), ref) < 1e-13
    q = bsm.synthetic.config2(n=600, nblocks=60)
    ri = [np.arange(r, r + b.shape[0]) for r, b in zip(q["rowstart"], q["blocks"])]
    ci = [np.arange(c, c + b.shape[1]) for c, b in zip(q["colstart"], q["blocks"])]
    B = bsm.BlockSparseMatrix(q["blocks"], ri, ci, q["size"], device=NODEV)
    V2 = bsm.VariableBlockCompressedRowStorage(B, device=NODEV)
    x, y0 = rand_vec(rng, 600, np.float64), rand_vec(rng, 600, np.float64)
    assert relerr(interpret_image(V2, N, x, y0), interpret_image(B, N, x, y0)) < 1e-13


def test_vbcrs_symmetric_view_equals_materialised_conversion(bsm, oracle):
    # VariableBlockCompressedRowStorage(sbm): the view keeps every off-diagonal block once; the
    # reference-style conversion materialises the transposes (src/vbcrs.jl:222-241).  Bookkeeping
    # must be identical, products equal, nnz equal (test/test_vbcrs.jl:65-87).
    p = bsm.synthetic.config5(n=3000, lo=1, hi=90, halfband=3)
    S = bsm.synthetic.build(p, device=NODEV)
    V = bsm.VariableBlockCompressedRowStorage(S, device=NODEV)
    M = bsm.VariableBlockCompressedRowStorage(S, device=NODEV, materialize=True)
    assert bsm.nnz(S) == bsm.nnz(V) == bsm.nnz(M)
    assert M.stats()["stored_entries"] == bsm.nnz(S)
    assert V.stats()["stored_entries"] == S.stats()["stored_entries"] < bsm.nnz(S)
    for f in ("perm", "rowptr", "colindices", "rowindices"):
        assert np.array_equal(getattr(V, f), getattr(M, f))
    # bookkeeping against the oracle's restatement of the constructor on the expanded list
    nd = len(p["diagonals"])
    rs = ([int(d[0]) for d in p["diagonalindices"]] + [int(r[0]) for r in p["rowindices"]]
          + [int(c[0]) for c in p["colindices"]])
    cs = ([int(d[0]) for d in p["diagonalindices"]] + [int(c[0]) for c in p["colindices"]]
          + [int(r[0]) for r in p["rowindices"]])
    perm, rowptr, colind, rowind = oracle.vbcrs_build(rs, cs)
    assert np.array_equal(V.perm, perm) and np.array_equal(V.rowptr, rowptr)
    assert np.array_equal(V.colindices, colind) and np.array_equal(V.rowindices, rowind)
    assert len(V.blocks) == nd + 2 * len(p["offdiagonals"])
    assert abs(bsm.sparse(V) - bsm.sparse(S)).max() == 0
    rng = np.random.default_rng(1)
    n = p["size"][0]
    x, y0 = rand_vec(rng, n, np.float64), rand_vec(rng, n, np.float64)
    ref = oracle_mul(oracle, p, N, x, y0, 0.5, 2.0, False)
    for A in (V, M):
        for op in (N, T):
            assert relerr(interpret_image(A, op, x, y0, 0.5, 2.0, False), ref) < 1e-13


def test_second_ordering_transposed_image(bsm, oracle):
    # bsm_options.transpose_image: op T / C run FORWARD on a second, transposed ordering
    # (the reference author's TODO, src/vbcrs.jl:124)
    rng = np.random.default_rng(12)
    p = bsm.synthetic.config2(n=4000, nblocks=300)
    A = bsm.synthetic.build(p, device=NODEV, transpose_image=True)
    A1 = bsm.synthetic.build(p, device=NODEV)
    assert A.stats()["device_bytes"] > 1.9 * A1.stats()["device_bytes"]
    x, y0 = rand_vec(rng, 4000, np.float64), rand_vec(rng, 4000, np.float64)
    for alpha, beta, strong in ((1, 0, True), (0.5, 2.0, False)):
        ref = oracle_mul(oracle, p, T, x, y0, alpha, beta, strong)
        assert relerr(interpret_image(A, T, x, y0, alpha, beta, strong, timage=True), ref) < 1e-13
    q = fixture_as_blocksparse("cuboid")
    B = bsm.synthetic.build(q, device=NODEV, transpose_image=True)
    n = q["size"][0]
    x, y0 = rand_vec(rng, n, np.complex128), rand_vec(rng, n, np.complex128)
    for op in (T, Cc):
        ref = oracle_mul(oracle, q, op, x, y0, 1j, 2j, False)
        assert relerr(interpret_image(B, op, x, y0, 1j, 2j, False, timage=True), ref) < 1e-13
    # rectangular, m > 64 chunks, odd sizes
    blocks = [np.asfortranarray(rng.standard_normal((130, 7))), np.asfortranarray(rng.standard_normal((5, 9)))]
    r = dict(kind="blocksparse", blocks=blocks, rowindices=[list(range(20, 150)), [1, 3, 5, 7, 9]],
             colindices=[[4, 3, 2, 1, 10, 11, 12], list(range(50, 59))], size=(210, 60))
    Cm = bsm.synthetic.build(r, device=NODEV, transpose_image=True)
    x, y0 = rand_vec(rng, 210, np.float64), rand_vec(rng, 60, np.float64)
    ref = oracle_mul(oracle, r, T, x, y0, 0.5, 2.0, False)
    assert relerr(interpret_image(Cm, T, x, y0, 0.5, 2.0, False, timage=True), ref) < 1e-13


def test_lds_window_descriptors_cover_their_workgroup(bsm):
    # workgroups of a symmetric operator that pack neighbouring small row groups carry an LDS
    # accumulation window: workgroup-uniform, at most window_entries(8) = 512 entries, placed on the
    # best-filled index range of what the workgroup emits in op N (forward rows of its leading waves +
    # KIND_OFF columns); it need not hold everything, but must merge at least a fifth of what it holds
    from _common import KIND_OFF, WORK_PANEL, get_image

    def check(p, cap, dt=np.float64):
        A = bsm.synthetic.build(p, device=NODEV)
        values, rows, cols, waves = get_image(A)
        st = A.stats()
        wg = waves.reshape(-1, st["ntasks"] // st["nworkgroups"])  # waves per workgroup
        nwin = tot = inside = flushed = 0
        for quad in wg:
            assert len(set(quad["win_span8"])) == 1 and len(set(quad["win_base"])) == 1
            span = int(quad["win_span8"][0]) * 8
            assert span <= cap
            base = int(quad["win_base"][0])
            groups, em = set(), []
            for W in quad[(quad["work"] == WORK_PANEL) & (quad["npieces"] > 0)]:
                m = int(W["m"])
                r = np.arange(W["rbase"], W["rbase"] + m) if W["rbase"] >= 0 else rows[W["row_off"]:W["row_off"] + m]
                pool = cols[W["first"]["col_off"]:W["first"]["col_off"] + W["first"]["ncols"]]
                kinds = int(W["first"]["kind"])
                nc = len(pool)
                if W["first"]["xbase"] >= 0:
                    wv = np.arange(nc)
                    ck = np.where(wv < W["seg1_w"], kinds & 3, np.where(wv < W["seg2_w"], (kinds >> 2) & 3, (kinds >> 4) & 3))
                else:
                    ck = np.where(pool < 0, 1, kinds & 3)
                if W["lead"]:
                    em += list(r)
                em += list((pool & 0x7fffffff)[ck == KIND_OFF])
                groups.add((int(W["rbase"]), int(W["row_off"])))
            tot += len(em)
            if span == 0:
                continue
            nwin += 1
            em = np.array(em)
            inw = em[(em >= base) & (em < base + span)]
            assert len(groups) >= 2 and len(inw) >= 32 and 5 * len(np.unique(inw)) <= 4 * len(inw)
            assert inw.min() == base and inw.max() >= base + span - 8
            inside += len(inw)
            flushed += len(np.unique(inw))
        assert nwin > 0
        assert (st["win_emissions"], st["win_inside"], st["win_flushed"]) == (tot, inside, flushed)
        return A, (tot - inside + flushed) / tot

    p = bsm.synthetic.config5(n=20000, lo=8, hi=28, halfband=6)
    check(p, 512)
    # coloured handles never use the window (plain RMW per colour class)
    B = bsm.synthetic.build(p, device=NODEV, accumulate="colored")
    assert not np.any(get_image(B)[3]["win_span8"])
    # the reference's BEM fixture (ComplexF64: 256 entries): panels with a few far columns still get a
    # window over their dense part, and fewer than 0.7 of the contributions leave a CU as atomics
    # (one wave per leaf, as the operator-dependent wave size gives a long launch of such panels)
    os.environ["BSM_WAVE_BYTES"] = "24576"
    try:
        _, frac = check(fixture_problem("cuboid"), 256)
    finally:
        del os.environ["BSM_WAVE_BYTES"]
    assert frac < 0.7, frac


@pytest.mark.parametrize("key", ["cuboid", "sphere"])
def test_gather_mode_image_matches_oracle(bsm, oracle, key):
    # BSM_ACC_GATHER: contributions are stored in workspace slots, an inverted index sums them
    p = fixture_problem(key)
    A = bsm.synthetic.build(p, device=NODEV, accumulate="gather")
    assert A.stats()["exclusive"] == 0
    _check_image(bsm, oracle, p, A, np.complex128)
    q = bsm.synthetic.config1()
    _check_image(bsm, oracle, q, bsm.synthetic.build(q, device=NODEV, accumulate="gather"), np.float64)
    r = bsm.synthetic.config5(n=3000, lo=1, hi=90, halfband=3)
    _check_image(bsm, oracle, r, bsm.synthetic.build(r, device=NODEV, accumulate="gather"), np.float64)
    v = bsm.synthetic.config2(n=3000, nblocks=200)
    _check_image(bsm, oracle, v, bsm.synthetic.build(v, device=NODEV, accumulate="gather"), np.float64)


def test_own_range_limits_scale_work(bsm):
    p = bsm.synthetic.config2(n=2000, nblocks=40)
    A = bsm.VariableBlockCompressedRowStorage(p["blocks"], p["rowstart"], p["colstart"], p["size"],
                                              device=NODEV, own=(501, 1000))
    from _common import WORK_SCALE, get_image
    waves = get_image(A)[3]
    sc = waves[waves["work"] == WORK_SCALE]
    assert len(sc) > 0
    assert sc["rbase"].min() >= 500 and (sc["rbase"] + sc["first"]["ncols"]).max() <= 1000


# ---- coloured accumulation (BSM_ACC_COLORED): launches of pairwise conflict-free row groups ----
@pytest.mark.parametrize("key", ["cuboid", "sphere"])
def test_coloured_mode_image_matches_oracle(bsm, oracle, key):
    p = fixture_problem(key)
    A = bsm.synthetic.build(p, device=NODEV, accumulate="colored")
    assert A.stats()["exclusive"] == 0
    _check_image(bsm, oracle, p, A, np.complex128)
    q = bsm.synthetic.config1()
    B = bsm.synthetic.build(q, device=NODEV, accumulate="colored")
    _check_image(bsm, oracle, q, B, np.float64)


def test_coloured_mode_rejects_repeated_indices(bsm):
    with pytest.raises(RuntimeError, match="repeats a row index"):
        bsm.BlockSparseMatrix([np.ones((2, 1))], [[3, 3]], [[1]], (4, 4), device=NODEV, accumulate="colored")
    with pytest.raises(RuntimeError, match="repeat a column index"):
        bsm.BlockSparseMatrix([np.ones((1, 2))], [[1]], [[2, 2]], (4, 4), device=NODEV, accumulate="colored")


# ---- C ABI argument checking: errors are status codes + bsm_last_error, never crashes --------------
def test_c_abi_rejects_bad_arguments(bsm):
    from bsm_amd import _lib as L
    lib = L.lib()
    I = C.POINTER(C.c_int64)
    o = L.BsmOptions()
    lib.bsm_options_default(C.byref(o))
    o.device = NODEV
    h = C.c_void_p()
    blk = np.asfortranarray(np.ones((2, 2)))
    ptrs = (C.c_void_p * 1)(blk.ctypes.data)
    one = np.array([1], dtype=np.int64)
    two = np.array([2], dtype=np.int64)

    def p(a):
        return a.ctypes.data_as(I)

    # null handle / null out
    assert lib.bsm_mul(None, 0, None, None, None, None, 1, 1, None) == -1
    assert b"null handle" in lib.bsm_last_error()
    assert lib.bsm_vbcrs_create(1, 4, 4, 1, ptrs, p(two), p(two), p(two), p(one), p(one), C.byref(o), None) == -1
    # zero blocks, bad dtype, ld < m, range outside matrix, wrong options size
    assert lib.bsm_vbcrs_create(1, 4, 4, 0, ptrs, p(two), p(two), p(two), p(one), p(one), C.byref(o), C.byref(h)) == -1
    assert lib.bsm_vbcrs_create(9, 4, 4, 1, ptrs, p(two), p(two), p(two), p(one), p(one), C.byref(o), C.byref(h)) == -1
    assert b"dtype" in lib.bsm_last_error()
    assert lib.bsm_vbcrs_create(1, 4, 4, 1, ptrs, p(two), p(two), p(one), p(one), p(one), C.byref(o), C.byref(h)) == -1
    assert b"ld" in lib.bsm_last_error()
    four = np.array([4], dtype=np.int64)
    assert lib.bsm_vbcrs_create(1, 4, 4, 1, ptrs, p(two), p(two), p(two), p(four), p(one), C.byref(o), C.byref(h)) == -1
    bad = L.BsmOptions()
    bad.struct_size = 8
    assert lib.bsm_vbcrs_create(1, 4, 4, 1, ptrs, p(two), p(two), p(two), p(one), p(one), C.byref(bad), C.byref(h)) == -1
    assert b"struct_size" in lib.bsm_last_error()
    # a good handle: op out of range, null vectors, multi-RHS argument checks, bookkeeping misuse
    assert lib.bsm_vbcrs_create(1, 4, 4, 1, ptrs, p(two), p(two), p(two), p(one), p(one), C.byref(o), C.byref(h)) == 0
    x = np.zeros(4)
    assert lib.bsm_mul(h, 7, x.ctypes.data, x.ctypes.data, None, None, 1, 0, None) == -1
    assert lib.bsm_mul(h, 0, None, x.ctypes.data, None, None, 1, 0, None) == -1
    assert lib.bsm_mul(h, 0, x.ctypes.data, x.ctypes.data, None, None, 1, 0, None) == -3  # no device image
    assert lib.bsm_mul_multi(h, 0, -1, x.ctypes.data, 4, x.ctypes.data, 4, None, None, 1, 0, None) == -1
    assert lib.bsm_mul_multi(h, 0, 0, x.ctypes.data, 4, x.ctypes.data, 4, None, None, 1, 0, None) == 0
    n = C.c_int64(0)
    assert lib.bsm_get_bookkeeping(h, 99, None, C.byref(n)) == -1
    assert lib.bsm_get_bookkeeping(h, L.BSM_BK_COLORS, None, C.byref(n)) == -1  # VBCRS has no colours
    assert lib.bsm_get_bookkeeping(h, L.BSM_BK_VBCRS_ROWPTR, None, C.byref(n)) == 0 and n.value == 2
    small = np.zeros(1, dtype=np.int64)
    n = C.c_int64(1)
    assert lib.bsm_get_bookkeeping(h, L.BSM_BK_VBCRS_ROWPTR, p(small), C.byref(n)) == -1  # buffer too small
    assert lib.bsm_destroy(h) == 0 and lib.bsm_destroy(None) == 0
    # colouring entry: 0-based index is rejected
    lst = np.array([0, 1], dtype=np.int64)
    lp = (C.c_void_p * 1)(lst.ctypes.data)
    out = np.zeros(1, dtype=np.int64)
    nc = C.c_int64(0)
    assert lib.bsm_color(1, lp, p(two), 0, p(out), C.byref(nc)) == -1


def test_auto_mode_splits_only_large_deep_operators(bsm):
    # conflict-free operators stay exclusive (one launch, direct stores) unless they are large AND
    # made of deep row groups (include/bsm_rocm.h: BSM_ACC_AUTO / BSM_ACC_DIRECT)
    small = bsm.synthetic.config4(ngrid=15625, row_lo=0, row_hi=20)      # 21 MB of 512 KB row groups
    assert bsm.synthetic.build(small, device=NODEV).stats()["exclusive"] == 1
    big = bsm.synthetic.config4(ngrid=15625, row_lo=0, row_hi=80)        # 84 MB of 512 KB row groups
    assert bsm.synthetic.build(big, device=NODEV).stats()["exclusive"] == 0
    assert bsm.synthetic.build(big, device=NODEV, accumulate="direct").stats()["exclusive"] == 1
    c2 = bsm.synthetic.config2(n=200000, nblocks=10000)                   # 110 MB of ~20 KB row groups
    assert bsm.synthetic.build(c2, device=NODEV).stats()["exclusive"] == 1


def test_partition_rows_c_abi(bsm):
    """bsm_partition_rows: contiguous key ranges, balanced weights, own ranges tile the rows, blocks
    with equal keys stay together, empty parts get empty ranges -- the partition both multi-GPU layers use."""
    rng = np.random.default_rng(0)
    keys = np.sort(rng.integers(1, 5000, 400))
    w = rng.integers(1, 1000, 400)
    for nparts in (1, 2, 3, 8):
        part, own = bsm.partition_rows(5000, keys, w, nparts)
        assert np.all(np.diff(part) >= 0), "sorted keys -> monotone parts"
        for k in np.unique(keys):
            assert len(set(part[keys == k])) == 1
        assert own[0][0] == 1 and own[-1][1] == 5000 or any(hi < lo for lo, hi in own)
        cover = np.zeros(5000, int)
        for p, (lo, hi) in enumerate(own):
            if hi >= lo:
                cover[lo - 1:hi] += 1
            assert np.all((keys[part == p] >= lo) & (keys[part == p] <= hi))
        assert np.all(cover == 1)
        loads = np.array([w[part == p].sum() for p in range(nparts)])
        assert loads.max() <= w.sum() / nparts + w.max() * 2
    # more parts than keys: the surplus parts are empty, everything is still owned exactly once
    part, own = bsm.partition_rows(100, [10, 40], [5, 5], 4)
    assert sorted(part.tolist()) == sorted(set(part.tolist())) and sum(hi >= lo for lo, hi in own) == 2
    cover = np.zeros(100, int)
    for lo, hi in own:
        if hi >= lo:
            cover[lo - 1:hi] += 1
    assert np.all(cover == 1)
    # no block at all: part 0 owns every row
    part, own = bsm.partition_rows(7, [], [], 3)
    assert own[0] == (1, 7) and all(hi < lo for lo, hi in own[1:])
    with pytest.raises(bsm._lib.BsmError):
        bsm.partition_rows(10, [11], [1], 2)


def test_vbcrs_from_blocksparse_c_abi_matches_the_reference_converter(bsm, oracle):
    """bsm_vbcrs_create_from_blocksparse (reference src/vbcrs.jl:150-160, 201-215): block i sits at
    (first(rowindices[i]), first(colindices[i])); bookkeeping bit-exact against the oracle's constructor."""
    rng = np.random.default_rng(4)
    n = 400
    cuts = np.sort(rng.choice(np.arange(2, n), 30, replace=False))
    segs = [np.arange(a, b) for a, b in zip(np.r_[1, cuts], np.r_[cuts, n + 1])]
    pairs = {(int(rng.integers(len(segs))), int(rng.integers(len(segs)))) for _ in range(60)}
    pairs = list(pairs)
    rng.shuffle(pairs)
    blocks = [np.asfortranarray(rng.standard_normal((len(segs[i]), len(segs[j])))) for i, j in pairs]
    B = bsm.BlockSparseMatrix(blocks, [segs[i] for i, _ in pairs], [segs[j] for _, j in pairs], (n, n), device=NODEV)
    V = bsm.VariableBlockCompressedRowStorage(B, device=NODEV)
    perm, rowptr, colind, rowind = oracle.vbcrs_build([int(segs[i][0]) for i, _ in pairs],
                                                      [int(segs[j][0]) for _, j in pairs])
    assert np.array_equal(V.perm, perm) and np.array_equal(V.rowptr, rowptr)
    assert np.array_equal(V.colindices, colind) and np.array_equal(V.rowindices, rowind)
    assert bsm.nnz(V) == bsm.nnz(B)
    x = rng.standard_normal(n)
    got = interpret_image(V, N, x, np.zeros(n))
    assert relerr(got, bsm.sparse(B) @ x) < 1e-13
    with pytest.raises(bsm._lib.BsmError, match="empty index list"):
        bsm.VariableBlockCompressedRowStorage(
            bsm.BlockSparseMatrix([np.zeros((0, 2))], [[]], [[1, 2]], (4, 4), device=NODEV), device=NODEV)
