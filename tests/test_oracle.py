"""CPU suite, part 1: pin the ORACLE (oracle/bsm_oracle.c).

The reference (pure Julia) cannot run in the build image and its tests store no expected
outputs, so the oracle is pinned exactly the way the reference's own tests pin the reference:
on the reference's fixture (test/assets/symmetricblockexamples.jld2 -> tests/golden/*.bin),
every product is compared with an independent sparse COO product (reference
test/test_symmetricblockmatrix.jl:48-97, test/test_vbcrs.jl:33-47) -- here scipy.sparse and the
oracle's own orc_coo_mul -- plus the hand-derived known answers of SURVEY.md section 8c.
"""
import numpy as np
import pytest

from _common import (Cc, N, T, coo_of, fixture_as_blocksparse, fixture_problem, oracle_mul, rand_vec,
                     relerr, scipy_mul)

OPS = [N, T, Cc]


# ---- hand-derived known answers -------------------------------------------------
def test_kat_vbcrs_constructor(oracle):
    # reference src/vbcrs.jl:84-117: (row0,col0) = [(5,1),(1,7),(1,3),(5,9)]
    perm, rowptr, colind, rowind = oracle.vbcrs_build([5, 1, 1, 5], [1, 7, 3, 9])
    assert perm.tolist() == [3, 2, 1, 4]
    assert rowptr.tolist() == [1, 3, 5]
    assert rowind.tolist() == [1, 5]
    assert colind.tolist() == [3, 7, 1, 9]


def test_kat_vbcrs_constructor_ties_are_stable(oracle):
    # sortperm is stable: equal (row, col) keys keep input order (src/vbcrs.jl:84)
    perm, rowptr, colind, rowind = oracle.vbcrs_build([2, 2, 1, 2], [4, 4, 1, 4])
    assert perm.tolist() == [3, 1, 2, 4]
    assert rowptr.tolist() == [1, 2, 5]
    assert rowind.tolist() == [1, 2]


def test_kat_blocksparse_mul(oracle):
    # reference src/blockmatrix.jl:231-244: 4x4, B1=[1 2;3 4] rows [1,3] cols [2,4], B2=[5] rows [3] cols [1]
    p = dict(kind="blocksparse", blocks=[np.array([[1., 2.], [3., 4.]]), np.array([[5.]])],
             rowindices=[[1, 3], [3]], colindices=[[2, 4], [1]], size=(4, 4))
    x = np.array([1., 2., 3., 4.])
    y = oracle_mul(oracle, p, N, x, np.full(4, np.nan))
    assert y.tolist() == [10., 0., 27., 0.]


def test_kat_symmetric_mul(oracle):
    # reference src/symmetricblockmatrix.jl:394-432: D=[7] on [2]; B=[1 2] rows [1] cols [3,4]
    p = dict(kind="symmetric", diagonals=[np.array([[7.]])], diagonalindices=[[2]],
             offdiagonals=[np.array([[1., 2.]])], rowindices=[[1]], colindices=[[3, 4]], size=(4, 4))
    x = np.array([1., 2., 3., 4.])
    y = oracle_mul(oracle, p, N, x, np.full(4, np.nan))
    assert y.tolist() == [11., 14., 1., 2.]


def test_strong_zero_vs_numeric_zero(oracle):
    # beta === false wipes NaN (src/abstractblockmatrix.jl:27-34); numeric 0.0 propagates it
    p = dict(kind="blocksparse", blocks=[np.array([[2.]])], rowindices=[[1]], colindices=[[1]], size=(2, 2))
    x = np.array([3., 0.])
    y0 = np.array([np.nan, np.nan])
    assert oracle_mul(oracle, p, N, x, y0, 1, 0, strong=True).tolist() == [6., 0.]
    assert np.all(np.isnan(oracle_mul(oracle, p, N, x, y0, 1, 0.0, strong=False)))


# ---- the reference's fixture, checked like the reference's tests check it ----------------------
@pytest.mark.parametrize("key", ["cuboid", "sphere"])
def test_fixture_shape_and_symmetry(key):
    import scipy.sparse as sp
    p = fixture_problem(key)
    stats = {"cuboid": (1344, 96, 92, 21264, 93842), "sphere": (1203, 106, 103, 16501, 87718)}[key]
    assert (p["size"][0], len(p["diagonals"]), len(p["offdiagonals"]),
            sum(b.size for b in p["diagonals"]), sum(b.size for b in p["offdiagonals"])) == stats
    r, c, v = coo_of(p)
    S = sp.coo_matrix((v, (r - 1, c - 1)), shape=p["size"]).tocsr()
    assert abs(S - S.T).max() == 0  # issymmetric(sparse(b)), test_symmetricblockmatrix.jl:49
    # nnz as the reference defines it (src/symmetricblockmatrix.jl:377-382) equals nnz(sparse)
    assert 2 * stats[4] + stats[3] == S.nnz


@pytest.mark.parametrize("key", ["cuboid", "sphere"])
@pytest.mark.parametrize("op", OPS)
def test_fixture_symmetric_products(oracle, key, op):
    p = fixture_problem(key)
    n = p["size"][0]
    rng = np.random.default_rng(1234 + op)
    for _ in range(3):
        x = rand_vec(rng, n, np.complex128)
        y0 = rand_vec(rng, n, np.complex128)
        # 3-arg form, then mul!(x, b, y, im, 2im)  (test_symmetricblockmatrix.jl:67-97)
        for alpha, beta, strong in ((1, 0, True), (1j, 2j, False)):
            got = oracle_mul(oracle, p, op, x, y0, alpha, beta, strong)
            ref = scipy_mul(p, op, x, y0, alpha, beta, strong)
            assert relerr(got, ref) < 1e-13
            r, c, v = coo_of(p)
            if op == T:
                r, c = c, r
            elif op == Cc:
                r, c, v = c, r, v.conj()
            ref2 = oracle.coo_mul(r, c, v, x, np.array(y0, copy=True), alpha, beta, strong)
            assert relerr(got, ref2) < 1e-13


@pytest.mark.parametrize("key", ["cuboid", "sphere"])
@pytest.mark.parametrize("op", OPS)
def test_fixture_blocksparse_products(oracle, key, op):
    p = fixture_as_blocksparse(key)
    n = p["size"][0]
    rng = np.random.default_rng(99 + op)
    x = rand_vec(rng, n, np.complex128)
    y0 = rand_vec(rng, n, np.complex128)
    for alpha, beta, strong in ((1, 0, True), (1j, 2j, False)):
        got = oracle_mul(oracle, p, op, x, y0, alpha, beta, strong)
        assert relerr(got, scipy_mul(p, op, x, y0, alpha, beta, strong)) < 1e-13


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.complex64, np.complex128])
def test_vbcrs_oracle_all_dtypes(oracle, dtype):
    rng = np.random.default_rng(5)
    sizes = [3, 8, 1, 17, 5]
    starts = np.concatenate([[0], np.cumsum(sizes)])
    n = int(starts[-1])
    blocks, rs, cs = [], [], []
    for (i, j) in [(4, 0), (0, 3), (0, 1), (2, 2), (3, 4), (1, 1), (4, 4)]:
        b = rand_vec(rng, sizes[i] * sizes[j], dtype).reshape((sizes[i], sizes[j]), order="F")
        blocks.append(np.asfortranarray(b))
        rs.append(starts[i] + 1)
        cs.append(starts[j] + 1)
    p = dict(kind="vbcrs", blocks=blocks, rowstart=np.array(rs), colstart=np.array(cs), size=(n, n))
    x, y0 = rand_vec(rng, n, dtype), rand_vec(rng, n, dtype)
    tol = 1e-5 if np.dtype(dtype).itemsize in (4, 8) and np.dtype(dtype) in (np.float32, np.complex64) else 1e-13
    for op in OPS:
        for alpha, beta, strong in ((1, 0, True), (0.5, -2, False)):
            got = oracle_mul(oracle, p, op, x, y0, alpha, beta, strong)
            assert relerr(got, scipy_mul(p, op, x, y0, alpha, beta, strong)) < tol


# ---- colouring contract -----------------------------------------------------------------------
@pytest.mark.parametrize("key", ["cuboid", "sphere"])
def test_dsatur_valid_on_fixture(oracle, key):
    p = fixture_problem(key)
    for lists in (p["rowindices"], p["colindices"], p["diagonalindices"]):
        classes = oracle.color_dsatur(lists)
        assert oracle.color_check(lists, classes)
    # row lists are mutually disjoint -> one colour; column lists overlap up to 11x / 15x
    assert len(oracle.color_dsatur(p["rowindices"])) == 1
    assert len(oracle.color_dsatur(p["colindices"])) >= (11 if key == "cuboid" else 15)


def test_color_check_rejects_bad_colorings(oracle):
    lists = [[1, 2], [2, 3], [4]]
    assert oracle.color_check(lists, [[1, 3], [2]])
    assert not oracle.color_check(lists, [[1, 2], [3]])   # blocks 1, 2 share index 2
    assert not oracle.color_check(lists, [[1], [2]])      # block 3 missing
    assert not oracle.color_check(lists, [[1, 3], [2, 3]])  # block 3 twice
