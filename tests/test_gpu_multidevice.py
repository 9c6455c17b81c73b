"""GPU suite (-m gpu): ONE handle spread over the devices of a bsm_ctx_t (include/bsm_rocm.h,
csrc/bsm_dist.cpp) -- the multi-GPU fan-out behind the C ABI, where the reference has its `@tasks`
loop (src/vbcrs.jl:275-276, src/symmetricblockmatrix.jl:395-432).

The GPU box has one MI355X, so the context lists device 0 several times (virtual devices): every
part gets its own packed image, stream, work vector and receive buffer, and the halo / reduce-scatter
/ delivery copies run exactly as between distinct GPUs (hipMemcpyPeerAsync degenerates to a
device-to-device copy).  Every product is compared with the CPU oracle on the WHOLE operator.
"""
import numpy as np
import pytest

from _common import Cc, N, T, fixture_as_blocksparse, fixture_problem, oracle_mul, rand_vec, relerr

pytestmark = pytest.mark.gpu
TOL = {np.dtype(np.float64): 1e-12, np.dtype(np.complex128): 1e-12,
       np.dtype(np.float32): 1e-5, np.dtype(np.complex64): 1e-5}


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU suite needs a GPU"
    from bsm_amd import _lib as L
    L.lib()
    return torch


def wrap(bsm, A, op):
    return A if op == N else (bsm.transpose(A) if op == T else bsm.adjoint(A))


def check(torch, bsm, oracle, problem, A, ops=(N, T, Cc)):
    dt = np.dtype(A.dtype)
    nr, nc = problem["size"]
    rng = np.random.default_rng(5)
    for op in ops:
        if op == Cc and dt.kind != "c":
            continue
        xl, yl = (nc, nr) if op == N else (nr, nc)
        x, y0 = rand_vec(rng, xl, dt), rand_vec(rng, yl, dt)
        y0[::7] = np.nan  # the strong zero must not let them through
        combos = [(1, 0, True), (0.75, -1.5, False)]
        if dt.kind == "c":
            combos.append((1j, 2j, False))
        for alpha, beta, strong in combos:
            yin = y0 if strong else np.nan_to_num(y0, nan=0.25)
            ref = oracle_mul(oracle, problem, op, x, yin, alpha, beta, strong)
            # host vectors: the library moves x / y over PCIe itself
            yh = np.array(yin, copy=True)
            bsm.mul(yh, wrap(bsm, A, op), x, alpha, False if strong else beta)
            assert relerr(yh, ref) < TOL[dt], ("host", op, alpha, beta)
            # device vectors on cuda:0
            xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(np.array(yin, copy=True)).cuda()
            bsm.mul(yd, wrap(bsm, A, op), xd, alpha, False if strong else beta)
            torch.cuda.synchronize()
            assert relerr(yd.cpu().numpy(), ref) < TOL[dt], ("device", op, alpha, beta)


def check_parts(A, nparts):
    parts = A.parts()
    assert len(parts) == nparts
    nrows = A.size[0]
    covered = np.zeros(nrows, dtype=np.int32)
    for p in parts:
        lo, hi = p["own"]
        if hi >= lo:
            covered[lo - 1:hi] += 1
            assert p["touched"][0] <= lo and p["touched"][1] >= hi
    assert np.all(covered == 1), "own ranges must tile the rows"


@pytest.mark.parametrize("ndev", [2, 3])
def test_vbcrs_over_virtual_devices(torch_cuda, bsm, oracle, ndev):
    prob = bsm.synthetic.config2(n=9000, nblocks=420)
    A = bsm.synthetic.build(prob, devices=[0] * ndev)
    check_parts(A, ndev)
    assert sum(p["nblocks"] for p in A.parts()) == len(prob["blocks"])
    # bookkeeping of the WHOLE operator is what the handle reports (bit-exact, src/vbcrs.jl:84-117)
    perm, rowptr, colind, rowind = oracle.vbcrs_build(prob["rowstart"], prob["colstart"])
    assert np.array_equal(A.perm, perm) and np.array_equal(A.rowptr, rowptr)
    assert np.array_equal(A.colindices, colind) and np.array_equal(A.rowindices, rowind)
    check(torch_cuda, bsm, oracle, prob, A, ops=(N, T))


@pytest.mark.parametrize("ndev", [2, 4])
def test_symmetric_halo_over_virtual_devices(torch_cuda, bsm, oracle, ndev):
    prob = bsm.synthetic.config5(n=40_000, lo=16, hi=96, halfband=3)
    A = bsm.synthetic.build(prob, devices=[0] * ndev)
    check_parts(A, ndev)
    # banded structure: a part touches rows of the part below it (the halo) and nothing else
    parts = A.parts()
    assert any(p["touched"][0] < p["own"][0] for p in parts[1:])
    check(torch_cuda, bsm, oracle, prob, A, ops=(N, T))


@pytest.mark.parametrize("key", ["cuboid", "sphere"])
def test_reference_fixture_over_virtual_devices(torch_cuda, bsm, oracle, key):
    """The reference's own fixture (ComplexF64, scattered index lists): SymmetricBlockMatrix and, from
    its off-diagonal panels, a BlockSparseMatrix -- every part reaches rows all over the matrix."""
    prob = fixture_problem(key)
    A = bsm.synthetic.build(prob, devices=[0, 0, 0])
    check_parts(A, 3)
    check(torch_cuda, bsm, oracle, prob, A)
    pb = fixture_as_blocksparse(key)
    B = bsm.synthetic.build(pb, devices=[0, 0])
    check(torch_cuda, bsm, oracle, pb, B)


def test_blocksparse_config1_and_more_devices_than_block_rows(torch_cuda, bsm, oracle):
    prob = bsm.synthetic.config1()
    A = bsm.synthetic.build(prob, devices=[0, 0, 0])
    check(torch_cuda, bsm, oracle, prob, A, ops=(N, T))
    # 2 block rows on 4 devices: two parts stay empty (no image) and the product is still complete
    rng = np.random.default_rng(2)
    blocks = [np.asfortranarray(rng.standard_normal((5, 7))), np.asfortranarray(rng.standard_normal((6, 4)))]
    small = dict(kind="vbcrs", blocks=blocks, rowstart=np.array([3, 20]), colstart=np.array([1, 9]), size=(30, 16))
    V = bsm.synthetic.build(small, devices=[0, 0, 0, 0])
    assert sorted(p["nblocks"] for p in V.parts()) == [0, 0, 1, 1]
    check(torch_cuda, bsm, oracle, small, V, ops=(N, T))


def test_vbcrs_view_of_symmetric_and_multi_rhs(torch_cuda, bsm, oracle):
    torch = torch_cuda
    prob = bsm.synthetic.config3(nseg=40, bs=24, halfband=3)
    S = bsm.synthetic.build(prob)
    V = bsm.VariableBlockCompressedRowStorage(S, devices=[0, 0])
    check(torch, bsm, oracle, prob, V, ops=(N, T))
    # A * X through the multi-device handle (one fan-out per column)
    n = prob["size"][0]
    rng = np.random.default_rng(3)
    X = np.asfortranarray(rng.standard_normal((n, 3)))
    Y = np.asarray(V @ X)
    for k in range(3):
        ref = oracle_mul(oracle, prob, N, X[:, k].copy(), np.zeros(n))
        assert relerr(Y[:, k], ref) < 1e-12


def test_single_device_host_path_keeps_rows_outside_the_owned_range(torch_cuda, bsm, oracle):
    """host vectors + own=(lo, hi) + beta = false must hand back the caller's y outside the
    owned range (rows no block reaches are left untouched, include/bsm_rocm.h)."""
    prob = bsm.synthetic.config2(n=4000, nblocks=150)
    keep = [b for b, r in enumerate(prob["rowstart"]) if 1000 <= r < 2500]
    sub = dict(kind="vbcrs", blocks=[prob["blocks"][b] for b in keep], rowstart=prob["rowstart"][keep],
               colstart=prob["colstart"][keep], size=prob["size"])
    lo = int(min(sub["rowstart"]))
    hi = int(max(r + b.shape[0] - 1 for r, b in zip(sub["rowstart"], sub["blocks"])))
    A = bsm.synthetic.build(sub, own=(lo, hi))
    x = prob["x"]
    ref = oracle_mul(oracle, sub, N, x, np.zeros(4000))
    for first in (3.5, -7.25):  # twice with different sentinels: stale staging memory would show
        y = np.full(4000, first)
        bsm.mul(y, A, x)
        assert np.all(y[:lo - 1] == first) and np.all(y[hi:] == first)
        assert relerr(y[lo - 1:hi], ref[lo - 1:hi]) < 1e-12


def test_tensor_on_the_wrong_device_is_an_error_not_a_fault(torch_cuda, bsm):
    prob = bsm.synthetic.config2(n=2000, nblocks=40)
    A = bsm.synthetic.build(prob)
    assert A.device == torch_cuda.cuda.current_device()
    A.device = A.device + 1  # pretend the handle lives elsewhere
    x = torch_cuda.from_numpy(prob["x"]).cuda()
    with pytest.raises(ValueError, match="lives on cuda"):
        bsm.mul(torch_cuda.zeros_like(x), A, x)


def test_concurrent_products_on_one_multi_device_handle(torch_cuda, bsm, oracle):
    """Two host threads drive the SAME four-part handle (worker-thread fan-out) with different x / y:
    the handle serialises the products (its work vectors belong to it), every result must be exact."""
    import threading
    torch = torch_cuda
    prob = bsm.synthetic.config5(n=30_000, lo=16, hi=96, halfband=3)
    A = bsm.synthetic.build(prob, devices=[0, 0, 0, 0])
    n = prob["size"][0]
    rng = np.random.default_rng(9)
    xs = [rng.standard_normal(n) for _ in range(2)]
    refs = [oracle_mul(oracle, prob, N, x, np.zeros(n)) for x in xs]
    errs = []

    def worker(k):
        try:
            xd = torch.from_numpy(xs[k]).cuda()
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for it in range(60):
                    if it % 2:
                        yd = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
                        bsm.mul(yd, A, xd)
                        s.synchronize()
                        got = yd.cpu().numpy()
                    else:
                        got = np.full(n, np.nan)
                        bsm.mul(got, A, xs[k])
                    e = relerr(got, refs[k])
                    if not e < 1e-12:
                        errs.append((k, it, e))
        except Exception as ex:  # pragma: no cover
            errs.append((k, repr(ex)))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errs, errs[:3]


@pytest.mark.parametrize("K", [11, 20])
@pytest.mark.parametrize("kind", ["symmetric", "vbcrs", "fixture"])
def test_multi_rhs_through_the_multi_device_handle(torch_cuda, bsm, oracle, kind, K):
    """mul!(Y, A, X, a, b) with matrices on a handle spread over three parts: every device streams its
    part once per batch of <= 8 columns, the halo and the delivery carry the batch's columns; 11 columns =
    a batch of 8 + one of 3; padded leading dimensions; host and device memory; each column against the
    oracle's single-vector product."""
    torch = torch_cuda
    if kind == "symmetric":
        prob = bsm.synthetic.config5(n=20_000, lo=16, hi=96, halfband=3)
    elif kind == "vbcrs":
        prob = bsm.synthetic.config2(n=9000, nblocks=400)
    else:
        prob = fixture_problem("cuboid")
    A = bsm.synthetic.build(prob, devices=[0, 0, 0])
    dt = np.dtype(A.dtype)
    n = prob["size"][0]
    rng = np.random.default_rng(21)
    ld = n + 5  # (K = 20 on a real operator: a fan-out of 16 columns -- the parts' matrix-pipe passes -- and one of 4)
    Xp = np.zeros((ld, K), dtype=dt, order="F")
    Yp = np.zeros((ld, K), dtype=dt, order="F")
    Xp[:n] = rng.standard_normal((n, K)) + (1j * rng.standard_normal((n, K)) if dt.kind == "c" else 0)
    Y0 = rng.standard_normal((n, K)).astype(dt)
    for op in (N, T):
        Aop = wrap(bsm, A, op)
        for alpha, beta in ((True, False), (0.5, -2.0)):
            refs = [oracle_mul(oracle, prob, op, np.ascontiguousarray(Xp[:n, k]), Y0[:, k].copy(),
                               1 if alpha is True else alpha, 0 if beta is False else beta, beta is False)
                    for k in range(K)]
            # device memory, padded leading dimension (views into a taller column-major matrix)
            Xd = torch.from_numpy(np.ascontiguousarray(Xp.T)).cuda().t()[:n]
            Yfull = torch.from_numpy(np.ascontiguousarray(Yp.T)).cuda().t()
            Yfull[:n] = torch.from_numpy(Y0).cuda()
            Yd = Yfull[:n]
            bsm.mul(Yd, Aop, Xd, alpha, beta)
            torch.cuda.synchronize()
            got = Yd.cpu().numpy()
            for k in range(K):
                assert relerr(got[:, k], refs[k]) < 1e-12, ("device", op, alpha, k)
            assert not bool(Yfull[n:].abs().max() > 0)  # the padding rows are not touched
            # host memory
            Yh = np.asfortranarray(Y0.copy())
            bsm.mul(Yh, Aop, np.asfortranarray(Xp[:n]), alpha, beta)
            for k in range(K):
                assert relerr(Yh[:, k], refs[k]) < 1e-12, ("host", op, alpha, k)


def test_single_product_then_a_wider_batch_without_a_sync_in_between(torch_cuda, bsm, oracle):
    """A K = 1 product of a fresh multi-device handle, then -- NO synchronisation -- an 8-column product on the same
    handle: the second call re-allocates the parts' work vectors for 8 columns while the fused finish kernels of the
    first may still be reading them on the CALLERS' streams (grow_buffers must reach every part's ev_done
    before the first free).  Both results against the oracle."""
    torch = torch_cuda
    prob = bsm.synthetic.config5(n=60_000, lo=16, hi=128, halfband=3)
    n = prob["size"][0]
    rng = np.random.default_rng(33)
    x1 = rng.standard_normal(n)
    X8 = np.asfortranarray(rng.standard_normal((n, 8)))
    ref1 = oracle_mul(oracle, prob, N, x1, np.zeros(n))
    ref8 = [oracle_mul(oracle, prob, N, np.ascontiguousarray(X8[:, k]), np.zeros(n)) for k in range(8)]
    for rep in range(3):  # a fresh handle every time: the buffers start at one column
        A = bsm.synthetic.build(prob, devices=[0, 0, 0])
        xd = torch.from_numpy(x1).cuda()
        yd = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        Xd = torch.from_numpy(np.ascontiguousarray(X8.T)).cuda().t()
        Yd = torch.full((8, n), float("nan"), dtype=torch.float64, device="cuda").t()
        torch.cuda.synchronize()
        bsm.mul(yd, A, xd)
        bsm.mul(Yd, A, Xd)  # grows the buffers; nothing has been waited for
        torch.cuda.synchronize()
        assert relerr(yd.cpu().numpy(), ref1) < 1e-12, rep
        got = Yd.cpu().numpy()
        for k in range(8):
            assert relerr(got[:, k], ref8[k]) < 1e-12, (rep, k)
        del A


@pytest.mark.parametrize("mode,own_streams,rezero", [("1", "1", "1"), ("0", "1", "1"), ("1", "0", "1"), ("1", "1", "0"), ("0", "0", "0")])
def test_flag_and_event_ordering_of_the_fan_out(torch_cuda, bsm, oracle, monkeypatch, mode, own_streams, rezero):
    """The cross-stream ordering of a multi-device product by stream memory operations (hipStreamWriteValue64 /
    hipStreamWaitValue64 on sequence counters, BSM_DIST_FLAGS=1: the default on virtual devices) and by events (=0),
    with the parts on streams of their own (BSM_DIST_ONE_STREAM=0: what distinct devices do) and all on the caller's
    stream (the default for parts that share the caller's device: no ordering packets at all), work vectors kept zero
    by the finish kernels (BSM_DIST_REZERO=1) or cleared in front of every product (=0):
    chained device products with no synchronisation in between, a host-vector product (copy path: events) in the middle
    -- the switch between the forms drains the streams --, partitioned vectors, a second caller stream; everything
    against the oracle."""
    torch = torch_cuda
    monkeypatch.setenv("BSM_DIST_FLAGS", mode)
    monkeypatch.setenv("BSM_DIST_ONE_STREAM", "0" if own_streams == "1" else "1")
    monkeypatch.setenv("BSM_DIST_REZERO", rezero)
    prob = bsm.synthetic.config5(n=40_000, lo=16, hi=96, halfband=3)
    n = prob["size"][0]
    A = bsm.synthetic.build(prob, devices=[0, 0, 0])
    rng = np.random.default_rng(8)
    xs = [rng.standard_normal(n) for _ in range(4)]
    refs = [oracle_mul(oracle, prob, N, x, np.zeros(n)) for x in xs]
    xd = [torch.from_numpy(x).cuda() for x in xs]
    yd = [torch.full((n,), float("nan"), dtype=torch.float64, device="cuda") for _ in xs]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    bsm.mul(yd[0], A, xd[0])
    bsm.mul(yd[1], A, xd[1])                      # back to back on the same handle
    yh = np.zeros(n)
    bsm.mul(yh, A, xs[2])                         # host vectors: the copy path (events)
    with torch.cuda.stream(side):                 # another caller stream
        side.wait_stream(torch.cuda.current_stream())
        bsm.mul(yd[3], A, xd[3])
    torch.cuda.current_stream().wait_stream(side)
    # y of one product as x of the next (alpha, beta): z = A (A x0) - 2 x1
    z = xd[1].clone()
    bsm.mul(z, A, yd[0], 1.0, -2.0)
    torch.cuda.synchronize()
    for k in (0, 1, 3):
        assert relerr(yd[k].cpu().numpy(), refs[k]) < 1e-12, (mode, k)
    assert relerr(yh, refs[2]) < 1e-12
    assert relerr(z.cpu().numpy(), oracle_mul(oracle, prob, N, refs[0], xs[1].copy(), 1.0, -2.0, False)) < 1e-12
    parts = A.parts()
    xp = [xd[0][p["cols"][0] - 1:p["cols"][1]].clone() for p in parts]
    yp = [torch.full((max(p["own"][1] - p["own"][0] + 1, 0),), float("nan"), dtype=torch.float64, device="cuda") for p in parts]
    bsm.mul_parts(yp, A, xp)
    bsm.mul(yd[1], A, xd[0])                      # full vectors right behind partitioned ones
    torch.cuda.synchronize()
    assert relerr(torch.cat(yp).cpu().numpy(), refs[0]) < 1e-12
    assert relerr(yd[1].cpu().numpy(), refs[0]) < 1e-12


@pytest.mark.parametrize("mode", ["1", "0"])
def test_a_fan_out_that_fails_half_way_leaves_a_usable_handle(torch_cuda, bsm, oracle, monkeypatch, mode):
    """A fused fan-out that fails between its two phases (injected: bsm_debug_dist_fail_after) has consumed a sequence
    number whose write packets were never issued.  The library releases every counter from the host, drains the devices
    and forgets the ordering form (csrc/bsm_dist.cpp: dist_mul_fused): the failing call reports an error, and the
    products after it neither hang on `flag >= seq` nor accumulate into a half-written work vector."""
    import ctypes as C
    from bsm_amd import _lib as L
    torch = torch_cuda
    monkeypatch.setenv("BSM_DIST_FLAGS", mode)
    monkeypatch.setenv("BSM_DIST_ONE_STREAM", "0")  # parts on streams of their own: ordering packets between them
    prob = bsm.synthetic.config5(n=30_000, lo=16, hi=96, halfband=3)
    n = prob["size"][0]
    A = bsm.synthetic.build(prob, devices=[0, 0, 0])
    rng = np.random.default_rng(11)
    x = rng.standard_normal(n)
    ref = oracle_mul(oracle, prob, N, x, np.zeros(n))
    xd = torch.from_numpy(x).cuda()
    yd = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    bsm.mul(yd, A, xd)
    bsm.mul(yd, A, xd)
    hook = L.lib().bsm_debug_dist_fail_after
    hook.argtypes, hook.restype = [C.c_int], None
    hook(0)  # the next fused fan-out fails after its products have been issued
    with pytest.raises(Exception):
        bsm.mul(yd, A, xd)
    for _ in range(3):  # must neither hang nor be wrong
        yd.fill_(float("nan"))
        bsm.mul(yd, A, xd)
    torch.cuda.synchronize()
    assert relerr(yd.cpu().numpy(), ref) < 1e-12
    z = torch.from_numpy(x.copy()).cuda()
    bsm.mul(z, A, yd, 0.5, -2.0)
    torch.cuda.synchronize()
    assert relerr(z.cpu().numpy(), oracle_mul(oracle, prob, N, ref, x.copy(), 0.5, -2.0, False)) < 1e-12


@pytest.mark.parametrize("own_streams", ["0", "1"])
def test_work_vectors_stay_zero_across_directions_batches_and_paths(torch_cuda, bsm, oracle, monkeypatch, own_streams):
    """The finish kernels write zeros behind what they read, so a product accumulates into its part's work vector
    without a `w = 0` launch in front (csrc/bsm_dist.cpp: DistState::w_clean).  The invariant has to survive
    everything that touches those vectors: the two directions of a VBCRS operator (different segments travel), a
    multi-RHS batch that re-allocates them, a copy-path product (host vectors) that leaves partial sums in them,
    NaN in the incoming y, numeric beta -- chained without synchronisation, each against the oracle."""
    torch = torch_cuda
    monkeypatch.setenv("BSM_DIST_ONE_STREAM", "0" if own_streams == "1" else "1")
    for prob in (bsm.synthetic.config2(n=12_000, nblocks=500), bsm.synthetic.config5(n=30_000, lo=16, hi=96, halfband=3)):
        n = prob["size"][0]
        A = bsm.synthetic.build(prob, devices=[0, 0, 0])
        rng = np.random.default_rng(31)
        x = rng.standard_normal(n)
        xd = torch.from_numpy(x).cuda()
        got, want = [], []

        def dev(op, alpha=1.0, beta=0.0, strong=True):
            y0 = rng.standard_normal(n)
            if strong:
                y0[::5] = np.nan
            yd = torch.from_numpy(y0.copy()).cuda()
            bsm.mul(yd, wrap(bsm, A, op), xd, alpha, False if strong else beta)
            got.append(yd)
            want.append(oracle_mul(oracle, prob, op, x, y0, alpha, beta, strong))

        dev(N)
        dev(T)                                    # the other plan: other segments of the work vectors travel
        dev(T, 0.5, -2.0, False)
        X = rng.standard_normal((n, 5))           # a batch: the work vectors are re-allocated (5 columns)
        Xd = torch.from_numpy(np.ascontiguousarray(X.T)).cuda().t()
        Yd = torch.full((5, n), float("nan"), dtype=torch.float64, device="cuda").t()
        bsm.mul(Yd, wrap(bsm, A, T), Xd)
        dev(N, 2.0, 1.0, False)
        yh = np.zeros(n)
        bsm.mul(yh, wrap(bsm, A, T), x)           # host vectors: the copy path leaves its sums in the work vectors
        dev(T)
        dev(N)
        torch.cuda.synchronize()
        for k, (g, w) in enumerate(zip(got, want)):
            assert relerr(g.cpu().numpy(), w) < 1e-12, (own_streams, prob["kind"], k)
        Yg = Yd.cpu().numpy()
        for k in range(5):
            assert relerr(Yg[:, k], oracle_mul(oracle, prob, T, np.ascontiguousarray(X[:, k]), np.zeros(n))) < 1e-12, k
        assert relerr(yh, oracle_mul(oracle, prob, T, x, np.zeros(n))) < 1e-12
        del A


# ---- partitioned vectors behind the C ABI: bsm_mul_parts ------------------------------------------------------------
def _scatter(torch, v, ranges):
    """the parts of a full host vector (1-based inclusive ranges) as CUDA tensors"""
    return [torch.from_numpy(np.ascontiguousarray(v[lo - 1:hi])).cuda() if hi >= lo else None for lo, hi in ranges]


@pytest.mark.parametrize("kind,ndev", [("vbcrs", 3), ("vbcrs_rect", 2), ("symmetric", 2), ("symmetric", 4), ("fixture", 3),
                                       ("blocksparse", 3)])
def test_partitioned_vectors_through_the_multi_device_handle(torch_cuda, bsm, oracle, kind, ndev):
    """bsm_mul_parts: every part holds only ITS x entries and receives only ITS y entries (reference: block rows own
    disjoint y ranges, src/vbcrs.jl:275-283); halos and partial-y segments move as fused peer-reading kernels.
    Against the oracle on the whole operator, every op, strong zero and numeric alpha / beta, twice (reused buffers)."""
    torch = torch_cuda
    if kind == "vbcrs":
        prob = bsm.synthetic.config2(n=9000, nblocks=420)
    elif kind == "vbcrs_rect":  # non-square: the column partition is its own (equal chunks)
        rng = np.random.default_rng(4)
        blocks, rs, cs = [], [], []
        for r0 in range(1, 600, 40):
            for c0 in rng.choice(np.arange(1, 900, 30), size=3, replace=False):
                blocks.append(np.asfortranarray(rng.standard_normal((int(rng.integers(8, 40)), int(rng.integers(5, 30))))))
                rs.append(r0)
                cs.append(int(c0))
        prob = dict(kind="vbcrs", blocks=blocks, rowstart=np.array(rs), colstart=np.array(cs), size=(640, 930))
    elif kind == "symmetric":
        prob = bsm.synthetic.config5(n=40_000, lo=16, hi=96, halfband=3)
    elif kind == "fixture":
        prob = fixture_problem("cuboid")
    else:
        prob = bsm.synthetic.config1(n=3000, nblocks=120, bs=24)
    A = bsm.synthetic.build(prob, devices=[0] * ndev)
    dt = np.dtype(A.dtype)
    nr, nc = prob["size"]
    parts = A.parts()
    rows, cols = [p["own"] for p in parts], [p["cols"] for p in parts]
    for rng_ in (rows, cols):  # both partitions tile their axis
        cov = np.zeros(max(nr, nc) + 1, dtype=int)
        for lo, hi in rng_:
            if hi >= lo:
                cov[lo:hi + 1] += 1
        assert set(cov[1:(nr if rng_ is rows else nc) + 1]) == {1}
    if nr == nc:
        assert rows == cols  # square: y parts of one product are x parts of the next
    rng = np.random.default_rng(5)
    ops = (N, T, Cc) if dt.kind == "c" else (N, T)
    for op in ops:
        xl, yl = (nc, nr) if op == N else (nr, nc)
        xr, yr = (cols, rows) if op == N else (rows, cols)
        x, y0 = rand_vec(rng, xl, dt), rand_vec(rng, yl, dt)
        y0[::7] = np.nan
        combos = [(1, 0, True), (0.75, -1.5, False)] + ([(1j, 2j, False)] if dt.kind == "c" else [])
        for alpha, beta, strong in combos:
            yin = y0 if strong else np.nan_to_num(y0, nan=0.25)
            ref = oracle_mul(oracle, prob, op, x, yin, alpha, beta, strong)
            for _ in range(2):
                xp, yp = _scatter(torch, x, xr), _scatter(torch, yin, yr)
                bsm.mul_parts(yp, wrap(bsm, A, op), xp, alpha, False if strong else beta)
                torch.cuda.synchronize()
                got = np.full(yl, np.nan, dtype=dt)
                for (lo, hi), t in zip(yr, yp):
                    if hi >= lo:
                        got[lo - 1:hi] = t.cpu().numpy()
                assert relerr(got, ref) < TOL[dt], (kind, op, alpha, beta)
    # chained: y parts of one product are the x parts of the next (square operators), no host round trip
    if nr == nc and dt.kind != "c":
        x = rand_vec(rng, nc, dt)
        xp = _scatter(torch, x, cols)
        y1 = [torch.empty_like(t) if t is not None else None for t in xp]
        y2 = [torch.empty_like(t) if t is not None else None for t in xp]
        bsm.mul_parts(y1, A, xp)
        bsm.mul_parts(y2, A, y1)
        torch.cuda.synchronize()
        ref = oracle_mul(oracle, prob, N, oracle_mul(oracle, prob, N, x, np.zeros(nr)), np.zeros(nr))
        got = np.concatenate([t.cpu().numpy() for t in y2 if t is not None])
        assert relerr(got, ref) < 1e-11


def test_copy_path_of_the_multi_device_handle_still_works(torch_cuda, bsm, oracle, monkeypatch):
    """Devices without peer access take the copy path (hipMemcpyPeerAsync + adds): forced here with BSM_DIST_COPIES,
    every op, host and device vectors, numeric beta (the late combine), products back to back on two streams."""
    monkeypatch.setenv("BSM_DIST_COPIES", "1")
    prob = bsm.synthetic.config5(n=30_000, lo=16, hi=96, halfband=3)
    A = bsm.synthetic.build(prob, devices=[0, 0, 0])
    check(torch_cuda, bsm, oracle, prob, A, ops=(N, T))
    xp = _scatter(torch_cuda, prob["x"], [p["cols"] for p in A.parts()])
    with pytest.raises(RuntimeError, match="peer access"):  # the partitioned-vector entry has no copy path
        bsm.mul_parts([torch_cuda.empty_like(t) for t in xp], A, xp)


def test_c_abi_null_scalars_and_leading_dimensions_on_a_multi_device_handle(torch_cuda, bsm, oracle, monkeypatch):
    """alpha = NULL means 1 and beta = NULL means 0 on EVERY path of a multi-device handle (the host
    numeric-beta path dereferenced beta); bsm_mul_multi rejects a leading dimension below the vector length."""
    import ctypes as C
    from bsm_amd import _lib as L
    torch = torch_cuda
    prob = bsm.synthetic.config5(n=20_000, lo=16, hi=96, halfband=3)
    n = prob["size"][0]
    x = prob["x"].copy()
    ref = oracle_mul(oracle, prob, N, x, np.zeros(n), 1, 0, False)
    for copies in ("0", "1"):
        monkeypatch.setenv("BSM_DIST_COPIES", copies)
        A = bsm.synthetic.build(prob, devices=[0, 0, 0])
        lib = L.lib()
        # host vectors, numeric beta (beta_strong_zero = 0) with beta = NULL: y = A x + 0 * y
        y = np.full(n, 3.0)
        L.check(lib.bsm_mul(A._h.ptr, N, x.ctypes.data, y.ctypes.data, None, None, 0, L.BSM_MEM_HOST, None))
        assert relerr(y, ref) < 1e-12
        xd, yd = torch.from_numpy(x).cuda(), torch.full((n,), 3.0, dtype=torch.float64, device="cuda")
        L.check(lib.bsm_mul(A._h.ptr, N, xd.data_ptr(), yd.data_ptr(), None, None, 0, L.BSM_MEM_DEVICE, None))
        torch.cuda.synchronize()
        assert relerr(yd.cpu().numpy(), ref) < 1e-12
        X = np.asfortranarray(np.stack([x, 2 * x], axis=1))
        Y = np.zeros((n, 2), order="F")
        rc = lib.bsm_mul_multi(A._h.ptr, N, 2, X.ctypes.data, n - 1, Y.ctypes.data, n, None, None, 1, L.BSM_MEM_HOST, None)
        assert rc == -1 and b"leading dimension" in lib.bsm_last_error()
        del A


def test_unequal_parts_driven_from_two_streams(torch_cuda, bsm, oracle):
    """products of one handle issued on two streams in turn, device vectors, parts of deliberately
    unequal work (one part holds almost everything): a product must not overwrite a work vector a peer is still
    reading.  Every result against the oracle."""
    torch = torch_cuda
    rng = np.random.default_rng(12)
    # a symmetric operator whose first rows carry nearly all the bytes: partition by stored entries puts a few
    # wide blocks in part 0 and hundreds of small ones in the others
    n = 24_000
    sz = np.concatenate([np.full(12, 250), np.full(300, 70)])
    start = np.concatenate([[0], np.cumsum(sz)[:-1]])
    diag, didx, off, ridx, cidx = [], [], [], [], []
    for s0, m in zip(start, sz):
        d = rng.standard_normal((m, m))
        diag.append(np.asfortranarray((d + d.T) / 2))
        didx.append(np.arange(s0 + 1, s0 + m + 1))
    for i in range(1, len(sz)):
        for k in (1, 2):
            if i - k >= 0:
                off.append(np.asfortranarray(rng.standard_normal((sz[i], sz[i - k]))))
                ridx.append(didx[i])
                cidx.append(didx[i - k])
    prob = dict(kind="symmetric", diagonals=diag, diagonalindices=didx, offdiagonals=off, rowindices=ridx,
                colindices=cidx, size=(n, n))
    A = bsm.synthetic.build(prob, devices=[0, 0, 0, 0])
    xs = [rng.standard_normal(n) for _ in range(2)]
    refs = [oracle_mul(oracle, prob, N, x, np.zeros(n), 0.5, 0, True) for x in xs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    xd = [torch.from_numpy(x).cuda() for x in xs]
    outs = [[], []]
    for it in range(40):
        k = it % 2
        with torch.cuda.stream(streams[k]):
            yd = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
            bsm.mul(yd, A, xd[k], 0.5, False)
            outs[k].append(yd)
    torch.cuda.synchronize()
    for k in range(2):
        for yd in outs[k]:
            assert relerr(yd.cpu().numpy(), refs[k]) < 1e-12


def test_multi_device_products_refuse_graph_capture(torch_cuda, bsm):
    """a product of a multi-device handle issues on several streams and waits for events of earlier
    products -- it must not be recorded into a graph; bsm_mul says so instead of corrupting the capture."""
    torch = torch_cuda
    prob = bsm.synthetic.config2(n=6000, nblocks=200)
    A = bsm.synthetic.build(prob, devices=[0, 0])
    x = torch.from_numpy(prob["x"]).cuda()
    y = torch.zeros_like(x)
    plan = bsm.MulPlan(y, A, x)
    plan()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    raised = False
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=s):
                try:
                    plan()
                except RuntimeError as e:
                    raised = "cannot be captured" in str(e)
        except Exception:
            pass  # an empty capture may fail to instantiate on some runtimes: irrelevant here
    torch.cuda.synchronize()
    assert raised
    plan()  # the handle is still usable
    torch.cuda.synchronize()


def test_partitioned_vectors_fp32_and_device_resident_blocks(torch_cuda, bsm, oracle):
    """bsm_mul_parts with fp32 elements and with blocks that already live in HBM (repacked by the pack kernel on every
    part's device): C4-shaped VBCRS over three parts, op N and T, against the oracle."""
    torch = torch_cuda
    prob = bsm.synthetic.config4(ngrid=24, bs=64, per_row=5)
    dev = dict(prob)
    dev["blocks"] = [torch.from_numpy(np.ascontiguousarray(b.T)).cuda().t() for b in prob["blocks"]]
    n = prob["size"][0]
    x = prob["x"]
    for p in (prob, dev):
        A = bsm.synthetic.build(p, devices=[0, 0, 0])
        parts = A.parts()
        rows, cols = [q["own"] for q in parts], [q["cols"] for q in parts]
        for op in (N, T):
            ref = oracle_mul(oracle, prob, op, x, np.zeros(n, dtype=np.float32))
            xp, yp = _scatter(torch, x, cols if op == N else rows), [torch.full((hi - lo + 1,), float("nan"), dtype=torch.float32, device="cuda")
                                                                      for lo, hi in (rows if op == N else cols)]
            bsm.mul_parts(yp, wrap(bsm, A, op), xp)
            torch.cuda.synchronize()
            got = np.concatenate([t.cpu().numpy() for t in yp])
            assert relerr(got, ref) < 1e-5, op
