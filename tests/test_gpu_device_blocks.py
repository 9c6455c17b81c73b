"""GPU suite (-m gpu): operators whose blocks are DEVICE-resident at construction
(bsm_options.blocks_memspace = BSM_MEM_DEVICE; SURVEY.md 8f2 "on-device construction"): the strip
layout is written by pack_kernel, no matrix byte crosses PCIe.  Also the in-HBM generator of the
synthetic configs (include/bsm_synth.h), which must reproduce the numpy streams bit for bit."""
import numpy as np
import pytest

from _common import N, T, oracle_mul, rand_vec, relerr, sampled_relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU suite needs a GPU"
    from bsm_amd import _lib as L
    L.lib()
    return torch


def to_host(prob):
    """host copy (numpy, column-major) of a problem generated in HBM"""
    out = dict(prob)
    for key in ("blocks", "diagonals", "offdiagonals"):
        if key in prob:
            out[key] = [np.asfortranarray(b.cpu().numpy()) for b in prob[key]]
    out["x"] = prob["x"].cpu().numpy()
    return out


def dev_copy(torch, b):
    """column-major CUDA copy of a numpy block"""
    t = torch.from_numpy(np.ascontiguousarray(b.T)).cuda()  # (n, m) row-major == (m, n) column-major
    return t.t()


def test_generator_in_hbm_is_bit_identical_to_the_numpy_streams(torch_cuda, bsm):
    S = bsm.synthetic
    for dt in (np.float64, np.float32):
        ids, ms, ns = [0, 7, 12345, 99], [5, 64, 17, 33], [9, 64, 17, 2]
        sym = [0, 1, 1, 0]
        got = S.device_blocks(0xB5A5, ids, ms, ns, dt, sym)
        for b, m, n, s, g in zip(ids, ms, ns, sym, got):
            ref = S.block_values(0xB5A5, b, m, n, dt)
            if s:
                ref = np.asfortranarray((ref + ref.T) / 2)
            assert g.shape == (m, n) and np.array_equal(g.cpu().numpy(), ref)
        assert np.array_equal(S.device_vector(0xB5A2, 1000, dt).cpu().numpy(), S.vector(0xB5A2, 1000, dt))
        assert np.array_equal(S.device_vector(0xB5A2, 100, dt, first=900).cpu().numpy(), S.vector(0xB5A2, 1000, dt)[900:])
    # whole configs: same structure, same values
    for host, dev in ((S.config2(n=3000, nblocks=80), S.config2(n=3000, nblocks=80, on_device=True)),
                      (S.config5(n=4000, lo=16, hi=80, halfband=3), S.config5(n=4000, lo=16, hi=80, halfband=3, on_device=True)),
                      (S.config3(nseg=12, bs=16, halfband=3), S.config3(nseg=12, bs=16, halfband=3, on_device=True)),
                      (S.config4(ngrid=12, bs=16, per_row=4), S.config4(ngrid=12, bs=16, per_row=4, on_device=True))):
        h = to_host(dev)
        for key in ("blocks", "diagonals", "offdiagonals"):
            if key in host:
                assert len(host[key]) == len(h[key])
                assert all(np.array_equal(a, b) for a, b in zip(host[key], h[key])), key
        for key in ("rowstart", "colstart"):
            if key in host:
                assert np.array_equal(host[key], dev[key])
        for key in ("rowindices", "colindices", "diagonalindices"):
            if key in host:
                assert all(np.array_equal(a, b) for a, b in zip(host[key], dev[key]))
        assert np.array_equal(host["x"], h["x"])


def test_vbcrs_from_device_blocks_equals_host_construction(torch_cuda, bsm, oracle):
    torch = torch_cuda
    S = bsm.synthetic
    for dt, tol in ((np.float64, 1e-12), (np.float32, 1e-5)):
        dev = S.config2(n=12_000, nblocks=500, dtype=dt, on_device=True)
        host = to_host(dev)
        Ad = S.build(dev, transpose_image=True)
        Ah = S.build(host, transpose_image=True)
        assert Ad.stats() == Ah.stats()
        assert np.array_equal(Ad.perm, Ah.perm) and np.array_equal(Ad.rowptr, Ah.rowptr)
        x = dev["x"]
        for op in (N, T):
            Aop_d = Ad if op == N else bsm.transpose(Ad)
            Aop_h = Ah if op == N else bsm.transpose(Ah)
            yd, yh = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
            bsm.mul(yd, Aop_d, x)
            bsm.mul(yh, Aop_h, x)
            torch.cuda.synchronize()
            assert torch.equal(yd, yh), "same packed image => bitwise the same product"
            ref = oracle_mul(oracle, host, op, host["x"], np.zeros(len(host["x"]), dtype=dt))
            assert relerr(yd.cpu().numpy(), ref) < tol


def test_symmetric_and_indexed_blocksparse_from_device_blocks(torch_cuda, bsm, oracle):
    torch = torch_cuda
    S = bsm.synthetic
    dev = S.config5(n=30_000, lo=16, hi=200, halfband=3, on_device=True)
    host = to_host(dev)
    A = S.build(dev)
    x = dev["x"]
    y = torch.zeros_like(x)
    for op in (N, T):
        bsm.mul(y, A if op == N else bsm.transpose(A), x, 0.5, False)
        ref = oracle_mul(oracle, host, op, host["x"], np.zeros(len(host["x"])), 0.5, 0, True)
        assert relerr(y.cpu().numpy(), ref) < 1e-12
    assert bsm.nnz(A) == sum(b.size for b in host["diagonals"]) + 2 * sum(b.size for b in host["offdiagonals"])
    # VBCRS view of the device-resident symmetric operator
    V = bsm.VariableBlockCompressedRowStorage(A)
    bsm.mul(y, V, x)
    ref = oracle_mul(oracle, host, N, host["x"], np.zeros(len(host["x"])))
    assert relerr(y.cpu().numpy(), ref) < 1e-12
    # scattered, unsorted index lists (the packer's permuted placement) with blocks taller than 64 rows
    rng = np.random.default_rng(11)
    n = 900
    blocks, rows, cols = [], [], []
    for m, k in ((70, 33), (5, 130), (64, 64), (130, 7)):
        blocks.append(np.asfortranarray(rng.standard_normal((m, k))))
        rows.append(rng.permutation(n)[:m] + 1)
        cols.append(rng.permutation(n)[:k] + 1)
    ph = dict(kind="blocksparse", blocks=blocks, rowindices=rows, colindices=cols, size=(n, n))
    pd = dict(ph, blocks=[dev_copy(torch, b) for b in blocks])
    B = S.build(pd, transpose_image=True)
    xh = rng.standard_normal(n)
    for op in (N, T):
        yh = np.zeros(n)
        bsm.mul(yh, B if op == N else bsm.transpose(B), xh)
        assert relerr(yh, oracle_mul(oracle, ph, op, xh, np.zeros(n))) < 1e-12
    assert np.allclose(bsm.sparse(B).toarray(), bsm.sparse(S.build(ph)).toarray())


def test_device_blocks_spread_over_a_context(torch_cuda, bsm, oracle):
    S = bsm.synthetic
    dev = S.config5(n=30_000, lo=16, hi=120, halfband=3, on_device=True)
    host = to_host(dev)
    A = S.build(dev, devices=[0, 0, 0])
    x = dev["x"]
    y = torch_cuda.zeros_like(x)
    bsm.mul(y, A, x)
    assert relerr(y.cpu().numpy(), oracle_mul(oracle, host, N, host["x"], np.zeros(len(host["x"])))) < 1e-12


def test_device_blocks_need_a_device_handle(torch_cuda, bsm):
    dev = bsm.synthetic.config2(n=2000, nblocks=30, on_device=True)
    with pytest.raises(bsm._lib.BsmError, match="device"):
        bsm.synthetic.build(dev, device=-2)


def test_rowcolvals_from_the_device_image(torch_cuda, bsm):
    """bsm_rowcolvals / sparse_device: the COO triples and the CSR matrix assembled on the GPU equal
    the mirror's host `sparse(A)` (reference src/sparse.jl) for all three types, wrappers included."""
    import scipy.sparse as sp
    from _common import fixture_problem, fixture_as_blocksparse
    S = bsm.synthetic
    cases = [S.build(S.config2(n=5000, nblocks=200)),
             S.build(S.config2(n=3000, nblocks=100, dtype=np.float32)),
             S.build(fixture_problem("cuboid")),                      # ComplexF64, scattered lists, diag + off
             S.build(fixture_as_blocksparse("sphere")),
             S.build(S.config5(n=6000, lo=16, hi=100, halfband=3)),
             S.build(S.config5(n=6000, lo=16, hi=100, halfband=3), devices=[0, 0]),
             bsm.VariableBlockCompressedRowStorage(S.build(S.config3(nseg=20, bs=24, halfband=3)))]
    for A in cases:
        ref = bsm.sparse(A) if not (isinstance(A, bsm.VariableBlockCompressedRowStorage) and A.perm.size != len(A.blocks)) else None
        r, c, v = bsm.rowcolvals_device(A, device=False)
        assert len(r) == bsm.nnz(A)
        got = sp.coo_matrix((v, (r - 1, c - 1)), shape=bsm.size(A)).tocsc()
        if ref is None:  # VBCRS view of a symmetric operator: compare with the symmetric matrix itself
            continue
        assert abs(got - ref).max() <= 1e-6 * abs(ref).max() if A.dtype == np.float32 else abs(got - ref).max() == 0
        if A.devices is None:
            csr = bsm.sparse_device(A)
            dense = csr.to_dense().cpu().numpy()
            assert np.allclose(dense, ref.toarray(), rtol=1e-6 if A.dtype == np.float32 else 1e-14, atol=0)
    # wrappers: rows / cols swap, adjoint conjugates
    A = cases[2]
    r, c, v = bsm.rowcolvals_device(bsm.adjoint(A), device=False)
    got = sp.coo_matrix((v, (r - 1, c - 1)), shape=bsm.size(A)).tocsc()
    assert abs(got - bsm.sparse(A).conj().T).max() == 0


def test_config5_FULL_size_against_the_oracle_on_sampled_rows(torch_cuda, bsm, oracle):
    """BASELINE.json configs[4] at its FULL size against the ORACLE: the operator is generated in HBM bit-identically to
    the numpy streams (test_generator_in_hbm_is_bit_identical_to_the_numpy_streams), so the host regenerates only the
    blocks that reach ~200 sampled diagonal segments (synthetic.config5_sample: diagonal block, forward blocks (I, J),
    and the blocks (I', I) whose transposes reach rows I), runs orc_sym_mul on that sub-problem with the full x, and the
    GPU's y is compared on those rows -- 3-argument form and mul!(y, A, x, alpha, beta); S^T through the same check.
    Reference procedure: test/test_vbcrs.jl:33-47 (product against an independent one, max |dy| / max |y|), 1e-12."""
    torch = torch_cuda
    S_ = bsm.synthetic
    p = S_.config5(on_device=True)
    n = p["size"][0]
    A = S_.build(p)
    assert A.stats()["stored_entries"] * 8 > 28e9
    nseg = len(p["diagonals"])
    sub, rows = S_.config5_sample(S_.sample_ids(200, nseg))
    assert sum(b - a + 1 for a, b in rows) > 20_000
    x = p["x"]
    assert torch.equal(x.cpu(), torch.from_numpy(sub["x"]))
    y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    bsm.mul(y, A, x)
    ref = oracle_mul(oracle, sub, N, sub["x"], np.zeros(n))
    assert sampled_relerr(y, ref, rows) < 1e-12
    bsm.mul(y, bsm.transpose(A), x)
    assert sampled_relerr(y, oracle_mul(oracle, sub, T, sub["x"], np.zeros(n)), rows) < 1e-12
    y0 = np.random.default_rng(2).standard_normal(n)
    yd = torch.from_numpy(y0).cuda()
    bsm.mul(yd, A, x, -0.5, 2.0)
    assert sampled_relerr(yd, oracle_mul(oracle, sub, N, sub["x"], y0, -0.5, 2.0, False), rows) < 1e-12
    # and through the multi-device handle (another partition, halo exchange)
    del A
    torch.cuda.empty_cache()
    B = S_.build(p, devices=[0, 0])
    y.fill_(float("nan"))
    bsm.mul(y, B, x)
    assert sampled_relerr(y, ref, rows) < 1e-12


def test_config4_FULL_size_against_the_oracle_on_sampled_rows(torch_cuda, bsm, oracle):
    """BASELINE.json configs[3] at its FULL size (16.4 GB fp32) against the oracle on ~200 sampled block rows (A x: the 16
    blocks of each sampled block row) and ~60 sampled block columns (A^T x: every block of the operator whose column is
    sampled), tolerance 1e-5."""
    torch = torch_cuda
    S_ = bsm.synthetic
    p = S_.config4(on_device=True)
    n = p["size"][0]
    A = S_.build(p)
    sub, rows, cols = S_.config4_sample(S_.sample_ids(200, 15625), S_.sample_ids(60, 15625, seed=3))
    x = p["x"]
    assert torch.equal(x.cpu(), torch.from_numpy(sub["x"]))
    y = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
    bsm.mul(y, A, x)
    assert sampled_relerr(y, oracle_mul(oracle, sub, N, sub["x"], np.zeros(n, np.float32)), rows) < 1e-5
    bsm.mul(y, bsm.transpose(A), x)
    assert sampled_relerr(y, oracle_mul(oracle, sub, T, sub["x"], np.zeros(n, np.float32)), cols) < 1e-5
    y0 = np.random.default_rng(2).standard_normal(n).astype(np.float32)
    yd = torch.from_numpy(y0).cuda()
    bsm.mul(yd, A, x, 0.75, -1.5)
    assert sampled_relerr(yd, oracle_mul(oracle, sub, N, sub["x"], y0, 0.75, -1.5, False), rows) < 1e-5


def test_config5_FULL_size_properties_on_one_gpu(torch_cuda, bsm):
    """BASELINE.json configs[4] at its FULL size (SymmetricBlockMatrix 5M x 5M, 36.8 k diagonal + 147 k
    off-diagonal blocks of 16-256 rows, 28.6 GB fp64) -- generated in HBM, packed by the device-side
    packer -- through size-independent properties: symmetry S x = S^T x, <S x, z> = <x, S z>, linearity,
    alpha / beta, and the SAME operator spread over two (virtual) devices (a different row partition,
    halo exchange) giving the same product.  The oracle cannot finish this size in seconds; the
    partitioned and reduced-size cases against it are above / in test_gpu_parity.py."""
    torch = torch_cuda
    S_ = bsm.synthetic
    p = S_.config5(on_device=True)
    n = p["size"][0]
    assert n == 5_000_000 and len(p["offdiagonals"]) > 140_000
    A = S_.build(p)
    assert A.stats()["stored_entries"] * 8 > 28e9
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) - 0.5
    z = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) - 0.5
    sx, stx, sz, comb = (torch.full_like(x, float("nan")) for _ in range(4))
    bsm.mul(sx, A, x)
    bsm.mul(stx, bsm.transpose(A), x)
    bsm.mul(sz, A, z)
    bsm.mul(comb, A, 2 * x - 3 * z)
    scale = float(sx.abs().max())
    assert scale > 0 and bool(torch.isfinite(sx).all())
    assert float((stx - sx).abs().max()) < 1e-12 * scale
    assert abs(float(torch.dot(sx, z)) - float(torch.dot(x, sz))) < 1e-11 * abs(float(torch.dot(sx, z)))
    assert float((comb - (2 * sx - 3 * sz)).abs().max()) < 1e-12 * scale
    y0 = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    ab = y0.clone()
    bsm.mul(ab, A, x, -0.5, 2.0)
    assert float((ab - (-0.5 * sx + 2.0 * y0)).abs().max()) < 1e-12 * scale
    del A
    torch.cuda.empty_cache()
    B = S_.build(p, devices=[0, 0])  # two parts, x / partial-y halo over the (virtual) link
    parts = B.parts()
    assert parts[1]["touched"][0] < parts[1]["own"][0]
    y2 = torch.full_like(x, float("nan"))
    bsm.mul(y2, B, x)
    assert float((y2 - sx).abs().max()) < 1e-12 * scale


def test_config4_FULL_size_properties_on_one_gpu(torch_cuda, bsm):
    """BASELINE.json configs[3] at its FULL size (VBCRS 2M x 2M, 250 000 blocks of 128x128 fp32, 16.4 GB):
    adjoint identity <A x, z> = <x, A^T z>, linearity, and the row-partitioned multi-device handle."""
    torch = torch_cuda
    S_ = bsm.synthetic
    p = S_.config4(on_device=True)
    n = p["size"][0]
    assert n == 2_000_000 and len(p["blocks"]) == 250_000
    A = S_.build(p)
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.rand(n, dtype=torch.float32, device="cuda", generator=g) - 0.5
    z = torch.rand(n, dtype=torch.float32, device="cuda", generator=g) - 0.5
    ax, az, atz, comb = (torch.full_like(x, float("nan")) for _ in range(4))
    bsm.mul(ax, A, x)
    bsm.mul(az, A, z)
    bsm.mul(atz, bsm.transpose(A), z)
    bsm.mul(comb, A, 2 * x - 3 * z)
    scale = float(ax.abs().max())
    assert scale > 0 and bool(torch.isfinite(ax).all())
    lhs, rhs = float(torch.dot(ax.double(), z.double())), float(torch.dot(x.double(), atz.double()))
    assert abs(lhs - rhs) < 1e-4 * max(abs(lhs), float(ax.double().norm() * z.double().norm()) * 1e-3)
    # (linearity between fp32 results, no oracle at this size: the combination 2x - 3z is itself rounded to fp32)
    assert float((comb - (2 * ax - 3 * az)).abs().max()) < 2e-5 * scale
    del A
    torch.cuda.empty_cache()
    B = S_.build(p, devices=[0, 0, 0])
    y2 = torch.full_like(x, float("nan"))
    bsm.mul(y2, B, x)
    assert float((y2 - ax).abs().max()) < 1e-5 * scale
