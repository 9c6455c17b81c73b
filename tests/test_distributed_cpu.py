"""CPU suite, part 3: the N > 1 path over `gloo` with world_size 2 (and 3).

No GPU exists here, so each rank's LOCAL product is executed by the packed-image interpreter of
tests/_common.py (the same image the HIP kernel walks); everything else -- the block-row partition,
the ownership ranges handed to the C ABI, the point-to-point halo reduce of the symmetric path and
the y all-gather -- is the real code of blocksparsematrices.jl_amd/distributed.py.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODEV = -2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, kind, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import bsm_amd as bsm
        from bsm_amd import distributed as D
        from _common import N, interpret_image, oracle_mul, relerr
        from oracle import load_oracle

        from _common import T
        op, axis = N, 0
        if kind == "vbcrs":
            prob = bsm.synthetic.config2(n=5000, nblocks=300)
            local, own = D.split_vbcrs(prob, rank, world)
            touched = own
        elif kind == "vbcrs_tiny":  # 2 block rows on 3 ranks: one rank owns nothing and creates no handle
            rng = np.random.default_rng(1)
            prob = dict(kind="vbcrs", blocks=[np.asfortranarray(rng.standard_normal((9, 12))),
                                              np.asfortranarray(rng.standard_normal((7, 5)))],
                        rowstart=np.array([4, 30]), colstart=np.array([2, 20]), size=(40, 40),
                        x=rng.standard_normal(40))
            local, own = D.split_vbcrs(prob, rank, world)
            touched = own
        elif kind == "vbcrs_T_across":  # transposed product of a ROW-partitioned operator: reduce-scatter
            prob = bsm.synthetic.config2(n=5000, nblocks=300)
            local, own = D.split_vbcrs(prob, rank, world)
            touched, op = own, T
        elif kind == "vbcrs_cols_T":  # column partition: the transposed product is collective-free
            prob = bsm.synthetic.config2(n=5000, nblocks=300)
            local, own = D.split_vbcrs(prob, rank, world, axis=1)
            touched, op, axis = own, T, 1
        elif kind == "blocksparse":
            prob = bsm.synthetic.config1(n=3000, nblocks=120, bs=24)
            local, own, touched = D.split_blocksparse(prob, rank, world)
        else:
            prob = bsm.synthetic.config5(n=5000, lo=16, hi=96, halfband=3)
            local, own, touched = D.split_symmetric(prob, rank, world)
        # `own` = the rows this handle is responsible for scaling by beta (C ABI bsm_options.own_lo/hi)
        A = None if D.is_empty(local) else \
            bsm.synthetic.build(local, device=NODEV, **({"own": touched} if axis == 0 else {}))
        n = prob["size"][0]
        x = torch.from_numpy(prob["x"].copy())
        y0 = np.random.default_rng(7).standard_normal(n)

        def local_mul(yy, xx, alpha, beta, lop=N):
            strong = beta is False
            a = 1 if alpha is True else alpha
            b = 0 if strong else (1 if beta is True else beta)
            out = interpret_image(A, lop, xx.numpy(), yy.numpy(), a, b, strong)
            if lop != N:
                yy[:] = torch.from_numpy(out)  # transposed products scale the whole y
            elif kind in ("vbcrs", "vbcrs_tiny"):
                lo, hi = own  # the handle only writes the rows it owns
                yy[lo - 1:hi] = torch.from_numpy(out[lo - 1:hi])
            else:
                lo, hi = touched
                yy[lo - 1:hi] = torch.from_numpy(out[lo - 1:hi])
            return yy

        results = []
        for gather in (True, False):
            P = D.RowPartitioned(A, own, touched, gather=gather, axis=axis, symmetric=(kind == "symmetric"))
            for alpha, beta in ((True, False), (0.5, -2.0)):
                y = torch.from_numpy(y0.copy())
                P.mul(y, x, alpha, beta, local_mul=(local_mul if A is not None else None), op=op)
                if not gather:  # only this rank's output range is final: keep it, take the rest from
                    lo, hi = own if (op == N) == (axis == 0) or kind == "symmetric" else P.out_range(n)
                    part = torch.zeros_like(y)
                    if hi >= lo:
                        part[lo - 1:hi] = y[lo - 1:hi]
                    dist.all_reduce(part)  # test-side assembly of the slices
                    y = part
                results.append(y.numpy().copy())
        if kind in ("vbcrs", "symmetric", "vbcrs_tiny"):
            # x and y PARTITIONED like the rows: x is valid on the own range only (NaN elsewhere); the
            # symmetric operator fetches its halo point-to-point, the VBCRS one all-gathers the slices
            sym = kind == "symmetric"
            P = D.RowPartitioned(A, own, touched, gather=False, symmetric=sym, xneed=touched if sym else None)
            for _ in range(2):
                xd = torch.full_like(x, float("nan"))
                if own[1] >= own[0]:
                    xd[own[0] - 1:own[1]] = x[own[0] - 1:own[1]]
                y = torch.from_numpy(y0.copy())
                P.mul(y, xd, 0.5, -2.0, x_distributed=True, local_mul=(local_mul if A is not None else None))
            part = torch.zeros_like(y)
            if own[1] >= own[0]:
                part[own[0] - 1:own[1]] = y[own[0] - 1:own[1]]
            dist.all_reduce(part)
            results.append(part.numpy().copy())
        combos = (((1, 0), (0.5, -2.0)) * 2 + ((0.5, -2.0),))[:len(results)]
        if kind in ("vbcrs", "symmetric", "blocksparse", "vbcrs_tiny"):
            # the same partitioned-vector product with the exchange OVERLAPPED with the interior rows:
            # two images per rank (interior / boundary blocks, distributed.split_interior), the boundary
            # one on the side of the exchange -- both interpreted from their packed images here
            interior, boundary, bt, bx = D.split_interior(local, own)
            ni = sum(len(interior.get(k, ())) for k in ("blocks", "diagonals", "offdiagonals"))
            nb = sum(len(boundary.get(k, ())) for k in ("blocks", "diagonals", "offdiagonals"))
            assert ni + nb == sum(len(local.get(k, ())) for k in ("blocks", "diagonals", "offdiagonals"))
            Ai = None if D.is_empty(interior) else bsm.synthetic.build(interior, device=NODEV, own=own)
            Ab = None if D.is_empty(boundary) else bsm.synthetic.build(boundary, device=NODEV, own=bt)

            def image_mul(H, rng):
                def f(yy, xx, alpha, beta):
                    strong = beta is False
                    a = 1 if alpha is True else alpha
                    b = 0 if strong else (1 if beta is True else beta)
                    out = interpret_image(H, N, xx.numpy(), yy.numpy(), a, b, strong)
                    yy[rng[0] - 1:rng[1]] = torch.from_numpy(out[rng[0] - 1:rng[1]])  # the handle's own range
                    return yy
                return f
            sym = kind == "symmetric"
            for xmode in (("halo", "allgather") if kind != "blocksparse" else ("allgather",)):
                P = D.RowPartitioned(Ab, own, bt, gather=False, symmetric=sym,
                                     xneed=(bx if xmode == "halo" else None), interior=Ai)
                for alpha, beta in ((True, False), (0.5, -2.0)):
                    for _ in range(2):  # second pass: cached plans and buffers
                        xd = torch.full_like(x, float("nan"))
                        if own[1] >= own[0]:
                            xd[own[0] - 1:own[1]] = x[own[0] - 1:own[1]]
                        y = torch.from_numpy(y0.copy())
                        P.mul_overlapped(y, xd, alpha, beta,
                                         local_mul=image_mul(Ab, bt) if Ab is not None else None,
                                         interior_mul=image_mul(Ai, own) if Ai is not None else (lambda yy, xx, a, b: P._combine(yy, slice(own[0] - 1, own[1]), 0, b) if own[1] >= own[0] else None))
                    part = torch.zeros_like(y)
                    if own[1] >= own[0]:
                        part[own[0] - 1:own[1]] = y[own[0] - 1:own[1]]
                    dist.all_reduce(part)
                    results.append(part.numpy().copy())
                    combos = combos + (((1, 0) if beta is False else (alpha, beta)),)
        multi = None
        if kind in ("vbcrs", "symmetric", "blocksparse", "vbcrs_tiny"):
            # A * X, three right-hand sides, X and Y (column-major) PARTITIONED like the rows: mul_multi -- one local product
            # for all columns, the columns of every halo segment in the one batch of the exchange
            sym = kind == "symmetric"
            K = 3
            Xf = np.stack([prob["x"] * (k + 1) + 0.25 * k for k in range(K)], axis=1)
            Y0 = np.stack([np.random.default_rng(11 + k).standard_normal(n) for k in range(K)], axis=1)

            def colmajor(a):
                return torch.from_numpy(np.ascontiguousarray(a.T)).t()

            def multi_hook(YY, XX, alpha, beta):
                for k in range(K):
                    local_mul(YY[:, k], XX[:, k], alpha, beta)
                return YY
            P = D.RowPartitioned(A, own, touched, gather=False, symmetric=sym, xneed=touched if sym else None)
            for _ in range(2):  # second pass: cached plans and buffers
                Xd = colmajor(np.full_like(Xf, np.nan))
                if own[1] >= own[0]:
                    Xd[own[0] - 1:own[1]] = torch.from_numpy(Xf[own[0] - 1:own[1]])
                Y = colmajor(Y0)
                P.mul_multi(Y, Xd, 0.5, -2.0, x_distributed=True, local_mul=(multi_hook if A is not None else None))
            part = torch.zeros((n, K), dtype=torch.float64)
            if own[1] >= own[0]:
                part[own[0] - 1:own[1]] = Y[own[0] - 1:own[1]]
            dist.all_reduce(part)
            multi = (Xf, Y0, part.numpy().copy())
        if rank == 0:
            orc = load_oracle()
            errs = []
            for (alpha, beta), got in zip(combos, results):
                ref = oracle_mul(orc, prob, op, prob["x"], y0, alpha, beta, strong=(beta == 0))
                errs.append(relerr(got, ref))
            if multi is not None:
                Xf, Y0, got = multi
                for k in range(Xf.shape[1]):
                    errs.append(relerr(got[:, k], oracle_mul(orc, prob, N, Xf[:, k].copy(), Y0[:, k].copy(), 0.5, -2.0, strong=False)))
            q.put(("ok", errs, own, touched))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put(("err", traceback.format_exc(), None, None))
        raise


@pytest.mark.parametrize("kind,world", [("vbcrs", 2), ("symmetric", 2), ("symmetric", 3), ("blocksparse", 2),
                                        ("blocksparse", 3), ("vbcrs_T_across", 2), ("vbcrs_cols_T", 2),
                                        ("vbcrs_tiny", 3)])
def test_row_partitioned_over_gloo(kind, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    status, errs, own, touched = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
    assert status == "ok", errs
    assert all(e < 1e-12 for e in errs), " ".join("%.2e" % e for e in errs)
    assert all(p.exitcode == 0 for p in procs)


def test_partition_is_a_partition():
    sys.path.insert(0, ROOT)
    import bsm_amd as bsm
    from bsm_amd import distributed as D
    prob = bsm.synthetic.config2(n=8000, nblocks=500)
    seen, prev_hi = 0, 0
    for r in range(4):
        local, own = D.split_vbcrs(prob, r, 4)
        seen += len(local["blocks"])
        assert own[0] == prev_hi + 1
        prev_hi = own[1]
        for rs, b in zip(local["rowstart"], local["blocks"]):
            assert own[0] <= rs and rs + b.shape[0] - 1 <= own[1]
    assert seen == len(prob["blocks"]) and prev_hi == 8000
    sp = bsm.synthetic.config3(nseg=30, bs=16, halfband=4)
    nd = no = 0
    for r in range(3):
        local, own, touched = D.split_symmetric(sp, r, 3)
        nd += len(local["diagonals"])
        no += len(local["offdiagonals"])
        assert touched[0] <= own[0] and touched[1] >= own[1]
    assert nd == len(sp["diagonals"]) and no == len(sp["offdiagonals"])
    assert D.balanced_cuts([1, 1, 1, 1], 2) == [0, 2, 4]
    bp = bsm.synthetic.config1(n=2000, nblocks=90, bs=16)
    nb, prev_hi = 0, 0
    for r in range(4):
        local, own, touched = D.split_blocksparse(bp, r, 4)
        nb += len(local["blocks"])
        assert own[0] == prev_hi + 1 and touched[0] <= own[0] and touched[1] >= own[1]
        prev_hi = own[1]
        for rows in local["rowindices"]:
            assert own[0] <= int(np.min(rows)) <= own[1]
    assert nb == len(bp["blocks"]) and prev_hi == 2000
    cp = bsm.synthetic.config2(n=8000, nblocks=500)
    for r in range(3):
        local, own = D.split_vbcrs(cp, r, 3, axis=1)
        for cs, b in zip(local["colstart"], local["blocks"]):
            assert own[0] <= cs and cs + b.shape[1] - 1 <= own[1]


def test_one_rank_loopback_rehearsal_logic():
    """`RowPartitioned(loopback=...)`: every collective / point-to-point branch against the rank itself (tests/_loopback.py).
    Here on the CPU over gloo (self send / recv replaced by an in-process copy: gloo has no pair to the own rank) with the
    image interpreter as local product -- the logic of what `-m gpu` runs over RCCL on the one GPU of the test box."""
    import _loopback
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_loopback.run, args=("gloo", _free_port(), q))
    p.start()
    status, res, extra = q.get(timeout=400)
    p.join(timeout=120)
    assert status == "ok", res
    assert len(res) >= 20 and all(e < 1e-12 for _, e in res), res
    assert p.exitcode == 0
