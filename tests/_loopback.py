"""One-rank LOOPBACK rehearsal of blocksparsematrices.jl_amd/distributed.py (test code only).

`RowPartitioned(..., loopback=...)` runs every collective and point-to-point branch of the process-per-GPU layer
against the rank itself instead of skipping it at world == 1.  With backend "nccl" on cuda:0 this is the first --
and on a one-GPU box the only possible -- execution of RCCL for this repository: device-tensor
`batch_isend_irecv` (grouped ncclSend / ncclRecv to the own rank), `all_gather_into_tensor`,
`reduce_scatter_tensor`, `all_reduce`, and the stream ordering between the collectives and the HIP product kernels
that distributed.py relies on ("RCCL orders its transfers by itself").  Every result is compared with the CPU
oracle on the WHOLE operator; sources of transfers are poisoned behind the send, so a transfer that did not happen
or was overtaken shows as NaN.

The same cases run on the CPU (backend "gloo", local products through the packed-image interpreter) to check the
loopback LOGIC without a GPU; gloo has no pair to the own rank, so there `batch_isend_irecv` is replaced by an
in-process copy of send i into receive i -- the matching rule of grouped NCCL point-to-point operations.

What replaces the reference's `@tasks` fan-out: src/vbcrs.jl:275-283, src/symmetricblockmatrix.jl:395-432.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODEV = -2


def loaded_libraries(sub):
    with open("/proc/self/maps") as f:
        return sorted({ln.split()[-1] for ln in f if sub in ln})


def run(backend, port, q):
    """child process: the loopback cases; puts ("ok", [(case, relerr)], extra) or ("err", traceback, None)"""
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist
        gpu = backend == "nccl"
        if gpu:
            torch.cuda.set_device(0)
            sys.path.insert(0, ROOT)
            from bsm_amd import distributed as D0
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), pg_options=D0.nccl_options())
        else:
            dist.init_process_group("gloo", rank=0, world_size=1)

            def in_process(ops):  # send i -> receive i (one peer: the order of posting), like a grouped NCCL exchange
                sends = [o for o in ops if o.op is dist.isend]
                recvs = [o for o in ops if o.op is dist.irecv]
                assert len(sends) == len(recvs) and sends, "every send needs its receive in the same batch"
                for s_, r_ in zip(sends, recvs):
                    assert s_.tensor.shape == r_.tensor.shape and s_.tensor.data_ptr() != r_.tensor.data_ptr()
                    r_.tensor.copy_(s_.tensor)
                return []
            dist.batch_isend_irecv = in_process
        import bsm_amd as bsm
        from bsm_amd import distributed as D
        from _common import N, T, interpret_image, oracle_mul, relerr
        from oracle import load_oracle
        orc = load_oracle()
        dev = "cuda" if gpu else "cpu"
        out = []

        def build(local, own=None):
            if D.is_empty(local):
                return None
            kw = {} if gpu else {"device": NODEV}
            if own is not None and own[1] >= own[0]:
                kw["own"] = own
            return bsm.synthetic.build(local, **kw)

        def image_mul(H, rng):
            """CPU: the handle's product through the image interpreter (writes the handle's own range only)"""
            if gpu or H is None:
                return None

            def f(yy, xx, alpha, beta, lop=N):
                strong = beta is False
                a = 1 if alpha is True else alpha
                b = 0 if strong else (1 if beta is True else beta)
                res = interpret_image(H, lop, xx.numpy(), yy.numpy(), a, b, strong)
                if lop != N or rng is None:
                    yy[:] = torch.from_numpy(res)
                else:
                    yy[rng[0] - 1:rng[1]] = torch.from_numpy(res[rng[0] - 1:rng[1]])
                return yy
            return f

        def sync():
            if gpu:
                torch.cuda.synchronize()

        y0_of = lambda n: np.random.default_rng(7).standard_normal(n)
        combos = ((True, False), (0.5, -2.0))

        def ref_of(prob, op, y0, alpha, beta):
            a = 1 if alpha is True else alpha
            return oracle_mul(orc, prob, op, prob["x"], y0, a, 0 if beta is False else beta, strong=beta is False)

        # ---- (a) VBCRS forward, y replicated by ONE all_gather_into_tensor; (a') x partitioned: all-gather of the x slices
        prob = bsm.synthetic.config2(n=20_000 if gpu else 4000, nblocks=900 if gpu else 250)
        n = prob["size"][0]
        own = (1, n)
        A = build(prob, own)
        x = torch.from_numpy(prob["x"].copy()).to(dev)
        y0 = y0_of(n)
        P = D.RowPartitioned(A, own, own, gather=True, loopback=True)
        for alpha, beta in combos:
            for rep in range(2):  # again: cached buffers
                y = torch.from_numpy(y0.copy()).to(dev)
                P.mul(y, x, alpha, beta, local_mul=image_mul(A, own))
            sync()
            out.append(("vbcrs gather=True (all_gather_into_tensor) %s" % (beta,), relerr(y.cpu().numpy(), ref_of(prob, N, y0, alpha, beta))))
        P = D.RowPartitioned(A, own, own, gather=False, loopback=True, xneed=None)
        for rep in range(2):
            y = torch.from_numpy(y0.copy()).to(dev)
            P.mul(y, x, 0.5, -2.0, x_distributed=True, local_mul=image_mul(A, own))
        sync()
        out.append(("vbcrs x partitioned (all-gather of the x slices)", relerr(y.cpu().numpy(), ref_of(prob, N, y0, 0.5, -2.0))))
        assert not bool(torch.isnan(x).any()), "x must be whole again after the loopback all-gather"

        # ---- (b) transposed product ACROSS the row partition: reduce_scatter_tensor / all_reduce
        for gather, what in ((False, "reduce_scatter_tensor"), (True, "all_reduce")):
            P = D.RowPartitioned(A, own, own, gather=gather, loopback=True)
            for alpha, beta in combos:
                for rep in range(2):
                    y = torch.from_numpy(y0.copy()).to(dev)
                    P.mul(y, x, alpha, beta, op=T, local_mul=image_mul(A, None))
                sync()
                out.append(("vbcrs transposed across (%s) %s" % (what, beta), relerr(y.cpu().numpy(), ref_of(prob, T, y0, alpha, beta))))

        # ---- (c) halo exchanges through grouped self send / recv: the rank owns the MIDDLE rows, a phantom neighbour the rest
        sym = bsm.synthetic.config5(n=60_000 if gpu else 5000, lo=16, hi=128 if gpu else 96, halfband=3)
        bsp = bsm.synthetic.config1(n=3000, nblocks=120, bs=24)
        for prob, kind in ((sym, "symmetric"), (bsp, "blocksparse"), (bsm.synthetic.config2(n=20_000 if gpu else 4000, nblocks=900 if gpu else 250), "vbcrs")):
            n = prob["size"][0]
            issym = kind == "symmetric"
            if issym:  # cut at segment boundaries (a diagonal block straddling the cut would be a boundary block: allowed, but keep it simple)
                starts = sorted(int(np.min(d)) for d in prob["diagonalindices"])
                own = (starts[len(starts) // 4], starts[3 * len(starts) // 4] - 1)
            else:
                own = (n // 4 + 1, 3 * n // 4)
            lists = {"symmetric": lambda: prob["diagonalindices"] + prob["rowindices"] + prob["colindices"],
                     "blocksparse": lambda: prob["rowindices"],
                     "vbcrs": lambda: [np.array([r, r + b.shape[0] - 1]) for r, b in zip(prob["rowstart"], prob["blocks"])]}[kind]()
            touched = D._touched(own, lists)
            x = torch.from_numpy(prob["x"].copy()).to(dev)
            y0 = y0_of(n)
            A = build(prob, touched)
            # x "partitioned": everything outside own belongs to the phantom and comes in through the exchange
            P = D.RowPartitioned(A, own, touched, gather=False, symmetric=issym, xneed=touched if issym else None, loopback=own)
            for alpha, beta in combos:
                for rep in range(3):
                    y = torch.from_numpy(y0.copy()).to(dev)
                    P.mul(y, x, alpha, beta, x_distributed=True, local_mul=image_mul(A, touched))
                sync()
                out.append(("%s halo, self send/recv (batch_isend_irecv) %s" % (kind, beta), relerr(y.cpu().numpy(), ref_of(prob, N, y0, alpha, beta))))
            assert torch.equal(x.cpu(), torch.from_numpy(prob["x"])), "x must be whole again after the loopback halo"
            # ---- (c') A * X, 5 right-hand sides through the same halo: mul_multi -- one local multi-RHS product, the columns of
            # every halo segment in the one grouped exchange
            K = 5
            Xf = np.stack([prob["x"] * (k + 1) + 0.25 * k for k in range(K)], axis=1)
            Y0m = np.stack([np.random.default_rng(11 + k).standard_normal(n) for k in range(K)], axis=1)
            colmajor = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(dev).t()
            Xm = colmajor(Xf)
            if gpu:
                hook = None
            else:
                one = image_mul(A, touched)

                def hook(YY, XX, a_, b_):
                    for k in range(K):
                        one(YY[:, k], XX[:, k], a_, b_)
                    return YY
            for alpha, beta in combos:
                for rep in range(2):
                    Ym = colmajor(Y0m)
                    P.mul_multi(Ym, Xm, alpha, beta, x_distributed=True, local_mul=hook)
                sync()
                a_ = 1 if alpha is True else alpha
                got = Ym.cpu().numpy()
                err = max(relerr(got[:, k], oracle_mul(orc, prob, N, Xf[:, k].copy(), Y0m[:, k].copy(), a_, 0 if beta is False else beta,
                                                       strong=beta is False)) for k in range(K))
                out.append(("%s halo, %d right-hand sides (mul_multi) %s" % (kind, K, beta), err))
            assert np.array_equal(Xm.cpu().numpy(), Xf), "X must be whole again after the loopback halo"
            # ---- (d) the same with the exchange OVERLAPPED with the interior rows (what bench.py --gpus N times)
            for xmode in (("halo", "allgather") if kind != "blocksparse" else ("auto",)):
                interior, boundary, bt, bx = D.split_interior(prob, own)
                Ai, Ab = build(interior, own), build(boundary, bt)
                assert Ab is not None, "the rehearsal needs boundary blocks"
                if xmode == "auto":
                    extra = max(0, own[0] - bx[0]) + max(0, bx[1] - own[1])
                    xmode = "halo" if extra <= own[1] - own[0] + 1 else "allgather"
                P = D.RowPartitioned(Ab, own, bt, gather=False, symmetric=issym, xneed=(bx if xmode == "halo" else None),
                                     interior=Ai, loopback=own)
                idle = (lambda yy, xx, a, b: P._combine(yy, slice(own[0] - 1, own[1]), 0, b))
                import contextlib
                for alpha, beta in combos:
                    # GPU: on the compute stream whose CU mask leaves one CU per XCD to RCCL (distributed.compute_stream)
                    with (torch.cuda.stream(D.compute_stream()) if gpu else contextlib.nullcontext()):
                        for rep in range(3):
                            y = torch.from_numpy(y0.copy()).to(dev)
                            P.mul_overlapped(y, x, alpha, beta, local_mul=image_mul(Ab, bt),
                                             interior_mul=(image_mul(Ai, own) if Ai is not None else idle) if not gpu else None)
                    sync()
                    out.append(("%s overlapped step, x %s %s" % (kind, xmode, beta), relerr(y.cpu().numpy(), ref_of(prob, N, y0, alpha, beta))))
                assert torch.equal(x.cpu(), torch.from_numpy(prob["x"]))
        extra = {"backend": dist.get_backend(), "librccl": loaded_libraries("librccl"), "libbsmrocm": loaded_libraries("libbsmrocm")}
        dist.barrier()
        dist.destroy_process_group()
        q.put(("ok", out, extra))
    except Exception:  # pragma: no cover
        import traceback
        q.put(("err", traceback.format_exc(), None))
        raise
