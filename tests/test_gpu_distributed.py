"""GPU suite (-m gpu): the process-per-GPU layer (blocksparsematrices.jl_amd/distributed.py) with the
REAL HIP local product.  world_size 2 (and 3) over `gloo`, every rank on cuda:0 (the GPU box has one
MI355X; RCCL refuses two ranks on one device, the collectives' code path is the same): each rank
creates its own single-device handle through the C ABI (own = the rows it touches, from
bsm_partition_rows), multiplies on the GPU, exchanges the halo / reduce-scatters / all-gathers, and
rank 0 compares the assembled result with the CPU oracle on the WHOLE operator."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        import bsm_amd as bsm
        from bsm_amd import distributed as D
        from _common import N, T, oracle_mul, relerr

        op, axis = N, 0
        if kind.startswith("vbcrs"):
            prob = bsm.synthetic.config2(n=20_000, nblocks=900)
            if kind == "vbcrs_tiny":  # 2 block rows on 3 ranks: one rank owns nothing
                rng = np.random.default_rng(1)
                prob = dict(kind="vbcrs", blocks=[np.asfortranarray(rng.standard_normal((9, 12))),
                                                  np.asfortranarray(rng.standard_normal((7, 5)))],
                            rowstart=np.array([4, 30]), colstart=np.array([2, 20]), size=(40, 40),
                            x=rng.standard_normal(40))
            local, own = D.split_vbcrs(prob, rank, world)
            touched = own
            if kind == "vbcrs_T":  # transposed product of a ROW-partitioned operator: reduce-scatter
                op = T
        elif kind == "blocksparse":
            prob = bsm.synthetic.config1(n=3000, nblocks=120, bs=24)
            local, own, touched = D.split_blocksparse(prob, rank, world)
        else:
            prob = bsm.synthetic.config5(n=60_000, lo=16, hi=128, halfband=3)
            local, own, touched = D.split_symmetric(prob, rank, world)
        A = D.build_local(local, touched)  # a real device handle (None for a rank without blocks)
        assert (A is None) == D.is_empty(local)
        n = prob["size"][0]
        x = torch.from_numpy(prob["x"].copy()).cuda()
        y0 = np.random.default_rng(7).standard_normal(n)
        results = []
        for gather in (True, False):
            P = D.RowPartitioned(A, own, touched, gather=gather, axis=axis,
                                 symmetric=(prob["kind"] == "symmetric"))
            for alpha, beta in ((True, False), (0.5, -2.0)):
                y = torch.from_numpy(y0.copy()).cuda()
                P.mul(y, x, alpha, beta, op=op)
                P.mul(y.copy_(torch.from_numpy(y0)), x, alpha, beta, op=op)  # again: cached plan / buffers
                torch.cuda.synchronize()
                y = y.cpu()
                if not gather:  # only this rank's output range is final
                    lo, hi = own if (op == N or prob["kind"] == "symmetric") else P.out_range(n)
                    part = torch.zeros_like(y)
                    if hi >= lo:
                        part[lo - 1:hi] = y[lo - 1:hi]
                    dist.all_reduce(part)  # test-side assembly of the slices
                    y = part
                results.append(y.numpy().copy())
        if kind in ("vbcrs", "symmetric", "vbcrs_tiny"):
            # x and y PARTITIONED like the rows: x is valid on the own range only (NaN elsewhere); the
            # symmetric operator fetches its halo point-to-point, the VBCRS one all-gathers the slices
            sym = prob["kind"] == "symmetric"
            P = D.RowPartitioned(A, own, touched, gather=False, symmetric=sym, xneed=touched if sym else None)
            for _ in range(2):
                xd = torch.full_like(x, float("nan"))
                if own[1] >= own[0]:
                    xd[own[0] - 1:own[1]] = x[own[0] - 1:own[1]]
                y = torch.from_numpy(y0.copy()).cuda()
                P.mul(y, xd, 0.5, -2.0, x_distributed=True)
            torch.cuda.synchronize()
            part = torch.zeros(n, dtype=torch.float64)
            if own[1] >= own[0]:
                part[own[0] - 1:own[1]] = y.cpu()[own[0] - 1:own[1]]
            dist.all_reduce(part)
            results.append(part.numpy().copy())
        combos = (((1, 0), (0.5, -2.0)) * 2 + ((0.5, -2.0),))[:len(results)]
        if kind in ("vbcrs", "symmetric", "blocksparse", "vbcrs_tiny"):
            # the exchange OVERLAPPED with the interior rows (what bench.py --gpus N times): interior and
            # boundary blocks as two real device handles, the boundary product and both exchanges on a
            # side stream beside the interior launch
            for xmode in (("halo", "allgather") if kind != "blocksparse" else ("auto",)):
                P = D.build_overlapped(local, own, symmetric=(prob["kind"] == "symmetric"), xmode=xmode)
                for alpha, beta in ((True, False), (0.5, -2.0)):
                    for _ in range(3):  # again: cached plans, reused receive buffers
                        xd = torch.full_like(x, float("nan"))
                        if own[1] >= own[0]:
                            xd[own[0] - 1:own[1]] = x[own[0] - 1:own[1]]
                        y = torch.from_numpy(y0.copy()).cuda()
                        P.mul_overlapped(y, xd, alpha, beta)
                    torch.cuda.synchronize()
                    part = torch.zeros(n, dtype=torch.float64)
                    if own[1] >= own[0]:
                        part[own[0] - 1:own[1]] = y.cpu()[own[0] - 1:own[1]]
                    dist.all_reduce(part)
                    results.append(part.numpy().copy())
                    combos = combos + (((1, 0) if beta is False else (alpha, beta)),)
        multi = None
        if kind in ("vbcrs", "symmetric", "blocksparse", "vbcrs_tiny"):
            # A * X with 6 right-hand sides, X and Y (column-major) PARTITIONED like the rows: mul_multi -- ONE local
            # multi-RHS HIP product per rank, the columns of every halo segment in the one batch of the exchange
            sym = prob["kind"] == "symmetric"
            K = 6
            Xf = np.stack([prob["x"] * (k + 1) + 0.25 * k for k in range(K)], axis=1)
            Y0 = np.stack([np.random.default_rng(11 + k).standard_normal(n) for k in range(K)], axis=1)
            colmajor = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).cuda().t()
            P = D.RowPartitioned(A, own, touched, gather=False, symmetric=sym, xneed=touched if sym else None)
            for _ in range(2):  # again: cached plans and buffers
                Xd = colmajor(np.full_like(Xf, np.nan))
                if own[1] >= own[0]:
                    Xd[own[0] - 1:own[1]] = torch.from_numpy(Xf[own[0] - 1:own[1]]).cuda()
                Y = colmajor(Y0)
                P.mul_multi(Y, Xd, 0.5, -2.0, x_distributed=True)
            torch.cuda.synchronize()
            part = torch.zeros((n, K), dtype=torch.float64)
            if own[1] >= own[0]:
                part[own[0] - 1:own[1]] = Y.cpu()[own[0] - 1:own[1]]
            dist.all_reduce(part)
            multi = (Xf, Y0, part.numpy().copy())
        if rank == 0:
            from oracle import load_oracle
            orc = load_oracle()
            errs = []
            for (alpha, beta), got in zip(combos, results):
                ref = oracle_mul(orc, prob, op, prob["x"], y0, alpha, beta, strong=(beta == 0))
                errs.append(relerr(got, ref))
            if multi is not None:
                Xf, Y0, got = multi
                for k in range(Xf.shape[1]):
                    errs.append(relerr(got[:, k], oracle_mul(orc, prob, N, Xf[:, k].copy(), Y0[:, k].copy(), 0.5, -2.0, strong=False)))
            q.put(("ok", errs))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put(("err", traceback.format_exc()))
        raise


@pytest.mark.parametrize("kind,world", [("vbcrs", 2), ("symmetric", 2), ("symmetric", 3), ("blocksparse", 2),
                                        ("vbcrs_T", 2), ("vbcrs_tiny", 3)])
def test_row_partitioned_with_the_hip_local_product(kind, world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    status, errs = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
    assert status == "ok", errs
    assert all(e < 1e-12 for e in errs), errs
    assert all(p.exitcode == 0 for p in procs)


def test_bench_launches_its_own_ranks_and_reports_parity():
    """`python bench.py --gpus 2` as the driver invokes it (no torch.distributed.run in front): the script starts
    its own rank processes, runs the overlapped C5 step + C4 in `extra` (gloo rehearsal: both ranks on cuda:0,
    operators shrunk) and prints ONE line with the parity number, the exchange time and the ranks it saw."""
    import json
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--device", "0",
                        "--scale", "0.02", "--steps", "4", "--warmup", "2"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["dtype"] == "f64"
    c = d["config"]
    assert c["ranks"] == 2 and c["overlap"] is True and len(c["devices"]) == 2
    assert c["parity_relerr"] <= 1e-12 and "exchange_us" in c and c["local_kernel_us_max"] > 0
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    c4 = d["extra"]["c4"]
    assert c4["parity_relerr"] <= 1e-5 and c4["overlap"] is True
    assert "value_invalid" not in d
    # the oracle saw the last timed step of both workloads (sampled block rows of every rank)
    assert c["parity_vs_oracle_sampled"] <= 1e-12 and c4["parity_vs_oracle_sampled"] <= 1e-5
    # the scaling anchor is a figure of THIS run: rank 0 alone on the same operator, same build, same code path
    a = d["n1_same_workload"]
    assert a["GBps"] > 0 and a["same_build"] is True and "THIS run" in a["source"]
    assert d["speedup_vs_n1_same_workload"] == round(d["value"] / a["GBps"], 3)


def test_rccl_executes_every_collective_branch_on_one_rank():
    """The `nccl` (= RCCL) backend really runs: world size 1 on cuda:0, in a fresh child process, with
    `RowPartitioned(loopback=...)` driving (a) all_gather_into_tensor of y and of the x slices, (b) reduce_scatter_tensor /
    all_reduce of the across-partition transpose, (c) the symmetric / index-list / VBCRS halo plans through grouped self
    send / recv (`batch_isend_irecv` on device tensors), (d) `mul_overlapped` with the exchange on its side stream --
    every result against the CPU oracle (tests/_loopback.py).  RCCL refuses two ranks on one device, so this is the only
    form in which the one-GPU box can execute the branches the 8-GPU run depends on."""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _loopback
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_loopback.run, args=("nccl", _free_port(), q))
    p.start()
    status, res, extra = q.get(timeout=500)
    p.join(timeout=120)
    assert status == "ok", res
    assert len(res) >= 20 and all(e < 1e-12 for _, e in res), res
    assert extra["backend"] == "nccl" and extra["librccl"], extra   # librccl.so is mapped into the process that ran them
    assert extra["libbsmrocm"], extra
    assert p.exitcode == 0


def test_segment_add_and_reserved_cu_stream():
    """The two C-ABI helpers of the process-per-GPU layer: bsm_vec_add_segments (delivery of the partial-y segments in one
    launch; overlapping segments refused) and bsm_stream_create_reserved (a compute stream whose CU mask leaves CUs to
    RCCL's kernels) -- a product on that stream against the oracle."""
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bsm_amd as bsm
    from bsm_amd import distributed as D
    from bsm_amd import matrices as M
    from _common import N, oracle_mul, relerr
    from oracle import load_oracle
    for dt in (torch.float64, torch.float32, torch.complex128, torch.complex64):
        g = torch.Generator(device="cpu").manual_seed(3)
        y = torch.randn(1000, dtype=dt, generator=g).cuda()
        ref = y.clone()
        srcs = [torch.randn(n, dtype=dt, generator=g).cuda() for n in (17, 300, 1, 64)]
        offs = (0, 100, 500, 936)
        dsts = [y[o:o + s_.shape[0]] for o, s_ in zip(offs, srcs)]
        add = M.SegmentAdd(y, dsts, srcs)
        add()
        add()
        for o, s_ in zip(offs, srcs):
            ref[o:o + s_.shape[0]] += 2 * s_
        torch.cuda.synchronize()
        assert torch.allclose(y, ref, rtol=1e-6, atol=0)
    # more segments than one launch takes (8), empty ones in between: every non-empty segment added exactly once
    y = torch.zeros(2000, dtype=torch.float64, device="cuda")
    lens = [5, 0, 7, 0, 0, 3, 11, 2, 0, 9, 4, 6, 0, 8, 1, 0, 13]
    offs = [100 * i for i in range(len(lens))]
    srcs = [torch.full((n,), float(i + 1), dtype=torch.float64, device="cuda") for i, n in enumerate(lens)]
    M.SegmentAdd(y, [y[o:o + n] for o, n in zip(offs, lens)], srcs)()
    ref = torch.zeros_like(y)
    for i, (o, n) in enumerate(zip(offs, lens)):
        ref[o:o + n] = i + 1
    torch.cuda.synchronize()
    assert torch.equal(y, ref)
    with pytest.raises(Exception):
        M.SegmentAdd(y, [y[0:10], y[5:15]], [srcs[6][:10], srcs[6][:10]])()
    st = D.compute_stream()
    assert st is D.compute_stream()  # cached
    prob = bsm.synthetic.config3(nseg=40)
    A = bsm.synthetic.build(prob)
    x = torch.from_numpy(prob["x"]).cuda()
    yd = torch.full((prob["size"][0],), float("nan"), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        for _ in range(3):
            bsm.mul(yd, A, x)
    st.synchronize()
    assert relerr(yd.cpu().numpy(), oracle_mul(load_oracle(), prob, N, prob["x"], np.zeros(prob["size"][0]))) < 1e-12
