#!/usr/bin/env python3
"""bench.py -- headline benchmark of the block-sparse mul! hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is ONE mul!(y, A, x) (3-argument form, alpha = 1, beta = Bool false) through the C ABI
(libbsmrocm.so: bsm_mul) with A, x and y resident in HBM.  Workload at N = 1 is BASELINE.json
configs[1]: VBCRS 100 000 x 100 000, 5 000 variable 8-64 sized fp64 blocks (SplitMix64 seed
0xB5A2).  For N > 1 (one process per GPU, launched by torch.distributed.run) the
GLOBAL operator is (N * 100 000)^2 with N * 5 000 blocks, row-partitioned: every rank owns the
block rows of its 1/N slice of the rows (weak scaling).  Block rows are independent units
(reference src/vbcrs.jl:275-283), y slices are disjoint, so the timed region contains NO collective
(`--allgather` adds the RCCL all-gather of the y slices a Krylov iteration would need).

metric value = algorithmic bytes of all ranks (every stored entry once + block
metadata + x once + y once) * K / max-over-ranks time.

Prints ONE JSON line on rank 0 with the `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# threads of the all-cores CPU variant (OpenMP runtime reads this when the oracle is loaded): the
# GPU box gives one GPU a 16-CPU share
os.environ.setdefault("OMP_NUM_THREADS", str(min(len(os.sched_getaffinity(0)), 16)))

HBM_PEAK_GBPS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); measured-copy ceiling 6290
# HBM-side bytes of one C2 launch from the PMC passes (profiles/r01_c2_pmc_summary.txt):
# FETCH_SIZE 27531.5 KiB x 2 (gfx950 counts 128-B requests as 64 B) + WRITE_SIZE 835.9 KiB
TRAFFIC_BYTES_PER_LAUNCH = 57_240_000


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--launch", choices=["graph", "eager"], default="graph",
                    help="graph: the K timed steps are one hipGraph replay of K bsm_mul launches")
    ap.add_argument("--allgather", action="store_true",
                    help="include an RCCL all-gather of the y slices in every step (N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to "
                    "rehearse the N > 1 code path on a single GPU)")
    ap.add_argument("--device", type=int, default=None, help="override LOCAL_RANK as the HIP device (rehearsal)")
    ap.add_argument("--cold", action="store_true", default=True,
                    help="also report a cold-cache figure (512 MiB flush before every launch); default on")
    ap.add_argument("--no-cold", dest="cold", action="store_false")
    args = ap.parse_args()

    import numpy as np
    import torch
    import bsm_amd as bsm
    from bsm_amd import _lib

    _lib.lib()  # the product path needs the HIP extension; no CPU fallback exists
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev_index = local_rank if args.device is None else args.device
    torch.cuda.set_device(dev_index)
    dist = None
    red_dev = "cuda"  # where the scalar reductions over ranks live
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
            red_dev = "cpu"

    # ---- workload -------------------------------------------------------------------------------
    prob = bsm.synthetic.config2(part=(rank, world) if world > 1 else None)
    A = bsm.VariableBlockCompressedRowStorage(prob["blocks"], prob["rowstart"], prob["colstart"],
                                              prob["size"], own=prob.get("own"))
    st = A.stats()
    assert st["exclusive"] == 1
    n = prob["size"][0]
    # algorithmic bytes of THIS rank: its stored entries + metadata + x once + its y rows once
    own = prob.get("own", (1, n))
    own_rows = own[1] - own[0] + 1
    if world > 1:
        # this rank reads only the x entries its blocks reference (each once) and writes its own rows
        touched = np.zeros(n, dtype=bool)
        for c0, blk in zip(prob["colstart"], prob["blocks"]):
            touched[c0 - 1:c0 - 1 + blk.shape[1]] = True
        alg_bytes = st["alg_bytes"] - 8 * n - 8 * n + 8 * int(touched.sum()) + 8 * own_rows
    else:
        alg_bytes = st["alg_bytes"]
    x = torch.from_numpy(prob["x"]).cuda()
    y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    plan = bsm.MulPlan(y, A, x)
    gather_out = None
    if args.allgather and world > 1:
        gather_out = [torch.empty(n // world, dtype=torch.float64, device="cuda") for _ in range(world)]

    def step():
        plan()
        if gather_out is not None:
            dist.all_gather(gather_out, y[rank * (n // world):(rank + 1) * (n // world)])

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up + (optional) graph capture ---------------------------------------------------------
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    graph = None
    launch = args.launch
    if launch == "graph" and gather_out is None:
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=s):
                    for _ in range(args.steps):
                        plan()
            torch.cuda.current_stream().wait_stream(s)
            graph.replay()  # one untimed replay
            torch.cuda.synchronize()
        except Exception as e:  # pragma: no cover
            print(f"[bench] graph capture unavailable ({e}); falling back to eager", file=sys.stderr)
            graph = None
    if graph is None:
        launch = "eager"

    # ---- timed region: EXACTLY K steps ----------------------------------------------------------------
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record()  # HIP events on the stream the kernels are launched on
    if graph is not None:
        graph.replay()
    else:
        for _ in range(args.steps):
            step()
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    dev_elapsed = ev0.elapsed_time(ev1) * 1e-3
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        b = torch.tensor([float(alg_bytes)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(b, op=dist.ReduceOp.SUM)
        total_bytes = float(b.item())
    else:
        total_bytes = float(alg_bytes)
    value = total_bytes * args.steps / elapsed / 1e9

    # ---- roofline of the dominant kernel --------------------------------------------------------
    # one step == one launch of bsm::panel_kernel<double,8,true,false,true> (forward-only, non-temporal matrix loads); its average duration is
    # the HIP-event time of the timed region / K (back-to-back launches on one stream; the
    # rocprofv3 --kernel-trace average in profiles/ must agree).
    kdur = dev_elapsed / args.steps
    achieved = alg_bytes / kdur / 1e9
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": TRAFFIC_BYTES_PER_LAUNCH,
                "kernel": "bsm::panel_kernel<double,8,true,false,true>",
                "alg_bytes_per_launch": int(alg_bytes), "avg_launch_us": round(kdur * 1e6, 3),
                "note": "warm: the 54 MB operator stays in the 256 MiB Infinity Cache between launches; "
                        "traffic = rocprofv3 FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE per launch, "
                        "profiles/r01_c2_pmc_summary.txt"}

    extra = {}
    if rank == 0 and launch == "graph":
        # the same K steps as individual bsm_mul calls (no graph), for transparency
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        extra["eager_us_per_step"] = round(e0.elapsed_time(e1) * 1e3 / args.steps, 3)
    if args.cold and rank == 0:
        flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
        cold = []
        for _ in range(20):
            flush.fill_(1)
            a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            plan()
            b_.record()
            torch.cuda.synchronize()
            cold.append(a.elapsed_time(b_) * 1e-3)
        cold.sort()
        extra["cold_median_us"] = round(cold[len(cold) // 2] * 1e6, 2)
        extra["cold_GBps"] = round(alg_bytes / cold[len(cold) // 2] / 1e9, 1)
        del flush

    # ---- CPU baseline: the oracle (reference loop structure, scalar C port), rank 0, N = 1 only -----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import load_oracle
        orc = load_oracle()
        perm, rowptr, colind, rowind = orc.vbcrs_build(prob["rowstart"], prob["colstart"])
        blocks = [prob["blocks"][p - 1] for p in perm]
        xh = prob["x"]
        yh = np.zeros(n)
        orc.vbcrs_mul(0, blocks, rowptr, colind, rowind, xh, yh)  # warm
        reps, t0 = 0, time.perf_counter()
        while True:
            orc.vbcrs_mul(0, blocks, rowptr, colind, rowind, xh, yh)
            reps += 1
            dt = time.perf_counter() - t0
            if dt > 10.0 or reps >= 2000:
                break
        # all host cores: one OpenMP task per block row == the reference's `@tasks for browidx`
        # with DynamicScheduler() (src/vbcrs.jl:275-276); reported beside, not as the baseline
        try:
            ncores = int(os.environ.get("OMP_NUM_THREADS", "1"))
            yp = np.zeros(n)
            orc.vbcrs_mul(0, blocks, rowptr, colind, rowind, xh, yp, parallel=True)
            preps, tp0 = 0, time.perf_counter()
            while True:
                orc.vbcrs_mul(0, blocks, rowptr, colind, rowind, xh, yp, parallel=True)
                preps += 1
                dtp = time.perf_counter() - tp0
                if dtp > 5.0 or preps >= 2000:
                    break
            extra["cpu_allcores"] = {"value": round(st["alg_bytes"] * preps / dtp / 1e9, 3), "unit": "GB/s",
                                     "cores": ncores, "kind": "port (OpenMP over block rows)",
                                     "sample": f"{preps} C2 mul! calls in {dtp:.1f} s"}
        except Exception as e:  # pragma: no cover
            extra["cpu_allcores"] = {"error": str(e)}
        # parity of the measured GPU result against this same oracle run
        err = float(np.max(np.abs(y.cpu().numpy() - yh)) / np.max(np.abs(yh)))
        cpu = {"value": round(st["alg_bytes"] * reps / dt / 1e9, 3), "unit": "GB/s", "cores": 1,
               "kind": "port",
               "sample": f"{reps} full C2 mul! calls of oracle/bsm_oracle.c (orc_vbcrs_mul_f64, ctypes "
                         f"marshalling included) in {dt:.1f} s on one host core",
               "gpu_vs_oracle_relerr": err}

    if rank == 0:
        out = {
            "metric": "fp64 block-SpMV GB/s (VBCRS mul!, algorithmic bytes / time)",
            "value": round(value, 1), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 6),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "C2: VBCRS 100000x100000 per GPU, 5000 variable 8-64 fp64 blocks per GPU "
                                   "(SplitMix64 seed 0xB5A2), mul!(y, A, x), x/y/A resident in HBM",
                       "global_rows": n, "blocks_per_gpu": len(prob["blocks"]),
                       "alg_bytes_per_gpu": int(alg_bytes), "launch": launch,
                       "partition": "block rows, no data-path collective" if not gather_out else
                                    "block rows + RCCL all-gather of y slices",
                       "frac_of_hbm_peak": round(value / (HBM_PEAK_GBPS * world), 4)},
            "roofline": roofline,
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if extra:
            out["extra"] = extra
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
