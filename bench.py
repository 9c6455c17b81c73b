#!/usr/bin/env python3
"""bench.py -- headline benchmark of the block-sparse mul! hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is ONE mul!(y, A, x) (3-argument form, alpha = 1, beta = Bool false) through the C ABI
(libbsmrocm.so: bsm_mul) with A, x and y resident in HBM.

N = 1 (default): BASELINE.json configs[1] = C2, VBCRS 100 000 x 100 000, 5 000 variable 8-64 sized
    fp64 blocks (SplitMix64 seed 0xB5A2); the K timed steps are one hipGraph replay of
    K bsm_mul launches.  `roofline` describes that kernel; `extra` adds driver-timed HBM-resident
    legs (a 1.1 GB C2-shaped VBCRS operator, the C3 fused symmetric product), a cold-cache figure and
    the CPU baselines.

N > 1 (one process per GPU, launched by torch.distributed.run, RCCL): the 8-GPU configuration the
    metric is quoted on in fp64 -- C5, SymmetricBlockMatrix 5M x 5M, mixed 16-256 block sizes (28.6 GB)
    -- STRONG-scaled: the diagonal segments are row-partitioned over the ranks (bsm_partition_rows),
    every rank generates its share in HBM (include/bsm_synth.h) and a step is the complete
    distributed product with x and y partitioned like the rows (what an iterative solver on N GPUs
    holds): point-to-point exchange of the x halo, fused A + A^T local product, point-to-point
    exchange + add of the partial-y segments that belong to the neighbouring rank (ncclSend/Recv over
    xGMI).  All of it is inside the timed region.  `extra.c4` reports C4 (VBCRS 2M x 2M, 128x128 fp32
    blocks, 16.4 GB, scattered block columns: all-gather of the x slices, then the local product).
    --workload c2 keeps the old weak-scaling C2 run.

metric value = algorithmic bytes of the whole job (every stored entry once + block
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
PMC_FILE = os.path.join(ROOT, "profiles", "r05_c2_pmc.json")  # written by tools/pmc_summary.py, stamped with the build id
MFMA_FILE = os.path.join(ROOT, "profiles", "r05_c4_mfma.json")
C5N1_FILE = os.path.join(ROOT, "profiles", "r05_c5_n1.json")  # the N > 1 workload on ONE GPU (tools/profile_round.sh)
ORACLE_SAMPLES = 48  # sampled block rows per rank of the full-size oracle check of the partitioned workloads
SETUP_STEPS = 30    # N > 1 / --workload c5: untimed steps of a freshly built operator before the W warm-up steps (config.setup_steps)
MIN_TIMED_S = 5e-3  # a timed region shorter than this is repeated (config.replays) so that host synchronisation stays below 1 %


def read_json(path):
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def counter_file(path, build, kernel=None, alg_bytes=None):
    """A rocprofv3 counter summary under profiles/ -- only if it was taken with THIS build of the kernels
    (and, where given, this kernel and byte count); else (None, why)."""
    d = read_json(path)
    if d is None:
        return None, "no counter file " + os.path.relpath(path, ROOT)
    if d.get("build") != build:
        return None, "%s was taken with build %s, this is %s" % (os.path.relpath(path, ROOT), d.get("build"), build)
    if kernel is not None and d.get("kernel") not in kernel:
        return None, "%s is for kernel %s" % (os.path.relpath(path, ROOT), d.get("kernel"))
    if alg_bytes is not None and d.get("alg_bytes_per_launch") != alg_bytes:
        return None, "%s is for %s algorithmic bytes per launch, this run has %d" % (os.path.relpath(path, ROOT), d.get("alg_bytes_per_launch"), alg_bytes)
    return d, None


def profiler_in_environment(env):
    """names what says that this process was started under rocprofv3 / rocprofiler (its preloaded tool library or its
    control variables), or None"""
    for k, v in env.items():
        if k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) or (k in ("LD_PRELOAD", "HSA_TOOLS_LIB") and "rocprof" in v.lower()):
            return k
    return None


def live_traffic(kernel_sub, alg_bytes):
    """HBM traffic of the C2 product measured NOW: two rocprofv3 passes (FETCH_SIZE, WRITE_SIZE -- one counter per
    pass, --kernel-trace only, MI355X_MICROARCH.md) over a child process that runs this script's C2 launches
    (--pmc-child); per-launch means, FETCH_SIZE KiB x 1024 x 2 (gfx950: the 128-B requests of 16-byte-per-lane
    streaming reads are tallied at 64 B) + WRITE_SIZE KiB x 1024.  (None, why) when rocprofv3 is not on the box or
    a pass fails: the caller falls back to the stamped file under profiles/."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 is not on this box"
    why = profiler_in_environment(os.environ)
    if why:  # this very process runs under a profiler: a child would inherit its preload and nest profilers
        return None, "not taken: bench.py itself runs under a profiler (%s)" % why
    means = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="bsm_pmc_", dir="/tmp")
        try:
            # (the program itself follows `--`: no shell, no env wrapper between the profiler and python)
            r = subprocess.run([exe, "--output-format", "csv", "--kernel-trace", "--pmc", counter, "-d", d, "-o", "p", "--",
                                sys.executable, os.path.abspath(__file__), "--pmc-child"], cwd="/tmp", env=env,
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=240)
            vals = []
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(path, newline="") as f:
                    for row in csv.DictReader(f):
                        if kernel_sub in row["Kernel_Name"] and row["Counter_Name"] == counter:
                            vals.append(float(row["Counter_Value"]))
            if r.returncode != 0 or len(vals) < 10:
                return None, "the %s pass gave %d dispatches (status %d): %s" % (counter, len(vals), r.returncode,
                                                                                r.stderr.decode(errors="replace")[-200:])
            means[counter] = (sum(vals) / len(vals), len(vals))
        except Exception as e:  # noqa: BLE001
            return None, "the %s pass failed: %r" % (counter, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    traffic = int(means["FETCH_SIZE"][0] * 1024 * 2 + means["WRITE_SIZE"][0] * 1024)
    return {"traffic_bytes_per_launch": traffic, "fetch_size_kib_mean": round(means["FETCH_SIZE"][0], 2),
            "write_size_kib_mean": round(means["WRITE_SIZE"][0], 2), "dispatches": [means["FETCH_SIZE"][1], means["WRITE_SIZE"][1]],
            "traffic_over_algorithmic": round(traffic / alg_bytes, 4)}, None


def pmc_child():
    """what live_traffic() profiles: 60 plain C2 launches (no graph: every dispatch gets its counter record)"""
    import torch
    import bsm_amd as bsm
    prob = bsm.synthetic.config2()
    A = bsm.VariableBlockCompressedRowStorage(prob["blocks"], prob["rowstart"], prob["colstart"], prob["size"])
    x = torch.from_numpy(prob["x"]).cuda()
    y = torch.zeros_like(x)
    plan = bsm.MulPlan(y, A, x)
    for _ in range(60):
        plan()
    torch.cuda.synchronize()


def graph_timed(fn, steps, torch):
    """Device time per call of `fn` when `steps` calls are captured into ONE hipGraph and replayed (HIP
    events around the replay, on the stream it runs on); None when capture is not available."""
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(steps):
                    fn()
        torch.cuda.current_stream().wait_stream(s)
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e-3 / steps
    except Exception:  # pragma: no cover
        return None


class StreamFloor:
    """Bare streaming read of a buffer (include/bsm_synth.h: bsm_bench_stream): the product kernels' request
    shape and nothing else -- what the memory system gives a launch of that size."""

    def __init__(self, torch, nbytes):
        import ctypes as C
        from bsm_amd import _lib
        self.C, self.torch, self.L = C, torch, _lib.lib()
        self.nbytes = int(nbytes) // 16 * 16
        self.buf = torch.zeros(self.nbytes // 8, dtype=torch.float64, device="cuda")
        nwaves = ((self.nbytes // 16 + 2047) // 2048) * 4
        self.scratch = torch.zeros((8192 + 64 * nwaves) // 8, dtype=torch.float64, device="cuda")
        self(1)  # uploads the record table of the hop variant
        torch.cuda.synchronize()

    def __call__(self, hop=0):
        C = self.C
        rc = self.L.bsm_bench_stream(C.c_void_p(self.buf.data_ptr()), self.nbytes, C.c_void_p(self.scratch.data_ptr()),
                                     self.scratch.numel() * 8, hop, C.c_void_p(self.torch.cuda.current_stream().cuda_stream))
        if rc:
            raise RuntimeError("bsm_bench_stream failed")


def cold_median(fn, flush, reps, torch):
    """median device time of fn() right after flush() (every cache holds something else)"""
    ts = []
    for _ in range(reps):
        flush()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    ts.sort()
    return ts[len(ts) // 2]


def timed(fn, reps, torch):
    """HIP-event time per call of `fn` over `reps` back-to-back calls on the current stream."""
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / reps


def leg(bsm, torch, prob, reps, multi_rhs=0, **kw):
    """One driver-timed single-GPU leg: warm device-event time of mul!(y, A, x).
    multi_rhs = K: also mul!(Y, A, X) with K right-hand sides (bsm_mul_multi: A streamed once per batch of <= 8),
    its time in single products and its worst column against K single products."""
    A = bsm.synthetic.build(prob, **kw)
    st = A.stats()
    x = prob["x"]
    y = torch.full((prob["size"][0],), float("nan"), dtype=x.dtype, device="cuda")
    plan = bsm.MulPlan(y, A, x)
    for _ in range(30):  # (the first tens of launches of a freshly built operator run 5-8 % slower: clocks, TLBs)
        plan()
    torch.cuda.synchronize()
    t = sorted(timed(plan, reps, torch) for _ in range(3))[1]  # median of three batches of `reps` launches
    out = {"us": round(t * 1e6, 2), "GBps": round(st["alg_bytes"] / t / 1e9, 1),
           "frac_of_hbm_peak": round(st["alg_bytes"] / t / 1e9 / HBM_PEAK_GBPS, 4),
           "alg_MB": round(st["alg_bytes"] / 1e6, 1), "device_MB": round(st["device_bytes"] / 1e6, 1)}
    # the same products as ONE captured hipGraph of `reps` launches (how the C2 headline is launched): without the host
    # issuing every `y .*= beta` and product launch of its own
    tg = graph_timed(plan, reps, torch)
    if tg is not None:
        out["graph_us"] = round(tg * 1e6, 2)
        out["graph_frac_of_hbm_peak"] = round(st["alg_bytes"] / tg / 1e9 / HBM_PEAK_GBPS, 4)
    if st.get("win_emissions"):  # fused symmetric launch: share of the y contributions that leave a CU as global atomics
        out["y_contributions_as_atomics"] = round((st["win_emissions"] - st["win_inside"] + st["win_flushed"]) / st["win_emissions"], 3)
    for K in ([multi_rhs] if isinstance(multi_rhs, int) else list(multi_rhs)) if multi_rhs else []:
        key = "multi_rhs" if K == 8 else f"multi_rhs_{K}"
        n = x.shape[0]
        X = torch.empty((K, n), dtype=x.dtype, device="cuda").t()  # column-major n x K
        for k in range(K):
            X[:, k] = x * (k + 1) / K
        Y = torch.full((K, n), float("nan"), dtype=x.dtype, device="cuda").t()
        many = lambda: bsm.mul(Y, A, X)
        for _ in range(30):
            many()
        torch.cuda.synchronize()
        tk = sorted(timed(many, max(reps // 3, 5), torch) for _ in range(3))[1]
        worst = 0.0
        for k in range(K):  # the reference semantics: LinearMaps applies _unsafe_mul! column by column
            bsm.mul(y, A, X[:, k].contiguous())
            worst = max(worst, float((Y[:, k] - y).abs().max() / y.abs().max()))
        plan()  # (y back to the single product of x)
        torch.cuda.synchronize()
        out[key] = {"nrhs": K, "us": round(tk * 1e6, 2), "single_products": round(tk / t, 3),
                    "relerr_vs_single_products": worst}
        # (csrc/bsm_kernels.hip: panel_kernel_il -- the interleaved pass wherever the product accumulates with atomics;
        # exclusive VBCRS products keep kMfmaReal / the vector kernels; counters: profiles/r05_il_counters.txt)
        mf = "f64" if x.dtype in (torch.complex128, torch.float64) else "f32"
        if x.dtype.is_complex and K == 8:
            out[key]["pipe"] = "matrix pipe: 8 complex columns = N = 16 of v_mfma_%s_16x16x4; X and Y row-major in work arrays (interleaved pass)" % mf
        if not x.dtype.is_complex and K in (8, 16):
            out[key]["pipe"] = "matrix pipe: %d real columns in N = 16 of v_mfma_%s_16x16x4 (interleaved pass where the product accumulates with atomics)" % (K, mf)
        del X, Y
    del plan, A
    return out, y


def self_launch(args):
    """Runs this script as N rank processes (python -m torch.distributed.run, one per GPU) and prints the
    JSON line of rank 0.  If the RCCL run dies or prints no line, ONE more attempt is made with the
    exchanges over gloo (host-staged) and the line says so: a slow measured line instead of none."""
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # this pool's driver only supports dmabuf IPC
    forwarded = [a for a in sys.argv[1:]]

    def attempt(extra, limit):
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + forwarded + extra
        # the launcher and its ranks form their own process group: on a timeout the WHOLE group is ended (killing
        # only the launcher would leave the ranks on the GPUs while the retry starts)
        import signal
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True)
        try:
            raw, _ = proc.communicate(timeout=limit)
            rc, out = proc.returncode, raw.decode(errors="replace")
        except subprocess.TimeoutExpired:
            for sig, grace in ((signal.SIGTERM, 10), (signal.SIGKILL, 20)):
                try:
                    os.killpg(proc.pid, sig)
                except ProcessLookupError:
                    break
                try:
                    proc.wait(timeout=grace)
                    break
                except subprocess.TimeoutExpired:
                    continue
            try:
                raw, _ = proc.communicate(timeout=10)
            except Exception:
                raw = b""
            rc, out = 124, (raw or b"").decode(errors="replace")
        line = None
        for ln in out.splitlines():
            ln = ln.strip()
            if ln.startswith("{") and '"metric"' in ln:
                line = ln
        return rc, line, out

    # both attempts together stay inside the driver's own limit for one bench run (25 minutes)
    rc, line, out = attempt([], 780)
    if line is None and args.backend == "nccl":
        print(f"[bench] the {args.gpus}-rank RCCL run ended with status {rc} and no result line; "
              "one more attempt with the exchanges over gloo", file=sys.stderr, flush=True)
        sys.stderr.write(out[-4000:])
        rc, line, out = attempt(["--backend", "gloo", "--note", f"the RCCL run ended with status {rc} before its line"], 600)
    if line is None:
        sys.stdout.write(out)
        return rc or 1
    print(line, flush=True)
    return (rc or 3) if '"value_invalid": true' in line else 0


class stdout_to_stderr:
    """RCCL prints a version banner ("RCCL version : ...", five lines) on STDOUT when its first communicator is created; this
    script's stdout carries ONE JSON line.  File descriptor 1 points at stderr while a process group is being set up."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def init_one_rank(backend, torch, dev_index):
    """torch.distributed with ONE rank (the loopback rehearsal): own rendezvous on 127.0.0.1"""
    import datetime
    import socket
    import torch.distributed as dist
    if dist.is_initialized():
        return dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "MASTER_PORT" not in os.environ:
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        sk.close()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kw = {}
    if backend == "nccl":
        from bsm_amd import distributed as D
        kw = {"device_id": torch.device("cuda", dev_index), "pg_options": D.nccl_options()}
    with stdout_to_stderr():
        dist.init_process_group(backend, rank=0, world_size=1, timeout=datetime.timedelta(minutes=5), **kw)
        if backend == "nccl":  # (the communicator -- and its banner -- may be created lazily, at the first collective)
            dist.barrier()
            torch.cuda.synchronize()
    return dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", choices=["auto", "c2", "c5"], default="auto",
                    help="auto: C2 at N = 1, strong-scaled C5 (+ C4 in `extra`) at N > 1")
    ap.add_argument("--launch", choices=["graph", "eager"], default="graph",
                    help="graph (N = 1): the K timed steps are one hipGraph replay of K bsm_mul launches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the additional legs in `extra`")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to "
                    "rehearse the N > 1 code path on a single GPU)")
    ap.add_argument("--device", type=int, default=None, help="override LOCAL_RANK as the HIP device (rehearsal)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the N > 1 operators (rehearsal on one GPU)")
    ap.add_argument("--note", default=None, help="free text carried into config.note (set by the self-launcher)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: strictly serial exchange / product / exchange "
                    "(the round-2 step) instead of the overlapped one")
    ap.add_argument("--no-live-pmc", action="store_true", help="take roofline.traffic from the stamped file under profiles/ "
                    "instead of two rocprofv3 passes of this run")
    ap.add_argument("--loopback", action="store_true", help="N = 1, --workload c5: initialise the backend (nccl = RCCL) with ONE rank and run "
                    "the N > 1 step for real -- the rank owns all but the first and last 8 diagonal segments, a phantom neighbour (this "
                    "same process) the rest, so the x halo and the partial-y halo travel through grouped self send / recv beside the "
                    "interior launch (distributed.RowPartitioned(loopback=...)): the only way a one-GPU box executes RCCL")
    ap.add_argument("--no-oracle-check", action="store_true", help="--workload c5 / N > 1: skip the sampled-row oracle check of the last timed step")
    ap.add_argument("--no-anchor", action="store_true", help="N > 1: do not let rank 0 measure the same workload alone afterwards (n1_same_workload)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: the parallel entry is INSIDE the call, like the reference's
        # `@tasks` fan-out (src/vbcrs.jl:275-276, src/symmetricblockmatrix.jl:395-432).  Nothing has
        # touched the GPU in this process: start N fresh rank processes and relay rank 0's line.
        raise SystemExit(self_launch(args))

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
    # comm: the process group the data path and the scalar reductions use.  `fallback` is a gloo group
    # over the same ranks: if the first RCCL exchange of a workload RAISES on any rank (this path has
    # never met real multi-GPU hardware before the driver's run), every rank switches to it together
    # and the JSON line says so -- a slow measured line instead of none.
    comm = {"group": None, "dev": "cuda", "name": args.backend, "fallback": None}
    if world == 1 and args.loopback:
        dist = init_one_rank(args.backend, torch, dev_index)
        if args.backend != "nccl":
            comm["dev"] = "cpu"
    if world > 1:
        import datetime
        import torch.distributed as dist
        with stdout_to_stderr():
            if args.backend == "nccl":
                from bsm_amd import distributed as D
                dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), pg_options=D.nccl_options(),
                                        timeout=datetime.timedelta(minutes=5))
            else:
                dist.init_process_group(args.backend)
                comm["dev"] = "cpu"
            comm["fallback"] = dist.new_group(backend="gloo", timeout=datetime.timedelta(minutes=5))
        seen = [None] * world
        dist.all_gather_object(seen, "rank %d: cuda:%d %s" % (rank, dev_index, torch.cuda.get_device_name(dev_index)),
                               group=comm["fallback"])
        comm["devices"] = seen
    workload = args.workload
    if workload == "auto":
        workload = "c2" if world == 1 else "c5"

    def barrier():
        if dist is not None:
            dist.barrier(group=comm["group"])
        torch.cuda.synchronize()

    def reduce_scalars(elapsed, nbytes):
        if dist is None:
            return elapsed, float(nbytes)
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm["dev"])
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=comm["group"])
        b = torch.tensor([float(nbytes)], dtype=torch.float64, device=comm["dev"])
        dist.all_reduce(b, op=dist.ReduceOp.SUM, group=comm["group"])
        return float(t.item()), float(b.item())
    reduce_scalars.comm = comm

    if workload == "c5":
        out = run_partitioned(args, bsm, torch, dist, np, rank, world, barrier, reduce_scalars)
    else:
        out = run_c2(args, bsm, torch, dist, np, rank, world, barrier, reduce_scalars)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier(group=comm["group"])
        dist.destroy_process_group()
    if out.get("value_invalid"):
        raise SystemExit("bench: the parity check of the distributed product failed (config.parity_relerr)")


# ------------------------------------------------------------------------------------------------
# N > 1: strong-scaled C5 (fused symmetric + halo + all-gather), C4 in `extra`
# ------------------------------------------------------------------------------------------------
def run_partitioned(args, bsm, torch, dist, np, rank, world, barrier, reduce_scalars):
    from bsm_amd import distributed as D
    S = bsm.synthetic
    comm = reduce_scalars.comm
    loopback = bool(getattr(args, "loopback", False)) and world == 1
    solo = dist is None  # (a process group may exist all the same: rank 0 measuring the N = 1 anchor after an N-rank run)

    def c5_share():
        n = int(5_000_000 * args.scale)
        start, sz = S.config5_segments(n=n)
        nseg = len(sz)
        halfband = 4
        # stored entries per diagonal segment: its diagonal block + the off-diagonal blocks of its rows.
        # All blocks of a segment share one row key, so this is bsm_partition_rows on the block list.
        w = sz.astype(np.float64) ** 2
        for k in range(1, halfband + 1):
            w[k:] += sz[k:] * sz[:-k]
        part, own = bsm.partition_rows(n, start + 1, w.astype(np.int64), world)
        segs = np.nonzero(part == rank)[0]
        lo, hi = (int(segs[0]), int(segs[-1]) + 1) if len(segs) else (0, 0)
        prob = S.config5(n=n, on_device=True, seg_lo=lo, seg_hi=hi)
        if loopback:
            # one rank plays a MIDDLE rank of a node: it owns all but the first and last `edge` segments, the phantom
            # neighbour (this process) the rows of those -- their blocks are this rank's boundary blocks
            edge = min(8, max(nseg // 4, 1))
            lb_own = (int(start[edge]) + 1, int(start[nseg - edge]))
            return prob, lb_own, True, "C5: SymmetricBlockMatrix %dx%d, segments U{16..256}, off-diagonal (I,J) J=I-1..I-4, fp64" % (n, n), (lambda: prob), \
                (lambda: S.config5_sample(S.sample_ids(ORACLE_SAMPLES, nseg), n=n)[:2])

        def verification():
            # everything that contributes to this rank's rows, built WITHOUT any exchange: the rank's own
            # segments plus the `halfband` segments behind them (their off-diagonal blocks reach back into
            # the rank's rows through B^T), applied to the full x
            ext = S.config5(n=n, on_device=True, seg_lo=hi, seg_hi=min(hi + halfband, nseg)) if hi < nseg else None
            ver = dict(prob)
            if ext is not None:
                for k in ("diagonals", "diagonalindices", "offdiagonals", "rowindices", "colindices"):
                    ver[k] = list(prob[k]) + list(ext[k])
            return ver
        # the rank's rows against the ORACLE on sampled diagonal segments (blocks regenerated on the host, bit-identical)
        sampler = (lambda: S.config5_sample(S.sample_ids(ORACLE_SAMPLES, nseg, lo=lo, hi=hi), n=n)[:2]) if hi > lo else None
        return prob, own[rank], True, "C5: SymmetricBlockMatrix %dx%d, segments U{16..256}, off-diagonal (I,J) J=I-1..I-4, fp64" % (n, n), verification, sampler

    def c4_share():
        ngrid = max(world, int(15625 * args.scale))
        lo, hi = rank * ngrid // world, (rank + 1) * ngrid // world  # uniform blocks: equal row counts == equal bytes
        prob = S.config4(ngrid=ngrid, on_device=True, row_lo=lo, row_hi=hi)
        own = (lo * 128 + 1, hi * 128)
        if loopback:
            edge = min(8, max(ngrid // 4, 1))
            own = (edge * 128 + 1, (ngrid - edge) * 128)
        sampler = (lambda: S.config4_sample(S.sample_ids(ORACLE_SAMPLES, ngrid, lo=lo, hi=hi), ngrid=ngrid)[:2]) if hi > lo else None
        return prob, own, False, "C4: VBCRS %dx%d, %d 128x128 fp32 blocks (16 per block row)" % (ngrid * 128, ngrid * 128, ngrid * 16), (lambda: prob), sampler

    def run(share, steps, warmup, overlap):
        # the rank's products run on a stream whose CU mask leaves one CU per XCD to RCCL's send / recv kernels (they
        # find no free CU beside a launch that fills the chip and would finish when the launch does: DESIGN.md 5b)
        reserve = int(os.environ.get("BSM_RESERVE_CUS", "8")) if (overlap and dist is not None and comm["name"] == "nccl") else 0
        if reserve:
            try:
                cs = D.compute_stream(reserve=reserve)
                torch.cuda.synchronize()
                prev_stream = torch.cuda.current_stream()
                torch.cuda.set_stream(cs)
            except Exception as e:  # (a stream with a CU mask is an optimisation: without it the step is measured all the same)
                print("[bench] no compute stream with reserved CUs (%r): the products run on the current stream" % (e,), file=sys.stderr, flush=True)
                reserve = 0
        try:
            r = run_on_current_stream(share, steps, warmup, overlap)
        finally:
            if reserve:
                torch.cuda.synchronize()
                torch.cuda.set_stream(prev_stream)
        r["reserved_cus"] = reserve
        return r

    def run_on_current_stream(share, steps, warmup, overlap):
        t0 = time.perf_counter()
        prob, own, sym, desc, verification, sampler = share()
        n = prob["size"][0]
        x = prob["x"]
        es = x.element_size()
        x_full = x.clone()  # the parity check at the end multiplies with the whole x, without any exchange
        if overlap:
            # interior blocks (read x[own], write y[own]) and boundary blocks as two handles: the exchanges
            # and the boundary product run on a side stream beside the interior launch
            P = D.build_overlapped(prob, own, group=comm["group"], symmetric=sym, xmode="halo" if sym else "allgather",
                                   loopback=own if loopback else None, solo=solo)
            handles = [h for h in (P.interior, P.local) if h is not None]
        else:
            touched = D._touched(own, prob["colindices"]) if sym else own
            if loopback:  # the rank holds every block; rows outside `own` are the phantom's
                touched = (1, n)
            A = D.build_local(prob, touched)
            P = D.RowPartitioned(A, own, touched, group=comm["group"], gather=False, symmetric=sym,
                                 xneed=touched if sym else None, loopback=own if loopback else None, solo=solo)
            handles = [A] if A is not None else []
        t_setup = time.perf_counter() - t0
        if not handles:
            raise SystemExit("bench: a rank received no block (more ranks than block rows)")
        sts = [h.stats() for h in handles]
        # this rank's part of the algorithmic bytes: stored entries + index metadata (x and y are
        # counted once for the whole job below)
        rank_bytes = sum(st["alg_bytes"] - 2 * n * es for st in sts)
        torch.cuda.empty_cache()
        # x and y stay PARTITIONED like the rows (what an iterative solver on N GPUs holds): outside its
        # own range a rank's x is NaN until the exchange of the step has filled what its blocks read
        if world > 1:
            keep = x[own[0] - 1:own[1]].clone()
            x.fill_(float("nan"))
            x[own[0] - 1:own[1]] = keep
        y = torch.full((n,), float("nan"), dtype=x.dtype, device="cuda")

        def step():
            # symmetric: x halo send/recv -> fused local product -> partial-y halo send/recv + add;
            # VBCRS (scattered columns): all-gather of the x slices -> local product
            if overlap:
                P.mul_overlapped(y, x)
            else:
                P.mul(y, x, x_distributed=True)

        if dist is not None and comm["fallback"] is not None and comm["group"] is None:
            # first exchange of this workload, guarded (see `comm` in main): every rank reports over gloo
            # that it has reached this point in good health BEFORE anybody enters the RCCL exchange (a rank
            # that died during set-up must not leave its peers waiting inside a collective)
            ok = torch.tensor([1.0], dtype=torch.float64)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=comm["fallback"])
            err = None
            try:
                if os.environ.get("BSM_BENCH_FAIL_FIRST"):  # rehearsal of the fallback (every rank raises:
                    # a rank that raised while its peers are inside the exchange would leave them waiting)
                    raise RuntimeError("simulated failure of the first exchange (BSM_BENCH_FAIL_FIRST)")
                step()
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001 -- anything the collective layer throws
                err = repr(e)
            ok = torch.tensor([0.0 if err else 1.0], dtype=torch.float64)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=comm["fallback"])
            if ok.item() < 1.0:
                comm.update(group=comm["fallback"], dev="cpu",
                            name="gloo (the %s exchange raised on a rank%s)" % (args.backend, ": " + err[:200] if err else ""))
                P.group = comm["group"]
                P._plan = P._xplan = P._ranges = None

        # set-up: a freshly built operator runs its first tens of launches 5-8 % slower (clocks, TLBs: DESIGN.md section 4;
        # per-step times of the first 30 steps in profiles/r05_loopback_first_steps.txt) -- SETUP_STEPS untimed steps of the
        # same kind first, as the single-GPU legs do with 30 launches; then the W warm-up steps and EXACTLY K timed ones
        for _ in range(SETUP_STEPS):
            step()
        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
        chk = (1, n) if loopback else own  # loopback: the phantom's rows are this process's too -- every row is checked
        y_step = y[chk[0] - 1:chk[1]].clone()
        # the local kernels alone (HIP events on the stream they are launched on), for the roofline object
        yk = torch.zeros_like(y)
        plans = [bsm.MulPlan(yk, h, x) for h in handles]

        def local_only():
            for pl in plans:
                pl()
        for _ in range(3):
            local_only()
        torch.cuda.synchronize()
        # (median of three batches: exchange_us is the difference of two millisecond figures)
        kdur = sorted(timed(local_only, max(5, steps // 4), torch) for _ in range(3))[1]
        del plans, yk
        elapsed, total = reduce_scalars(elapsed, rank_bytes)
        total += 2 * n * es
        kmax, _ = reduce_scalars(kdur, 0)
        # parity of the distributed result: this rank's y slice against a product that needs NO exchange --
        # every block that contributes to the rank's rows in one ordinary handle, applied to the full x
        del P, handles
        torch.cuda.empty_cache()
        Aver = S.build(verification())
        yv = torch.full((n,), float("nan"), dtype=x.dtype, device="cuda")
        bsm.mul(yv, Aver, x_full)
        ref = yv[chk[0] - 1:chk[1]]
        scale = float(ref.abs().max().item())
        diff = float((y_step - ref).abs().max().item())
        rel = diff / scale if scale > 0 else float("nan")
        if not (rel == rel):  # NaN: a missing x entry or an undefined y row
            rel = float("inf")
        del Aver, yv
        pmax, _ = reduce_scalars(rel, 0)
        # ... and against the CPU ORACLE (the checker, after every timed figure is final): the operator is generated in HBM
        # bit-identically to the numpy streams, so the host regenerates only the blocks reaching ORACLE_SAMPLES sampled block
        # rows of this rank, runs the reference loop restatement on that sub-problem with the full x, and this rank's y of
        # the last timed step is compared on those rows (max |dy| / max |y| over them; max over ranks)
        osamp = float("nan")
        if sampler is not None and not getattr(args, "no_oracle_check", False):
            from oracle import load_oracle
            orc = None
            if rank == 0:  # rank 0 first: builds oracle/libbsm_oracle.so if the tree has none
                try:
                    orc = load_oracle()
                except Exception as e:  # pragma: no cover
                    print("[bench] the oracle could not be loaded: %r" % (e,), file=sys.stderr, flush=True)
            if dist is not None and world > 1:
                dist.barrier(group=comm["fallback"] if comm["fallback"] is not None else comm["group"])
            try:
                if orc is None:
                    orc = load_oracle(build=False)
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                from _common import N as OPN, oracle_mul, sampled_relerr
                sub, ranges = sampler()
                yref = oracle_mul(orc, sub, OPN, sub["x"], np.zeros(n, sub["x"].dtype))
                yfull = np.full(n, np.nan, sub["x"].dtype)
                yfull[chk[0] - 1:chk[1]] = y_step.cpu().numpy()
                osamp = sampled_relerr(yfull, yref, ranges)
                del yfull, yref, sub
            except Exception as e:  # pragma: no cover
                print("[bench] sampled oracle check failed to run: %r" % (e,), file=sys.stderr, flush=True)
            osamp_max, _ = reduce_scalars(osamp if osamp == osamp else float("inf"), 0)
        else:
            osamp_max = None
        return dict(oracle_sampled=osamp_max, desc=desc, elapsed=elapsed, total_bytes=total, kdur=kmax, rank_alg=rank_bytes + 2 * n * es // world,
                    setup_s=t_setup, parity=pmax, n=n, own=own, overlap=overlap,
                    exchange_us=(elapsed / steps - kmax) * 1e6)

    def run_checked(share, steps, warmup, tol):
        r = run(share, steps, warmup, not args.no_overlap)
        if r["overlap"] and not (r["parity"] <= tol):
            # the overlapped step gave a wrong y on some rank (every rank sees the same maximum): measure the
            # strictly serial step instead and say so -- a wrong y must not pass as a measurement
            first = r["parity"]
            torch.cuda.empty_cache()
            r = run(share, steps, warmup, False)
            r["note"] = "the overlapped step FAILED its parity check (rel-err %.3g); this line is the serial step" % first
        return r

    steps, warmup = args.steps, args.warmup
    tol = 1e-12
    r5 = run_checked(c5_share, steps, warmup, tol)
    value = r5["total_bytes"] * steps / r5["elapsed"] / 1e9
    extra = {}
    if not args.no_extra:
      try:  # (an additional figure must never cost the headline line)
        torch.cuda.empty_cache()
        r4 = run_checked(c4_share, steps, warmup, 1e-5)
        extra["c4"] = {"workload": r4["desc"] + ", rows and vectors partitioned over %d GPUs, RCCL all-gather of the x slices "
                                                "(beside the product of the blocks that read own x entries), then the rest" % world,
                       "dtype": "f32", "value": round(r4["total_bytes"] * steps / r4["elapsed"] / 1e9, 1), "unit": "GB/s",
                       "ms_per_step": round(r4["elapsed"] / steps * 1e3, 4),
                       "local_kernel_us_max": round(r4["kdur"] * 1e6, 1),
                       "exchange_us": round(r4["exchange_us"], 1),
                       "parity_relerr": r4["parity"], "parity_tol": 1e-5, "overlap": r4["overlap"],
                       "parity_vs_oracle_sampled": r4["oracle_sampled"],
                       "frac_of_hbm_peak": round(r4["total_bytes"] * steps / r4["elapsed"] / 1e9 / (HBM_PEAK_GBPS * world), 4),
                       "setup_s": round(r4["setup_s"], 2)}
      except Exception as e:  # pragma: no cover
        extra["c4"] = {"error": repr(e)}
        if dist is not None:
            raise  # a rank that skipped collectives would hang the others: fail loudly instead
    kb = r5["rank_alg"]
    roofline = {"bound": "hbm", "achieved": round(kb / r5["kdur"] / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(kb / r5["kdur"] / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                "kernel": "bsm::panel_kernel<double,8,true,true,true> (fused A + A^T, one rank's share: interior launch + boundary launch)",
                "alg_bytes_per_launch": int(kb), "avg_launch_us": round(r5["kdur"] * 1e6, 2),
                "note": "slowest rank's local launches alone (HIP events, back to back on one stream); a step adds the "
                        "x-halo and partial-y-halo exchanges, overlapped with the interior launch (exchange_us = step - this)"}
    ranks_seen = world
    backend = args.backend
    if dist is not None:
        ranks_seen = dist.get_world_size(comm["group"])
        backend = dist.get_backend(comm["group"])
    out = {
        "metric": "fp64 block-SpMV GB/s (SymmetricBlockMatrix mul!, algorithmic bytes / time)",
        "value": round(value, 1), "unit": "GB/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(r5["elapsed"] / steps * 1e3, 6), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": r5["desc"] + ": mul!(y, A, x) with A, x and y row-partitioned over %d GPUs; a step = "
                               "RCCL send/recv of the x halo, fused A + A^T local product (interior blocks beside the "
                               "exchange, boundary blocks after it), RCCL send/recv + add of the partial-y halo (the "
                               "overlapping y segments), all in the timed region; operator generated in HBM "
                               "(SplitMix64 seed 0xB5A5)" % world,
                   "global_rows": r5["n"], "alg_bytes_total": int(r5["total_bytes"]),
                   "partition": "diagonal segments by stored bytes (bsm_partition_rows)",
                   "collectives": "ncclSend/ncclRecv (x halo, partial-y halo)" if comm["name"] == "nccl" else comm["name"],
                   "backend": backend, "ranks": ranks_seen, "devices": comm.get("devices"),
                   "overlap": r5["overlap"], "reserved_cus": r5.get("reserved_cus", 0), "setup_steps": SETUP_STEPS,
                   "parity_relerr": r5["parity"], "parity_tol": tol,
                   "parity_check": "every rank: its y slice of the last timed step vs an exchange-free product of all blocks "
                                   "reaching its rows with the full x (max |dy| / max |y|, max over ranks)",
                   "parity_vs_oracle_sampled": r5["oracle_sampled"],
                   "parity_vs_oracle_sampled_check": "every rank: its y of the last timed step vs oracle/bsm_oracle.c (orc_sym_mul, the reference's "
                                                     "three sweeps) on %d sampled diagonal segments of its rows -- the blocks reaching them regenerated "
                                                     "on the host from the same SplitMix64 streams (synthetic.config5_sample); max over ranks" % ORACLE_SAMPLES,
                   "exchange_us": round(r5["exchange_us"], 1),
                   "local_kernel_us_max": round(r5["kdur"] * 1e6, 1),
                   "frac_of_hbm_peak": round(value / (HBM_PEAK_GBPS * world), 4)},
        "roofline": roofline,
    }
    notes = [t for t in (args.note, r5.get("note")) if t]
    if loopback:
        out["config"]["loopback"] = {
            "what": "ONE rank over the %s backend playing a middle rank of a node: it owns rows %d..%d (all but the first and last 8 "
                    "diagonal segments), a phantom neighbour -- this same process -- the rest; the x halo and the partial-y halo of "
                    "the boundary blocks travel through grouped self send / recv (batch_isend_irecv on device tensors) on the side "
                    "stream beside the interior launch; every row of y is checked against an ordinary single-handle product" % (backend, r5["own"][0], r5["own"][1]),
            "librccl_loaded": bool([ln for ln in open("/proc/self/maps") if "librccl" in ln])}
    if world > 1:
        # The SAME operator on ONE GPU through the same code path -- what this line's value has to be divided by for a
        # speed-up (the driver's own N = 1 run times C2: another operator, Infinity-Cache-resident).  Sources, best first:
        #   1. measured NOW: rank 0 alone, on its own GPU, after the N-rank run (same box, same build; the other ranks wait
        #      at a host-side barrier); the whole operator (29 GB at the full size) fits one GPU;
        #   2. the driver-timed extra.c5_n1 of the newest BENCH_r*.json in the tree;
        #   3. the committed profiles/r0N_c5_n1.json.
        from bsm_amd import _lib
        build = _lib.lib().bsm_version().decode().split("build ")[-1]
        anchor = None
        if not getattr(args, "no_anchor", False):
            wait_group = comm["fallback"] if comm["fallback"] is not None else comm["group"]
            if rank == 0:
                try:
                    torch.cuda.empty_cache()
                    a1 = argparse.Namespace(**vars(args))
                    a1.no_extra, a1.no_overlap, a1.note, a1.loopback, a1.no_anchor = True, False, None, False, True
                    a1.steps, a1.warmup = max(args.steps, 10), max(args.warmup, 3)

                    def bar1():
                        torch.cuda.synchronize()

                    def red1(elapsed, nbytes):
                        return elapsed, float(nbytes)
                    red1.comm = {"group": None, "dev": "cuda", "name": "none", "fallback": None}
                    o1 = run_partitioned(a1, bsm, torch, None, np, 0, 1, bar1, red1)
                    if not o1.get("value_invalid"):
                        anchor = {"GBps": o1["value"], "ms_per_step": o1["ms_per_step"], "steps": a1.steps, "build": build, "same_build": True,
                                  "parity_vs_oracle_sampled": o1["config"]["parity_vs_oracle_sampled"],
                                  "source": "measured in THIS run: rank 0 alone on its GPU after the %d-rank run, same box, same build, same code "
                                            "path (python bench.py --gpus 1 --workload c5)" % world}
                except Exception as e:  # pragma: no cover
                    print("[bench] the live N = 1 anchor failed: %r" % (e,), file=sys.stderr, flush=True)
                torch.cuda.empty_cache()
            dist.barrier(group=wait_group)
        if anchor is None and args.scale == 1.0 and rank == 0:
            import glob
            for path in sorted(glob.glob(os.path.join(ROOT, "BENCH_r*.json")), reverse=True):
                d = read_json(path) or {}
                c5 = ((d.get("parsed") or d).get("extra") or {}).get("c5_n1") or {}
                if c5.get("value"):
                    anchor = {"GBps": c5["value"], "ms_per_step": c5.get("ms_per_step"), "build": c5.get("build"),
                              "same_build": c5.get("build") == build,
                              "source": "driver-timed extra.c5_n1 of %s (an earlier round's driver run)" % os.path.basename(path)}
                    break
            if anchor is None:
                n1 = read_json(C5N1_FILE)
                if n1 and n1.get("value"):
                    anchor = {"GBps": n1["value"], "ms_per_step": n1.get("ms_per_step"), "build": n1.get("build"),
                              "same_build": n1.get("build") == build, "source": "committed " + os.path.relpath(C5N1_FILE, ROOT)}
        # top-level, so that a reader needs no knowledge of `config`
        out["n1_same_workload"] = anchor
        out["speedup_vs_n1_same_workload"] = round(value / anchor["GBps"], 3) if anchor else None
        out["config"]["n1_same_workload_GBps"] = anchor["GBps"] if anchor else None
        out["config"]["n1_same_workload"] = anchor
        out["config"]["speedup_vs_n1_same_workload"] = out["speedup_vs_n1_same_workload"]
        notes.append("the default N = 1 line of this script times C2 (BASELINE.json's 1-GPU configuration: another operator, "
                     "Infinity-Cache-resident); the N = 1 figure of THIS workload is n1_same_workload (its `source` says where it comes "
                     "from) and speedup_vs_n1_same_workload divides by it")
    if notes:
        out["config"]["note"] = "; ".join(notes)
    if extra:
        if "c4" in extra and "note" in locals().get("r4", {}):
            extra["c4"]["note"] = r4["note"]
        out["extra"] = extra
    if r5["oracle_sampled"] is not None and not (r5["oracle_sampled"] <= tol):
        out["config"]["oracle_check_failed"] = True
        out["value_invalid"] = True
    if not (r5["parity"] <= tol):
        # a wrong y must not pass as a measurement: the line is printed for diagnosis, flagged, and the run fails
        out["config"]["parity_failed"] = True
        out["value_invalid"] = True
    return out


# ------------------------------------------------------------------------------------------------
# N = 1 (and --workload c2 at N > 1: weak scaling, no collective): C2
# ------------------------------------------------------------------------------------------------
BEM_TYPES = (("c128", "complex128", "full", 1e-12), ("f64", "float64", "real", 1e-12),
             ("c64", "complex64", "full", 1e-5), ("f32", "float32", "real", 1e-5))


def bem_tiled_problem(torch, np, K=400, dtype="complex128", part="full"):
    """The reference's own workload: its BEM test fixture (test/assets/symmetricblockexamples.jld2 "cuboid",
    decoded to tests/golden/symmetric_cuboid.bin: ComplexF64, 96 leaves of 3-28 rows, wide near-field panels
    with scattered columns) tiled K times along the diagonal -- the true block shapes at a size that
    streams from HBM (SURVEY.md 8d, C3').  The 188 blocks are uploaded once and referenced K times."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _common import fixture_problem
    dtype = np.dtype(dtype)
    p = fixture_problem("cuboid", dtype, part)
    n0 = p["size"][0]
    dd = [torch.from_numpy(np.ascontiguousarray(b.T)).cuda().t() for b in p["diagonals"]]   # column-major on the GPU
    oo = [torch.from_numpy(np.ascontiguousarray(b.T)).cuda().t() for b in p["offdiagonals"]]
    tile = lambda lists: [l + k * n0 for k in range(K) for l in lists]
    prob = dict(kind="symmetric", diagonals=dd * K, diagonalindices=tile(p["diagonalindices"]),
                offdiagonals=oo * K, rowindices=tile(p["rowindices"]), colindices=tile(p["colindices"]),
                size=(n0 * K, n0 * K))
    rng = np.random.default_rng(0)
    xh = rng.standard_normal(n0 * K)
    if dtype.kind == "c":  # a FULL complex x (zero imaginary parts run 5-8 % faster: less switching, higher clock)
        xh = xh + 1j * rng.standard_normal(n0 * K)
    prob["x"] = torch.from_numpy(xh.astype(dtype)).cuda()
    return prob, p, n0


def fixture_coo(np, fx, n0):
    """the fixture's operator through an independent COO sum (the reference's test oracle, src/sparse.jl), in double"""
    import scipy.sparse as sp
    rr, cc_, vv = [], [], []
    for blk, idx in zip(fx["diagonals"], fx["diagonalindices"]):
        R, Cq = np.meshgrid(idx - 1, idx - 1, indexing="ij")
        rr.append(R.ravel()); cc_.append(Cq.ravel()); vv.append(blk.ravel())
    for blk, ri, ci in zip(fx["offdiagonals"], fx["rowindices"], fx["colindices"]):
        R, Cq = np.meshgrid(ri - 1, ci - 1, indexing="ij")
        rr += [R.ravel(), Cq.ravel()]; cc_ += [Cq.ravel(), R.ravel()]; vv += [blk.ravel(), blk.ravel()]
    vals = np.concatenate(vv)
    vals = vals.astype(np.complex128 if vals.dtype.kind == "c" else np.float64)
    return sp.coo_matrix((vals, (np.concatenate(rr), np.concatenate(cc_))), shape=(n0, n0)).tocsr()


def run_c2(args, bsm, torch, dist, np, rank, world, barrier, reduce_scalars):
    from bsm_amd import _lib
    build = _lib.lib().bsm_version().decode().split("build ")[-1]
    prob = bsm.synthetic.config2(part=(rank, world) if world > 1 else None)
    A = bsm.VariableBlockCompressedRowStorage(prob["blocks"], prob["rowstart"], prob["colstart"],
                                              prob["size"], own=prob.get("own"))
    st = A.stats()
    assert st["exclusive"] == 1
    n = prob["size"][0]
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

    for _ in range(args.warmup):
        plan()
    torch.cuda.synchronize()
    graph = None
    launch = args.launch
    if launch == "graph":
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

    # ---- timed region: the K steps, R times --------------------------------------------------------------
    # K launches of a 10 us product are 0.2 ms at the driver's K = 20: the barrier + synchronize on both sides would be
    # 8 % of that.  When K steps take less than MIN_TIMED_S the K-step region is repeated R times back to back (the
    # same captured graph replayed R times) and ms_per_step = total / (K R); `steps` stays K, config.replays = R.
    def k_steps():
        if graph is not None:
            graph.replay()
        else:
            for _ in range(args.steps):
                plan()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    k_steps()  # (untimed: sizes R)
    ev1.record()
    torch.cuda.synchronize()
    once = max(ev0.elapsed_time(ev1) * 1e-3, 1e-6)
    replays = 1 if once >= MIN_TIMED_S else min(1000, int(MIN_TIMED_S / once) + 1)
    if dist is not None:  # every rank repeats the same number of times
        rr = torch.tensor([float(replays)], dtype=torch.float64, device=reduce_scalars.comm["dev"])
        dist.all_reduce(rr, op=dist.ReduceOp.MAX, group=reduce_scalars.comm["group"])
        replays = int(rr.item())
    # three such regions, the median reported (a host hiccup during the enqueue of one region -- another process of the
    # box waking up -- cost 25 % of the wall-clock figure in one of this round's runs while the device time did not move)
    regions = []
    for _ in range(3):
        barrier()
        t0 = time.perf_counter()
        ev0.record()  # HIP events on the stream the kernels are launched on
        for _ in range(replays):
            k_steps()
        ev1.record()
        barrier()
        regions.append(((time.perf_counter() - t0) / replays, ev0.elapsed_time(ev1) * 1e-3 / replays))
    regions.sort()
    elapsed, dev_elapsed = regions[1]
    elapsed, total_bytes = reduce_scalars(elapsed, alg_bytes)
    value = total_bytes * args.steps / elapsed / 1e9

    # ---- roofline of the dominant kernel --------------------------------------------------------
    # one step == one launch of bsm::panel_kernel<double,8,true,false,true> (forward-only, non-temporal
    # matrix loads); its average duration is the HIP-event time of the timed region / K (back-to-back
    # launches on one stream; the rocprofv3 --kernel-trace average in profiles/ must agree).
    KERNEL = "bsm::panel_kernel<double,8,true,false,true>"
    kdur = dev_elapsed / args.steps
    achieved = alg_bytes / kdur / 1e9
    pmc, why, traffic_src = None, None, None
    if rank == 0 and world == 1 and not args.no_live_pmc:
        pmc, why_live = live_traffic("panel_kernel<double, 8, true, false", int(alg_bytes))
        traffic_src = "two rocprofv3 --pmc passes of THIS run (FETCH_SIZE, WRITE_SIZE; 60 eager launches each)" if pmc else None
        if pmc is None:
            why = "live passes: " + why_live
    if pmc is None:
        pmc, why_file = counter_file(PMC_FILE, build, kernel=KERNEL, alg_bytes=int(alg_bytes))
        if pmc is not None:
            traffic_src = os.path.relpath(PMC_FILE, ROOT) + " (same build, kernel and byte count)" + ("; " + why if why else "")
        else:
            why = (why + "; " if why else "") + why_file
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": pmc["traffic_bytes_per_launch"] if pmc else None,
                "traffic_source": traffic_src if pmc else "none -- " + str(why),
                "kernel": KERNEL, "build": build,
                "alg_bytes_per_launch": int(alg_bytes), "avg_launch_us": round(kdur * 1e6, 3),
                "note": "warm: the 54 MB operator stays in the 256 MiB Infinity Cache between launches, so `frac` divides "
                        "a cache-resident launch by the HBM peak -- stream_floor_us is what a BARE streaming read of the "
                        "same byte count takes in the same state (same request shape, nothing else to do) and "
                        "frac_of_stream_floor the honest distance; extra.hbm_vbcrs_fp64 / extra.c3_fused / "
                        "extra.bem_tiled are HBM-streaming legs.  traffic = rocprofv3 FETCH_SIZE x2 (gfx950 "
                        "correction) + WRITE_SIZE per launch (traffic_source)"}
    floor = None
    if rank == 0:
        try:
            floor = StreamFloor(torch, alg_bytes)
            for _ in range(20):
                floor()
            torch.cuda.synchronize()
            f_us = graph_timed(floor, args.steps, torch) if launch == "graph" else None
            if f_us is None:
                f_us = timed(floor, args.steps, torch)
            roofline["stream_floor_us"] = round(f_us * 1e6, 3)
            roofline["frac_of_stream_floor"] = round(f_us / kdur, 4)
        except Exception as e:  # pragma: no cover
            roofline["stream_floor_error"] = repr(e)

    extra = {}
    if rank == 0 and launch == "graph":
        for _ in range(20):
            plan()
        torch.cuda.synchronize()
        extra["eager_us_per_step"] = round(timed(plan, args.steps, torch) * 1e6, 3)
    if rank == 0 and world == 1 and not args.no_extra:
        flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
        dirty = lambda: flush.fill_(1)
        clean = lambda: flush.sum()
        c = cold_median(plan, dirty, 20, torch)
        extra["cold_median_us"] = round(c * 1e6, 2)
        extra["cold_GBps"] = round(alg_bytes / c / 1e9, 1)
        # the same with a READ sweep of 512 MiB as the flush: the operator is out of every cache, but
        # the launch does not have to evict 256 MiB of dirty lines first
        cc = cold_median(plan, clean, 20, torch)
        extra["cold_clean_median_us"] = round(cc * 1e6, 2)
        extra["cold_clean_GBps"] = round(alg_bytes / cc / 1e9, 1)
        if floor is not None:
            # the cold FLOOR: the bare streaming read of the same byte count right after the same flush, and
            # the same with the one dependent descriptor load every product wave starts with
            f0 = cold_median(lambda: floor(0), dirty, 20, torch)
            f1 = cold_median(lambda: floor(1), dirty, 20, torch)
            extra["cold_floor_us"] = round(f0 * 1e6, 2)
            extra["cold_floor_hop_us"] = round(f1 * 1e6, 2)
            extra["cold_over_floor_hop"] = round(c / f1, 3)
        del flush
        # driver-timed HBM-streaming legs (operators far larger than the 256 MiB Infinity Cache),
        # generated in HBM and packed by the device-side packer
        S = bsm.synthetic
        try:  # (additional figures must never cost the headline line)
            extra["hbm_vbcrs_fp64"], _ = leg(bsm, torch, S.config2(n=2_000_000, nblocks=100_000, on_device=True), 50)
            extra["hbm_vbcrs_fp64"]["workload"] = "C2-shaped VBCRS 2M x 2M, 100 000 fp64 blocks 8-64 (20 x C2), forward mul!"
            extra["c3_fused"], _ = leg(bsm, torch, S.config3(on_device=True), 50, multi_rhs=(8, 16))
            extra["c3_fused"]["workload"] = "C3: SymmetricBlockMatrix 200k x 200k, 64x64 fp64 blocks, half-bandwidth 8, fused A + A^T mul!"
        except Exception as e:  # pragma: no cover
            extra["legs_error"] = repr(e)
        torch.cuda.empty_cache()
        # the reference's own workload in the four element types of the library (its fixture is ComplexF64; Float64 =
        # its real part; the 4-byte types the same data rounded): driver-timed, each with its parity number
        bem = {"workload": "the reference's BEM fixture (test/assets/symmetricblockexamples.jld2 'cuboid', 96 leaves of 3-28 rows, "
                           "scattered near-field columns) tiled 400 x along the diagonal, fused A + A^T mul!; legs c128 (the "
                           "fixture's own type), f64 (real part), c64, f32 (rounded)"}
        for tname, dtn, part, tol in BEM_TYPES:
            try:
                bp, fx, n0 = bem_tiled_problem(torch, np, 400, dtn, part)
                one, yb = leg(bsm, torch, bp, 50, multi_rhs=8)
                one["dtype"] = tname
                # parity of the leg: tile 0 of y against the fixture's own product through an independent COO sum
                ref = fixture_coo(np, fx, n0) @ bp["x"][:n0].cpu().numpy().astype(np.complex128 if np.dtype(dtn).kind == "c" else np.float64)
                got = yb[:n0].cpu().numpy()
                one["relerr_vs_coo"] = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
                one["parity_tol"] = tol
                one["parity_ok"] = bool(one["relerr_vs_coo"] <= tol)
                bem[tname] = one
                del bp, yb
            except Exception as e:  # pragma: no cover
                bem[tname] = {"error": repr(e)}
            torch.cuda.empty_cache()
        if "us" in bem.get("c128", {}):  # (the round-3 fields of the ComplexF64 leg stay where they were)
            bem.update({k: v for k, v in bem["c128"].items() if k not in bem})
        extra["bem_tiled"] = bem
        torch.cuda.empty_cache()
        # the workload of the N > 1 lines (C5, strong-scaled) on THIS one GPU, through the same code path (the overlapped
        # step with no neighbour): the same-operator anchor of a scaling curve -- the headline of this line is C2
        # (BASELINE.json's 1-GPU configuration), a different operator in a different cache state
        try:
            a5 = argparse.Namespace(**vars(args))
            a5.steps, a5.warmup, a5.no_extra, a5.scale, a5.no_overlap, a5.note = 20, 3, True, 1.0, False, None
            o5 = run_partitioned(a5, bsm, torch, None, np, 0, 1, barrier, reduce_scalars)
            extra["c5_n1"] = {"workload": o5["config"]["workload"], "value": o5["value"], "unit": "GB/s", "ms_per_step": o5["ms_per_step"],
                              "steps": 20, "exchange_us": o5["config"]["exchange_us"], "local_kernel_us_max": o5["config"]["local_kernel_us_max"],
                              "parity_relerr": o5["config"]["parity_relerr"], "parity_vs_oracle_sampled": o5["config"]["parity_vs_oracle_sampled"],
                              "frac_of_hbm_peak": o5["config"]["frac_of_hbm_peak"],
                              "build": build, "note": "what `python bench.py --gpus 1 --workload c5` prints; N > 1 lines carry the "
                                                      "committed copy of this figure as config.n1_same_workload"}
        except Exception as e:  # pragma: no cover
            extra["c5_n1"] = {"error": repr(e)}
        torch.cuda.empty_cache()
        # the same step with the exchanges REALLY running over RCCL: one rank, loopback (see --loopback).  What the halo
        # exchange of a middle rank costs beside the interior launch on this box; the first and only execution of the
        # nccl branches of distributed.py a one-GPU box allows.
        try:
            why = profiler_in_environment(os.environ)
            if why:  # (rocprofv3 + RCCL + a stream with a CU mask in one process: the run completes, the process then dies in its
                # exit handlers -- seen with tools/profile_round5.sh; the leg belongs to un-profiled runs)
                raise RuntimeError("skipped under a profiler (%s)" % why)
            d1 = init_one_rank("nccl", torch, torch.cuda.current_device())
            a5.loopback = True
            cm = {"group": None, "dev": "cuda", "name": "nccl", "fallback": None}

            def bar1():
                d1.barrier()
                torch.cuda.synchronize()

            def red1(elapsed, nbytes):
                return elapsed, float(nbytes)
            red1.comm = cm
            o6 = run_partitioned(a5, bsm, torch, d1, np, 0, 1, bar1, red1)
            extra["c5_n1_rccl_loopback"] = {"value": o6["value"], "unit": "GB/s", "ms_per_step": o6["ms_per_step"], "steps": 20,
                                            "exchange_us": o6["config"]["exchange_us"], "local_kernel_us_max": o6["config"]["local_kernel_us_max"],
                                            "parity_relerr": o6["config"]["parity_relerr"], "parity_vs_oracle_sampled": o6["config"]["parity_vs_oracle_sampled"],
                                            "backend": o6["config"]["backend"],
                                            "loopback": o6["config"]["loopback"]}
            d1.destroy_process_group()
        except Exception as e:  # pragma: no cover
            extra["c5_n1_rccl_loopback"] = {"error": repr(e)}
        torch.cuda.empty_cache()
    mf, why_mf = counter_file(MFMA_FILE, build)
    if rank == 0:
        extra["c4_mfma"] = mf if mf else {"unavailable": why_mf}

    out = {
        "metric": "fp64 block-SpMV GB/s (VBCRS mul!, algorithmic bytes / time)",
        "value": round(value, 1), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 6),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "C2: VBCRS 100000x100000 per GPU, 5000 variable 8-64 fp64 blocks per GPU "
                               "(SplitMix64 seed 0xB5A2), mul!(y, A, x), x/y/A resident in HBM",
                   "global_rows": n, "blocks_per_gpu": len(prob["blocks"]),
                   "alg_bytes_per_gpu": int(alg_bytes), "launch": launch, "replays": replays,
                   "timed_region": "%d x (K = %d steps) between two barriers, three times; ms_per_step = median wall time / (K x replays); "
                                   "wall times of the three regions in ms per step: %s" % (replays, args.steps, ", ".join("%.6f" % (r[0] / args.steps * 1e3) for r in regions)),
                   "partition": "block rows, no data-path collective",
                   "frac_of_hbm_peak": round(value / (HBM_PEAK_GBPS * world), 4)},
        "roofline": roofline,
    }
    if extra:
        out["extra"] = extra

    # ---- CPU baseline: the oracle (reference loop structure, C port), rank 0, N = 1 only -------------
    # LAST: every GPU figure above is complete (and kept on stderr) before these ~15 s of host work start
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        print("[bench] GPU legs complete, CPU baseline next: " + json.dumps(out), file=sys.stderr, flush=True)
        from oracle.oracle import load_oracle_native
        orc = load_oracle_native()  # -O3 -march=native build of oracle/bsm_oracle.c, compiled on this box
        perm, rowptr, colind, rowind = orc.vbcrs_build(prob["rowstart"], prob["colstart"])
        blocks = [prob["blocks"][p - 1] for p in perm]
        xh = prob["x"]
        yh = np.zeros(n)
        # everything is marshalled once; the repetitions and the clock are inside C (orc_vbcrs_bench_f64)
        reps, dt = orc.vbcrs_bench(blocks, rowptr, colind, rowind, xh, yh, seconds=10.0)
        err = float(np.max(np.abs(y.cpu().numpy() - yh)) / np.max(np.abs(yh)))
        out["cpu_baseline"] = {"value": round(st["alg_bytes"] * reps / dt / 1e9, 3), "unit": "GB/s", "cores": 1, "kind": "port",
                               "sample": f"{reps} full C2 mul! calls of oracle/bsm_oracle.c (orc_vbcrs_mul_f64, gcc -O3 "
                                         f"-march=native, loop timed in C, no marshalling) in {dt:.1f} s on one host core",
                               "gpu_vs_oracle_relerr": err}
        # all host cores: one OpenMP task per block row == the reference's `@tasks for browidx` with
        # DynamicScheduler() (src/vbcrs.jl:275-276); reported beside, not as the baseline
        try:
            ncores = int(os.environ.get("OMP_NUM_THREADS", "1"))
            yp = np.zeros(n)
            preps, dtp = orc.vbcrs_bench(blocks, rowptr, colind, rowind, xh, yp, seconds=5.0, parallel=True)
            out.setdefault("extra", {})["cpu_allcores"] = {
                "value": round(st["alg_bytes"] * preps / dtp / 1e9, 3), "unit": "GB/s",
                "cores": ncores, "kind": "port (OpenMP over block rows)",
                "sample": f"{preps} C2 mul! calls in {dtp:.1f} s, timed in C",
                "relerr_vs_1core": float(np.max(np.abs(yp - yh)) / np.max(np.abs(yh)))}
        except Exception as e:  # pragma: no cover
            out.setdefault("extra", {})["cpu_allcores"] = {"error": str(e)}
        # the reference's three-sweep symmetric product (src/symmetricblockmatrix.jl:386-435) beside the fused legs, on
        # bounded samples of the same operators, loop and clock in C (orc_sym_bench_*), one host core
        if not args.no_extra:
            def sym_cpu(prob, seconds, what):
                dt = prob["diagonals"][0].dtype
                es = dt.itemsize
                nn = prob["size"][0]
                xs_ = np.random.default_rng(1).standard_normal(nn).astype(dt)
                ys_ = np.zeros(nn, dt)
                reps, secs = orc.sym_bench(prob["diagonals"], prob["diagonalindices"], prob["offdiagonals"], prob["rowindices"],
                                           prob["colindices"], xs_, ys_, seconds=seconds)
                stored = sum(b.size for b in prob["diagonals"]) + sum(b.size for b in prob["offdiagonals"])
                meta = 8 * (sum(len(d) for d in prob["diagonalindices"]) + sum(len(r) + len(c) for r, c in zip(prob["rowindices"], prob["colindices"])))
                alg = stored * es + meta + 2 * nn * es  # every stored entry once (the reference reads the off-diagonal ones twice)
                return {"value": round(alg * reps / secs / 1e9, 3), "unit": "GB/s", "cores": 1, "kind": "port",
                        "sample": f"{reps} mul! calls of {what} ({alg / 1e6:.0f} MB algorithmic) in {secs:.1f} s, oracle/bsm_oracle.c "
                                  "orc_sym_mul (three sweeps, serial colour sets), timed in C"}
            try:
                ex = out.setdefault("extra", {})
                if "c3_fused" in ex and "error" not in ex["c3_fused"]:
                    ex["c3_fused"]["cpu_baseline"] = sym_cpu(bsm.synthetic.config3(nseg=300), 4.0, "C3 with 300 of its 3 125 diagonal segments")
                if "bem_tiled" in ex:
                    sys.path.insert(0, os.path.join(ROOT, "tests"))
                    from _common import fixture_problem
                    for tname, dtn, part, _tol in BEM_TYPES:
                        if "error" in ex["bem_tiled"].get(tname, {"error": 1}):
                            continue
                        fx = fixture_problem("cuboid", np.dtype(dtn), part)
                        n0, KC = fx["size"][0], 24  # 24 tiles with their OWN copies of the blocks (a CPU cache must not hold the operator)
                        tile = lambda lists: [l + k * n0 for k in range(KC) for l in lists]
                        pc = dict(diagonals=[b.copy() for _ in range(KC) for b in fx["diagonals"]], diagonalindices=tile(fx["diagonalindices"]),
                                  offdiagonals=[b.copy() for _ in range(KC) for b in fx["offdiagonals"]], rowindices=tile(fx["rowindices"]),
                                  colindices=tile(fx["colindices"]), size=(n0 * KC, n0 * KC))
                        ex["bem_tiled"][tname]["cpu_baseline"] = sym_cpu(pc, 2.0, f"the fixture tiled {KC} x in {tname}")
            except Exception as e:  # pragma: no cover
                out.setdefault("extra", {})["cpu_symmetric_error"] = repr(e)
    return out


if __name__ == "__main__":
    main()
