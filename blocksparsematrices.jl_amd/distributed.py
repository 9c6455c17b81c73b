"""One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI): the block rows of an
operator partitioned over the ranks of a single 8 x MI355X node.

The reference has no distributed code (one process, shared-memory tasks); this layer and the
in-library multi-device handles (bsm_ctx_t, csrc/bsm_dist.cpp) are the two MI355X counterparts of its
`@tasks` fan-out over block rows / colour classes.  Both use the SAME row partition
(`bsm_partition_rows` in the C ABI) and the same exchange pattern:

  * VBCRS forward: block rows own disjoint y ranges (reference src/vbcrs.jl:275-283), so every
    rank multiplies its own rows and NO collective is needed; `gather=True` adds one all-gather of
    the y slices for callers that need the whole y on every GPU (a Krylov iteration: y is the next x).
  * SymmetricBlockMatrix: the off-diagonal block (I, J) lives with the owner of row set I but also
    contributes B^T x_I to y_J, which may belong to another rank (for banded operators: the
    previous rank only).  Each rank accumulates into a work vector over the rows it TOUCHES and the
    overlaps are exchanged point-to-point (batched isend/irecv == ncclSend/ncclRecv over the direct
    xGMI links) and added -- never a ring all-reduce of the full y, which would be bound by one link.
  * BlockSparseMatrix (index lists): blocks go with the rank that owns their smallest row index;
    rows a block reaches outside its rank's range travel through the same point-to-point exchange.
  * Products that run ACROSS the partition (transpose(A)*x of a row-partitioned VBCRS /
    BlockSparseMatrix, A*x of a column-partitioned one) give a full-length partial result on every
    rank: one reduce-scatter onto equal chunks (all-reduce when the caller wants the whole y).
    `split_vbcrs(..., axis=1)` is the column partition that makes the TRANSPOSED products
    collective-free (mirror of the forward case, reference src/vbcrs.jl:303-329).

Vectors: either x holds the FULL vector on every rank (and gather=True replicates the result again),
or -- what scales -- x and y stay PARTITIONED like the rows (mul(..., x_distributed=True)): a rank
then fetches only the x entries its blocks read (`xneed`: own range + halo for banded / symmetric
operators, one batched send/recv; everything for operators with scattered columns: all-gather).
Every buffer of the exchange is allocated once, at the first product of a given kind.
"""
import numpy as np

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None

from . import matrices as M


def nccl_options():
    """`pg_options` for `init_process_group("nccl", ...)`: the collective layer's internal stream as a HIGH-PRIORITY
    stream.  The runtime deals ordinary streams onto a handful of hardware queues per device; in the kernel trace of the
    RCCL loopback step (profiles/r05_loopback_timeline.txt) RCCL's stream shared the main stream's queue, so the
    partial-y send / recv kernel waited for the whole interior launch it was meant to run beside.  Priority streams get
    queues of their own.  None when this torch build has no such option."""
    try:
        opts = dist.ProcessGroupNCCL.Options()
        opts.is_high_priority_stream = True
        return opts
    except Exception:  # pragma: no cover
        return None


_compute_streams = {}


def compute_stream(device=None, reserve=None):
    """The stream a rank's products should run on when `mul_overlapped` exchanges beside them: a torch ExternalStream over
    `bsm_stream_create_reserved` whose CU mask leaves `reserve` CUs (default: BSM_RESERVE_CUS or 8 = one per XCD) to RCCL's
    kernels.  Without it the send / recv kernels find no free CU beside the interior launch and the exchange ends when
    the launch does (profiles/r05_loopback_*.txt: C5 share of one rank of eight, step 698 -> 656 us).  Use as
    `with torch.cuda.stream(D.compute_stream()): ...` around the solver loop.  One stream per (device, reserve), kept for
    the life of the process."""
    import ctypes as C
    import os
    from . import _lib as L
    dev = torch.cuda.current_device() if device is None else int(device)
    if reserve is None:
        reserve = int(os.environ.get("BSM_RESERVE_CUS", "8"))
    key = (dev, int(reserve))
    if key not in _compute_streams:
        st = C.c_void_p()
        L.check(L.lib().bsm_stream_create_reserved(dev, int(reserve), C.byref(st)))
        _compute_streams[key] = torch.cuda.ExternalStream(st.value, device=torch.device("cuda", dev))
    return _compute_streams[key]


def balanced_cuts(weights, nparts):
    """Cut len(weights) consecutive units into nparts contiguous ranges of ~equal total weight.
    Returns nparts+1 boundaries (unit indices).  (Same rule as bsm_partition_rows.)"""
    w = np.asarray(weights, dtype=np.float64)
    csum = np.concatenate([[0.0], np.cumsum(w)])
    total = csum[-1]
    cuts = [0]
    for p in range(1, nparts):
        target = total * p / nparts
        k = int(np.searchsorted(csum, target, side="left"))
        k = min(max(k, cuts[-1]), len(w))
        cuts.append(k)
    cuts.append(len(w))
    return cuts


def _minrow(lst):
    return int(np.min(lst)) if len(lst) else 1


def split_vbcrs(problem, rank, nparts, axis=0):
    """Partition a VBCRS problem dict into contiguous ranges of block rows (axis=0) or block columns
    (axis=1: for transposed products) balanced by stored entries (bsm_partition_rows).
    Returns (local problem, own=(lo, hi) 1-based inclusive along `axis`; hi = lo - 1: nothing)."""
    keys = np.asarray(problem["rowstart" if axis == 0 else "colstart"], dtype=np.int64)
    weights = [b.size for b in problem["blocks"]]
    part, own = M.partition_rows(problem["size"][axis], keys, weights, nparts)
    keep = np.nonzero(part == rank)[0]
    local = dict(kind="vbcrs", blocks=[problem["blocks"][b] for b in keep],
                 rowstart=np.asarray(problem["rowstart"], dtype=np.int64)[keep],
                 colstart=np.asarray(problem["colstart"], dtype=np.int64)[keep], size=problem["size"])
    return local, own[rank]


def _touched(own, lists):
    lo, hi = own
    for lst in lists:
        if len(lst):
            lo, hi = min(lo, int(np.min(lst))), max(hi, int(np.max(lst)))
    return (lo, hi) if hi >= lo else (own[0], own[0] - 1)


def split_blocksparse(problem, rank, nparts):
    """Partition a BlockSparseMatrix problem (arbitrary index lists): a block goes with the rank that
    owns its smallest row index.  Returns (local problem, own, touched), 1-based inclusive."""
    keys = [_minrow(r) for r in problem["rowindices"]]
    weights = [b.size for b in problem["blocks"]]
    part, own = M.partition_rows(problem["size"][0], keys, weights, nparts)
    keep = [int(b) for b in np.nonzero(part == rank)[0]]
    local = dict(kind="blocksparse", blocks=[problem["blocks"][b] for b in keep],
                 rowindices=[problem["rowindices"][b] for b in keep],
                 colindices=[problem["colindices"][b] for b in keep], size=problem["size"])
    return local, own[rank], _touched(own[rank], local["rowindices"])


def split_symmetric(problem, rank, nparts):
    """Partition a SymmetricBlockMatrix problem: diagonal block d and off-diagonal block b go with the
    rank that owns their smallest (row) index, balanced by stored entries.
    Returns (local problem, own=(lo, hi), touched=(lo, hi)), 1-based inclusive."""
    nd = len(problem["diagonals"])
    keys = [_minrow(d) for d in problem["diagonalindices"]] + [_minrow(r) for r in problem["rowindices"]]
    weights = [b.size for b in problem["diagonals"]] + [b.size for b in problem["offdiagonals"]]
    part, own = M.partition_rows(problem["size"][0], keys, weights, nparts)
    dkeep = [int(i) for i in np.nonzero(part[:nd] == rank)[0]]
    okeep = [int(b) for b in np.nonzero(part[nd:] == rank)[0]]
    local = dict(kind="symmetric", diagonals=[problem["diagonals"][i] for i in dkeep],
                 diagonalindices=[problem["diagonalindices"][i] for i in dkeep],
                 offdiagonals=[problem["offdiagonals"][b] for b in okeep],
                 rowindices=[problem["rowindices"][b] for b in okeep],
                 colindices=[problem["colindices"][b] for b in okeep], size=problem["size"])
    touched = _touched(own[rank], local["diagonalindices"] + local["rowindices"] + local["colindices"])
    return local, own[rank], touched


def is_empty(local):
    """A rank may receive no block at all (more ranks than block rows): it creates no handle
    (`RowPartitioned(None, ...)`) and still takes part in every collective."""
    return (len(local.get("blocks", ())) + len(local.get("diagonals", ())) + len(local.get("offdiagonals", ()))) == 0


def build_local(local, touched=None, **kw):
    """The rank's handle: built with own = the rows it TOUCHES, so that its `y .*= beta` pass covers
    exactly the rows it writes.  None for a rank without blocks."""
    if is_empty(local):
        return None
    from . import synthetic
    if touched is not None and touched[1] >= touched[0]:
        kw["own"] = touched
    return synthetic.build(local, **kw)


def _within(lst, lo, hi):
    return len(lst) == 0 or (int(np.min(lst)) >= lo and int(np.max(lst)) <= hi)


def _range_list(start, length):
    return (int(start), int(start) + int(length) - 1)


def split_interior(local, own):
    """Cuts a rank's blocks into INTERIOR blocks -- they read only x[own] and write only y[own], so they
    need nothing from another rank and nothing of theirs leaves the rank -- and BOUNDARY blocks (the
    rest: they read x entries of a neighbour and / or produce y segments for it).  The interior
    product can run while the x halo of the boundary blocks is still travelling, and the partial-y
    segments leave as soon as the (small) boundary product is done.
    Returns (interior, boundary, btouched, bxneed): two problem dicts of local's kind, the hull of the
    rows the boundary blocks write and the hull of the x entries they read (1-based inclusive;
    (own_lo, own_lo - 1) when there is no boundary block)."""
    lo, hi = int(own[0]), int(own[1])
    kind = local["kind"]
    sym = kind == "symmetric"
    empty = (lo, lo - 1)
    tl, th, xl, xh = None, None, None, None

    def grow(rng_rows, rng_cols):
        nonlocal tl, th, xl, xh
        # rows written / x entries read by one boundary block (a symmetric off-diagonal block writes and
        # reads both its row and its column lists)
        wr = [rng_rows] + ([rng_cols] if sym else [])
        rd = [rng_cols] + ([rng_rows] if sym else [])
        for a, b in wr:
            tl, th = (a, b) if tl is None else (min(tl, a), max(th, b))
        for a, b in rd:
            xl, xh = (a, b) if xl is None else (min(xl, a), max(xh, b))

    def span(lst):
        return (int(np.min(lst)), int(np.max(lst)))

    if kind == "vbcrs":
        inner, outer = [], []
        for b, blk in enumerate(local["blocks"]):
            rr = _range_list(local["rowstart"][b], blk.shape[0])
            cr = _range_list(local["colstart"][b], blk.shape[1])
            if lo <= rr[0] and rr[1] <= hi and lo <= cr[0] and cr[1] <= hi:
                inner.append(b)
            else:
                outer.append(b)
                grow(rr, cr)

        def sub(keep):
            return dict(kind="vbcrs", blocks=[local["blocks"][b] for b in keep],
                        rowstart=np.asarray(local["rowstart"], dtype=np.int64)[keep],
                        colstart=np.asarray(local["colstart"], dtype=np.int64)[keep], size=local["size"])
        interior, boundary = sub(inner), sub(outer)
    elif kind == "blocksparse":
        inner, outer = [], []
        for b in range(len(local["blocks"])):
            r, c = local["rowindices"][b], local["colindices"][b]
            if len(r) == 0 or len(c) == 0 or (_within(r, lo, hi) and _within(c, lo, hi)):
                inner.append(b)
            else:
                outer.append(b)
                grow(span(r), span(c))

        def sub(keep):
            return dict(kind="blocksparse", blocks=[local["blocks"][b] for b in keep],
                        rowindices=[local["rowindices"][b] for b in keep],
                        colindices=[local["colindices"][b] for b in keep], size=local["size"])
        interior, boundary = sub(inner), sub(outer)
    elif kind == "symmetric":
        din, dout, oin, oout = [], [], [], []
        for d in range(len(local["diagonals"])):
            idx = local["diagonalindices"][d]
            if _within(idx, lo, hi):
                din.append(d)
            else:
                dout.append(d)
                grow(span(idx), span(idx))
        for b in range(len(local["offdiagonals"])):
            r, c = local["rowindices"][b], local["colindices"][b]
            if len(r) == 0 or len(c) == 0 or (_within(r, lo, hi) and _within(c, lo, hi)):
                oin.append(b)
            else:
                oout.append(b)
                grow(span(r), span(c))

        def sub(dk, ok):
            return dict(kind="symmetric", diagonals=[local["diagonals"][d] for d in dk],
                        diagonalindices=[local["diagonalindices"][d] for d in dk],
                        offdiagonals=[local["offdiagonals"][b] for b in ok],
                        rowindices=[local["rowindices"][b] for b in ok],
                        colindices=[local["colindices"][b] for b in ok], size=local["size"])
        interior, boundary = sub(din, oin), sub(dout, oout)
    else:
        raise ValueError(kind)
    btouched = empty if tl is None else (tl, th)
    bxneed = empty if xl is None else (xl, xh)
    return interior, boundary, btouched, bxneed


def build_overlapped(local, own, group=None, symmetric=None, xmode="auto", loopback=None, solo=False, **kw):
    """RowPartitioned for the partitioned-vector forward product with the exchange OVERLAPPED with the
    interior rows (mul_overlapped): two handles per rank -- interior blocks with own = the rank's
    rows, boundary blocks with own = the rows they touch.  xmode: "halo" (the x entries the boundary
    blocks read form a neighbourhood of the own range: point-to-point), "allgather" (scattered block
    columns), "auto": halo when that neighbourhood is at most as long as the own range itself."""
    interior, boundary, btouched, bxneed = split_interior(local, own)
    sym = (local["kind"] == "symmetric") if symmetric is None else symmetric
    A_int = build_local(interior, own if own[1] >= own[0] else None, **kw)
    A_bnd = build_local(boundary, btouched, **kw)
    if xmode == "auto":
        extra = max(0, own[0] - bxneed[0]) + max(0, bxneed[1] - own[1]) if bxneed[1] >= bxneed[0] else 0
        xmode = "halo" if extra <= max(own[1] - own[0] + 1, 0) else "allgather"
    P = RowPartitioned(A_bnd, own, btouched, group=group, gather=False, symmetric=sym,
                       xneed=(bxneed if xmode == "halo" else None), interior=A_int, loopback=loopback, solo=solo)
    P.xmode = xmode
    return P


class RowPartitioned:
    """y = alpha*op(A)*x + beta*y with A's blocks spread over the ranks of `group`.

    `axis` = 0: `own` / `touched` are ROW ranges (forward products are local up to the halo);
    `axis` = 1: column ranges of a column-partitioned operator (the transposed products are).
    `symmetric`: op(A) has A's structure, every op runs along the partition.

    x must hold the FULL vector on every rank.  After mul() the rank's output range is final in y
    (the whole y when gather=True): `own` for products along the partition, `out_range(n)` (equal
    chunks) for products across it.  `local` is this rank's matrix (built with own=touched range so
    its beta pass covers exactly the rows it touches), or None for a rank without blocks."""

    def __init__(self, local, own, touched=None, group=None, gather=False, axis=0, symmetric=None, xneed=None,
                 interior=None, loopback=None, solo=False):
        self.local = local
        # solo: this object takes no part in the process group although one is initialised (rank 0 of a finished N-rank
        # run measuring the whole operator alone: bench.py's n1_same_workload) -- rank 0 of a world of one
        self.solo = bool(solo)
        # LOOPBACK rehearsal (one rank, any backend -- meant for "nccl" on the single GPU of a test box, where RCCL
        # refuses two ranks per device): every collective and point-to-point branch of this class runs against the
        # rank ITSELF instead of being skipped at world == 1.
        #   loopback=True     the world == 1 short cuts are off: all_gather_into_tensor / reduce_scatter_tensor /
        #                     all_reduce really run (one rank), and what they deliver is what ends up in y (the
        #                     source slice is poisoned with NaN in between);
        #   loopback=(lo, hi) additionally the rank plays TWO roles: itself, owning rows lo..hi only (`own` must be
        #                     that range), and a block-less phantom neighbour owning every other row.  x entries
        #                     outside lo..hi and partial-y segments outside lo..hi then travel through grouped self
        #                     send / recv (batch_isend_irecv on device tensors: ncclSend / ncclRecv to the own rank
        #                     inside one group are legal) exactly like the halo between two ranks; the source of
        #                     every transfer is poisoned or zeroed behind the send, so a transfer that did not
        #                     happen, or one that was overtaken by the kernels around it, shows in the result.
        self.loopback = bool(loopback)
        self._phantom = None
        if isinstance(loopback, (tuple, list)):
            if (int(loopback[0]), int(loopback[1])) != (int(own[0]), int(own[1])):
                raise ValueError("loopback=(lo, hi) must be the `own` range of the rank's own role")
        # mul_overlapped only: handle of the rank's INTERIOR blocks (own = the rank's rows); `local` then
        # holds the boundary blocks and `touched` the rows THEY write (see split_interior)
        self.interior = interior
        self._side = None
        self.own = (int(own[0]), int(own[1]))
        self.touched = self.own if touched is None else (int(touched[0]), int(touched[1]))
        # x entries this rank's blocks read (1-based inclusive) when x arrives PARTITIONED like y
        # (mul(..., x_distributed=True)); None: everything (the x slices are all-gathered first)
        self.xneed = None if xneed is None else (int(xneed[0]), int(xneed[1]))
        self._xplan = None
        self.group = group
        self.gather = gather
        self.axis = axis
        self.symmetric = isinstance(local, M.SymmetricBlockMatrix) if symmetric is None else symmetric
        grouped = dist is not None and dist.is_initialized() and not self.solo
        self.rank = dist.get_rank(group) if grouped else 0
        self.world = dist.get_world_size(group) if grouped else 1
        if self.loopback:
            if self.world != 1:
                raise ValueError("loopback is a one-rank rehearsal")
            self._phantom = isinstance(loopback, (tuple, list))
        self._ranges = None
        self._plan = None   # halo exchange of products along the partition (buffers included)
        self._work = None
        self._gbuf = self._sbuf = self._pad = self._rsout = None
        self._ops = {}
        self._segadd = None
        self._xplan_m = self._plan_m = self._work_m = None  # mul_multi: plans and the (n, K) work matrix

    def out_range(self, n, rank=None):
        """Output rows (1-based, inclusive) rank `rank` holds after a product ACROSS the partition."""
        r = self.rank if rank is None else rank
        chunk = -(-n // self.world)
        return (min(r * chunk, n) + 1, min((r + 1) * chunk, n))

    def _exchange_ranges(self, device):
        """(own, touched) of every rank -- one small all_gather at first use."""
        if self._ranges is None:
            xn = self.xneed if self.xneed is not None else (0, -1)
            mine = torch.tensor([self.own[0], self.own[1], self.touched[0], self.touched[1], xn[0], xn[1]],
                                dtype=torch.int64)
            if self.world > 1 or self.loopback:
                out = [torch.zeros(6, dtype=torch.int64, device=device) for _ in range(self.world)]
                dist.all_gather(out, mine.to(device), group=self.group)
                full = [tuple(int(v) for v in t.cpu()) for t in out]
            else:
                full = [tuple(int(v) for v in mine)]
            if self._phantom:
                # logical ranks of the loopback rehearsal: 0 = this rank's own role, 1 / 2 = the phantom neighbour's
                # rows below / above it (no blocks: they touch nothing and read nothing); all three are this process
                lo, hi = self.own
                big = 1 << 62
                full = [full[0], (1, lo - 1, 1, lo - 1, 1, 0), (hi + 1, big, hi + 1, big, 1, 0)]
                if self.xneed is None:
                    full[0] = full[0][:4] + (1, big)  # "reads everything": every x entry outside own comes from the phantom
            self._ranges = [t[:4] for t in full]
            self._xneeds = [t[4:] for t in full]
        return self._ranges

    def _peer(self, r):
        """physical rank behind logical rank r (loopback: every logical rank is this process)"""
        return self.rank if self._phantom else r

    def fetch_x(self, x):
        """x arrives PARTITIONED like y (every rank holds x[own] only, the rest of the full-length
        tensor is undefined): bring in what this rank's blocks read.  With `xneed` ranges (banded /
        symmetric operators: own range + a halo) that is one batched point-to-point exchange with the
        owners, received straight into x; without, the all-gather of the x slices."""
        ranges = self._exchange_ranges(x.device)
        if self.world == 1 and not self.loopback:
            return x
        own_ranges = [(rl, rh) for rl, rh, _, _ in ranges]
        if any(xn == (0, -1) for xn in self._xneeds):  # somebody reads everything: all-gather (collective)
            return self._allgather(x, own_ranges)
        if self._xplan is None or self._xplan[0] is not x:
            olo, ohi = self.own
            nlo, nhi = self._xneeds[self.rank]
            sends, recvs, staged = [], [], []
            for r, (rlo, rhi) in enumerate(own_ranges):
                if r == self.rank:
                    continue
                a, b = max(self._xneeds[r][0], olo), min(self._xneeds[r][1], ohi)  # what rank r reads of mine
                if a <= b:
                    sends.append((r, x[a - 1:b]))
                a, b = max(nlo, rlo), min(nhi, min(rhi, x.shape[0]))  # what I read of rank r's
                if a <= b:
                    recvs.append((r, x[a - 1:b]))
                    if self._phantom:
                        # the phantom owner's half of the transfer: its entries leave from a buffer of their own, and
                        # the place they are received into holds NaN until they have arrived
                        staged.append((torch.empty_like(x[a - 1:b]), x[a - 1:b]))
                        sends.append((r, staged[-1][0]))
            ops = [dist.P2POp(dist.isend, v, self._peer(r), group=self.group) for r, v in sends]
            ops += [dist.P2POp(dist.irecv, v, self._peer(r), group=self.group) for r, v in recvs]
            self._xplan = (x, ops, staged)  # the descriptors point at fixed views of x: built once
        ops = self._xplan[1]
        if ops:
            for src, view in self._xplan[2]:
                src.copy_(view)
                view.fill_(float("nan"))
            self._host_mediated_fence(x)
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return x

    def _workvec(self, y):
        if self._work is None or self._work.shape != y.shape or self._work.device != y.device or \
                self._work.dtype != y.dtype:
            self._work = torch.zeros_like(y)
            self._plan = None
        return self._work

    def _halo_plan(self, w, ranges):
        """Send views into the work vector and preallocated receive buffers, one pair per peer whose
        owned rows this rank touches / that touches this rank's rows."""
        if self._plan is None:
            olo, ohi = self.own
            tlo, thi = self.touched
            sends, recvs = [], []
            for r, (rlo, rhi, rtlo, rthi) in enumerate(ranges):
                if r == self.rank:
                    continue
                a, b = max(tlo, rlo), min(thi, rhi)  # my contributions to rank r's rows
                if a <= b:
                    sends.append((r, w[a - 1:b]))
                a, b = max(rtlo, olo), min(rthi, ohi)  # rank r's contributions to my rows
                if a <= b:
                    recvs.append((r, a, b, torch.empty(b - a + 1, dtype=w.dtype, device=w.device)))
            if self._phantom:
                # the phantom owners' half: they receive what this rank produced for their rows (and add it, below)
                n = w.shape[0]
                sends = [(r, v) for r, v in sends if v.shape[0] > 0]
                for r, v in sends:
                    a = int(v.storage_offset() - w.storage_offset()) + 1
                    recvs.append((r, a, a + v.shape[0] - 1, torch.empty_like(v)))
                assert all(b <= n for _, _, b, _ in recvs)
            ops = [dist.P2POp(dist.isend, v, self._peer(r), group=self.group) for r, v in sends]
            ops += [dist.P2POp(dist.irecv, buf, self._peer(r), group=self.group) for r, _, _, buf in recvs]
            self._plan = (ops, recvs)  # the descriptors point at fixed views / buffers: built once
        return self._plan

    def _phantom_rows(self, y, beta):
        """loopback=(lo, hi): the phantom neighbour's own part of a product -- it has no blocks, so its rows (everything
        outside lo..hi) are just scaled by beta; what this rank produced for them is added from the receive buffers"""
        lo, hi = self.own
        n = y.shape[0]
        if lo > 1:
            self._combine(y, slice(0, lo - 1), 0, beta)
        if hi < n:
            self._combine(y, slice(hi, n), 0, beta)

    def _exchange(self, ops, recvs, t):
        """one batched point-to-point exchange; loopback: receive buffers hold NaN until their transfer has arrived"""
        if self._phantom:
            for _, _, _, buf in recvs:
                buf.fill_(float("nan"))
        self._host_mediated_fence(t)
        for req in dist.batch_isend_irecv(ops):
            req.wait()

    def _host_mediated_fence(self, t):
        """RCCL orders its transfers against the compute stream by itself.  `gloo` (the CPU rehearsal
        backend) moves CUDA tensors through host staging on its own side streams and does not order a
        receive against kernels still reading the REUSED receive buffer: drain the stream first."""
        if t.is_cuda and dist.get_backend(self.group) != "nccl":
            torch.cuda.current_stream(t.device).synchronize()

    @staticmethod
    def _combine(y, sl, w, beta):
        if beta is False:
            y[sl] = w
        elif beta is True or beta == 1:
            y[sl] += w
        else:
            y[sl].mul_(beta).add_(w)

    def _local_op(self, op):
        if op not in self._ops:
            A = self.local
            self._ops[op] = A if op == M.L.BSM_OP_N else (M.transpose(A) if op == M.L.BSM_OP_T else M.adjoint(A))
        return self._ops[op]

    def mul(self, y, x, alpha=True, beta=False, local_mul=None, op=M.L.BSM_OP_N, x_distributed=False):
        """local_mul(work_or_y, x, alpha, beta[, op]): test hook replacing the HIP product (CPU gloo tests).
        x_distributed: x is valid only on this rank's own range (see fetch_x); products along the
        partition only."""
        N_ = M.L.BSM_OP_N
        if x_distributed:
            self.fetch_x(x)
        if local_mul is not None:
            lm = (lambda yy, xx, a, b: local_mul(yy, xx, a, b)) if op == N_ and self.axis == 0 else \
                (lambda yy, xx, a, b: local_mul(yy, xx, a, b, op))
        elif self.local is None:
            lm = None  # a rank without blocks: nothing to multiply, the collectives still run
        else:
            Aop = self._local_op(op)
            lm = lambda yy, xx, a, b: M.mul(yy, Aop, xx, a, b)
        along = self.symmetric or ((op == N_) == (self.axis == 0))
        ranges = self._exchange_ranges(y.device)
        if not along:
            return self._mul_across(y, x, alpha, beta, lm)
        # collective decision: if ANY rank touches rows it does not own, every rank takes part in
        # the exchange (a rank without a halo of its own may still receive contributions)
        halo = any((rl, rh) != (tl, th) for rl, rh, tl, th in ranges)
        olo, ohi = self.own
        if not halo and self.axis == 0 and op == N_:
            if lm is not None:
                lm(y, x, alpha, beta)  # rows outside `own` are left untouched by the handle
            elif ohi >= olo:
                self._combine(y, slice(olo - 1, ohi), 0, beta)
            if self._phantom:
                self._phantom_rows(y, beta)
        else:
            w = self._workvec(y)
            if lm is not None:
                lm(w, x, alpha, False)  # strong zero over the touched range, then accumulate
            elif ohi >= olo:
                w[olo - 1:ohi] = 0
            ops, recvs = self._halo_plan(w, ranges)
            if halo and ops:
                self._exchange(ops, recvs, w)
            if ohi >= olo:
                own_slice = slice(olo - 1, ohi)
                self._combine(y, own_slice, w[own_slice], beta)
            if self._phantom:
                self._phantom_rows(y, beta)
            for _, a, b, buf in recvs:
                y[a - 1:b] += buf
        if self.gather and (self.world > 1 or self.loopback):
            self._allgather(y, [(rl, rh) for rl, rh, _, _ in ranges])
        return y

    # ---- A * X: several right-hand sides -------------------------------------------------------------------------
    @staticmethod
    def _colmajor_like(t, rows):
        """(rows, K) matrix whose columns are contiguous, like a column-major t"""
        return torch.empty((t.shape[1], rows), dtype=t.dtype, device=t.device).t()

    def _p2p_columns(self, pairs, op):
        """one point-to-point descriptor per COLUMN of every (peer, matrix piece): the row slice of a column-major matrix
        is K contiguous runs, and all of them travel in the one batch of the exchange (one grouped RCCL call)"""
        return [dist.P2POp(op, v[:, k], self._peer(r), group=self.group) for r, v in pairs for k in range(v.shape[1])]

    def _fetch_x_multi(self, X):
        """fetch_x for an (n, K) column-major X partitioned like the rows"""
        ranges = self._exchange_ranges(X.device)
        if self.world == 1 and not self.loopback:
            return X
        own_ranges = [(rl, rh) for rl, rh, _, _ in ranges]
        if any(xn == (0, -1) for xn in self._xneeds):  # somebody reads everything: all-gather, column by column
            for k in range(X.shape[1]):
                self._allgather(X[:, k], own_ranges)
            return X
        if self._xplan_m is None or self._xplan_m[0] is not X:
            olo, ohi = self.own
            nlo, nhi = self._xneeds[self.rank]
            sends, recvs, staged = [], [], []
            for r, (rlo, rhi) in enumerate(own_ranges):
                if r == self.rank:
                    continue
                a, b = max(self._xneeds[r][0], olo), min(self._xneeds[r][1], ohi)  # what rank r reads of mine
                if a <= b:
                    sends.append((r, X[a - 1:b]))
                a, b = max(nlo, rlo), min(nhi, min(rhi, X.shape[0]))  # what I read of rank r's
                if a <= b:
                    recvs.append((r, X[a - 1:b]))
                    if self._phantom:  # (see fetch_x)
                        staged.append((self._colmajor_like(X, b - a + 1), X[a - 1:b]))
                        sends.append((r, staged[-1][0]))
            ops = self._p2p_columns(sends, dist.isend) + self._p2p_columns(recvs, dist.irecv)
            self._xplan_m = (X, ops, staged)
        ops = self._xplan_m[1]
        if ops:
            for src, view in self._xplan_m[2]:
                src.copy_(view)
                view.fill_(float("nan"))
            self._host_mediated_fence(X)
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return X

    def _halo_plan_multi(self, W, ranges):
        """_halo_plan for the (n, K) work matrix: send views of its rows, receive buffers with contiguous columns"""
        if self._plan_m is None or self._plan_m[0] is not W:
            olo, ohi = self.own
            tlo, thi = self.touched
            sends, recvs = [], []
            for r, (rlo, rhi, rtlo, rthi) in enumerate(ranges):
                if r == self.rank:
                    continue
                a, b = max(tlo, rlo), min(thi, min(rhi, W.shape[0]))  # my contributions to rank r's rows
                if a <= b:
                    sends.append((r, a, b))
                a, b = max(rtlo, olo), min(rthi, ohi)  # rank r's contributions to my rows
                if a <= b:
                    recvs.append((r, a, b, self._colmajor_like(W, b - a + 1)))
            if self._phantom:  # the phantom owners receive what this rank produced for their rows
                for r, a, b in sends:
                    recvs.append((r, a, b, self._colmajor_like(W, b - a + 1)))
            ops = self._p2p_columns([(r, W[a - 1:b]) for r, a, b in sends], dist.isend)
            ops += self._p2p_columns([(r, buf) for r, _, _, buf in recvs], dist.irecv)
            self._plan_m = (W, ops, recvs)
        return self._plan_m[1], self._plan_m[2]

    def mul_multi(self, Y, X, alpha=True, beta=False, local_mul=None, x_distributed=False):
        """Y = alpha * A * X + beta * Y for (n, K) COLUMN-major matrices (torch: `.t()` of a contiguous (K, n) tensor),
        products ALONG the partition (A * X of a row-partitioned operator, any op of a symmetric one): the local product
        is ONE multi-RHS product (bsm_mul_multi: A streamed once per batch of columns -- what LinearMaps' column loop of
        the reference, src/abstractblockmatrix.jl:27-34, does K times), and the K columns of every halo segment travel
        in the ONE batch of the exchange.  x_distributed: X is valid on the own rows only, as in mul().  After the call
        the own rows of Y are final (all of Y when gather=True).  local_mul(W_or_Y, X, alpha, beta): test hook."""
        if Y.dim() != 2 or X.dim() != 2 or X.shape[1] != Y.shape[1]:
            raise ValueError("mul_multi takes (n, K) matrices with the same number of columns")
        if Y.stride(0) != 1 or X.stride(0) != 1:
            raise ValueError("mul_multi takes COLUMN-major matrices (the columns are what the exchanges send)")
        if not (self.symmetric or self.axis == 0):
            raise ValueError("mul_multi: products along the partition only (mul() column by column runs across it)")
        if x_distributed:
            self._fetch_x_multi(X)
        if local_mul is not None:
            lm = local_mul
        elif self.local is None:
            lm = None
        else:
            A = self.local
            lm = lambda yy, xx, a, b: M.mul(yy, A, xx, a, b)
        ranges = self._exchange_ranges(Y.device)
        halo = any((rl, rh) != (tl, th) for rl, rh, tl, th in ranges)
        olo, ohi = self.own
        own_slice = slice(olo - 1, ohi) if ohi >= olo else None
        if not halo and self.axis == 0:
            if lm is not None:
                lm(Y, X, alpha, beta)  # rows outside `own` are left untouched by the handle
            elif own_slice is not None:
                self._combine(Y, own_slice, 0, beta)
            if self._phantom:
                self._phantom_rows(Y, beta)
        else:
            if self._work_m is None or self._work_m.shape != Y.shape or self._work_m.device != Y.device or \
                    self._work_m.dtype != Y.dtype:
                self._work_m = torch.zeros((Y.shape[1], Y.shape[0]), dtype=Y.dtype, device=Y.device).t()
                self._plan_m = None
            W = self._work_m
            if lm is not None:
                lm(W, X, alpha, False)  # strong zero over the touched rows, then accumulate
            elif own_slice is not None:
                W[own_slice] = 0
            ops, recvs = self._halo_plan_multi(W, ranges)
            if halo and ops:
                self._exchange(ops, recvs, W)
            if own_slice is not None:
                self._combine(Y, own_slice, W[own_slice], beta)
            if self._phantom:
                self._phantom_rows(Y, beta)
            for _, a, b, buf in recvs:
                Y[a - 1:b] += buf
        if self.gather and (self.world > 1 or self.loopback):
            for k in range(Y.shape[1]):
                self._allgather(Y[:, k], [(rl, rh) for rl, rh, _, _ in ranges])
        return Y

    def mul_overlapped(self, y, x, alpha=True, beta=False, local_mul=None, interior_mul=None):
        """Forward product with x and y PARTITIONED like the rows and the exchange overlapped with the
        interior rows (build_overlapped):

            side stream : x exchange -> boundary product into the work vector -> partial-y exchange
            main stream : interior product straight into y[own]          (needs nothing from anybody)
            main stream : y[own] += work vector, y[own] += received segments   (after the side stream)

        With RCCL the transfers run on the collective layer's own stream, ordered against the side stream
        only, so the interior launch -- nearly all of the rank's bytes -- streams while the halo
        travels; the reference has no counterpart (one process, `@tasks`: src/symmetricblockmatrix.jl:
        394-418 reads x and writes y in shared memory).  local_mul / interior_mul(y, x, alpha, beta):
        test hooks replacing the HIP products of the boundary / interior handle (CPU gloo tests)."""
        ranges = self._exchange_ranges(y.device)
        cuda = y.is_cuda
        olo, ohi = self.own
        tlo, thi = self.touched
        own_slice = slice(olo - 1, ohi) if ohi >= olo else None
        halo = any(th >= tl and (tl < rl or th > rh) for rl, rh, tl, th in ranges)
        if not halo and self.local is None and local_mul is None and self.world == 1:
            # nothing to exchange and no boundary block (one rank): the step IS the interior product -- no side stream to
            # fork and join, no work vector (a fork / join pair costs ~30 us of an otherwise idle queue per step)
            if interior_mul is not None:
                interior_mul(y, x, alpha, beta)
            elif self.interior is not None:
                M.mul(y, self.interior, x, alpha, beta)
            elif own_slice is not None:
                self._combine(y, own_slice, 0, beta)
            return y
        if cuda:
            main = torch.cuda.current_stream(y.device)
            if self._side is None:
                # a HIGH-PRIORITY stream: the runtime deals ordinary streams onto a handful of hardware queues per device
                # and the side stream landed on the main stream's own queue (kernel trace of the RCCL loopback step,
                # profiles/r05_loopback_timeline.txt: the interior launch sat behind the side stream's wait for the x
                # halo, nothing overlapped); priority streams get queues of their own
                self._side = torch.cuda.Stream(device=y.device, priority=-1)
            side = self._side
            side.wait_stream(main)
            side_ctx = torch.cuda.stream(side)
        else:
            import contextlib
            side_ctx = contextlib.nullcontext()
        with side_ctx:
            self.fetch_x(x)
        # interior rows: y[own] = alpha * A_int x[own] + beta * y[own]
        if interior_mul is not None:
            interior_mul(y, x, alpha, beta)
        elif self.interior is not None:
            M.mul(y, self.interior, x, alpha, beta)
        elif own_slice is not None:
            self._combine(y, own_slice, 0, beta)
        with side_ctx:
            w = self._workvec(y)
            if local_mul is not None:
                local_mul(w, x, alpha, False)
            elif self.local is not None:
                M.mul(w, self.local, x, alpha, False)  # strong zero over the rows the boundary blocks touch
            ops, recvs = self._halo_plan(w, ranges)
            if halo and ops:
                self._exchange(ops, recvs, w)
        if cuda:
            main.wait_stream(side)
        if self._phantom:
            self._phantom_rows(y, beta)
        # y[own] += the boundary blocks' sums for own rows + every received segment: ONE multi-tensor launch (each separate
        # add is a ~5 us launch on the critical path behind the join -- with 8 ranks the interior launch is ~0.57 ms)
        dst, src = [], []
        if own_slice is not None and thi >= tlo:
            a, b = max(olo, tlo), min(ohi, thi)
            if a <= b:
                dst.append(y[a - 1:b])
                src.append(w[a - 1:b])
        for _, a, b, buf in recvs:
            dst.append(y[a - 1:b])
            src.append(buf)
        if len(dst) == 1 or not cuda or any(self._overlap(p, q) for i, p in enumerate(dst) for q in dst[i + 1:]):
            for d_, s_ in zip(dst, src):  # (two segments for the same rows must not race inside one launch)
                d_ += s_
        elif dst:
            # (torch._foreach_add_ was tried first: its multi-tensor kernel took 38 us for three short segments)
            if self._segadd is None or not self._segadd.matches(y, dst, src):
                self._segadd = M.SegmentAdd(y, dst, src)
            self._segadd()
        return y

    @staticmethod
    def _overlap(p, q):
        a0, a1 = p.storage_offset(), p.storage_offset() + p.shape[0]
        b0, b1 = q.storage_offset(), q.storage_offset() + q.shape[0]
        return a0 < b1 and b0 < a1

    def _mul_across(self, y, x, alpha, beta, lm):
        """Every rank holds a full-length partial result: reduce-scatter onto equal chunks (or one
        all-reduce when the whole y is wanted on every rank)."""
        n = y.shape[0]
        w = self._workvec(y)
        if lm is not None:
            lm(w, x, alpha, False)
        else:
            w.zero_()
        if self.world == 1 and not self.loopback:
            self._combine(y, slice(0, n), w, beta)
            return y
        if self.gather:
            dist.all_reduce(w, group=self.group)
            self._combine(y, slice(0, n), w, beta)
            return y
        chunk = -(-n // self.world)
        if self._pad is None or self._pad.shape[0] != chunk * self.world or self._pad.device != y.device or \
                self._pad.dtype != y.dtype:
            self._pad = torch.zeros(chunk * self.world, dtype=y.dtype, device=y.device)
            self._rsout = torch.empty(chunk, dtype=y.dtype, device=y.device)
        self._pad[:n] = w
        if self.loopback:
            self._rsout.fill_(float("nan"))  # what ends up in y is what the collective delivered
        self._host_mediated_fence(w)
        dist.reduce_scatter_tensor(self._rsout, self._pad, group=self.group)
        lo, hi = self.out_range(n)
        if hi >= lo:
            self._combine(y, slice(lo - 1, hi), self._rsout[:hi - lo + 1], beta)
        return y

    def _allgather(self, y, own_ranges):
        # one all-gather of the (padded) own slices instead of one broadcast per rank: a single
        # collective whose per-peer messages (~n/N entries) use all xGMI links at once
        if self.loopback:
            own_ranges = own_ranges[:1]  # (the phantom roles hold their rows in this very y)
        maxlen = max(max(rh - rl + 1, 0) for rl, rh in own_ranges)
        if maxlen == 0:
            return y
        if self._gbuf is None or self._gbuf.shape[0] != self.world * maxlen or self._gbuf.device != y.device or \
                self._gbuf.dtype != y.dtype:
            self._gbuf = torch.empty(self.world * maxlen, dtype=y.dtype, device=y.device)
            self._sbuf = torch.zeros(maxlen, dtype=y.dtype, device=y.device)
        olo, ohi = self.own
        if ohi >= olo:
            self._sbuf[:ohi - olo + 1] = y[olo - 1:ohi]
            if self.loopback:
                y[olo - 1:ohi] = float("nan")  # comes back through the collective, or shows
        self._host_mediated_fence(y)
        dist.all_gather_into_tensor(self._gbuf, self._sbuf, group=self.group)
        for r, (rlo, rhi) in enumerate(own_ranges):
            if (r != self.rank or self.loopback) and rhi >= rlo:
                y[rlo - 1:rhi] = self._gbuf[r * maxlen:r * maxlen + (rhi - rlo + 1)]
        return y
