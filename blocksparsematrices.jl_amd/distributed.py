"""Single-node multi-GPU layer: one process per GPU (torch.distributed, backend "nccl" == RCCL
over xGMI), block rows partitioned over the ranks.

The reference has no distributed code at all (one process, shared-memory tasks); this layer is the
MI355X-first counterpart of its `@tasks` fan-out over block rows / colour classes.

  * VBCRS forward: block rows own disjoint y ranges (reference src/vbcrs.jl:275-283), so every
    rank multiplies its own rows and NO collective is needed; `gather=True` adds one all-gather of
    the y slices for callers that need the whole y on every GPU (a Krylov iteration).
  * SymmetricBlockMatrix: the off-diagonal block (I, J) lives with the owner of row set I but also
    contributes B^T x_I to y_J, which may belong to another rank (for banded operators: the
    previous rank only).  Each rank accumulates into a work vector over the rows it TOUCHES and the
    overlaps are exchanged point-to-point (isend/irecv == ncclSend/ncclRecv over the direct xGMI
    links) and added -- never a ring all-reduce of the full y, which would be bound by one link.
  * BlockSparseMatrix (index lists): blocks go with the rank that owns their smallest row index;
    rows a block reaches outside its rank's range travel through the same point-to-point exchange.
  * Products that run ACROSS the partition (transpose(A)*x of a row-partitioned VBCRS /
    BlockSparseMatrix, A*x of a column-partitioned one) give a full-length partial result on every
    rank: one reduce-scatter onto equal chunks (all-reduce when the caller wants the whole y).
    `split_vbcrs(..., axis=1)` is the column partition that makes the TRANSPOSED products
    collective-free (mirror of the forward case, reference src/vbcrs.jl:303-329).
"""
import numpy as np

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None

from . import matrices as M


def balanced_cuts(weights, nparts):
    """Cut len(weights) consecutive units into nparts contiguous ranges of ~equal total weight.
    Returns nparts+1 boundaries (unit indices)."""
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


def split_vbcrs(problem, rank, nparts, axis=0):
    """Partition a VBCRS problem dict into contiguous ranges of block rows (axis=0) or block columns
    (axis=1: for transposed products) balanced by stored bytes.
    Returns (local problem, own=(lo, hi) 1-based inclusive along `axis`)."""
    rs = np.asarray(problem["rowstart" if axis == 0 else "colstart"], dtype=np.int64)
    n = problem["size"][axis]
    starts = np.unique(rs)  # block rows, sorted
    bytes_per_row = np.zeros(len(starts))
    idx = np.searchsorted(starts, rs)
    for b, blk in enumerate(problem["blocks"]):
        bytes_per_row[idx[b]] += blk.size
    cuts = balanced_cuts(bytes_per_row, nparts)
    lo_row = int(starts[cuts[rank]]) if cuts[rank] < len(starts) else n + 1
    hi_row = int(starts[cuts[rank + 1]]) - 1 if cuts[rank + 1] < len(starts) else n
    if rank == 0:
        lo_row = 1
    keep = [b for b in range(len(rs)) if cuts[rank] <= idx[b] < cuts[rank + 1]]
    local = dict(kind="vbcrs", blocks=[problem["blocks"][b] for b in keep],
                 rowstart=np.asarray(problem["rowstart"], dtype=np.int64)[keep],
                 colstart=np.asarray(problem["colstart"], dtype=np.int64)[keep], size=problem["size"])
    return local, (lo_row, hi_row)


def split_blocksparse(problem, rank, nparts):
    """Partition a BlockSparseMatrix problem (arbitrary index lists): blocks sorted by their smallest
    row index, cut into ranges balanced by stored bytes; a rank owns the rows from its first block's
    smallest row up to the next rank's.  Returns (local problem, own, touched), 1-based inclusive."""
    n = problem["size"][0]
    first = np.array([int(np.min(r)) if len(r) else 1 for r in problem["rowindices"]], dtype=np.int64)
    order = np.argsort(first, kind="stable")
    w = np.array([problem["blocks"][b].size for b in order], dtype=np.float64)
    cuts = balanced_cuts(w, nparts)
    # blocks with the same smallest row stay together, so that the row ranges are disjoint
    for p in range(1, nparts):
        k = cuts[p]
        while 0 < k < len(order) and first[order[k]] == first[order[k - 1]]:
            k += 1
        cuts[p] = max(k, cuts[p - 1])
    lo = int(first[order[cuts[rank]]]) if cuts[rank] < len(order) else n + 1
    hi = int(first[order[cuts[rank + 1]]]) - 1 if cuts[rank + 1] < len(order) else n
    if rank == 0:
        lo = 1
    keep = [int(order[k]) for k in range(cuts[rank], cuts[rank + 1])]
    local = dict(kind="blocksparse", blocks=[problem["blocks"][b] for b in keep],
                 rowindices=[problem["rowindices"][b] for b in keep],
                 colindices=[problem["colindices"][b] for b in keep], size=problem["size"])
    tlo, thi = lo, hi
    for lst in local["rowindices"]:
        if len(lst):
            tlo, thi = min(tlo, int(np.min(lst))), max(thi, int(np.max(lst)))
    if thi < tlo:
        tlo, thi = lo, lo - 1
    return local, (lo, hi), (tlo, thi)


def split_symmetric(problem, rank, nparts):
    """Partition a SymmetricBlockMatrix problem by diagonal segments (balanced by stored bytes of the
    segment's diagonal block + the off-diagonal blocks whose rows start in it).
    Returns (local problem, own=(lo, hi), touched=(lo, hi)), 1-based inclusive."""
    n = problem["size"][0]
    dfirst = np.array([int(np.min(d)) for d in problem["diagonalindices"]], dtype=np.int64)
    order = np.argsort(dfirst, kind="stable")
    seg_start = dfirst[order]
    w = np.array([problem["diagonals"][i].size for i in order], dtype=np.float64)
    ofirst = np.array([int(np.min(r)) for r in problem["rowindices"]], dtype=np.int64)
    oseg = np.clip(np.searchsorted(seg_start, ofirst, side="right") - 1, 0, len(seg_start) - 1)
    for b, blk in enumerate(problem["offdiagonals"]):
        w[oseg[b]] += blk.size
    cuts = balanced_cuts(w, nparts)
    lo = int(seg_start[cuts[rank]]) if cuts[rank] < len(seg_start) else n + 1
    hi = int(seg_start[cuts[rank + 1]]) - 1 if cuts[rank + 1] < len(seg_start) else n
    if rank == 0:
        lo = 1
    dkeep = [int(order[k]) for k in range(cuts[rank], cuts[rank + 1])]
    okeep = [b for b in range(len(ofirst)) if cuts[rank] <= oseg[b] < cuts[rank + 1]]
    local = dict(kind="symmetric", diagonals=[problem["diagonals"][i] for i in dkeep],
                 diagonalindices=[problem["diagonalindices"][i] for i in dkeep],
                 offdiagonals=[problem["offdiagonals"][b] for b in okeep],
                 rowindices=[problem["rowindices"][b] for b in okeep],
                 colindices=[problem["colindices"][b] for b in okeep], size=problem["size"])
    tlo, thi = lo, hi
    for lst in local["diagonalindices"] + local["rowindices"] + local["colindices"]:
        tlo, thi = min(tlo, int(np.min(lst))), max(thi, int(np.max(lst)))
    if thi < tlo:
        tlo, thi = lo, lo - 1
    return local, (lo, hi), (tlo, thi)


class RowPartitioned:
    """y = alpha*op(A)*x + beta*y with A's blocks spread over the ranks of `group`.

    `axis` = 0: `own` / `touched` are ROW ranges (forward products are local up to the halo);
    `axis` = 1: column ranges of a column-partitioned operator (the transposed products are).
    `symmetric`: op(A) has A's structure, every op runs along the partition.

    x must hold the FULL vector on every rank.  After mul() the rank's output range is final in y
    (the whole y when gather=True): `own` for products along the partition, `out_range(n)` (equal
    chunks) for products across it.  `local` is this rank's matrix (built with own=touched range so
    its beta pass covers exactly the rows it touches)."""

    def __init__(self, local, own, touched=None, group=None, gather=False, axis=0, symmetric=None):
        self.local = local
        self.own = (int(own[0]), int(own[1]))
        self.touched = self.own if touched is None else (int(touched[0]), int(touched[1]))
        self.group = group
        self.gather = gather
        self.axis = axis
        self.symmetric = isinstance(local, M.SymmetricBlockMatrix) if symmetric is None else symmetric
        self.rank = dist.get_rank(group) if dist is not None and dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist is not None and dist.is_initialized() else 1
        self._ranges = None
        self._work = None
        self._gbuf = None
        self._sbuf = None
        self._pad = None

    def out_range(self, n, rank=None):
        """Output rows (1-based, inclusive) rank `rank` holds after a product ACROSS the partition."""
        r = self.rank if rank is None else rank
        chunk = -(-n // self.world)
        return (min(r * chunk, n) + 1, min((r + 1) * chunk, n))

    def _exchange_ranges(self):
        """(own, touched) of every rank -- one small all_gather at first use."""
        if self._ranges is None:
            mine = torch.tensor([self.own[0], self.own[1], self.touched[0], self.touched[1]], dtype=torch.int64)
            if self.world > 1:
                dev = self._device
                out = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(self.world)]
                dist.all_gather(out, mine.to(dev), group=self.group)
                self._ranges = [tuple(int(v) for v in t.cpu()) for t in out]
            else:
                self._ranges = [tuple(int(v) for v in mine)]
        return self._ranges

    def _workvec(self, y):
        if self._work is None or self._work.shape != y.shape or self._work.device != y.device or \
                self._work.dtype != y.dtype:
            self._work = torch.zeros_like(y)
        return self._work

    @staticmethod
    def _combine(y, sl, w, beta):
        if beta is False:
            y[sl] = w
        else:
            y[sl] = y[sl] * (1 if beta is True else beta) + w

    def mul(self, y, x, alpha=True, beta=False, local_mul=None, op=M.L.BSM_OP_N):
        """local_mul(work_or_y, x, alpha, beta[, op]): test hook replacing the HIP product (CPU gloo tests)."""
        self._device = y.device
        N_ = M.L.BSM_OP_N
        if local_mul is not None:
            lm = (lambda yy, xx, a, b: local_mul(yy, xx, a, b)) if op == N_ and self.axis == 0 else \
                (lambda yy, xx, a, b: local_mul(yy, xx, a, b, op))
        else:
            Aop = self.local if op == N_ else (M.transpose(self.local) if op == M.L.BSM_OP_T else M.adjoint(self.local))
            lm = lambda yy, xx, a, b: M.mul(yy, Aop, xx, a, b)
        along = self.symmetric or ((op == N_) == (self.axis == 0))
        ranges = self._exchange_ranges()
        if not along:
            return self._mul_across(y, x, alpha, beta, lm)
        # collective decision: if ANY rank touches rows it does not own, every rank takes part in
        # the exchange (a rank without a halo of its own may still receive contributions)
        halo = any((rl, rh) != (tl, th) for rl, rh, tl, th in ranges)
        if not halo and self.axis == 0 and op == N_:
            lm(y, x, alpha, beta)  # rows outside `own` are left untouched by the handle
        else:
            w = self._workvec(y)
            lm(w, x, alpha, False)  # strong zero over the touched range, then accumulate
            ops, recvs = [], []
            olo, ohi = self.own
            tlo, thi = self.touched
            for r, (rlo, rhi, rtlo, rthi) in enumerate(ranges):
                if r == self.rank or not halo:
                    continue
                a, b = max(tlo, rlo), min(thi, rhi)  # my contributions to rank r's rows
                if a <= b:
                    ops.append(dist.P2POp(dist.isend, w[a - 1:b].contiguous(), r, group=self.group))
                a, b = max(rtlo, olo), min(rthi, ohi)  # rank r's contributions to my rows
                if a <= b:
                    buf = torch.empty(b - a + 1, dtype=y.dtype, device=y.device)
                    recvs.append((a, b, buf))
                    ops.append(dist.P2POp(dist.irecv, buf, r, group=self.group))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            own_slice = slice(olo - 1, ohi)
            self._combine(y, own_slice, w[own_slice], beta)
            for a, b, buf in recvs:
                y[a - 1:b] += buf
        if self.gather and self.world > 1:
            self._allgather(y, [(rl, rh) for rl, rh, _, _ in ranges])
        return y

    def _mul_across(self, y, x, alpha, beta, lm):
        """Every rank holds a full-length partial result: reduce-scatter onto equal chunks (or one
        all-reduce when the whole y is wanted on every rank)."""
        n = y.shape[0]
        w = self._workvec(y)
        lm(w, x, alpha, False)
        if self.world == 1:
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
        self._pad[:n] = w
        out = torch.empty(chunk, dtype=y.dtype, device=y.device)
        dist.reduce_scatter(out, list(self._pad.view(self.world, chunk).unbind(0)), group=self.group)
        lo, hi = self.out_range(n)
        if hi >= lo:
            self._combine(y, slice(lo - 1, hi), out[:hi - lo + 1], beta)
        return y

    def _allgather(self, y, own_ranges):
        # one all-gather of the (padded) own slices instead of one broadcast per rank: a single
        # collective whose per-peer messages (~n/N entries) use all xGMI links at once
        maxlen = max(max(rh - rl + 1, 0) for rl, rh in own_ranges)
        if self._gbuf is None or self._gbuf.shape[0] != self.world * maxlen or self._gbuf.device != y.device or \
                self._gbuf.dtype != y.dtype:
            self._gbuf = torch.empty(self.world * maxlen, dtype=y.dtype, device=y.device)
            self._sbuf = torch.zeros(maxlen, dtype=y.dtype, device=y.device)
        olo, ohi = self.own
        if ohi >= olo:
            self._sbuf[:ohi - olo + 1] = y[olo - 1:ohi]
        dist.all_gather(list(self._gbuf.view(self.world, maxlen).unbind(0)), self._sbuf, group=self.group)
        for r, (rlo, rhi) in enumerate(own_ranges):
            if r != self.rank and rhi >= rlo:
                y[rlo - 1:rhi] = self._gbuf[r * maxlen:r * maxlen + (rhi - rlo + 1)]
        return y
