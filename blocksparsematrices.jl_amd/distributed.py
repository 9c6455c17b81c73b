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


def split_vbcrs(problem, rank, nparts):
    """Row-partition a VBCRS problem dict: contiguous ranges of block rows balanced by stored bytes.
    Returns (local problem, own=(lo, hi) 1-based inclusive)."""
    rs = np.asarray(problem["rowstart"], dtype=np.int64)
    n = problem["size"][0]
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
    local = dict(kind="vbcrs", blocks=[problem["blocks"][b] for b in keep], rowstart=rs[keep],
                 colstart=np.asarray(problem["colstart"], dtype=np.int64)[keep], size=problem["size"])
    return local, (lo_row, hi_row)


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
    """y = alpha*A*x + beta*y with A's block rows spread over the ranks of `group`.

    x must hold the FULL vector on every rank; after mul(), y[own range] is final on every rank
    (the whole y when gather=True).  `local` is this rank's matrix (built with own=touched range so
    its beta pass covers exactly the rows it touches)."""

    def __init__(self, local, own, touched=None, group=None, gather=False):
        self.local = local
        self.own = (int(own[0]), int(own[1]))
        self.touched = self.own if touched is None else (int(touched[0]), int(touched[1]))
        self.group = group
        self.gather = gather
        self.rank = dist.get_rank(group) if dist is not None and dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist is not None and dist.is_initialized() else 1
        self._ranges = None
        self._work = None
        self._gbuf = None
        self._sbuf = None

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

    def mul(self, y, x, alpha=True, beta=False, local_mul=None):
        """local_mul(work_or_y, x, alpha, beta): test hook replacing the HIP product (CPU gloo tests)."""
        self._device = y.device
        lm = local_mul if local_mul is not None else (lambda yy, xx, a, b: M.mul(yy, self.local, xx, a, b))
        # collective decision: if ANY rank touches rows it does not own, every rank takes part in
        # the exchange (a rank without a halo of its own may still receive contributions)
        ranges = self._exchange_ranges()
        halo = any((rl, rh) != (tl, th) for rl, rh, tl, th in ranges)
        if not halo:
            lm(y, x, alpha, beta)  # rows outside `own` are left untouched by the handle
        else:
            if self._work is None or self._work.shape != y.shape or self._work.device != y.device:
                self._work = torch.zeros_like(y)
            w = self._work
            lm(w, x, alpha, False)  # strong zero over the touched range, then accumulate
            ops, recvs = [], []
            olo, ohi = self.own
            tlo, thi = self.touched
            for r, (rlo, rhi, rtlo, rthi) in enumerate(ranges):
                if r == self.rank:
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
            if beta is False:
                y[own_slice] = w[own_slice]
            else:
                y[own_slice] = y[own_slice] * (1 if beta is True else beta) + w[own_slice]
            for a, b, buf in recvs:
                y[a - 1:b] += buf
        if self.gather and self.world > 1:
            # one all-gather of the (padded) own slices instead of one broadcast per rank: a single
            # collective whose per-peer messages (~n/N entries) use all xGMI links at once
            ranges = self._exchange_ranges()
            maxlen = max(max(rh - rl + 1, 0) for rl, rh, _, _ in ranges)
            if self._gbuf is None or self._gbuf.shape[0] != self.world * maxlen or self._gbuf.device != y.device:
                self._gbuf = torch.empty(self.world * maxlen, dtype=y.dtype, device=y.device)
                self._sbuf = torch.zeros(maxlen, dtype=y.dtype, device=y.device)
            olo, ohi = self.own
            if ohi >= olo:
                self._sbuf[:ohi - olo + 1] = y[olo - 1:ohi]
            dist.all_gather(list(self._gbuf.view(self.world, maxlen).unbind(0)), self._sbuf, group=self.group)
            for r, (rlo, rhi, _, _) in enumerate(ranges):
                if r != self.rank and rhi >= rlo:
                    y[rlo - 1:rhi] = self._gbuf[r * maxlen:r * maxlen + (rhi - rlo + 1)]
        return y
