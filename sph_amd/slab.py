"""Slab domain decomposition along x: host-side orchestration of one rank per GPU.

The reference is a single shared-memory process (OpenMP only, reference compute/compute.cpp:131-287);
it has no distributed counterpart.  Interactions are strictly local (every kernel vanishes beyond
r = h, compute.cpp:44,53,64,76,88), so the tank is cut into slabs on cell-column boundaries and each
rank talks only to its two neighbours:

    records_left, records_right = backend.pack()          # migrants + 2 ghost columns per face
    from_left, from_right = exchange(records_left, records_right)
    backend.append(from_left); backend.append(from_right)
    backend.step()                                        # afterwards: own particles only

The exchange is point-to-point (torch.distributed isend/irecv; backend "nccl" is RCCL over xGMI, each
neighbour pair on its own direct link); no collective sits on the data path.  Load balancing moves
the cut planes by at most one column per step towards equal particle counts, using a per-column
histogram all-gathered every `rebalance_every` steps (a few KB, off the data path).

This module contains no physics: `backend` is sph_amd.capi.SlabSimulation (HIP) in production.
"""
import numpy as np

from .capi import MAX_FACE_MOVE, MIN_SLAB_COLUMNS, RECORD_WORDS  # noqa: F401


# ---- cut planes ---------------------------------------------------------------------------------------

def partition_columns(hist, nranks, min_cols=MIN_SLAB_COLUMNS):
    """Cut `len(hist)` columns into nranks contiguous slabs with near-equal particle counts.

    Returns cuts[0..nranks] with cuts[0] = 0, cuts[-1] = len(hist); slab r owns columns
    [cuts[r], cuts[r+1]).  Every slab gets at least min_cols columns.
    """
    hist = np.asarray(hist, dtype=np.int64)
    ncols = hist.size
    if nranks * min_cols > ncols:
        raise ValueError(f"{ncols} columns cannot hold {nranks} slabs of >= {min_cols} columns")
    csum = np.concatenate([[0], np.cumsum(hist)])
    total = csum[-1]
    cuts = [0]
    for r in range(1, nranks):
        target = total * r / nranks
        c = int(np.searchsorted(csum, target, side="left"))
        # pick the boundary whose cumulative count is closest to the target
        if c > 0 and abs(csum[c - 1] - target) <= abs(csum[min(c, ncols)] - target):
            c -= 1
        lo = cuts[-1] + min_cols
        hi = ncols - (nranks - r) * min_cols
        cuts.append(int(min(max(c, lo), hi)))
    cuts.append(ncols)
    return cuts


def min_columns_for_shift(max_shift_columns, face_move=1):
    """Narrowest slab that one exchange per step between direct neighbours can serve.

    A particle that crosses D columns in a step must still land in its old owner's own or neighbouring
    slab (+- 2 ghost columns): every slab has to be ceil(D) + 2 columns wide, + the columns the cut planes
    themselves move by in a step.
    """
    return max(MIN_SLAB_COLUMNS, int(np.ceil(max_shift_columns)) + 2 + int(face_move))


def face_move_for_shift(max_shift_columns, cuts=None):
    """Columns a cut plane may travel per step: as many as the fastest particle crosses, so that the planes can follow a
    front (a dam-break front runs at 2-4 columns per step), within what one exchange covers.  Every column of travel
    is a column of minimum slab width (min_columns_for_shift), so where fluid piles up against a wall and the slabs
    there want to become narrow, the planes slow down rather than keep the slabs wide: given the cuts, the move is
    limited to what the narrowest slab can afford."""
    move = int(min(MAX_FACE_MOVE, max(1, np.ceil(max_shift_columns))))
    if cuts is not None:
        narrowest = min(b - a for a, b in zip(cuts, cuts[1:]))
        # one column less than the narrowest slab could afford: a slab under pressure can then still narrow by a column
        # per call, down to the width that planes moving one column per step allow
        move = max(1, min(move, narrowest - int(np.ceil(max_shift_columns)) - 3))
    return move


def rebalance_cuts(cuts, hist, max_shift=1, min_cols=MIN_SLAB_COLUMNS, max_cols=None):
    """Move every interior cut by at most max_shift columns towards the balanced partition.  max_cols: the widest each
    slab may become (a context's column capacity), one number or one per slab."""
    nranks = len(cuts) - 1
    ncols = cuts[-1]
    min_cols = min(min_cols, ncols // nranks)  # never ask for more than the tank can give
    ideal = partition_columns(hist, nranks, min_cols)
    new = list(cuts)
    for r in range(1, nranks):
        step = int(np.clip(ideal[r] - cuts[r], -max_shift, max_shift))
        new[r] = cuts[r] + step
    # keep the minimum width after the individual moves
    for r in range(1, nranks):
        new[r] = max(new[r], new[r - 1] + min_cols)
    for r in range(nranks - 1, 0, -1):
        new[r] = min(new[r], new[r + 1] - min_cols)
    # ... but never by more than max_shift per call: the exchange of one step covers a face that moved one
    # column, not more (a slab that is still too narrow keeps widening on the following calls)
    for r in range(1, nranks):
        new[r] = int(np.clip(new[r], cuts[r] - max_shift, cuts[r] + max_shift))
    if max_cols is not None:
        cap = [int(max_cols)] * nranks if np.isscalar(max_cols) else [int(c) for c in max_cols]
        fit = list(new)
        for r in range(nranks - 1):          # a slab that would outgrow its capacity hands columns to its right neighbour
            fit[r + 1] = min(fit[r + 1], fit[r] + cap[r])
        for r in range(nranks - 1, 0, -1):   # ... and the last ones to their left neighbours
            fit[r] = max(fit[r], fit[r + 1] - cap[r])
        ok = all(fit[r + 1] - fit[r] <= cap[r] and fit[r + 1] - fit[r] >= min_cols for r in range(nranks)) and \
            all(abs(fit[r] - cuts[r]) <= max_shift for r in range(1, nranks))
        if ok:
            new = fit
        elif any(new[r + 1] - new[r] > cap[r] for r in range(nranks)):
            # plane by plane from where they stand: a plane takes its move if both slabs beside it stay admissible
            work = list(cuts)
            for r in range(1, nranks):
                left, right = new[r] - work[r - 1], work[r + 1] - new[r]
                if min_cols <= left <= cap[r - 1] and min_cols <= right <= cap[r]:
                    work[r] = new[r]
            new = work
    return new


# ---- exchanges --------------------------------------------------------------------------------------------

class LocalExchange:
    """All slabs live in one process (tests; several slabs on one GPU): hand the buffers over directly."""

    def __init__(self, nranks):
        self.nranks = nranks
        self.outbox = [None] * nranks

    def post(self, rank, left, right):
        self.outbox[rank] = (left, right)

    def collect(self, rank):
        from_left = self.outbox[rank - 1][1] if rank > 0 else None
        from_right = self.outbox[rank + 1][0] if rank < self.nranks - 1 else None
        return from_left, from_right


class DistExchange:
    """Neighbour exchange over torch.distributed point-to-point operations.

    With the "nccl" backend (RCCL on ROCm) the record tensors stay in HBM and travel over xGMI;
    with "gloo" they are staged through host memory (CPU tests, single-GPU rehearsals).

    A message's size must be known to both ends before it is posted.  The first step therefore takes
    two rounds (counts, then payload of exactly that size).  From then on one round per step: both ends
    size the next message from the count of the last one plus `headroom` - the sender knows what it sent,
    the receiver what it got, so they agree without talking - and the message carries its true record
    count in a header record.  The rare message that outgrows its agreed size sends the remainder in a
    second round, whose size both ends derive from that header.
    """

    def __init__(self, rank, nranks, device=None, group=None, headroom=1.25, min_capacity=1024):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.nranks, self.group = rank, nranks, group
        self.on_device = dist.get_backend(group) == "nccl"
        self.device = device if self.on_device else torch.device("cpu")
        self.data_device = device if device is not None else torch.device("cpu")
        self.bytes_sent = 0
        self.rounds = 0
        self.headroom, self.min_capacity = float(headroom), int(min_capacity)
        self.cap_out = {}   # neighbour rank -> records the next message to it holds
        self.cap_in = {}    # neighbour rank -> records the next message from it holds
        self.buf_out, self.buf_in = {}, {}

    def _neighbours(self):
        left = self.rank - 1 if self.rank > 0 else None
        right = self.rank + 1 if self.rank < self.nranks - 1 else None
        return left, right

    def _capacity_after(self, count):
        want = int(count * self.headroom) + 1
        return max(self.min_capacity, (want + 255) // 256 * 256)

    def _buffer(self, table, peer, rows):
        t = table.get(peer)
        if t is None or t.shape[0] < rows:
            t = self.torch.empty((rows, RECORD_WORDS), dtype=self.torch.float32, device=self.device)
            table[peer] = t
        return t

    def _batch(self, ops):
        if ops:
            self.rounds += 1
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()

    def exchange(self, send_left, send_right):
        left, right = self._neighbours()
        peers = [(left, send_left), (right, send_right)]
        if any(p is not None and p not in self.cap_out for p, _ in peers):
            return self._exchange_two_rounds(send_left, send_right)
        torch, dist = self.torch, self.dist
        ops, sent = [], {}
        for peer, send in peers:
            if peer is None:
                continue
            k, cap = int(send.shape[0]), self.cap_out[peer]
            msg = self._buffer(self.buf_out, peer, cap + 1)
            msg[0].fill_(float(k))  # header record: the true count (exact in fp32 far beyond any halo)
            m = min(k, cap)
            if m:
                msg[1:1 + m].copy_(send[:m] if self.on_device else send[:m].detach().to("cpu"))
            sent[peer] = (k, send)
            ops.append(dist.P2POp(dist.isend, msg[:cap + 1], peer, self.group))
            ops.append(dist.P2POp(dist.irecv, self._buffer(self.buf_in, peer, self.cap_in[peer] + 1)[:self.cap_in[peer] + 1],
                                  peer, self.group))
            self.bytes_sent += 4 * RECORD_WORDS * (cap + 1)
        self._batch(ops)
        heads = [self.buf_in[p][0, 0] for p, _ in peers if p is not None]
        counts = [int(v) for v in torch.stack(heads).cpu().tolist()] if heads else []
        got, more = {}, []
        for (peer, _), k_in in zip([x for x in peers if x[0] is not None], counts):
            cap = self.cap_in[peer]
            got[peer] = [self.buf_in[peer][1:1 + min(k_in, cap)]]
            if k_in > cap:  # the remainder follows, exactly k_in - cap records
                rest = torch.empty((k_in - cap, RECORD_WORDS), dtype=torch.float32, device=self.device)
                got[peer].append(rest)
                more.append(dist.P2POp(dist.irecv, rest, peer, self.group))
            self.cap_in[peer] = self._capacity_after(k_in)
        for peer, (k, send) in sent.items():
            cap = self.cap_out[peer]
            if k > cap:
                more.append(dist.P2POp(dist.isend, self._out(send[cap:]), peer, self.group))
                self.bytes_sent += 4 * RECORD_WORDS * (k - cap)
            self.cap_out[peer] = self._capacity_after(k)
        self._batch(more)
        empty = torch.empty((0, RECORD_WORDS), dtype=torch.float32, device=self.device)
        out = []
        for peer in (left, right):
            parts = got.get(peer, [empty])
            out.append(self._in(parts[0] if len(parts) == 1 else torch.cat(parts)))
        return out[0], out[1]

    def _exchange_two_rounds(self, send_left, send_right):
        torch, dist = self.torch, self.dist
        left, right = self._neighbours()
        # 1. counts (two ints per neighbour)
        counts_out = torch.tensor([send_left.shape[0] if left is not None else 0,
                                   send_right.shape[0] if right is not None else 0],
                                  dtype=torch.int64, device=self.device)
        from_left_n = torch.zeros(1, dtype=torch.int64, device=self.device)
        from_right_n = torch.zeros(1, dtype=torch.int64, device=self.device)
        ops = []
        if left is not None:
            ops.append(dist.P2POp(dist.isend, counts_out[0:1], left, self.group))
            ops.append(dist.P2POp(dist.irecv, from_left_n, left, self.group))
        if right is not None:
            ops.append(dist.P2POp(dist.isend, counts_out[1:2], right, self.group))
            ops.append(dist.P2POp(dist.irecv, from_right_n, right, self.group))
        self._batch(ops)
        nl, nr = int(from_left_n.item()), int(from_right_n.item())
        # 2. payload, exact sizes
        recv_left = torch.empty((nl, RECORD_WORDS), dtype=torch.float32, device=self.device)
        recv_right = torch.empty((nr, RECORD_WORDS), dtype=torch.float32, device=self.device)
        ops = []
        if left is not None:
            if send_left.shape[0]:
                ops.append(dist.P2POp(dist.isend, self._out(send_left), left, self.group))
            if nl:
                ops.append(dist.P2POp(dist.irecv, recv_left, left, self.group))
            self.cap_out[left] = self._capacity_after(send_left.shape[0])
            self.cap_in[left] = self._capacity_after(nl)
        if right is not None:
            if send_right.shape[0]:
                ops.append(dist.P2POp(dist.isend, self._out(send_right), right, self.group))
            if nr:
                ops.append(dist.P2POp(dist.irecv, recv_right, right, self.group))
            self.cap_out[right] = self._capacity_after(send_right.shape[0])
            self.cap_in[right] = self._capacity_after(nr)
        self._batch(ops)
        self.bytes_sent += 4 * RECORD_WORDS * (send_left.shape[0] + send_right.shape[0])
        return self._in(recv_left), self._in(recv_right)

    def _out(self, t):
        return t.contiguous() if self.on_device else t.detach().to("cpu").contiguous()

    def _in(self, t):
        return t if t.device == self.data_device else t.to(self.data_device)

    def allgather_histogram(self, local_hist, cuts, ncols, local_shift=0.0):
        """Global per-column particle histogram from every rank's own-column counts, and the largest of the
        ranks' column shifts, in ONE all-reduce: counts are summed into their columns, and every rank adds its
        shift (in 1/1024 column) into a slot of its own behind them."""
        torch, dist = self.torch, self.dist
        full = torch.zeros(ncols + self.nranks, dtype=torch.int64)
        lo = cuts[self.rank]
        full[lo:lo + len(local_hist)] = torch.as_tensor(np.asarray(local_hist, dtype=np.int64))
        full[ncols + self.rank] = int(np.ceil(float(local_shift) * 1024.0))
        full = full.to(self.device)
        dist.all_reduce(full, group=self.group)
        full = full.cpu().numpy()
        return full[:ncols], float(full[ncols:].max()) / 1024.0


# ---- per-rank driver ----------------------------------------------------------------------------------------

class SlabRunner:
    """Drives one slab.  `backend` supplies pack / append / step / column_counts / set_columns."""

    def __init__(self, backend, exchange, rank, nranks, cuts, rebalance_every=0):
        self.backend, self.exchange = backend, exchange
        self.rank, self.nranks = rank, nranks
        self.cuts = list(cuts)
        self.rebalance_every = rebalance_every
        self.steps_done = 0
        self.min_cols = MIN_SLAB_COLUMNS

    def step(self):
        b = self.backend
        send_left, send_right = b.pack()
        from_left, from_right = self.exchange.exchange(send_left, send_right)
        b.append(from_left)
        b.append(from_right)
        b.step()
        self.steps_done += 1
        if self.rebalance_every and self.steps_done % self.rebalance_every == 0 and self.nranks > 1:
            self.rebalance()

    def rebalance(self):
        b = self.backend
        if hasattr(b, "load_info"):
            counts, own_shift = b.load_info()
        else:
            counts, own_shift = b.column_counts(), (b.max_column_shift() if hasattr(b, "max_column_shift") else 0.0)
        hist, shift = self.exchange.allgather_histogram(counts, self.cuts, self.cuts[-1], own_shift)
        self.min_cols = min_columns_for_shift(shift)
        new = rebalance_cuts(self.cuts, hist, min_cols=self.min_cols)
        if new != self.cuts:
            self.cuts = new
            b.set_columns(new[self.rank], new[self.rank + 1])


def step_local_slabs(backends, exchange):
    """One step of several slabs that live in ONE process (LocalExchange)."""
    for r, b in enumerate(backends):
        left, right = b.pack()
        exchange.post(r, left, right)
    for r, b in enumerate(backends):
        from_left, from_right = exchange.collect(r)
        b.append(from_left)
        b.append(from_right)
    for b in backends:
        b.step()


def rebalance_local_slabs(backends, cuts):
    """Rebalancing counterpart of step_local_slabs: returns the new cuts and applies them."""
    hist = np.zeros(cuts[-1], np.int64)
    for r, b in enumerate(backends):
        c = b.column_counts()
        hist[cuts[r]:cuts[r] + len(c)] = c
    shift = max((b.max_column_shift() for b in backends if hasattr(b, "max_column_shift")), default=0.0)
    new = rebalance_cuts(cuts, hist, min_cols=min_columns_for_shift(shift))
    if new != list(cuts):
        for r, b in enumerate(backends):
            b.set_columns(new[r], new[r + 1])
    return new


# ---- exchange issued by the library (sph_slab_step_begin / _end) ---------------------------------------------------------

def halo_local_slabs(backends):
    """Initial halo exchange of several attached slabs that live in ONE process: begin on all, then end on all."""
    for b in backends:
        b.halo_begin()
    for b in backends:
        b.halo_end()


def step_local_slabs_overlapped(backends):
    """One overlapped step of several attached slabs that live in ONE process."""
    for b in backends:
        b.step_begin()
    for b in backends:
        b.step_end()


def rebalance_next_local_slabs(backends, cuts, by="work", stats=None, max_move=None):
    """Load balancing for the library-driven exchange: the new cuts apply to the records of the next step.
    by="work" balances the sum of (9 + neighbours) per slab - what a step costs - instead of the particle count.
    `stats`, a dict, receives the load per slab under the cuts as they stood (key "load") and the column shift.
    max_move: columns a face may travel in this call (default: as many as the fastest particle crosses per step)."""
    hist = np.zeros(cuts[-1], np.int64)
    shift = 0.0
    for r, b in enumerate(backends):
        first, counts, work, s = b.load_work()  # the columns of the last sort: they lag the cut planes by a step
        c = work if by == "work" else counts
        hist[first:first + len(c)] = c
        shift = max(shift, s)
    if stats is not None:
        stats["load"] = [int(hist[cuts[r]:cuts[r + 1]].sum()) for r in range(len(backends))]
        stats["shift"] = shift
    move = face_move_for_shift(shift, cuts) if max_move is None else max_move
    caps = [getattr(b, "cols_capacity", cuts[-1]) or cuts[-1] for b in backends]
    new = rebalance_cuts(cuts, hist, max_shift=move, min_cols=min_columns_for_shift(shift, move), max_cols=caps)
    for r, b in enumerate(backends):
        b.set_face_shift(shift)
        b.set_next_columns(new[r], new[r + 1])
    return new


class LibraryRunner:
    """Drives one slab per process with the exchange issued by the library (RCCL from C++, overlapped with the interior
    columns).  torch.distributed is used only off the data path: to hand the communicator id to the ranks and to
    all-reduce the column histogram when the cut planes are rebalanced."""

    def __init__(self, backend, rank, nranks, cuts, rebalance_every=0, group=None, reduce_device=None, balance_by="work"):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.balance_by = balance_by  # "work": sum of (9 + neighbours) per slab; "count": particles per slab
        self.backend, self.rank, self.nranks = backend, rank, nranks
        self.cuts = list(cuts)
        self.rebalance_every = rebalance_every
        self.steps_done = 0
        self._pending = None  # (tensor, work handle) of an all-reduce that the next step's end collects
        self.reduce_device = reduce_device if reduce_device is not None else torch.device("cpu")

    def prime(self):
        self.backend.halo_begin()
        self.backend.halo_end()

    def step(self):
        b = self.backend
        b.step_begin()
        b.step_end()
        self.steps_done += 1
        if self._pending is not None:
            self._finish_rebalance()
        if self.rebalance_every and self.steps_done % self.rebalance_every == 0 and self.nranks > 1:
            self._post_rebalance()

    def drain(self):
        """Collect an all-reduce that is still outstanding (end of a run)."""
        if self._pending is not None:
            self._finish_rebalance()

    def rebalance(self):
        """Synchronous form: histogram, all-reduce and new cut planes between two steps."""
        self._post_rebalance()
        self._finish_rebalance()

    def _post_rebalance(self):
        # The all-reduce (host, a few KB) is started now and collected after the NEXT step: it runs beside that step's
        # launches instead of between two steps, at the price of cut planes that follow the load one step later.
        b = self.backend
        lo, counts, work, own_shift = b.load_work()  # the columns of the last sort: they lag the cut planes by a step
        if self.balance_by == "work":
            counts = work
        ncols = self.cuts[-1]
        full = self.torch.zeros(ncols + self.nranks, dtype=self.torch.int64)
        full[lo:lo + len(counts)] = self.torch.as_tensor(np.asarray(counts, dtype=np.int64))
        full[ncols + self.rank] = int(np.ceil(float(own_shift) * 1024.0))
        full = full.to(self.reduce_device)
        self._pending = (full, self.dist.all_reduce(full, group=self.group, async_op=True))

    def _finish_rebalance(self):
        b = self.backend
        full, work = self._pending
        self._pending = None
        work.wait()
        full = full.cpu().numpy()
        ncols = self.cuts[-1]
        # one column of margin: the shift was measured up to rebalance_every + 1 steps before the faces it sizes are used
        raw = float(full[ncols:].max()) / 1024.0
        hist, shift = full[:ncols], raw + 1.0
        move = face_move_for_shift(raw, self.cuts)
        new = rebalance_cuts(self.cuts, hist, max_shift=move, min_cols=min_columns_for_shift(shift, move))
        b.set_face_shift(shift)
        b.set_next_columns(new[self.rank], new[self.rank + 1])
        self.cuts = new
