"""One-process-per-GPU sharding of the particle-particle ladder (ccd.py:187).

The ladder is a GEMM whose rows are the (a,b) virtual pairs: rows of R[(a,b),(i,j)] for the plain form
(a-slabs), rows P(a,b), a >= b, of the pair-packed result for the symmetric form.  Rows are cut into
``world`` equal chunks of ceil(n/world) rows (the last chunk may be short), each rank computes its
chunk with the fp64 MFMA GEMM and the chunks are exchanged with ONE all-gather over RCCL/xGMI
(``torch.distributed``; backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests).  Chunks are disjoint,
so the all-gather moves 1/world of the bytes of the zero-padded all-reduce the north star names.
"""
import os

# ---- rehearsal of one rank's share on a single GPU --------------------------------------------------------------------
# ``stub(rank, world)``: the solvers take their one-process-per-GPU path as rank ``rank`` of ``world`` while every
# collective is a no-op (the buffers of the other ranks stay as they are: timings are those of the rank's compute,
# energies are meaningless).  bench.py --stub-collectives --as-rank r --of N: a compute-only 1/2/4/8 curve on one GPU.
_STUB = None


def stub(rank, world_size):
    global _STUB
    if not (0 <= int(rank) < int(world_size)):
        raise ValueError("stub: need 0 <= rank < world")
    _STUB = (int(rank), int(world_size))


def stubbed():
    return _STUB is not None


class Trace:
    """Per-phase device time and exposed communication of the sharded iteration (bench.py switches it on).

    ``mark(name)`` records an event on the engine's stream (= torch's current stream under torch.distributed,
    ``bind_stream``); the time between two marks is charged to the later mark's name.  Every collective handle records
    an event pair around its ``wait()``: what the stream stood still for it — the EXPOSED part of the transfer (≈ 0 when
    it was hidden behind kernels) — and the bytes it put on the wire are summed per label."""

    def __init__(self):
        self.on = False
        self.reset()

    def reset(self):
        self._marks, self._waits, self.bytes, self.calls, self._host = [], [], {}, {}, []

    def enable(self, on=True):
        self.on = bool(on)
        self.reset()

    stream = None        # torch stream the events are recorded on (None: torch's current stream, which the engine is bound
                         # to under torch.distributed); bench.py sets the engine's own stream for single-rank runs

    def _event(self):
        import torch
        if not torch.cuda.is_available():
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(self.stream if self.stream is not None else torch.cuda.current_stream())
        return ev

    def mark(self, name):
        if self.on:
            import time
            self._marks.append((name, self._event()))
            self._host.append((name, time.perf_counter()))

    def sent(self, label, nbytes):
        if self.on:
            self.bytes[label] = self.bytes.get(label, 0) + int(nbytes)
            self.calls[label] = self.calls.get(label, 0) + 1

    def waiting(self, label):
        return _WaitSpan(self, label)

    def summary(self, steps):
        """({phase: ms per step}, {label: {"exposed_wait_ms", "wire_bytes", "calls"} per step}); synchronises."""
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        phases, waits = {}, {}
        for (_, e0), (name, e1) in zip(self._marks, self._marks[1:]):
            if name == "begin" or e0 is None or e1 is None:
                continue
            phases[name] = phases.get(name, 0.0) + e0.elapsed_time(e1)
        for label, e0, e1 in self._waits:
            if e0 is not None and e1 is not None:
                waits[label] = waits.get(label, 0.0) + e0.elapsed_time(e1)
        steps = max(1, int(steps))
        # host wall time between the same marks: a phase whose host time is close to its device time is host-bound (the
        # GPU stood idle waiting for launches or for a host decision), one with little host time was enqueued ahead
        host = {}
        for (_, t0), (name, t1) in zip(self._host, self._host[1:]):
            if name != "begin":
                host[name] = host.get(name, 0.0) + 1e3 * (t1 - t0)
        self.host_ms = {k: v / steps for k, v in host.items()}
        coll = {label: {"exposed_wait_ms": waits.get(label, 0.0) / steps, "wire_bytes": self.bytes.get(label, 0) / steps,
                        "calls": self.calls.get(label, 0) / steps} for label in sorted(set(self.bytes) | set(waits))}
        return {k: v / steps for k, v in phases.items()}, coll


class _WaitSpan:
    def __init__(self, tr, label):
        self.tr, self.label = tr, label

    def __enter__(self):
        self.e0 = self.tr._event() if self.tr.on else None

    def __exit__(self, *exc):
        if self.tr.on:
            self.tr._waits.append((self.label, self.e0, self.tr._event()))
        return False


trace = Trace()


def world():
    """(rank, world_size, local_rank) from torch.distributed if initialised, else (0, 1, 0)."""
    if _STUB is not None:
        return _STUB[0], _STUB[1], 0
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size(), int(os.environ.get("LOCAL_RANK", dist.get_rank()))
    except ImportError:
        pass
    return 0, 1, 0


def chunk_rows(n_rows, world_size):
    return -(-int(n_rows) // int(world_size))


def slab_rows(n_rows, rank, world_size):
    """Rows [lo, hi) owned by ``rank``; chunks are equal (padded), so hi - lo may be short or 0 at the end."""
    c = chunk_rows(n_rows, world_size)
    lo = min(rank * c, n_rows)
    return lo, min(lo + c, n_rows)


def padded_rows(n_rows, world_size):
    """Row count of the exchange buffer: world_size equal chunks."""
    return chunk_rows(n_rows, world_size) * world_size


def forced():
    """PYMES_FORCE_SHARDED=1: take the one-process-per-GPU code path (exchange buffers, collectives, pair-sharded tail)
    even in a world of one rank.  This is how the RCCL calls are exercised on a box with a single GPU (a communicator
    of one rank still goes through ``init_process_group("nccl")`` and every collective entry point)."""
    return bool(os.environ.get("PYMES_FORCE_SHARDED"))


def sharded():
    """True when the solvers should run their sharded form."""
    if _STUB is not None:
        return True
    try:
        import torch.distributed as dist
    except ImportError:
        return False
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or forced()


def _staged(t):
    """The exchange goes through host memory (test rigs: gloo cannot touch device memory; CPU tensors of the host
    simulator).  Only then are host fences needed: under RCCL the
    engine runs on torch's current stream (``bind_stream``), the collective is ordered behind the kernels already
    enqueued there and ``wait()`` orders the stream behind the collective — no host synchronisation at all."""
    import torch.distributed as dist
    return (not t.is_cuda) or dist.get_backend() != "nccl"


def bind_stream(ctx):
    """One process per GPU under RCCL: run the engine on torch's current stream, so that engine kernels and
    collectives are ordered on the device without host fences.  No-op for the host simulator / CPU tensors."""
    import torch
    if ctx.lib.backend.startswith("hip") and torch.cuda.is_available():
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)


def _fence_before(t, ctx):
    if ctx is not None:
        ctx.sync()


def _fence_after(t):
    if t.is_cuda:
        import torch
        torch.cuda.current_stream().synchronize()


def exchange_rows(full, rank, world_size, ctx=None):
    """``full`` is a torch tensor [padded_rows, ...] whose chunk ``rank`` was computed locally (by ``ctx``'s engine).
    On return every rank holds every chunk (stream-ordered under RCCL, complete on the host otherwise)."""
    import torch.distributed as dist
    if not sharded() or _STUB is not None:
        return full
    c = full.shape[0] // world_size
    if _staged(full):
        _fence_before(full, ctx)
        host = full.cpu() if full.is_cuda else full
        mine = host[rank * c:(rank + 1) * c].clone()      # all_gather_into_tensor must not alias its output
        dist.all_gather_into_tensor(host.view(-1), mine.view(-1))
        if full.is_cuda:
            full.copy_(host)
            _fence_after(full)
        return full
    mine = full[rank * c:(rank + 1) * c].clone()
    dist.all_gather_into_tensor(full.view(-1), mine.view(-1))
    return full


def allreduce_sum(vec):
    """Sum of a small float64 numpy vector over the ranks (identical result on every rank)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    if not sharded() or _STUB is not None:
        return np.asarray(vec, dtype=np.float64)
    trace.sent("scalars", 2 * 8 * len(vec))
    t = torch.from_numpy(np.array(vec, dtype=np.float64, copy=True))
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def _ring_bytes(t, world_size, passes):
    """Bytes one rank puts on the wire for a ring collective over ``t``: (world-1)/world of the buffer per pass
    (all-gather: 1 pass; all-reduce = reduce-scatter + all-gather: 2)."""
    return passes * t.numel() * t.element_size() * (world_size - 1) // max(world_size, 1)


def allreduce_tensor_start(t, ctx=None, label="allreduce"):
    """In-place sum of a torch tensor over the ranks; returns a handle with ``wait()`` (async under RCCL)."""
    import torch.distributed as dist
    if not sharded():
        return _Done()
    trace.sent(label, _ring_bytes(t, world()[1], 2))
    if _STUB is not None:
        return _Done()
    if _staged(t):
        _fence_before(t, ctx)
        if t.is_cuda:                               # test rig: several ranks on one GPU, gloo cannot reduce device memory
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)
            _fence_after(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return _Done()
    return _Pending(dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True), t, label)


class _Done:
    def wait(self):
        return True


class _Pending:
    """A collective in flight; keeps its send buffer alive until waited for.  ``wait()`` makes torch's current stream
    (= the engine's stream) wait for the collective; the host does not block."""

    def __init__(self, work, send, label="collective", after=None):
        self.work, self.send, self.label, self.after = work, send, label, after

    def wait(self):
        with trace.waiting(self.label):
            self.work.wait()
        if self.after is not None:          # e.g. received tiles copied to their places, on the stream that just waited
            self.after()
            self.after = None
        self.send = None
        return True


def exchange_rows_start(full, rank, world_size, ctx=None, label="allgather"):
    """Asynchronous form of ``exchange_rows``: returns a handle whose ``wait()`` makes the current stream wait for
    the all-gather (RCCL runs it on its own stream, so kernels enqueued in between overlap with the transfer).
    The staged test rigs fall back to the blocking exchange."""
    import torch.distributed as dist
    if not sharded():
        return _Done()
    trace.sent(label, _ring_bytes(full, world_size, 1))
    if _STUB is not None:
        return _Done()
    if _staged(full):
        exchange_rows(full, rank, world_size, ctx)
        _fence_after(full)
        return _Done()
    c = full.shape[0] // world_size
    # IN PLACE: the rank's rows are sent from where they lie in the gathered buffer (ncclAllGather with sendbuff = recvbuff +
    # rank * count is the in-place form RCCL defines) — no send clone: at (50,200) on eight ranks 0.3 GB of copies per
    # iteration, and in the one-rank rehearsal (PYMES_FORCE_SHARDED) the whole 2.6 GB twice, once into the clone and once
    # "through the wire" back.  PYMES_ALLGATHER_CLONE=1 restores the separate send buffer.
    mine = full[rank * c:(rank + 1) * c]
    if os.environ.get("PYMES_ALLGATHER_CLONE"):
        mine = mine.clone()
    return _Pending(dist.all_gather_into_tensor(full.view(-1), mine.view(-1), async_op=True), mine, label)


def alltoall_start(send, send_n, recv, recv_n, ctx=None, label="owner tiles"):
    """All-to-all of contiguous pieces (``send_n[q]`` doubles for rank q out of ``send``, ``recv_n[p]`` from rank p into ``recv``,
    both in rank order): the exchange behind ``pymes_set_alltoallv``.  Asynchronous under RCCL (ncclSend / ncclRecv in a group),
    blocking and staged through the host in the gloo rigs."""
    import torch.distributed as dist
    if not sharded():
        return _Done()
    trace.sent(label, 8 * sum(send_n))
    if _STUB is not None:
        return _Done()
    if _staged(send):
        _fence_before(send, ctx)
        hs = send.cpu() if send.is_cuda else send
        hr = recv.cpu() if recv.is_cuda else recv
        dist.all_to_all_single(hr, hs, list(recv_n), list(send_n))
        if recv.is_cuda:
            recv.copy_(hr)
        _fence_after(recv)
        return _Done()
    return _Pending(dist.all_to_all_single(recv, send, list(recv_n), list(send_n), async_op=True), (send, recv), label)


# ---- the collective table of include/pymes_amd.h (pymes_collectives), filled with torch.distributed ------------------------------
class Collectives:
    """What a host program hands to ``pymes_set_collectives``: the library's one-process-per-GPU step
    (``pymes_ccsd_sharded_residuals`` / ``_finish``) calls back for every all-reduce / all-gather, naming device memory by
    pointer.  Here the exchange buffers are torch tensors (``buffers``: name -> tensor, the memory the solver allocated
    for them), the pointer is mapped back to a view of its tensor and the collective is the same ``torch.distributed`` call
    the Python-sequenced path makes (RCCL on the GPUs, gloo staged through the host in the test rigs, a no-op under
    ``stub``).  An RCCL host in another language fills the same table with ncclAllReduce / ncclAllGather."""

    LABELS = {"W": "fock intermediates", "Xvv": "X_ac", "R1": "R1", "S": "scalars", "ETd": "ETd", "ETx": "ETx", "QK": "QK",
              "Tall": "new T2", "L": "L"}

    def __init__(self, ctx, buffers, rank, world_size):
        import ctypes as C
        self.ctx, self.rank, self.world = ctx, int(rank), int(world_size)
        self._buffers = sorted(((t.data_ptr(), t.numel(), name, t) for name, t in buffers.items()), key=lambda x: x[0])
        self._tickets, self._next, self.error = {}, 1, None
        start_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64))
        wait_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_void_p)
        mark_t = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p)

        class Table(C.Structure):
            _fields_ = [("user", C.c_void_p), ("rank", C.c_int), ("world", C.c_int), ("allreduce_start", start_t),
                        ("allgather_start", start_t), ("wait", wait_t), ("mark", mark_t)]
        self._callbacks = (start_t(self._allreduce), start_t(self._allgather), wait_t(self._wait), mark_t(self._mark))
        self.table = Table(None, self.rank, self.world, *self._callbacks)
        ctx.lib.call("pymes_set_collectives", ctx.handle, C.byref(self.table))
        ctx._collectives = self              # the callbacks must outlive the context's use of them

    def _view(self, ptr, n):
        for base, numel, name, t in self._buffers:
            if base <= ptr < base + 8 * numel:
                off = (ptr - base) // 8
                if off + n > numel:
                    raise ValueError("collective hook: %d doubles at offset %d run past buffer %s" % (n, off, name))
                return t.view(-1)[off:off + n], name, off
        raise ValueError("collective hook: pointer %#x is none of the exchange buffers" % ptr)

    def _ticket(self, work, ticket_p):
        k = self._next
        self._next += 1
        self._tickets[k] = work
        ticket_p[0] = k
        return 0

    def _allreduce(self, user, buf, n, stream, ticket_p):
        try:
            t, name, off = self._view(int(buf), int(n))
            label = ("X_ki" if off == 0 else "hole-ladder J") if name == "P" else self.LABELS.get(name, name)
            return self._ticket(allreduce_tensor_start(t, self.ctx, label=label), ticket_p)
        except BaseException as exc:         # nothing may propagate through the C frames: the library reports the failure
            self.error = exc
            return 1

    def _allgather(self, user, buf, chunk, stream, ticket_p):
        try:
            t, name, _ = self._view(int(buf), int(chunk) * self.world)
            full = t.view(self.world, int(chunk))
            return self._ticket(exchange_rows_start(full, self.rank, self.world, self.ctx, label=self.LABELS.get(name, name)),
                                ticket_p)
        except BaseException as exc:
            self.error = exc
            return 1

    def enable_owner_tiles(self, device):
        """The optional all-to-all of the table (``pymes_set_alltoallv``) and the two staging buffers of the owner-tile
        exchange (``pymes_owner_tile_sizes``): the rows of the ring products then travel as the tiles each pair owner reads
        (flag PYMES_OWNER_TILES of the sharded residual steps)."""
        import ctypes as C
        import torch
        ns, nr = C.c_int64(), C.c_int64()
        self.ctx.lib.call("pymes_owner_tile_sizes", self.ctx.handle, self.rank, self.world, C.byref(ns), C.byref(nr))
        self.xs = torch.zeros((int(ns.value),), dtype=torch.float64, device=device)
        self.xr = torch.zeros((int(nr.value),), dtype=torch.float64, device=device)
        a2a_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64), C.c_void_p,
                            C.POINTER(C.c_int64))
        self._a2a = a2a_t(self._alltoallv)
        self.ctx.lib.call("pymes_set_alltoallv", self.ctx.handle, C.cast(self._a2a, C.c_void_p))
        self.ctx.lib.call("pymes_set_owner_tile_buffers", self.ctx.handle, C.c_void_p(self.xs.data_ptr()), C.c_void_p(self.xr.data_ptr()))

    def _alltoallv(self, user, send, send_counts, recv, recv_counts, stream, ticket_p):
        try:
            ns = [int(send_counts[q]) for q in range(self.world)]
            nr = [int(recv_counts[q]) for q in range(self.world)]
            if int(send) != self.xs.data_ptr() or int(recv) != self.xr.data_ptr() or sum(ns) > self.xs.numel() or sum(nr) > self.xr.numel():
                raise ValueError("collective hook: all-to-all outside the owner-tile staging buffers")
            return self._ticket(alltoall_start(self.xs[:sum(ns)], ns, self.xr[:sum(nr)], nr, self.ctx, label="ETd+ETx owner tiles"),
                                ticket_p)
        except BaseException as exc:
            self.error = exc
            return 1

    def _wait(self, user, ticket, stream):
        try:
            self._tickets.pop(int(ticket)).wait()
            return 0
        except BaseException as exc:
            self.error = exc
            return 1

    def _mark(self, user, phase):
        trace.mark(phase.decode())

    def call(self, name, *args):
        """``ctx.lib.call`` for an entry that may call back: a failure inside a callback is re-raised as what it was."""
        self.error = None
        try:
            return self.ctx.lib.call(name, *args)
        except Exception as exc:
            if self.error is not None:
                raise self.error from exc
            raise


# ---- owner-tile exchange of the ring-product rows ------------------------------------------------------------------------
# In the pair-sharded tail rank q assembles R only for its virtual pairs P(a,b), a >= b, a in [a0_q, a1_q): of the pair
# matrices ETd / ETx (rows = this rank's column slab, see Engine::residual_slab) it reads the tiles [(a,.),(b,.)] and
# [(b,.),(a,.)] only, i.e. rows [0, a1 o) x columns [a0 o, a1 o) and rows [a0 o, a1 o) x columns [0, a0 o).  Sending
# every rank just those rectangles is an all-to-all of ~2 (a1-a0) a1 o^2 doubles per rank and matrix (0.2 GB at
# (50,200) on 8 ranks) instead of the all-gather of the whole matrix (0.8 GB each): PYMES_OWNER_TILES=1.
def owner_tile_plan(no, nv, world_size):
    """plan[p][q] = rectangles (r0, r1, c0, c1) of a pair matrix [ov x ov] that the owner p of rows
    ``slab_rows(ov, p)`` sends to rank q (empty for p == q: those rows are already in place)."""
    ov, npp = no * nv, nv * (nv + 1) // 2
    plan = [[[] for _ in range(world_size)] for _ in range(world_size)]
    for q in range(world_size):
        a0, a1 = a_range_of_pair_rows(*slab_rows(npp, q, world_size))
        if a1 <= a0:
            continue
        A0, A1 = a0 * no, a1 * no
        for p in range(world_size):
            if p == q:
                continue
            r0, r1 = slab_rows(ov, p, world_size)
            lo, hi = r0, min(r1, A1)
            if hi > lo:
                plan[p][q].append((lo, hi, A0, A1))                      # rows (b,.), b < a1;  columns (a,.)
            lo, hi = max(r0, A0), min(r1, A1)
            if hi > lo and A0 > 0:
                plan[p][q].append((lo, hi, 0, A0))                       # rows (a,.);  columns (b,.), b < a0
    return plan


def exchange_pair_tiles_start(mats, no, nv, rank, world_size, ctx=None, label="owner tiles"):
    """Owner-tile exchange of the pair matrices ``mats`` (torch tensors [padded ov, ov], this rank's row slab filled in):
    one all-to-all for all of them.  Returns a handle; after ``wait()`` every rectangle this rank's assembly reads is in
    place (bit-identical to what the all-gather would have put there)."""
    import torch
    import torch.distributed as dist
    if not sharded():
        return _Done()
    plan = owner_tile_plan(no, nv, world_size)
    area = lambda rects: sum((r1 - r0) * (c1 - c0) for r0, r1, c0, c1 in rects)
    send_n = [len(mats) * area(plan[rank][q]) for q in range(world_size)]
    recv_n = [len(mats) * area(plan[p][rank]) for p in range(world_size)]
    trace.sent(label, 8 * sum(send_n))
    if _STUB is not None:
        return _Done()
    staged = _staged(mats[0])
    if staged:
        _fence_before(mats[0], ctx)
    src = [m.cpu() if (staged and m.is_cuda) else m for m in mats]
    pieces = [m[r0:r1, c0:c1].reshape(-1) for q in range(world_size) for m in src for (r0, r1, c0, c1) in plan[rank][q]]
    dev = src[0].device
    send = torch.cat(pieces) if pieces else torch.empty(0, dtype=torch.float64, device=dev)
    recv = torch.empty(sum(recv_n), dtype=torch.float64, device=dev)

    def scatter():
        off = 0
        for p in range(world_size):
            for m in mats:
                for (r0, r1, c0, c1) in plan[p][rank]:
                    n = (r1 - r0) * (c1 - c0)
                    m[r0:r1, c0:c1].copy_(recv[off:off + n].view(r1 - r0, c1 - c0))
                    off += n
    if staged:
        dist.all_to_all_single(recv, send, recv_n, send_n)
        scatter()
        _fence_after(mats[0])
        return _Done()
    return _Pending(dist.all_to_all_single(recv, send, recv_n, send_n, async_op=True), (send, recv), label, after=scatter)


def a_range_of_pair_rows(lo, hi):
    """Virtual indices a touched by the packed pair rows P(a,b) = a(a+1)/2 + b in [lo, hi)."""
    if hi <= lo:
        return 0, 0

    def a_of(r):
        a = int(((8.0 * r + 1.0) ** 0.5 - 1.0) / 2.0)
        while a * (a + 1) // 2 > r:
            a -= 1
        while (a + 1) * (a + 2) // 2 <= r:
            a += 1
        return a
    return a_of(lo), a_of(hi - 1) + 1
