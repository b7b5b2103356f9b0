"""One-process-per-GPU sharding of the particle-particle ladder (ccd.py:187).

The ladder is a GEMM whose rows are the (a,b) virtual pairs: rows of R[(a,b),(i,j)] for the plain form
(a-slabs), rows P(a,b), a >= b, of the pair-packed result for the symmetric form.  Rows are cut into
``world`` equal chunks of ceil(n/world) rows (the last chunk may be short), each rank computes its
chunk with the fp64 MFMA GEMM and the chunks are exchanged with ONE all-gather over RCCL/xGMI
(``torch.distributed``; backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests).  Chunks are disjoint,
so the all-gather moves 1/world of the bytes of the zero-padded all-reduce the north star names.
"""
import os


def world():
    """(rank, world_size, local_rank) from torch.distributed if initialised, else (0, 1, 0)."""
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
    if not sharded():
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
    if not sharded():
        return np.asarray(vec, dtype=np.float64)
    t = torch.from_numpy(np.array(vec, dtype=np.float64, copy=True))
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def allreduce_tensor_start(t, ctx=None):
    """In-place sum of a torch tensor over the ranks; returns a handle with ``wait()`` (async under RCCL)."""
    import torch.distributed as dist
    if not sharded():
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
    if os.environ.get("PYMES_SYNC_EXCHANGE"):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        _fence_after(t)
        return _Done()
    return _Pending(dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True), t)


class _Done:
    def wait(self):
        return True


class _Pending:
    """A collective in flight; keeps its send buffer alive until waited for.  ``wait()`` makes torch's current stream
    (= the engine's stream) wait for the collective; the host does not block."""

    def __init__(self, work, send):
        self.work, self.send = work, send

    def wait(self):
        self.work.wait()
        self.send = None
        return True


def exchange_rows_start(full, rank, world_size, ctx=None):
    """Asynchronous form of ``exchange_rows``: returns a handle whose ``wait()`` makes the current stream wait for
    the all-gather (RCCL runs it on its own stream, so kernels enqueued in between overlap with the transfer).
    The staged test rigs and PYMES_SYNC_EXCHANGE=1 fall back to the blocking exchange."""
    import torch.distributed as dist
    if not sharded():
        return _Done()
    if _staged(full) or os.environ.get("PYMES_SYNC_EXCHANGE"):
        exchange_rows(full, rank, world_size, ctx)
        _fence_after(full)
        return _Done()
    c = full.shape[0] // world_size
    mine = full[rank * c:(rank + 1) * c].clone()
    return _Pending(dist.all_gather_into_tensor(full.view(-1), mine.view(-1), async_op=True), mine)


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
