"""One-process-per-GPU sharding of the particle-particle ladder (ccd.py:187).

The virtual index ``a`` of R[a,b,i,j] = sum_cd V[a,b,c,d] T[c,d,i,j] is split into one
contiguous slab per rank; every rank holds all of T2 and (in this round) all integral
blocks, computes its slab with the fp64 MFMA GEMM and the slabs are exchanged with ONE
collective over RCCL/xGMI (``torch.distributed``, backend "nccl" = RCCL on ROCm; "gloo" in
the CPU tests).  Slabs are disjoint, so an all-gather of nv/world rows each moves 1/world
of the bytes of the all-reduce the north star names; the zero-padded all-reduce is used
only when nv is not divisible by the world size.
"""
import os


def world():
    """(rank, world_size, local_rank) from torch.distributed if initialised, else the env, else (0,1,0)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size(), int(os.environ.get("LOCAL_RANK", dist.get_rank()))
    except ImportError:
        pass
    return 0, 1, 0


def slab_bounds(nv, rank, world_size):
    """Contiguous, balanced a-ranges: the first nv % world ranks get one extra row."""
    base, extra = divmod(int(nv), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def exchange_slabs(full, nv, rank, world_size):
    """``full`` is a torch tensor [nv, ...] whose rows slab_bounds(nv, rank, world) hold this
    rank's ladder slab (other rows are ignored / must be zero for the all-reduce path).
    On return every rank holds all slabs."""
    import torch.distributed as dist
    if world_size == 1:
        return full
    if nv % world_size == 0:
        lo, hi = slab_bounds(nv, rank, world_size)
        mine = full[lo:hi].clone()          # all_gather_into_tensor must not alias its output
        dist.all_gather_into_tensor(full.view(-1), mine.view(-1))
    else:
        dist.all_reduce(full, op=dist.ReduceOp.SUM)
    return full


class TorchAllocator:
    """Backs DeviceArrays with torch CUDA tensors so they can be handed to torch.distributed."""

    def __init__(self, device_index):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", device_index)

    def __call__(self, n_doubles):
        t = self.torch.empty(max(int(n_doubles), 1), dtype=self.torch.float64, device=self.device)
        return t.data_ptr(), t
