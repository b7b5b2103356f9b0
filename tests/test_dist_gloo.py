"""CPU, world_size 2, gloo: the N>1 path's host logic — slab bounds and the slab exchange
(all-gather when nv divides evenly, zero-padded all-reduce otherwise) — with the oracle's
ladder contraction standing in for the HIP kernel."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pymes_amd import dist as pdist


def test_slab_bounds_cover_range():
    for nv in (1, 7, 25, 200, 203):
        for w in (1, 2, 3, 8):
            b = [pdist.slab_bounds(nv, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == nv
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, nv, no, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        V = rng.standard_normal((nv, nv, nv, nv))
        T = rng.standard_normal((nv, nv, no, no))
        lo, hi = pdist.slab_bounds(nv, rank, world)
        full = torch.zeros((nv, nv * no * no), dtype=torch.float64)
        full[lo:hi] = torch.from_numpy(np.einsum("abcd,cdij->abij", V[lo:hi], T).reshape(hi - lo, -1))
        assert pdist.world()[:2] == (rank, world)
        pdist.exchange_slabs(full, nv, rank, world)
        ref = np.einsum("abcd,cdij->abij", V, T).reshape(nv, -1)
        out[rank] = float(np.abs(full.numpy() - ref).max())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nv", [6, 7])
def test_exchange_two_ranks(nv):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, nv, 2, out), nprocs=2, join=True)
    assert len(out) == 2 and max(out.values()) < 1e-12
