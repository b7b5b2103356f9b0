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


def test_slab_rows_cover_range():
    for n in (1, 7, 25, 200, 203, 20100):
        for w in (1, 2, 3, 8):
            b = [pdist.slab_rows(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            c = pdist.chunk_rows(n, w)
            assert all(hi - lo <= c for lo, hi in b) and pdist.padded_rows(n, w) == c * w >= n


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _pairs(n):
    return [(x, y) for x in range(n) for y in range(x + 1)]


def _worker(rank, world, port, nv, no, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        V = rng.standard_normal((nv, nv, nv, nv))
        V = 0.5 * (V + V.transpose(1, 0, 3, 2))           # V_abcd = V_badc
        T = rng.standard_normal((nv, nv, no, no))
        T = 0.5 * (T + T.transpose(1, 0, 3, 2))           # T_cdij = T_dcji
        ref = np.einsum("abcd,cdij->abij", V, T)
        assert pdist.world()[:2] == (rank, world)
        # (1) plain rows (a,b) of R
        rows = nv * nv
        lo, hi = pdist.slab_rows(rows, rank, world)
        full = torch.zeros((pdist.padded_rows(rows, world), no * no), dtype=torch.float64)
        full[lo:hi] = torch.from_numpy(V.reshape(rows, -1)[lo:hi] @ T.reshape(nv * nv, -1))
        pdist.exchange_rows(full, rank, world)
        err1 = float(np.abs(full.numpy()[:rows] - ref.reshape(rows, -1)).max())
        # (2) pair-packed rows P(a,b): the oracle's ladder restricted to a >= b, i >= j
        pr = _pairs(nv)
        lo, hi = pdist.slab_rows(len(pr), rank, world)
        L = torch.zeros((pdist.padded_rows(len(pr), world), no * no), dtype=torch.float64)
        for r in range(lo, hi):
            a, b = pr[r]
            k = 0
            for i, j in _pairs(no):
                L[r, k] = 0.5 * (ref[a, b, i, j] + ref[a, b, j, i]); k += 1          # LS
            for i, j in _pairs(no):
                if i > j:
                    L[r, k] = 0.5 * (ref[a, b, i, j] - ref[a, b, j, i]); k += 1      # LA
        # the overlapped form: two exchanges in flight, local work in between, then wait (as CCSD.iterate does)
        full2 = torch.zeros_like(full)
        full2[slice(*pdist.slab_rows(rows, rank, world))] = float(rank + 1)
        pending = [pdist.exchange_rows_start(L, rank, world), pdist.exchange_rows_start(full2, rank, world)]
        _ = float((full * 2.0).sum())
        for w in pending:
            assert w.wait()
        c = pdist.chunk_rows(rows, world)
        for r in range(world):
            blk = full2[r * c:min((r + 1) * c, rows)]
            assert blk.numel() == 0 or bool((blk == float(r + 1)).all())
        Ln = L.numpy()
        opp = no * (no + 1) // 2
        rec = np.zeros_like(ref)
        for a in range(nv):
            for b in range(nv):
                row = Ln[max(a, b) * (max(a, b) + 1) // 2 + min(a, b)]
                for i in range(no):
                    for j in range(no):
                        ih, il = max(i, j), min(i, j)
                        v = row[ih * (ih + 1) // 2 + il]
                        if a != b and i != j:
                            x = row[opp + ih * (ih - 1) // 2 + il]
                            v += x if (a > b) == (i > j) else -x
                        rec[a, b, i, j] = v
        out[rank] = max(err1, float(np.abs(rec - ref).max()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nv", [6, 7])
def test_exchange_two_ranks(nv):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, nv, 3, out), nprocs=2, join=True)
    assert len(out) == 2 and max(out.values()) < 1e-12


def _solver_worker(rank, world, port, libpath, out):
    """The distributed CCSD/DCSD loop end to end on the host simulator (CPU stand-in for the kernels): slab residual,
    overlapped all-gathers, pair-sharded tail (compact R2 / update / DIIS, all-reduced overlaps, T2 all-gather)."""
    import contextlib
    import io
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.cases import synthetic_case
        from pymes_amd import _lib
        from pymes_amd.solver.ccsd import CCSD
        from tests.conftest import hostsim_library
        _lib._default = hostsim_library(libpath)
        res = {}
        for no, nv, dcsd, diis in ((3, 7, False, True), (2, 5, True, True), (3, 6, False, False)):
            f, V, B, eps = synthetic_case(no, nv, seed=3, scale=0.3)
            s = CCSD(no, delta_e=1e-10, is_dcsd=dcsd, is_diis=diis)
            with contextlib.redirect_stdout(io.StringIO()):
                r = s.solve(f, V)
            assert s.pair_sharded and s.hooked and s.collective_calls >= 10 * s.iterations
            res[(no, nv, dcsd, diis)] = (float(r["ccsd e"]), int(s.iterations), float(np.abs(r["t2"]).sum()),
                                         float(np.abs(r["t2"] - r["t2"].transpose(1, 0, 3, 2)).max()))
        # symmetric user amplitudes: sharded residual with the replicated tail (the caller's arrays are updated in place)
        no, nv = 3, 6
        f, V, B, eps = synthetic_case(no, nv, seed=5, scale=0.3)
        rng = np.random.default_rng(6)
        x = rng.standard_normal((nv, nv, no, no)) * 1e-2
        amps = [rng.standard_normal((nv, no)) * 1e-2, x + x.transpose(1, 0, 3, 2)]
        s = CCSD(no, delta_e=1e-10)
        with contextlib.redirect_stdout(io.StringIO()):
            r = s.solve(f, V, amps=[a.copy() for a in amps])
        assert not s.pair_sharded
        res[("amps", no, nv)] = (float(r["ccsd e"]), int(s.iterations), float(np.abs(r["t2"]).sum()), 0.0)
        from pymes_amd.solver.ccd import CCD
        for no, nv, dcd, diis in ((3, 7, False, True), (2, 6, True, True), (3, 5, True, False)):
            f, V, B, eps = synthetic_case(no, nv, seed=4, scale=0.3)
            s = CCD(no, delta_e=1e-10, is_dcd=dcd, is_diis=diis)
            with contextlib.redirect_stdout(io.StringIO()):
                r = s.solve(f, V)
            assert s.pair_sharded and s.hooked          # (pymes_ccd_sharded_residuals + the finish / energy / await steps)
            res[("ccd", no, nv, dcd, diis)] = (float(r["ccd e"]), int(s.iterations), float(np.abs(r["t2 amp"]).sum()), 0.0)
        # owner-tile exchange (PYMES_OWNER_TILES=1): an all-to-all of the tiles each pair owner reads instead of the two
        # all-gathers of the ring-product rows — must give the very same numbers, bit for bit
        os.environ["PYMES_OWNER_TILES"] = "1"
        try:
            tiles = {}
            for no, nv, dcsd, diis in ((3, 7, False, True), (2, 5, True, True)):
                f, V, B, eps = synthetic_case(no, nv, seed=3, scale=0.3)
                s = CCSD(no, delta_e=1e-10, is_dcsd=dcsd, is_diis=diis)
                with contextlib.redirect_stdout(io.StringIO()):
                    r = s.solve(f, V)
                assert s.pair_sharded and s.hooked      # (the all-to-all is a callback of the table too: pymes_set_alltoallv)
                tiles[(no, nv, dcsd, diis)] = (float(r["ccsd e"]), int(s.iterations), float(np.abs(r["t2"]).sum()),
                                               float(np.abs(r["t2"] - r["t2"].transpose(1, 0, 3, 2)).max()))
            f, V, B, eps = synthetic_case(3, 7, seed=4, scale=0.3)
            s = CCD(3, delta_e=1e-10)
            with contextlib.redirect_stdout(io.StringIO()):
                r = s.solve(f, V)
            tiles[("ccd", 3, 7, False, True)] = (float(r["ccd e"]), int(s.iterations), float(np.abs(r["t2 amp"]).sum()), 0.0)
            # ... and the Python-sequenced forms of both (PYMES_PY_SEQUENCED=1: one library call per step, torch.distributed in
            # between — what the whole steps with the collective table replaced): the same numbers again
            os.environ["PYMES_PY_SEQUENCED"] = "1"
            try:
                f, V, B, eps = synthetic_case(3, 7, seed=3, scale=0.3)
                s = CCSD(3, delta_e=1e-10)
                with contextlib.redirect_stdout(io.StringIO()):
                    r = s.solve(f, V)
                assert s.pair_sharded and not s.hooked
                tiles[("py", 3, 7, False, True)] = (float(r["ccsd e"]), int(s.iterations), float(np.abs(r["t2"]).sum()),
                                                    float(np.abs(r["t2"] - r["t2"].transpose(1, 0, 3, 2)).max()))
                f, V, B, eps = synthetic_case(3, 7, seed=4, scale=0.3)
                s = CCD(3, delta_e=1e-10)
                with contextlib.redirect_stdout(io.StringIO()):
                    r = s.solve(f, V)
                assert s.pair_sharded and not s.hooked
                tiles[("pyccd", 3, 7, False, True)] = (float(r["ccd e"]), int(s.iterations), float(np.abs(r["t2 amp"]).sum()), 0.0)
            finally:
                del os.environ["PYMES_PY_SEQUENCED"]
        finally:
            del os.environ["PYMES_OWNER_TILES"]
        for key, val in tiles.items():
            ref_key = (3, 7, False, True) if key[0] == "py" else (("ccd", 3, 7, False, True) if key[0] == "pyccd" else key)
            assert val == res[ref_key], (key, val, res[ref_key])
        out[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("no,nv,world", [(3, 7, 2), (2, 5, 3), (4, 9, 8), (5, 20, 8), (50, 200, 8), (1, 3, 4)])
def test_owner_tile_plan_covers_what_the_assembly_reads(no, nv, world):
    """Every tile [(a,.),(b,.)] / [(b,.),(a,.)] of a pair matrix that residual_assemble_pairs reads for the pairs a rank
    owns is either in the rank's own row slab or inside a rectangle some owner sends it; no rectangle leaves the rows
    of its sender."""
    ov, npp = no * nv, nv * (nv + 1) // 2
    plan = pdist.owner_tile_plan(no, nv, world)
    for q in range(world):
        have = np.zeros((nv, nv), dtype=bool)             # tile (x, y): rows (x,.) columns (y,.) present on rank q
        own = pdist.slab_rows(ov, q, world)
        rows_present = np.zeros((ov, nv), dtype=bool)     # (row, column tile)
        rows_present[own[0]:own[1], :] = True
        for p in range(world):
            assert not plan[q][q]
            r0p, r1p = pdist.slab_rows(ov, p, world)
            for (r0, r1, c0, c1) in plan[p][q]:
                assert r0p <= r0 < r1 <= r1p and 0 <= c0 < c1 <= ov and c0 % no == 0 and c1 % no == 0
                rows_present[r0:r1, c0 // no:c1 // no] = True
        have = rows_present.reshape(nv, no, nv).all(axis=1)
        lo, hi = pdist.slab_rows(npp, q, world)
        for r in range(lo, hi):
            a = int((np.sqrt(8.0 * r + 1.0) - 1.0) / 2.0)
            while a * (a + 1) // 2 > r:
                a -= 1
            while (a + 1) * (a + 2) // 2 <= r:
                a += 1
            b = r - a * (a + 1) // 2
            assert have[a, b] and have[b, a], (q, a, b)
    if world == 8 and (no, nv) == (50, 200):      # the point of it: ~0.2 GB per rank instead of 1.6 GB of all-gathers
        recv = [2 * 8 * sum((r1 - r0) * (c1 - c0) for p in range(world) for r0, r1, c0, c1 in plan[p][q]) for q in range(world)]
        assert max(recv) < 0.21e9 and 2 * 8 * ov * ov * 7 // 8 > 1.3e9


@pytest.mark.parametrize("world,dress", [(2, "0"), (3, "0"), (8, "0"), (2, "1"), (8, "1")])
def test_distributed_solver_host_logic(hostsim_lib, world, dress, monkeypatch):
    """8: more ranks than some index ranges have chunks (empty shares).  dress = "1": every rank dresses the bra of its
    rows of the packed V_abcd (PYMES_LADDER_DRESS, inherited by the spawned ranks) instead of forming its Q_kb share."""
    from oracle import cc_oracle as oc
    from oracle.cases import synthetic_case
    monkeypatch.setenv("PYMES_LADDER_DRESS", dress)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_solver_worker, args=(world, _free_port(), hostsim_lib.path, out), nprocs=world, join=True)
    assert len(out) == world and all(out[r] == out[0] for r in range(world))          # ranks agree bit for bit
    for key, (e, it, t2sum, asym) in out[0].items():
        if key[0] == "amps":
            _, no, nv = key
            f, V, B, eps = synthetic_case(no, nv, seed=5, scale=0.3)
            rng = np.random.default_rng(6)
            x = rng.standard_normal((nv, nv, no, no)) * 1e-2
            amps = [rng.standard_normal((nv, no)) * 1e-2, x + x.transpose(1, 0, 3, 2)]
            ref = oc.ccsd_solve(no, f, V, delta_e=1e-10, amps=amps)
            assert abs(e - ref["e"]) < 1e-10 and it == ref["iterations"], (key, e, ref["e"], it, ref["iterations"])
            continue
        if key[0] == "ccd":
            _, no, nv, dcd, diis = key
            f, V, B, eps = synthetic_case(no, nv, seed=4, scale=0.3)
            ref = oc.ccd_solve(no, f, V, is_dcd=dcd, is_diis=diis, delta_e=1e-10)
            assert abs(e - ref["e"]) < 1e-10 and it == ref["iterations"], (key, e, ref["e"], it, ref["iterations"])
            assert abs(t2sum - np.abs(ref["t2"]).sum()) < 1e-8
            continue
        no, nv, dcsd, diis = key
        f, V, B, eps = synthetic_case(no, nv, seed=3, scale=0.3)
        ref = oc.ccsd_solve(no, f, V, is_dcsd=dcsd, is_diis=diis, delta_e=1e-10)
        assert abs(e - ref["e"]) < 1e-10 and it == ref["iterations"], (no, nv, dcsd, diis, e, ref["e"], it)
        assert abs(t2sum - np.abs(ref["t2"]).sum()) < 1e-8 and asym < 1e-12
