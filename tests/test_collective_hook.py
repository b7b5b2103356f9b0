"""The one-process-per-GPU loop body as whole library steps with a collective table (include/pymes_amd.h:
pymes_set_collectives, pymes_ccsd_sharded_residuals / _finish / _energy / _await) — driven here WITHOUT torch: the table is
filled with plain ctypes callbacks.  World of one against the single-rank entry points; a true two-rank run on two host
threads (one context each, the callbacks meet at a barrier and exchange through host memory) against the same; a failing
callback; the order in which the collectives are issued."""
import ctypes as C
import threading

import numpy as np
import pytest

from pymes_amd import _lib
from pymes_amd.device import Context, PymesError

from oracle.cases import synthetic_case

START_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64))
WAIT_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_void_p)
MARK_T = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p)
A2A_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64), C.c_void_p,
                    C.POINTER(C.c_int64))


class Table(C.Structure):
    _fields_ = [("user", C.c_void_p), ("rank", C.c_int), ("world", C.c_int), ("allreduce_start", START_T),
                ("allgather_start", START_T), ("wait", WAIT_T), ("mark", MARK_T)]


NAMES = ("ETd", "ETx", "L", "QK", "Tall", "W", "Xvv", "P", "R1", "S")


class Rank:
    """One rank's context, exchange buffers and table.  ``peers``: the Rank objects of the whole world (shared list) — the
    callbacks exchange through the host copies the peers publish at a barrier; a world of one needs neither."""

    def __init__(self, lib, no, nv, f, V, rank, world, peers=None, barrier=None, fail=None):
        self.rank, self.world, self.peers, self.barrier, self.fail = rank, world, peers, barrier, fail
        self.log, self.published = [], None
        self.ctx = ctx = Context(no, nv, lib=lib)
        ctx.set_V_pqrs(V)
        ctx.set_orbital_energies(f.diagonal()[:no].copy(), f.diagonal()[no:].copy())
        sizes = (C.c_int64 * 10)()
        ctx.lib.call("pymes_shard_buffer_sizes", ctx.handle, world, sizes)
        self.sizes = dict(zip(NAMES, sizes))
        self.arr = {k: ctx.zeros((int(self.sizes[k]),)) for k in NAMES}
        self.bufs = _lib.ShardBuffers(*[self.arr[k].ptr for k in NAMES])
        self.cbs = (START_T(self._allreduce), START_T(self._allgather), WAIT_T(self._wait), MARK_T(self._mark))
        self.table = Table(None, rank, world, *self.cbs)
        ctx.lib.call("pymes_set_collectives", ctx.handle, C.byref(self.table))
        # the optional all-to-all of the table and the staging buffers of the owner-tile exchange (include/pymes_amd.h)
        ns, nr = C.c_int64(), C.c_int64()
        ctx.lib.call("pymes_owner_tile_sizes", ctx.handle, rank, world, C.byref(ns), C.byref(nr))
        self.xs, self.xr = ctx.zeros((int(ns.value),)), ctx.zeros((int(nr.value),))
        self.a2a = A2A_T(self._alltoallv)
        ctx.lib.call("pymes_set_alltoallv", ctx.handle, C.cast(self.a2a, C.c_void_p))
        ctx.lib.call("pymes_set_owner_tile_buffers", ctx.handle, C.c_void_p(self.xs.ptr), C.c_void_p(self.xr.ptr))

    # device memory by pointer: which buffer, which offset (doubles)
    def _locate(self, ptr):
        for k in NAMES:
            a = self.arr[k]
            if a.ptr <= ptr < a.ptr + 8 * a.size:
                return k, (ptr - a.ptr) // 8
        raise AssertionError("pointer outside the exchange buffers")

    def _exchange(self, name, off, n, combine):
        """Blocking collective through the host: publish this rank's n doubles, meet, combine all ranks', write back."""
        self.ctx.sync()
        mine = self.arr[name].get().ravel()[off:off + n].copy()
        if self.world > 1:
            self.published = mine
            self.barrier.wait()
            parts = [p.published for p in self.peers]
            self.barrier.wait()             # everybody has read before anybody publishes again
        else:
            parts = [mine]
        full = self.arr[name].get().ravel()
        full[off:off + n] = combine(parts)
        self.arr[name].set(full.reshape(self.arr[name].shape))

    def _allreduce(self, user, buf, n, stream, ticket):
        name, off = self._locate(int(buf))
        self.log.append(("allreduce", name, off, int(n)))
        if self.fail == "allreduce":
            return 7
        self._exchange(name, off, int(n), lambda parts: np.sum(parts, axis=0))
        ticket[0] = len(self.log)
        return 0

    def _allgather(self, user, buf, chunk, stream, ticket):
        name, off = self._locate(int(buf))
        chunk = int(chunk)
        self.log.append(("allgather", name, off, chunk))
        assert off == 0 and chunk * self.world <= self.sizes[name]

        def combine(parts):      # rank r contributes its own chunk r of the buffer
            return np.concatenate([p[r * chunk:(r + 1) * chunk] for r, p in enumerate(parts)])
        self._exchange(name, 0, chunk * self.world, combine)
        ticket[0] = len(self.log)
        return 0

    def _alltoallv(self, user, send, send_counts, recv, recv_counts, stream, ticket):
        """Blocking all-to-all through the host: every rank publishes its send buffer and counts, meets the others, and picks
        the piece addressed to it out of each (pieces contiguous in rank order on both sides)."""
        assert int(send) == self.xs.ptr and int(recv) == self.xr.ptr
        ns = [int(send_counts[q]) for q in range(self.world)]
        nr = [int(recv_counts[p]) for p in range(self.world)]
        self.log.append(("alltoallv", "X", sum(ns), sum(nr)))
        assert ns[self.rank] == 0 and nr[self.rank] == 0 and sum(ns) <= self.xs.size and sum(nr) <= self.xr.size
        self.ctx.sync()
        if self.world > 1:
            self.published = (self.xs.get().ravel().copy(), ns)
            self.barrier.wait()
            got = []
            for p, peer in enumerate(self.peers):
                buf, counts = peer.published
                off = sum(counts[:self.rank])
                assert counts[self.rank] == nr[p]
                got.append(buf[off:off + counts[self.rank]])
            self.barrier.wait()
            full = self.xr.get().ravel()
            cat = np.concatenate(got) if got else np.zeros(0)
            full[:cat.size] = cat
            self.xr.set(full)
        ticket[0] = len(self.log)
        return 0

    def _wait(self, user, ticket, stream):
        self.log.append(("wait", int(ticket)))
        return 0

    def _mark(self, user, phase):
        self.log.append(("mark", phase.decode()))

    def call(self, name, *args):
        self.ctx.lib.call(name, self.ctx.handle, *args)

    def close(self):
        self.ctx.close()


def amplitudes(no, nv, seed):
    rng = np.random.default_rng(seed)
    t1 = 0.1 * rng.standard_normal((nv, no))
    t2 = 0.1 * rng.standard_normal((nv, nv, no, no))
    t2 = t2 + t2.transpose(1, 0, 3, 2)
    dt2 = 0.1 * rng.standard_normal((nv, nv, no, no))
    dt2 = dt2 + dt2.transpose(1, 0, 3, 2)
    return t1, t2, dt2


def single_rank(lib, no, nv, f, V, t1, t2, dt2, dcsd):
    ctx = Context(no, nv, lib=lib)
    try:
        ctx.set_V_pqrs(V)
        ctx.set_orbital_energies(f.diagonal()[:no].copy(), f.diagonal()[no:].copy())
        fdev, a1, a2 = ctx.array(f), ctx.array(t1), ctx.array(t2)
        r1, r2 = ctx.empty(t1.shape), ctx.empty(t2.shape)
        ctx.ccsd_residuals(fdev, a1, a2, r1, r2, is_dcd=dcsd)
        en = ctx.energy_norms(fdev, a1, a2, ctx.array(dt2))
        ctx.ccsd_release()
        return r1.get(), r2.get(), np.array(en)
    finally:
        ctx.close()


def run_rank(rk, no, nv, f, t1, t2, dt2, dcsd, out, owner=False):
    """The sequence of include/pymes_amd.h on one rank: residuals; finish (energies of the amplitudes as they stand + the
    exchange of the compact tiles); the energy read-back; await (the replicated array rebuilt from every rank's tiles)."""
    ctx, world, rank = rk.ctx, rk.world, rk.rank
    fdev, fd, a1, a2 = ctx.array(f), ctx.empty(f.shape), ctx.array(t1), ctx.array(t2)
    npp = nv * (nv + 1) // 2
    c = -(-npp // world)
    lo, hi = min(rank * c, npp), min(rank * c + c, npp)
    shape = (max(hi - lo, 1), 2, no * no)
    rc, tc, dtc = ctx.zeros(shape), ctx.zeros(shape), ctx.zeros(shape)
    ctx.pairs_pack(a2, tc, rank, world)
    ctx.pairs_pack(ctx.array(dt2), dtc, rank, world)
    flags = (_lib.PYMES_DCD if dcsd else 0) | (_lib.PYMES_OWNER_TILES if owner else 0)
    rk.call("pymes_ccsd_sharded_residuals", C.c_void_p(fdev.ptr), C.c_void_p(fd.ptr), C.c_void_p(a1.ptr), C.c_void_p(a2.ptr),
            C.byref(rk.bufs), flags, C.c_void_p(rc.ptr))
    r1 = rk.arr["R1"].get()[:nv * no].reshape(nv, no).copy()
    slot = C.c_int()
    rk.call("pymes_ccsd_sharded_finish", C.c_void_p(fdev.ptr), C.c_void_p(a1.ptr), C.c_void_p(tc.ptr), C.c_void_p(dtc.ptr),
            C.byref(rk.bufs), C.byref(slot))
    en = (C.c_double * 6)()
    rk.call("pymes_ccsd_sharded_energy", slot.value, en)
    back = ctx.zeros(t2.shape)
    rk.call("pymes_ccsd_sharded_await", C.c_void_p(back.ptr), C.byref(rk.bufs))
    rk.call("pymes_ccsd_sharded_await", C.c_void_p(back.ptr), C.byref(rk.bufs))        # nothing in flight any more: a no-op
    out[rank] = dict(r1=r1, rc=rc.get()[:max(hi - lo, 0)], lo=lo, hi=hi, en=np.array(en[:]), t2_back=back.get())


def run_rank_ccd(rk, no, nv, f, t2, dt2, dcd, owner, out):
    """CCD / DCD on one rank (pymes/solver/ccd.py:100-132): pymes_ccd_sharded_residuals; the finish, energy read-back and await
    of the CCSD steps with f = t1 = NULL."""
    ctx, world, rank = rk.ctx, rk.world, rk.rank
    fdev, a2 = ctx.array(f), ctx.array(t2)
    npp = nv * (nv + 1) // 2
    c = -(-npp // world)
    lo, hi = min(rank * c, npp), min(rank * c + c, npp)
    shape = (max(hi - lo, 1), 2, no * no)
    rc, tc, dtc = ctx.zeros(shape), ctx.zeros(shape), ctx.zeros(shape)
    ctx.pairs_pack(a2, tc, rank, world)
    ctx.pairs_pack(ctx.array(dt2), dtc, rank, world)
    flags = (_lib.PYMES_DCD if dcd else 0) | (_lib.PYMES_OWNER_TILES if owner else 0)
    rk.call("pymes_ccd_sharded_residuals", C.c_void_p(fdev.ptr), C.c_void_p(a2.ptr), C.byref(rk.bufs), flags, C.c_void_p(rc.ptr))
    slot = C.c_int()
    rk.call("pymes_ccsd_sharded_finish", None, None, C.c_void_p(tc.ptr), C.c_void_p(dtc.ptr), C.byref(rk.bufs), C.byref(slot))
    en = (C.c_double * 6)()
    rk.call("pymes_ccsd_sharded_energy", slot.value, en)
    back = ctx.zeros(t2.shape)
    rk.call("pymes_ccsd_sharded_await", C.c_void_p(back.ptr), C.byref(rk.bufs))
    out[rank] = dict(rc=rc.get()[:max(hi - lo, 0)], lo=lo, hi=hi, en=np.array(en[:]), t2_back=back.get())


def check_world_ccd(lib, world, dcd, owner, no=3, nv=7, tol=1e-12):
    """CCD's N > 1 loop body through the collective table, torch-free: against the single-rank residual
    (ccd.py:164-254) and energies; with the two all-gathers of the ring rows and with the owner-tile all-to-all."""
    f, V, _, _ = synthetic_case(no, nv, seed=6, scale=0.3 if nv < 20 else 0.15)
    _, t2, dt2 = amplitudes(no, nv, 11)
    ctx = Context(no, nv, lib=lib)
    try:
        ctx.set_V_pqrs(V)
        ctx.set_orbital_energies(f.diagonal()[:no].copy(), f.diagonal()[no:].copy())
        fdev, a2 = ctx.array(f), ctx.array(t2)
        r2 = ctx.empty(t2.shape)
        ctx.doubles_residual(fdev, a2, r2, is_dcd=dcd, sym_ladder=True)
        want_r2 = r2.get()
        want_en = np.array(ctx.energy_norms(None, None, a2, ctx.array(dt2)))
    finally:
        ctx.close()
    peers, barrier, out = [], threading.Barrier(world), {}
    for r in range(world):
        peers.append(Rank(lib, no, nv, f, V, r, world, peers, barrier))
    try:
        errors = []

        def body(rk):
            try:
                run_rank_ccd(rk, no, nv, f, t2, dt2, dcd, owner, out)
            except BaseException as exc:
                errors.append(exc)
                barrier.abort()
        threads = [threading.Thread(target=body, args=(rk,)) for rk in peers]
        for th in threads:
            th.start()
        for th in threads:
            th.join(timeout=300)
        assert not errors, errors
        got_r2 = np.zeros_like(want_r2)
        for r in range(world):
            o = out[r]
            assert np.abs(o["en"] - want_en).max() < tol * max(1.0, np.abs(want_en).max()), (o["en"], want_en)
            assert np.abs(o["t2_back"] - t2).max() == 0.0
            unpack(no, nv, o["lo"], o["rc"], got_r2)
        assert np.abs(got_r2 - want_r2).max() < tol
        seqs = [[e[:2] for e in rk.log if e[0] in ("allreduce", "allgather", "alltoallv")] for rk in peers]
        assert all(s == seqs[0] for s in seqs)
        ring = [("alltoallv", "X")] if owner else [("allgather", "ETd"), ("allgather", "ETx")]
        assert seqs[0] == ring + [("allreduce", "S"), ("allgather", "Tall")]
    finally:
        for rk in peers:
            rk.close()


def unpack(no, nv, lo, tiles, full):
    """Compact tiles [pair][2][o*o] of the pairs P(a,b) = a(a+1)/2 + b, lo <= P: tile 0 = X[a,b], tile 1 = X[b,a]."""
    for k, tile in enumerate(tiles):
        p = lo + k
        a = int((np.sqrt(8.0 * p + 1.0) - 1.0) / 2.0)
        while a * (a + 1) // 2 > p:
            a -= 1
        while (a + 1) * (a + 2) // 2 <= p:
            a += 1
        b = p - a * (a + 1) // 2
        full[a, b] = tile[0].reshape(no, no)
        if a != b:
            full[b, a] = tile[1].reshape(no, no)


def check_world(lib, world, dcsd, no=3, nv=7, tol=1e-12, owner=False):
    f, V, _, _ = synthetic_case(no, nv, seed=6, scale=0.3 if nv < 20 else 0.15)
    t1, t2, dt2 = amplitudes(no, nv, 11)
    want_r1, want_r2, want_en = single_rank(lib, no, nv, f, V, t1, t2, dt2, dcsd)
    peers, barrier, out = [], threading.Barrier(world), {}
    for r in range(world):
        peers.append(Rank(lib, no, nv, f, V, r, world, peers, barrier))
    try:
        errors = []

        def body(rk):
            try:
                run_rank(rk, no, nv, f, t1, t2, dt2, dcsd, out, owner)
            except BaseException as exc:          # a rank that dies must not leave the others at the barrier
                errors.append(exc)
                barrier.abort()
        threads = [threading.Thread(target=body, args=(rk,)) for rk in peers]
        for th in threads:
            th.start()
        for th in threads:
            th.join(timeout=300)
        assert not errors, errors
        got_r2 = np.zeros_like(want_r2)
        for r in range(world):
            o = out[r]
            assert np.abs(o["r1"] - want_r1).max() < tol                    # all-reduced: complete on every rank
            assert np.abs(o["en"] - want_en).max() < tol * max(1.0, np.abs(want_en).max())
            assert np.abs(o["t2_back"] - t2).max() == 0.0                   # the exchange of the compact tiles, unpacked
            unpack(no, nv, o["lo"], o["rc"], got_r2)
        assert np.abs(got_r2 - want_r2).max() < tol
        # the order of the collectives is the same on every rank (a communicator runs them in order), the big all-reduce of
        # the hole-ladder intermediate is waited for after the ring rows have been handed over, the new T2 goes last
        seqs = [[e[:2] for e in rk.log if e[0] in ("allreduce", "allgather", "alltoallv")] for rk in peers]
        assert all(s == seqs[0] for s in seqs)
        ring = [("alltoallv", "X")] if owner else [("allgather", "ETd"), ("allgather", "ETx")]
        assert seqs[0] == [("allreduce", "W"), ("allreduce", "P"), ("allreduce", "P")] + ring + [
            ("allgather", "QK"), ("allreduce", "Xvv"), ("allreduce", "R1"), ("allreduce", "S"), ("allgather", "Tall")]
        marks = [e[1] for e in peers[0].log if e[0] == "mark"]
        assert marks[0] == "begin" and "ring products" in marks and marks[-1] == "energy + norms (pairs)"
    finally:
        for rk in peers:
            rk.close()


@pytest.mark.parametrize("world,dcsd", [(1, False), (1, True), (2, False), (2, True), (3, False)])
def test_sharded_steps_with_a_plain_table_host_logic(hostsim_lib, world, dcsd):
    check_world(hostsim_lib, world, dcsd)


@pytest.mark.parametrize("world,dcd,owner", [(1, False, False), (1, True, True), (2, False, False), (2, True, True), (3, False, True),
                                             (3, True, False)])
def test_ccd_sharded_steps_with_a_plain_table_host_logic(hostsim_lib, world, dcd, owner):
    check_world_ccd(hostsim_lib, world, dcd, owner)


@pytest.mark.parametrize("world", [2, 3])
def test_ccsd_owner_tiles_with_a_plain_table_host_logic(hostsim_lib, world):
    check_world(hostsim_lib, world, False, owner=True)


@pytest.mark.gpu
def test_ccd_sharded_steps_with_a_plain_table_gpu(gpu_lib):
    check_world_ccd(gpu_lib, 1, False, False)
    check_world_ccd(gpu_lib, 1, True, True, no=20, nv=80, tol=1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("dcsd", [False, True])
def test_sharded_steps_with_a_plain_table_gpu(gpu_lib, dcsd):
    # (a world of one: the library is one context per GPU and process — the reduction workspace is per device — so several
    # ranks on the one card of the test box would have to take turns; the multi-rank sequence is the host-logic test above)
    check_world(gpu_lib, 1, dcsd)
    # ... and at config 2's size, where the LDS-DMA launches, their cut tails and the bra dressing of the packed V_abcd are in play
    check_world(gpu_lib, 1, dcsd, no=20, nv=80, tol=1e-11)


def check_failures(lib):
    no, nv = 2, 4
    f, V, _, _ = synthetic_case(no, nv, seed=2, scale=0.3)
    t1, t2, dt2 = amplitudes(no, nv, 3)
    rk = Rank(lib, no, nv, f, V, 0, 1, fail="allreduce")
    try:
        ctx = rk.ctx
        fdev, fd, a1, a2 = ctx.array(f), ctx.empty(f.shape), ctx.array(t1), ctx.array(t2)
        rc = ctx.zeros((nv * (nv + 1) // 2, 2, no * no))
        args = (C.c_void_p(fdev.ptr), C.c_void_p(fd.ptr), C.c_void_p(a1.ptr), C.c_void_p(a2.ptr), C.byref(rk.bufs), 0,
                C.c_void_p(rc.ptr))
        with pytest.raises(PymesError, match="collective hook"):
            rk.call("pymes_ccsd_sharded_residuals", *args)
        with pytest.raises(PymesError, match="PYMES_DCD"):
            rk.call("pymes_ccsd_sharded_residuals", *args[:5], 64, args[6])
        ctx.lib.call("pymes_set_alltoallv", ctx.handle, None)                   # no all-to-all in the table: owner tiles refuse
        rk.fail = None
        with pytest.raises(PymesError, match="owner tiles"):
            rk.call("pymes_ccsd_sharded_residuals", *args[:5], _lib.PYMES_OWNER_TILES, args[6])
        rk.fail = "allreduce"
        ctx.lib.call("pymes_set_collectives", ctx.handle, None)                  # table removed: the steps refuse
        with pytest.raises(PymesError, match="no collectives"):
            rk.call("pymes_ccsd_sharded_residuals", *args)
        bad = Table(None, 2, 2, *rk.cbs)
        with pytest.raises(PymesError, match="rank"):
            ctx.lib.call("pymes_set_collectives", ctx.handle, C.byref(bad))
    finally:
        rk.close()


def check_abandoned_loop(lib):
    """A loop left between finish and the next residuals: the exchange of the amplitudes it started is completed through the
    table that started it when that table is replaced or removed (nobody reads the result)."""
    no, nv = 2, 4
    f, V, _, _ = synthetic_case(no, nv, seed=2, scale=0.3)
    t1, t2, dt2 = amplitudes(no, nv, 3)
    rk = Rank(lib, no, nv, f, V, 0, 1)
    try:
        ctx = rk.ctx
        fdev, a1, a2 = ctx.array(f), ctx.array(t1), ctx.array(t2)
        shape = (nv * (nv + 1) // 2, 2, no * no)
        tc, dtc = ctx.zeros(shape), ctx.zeros(shape)
        ctx.pairs_pack(a2, tc, 0, 1)
        slot = C.c_int()
        rk.call("pymes_ccsd_sharded_finish", C.c_void_p(fdev.ptr), C.c_void_p(a1.ptr), C.c_void_p(tc.ptr), C.c_void_p(dtc.ptr),
                C.byref(rk.bufs), C.byref(slot))
        with pytest.raises(PymesError, match="never awaited"):                   # a second finish without residuals / await
            rk.call("pymes_ccsd_sharded_finish", C.c_void_p(fdev.ptr), C.c_void_p(a1.ptr), C.c_void_p(tc.ptr),
                    C.c_void_p(dtc.ptr), C.byref(rk.bufs), C.byref(slot))
        gather = [i + 1 for i, e in enumerate(rk.log) if e[:2] == ("allgather", "Tall")][-1]      # its ticket (Rank._allgather)
        waits = len([e for e in rk.log if e[0] == "wait"])
        ctx.lib.call("pymes_set_collectives", ctx.handle, C.byref(rk.table))     # the same table again: old exchange completed
        assert [e for e in rk.log if e[0] == "wait"][waits:] == [("wait", gather)]
        ctx.lib.call("pymes_set_collectives", ctx.handle, None)                  # nothing in flight any more
        assert len([e for e in rk.log if e[0] == "wait"]) == waits + 1
    finally:
        rk.close()


def test_abandoned_loop_host_logic(hostsim_lib):
    check_abandoned_loop(hostsim_lib)


def test_failures_host_logic(hostsim_lib):
    check_failures(hostsim_lib)


@pytest.mark.gpu
def test_failures_gpu(gpu_lib):
    check_failures(gpu_lib)
