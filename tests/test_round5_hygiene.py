"""Round-5 regression tests for the advisor's findings of round 4: read-back tickets, nested product groups, a DIIS
step without a finite solution on the device-resident path, stale / partial dressed-integral hand-overs, and the
pipelined residual build that a solve leaves in flight when it ends."""
import contextlib
import io
import os

import numpy as np
import pytest

from pymes_amd import _lib
from pymes_amd.device import Context, PymesError

from oracle.cases import eom_davidson_case, synthetic_case


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


# ---- read-back tickets (kernels.hip: readback_start_impl / readback_wait_impl) ------------------------------------------
def check_readback_tickets(lib):
    ctx = Context(2, 3, lib=lib)
    try:
        a = ctx.array(np.arange(8.0))
        first = ctx.readback_start(a, 8)
        assert np.array_equal(ctx.readback_wait(first, 8), np.arange(8.0))
        assert np.array_equal(ctx.readback_wait(first, 8), np.arange(8.0))        # waiting twice is fine
        later = [ctx.readback_start(a, 4) for _ in range(16)]                     # the ring has 16 slots: `first` is gone
        with pytest.raises(PymesError, match="reused"):
            ctx.readback_wait(first, 8)
        assert np.array_equal(ctx.readback_wait(later[-1], 4), np.arange(4.0))
        with pytest.raises(PymesError):
            ctx.readback_wait(5, 4)                                               # generation 0: never started
        with pytest.raises(PymesError):
            ctx.readback_wait(-1, 4)
    finally:
        ctx.close()


def test_readback_tickets_host_logic(hostsim_lib):
    check_readback_tickets(hostsim_lib)


@pytest.mark.gpu
def test_readback_tickets_gpu(gpu_lib):
    check_readback_tickets(gpu_lib)


# ---- nested product groups (dev::gemm_group_begin / _end) ---------------------------------------------------------------
def check_nested_groups(lib):
    """An engine-internal group scope inside a caller's (ladder_t1 inside ``ctx.gemm_group()``) must not close the outer
    one, and a failing flush must not leave the group open."""
    rng = np.random.default_rng(3)
    ctx = Context(6, 20, lib=lib)
    try:
        A, B = rng.standard_normal((70, 90)), rng.standard_normal((90, 80))
        a, b = ctx.array(A), ctx.array(B)
        with ctx.gemm_group():
            c1 = ctx.contract("mk,kn->mn", a, b)
            with ctx.gemm_group():                                 # nested: what an engine-internal scope does
                c2 = ctx.contract("mk,kn->mn", a, b, alpha=2.0)
            c3 = ctx.contract("mk,kn->mn", a, b, alpha=3.0)        # still inside the outer group
        for c, al in ((c1, 1.0), (c2, 2.0), (c3, 3.0)):
            assert np.abs(c.get() - al * (A @ B)).max() < 1e-11
        # outside every group a product runs at once again
        assert np.abs(ctx.contract("mk,kn->mn", a, b).get() - A @ B).max() < 1e-11
    finally:
        ctx.close()


def test_nested_groups_host_logic(hostsim_lib):
    check_nested_groups(hostsim_lib)


@pytest.mark.gpu
def test_nested_groups_gpu(gpu_lib):
    check_nested_groups(gpu_lib)


# ---- DIIS without a finite solution: the C algebra shared by host, simulator and kernel ---------------------------------
def check_diis_singular(lib):
    """A subspace matrix full of NaN: `pymes_diis_solve` reports status 2 and its coefficients select the newest amplitudes
    (what an extrapolation enqueued behind a device-resident step then computes), the mixer raises LinAlgError."""
    from pymes_amd.mixer.diis import DIIS
    buf = np.zeros(96)
    n = 3                              # two stored pairs + the Lagrange row; a third pair arrives with NaN overlaps
    buf[0] = n
    pad = np.zeros((9, 9))
    pad[:n, :n] = [[1.0, 0.5, -1.0], [0.5, 2.0, -1.0], [-1.0, -1.0, 0.0]]
    buf[1:82] = pad.ravel()
    overlaps = np.array([np.nan, np.nan])
    rc = lib.dll.pymes_diis_solve(_lib.host_ptr(buf), _lib.host_ptr(overlaps), 1, 2, 0)
    assert rc == 0 and buf[91] == 2.0, (rc, buf[91])
    m_new = int(buf[0]) - 1
    assert np.array_equal(buf[82:82 + m_new + 1], np.eye(m_new + 1)[m_new - 1])      # the newest amplitudes, unchanged
    # the mixer's read-back of a failed device step raises like numpy.linalg in the reference (diis.py:85-95)
    m = DIIS(dim_space=6)

    class FakeState:
        def __init__(self, ctx):
            self.ctx = ctx

        def get(self):
            b = np.zeros(96)
            b[0], b[91] = 2, 2.0
            return b
    ctx = Context(2, 3, lib=lib)
    try:
        m._state, m._stale, m._log_slot = FakeState(ctx), True, None
        with pytest.raises(np.linalg.LinAlgError):
            m._refresh_host()
    finally:
        ctx.close()


def test_diis_singular_host_logic(hostsim_lib):
    check_diis_singular(hostsim_lib)


# ---- dressed-integral hand-over: subset / stale ------------------------------------------------------------------------------
def check_dressed_handover(lib, monkeypatch):
    from pymes_amd.integral.device import DeviceIntegrals
    from pymes_amd.solver.ccsd import CCSD
    from pymes_amd.solver.eom_ccsd import EOM_CCSD
    monkeypatch.setattr(_lib, "_default", lib)
    no, nv = 3, 7
    f, V = eom_davidson_case(no, nv, seed=0, scale=0.3)
    ints = DeviceIntegrals.from_V_pqrs(no, V)
    try:
        cc = CCSD(no, delta_e=1e-9)
        res = quiet(cc.solve, f, ints, device_amplitudes=True)
        fd = cc.get_T1_dressed_fock(f, res["t1"], ints)
        part = cc.get_T1_dressed_V(res["t1"], ints, {"ijab": None, "klij": None})
        eom = EOM_CCSD(no, n_excit=2)
        eom.max_iter = 3
        with pytest.raises(KeyError):
            quiet(eom.solve, fd, part, res["t2"])                       # a subset: the sigma build would read stale blocks
        full = cc.get_T1_dressed_V(res["t1"], ints)
        quiet(eom.solve, fd, full, res["t2"])                           # fine
        cc.get_T1_dressed_V(res["t1"], ints, {"klij": None})            # a later dressing on the same context ...
        with pytest.raises(RuntimeError, match="stale"):
            quiet(eom.solve, fd, full, res["t2"])                       # ... invalidates the earlier hand-over
        # ... and so does a CCSD solve on the same DeviceIntegrals: its loop body dresses V_klij / V_iajb / V_iabj INSIDE the
        # library (pymes_ccsd_residuals, recorded and replayed as a launch graph) — the engine counts those too
        full = cc.get_T1_dressed_V(res["t1"], ints)
        quiet(eom.solve, fd, full, res["t2"])
        f2 = f + np.diag(np.linspace(0.0, 0.05, no + nv))
        quiet(CCSD(no, delta_e=1e-9).solve, f2, ints, device_amplitudes=True)
        with pytest.raises(RuntimeError, match="stale"):
            quiet(eom.solve, fd, full, res["t2"])
    finally:
        ints.ctx.close()


def test_dressed_handover_host_logic(hostsim_lib, monkeypatch):
    check_dressed_handover(hostsim_lib, monkeypatch)


@pytest.mark.gpu
def test_dressed_handover_gpu(gpu_lib, monkeypatch):
    check_dressed_handover(gpu_lib, monkeypatch)


# ---- handles that outlive their context; staging buffers released while a recorded graph still replays into them ----
def check_handle_and_release_order(lib):
    import ctypes as C
    no, nv = 3, 6
    f, V, _, _ = synthetic_case(no, nv, seed=1, scale=0.3)
    rng = np.random.default_rng(2)
    t1h = 0.05 * rng.standard_normal((nv, no))
    x = 0.05 * rng.standard_normal((nv, nv, no, no))
    t2h = x + x.transpose(1, 0, 3, 2)
    ctx = Context(no, nv, lib=lib)
    ctx.set_V_pqrs(V)
    ctx.set_orbital_energies(f.diagonal()[:no].copy(), f.diagonal()[no:].copy())
    fd, t1, t2 = ctx.array(f), ctx.array(t1h), ctx.array(t2h)
    r1, r2 = ctx.empty(t1.shape), ctx.empty(t2.shape)
    ctx.ccsd_residuals(fd, t1, t2, r1, r2)                       # eager: staging buffers allocated, statics built
    want = r2.get()
    if ctx.graphs_supported():
        ctx.graph_begin()
        ctx.ccsd_residuals(fd, t1, t2, r1, r2)
        g = ctx.graph_end()
        ctx.ccsd_release()                                       # another solver of the context finishing: must NOT free them yet
        ctx.dress_V(t1, ("klij", "ijka", "ijak", "iajb", "iabj", "iabc", "abic", "iajk", "abcd", "abij", "ijab"))
        h = C.c_void_p()
        ctx.lib.call("pymes_eom_sigma_prepare", ctx.handle, _lib.host_ptr(np.ascontiguousarray(f)), C.c_void_p(t2.ptr), 1, C.byref(h))
        r2.zero_()
        ctx.graph_launch(g)                                      # replays into the staging buffers: nobody else got them
        assert np.abs(r2.get() - want).max() < 1e-13
        ctx.graph_destroy(g)                                     # now they go back to the pool
    else:
        ctx.dress_V(t1, ("klij", "ijka", "ijak", "iajb", "iabj", "iabc", "abic", "iajk", "abcd", "abij", "ijab"))
        h = C.c_void_p()
        ctx.lib.call("pymes_eom_sigma_prepare", ctx.handle, _lib.host_ptr(np.ascontiguousarray(f)), C.c_void_p(t2.ptr), 1, C.byref(h))
    # the context goes first: the handle is invalidated, using it is an error, destroying it is fine
    ctx.close()
    flags = C.c_int()
    with pytest.raises(PymesError, match="null EOM handle"):
        lib.call("pymes_eom_sigma_flags", h, C.byref(flags))
    lib.call("pymes_eom_sigma_destroy", h)


def test_handle_and_release_order_host_logic(hostsim_lib):
    check_handle_and_release_order(hostsim_lib)


@pytest.mark.gpu
def test_handle_and_release_order_gpu(gpu_lib):
    check_handle_and_release_order(gpu_lib)


# ---- a solve that ends with a speculative residual build in flight ----------------------------------------------------------
@pytest.mark.gpu
def test_solve_ends_with_residuals_in_flight(gpu_lib, monkeypatch):
    """max_iter cuts the loop while |dE| is still large: the last pass has enqueued the residual graph of a pass that never
    comes.  With device_amplitudes=True nothing reads back, so destroy_graphs must drain the stream before it destroys the
    graph exec; the amplitudes handed over must be those of the unpipelined run."""
    from pymes_amd.integral.device import DeviceIntegrals
    from pymes_amd.solver.ccsd import CCSD
    monkeypatch.setattr(_lib, "_default", gpu_lib)
    no, nv = 8, 30
    f, V, _, _ = synthetic_case(no, nv, seed=1, scale=0.3)
    ints = DeviceIntegrals.from_V_pqrs(no, V)
    try:
        cc = CCSD(no, delta_e=1e-14)
        r1 = quiet(cc.solve, f, ints, device_amplitudes=True, max_iter=4)
        t2a = r1["t2"].get()
        assert ints.ctx.graph_count() == 0 if hasattr(ints.ctx, "graph_count") else True
        monkeypatch.setenv("PYMES_NO_PIPELINE", "1")
        r2 = quiet(CCSD(no, delta_e=1e-14).solve, f, ints, device_amplitudes=True, max_iter=4)
        assert r1["ccsd e"] == r2["ccsd e"]
        assert np.array_equal(t2a, r2["t2"].get())
    finally:
        ints.ctx.close()


# ---- whole-step entry points of the C-ABI (ccsd_residuals / ccsd_iterate) ---------------------------------------
def check_ccsd_iterate_entry(lib, dcsd):
    """pymes_ccsd_iterate — one fixed-point pass without a mixer (ccsd.py:159-197, is_diis = False) per call — three passes
    from the MP2 start against the oracle's loop (= the reference's algebra), and pymes_ccsd_residuals against the residuals
    the engine's separate entry points give."""
    from oracle import cc_oracle as oc
    no, nv = 3, 6
    f, V, _, _ = synthetic_case(no, nv, seed=4, scale=0.3)
    ref = oc.ccsd_solve(no, f, V, is_dcsd=dcsd, is_diis=False, delta_e=1e-30, max_iter=2)
    ctx = Context(no, nv, lib=lib)
    try:
        ctx.set_V_pqrs(V)
        assert ctx.V_exchange_symmetric()
        ctx.set_orbital_energies(f.diagonal()[:no].copy(), f.diagonal()[no:].copy())
        fdev = ctx.array(f)
        t1, t2 = ctx.zeros((nv, no)), ctx.empty((nv, nv, no, no))
        ctx.mp2(t2, 0.0)
        dt1, dt2 = ctx.empty(t1.shape), ctx.empty(t2.shape)
        e = None
        for it in range(3):
            out = ctx.ccsd_iterate(fdev, t1, t2, dt1, dt2, is_dcd=dcsd, t1_zero=(it == 0))
            e = out[0] + out[1] + out[2]
            assert abs(out[3] - np.vdot(t2.get(), t2.get())) < 1e-12
        assert abs(e - ref["e"]) < 1e-12
        assert np.abs(t2.get() - ref["t2"]).max() < 1e-12 and np.abs(t1.get() - ref["t1"]).max() < 1e-12
        # the residuals alone, against the general (no symmetry assumed) entry points on explicitly dressed blocks
        r1, r2 = ctx.empty(t1.shape), ctx.empty(t2.shape)
        ctx.ccsd_residuals(fdev, t1, t2, r1, r2, is_dcd=dcsd)
        fd = ctx.empty(f.shape)
        ctx.dress_fock(fdev, t1, fd)
        g1, g2 = ctx.empty(t1.shape), ctx.empty(t2.shape)
        ctx.singles_residual(fd, t1, t2, g1)
        ctx.dress_V(t1, ("abij", "klij", "iajb", "iabj", "abcd"))
        ctx.doubles_residual(fd, t2, g2, is_dcd=dcsd, dressed=True, sym_ladder=False)
        assert np.abs(r1.get() - g1.get()).max() < 1e-12 and np.abs(r2.get() - g2.get()).max() < 1e-12
        ctx.ccsd_release()
    finally:
        ctx.close()


@pytest.mark.parametrize("dcsd", [False, True])
def test_ccsd_iterate_entry_host_logic(hostsim_lib, dcsd):
    check_ccsd_iterate_entry(hostsim_lib, dcsd)


@pytest.mark.gpu
@pytest.mark.parametrize("dcsd", [False, True])
def test_ccsd_iterate_entry_gpu(gpu_lib, dcsd):
    check_ccsd_iterate_entry(gpu_lib, dcsd)
