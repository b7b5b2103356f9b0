"""CPU (host simulator standing in for the kernels): host logic added in round 2 — exchange-symmetry guard of the
symmetry-reduced path, buffers released with their context, block-shape validation, fused reductions, fixed-buffer
iteration with and without DIIS, in-place contract of ``amps``."""
import contextlib
import ctypes as C
import io

import numpy as np
import pytest

from oracle import cc_oracle as oc
from oracle.cases import random_case, synthetic_case
from pymes_amd import _lib
from pymes_amd.device import Context
from pymes_amd.solver.ccd import CCD
from pymes_amd.solver.ccsd import CCSD


@pytest.fixture()
def sim(hostsim_lib, monkeypatch):
    monkeypatch.setattr(_lib, "_default", hostsim_lib)
    return hostsim_lib


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def live(lib):
    n = C.c_int64()
    lib.call("pymes_live_allocations", C.byref(n))
    return n.value


def test_exchange_asymmetry(sim):
    no, nv = 3, 4
    f, V, t1, t2 = random_case(no, nv, 5, symmetric=True)
    ctx = Context(no, nv)
    ctx.set_V_pqrs(V)
    asym, vmax = ctx.V_exchange_asymmetry()
    assert asym < 1e-14 and abs(vmax - np.abs(V).max()) < 1e-15 and ctx.V_exchange_symmetric()
    V2 = V.copy()
    V2[no + 1, 0, no + 2, 1] += 1e-6            # one element of V_aibj: its partner sits in V_iajb
    ctx.set_V_pqrs(V2)
    asym, _ = ctx.V_exchange_asymmetry()
    assert abs(asym - 1e-6) < 1e-12 and not ctx.V_exchange_symmetric()
    assert ctx.exchange_symmetric(ctx.array(t2))
    t2[1, 0, 2, 1] += 1e-9
    assert not ctx.exchange_symmetric(ctx.array(t2))
    ctx.close()
    # a context with only some blocks: a missing partner block means "unknown" -> not symmetric
    ctx = Context(no, nv)
    ctx.set_V_block("iajb", oc.split_blocks(no, V)["iajb"])
    assert ctx.V_exchange_asymmetry()[0] == np.inf
    ctx.close()


@pytest.mark.parametrize("kind", ["ccsd", "dcsd", "ccd"])
def test_non_symmetric_V_takes_the_general_path(sim, kind):
    """V without V_pqrs = V_qpsr must not be fed to the symmetry-reduced residual.  The oracle (= the
    reference's algebra) makes no symmetry assumption; three iterations without DIIS must agree with it."""
    no, nv = 3, 5
    f, V, _, _ = synthetic_case(no, nv, seed=1, scale=0.3)
    V = V + 0.01 * np.random.default_rng(3).standard_normal(V.shape)
    if kind == "ccd":
        ref = oc.ccd_solve(no, f, V, is_diis=False, delta_e=1e-30, max_iter=2)
        s = CCD(no, is_diis=False, delta_e=1e-30)
        s.max_iter = 2
        res = quiet(s.solve, f, V)
        e, t2 = res["ccd e"], res["t2 amp"]
    else:
        ref = oc.ccsd_solve(no, f, V, is_dcsd=(kind == "dcsd"), is_diis=False, delta_e=1e-30, max_iter=2)
        s = CCSD(no, is_diis=False, is_dcsd=(kind == "dcsd"), delta_e=1e-30)
        s.max_iter = 2
        res = quiet(s.solve, f, V)
        e, t2 = res["ccsd e"], res["t2"]
    assert abs(e - ref["e"]) < 1e-12
    assert np.abs(t2 - ref["t2"]).max() < 1e-12


def test_context_releases_every_buffer(sim):
    """buffers handed out by a context (DIIS history, state arrays, pool) die with it."""
    no, nv = 4, 12
    f, V, _, _ = synthetic_case(no, nv, seed=0, scale=0.3)
    base = live(sim)
    for _ in range(2):
        quiet(CCSD(no, delta_e=1e-9).solve, f, V)
        quiet(CCD(no, delta_e=1e-9).solve, f, V)
        assert live(sim) == base
    ctx = Context(no, nv)
    keep = [ctx.empty((100,)) for _ in range(5)]      # still referenced by the caller when the context closes
    assert live(sim) > base
    ctx.close()
    assert live(sim) == base
    del keep                                          # dead handles: freeing them is a no-op, not a double free
    assert live(sim) == base


def test_block_shape_is_validated(sim):
    no, nv = 2, 3
    ctx = Context(no, nv)
    with pytest.raises(ValueError, match="must have shape"):
        ctx.set_V_block("abij", np.zeros((nv, nv, no, no + 1)))
    with pytest.raises(ValueError, match="must have shape"):
        ctx.set_V_block("iajb", np.zeros((nv, no, nv, no)))         # transposed block
    bad = np.zeros(7)
    with pytest.raises(_lib.PymesError, match="elements for this context"):
        ctx.lib.call("pymes_set_V_block", ctx.handle, b"abij", _lib.host_ptr(bad), 7, 0, None)
    ctx.close()


def test_fused_reductions(sim):
    no, nv = 3, 4
    f, V, t1, t2 = random_case(no, nv, 8, symmetric=False)
    rng = np.random.default_rng(0)
    dt2 = rng.standard_normal(t2.shape)
    ctx = Context(no, nv)
    ctx.set_V_pqrs(V)
    Vb = oc.split_blocks(no, V)
    e1, ed, ex, nt, nr, n1 = ctx.energy_norms(ctx.array(f), ctx.array(t1), ctx.array(t2), ctx.array(dt2))
    assert abs(n1 - (t1 ** 2).sum()) < 1e-13
    ref = oc.ccsd_energy(f[:no, no:], t1, t2, Vb["ijab"])
    assert np.allclose([e1, ed, ex], ref, rtol=0, atol=1e-13)
    assert abs(nt - (t2 ** 2).sum()) < 1e-12 and abs(nr - (dt2 ** 2).sum()) < 1e-12
    _, ed, ex, _, nr0, n10 = ctx.energy_norms(None, None, ctx.array(t2))
    assert np.allclose([ed, ex], oc.ccd_energy(t2, Vb["ijab"]), rtol=0, atol=1e-13) and nr0 == 0.0 and n10 == 0.0
    # pairs of different lengths in one call
    a, b = rng.standard_normal(7), rng.standard_normal(1000)
    out = ctx.dots([ctx.array(a), ctx.array(b)], [ctx.array(a), ctx.array(b)])
    assert np.allclose(out, [a @ a, b @ b], rtol=1e-14)
    # out-of-place update
    eps = f.diagonal()
    ctx.set_orbital_energies(eps[:no], eps[no:])
    r = rng.standard_normal(t2.shape)
    tn, dt = ctx.empty(t2.shape), ctx.empty(t2.shape)
    ctx.cc_update_to(tn, dt, ctx.array(t2), ctx.array(r), 0.3, 0.7)
    D = oc.denominators(eps[:no], eps[no:], 0.3)[1]
    assert np.abs(dt.get() - r * D).max() < 1e-14 and np.abs(tn.get() - (t2 + 0.7 * r * D)).max() < 1e-14
    ctx.close()


@pytest.mark.parametrize("diis", [True, False])
def test_amps_are_updated_in_place(sim, diis):
    """ccsd.py:132,178-179: the caller's arrays are the iteration's arrays — after the first update with DIIS (the
    mixer then hands back fresh arrays, diis.py:97-103), after every update without it."""
    no, nv = 3, 5
    f, V, _, _ = synthetic_case(no, nv, seed=2, scale=0.3)
    r0 = oc.ccsd_solve(no, f, V, is_diis=diis, delta_e=1e-30, max_iter=0)      # one pass from MP2
    t1, t2 = r0["t1"].copy(), r0["t2"].copy()
    ref = oc.ccsd_solve(no, f, V, is_diis=diis, delta_e=1e-30, max_iter=2, amps=[t1.copy(), t2.copy()])
    a1, a2 = t1.copy(), t2.copy()
    s = CCSD(no, is_diis=diis, delta_e=1e-30)
    s.max_iter = 2
    res = quiet(s.solve, f, V, amps=[a1, a2])
    assert not np.array_equal(a2, t2)               # updated in place
    if not diis:
        assert np.array_equal(a2, res["t2"]) and np.array_equal(a1, res["t1"])
    assert abs(res["ccsd e"] - ref["e"]) < 1e-12


def test_t1_zero_shortcut_is_exact(sim, monkeypatch):
    """T1 = 0 exactly lets the solver skip every T1 dressing (exp(-T1) H exp(T1) = H): same energies and amplitudes as
    the full path, on a momentum-conserving-like problem (T1 stays zero: diagonal blocks only) and on a generic one
    (T1 = 0 in the first iteration only)."""
    no, nv = 3, 5
    f, V, _, _ = synthetic_case(no, nv, seed=4, scale=0.3)
    for dcsd in (False, True):
        res = {}
        for off in ("", "1"):
            if off:
                monkeypatch.setenv("PYMES_NO_T1_SHORTCUT", "1")
            else:
                monkeypatch.delenv("PYMES_NO_T1_SHORTCUT", raising=False)
            s = CCSD(no, delta_e=1e-11, is_dcsd=dcsd)
            res[off] = (quiet(s.solve, f, V), s.iterations)
        assert res[""][1] == res["1"][1]
        assert abs(res[""][0]["ccsd e"] - res["1"][0]["ccsd e"]) < 1e-13
        assert np.abs(res[""][0]["t2"] - res["1"][0]["t2"]).max() < 1e-12
        ref = oc.ccsd_solve(no, f, V, is_dcsd=dcsd, delta_e=1e-11)
        assert abs(res[""][0]["ccsd e"] - ref["e"]) < 1e-11 and res[""][1] == ref["iterations"]


def test_mixer_history_survives_solve_calls(sim):
    """ccsd.py:42 / diis.py:16-112: the mixer of a solver instance is never reset, a second solve() starts from the
    history of the first.  The oracle restates that with a shared mixer object."""
    no, nv = 3, 5
    f, V, _, _ = synthetic_case(no, nv, seed=6, scale=0.3)
    f2, V2, _, _ = synthetic_case(no, nv, seed=7, scale=0.3)
    mixer = oc.Diis(6)
    ref1 = oc.ccsd_solve(no, f, V, delta_e=1e-6, mixer=mixer)
    ref2 = oc.ccsd_solve(no, f2, V2, delta_e=1e-10, mixer=mixer)
    fresh = oc.ccsd_solve(no, f2, V2, delta_e=1e-10)
    s = CCSD(no, delta_e=1e-6)
    r1 = quiet(s.solve, f, V)
    assert abs(r1["ccsd e"] - ref1["e"]) < 1e-12
    s.delta_e = 1e-10
    r2 = quiet(s.solve, f2, V2)
    assert abs(r2["ccsd e"] - ref2["e"]) < 1e-11 and s.iterations == ref2["iterations"]
    # (upstream's never-reset history steers the second solve somewhere else entirely — a stale subspace of another
    # problem; the point here is only that the drop-in does what the reference does)
    assert abs(ref2["e"] - fresh["e"]) > 1e-6
    # a different problem size drops the parked history instead of mixing incompatible vectors
    f3, V3, _, _ = synthetic_case(no, nv + 1, seed=8, scale=0.3)
    r3 = quiet(s.solve, f3, V3)
    assert abs(r3["ccsd e"] - oc.ccsd_solve(no, f3, V3, delta_e=1e-10)["e"]) < 1e-9


def check_mixer_against_reference_golden(lib):
    """pymes_amd.mixer.diis.DIIS — host solve and the device-resident step (pymes_diis_step) — on the seeded sequence of
    tests/golden/diis.json, which the reference's DIIS.mix produced (oracle/make_golden.py): coefficients of every call,
    the final L including the full-subspace quirk."""
    import json
    import os
    from pymes_amd.mixer.diis import DIIS
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "diis.json")))
    for mode in ("numpy", "native", "device"):          # host numpy.linalg / one native call / device-resident step
        on_device = mode == "device"
        rng = np.random.default_rng(g["seed"])
        ctx = Context(2, 3, lib=lib)
        mixer = DIIS(dim_space=6)
        try:
            for it in range(10):
                err = [rng.standard_normal((3, 2)) * 0.5 ** it, rng.standard_normal((3, 3, 2, 2)) * 0.5 ** it]
                amp = [rng.standard_normal((3, 2)), rng.standard_normal((3, 3, 2, 2))]
                out = quiet(mixer.mix, [ctx.array(e) for e in err], [ctx.array(a) for a in amp], on_device=on_device,
                            native=(mode == "native"))
                if on_device:
                    quiet(mixer.log_last)
                assert np.allclose(mixer.last_coefficients, g["coeffs"][it], rtol=1e-9, atol=1e-11), (mode, it)
                if it >= 2:      # the extrapolated amplitudes are that combination of the stored ones
                    want = sum(c * a[1].get() for c, a in zip(mixer.last_coefficients, mixer.amplitude_list))
                    assert np.abs(out[1].get() - want).max() < 1e-12, (mode, it)
            mixer._refresh_host()
            assert np.allclose(mixer.L, np.array(g["L_final"]), rtol=1e-12, atol=1e-14), mode
            assert mixer.L[4, 4] == 0.0 and np.all(mixer.L[4, :4] == 0.0)
            assert all(o.get().shape == s for o, s in zip(out, ((3, 2), (3, 3, 2, 2))))
        finally:
            ctx.close()


def test_mixer_against_reference_golden(sim):
    check_mixer_against_reference_golden(sim)


def test_mixer_near_singular_and_non_finite_subspace(sim):
    """the native step (csrc/diis_small.h: cyclic Jacobi + LU in place of numpy's eigh / inv) on a linearly
    dependent history — the same error vector twice, |lambda_min| < 1e-12, the pseudo-inverse branch of diis.py:85-93 — gives
    numpy's coefficients; a non-finite overlap (a diverged iteration) is refused like numpy.linalg refuses it, instead of
    writing NaN amplitudes."""
    from pymes_amd.device import PymesError
    from pymes_amd.mixer.diis import DIIS
    rng = np.random.default_rng(3)
    errs = [rng.standard_normal((3, 2)), rng.standard_normal((3, 3, 2, 2))]
    amps = [[rng.standard_normal((3, 2)), rng.standard_normal((3, 3, 2, 2))] for _ in range(3)]
    coeffs = {}
    for mode in ("numpy", "native"):
        ctx = Context(2, 3, lib=sim)
        mixer = DIIS(dim_space=6)
        try:
            for it in range(3):          # identical error vectors: L is singular from the second call on
                quiet(mixer.mix, [ctx.array(e) for e in errs], [ctx.array(a) for a in amps[it]], native=(mode == "native"))
            assert mode == "numpy" or mixer.last_dependent          # (the pseudo-inverse branch was taken)
            coeffs[mode] = np.array(mixer.last_coefficients)
        finally:
            ctx.close()
    assert np.allclose(coeffs["native"], coeffs["numpy"], rtol=1e-9, atol=1e-11), coeffs
    ctx = Context(2, 3, lib=sim)
    mixer = DIIS(dim_space=6)
    try:
        quiet(mixer.mix, [ctx.array(e) for e in errs], [ctx.array(a) for a in amps[0]], native=True)
        bad = [errs[0].copy(), errs[1].copy()]
        bad[1][0, 0, 0, 0] = np.nan
        with pytest.raises(PymesError):
            quiet(mixer.mix, [ctx.array(e) for e in bad], [ctx.array(a) for a in amps[1]], native=True)
    finally:
        ctx.close()


@pytest.mark.parametrize("solver", ["ccsd", "ccd"])
def test_device_resident_diis_gives_the_same_solve(sim, monkeypatch, solver):
    """PYMES_DEVICE_DIIS=1: overlaps, subspace solve (pymes_diis_step) and extrapolation without a host round trip — same
    iteration count and energy as the oracle's run of the reference algorithm (quirk included), and as the host solve."""
    no, nv = 3, 6
    f, V, _, _ = synthetic_case(no, nv, seed=4, scale=0.3)
    ref = oc.ccsd_solve(no, f, V, delta_e=1e-11) if solver == "ccsd" else oc.ccd_solve(no, f, V, delta_e=1e-11)
    monkeypatch.setenv("PYMES_DEVICE_DIIS", "1")
    s = CCSD(no, delta_e=1e-11) if solver == "ccsd" else CCD(no, delta_e=1e-11)
    r = quiet(s.solve, f, V)
    e = r["ccsd e"] if solver == "ccsd" else r["ccd e"]
    assert s.iterations == ref["iterations"] and abs(e - ref["e"]) < 1e-11
    assert s.mixer._state is None and s.mixer.L.shape == (7, 7)          # the state came back to the host when the context closed


def test_mixer_history_follows_caller_owned_contexts(sim):
    """with caller-owned DeviceIntegrals the DIIS history stays on that context after solve().  A second solve()
    on another context must neither read vectors of a closed context nor pool foreign pointers: the history is parked on
    the host when its context closes (Context.on_close) or migrated from a live one, and the reference's never-reset
    semantics (diis.py:16-112) is kept either way."""
    from pymes_amd.integral.device import DeviceIntegrals
    no, nv = 3, 5
    f, V, _, _ = synthetic_case(no, nv, seed=6, scale=0.3)
    f2, V2, _, _ = synthetic_case(no, nv, seed=7, scale=0.3)
    base = live(sim)
    for close_first in (True, False):
        mixer = oc.Diis(6)
        ref1 = oc.ccsd_solve(no, f, V, delta_e=1e-6, mixer=mixer)
        ref2 = oc.ccsd_solve(no, f2, V2, delta_e=1e-10, mixer=mixer)
        s = CCSD(no, delta_e=1e-6)
        A = DeviceIntegrals.from_V_pqrs(no, V)
        r1 = quiet(s.solve, f, A)
        assert abs(r1["ccsd e"] - ref1["e"]) < 1e-12
        assert all(a.ctx is A.ctx for vec in s.mixer.error_list for a in vec)        # device-resident, on A
        if close_first:
            A.ctx.close()                                                            # parks the history on the host
            assert all(isinstance(a, np.ndarray) for vec in s.mixer.error_list for a in vec)
        B = DeviceIntegrals.from_V_pqrs(no, V2)
        s.delta_e = 1e-10
        r2 = quiet(s.solve, f2, B)
        assert abs(r2["ccsd e"] - ref2["e"]) < 1e-11 and s.iterations == ref2["iterations"]
        assert all(a.ctx is B.ctx for vec in s.mixer.error_list + s.mixer.amplitude_list for a in vec)
        # a history whose context died without parking (handle gone) is dropped, not dereferenced
        h = B.ctx.handle
        B.ctx._closing = []
        B.ctx.close()
        assert h is not None and B.ctx.handle is None
        C_ = DeviceIntegrals.from_V_pqrs(no, V)
        s.delta_e = 1e-9
        r3 = quiet(s.solve, f, C_)
        assert abs(r3["ccsd e"] - oc.ccsd_solve(no, f, V, delta_e=1e-9)["e"]) < 1e-9
        C_.ctx.close()
        if not close_first:
            A.ctx.close()
        del s, r1, r2, r3
        assert live(sim) == base


def test_recycled_buffers_are_given_back_on_allocation_failure(sim, monkeypatch):
    """Context.empty retries after trimming its spare list when pymes_malloc fails."""
    ctx = Context(2, 3)
    a = ctx.empty((1000,))
    a.free()
    assert ctx._spare_bytes == 8000
    calls = {"n": 0}
    real = ctx.lib.call

    def flaky(name, *args):
        if name == "pymes_malloc":
            calls["n"] += 1
            if calls["n"] == 1:
                raise _lib.PymesError("out of memory")
        return real(name, *args)
    monkeypatch.setattr(ctx.lib, "call", flaky)
    b = ctx.empty((77,))
    assert calls["n"] == 2 and ctx._spare_bytes == 0 and b.ptr
    monkeypatch.setattr(ctx.lib, "call", real)
    ctx.close()


def test_singles_sums_are_partial_traces_of_the_ring_intermediate():
    """for V_pqrs = V_qpsr and T_abij = T_baji the V.T sums of ccsd.py:434 / :436 equal those of X_ki / X_ac
    (ccd.py:213-220), and both are partial traces of Y = Vd Tt_d (ccd.py:202) — the identity the product path relies on
    when it drops the four products.  Pure numpy on the oracle's index conventions."""
    rng = np.random.default_rng(7)
    o, v = 3, 5
    n = o + v
    V = rng.standard_normal((n, n, n, n))
    V = V + V.transpose(1, 0, 3, 2)                       # electron-exchange symmetry only (not hermitian: TC-like)
    Vijab = V[:o, :o, o:, o:]
    T = rng.standard_normal((v, v, o, o))
    T = T + T.transpose(1, 0, 3, 2)
    Tt = 2.0 * T - T.transpose(1, 0, 2, 3)                # ccd.py:199
    Tp = 2.0 * T - T.transpose(0, 1, 3, 2)                # ccsd.py:430
    S_ac = np.einsum("adkl,lkdc->ac", Tt, Vijab)          # ccd.py:213
    S_ki = np.einsum("cdil,lkdc->ki", Tt, Vijab)          # ccd.py:215
    assert np.abs(np.einsum("jkcb,abjk->ac", Vijab, Tp) - S_ac).max() < 1e-12       # ccsd.py:436
    assert np.abs(np.einsum("kjbc,bcij->ki", Vijab, Tp) - S_ki).max() < 1e-12       # ccsd.py:434
    Y = np.einsum("klcd,dblj->ckbj", Vijab, Tt)           # ccd.py:202 as the pair matrix [(c,k),(b,j)]
    assert np.abs(np.einsum("ckak->ac", Y) - S_ac).max() < 1e-12
    assert np.abs(np.einsum("ckci->ki", Y) - S_ki).max() < 1e-12


def test_slab_halves_equal_the_whole(sim):
    """include/pymes_amd.h: PYMES_SLAB_RINGS_ONLY + PYMES_SLAB_LADDERS_ONLY (one process per GPU issues them separately so
    that the all-gathers of the ring rows fly during the ladders) produce what the single call produces, for every
    simulated rank, with and without the T1 dressing."""
    no, nv = 3, 5
    f, V, t1, t2 = random_case(no, nv, 11, symmetric=True)
    ctx = Context(no, nv)
    ctx.set_V_pqrs(V)
    ov, npp = no * nv, nv * (nv + 1) // 2
    dF, dT1, dT2 = ctx.array(f), ctx.array(t1), ctx.array(t2)
    ctx.dress_V(dT1, ["klij", "iajb", "iabj", "abcd"])          # abcd: read by the ladders of the mode without t1 / QK
    for world in (1, 3):
        pad = lambda n: -(-n // world) * world
        for dcd in (False, True):
            for with_t1 in (False, True):
                outs = []
                for split in (False, True):
                    ETd, ETx = ctx.zeros((pad(ov), ov)), ctx.zeros((pad(ov), ov))
                    L, QK = ctx.zeros((pad(npp), no * no)), ctx.zeros((pad(ov), no * no))
                    kw = dict(is_dcd=dcd, dressed=True)
                    if with_t1:
                        kw.update(t1=dT1, QK=QK)
                    for rank in range(world):
                        if split:
                            ctx.residual_slab(dF, dT2, ETd, ETx, L, rank, world, part="rings", **kw)
                            ctx.residual_slab(dF, dT2, ETd, ETx, L, rank, world, part="ladders", **kw)
                        else:
                            ctx.residual_slab(dF, dT2, ETd, ETx, L, rank, world, **kw)
                    outs.append([x.get() for x in (ETd, ETx, L, QK)])
                for a, b in zip(*outs):
                    assert np.array_equal(a, b), (world, dcd, with_t1)
    ctx.close()


def test_energy_norms_over_pairs(sim):
    """Host logic of pymes_energy_norms_pairs (the GPU kernel has the same test in test_gpu_kernels.py)."""
    no, nv = 3, 5
    f, V, t1, t2 = random_case(no, nv, 3, symmetric=True)
    rng = np.random.default_rng(5)
    ctx = Context(no, nv)
    ctx.set_V_pqrs(V)
    dF, dT1, dT2, dD = ctx.array(f), ctx.array(t1), ctx.array(t2), ctx.array(rng.standard_normal(t2.shape))
    full = np.array(ctx.energy_norms(dF, dT1, dT2, dD))
    npp = nv * (nv + 1) // 2
    for world in (1, 2, 4):
        chunk = -(-npp // world)
        tot = np.zeros(6)
        for rank in range(world):
            tc, dtc = ctx.zeros((chunk, 2, no * no)), ctx.zeros((chunk, 2, no * no))
            ctx.pairs_pack(dT2, tc, rank, world)
            ctx.pairs_pack(dD, dtc, rank, world)
            tot += ctx.energy_norms_pairs(dF, dT1, tc, dtc, rank, world)
        assert np.allclose(tot, full, rtol=1e-12, atol=1e-13), world
    ctx.close()


def test_hole_ladder_packed(sim):
    """pymes_hole_ladder_packed: rows of the pair-packed L += sum_kl I_klij X_abkl for symmetric I and X; unpacked, it is
    the plain product (eom_ccsd.py:380-382)."""
    no, nv = 3, 5
    rng = np.random.default_rng(9)
    X = rng.standard_normal((nv, nv, no, no))
    X = X + X.transpose(1, 0, 3, 2)
    I = rng.standard_normal((no, no, no, no))
    I = I + I.transpose(1, 0, 3, 2)
    ctx = Context(no, nv)
    npp = nv * (nv + 1) // 2
    L = ctx.zeros((npp, no * no))
    dX, dI = ctx.array(X), ctx.array(I)
    for lo, hi in ((0, 4), (4, npp)):                     # two row chunks, as two ranks would
        ctx.hole_ladder_packed(dX, dI, L, lo, hi)
    out = ctx.zeros(X.shape)
    ctx.ladder_sym_unpack(L, out, beta=0.0)
    assert np.abs(out.get() - np.einsum("abkl,klij->abij", X, I)).max() < 1e-12
    # with y: I + V_klcd y_cdij, the V.T part formed pair-packed too
    f, V, _, _ = random_case(no, nv, 4, symmetric=True)
    ctx.set_V_pqrs(V)
    Y = rng.standard_normal((nv, nv, no, no))
    Y = Y + Y.transpose(1, 0, 3, 2)
    L.zero_()
    ctx.hole_ladder_packed(dX, dI, L, 0, npp, y=ctx.array(Y))
    ctx.ladder_sym_unpack(L, out, beta=0.0)
    Ifull = I + np.einsum("klcd,cdij->klij", V[:no, :no, no:, no:], Y)
    assert np.abs(out.get() - np.einsum("abkl,klij->abij", X, Ifull)).max() < 1e-11
    ctx.close()


def test_unset_orbital_energies_and_non_finite_amplitudes_fail_loudly(sim):
    """mp2 / the amplitude updates divide by orbital-energy differences: calling them on a context whose energies were
    never set is an error, not a read of uninitialised memory; inf / NaN amplitudes never count as exchange-symmetric."""
    from pymes_amd._lib import PymesError
    no, nv = 2, 3
    f, V, t1, t2 = random_case(no, nv, 2, symmetric=True)
    ctx = Context(no, nv)
    ctx.set_V_pqrs(V)
    with pytest.raises(PymesError, match="orbital energies"):
        ctx.mp2(ctx.empty(t2.shape), 0.0)
    with pytest.raises(PymesError, match="orbital energies"):
        ctx.cc_update(ctx.array(t2), ctx.empty(t2.shape), ctx.array(t2), 0.0, 1.0)
    ctx.set_orbital_energies(np.diag(f)[:no], np.diag(f)[no:])
    ctx.mp2(ctx.empty(t2.shape), 0.0)
    bad = t2.copy()
    bad[0, 0, 0, 0] = np.inf
    assert ctx.exchange_symmetric(ctx.array(t2)) and not ctx.exchange_symmetric(ctx.array(bad))
    ctx.close()


def test_ring_terms_in_four_products():
    """for V_pqrs = V_qpsr and T_abij = T_baji the ten o^3v^3 products of ccd.py:190-191, :199-204, :233-240 equal
    the C / D form  Ex_d = 1/2 Tt_d (2 Wd - Ud^T + 1/2 Ld Tt_d) - 1/2 Xc,  Ex_x = -Xc,  Xc = Tx (Ud^T - 1/2 Vx Tx)  (four
    products; DCSD: Vd in place of Ld/2 and no Vx Tx: three), and the small V.T sums are (3 tr(Vx Tx) + tr(Ld Tt_d)) / 4.
    Pure numpy on the reference's expressions — what cc.cpp::residual_slab relies on."""
    rng = np.random.default_rng(3)
    o, v = 3, 4
    n, ov = o + v, o * v
    V = rng.standard_normal((n, n, n, n))
    V = V + V.transpose(1, 0, 3, 2)                               # electron-exchange symmetry only
    T = rng.standard_normal((v, v, o, o))
    T = T + T.transpose(1, 0, 3, 2)
    Vijab, Viajb, Viabj = V[:o, :o, o:, o:], V[:o, o:, :o, o:], V[:o, o:, o:, :o]
    Tt = 2.0 * T - T.transpose(1, 0, 2, 3)
    Td = T.transpose(0, 2, 1, 3).reshape(ov, ov)
    Tx = T.transpose(0, 3, 1, 2).reshape(ov, ov)
    Ttd = Tt.transpose(0, 2, 1, 3).reshape(ov, ov)
    Vd = np.einsum("klcd->ckdl", Vijab).reshape(ov, ov)
    Vx = np.einsum("klcd->cldk", Vijab).reshape(ov, ov)
    Wd = np.einsum("kbcj->ckbj", Viabj).reshape(ov, ov)
    UdT = np.einsum("kbjc->ckbj", Viajb).reshape(ov, ov)
    for quad in (True, False):
        # the reference's sequence
        R = np.zeros_like(T)
        if quad:
            R += np.einsum("alcj,cbil->abij", np.einsum("klcd,adkj->alcj", Vijab, T), T)            # :190-191
        R += np.einsum("acik,cbkj->abij", Tt, np.einsum("klcd,dblj->cbkj", Vijab, Tt))               # :202-204
        Ex = -np.einsum("kaic,cbkj->abij", Viajb, T) - np.einsum("kbic,ackj->abij", Viajb, T)        # :233-234
        Ex += np.einsum("acik,kbcj->abij", Tt, Viabj)                                                # :235
        if quad:
            Xa = np.einsum("klcd,daki->alci", Vijab, T)                                              # :238
            Ex += np.einsum("alci,bclj->abij", Xa, T) - np.einsum("alci,cblj->abij", Xa, T)          # :239-240
        ref = R + Ex + Ex.transpose(1, 0, 3, 2)
        # the product path
        M = 2.0 * Wd - UdT + (0.5 * (2.0 * Vd - Vx) if quad else Vd) @ Ttd
        X3 = -UdT + (0.5 * Vx @ Tx if quad else 0.0)
        ETx = X3.T @ Tx.T
        ETd = 0.5 * (M.T @ Ttd.T) + 0.5 * ETx
        new = np.einsum("aibj->abij", ETd.reshape(v, o, v, o)) + np.einsum("ajbi->abij", ETx.reshape(v, o, v, o))
        new = new + new.transpose(1, 0, 3, 2)
        assert np.abs(new - ref).max() < 1e-11 * np.abs(ref).max(), quad
    S_ac = np.einsum("adkl,lkdc->ac", Tt, Vijab)
    S_ki = np.einsum("cdil,lkdc->ki", Tt, Vijab)
    Z4, U4 = ((2.0 * Vd - Vx) @ Ttd).reshape(v, o, v, o), (Vx @ Tx).reshape(v, o, v, o)
    assert np.abs((3.0 * np.einsum("ckak->ac", U4) + np.einsum("ckak->ac", Z4)) / 4.0 - S_ac).max() < 1e-12
    assert np.abs((3.0 * np.einsum("ckci->ki", U4) + np.einsum("ckci->ki", Z4)) / 4.0 - S_ki).max() < 1e-12


@pytest.mark.parametrize("minus", [False, True])
def test_ladder_dress_host_statement(hostsim_lib, minus):
    """The CPU stand-in of the dressing kernel against the numpy statement the GPU test uses."""
    from tests.test_gpu_kernels import DRESS_CASES, check_ladder_dress
    for i, (no, nv, ld, r0, r1) in enumerate(DRESS_CASES[:5]):
        check_ladder_dress(hostsim_lib, no, nv, ld, r0, r1, minus, seed=i)


def test_dressed_fock_from_its_six_blocks(hostsim_lib):
    """Host logic of dress_fock with the fused direct / exchange sums (CPU stand-in), from the six blocks alone."""
    from tests.test_gpu_kernels import check_dressed_fock_from_blocks
    for i, (no, nv) in enumerate(((3, 7), (2, 9), (1, 4))):
        check_dressed_fock_from_blocks(hostsim_lib, no, nv, i)


def test_host_diis_solve_matches_numpy(sim):
    """pymes_diis_solve (the small algebra of a DIIS step in C on host arrays — what the one-process-per-GPU path calls after
    it has summed the overlaps over the ranks) against the mixer's numpy.linalg route (diis.py:56-103): subspace growing to
    its full size, the full-subspace quirk, overlaps shrinking into the pseudo-inverse branch."""
    import ctypes as C
    from pymes_amd import _lib
    from pymes_amd.mixer.diis import DIIS
    rng = np.random.default_rng(17)
    ref = DIIS(dim_space=6)
    state = np.zeros(96)
    state[0] = 1.0
    scale = 1.0
    for it in range(14):
        was_full = it >= 6
        m = min(it + 1, 6)
        vecs = rng.standard_normal((m, 40)) * scale
        overlaps = vecs @ vecs[-1]                      # <e_i, e_new>, the newest vector last
        if was_full:                                     # the mixer drops the oldest vector before it extends L
            pass
        ref._update_L(overlaps, was_full)
        c_ref = ref._solve_on_this_thread()
        sim.call("pymes_diis_solve", _lib.host_ptr(state), _lib.host_ptr(np.ascontiguousarray(overlaps)), 1, m, int(was_full))
        n = int(state[0])
        assert n == m + 1
        assert np.abs(state[1:82].reshape(9, 9)[:n, :n] - ref.L).max() == 0.0
        c = state[82:82 + n]
        assert np.abs(c - c_ref).max() <= 1e-9 * max(1.0, np.abs(c_ref).max()), (it, c, c_ref)
        assert state[91] == (1.0 if np.any(np.abs(np.linalg.eigvalsh(ref.L)) < 1e-12) else 0.0)
        scale *= 0.1                                     # overlaps fall below 1e-12: the pseudo-inverse branch is taken
    # two types: the overlaps are summed over the types in the reference's order
    st2 = np.zeros(96); st2[0] = 1.0
    sim.call("pymes_diis_solve", _lib.host_ptr(st2), _lib.host_ptr(np.array([0.3, 0.2])), 2, 1, 0)
    assert st2[1] == 0.5 and abs(st2[82] - 1.0) < 1e-15
    with pytest.raises(Exception):
        sim.call("pymes_diis_solve", _lib.host_ptr(st2), _lib.host_ptr(np.array([0.3])), 1, 9, 0)
