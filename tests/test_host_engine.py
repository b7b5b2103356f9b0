"""CPU: HOST logic of the engine (contraction planner, CC term sequences, C-ABI, Python
drop-in classes) linked against the host simulator (tests/hostsim).  The kernels
themselves are tested on the GPU (test_gpu_*.py)."""
import json
import contextlib
import io
import os

import numpy as np
import pytest

from oracle import cc_oracle as oc
from oracle import io_oracle as oio
from oracle.cases import random_case, synthetic_case
from pymes_amd import _lib
from pymes_amd.device import Context, DeviceArray

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SOLVES = json.load(open(os.path.join(GOLD, "solves.json")))


@pytest.fixture()
def sim(hostsim_lib, monkeypatch):
    """Route the package's default library to the host simulator for this test only."""
    monkeypatch.setattr(_lib, "_default", hostsim_lib)
    return hostsim_lib


def test_product_loader_rejects_non_hip_backend(hostsim_lib):
    with pytest.raises(_lib.PymesError):
        _lib.Library(hostsim_lib.path)          # the package itself never accepts the simulator


@pytest.mark.parametrize("spec,shapes,batch", [
    ("abcd,cdij->abij", ((5, 5, 5, 5), (5, 5, 3, 3)), ""),
    ("klcd,adkj->alcj", ((3, 3, 5, 5), (5, 5, 3, 3)), ""),
    ("klij,abkl->abij", ((3, 3, 3, 3), (5, 5, 3, 3)), ""),
    ("ki,akbj->aibj", ((3, 3), (5, 3, 5, 3)), "a"),
    ("bj,jabc->ac", ((5, 3), (3, 5, 5, 5)), ""),
    ("pqxs,xr->pqrs", ((4, 3, 6, 2), (6, 5)), "pq"),
    ("ai,bj->abij", ((5, 3), (4, 2)), ""),
    ("Qpr,Qqs->pqrs", ((7, 3, 4), (7, 5, 2)), "pq"),
    ("zab,zbc->zac", ((4, 3, 5), (4, 5, 2)), ""),
])
def test_contraction_planner(sim, spec, shapes, batch):
    rng = np.random.default_rng(0)
    A, B = rng.standard_normal(shapes[0]), rng.standard_normal(shapes[1])
    ctx = Context(2, 2, workspace_bytes=1 << 22)
    ref = np.einsum(spec, A, B)
    assert np.abs(ctx.contract(spec, ctx.array(A), ctx.array(B), batch=batch).get() - ref).max() < 1e-13
    C0 = rng.standard_normal(ref.shape)
    dC = ctx.array(C0)
    ctx.contract(spec, ctx.array(A), ctx.array(B), out=dC, alpha=-0.5, beta=2.0, batch=batch)
    assert np.abs(dC.get() - (2.0 * C0 - 0.5 * ref)).max() < 1e-13
    ctx.close()


def test_planner_errors(sim):
    ctx = Context(2, 2, workspace_bytes=1 << 20)
    A, B = ctx.zeros((2, 3)), ctx.zeros((3, 4))
    with pytest.raises(_lib.PymesError, match="only one tensor"):
        ctx.lib.call("pymes_contract", ctx.handle, 1.0, A.ptr, b"ab", _lib.i64_array((2, 3)), None, B.ptr, b"bc",
                     _lib.i64_array((3, 4)), None, 0.0, A.ptr, b"ad", _lib.i64_array((2, 5)), None, b"")
    with pytest.raises(_lib.PymesError, match="extent mismatch"):
        ctx.lib.call("pymes_contract", ctx.handle, 1.0, A.ptr, b"ab", _lib.i64_array((2, 3)), None, B.ptr, b"bc",
                     _lib.i64_array((4, 4)), None, 0.0, A.ptr, b"ac", _lib.i64_array((2, 4)), None, b"")
    with pytest.raises(_lib.PymesError, match="has not been set"):
        ctx.V_block("abcd")
    with pytest.raises(_lib.PymesError, match="workspace exhausted"):
        big = Context(8, 8, workspace_bytes=4096)
        big.set_V_pqrs(np.zeros((16,) * 4))
        big.doubles_residual(big.zeros((16, 16)), big.zeros((8, 8, 8, 8)), big.zeros((8, 8, 8, 8)))
    ctx.close()


@pytest.mark.parametrize("no,nv,seed", [(1, 1, 3), (2, 3, 11), (3, 5, 12), (4, 7, 9)])
def test_cc_terms_vs_oracle(sim, no, nv, seed):
    f, V, t1, t2 = random_case(no, nv, seed, symmetric=False)
    Vb = oc.split_blocks(no, V)
    ctx = Context(no, nv)
    ctx.set_V_pqrs(V)
    dF, dT1, dT2 = ctx.array(f), ctx.array(t1), ctx.array(t2)
    fd = ctx.empty(f.shape)
    ctx.dress_fock(dF, dT1, fd)
    fd_ref = oc.dressed_fock(no, f, t1, Vb)
    assert np.abs(fd.get() - fd_ref).max() < 1e-13
    ctx.dress_V(dT1, oc.DRESSED_KEYS)
    Vd = oc.dressed_V(t1, Vb)
    for k in oc.DRESSED_KEYS:
        assert np.abs(ctx.V_block(k, dressed=True).get() - Vd[k]).max() < 1e-13, k
    r1 = ctx.empty(t1.shape)
    ctx.singles_residual(fd, dT1, dT2, r1)
    assert np.abs(r1.get() - oc.singles_residual(no, fd_ref, t1, t2, Vb)).max() < 1e-13
    for dcd in (False, True):
        r2 = ctx.empty(t2.shape)
        ctx.doubles_residual(fd, dT2, r2, is_dcd=dcd, dressed=True)
        assert np.abs(r2.get() - oc.ccsd_doubles_residual(no, fd_ref, t2, Vd, is_dcsd=dcd)).max() < 1e-12
        r2s = ctx.empty(t2.shape)
        ctx.doubles_residual(fd, dT2, r2s, is_dcd=dcd, dressed=True, skip_ladder=True)
        for lo, hi in ((0, nv // 2), (nv // 2, nv)):
            ctx.ladder(dT2, r2s, lo, hi, dressed=True, beta=1.0)
        assert np.abs(r2s.get() - r2.get()).max() < 1e-13
    ctx.close()


@pytest.mark.parametrize("no,nv,seed", [(1, 1, 1), (2, 3, 2), (3, 5, 3), (4, 6, 4)])
def test_symmetry_reduced_residual(sim, no, nv, seed):
    """Pair-packed ladder + merged ring products on exchange-symmetric (also non-hermitian) input."""
    f, V, t1, t2 = random_case(no, nv, seed, symmetric=True)
    if seed % 2:
        V = V + 0.05 * np.random.default_rng(seed).standard_normal(V.shape)
        V = 0.5 * (V + V.transpose(1, 0, 3, 2))
    Vb = oc.split_blocks(no, V)
    ctx = Context(no, nv)
    ctx.set_V_pqrs(V)
    dF, dT2 = ctx.array(f), ctx.array(t2)
    for dcd in (False, True):
        ref = oc.doubles_residual(no, f, t2, Vb["klij"], Vb["ijab"], Vb["abij"], Vb["iajb"], Vb["iabj"], Vb["abcd"],
                                  is_dcd=dcd)
        for rings in (False, True):
            r2 = ctx.empty(t2.shape)
            ctx.doubles_residual(dF, dT2, r2, is_dcd=dcd, sym_ladder=True, sym_rings=rings)
            assert np.abs(r2.get() - ref).max() < 1e-12
        calls = ctx.stats(reset=True)
    hole_ladder_check(ctx, no, nv, Vb, t2, 1e-12)
    ctx.close()


def hole_ladder_check(ctx, no, nv, Vb, t2, tol):
    """pymes_ladder_sym with hole_ladder = 1 / 2: ccd.py:175-187 pair-packed, in two row chunks."""
    npp = nv * (nv + 1) // 2
    dT2 = ctx.array(t2)
    for mode in (0, 1, 2):
        ref = np.einsum("abcd,cdij->abij", Vb["abcd"], t2)
        if mode:
            I = Vb["klij"] + (np.einsum("klcd,cdij->klij", Vb["ijab"], t2) if mode == 1 else 0.0)
            ref = ref + np.einsum("klij,abkl->abij", I, t2)
        L = ctx.zeros((npp, no * no))
        cut = npp // 2
        ctx.ladder_sym(dT2, L, 0, cut, hole_ladder=mode)
        ctx.ladder_sym(dT2, L, cut, npp, hole_ladder=mode)
        R = ctx.empty(t2.shape)
        ctx.ladder_sym_unpack(L, R, beta=0.0)
        assert np.abs(R.get() - ref).max() < tol * max(1.0, np.abs(ref).max()), mode


def sharded_residual_check(lib, cases, worlds, tol):
    """Every rank's slab computed one after the other into shared buffers (= a perfect exchange), then the
    replicated finish: must equal the oracle for any world size, with abcd dressed row-block-wise."""
    for no, nv, seed in cases:
        f, V, t1, t2 = random_case(no, nv, seed, symmetric=True)
        V = V + 0.05 * np.random.default_rng(seed).standard_normal(V.shape)
        V = 0.5 * (V + V.transpose(1, 0, 3, 2))
        Vb = oc.split_blocks(no, V)
        fd_ref = oc.dressed_fock(no, f, t1, Vb)
        Vd_ref = oc.dressed_V(t1, Vb)
        ctx = Context(no, nv, lib=lib)
        ctx.set_V_pqrs(V)
        ctx.set_orbital_energies(-1.0 - np.arange(no, dtype=float), 1.0 + np.arange(nv, dtype=float))
        dT1, dT2, dF = ctx.array(t1), ctx.array(t2), ctx.array(fd_ref)
        ov, npp = no * nv, nv * (nv + 1) // 2
        for world in worlds:
            ctx.dress_V(dT1, ["abij", "klij", "iajb", "iabj"])
            for dcd in (False, True):
                pad = lambda n: -(-n // world) * world
                ETd, ETx, L = ctx.zeros((pad(ov), ov)), ctx.zeros((pad(ov), ov)), ctx.zeros((pad(npp), no * no))
                for rank in range(world):
                    lo = min(rank * (-(-npp // world)), npp)
                    hi = min(lo + (-(-npp // world)), npp)
                    if hi > lo:      # rows a of V_abcd that the packed rows [lo,hi) touch
                        a_of = lambda r: int((np.sqrt(8.0 * r + 1.0) - 1.0) / 2.0 + 1e-9)
                        ctx.dress_abcd_rows(dT1, a_of(lo), a_of(hi - 1) + 1, lower_only=(world % 2 == 0))
                    ctx.residual_slab(dF, dT2, ETd, ETx, L, rank, world, is_dcd=dcd, dressed=True)
                r2 = ctx.empty(t2.shape)
                ctx.residual_finish(dF, dT2, ETd, ETx, L, r2, is_dcd=dcd, dressed=True)
                ref = oc.ccsd_doubles_residual(no, fd_ref, t2, Vd_ref, is_dcsd=dcd)
                assert np.abs(r2.get() - ref).max() < tol, (no, nv, world, dcd)
            # the same with the T1 dressing of V_abcd carried by the amplitudes (never dresses abcd)
            for dcd in (False, True):
                ctx.dress_V(dT1, ["klij", "iajb", "iabj"])      # V_abij and V_abcd are never dressed in this mode
                ctx.V_block("abij", dressed=True).zero_()
                ETd, ETx, L = ctx.zeros((pad(ov), ov)), ctx.zeros((pad(ov), ov)), ctx.zeros((pad(npp), no * no))
                QK = ctx.zeros((pad(ov), no * no))
                # every rank dresses only the range of the second index of V_iajb / V_iabj that its column slab reads
                ctx.V_block("iajb", dressed=True).zero_()
                ctx.V_block("iabj", dressed=True).zero_()
                Psum = np.zeros(ctx.slab_prepare_ws())       # K-sharded partial sums of the slab's small intermediates
                for rank in range(world):
                    Psum += ctx.slab_prepare(dT2, ctx.empty((ctx.slab_prepare_ws(),)), rank, world, is_dcd=dcd).get()
                dP = ctx.array(Psum)
                for rank in range(world):
                    c0 = min(rank * (-(-ov // world)), ov)
                    c1 = min(c0 + (-(-ov // world)), ov)
                    if c1 > c0:
                        ctx.dress_V(dT1, ["iajb", "iabj"], q_range=(c0 // no, -(-c1 // no)))
                    ctx.residual_slab(dF, dT2, ETd, ETx, L, rank, world, is_dcd=dcd, dressed=True, t1=dT1, QK=QK,
                                      P=dP if rank % 2 == 0 else None)
                r2 = ctx.empty(t2.shape)
                ctx.residual_finish(dF, dT2, ETd, ETx, L, r2, is_dcd=dcd, dressed=True, t1=dT1, QK=QK)
                ref = oc.ccsd_doubles_residual(no, fd_ref, t2, Vd_ref, is_dcsd=dcd)
                assert np.abs(r2.get() - ref).max() < tol, ("t1-side", no, nv, world, dcd)
                # K-sharded small intermediates: the partial sums of all ranks add up to the replicated result
                Wsum = np.zeros(ctx.dress_fock_ws())
                Xsum = np.zeros((nv, nv))
                for rank in range(world):
                    Wsum += ctx.dress_fock_partial(dT1, ctx.empty((ctx.dress_fock_ws(),)), rank, world).get()
                    Xsum += ctx.xvv_partial(dF, dT2, ctx.empty((nv, nv)), rank, world, is_dcd=dcd).get()
                fd2 = ctx.dress_fock_finish(ctx.array(f), dT1, ctx.array(Wsum), ctx.empty(f.shape)).get()
                assert np.abs(fd2 - fd_ref).max() < 1e-12
                Tt = 2.0 * t2 - t2.transpose(1, 0, 2, 3)
                Xref = fd_ref[no:, no:] - (0.5 if dcd else 1.0) * np.einsum("adkl,lkdc->ac", Tt, Vb["ijab"])
                assert np.abs(Xsum - Xref).max() < 1e-12
                # singles residual (ccsd.py:423-438) as K-sharded partial sums over the occupied summation index
                for reuse in (False, True):       # True: read the pair layouts residual_slab left behind for this t2
                    Rsum = np.zeros((nv, no))
                    for rank in range(world):
                        Rsum += ctx.singles_residual_partial(dF, dT1, dT2, ctx.empty((nv, no)), rank, world,
                                                             reuse_layouts=reuse).get()
                    assert np.abs(Rsum - oc.singles_residual(no, fd_ref, t1, t2, Vb)).max() < 1e-12, reuse
                dXvv = ctx.array(Xsum)
                if not lib.dll.pymes_backend().decode().startswith("hip") and no > 3:
                    continue            # the host simulator declares the fused pair kernels available for no <= 3 only
                # pair-sharded tail: every rank assembles R for its own pairs (compact), then update + unpack
                chunk = -(-npp // world)
                Rall = ctx.zeros((world * chunk, 2, no * no))
                Tall, dTall = ctx.zeros(Rall.shape), ctx.zeros(Rall.shape)
                piece = lambda arr, rank: DeviceArray(ctx, arr.ptr + 8 * rank * chunk * 2 * no * no,
                                                      (chunk, 2, no * no), owned=False, keepalive=arr)
                for rank in range(world):
                    ctx.residual_finish_pairs(dF, dT2, ETd, ETx, L, piece(Rall, rank), rank, world, dT1, QK, is_dcd=dcd,
                                              dressed=True, Xvv=dXvv if rank % 2 else None)
                    ctx.pairs_pack(dT2, piece(Tall, rank), rank, world)
                    ctx.cc_update_pairs(piece(Tall, rank), piece(dTall, rank), piece(Rall, rank), 0.25, 0.5, rank, world)
                full = ctx.zeros(t2.shape)
                assert np.abs(ctx.pairs_unpack(Rall, full, world).get() - ref).max() < tol, ("pairs", no, nv, world, dcd)
                # compact dots summed over the ranks = dots over the full arrays
                assert abs(float(ctx.dots([Rall], [Rall])[0]) - float(np.vdot(ref, ref))) < 1e-9 * max(1.0, np.vdot(ref, ref))
                tt, dtt = ctx.array(t2), ctx.empty(t2.shape)
                ctx.cc_update(tt, dtt, ctx.array(ref), 0.25, 0.5)
                assert np.abs(ctx.pairs_unpack(Tall, full, world).get() - tt.get()).max() < 1e-13
                assert np.abs(ctx.pairs_unpack(dTall, full, world).get() - dtt.get()).max() < 1e-12 * max(1.0, np.abs(dtt.get()).max())
        ctx.close()


def test_sharded_residual_host_logic(sim):
    sharded_residual_check(sim, [(2, 3, 1), (3, 5, 2)], (1, 2, 3, 8), 1e-12)


def test_sharded_residual_host_logic_bra_dressed(sim, monkeypatch):
    """The same with every rank dressing the bra of its rows of the packed V_abcd (QK then carries no Q_kb part)."""
    monkeypatch.setenv("PYMES_LADDER_DRESS", "1")
    sharded_residual_check(sim, [(2, 3, 1), (3, 5, 2)], (1, 2, 3, 8), 1e-12)


def _problem(tag):
    if tag.startswith("syn_"):
        no, nv = (int(x) for x in tag.split("_")[1:])
        rec = SOLVES[tag]["recipe"]
        f, V, _, _ = synthetic_case(no, nv, seed=rec["seed"], scale=rec["scale"], gap=rec["gap"])
        return no, f, V
    ne, n, ec, eps, h, V = oio.read_fcidump(os.path.join(GOLD, "fcidump", "FCIDUMP." + tag))
    return ne // 2, oio.fock_matrix(ne // 2, h, V), V


@pytest.mark.parametrize("tag,kind", [("LiH.sto6g", "ccsd"), ("H2.321g", "dcsd"), ("syn_4_12", "ccsd"),
                                      ("LiH.bare", "ccd"), ("LiH.sto6g", "dcd")])
def test_solver_classes_reproduce_reference_energies(sim, capsys, tag, kind):
    """The drop-in classes (host logic + DIIS bookkeeping) against the reference's energies."""
    from pymes_amd.solver import ccd, ccsd
    ref = SOLVES[tag][kind]
    no, f, V = _problem(tag)
    if kind in ("ccd", "dcd"):
        s = ccd.CCD(no, delta_e=ref["delta_e"], is_dcd=(kind == "dcd"))
        res = s.solve(f, V)
        e = res["ccd e"]
        assert set(res) == {"ccd e", "t2 amp", "hole e", "particle e", "dE"}
    else:
        s = ccsd.CCSD(no, delta_e=ref["delta_e"], is_dcsd=(kind == "dcsd"))
        res = s.solve(f, V)
        e = res["ccsd e"]
        assert set(res) == {"ccsd e", "t1", "t2", "hole e", "particle e", "dE"}
        assert res["t1"].shape == (f.shape[0] - no, no)
    assert s.iterations == ref["iterations"]
    assert abs(e - ref["e"]) < 1e-9
    assert abs(np.linalg.norm(res["t2"] if "t2" in res else res["t2 amp"]) - ref["t2_norm"]) < 1e-7
    out = capsys.readouterr().out
    assert "Correlation Energy" in out and out.count("Iteration = ") == min(ref["iterations"], 50) + 1


@pytest.mark.parametrize("tag,kind", [("LiH.sto6g", "ccsd"), ("H2.321g", "dcsd"), ("syn_4_12", "ccsd")])
def test_bra_dressed_ladder_reproduces_reference_energies(sim, monkeypatch, tag, kind):
    """PYMES_LADDER_DRESS=1: the T1 dressing of the bra of the pair-packed V_abcd (ccsd.py:414-419) in place of the Q_kb
    products — same iteration history as the reference."""
    from pymes_amd.solver import ccsd
    monkeypatch.setenv("PYMES_LADDER_DRESS", "1")
    ref = SOLVES[tag][kind]
    no, f, V = _problem(tag)
    s = ccsd.CCSD(no, delta_e=ref["delta_e"], is_dcsd=(kind == "dcsd"))
    with contextlib.redirect_stdout(io.StringIO()):
        res = s.solve(f, V)
    assert s.iterations == ref["iterations"]
    assert abs(res["ccsd e"] - ref["e"]) < 1e-9
    assert abs(np.linalg.norm(res["t2"]) - ref["t2_norm"]) < 1e-7


def test_bra_dressing_falls_back_when_its_buffers_do_not_fit(sim, monkeypatch):
    """Out of device memory for the dressed copy of the packed V_abcd (injected in the host simulator): one rank goes on in
    the Q_kb form — same history, nothing leaked."""
    import ctypes as C
    from pymes_amd.solver import ccsd
    monkeypatch.setenv("PYMES_LADDER_DRESS", "1")
    monkeypatch.setenv("PYMES_HOSTSIM_ALLOC_LIMIT", "64")
    ref = SOLVES["syn_4_12"]["ccsd"]
    no, f, V = _problem("syn_4_12")
    n0 = C.c_int64()
    sim.call("pymes_live_allocations", C.byref(n0))
    s = ccsd.CCSD(no, delta_e=ref["delta_e"])
    with contextlib.redirect_stdout(io.StringIO()):
        res = s.solve(f, V)
    assert s.iterations == ref["iterations"] and abs(res["ccsd e"] - ref["e"]) < 1e-9
    n1 = C.c_int64()
    sim.call("pymes_live_allocations", C.byref(n1))
    assert n1.value == n0.value


def test_public_helper_methods_match_oracle(sim, capsys):
    from pymes_amd.integral.partition import part_2_body_int
    from pymes_amd.solver import ccsd
    no, nv = 3, 5
    f, V, t1, t2 = random_case(no, nv, 12, symmetric=False)
    dV = part_2_body_int(no, V)
    assert all(np.shares_memory(v, V) for v in dV.values()) and list(dV) == list(oc.BLOCK_NAMES)
    cc = ccsd.CCSD(no)
    fd = cc.get_T1_dressed_fock(f, t1, dV)
    assert np.abs(fd - oc.dressed_fock(no, f, t1, dV)).max() < 1e-13
    Vd = cc.get_T1_dressed_V(t1, dV)
    ref = oc.dressed_V(t1, dV)
    assert list(Vd) == list(ref)
    for k in ref:
        assert (Vd[k] is None) == (ref[k] is None)
        if ref[k] is not None:
            assert np.abs(Vd[k] - ref[k]).max() < 1e-13
    sub = cc.get_T1_dressed_V(t1, dV, {"abcd": None, "klij": None})
    assert set(sub) == {"abcd", "klij"} and np.abs(sub["abcd"] - ref["abcd"]).max() < 1e-13
    assert np.abs(cc.get_singles_residual(fd, t1, t2, dV) - oc.singles_residual(no, fd, t1, t2, dV)).max() < 1e-13
    assert np.abs(cc.get_doubles_residual(fd, t2, Vd) - oc.ccsd_doubles_residual(no, fd, t2, ref)).max() < 1e-12
    assert np.abs(np.array(cc.get_energy(f[:no, no:], t1, t2, dV["ijab"])) -
                  np.array(oc.ccsd_energy(f[:no, no:], t1, t2, dV["ijab"]))).max() < 1e-13


def test_amps_warm_start_and_kwargs(sim, capsys):
    from pymes_amd.solver import ccsd
    no, f, V = _problem("syn_4_12")
    first = ccsd.CCSD(no, delta_e=1e-6).solve(f, V, maxIter=3, epsilon_e=1)     # unknown kwargs are ignored
    t1, t2 = first["t1"].copy(), first["t2"].copy()
    t2_before = t2.copy()
    res = ccsd.CCSD(no, delta_e=1e-10).solve(f, V, amps=[t1, t2], max_iter=30)
    assert abs(res["ccsd e"] - SOLVES["syn_4_12"]["ccsd"]["e"]) < 5e-8     # a different (warm-started) path
    assert not np.array_equal(t2, t2_before)      # caller's arrays are updated in place (ccsd.py:178-179)
