"""FEAST-EOM-CCSD (pymes/solver/feast_eom_ccsd.py:72-181, :293-350): oracle vs the reference run recorded by
oracle/make_golden_feast.py (CPU), the product through the host simulator (CPU) and on the HIP path (GPU).

What is compared (see the module docstring of pymes_amd/solver/feast_eom_ccsd.py): the Ritz values of the first pass, and
in the last pass the values inside the window that have settled — the reference's linear solves stop at a relative residual
of 1e-4, so an unconverged Ritz value moves by ~1e-5 when one inner iteration count flips by rounding."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

from oracle import cc_oracle as oc, feast_oracle as fo, io_oracle as oio
from pymes_amd import _lib

GOLD = os.path.join(os.path.dirname(__file__), "golden")
G = json.load(open(os.path.join(GOLD, "feast.json")))


def ground_state(tag):
    ne, n, ec, eps, h, V = oio.read_fcidump(os.path.join(GOLD, "fcidump", "FCIDUMP." + tag))
    no = ne // 2
    f = oio.fock_matrix(no, h, V)
    r = oc.ccsd_solve(no, f, V, delta_e=1e-12, max_iter=200)
    Vb = oc.split_blocks(no, V)
    return no, oc.dressed_fock(no, f, r["t1"], Vb), oc.dressed_V(r["t1"], Vb), r["t2"]


def cplx(rows):
    return np.array([complex(a, b) for a, b in rows])


def compare(ref, history, first_tol, settled_tol):
    assert len(history) == len(ref["history"])
    first = np.abs(np.sort_complex(cplx(ref["history"][0])) - np.sort_complex(np.asarray(history[0]))).max()
    assert first < first_tol, (first, history[0])
    for x in cplx(ref["settled_in_window"]):
        assert min(abs(x - y) for y in history[-1]) < settled_tol, (x, history[-1])


def test_oracle_matches_reference():
    ref = G["LiH.sto6g|seed7"]
    no, fd, Vd, t2 = ground_state(ref["tag"])
    np.random.seed(ref["seed"])
    o = fo.feast_solve(no, fd, Vd, t2, e_c=ref["e_c"], e_r=ref["e_r"], n_trial=ref["n_trial"], max_iter=ref["max_iter"])
    compare(ref, o["history"], 1e-8, 1e-8)
    # the settled value is an excitation energy of H̄: the Davidson driver's second root (tests/golden/eom_solves.json)
    ee = json.load(open(os.path.join(GOLD, "eom_solves.json")))["LiH.sto6g"]["ee"]
    assert min(abs(x - ee[1]) for x in cplx(ref["settled_in_window"])) < 1e-7


def run_product(lib, monkeypatch, keys):
    from pymes_amd.solver.feast_eom_ccsd import FEAST_EOM_CCSD
    monkeypatch.setattr(_lib, "_default", lib)
    for key in keys:
        ref = G[key]
        no, fd, Vd, t2 = ground_state(ref["tag"])
        s = FEAST_EOM_CCSD(no, e_c=ref["e_c"], e_r=ref["e_r"], n_trial=ref["n_trial"], max_iter=ref["max_iter"])
        np.random.seed(ref["seed"])
        with contextlib.redirect_stdout(io.StringIO()):
            ev = s.solve(fd, Vd, t2)
        assert s.iterations == ref["iterations"] and np.array_equal(ev, s.history[-1])
        compare(ref, s.history, 1e-7, 1e-8)
        assert all(info == 0 for info, _ in s.linear_solver_info)          # every linear solve met its tolerance
    return s


def test_product_host_logic(hostsim_lib, monkeypatch):
    run_product(hostsim_lib, monkeypatch, ["LiH.sto6g|seed7"])


def test_linear_solvers_host_forms(hostsim_lib, monkeypatch):
    """_gcrotmk / _jacobi with the reference's host-array signatures (:252-350): both must solve (z - H̄) Q = u_l."""
    from oracle import eom_oracle as eo
    from pymes_amd.solver.feast_eom_ccsd import FEAST_EOM_CCSD
    monkeypatch.setattr(_lib, "_default", hostsim_lib)
    no, fd, Vd, t2 = ground_state("LiH.sto6g")
    nv = fd.shape[0] - no
    s = FEAST_EOM_CCSD(no, e_c=0.15, e_r=0.04)
    rng = np.random.default_rng(5)
    s.u_singles, s.u_doubles = [rng.standard_normal((nv, no))], [rng.standard_normal((nv, nv, no, no)) * 0.01]
    d1, d2 = s.get_diag_singles(fd, Vd, t2), s.get_diag_doubles(fd, Vd, t2)
    ze = 0.15 + 0.04j
    q1, q2 = s._gcrotmk(0, ze, d1, d2, fd, Vd, t2)
    r1 = ze * q1 - eo.sigma_singles(no, fd, Vd, q1, q2, t2) - s.u_singles[0]
    r2 = ze * q2 - eo.sigma_doubles(no, fd, Vd, q1, q2, t2) - s.u_doubles[0]
    bnorm = np.sqrt(np.vdot(s.u_singles[0], s.u_singles[0]) + np.vdot(s.u_doubles[0], s.u_doubles[0])).real
    assert np.sqrt(np.vdot(r1, r1) + np.vdot(r2, r2)).real <= 1.0001e-4 * bnorm
    # the oracle's call of scipy's gcrotmk on the same system: same solution up to the stopping tolerance
    o1, o2, info = fo.linear_solve(no, fd, Vd, t2, ze, d1, d2, s.u_singles[0], s.u_doubles[0])
    assert info == 0 and np.abs(o1 - q1).max() < 1e-7 and np.abs(o2 - q2).max() < 1e-7
    # get_residual is what _jacobi iterates on (:183-218)
    g1, g2 = s.get_residual(0, ze, q1, q2, fd, Vd, t2)
    assert np.abs(g1 + r1).max() < 1e-10 and np.abs(g2 + r2).max() < 1e-10
    # _jacobi (:252-291): 200 damped sweeps, restated here with the oracle's sigma
    j1, j2 = s._jacobi(0, ze, d1, d2, fd, Vd, t2)
    p1, p2 = np.zeros((nv, no), dtype=complex), np.zeros((nv, nv, no, no), dtype=complex)
    for _ in range(200):
        e1 = s.u_singles[0] - ze * p1 + eo.sigma_singles(no, fd, Vd, p1, p2, t2)
        e2 = s.u_doubles[0] - ze * p2 + eo.sigma_doubles(no, fd, Vd, p1, p2, t2)
        p1 = p1 + 0.01 * e1 / (ze - d1 + 0.01)
        p2 = p2 + 0.01 * e2 / (ze - d2 + 0.01)
    assert np.abs(j1 - p1).max() < 1e-11 and np.abs(j2 - p2).max() < 1e-11


def check_device_form(lib, monkeypatch):
    """FEAST_EOM_CCSD.solve on the device-resident hand-over of a CCSD solve (DeviceIntegrals -> device amplitudes ->
    DressedDeviceIntegrals): same Ritz values as the host-dictionary form, diagonals from strided device gathers."""
    from pymes_amd.integral.device import DeviceIntegrals
    from pymes_amd.solver.ccsd import CCSD
    from pymes_amd.solver.feast_eom_ccsd import FEAST_EOM_CCSD
    monkeypatch.setattr(_lib, "_default", lib)
    ref = G["LiH.sto6g|seed7"]
    ne, n, ec, eps, h, V = oio.read_fcidump(os.path.join(GOLD, "fcidump", "FCIDUMP." + ref["tag"]))
    no = ne // 2
    f = oio.fock_matrix(no, h, V)
    ints = DeviceIntegrals.from_V_pqrs(no, V)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            cc = CCSD(no, delta_e=1e-12)
            res = cc.solve(f, ints, max_iter=200, device_amplitudes=True)
            fd = cc.get_T1_dressed_fock(f, res["t1"], ints)
            Vd = cc.get_T1_dressed_V(res["t1"], ints)
            s = FEAST_EOM_CCSD(no, e_c=ref["e_c"], e_r=ref["e_r"], n_trial=ref["n_trial"], max_iter=ref["max_iter"])
            host = Vd.to_host()
            t2h = res["t2"].get()
            assert np.abs(s.get_diag_singles(fd, Vd, res["t2"]) - s.get_diag_singles(fd, host, t2h)).max() < 1e-13
            assert np.abs(s.get_diag_doubles(fd, Vd, res["t2"]) - s.get_diag_doubles(fd, host, t2h)).max() < 1e-13
            np.random.seed(ref["seed"])
            s.solve(fd, Vd, res["t2"])
        compare(ref, s.history, 1e-7, 1e-8)
        assert s.u_doubles[0].ctx is ints.ctx and s.Q_doubles[0].ctx is ints.ctx
    finally:
        ints.ctx.close()


def test_device_form_host_logic(hostsim_lib, monkeypatch):
    check_device_form(hostsim_lib, monkeypatch)


@pytest.mark.gpu
def test_device_form_gpu(gpu_lib, monkeypatch):
    check_device_form(gpu_lib, monkeypatch)


def check_real_time_hooks(lib, monkeypatch, tol):
    """get_residual / _gcrotmk / _jacobi with ``is_rt=True, dt, phase`` (feast_eom_ccsd.py:197-214, :276-278, :321-334) against
    the outputs of the reference's own methods on the same seeded inputs (oracle/make_golden_feast.py rt)."""
    from pymes_amd.solver.feast_eom_ccsd import FEAST_EOM_CCSD
    monkeypatch.setattr(_lib, "_default", lib)
    g = np.load(os.path.join(GOLD, "feast_rt.npz"))
    no, fd, Vd, t2 = ground_state("LiH.sto6g")
    nv = fd.shape[0] - no
    s = FEAST_EOM_CCSD(no, e_c=0.15, e_r=0.04)
    rng = np.random.default_rng(int(g["seed"]))
    s.u_singles = [rng.standard_normal((nv, no))]
    s.u_doubles = [rng.standard_normal((nv, nv, no, no)) * 0.05]
    q1 = rng.standard_normal((nv, no)) + 1j * rng.standard_normal((nv, no))
    q2 = 0.05 * (rng.standard_normal((nv, nv, no, no)) + 1j * rng.standard_normal((nv, nv, no, no)))
    d1, d2 = s.get_diag_singles(fd, Vd, t2), s.get_diag_doubles(fd, Vd, t2)
    ze, dt, phase = complex(g["ze"]), float(g["dt"]), complex(g["phase"])
    g1, g2 = s.get_residual(0, ze, q1, q2, fd, Vd, t2, phase=phase, is_rt=True, dt=dt)
    assert np.abs(g1 - g["g1"]).max() < 1e-11 and np.abs(g2 - g["g2"]).max() < 1e-11
    with contextlib.redirect_stdout(io.StringIO()):
        k1, k2 = s._gcrotmk(0, ze, d1, d2, fd, Vd, t2, phase=phase, is_rt=True, dt=dt)
        j1, j2 = s._jacobi(0, ze, d1, d2, fd, Vd, t2, phase=phase, is_rt=True, dt=dt)
    assert np.abs(k1 - g["k1"]).max() < tol and np.abs(k2 - g["k2"]).max() < tol
    assert np.abs(j1 - g["j1"]).max() < 1e-10 and np.abs(j2 - g["j2"]).max() < 1e-10
    # without dt the switch is off (the reference tests `is_rt and dt is not None`)
    h1, _ = s.get_residual(0, ze, q1, q2, fd, Vd, t2, phase=phase, is_rt=True, dt=None)
    f1, _ = s.get_residual(0, ze, q1, q2, fd, Vd, t2, phase=phase)
    assert np.array_equal(h1, f1)


def test_real_time_hooks_host_logic(hostsim_lib, monkeypatch):
    check_real_time_hooks(hostsim_lib, monkeypatch, 1e-7)


@pytest.mark.gpu
def test_real_time_hooks_gpu(gpu_lib, monkeypatch):
    check_real_time_hooks(gpu_lib, monkeypatch, 1e-7)


@pytest.mark.gpu
def test_product_gpu(gpu_lib, monkeypatch):
    run_product(gpu_lib, monkeypatch, list(G))


SYN_PATH = os.path.join(GOLD, "feast_synthetic.json")       # oracle/make_golden_feast.py syn
SYN = json.load(open(SYN_PATH)) if os.path.exists(SYN_PATH) else {}


def check_synthetic_chain(lib, monkeypatch, tag, first_tol, settled_tol):
    """Beyond a toy molecule: CCSD.solve -> get_T1_dressed_* -> FEAST_EOM_CCSD.solve, device-resident, on the synthetic problems
    of the Davidson golden, against the reference's own run of the same chain (tests/golden/feast_synthetic.json): Ritz values
    of the first pass, the settled values of the last, and those against the Davidson roots inside the window."""
    from oracle.cases import eom_davidson_case
    from pymes_amd.integral.device import DeviceIntegrals
    from pymes_amd.solver.ccsd import CCSD
    from pymes_amd.solver.feast_eom_ccsd import FEAST_EOM_CCSD
    monkeypatch.setattr(_lib, "_default", lib)
    ref = SYN[tag]
    no, nv = ref["no"], ref["nv"]
    f, V = eom_davidson_case(no, nv, seed=0, scale=ref["scale"])
    ints = DeviceIntegrals.from_V_pqrs(no, V)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            cc = CCSD(no, delta_e=1e-11)
            res = cc.solve(f, ints, max_iter=100, device_amplitudes=True)
            fd = cc.get_T1_dressed_fock(f, res["t1"], ints)
            Vd = cc.get_T1_dressed_V(res["t1"], ints)
            s = FEAST_EOM_CCSD(no, e_c=ref["e_c"], e_r=ref["e_r"], n_trial=ref["n_trial"], max_iter=ref["max_iter"])
            np.random.seed(ref["seed"])
            s.solve(fd, Vd, res["t2"])
        assert abs(res["ccsd e"] - ref["ccsd_e"]) < 1e-9
        assert len(ref["settled_in_window"]) == len(ref["davidson_roots_in_window"]) == 2
        compare(ref, s.history, first_tol, settled_tol)
        for e in ref["davidson_roots_in_window"]:
            assert min(abs(x - e) for x in s.history[-1]) < 2e-7, (e, s.history[-1])
        assert all(info == 0 for info, _ in s.linear_solver_info)
    finally:
        ints.ctx.close()
    return s


def test_synthetic_chain_host_logic(hostsim_lib, monkeypatch):
    check_synthetic_chain(hostsim_lib, monkeypatch, "small", 1e-7, 1e-7)


@pytest.mark.gpu
def test_synthetic_chain_gpu(gpu_lib, monkeypatch):
    """(12,48): LDS-DMA GEMMs, batched pair-packed ladders and the stacked complex sigma (24 right-hand sides per quadrature
    sweep) inside a FEAST solve that is compared with the reference."""
    check_synthetic_chain(gpu_lib, monkeypatch, "small", 1e-7, 1e-7)
    if "big" in SYN:
        check_synthetic_chain(gpu_lib, monkeypatch, "big", 1e-7, 1e-7)
