"""Explicit three-body (transcorrelated) path: TCDUMP reader + mean-field foldings + CCSD on the folded
Hamiltonian (pymes/util/tcdump.py, pymes/integral/contraction.py, pymes/test/test_tc_ccsd/test_tc_ccsd.py).
Oracle vs the reference's recorded outputs (CPU), host logic through the host simulator (CPU), HIP kernels (GPU)."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

from oracle import io_oracle as oio, tc_oracle as tco
from pymes_amd import _lib

GOLD = os.path.join(os.path.dirname(__file__), "golden")
G = json.load(open(os.path.join(GOLD, "tc.json")))
MOLS = ("H2", "LiH")


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


@pytest.mark.parametrize("tag", MOLS)
def test_oracle_matches_reference(tag):
    ref = G[tag]
    L = tco.read_tcdump(os.path.join(GOLD, "tc", ref["tcdump"]))
    assert L.shape == (ref["nb"],) * 6 and np.count_nonzero(L) == ref["L_nnz"]
    assert abs(np.abs(L).sum() - ref["L_abs_sum"]) < 1e-14
    no = ref["no"]
    assert abs(tco.triple_contraction(no, L) - ref["T0"]) < 1e-15
    assert np.abs(tco.double_contraction(no, L) - np.array(ref["S"])).max() < 1e-15
    assert np.abs(tco.single_contraction(no, L) - np.array(ref["D"])).max() < 1e-15


def test_oracle_random_tensor():
    ref = G["random"]
    L = np.random.default_rng(ref["seed"]).standard_normal((ref["nb"],) * 6)
    assert abs(tco.triple_contraction(ref["no"], L) - ref["T0"]) < 1e-13
    assert np.abs(tco.double_contraction(ref["no"], L) - np.array(ref["S"])).max() < 1e-13
    assert np.abs(tco.single_contraction(ref["no"], L) - np.array(ref["D"])).max() < 1e-13


def check_product(lib, monkeypatch, solve):
    from pymes_amd.integral import contraction
    from pymes_amd.mean_field import hf
    from pymes_amd.solver import ccsd
    from pymes_amd.util import fcidump, tcdump
    monkeypatch.setattr(_lib, "_default", lib)
    ref = G["random"]
    L = np.random.default_rng(ref["seed"]).standard_normal((ref["nb"],) * 6)
    assert abs(quiet(contraction.get_triple_contraction, ref["no"], L) - ref["T0"]) < 1e-12
    assert np.abs(contraction.get_double_contraction(ref["no"], L) - np.array(ref["S"])).max() < 1e-12
    assert np.abs(contraction.get_single_contraction(ref["no"], L) - np.array(ref["D"])).max() < 1e-12
    assert np.abs(contraction.get_single_contraction(0, L)).max() == 0.0            # no occupied orbitals
    with pytest.raises(ValueError):
        contraction.get_double_contraction(1, L[:, :, :, :, :, :2])
    for tag in MOLS:
        ref = G[tag]
        d = os.path.join(GOLD, "tc")
        L = quiet(tcdump.read, os.path.join(d, ref["tcdump"]), sp=0)
        assert np.array_equal(L, tco.read_tcdump(os.path.join(d, ref["tcdump"])))
        n_elec, nb, e_core, e_orb, h, V = quiet(fcidump.read, os.path.join(d, ref["fcidump"]), is_tc=True)
        no = n_elec // 2
        assert (nb, no) == (ref["nb"], ref["no"])
        T0 = quiet(contraction.get_triple_contraction, no, L)
        S = contraction.get_double_contraction(no, L)
        D = contraction.get_single_contraction(no, L)
        assert abs(T0 - ref["T0"]) < 1e-14
        assert np.abs(S - np.array(ref["S"])).max() < 1e-14 and np.abs(D - np.array(ref["D"])).max() < 1e-14
        assert abs(hf.calc_hf_e(no, e_core, h, V) + T0 - ref["e_hf_plus_T0"]) < 1e-11      # test_tc_ccsd.py:17-38
        if solve:                                                                           # test_tc_ccsd.py:41-63
            f = hf.construct_hf_matrix(no, h, V) + S
            r = quiet(ccsd.CCSD(no).solve, f, V + D, delta_e=1e-11)
            assert abs(r["ccsd e"] - ref["ccsd_e"]) < 1e-9


def test_host_logic(hostsim_lib, monkeypatch):
    check_product(hostsim_lib, monkeypatch, solve=True)


def test_tcdump_edge_cases(hostsim_lib, monkeypatch, tmp_path):
    from pymes_amd.util import tcdump
    monkeypatch.setattr(_lib, "_default", hostsim_lib)
    p = tmp_path / "TCDUMP"
    p.write_text("3\n")                                                    # no integrals at all
    assert np.abs(quiet(tcdump.read, str(p))).max() == 0.0
    # duplicate targets: the later line wins, as in the reference's sequential assignment; a blank line ends the file
    p.write_text("3\n 1.0 1 2 3 1 2 3\n 2.0 2 1 3 2 1 3\n\n 5.0 1 1 1 1 1 1\n")
    L = quiet(tcdump.read, str(p))
    assert np.array_equal(L, tco.read_tcdump(str(p))) and L[0, 0, 1, 1, 2, 2] == -6.0 and L[0, 0, 0, 0, 0, 0] == 0.0
    p.write_text("2\n 1.0 1 2 3 1 1 1\n")
    with pytest.raises(IndexError):
        quiet(tcdump.read, str(p))
    p.write_text("2\n 1.0 1 2 1 1\n")
    with pytest.raises(ValueError):
        quiet(tcdump.read, str(p))


@pytest.mark.gpu
def test_gpu_tc_pipeline(gpu_lib, monkeypatch):
    check_product(gpu_lib, monkeypatch, solve=True)
