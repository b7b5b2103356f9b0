"""UEG model (BASELINE config 4; pymes/model/ueg.py): oracle vs the reference's golden values (CPU),
host logic through the host simulator (CPU), HIP integral kernels + TC-DCSD solves (GPU)."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

from oracle import cc_oracle as oc
from oracle.ueg_oracle import Ueg
from pymes_amd import _lib

GOLD = os.path.join(os.path.dirname(__file__), "golden")
G = json.load(open(os.path.join(GOLD, "ueg.json")))


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


@pytest.mark.parametrize("cutoff", [2, 3])
def test_oracle_matches_reference(cutoff):
    ref = G[f"tc_N14_rs1.0_c{cutoff}"]
    u = Ueg(14, 1.0)
    assert u.init_basis(cutoff) == ref["n_pw"]
    u.k_cutoff = ref["k_cutoff"]
    gold = np.load(os.path.join(GOLD, f"ueg_tc_c{cutoff}.npz"))
    V = u.two_body("only_2b") + u.two_body("effect_2b")
    assert np.abs(V - gold["V"]).max() < 1e-13
    assert np.count_nonzero(V) == ref["V_nnz"] and abs(np.abs(V - V.transpose(2, 3, 0, 1)).max() - ref["nonhermiticity"]) < 1e-12
    assert np.abs(u.double_contractions() - np.array(ref["double_contractions"])).max() < 1e-14
    assert abs(u.triple_contractions() - ref["e_3b"]) < 1e-14
    assert abs(np.abs(u.two_body("coulomb")).sum() - ref["coulomb_V_abs_sum"]) < 1e-10


def tc_problem(model_cls, nel, rs, cutoff, kc, ctx_kwargs=None):
    """The calling sequence of pymes/test/test_ueg/test_symmetrised_2body_integral.py:39-170."""
    from pymes_amd.mean_field import hf
    m = model_cls(nel, nel // 2, nel // 2, rs)
    m.init_single_basis(cutoff)
    m.k_cutoff = kc
    no, n_p = nel // 2, len(m.basis_fns) // 2
    kin = np.array([m.basis_fns[2 * i].kinetic for i in range(n_p)])
    V = quiet(m.eval_2b_integrals, correlator=m.trunc, is_only_2b=True, sp=0)
    eps_i = hf.calcOccupiedOrbE(kin, V[:no, :no, :no, :no], no)
    eps_a = hf.calcVirtualOrbE(kin, V[no:, :no, no:, :no], V[no:, :no, :no, no:], no, n_p - no)
    f = hf.construct_hf_matrix(no, np.diag(kin), V)
    occ = V[:no, :no, :no, :no]
    e_hf = 2 * eps_i.sum() - (2.0 * np.einsum("jiji->", occ) - np.einsum("ijji->", occ))
    Va = quiet(m.eval_2b_integrals, correlator=m.trunc, is_effect_2b=True, sp=0)
    V = V + 0.5 * (Va + Va.transpose(1, 0, 3, 2))
    d2 = quiet(m.double_contractions_in_3_body)
    e3 = quiet(m.triple_contractions_in_3_body)
    return no, V, f + np.diag(d2), e_hf, d2, e3, eps_i + d2[:no], eps_a + d2[no:]


def check_product(lib, monkeypatch, cutoffs, solve):
    from pymes_amd.model.ueg import UEG
    from pymes_amd.solver import ccd, ccsd, mp2
    monkeypatch.setattr(_lib, "_default", lib)
    for cutoff in cutoffs:
        key = f"tc_N14_rs1.0_c{cutoff}" if cutoff < 5 else "tc_N14_rs0.5_c5"
        if key not in G:
            pytest.skip(f"{key} not in golden file")
        ref = G[key]
        no, V, f, e_hf, d2, e3, eps_i, eps_a = tc_problem(UEG, ref["nel"], ref["rs"], cutoff, ref["k_cutoff"])
        assert V.shape[0] == ref["n_pw"] and np.count_nonzero(V) == ref["V_nnz"]
        assert abs(np.abs(V).sum() - ref["V_abs_sum"]) < 1e-9
        assert abs(e_hf - ref["e_hf"]) < 1e-9 and abs(e3 - ref["e_3b"]) < 1e-12
        assert np.abs(d2 - np.array(ref["double_contractions"])).max() < 1e-12
        gold = os.path.join(GOLD, f"ueg_tc_c{cutoff}.npz")
        if os.path.exists(gold):
            g = np.load(gold)
            assert np.abs(V - g["V"]).max() < 1e-12 and np.abs(f - g["f"]).max() < 1e-12
        if not solve:
            continue
        if "mp2" in ref:
            e_mp2, _ = quiet(mp2.solve, eps_i, eps_a, V[:no, :no, no:, no:], V[no:, no:, :no, :no])
            assert abs(e_mp2 - ref["mp2"]) < 1e-9
        en = ref["energies"]
        if "ccd" in en:
            delta = 1e-8 if cutoff == 5 else 1e-10
            assert abs(quiet(ccd.CCD(no, delta_e=delta).solve, f, V)["ccd e"] - en["ccd"]) < (1e-8 if cutoff == 5 else 1e-9)
        if "dcd" in en:
            assert abs(quiet(ccd.CCD(no, delta_e=1e-10, is_dcd=True).solve, f, V)["ccd e"] - en["dcd"]) < 1e-9
        if "dcsd" in en:       # "transcorrelated DCSD": T1 stays exactly zero by momentum conservation
            r = quiet(ccsd.CCSD(no, delta_e=1e-10, is_dcsd=True).solve, f, V)
            assert abs(r["ccsd e"] - en["dcsd"]) < 1e-9 and np.abs(r["t1"]).max() < 1e-14


def test_product_host_logic(hostsim_lib, monkeypatch):
    check_product(hostsim_lib, monkeypatch, (2,), solve=False)


@pytest.mark.gpu
def test_product_gpu(gpu_lib, monkeypatch):
    check_product(gpu_lib, monkeypatch, (2, 3), solve=True)


@pytest.mark.gpu
def test_product_gpu_57_plane_waves(gpu_lib, monkeypatch):
    """The reference's own driver literals (N=14, rs=0.5, cutoff=5; test_symmetrised_2body_integral.py:205-220)."""
    check_product(gpu_lib, monkeypatch, (5,), solve=True)
