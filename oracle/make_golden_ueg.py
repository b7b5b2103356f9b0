#!/usr/bin/env python3
"""Pin oracle/ueg_oracle.py against the reference's UEG model (pymes/model/ueg.py) and write
tests/golden/ueg_*.  BUILD CONTAINER ONLY:

    PYTHONPATH=/root/reference:/root/repo python oracle/make_golden_ueg.py [--full]

--full also reruns the reference's own N=14, rs=0.5, cutoff=5 TC driver
(test_ueg/test_symmetrised_2body_integral.py:39-222, ~3 minutes) for its known answers.
--c4 runs the same calling sequence at the size BASELINE config 4 is TIMED at (N=14, rs=1.0, cutoff=5 -> 57 plane
waves, k_cutoff of test_ueg/test_ccd_dcd.py:99) followed by the reference's CCSD(no, is_dcsd=True) ("TC-DCSD").
--coulomb57 reruns test_ueg/test_ccd_dcd.py:60-209 (plain Coulomb integrals, 57 plane waves, CCD then DCD warm-started
from the CCD amplitudes, level_shift = -1, max_iter = 60) and checks its two literals (:208-209).
"""
import contextlib
import io
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle.ueg_oracle import Ueg                     # noqa: E402
from oracle import cc_oracle as oc                    # noqa: E402
from pymes.model import ueg as ref_ueg                # noqa: E402
from pymes.mean_field import hf as ref_hf             # noqa: E402
from pymes.solver import ccd as ref_ccd, mp2 as ref_mp2, ccsd as ref_ccsd   # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def ref_model(nel, rs, cutoff, kc):
    m = ref_ueg.UEG(nel, nel // 2, nel // 2, rs)
    m.init_single_basis(cutoff)
    m.k_cutoff = kc
    return m


def tc_problem_ref(nel, rs, cutoff, kc):
    """The calling sequence of test_symmetrised_2body_integral.py:39-170 on the reference."""
    m = ref_model(nel, rs, cutoff, kc)
    no = nel // 2
    n_p = len(m.basis_fns) // 2
    kin = np.array([m.basis_fns[2 * i].kinetic for i in range(n_p)])
    V = quiet(m.eval_2b_integrals, correlator=m.trunc, is_only_2b=True, sp=0)
    eps_i = ref_hf.calcOccupiedOrbE(kin, V[:no, :no, :no, :no], no)
    eps_a = ref_hf.calcVirtualOrbE(kin, V[no:, :no, no:, :no], V[no:, :no, :no, no:], no, n_p - no)
    f = ref_hf.construct_hf_matrix(no, np.diag(kin), V)
    e_hf = 2 * eps_i.sum() - (2.0 * np.einsum("jiji->", V[:no, :no, :no, :no]) - np.einsum("ijji->", V[:no, :no, :no, :no]))
    Va = quiet(m.eval_2b_integrals, correlator=m.trunc, is_effect_2b=True, sp=0)
    V = V + 0.5 * (Va + Va.transpose(1, 0, 3, 2))
    d2 = quiet(m.double_contractions_in_3_body)
    e3 = quiet(m.triple_contractions_in_3_body)
    f = f + np.diag(d2)
    return m, no, kin, V, f, float(e_hf), d2, float(e3), eps_i + d2[:no], eps_a + d2[no:]


def tc_problem_oracle(nel, rs, cutoff, kc):
    u = Ueg(nel, rs)
    u.init_basis(cutoff)
    u.k_cutoff = kc
    no = nel // 2
    from oracle import io_oracle as oio
    V = u.two_body("only_2b")
    f = oio.fock_matrix(no, np.diag(u.kinetic), V)
    occ = V[:no, :no, :no, :no]
    eps_i = u.kinetic[:no] + 2.0 * np.einsum("ijij->i", occ) - np.einsum("ijji->i", occ)
    e_hf = 2 * eps_i.sum() - (2.0 * np.einsum("jiji->", occ) - np.einsum("ijji->", occ))
    V = V + u.two_body("effect_2b")      # already symmetrised once inside (ueg.py:509), the driver's second pass is a no-op
    d2 = u.double_contractions()
    return u, no, V, f + np.diag(d2), float(e_hf), d2, float(u.triple_contractions())


def main():
    out = {}
    for cutoff, kc in ((2, 1.0), (3, 1.0)):
        nel, rs = 14, 1.0
        m, no, kin, V, f, e_hf, d2, e3, eps_i, eps_a = tc_problem_ref(nel, rs, cutoff, kc)
        u, no2, Vo, fo, e_hf_o, d2o, e3o = tc_problem_oracle(nel, rs, cutoff, kc)
        n_p = len(kin)
        assert np.array_equal(u.k_int, np.array([m.basis_fns[2 * i].k for i in range(n_p)]))
        assert np.abs(Vo - V).max() < 1e-13, np.abs(Vo - V).max()
        assert np.abs(fo - f).max() < 1e-13 and abs(e_hf - e_hf_o) < 1e-11
        assert np.abs(d2o - d2).max() < 1e-14 and abs(e3 - e3o) < 1e-14
        # plain Coulomb integrals too
        m2 = ref_model(nel, rs, cutoff, kc)
        Vc = quiet(m2.eval_2b_integrals, sp=0)
        assert np.abs(u.two_body("coulomb") - Vc).max() < 1e-14
        res = {}
        for kind, solver in (("ccd", ref_ccd.CCD(no, delta_e=1e-10)), ("dcd", ref_ccd.CCD(no, delta_e=1e-10, is_dcd=True)),
                             ("dcsd", ref_ccsd.CCSD(no, delta_e=1e-10, is_dcsd=True))):
            r = quiet(solver.solve, f, V)
            res[kind] = float(r["ccd e"] if "ccd e" in r else r["ccsd e"])
            if kind == "dcsd":
                assert np.abs(r["t1"]).max() == 0.0      # momentum conservation: T1 = 0 exactly
        o = oc.ccd_solve(no, fo, Vo, is_dcd=True, delta_e=1e-10)
        assert abs(o["e"] - res["dcd"]) < 1e-10
        out[f"tc_N{nel}_rs{rs}_c{cutoff}"] = {
            "nel": nel, "rs": rs, "cutoff": cutoff, "k_cutoff": kc, "n_pw": n_p, "e_hf": e_hf, "e_3b": e3,
            "double_contractions": d2.tolist(), "V_sum": float(V.sum()), "V_abs_sum": float(np.abs(V).sum()),
            "V_nnz": int(np.count_nonzero(V)), "nonhermiticity": float(np.abs(V - V.transpose(2, 3, 0, 1)).max()),
            "coulomb_V_abs_sum": float(np.abs(Vc).sum()), "energies": res}
        np.savez_compressed(os.path.join(GOLD, f"ueg_tc_c{cutoff}.npz"), V=V, f=f)
        print(f"UEG N=14 rs=1.0 cutoff={cutoff}: {n_p} PW, E_HF={e_hf:.10f} E_3b={e3:.10f} {res}  oracle == reference")
    if "--full" in sys.argv:
        # the reference's own driver literals (test_symmetrised_2body_integral.py:205-220): rs=0.5, cutoff=5
        m, no, kin, V, f, e_hf, d2, e3, eps_i, eps_a = tc_problem_ref(14, 0.5, 5, 1.0)
        assert abs(e_hf - 58.143779330795965) < 1e-8 and abs(e3 - 0.07218268772824925) < 1e-8
        e_mp2, _ = quiet(ref_mp2.solve, eps_i, eps_a, t_V_abij=V[no:, no:, :no, :no], t_V_ijab=V[:no, :no, no:, no:])
        r = quiet(ref_ccd.CCD(no).solve, f, V)
        assert abs(e_mp2 - -0.327226965969) < 1e-8 and abs(r["ccd e"] - -0.256670836708) < 1e-8
        out["tc_N14_rs0.5_c5"] = {"nel": 14, "rs": 0.5, "cutoff": 5, "k_cutoff": 1.0, "n_pw": len(kin), "e_hf": e_hf,
                                  "e_3b": e3, "double_contractions": d2.tolist(), "mp2": float(e_mp2),
                                  "energies": {"ccd": float(r["ccd e"])}, "V_abs_sum": float(np.abs(V).sum()),
                                  "V_nnz": int(np.count_nonzero(V))}
        print("UEG N=14 rs=0.5 cutoff=5 (57 PW): reference literals reproduced", e_mp2, r["ccd e"])
    if "--c4" in sys.argv:
        nel, rs, cutoff = 14, 1.0, 5
        m0 = ref_model(nel, rs, cutoff, None)
        kc = m0.L / (2 * np.pi) * 2.3225029893472993 / rs                      # test_ccd_dcd.py:99
        m, no, kin, V, f, e_hf, d2, e3, eps_i, eps_a = tc_problem_ref(nel, rs, cutoff, kc)
        e_mp2, _ = quiet(ref_mp2.solve, eps_i, eps_a, t_V_abij=V[no:, no:, :no, :no], t_V_ijab=V[:no, :no, no:, no:])
        s = ref_ccsd.CCSD(no, delta_e=1e-10, is_dcsd=True)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            r = s.solve(f, V)
        import re
        hist = [float(x) for x in re.findall(r"Correlation Energy = (-?[0-9.eE+-]+)", buf.getvalue())]
        assert np.abs(r["t1"]).max() == 0.0
        rd = quiet(ref_ccd.CCD(no, delta_e=1e-10, is_dcd=True).solve, f, V)
        out["tc_N14_rs1.0_c5"] = {"nel": nel, "rs": rs, "cutoff": cutoff, "k_cutoff": float(kc), "n_pw": len(kin),
                                  "e_hf": e_hf, "e_3b": e3, "double_contractions": d2.tolist(), "mp2": float(e_mp2),
                                  "energies": {"dcsd": float(r["ccsd e"]), "dcd": float(rd["ccd e"])},
                                  "dcsd_history": hist, "t2_norm": float(np.linalg.norm(r["t2"])),
                                  "V_abs_sum": float(np.abs(V).sum()), "V_nnz": int(np.count_nonzero(V)),
                                  "nonhermiticity": float(np.abs(V - V.transpose(2, 3, 0, 1)).max())}
        print("UEG N=14 rs=1.0 cutoff=5 (57 PW, the size config 4 is timed at): TC-DCSD", r["ccsd e"], len(hist), "iterations")
    if "--coulomb57" in sys.argv:
        nel, rs, cutoff = 14, 0.5, 5
        m = ref_model(nel, rs, cutoff, None)
        m.k_cutoff = m.L / (2 * np.pi) * 2.3225029893472993 / rs               # :99 (unused by the Coulomb integrals)
        no = nel // 2
        n_p = len(m.basis_fns) // 2
        kin = np.array([m.basis_fns[2 * i].kinetic for i in range(n_p)])
        V = quiet(m.eval_2b_integrals, sp=1)                                     # :107
        eps_i = ref_hf.calcOccupiedOrbE(kin, V[:no, :no, :no, :no], no)
        eps_a = ref_hf.calcVirtualOrbE(kin, V[no:, :no, no:, :no], V[no:, :no, :no, no:], no, n_p - no)
        e_mp2, _ = quiet(ref_mp2.solve, eps_i, eps_a, V[:no, :no, no:, no:], V[no:, no:, :no, :no])
        f = ref_hf.construct_hf_matrix(no, np.diag(kin), V)
        rc = quiet(ref_ccd.CCD(no, is_diis=True).solve, f, V, level_shift=-1., sp=0, max_iter=60)       # :174-178
        amp = rc["t2 amp"].copy()
        rdd = quiet(ref_ccd.CCD(no, is_dcd=True, is_diis=True).solve, f, V, level_shift=-1., sp=0, max_iter=60, amps=amp)
        assert abs(rc["ccd e"] - -0.5120153512190824) < 1e-6 and abs(rdd["ccd e"] - -0.515296499349519) < 1e-6   # :208-209
        out["coulomb_N14_rs0.5_c5"] = {"nel": nel, "rs": rs, "cutoff": cutoff, "n_pw": n_p, "level_shift": -1.0, "max_iter": 60,
                                       "mp2": float(e_mp2), "V_abs_sum": float(np.abs(V).sum()),
                                       "energies": {"ccd": float(rc["ccd e"]), "dcd_from_ccd_amps": float(rdd["ccd e"])}}
        print("Coulomb UEG 57 PW, level_shift -1: CCD", rc["ccd e"], "DCD", rdd["ccd e"], "(reference literals reproduced)")
    path = os.path.join(GOLD, "ueg.json")
    if os.path.exists(path):
        old = json.load(open(path))
        for k, v in old.items():
            out.setdefault(k, v)
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print("written")


if __name__ == "__main__":
    main()
