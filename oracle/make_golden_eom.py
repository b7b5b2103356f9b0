#!/usr/bin/env python3
"""Pin oracle/eom_oracle.py against the reference's EOM-CCSD (pymes/solver/eom_ccsd.py) and
write tests/golden/eom_*.  BUILD CONTAINER ONLY:

    PYTHONPATH=/root/reference:/root/repo python oracle/make_golden_eom.py
"""
import contextlib
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle import cc_oracle as oc, eom_oracle as eo, io_oracle as oio      # noqa: E402
from oracle.cases import random_case                                          # noqa: E402
from pymes.solver import ccsd as ref_ccsd, eom_ccsd as ref_eom               # noqa: E402
from pymes.util import fcidump as ref_fcidump                                 # noqa: E402
from pymes.mean_field import hf as ref_hf                                     # noqa: E402
from pymes.integral.partition import part_2_body_int                          # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def sigma_case(no, nv, seed):
    f, V, t1, t2 = random_case(no, nv, seed, symmetric=False)
    rng = np.random.default_rng(seed + 100)
    u1 = rng.standard_normal((nv, no))
    u2 = rng.standard_normal((nv, nv, no, no))
    Vb = part_2_body_int(no, V)          # every block "dressed" = random here: all 16 are exercised
    e = ref_eom.EOM_CCSD(no, n_excit=2)
    s1 = e.update_singles(f, Vb, u1, u2, t2)
    s2 = e.update_doubles(f, Vb, u1, u2, t2)
    o1 = eo.sigma_singles(no, f, Vb, u1, u2, t2)
    o2 = eo.sigma_doubles(no, f, Vb, u1, u2, t2)
    assert np.abs(o1 - s1).max() < 1e-12 and np.abs(o2 - s2).max() < 1e-12
    # diagonals used as preconditioner by the FEAST / real-time callers (eom_ccsd.py:169-266)
    d1 = e.get_diag_singles(f, Vb, t2)
    d2 = e.get_diag_doubles(f, Vb, t2)
    assert np.abs(eo.diag_singles(no, f, Vb, t2) - d1).max() < 1e-12
    assert np.abs(eo.diag_doubles(no, f, Vb, t2) - d2).max() < 1e-12
    # complex trial vectors (feast_eom_ccsd.py:309-350 calls update_* with complex u)
    w1 = rng.standard_normal((nv, no))
    w2 = rng.standard_normal((nv, nv, no, no))
    c1 = e.update_singles(f, Vb, u1 + 1j * w1, u2 + 1j * w2, t2)
    c2 = e.update_doubles(f, Vb, u1 + 1j * w1, u2 + 1j * w2, t2)
    assert np.abs(eo.sigma_singles(no, f, Vb, u1 + 1j * w1, u2 + 1j * w2, t2) - c1).max() < 1e-12
    assert np.abs(eo.sigma_doubles(no, f, Vb, u1 + 1j * w1, u2 + 1j * w2, t2) - c2).max() < 1e-12
    np.savez_compressed(os.path.join(GOLD, f"eom_sigma_{no}_{nv}.npz"), seed=seed, s1=s1, s2=s2, d1=d1, d2=d2,
                        c1=c1, c2=c2)
    print(f"sigma ({no},{nv}): oracle == reference")


def solve_case(tag, n_excit):
    path = os.path.join(GOLD, "fcidump", "FCIDUMP." + tag)
    ne, n, ec, eps, h, V = quiet(ref_fcidump.read, path)
    no = ne // 2
    f = ref_hf.construct_hf_matrix(no, h, V)
    cc = ref_ccsd.CCSD(no)
    cc.delta_e = 1e-12
    res = quiet(cc.solve, f, V, max_iter=200)
    t1, t2 = res["t1"].copy(), res["t2"].copy()
    Vb = part_2_body_int(no, V)
    fd = cc.get_T1_dressed_fock(f, t1, Vb)
    Vd = cc.get_T1_dressed_V(t1, Vb)
    e = ref_eom.EOM_CCSD(no, n_excit=n_excit)
    e.max_iter = 1000
    ee = quiet(e.solve, fd, Vd, t2)
    mine = eo.eom_solve(no, fd, Vd, t2, n_excit=n_excit, max_iter=1000)
    assert np.abs(np.asarray(mine["e"]) - np.asarray(ee)).max() < 1e-8, (mine["e"], ee)
    print(f"eom {tag}: EE = {ee}  ({mine['iterations']} it) oracle == reference")
    return {"n_excit": n_excit, "ee": [float(x) for x in ee], "ccsd_e": float(res["ccsd e"]),
            "iterations": mine["iterations"]}


def main():
    sigma_case(2, 3, 31)
    sigma_case(3, 5, 32)
    out = {"LiH.321g": solve_case("LiH.321g", 2), "LiH.sto6g": solve_case("LiH.sto6g", 2),
           "H2.ccpvdz": solve_case("H2.ccpvdz", 2)}
    # the literal of the reference's own test (test_eom_ccsd.py:9)
    assert np.allclose(out["LiH.321g"]["ee"], [0.1180867117168979, 0.154376205595602])
    with open(os.path.join(GOLD, "eom_solves.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("written")


if __name__ == "__main__":
    main()
