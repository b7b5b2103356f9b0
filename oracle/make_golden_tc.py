#!/usr/bin/env python3
"""Pin oracle/tc_oracle.py against the imported reference and write tests/golden/tc.json.

    PYTHONPATH=/root/reference:/root/repo python oracle/make_golden_tc.py

Runs the reference's own explicit-3-body pipeline (pymes/test/test_tc_ccsd/test_tc_ccsd.py:17-68: fcidump.read(is_tc)
+ tcdump.read + contraction.get_*_contraction + CCSD on the folded Hamiltonian) on its two fixtures and on a seeded
random 6-index tensor, asserts oracle == reference, and records the reference's outputs.
"""
import contextlib
import io
import json
import os

import numpy as np

from pymes.integral import contraction
from pymes.mean_field import hf
from pymes.solver import ccsd
from pymes.util import fcidump, tcdump

from oracle import tc_oracle as tco

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "..", "tests", "golden")


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


out = {}
for tag, fc, tc in (("H2", "FCIDUMP.H2.tc", "TCDUMP.H2.tc"), ("LiH", "FCIDUMP.LiH.tc", "TCDUMP.LiH_FNO")):
    d = os.path.join(GOLD, "tc")
    n_elec, nb, e_core, e_orb, h, V = quiet(fcidump.read, os.path.join(d, fc), is_tc=True)
    no = n_elec // 2
    L = quiet(tcdump.read, os.path.join(d, tc), sp=0)
    Lo = tco.read_tcdump(os.path.join(d, tc))
    assert np.array_equal(L, Lo), tag
    T0 = quiet(contraction.get_triple_contraction, no, L)
    S = contraction.get_double_contraction(no, L)
    D = contraction.get_single_contraction(no, L)
    assert abs(tco.triple_contraction(no, L) - T0) < 1e-15
    assert np.abs(tco.double_contraction(no, L) - S).max() < 1e-15
    assert np.abs(tco.single_contraction(no, L) - D).max() < 1e-15
    f = hf.construct_hf_matrix(no, h, V) + S
    r = quiet(ccsd.CCSD(no).solve, f, V + D, delta_e=1e-11)
    out[tag] = {"fcidump": fc, "tcdump": tc, "nb": int(nb), "no": int(no), "L_nnz": int(np.count_nonzero(L)),
                "L_abs_sum": float(np.abs(L).sum()), "T0": float(T0), "S": S.tolist(), "D": D.tolist(),
                "e_hf_plus_T0": float(hf.calc_hf_e(no, e_core, h, V) + T0), "ccsd_e": float(r["ccsd e"])}
    print(tag, "T0", T0, "E_ref", out[tag]["e_hf_plus_T0"], "CCSD", out[tag]["ccsd_e"])

# seeded dense tensor without any symmetry: every index position of every term is exercised
rng = np.random.default_rng(11)
nb, no = 5, 2
L = rng.standard_normal((nb,) * 6)
out["random"] = {"seed": 11, "nb": nb, "no": no, "T0": float(quiet(contraction.get_triple_contraction, no, L)),
                 "S": contraction.get_double_contraction(no, L).tolist(),
                 "D": contraction.get_single_contraction(no, L).tolist()}
assert abs(tco.triple_contraction(no, L) - out["random"]["T0"]) < 1e-13
assert np.abs(tco.double_contraction(no, L) - np.array(out["random"]["S"])).max() < 1e-13
assert np.abs(tco.single_contraction(no, L) - np.array(out["random"]["D"])).max() < 1e-13
json.dump(out, open(os.path.join(GOLD, "tc.json"), "w"))
print("wrote tests/golden/tc.json")
