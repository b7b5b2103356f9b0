#!/usr/bin/env python3
"""Twist-shifted plane-wave basis (pymes/model/ueg.py:128-164, ``init_single_basis(cutoff, k_shift)``): the reference's
basis, kinetic energies and two-body integrals (Coulomb, TC ``is_only_2b``, TC ``is_effect_2b``) and its mean-field 3-body
pieces for N = 14, rs = 1.0, cutoff 2, k_shift = (0.1, 0.25, -0.05) -> tests/golden/ueg_twist.npz.  BUILD CONTAINER ONLY:

    PYTHONPATH=/root/reference:/root/repo python oracle/make_golden_ueg_twist.py

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from pymes.model import ueg as ref_ueg                # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def main():
    shift = [0.1, 0.25, -0.05]
    m = ref_ueg.UEG(14, 7, 7, 1.0)
    m.init_single_basis(2, k_shift=shift)
    m.k_cutoff = 1.0
    out = {"k_shift": np.array(shift), "cutoff": 2, "rs": 1.0, "nel": 14, "k_cutoff": 1.0,
           "k": np.array([b.k for b in m.basis_fns[::2]]), "kinetic": np.array([b.kinetic for b in m.basis_fns[::2]]),
           "coulomb": quiet(m.eval_2b_integrals),
           "only_2b": quiet(m.eval_2b_integrals, correlator=m.trunc, is_only_2b=True, sp=0),
           "effect_2b": quiet(m.eval_2b_integrals, correlator=m.trunc, is_effect_2b=True, sp=0),
           "double_contractions": np.array(quiet(m.double_contractions_in_3_body)),
           "triple_contractions": float(quiet(m.triple_contractions_in_3_body))}
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ueg_twist.npz"), **out)
    print("written ueg_twist.npz:", len(out["kinetic"]), "plane waves")


if __name__ == "__main__":
    main()
