#!/usr/bin/env python3
"""Pin oracle/ueg_oracle.py against the reference's UEG integral builder for the correlators other than `trunc`
(pymes/model/ueg.py:740-770 yukawa, 802-834 gaskell_modified, 836-883 gaskell, 885-903 smooth, 905-915 coulomb,
917-935 stg) and write tests/golden/ueg_correlators.npz.  BUILD CONTAINER ONLY:

    PYTHONPATH=/root/reference:/root/repo python oracle/make_golden_ueg_correlators.py

N = 14, rs = 1.0, cutoff = 2 (19 plane waves); for every (correlator, k_cutoff, gamma) case the reference's
eval_2b_integrals in the modes is_only_2b / is_effect_2b / is_rpa_approx plus double / triple contractions of the
three-body operator.  Golden content per case and mode: 4096 sampled entries of V, its sum, sum of absolute values and a
seeded random projection; the mean-field vectors in full.
"""
import contextlib
import io
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle.ueg_oracle import Ueg                     # noqa: E402
from pymes.model import ueg as ref_ueg                # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
# (correlator, k_cutoff, gamma): None = the correlator's own default.  gaskell with its default cut-off 4 k_F^2 puts the
# jump exactly on the lattice shell |n|^2 = 8 for N = 14 (k_F^2 is the shell |n|^2 = 2): scalar and array forms differ there.
CASES = (("gaskell", None, None), ("gaskell", 1.5, 0.8), ("gaskell_modified", 1.0, None), ("coulomb", None, 0.7),
         ("yukawa", None, None), ("yukawa", 1.0, 2.0), ("stg", 1.0, None), ("smooth", None, None))
MODES = (("only_2b", dict(is_only_2b=True)), ("effect_2b", dict(is_effect_2b=True)), ("rpa", dict(is_rpa_approx=True)))


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def main():
    nel, rs, cutoff = 14, 1.0, 2
    rng = np.random.default_rng(2024)
    out = {"cases": np.array(["|".join(map(str, c)) for c in CASES])}
    proj = None
    for ci, (name, kc, gamma) in enumerate(CASES):
        t0 = time.time()
        m = ref_ueg.UEG(nel, nel // 2, nel // 2, rs)
        m.init_single_basis(cutoff)
        m.k_cutoff, m.gamma = kc, gamma
        u = Ueg(nel, rs)
        u.init_basis(cutoff)
        u.k_cutoff, u.gamma, u.correlator = kc, gamma, name
        n_p = u.n_p
        if proj is None:
            proj = rng.standard_normal((n_p,) * 4)
            idx = rng.integers(0, n_p ** 4, size=4096)
            out["sample_idx"] = idx
        for mode, flags in MODES:
            with np.errstate(all="ignore"):
                V = quiet(m.eval_2b_integrals, correlator=getattr(m, name), sp=0, **flags)
                Vo = u.two_body(mode)
            err = np.abs(Vo - V).max()
            assert err < 1e-12 * max(1.0, np.abs(V).max()), (name, kc, gamma, mode, err)
            out[f"c{ci}_{mode}_samples"] = V.reshape(-1)[idx]
            out[f"c{ci}_{mode}_sums"] = np.array([V.sum(), np.abs(V).sum(), (V * proj).sum(), float(np.count_nonzero(V))])
        with np.errstate(all="ignore"):
            d2, e3 = quiet(m.double_contractions_in_3_body), quiet(m.triple_contractions_in_3_body)
            d2o, e3o = u.double_contractions(), u.triple_contractions()
        assert np.abs(d2o - d2).max() < 1e-13 * max(1.0, np.abs(d2).max()) and abs(e3 - e3o) < 1e-13 * max(1.0, abs(e3)), (name, "mean field")
        out[f"c{ci}_double"], out[f"c{ci}_triple"] = d2, np.array([e3])
        out[f"c{ci}_params"] = np.array([np.nan if m.k_cutoff is None else m.k_cutoff, np.nan if m.gamma is None else m.gamma])
        print(f"{name:17s} k_cutoff={kc} gamma={gamma}: oracle == reference in 3 modes + mean field ({time.time() - t0:.0f} s)", flush=True)
    out["proj_seed"] = np.array([2024])
    np.savez_compressed(os.path.join(GOLD, "ueg_correlators.npz"), **out)
    print("written")


if __name__ == "__main__":
    main()
