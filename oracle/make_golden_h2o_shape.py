#!/usr/bin/env python3
"""BASELINE config 1 ("H2O cc-pVDZ CCSD from FCIDUMP") as far as it can be had here: the reference ships no H2O FCIDUMP
and this image has no integral program ("Config 1 caveat"), so the PLUMBING of that config — text FCIDUMP ->
`fcidump.read` -> `construct_hf_matrix` -> `CCSD.solve` — is pinned on a file of H2O/cc-pVDZ's SHAPE: 24 orbitals, 10
electrons, (nocc, nvirt) = (5, 19), integrals from the synthetic recipe of SURVEY 8(d), written by the product's FCIDUMP
writer, read back and solved by the REFERENCE.

Run in the BUILD CONTAINER ONLY (the reference does not travel):

    PYTHONPATH=/root/reference:/root/repo python oracle/make_golden_h2o_shape.py

Writes tests/golden/fcidump/FCIDUMP.syn_5_19.gz (the data file; lines with p <= r and q <= s only, the images the
reference's reader restores, pymes/util/fcidump.py:143-146) and tests/golden/h2o_shape.json (what the reference made of it:
header values, checksums of h and V, E_HF, the Fock diagonal, CCSD and DCSD energies with their iteration histories).
The oracle is checked against the reference on the way.
"""
import contextlib
import gzip
import io
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle import cc_oracle as oc          # noqa: E402
from oracle import io_oracle as oio         # noqa: E402
from oracle.cases import synthetic_case     # noqa: E402
from pymes_amd.util import fcidump as my_fcidump        # noqa: E402  (host-side writer: pure Python)

from pymes.util import fcidump as ref_fcidump           # noqa: E402
from pymes.mean_field import hf as ref_hf                # noqa: E402
from pymes.solver import ccsd as ref_ccsd                # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
NO, NV, SEED, SCALE, E_CORE = 5, 19, 3, 0.3, 9.18953


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **k)
    return out, buf.getvalue()


def history(text):           # the energies the reference logs, one per pass (ccsd.py:199)
    return [float(x) for x in re.findall(r"Correlation Energy = (-?[0-9.eE+-]+)", text)]


def main():
    f, V, _, eps = synthetic_case(NO, NV, seed=SEED, scale=SCALE)
    n = NO + NV
    # h such that the HF matrix of (h, V) is f: h = f - (2 J - K)
    J = np.einsum("piqi->pq", V[:, :NO, :, :NO])
    K = np.einsum("piiq->pq", V[:, :NO, :NO, :])
    h = f - (2.0 * J - K)
    h = 0.5 * (h + h.T)
    p, q, r, s = np.indices(V.shape, sparse=True)
    Vw = np.where((p <= r) & (q <= s), V, 0.0)            # one representative per image group of the reference's reader
    tmp = os.path.join(GOLD, "fcidump", "FCIDUMP.syn_5_19")
    my_fcidump.write(Vw, h, NO, e_nuc=E_CORE, file=tmp)
    (ne, norb, ec, e1, h_r, V_r), _ = quiet(ref_fcidump.read, tmp)
    assert ne == 2 * NO and norb == n and ec == E_CORE
    assert np.array_equal(V_r, V), np.abs(V_r - V).max()            # %.17g round-trips; the four images restore all of V
    assert np.array_equal(h_r, h)
    mine = oio.read_fcidump(tmp)
    assert np.array_equal(mine[5], V_r) and np.array_equal(mine[4], h_r)
    e_hf = ref_hf.calc_hf_e(NO, ec, h_r, V_r)
    fm = ref_hf.construct_hf_matrix(NO, h_r, V_r)
    assert np.abs(fm - f).max() < 1e-12
    out = {"recipe": {"nocc": NO, "nvirt": NV, "seed": SEED, "scale": SCALE, "gap": 3.0, "e_core": E_CORE},
           "n_elec": ne, "n_orb": norb, "e_core": ec, "e_hf": float(e_hf),
           "V_sum": float(V_r.sum()), "V_abs_sum": float(np.abs(V_r).sum()), "V_nnz": int(np.count_nonzero(V_r)),
           "h_sum": float(h_r.sum()), "h_abs_sum": float(np.abs(h_r).sum()), "fock_diag": fm.diagonal().tolist(),
           "fock_offdiag_max": float(np.max(np.abs(fm - np.diag(fm.diagonal()))))}
    for kind in ("ccsd", "dcsd"):
        s = ref_ccsd.CCSD(NO, delta_e=1e-10, is_dcsd=(kind == "dcsd"))
        res, log = quiet(s.solve, fm, V_r)
        o = oc.ccsd_solve(NO, fm, V_r, delta_e=1e-10, is_dcsd=(kind == "dcsd"))
        assert abs(o["e"] - res["ccsd e"]) < 1e-11, (o["e"], res["ccsd e"])
        assert np.abs(o["t2"] - res["t2"]).max() < 1e-10
        hist = history(log)
        assert len(hist) == o["iterations"] and np.abs(np.array(hist) - np.array([x[0] for x in o["history"]])).max() < 1e-10
        out[kind] = {"e": float(res["ccsd e"]), "dE": float(res["dE"]), "delta_e": 1e-10, "iterations": int(o["iterations"]),
                     "t1_norm": float(np.linalg.norm(res["t1"])), "t2_norm": float(np.linalg.norm(res["t2"])),
                     "history": hist}
        print(f"{kind}: E = {res['ccsd e']:.12f}  ({o['iterations']} iterations), |E_oracle - E_ref| = {abs(o['e'] - res['ccsd e']):.1e}")
    with open(tmp, "rb") as fh:
        data = fh.read()
    with gzip.GzipFile(tmp + ".gz", "wb", mtime=0) as gz:
        gz.write(data)
    os.remove(tmp)
    out["file"] = {"name": "fcidump/FCIDUMP.syn_5_19.gz", "bytes": len(data), "lines": data.count(b"\n")}
    with open(os.path.join(GOLD, "h2o_shape.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("E_HF = %.12f; %d lines, %d bytes (%d gzipped)" % (e_hf, out["file"]["lines"], len(data), os.path.getsize(tmp + ".gz")))


if __name__ == "__main__":
    main()
