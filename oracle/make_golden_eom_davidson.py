#!/usr/bin/env python3
"""A reference Davidson run larger than a toy molecule (missing #4).  BUILD CONTAINER ONLY:

    PYTHONPATH=/root/reference:/root/repo python oracle/make_golden_eom_davidson.py [small] [big] [--oracle]

Runs the reference's own calling chain (pymes/test/test_eom_ccsd/test_eom_ccsd.py:24-48) on a synthetic closed-shell
problem (oracle/cases.py::eom_davidson_case: the integrals of SURVEY 8(d), orbital energies with isolated frontier levels so
that the driver converges):  ``CCSD.solve`` -> ``get_T1_dressed_fock`` / ``get_T1_dressed_V`` -> ``EOM_CCSD(no, 3).solve``
(pymes/solver/eom_ccsd.py:46-167) and writes what the run printed and returned — CCSD energy, excitation energies, number of
Davidson passes, the Ritz values of every pass — to tests/golden/eom_davidson.json:

  small  (nocc, nvirt) = (4, 12),  s = 0.3   (host-logic test through the host simulator, CPU)
  big    (nocc, nvirt) = (12, 48), s = 0.19  (GPU test: LDS-DMA GEMMs, batched pair-packed ladders, multi-vector sigma)
  mid    (nocc, nvirt) = (20, 80), s = 0.15  (round 5: config 2's size, the nearest to config 5 an hour of CPU reaches)

``--oracle`` pins oracle/eom_oracle.py::eom_solve on the same inputs (energies 1e-9, pass count equal).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import contextlib
import io
import json
import os
import re
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle import eom_oracle as eo                                           # noqa: E402
from oracle.cases import eom_davidson_case                                      # noqa: E402
from pymes.solver import ccsd as ref_ccsd, eom_ccsd as ref_eom               # noqa: E402
from pymes.integral.partition import part_2_body_int                          # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CASES = {"small": (4, 12, 0.3, 3), "big": (12, 48, 0.19, 3), "mid": (20, 80, 0.15, 3)}


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **k)
    return out, buf.getvalue()


def run(tag, with_oracle):
    no, nv, scale, n_excit = CASES[tag]
    f, V = eom_davidson_case(no, nv, seed=0, scale=scale)
    t0 = time.time()
    cc = ref_ccsd.CCSD(no, delta_e=1e-11)
    res, _ = quiet(cc.solve, f, V, max_iter=100)
    t1, t2 = res["t1"].copy(), res["t2"].copy()
    Vb = part_2_body_int(no, V)
    fd = cc.get_T1_dressed_fock(f, t1, Vb)
    Vd = cc.get_T1_dressed_V(t1, Vb)
    t_cc = time.time() - t0
    e = ref_eom.EOM_CCSD(no, n_excit=n_excit)
    e.max_iter = 400
    t0 = time.time()
    ee, log = quiet(e.solve, fd, Vd, t2)
    t_eom = time.time() - t0
    converged = "Iterative solver converged." in log
    passes = len(re.findall(r"^\s*Iteration = ", log, flags=re.M)) + (1 if converged else 0)
    ritz = [float(x) for x in re.findall(r"Excited state \d+ energy = (-?[0-9.]+)", log)]
    # every pass prints n_excit lines; the summary after the loop prints them once more
    per_pass = [ritz[i:i + n_excit] for i in range(0, n_excit * passes, n_excit)]
    out = {"no": no, "nv": nv, "scale": scale, "seed": 0, "n_excit": n_excit, "ccsd_e": float(res["ccsd e"]),
           "t1_norm": float(np.linalg.norm(t1)), "t2_norm": float(np.linalg.norm(t2)),
           "ee": [float(x) for x in ee], "passes": passes, "converged": converged, "ritz_per_pass": per_pass,
           "reference_seconds": {"ccsd+dressing": round(t_cc, 1), "eom": round(t_eom, 1)}}
    print(f"{tag} ({no},{nv}): CCSD {res['ccsd e']:.12f}; EE {ee}; {passes} passes, converged {converged}; "
          f"{t_cc:.0f} s + {t_eom:.0f} s", flush=True)
    if with_oracle:
        mine = eo.eom_solve(no, fd, Vd, t2, n_excit=n_excit, max_iter=400)
        assert np.abs(np.asarray(mine["e"]) - np.asarray(ee)).max() < 1e-9, (mine["e"], ee)
        assert mine["iterations"] == passes, (mine["iterations"], passes)
        out["oracle_pinned"] = True
        print(f"{tag}: oracle == reference ({mine['iterations']} passes)", flush=True)
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")] or ["small", "big"]
    path = os.path.join(GOLD, "eom_davidson.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    for tag in args:
        data[tag] = run(tag, "--oracle" in sys.argv)
        with open(path, "w") as fh:
            json.dump(data, fh, indent=1)
    print("written", path)


if __name__ == "__main__":
    main()
