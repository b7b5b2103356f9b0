#!/usr/bin/env python3
"""Reference-generated golden vectors at BASELINE.json's configuration sizes.

Run in the BUILD CONTAINER ONLY (the reference does not travel to the GPU box):

    PYTHONPATH=/root/reference:/root/repo python oracle/make_golden_big.py [c2] [c5]

  c2  config 2, synthetic (nocc=20, nvirt=80), seed 0, s = 0.15: the reference's own
      ``CCSD.solve`` (pymes/solver/ccsd.py:47-224, ~80 s per iteration here) -> per-iteration energies,
      final energy, |T1|, |T2| into tests/golden/solves.json["syn_20_80"]; the oracle solve is pinned
      against it on the way (aborts on mismatch).
  c2dcsd / c2ccd / c2dcd  the other solver variants (DCSD; CCD.solve / DCD, ccd.py:24-162) on the same problem.
  c5  config 5, EOM-CCSD sigma build (pymes/solver/eom_ccsd.py:268-385) at (nocc=30, nvirt=120): the reference's
      ``update_singles`` / ``update_doubles`` on seeded inputs -> checksums and sampled entries of sigma1 / sigma2
      into tests/golden/eom_sigma_30_120.npz; the oracle is pinned against it on the way.

  c5gen  the same build for a trial vector WITHOUT the exchange symmetry u2_abij = u2_baji (one entry of the c5 vector
      displaced: the general sigma path — plain particle ladder, five (ov)^3 products) ->
      tests/golden/eom_sigma_30_120_general.npz.

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
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import cc_oracle as oc                      # noqa: E402
from oracle.cases import synthetic_case                 # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **k)
    return out, buf.getvalue()


def history_from_log(text):
    return [float(x) for x in re.findall(r"Correlation Energy = (-?[0-9.eE+-]+)", text)]


def c2(kind="ccsd"):
    """kind: ccsd (the config itself), dcsd, ccd, dcd — the other three solver variants on the same problem."""
    from pymes.solver import ccd as ref_ccd, ccsd as ref_ccsd
    no, nv, scale, delta_e = 20, 80, 0.15, 1e-10
    f, V, _, _ = synthetic_case(no, nv, seed=0, scale=scale)
    t0 = time.time()
    if kind in ("ccsd", "dcsd"):
        s = ref_ccsd.CCSD(no, delta_e=delta_e, is_dcsd=(kind == "dcsd"))
        res, log = quiet(s.solve, f, V)
        e_ref, t2, t1 = res["ccsd e"], res["t2"], res["t1"]
    else:
        s = ref_ccd.CCD(no, delta_e=delta_e, is_dcd=(kind == "dcd"))
        res, log = quiet(s.solve, f, V)
        e_ref, t2, t1 = res["ccd e"], res["t2 amp"], None
    hist = history_from_log(log)
    t_ref = time.time() - t0
    print(f"reference {kind} solve (20,80): E = {e_ref:+.15f}, {len(hist)} iterations, {t_ref:.0f} s", flush=True)
    t0 = time.time()
    opt = lambda *a: np.einsum(*a, optimize=True)          # the oracle's BLAS-backed mode: same algebra, minutes instead of tens
    if kind in ("ccsd", "dcsd"):
        r = oc.ccsd_solve(no, f, V, is_dcsd=(kind == "dcsd"), delta_e=delta_e, ein=opt)
    else:
        r = oc.ccd_solve(no, f, V, is_dcd=(kind == "dcd"), delta_e=delta_e, ein=opt)
    print(f"oracle {kind} solve (20,80): E = {r['e']:+.15f}, {r['iterations']} iterations, {time.time() - t0:.0f} s", flush=True)
    assert r["iterations"] == len(hist), (r["iterations"], len(hist))
    h_or = [h[0] for h in r["history"]]
    assert np.max(np.abs(np.array(h_or) - np.array(hist))) < 1e-9, "oracle history != reference"
    assert abs(r["e"] - e_ref) < 1e-10, "oracle energy != reference"
    assert np.max(np.abs(r["t2"] - t2)) < 1e-8, "oracle t2 != reference"
    if t1 is not None:
        assert np.max(np.abs(r["t1"] - t1)) < 1e-8, "oracle t1 != reference"
    path = os.path.join(GOLD, "solves.json")
    solves = json.load(open(path))
    entry = solves.setdefault("syn_20_80", {})
    entry[kind] = {"e": float(e_ref), "iterations": len(hist), "converged": True, "history": hist,
                   "delta_e": delta_e, "level_shift": 0.0, "t2_norm": float(np.linalg.norm(t2)),
                   "t1_norm": None if t1 is None else float(np.linalg.norm(t1)),
                   # a few entries of the converged amplitudes (index -> value) for a cheap element-wise check
                   "t2_samples": [[a, b, i, j, float(t2[a, b, i, j])] for (a, b, i, j) in
                                  ((0, 0, 0, 0), (3, 7, 2, 5), (79, 0, 19, 0), (41, 40, 10, 11), (17, 63, 19, 3))],
                   "reference_seconds": t_ref}
    entry["recipe"] = {"seed": 0, "scale": scale, "gap": 3.0}
    with open(path, "w") as fh:
        json.dump(solves, fh, indent=1)
    print(f"syn_20_80 / {kind} written to", path)


def c5(general=False):
    from oracle import eom_oracle as eo
    from oracle.cases import eom_sigma_case
    from pymes.solver import eom_ccsd as ref_eom
    from pymes.integral.partition import part_2_body_int
    no, nv, seed = 30, 120, 31
    fd, V, t2, u1, u2 = eom_sigma_case(no, nv, seed, scale=0.12)
    if general:         # what tests/test_gpu_big.py displaces: the vector loses u2_abij = u2_baji
        u2[3, 5, 1, 2] += 0.25
    dictV = part_2_body_int(no, V)
    eom = ref_eom.EOM_CCSD(no)
    t0 = time.time()
    (s1), _ = quiet(eom.update_singles, fd, dictV, u1, u2, t2)
    (s2), _ = quiet(eom.update_doubles, fd, dictV, u1, u2, t2)
    t_ref = time.time() - t0
    print(f"reference sigma build (30,120): {t_ref:.0f} s", flush=True)
    t0 = time.time()
    o1 = eo.sigma_singles(no, fd, dictV, u1, u2, t2)
    o2 = eo.sigma_doubles(no, fd, dictV, u1, u2, t2)
    print(f"oracle sigma build: {time.time() - t0:.0f} s", flush=True)
    e1 = np.max(np.abs(o1 - s1)) / np.max(np.abs(s1))
    e2 = np.max(np.abs(o2 - s2)) / np.max(np.abs(s2))
    print("oracle vs reference: rel err sigma1", e1, "sigma2", e2)
    assert e1 < 1e-11 and e2 < 1e-11
    # fixtures: sigma1 in full (3600 numbers), sigma2 as an a-slab + strided samples + checksums
    idx = np.random.default_rng(seed + 1).integers(0, s2.size, size=4096)
    name = "eom_sigma_30_120_general.npz" if general else "eom_sigma_30_120.npz"
    np.savez_compressed(os.path.join(GOLD, name), seed=seed, scale=0.12, sigma1=s1,
                        sigma2_slab=s2[7:8], sigma2_idx=idx, sigma2_val=s2.reshape(-1)[idx],
                        sigma2_sums=np.array([s2.sum(), np.abs(s2).sum(), np.linalg.norm(s2)]),
                        reference_seconds=t_ref)
    print(name, "written")


if __name__ == "__main__":
    which = sys.argv[1:] or ["c2", "c5"]
    if "c2" in which:
        c2()
    for kind in ("dcsd", "ccd", "dcd"):
        if "c2" + kind in which:
            c2(kind)
    if "c5" in which:
        c5()
    if "c5gen" in which:
        c5(general=True)
