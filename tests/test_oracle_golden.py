"""CPU: the oracle against the golden vectors generated from the reference
(oracle/make_golden.py) — the pin that makes the oracle trustworthy on the GPU box,
where the reference itself is absent."""
import json
import os

import numpy as np
import pytest

from oracle import cc_oracle as oc
from oracle import io_oracle as oio
from oracle.cases import random_case, synthetic_case

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("no,nv,seed", [(2, 3, 11), (3, 5, 12), (4, 12, 13)])
def test_functions(no, nv, seed):
    g = np.load(os.path.join(GOLD, f"functions_{no}_{nv}.npz"))
    assert int(g["seed"]) == seed
    f, V, t1, t2 = random_case(no, nv, seed, symmetric=False)
    Vb = oc.split_blocks(no, V)
    eo, ev = f.diagonal()[:no], f.diagonal()[no:]
    e, T = oc.mp2(eo, ev, Vb["ijab"], Vb["abij"], 0.1)
    assert abs(e - g["mp2_e"]) < 1e-13 and np.abs(T - g["mp2_t2"]).max() < 1e-13
    fd = oc.dressed_fock(no, f, t1, Vb)
    assert np.abs(fd - g["dressed_fock"]).max() < 1e-13
    Vd = oc.dressed_V(t1, Vb)
    assert [k for k, v in Vd.items() if v is None] == [k for k in Vb if k not in oc.DRESSED_KEYS]
    for k in oc.DRESSED_KEYS:
        assert abs(Vd[k].sum() - g["dressed_sum_" + k][0]) < 1e-11
        if "dressed_" + k in g:
            assert np.abs(Vd[k] - g["dressed_" + k]).max() < 1e-13
    assert np.abs(oc.singles_residual(no, fd, t1, t2, Vb) - g["r1"]).max() < 1e-13
    for flag, tag in ((False, "ccsd"), (True, "dcsd")):
        assert np.abs(oc.ccsd_doubles_residual(no, fd, t2, Vd, is_dcsd=flag) - g["r2_" + tag]).max() < 1e-12
    for flag, tag in ((False, "ccd"), (True, "dcd")):
        r = oc.doubles_residual(no, f, t2, Vb["klij"], Vb["ijab"], Vb["abij"], Vb["iajb"], Vb["iabj"], Vb["abcd"],
                                is_dcd=flag)
        assert np.abs(r - g["r2_ccd_" + tag]).max() < 1e-12
    assert np.abs(np.array(oc.ccsd_energy(f[:no, no:], t1, t2, Vb["ijab"])) - g["energy"]).max() < 1e-13


def test_diis_bookkeeping():
    g = json.load(open(os.path.join(GOLD, "diis.json")))
    rng = np.random.default_rng(g["seed"])
    mine = oc.Diis(6)
    for it in range(10):
        err = [rng.standard_normal((3, 2)) * 0.5 ** it, rng.standard_normal((3, 3, 2, 2)) * 0.5 ** it]
        amp = [rng.standard_normal((3, 2)), rng.standard_normal((3, 3, 2, 2))]
        mine.mix(err, amp)
        assert np.allclose(mine.last_coeff, g["coeffs"][it], rtol=1e-9, atol=1e-11)
    assert np.allclose(mine.L, np.array(g["L_final"]), rtol=1e-12, atol=1e-14)
    # the quirk: overlaps of the second-newest vector are dropped once the subspace is full
    assert mine.L[4, 4] == 0.0 and np.all(mine.L[4, :4] == 0.0)


def test_fcidump_and_hf():
    g = json.load(open(os.path.join(GOLD, "fcidump.json")))
    for tag, ref in g.items():
        ne, n, ec, eps, h, V = oio.read_fcidump(os.path.join(GOLD, "fcidump", "FCIDUMP." + tag))
        assert (ne, n) == (ref["n_elec"], ref["n_orb"]) and ec == ref["e_core"]
        assert abs(V.sum() - ref["V_sum"]) < 1e-12 and np.count_nonzero(V) == ref["V_nnz"]
        assert abs(np.abs(h).sum() - ref["h_abs_sum"]) < 1e-13
        no = ne // 2
        assert abs(oio.hf_energy(no, ec, h, V) - ref["e_hf"]) < 1e-12
        assert np.abs(oio.fock_matrix(no, h, V).diagonal() - np.array(ref["fock_diag"])).max() < 1e-13


SOLVES = json.load(open(os.path.join(GOLD, "solves.json")))


@pytest.mark.parametrize("tag,kind", [("LiH.sto6g", "ccsd"), ("LiH.sto6g", "dcd"), ("H2.321g", "dcsd"),
                                      ("LiH.bare", "ccd"), ("syn_4_12", "ccsd"), ("syn_6_20", "dcsd"),
                                      ("tc_like_3_6", "dcd")])
def test_solves(tag, kind):
    ref = SOLVES[tag][kind]
    if tag.startswith("syn_"):
        no, nv = (int(x) for x in tag.split("_")[1:])
        rec = SOLVES[tag]["recipe"]
        f, V, _, _ = synthetic_case(no, nv, seed=rec["seed"], scale=rec["scale"], gap=rec["gap"])
    elif tag.startswith("tc_like"):
        f, V, _, _ = random_case(3, 6, 21, symmetric=True)
        V = V + 0.02 * np.random.default_rng(5).standard_normal(V.shape)
        V = 0.5 * (V + V.transpose(1, 0, 3, 2))
        f = np.diag(f.diagonal())
        no = 3
    else:
        ne, n, ec, eps, h, V = oio.read_fcidump(os.path.join(GOLD, "fcidump", "FCIDUMP." + tag))
        no = ne // 2
        f = oio.fock_matrix(no, h, V)
    kw = dict(delta_e=ref["delta_e"], level_shift=ref["level_shift"])
    if kind in ("ccd", "dcd"):
        r = oc.ccd_solve(no, f, V, is_dcd=(kind == "dcd"), **kw)
    else:
        r = oc.ccsd_solve(no, f, V, is_dcsd=(kind == "dcsd"), **kw)
    assert r["iterations"] == ref["iterations"]
    assert abs(r["e"] - ref["e"]) < 1e-9
    hist = [h[0] for h in r["history"]][:len(ref["history"])]
    assert np.abs(np.array(hist[:6]) - np.array(ref["history"][:6])).max() < 1e-11


@pytest.mark.parametrize("no,nv,a0,a1,dcsd", [(3, 7, 2, 4, False), (3, 7, 0, 7, True), (4, 9, 8, 9, False), (2, 5, 1, 3, True)])
def test_slab_oracle_equals_full_oracle(no, nv, a0, a1, dcsd):
    """oracle/slab_oracle.py (factor-built dressed blocks, slab-linear contraction order) == rows of the full oracle
    residual, which make_golden.py pins to the reference.  The GPU tests use it at (50,200)."""
    from oracle import slab_oracle as so
    from oracle.cases import synthetic_case
    rng = np.random.default_rng(no * 100 + nv)
    f, V, B, eps = synthetic_case(no, nv, seed=3, scale=0.4)
    t1 = 0.1 * rng.standard_normal((nv, no))
    t2 = 0.1 * rng.standard_normal((nv, nv, no, no))
    t2 = 0.5 * (t2 + t2.transpose(1, 0, 3, 2))
    Vb = oc.split_blocks(no, V)
    Vd = oc.dressed_V(t1, Vb)
    fb = so.FactorBlocks(no, so.dressed_factors(no, B, t1))
    for k in oc.DRESSED_KEYS:
        assert np.abs(fb(k) - Vd[k]).max() < 1e-13, k
    assert np.abs(fb("abcd", {0: (a0, a1)}) - Vd["abcd"][a0:a1]).max() < 1e-13
    for k, blk in so.fock_blocks(no, B).items():
        assert np.abs(blk - Vb[k]).max() < 1e-14, k
    fd = oc.dressed_fock(no, f, t1, Vb)
    full = oc.ccsd_doubles_residual(no, fd, t2, Vd, is_dcsd=dcsd)
    slab = so.residual_slab(no, fd, t1, t2, B, a0, a1, is_dcsd=dcsd)
    assert slab.shape == (a1 - a0, nv, no, no)
    assert np.abs(slab - full[a0:a1]).max() < 1e-13
