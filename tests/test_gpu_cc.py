"""Parity of the HIP CC path (through the C-ABI) against the oracle and the golden
vectors generated from the reference (tests/golden, oracle/make_golden.py)."""
import os

import numpy as np
import pytest

from oracle import cc_oracle as oc
from oracle.cases import random_case
from pymes_amd.device import Context

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-12   # fp64 contraction in a different summation order; values are O(1)


def run_functions(lib, no, nv, seed, gold=None):
    f, V, t1, t2 = random_case(no, nv, seed, symmetric=False)
    Vb = oc.split_blocks(no, V)
    ctx = Context(no, nv, lib=lib)
    try:
        ctx.set_V_pqrs(V)
        for nm, blk in Vb.items():
            assert np.array_equal(ctx.V_block(nm).get(), blk), nm
        eo, ev = f.diagonal()[:no].copy(), f.diagonal()[no:].copy()
        ctx.set_orbital_energies(eo, ev)
        dF, dT1, dT2 = ctx.array(f), ctx.array(t1), ctx.array(t2)
        # mp2
        tm = ctx.empty(t2.shape)
        e_dir, e_ex = ctx.mp2(tm, 0.1)
        e_ref, T_ref = oc.mp2(eo, ev, Vb["ijab"], Vb["abij"], 0.1)
        assert abs(e_dir + e_ex - e_ref) < TOL and np.abs(tm.get() - T_ref).max() < TOL
        # dressing
        fd = ctx.empty(f.shape)
        ctx.dress_fock(dF, dT1, fd)
        fd_ref = oc.dressed_fock(no, f, t1, Vb)
        assert np.abs(fd.get() - fd_ref).max() < TOL
        ctx.dress_V(dT1, oc.DRESSED_KEYS)
        Vd_ref = oc.dressed_V(t1, Vb)
        for k in oc.DRESSED_KEYS:
            assert np.abs(ctx.V_block(k, dressed=True).get() - Vd_ref[k]).max() < TOL, k
        # residuals
        r1 = ctx.empty(t1.shape)
        ctx.singles_residual(fd, dT1, dT2, r1)
        r1_ref = oc.singles_residual(no, fd_ref, t1, t2, Vb)
        assert np.abs(r1.get() - r1_ref).max() < TOL
        out = {}
        for dcd in (False, True):
            r2 = ctx.empty(t2.shape)
            ctx.doubles_residual(fd, dT2, r2, is_dcd=dcd, dressed=True)
            ref = oc.ccsd_doubles_residual(no, fd_ref, t2, Vd_ref, is_dcsd=dcd)
            assert np.abs(r2.get() - ref).max() < 10 * TOL
            out["r2_dcsd" if dcd else "r2_ccsd"] = r2.get()
            r2c = ctx.empty(t2.shape)
            ctx.doubles_residual(dF, dT2, r2c, is_dcd=dcd, dressed=False)
            refc = oc.doubles_residual(no, f, t2, Vb["klij"], Vb["ijab"], Vb["abij"], Vb["iajb"], Vb["iabj"],
                                       Vb["abcd"], is_dcd=dcd)
            assert np.abs(r2c.get() - refc).max() < 10 * TOL
            out["r2_ccd_dcd" if dcd else "r2_ccd_ccd"] = r2c.get()
            # ladder split off and added per a-slab (the sharded form) gives the same residual
            r2s = ctx.empty(t2.shape)
            ctx.doubles_residual(dF, dT2, r2s, is_dcd=dcd, dressed=False, skip_ladder=True)
            half = nv // 2
            ctx.ladder(dT2, r2s, 0, half, dressed=False, beta=1.0)
            ctx.ladder(dT2, r2s, half, nv, dressed=False, beta=1.0)
            assert np.abs(r2s.get() - refc).max() < 10 * TOL
        en = ctx.ccsd_energy(dF, dT1, dT2)
        assert np.abs(np.array(en) - np.array(oc.ccsd_energy(f[:no, no:], t1, t2, Vb["ijab"]))).max() < TOL
        en2 = ctx.ccd_energy(dT2)
        assert np.abs(np.array(en2) - np.array(oc.ccd_energy(t2, Vb["ijab"]))).max() < TOL
        if gold is not None:   # the reference's own outputs
            g = np.load(gold)
            assert int(g["seed"]) == seed
            assert np.abs(fd.get() - g["dressed_fock"]).max() < TOL
            assert np.abs(r1.get() - g["r1"]).max() < TOL
            for k in ("r2_ccsd", "r2_dcsd", "r2_ccd_ccd", "r2_ccd_dcd"):
                assert np.abs(out[k] - g[k]).max() < 10 * TOL, k
            for k in oc.DRESSED_KEYS:
                got = ctx.V_block(k, dressed=True).get()
                assert abs(got.sum() - g["dressed_sum_" + k][0]) < 1e-10
                if "dressed_" + k in g:
                    assert np.abs(got - g["dressed_" + k]).max() < TOL
            assert np.abs(np.array(en) - g["energy"]).max() < TOL
    finally:
        ctx.close()


@pytest.mark.parametrize("no,nv,seed", [(2, 3, 11), (3, 5, 12), (4, 12, 13)])
def test_functions_vs_oracle_and_golden(gpu_lib, no, nv, seed):
    run_functions(gpu_lib, no, nv, seed, gold=os.path.join(GOLD, f"functions_{no}_{nv}.npz"))


@pytest.mark.parametrize("no,nv,seed", [(1, 1, 5), (7, 9, 6), (8, 24, 7), (6, 40, 8)])
def test_functions_vs_oracle_more_shapes(gpu_lib, no, nv, seed):
    run_functions(gpu_lib, no, nv, seed)


def test_factor_built_integrals(gpu_lib):
    from oracle.io_oracle import synthetic_factors, eri_from_factors
    no, nv = 5, 11
    B, eps = synthetic_factors(no, nv, seed=3, scale=0.3)
    V = eri_from_factors(B)
    ctx = Context(no, nv, lib=gpu_lib)
    try:
        ctx.set_V_from_factors(B)
        for nm, blk in oc.split_blocks(no, V).items():
            assert np.abs(ctx.V_block(nm).get() - blk).max() < 1e-13, nm
    finally:
        ctx.close()


@pytest.mark.parametrize("no,nv,seed", [(1, 1, 1), (2, 3, 2), (5, 9, 3), (8, 24, 4), (7, 33, 5)])
def test_pair_packed_ladder(gpu_lib, no, nv, seed):
    """ccd.py:187 in pair-packed form (exchange-symmetric V and T) against the oracle's plain einsum,
    computed in two row slabs (the sharded form) and through the residual flag."""
    f, V, t1, t2 = random_case(no, nv, seed, symmetric=True)
    if seed % 2:        # transcorrelated-like: non-hermitian, only V_pqrs = V_qpsr survives
        V = V + 0.05 * np.random.default_rng(seed).standard_normal(V.shape)
        V = 0.5 * (V + V.transpose(1, 0, 3, 2))
    Vb = oc.split_blocks(no, V)
    ctx = Context(no, nv, lib=gpu_lib)
    try:
        ctx.set_V_pqrs(V)
        dT2, dF = ctx.array(t2), ctx.array(f)
        ref = np.einsum("abcd,cdij->abij", Vb["abcd"], t2)
        npp = nv * (nv + 1) // 2
        L = ctx.zeros((npp, no * no))
        cut = npp // 3
        ctx.ladder_sym(dT2, L, 0, cut)
        ctx.ladder_sym(dT2, L, cut, npp)
        R0 = np.random.default_rng(seed).standard_normal(t2.shape)
        R = ctx.array(R0)
        ctx.ladder_sym_unpack(L, R, beta=0.5)
        assert np.abs(R.get() - (ref + 0.5 * R0)).max() < TOL * max(1.0, np.abs(ref).max())
        for dcd in (False, True):
            refr = oc.doubles_residual(no, f, t2, Vb["klij"], Vb["ijab"], Vb["abij"], Vb["iajb"], Vb["iabj"],
                                       Vb["abcd"], is_dcd=dcd)
            for rings in (False, True):      # packed ladder alone, and with the symmetry-merged ring products
                r2 = ctx.empty(t2.shape)
                ctx.doubles_residual(dF, dT2, r2, is_dcd=dcd, sym_ladder=True, sym_rings=rings)
                assert np.abs(r2.get() - refr).max() < 10 * TOL
        from tests.test_host_engine import hole_ladder_check
        hole_ladder_check(ctx, no, nv, Vb, t2, TOL)
        # dressed blocks: the packed copy must follow a re-dressing
        dT1 = ctx.array(t1)
        for scale in (1.0, -0.5):
            ctx.dress_V(ctx.array(scale * t1), ["abcd"])
            Vd = oc.dressed_block("abcd", scale * t1, Vb)
            ctx.ladder_sym(dT2, L, 0, npp, dressed=True)
            ctx.ladder_sym_unpack(L, R, beta=0.0)
            assert np.abs(R.get() - np.einsum("abcd,cdij->abij", Vd, t2)).max() < TOL * 10
    finally:
        ctx.close()


def test_sharded_residual_simulated_ranks(gpu_lib):
    from tests.test_host_engine import sharded_residual_check
    sharded_residual_check(gpu_lib, [(3, 5, 2), (6, 17, 3), (8, 24, 4)], (1, 2, 8), 1e-11)


def test_sharded_residual_simulated_ranks_bra_dressed(gpu_lib, monkeypatch):
    """Every rank dresses the bra of its rows of the pair-packed V_abcd (ladder_dress_kernel with a row range) instead of
    forming its share of Q_kb."""
    from tests.test_host_engine import sharded_residual_check
    monkeypatch.setenv("PYMES_LADDER_DRESS", "1")
    sharded_residual_check(gpu_lib, [(3, 5, 2), (6, 17, 3), (8, 24, 4)], (1, 2, 8), 1e-11)
