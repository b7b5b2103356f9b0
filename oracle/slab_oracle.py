"""Slab-sampled oracle for benchmark sizes.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

At (nocc=50, nvirt=200) one evaluation of the reference's T2 residual (pymes/solver/ccd.py:164-254 through
ccsd.py:440-456) is 1.5e13 FMA and V_pqrs is 31 GB: beyond a CPU oracle.  Every term of the residual has the first
virtual index ``a`` of R_abij as a free index, so the rows R[a0:a1] cost (a1-a0)/nvirt of the whole — provided each term is
contracted in an order that never forms a full o^2 v^2 intermediate.  This module restates ccd.py:175-252 term by term
as multi-operand einsums on an a-slab (``optimize=True`` picks that order) and builds the T1-dressed blocks it needs
(ccsd.py:290-421) directly from density-fitting factors:

    V[p,q,r,s] = sum_Q B[Q,p,r] B[Q,q,s]      (SURVEY 8(d) synthetic recipe)
    V~[p,q,r,s] = sum_Q B~[Q,p,r] B~[Q,q,s],   B~[Q] = X B[Q] Y,   X = 1 - t (virt x occ),  Y = 1 + t   

It is pinned to oracle/cc_oracle.py (itself pinned to the imported reference) by tests/test_oracle_golden.py on small
cases, where the slab must equal the corresponding rows of the full residual.
"""
import numpy as np


def ein(spec, *ops):
    return np.einsum(spec, *ops, optimize=True)


def dressed_factors(no, B, t1):
    """B~[Q,p,r] = sum_{P,R} X[p,P] B[Q,P,R] Y[R,r]  with  X[a,k] = -t[a,k] (bra virtual <- occupied),
    Y[c,i] = +t[c,i] (ket occupied <- virtual), identity elsewhere (ccsd.py:322-419 in compact form)."""
    n = B.shape[1]
    X, Y = np.eye(n), np.eye(n)
    X[no:, :no] -= t1
    Y[no:, :no] += t1
    return ein("pP,QPR,Rr->Qpr", X, B, Y)


class FactorBlocks:
    """Blocks of V (or V~) from factors, by partition.py name, optionally with index ranges (slabs)."""

    def __init__(self, no, B):
        self.no, self.B, self.n = no, B, B.shape[1]

    def _range(self, ch, cut):
        if cut is not None:
            lo, hi = cut
            return slice(self.no + lo, self.no + hi) if ch in "abcd" else slice(lo, hi)
        return slice(self.no, self.n) if ch in "abcd" else slice(0, self.no)

    def __call__(self, name, cuts=None):
        """name e.g. "iajb"; cuts = {position: (lo, hi)} restricts that index (block-local numbering)."""
        cuts = cuts or {}
        r = [self._range(ch, cuts.get(pos)) for pos, ch in enumerate(name)]
        return ein("Qpr,Qqs->pqrs", self.B[:, r[0], r[2]], self.B[:, r[1], r[3]])


def residual_slab(no, f_dressed, t1, T, B, a0, a1, is_dcsd=False):
    """Rows R[a0:a1, :, :, :] of the CCSD/DCSD T2 residual (ccsd.py:440-456 -> ccd.py:164-254) for integrals given by
    factors B [naux, n, n]; f_dressed is the T1-dressed Fock matrix (ccsd.py:226-288)."""
    quad = not is_dcsd
    nv = T.shape[0]
    S, A = slice(a0, a1), slice(0, nv)
    Vd = FactorBlocks(no, dressed_factors(no, B, t1))
    Vu = FactorBlocks(no, B)
    V_ijab = Vu("ijab")                                   # dressed ijab = undressed ijab (ccsd.py:355-357)
    V_iajb, V_iabj = Vd("iajb"), Vd("iabj")
    f_oo, f_vv = f_dressed[:no, :no], f_dressed[no:, no:]

    hole = Vd("klij")                                                          # :175-180
    if quad:
        hole = hole + ein("klcd,cdij->klij", V_ijab, T)
    R = Vd("abij", {0: (a0, a1)}) + ein("klij,abkl->abij", hole, T[S])         # :185-186
    R = R + ein("abcd,cdij->abij", Vd("abcd", {0: (a0, a1)}), T)               # :187
    if quad:                                                                   # :189-191
        R = R + ein("klcd,adkj,cbil->abij", V_ijab, T[S], T)
    Tt = 2.0 * T - T.transpose(1, 0, 2, 3)                                     # :199
    R = R + ein("acik,klcd,dblj->abij", Tt[S], V_ijab, Tt)                     # :202-204
    w = 1.0 if quad else 0.5                                                   # :213-220
    X_vv = f_vv - w * ein("adkl,lkdc->ac", Tt, V_ijab)
    X_oo = f_oo + w * ein("cdil,lkdc->ki", Tt, V_ijab)

    def ex(sa, sb):
        """Ex[sa, sb, :, :] of ccd.py:231-240."""
        def cut(blk, s):      # second index of an "ia.." block restricted like the amplitude index it pairs with
            return blk[:, s]
        e = ein("ac,cbij->abij", X_vv[sa], T[:, sb])
        e = e - ein("ki,abkj->abij", X_oo, T[sa, sb])
        e = e - ein("kaic,cbkj->abij", cut(V_iajb, sa), T[:, sb])
        e = e - ein("kbic,ackj->abij", cut(V_iajb, sb), T[sa])
        e = e + ein("acik,kbcj->abij", Tt[sa], cut(V_iabj, sb))
        if quad:
            e = e - ein("klcd,daki,cblj->abij", V_ijab, T[:, sa], T[:, sb])
            e = e + ein("klcd,daki,bclj->abij", V_ijab, T[:, sa], T[sb])
        return e
    return R + ex(S, A) + ex(A, S).transpose(1, 0, 3, 2)                       # :249-252


def fock_blocks(no, B):
    """The undressed blocks get_T1_dressed_fock reads (ccsd.py:257-286), from factors."""
    Vu = FactorBlocks(no, B)
    return {k: Vu(k) for k in ("iabj", "ijab", "ijak", "iabc", "iajb", "ijka")}


def singles_blocks(no, B):
    """The undressed blocks get_singles_residual reads (ccsd.py:431-436), from factors."""
    Vu = FactorBlocks(no, B)
    return {k: Vu(k) for k in ("aibc", "ijab", "ijka")}
