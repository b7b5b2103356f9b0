"""Closed-shell EOM-CCSD (pymes/solver/eom_ccsd.py) on the MI355X engine.

Drop-in for ``pymes.solver.eom_ccsd.EOM_CCSD``: ``EOM_CCSD(no, n_excit).solve(f_dressed,
dict_t_V_dressed, t_T_abij)`` returns the excitation energies; ``update_singles`` /
``update_doubles`` keep the reference's host-array call forms.

The sigma build H̄·u (eom_ccsd.py:268-385, 18 + 44 terms) runs on the device through the
contraction engine (fp64 MFMA GEMM).  Every V·T product that does not depend on the trial
vector is hoisted into per-solve intermediates (``_Sigma.__init__``), so one sigma costs the
particle ladder ``V_abcd·u2`` plus eight (ov)^3 products instead of the reference's 34
three-operand einsums.  The subspace algebra (QR, B = Uᵀσ, eig, collapse/expand) follows
eom_ccsd.py:46-167 and stays on the host: the trial vectors live on the device, the host sees
only overlaps and coefficients.
"""
import os
import time

import numpy as np

from pymes_amd.device import Context, DeviceArray, PymesError
from pymes_amd.integral.device import DressedDeviceIntegrals
from pymes_amd.integral.partition import BLOCK_NAMES
from pymes_amd.log import print_logging_info, print_title
from pymes_amd.mixer.diis import _single_threaded_blas


class _Sigma:
    """Device-resident H̄·u with hoisted u-independent intermediates.

    Pair layouts (ov x ov matrices): Xd[(a,i),(b,j)] = X[a,b,i,j], Xx[(a,j),(b,i)] = X[a,b,i,j]."""

    # the blocks a sigma build reads (eom_ccsd.py:268-385)
    BLOCKS = ("ijab", "iabj", "iajb", "ijka", "ijak", "iabc", "iajk", "abic", "klij", "abcd")

    def __init__(self, ctx, f, t2, dressed=False):
        """``dressed``: read the context's T1-DRESSED blocks (the context of a CCSD solve whose integrals were dressed in
        place, ``CCSD.get_T1_dressed_V`` on a ``DeviceIntegrals``) instead of blocks uploaded as they are."""
        self.ctx, c = ctx, ctx
        no, nv = ctx.no, ctx.nv
        self.no, self.nv = no, nv
        self.dressed = bool(dressed)
        f = np.asarray(f, dtype=np.float64)
        self.foo, self.fov, self.fvv = c.array(f[:no, :no]), c.array(f[:no, no:]), c.array(f[no:, no:])
        V = {nm: c.V_block(nm, self.dressed) for nm in self.BLOCKS}
        self.V = V
        T = t2
        self.T = T
        self.Td = c.permute("abij->aibj", T)
        self.Tx = c.permute("abij->ajbi", T)
        Vd = c.permute("klcd->ckdl", V["ijab"])          # [(c,k),(d,l)]
        Vx = c.permute("klcd->cldk", V["ijab"])          # [(c,l),(d,k)]
        # ---- singles (eom_ccsd.py:288-308) ----------------------------------------------------
        # W1[(c,k),(a,i)] = sum_jb (2V[j,k,b,c]-V[j,k,c,b]) (2T[b,a,j,i]-T[a,b,j,i]) + 2V_iabj[k,a,c,i] - V_iajb[k,a,i,c]
        Vq = c.permute("jkbc->ckbj", V["ijab"], alpha=2.0)
        c.permute("jkcb->ckbj", V["ijab"], out=Vq, alpha=-1.0, beta=1.0)
        Tq = c.permute("baji->bjai", T, alpha=2.0)
        c.permute("abji->bjai", T, out=Tq, alpha=-1.0, beta=1.0)
        self.W1 = c.contract("ckbj,bjai->ckai", Vq, Tq)
        c.permute("kaci->ckai", V["iabj"], out=self.W1, alpha=2.0, beta=1.0)
        c.permute("kaic->ckai", V["iajb"], out=self.W1, alpha=-1.0, beta=1.0)
        # Gvv_s[a,c] = fvv + sum V[j,k,b,c] (-2T[b,a,j,k] + T[a,b,j,k]);  Goo_s[k,i] = -foo + sum (-2V[j,k,b,c]+V[j,k,c,b]) T[b,c,j,i]
        self.Gvv_s = c.array(f[no:, no:])
        c.contract("jkbc,bajk->ac", V["ijab"], T, out=self.Gvv_s, alpha=-2.0, beta=1.0)
        c.contract("jkbc,abjk->ac", V["ijab"], T, out=self.Gvv_s, alpha=1.0, beta=1.0)
        self.Goo_s = c.array(-f[:no, :no])
        c.contract("jkbc,bcji->ki", V["ijab"], T, out=self.Goo_s, alpha=-2.0, beta=1.0)
        c.contract("jkcb,bcji->ki", V["ijab"], T, out=self.Goo_s, alpha=1.0, beta=1.0)
        # ---- doubles: (V.T) pair matrices (eom_ccsd.py:352-372) ------------------------------------
        M_A = c.contract("ckai,ckdl->aidl", self.Td, Vd)       # sum_kc V[k,l,c,d] T[c,a,k,i]
        M_B = c.contract("aick,ckdl->aidl", self.Tx, Vd)       # sum_kc V[k,l,c,d] T[a,c,k,i]
        self.M_C = c.contract("ckai,dlck->aidl", self.Td, Vx)  # sum_kc V[k,l,d,c] T[c,a,k,i]
        self.M_D = c.contract("aick,dlck->aidl", self.Tx, Vx)  # sum_kc V[k,l,d,c] T[a,c,k,i]
        self.M1 = c.permute("kaci->aick", V["iabj"])           # Wd'[(a,i),(c,k)] = V_iabj[k,a,c,i]
        c.lincomb(self.M1, [self.M1, M_A, M_B], [1.0, 2.0, -1.0])
        self.Ud = c.permute("kaic->aick", V["iajb"])           # Ud[(a,i),(c,k)] = V_iajb[k,a,i,c]
        self.M2 = c.empty(self.M1.shape)
        c.lincomb(self.M2, [self.M_D, self.M_C, self.Ud], [1.0, -2.0, -1.0])
        # for exchange-symmetric trial doubles ut = 2 u2 - u2^(ab) is 2 u2d - u2x in the pair layout, so
        # M1.utd + M2.u2d + M_C.u2x = (2 M1 + M2).utd / 2 + (M_D - Ud).u2x / 2 (see doubles()): one product + half of Dx
        self.M12 = c.empty(self.M1.shape)
        c.lincomb(self.M12, [self.M1, self.M2], [2.0, 1.0])
        # crossed layout: the result is symmetrised by P(ijab, jiba) (:377), i.e. only Dx + Dx^T counts, and u2x is a
        # symmetric matrix for exchange-symmetric u2, so -u2x.Ud^T (:364) may be replaced by its transpose -Ud.u2x:
        # M_D.u2x - Ud.u2x = (M_D - Ud).u2x, one product instead of two
        self.MDU = c.empty(self.M1.shape)
        c.lincomb(self.MDU, [self.M_D, self.Ud], [1.0, -1.0])
        del M_A, M_B, Vd, Vx, Vq, Tq
        # V_kacd.T products of the u1 terms (eom_ccsd.py:334, :343, :345, :346), u-independent like the pair matrices above:
        #   WA[a,d,b,j] = sum_ck (2 V[k,a,c,d] - V[k,a,d,c]) T[c,b,k,j] - V[k,a,c,d] T[b,c,k,j],   W3[a,d,b,i] = sum_ck V[k,a,d,c] T[b,c,k,i]
        # so that a sigma build contracts them with u1 over d (o^2 v^3, HBM-bound) instead of forming V.u1 first and paying
        # three (ov)^3 products per trial vector.  Two v^3 o arrays (0.4 GB each at (30,120)).
        self.WA = c.contract("kacd,cbkj->adbj", V["iabc"], T, alpha=2.0)
        c.contract("kadc,cbkj->adbj", V["iabc"], T, out=self.WA, alpha=-1.0, beta=1.0)
        c.contract("kacd,bckj->adbj", V["iabc"], T, out=self.WA, alpha=-1.0, beta=1.0)
        self.W3 = c.contract("kadc,bcki->adbi", V["iabc"], T)
        # small hoisted V.T blocks
        self.A3 = c.contract("klci,cbkj->libj", V["ijak"], T, alpha=-2.0)                 # A_oovo
        c.contract("klic,cbkj->libj", V["ijka"], T, out=self.A3, alpha=1.0, beta=1.0)
        c.contract("kldi,bdkj->libj", V["ijak"], T, out=self.A3, alpha=1.0, beta=1.0)
        self.A4 = c.contract("klid,adkj->liaj", V["ijka"], T)
        self.A6 = c.contract("lacd,cdji->laji", V["iabc"], T)
        self.Gvv = c.array(f[no:, no:])
        c.contract("klcd,cakl->ad", V["ijab"], T, out=self.Gvv, alpha=-2.0, beta=1.0)
        c.contract("klcd,ackl->ad", V["ijab"], T, out=self.Gvv, alpha=1.0, beta=1.0)
        self.Goo = c.array(-f[:no, :no])
        c.contract("klcd,cdki->li", V["ijab"], T, out=self.Goo, alpha=-2.0, beta=1.0)
        c.contract("kldc,cdki->li", V["ijab"], T, out=self.Goo, alpha=1.0, beta=1.0)
        self.B2 = c.permute("klij->klij", V["klij"])
        c.contract("klcd,cdij->klij", V["ijab"], T, out=self.B2, alpha=1.0, beta=1.0)
        # particle ladder (:383): pair-packed form (1/4 of the flops) whenever V_abcd = V_badc and the trial doubles
        # are exchange-symmetric, u2_abij = u2_baji — true for every vector the Davidson driver generates
        self.v_sym = c.exchange_symmetric(V["abcd"])
        # T_abij = T_baji (every CCSD solution): P(ijab,jiba)[T B5] = T (B5 + B5^(lkji)), so that term rides in the
        # product with B' of eom_ccsd.py:381 — one v^2 o^4 product less per sigma
        self.fused_ok = c.pairs_supported()
        self.t_sym = c.exchange_symmetric(T)
        # eom_ccsd.py:380-382 in pair-packed rows needs B2_klij = B2_lkji and V_klcd = V_lkdc (then B' has it for symmetric u2)
        self.hole_sym = self.t_sym and c.exchange_symmetric(self.B2) and c.exchange_symmetric(V["ijab"])
        self.L = c.empty((nv * (nv + 1) // 2, no * no)) if self.v_sym else None
        # ---- multi-vector sigma (apply_many): what the stacked products read --------------------------------------------
        self.many_ok = bool(self.v_sym and self.hole_sym and self.fused_ok and self.t_sym)
        if self.many_ok:
            self.fovT = c.array(np.ascontiguousarray(f[:no, no:].T))
            # WA / W3 with the contracted index third: u1[d,i] then meets WA[a,b,d,:] as a batch of (a,b) products whose
            # o x o results ARE the tiles D[z,a,b,:,:] — no transposed copy of a v^3 o array and no permuted accumulation of
            # the result per build (0.7 ms each at (30,120), rocprofv3 round 4)
            self.WAt = c.permute("adbj->abdj", self.WA)
            self.W3t = c.permute("adbi->abdi", self.W3)
            # The four u1 terms with one free index on u1 (:336-338, :341 region: A_oovo, A4, A6 and V_iajk) as ONE product
            # u1[a,l] A346[l,b,i,j].  Everything added to D is symmetrised by P(ijab,jiba) afterwards (:377), so a term
            # X_abij may be replaced by its partner X_baji: sum_l u1[b,l] A4[l,i,a,j] -> sum_l u1[a,l] A4[l,j,b,i], etc.
            self.A346 = c.permute("libj->lbij", self.A3)
            c.permute("ljbi->lbij", self.A4, out=self.A346, beta=1.0)
            c.lincomb(self.A346, [self.A346, self.A6, V["iajk"]], [1.0, -1.0, -1.0])

    def exchange_symmetric(self, u2):
        return self.ctx.exchange_symmetric(u2)      # one reduction kernel, no temporary

    # ------------------------------------------------------------------------------------------
    def singles(self, u1, u2):
        """eom_ccsd.py:268-310."""
        c, V = self.ctx, self.V
        ut = c.permute("abij->abij", u2, alpha=2.0)                   # 2 u2[a,b,i,j] - u2[b,a,i,j]
        c.permute("baij->abij", u2, out=ut, alpha=-1.0, beta=1.0)
        s = c.contract("ck,ckai->ai", u1, self.W1)
        c.contract("ac,ci->ai", self.Gvv_s, u1, out=s, beta=1.0)
        c.contract("ak,ki->ai", u1, self.Goo_s, out=s, beta=1.0)
        c.contract("jb,baji->ai", self.fov, ut, out=s, beta=1.0)
        c.contract("jkib,abjk->ai", V["ijka"], ut, out=s, alpha=-1.0, beta=1.0)
        c.contract("jabc,bcji->ai", V["iabc"], ut, out=s, beta=1.0)
        return s

    def doubles(self, u1, u2, u2_sym=None):
        """eom_ccsd.py:312-385.  ``u2_sym``: the caller's knowledge that u2_abij = u2_baji (checked here, with a
        device-to-host synchronisation, when None)."""
        c, V, T = self.ctx, self.V, self.T
        u2x = c.permute("abij->ajbi", u2)
        if u2_sym is None:
            u2_sym = self.exchange_symmetric(u2)
        # ---- (ov)^3 products -----------------------------------------------------------------------
        utd = c.permute("abij->aibj", u2, alpha=2.0)                  # ut[d,b,l,j] = 2u2[d,b,l,j] - u2[b,d,l,j]
        c.permute("baij->aibj", u2, out=utd, alpha=-1.0, beta=1.0)
        if u2_sym:
            # exchange-symmetric u2: utd = 2 u2d - u2x as matrices, hence M1.utd + M2.u2d + M_C.u2x =
            # (2 M1 + M2).utd / 2 + (M_D - Ud).u2x / 2, and the second product IS Dx (the C / D form of the ring terms, as
            # in the CCSD residual): TWO (ov)^3 products per sigma, Dd = M12.utd / 2 + Dx / 2
            Dx = c.contract("ajdl,dlbi->ajbi", self.MDU, u2x)                     # :372 and :364 (transposed)
            Dd = c.permute("ajbi->ajbi", Dx, alpha=0.5)                           # same memory layout as "aibj"
            c.contract("aidl,dlbj->aibj", self.M12, utd, out=Dd, alpha=0.5, beta=1.0)
        else:
            u2d = c.permute("abij->aibj", u2)
            Dd = c.contract("aidl,dlbj->aibj", self.M1, utd)
            c.contract("aidl,dlbj->aibj", self.M2, u2d, out=Dd, beta=1.0)
            c.contract("aidl,dlbj->aibj", self.M_C, u2x, out=Dd, beta=1.0)      # u2x[(d,l),(b,j)] = u2[d,b,j,l]
            Dx = c.contract("ajdl,dlbi->ajbi", self.M_D, u2x)                     # :372  u2[d,b,i,l]
            c.contract("ajck,bick->ajbi", u2x, self.Ud, out=Dx, alpha=-1.0, beta=1.0)  # :364
        # ---- one-index dressings -----------------------------------------------------------------------
        Xoo = c.contract("klid,dl->ki", V["ijka"], u1, alpha=-2.0)
        c.contract("kldi,dl->ki", V["ijak"], u1, out=Xoo, beta=1.0)
        c.contract("kd,di->ki", self.fov, u1, out=Xoo, alpha=-1.0, beta=1.0)
        c.contract("kldc,dcil->ki", V["ijab"], u2, out=Xoo, alpha=-2.0, beta=1.0)
        c.contract("kldc,dcli->ki", V["ijab"], u2, out=Xoo, beta=1.0)
        c.contract("ki,akbj->aibj", Xoo, self.Td, out=Dd, beta=1.0, batch="a")
        Xvv = c.contract("ladc,dl->ac", V["iabc"], u1, alpha=2.0)
        c.contract("lacd,dl->ac", V["iabc"], u1, out=Xvv, alpha=-1.0, beta=1.0)
        c.contract("al,lc->ac", u1, self.fov, out=Xvv, alpha=-1.0, beta=1.0)
        c.contract("lkcd,adlk->ac", V["ijab"], u2, out=Xvv, alpha=-2.0, beta=1.0)
        c.contract("lkcd,dalk->ac", V["ijab"], u2, out=Xvv, beta=1.0)
        D = c.contract("ac,cbij->abij", Xvv, T)
        # V_kacd.T.u1 terms (:334, :343, :345, :346) through the hoisted V.T intermediates: o^2 v^3 instead of (ov)^3 each
        c.contract("adbj,di->abij", self.WA, u1, out=D, beta=1.0)
        c.contract("adbi,dj->abij", self.W3, u1, out=D, alpha=-1.0, beta=1.0)
        c.contract("ad,dbij->abij", self.Gvv, u2, out=D, beta=1.0)
        c.contract("li,ablj->abij", self.Goo, u2, out=D, beta=1.0, batch="ab")
        c.contract("al,libj->abij", u1, self.A3, out=D, beta=1.0)
        c.contract("bl,liaj->abij", u1, self.A4, out=D, beta=1.0)
        c.contract("bl,laji->abij", u1, self.A6, out=D, alpha=-1.0, beta=1.0)
        B5 = c.contract("klid,dj->klij", V["ijka"], u1)
        if not self.t_sym:
            c.contract("abkl,klij->abij", T, B5, out=D, beta=1.0)
        c.contract("ak,kbij->abij", u1, V["iajk"], out=D, alpha=-1.0, beta=1.0)
        c.contract("abic,cj->abij", V["abic"], u1, out=D, beta=1.0)
        packed = self.v_sym and u2_sym and self.hole_sym
        if packed and self.fused_ok:
            # the terms (:380-383) that stay outside P(ijab,jiba) all live in the pair-packed rows L here, so the
            # symmetrisation (:377) of D and of the two pair matrices and the unpacking of L are ONE pass (the assembly
            # kernel of the CCSD residual) instead of two transposed accumulations, a transposition, a sum and an unpack
            npp = self.L.shape[0]
            B5s = c.permute("klij->klij", B5)
            c.permute("lkji->klij", B5, out=B5s, beta=1.0)
            c.ladder_sym(u2, self.L, 0, npp, dressed=self.dressed)
            c.hole_ladder_packed(u2, self.B2, self.L, 0, npp)
            c.hole_ladder_packed(T, B5s, self.L, 0, npp, y=u2)
            return c.symmetrised_assemble(D, Dd, Dx, c.empty(D.shape), L=self.L)
        c.permute("aibj->abij", Dd, out=D, beta=1.0)
        c.permute("ajbi->abij", Dx, out=D, beta=1.0)
        # ---- P(ijab, jiba) (:377), then the unpermuted terms (:380-383) ----------------------------------
        S = c.permute("baji->abij", D)
        c.lincomb(D, [D, S], [1.0, 1.0])
        if packed:
            # all three remaining terms in the pair-packed rows (a >= b, i >= j): the particle ladder (:383) and the two
            # hole-ladder-shaped products (:380-382; B2 and B' are symmetric under (kl)(ij) -> (lk)(ji)) — 1/4 of their
            # flops; B' = V_kldc u2_dcij itself is formed pair-packed inside the second call, on top of the symmetrised
            # u1 term that was held back above
            npp = self.L.shape[0]
            B5s = c.permute("klij->klij", B5)
            c.permute("lkji->klij", B5, out=B5s, beta=1.0)
            c.ladder_sym(u2, self.L, 0, npp, dressed=self.dressed)
            c.hole_ladder_packed(u2, self.B2, self.L, 0, npp)
            c.hole_ladder_packed(T, B5s, self.L, 0, npp, y=u2)
            c.ladder_sym_unpack(self.L, D, beta=1.0)
            return D
        Bn = c.contract("kldc,dcij->klij", V["ijab"], u2)
        if self.t_sym:           # + the symmetrised u1 term that was held back above
            c.permute("klij->klij", B5, out=Bn, beta=1.0)
            c.permute("lkji->klij", B5, out=Bn, beta=1.0)
        c.contract("abkl,klij->abij", u2, self.B2, out=D, beta=1.0)               # :380, :382
        c.contract("abkl,klij->abij", T, Bn, out=D, beta=1.0)                     # :381
        if self.v_sym and u2_sym:                                                 # :383
            c.ladder_sym(u2, self.L, 0, self.L.shape[0], dressed=self.dressed)
            c.ladder_sym_unpack(self.L, D, beta=1.0)
        else:
            c.contract("abcd,cdij->abij", V["abcd"], u2, out=D, beta=1.0)
        return D

    def apply(self, u1, u2, u2_sym=None):
        return self.singles(u1, u2), self.doubles(u1, u2, u2_sym)

    # ------------------------------------------------------------------------------------------
    MAX_STACK = 16          # vectors per stacked build (the batched ladder launches take at most 64)

    def stack_limit(self):
        """How many trial vectors one stacked build may take: nine (ov)^2-sized temporaries per vector (X, Tt, DxT, DdT, D,
        S2, the packed ladder rows and their operands) must fit in half of what the device has free right now."""
        free = self.ctx.mem_info()[0] + self.ctx._spare_bytes
        per_vector = 9 * 8 * (self.no * self.nv) ** 2
        return int(max(1, min(self.MAX_STACK, (free // 2) // max(per_vector, 1))))

    def apply_many(self, u1s, u2s, syms=None, out1=None, out2=None):
        """sigma for any number of trial vectors: stacked builds (``_apply_stack``) over chunks of at most ``stack_limit()``
        vectors — every batched kernel and every temporary is sized by the chunk, not by the subspace; a chunk whose
        temporaries cannot be allocated after all is built vector by vector.  ``out1`` / ``out2``: device arrays that receive
        sigma1_z / sigma2_z (e.g. the two parts of a flat subspace vector) instead of fresh ones."""
        k = len(u1s)
        if syms is None:
            syms = [self.exchange_symmetric(u2) for u2 in u2s]
        res = []
        step = self.stack_limit()
        for lo in range(0, k, step):
            hi = min(k, lo + step)
            o1 = out1[lo:hi] if out1 is not None else None
            o2 = out2[lo:hi] if out2 is not None else None
            try:
                res += self._apply_stack(u1s[lo:hi], u2s[lo:hi], syms[lo:hi], o1, o2)
            except PymesError as err:
                if hi - lo < 2 or "memory" not in str(err).lower():
                    raise
                self.ctx.trim()          # out of device memory in the middle of a stacked build: one vector at a time
                for z in range(lo, hi):
                    res += self._apply_stack(u1s[z:z + 1], u2s[z:z + 1], syms[z:z + 1],
                                             o1[z - lo:z - lo + 1] if o1 is not None else None,
                                             o2[z - lo:z - lo + 1] if o2 is not None else None)
        return res

    def _into(self, pairs, out1, out2):
        """Results of the vector-by-vector build copied into the caller's output arrays, if any."""
        if out1 is None and out2 is None:
            return pairs
        res = []
        for z, (s1, s2) in enumerate(pairs):
            if out1 is not None:
                s1 = out1[z].copy_from(s1)
            if out2 is not None:
                s2 = out2[z].copy_from(s2)
            res.append((s1, s2))
        return res

    def _apply_stack(self, u1s, u2s, syms, out1=None, out2=None):
        """sigma for k trial vectors at once: [(sigma1_z, sigma2_z)].  The reference builds sigma vector by vector for
        the whole Davidson subspace (eom_ccsd.py:95-101); here the k vectors are stacked, so that every operand that does not
        depend on the trial vector — the hoisted (ov)^2 pair matrices, V_abcd (pair-packed), T, the V.T intermediates — is
        read ONCE for all of them: the (ov)^3 products become [k ov x ov x ov] GEMMs, the particle ladders one batched
        launch, the one-index terms GEMMs with M = k v.  Needs exchange-symmetric vectors and the pair-packed forms
        (``many_ok``); anything else goes vector by vector through ``apply``."""
        k = len(u1s)
        if k < 2 or not self.many_ok or not all(syms):
            return self._into([self.apply(u1, u2, u2_sym=sy) for u1, u2, sy in zip(u1s, u2s, syms)], out1, out2)
        c, V, T = self.ctx, self.V, self.T
        no, nv = self.no, self.nv

        def part(stack, z):
            n = stack.size // k
            return DeviceArray(c, stack.ptr + 8 * z * n, stack.shape[1:], owned=False, keepalive=stack)
        U1 = c.empty((k, nv, no))
        X = c.empty((k, nv, no, nv, no))        # X[z,(a,j),(b,i)] = u2_z[a,b,i,j]                   (symmetric matrices)
        Tt = c.empty((k, nv, no, nv, no))       # Tt[z,(a,i),(b,j)] = 2 u2_z[a,b,i,j] - u2_z[b,a,i,j]  (symmetric matrices)
        for z in range(k):
            part(U1, z).copy_from(u1s[z])
            c.pair_layouts(u2s[z], part(X, z), part(Tt, z))
        # ---- singles (eom_ccsd.py:268-310), ut[a,b,i,j] = Tt[(a,i),(b,j)] --------------------------------------------------
        S1 = c.contract("zck,ckai->zai", U1, self.W1)
        c.contract("ac,zci->zai", self.Gvv_s, U1, out=S1, beta=1.0, batch="z")
        c.contract("zak,ki->zai", U1, self.Goo_s, out=S1, beta=1.0)
        c.contract("zaibj,bj->zai", Tt, self.fovT, out=S1, beta=1.0)
        c.contract("zajbk,jkib->zai", Tt, V["ijka"], out=S1, alpha=-1.0, beta=1.0)
        c.contract("jabc,zbjci->zai", V["iabc"], Tt, out=S1, beta=1.0, batch="z")        # (z as a batch: Tt is read in place)
        # ---- (ov)^3 products, transposed: only Dx + Dx^T and Dd + Dd^T enter (:377), X and Tt are symmetric matrices ----------
        DxT = c.contract("zajdl,bidl->zajbi", X, self.MDU)                        # (MDU . u2x)^T per vector
        DdT = c.permute("zajbi->zajbi", DxT, alpha=0.5)                           # same memory layout as "zaibj"
        c.contract("zaidl,bjdl->zaibj", Tt, self.M12, out=DdT, alpha=0.5, beta=1.0)
        # ---- one-index dressings ---------------------------------------------------------------------------------------------
        Xoo = c.contract("klid,zdl->zki", V["ijka"], U1, alpha=-2.0)
        c.contract("kldi,zdl->zki", V["ijak"], U1, out=Xoo, beta=1.0)
        c.contract("kd,zdi->zki", self.fov, U1, out=Xoo, alpha=-1.0, beta=1.0, batch="z")
        c.contract("kldc,zdlci->zki", V["ijab"], X, out=Xoo, alpha=-2.0, beta=1.0, batch="z")      # u2[d,c,i,l] = X[(d,l),(c,i)]
        c.contract("kldc,zdicl->zki", V["ijab"], X, out=Xoo, beta=1.0, batch="z")                  # u2[d,c,l,i] = X[(d,i),(c,l)]
        c.contract("zki,akbj->zaibj", Xoo, self.Td, out=DdT, beta=1.0, batch="za")
        Xvv = c.contract("ladc,zdl->zac", V["iabc"], U1, alpha=2.0)
        c.contract("lacd,zdl->zac", V["iabc"], U1, out=Xvv, alpha=-1.0, beta=1.0)
        c.contract("zal,lc->zac", U1, self.fov, out=Xvv, alpha=-1.0, beta=1.0)
        c.contract("lkcd,zakdl->zac", V["ijab"], X, out=Xvv, alpha=-2.0, beta=1.0, batch="z")      # u2[a,d,l,k] = X[(a,k),(d,l)]
        c.contract("lkcd,zdkal->zac", V["ijab"], X, out=Xvv, beta=1.0, batch="z")                  # u2[d,a,l,k] = X[(d,k),(a,l)]
        D = c.contract("zac,cbij->zabij", Xvv, T)
        c.contract("abdj,zdi->zabij", self.WAt, U1, out=D, beta=1.0, batch="zab")
        c.contract("abdi,zdj->zabij", self.W3t, U1, out=D, alpha=-1.0, beta=1.0, batch="zab")
        c.contract("zal,lbij->zabij", U1, self.A346, out=D, beta=1.0)
        c.contract("abic,zcj->zabij", V["abic"], U1, out=D, beta=1.0, batch="z")
        npp = self.L.shape[0]
        Lall = c.empty((k, npp, no * no))
        c.ladder_sym_multi(u2s, Lall, dressed=self.dressed)                                                   # :383, all vectors
        out = []
        S2 = c.empty(D.shape) if out2 is None else None
        B5s = []
        for z in range(k):
            Dz, u2, u1 = part(D, z), u2s[z], part(U1, z)
            c.contract("ad,dbij->abij", self.Gvv, u2, out=Dz, beta=1.0)
            c.contract("li,ablj->abij", self.Goo, u2, out=Dz, beta=1.0, batch="ab")
            B5 = c.contract("klid,dj->klij", V["ijka"], u1)
            B5s.append(c.permute("klij->klij", B5))
            c.permute("lkji->klij", B5, out=B5s[z], beta=1.0)
        # the hole-ladder-shaped terms of all vectors in batched launches (the shared side — V_klij + V_klcd T_cdij, then T —
        # packed once)
        c.hole_ladder_packed_multi(u2s, [self.B2] * k, Lall)                            # :380, :382
        c.hole_ladder_packed_multi([T] * k, B5s, Lall, ys=u2s)                          # :381 (+ the symmetrised u1 term)
        for z in range(k):
            s2 = part(S2, z) if out2 is None else out2[z]
            c.symmetrised_assemble(part(D, z), part(DdT, z), part(DxT, z), s2, L=part(Lall, z))            # :377 + unpacking
            s1 = part(S1, z) if out1 is None else out1[z].copy_from(part(S1, z))
            out.append((s1, s2))
        return out


class EOM_CCSD:
    def __init__(self, no, n_excit=3, device=0):
        self.algo_name = "EOM-CCSD"
        self.no = no
        self.n_excit = n_excit
        self.u_singles = []
        self.u_doubles = []
        self.e_excit = np.zeros(n_excit)
        self.max_dim = n_excit * 4
        self.e_epsilon = 1.e-8
        self.max_iter = 500
        self.device = device

    def write_logging_info(self):
        return

    # ---- device plumbing ------------------------------------------------------------------
    def _context(self, dict_t_V, nv):
        ctx = Context(self.no, nv, device=self.device)
        for name in BLOCK_NAMES:
            blk = dict_t_V.get(name)
            if blk is not None:
                ctx.set_V_block(name, np.ascontiguousarray(blk, dtype=np.float64))
        return ctx

    def solve(self, t_fock_dressed_pq, dict_t_V_dressed, t_T_abij):
        """eom_ccsd.py:46-167.

        Call forms: the reference's (dressed Fock matrix, dictionary of dressed host blocks, host T2) — a context is built,
        the blocks are uploaded, and the context dies with the call —, or the device-resident hand-over from a CCSD solve
        (``DressedDeviceIntegrals`` from ``CCSD.get_T1_dressed_V(t1, DeviceIntegrals)``, T2 as a DeviceArray of the same
        context; the Fock matrix may be a host array or a DeviceArray): nothing crosses PCIe but n^2 numbers of the Fock
        matrix before the loop and the subspace matrices inside it.

        The subspace bookkeeping is the reference's (QR of the trial vectors :91, B = U^T sigma(U) :103-109, eig :112,
        collapse at 4 n_excit vectors :122-133, expansion by (W v - e U v) / (e - D_ai[guess] + 1e-5) :135-147), evaluated
        incrementally: sigma is linear and the orthonormalisation leaves the vectors it already made orthonormal unchanged,
        so each pass builds sigma for the n_excit NEW vectors only and extends B by their rows and columns; after a
        collapse U v, sigma(U v) = W v.  (The reference rebuilds all <= 4 n_excit sigma vectors every pass, :95-101.)
        ``self.reuse_sigma = False`` (or PYMES_EOM_REBUILD_ALL=1) restores that schedule."""
        print_title("EOM-CCSD Solver", )
        time_init = time.time()
        no = self.no
        device_form = isinstance(dict_t_V_dressed, DressedDeviceIntegrals)
        if isinstance(t_fock_dressed_pq, DeviceArray):
            t_fock_dressed_pq = t_fock_dressed_pq.get()
        f = np.asarray(t_fock_dressed_pq, dtype=np.float64)
        eps_i, eps_a = f.diagonal()[:no], f.diagonal()[no:]
        nv = eps_a.shape[0]
        D_ai = -(eps_i[None, :] - eps_a[:, None]).ravel()
        lowest_ex_ind_init = np.argsort(D_ai)[:self.n_excit]
        if device_form:
            ctx = dict_t_V_dressed.ctx
            if ctx.no != no or ctx.nv != nv:
                raise ValueError("the integrals' context does not match (no, nv) of the Fock matrix")
            t2 = t_T_abij if isinstance(t_T_abij, DeviceArray) else ctx.array(t_T_abij)
            if isinstance(t_T_abij, DeviceArray) and t_T_abij.ctx is not ctx:
                raise ValueError("t_T_abij lives in another context than the dressed integrals")
            dict_t_V_dressed.require(_Sigma.BLOCKS)      # a subset dressing / a later dressing on the same context: refuse
        else:
            ctx = self._context(dict_t_V_dressed, nv)
            t2 = ctx.array(t_T_abij)
        from pymes_amd.solver.ccd import quiet_collector
        collector = quiet_collector().__enter__()
        reuse = getattr(self, "reuse_sigma", True) and not os.environ.get("PYMES_EOM_REBUILD_ALL")
        self.history = []
        self.timings = {"hoist_s": 0.0, "sigma_s": 0.0, "orth_s": 0.0, "subspace_s": 0.0, "sigma_vectors": 0, "passes": 0}
        tm, timed = self.timings, bool(getattr(self, "profile_phases", False))

        self.pass_log = []         # with profile_phases: one record per pass (subspace dimension, new vectors, phase times)
        cur = {}

        def lap(key, t0):          # per-phase wall time (with a device synchronisation) only when asked for
            if timed:
                ctx.sync()
                dt = time.perf_counter() - t0
                tm[key] += dt
                cur[key] = cur.get(key, 0.0) + dt
            return time.perf_counter()
        try:
            t0 = time.perf_counter()
            sig = _Sigma(ctx, f, t2, dressed=device_form)
            t0 = lap("hoist_s", t0)
            print_logging_info("Initialising u tensors...", level=1)
            lay = self._layout(no, nv)
            n1, off2, nflat = lay
            # (a combination of vectors with a zero pad has a zero pad; a sigma vector gets its parts written one by one)
            fresh = lambda: ctx.empty((nflat,))
            new = []                                  # raw new trial vectors of this pass [u1 | pad | u2], flat
            for i in range(self.n_excit):
                vec = ctx.zeros((nflat,))
                one = np.zeros(n1)
                one[lowest_ex_ind_init[i]] = 1.0
                self._u1(ctx, vec, lay).set(one.reshape(nv, no))
                new.append(vec)
            us, ws, B = [], [], np.zeros((0, 0))      # orthonormal basis, its sigma vectors, U^T W
            # exchange symmetry u2_abij = u2_baji: the start vectors have it (zero doubles), and the stacked sigma build,
            # the projections and the expansions keep it exactly — tested only where that chain is broken
            all_sym = sig.many_ok
            e = self.e_excit
            e_old = self.e_excit
            e_imag = np.zeros(self.n_excit)
            diff_e_norm = np.inf
            for it in range(self.max_iter):
                time_iter_init = time.time()
                t0 = time.perf_counter()
                cur.clear()
                n_new = len(new) if (reuse or not us) else len(us) + len(new)
                if not reuse and us:                                                 # the reference's schedule: everything anew
                    new, us, ws, B = us + new, [], [], np.zeros((0, 0))
                if new:
                    new = self._orthonormalise_block(ctx, us, new, lay)              # :91
                    t0 = lap("orth_s", t0)
                    u2s = [self._u2(ctx, u, lay) for u in new]
                    sym = [True] * len(new) if all_sym else [sig.exchange_symmetric(u2) for u2 in u2s]
                    wn = [self._zero_pad(ctx, fresh(), lay) for _ in new]
                    sig.apply_many([self._u1(ctx, u, lay) for u in new], u2s, sym,                  # :95-101, new vectors only
                                   out1=[self._u1(ctx, w, lay) for w in wn], out2=[self._u2(ctx, w, lay) for w in wn])
                    tm["sigma_vectors"] += len(new)
                    t0 = lap("sigma_s", t0)
                    d0 = len(us)
                    us, ws = us + new, ws + wn
                    Bn = np.zeros((len(us), len(us)))                                # :103-109, the new rows and columns
                    Bn[:d0, :d0] = B
                    Bn[:, d0:] = ctx.gram(us, wn)
                    if d0:
                        Bn[d0:, :d0] = ctx.gram(new, ws[:d0])
                    B, new = Bn, []
                dim = len(us)
                e_old = self.e_excit                                                 # :110 (every pass, so the collapse
                with _single_threaded_blas():                                        # (a <= 12 x 12 matrix: no thread pool)
                    lam, vec = np.linalg.eig(B)                                      # :112  branch's restore is a no-op)
                pick = lam.argsort()[:self.n_excit]
                e_imag = np.imag(lam[pick])
                e = np.real(lam[pick])
                v = np.real(vec[:, pick])
                self.history.append(np.array(e))      # (the Ritz values of the pass; the reference only logs them)
                if dim >= self.max_dim:                                              # collapse :122-133
                    # The next pass of the reference orthonormalises the Ritz vectors U v (:91) and builds their sigma vectors
                    # again.  U is orthonormal, so the Gram matrix of U v is v^T v: its Cholesky factor R is known without
                    # touching a vector, U v R^-1 and sigma(U v R^-1) = W v R^-1 are ONE combination each, and the subspace
                    # matrix of the new basis is (v R^-1)^T B (v R^-1) — the pass after a collapse costs no sigma build.
                    with _single_threaded_blas():
                        Rc = np.linalg.cholesky(v.T @ v).T
                        cmat = v @ np.linalg.inv(Rc)
                    cu = [fresh() for _ in range(self.n_excit)]
                    cw = [fresh() for _ in range(self.n_excit)]
                    ctx.lincomb_multi(cu, us, cmat)
                    ctx.lincomb_multi(cw, ws, cmat)
                    B = cmat.T @ B @ cmat
                    G = ctx.gram(cu, cu)
                    if np.abs(G - np.eye(self.n_excit)).max() > self.ORTH_TOL:       # (nearly parallel Ritz vectors)
                        cu, cw = self._orthonormalise_block(ctx, [], cu, lay, shadows=cw)
                        B = ctx.gram(cu, cw)
                    us, ws = cu, cw
                    self.e_excit = e_old
                else:                                                                # expand :135-147
                    coef = np.zeros((2 * dim, self.n_excit))
                    for n in range(self.n_excit):
                        den = e[n] - D_ai[lowest_ex_ind_init[n]] + 1e-5
                        coef[:dim, n] = v[:, n] / den
                        coef[dim:, n] = -e[n] * v[:, n] / den
                    new = [fresh() for _ in range(self.n_excit)]
                    ctx.lincomb_multi(new, ws + us, coef)
                    e_old = self.e_excit
                    diff_e_norm = np.linalg.norm(self.e_excit - e)
                    self.e_excit = e
                t0 = lap("subspace_s", t0)
                tm["passes"] += 1
                if timed:
                    self.pass_log.append(dict(cur, dim=dim, new_vectors=n_new, collapse=bool(dim >= self.max_dim)))
                if diff_e_norm < self.e_epsilon:
                    print_logging_info("Iterative solver converged.", level=1)
                    print_logging_info("Norm of energy difference = {:.12f}".format(diff_e_norm), level=2)
                    for r in range(self.n_excit):
                        print_logging_info("Excited state {:d} energy = {:.12f}".format(r, e[r]), level=2)
                    print_logging_info("Excited states energies imaginary part = ", e_imag, level=2)
                    break
                print_logging_info("Iteration = ", it, level=1)
                print_logging_info("Norm of energy difference = ", diff_e_norm, level=2)
                for r in range(self.n_excit):
                    print_logging_info("Excited state {:d} energy = {:.12f}".format(r, e[r]), level=2)
                print_logging_info("Excited states energies imaginary part = ", e_imag, level=2)
                print_logging_info("Took {:.3f} seconds ".format(time.time() - time_iter_init), level=2)
            print_logging_info("EOM-CCSD finished in {:.3f} seconds".format(time.time() - time_init), level=1)
            print_logging_info("Converged excited states energies:", level=1)
            for r in range(self.n_excit):
                print_logging_info("Excited state {:d} energy = {:.12f}".format(r, e[r]), level=2)
            self.iterations = it + 1
            # (as in the reference the attributes hold the trial space as the loop left it: basis first, then the expansion
            # vectors; in the device form they stay device arrays of the caller's context)
            keep = (us + new)[:self.n_excit]
            self.u_singles = [self._u1(ctx, u, lay) for u in keep]
            self.u_doubles = [self._u2(ctx, u, lay) for u in keep]
            if not device_form:
                self.u_singles = [x.get() for x in self.u_singles]
                self.u_doubles = [x.get() for x in self.u_doubles]
            return self.e_excit
        finally:
            collector.__exit__()
            if not device_form:
                ctx.close()

    @staticmethod
    def _part(ctx, vec, offset, shape):
        return DeviceArray(ctx, vec.ptr + 8 * offset, shape, owned=False, keepalive=vec)

    # ---- flat subspace vectors [u1 (nv no) | zero pad | u2 (nv^2 no^2)], the doubles on a 256-byte boundary ----------------
    @staticmethod
    def _layout(no, nv):
        n1 = nv * no
        off2 = -(-n1 // 32) * 32
        return n1, off2, off2 + nv * nv * no * no

    def _u1(self, ctx, vec, lay):
        return self._part(ctx, vec, 0, (ctx.nv, ctx.no))

    def _u2(self, ctx, vec, lay):
        return self._part(ctx, vec, lay[1], (ctx.nv, ctx.nv, ctx.no, ctx.no))

    def _orthonormalise_block(self, ctx, us, ys, lay, shadows=None):
        """EOM_CCSD.QR (eom_ccsd.py:512-541) for a trial space [us | ys] whose leading vectors ``us`` are orthonormal already
        (Householder QR leaves those as they are, up to a sign the Rayleigh-Ritz step does not see): the block ``ys`` is
        projected against ``us`` and orthonormalised in itself by rounds of block Gram-Schmidt in its Pythagorean form
        — ONE Gram product [us | ys]^T ys (every vector read once), the Cholesky factor of ys^T ys - P^T P on the host, ONE
        multi-output combination (ys - us P) R^-1 — so a round costs two passes over the subspace instead of a dot product
        and an update per pair of vectors; a round is repeated only when the check of the result asks for it.  Returns the
        new orthonormal block; with ``shadows`` (vectors that any linear map of ``ys`` must follow, e.g. their sigma vectors;
        only for an empty ``us``) returns (block, mapped shadows)."""
        assert shadows is None or not us
        k, d, nflat = len(ys), len(us), lay[2]
        eye = np.vstack([np.zeros((d, k)), np.eye(k)])
        for rnd in range(4):
            G = ctx.gram(us + ys, ys)
            # a-posteriori check = the Gram product the next round needs anyway: expansion vectors are residuals, orthogonal
            # to the basis up to rounding (U^T (W v - e U v) = B v - e v), and one round brings them to the unit matrix
            if rnd > 0 and np.abs(G - eye).max() <= self.ORTH_TOL:
                break
            P, S = G[:d], G[d:] - G[:d].T @ G[:d]
            S = 0.5 * (S + S.T)
            scale = np.sqrt(np.abs(np.diag(S)))
            ok = bool(np.all(np.isfinite(scale)) and np.all(scale > 0.0) and np.all(np.diag(S) > 0.0))
            if ok:
                try:
                    with _single_threaded_blas():
                        Lc = np.linalg.cholesky(S / np.outer(scale, scale))          # equilibrated: S = D L L^T D
                        ok = bool(np.diag(Lc).min() > 1e-7)                          # (condition number of ys below ~1e7)
                        Rinv = np.linalg.inv(Lc.T * scale[None, :]) if ok else None  # R = L^T D, ys_new = ys' R^-1
                except np.linalg.LinAlgError:
                    ok = False
            if not ok or rnd == 3:       # (numerically) dependent new vectors: vector by vector, null vectors replaced
                return self._orthonormalise_sequential(ctx, us, ys, lay, shadows)
            coef = np.vstack([-P @ Rinv, Rinv])
            out = [ctx.empty((nflat,)) for _ in range(k)]
            ctx.lincomb_multi(out, us + ys, coef)
            if shadows is not None:
                sh = [ctx.empty((nflat,)) for _ in range(k)]
                ctx.lincomb_multi(sh, shadows, Rinv)
                shadows = sh
            ys = out
        return ys if shadows is None else (ys, shadows)

    ORTH_TOL = 1e-13       # max |U^T U - 1| accepted for the trial space (numpy's Householder QR: ~1e-15)

    @staticmethod
    def _zero_pad(ctx, vec, lay):
        if lay[1] > lay[0]:
            DeviceArray(ctx, vec.ptr + 8 * lay[0], (lay[1] - lay[0],), owned=False, keepalive=vec).zero_()
        return vec

    def _orthonormalise_sequential(self, ctx, us, ys, lay, shadows=None):
        """The fall-back of ``_orthonormalise_block``: modified Gram-Schmidt with re-orthogonalisation, one vector at a time
        (one Gram product and one combination per sweep).  A vector that vanishes against the others — the reference's
        Householder QR would return an arbitrary unit vector orthogonal to them — is replaced by a seeded random,
        exchange-symmetric direction."""
        done, sh_done = list(us), []
        n1, off2, nflat = lay
        rng = np.random.default_rng(len(us) + 1000 * len(ys))
        for z, y in enumerate(ys):
            q = ctx.empty((nflat,)).copy_from(y)
            sh = None if shadows is None else ctx.empty((nflat,)).copy_from(shadows[z])
            for attempt in range(3):
                nrm0 = np.sqrt(ctx.gram([q], [q])[0, 0])
                for _ in range(2):
                    if done:
                        proj = ctx.gram(done, [q])[:, 0]
                        ctx.lincomb_multi([q], done, -proj[:, None], beta=[1.0])
                        if sh is not None:
                            ctx.lincomb_multi([sh], sh_done, -proj[len(us):, None], beta=[1.0])
                nrm = np.sqrt(ctx.gram([q], [q])[0, 0])
                if np.isfinite(nrm) and nrm > 1e-12 * max(nrm0, 1e-300) and nrm > 0.0:
                    break
                if shadows is not None:
                    raise np.linalg.LinAlgError("linearly dependent Ritz vectors in the Davidson collapse")
                r1 = rng.standard_normal((ctx.nv, ctx.no))
                r2 = rng.standard_normal((ctx.nv, ctx.nv, ctx.no, ctx.no))
                q.zero_()
                self._u1(ctx, q, lay).set(r1)
                self._u2(ctx, q, lay).set(r2 + r2.transpose(1, 0, 3, 2))
            ctx.lincomb_multi([q], [], np.zeros((0, 1)), beta=[1.0 / nrm])
            if sh is not None:
                ctx.lincomb_multi([sh], [], np.zeros((0, 1)), beta=[1.0 / nrm])
                sh_done.append(sh)
            done.append(q)
        block = done[len(us):]
        return block if shadows is None else (block, sh_done)

    # ---- the reference's host-array call forms (eom_ccsd.py:268-385) ---------------------------
    def _host_sigma(self, t_fock_pq, dict_t_V, t_u_ai, t_u_abij, t_T_abij, which):
        """Complex trial vectors (the FEAST / real-time callers, feast_eom_ccsd.py:309-350) are propagated as two
        real ones: the operator is real and linear, sigma(u + i w) = sigma(u) + i sigma(w)."""
        if np.iscomplexobj(t_fock_pq) or np.iscomplexobj(t_T_abij):
            raise TypeError("complex Fock matrix / amplitudes are not supported")
        nv = t_u_ai.shape[0]
        ctx = self._context(dict_t_V, nv)
        try:
            sig = _Sigma(ctx, t_fock_pq, ctx.array(t_T_abij))
            apply = sig.singles if which == 1 else sig.doubles
            cplx = np.iscomplexobj(t_u_ai) or np.iscomplexobj(t_u_abij)
            out = apply(ctx.array(np.real(t_u_ai)), ctx.array(np.real(t_u_abij))).get()
            if cplx:
                out = out + 1j * apply(ctx.array(np.imag(t_u_ai)), ctx.array(np.imag(t_u_abij))).get()
            return out
        finally:
            ctx.close()

    def _diag_inputs(self, dict_t_V, t_T_abij):
        """What the two diagonals read, as host arrays: V_ijab, T and four diagonal slices of other blocks.  For the
        reference's dictionary these are einsum views; for the device-resident hand-over (``DressedDeviceIntegrals``) V_ijab
        and T come down once per solve (2 o^2 v^2 numbers) and the slices — V_iabj[i,a,a,i], V_iajb[i,a,i,a], V_klij[i,j,i,j],
        V_abcd[a,b,a,b] — are gathered on the device through strided views (o v / o^2 / v^2 numbers), so neither the
        16 blocks nor V_abcd ever cross PCIe."""
        if isinstance(dict_t_V, DressedDeviceIntegrals):
            c, o, v = dict_t_V.ctx, dict_t_V.no, dict_t_V.nv
            T = t_T_abij.get() if isinstance(t_T_abij, DeviceArray) else np.asarray(t_T_abij)
            gather = lambda name, dims, strides: c.permute("xy->xy", dict_t_V[name], in_view=(dims, strides)).get()
            return {"V": dict_t_V["ijab"].get(), "T": T,
                    "iaai": gather("iabj", (v, o), (v * o + o, v * v * o + 1)),          # [a,i] <- V[i,a,a,i]
                    "iaia": gather("iajb", (v, o), (o * v + 1, v * o * v + v)),          # [a,i] <- V[i,a,i,a]
                    "ijij": gather("klij", (o, o), (o * o * o + o, o * o + 1)),          # [i,j] <- V[i,j,i,j]
                    "abab": gather("abcd", (v, v), (v * v * v + v, v * v + 1))}          # [a,b] <- V[a,b,a,b]
        return {"V": np.asarray(dict_t_V["ijab"]), "T": np.asarray(t_T_abij),
                "iaai": np.einsum("iaai->ai", dict_t_V["iabj"]), "iaia": np.einsum("iaia->ai", dict_t_V["iajb"]),
                "ijij": np.einsum("ijij->ij", dict_t_V["klij"]), "abab": np.einsum("abab->ab", dict_t_V["abcd"])}

    def get_diag_singles(self, t_fock_pq, dict_t_V, t_T_abij, _inputs=None):
        """eom_ccsd.py:169-198, terms grouped: with V~ = 2V - V^(ab), T~ = 2T - T^(ab) the four (a,i)-resolved
        V.T terms are one Hadamard sum.  O(o^2 v^2) work on host arrays (preconditioner data, once per solve)."""
        no = self.no
        g = _inputs or self._diag_inputs(dict_t_V, t_T_abij)
        V, T = g["V"], g["T"]
        f = t_fock_pq.get() if isinstance(t_fock_pq, DeviceArray) else np.asarray(t_fock_pq)
        Vt = 2.0 * V - V.transpose(0, 1, 3, 2)
        Tt = 2.0 * T - T.transpose(1, 0, 2, 3)
        d = f.diagonal()[no:][:, None] - f.diagonal()[:no][None, :]
        d = d + 2.0 * g["iaai"] - g["iaia"]
        d = d + np.einsum("jiba,baji->ai", Vt, Tt)
        d = d - np.einsum("jkba,abjk->a", Vt, T)[:, None]
        d = d - np.einsum("jicb,bcji->i", V, T)[None, :]
        return d

    def get_diag_doubles(self, t_fock_pq, dict_t_V, t_T_abij, _inputs=None):
        """eom_ccsd.py:200-266 (same grouping; the reference's placement of the `ibib` term on the (a,i) axes is kept)."""
        no = self.no
        g = _inputs or self._diag_inputs(dict_t_V, t_T_abij)
        V, T = g["V"], g["T"]
        f = t_fock_pq.get() if isinstance(t_fock_pq, DeviceArray) else np.asarray(t_fock_pq)
        Vx = V.transpose(0, 1, 3, 2)                         # V[k,i,a,c] read as [k,i,c,a]
        Tx = T.transpose(1, 0, 2, 3)                         # T[a,c,k,i] read as [c,a,k,i]
        ai = f.diagonal()[no:][:, None] - f.diagonal()[:no][None, :]
        ai = ai + g["iaai"] - 2.0 * g["iaia"]
        ai = ai + np.einsum("kica,caki->ai", 2.0 * V - 2.0 * Vx, T) + np.einsum("kica,caki->ai", Vx - 2.0 * V, Tx)
        ai = ai + np.einsum("kicb,acki->ai", V, T)
        a_ = np.einsum("klca,cakl->a", V, Tx - 2.0 * T)
        i_ = np.einsum("kicd,cdki->i", Vx - 2.0 * V, T)
        d = ai[:, None, :, None] + a_[:, None, None, None] + i_[None, None, :, None]
        d = d - 2.0 * np.einsum("kjab,abkj->abj", V, T)[:, :, None, :]
        d = d - 2.0 * np.einsum("ijcb,cbij->ij", V, T)[None, None, :, :]
        d = d + np.einsum("kiab,abkj->abij", V, T) + np.einsum("ijca,cbij->abij", V, T)
        d = d + np.einsum("kjac,caki->aij", V, T)[:, None, :, :] + np.einsum("kjac,ackj->aj", V, T)[:, None, None, :]
        d = d + d.transpose(1, 0, 3, 2)                                                           # P(ijab, jiba), :253
        d = d + (g["ijij"] + np.einsum("ijcd,cdij->ij", V, T))[None, None, :, :]
        d = d + (np.einsum("klab,abkl->ab", V, T) + g["abab"])[:, :, None, None]
        return d

    def update_singles(self, t_fock_pq, dict_t_V, t_u_ai, t_u_abij, t_T_abij):
        return self._host_sigma(t_fock_pq, dict_t_V, t_u_ai, t_u_abij, t_T_abij, 1)

    def update_doubles(self, t_fock_pq, dict_t_V, t_u_ai, t_u_abij, t_T_abij):
        return self._host_sigma(t_fock_pq, dict_t_V, t_u_ai, t_u_abij, t_T_abij, 2)
