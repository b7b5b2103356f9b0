"""numpy oracle of the closed-shell CCD/DCD/CCSD/DCSD amplitude-update path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Independent restatement of
nickirk/pymes @ 2024_10_08; each function cites the reference lines it follows
(paths relative to the reference root).  Pinned against the imported reference
by oracle/make_golden.py -> tests/golden/*.npz.

Conventions (identical to the reference): V[p,q,r,s] = <pq|rs>, C-contiguous
fp64, T2 is [a,b,i,j] (nv,nv,no,no), T1 is [a,i].  Block names spell the index
types position by position: i,j,k,l occupied; a,b,c,d virtual.
"""
import itertools
import time

import numpy as np

OCC = "ijkl"
VIRT = "abcd"

# the 16 names used by pymes/integral/partition.py:4-39, one per occ/virt pattern
BLOCK_NAMES = ("abci", "iabj", "iajk", "aijk", "klij", "aibj", "ijak", "abic",
               "iajb", "abcd", "iabc", "aijb", "ijka", "aibc", "ijab", "abij")


def _pattern(name):
    return "".join("o" if ch in OCC else "v" for ch in name)


NAME_OF_PATTERN = {_pattern(nm): nm for nm in BLOCK_NAMES}
assert len(NAME_OF_PATTERN) == 16


def split_blocks(no, V):
    """pymes/integral/partition.py:4-39 — the 16 occ/virt block views of V."""
    sl = {"o": slice(0, no), "v": slice(no, None)}
    return {nm: V[tuple(sl[c] for c in _pattern(nm))] for nm in BLOCK_NAMES}


def denominators(eps_o, eps_v, shift=0.0):
    """ccsd.py:149-156 — D_ai and D_abij (already inverted)."""
    d2 = (eps_o[None, None, :, None] + eps_o[None, None, None, :]
          - eps_v[:, None, None, None] - eps_v[None, :, None, None])
    d1 = eps_o[None, :] - eps_v[:, None]
    return 1.0 / (d1 + shift), 1.0 / (d2 + shift)


def mp2(eps_o, eps_v, V_ijab, V_abij, shift=0.0):
    """pymes/solver/mp2.py:9-22 — first-order doubles and the MP2 energy."""
    _, inv_d2 = denominators(eps_o, eps_v, shift)
    T = V_abij * inv_d2
    e = 2.0 * np.einsum("abij,ijab->", T, V_ijab) - np.einsum("abij,jiab->", T, V_ijab)
    return e, T


# --------------------------------------------------------------------------
# T2 residual (pymes/solver/ccd.py:164-254)
# --------------------------------------------------------------------------
def doubles_residual(no, f, T, V_klij, V_ijab, V_abij, V_iajb, V_iabj, V_abcd,
                     is_dcd=False, ein=np.einsum):
    """Closed-shell CCD/DCD T2 residual.  ``ein`` selects the contraction
    engine: plain ``np.einsum`` is what the reference calls (ccd.py:180-240),
    ``partial(np.einsum, optimize=True)`` is the BLAS-backed best-effort mode."""
    quad = not is_dcd
    f_oo, f_vv = f[:no, :no], f[no:, no:]

    hole = V_klij.copy()                                           # :175-180
    if quad:
        hole = hole + ein("klcd,cdij->klij", V_ijab, T)
    R = V_abij + ein("klij,abkl->abij", hole, T)                   # :185-186
    R = R + ein("abcd,cdij->abij", V_abcd, T)                      # :187  ladder
    if quad:                                                       # :189-191
        R = R + ein("alcj,cbil->abij", ein("klcd,adkj->alcj", V_ijab, T), T)

    Tt = 2.0 * T - T.transpose(1, 0, 2, 3)                         # :199
    R = R + ein("acik,cbkj->abij", Tt, ein("klcd,dblj->cbkj", V_ijab, Tt))  # :202-204

    w = 1.0 if quad else 0.5                                       # :213-220
    X_vv = f_vv - w * ein("adkl,lkdc->ac", Tt, V_ijab)
    X_oo = f_oo + w * ein("cdil,lkdc->ki", Tt, V_ijab)

    Ex = ein("ac,cbij->abij", X_vv, T)                             # :231-235
    Ex = Ex - ein("ki,abkj->abij", X_oo, T)
    Ex = Ex - ein("kaic,cbkj->abij", V_iajb, T)
    Ex = Ex - ein("kbic,ackj->abij", V_iajb, T)
    Ex = Ex + ein("acik,kbcj->abij", Tt, V_iabj)
    if quad:                                                       # :237-240
        Y = ein("klcd,daki->alci", V_ijab, T)
        Ex = Ex - ein("alci,cblj->abij", Y, T) + ein("alci,bclj->abij", Y, T)
    return R + Ex + Ex.transpose(1, 0, 3, 2)                       # :249-252


def ccd_energy(T, V_ijab):
    """ccd.py:256-262 — (direct, exchange) parts."""
    return (2.0 * np.einsum("abij,ijab->", T, V_ijab),
            -1.0 * np.einsum("abij,ijba->", T, V_ijab))


# --------------------------------------------------------------------------
# T1 dressing (pymes/solver/ccsd.py:226-421)
# --------------------------------------------------------------------------
# (target block, coefficient, subscripts, operands); "t" = T1, f?? = blocks of the
# UNDRESSED Fock matrix, other names = undressed V blocks.  ccsd.py:257-286.
_FOCK_TERMS = (
    ("ov", +2.0, "bj,jabi->ia", ("t", "iabj")),
    ("ov", -1.0, "bj,jiab->ia", ("t", "ijab")),
    ("vo", -1.0, "ji,aj->ai", ("foo", "t")),
    ("vo", +1.0, "ab,bi->ai", ("fvv", "t")),
    ("vo", -1.0, "jb,bi,aj->ai", ("fov", "t", "t")),
    ("vo", +2.0, "bj,jabi->ai", ("t", "iabj")),
    ("vo", -2.0, "bj,jkbi,ak->ai", ("t", "ijak", "t")),
    ("vo", +2.0, "bj,jabc,ci->ai", ("t", "iabc", "t")),
    ("vo", -2.0, "bj,jkbc,ci,ak->ai", ("t", "ijab", "t", "t")),
    ("vo", -1.0, "bj,jaib->ai", ("t", "iajb")),
    ("vo", +1.0, "bj,jkib,ak->ai", ("t", "ijka", "t")),
    ("vo", -1.0, "bj,jacb,ci->ai", ("t", "iabc", "t")),
    ("vo", +1.0, "bj,jkcb,ci,ak->ai", ("t", "ijab", "t", "t")),
    ("oo", +2.0, "ck,kicj->ij", ("t", "ijak")),
    ("oo", -1.0, "ck,kijc->ij", ("t", "ijka")),
    ("oo", +1.0, "ib,bj->ij", ("fov", "t")),
    ("oo", +2.0, "ck,kicb,bj->ij", ("t", "ijab", "t")),
    ("oo", -1.0, "ck,kibc,bj->ij", ("t", "ijab", "t")),
    ("vv", +2.0, "ci,iacb->ab", ("t", "iabc")),
    ("vv", -1.0, "ci,iabc->ab", ("t", "iabc")),
    ("vv", -1.0, "ib,ai->ab", ("fov", "t")),
    ("vv", -2.0, "ck,klcb,al->ab", ("t", "ijab", "t")),
    ("vv", +1.0, "ck,kibc,ai->ab", ("t", "ijab", "t")),
)


def dressed_fock(no, f, t1, Vb):
    """ccsd.py:226-288 — exp(-T1) f exp(T1) block by block, term by term."""
    sl = {"o": slice(0, no), "v": slice(no, None)}
    env = dict(Vb)
    env.update(t=t1, foo=f[:no, :no], fvv=f[no:, no:], fov=f[:no, no:])
    out = f.copy()
    for blk, coef, spec, names in _FOCK_TERMS:
        out[sl[blk[0]], sl[blk[1]]] += coef * np.einsum(spec, *[env[n] for n in names], optimize=True)
    return out


# blocks the reference produces (ccsd.py:322-419); the other 5 names stay None (:317)
DRESSED_KEYS = ("abij", "klij", "ijab", "ijka", "ijak", "iajb", "iabj", "iabc", "abic",
                "iajk", "abcd")


def dressed_block(key, t1, Vb):
    """One block of exp(-T1) V exp(T1) (ccsd.py:322-419).

    Restated compactly (checked against the reference's
    spelled-out terms on random unsymmetric V by make_golden.py): a bra index
    that is virtual in the target picks up  -t[a,k] x (occupied k source);  a
    ket index that is occupied in the target picks up  +(virtual c source) x
    t[c,i];  bra-occupied and ket-virtual indices are untouched.  Sources are
    always UNDRESSED blocks."""
    pat = _pattern(key)
    out_sub = "pqrs"
    choices = []
    for pos, kind in enumerate(pat):
        if pos < 2 and kind == "v":
            choices.append(("same", "mix"))
        elif pos >= 2 and kind == "o":
            choices.append(("same", "mix"))
        else:
            choices.append(("same",))
    total = None
    for combo in itertools.product(*choices):
        src_pat, src_sub, ops, subs, sign = "", "", [], [], 1.0
        for pos, how in enumerate(combo):
            if how == "same":
                src_pat += pat[pos]
                src_sub += out_sub[pos]
            else:
                dummy = "wxyz"[pos]
                src_sub += dummy
                ops.append(t1)
                if pos < 2:            # bra virtual <- occupied source, -t[a,k]
                    src_pat += "o"
                    subs.append(out_sub[pos] + dummy)
                    sign = -sign
                else:                  # ket occupied <- virtual source, +t[c,i]
                    src_pat += "v"
                    subs.append(dummy + out_sub[pos])
        src = Vb[NAME_OF_PATTERN[src_pat]]
        term = sign * np.einsum(",".join([src_sub] + subs) + "->" + out_sub, src, *ops,
                                optimize=True)
        total = term if total is None else total + term
    return total


def dressed_V(t1, Vb, keys=None):
    """ccsd.py:290-421.  ``keys`` mirrors the optional third argument (:316-317)."""
    if not keys:
        out = dict.fromkeys(Vb, None)
        keys = DRESSED_KEYS
    else:
        out = dict.fromkeys(keys, None)
    for k in keys:
        if k in DRESSED_KEYS:
            out[k] = dressed_block(k, t1, Vb)
    return out


def singles_residual(no, f_dressed, t1, T, Vb):
    """ccsd.py:423-438 — T1 residual from the dressed Fock and UNDRESSED V."""
    ein = lambda *a: np.einsum(*a, optimize=True)
    Tt = 2.0 * T - T.transpose(0, 1, 3, 2)
    r = f_dressed[no:, :no].copy()
    r += ein("jb,abij->ai", f_dressed[:no, no:], Tt)
    r += ein("ajbc,bcij->ai", Vb["aibc"], Tt)
    r -= ein("kjbc,ak,bcij->ai", Vb["ijab"], t1, Tt)
    r -= ein("jkib,abjk->ai", Vb["ijka"], Tt)
    r -= ein("jkcb,ci,abjk->ai", Vb["ijab"], t1, Tt)
    return r


def ccsd_doubles_residual(no, f_dressed, T, Vd, is_dcsd=False, ein=np.einsum):
    """ccsd.py:440-456 — forwards six dressed blocks to the CCD residual."""
    return doubles_residual(no, f_dressed, T, Vd["klij"], Vd["ijab"], Vd["abij"], Vd["iajb"],
                            Vd["iabj"], Vd["abcd"], is_dcd=is_dcsd, ein=ein)


def ccsd_energy(f_ov, t1, T, V_ijab):
    """ccsd.py:458-466 — (one-body, direct, exchange)."""
    tau = T + np.einsum("ai,bj->abij", t1, t1)
    return (2.0 * np.einsum("ia,ai->", f_ov, t1),
            2.0 * np.einsum("abij,ijab->", tau, V_ijab),
            -1.0 * np.einsum("abij,ijba->", tau, V_ijab))


# --------------------------------------------------------------------------
# DIIS (pymes/mixer/diis.py:16-112)
# --------------------------------------------------------------------------
class Diis:
    """Pulay mixing with the reference's exact bookkeeping, including its
    full-subspace quirk: once ``dim`` vectors are stored, the shifted copy of
    the old overlap matrix (diis.py:59-60) leaves out the row/column of the
    second-newest vector, so those overlaps become zeros."""

    def __init__(self, dim=6):
        self.dim = dim
        self.L = np.zeros((1, 1))
        self.errs, self.amps = [], []
        self.last_coeff = None

    def push_overlaps(self, overlaps_with_new, was_full):
        """Pure bookkeeping on L given <e_i, e_new> for the m stored vectors."""
        m = len(overlaps_with_new)
        L = np.zeros((m + 1, m + 1))
        L[m, :m] = -1.0
        L[:m, m] = -1.0
        if was_full:
            L[:m - 2, :m - 2] = self.L[1:m - 1, 1:m - 1]           # diis.py:60
        else:
            L[:m - 1, :m - 1] = self.L[:m - 1, :m - 1]             # diis.py:62
        L[:m, m - 1] += overlaps_with_new                          # :65-78
        L[m - 1, :] = L[:, m - 1]                                  # :80
        self.L = L
        return L

    @staticmethod
    def coefficients(L):
        """diis.py:83-95."""
        rhs = np.zeros(L.shape[0])
        rhs[-1] = -1.0
        lam, vec = np.linalg.eigh(L)
        if np.any(np.abs(lam) < 1e-12):
            keep = np.abs(lam) > 1e-12
            return (vec[:, keep] / lam[keep]) @ (vec[:, keep].T.conj() @ rhs)
        return np.linalg.inv(L) @ rhs

    def mix(self, err, amp):
        was_full = len(self.errs) == self.dim
        if was_full:
            self.errs.pop(0)
            self.amps.pop(0)
        self.errs.append(err)
        self.amps.append(amp)
        ov = np.array([sum(np.real(np.vdot(e_old[k], err[k])) for k in range(len(err)))
                       for e_old in self.errs])
        c = self.coefficients(self.push_overlaps(ov, was_full))
        self.last_coeff = c
        return [sum(c[a] * self.amps[a][k] for a in range(len(self.errs)))
                for k in range(len(amp))]


# --------------------------------------------------------------------------
# iteration drivers
# --------------------------------------------------------------------------
def ccd_solve(no, f, V, is_dcd=False, is_diis=True, delta_e=1e-8, max_iter=50,
              level_shift=0.0, amps=None, mixer=None, ein=np.einsum):
    """pymes/solver/ccd.py:24-162 (Brueckner / dr-CCD branches not restated)."""
    eps_o, eps_v = f.diagonal()[:no].copy(), f.diagonal()[no:].copy()
    Vb = split_blocks(no, V)
    e_mp2, T = mp2(eps_o, eps_v, Vb["ijab"], Vb["abij"], level_shift)
    if amps is not None:
        T = amps
    _, inv_d2 = denominators(eps_o, eps_v, level_shift)
    mixer = mixer or (Diis(6) if is_diis else None)
    dE, e_last, it, hist, e = abs(e_mp2), e_mp2, 0, [], 0.0
    dT = None
    while abs(dE) > delta_e and it <= max_iter:
        it += 1
        R = doubles_residual(no, f, T, Vb["klij"], Vb["ijab"], Vb["abij"], Vb["iajb"],
                             Vb["iabj"], Vb["abcd"], is_dcd=is_dcd, ein=ein)
        dT = R * inv_d2
        T = T + dT
        if mixer is not None:
            T = mixer.mix([dT], [T])[0]
        e_dir, e_ex = ccd_energy(T, Vb["ijab"])
        e = float(np.real(e_dir + e_ex))
        dE, e_last = e - e_last, e
        hist.append((e, np.linalg.norm(T), np.linalg.norm(dT)))
    return {"e": e, "e_mp2": float(e_mp2), "t2": T, "dE": dE, "history": hist, "iterations": it}


def ccsd_solve(no, f, V, is_dcsd=False, is_diis=True, delta_e=1e-8, max_iter=50,
               level_shift=0.0, amps=None, mixer=None, ein=np.einsum, timings=None):
    """pymes/solver/ccsd.py:47-224."""
    nv = f.shape[0] - no
    eps_o, eps_v = f.diagonal()[:no].copy(), f.diagonal()[no:].copy()
    Vb = split_blocks(no, V)
    e_mp2, T2 = mp2(eps_o, eps_v, Vb["ijab"], Vb["abij"], level_shift)
    T1 = np.zeros((nv, no))
    if amps is not None:
        T1, T2 = amps
    inv_d1, inv_d2 = denominators(eps_o, eps_v, level_shift)
    mixer = mixer or (Diis(6) if is_diis else None)
    dE, e_last, it, hist, e = abs(e_mp2), e_mp2, 0, [], 0.0
    while abs(dE) > delta_e and it <= max_iter:
        it += 1
        t0 = time.time()
        fd = dressed_fock(no, f, T1, Vb)                           # :163
        Vd = dressed_V(T1, Vb)                                     # :165
        R1 = singles_residual(no, fd, T1, T2, Vb)                  # :167
        R2 = ccsd_doubles_residual(no, fd, T2, Vd, is_dcsd, ein)   # :171
        dT1, dT2 = R1 * inv_d1, R2 * inv_d2                        # :176-177
        T1, T2 = T1 + dT1, T2 + dT2
        if mixer is not None:
            T1, T2 = mixer.mix([dT1, dT2], [T1, T2])               # :181-183
        e1, ed, ex = ccsd_energy(f[:no, no:], T1, T2, Vb["ijab"])  # :189-192
        e = float(np.real(e1 + ed + ex))
        dE, e_last = e - e_last, e
        hist.append((e, np.linalg.norm(T2), np.linalg.norm(dT2)))
        if timings is not None:
            timings.append(time.time() - t0)
    return {"e": e, "e_mp2": float(e_mp2), "t1": T1, "t2": T2, "dE": dE, "history": hist,
            "iterations": it}
