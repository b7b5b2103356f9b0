"""numpy oracle of the 3D uniform-electron-gas integral builder (pymes/model/ueg.py,
pymes/basis_set/planewave.py).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Vectorised restatement (per (p,r) pair, all q at once; the k'-lattice sum is cached per
momentum transfer).  Pinned against the imported reference by oracle/make_golden_ueg.py.
"""
import numpy as np


class Ueg:
    def __init__(self, n_ele, rs):
        """ueg.py:16-80."""
        self.n_ele, self.rs = int(n_ele), rs
        self.L = rs * ((4 * np.pi * self.n_ele) / 3) ** (1.0 / 3.0)
        self.Omega = self.L ** 3
        self.k_cutoff = None
        self.gamma = None
        self.correlator = "trunc"          # name of the correlator u(k^2) (ueg.py:740-984)
        self._lattice = None
        self._nabla_cache = {}

    # ---- basis (ueg.py:128-172, planewave.py:12-26) -------------------------------------
    def init_basis(self, cutoff):
        self.cutoff = cutoff
        self.imax = int(np.ceil(np.sqrt(cutoff))) + 1
        limit = cutoff * (2 * np.pi / self.L) ** 2 / 2.0
        ks, kin = [], []
        r = range(-self.imax, self.imax + 1)
        for i in r:
            for j in r:
                for k in r:
                    kp = (np.array([i, j, k]) + np.zeros(3)) * 2 * np.pi / self.L
                    e = np.dot(kp, kp) / 2.0          # the exact float expression of planewave.py:19
                    if e <= limit:
                        ks.append((i, j, k))
                        kin.append(e)
        order = np.argsort(np.array(kin), kind="stable")       # list.sort() on BasisFunc.__lt__ is stable
        self.k_int = np.array(ks)[order]
        self.kinetic = np.array(kin)[order]
        self.kp = self.k_int * 2 * np.pi / self.L
        self.n_p = len(order)
        m = 2 * self.imax + 1
        self.index_of = -np.ones(m ** 3, dtype=int)            # ueg.py:104-125
        loc = m * m * (self.k_int[:, 0] + self.imax) + m * (self.k_int[:, 1] + self.imax) + self.k_int[:, 2] + self.imax
        self.index_of[loc] = np.arange(self.n_p)
        return self.n_p

    # ---- correlator (ueg.py:772-800) ------------------------------------------------------
    def trunc(self, k2):
        if self.k_cutoff is None:
            self.k_cutoff = int(np.ceil(np.sqrt(self.cutoff)))
        if self.gamma is None:
            self.gamma = 1.0
        kc2 = (self.k_cutoff * 2 * np.pi / self.L) ** 2
        k2 = np.array(k2, dtype=np.float64, copy=True)
        k2[k2 <= kc2 * (1 + 0.00001)] = 0.0
        out = np.zeros_like(k2)
        np.divide(-4.0 * np.pi, k2 ** 2, out=out, where=(k2 > 1e-12))
        return out * self.gamma

    # ---- the other correlators of the reference (ueg.py:740-770, 802-984), restated branch by branch.  Each has an
    # "array" form (called with an ndarray) and a "scalar" form (called with a float; what `k.dot(k)` and
    # einsum("i,i->") produce) — the two differ AT the cut-off in gaskell / gaskell_modified, so both are kept.
    def _rho(self):
        return self.n_ele / self.Omega

    def _masked(self, num, den, keep):
        out = np.zeros_like(den, dtype=np.float64)
        np.divide(num, den, out=out, where=keep)
        return out

    def u(self, k2, scalar=False):
        """u(k^2) of the selected correlator.  Returns (values, k2 as the reference leaves it): `trunc` zeroes the
        sub-cutoff entries of an array argument in place (ueg.py:794), the others do not touch it."""
        name = self.correlator
        tw = 2 * np.pi / self.L
        if name == "trunc":
            if self.k_cutoff is None:
                self.k_cutoff = int(np.ceil(np.sqrt(self.cutoff)))
            kc2 = (self.k_cutoff * 2 * np.pi / self.L) ** 2
            k2m = np.array(k2, dtype=np.float64, copy=True)
            k2m[k2m <= kc2 * (1 + 0.00001)] = 0.0
            return self.trunc(k2), k2m
        k2 = np.array(k2, dtype=np.float64, copy=True)
        if name == "gaskell":                                               # ueg.py:836-883
            mu = np.sqrt(4.0 * np.pi / self._rho()) * (self.gamma if self.gamma is not None else 1.0)
            kf = self.kp[self.n_ele // 2]
            kf2 = kf.dot(kf)
            cut = (self.k_cutoff ** 2 if self.k_cutoff is not None else 4.0) * kf2
            if scalar:
                res = np.where((k2 < cut) & (k2 > 1e-12), mu / np.where(k2 > 1e-12, k2, 1.0), 0.0)
            else:
                res = self._masked(mu, k2, k2 > 1e-12)
                res[k2 > cut] = 0.0
            return -res, k2
        if name == "gaskell_modified":                                      # ueg.py:802-834
            cut = (self.k_cutoff * tw) ** 2 if self.k_cutoff is not None else 2.0
            mu = np.pi
            if scalar:
                inner = (k2 < cut) & (k2 > 1e-12)
                with np.errstate(divide="ignore"):
                    res = np.where(inner, 0.0, 4 * mu / k2 ** 2)
            else:
                res = self._masked(4 * mu, k2 ** 2, k2 >= cut)
            return -res, k2
        if name == "coulomb":                                               # ueg.py:905-915
            g = 1.0 if self.gamma is None else self.gamma
            return self._masked(-4.0 * np.pi * g, k2, k2 > 1e-12), k2
        if name == "yukawa":                                                # ueg.py:740-770
            g0 = np.sqrt(self._rho() / 4.0 * np.pi)
            g = g0 if self.gamma is None else self.gamma * g0
            den_cut = (self.k_cutoff * tw ** 2 + g) if self.k_cutoff is not None else 1e-12
            b = k2 + g
            return self._masked(-4.0 * np.pi, b, np.abs(b) > den_cut), k2
        if name == "stg":                                                   # ueg.py:917-935
            g = np.sqrt(4.0 * np.pi * self._rho()) if self.gamma is None else self.gamma
            den_cut = (self.k_cutoff * tw ** 2 + g ** 2) ** 2 if self.k_cutoff is not None else 1e-12
            b = (k2 + g ** 2) ** 2
            return self._masked(-4.0 * np.pi / g, b, np.abs(b) > den_cut), k2
        if name == "smooth":                                                # ueg.py:885-903
            from scipy import special
            if self.k_cutoff is None:
                self.k_cutoff = int(np.ceil(np.sqrt(self.cutoff)))
            if self.gamma is None:
                self.gamma = 0.01
            kc = np.sqrt((self.k_cutoff * tw) ** 2)
            k = np.sqrt(k2)
            num = -4.0 * np.pi * (1.0 + special.erf((k - kc) / (kc * self.gamma))) / 2.0
            return self._masked(num, k2 ** 2, k2 > (kc * self.gamma) ** 2), k2
        raise ValueError(name)

    def sum_nabla_u_square(self, dk, cutoff=30):
        """ueg.py:581-596: sum_k' (k'.(k-k')) u(k'^2) u((k-k')^2) / Omega over a (2*30+1)^3 lattice."""
        key = tuple(np.round(dk * self.L / (2 * np.pi)).astype(int))
        if key not in self._nabla_cache:
            if self._lattice is None:
                g = np.arange(-cutoff, cutoff + 1)
                self._lattice = np.array([[i, j, k] for i in g for j in g for k in g])
            k1 = 2 * np.pi * self._lattice / self.L
            k2 = dk - k1
            k1s, k2s = np.einsum("ni,ni->n", k1, k1), np.einsum("ni,ni->n", k2, k2)
            val = np.einsum("ni,ni->n", k1, k2) * self.u(k1s)[0] * self.u(k2s)[0]
            self._nabla_cache[key] = np.einsum("n->", val) / self.Omega
        return self._nabla_cache[key]

    def _occ(self):
        return self.kp[: self.n_ele // 2]

    def exchange_3b(self, pvec, kvec):
        """ueg.py:518-543.  k.k comes out of einsum(optimize=True) as a 0-d ARRAY there, so it takes the correlator's array
        branch — only the `d_k_vec.dot(d_k_vec)` of the main loop (ueg.py:409) is a scalar."""
        d = pvec - self._occ()
        return np.sum((d @ kvec) * self.u(kvec @ kvec)[0] * self.u(np.einsum("ni,ni->n", d, d))[0]) / self.Omega

    def p_k_with_q(self, pvec, kvec):
        """ueg.py:545-573."""
        v1, v2 = pvec - kvec - self._occ(), pvec - self._occ()
        return np.sum(np.einsum("ni,ni->n", v1, v2) * self.u(np.einsum("ni,ni->n", v1, v1))[0]
                      * self.u(np.einsum("ni,ni->n", v2, v2))[0]) / self.Omega

    # ---- two-body integrals (ueg.py:265-516) --------------------------------------------------
    def two_body(self, mode="coulomb"):
        """mode: 'coulomb' (correlator None), 'only_2b', 'effect_2b', 'rpa'."""
        n, m = self.n_p, 2 * self.imax + 1
        V = np.zeros((n, n, n, n))
        qs = np.arange(n)
        for p in range(n):
            for r in range(n):
                d_int = self.k_int[r] - self.k_int[p]
                dk = self.kp[r] - self.kp[p]
                ks = self.k_int - d_int                                     # k_s = k_q - (k_r - k_p)
                # like the reference (ueg.py:392-401): only the FLATTENED index is range-checked
                loc = m * m * (ks[:, 0] + self.imax) + m * (ks[:, 1] + self.imax) + ks[:, 2] + self.imax
                ok = (loc >= 0) & (loc < m ** 3)
                s = np.where(ok, self.index_of[np.clip(loc, 0, m ** 3 - 1)], -1)
                ok &= s >= 0
                if not ok.any():
                    continue
                dk2 = dk @ dk
                if mode == "coulomb":
                    w = np.full(n, 4 * np.pi / dk2 / self.Omega if abs(dk2) > 0 else 0.0)
                elif mode == "rpa":
                    w = np.full(n, -self.n_ele * dk2 * float(self.u(dk2, scalar=True)[0]) ** 2 / self.Omega / self.Omega if abs(dk2) > 0 else 0.0)
                elif mode == "only_2b":
                    u_mat = self.sum_nabla_u_square(dk)
                    if abs(dk2) > 0:
                        rs_dk = self.kp[r] - self.kp[np.clip(s, 0, n - 1)]
                        u = float(self.u(dk2, scalar=True)[0])
                        w = (4 * np.pi / dk2 + u_mat + dk2 * u - (rs_dk @ dk) * u) / self.Omega
                    else:
                        w = np.full(n, u_mat / self.Omega)
                elif mode == "effect_2b":
                    if abs(dk2) > 0:
                        val = (-self.n_ele * dk2 * float(self.u(dk2, scalar=True)[0]) ** 2 / self.Omega
                               + 2.0 * self.exchange_3b(self.kp[r], dk) - 2.0 * self.exchange_3b(self.kp[p], dk)
                               + 2.0 * self.p_k_with_q(self.kp[r], dk))
                    else:
                        val = 2.0 * self.p_k_with_q(self.kp[r], dk)
                    w = np.full(n, val / self.Omega)
                else:
                    raise ValueError(mode)
                V[p, qs[ok], r, s[ok]] = np.asarray(w)[ok] if np.ndim(w) else w
        if mode == "effect_2b":
            V = 0.5 * (V + V.transpose(1, 0, 3, 2))                          # ueg.py:509-513
        return V

    # ---- 3-body mean-field pieces (ueg.py:598-733) --------------------------------------------
    def triple_contractions(self):
        occ = self._occ()
        d = occ[:, None, :] - occ[None, :, :]
        d2 = np.einsum("pqi,pqi->pq", d, d)
        u, d2m = self.u(d2)                       # trunc zeroes the sub-cutoff entries of its input (ueg.py:794)
        dir_e = np.sum(u ** 2 * d2m) * self.n_ele / 2 / self.Omega ** 2 * 2
        exc_e = -2 * 2 * np.einsum("pqo,pqo->", np.einsum("poi,pqi->pqo", d, d), np.einsum("pq,po->pqo", u, u)) / 2.0 / self.Omega ** 2
        return dir_e + exc_e

    def double_contractions(self):
        no = self.n_ele // 2
        kp, ki = self.kp, self.kp[:no]
        dpi = kp[:, None, :] - ki[None, :, :]
        dpi2 = np.einsum("pij,pij->pi", dpi, dpi)
        u_pi, dpi2m = self.u(dpi2)
        e_perl = 2.0 * self.n_ele / self.Omega ** 2 / 2 * np.sum(u_pi ** 2 * dpi2m, axis=1)
        e_wave = -np.einsum("pij,pij->p", np.einsum("pik,pjk->pij", dpi, dpi), np.einsum("pi,pj->pij", u_pi, u_pi)) * 2 / self.Omega ** 2 / 2
        dij = ki[:, None, :] - ki[None, :, :]
        dij2 = np.einsum("ijk,ijk->ij", dij, dij)
        u_ij, dij2m = self.u(dij2)
        e_shield = np.ones(self.n_p) * np.einsum("ij,ij->", u_ij ** 2, dij2m) * 2 / 2 / self.Omega ** 2
        e_frog = -np.einsum("ijp,ijp->p", np.einsum("ijk,pik->ijp", dij, -dpi), np.einsum("ij,pi->ijp", u_ij, u_pi)) * 4 / self.Omega ** 2 / 2
        return e_perl + e_wave + e_shield + e_frog
