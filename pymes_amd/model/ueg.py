"""3D uniform electron gas (pymes/model/ueg.py): plane-wave basis, Coulomb and transcorrelated
two-body integrals, mean-field pieces of the three-body operator.

Drop-in for the parts of ``pymes.model.ueg.UEG`` that feed the CCD/DCSD path (BASELINE config 4):
``init_single_basis``, ``eval_2b_integrals`` (Coulomb, ``is_only_2b``, ``is_effect_2b``,
``is_rpa_approx``; the reference's correlators ``trunc``, ``yukawa``, ``gaskell``, ``gaskell_modified``, ``smooth``,
``coulomb``, ``stg`` are evaluated inside the kernels from the same float k^2 the reference forms, a callable of the
caller's own is tabulated over the lattice shells and looked up on the device), ``double_contractions_in_3_body``,
``triple_contractions_in_3_body``.  The O(n_pw^3) integral evaluation — a Python triple loop with a
(2*30+1)^3 lattice sum per (p,r) pair in the reference — runs as HIP kernels (``pymes_ueg_eval_2b``);
the O(n_occ^2 n_pw) mean-field contractions stay on the host like the reference's.
"""
import ctypes as C
import time
import warnings

import numpy as np

from pymes_amd import _lib
from pymes_amd.basis_set import planewave
from pymes_amd.device import Context
from pymes_amd.log import print_logging_info


class UEG:
    def __init__(self, n_ele, n_alpha, n_beta, rs, device=0):
        if n_ele % 2 != 0 or n_alpha != n_beta:
            warnings.warn("only closed-shell systems are supported")
        self.n_ele, self.n_alpha, self.n_beta = int(n_ele), int(n_alpha), int(n_beta)
        self.rs = rs
        self.L = self.rs * ((4 * np.pi * self.n_ele) / 3) ** (1.0 / 3.0)
        self.Omega = self.L ** 3
        self.basis_fns = None
        self.imax = 0
        self.cutoff = 0.
        self.basis_indices_map = None
        self.kPrime = None
        self.correlator = None
        self.k_cutoff = None
        self.gamma = None
        self.device = device
        self._k_shift = np.zeros(3)

    # ---- basis (ueg.py:82-172) -----------------------------------------------------------------
    def is_k_in_basis(self, ke):
        return ke <= self.cutoff * (2 * np.pi / self.L) ** 2 / 2.

    def init_basis_indices_map(self):
        m = self.imax * 2 + 1
        self.basis_indices_map = -1 * np.ones(m ** 3).astype(int)
        for i in range(len(self.basis_fns) // 2):
            k = self.basis_fns[i * 2].k
            self.basis_indices_map[m * m * (k[0] + self.imax) + m * (k[1] + self.imax) + k[2] + self.imax] = i

    def init_single_basis(self, cutoff, k_shift=(0., 0., 0.)):
        k_shift = np.array(k_shift, dtype=float)
        self._k_shift = k_shift
        self.cutoff = cutoff
        self.imax = int(np.ceil(np.sqrt(cutoff + k_shift.dot(k_shift)))) + 1
        fns = []
        rng = range(-self.imax, self.imax + 1)
        for i in rng:
            for j in rng:
                for k in rng:
                    bfn = planewave.BasisFunc(i, j, k, self.L, 1, k_shift)
                    if self.is_k_in_basis(bfn.kinetic):
                        fns.append(bfn)
                        fns.append(planewave.BasisFunc(i, j, k, self.L, -1, k_shift))
        fns.sort()                               # stable: degenerate shells keep their generation order
        self.basis_fns = tuple(fns)
        self.init_basis_indices_map()
        return self.basis_fns

    # ---- correlator (ueg.py:772-800); like the reference it zeroes the small entries of an array argument
    def trunc(self, kSquare):
        if self.k_cutoff is None:
            self.k_cutoff = int(np.ceil(np.sqrt(self.cutoff)))
        if self.gamma is None:
            self.gamma = 1.0
        kc2 = (self.k_cutoff * 2 * np.pi / self.L) ** 2
        if not isinstance(kSquare, np.ndarray):
            if kSquare <= kc2 * (1 + 0.00001):
                kSquare = 0.
        else:
            kSquare[kSquare <= kc2 * (1 + 0.00001)] = 0.
        result = np.divide(-4. * np.pi, kSquare ** 2, out=np.zeros_like(kSquare), where=(kSquare > 1e-12))
        return result * self.gamma

    # ---- the other correlators of the reference, with its two call forms: an ndarray argument goes through np.divide
    # masks, a float through an if/else — the two agree except AT the cut-off of gaskell / gaskell_modified
    @staticmethod
    def _masked(num, den, keep):
        den = np.asarray(den, dtype=np.float64)
        return np.divide(num, den, out=np.zeros_like(den), where=keep)

    def yukawa(self, kSquare, multiply_by_k_square=False):
        """ueg.py:740-770: -4 pi / (k^2 + gamma), gamma in units of sqrt(rho pi / 4)."""
        if multiply_by_k_square:
            raise NotImplementedError("multiply_by_k_square is not used by the integral builder")
        g0 = np.sqrt(self.n_ele / self.Omega / 4. * np.pi)
        g = g0 if self.gamma is None else self.gamma * g0
        floor = self.k_cutoff * (2 * np.pi / self.L) ** 2 + g if self.k_cutoff is not None else 1e-12
        b = kSquare + g
        return self._masked(-4. * np.pi, b, np.abs(b) > floor)

    def gaskell(self, kSquare, multiply_by_k_square=False):
        """ueg.py:836-883: -mu / k^2 below k_cutoff^2 k_F^2 (default 4 k_F^2), mu = gamma sqrt(4 pi / rho)."""
        mu = np.sqrt(4. * np.pi / (self.n_ele / self.Omega)) * (self.gamma if self.gamma is not None else 1.)
        kf = self.basis_fns[int(self.n_ele / 2) * 2].kp
        cut = (self.k_cutoff ** 2 if self.k_cutoff is not None else 4.) * kf.dot(kf)
        if not isinstance(kSquare, np.ndarray):
            return -(mu / kSquare) if (kSquare < cut and kSquare > 1e-12) else -0.
        res = self._masked(mu, kSquare, kSquare > 1e-12)
        res[kSquare > cut] = 0.
        return -res

    def gaskell_modified(self, kSquare, multiply_by_k_square=False):
        """ueg.py:802-834: -4 pi / k^4 from (k_cutoff 2 pi / L)^2 (default 2) on."""
        cut = (self.k_cutoff * (2 * np.pi / self.L)) ** 2 if self.k_cutoff is not None else 2
        if not isinstance(kSquare, np.ndarray):
            if kSquare < cut and kSquare > 1e-12:
                return -0.
            with np.errstate(divide="ignore"):
                return -(4 * np.pi / np.float64(kSquare) ** 2)
        return -self._masked(4 * np.pi, kSquare ** 2, kSquare >= cut)

    def smooth(self, kSquare, multiply_by_k_square=False):
        """ueg.py:885-903: trunc with an error-function switch of relative width gamma (default 0.01)."""
        from scipy import special
        if self.k_cutoff is None:
            self.k_cutoff = int(np.ceil(np.sqrt(self.cutoff)))
        if self.gamma is None:
            self.gamma = 0.01
        kc = np.sqrt((self.k_cutoff * 2 * np.pi / self.L) ** 2)
        k = np.sqrt(kSquare)
        return self._masked(-4. * np.pi * (1. + special.erf((k - kc) / (kc * self.gamma))) / 2., np.asarray(kSquare) ** 2,
                            kSquare > (kc * self.gamma) ** 2)

    def coulomb(self, kSquare, multiply_by_k_square=False):
        """ueg.py:905-915: -4 pi gamma / k^2."""
        return self._masked(-4. * np.pi * (1. if self.gamma is None else self.gamma), kSquare, np.asarray(kSquare) > 1e-12)

    def stg(self, kSquare, multiply_by_k_square=False):
        """ueg.py:917-935: Slater-type geminal, -4 pi / gamma / (k^2 + gamma^2)^2, gamma default sqrt(4 pi rho)."""
        g = np.sqrt(4. * np.pi * self.n_ele / self.Omega) if self.gamma is None else self.gamma
        floor = (self.k_cutoff * (2 * np.pi / self.L) ** 2 + g ** 2) ** 2 if self.k_cutoff is not None else 1e-12
        b = (kSquare + g ** 2) ** 2
        return self._masked(-4. * np.pi / g, b, np.abs(b) > floor)

    def _correlator_spec(self, correlator):
        """(kind, params) of pymes_ueg_eval_2b_corr for the reference's named correlators bound to this model, with the
        parameters formed by the same float expressions as the methods above; None for anything else (-> tables)."""
        if getattr(correlator, "__self__", None) is not self:
            return None
        name = getattr(correlator, "__name__", "")
        tw2 = (2 * np.pi / self.L) ** 2
        if name == "gaskell":
            mu = np.sqrt(4. * np.pi / (self.n_ele / self.Omega)) * (self.gamma if self.gamma is not None else 1.)
            kf = self.basis_fns[int(self.n_ele / 2) * 2].kp
            return 1, [mu, (self.k_cutoff ** 2 if self.k_cutoff is not None else 4.) * kf.dot(kf)]
        if name == "gaskell_modified":
            return 2, [(self.k_cutoff * (2 * np.pi / self.L)) ** 2 if self.k_cutoff is not None else 2.]
        if name == "coulomb":
            return 3, [-4. * np.pi * (1. if self.gamma is None else self.gamma)]
        if name == "yukawa":
            g0 = np.sqrt(self.n_ele / self.Omega / 4. * np.pi)
            g = g0 if self.gamma is None else self.gamma * g0
            return 4, [g, self.k_cutoff * tw2 + g if self.k_cutoff is not None else 1e-12]
        if name == "stg":
            g = np.sqrt(4. * np.pi * self.n_ele / self.Omega) if self.gamma is None else self.gamma
            return 5, [g ** 2, (self.k_cutoff * tw2 + g ** 2) ** 2 if self.k_cutoff is not None else 1e-12, -4. * np.pi / g]
        if name == "smooth":
            kc = np.sqrt((self.k_cutoff * 2 * np.pi / self.L) ** 2)
            return 6, [kc, kc * self.gamma, (kc * self.gamma) ** 2]
        return None

    def _correlator_tables(self, correlator, lattice_cutoff=30):
        """u over the lattice shells m = |n|^2 (every argument the integral builder passes to the correlator is
        |2 pi n / L|^2 for an integer n): (called with a float, called with an ndarray).  The argument of shell m is the
        reference's own float expression kp.dot(kp) for the first vector of the shell; a correlator with a jump exactly
        on a shell whose members round differently would be evaluated at that member's value."""
        r = lattice_cutoff + 2 * self.imax
        g = np.arange(-r, r + 1)
        vec = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
        m_all = np.einsum("ni,ni->n", vec, vec)
        m_max = 3 * r * r
        shells, first = np.unique(m_all, return_index=True)
        x = (2 * np.pi / self.L) ** 2 * np.arange(m_max + 1, dtype=np.float64)     # shells without a vector: never looked up
        kp = vec[first] * 2 * np.pi / self.L
        x[shells] = np.einsum("ni,ni->n", kp, kp)
        with np.errstate(all="ignore"):
            arr = np.asarray(correlator(x.copy()), dtype=np.float64)
            try:
                sca = np.array([float(correlator(np.float64(v))) for v in x])
            except (TypeError, IndexError, ValueError):      # a callable written for arrays only
                sca = arr.copy()
        arr[~np.isfinite(arr)] = 0.0          # u(0) of a float call may divide by zero; k = 0 terms are never used
        sca[~np.isfinite(sca)] = 0.0
        return np.ascontiguousarray(sca), np.ascontiguousarray(arr)

    # ---- two-body integrals (ueg.py:265-516) on the device -----------------------------------------
    def _mode(self, correlator, flags):
        if correlator is None:
            return 0
        if not callable(correlator):
            raise TypeError("correlator must be a callable u(k^2)")
        on = [k for k, v in flags.items() if v]
        if on == ["is_rpa_approx"] or (flags["is_rpa_approx"]):
            return 3
        if flags["is_only_2b"]:
            return 1
        if flags["is_effect_2b"]:
            return 2
        raise NotImplementedError("supported TC modes: is_only_2b, is_effect_2b, is_rpa_approx "
                                  "(is_only_(non_)hermi_2b and is_exchange_1..3 are the reference's test-only switches)")

    def eval_2b_integrals(self, correlator=None, is_rpa_approx=False, is_only_2b=False, is_only_non_hermi_2b=False,
                          is_only_hermi_2b=False, is_effect_2b=False, is_exchange_1=False, is_exchange_2=False,
                          is_exchange_3=False, dtype=np.float64, sp=1, on_device=False, ctx=None):
        """Returns V_pqrs (numpy [n_p]^4 as the reference; a DeviceArray with ``on_device=True``)."""
        start_time = time.time()
        print_logging_info(__name__, level=0)
        if self.basis_fns is None:
            raise ValueError("Basis functions not initialized!")
        # (a twist k_shift of init_single_basis moves the kinetic energies and the shell order of the basis; every integral
        # below depends on DIFFERENCES of the integer plane-wave vectors only, so the kernels take the shifted basis as it is
        # — checked against the reference: tests/golden/ueg_twist.npz)
        if is_only_non_hermi_2b or is_only_hermi_2b or is_exchange_1 or is_exchange_2 or is_exchange_3:
            raise NotImplementedError("test-only switches of the reference are not on the HIP path")
        mode = self._mode(correlator, dict(is_rpa_approx=is_rpa_approx, is_only_2b=is_only_2b, is_effect_2b=is_effect_2b))
        is_trunc = correlator is not None and getattr(correlator, "__name__", "") == "trunc" and \
            getattr(correlator, "__self__", None) is self
        if correlator is not None:
            self.correlator = correlator
            print_logging_info("Using TC method", level=1)
            print_logging_info("Using correlator: ", getattr(correlator, "__name__", repr(correlator)), level=1)
            if self.k_cutoff is not None:
                print_logging_info("k_cutoff in correlator = {:.8f}".format(self.k_cutoff), level=1)
            if self.gamma is not None:
                print_logging_info("Gamma in correlator = {:.8f}".format(self.gamma), level=1)
            correlator(np.float64(1.0))          # fixes the correlator's k_cutoff / gamma defaults like its first call would
        n_p = len(self.basis_fns) // 2
        k_int = np.ascontiguousarray([self.basis_fns[2 * i].k for i in range(n_p)], dtype=np.int32)
        imap = np.ascontiguousarray(self.basis_indices_map, dtype=np.int32)
        own = ctx is None
        if own:
            ctx = Context(1, 1, device=self.device, workspace_bytes=1 << 20)
        try:
            V = ctx.empty((n_p,) * 4)
            if correlator is None or is_trunc:       # trunc is evaluated inside the kernels
                ctx.lib.call("pymes_ueg_eval_2b", ctx.handle, n_p, self.n_ele, self.imax, mode, float(self.L),
                             float(self.k_cutoff or 0.0), float(self.gamma or 1.0), 30,
                             k_int.ctypes.data_as(C.c_void_p), imap.ctypes.data_as(C.c_void_p), C.c_void_p(V.ptr))
            elif self._correlator_spec(correlator) is not None:       # the reference's other correlators, in the kernels
                kind, prm = self._correlator_spec(correlator)
                prm = (C.c_double * 4)(*(list(map(float, prm)) + [0.0] * (4 - len(prm))))
                ctx.lib.call("pymes_ueg_eval_2b_corr", ctx.handle, n_p, self.n_ele, self.imax, mode, float(self.L), 30, kind,
                             prm, k_int.ctypes.data_as(C.c_void_p), imap.ctypes.data_as(C.c_void_p), C.c_void_p(V.ptr))
            else:                                    # a u(k^2) of the caller's own: tabulated over the lattice shells
                tab_s, tab_a = self._correlator_tables(correlator, 30)
                ctx.lib.call("pymes_ueg_eval_2b_tab", ctx.handle, n_p, self.n_ele, self.imax, mode, float(self.L), 30,
                             k_int.ctypes.data_as(C.c_void_p), imap.ctypes.data_as(C.c_void_p),
                             tab_s.ctypes.data_as(C.c_void_p), tab_a.ctypes.data_as(C.c_void_p), len(tab_a),
                             C.c_void_p(V.ptr))
            if mode == 2:                        # ueg.py:509-513: symmetrise w.r.t. the electron labels
                Vs = ctx.permute("pqrs->pqrs", V, alpha=0.5)
                ctx.permute("qpsr->pqrs", V, out=Vs, alpha=0.5, beta=1.0)
                V = Vs
            print_logging_info("{:.3f} s spent on ".format(time.time() - start_time) + __name__, level=1)
            if on_device:
                if own:
                    raise ValueError("on_device=True needs the caller's ctx (the array lives in its context)")
                return V
            return V.get().astype(dtype, copy=False)
        finally:
            if own:
                ctx.close()

    # ---- host mirrors of the per-element helpers (ueg.py:518-596) ----------------------------------
    def _occ_kp(self):
        return np.array([self.basis_fns[i * 2].kp for i in range(self.n_ele // 2)])

    def contract_exchange_3_body(self, p_vec, kVec):
        d = p_vec - self._occ_kp()
        return np.sum((d @ kVec) * self.correlator(kVec @ kVec) * self.correlator(np.einsum("ni,ni->n", d, d))) / self.Omega

    def contractP_KWithQ(self, pVec, kVec):
        v1, v2 = pVec - kVec - self._occ_kp(), pVec - self._occ_kp()
        return np.sum(np.einsum("ni,ni->n", v1, v2) * self.correlator(np.einsum("ni,ni->n", v1, v1))
                      * self.correlator(np.einsum("ni,ni->n", v2, v2))) / self.Omega

    def sumNablaUSquare(self, k, cutoff=30):
        if self.kPrime is None:
            g = np.arange(-cutoff, cutoff + 1)
            self.kPrime = np.array([[a, b, c] for a in g for b in g for c in g])
        k1 = 2 * np.pi * self.kPrime / self.L
        k2 = k - k1
        val = np.einsum("ni,ni->n", k1, k2) * self.correlator(np.einsum("ni,ni->n", k1, k1)) \
            * self.correlator(np.einsum("ni,ni->n", k2, k2))
        return np.sum(val) / self.Omega

    # ---- mean-field pieces of the 3-body operator (ueg.py:598-733), O(n_occ^2 n_pw) on the host --------
    def triple_contractions_in_3_body(self):
        print_logging_info("UEG.triple_contractions_in_3_body", level=1)
        occ = self._occ_kp()
        d = occ[:, None, :] - occ[None, :, :]
        d2 = np.einsum("pqi,pqi->pq", d, d)
        u = self.correlator(d2)                  # zeroes the sub-cutoff entries of d2 as well (reference behaviour)
        dirE = np.sum(u ** 2 * d2) * self.n_ele / 2 / self.Omega ** 2 * 2
        excE = -2 * 2 * np.einsum("pqo,pqo->", np.einsum("poi,pqi->pqo", d, d), np.einsum("pq,po->pqo", u, u)) \
            / 2. / self.Omega ** 2
        print_logging_info("Direct E = {:.8f}".format(dirE), level=2)
        print_logging_info("Exchange E = {:.8f}".format(excE), level=2)
        return dirE + excE

    def double_contractions_in_3_body(self):
        print_logging_info("UEG.double_contractions_in_3_body", level=1)
        no, n_p = self.n_ele // 2, len(self.basis_fns) // 2
        kp = np.array([self.basis_fns[i * 2].kp for i in range(n_p)])
        ki = kp[:no]
        dpi = kp[:, None, :] - ki[None, :, :]
        dpi2 = np.einsum("pij,pij->pi", dpi, dpi)
        u_pi = self.correlator(dpi2)             # dpi2 is cut in place, as in the reference
        e_perl = 2.0 * self.n_ele / self.Omega ** 2 / 2 * np.sum(u_pi ** 2 * dpi2, axis=1)
        e_wave = -np.einsum("pij,pij->p", np.einsum("pik,pjk->pij", dpi, dpi), np.einsum("pi,pj->pij", u_pi, u_pi)) \
            * 2 / self.Omega ** 2 / 2
        dij = ki[:, None, :] - ki[None, :, :]
        dij2 = np.einsum("ijk,ijk->ij", dij, dij)
        u_ij = self.correlator(dij2)
        e_shield = np.ones(n_p) * np.einsum("ij,ij->", u_ij ** 2, dij2) * 2 / 2 / self.Omega ** 2
        e_frog = -np.einsum("ijp,ijp->p", np.einsum("ijk,pik->ijp", dij, -dpi), np.einsum("ij,pi->ijp", u_ij, u_pi)) \
            * 4 / self.Omega ** 2 / 2
        return e_perl + e_wave + e_shield + e_frog
