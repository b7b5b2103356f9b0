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
import ctypes as C
import os
import time

import numpy as np

from pymes_amd import _lib
from pymes_amd.device import Context, DeviceArray, PymesError
from pymes_amd.integral.device import DressedDeviceIntegrals
from pymes_amd.integral.partition import BLOCK_NAMES
from pymes_amd.log import print_logging_info, print_title
from pymes_amd.mixer.diis import _single_threaded_blas


class _Sigma:
    """Device-resident H-bar . u (eom_ccsd.py:268-385): a handle of the engine's EOM sigma build (csrc/eom.cpp,
    ``pymes_eom_sigma_prepare / _apply``).  Every V.T product that does not depend on the trial vector is hoisted once per solve;
    one ``apply_many`` call builds sigma for any number of trial vectors — exchange-symmetric ones stacked, so that every shared
    operand is read once — in ONE library call (round 4 issued the ~130 contractions of a build one ctypes call at a time)."""

    # the blocks a sigma build reads (eom_ccsd.py:268-385)
    BLOCKS = ("ijab", "iabj", "iajb", "ijka", "ijak", "iabc", "iajk", "abic", "klij", "abcd")

    def __init__(self, ctx, f, t2, dressed=False):
        """``dressed``: read the context's T1-DRESSED blocks (the context of a CCSD solve whose integrals were dressed in
        place, ``CCSD.get_T1_dressed_V`` on a ``DeviceIntegrals``) instead of blocks uploaded as they are."""
        self.ctx = ctx
        self.no, self.nv = ctx.no, ctx.nv
        self.dressed = bool(dressed)
        self.T = t2                                   # (kept alive: the handle reads it in every build)
        self._f = np.ascontiguousarray(f, dtype=np.float64)
        if self._f.shape != (ctx.n, ctx.n):
            raise ValueError("the dressed Fock matrix must be [n, n]")
        h = C.c_void_p()
        ctx.lib.call("pymes_eom_sigma_prepare", ctx.handle, _lib.host_ptr(self._f), C.c_void_p(t2.ptr), int(self.dressed),
                     C.byref(h))
        self._h = h
        ctx.on_close(self._ctx_closing)               # the handle dies before its context
        flags = C.c_int()
        ctx.lib.call("pymes_eom_sigma_flags", self._h, C.byref(flags))
        fl = flags.value
        self.v_sym, self.t_sym, self.hole_sym = bool(fl & 1), bool(fl & 2), bool(fl & 4)
        self.fused_ok, self.many_ok = bool(fl & 8), bool(fl & 16)

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None:        # (also after the context has gone: the library invalidated the handle then, this frees its shell)
            self.ctx.lib.call("pymes_eom_sigma_destroy", h)

    def _ctx_closing(self, ctx):
        try:
            self.close()
        except Exception:
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def exchange_symmetric(self, u2):
        return self.ctx.exchange_symmetric(u2)      # one reduction kernel, no temporary

    def apply_many(self, u1s, u2s, syms=None, out1=None, out2=None):
        """[(sigma1_z, sigma2_z)] for the trial vectors (u1s[z], u2s[z]).  ``syms[z]``: the caller's knowledge that u2_z has
        the exchange symmetry u2_abij = u2_baji (tested on the device when ``syms`` is None); ``out1`` / ``out2``: device
        arrays that receive sigma1_z / sigma2_z (e.g. the two parts of a flat subspace vector) instead of fresh ones."""
        if self._h is None:
            raise PymesError("the EOM sigma handle has been destroyed (its context was closed)")
        k = len(u1s)
        c, no, nv = self.ctx, self.no, self.nv
        s1 = [out1[z] if out1 is not None else c.empty((nv, no)) for z in range(k)]
        s2 = [out2[z] if out2 is not None else c.empty((nv, nv, no, no)) for z in range(k)]
        sym = None if syms is None else (C.c_int * max(k, 1))(*[int(bool(x)) for x in syms])
        c.lib.call("pymes_eom_sigma_apply", self._h, k, _lib.ptr_array([u.ptr for u in u1s]), _lib.ptr_array([u.ptr for u in u2s]),
                   sym, _lib.ptr_array([x.ptr for x in s1]), _lib.ptr_array([x.ptr for x in s2]))
        return list(zip(s1, s2))

    def apply(self, u1, u2, u2_sym=None):
        return self.apply_many([u1], [u2], None if u2_sym is None else [u2_sym])[0]

    def singles(self, u1, u2):
        """eom_ccsd.py:268-310."""
        return self.apply(u1, u2)[0]

    def doubles(self, u1, u2, u2_sym=None):
        """eom_ccsd.py:312-385."""
        return self.apply(u1, u2, u2_sym)[1]

    def trim(self):
        self.ctx.trim()


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

    def ritz_residuals(self, t_fock_dressed_pq, dict_t_V_dressed, t_T_abij):
        """Certificate of the last device-resident ``solve``: for every returned root e_n the Ritz vector r_n = U v_n is
        rebuilt from the basis of the last pass, the sigma build (eom_ccsd.py:268-385; pinned to the reference's own output at
        this size) is applied to it ONCE MORE — a fresh ``_Sigma``, not the W = sigma(U) the driver carried along — and
        ``|sigma(r_n) - e_n r_n| / |r_n|`` is returned.  Independent of the driver's bookkeeping: a wrong projection,
        orthonormalisation, collapse or root choice shows up here as an O(1) number.  Not part of the reference's interface."""
        if not isinstance(dict_t_V_dressed, DressedDeviceIntegrals) or getattr(self, "_ritz", None) is None:
            raise RuntimeError("ritz_residuals: needs a finished device-resident solve() on these integrals")
        ctx = dict_t_V_dressed.ctx
        dict_t_V_dressed.require(_Sigma.BLOCKS)
        f = t_fock_dressed_pq.get() if isinstance(t_fock_dressed_pq, DeviceArray) else np.asarray(t_fock_dressed_pq, dtype=np.float64)
        t2 = t_T_abij if isinstance(t_T_abij, DeviceArray) else ctx.array(t_T_abij)
        us, v, e = self._ritz
        lay = self._layout(self.no, ctx.nv)
        nflat = lay[2]
        sig = _Sigma(ctx, f, t2, dressed=True)
        rz = [ctx.empty((nflat,)) for _ in range(self.n_excit)]
        ctx.lincomb_multi(rz, us, v)
        wz = [self._zero_pad(ctx, ctx.empty((nflat,)), lay) for _ in rz]
        u2s = [self._u2(ctx, r, lay) for r in rz]
        sig.apply_many([self._u1(ctx, r, lay) for r in rz], u2s, [sig.exchange_symmetric(u2) for u2 in u2s],
                       out1=[self._u1(ctx, w, lay) for w in wz], out2=[self._u2(ctx, w, lay) for w in wz])
        out = []
        for n in range(self.n_excit):        # the residual vector itself (its norm from the three inner products cancels at 1e-8)
            z = ctx.empty((nflat,))
            ctx.lincomb_multi([z], [wz[n], rz[n]], np.array([[1.0], [-e[n]]]))
            out.append(float(np.sqrt(ctx.gram([z], [z])[0, 0] / ctx.gram([rz[n]], [rz[n]])[0, 0])))
        return out

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
                self._ritz = (list(us), v, np.array(e)) if device_form else None     # for ritz_residuals()
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

    def _device_diagonals(self, t_fock_pq, dict_t_V, t_T_abij):
        """Both diagonals (eom_ccsd.py:169-198, :200-266) on the device for the device-resident hand-over: (d1 [v,o], d2
        [v,v,o,o]) as DeviceArrays of the integrals' context (``pymes_eom_diagonals``: the V.T sums, the four diagonal
        slices and the assembly in three kernels; nothing of size o^2 v^2 crosses PCIe)."""
        c = dict_t_V.ctx
        dict_t_V.require(("ijab", "iabj", "iajb", "klij", "abcd"))
        f = t_fock_pq.get() if isinstance(t_fock_pq, DeviceArray) else np.asarray(t_fock_pq)
        f = np.ascontiguousarray(f, dtype=np.float64)
        t2 = t_T_abij if isinstance(t_T_abij, DeviceArray) else c.array(t_T_abij)
        d1, d2 = c.empty((c.nv, c.no)), c.empty((c.nv, c.nv, c.no, c.no))
        c.lib.call("pymes_eom_diagonals", c.handle, _lib.host_ptr(f), C.c_void_p(t2.ptr), 1, C.c_void_p(d1.ptr),
                   C.c_void_p(d2.ptr))
        return d1, d2

    def get_diag_singles(self, t_fock_pq, dict_t_V, t_T_abij, _inputs=None):
        """eom_ccsd.py:169-198, terms grouped: with V~ = 2V - V^(ab), T~ = 2T - T^(ab) the four (a,i)-resolved
        V.T terms are one Hadamard sum.  O(o^2 v^2) work on host arrays (preconditioner data, once per solve)."""
        no = self.no
        if _inputs is None and isinstance(dict_t_V, DressedDeviceIntegrals):
            return self._device_diagonals(t_fock_pq, dict_t_V, t_T_abij)[0].get()
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
        if _inputs is None and isinstance(dict_t_V, DressedDeviceIntegrals):
            return self._device_diagonals(t_fock_pq, dict_t_V, t_T_abij)[1].get()
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
