"""FEAST-EOM-CCSD (pymes/solver/feast_eom_ccsd.py) on the MI355X engine.

Drop-in for ``pymes.solver.feast_eom_ccsd.FEAST_EOM_CCSD``: same constructor and attributes, ``solve(f_dressed,
dict_t_V_dressed, t_T_abij)`` returns the Ritz values of the last pass.  The reference integrates the resolvent
(z - H̄)^-1 over the upper half of the circle |z - e_c| = e_r with an 8-point Gauss-Legendre rule (:98-100), solves every
(z_e - H̄) Q = Y with scipy's flexible GCROT(m,k) under a diagonal preconditioner (:293-350) and diagonalises H̄ in the span
of the real parts (:124-149); its trial space starts from two random vectors and is doubled while it holds fewer than
``n_trial`` vectors (:152-165).  All of that is kept as it is — including the draws from the global ``np.random.rand``
(:90-91), so ``np.random.seed`` before ``solve`` reproduces a reference run.

What runs where: every vector of length nv*no + nv^2*no^2 — trial vectors, Krylov bases, the C/U pairs of GCROT — lives in
HBM; a complex vector is a pair of real device arrays and H̄ acts on both through the sigma build of eom_ccsd.py (one
``apply_many`` call: the operator is real and linear).  The host sees inner products, the small Hessenberg / least-squares
problems of the Krylov solver (scipy.linalg.qr_insert / lstsq on (m+1) x m matrices, as scipy's own solver does) and the
projected n_trial x n_trial eigenproblem.

The linear solver restates scipy.sparse.linalg.gcrotmk (scipy 1.15.3, _isolve/_gcrotmk.py: E. de Sturler, SIAM J. Sci.
Comput. 20, 864 (1999); J. Hicken, D. Zingg, ibid. 32, 1672 (2010)) with the reference's parameters: zero start, m = 20,
k = m, truncate = "oldest", right preconditioner 1 / (z - diag + 0.01), relative tolerance 1e-4, ``ls_max_iter`` outer
cycles.  Parity caveat (DESIGN): the solves stop at 1e-4, so a Ritz value that FEAST has not converged moves by ~1e-5 when
one inner iteration count changes by rounding — the reference run against itself with another summation order shows it;
converged values inside the window and the first pass are reproducible to 1e-8.
"""
import time

import numpy as np
from scipy.linalg import eig, lstsq, qr_insert

from pymes_amd.device import DeviceArray
from pymes_amd.integral.device import DressedDeviceIntegrals
from pymes_amd.log import print_logging_info, print_title
from pymes_amd.mixer.diis import _single_threaded_blas
from pymes_amd.solver.eom_ccsd import EOM_CCSD, _Sigma


class _CVec:
    """Complex device vector: two real flat arrays."""
    __slots__ = ("re", "im")

    def __init__(self, re, im):
        self.re, self.im = re, im


class _Ops:
    """BLAS-1 on complex device vectors through the context's fused dots / linear combinations."""

    def __init__(self, ctx, n):
        self.c, self.n = ctx, n

    def zeros(self):
        return _CVec(self.c.zeros((self.n,)), self.c.zeros((self.n,)))

    def copy(self, x):
        return _CVec(self.c.empty((self.n,)).copy_from(x.re), self.c.empty((self.n,)).copy_from(x.im))

    def dotc(self, a, b):                      # sum conj(a) b  (zdotc, what scipy's 'dot' is for complex vectors)
        d = self.c.dots([a.re, a.im, a.re, a.im], [b.re, b.im, b.im, b.re])
        return complex(d[0] + d[1], d[2] - d[3])

    def nrm2(self, a):
        d = self.c.dots([a.re, a.im], [a.re, a.im])
        return float(np.sqrt(d[0] + d[1]))

    def axpy(self, alpha, x, y):               # y += alpha x
        ar, ai = float(np.real(alpha)), float(np.imag(alpha))
        self.c.lincomb(y.re, [y.re, x.re, x.im], [1.0, ar, -ai])
        self.c.lincomb(y.im, [y.im, x.im, x.re], [1.0, ar, ai])
        return y

    def scaled(self, alpha, x):                # alpha x as a new vector
        ar, ai = float(np.real(alpha)), float(np.imag(alpha))
        out = _CVec(self.c.empty((self.n,)), self.c.empty((self.n,)))
        self.c.lincomb(out.re, [x.re, x.im], [ar, -ai])
        self.c.lincomb(out.im, [x.im, x.re], [ar, ai])
        return out

    def combine(self, vecs, coeffs):           # sum_k coeffs[k] vecs[k]
        out = self.scaled(coeffs[0], vecs[0])
        for v, a in zip(vecs[1:], coeffs[1:]):
            self.axpy(a, v, out)
        return out


class FEAST_EOM_CCSD(EOM_CCSD):
    def __init__(self, no, e_c=0., e_r=1, n_trial=5, max_iter=20, tol=1e-12, device=0, **kwargs):
        """feast_eom_ccsd.py:29-64."""
        self.no = no
        self.e_c = e_c
        self.e_r = e_r
        self.n_trial = n_trial
        self.n_excit = 2
        self.max_iter = max_iter
        self.tol = tol
        self.linear_solver = "Jacobi"          # (the reference's default; its solve() runs GCROT(m,k) for it, :116-119)
        self.ls_max_iter = 20
        self.u_singles = []
        self.u_doubles = []
        self.eigvals = np.array([self.e_c - self.e_r, self.e_c + self.e_r])
        self.eigvecs = None
        self.device = device
        self.history = []                      # Ritz values of every pass (the reference only logs them)
        self.linear_solver_info = []           # (outer cycles, matvecs) per linear solve of the last solve()

    def dump_log(self):
        pass

    # ---- device pieces -------------------------------------------------------------------------------------------------
    def _matvec(self, sig, ops, ze, x, n1, shapes, hs=1.0):
        """(ze - hs H̄) x for a complex device vector (feast_eom_ccsd.py:309-340): H̄ on the real and the imaginary part in
        one sigma call.  ``hs``: 1 for FEAST, 1j dt for the real-time form of the reference's solvers (:321-334)."""
        c = ops.c
        parts = lambda v: (EOM_CCSD._part(c, v, 0, shapes[0]), EOM_CCSD._part(c, v, n1, shapes[1]))
        (r1, r2), (i1, i2) = parts(x.re), parts(x.im)
        # (random trial vectors have no exchange symmetry, and neither have their Krylov iterates: the general sigma form)
        (sr1, sr2), (si1, si2) = sig.apply_many([r1, i1], [r2, i2], syms=[False, False])
        zr, zi = float(np.real(ze)), float(np.imag(ze))
        hr, hi = float(np.real(hs)), float(np.imag(hs))
        y = _CVec(c.empty((ops.n,)), c.empty((ops.n,)))
        (yr1, yr2), (yi1, yi2) = parts(y.re), parts(y.im)
        flat = lambda t: t.reshape(t.size)
        # y = ze x - hs (sr + i si):  re = zr xr - zi xi - hr sr + hi si,  im = zr xi + zi xr - hr si - hi sr
        for out, a, b, s, t, sa, sb, ss, st in ((yr1, r1, i1, sr1, si1, zr, -zi, -hr, hi), (yr2, r2, i2, sr2, si2, zr, -zi, -hr, hi),
                                                (yi1, i1, r1, si1, sr1, zr, zi, -hr, -hi), (yi2, i2, r2, si2, sr2, zr, zi, -hr, -hi)):
            if st == 0.0:
                c.lincomb(flat(out), [flat(a), flat(b), flat(s)], [sa, sb, ss])
            else:
                c.lincomb(flat(out), [flat(a), flat(b), flat(s), flat(t)], [sa, sb, ss, st])
        return y

    def _fgmres(self, ops, matvec, psolve, v0, m, atol, cs):
        """Flexible GMRES inner cycle with projection against the columns C (scipy _gcrotmk.py:_fgmres)."""
        vs, zs, res, j = [v0], [], np.nan, -1
        B = np.zeros((len(cs), m), dtype=complex)
        Q = np.ones((1, 1), dtype=complex)
        R = np.zeros((1, 0), dtype=complex)
        eps = np.finfo(float).eps
        breakdown = False
        for j in range(m):
            z = psolve(vs[-1])
            w = matvec(z)
            self._matvecs += 1
            w_norm = ops.nrm2(w)
            for i, cvec in enumerate(cs):                              # (1 - C C^H) A z
                alpha = ops.dotc(cvec, w)
                B[i, j] = alpha
                ops.axpy(-alpha, cvec, w)
            hcur = np.zeros(j + 2, dtype=complex)
            for i, v in enumerate(vs):                                 # modified Gram-Schmidt against V
                alpha = ops.dotc(v, w)
                hcur[i] = alpha
                ops.axpy(-alpha, v, w)
            hcur[j + 1] = ops.nrm2(w)
            with np.errstate(over="ignore", divide="ignore"):
                alpha = 1 / hcur[-1]
            if np.isfinite(alpha):
                w = ops.scaled(alpha, w)
            if not (np.real(hcur[-1]) > eps * w_norm):
                breakdown = True
            vs.append(w)
            zs.append(z)
            Q2 = np.zeros((j + 2, j + 2), dtype=complex, order="F")
            Q2[:j + 1, :j + 1] = Q
            Q2[j + 1, j + 1] = 1
            R2 = np.zeros((j + 2, j), dtype=complex, order="F")
            R2[:j + 1, :] = R
            with _single_threaded_blas():
                Q, R = qr_insert(Q2, R2, hcur, j, which="col", overwrite_qru=True, check_finite=False)
            res = abs(Q[0, -1])
            if res < atol or breakdown:
                break
        if not np.isfinite(R[j, j]):
            raise np.linalg.LinAlgError()
        with _single_threaded_blas():
            y, _, _, _ = lstsq(R[:j + 1, :j + 1], Q[0, :j + 1].conj())
        return Q, R, B[:, :j + 1], vs, zs, y, res

    def _gcrotmk_device(self, ops, matvec, psolve, b, rtol=1e-4, maxiter=20, m=20, k=None):
        """(ze - H̄) x = b from x = 0 (scipy _gcrotmk.py:gcrotmk as feast_eom_ccsd.py:344 calls it).  Returns (x, info)."""
        k = m if k is None else k
        x = ops.zeros()
        r = ops.copy(b)                                                # b - A 0
        b_norm = ops.nrm2(b)
        if b_norm == 0:
            return x, 0
        atol = rtol * b_norm
        CU = []
        j_outer = -1
        for j_outer in range(maxiter):
            beta = ops.nrm2(r)
            if beta <= atol and (j_outer > 0 or CU):                   # recompute the true residual before accepting it
                r = ops.copy(b)
                ops.axpy(-1.0, matvec(x), r)
                self._matvecs += 1
                beta = ops.nrm2(r)
            if beta <= atol:
                j_outer = -1
                break
            ml = m + max(k - len(CU), 0)
            cs = [cu[0] for cu in CU]
            try:
                Q, R, B, vs, zs, y, _ = self._fgmres(ops, matvec, psolve, ops.scaled(1.0 / beta, r), ml, atol / beta, cs)
            except np.linalg.LinAlgError:
                break
            y = y * beta
            ux = ops.combine(zs, y)                                    # GCROT(m,k) update: u = Z y - U B y, c = V H y
            by = B.dot(y)
            for (_, u), byc in zip(CU, by):
                ops.axpy(-byc, u, ux)
            with np.errstate(invalid="ignore"):
                hy = Q.dot(R.dot(y))
            cx = ops.combine(vs, hy)
            try:
                alpha = 1 / ops.nrm2(cx)
                if not np.isfinite(alpha):
                    raise FloatingPointError()
            except (FloatingPointError, ZeroDivisionError):
                continue
            cx, ux = ops.scaled(alpha, cx), ops.scaled(alpha, ux)
            gamma = ops.dotc(cx, r)
            ops.axpy(-gamma, cx, r)
            ops.axpy(gamma, ux, x)
            while len(CU) >= k and CU:                                 # truncate = "oldest"
                del CU[0]
            CU.append((cx, ux))
        return x, j_outer + 1

    def _jacobi_device(self, ops, matvec, minv, b, n_iter=200):
        """feast_eom_ccsd.py:252-291: damped preconditioned Richardson sweeps Q += 0.01 (b - (z - H̄) Q) / (z - diag + 0.01)."""
        q = ops.zeros()
        for _ in range(n_iter):
            delta = ops.copy(b)
            ops.axpy(-1.0, matvec(q), delta)
            ops.c.cmul(minv.re, minv.im, delta.re, delta.im, delta.re, delta.im)
            ops.axpy(0.01, delta, q)
        return q

    # ---- driver ----------------------------------------------------------------------------------------------------------
    def solve(self, t_fock_dressed_pq, dict_t_V_dressed, t_T_abij):
        """feast_eom_ccsd.py:72-181."""
        print_title("FEAST-EOM-CCSD Solver")
        time_init = time.time()
        no = self.no
        device_form = isinstance(dict_t_V_dressed, DressedDeviceIntegrals)      # the hand-over of EOM_CCSD.solve: everything in HBM
        if device_form:
            from pymes_amd.solver.eom_ccsd import _Sigma as _S
            dict_t_V_dressed.require(_S.BLOCKS)       # a subset dressing / a later dressing on the same context: refuse
        if isinstance(t_fock_dressed_pq, DeviceArray):
            t_fock_dressed_pq = t_fock_dressed_pq.get()
        f = np.asarray(t_fock_dressed_pq, dtype=np.float64)
        nv = f.shape[0] - no
        if device_form:
            # both diagonals on the device (pymes_eom_diagonals): the preconditioners below are formed from them in HBM
            diag_ai, diag_abij = self._device_diagonals(f, dict_t_V_dressed, t_T_abij)
        else:
            dg = self._diag_inputs(dict_t_V_dressed, t_T_abij)
            diag_ai = self.get_diag_singles(f, dict_t_V_dressed, t_T_abij, _inputs=dg)
            diag_abij = self.get_diag_doubles(f, dict_t_V_dressed, t_T_abij, _inputs=dg)
            del dg
        print_logging_info("Initialising u tensors...", level=1)
        # the reference APPENDS its two random vectors to whatever self.u_singles / u_doubles hold (:89-91): a second solve()
        # on the same object continues from the trial space of the first
        host_us = []
        for a0, b0 in zip(self.u_singles, self.u_doubles):
            a0 = a0.get() if isinstance(a0, DeviceArray) else np.asarray(a0)
            b0 = b0.get() if isinstance(b0, DeviceArray) else np.asarray(b0)
            host_us.append(np.concatenate((np.real(a0).ravel(), np.real(b0).ravel())))
        for _ in range(self.n_excit):                                              # :89-91 (global numpy generator)
            a = 0.5 - np.random.rand(*diag_ai.shape)
            b = (0.5 - np.random.rand(*diag_abij.shape)) * 0.01
            host_us.append(np.concatenate((a.ravel(), b.ravel())))
        x, w = np.polynomial.legendre.leggauss(8)                                  # :98-100
        theta = -np.pi / 2 * (x - 1)
        z = self.e_c + self.e_r * np.exp(1j * theta)
        n1, n2 = nv * no, nv * nv * no * no
        n = n1 + n2
        shapes = ((nv, no), (nv, nv, no, no))
        ctx = dict_t_V_dressed.ctx if device_form else self._context(dict_t_V_dressed, nv)
        if device_form:
            diag = ctx.empty((n,))                     # [d1 | d2], flat, on the device
            self._part(ctx, diag, 0, shapes[0]).copy_from(diag_ai)
            self._part(ctx, diag, n1, shapes[1]).copy_from(diag_abij)
        else:
            diag = np.concatenate((diag_ai.ravel(), diag_abij.ravel()))
        self.history, self.linear_solver_info = [], []
        from pymes_amd.solver.ccd import quiet_collector
        collector = quiet_collector().__enter__()
        try:
            t2 = t_T_abij if isinstance(t_T_abij, DeviceArray) else ctx.array(t_T_abij)
            sig = _Sigma(ctx, f, t2, dressed=device_form)
            ops = _Ops(ctx, n)
            zero = ctx.zeros((n,))
            us = [ctx.array(u) for u in host_us]

            def normalise(u):                                                      # :625-630
                ctx.lincomb(u, [u], [1.0 / ctx.norm(u)])
            for u in us:
                normalise(u)
            minv = []                                                              # 1 / (z_e - diag + 0.01) per node (:342)
            for ze in z:
                if device_form:
                    mv = _CVec(ctx.empty((n,)), ctx.empty((n,)))
                    ctx.cshift_inv(diag, ze, 1.0, 0.01, mv.re, mv.im)
                    minv.append(mv)
                    continue
                mv = 1.0 / (ze - diag + 0.01)
                minv.append(_CVec(ctx.array(np.ascontiguousarray(mv.real)), ctx.array(np.ascontiguousarray(mv.imag))))
            e_norm_prev = 1e10
            for it in range(self.max_iter):
                time_iter_init = time.time()
                m = len(us)
                Qs = [ctx.zeros((n,)) for _ in range(m)]
                for u in us:                                                       # :109-110
                    normalise(u)
                for e in range(len(z)):                                            # :113-121
                    print_logging_info(f"e = {e}, z = {z[e]}, theta = {theta[e]}, w = {w[e]}", level=1)
                    matvec = lambda v, ze=z[e]: self._matvec(sig, ops, ze, v, n1, shapes)
                    # linear_solver: every setting the reference can run goes through self._gcrotmk (:118-119, its default
                    # "Jacobi" included); its "BICGSTAB" branch (:116-117) calls a method whose body reads names that are not
                    # in its scope (:377-382, a NameError upstream) and whose last lines are the same gcrotmk call — here it
                    # runs GCROT(m,k) as well.  "RICHARDSON" (not a value of the reference) selects the damped sweeps of its
                    # _jacobi method (:252-291), which no reference driver reaches
                    if self.linear_solver.upper() == "RICHARDSON":
                        solver = lambda b, e=e: (self._jacobi_device(ops, matvec, minv[e], b), 0)
                    else:
                        def psolve(v, e=e):
                            out = _CVec(ctx.empty((n,)), ctx.empty((n,)))
                            ctx.cmul(minv[e].re, minv[e].im, v.re, v.im, out.re, out.im)
                            return out
                        solver = lambda b: self._gcrotmk_device(ops, matvec, psolve, b, rtol=1e-4, maxiter=self.ls_max_iter)
                    ph = self.e_r * np.exp(1j * theta[e])
                    for l in range(m):
                        self._matvecs = 0
                        qe, info = solver(_CVec(us[l], zero))
                        self.linear_solver_info.append((info, self._matvecs))
                        print_logging_info("Linear Solver Info = ", info, level=2)
                        # Q_l -= w/2 Re(ph Qe)
                        ctx.lincomb(Qs[l], [Qs[l], qe.re, qe.im], [1.0, -w[e] / 2 * ph.real, w[e] / 2 * ph.imag])
                # projected problem (:124-149): H[i,j] = <Q_i, H̄ Q_j>, B[i,j] = <Q_i, Q_j>
                part = lambda v: (self._part(ctx, v, 0, shapes[0]), self._part(ctx, v, n1, shapes[1]))
                self.Q_singles, self.Q_doubles = [part(q)[0] for q in Qs], [part(q)[1] for q in Qs]      # (:104-105; device arrays)
                sigmas = sig.apply_many(self.Q_singles, self.Q_doubles, syms=[False] * m)
                Ws = []
                for s1, s2 in sigmas:
                    wv = ctx.empty((n,))
                    part(wv)[0].copy_from(s1)
                    part(wv)[1].copy_from(s2)
                    Ws.append(wv)
                H = np.zeros((m, m))
                Bm = np.zeros((m, m))
                for j in range(m):
                    H[:, j] = ctx.dots(Qs, [Ws[j]] * m)
                    Bm[j:, j] = ctx.dots(Qs[j:], [Qs[j]] * (m - j))
                Bm = np.tril(Bm) + np.tril(Bm, -1).T
                with _single_threaded_blas():
                    self.eigvals, self.eigvecs = eig(H, Bm)
                self.history.append(np.array(self.eigvals))
                vr = np.real(self.eigvecs)
                if m < self.n_trial:                                               # :152-160
                    for l in range(m):
                        new = ctx.empty((n,))
                        ctx.lincomb(new, Qs, list(vr[:, l]))
                        us.append(new)
                else:                                                              # :161-165
                    for l in range(m):
                        ctx.lincomb(us[l], [us[l]] + Qs, [1.0] + list(vr[:, l]))
                e_norm = np.linalg.norm(self.eigvals)
                if np.abs(e_norm - e_norm_prev) < self.tol:
                    break
                print_logging_info(f"Iter = {it}, Eigenvalues: {self.eigvals}", level=1)
                print_logging_info(f"Norm of eigenvalues: {e_norm}, Difference: {np.abs(e_norm - e_norm_prev)}", level=1)
                print_logging_info("Took {:.3f} seconds ".format(time.time() - time_iter_init), level=2)
                e_norm_prev = e_norm
            self.iterations = it + 1
            self.u_singles = [self._part(ctx, u, 0, shapes[0]) for u in us]
            self.u_doubles = [self._part(ctx, u, n1, shapes[1]) for u in us]
            if not device_form:          # (device form: the trial space stays in the caller's context)
                self.u_singles = [x.get() for x in self.u_singles]
                self.u_doubles = [x.get() for x in self.u_doubles]
                self.Q_singles = [x.get() for x in self.Q_singles]
                self.Q_doubles = [x.get() for x in self.Q_doubles]
        finally:
            collector.__exit__()
            if not device_form:
                ctx.close()
        print_logging_info(f"FEAST-EOM-CCSD finished in {time.time() - time_init:.2f} seconds.", level=0)
        self.e_excit = self.eigvals
        return self.eigvals

    # ---- the reference's host-array call forms of the linear solvers (:252-350) ----------------------------------------------
    def _host_linear_solve(self, which, l, ze, diag_ai, diag_abij, f, dict_t_V, t_T_abij, phase=None, hs=1.0):
        """``hs``: the factor of H̄ in the operator ze - hs H̄ (1j dt in the real-time form).  The preconditioner follows the
        reference: 1 / (ze - hs diag + 0.01) for _jacobi (:276-278, :288-289), the UNSCALED 1 / (ze - diag + 0.01) for
        _gcrotmk in either form (:342 reads diag_ai / diag_abij, not the shifted copies of :301-302)."""
        no = self.no
        nv = diag_ai.shape[0]
        n1, n = diag_ai.size, diag_ai.size + diag_abij.size
        shapes = (tuple(diag_ai.shape), tuple(diag_abij.shape))
        ctx = self._context(dict_t_V, nv)
        try:
            sig = _Sigma(ctx, np.asarray(f, dtype=np.float64), ctx.array(t_T_abij))
            ops = _Ops(ctx, n)
            b = np.concatenate((np.asarray(self.u_singles[l]).ravel(), np.asarray(self.u_doubles[l]).ravel())).astype(complex)
            if phase is not None:
                b = b * phase
            bv = _CVec(ctx.array(np.ascontiguousarray(b.real)), ctx.array(np.ascontiguousarray(b.imag)))
            mv = 1.0 / (ze - (hs if which == "jacobi" else 1.0) * np.concatenate((diag_ai.ravel(), diag_abij.ravel())) + 0.01)
            minv = _CVec(ctx.array(np.ascontiguousarray(mv.real)), ctx.array(np.ascontiguousarray(mv.imag)))
            matvec = lambda v: self._matvec(sig, ops, ze, v, n1, shapes, hs)
            self._matvecs = 0
            if which == "jacobi":
                q = self._jacobi_device(ops, matvec, minv, bv)
            else:
                def psolve(v):
                    out = _CVec(ctx.empty((n,)), ctx.empty((n,)))
                    ctx.cmul(minv.re, minv.im, v.re, v.im, out.re, out.im)
                    return out
                q, info = self._gcrotmk_device(ops, matvec, psolve, bv, rtol=1e-4, maxiter=self.ls_max_iter)
                print_logging_info("Linear Solver Info = ", info, level=2)
            qh = q.re.get() + 1j * q.im.get()
            return qh[:n1].reshape(shapes[0]), qh[n1:].reshape(shapes[1])
        finally:
            ctx.close()

    @staticmethod
    def _h_scale(is_rt, dt):
        """The reference's real-time switch (:197-200, :321-334): H̄ enters as 1j dt H̄ when ``is_rt and dt is not None``."""
        return 1j * dt if (is_rt and dt is not None) else 1.0

    def _gcrotmk(self, l, ze, diag_ai, diag_abij, t_fock_dressed_pq, dict_t_V_dressed, t_T_abij, phase=None, is_rt=False,
                 dt=None, **kwargs):
        """feast_eom_ccsd.py:293-350: (ze - H̄) Q = phase u_l, or (ze - 1j dt H̄) Q = phase u_l in the real-time form
        (``is_rt`` / ``dt``, :321-334 — the hook of rt_eom_ccsd.py, whose own driver does not run upstream)."""
        return self._host_linear_solve("gcrotmk", l, ze, diag_ai, diag_abij, t_fock_dressed_pq, dict_t_V_dressed, t_T_abij, phase,
                                       self._h_scale(is_rt, dt))

    def _jacobi(self, l, ze, diag_ai, diag_abij, t_fock_dressed_pq, dict_t_V_dressed, t_T_abij, phase=None, is_rt=False,
                dt=None, **kwargs):
        """feast_eom_ccsd.py:252-291 (real-time form: the diagonal shift is 1j dt diag, :276-278)."""
        return self._host_linear_solve("jacobi", l, ze, diag_ai, diag_abij, t_fock_dressed_pq, dict_t_V_dressed, t_T_abij, phase,
                                       self._h_scale(is_rt, dt))

    def get_residual(self, l, ze, trial_singles, trial_doubles, t_fock_dressed_pq, dict_t_V_dressed, t_T_abij, phase=None,
                     is_rt=False, dt=None):
        """feast_eom_ccsd.py:183-218: u_l phase - ze Q + H̄ Q (real-time form: + 1j dt H̄ Q, :197-200, :211-214) for host
        arrays."""
        ph = 1.0 if phase is None else phase
        hs = self._h_scale(is_rt, dt)
        s1 = self.update_singles(t_fock_dressed_pq, dict_t_V_dressed, trial_singles, trial_doubles, t_T_abij)
        s2 = self.update_doubles(t_fock_dressed_pq, dict_t_V_dressed, trial_singles, trial_doubles, t_T_abij)
        return (self.u_singles[l] * ph - ze * trial_singles + hs * s1, self.u_doubles[l] * ph - ze * trial_doubles + hs * s2)
