"""numpy oracle of the FEAST-EOM-CCSD driver (pymes/solver/feast_eom_ccsd.py:72-181, :293-350).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference draws real random trial vectors, integrates the resolvent (z - H̄)^-1 over the upper half of the circle
|z - e_c| = e_r with an 8-point Gauss-Legendre rule, solves every (z_e - H̄) Q = Y with scipy's flexible GCROT(m,k)
(diagonal preconditioner, at most ``ls_max_iter`` outer cycles, relative tolerance 1e-4) and diagonalises H̄ in the span of
the real parts.  Its trial space is not the textbook one: it starts from ``n_excit`` = 2 vectors and, while it holds fewer
than ``n_trial`` vectors, APPENDS the Ritz vectors (2 -> 4 -> 8 ...), afterwards it ADDS them onto the stored vectors
(:158-171).  All of that is restated here as it is.

Third-party arithmetic: the linear solver is ``scipy.sparse.linalg.gcrotmk`` (scipy is unpinned in the reference's setup.py;
1.15.3 in the build container) — E. de Sturler, SIAM J. Sci. Comput. 20, 864 (1999); J. Hicken and D. Zingg, SIAM J. Sci.
Comput. 32, 1672 (2010) — called here exactly as the reference calls it (:344), except for two API changes of newer scipy
that its call does not survive: ``tol=`` is now ``rtol=``, and a ``LinearOperator`` without ``dtype`` is probed with an
int8 vector, which the in-place updates of ``update_singles`` reject (oracle/make_golden_feast.py shims exactly these two
entry points to run the reference itself).  The sigma build is oracle/eom_oracle.py (pinned by make_golden_eom.py).
"""
import numpy as np
from scipy.linalg import eig
from scipy.sparse import diags
from scipy.sparse.linalg import LinearOperator, gcrotmk

from . import eom_oracle as eo


def normalize(u1, u2):
    """feast_eom_ccsd.py:625-630."""
    n = np.sqrt(np.vdot(u1, u1) + np.vdot(u2, u2))
    return u1 / n, u2 / n


def quadrature(e_c, e_r, n=8):
    """:98-100 — nodes on the upper half circle, theta from 0 (x = 1) to pi (x = -1)."""
    x, w = np.polynomial.legendre.leggauss(n)
    theta = -np.pi / 2 * (x - 1)
    return theta, w, e_c + e_r * np.exp(1j * theta)


def linear_solve(no, fd, Vd, t2, ze, d1, d2, b1, b2, ls_max_iter=20, rtol=1e-4, sigma=None):
    """:293-350 — (ze - H̄) Q = b by preconditioned GCROT(m,k) from a zero start."""
    sigma = sigma or (lambda u1, u2: (eo.sigma_singles(no, fd, Vd, u1, u2, t2), eo.sigma_doubles(no, fd, Vd, u1, u2, t2)))
    n1 = d1.size

    def matvec(q):
        u1, u2 = q[:n1].reshape(d1.shape), q[n1:].reshape(d2.shape)
        s1, s2 = sigma(u1, u2)
        return np.concatenate(((ze * u1 - s1).ravel(), (ze * u2 - s2).ravel()))
    n = n1 + d2.size
    A = LinearOperator((n, n), matvec=matvec, dtype=complex)
    M = diags(np.concatenate((1.0 / (ze - d1.ravel() + 0.01), 1.0 / (ze - d2.ravel() + 0.01))), offsets=0)
    b = np.concatenate((b1.ravel(), b2.ravel())).astype(complex)
    q, info = gcrotmk(A, b, x0=np.zeros(n, dtype=complex), M=M, maxiter=ls_max_iter, rtol=rtol)
    return q[:n1].reshape(d1.shape), q[n1:].reshape(d2.shape), info


def feast_solve(no, fd, Vd, t2, e_c=0.0, e_r=1.0, n_trial=5, max_iter=20, tol=1e-12, ls_max_iter=20, rand=None):
    """:72-181.  ``rand(*shape)``: the uniform draws (np.random.rand in the reference, :90-91: singles then doubles, vector by
    vector).  Returns {"eigvals", "history" (eigenvalues of every pass), "iterations"}."""
    rand = rand or np.random.rand
    d1, d2 = eo.diag_singles(no, fd, Vd, t2), eo.diag_doubles(no, fd, Vd, t2)
    us = []
    for _ in range(2):                                                        # n_excit = 2 (:53)
        a = 0.5 - rand(*d1.shape)
        b = (0.5 - rand(*d2.shape)) * 0.01
        us.append(normalize(a, b))
    theta, w, z = quadrature(e_c, e_r)

    def sigma(u1, u2):
        return eo.sigma_singles(no, fd, Vd, u1, u2, t2), eo.sigma_doubles(no, fd, Vd, u1, u2, t2)
    prev, history, eigvals = 1e10, [], None
    it = 0
    for it in range(max_iter):
        us = [normalize(a, b) for a, b in us]                                 # :109-110
        Q = [[np.zeros(d1.shape), np.zeros(d2.shape)] for _ in us]
        for e in range(len(z)):                                               # :113-121
            for l, (a, b) in enumerate(us):
                q1, q2, _ = linear_solve(no, fd, Vd, t2, z[e], d1, d2, a, b, ls_max_iter, sigma=sigma)
                ph = e_r * np.exp(1j * theta[e])
                Q[l][0] = Q[l][0] - w[e] / 2 * np.real(ph * q1)
                Q[l][1] = Q[l][1] - w[e] / 2 * np.real(ph * q2)
        m = len(us)
        W = [sigma(q1, q2) for q1, q2 in Q]                                   # :128-134
        H = np.array([[np.vdot(Q[i][0], W[j][0]) + np.vdot(Q[i][1], W[j][1]) for j in range(m)] for i in range(m)])
        B = np.array([[np.vdot(Q[i][0], Q[j][0]) + np.vdot(Q[i][1], Q[j][1]) for j in range(m)] for i in range(m)])
        B = np.tril(B) + np.tril(B, -1).T                                     # B[j,i] = B[i,j] for j < i (:141-142)
        eigvals, vecs = eig(H, B)                                             # :149
        if m < n_trial:                                                       # :152-160
            for l in range(m):
                us.append((sum(np.real(vecs[i, l]) * Q[i][0] for i in range(m)),
                           sum(np.real(vecs[i, l]) * Q[i][1] for i in range(m))))
        else:                                                                 # :161-165
            for l in range(m):
                us[l] = (us[l][0] + sum(np.real(vecs[i, l]) * Q[i][0] for i in range(m)),
                         us[l][1] + sum(np.real(vecs[i, l]) * Q[i][1] for i in range(m)))
        history.append(np.array(eigvals))
        e_norm = np.linalg.norm(eigvals)
        if abs(e_norm - prev) < tol:
            break
        prev = e_norm
    return {"eigvals": eigvals, "history": history, "iterations": it + 1}
