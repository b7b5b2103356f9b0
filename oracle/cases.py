"""Seeded input generators shared by oracle/make_golden.py and tests/.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import numpy as np

from .io_oracle import eri_from_factors, synthetic_factors


def random_case(no, nv, seed, symmetric=False, amp=0.1):
    """Random Fock/V/T1/T2 for per-function parity.  ``symmetric=False`` gives a
    V with no permutational symmetry at all (stress-tests index bookkeeping;
    the TC Hamiltonian of the reference has only V_pqrs = V_qpsr)."""
    rng = np.random.default_rng(seed)
    n = no + nv
    if symmetric:
        B = rng.standard_normal((2 * n, n, n)) * (0.3 / np.sqrt(2 * n))
        B = 0.5 * (B + B.transpose(0, 2, 1))
        V = eri_from_factors(B)
    else:
        V = rng.standard_normal((n, n, n, n)) * 0.1
    f = rng.standard_normal((n, n)) * 0.05
    f = 0.5 * (f + f.T)
    f[np.diag_indices(n)] = np.concatenate([np.sort(-1.0 - rng.random(no)), np.sort(1.0 + rng.random(nv))])
    t1 = rng.standard_normal((nv, no)) * amp
    t2 = rng.standard_normal((nv, nv, no, no)) * amp
    t2 = 0.5 * (t2 + t2.transpose(1, 0, 3, 2))
    return f, V, t1, t2


def synthetic_case(no, nv, seed=0, scale=0.3, gap=3.0):
    """SURVEY §8(d) synthetic closed-shell problem: f = diag(eps), 8-fold symmetric PSD V."""
    B, eps = synthetic_factors(no, nv, seed, scale, gap)
    return np.diag(eps), eri_from_factors(B), B, eps


def eom_sigma_case(no, nv, seed, scale):
    """Inputs of one EOM-CCSD sigma build at a benchmark size (config 5): synthetic 8-fold symmetric V standing in for
    the dressed integrals, a dressed-like (non-diagonal) Fock matrix, exchange-symmetric T2 and trial vector (u1, u2) —
    what ``EOM_CCSD.solve`` hands to ``update_singles`` / ``update_doubles`` (eom_ccsd.py:95-101).
    Returns (f~, V_pqrs, T2, u1, u2)."""
    rng = np.random.default_rng(seed)
    f, V, _, _ = synthetic_case(no, nv, seed=0, scale=scale)
    n = no + nv
    fd = f + 0.02 * rng.standard_normal((n, n))
    t2 = rng.standard_normal((nv, nv, no, no)) * 0.02
    t2 = 0.5 * (t2 + t2.transpose(1, 0, 3, 2))
    u1 = rng.standard_normal((nv, no)) * 0.3
    u2 = rng.standard_normal((nv, nv, no, no)) * 0.05
    u2 = 0.5 * (u2 + u2.transpose(1, 0, 3, 2))
    return fd, V, t2, u1, u2


def eom_davidson_case(no, nv, seed=0, scale=0.3):
    """A synthetic closed-shell problem on which the reference's Davidson driver (eom_ccsd.py:46-167) CONVERGES: V as in
    ``synthetic_case``; orbital energies with hand-set frontier levels — the three lowest single excitations out of the HOMO
    are isolated (3.0, 3.4, 3.85; everything else above 4.1).  The driver preconditions with one number per root
    (e - D_ai[guess] + 1e-5, :139), i.e. it is a restarted block Krylov method, and on the dense spectrum of the SURVEY 8(d)
    orbital energies (spacing 0.02-0.08) it stalls at |dE| ~ 1e-5 for hundreds of passes, until rounding noise in the
    exchange-antisymmetric doubles — whose Ritz values are O(s^2), below every physical root — takes the subspace over.
    Returns (f = diag(eps), V_pqrs)."""
    _, V, _, _ = synthetic_case(no, nv, seed=seed, scale=scale)
    rng = np.random.default_rng(seed + 5)
    eps_o = np.sort(np.concatenate([[-1.5], -2.7 - 0.8 * rng.random(no - 1)]))
    eps_v = np.sort(np.concatenate([[1.5, 1.9, 2.35], 3.2 + 1.0 * rng.random(nv - 3)]))
    return np.diag(np.concatenate([eps_o, eps_v])), V
