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
