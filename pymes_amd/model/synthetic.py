"""Seeded synthetic closed-shell integrals for benchmarks (the reference has no
generator; recipe of SURVEY §8(d), validated there against the reference solver).

(pr|qs) = sum_Q B_Qpr B_Qqs with B symmetric in (p,r): an 8-fold symmetric, positive
semi-definite ERI tensor in factorised (density-fitting) form; f = diag(eps) with a gap.
"""
import numpy as np


def factors(no, nv, seed=0, scale=None, gap=3.0):
    n = no + nv
    naux = 2 * n
    if scale is None:
        scale = 1.5 / np.sqrt(n)
    rng = np.random.default_rng(seed)
    B = rng.standard_normal((naux, n, n)) * (scale / np.sqrt(naux))
    B = 0.5 * (B + B.transpose(0, 2, 1))
    eps = np.concatenate([np.sort(-gap / 2 - rng.random(no)), np.sort(gap / 2 + rng.random(nv))])
    return B, eps


def dense_eri(B):
    """V[p,q,r,s] = (pr|qs) on the host (small n only)."""
    return np.einsum("Qpr,Qqs->pqrs", B, B, optimize=True)
