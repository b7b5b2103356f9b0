"""numpy oracle of the integral-side callers of the CC path (FCIDUMP reader,
Hartree-Fock matrix).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import numpy as np


def read_fcidump(path, is_tc=False):
    """pymes/util/fcidump.py:59-163.

    Header: everything up to the first line containing '/' or 'END'; NORB and
    NELEC are found by substring match on comma-separated fields (:108-116).
    Body lines are ``value i j k l`` in chemists' order (ij|kl); the reference
    renames them p,r,q,s (:130) and stores <pq|rs> plus three images
    (:143-146) — the electron-exchange image [q,p,s,r] is NOT restored unless
    ``is_tc`` (:148-149).  |value| < 1e-19 is skipped (:138)."""
    with open(path) as fh:
        head = fh.readline().strip()
        while "/" not in head and "end" not in head.lower():
            head += fh.readline().strip()
        found = {"norb": 0, "nelec": 0}
        for field in head.split(","):
            for key in found:
                if key in field.lower():
                    for word in field.split("="):
                        if word.strip().isdigit():
                            found[key] = int(word.strip())
        n = found["norb"]
        eps, h, V = np.zeros(n), np.zeros((n, n)), np.zeros((n, n, n, n))
        e_core = 0.0
        for line in fh:
            val, p, r, q, s = line.split()
            val, p, r, q, s = float(val), int(p) - 1, int(r) - 1, int(q) - 1, int(s) - 1
            if abs(val) < 1e-19:
                continue
            if min(p, q, r, s) >= 0:
                if is_tc:
                    V[q, p, s, r] = val
                    V[p, q, r, s] = val
                else:
                    V[p, q, r, s] = V[r, q, p, s] = V[r, s, p, q] = V[p, s, r, q] = val
            elif p < 0 and q < 0 and r < 0 and s < 0:
                e_core = val
            elif p >= 0 and q < 0 and r < 0 and s < 0:
                eps[p] = val
            elif p >= 0 and r >= 0 and q < 0 and s < 0:
                h[p, r] = h[r, p] = val
    return found["nelec"], n, e_core, eps, h, V


def hf_energy(no, e_core, h, V):
    """pymes/mean_field/hf.py:5-11."""
    oooo = V[:no, :no, :no, :no]
    return (2.0 * np.trace(h[:no, :no]) + 2.0 * np.einsum("jiji->", oooo)
            - np.einsum("ijji->", oooo) + e_core)


def fock_matrix(no, h, V):
    """pymes/mean_field/hf.py:14-18 — f = h + 2 V_piqi - V_piiq."""
    return h + 2.0 * np.einsum("piqi->pq", V[:, :no, :, :no]) - np.einsum("piiq->pq", V[:, :no, :no, :])


def synthetic_factors(no, nv, seed=0, scale=None, gap=3.0):
    """Seeded synthetic closed-shell problem (the reference has no
    generator).  Returns (B[naux,n,n], eps[n]) with (pr|qs) = sum_Q B_Qpr B_Qqs."""
    n = no + nv
    naux = 2 * n
    if scale is None:
        scale = 1.5 / np.sqrt(n)
    rng = np.random.default_rng(seed)
    B = rng.standard_normal((naux, n, n)) * (scale / np.sqrt(naux))
    B = 0.5 * (B + B.transpose(0, 2, 1))
    eps = np.concatenate([np.sort(-gap / 2 - rng.random(no)), np.sort(gap / 2 + rng.random(nv))])
    return B, eps


def eri_from_factors(B):
    """Physicists' V[p,q,r,s] = (pr|qs) = sum_Q B[Q,p,r] B[Q,q,s]."""
    return np.einsum("Qpr,Qqs->pqrs", B, B, optimize=True)
