"""numpy oracle of the explicit 3-body (transcorrelated) path: TCDUMP reader (pymes/util/tcdump.py:30-139)
and the mean-field foldings of the 3-body operator (pymes/integral/contraction.py:17-95).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restated with explicit index loops / diagonal slicing instead of the reference's einsum strings.
Pinned against the imported reference by oracle/make_golden_tc.py -> tests/golden/tc.json.
"""
import itertools

import numpy as np


def read_tcdump(path):
    """tcdump.py:30-97: first line = number of orbitals, then ``value o p q r s t`` (1-based, physicists'
    order).  The stored value is -3 * value; the six simultaneous permutations of (o,p,q) and (r,s,t) are
    written to L[o,r,p,s,q,t] (chemists' order (or|ps|qt)).  Later lines overwrite earlier ones."""
    with open(path) as fh:
        nb = int(fh.readline().split()[0])
        L = np.zeros((nb,) * 6)
        for line in fh:
            tok = line.split()
            if not tok:
                break                                                  # the reference stops at a blank line
            val = -3.0 * float(tok[0])
            idx = [int(x) - 1 for x in tok[1:7]]
            for perm in itertools.permutations(range(3)):
                a = [idx[k] for k in perm]
                b = [idx[3 + k] for k in perm]
                L[a[0], b[0], a[1], b[1], a[2], b[2]] = val
    return L


def single_contraction(no, L):
    """contraction.py:17-39 -> D[p,r,q,s]."""
    nb = L.shape[0]
    D = np.zeros((nb,) * 4)
    for i in range(no):
        x = L[:, :, :, i, i, :]                     # [p,q,r,s]
        D += 0.5 * (-6.0) * (x.transpose(0, 2, 1, 3) + x.transpose(2, 0, 3, 1))   # pqr(ii)s->prqs, rsp(ii)q->prqs
        D += 6.0 * L[:, :, :, :, i, i].transpose(0, 2, 1, 3)                      # pqrs(ii)->prqs
    return -D / 3.0


def double_contraction(no, L):
    """contraction.py:41-65 -> S[p,q]."""
    nb = L.shape[0]
    S = np.zeros((nb, nb))
    for i in range(no):
        for j in range(no):
            S += 12.0 * L[i, i, j, j, :, :]
            S -= 12.0 * L[i, i, :, j, j, :]
            S += 6.0 * L[:, i, j, :, i, j]
            S -= 6.0 * L[i, j, j, i, :, :]
    return -S / 6.0


def triple_contraction(no, L):
    """contraction.py:67-95 -> scalar."""
    t = 0.0
    for i in range(no):
        for j in range(no):
            for k in range(no):
                t += 8.0 * L[i, i, j, j, k, k] - 12.0 * L[i, j, j, i, k, k] + 4.0 * L[i, j, j, k, k, i]
    return -t / 6.0
