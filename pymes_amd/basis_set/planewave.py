"""Plane-wave basis function record (pymes/basis_set/planewave.py:3-26)."""
import numpy as np


class BasisFunc:
    """Plane wave with wavevector 2*pi*(i,j,k)/L and spin +1/-1; ordered by kinetic energy."""

    def __init__(self, i, j, k, L, spin, k_shift=(0., 0., 0.)):
        if spin not in (-1, 1):
            raise RuntimeError('spin not +1 or -1')
        self.k = np.array([i, j, k])
        self.L = L
        self.kp = (self.k + np.asarray(k_shift)) * 2 * np.pi / L
        self.kinetic = np.dot(self.kp, self.kp) / 2.        # this exact float expression fixes the shell order
        self.spin = spin

    def __repr__(self):
        return (self.k, self.kinetic, self.spin).__repr__()

    def __lt__(self, other):
        return self.kinetic < other.kinetic
