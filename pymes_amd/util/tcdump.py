"""TCDUMP reader (pymes/util/tcdump.py:30-139): the explicit three-body integrals of a transcorrelated
Hamiltonian as a dense [nb]^6 tensor in the chemists' order (or|ps|qt).

Text format: first line = number of orbitals, then ``value o p q r s t`` (1-based, physicists' order).
The stored value is -3 * value (tcdump.py:75) and the six simultaneous permutations of (o,p,q) / (r,s,t)
are filled in (``restore_6_fold_sym``, :116-139).  The file is parsed vectorised on the host; the dense fill
runs on the device (``pymes_scatter``).  hdf5 input needs h5py like the reference and is refused without it.
"""
import ctypes as C
import itertools

import numpy as np

from pymes_amd.device import Context
from pymes_amd.log import print_logging_info

_PERMS = list(itertools.permutations(range(3)))


def _parse_txt(file_name):
    """tcdump.py:58-86 -> (flat target indices, values, nb): targets are unique, a later line wins."""
    with open(file_name, "r") as reader:
        nb = int(reader.readline().strip())
        rows = []
        for line in reader:
            if not line.strip():
                break
            tok = line.split()
            if len(tok) != 7:
                raise ValueError("not enough values to unpack (expected 7): " + line.rstrip())
            rows.append(tok)
    if not rows:
        return np.zeros(0, dtype=np.int64), np.zeros(0), nb
    vals = -3.0 * np.array([float(r[0]) for r in rows])
    idx = np.array([[int(x) - 1 for x in r[1:]] for r in rows], dtype=np.int64)
    if idx.min() < 0 or idx.max() >= nb:
        raise IndexError("TCDUMP orbital index out of range")
    flat = []
    for perm in _PERMS:                                             # restore_6_fold_sym, in its order
        a, b = idx[:, list(perm)], idx[:, [3 + k for k in perm]]
        six = np.stack([a[:, 0], b[:, 0], a[:, 1], b[:, 1], a[:, 2], b[:, 2]], axis=1)
        flat.append(np.ravel_multi_index(six.T, (nb,) * 6))
    flat = np.stack(flat, axis=1).ravel()                           # entry-major, permutation-minor: file order
    v = np.repeat(vals, len(_PERMS))
    # the reference assigns sequentially, so the LAST write to a target survives
    _, first_of_reversed = np.unique(flat[::-1], return_index=True)
    keep = np.sort(len(flat) - 1 - first_of_reversed)
    return flat[keep], v[keep], nb


def read_to_device(ctx, file_name="TCDUMP"):
    """Dense L[nb]^6 as a DeviceArray of ``ctx``."""
    if "h5" in file_name or "hdf5" in file_name:
        raise ImportError("hdf5 TCDUMP files need h5py, which is not available here")
    flat, vals, nb = _parse_txt(file_name)
    L = ctx.zeros((nb,) * 6)
    ctx.lib.call("pymes_scatter", ctx.handle, C.c_void_p(L.ptr), int(nb) ** 6,
                 flat.ctypes.data_as(C.POINTER(C.c_int64)), vals.ctypes.data_as(C.POINTER(C.c_double)), len(flat))
    return L


def read(file_name="TCDUMP", sym=True, sp=1, device=0):
    """tcdump.read: returns the dense numpy tensor t_V_orpsqt like the reference (``sym``/``sp`` are CTF-era
    switches without effect there either)."""
    print_logging_info("Reading in TCDUMP", level=1)
    if "h5" in file_name or "hdf5" in file_name:
        print_logging_info("Integral file in hdf5 format.", level=1)
    else:
        print_logging_info("Assuming integral file in txt format.", level=1)
    ctx = Context(1, 1, device=device, workspace_bytes=1 << 20)
    try:
        return read_to_device(ctx, file_name).get()
    finally:
        ctx.close()


def unique_index(p, q):
    """tcdump.py:113-114."""
    return int(min(p, q) + (max(p, q) - 1) * max(p, q) / 2)
