"""FCIDUMP text I/O with the semantics of pymes/util/fcidump.py:59-163 (reader) and a
working writer (the reference's writer, fcidump.py:8-56, still calls the removed CTF
API)."""
import ctypes as C

import numpy as np

from pymes_amd import _lib
from pymes_amd.log import print_logging_info


def read(fcidump_file="FCIDUMP", is_tc=False):
    """Returns (n_elec, n_orb, e_core, epsilon_p, h_pq, V_pqrs), V[p,q,r,s] = <pq|rs> = (pr|qs).

    Lines are ``value i j k l`` = (ij|kl); for ``is_tc=False`` the images [r,q,p,s],
    [r,s,p,q], [p,s,r,q] are restored but NOT the electron-exchange image [q,p,s,r]
    (fcidump.py:143-146), for ``is_tc=True`` only [q,p,s,r] (:148-149); |value| < 1e-19 is
    skipped (:138); a blank line in the body is an error, as in the reference.  The text is parsed by
    the native reader behind ``pymes_fcidump_read_host`` (the reference's per-line Python loop takes
    ~1 us per integral)."""
    print_logging_info("Reading " + fcidump_file + "...", level=1)
    print_logging_info("Using TC integrals: ", is_tc, level=2)
    lib = _lib.default_library()
    n_elec, n = _header(lib, fcidump_file)
    e_core = C.c_double()
    eps, h, V = np.zeros(n), np.zeros((n, n)), np.zeros((n, n, n, n))
    _call(lib, "pymes_fcidump_read_host", fcidump_file.encode(), int(bool(is_tc)), C.byref(e_core),
          eps.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p), V.ctypes.data_as(C.c_void_p))
    return n_elec, n, e_core.value, eps, h, V


def read_to_device(fcidump_file="FCIDUMP", is_tc=False, device=0, **ctx_kwargs):
    """Like ``read`` but V_pqrs goes straight into the 16 device blocks of a new context (for NORB > 64 it never exists
    on the host).  Returns (n_elec, n_orb, e_core, epsilon_p, h_pq, DeviceIntegrals)."""
    from pymes_amd.device import Context
    from pymes_amd.integral.device import DeviceIntegrals
    lib = _lib.default_library()
    n_elec, n = _header(lib, fcidump_file)
    no = n_elec // 2
    ctx = Context(no, n - no, device=device, **ctx_kwargs)
    try:
        e_core, lines = C.c_double(), C.c_int64()
        eps, h = np.zeros(n), np.zeros((n, n))
        ctx.lib.call("pymes_fcidump_load", ctx.handle, fcidump_file.encode(), int(bool(is_tc)), C.byref(e_core),
                     eps.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p), C.byref(lines))
    except Exception:
        ctx.close()
        raise
    print_logging_info("Read %d two-electron lines of %s into the device blocks" % (lines.value, fcidump_file), level=1)
    return n_elec, n, e_core.value, eps, h, DeviceIntegrals(ctx)


def _call(lib, name, *args):
    try:
        lib.call(name, *args)
    except _lib.PymesError as exc:
        msg = str(exc)
        if "cannot open" in msg:
            raise FileNotFoundError(msg) from None
        raise ValueError(msg) from None


def _header(lib, path):
    n_elec, n = C.c_int(), C.c_int()
    _call(lib, "pymes_fcidump_header", path.encode(), C.byref(n_elec), C.byref(n))
    return n_elec.value, n.value


def write(integrals, h, no, e_nuc=0.0, ms2=0, orbsym=1, isym=1, dtype="r", file="FCIDUMP", threshold=1e-19):
    """Write every |V[p,q,r,s]| >= threshold as ``value p r q s`` (1-based), the one-body
    part as ``value i j 0 0`` (i >= j) and the core energy; ``read`` restores V exactly when V has
    the symmetry the chosen read mode assumes."""
    n = integrals.shape[0]
    with open(file, "w") as f:
        f.write("&FCI NORB=%d,NELEC=%d,MS2=%d,\n" % (n, 2 * no, ms2))
        f.write(" ORBSYM=" + ",".join([str(orbsym)] * n) + ",\n ISYM=%d,\n&END\n" % isym)
        P, Q, R, S = np.nonzero(np.abs(integrals) >= threshold)
        for p, q, r, s in zip(P, Q, R, S):
            f.write(" %.17g %d %d %d %d\n" % (integrals[p, q, r, s], p + 1, r + 1, q + 1, s + 1))
        for i in range(n):
            for j in range(i + 1):
                if abs(h[i, j]) > 1e-19:
                    f.write(" %.17g %d %d 0 0\n" % (h[i, j], i + 1, j + 1))
        f.write(" %.17g 0 0 0 0\n" % e_nuc)
