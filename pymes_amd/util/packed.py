"""Packed binary integral files ("PYMESPK1", pymes_amd/csrc/packed.h): the on-disk form of the 16 partition.py blocks
or of density-fitting factors, read straight into the device blocks.

The reference reads integrals only as FCIDUMP text, one Python loop iteration per line (pymes/util/fcidump.py:124-161;
its own comment puts the limit at ~300 orbitals), plus an hdf5 branch in the TCDUMP reader (tcdump.py:44-48,88-92).  The
return tuples here are those of ``fcidump.read`` (``pymes_amd/util/fcidump.py``), so a driver switches formats by
switching the reader."""
import ctypes as C

import numpy as np

from pymes_amd import _lib
from pymes_amd.device import BLOCK_NAMES, Context, pattern_of
from pymes_amd.integral.device import DeviceIntegrals
from pymes_amd.log import print_logging_info

HEADER_BYTES = 64
KIND_BLOCKS, KIND_FACTORS = 1, 2
_NAME_OF_PATTERN = {pattern_of(nm): nm for nm in BLOCK_NAMES}


def _wrap(fn, *args):
    try:
        fn(*args)
    except _lib.PymesError as exc:
        msg = str(exc)
        if "cannot open" in msg:
            raise FileNotFoundError(msg) from None
        raise ValueError(msg) from None


def header(path):
    """(kind, n_elec, n_orb, naux)."""
    lib = _lib.default_library()
    kind, ne, n, naux = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    _wrap(lib.call, "pymes_packed_header", path.encode(), C.byref(kind), C.byref(ne), C.byref(n), C.byref(naux))
    return kind.value, ne.value, n.value, naux.value


def write_packed(path, n_elec, e_core, epsilon_p, h_pq, integrals, device=0):
    """``integrals``: a dense host ``V_pqrs`` (uploaded and packed first) or ``DeviceIntegrals`` (written from HBM)."""
    own = not isinstance(integrals, DeviceIntegrals)
    ints = DeviceIntegrals.from_V_pqrs(n_elec // 2, integrals, device=device) if own else integrals
    try:
        eps = np.ascontiguousarray(epsilon_p, dtype=np.float64)
        h = np.ascontiguousarray(h_pq, dtype=np.float64)
        if eps.shape != (ints.ctx.n,) or h.shape != (ints.ctx.n, ints.ctx.n):
            raise ValueError("epsilon_p / h_pq do not match the integrals")
        _wrap(ints.ctx.lib.call, "pymes_packed_write", ints.ctx.handle, path.encode(), int(n_elec), float(e_core),
              _lib.host_ptr(eps), _lib.host_ptr(h))
    finally:
        if own:
            ints.ctx.close()


def write_factors(path, n_elec, e_core, epsilon_p, h_pq, B):
    """Density-fitted form: B[naux,n,n] with V[p,q,r,s] = sum_Q B[Q,p,r] B[Q,q,s] (n^2 naux numbers instead of n^4)."""
    B = np.ascontiguousarray(B, dtype=np.float64)
    eps = np.ascontiguousarray(epsilon_p, dtype=np.float64)
    h = np.ascontiguousarray(h_pq, dtype=np.float64)
    n = B.shape[1]
    if B.ndim != 3 or B.shape[2] != n or eps.shape != (n,) or h.shape != (n, n):
        raise ValueError("B must be [naux,n,n], epsilon_p [n], h_pq [n,n]")
    _wrap(_lib.default_library().call, "pymes_packed_write_factors", path.encode(), int(n_elec), n, int(B.shape[0]),
          float(e_core), _lib.host_ptr(eps), _lib.host_ptr(h), _lib.host_ptr(B))


def read_packed(path):
    """Host form, the tuple of ``fcidump.read``: (n_elec, n_orb, e_core, epsilon_p, h_pq, V_pqrs) with a dense V."""
    kind, n_elec, n, naux = header(path)
    no = n_elec // 2
    with open(path, "rb") as f:
        f.seek(40)
        e_core = float(np.fromfile(f, dtype="<f8", count=1)[0])
        f.seek(HEADER_BYTES)
        eps = np.fromfile(f, dtype="<f8", count=n)
        h = np.fromfile(f, dtype="<f8", count=n * n).reshape(n, n)
        if kind == KIND_FACTORS:
            B = np.fromfile(f, dtype="<f8", count=naux * n * n).reshape(naux, n, n)
            V = np.einsum("Qpr,Qqs->pqrs", B, B, optimize=True)
        else:
            V = np.empty((n, n, n, n))
            sl = {False: slice(0, no), True: slice(no, n)}
            for pat in range(16):
                virt = [bool(pat >> (3 - i) & 1) for i in range(4)]
                shape = tuple(n - no if v else no for v in virt)
                blk = np.fromfile(f, dtype="<f8", count=int(np.prod(shape))).reshape(shape)
                V[tuple(sl[v] for v in virt)] = blk
    return n_elec, n, e_core, eps, h, V


def read_packed_to_device(path, device=0, **ctx_kwargs):
    """(n_elec, n_orb, e_core, epsilon_p, h_pq, DeviceIntegrals): the payload goes block by block into HBM (factors are
    expanded by the fp64 MFMA GEMM); V_pqrs never exists as a whole, neither on the host nor on the device."""
    kind, n_elec, n, naux = header(path)
    no = n_elec // 2
    ctx = Context(no, n - no, device=device, **ctx_kwargs)
    try:
        e_core = C.c_double()
        eps, h = np.zeros(n), np.zeros((n, n))
        _wrap(ctx.lib.call, "pymes_packed_load", ctx.handle, path.encode(), C.byref(e_core), _lib.host_ptr(eps),
              _lib.host_ptr(h))
    except Exception:
        ctx.close()
        raise
    print_logging_info("Read packed integrals %s (%s) into the device blocks" %
                       (path, "factors, naux=%d" % naux if kind == KIND_FACTORS else "16 blocks"), level=1)
    return n_elec, n, e_core.value, eps, h, DeviceIntegrals(ctx)
