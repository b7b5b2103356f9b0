"""Mean-field foldings of the explicit three-body operator -L^{opq}_{rst} of a transcorrelated Hamiltonian
(pymes/integral/contraction.py:17-95): effective two-body integrals, Fock correction and constant.

``t_L`` is the dense [nb]^6 tensor of ``tcdump.read`` (numpy array, uploaded here) or a DeviceArray
(``tcdump.read_to_device``).  The traces over the occupied indices run as HIP kernels
(``pymes_tc_{single,double,triple}_contraction``); results come back as numpy arrays like the reference's.
"""
import ctypes as C

import numpy as np

from pymes_amd.device import Context, DeviceArray
from pymes_amd.log import print_logging_info


def _on_device(t_L, device):
    """(ctx, L, owns_ctx)"""
    if isinstance(t_L, DeviceArray):
        return t_L.ctx, t_L, False
    t_L = np.asarray(t_L)
    if np.iscomplexobj(t_L):
        raise TypeError("complex three-body integrals are not supported")
    if t_L.ndim != 6 or len(set(t_L.shape)) != 1:
        raise ValueError("expected a [nb]^6 tensor")
    ctx = Context(1, 1, device=device, workspace_bytes=1 << 20)
    return ctx, ctx.array(t_L), True


def get_single_contraction(no, t_L_opqrst, device=0):
    """contraction.py:17-39 -> t_D_pqrs [nb,nb,nb,nb]."""
    ctx, L, own = _on_device(t_L_opqrst, device)
    try:
        nb = L.shape[0]
        D = ctx.empty((nb,) * 4)
        ctx.lib.call("pymes_tc_single_contraction", ctx.handle, C.c_void_p(L.ptr), nb, int(no), C.c_void_p(D.ptr))
        return D.get()
    finally:
        if own:
            ctx.close()


def get_double_contraction(no, t_L_opqrst, device=0):
    """contraction.py:41-65 -> t_S_pq [nb,nb]."""
    ctx, L, own = _on_device(t_L_opqrst, device)
    try:
        nb = L.shape[0]
        S = ctx.empty((nb, nb))
        ctx.lib.call("pymes_tc_double_contraction", ctx.handle, C.c_void_p(L.ptr), nb, int(no), C.c_void_p(S.ptr))
        return S.get()
    finally:
        if own:
            ctx.close()


def get_triple_contraction(no, t_L_orpsqt, device=0):
    """contraction.py:67-95 -> float."""
    print_logging_info("Triple contraction")
    ctx, L, own = _on_device(t_L_orpsqt, device)
    try:
        t0 = C.c_double()
        ctx.lib.call("pymes_tc_triple_contraction", ctx.handle, C.c_void_p(L.ptr), L.shape[0], int(no), C.byref(t0))
        return t0.value
    finally:
        if own:
            ctx.close()
