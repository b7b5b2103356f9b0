"""Hartree-Fock helpers of the caller side (pymes/mean_field/hf.py); O(n^3) host work."""
import numpy as np


def calc_hf_e(no, e_core, t_h_pq, t_V_pqrs):
    """pymes/mean_field/hf.py:5-11."""
    occ = t_V_pqrs[:no, :no, :no, :no]
    return 2.0 * np.trace(t_h_pq[:no, :no]) + 2.0 * np.einsum("jiji->", occ) - np.einsum("ijji->", occ) + e_core


def construct_hf_matrix(no, t_h_pq, t_V_pqrs):
    """pymes/mean_field/hf.py:14-18: f = h + 2 V_piqi - V_piiq.  ``t_V_pqrs`` may be a
    ``pymes_amd.integral.device.DeviceIntegrals`` (e.g. from ``fcidump.read_to_device``): the traces then run on the
    device blocks (``pymes_hf_fock_matrix``) and V_pqrs is never needed on the host."""
    if hasattr(t_V_pqrs, "ctx"):
        import ctypes as C
        ctx = t_V_pqrs.ctx
        if ctx.no != no:
            raise ValueError("number of occupied orbitals does not match the device integrals")
        h = np.ascontiguousarray(t_h_pq, dtype=np.float64)
        f = np.empty_like(h)
        ctx.lib.call("pymes_hf_fock_matrix", ctx.handle, h.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p))
        return f
    f = np.array(t_h_pq, dtype=np.float64, copy=True)
    f += 2.0 * np.einsum("piqi->pq", t_V_pqrs[:, :no, :, :no])
    f -= np.einsum("piiq->pq", t_V_pqrs[:, :no, :no, :])
    return f


def calcOccupiedOrbE(kinetic_G, tV_ijkl, no):
    """pymes/mean_field/hf.py:21-30."""
    return kinetic_G[0:no] + 2.0 * np.einsum("ijij->i", tV_ijkl) - np.einsum("ijji->i", tV_ijkl)


def calcVirtualOrbE(kinetic_G, t_V_aibj, t_V_aijb, no, nv):
    """pymes/mean_field/hf.py:33-43."""
    return kinetic_G[no:] + 2.0 * np.einsum("aiai->a", t_V_aibj) - np.einsum("aiia->a", t_V_aijb)
