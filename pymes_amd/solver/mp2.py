"""MP2 amplitudes/energy = iteration-0 guess (pymes/solver/mp2.py:9-22)."""
import numpy as np

from pymes_amd.device import Context


def solve(t_epsilon_i, t_epsilon_a, t_V_ijab, t_V_abij, leve_shift=0.0, **kwargs):
    """Same call form as the reference: host blocks in, [e_total, t_T_abij] out.
    (The keyword is spelled ``leve_shift`` in the reference, mp2.py:9.)"""
    no, nv = len(t_epsilon_i), len(t_epsilon_a)
    ctx = kwargs.get("ctx") or Context(no, nv, device=kwargs.get("device", 0))
    try:
        ctx.set_V_block("ijab", np.ascontiguousarray(t_V_ijab))
        ctx.set_V_block("abij", np.ascontiguousarray(t_V_abij))
        ctx.set_orbital_energies(t_epsilon_i, t_epsilon_a)
        t2 = ctx.empty((nv, nv, no, no))
        e_dir, e_exc = ctx.mp2(t2, leve_shift)
        return [e_dir + e_exc, t2.get()]
    finally:
        if "ctx" not in kwargs:
            ctx.close()
