"""Device-resident two-body integrals: the 16 partition.py blocks packed in HBM.

Pass an instance in place of the dense ``t_V_pqrs`` numpy array to ``CCSD.solve`` /
``CCD.solve`` when the integrals should never exist on the host (at (50 occ, 200 virt)
V_pqrs alone is 31 GB): build them from density-fitting / synthetic factors on the GPU.
"""
import numpy as np

from pymes_amd.device import Context


class DeviceIntegrals:
    def __init__(self, ctx):
        self.ctx = ctx
        self.no, self.nv = ctx.no, ctx.nv
        self.dtype = np.dtype(np.float64)
        self.shape = (ctx.n,) * 4

    @classmethod
    def from_V_pqrs(cls, no, t_V_pqrs, **ctx_kwargs):
        """pymes/integral/partition.py:4-39 on the device (one upload + 16 pack kernels)."""
        V = np.asarray(t_V_pqrs)
        if np.iscomplexobj(V):
            raise NotImplementedError("complex integrals are not supported by the fp64 HIP path")
        n = V.shape[0]
        ctx = Context(no, n - no, **ctx_kwargs)
        ctx.set_V_pqrs(V)
        return cls(ctx)

    @classmethod
    def from_factors(cls, no, B, **ctx_kwargs):
        """V[p,q,r,s] = sum_Q B[Q,p,r] B[Q,q,s], formed block by block with the fp64 MFMA GEMM."""
        n = B.shape[1]
        ctx = Context(no, n - no, **ctx_kwargs)
        ctx.set_V_from_factors(B)
        return cls(ctx)

    def block(self, name, dressed=False):
        return self.ctx.V_block(name, dressed)
