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


class DressedDeviceIntegrals:
    """The T1-dressed blocks of ``get_T1_dressed_V`` (pymes/solver/ccsd.py:290-421) held in HBM, in the context of the
    integrals they were dressed from: what ``CCSD.get_T1_dressed_V(t1, ints)`` returns for a ``DeviceIntegrals`` and what
    ``EOM_CCSD.solve`` / ``FEAST_EOM_CCSD.solve`` take in place of the reference's dictionary of host arrays
    (pymes/test/test_eom_ccsd/test_eom_ccsd.py:24-48: CCSD.solve -> get_T1_dressed_* -> EOM_CCSD.solve) — no block crosses
    PCIe.  Reads like the reference's dictionary: ``d["ijab"]`` is a DeviceArray, the five undressed names are None."""

    def __init__(self, ints, keys):
        self.ints, self.ctx = ints, ints.ctx
        self.no, self.nv = ints.no, ints.nv
        self._keys = tuple(keys)
        # The object ALIASES the context's dressed storage: any later dressing on the same context (another
        # get_T1_dressed_V, a CCSD iteration) overwrites what it stands for.  The generation of the dressing it belongs to:
        self._generation = getattr(ints.ctx, "dress_generation", 0)

    def require(self, names):
        """Raise unless every block of ``names`` was dressed for this object and the context's dressed storage still holds
        that dressing (``get_T1_dressed_V(t1, ints, subset)`` may have dressed fewer blocks; a later dressing on the same
        ``DeviceIntegrals`` replaces them underneath)."""
        missing = [nm for nm in names if nm not in self._keys]
        if missing:
            raise KeyError("T1-dressed blocks %s were not requested from get_T1_dressed_V" % ", ".join(missing))
        if getattr(self.ctx, "dress_generation", 0) != self._generation:
            raise RuntimeError("the dressed integrals are stale: the context's blocks have been dressed again since "
                               "(another get_T1_dressed_V / CCSD iteration on the same DeviceIntegrals)")

    def keys(self):
        from pymes_amd.integral.partition import BLOCK_NAMES
        return BLOCK_NAMES

    def __contains__(self, name):
        return name in self.keys()

    def __getitem__(self, name):
        if name not in self.keys():
            raise KeyError(name)
        return self.ctx.V_block(name, dressed=True) if name in self._keys else None

    def get(self, name, default=None):
        return self[name] if name in self else default

    def to_host(self):
        """The reference's dictionary (host arrays) — for callers that leave the device."""
        return {k: (self[k].get() if k in self._keys else None) for k in self.keys()}
