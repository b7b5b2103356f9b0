"""Device context and device-resident fp64 arrays on top of the C-ABI (include/pymes_amd.h).

``Context`` owns one ``pymes_ctx`` (one GPU, one HIP stream) for a fixed (no, nv).
``DeviceArray`` is a thin handle (pointer + shape); numpy arrays enter and leave the
device only through ``Context.array`` / ``DeviceArray.get``.
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import PymesError, i64_array, ptr_array

# partition.py block name <-> pattern id (bit 3-pos set when index `pos` is virtual)
BLOCK_NAMES = ("abci", "iabj", "iajk", "aijk", "klij", "aibj", "ijak", "abic",
               "iajb", "abcd", "iabc", "aijb", "ijka", "aibc", "ijab", "abij")


def pattern_of(name):
    return sum((1 << (3 - p)) for p, ch in enumerate(name) if ch in "abcd")


class DeviceArray:
    """fp64 C-contiguous array in HBM."""

    def __init__(self, ctx, ptr, shape, owned=True, keepalive=None):
        self.ctx, self.ptr, self.shape = ctx, int(ptr), tuple(int(s) for s in shape)
        self._owned, self._keep = owned, keepalive
        self.size = int(np.prod(self.shape)) if self.shape else 1

    @property
    def nbytes(self):
        return 8 * self.size

    def get(self):
        out = np.empty(self.shape, dtype=np.float64)
        self.ctx.lib.call("pymes_download", self.ctx.handle, _lib.host_ptr(out), C.c_void_p(self.ptr), self.nbytes)
        return out

    def set(self, host):
        host = np.ascontiguousarray(host, dtype=np.float64)
        if host.shape != self.shape:
            raise ValueError(f"shape mismatch: {host.shape} vs {self.shape}")
        self.ctx.lib.call("pymes_upload", self.ctx.handle, C.c_void_p(self.ptr), _lib.host_ptr(host), self.nbytes)
        return self

    def copy_from(self, other):
        if other.size != self.size:
            raise ValueError("size mismatch")
        self.ctx.lib.call("pymes_copy", self.ctx.handle, C.c_void_p(self.ptr), C.c_void_p(other.ptr), self.nbytes)
        return self

    def zero_(self):
        self.ctx.lib.call("pymes_memset_zero", self.ctx.handle, C.c_void_p(self.ptr), self.nbytes)
        return self

    def reshape(self, *shape):
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        if int(np.prod(shape)) != self.size:
            raise ValueError("cannot reshape")
        return DeviceArray(self.ctx, self.ptr, shape, owned=False, keepalive=self)

    def free(self):
        # a closed context has released every buffer it handed out (pymes_ctx_destroy); a live one keeps the buffer for
        # the next array of that size (Context._recycle): all work of a context is ordered on one stream, so reuse needs
        # neither hipFree nor the stream synchronisation that goes with it
        if self._owned and self.ptr and self.ctx.handle:
            self.ctx._recycle(self.ptr, self.size)
        self.ptr, self._owned = 0, False

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    def __init__(self, no, nv, device=0, workspace_bytes=0, lib=None, stream=None, allocator=None):
        self.lib = lib or _lib.default_library()
        self.no, self.nv, self.n = int(no), int(nv), int(no) + int(nv)
        self.device = int(device)
        h = C.c_void_p()
        self.handle = None
        self.lib.call("pymes_ctx_create", C.byref(h), self.device, self.no, self.nv, int(workspace_bytes))
        self.handle = h
        self._allocator = allocator      # optional callable(n_doubles) -> (ptr, keepalive), e.g. torch-backed
        self._pool = {}
        self._spare, self._spare_bytes = {}, 0      # released buffers by size, for reuse (see _recycle)
        self._closing = []               # weak callbacks run by close() while the buffers are still alive (on_close)
        self.profiling = False           # per-GEMM event timing is on (launch graphs are bypassed then)
        if stream is not None:
            self.set_stream(stream)

    # ---- lifetime -----------------------------------------------------------------
    def close(self):
        """Destroys the context; every buffer it handed out (pool, arrays still referenced by the caller) is released
        with it, DeviceArrays of a closed context are dead."""
        if self.handle:
            for ref in self._closing:        # e.g. a DIIS mixer parking the vectors it keeps in this context
                cb = ref()
                if cb is not None:
                    try:
                        cb(self)
                    except Exception:
                        pass
        self._closing = []
        self._pool = {}
        self._spare, self._spare_bytes = {}, 0
        if self.handle:
            self.lib.call("pymes_ctx_destroy", self.handle)
            self.handle = None

    def on_close(self, method):
        """Register a bound method ``method(ctx)`` to be called (once, weakly referenced) right before the context is
        destroyed — holders of device vectors that must outlive it move them to the host there."""
        if not any(ref() == method for ref in self._closing):
            self._closing.append(weakref.WeakMethod(method))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        self.lib.call("pymes_ctx_set_stream", self.handle, C.c_void_p(int(hip_stream)))

    def sync(self):
        self.lib.call("pymes_ctx_sync", self.handle)

    def workspace(self):
        cap, high = C.c_uint64(), C.c_uint64()
        self.lib.call("pymes_ctx_workspace", self.handle, C.byref(cap), C.byref(high))
        return cap.value, high.value

    # ---- arrays -------------------------------------------------------------------
    RECYCLE_BYTES = 16 << 30       # at most this much released memory is kept for reuse; beyond it buffers are freed
    RECYCLE_FRACTION = 0.25        # ... and never more than this fraction of what the device has free right now

    def mem_info(self):
        """(free, total) bytes of the context's device."""
        free, total = C.c_uint64(), C.c_uint64()
        self.lib.call("pymes_mem_info", self.handle, C.byref(free), C.byref(total))
        return free.value, total.value

    def empty(self, shape):
        shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        n = int(np.prod(shape)) if shape else 1
        if self._allocator is not None:
            ptr, keep = self._allocator(n)
            return DeviceArray(self, ptr, shape, owned=False, keepalive=keep)
        spare = self._spare.get(max(n, 1))
        if spare:
            self._spare_bytes -= 8 * max(n, 1)
            return DeviceArray(self, spare.pop(), shape)
        p = C.c_void_p()
        try:
            self.lib.call("pymes_malloc", self.handle, 8 * max(n, 1), C.byref(p))
        except PymesError:
            if not self._spare_bytes:
                raise
            self.trim()                  # the memory may be sitting in the spare list under other sizes
            self.lib.call("pymes_malloc", self.handle, 8 * max(n, 1), C.byref(p))
        return DeviceArray(self, p.value, shape)

    def _recycle(self, ptr, n):
        """A released owned buffer: kept for the next ``empty`` of the same size (the drivers written in Python — EOM sigma,
        Davidson, UEG — create and drop dozens of temporaries per step; hipMalloc / hipFree + a stream synchronisation for
        each of them cost more than the kernels in between)."""
        n = max(int(n), 1)
        cap = self.RECYCLE_BYTES
        if 8 * n >= (64 << 20):          # large buffers: never sit on more than a fraction of what is still free
            self._free_probe = getattr(self, "_free_probe", 0) - 1        # (hipMemGetInfo every 32nd large release)
            if self._free_probe <= 0:
                self._free_seen, self._free_probe = self.mem_info()[0] + self._spare_bytes, 32
            cap = min(cap, int(self.RECYCLE_FRACTION * self._free_seen))
        if self._spare_bytes + 8 * n > cap:
            self.lib.call("pymes_free", self.handle, C.c_void_p(ptr))
            return
        self._spare.setdefault(n, []).append(ptr)
        self._spare_bytes += 8 * n

    def trim(self):
        """Return the recycled buffers to the device allocator (the engine's own allocations — integral blocks, packed
        integrals, pair layouts — cannot see the spare list: the solvers call this before their set-up)."""
        for ptrs in self._spare.values():
            for ptr in ptrs:
                self.lib.call("pymes_free", self.handle, C.c_void_p(ptr))
        self._spare, self._spare_bytes = {}, 0
        self.lib.call("pymes_scratch_trim", self.handle)       # ... and the engine's pooled scratch (EOM sigma temporaries)

    def zeros(self, shape):
        return self.empty(shape).zero_()

    # amplitude-sized buffers are recycled instead of hipFree'd/hipMalloc'ed every iteration
    def pool_get(self, shape):
        shape = tuple(int(s) for s in shape)
        free = self._pool.setdefault(int(np.prod(shape)), [])
        if free:
            arr = free.pop()
            return DeviceArray(self, arr.ptr, shape, owned=False, keepalive=arr)
        return self.empty(shape)

    def pool_reserve(self, shape, count):
        """Make sure ``count`` buffers of this size exist (in use or free): device allocations (page-table set-up and
        clearing of fresh HBM pages) then happen at set-up time, not inside the first iterations of a solve."""
        n = int(np.prod(tuple(int(s) for s in shape)))
        free = self._pool.setdefault(n, [])
        while len(free) < count:
            free.append(self.empty((n,)))

    def pool_put(self, arr):
        base = arr
        while not base._owned and isinstance(base._keep, DeviceArray):
            base = base._keep
        self._pool.setdefault(base.size, []).append(base)

    def array(self, host):
        host = np.ascontiguousarray(host, dtype=np.float64)
        return self.empty(host.shape).set(host)

    # ---- tensor engine ------------------------------------------------------------
    def contract(self, spec, A, B, out=None, alpha=1.0, beta=0.0, batch=""):
        """Binary einsum on the device, e.g. ``contract("abcd,cdij->abij", V, T)``."""
        ins, lc = spec.replace(" ", "").split("->")
        la, lb = ins.split(",")
        dims = {}
        for lab, arr in ((la, A), (lb, B)):
            if len(lab) != len(arr.shape):
                raise ValueError(f"labels '{lab}' do not match rank {len(arr.shape)}")
            for ch, d in zip(lab, arr.shape):
                if dims.setdefault(ch, d) != d:
                    raise ValueError(f"extent mismatch for label '{ch}'")
        shape_c = tuple(dims[ch] for ch in lc)
        if out is None:
            out = self.empty(shape_c)
            beta = 0.0
        elif out.shape != shape_c:
            raise ValueError("output shape mismatch")
        self.lib.call("pymes_contract", self.handle, float(alpha),
                      C.c_void_p(A.ptr), la.encode(), i64_array(A.shape), None,
                      C.c_void_p(B.ptr), lb.encode(), i64_array(B.shape), None,
                      float(beta), C.c_void_p(out.ptr), lc.encode(), i64_array(out.shape), None, batch.encode())
        return out

    def permute(self, spec, A, out=None, alpha=1.0, beta=0.0, in_view=None):
        """``permute("abij->aibj", T)``: out[lo] = alpha*A[li] + beta*out[lo].  ``in_view`` = (dims, strides in doubles): read
        A through a strided view instead of its own shape (``li`` then labels the view), e.g. the diagonal V[a,b,a,b] of a
        [v,v,v,v] block as dims (v, v), strides (v^3 + v, v^2 + 1)."""
        li, lo = spec.replace(" ", "").split("->")
        dims = tuple(A.shape) if in_view is None else tuple(int(d) for d in in_view[0])
        shape_o = tuple(dims[li.index(ch)] for ch in lo)
        if out is None:
            out = self.empty(shape_o)
            beta = 0.0
        self.lib.call("pymes_permute", self.handle, float(alpha), C.c_void_p(A.ptr), li.encode(), i64_array(dims),
                      None if in_view is None else i64_array(in_view[1]), float(beta), C.c_void_p(out.ptr), lo.encode(), None)
        return out

    def dgemm(self, M, N, K, alpha, A, a_sm, a_sk, B, b_sk, b_sn, beta, Cmat, ldc):
        self.lib.call("pymes_dgemm", self.handle, M, N, K, float(alpha), C.c_void_p(A.ptr), a_sm, a_sk,
                      C.c_void_p(B.ptr), b_sk, b_sn, float(beta), C.c_void_p(Cmat.ptr), ldc)

    # ---- integrals ----------------------------------------------------------------
    def set_V_pqrs(self, V):
        """partition.py:4-39 — V is a host numpy [n,n,n,n] or a DeviceArray of that shape."""
        if isinstance(V, DeviceArray):
            self.lib.call("pymes_set_V_pqrs", self.handle, C.c_void_p(V.ptr), 1, None)
        else:
            V = np.ascontiguousarray(V, dtype=np.float64)
            if V.shape != (self.n,) * 4:
                raise ValueError(f"V_pqrs must have shape {(self.n,) * 4}")
            self.lib.call("pymes_set_V_pqrs", self.handle, _lib.host_ptr(V), 0, None)

    def set_V_block(self, name, data):
        want = self.block_shape(name)
        if tuple(data.shape) != want:      # the reference would raise a numpy/einsum shape error at first use
            raise ValueError(f"integral block '{name}' must have shape {want} for (no, nv) = ({self.no}, {self.nv}), "
                             f"got {tuple(data.shape)}")
        if isinstance(data, DeviceArray):
            self.lib.call("pymes_set_V_block", self.handle, name.encode(), C.c_void_p(data.ptr), data.size, 1, None)
        else:
            data = np.ascontiguousarray(data, dtype=np.float64)
            self.lib.call("pymes_set_V_block", self.handle, name.encode(), _lib.host_ptr(data), data.size, 0, None)

    def V_exchange_asymmetry(self):
        """(max |V_pqrs - V_qpsr| over the blocks that are set — inf if a partner block is missing —, max |V|)."""
        out = (C.c_double * 2)()
        self.lib.call("pymes_V_exchange_asymmetry", self.handle, out)
        return out[0], out[1]

    def V_exchange_symmetric(self, rtol=1e-12):
        """V_pqrs = V_qpsr to rounding: the precondition of the symmetry-reduced residual (pair-packed ladders, merged
        ring products, amplitude-side dressing).  The reference makes no such assumption, so inputs that violate it
        (a user-built TC Hamiltonian, an FCIDUMP that lists only one of (ij|kl) / (kl|ij)) take the general path."""
        asym, vmax = self.V_exchange_asymmetry()
        return bool(np.isfinite(vmax) and asym <= rtol * max(1.0, vmax))

    def exchange_symmetric(self, x, rtol=1e-13):
        """X[p,q,r,s] = X[q,p,s,r] (amplitudes: T_abij = T_baji) for a DeviceArray with shape (d0,d0,d2,d2)."""
        if len(x.shape) != 4 or x.shape[0] != x.shape[1] or x.shape[2] != x.shape[3]:
            raise ValueError("exchange_symmetric: need a [p,p,r,r] array")
        out = (C.c_double * 2)()
        self.lib.call("pymes_exchange_asymmetry", self.handle, C.c_void_p(x.ptr), C.c_void_p(x.ptr), i64_array(x.shape), out)
        return bool(np.isfinite(out[1]) and out[0] <= rtol * max(1.0, out[1]))     # inf / NaN entries: never "symmetric"

    def set_V_from_factors(self, B):
        B = np.ascontiguousarray(B, dtype=np.float64)
        if B.ndim != 3 or B.shape[1:] != (self.n, self.n):
            raise ValueError("B must be [naux, n, n]")
        self.lib.call("pymes_set_V_from_factors", self.handle, _lib.host_ptr(B), int(B.shape[0]))

    def block_shape(self, name):
        return tuple(self.nv if ch in "abcd" else self.no for ch in name)

    def V_block(self, name, dressed=False):
        p, nel = C.c_void_p(), C.c_int64()
        self.lib.call("pymes_V_block_ptr", self.handle, name.encode(), int(dressed), C.byref(p), C.byref(nel))
        return DeviceArray(self, p.value, self.block_shape(name), owned=False, keepalive=self)

    def set_orbital_energies(self, eps_o, eps_v):
        eo = np.ascontiguousarray(eps_o, dtype=np.float64)
        ev = np.ascontiguousarray(eps_v, dtype=np.float64)
        assert eo.shape == (self.no,) and ev.shape == (self.nv,)
        self.lib.call("pymes_set_orbital_energies", self.handle, _lib.host_ptr(eo), _lib.host_ptr(ev))

    def phase_enable(self, mode):
        """Phase launches of the calling thread (include/pymes_amd.h): 1 on, 0 off, -1 as PYMES_PHASE says."""
        self.lib.call("pymes_phase_enable", int(mode))

    def phase_hold(self):
        """``with ctx.phase_hold():`` — the small operations of the calls inside are recorded together and launched level by
        level when the block ends (include/pymes_amd.h, pymes_phase_hold)."""
        import contextlib

        @contextlib.contextmanager
        def hold():
            self.lib.call("pymes_phase_hold", 1)
            try:
                yield
            finally:
                self.lib.call("pymes_phase_hold", 0)
        return hold()

    def phase_stats(self):
        """(tasks recorded, grids launched, levels, flushes) by the calling thread so far."""
        v = [C.c_int64() for _ in range(4)]
        self.lib.call("pymes_phase_stats", *[C.byref(x) for x in v])
        return dict(zip(("tasks", "launches", "levels", "flushes"), (int(x.value) for x in v)))

    @property
    def dress_generation(self):
        """How often the context's dressed blocks have been (re)written — counted by the engine itself, so that the
        dressings inside ``pymes_ccsd_residuals`` / ``_iterate`` / ``_sharded_residuals`` and replayed launch graphs are
        seen too; ``DressedDeviceIntegrals.require`` compares."""
        n = C.c_uint64()
        self.lib.call("pymes_dress_generation", self.handle, C.byref(n))
        return int(n.value)

    # ---- CC hot path ----------------------------------------------------------------
    def mp2(self, t2, level_shift=0.0):
        e = (C.c_double * 2)()
        self.lib.call("pymes_mp2", self.handle, float(level_shift), C.c_void_p(t2.ptr), e)
        return e[0], e[1]

    def dress_fock(self, f, t1, out):
        self.lib.call("pymes_ccsd_dress_fock", self.handle, C.c_void_p(f.ptr), C.c_void_p(t1.ptr), C.c_void_p(out.ptr))
        return out

    def dress_V(self, t1, names, reduced_abij=False, q_range=None, p_range=None):
        """ccsd.py:290-421 for the named blocks; ``reduced_abij``: the form of V~_abij that goes with
        ``residual_slab(..., t1=, QK=)``; ``p_range`` / ``q_range``: only these ranges of the first / second (virtual) index of the blocks
        (include/pymes_amd.h)."""
        mask = _lib.PYMES_DRESS_ABIJ_REDUCED if reduced_abij else 0
        for nm in names:
            mask |= 1 << pattern_of(nm)
        if q_range is not None or p_range is not None:
            p0, p1 = p_range if p_range is not None else (0, 0)
            q0, q1 = q_range if q_range is not None else (0, 0)
            self.lib.call("pymes_ccsd_dress_V_slab", self.handle, C.c_void_p(t1.ptr), mask, int(p0), int(p1), int(q0), int(q1))
        else:
            self.lib.call("pymes_ccsd_dress_V", self.handle, C.c_void_p(t1.ptr), mask)

    def ccsd_residuals(self, f, t1, t2, r1, r2, is_dcd=False, t1_zero=False):
        """ccsd.py:161-171 in one call (symmetry-reduced form, one rank; include/pymes_amd.h): R1, R2 from (f, t1, t2)."""
        self.lib.call("pymes_ccsd_residuals", self.handle, C.c_void_p(f.ptr), C.c_void_p(t1.ptr), C.c_void_p(t2.ptr),
                      (_lib.PYMES_DCD if is_dcd else 0) | (_lib.PYMES_T1_ZERO if t1_zero else 0), C.c_void_p(r1.ptr),
                      C.c_void_p(r2.ptr))

    def ccsd_iterate(self, f, t1, t2, dt1, dt2, level_shift=0.0, delta=1.0, is_dcd=False, t1_zero=False):
        """One fixed-point pass without a mixer (ccsd.py:159-197, is_diis = False): (e_1b, e_dir, e_ex, |t2|^2, |dt2|^2, |t1|^2)."""
        out = (C.c_double * 6)()
        self.lib.call("pymes_ccsd_iterate", self.handle, C.c_void_p(f.ptr), C.c_void_p(t1.ptr), C.c_void_p(t2.ptr),
                      (_lib.PYMES_DCD if is_dcd else 0) | (_lib.PYMES_T1_ZERO if t1_zero else 0), float(level_shift), float(delta),
                      C.c_void_p(dt1.ptr), C.c_void_p(dt2.ptr), out)
        return tuple(out)

    def ccsd_release(self):
        """Give back the staging buffers pymes_ccsd_residuals holds (recorded launch graphs that replay it must be gone)."""
        self.lib.call("pymes_ccsd_release", self.handle)

    def singles_residual(self, fd, t1, t2, out):
        self.lib.call("pymes_ccsd_singles_residual", self.handle, C.c_void_p(fd.ptr), C.c_void_p(t1.ptr),
                      C.c_void_p(t2.ptr), C.c_void_p(out.ptr))
        return out

    def singles_residual_partial(self, fd, t1, t2, out, rank, world, reuse_layouts=False):
        """This rank's K-sharded share of the singles residual (exchange-symmetric T2; the caller all-reduces).
        ``reuse_layouts``: t2 is unchanged since the preceding ``residual_slab`` call on it (its pair layouts are read)."""
        self.lib.call("pymes_ccsd_singles_residual_partial", self.handle, C.c_void_p(fd.ptr), C.c_void_p(t1.ptr),
                      C.c_void_p(t2.ptr), C.c_void_p(out.ptr), int(rank), int(world),
                      _lib.PYMES_REUSE_LAYOUTS if reuse_layouts else 0)
        return out

    def doubles_residual(self, f, t2, out, is_dcd=False, dressed=False, skip_ladder=False, sym_ladder=False,
                         sym_rings=None):
        """``sym_ladder`` / ``sym_rings``: T_abij = T_baji and V_pqrs = V_qpsr hold, use the symmetry-reduced
        forms (pair-packed ladder, merged ring products); ``sym_rings`` defaults to ``sym_ladder``."""
        sym_rings = sym_ladder if sym_rings is None else sym_rings
        flags = (_lib.PYMES_DCD if is_dcd else 0) | (_lib.PYMES_USE_DRESSED if dressed else 0) | \
                (_lib.PYMES_SKIP_LADDER if skip_ladder else 0) | (_lib.PYMES_SYM_LADDER if sym_ladder else 0) | \
                (_lib.PYMES_SYM_RINGS if sym_rings else 0)
        self.lib.call("pymes_doubles_residual", self.handle, C.c_void_p(f.ptr), C.c_void_p(t2.ptr),
                      C.c_void_p(out.ptr), flags)
        return out

    def ladder(self, t2, out, a_begin, a_end, dressed=False, beta=0.0):
        self.lib.call("pymes_ladder", self.handle, C.c_void_p(t2.ptr), C.c_void_p(out.ptr), int(a_begin), int(a_end),
                      int(dressed), float(beta))
        return out

    @staticmethod
    def _flags(is_dcd, dressed, skip_ladder=False, sym_ladder=False, sym_rings=False):
        return (_lib.PYMES_DCD if is_dcd else 0) | (_lib.PYMES_USE_DRESSED if dressed else 0) | \
               (_lib.PYMES_SKIP_LADDER if skip_ladder else 0) | (_lib.PYMES_SYM_LADDER if sym_ladder else 0) | \
               (_lib.PYMES_SYM_RINGS if sym_rings else 0)

    def residual_slab(self, f, t2, ETd, ETx, L, rank, world, is_dcd=False, dressed=False, t1=None, QK=None, P=None,
                      part=None):
        """This rank's share of the symmetry-reduced residual (include/pymes_amd.h).  ``t1`` + ``QK``: T1 dressing
        of V_abcd on the amplitude side; ``P``: all-reduced output of ``slab_prepare``; ``part``: "rings" / "ladders" to
        compute only that half (the caller overlaps the exchange of the first with the second)."""
        extra = {None: 0, "rings": _lib.PYMES_SLAB_RINGS_ONLY, "ladders": _lib.PYMES_SLAB_LADDERS_ONLY}[part]
        self.lib.call("pymes_residual_slab", self.handle, C.c_void_p(f.ptr), C.c_void_p(t2.ptr), C.c_void_p(ETd.ptr),
                      C.c_void_p(ETx.ptr), C.c_void_p(L.ptr if L is not None else 0), int(rank), int(world),
                      self._flags(is_dcd, dressed, False, True, True) | extra, C.c_void_p(t1.ptr if t1 is not None else 0),
                      C.c_void_p(QK.ptr if QK is not None else 0), C.c_void_p(P.ptr if P is not None else 0))

    def slab_prepare_ws(self):
        n = C.c_int64()
        self.lib.call("pymes_slab_prepare_ws", self.handle, C.byref(n))
        return n.value

    def slab_prepare(self, t2, P, rank, world, is_dcd=False):
        self.lib.call("pymes_slab_prepare", self.handle, C.c_void_p(t2.ptr), C.c_void_p(P.ptr), int(rank), int(world),
                      _lib.PYMES_DCD if is_dcd else 0)
        return P

    def residual_finish(self, f, t2, ETd, ETx, L, out, is_dcd=False, dressed=False, t1=None, QK=None, reuse_layouts=False):
        self.lib.call("pymes_residual_finish", self.handle, C.c_void_p(f.ptr), C.c_void_p(t2.ptr), C.c_void_p(ETd.ptr),
                      C.c_void_p(ETx.ptr), C.c_void_p(L.ptr if L is not None else 0), C.c_void_p(out.ptr),
                      self._flags(is_dcd, dressed, False, True, True) | (_lib.PYMES_REUSE_LAYOUTS if reuse_layouts else 0),
                      C.c_void_p(t1.ptr if t1 is not None else 0),
                      C.c_void_p(QK.ptr if QK is not None else 0))
        return out

    # ---- pair-sharded tail (include/pymes_amd.h) ----------------------------------------------------------------
    def residual_finish_pairs(self, f, t2, ETd, ETx, L, Rc, rank, world, t1=None, QK=None, is_dcd=False, dressed=False,
                              Xvv=None):
        self.lib.call("pymes_residual_finish_pairs", self.handle, C.c_void_p(f.ptr), C.c_void_p(t2.ptr),
                      C.c_void_p(ETd.ptr), C.c_void_p(ETx.ptr), C.c_void_p(L.ptr), C.c_void_p(Rc.ptr),
                      self._flags(is_dcd, dressed, False, True, True), C.c_void_p(t1.ptr if t1 is not None else 0),
                      C.c_void_p(QK.ptr if QK is not None else 0), int(rank), int(world),
                      C.c_void_p(Xvv.ptr if Xvv is not None else 0))
        return Rc

    def xvv_partial(self, f, t2, Xvv, rank, world, is_dcd=False):
        self.lib.call("pymes_xvv_partial", self.handle, C.c_void_p(f.ptr), C.c_void_p(t2.ptr), C.c_void_p(Xvv.ptr),
                      int(rank), int(world), _lib.PYMES_DCD if is_dcd else 0)
        return Xvv

    def dress_fock_ws(self):
        n = C.c_int64()
        self.lib.call("pymes_ccsd_dress_fock_ws", self.handle, C.byref(n))
        return n.value

    def dress_fock_partial(self, t1, W, rank, world):
        self.lib.call("pymes_ccsd_dress_fock_partial", self.handle, C.c_void_p(t1.ptr), C.c_void_p(W.ptr), int(rank), int(world))
        return W

    def dress_fock_finish(self, f, t1, W, fd):
        self.lib.call("pymes_ccsd_dress_fock_finish", self.handle, C.c_void_p(f.ptr), C.c_void_p(t1.ptr), C.c_void_p(W.ptr),
                      C.c_void_p(fd.ptr))
        return fd

    def cc_update_pairs(self, tc, dtc, rc, shift, delta, rank, world):
        self.lib.call("pymes_cc_update_pairs", self.handle, C.c_void_p(tc.ptr), C.c_void_p(dtc.ptr), C.c_void_p(rc.ptr),
                      float(shift), float(delta), int(rank), int(world))

    def pairs_supported(self):
        yes = C.c_int()
        self.lib.call("pymes_pairs_supported", self.handle, C.byref(yes))
        return bool(yes.value)

    def pairs_pack(self, full, xc, rank, world):
        self.lib.call("pymes_pairs_pack", self.handle, C.c_void_p(full.ptr), C.c_void_p(xc.ptr), int(rank), int(world))
        return xc

    def pairs_unpack(self, xc_all, full, world):
        self.lib.call("pymes_pairs_unpack", self.handle, C.c_void_p(xc_all.ptr), C.c_void_p(full.ptr), int(world))
        return full

    def dress_abcd_rows(self, t1, a_begin, a_end, lower_only=False):
        self.lib.call("pymes_ccsd_dress_abcd_rows", self.handle, C.c_void_p(t1.ptr), int(a_begin), int(a_end),
                      int(lower_only))

    def ladder_sym(self, t2, L, row_begin, row_end, dressed=False, hole_ladder=0):
        """Rows [row_begin,row_end) of the pair-packed ladder L[v(v+1)/2, o*o] (include/pymes_amd.h);
        hole_ladder = 1 (CCSD) / 2 (DCSD) adds ccd.py:175-186 to the same rows."""
        self.lib.call("pymes_ladder_sym", self.handle, C.c_void_p(t2.ptr), C.c_void_p(L.ptr), int(row_begin),
                      int(row_end), int(dressed), int(hole_ladder))
        return L

    def ladder_sym_multi(self, xs, L_all, dressed=False):
        """Particle ladder (ccd.py:187 / eom_ccsd.py:383, pair-packed) of the k exchange-symmetric arrays ``xs`` in one
        batched launch per half; ``L_all`` is [k, v(v+1)/2, o*o] (include/pymes_amd.h)."""
        self.lib.call("pymes_ladder_sym_multi", self.handle, ptr_array([x.ptr for x in xs]), len(xs), C.c_void_p(L_all.ptr),
                      int(dressed))
        return L_all

    def ladder_dress(self, V, Pk, t1, W, ld, r0, r1, minus_half=False):
        """T1 dressing of the bra of pair-packed rows [r0,r1) of V_abcd (ccsd.py:414-419; include/pymes_amd.h)."""
        self.lib.call("pymes_ladder_dress", self.handle, C.c_void_p(V.ptr), C.c_void_p(Pk.ptr), C.c_void_p(t1.ptr),
                      C.c_void_p(W.ptr), int(ld), int(r0), int(r1), int(minus_half))
        return W

    def pair_layouts(self, x, Xx, Xt, Xd=None):
        """Xx[(a,j),(b,i)] = x_abij, Xt[(a,i),(b,j)] = 2 x_abij - x_baij (and Xd[(a,i),(b,j)] = x_abij) in one pass."""
        self.lib.call("pymes_pair_layouts", self.handle, C.c_void_p(x.ptr), C.c_void_p(Xd.ptr if Xd is not None else 0),
                      C.c_void_p(Xx.ptr), C.c_void_p(Xt.ptr))

    def symmetrised_assemble(self, N, D, X, out, V=None, L=None):
        """out_abij = V + unpack(L) + N_abij + N_baji + D[(a,i),(b,j)] + D[(b,j),(a,i)] + X[(a,j),(b,i)] + X[(b,i),(a,j)] in one
        pass (include/pymes_amd.h); V / L optional."""
        self.lib.call("pymes_symmetrised_assemble", self.handle, C.c_void_p(V.ptr if V is not None else 0),
                      C.c_void_p(L.ptr if L is not None else 0), C.c_void_p(N.ptr), C.c_void_p(D.ptr), C.c_void_p(X.ptr),
                      C.c_void_p(out.ptr))
        return out

    def hole_ladder_packed(self, x, I, L, row_begin, row_end, y=None):
        """Rows of the pair-packed L += sum_kl (I_klij + sum_cd V_klcd y_cdij) X_abkl (I_klij = I_lkji, X_abkl = X_balk,
        y optional and exchange-symmetric); include/pymes_amd.h."""
        self.lib.call("pymes_hole_ladder_packed", self.handle, C.c_void_p(x.ptr), C.c_void_p(I.ptr), C.c_void_p(L.ptr),
                      int(row_begin), int(row_end), C.c_void_p(y.ptr if y is not None else 0))
        return L

    def hole_ladder_packed_multi(self, xs, Is, L_all, ys=None):
        """``hole_ladder_packed`` over all rows for the k vectors at once (batched launches; entries of ``xs`` / ``Is`` that
        are all the same array are packed once): L_all[z] += rows(xs[z]) . (2 Is[z] [+ 2 V_klcd ys[z]_cdij])."""
        k = len(xs)
        self.lib.call("pymes_hole_ladder_packed_multi", self.handle, ptr_array([x.ptr for x in xs]),
                      ptr_array([i.ptr for i in Is]), ptr_array([y.ptr for y in ys]) if ys is not None else None, k,
                      C.c_void_p(L_all.ptr))
        return L_all

    def ladder_sym_unpack(self, L, out, beta=1.0):
        self.lib.call("pymes_ladder_sym_unpack", self.handle, C.c_void_p(L.ptr), C.c_void_p(out.ptr), float(beta))
        return out

    def cc_update(self, t, dt, r, level_shift=0.0, delta=1.0):
        rank = len(t.shape)
        self.lib.call("pymes_cc_update", self.handle, C.c_void_p(t.ptr), C.c_void_p(dt.ptr), C.c_void_p(r.ptr),
                      float(level_shift), float(delta), rank)

    def cc_update_to(self, t_out, dt, t_in, r, level_shift=0.0, delta=1.0):
        """dt = r/(D+shift), t_out = t_in + delta*dt (t_out may be t_in)."""
        self.lib.call("pymes_cc_update_to", self.handle, C.c_void_p(t_out.ptr), C.c_void_p(dt.ptr), C.c_void_p(t_in.ptr),
                      C.c_void_p(r.ptr), float(level_shift), float(delta), len(t_in.shape))

    def energy_norms(self, f, t1, t2, dt2=None):
        """(one-body, direct, exchange, |t2|^2, |dt2|^2, |t1|^2) in one pass and one synchronisation; f/t1 None for CCD."""
        out = (C.c_double * 6)()
        self.lib.call("pymes_energy_norms", self.handle, C.c_void_p(f.ptr if f is not None else 0),
                      C.c_void_p(t1.ptr if t1 is not None else 0), C.c_void_p(t2.ptr),
                      C.c_void_p(dt2.ptr if dt2 is not None else 0), out)
        return tuple(out[:])

    def energy_norms_start(self, f, t1, t2, dt2=None):
        """``energy_norms`` enqueued only: returns a read-back slot for ``energy_norms_wait`` (include/pymes_amd.h)."""
        slot = C.c_int()
        self.lib.call("pymes_energy_norms_start", self.handle, C.c_void_p(f.ptr if f is not None else 0),
                      C.c_void_p(t1.ptr if t1 is not None else 0), C.c_void_p(t2.ptr),
                      C.c_void_p(dt2.ptr if dt2 is not None else 0), C.byref(slot))
        return slot.value

    def energy_norms_wait(self, slot):
        out = (C.c_double * 6)()
        self.lib.call("pymes_energy_norms_wait", self.handle, int(slot), out)
        return tuple(out[:])

    def readback_start(self, arr, n=None, offset=0):
        """Start the read-back of n (<= 128) doubles of ``arr`` from ``offset`` on; ``readback_wait(slot, n)`` returns them
        as a numpy vector once that copy — not the whole stream — has completed."""
        n = arr.size - offset if n is None else int(n)
        slot = C.c_int()
        self.lib.call("pymes_readback_start", self.handle, C.c_void_p(arr.ptr + 8 * int(offset)), n, C.byref(slot))
        return slot.value

    def readback_wait(self, slot, n):
        buf = (C.c_double * int(n))()
        self.lib.call("pymes_readback_wait", self.handle, int(slot), buf, int(n))
        return np.frombuffer(buf, dtype=np.float64).copy()

    def energy_norms_pairs(self, f, t1, tc, dtc, rank, world):
        """This rank's partial sums of ``energy_norms`` from the compact tiles of its pairs (to be all-reduced)."""
        out = (C.c_double * 6)()
        self.lib.call("pymes_energy_norms_pairs", self.handle, C.c_void_p(f.ptr if f is not None else 0),
                      C.c_void_p(t1.ptr if t1 is not None else 0), C.c_void_p(tc.ptr),
                      C.c_void_p(dtc.ptr if dtc is not None else 0), int(rank), int(world), out)
        return np.array(out[:])

    # ---- launch graphs ----------------------------------------------------------------
    def graph_begin(self):
        self.lib.call("pymes_graph_begin", self.handle)

    def graph_end(self):
        g = C.c_void_p()
        self.lib.call("pymes_graph_end", self.handle, C.byref(g))
        return g

    def graph_abort(self):
        self.lib.call("pymes_graph_abort", self.handle)

    def graph_launch(self, g):
        self.lib.call("pymes_graph_launch", self.handle, g)

    def graph_destroy(self, g):
        self.lib.call("pymes_graph_destroy", self.handle, g)

    def graphs_supported(self):
        return self.lib.backend.startswith("hip")

    def ccsd_energy(self, f, t1, t2):
        e = (C.c_double * 3)()
        self.lib.call("pymes_ccsd_energy", self.handle, C.c_void_p(f.ptr), C.c_void_p(t1.ptr), C.c_void_p(t2.ptr), e)
        return e[0], e[1], e[2]

    def ccd_energy(self, t2):
        e = (C.c_double * 2)()
        self.lib.call("pymes_ccd_energy", self.handle, C.c_void_p(t2.ptr), e)
        return e[0], e[1]

    # ---- vector helpers ---------------------------------------------------------------
    def dots(self, xs, ys):
        """out[p] = <xs[p], ys[p]>; the pairs may differ in length (T1 and T2 overlaps in one launch)."""
        assert len(xs) == len(ys) and all(x.size == y.size for x, y in zip(xs, ys))
        out = np.empty(len(xs))
        for lo in range(0, len(xs), 16):
            hi = min(len(xs), lo + 16)
            buf = (C.c_double * (hi - lo))()
            self.lib.call("pymes_dots_var", self.handle, hi - lo, ptr_array([x.ptr for x in xs[lo:hi]]),
                          ptr_array([y.ptr for y in ys[lo:hi]]), i64_array([x.size for x in xs[lo:hi]]), buf)
            out[lo:hi] = buf[:]
        return out

    def norm(self, x):
        return float(np.sqrt(self.dots([x], [x])[0]))

    def lincomb(self, out, xs, coeffs):
        assert len(xs) == len(coeffs) and all(x.size == out.size for x in xs)
        first = True
        for lo in range(0, len(xs), 7):
            chunk, cc = list(xs[lo:lo + 7]), [float(c) for c in coeffs[lo:lo + 7]]
            if not first:
                chunk, cc = [out] + chunk, [1.0] + cc
            cbuf = (C.c_double * len(cc))(*cc)
            self.lib.call("pymes_lincomb", self.handle, C.c_void_p(out.ptr), len(chunk),
                          ptr_array([x.ptr for x in chunk]), cbuf, out.size)
            first = False
        return out

    def gemm_group(self):
        """``with ctx.gemm_group() as g: ...``: the contract / dgemm calls inside are independent of each other (none reads or
        accumulates into the output of another) and the small ones share launches (include/pymes_amd.h,
        pymes_gemm_group_begin).  After the block ``g.launches`` / ``g.products`` say what was grouped."""
        ctx = self

        class _Group:
            launches = products = 0

            def __enter__(self):
                ctx.lib.call("pymes_gemm_group_begin", ctx.handle)
                return self

            def __exit__(self, *exc):
                l, p = (C.c_int64 * 1)(), (C.c_int64 * 1)()
                ctx.lib.call("pymes_gemm_group_end", ctx.handle, l, p)
                self.launches, self.products = l[0], p[0]
                return False
        return _Group()

    def gram(self, xs, ys):
        """G[i, j] = <xs[i], ys[j]> as a host array; every vector is read once per call (include/pymes_amd.h, pymes_gram)."""
        m, n = len(xs), len(ys)
        out = np.zeros((m, n))
        if m == 0 or n == 0:
            return out
        size = xs[0].size
        assert all(v.size == size for v in list(xs) + list(ys))
        for i0 in range(0, m, 64):
            for j0 in range(0, n, 64):
                xi, yj = xs[i0:i0 + 64], ys[j0:j0 + 64]
                buf = (C.c_double * (len(xi) * len(yj)))()
                self.lib.call("pymes_gram", self.handle, len(xi), len(yj), ptr_array([x.ptr for x in xi]),
                              ptr_array([y.ptr for y in yj]), size, buf)
                out[i0:i0 + len(xi), j0:j0 + len(yj)] = np.frombuffer(buf, dtype=np.float64).reshape(len(xi), len(yj))
        return out

    def lincomb_multi(self, outs, xs, coeff, beta=None):
        """outs[j] = sum_i coeff[i, j] xs[i] (+ beta[j] outs[j]); the inputs are read once for all outputs.  An output may be
        one of the inputs only for len(xs) <= 16 and len(outs) <= 4 (include/pymes_amd.h, pymes_lincomb_multi)."""
        m, n = len(xs), len(outs)
        if n == 0:
            return outs
        coeff = np.ascontiguousarray(np.asarray(coeff, dtype=np.float64).reshape(m, n))
        size = outs[0].size
        assert all(v.size == size for v in list(xs) + list(outs))
        bet = None if beta is None else np.ascontiguousarray(beta, dtype=np.float64)
        for j0 in range(0, n, 64):
            oj = outs[j0:j0 + 64]
            for i0 in range(0, max(m, 1), 64):
                xi = xs[i0:i0 + 64]
                cc = np.ascontiguousarray(coeff[i0:i0 + 64, j0:j0 + 64])
                bb = np.ones(len(oj)) if i0 > 0 else (bet[j0:j0 + 64].copy() if bet is not None else None)
                self.lib.call("pymes_lincomb_multi", self.handle, len(xi), len(oj),
                              ptr_array([x.ptr for x in xi]) if xi else None, _lib.host_ptr(cc) if xi else None,
                              _lib.host_ptr(bb) if bb is not None else None, ptr_array([y.ptr for y in oj]), size)
        return outs

    def diis_mix(self, state_host, err_hist, err_new, amp_hist, outs, m, was_full):
        """One DIIS step in one library call (include/pymes_amd.h, pymes_diis_mix): ``err_hist`` / ``amp_hist`` are the
        stored vectors type-major ([t][i]), ``state_host`` a float64 numpy array of 96 (L and the coefficients)."""
        ntypes = len(err_new)
        assert len(err_hist) == len(amp_hist) == ntypes * m and len(outs) == ntypes and state_host.size >= 96
        self.lib.call("pymes_diis_mix", self.handle, _lib.host_ptr(state_host), ntypes, int(m), int(bool(was_full)),
                      ptr_array([x.ptr for x in err_hist]), ptr_array([x.ptr for x in err_new]),
                      i64_array([x.size for x in err_new]), ptr_array([x.ptr for x in amp_hist]),
                      ptr_array([x.ptr for x in outs]))

    def diis_step(self, state, xs, ys, ntypes, m, was_full):
        """One DIIS step on the device (include/pymes_amd.h): overlaps <xs[p], ys[p]>, p = t * m + i, into ``state``."""
        assert len(xs) == len(ys) == ntypes * m and state.size >= 96
        self.lib.call("pymes_diis_step", self.handle, C.c_void_p(state.ptr), len(xs), ptr_array([x.ptr for x in xs]),
                      ptr_array([y.ptr for y in ys]), i64_array([x.size for x in xs]), int(ntypes), int(m), int(bool(was_full)))

    def lincomb_dev(self, out, xs, coeff_ptr):
        """out = sum_k c[k] xs[k] with c read from device memory at ``coeff_ptr`` (at most 8 terms)."""
        assert len(xs) <= 8 and all(x.size == out.size for x in xs)
        self.lib.call("pymes_lincomb_dev", self.handle, C.c_void_p(out.ptr), len(xs), ptr_array([x.ptr for x in xs]),
                      C.c_void_p(int(coeff_ptr)), out.size)
        return out

    def cmul(self, mr, mi, xr, xi, yr, yi):
        """(yr + i yi) = (mr + i mi) * (xr + i xi) element by element (y may alias x)."""
        assert mr.size == mi.size == xr.size == xi.size == yr.size == yi.size
        self.lib.call("pymes_cmul", self.handle, C.c_void_p(mr.ptr), C.c_void_p(mi.ptr), C.c_void_p(xr.ptr), C.c_void_p(xi.ptr),
                      C.c_void_p(yr.ptr), C.c_void_p(yi.ptr), xr.size)

    def cshift_inv(self, d, z, hs, shift, mr, mi):
        """(mr + i mi) = 1 / (z - hs d + shift) element by element: the FEAST preconditioner from a device-resident diagonal."""
        z, hs = complex(z), complex(hs)
        self.lib.call("pymes_cshift_inv", self.handle, C.c_void_p(d.ptr), z.real, z.imag, hs.real, hs.imag, float(shift),
                      C.c_void_p(mr.ptr), C.c_void_p(mi.ptr), d.size)

    # ---- measurement --------------------------------------------------------------------
    def stats(self, reset=False):
        gc, pc, gf, pb = C.c_int64(), C.c_int64(), C.c_double(), C.c_double()
        self.lib.call("pymes_stats", self.handle, int(reset), C.byref(gc), C.byref(gf), C.byref(pc), C.byref(pb))
        return {"gemm_calls": gc.value, "gemm_flops": gf.value, "permute_calls": pc.value, "permute_bytes": pb.value}

    def prof_enable(self, on=True):
        self.lib.call("pymes_prof_enable", self.handle, int(on))
        self.profiling = bool(on)

    def prof_reset(self):
        self.lib.call("pymes_prof_reset", self.handle)

    def prof_query(self, kernel_class=0):
        """HIP-event totals of the fp64 GEMM calls since prof_reset (kernel_class 1: LDS-DMA kernel only)."""
        n, nk, ms, fl = C.c_int64(), C.c_int64(), C.c_double(), C.c_double()
        self.lib.call("pymes_prof_query", self.handle, int(kernel_class), C.byref(n), C.byref(nk), C.byref(ms),
                      C.byref(fl))
        return {"launches": n.value, "kernel_launches": nk.value, "ms": ms.value, "flops": fl.value}


__all__ = ["Context", "DeviceArray", "PymesError", "BLOCK_NAMES", "pattern_of"]
