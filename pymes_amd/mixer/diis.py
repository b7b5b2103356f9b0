"""DIIS mixer with the bookkeeping of pymes/mixer/diis.py:16-112, on device arrays.

The stored error/amplitude vectors stay in HBM; per call the host sees only the
overlaps <e_i, e_new> (one fused multi-dot) and the (<=7)x(<=7) system.  The reference's
full-subspace quirk is reproduced on purpose: once ``dim_space`` vectors
are stored, the shifted copy of the old L (diis.py:59-60) omits the row/column of the
second-newest vector, which therefore become zeros — iteration histories depend on it.
"""
import os

import numpy as np

from pymes_amd import _lib
from pymes_amd.log import print_logging_info

_blas_threads = None


def _single_threaded_blas():
    """Context manager: the host BLAS / LAPACK runs the (<= 7) x (<= 7) algebra below on one thread.  A GPU node shows
    hundreds of cores to OpenBLAS while the process may own a small share of them; waking that pool for a 7 x 7 matrix
    costs more than the solve and — measured — leaves the GPU idle for milliseconds between the overlaps and the
    extrapolation."""
    global _blas_threads
    if _blas_threads is None:
        try:
            from threadpoolctl import ThreadpoolController
            _blas_threads = ThreadpoolController()
        except Exception:          # not installed: run as numpy is configured
            _blas_threads = False
    if _blas_threads is False:
        import contextlib
        return contextlib.nullcontext()
    return _blas_threads.limit(limits=1, user_api="blas")


class DIIS:
    def __init__(self, dim_space=5):
        self.dim_space = dim_space
        self.L = np.zeros((1, 1))
        self.error_list = []
        self.amplitude_list = []
        self.last_coefficients = None
        self._state, self._stale, self._log_pending = None, False, False

    # -- history across contexts: the reference's mixer outlives a solve() call ---------
    # Device vectors cannot outlive their context.  The mixer registers itself with every context it stores vectors in
    # (``Context.on_close``); when such a context closes, its stored vectors are parked on the host — up to PARK_LIMIT
    # bytes; a larger history (12 amplitude sets, 9.6 GB at (50,200)) is dropped instead: a second solve() on the same
    # solver instance then starts from an empty subspace, the only place where this drop-in knowingly leaves the
    # reference's never-reset semantics (``DIIS.PARK_LIMIT`` moves the limit).
    PARK_LIMIT = 2 << 30

    def _reset(self):
        self.error_list, self.amplitude_list, self.L = [], [], np.zeros((1, 1))
        self._state, self._stale = None, False

    # -- the small algebra on the device (pymes_diis_step): L and the coefficients live in a 96-double device array; the
    # host copies (self.L, self.last_coefficients) are refreshed only when somebody asks (logging, parking)
    def _device_state(self, ctx):
        st = getattr(self, "_state", None)
        if st is None or st.ctx is not ctx or ctx.handle is None:
            self._refresh_host()
            buf = np.zeros(96)
            n = self.L.shape[0]
            buf[0] = n
            pad = np.zeros((9, 9))
            pad[:n, :n] = self.L
            buf[1:82] = pad.ravel()
            st = ctx.array(buf)
            self._state, self._stale = st, False
            ctx.on_close(self._state_closing)
        return st

    def _refresh_host(self):
        """Bring self.L / self.last_coefficients up to date with the device state (one small download)."""
        st = getattr(self, "_state", None)
        if st is not None and getattr(self, "_stale", False) and st.ctx.handle is not None:
            slot, self._log_slot = getattr(self, "_log_slot", None), None
            # (the copy that mix() started right behind the step kernel: waiting for it does not drain the stream, on which
            # the caller may already have enqueued its next residual)
            buf = st.ctx.readback_wait(slot, 96) if slot is not None else st.get()
            n = int(buf[0])
            self.L = buf[1:82].reshape(9, 9)[:n, :n].copy()
            self.last_coefficients = buf[82:82 + n].copy()
            self.last_dependent = bool(buf[91] == 1.0)
            if buf[91] == 2.0:
                # the step on the device found no finite solution (its extrapolation kept the newest amplitudes): the
                # reference's numpy.linalg raises at this point (diis.py:85-95)
                self._stale = False
                raise np.linalg.LinAlgError("DIIS: singular or non-finite subspace matrix")
        self._stale = False

    def _state_closing(self, ctx):
        try:
            if getattr(self, "_state", None) is not None and self._state.ctx is ctx:
                self._refresh_host()
                self._state = None
        except Exception:
            self._state = None

    def _stored(self):
        return [arr for lst in (self.error_list, self.amplitude_list) for vec in lst for arr in vec]

    def park(self, ctx):
        """Move the stored vectors that live in ``ctx`` to host memory (the context is about to be destroyed).  Never
        raises: called from ``finally`` blocks and from ``Context.close``."""
        import os
        self._state_closing(ctx)
        try:
            mine = [a for a in self._stored() if not isinstance(a, np.ndarray) and a.ctx is ctx]
            if not mine:
                return
            limit = int(self.PARK_LIMIT)
            if ctx.handle is None or sum(a.nbytes for a in mine) > limit:
                self._reset()
                return
            for lst in (self.error_list, self.amplitude_list):
                for vec in lst:
                    for k, arr in enumerate(vec):
                        if not isinstance(arr, np.ndarray) and arr.ctx is ctx:
                            vec[k] = arr.get()
        except Exception:
            self._reset()

    def _adopt(self, ctx, like):
        """Make every stored vector a live array of ``ctx``: parked host copies are uploaded, vectors of another live
        context are migrated through the host, and a history that cannot be mixed with — vectors of a closed context, or
        of another problem size / sharding (the reference would fail in its einsum there) — is dropped."""
        shapes = [tuple(a.shape) for a in like]
        for arr in self._stored():
            if not isinstance(arr, np.ndarray) and arr.ctx is not ctx and arr.ctx.handle is None:
                return self._reset()
        for lst in (self.error_list, self.amplitude_list):
            for vec in lst:
                if [tuple(a.shape) for a in vec] != shapes:
                    return self._reset()
        for lst in (self.error_list, self.amplitude_list):
            for vec in lst:
                for k, arr in enumerate(vec):
                    if isinstance(arr, np.ndarray):
                        vec[k] = ctx.pool_get(arr.shape).set(arr)
                    elif arr.ctx is not ctx:          # its old buffer goes back to its own context when the handle dies
                        vec[k] = ctx.pool_get(arr.shape).set(arr.get())
        ctx.on_close(self.park)

    # -- host part: identical arithmetic to the reference ------------------------------
    def _update_L(self, overlaps, was_full):
        m = len(overlaps)
        L = np.zeros((m + 1, m + 1))
        L[-1, :-1] = -1.0
        L[:-1, -1] = -1.0
        if was_full:
            L[:-3, :-3] = self.L[1:-2, 1:-2]
        else:
            L[:-2, :-2] = self.L[:-1, :-1]
        L[:m, -2] += overlaps
        L[-2, :] = L[:, -2]
        self.L = L.copy()

    def _solve(self):
        with _single_threaded_blas():
            return self._solve_on_this_thread()

    def _solve_on_this_thread(self):
        unit = np.zeros(self.L.shape[0])
        unit[-1] = -1.0
        lam, vec = np.linalg.eigh(self.L)
        if np.any(np.abs(lam) < 1e-12):
            print_logging_info("Linear dependence found in DIIS subspace.", level=2)
            ok = np.abs(lam) > 1e-12
            return np.dot(vec[:, ok] * (1.0 / lam[ok]), np.dot(vec[:, ok].T.conj(), unit))
        return np.linalg.inv(self.L).dot(unit)

    def mix(self, error, amplitude, release=None, sharded=(), allreduce=None, out=None, mark=None, on_device=False,
            defer_log=False, native=False):
        """error / amplitude: lists of DeviceArray (one entry per amplitude type).
        Returns freshly allocated DeviceArrays with the extrapolated amplitudes.  The
        mixer keeps references to the arrays passed in (like the reference): the caller must
        not modify them afterwards.  ``release(arr)`` is called for vectors that leave the
        subspace so a caller-side pool can recycle their memory.  ``sharded``: indices of the amplitude types of which
        this process holds only its share (one process per GPU); their overlaps are summed over the ranks with
        ``allreduce`` (a callable on a small numpy vector), the extrapolation itself is local.  ``out``: arrays to
        write the extrapolated amplitudes into instead of fresh ones (the solvers keep T1/T2 in fixed buffers so that
        their loop body can be replayed as a launch graph); they must not be among the stored vectors."""
        self._adopt(error[0].ctx, error)
        was_full = len(self.error_list) == self.dim_space
        if was_full:
            old_e, old_a = self.error_list.pop(0), self.amplitude_list.pop(0)
            if release is not None:
                for arr in list(old_e) + list(old_a):
                    release(arr)
        self.error_list.append(list(error))
        self.amplitude_list.append(list(amplitude))
        ctx = error[0].ctx
        m, ntypes = len(self.error_list), len(error)
        if on_device and not sharded and ntypes * m <= 16 and m <= 8:
            # ``on_device``: overlaps, the (m+1) x (m+1) solve and the extrapolation without a host round trip; returns the
            # arrays and leaves the log lines to ``self.log_last()`` (the caller prints them once it has synchronised anyway)
            state = self._device_state(ctx)
            ctx.diis_step(state, [self.error_list[i][nt] for nt in range(ntypes) for i in range(m)],
                          [error[nt] for nt in range(ntypes) for _ in range(m)], ntypes, m, was_full)
            self._stale = True
            self._log_slot = ctx.readback_start(state, 96)       # L and the coefficients for log_last(), read back on the side
            res = []
            for nt in range(ntypes):
                dst = out[nt] if out is not None else ctx.pool_get(amplitude[nt].shape)
                ctx.lincomb_dev(dst, [self.amplitude_list[a][nt] for a in range(m)], state.ptr + 8 * 82)
                res.append(dst)
            self._log_pending = True
            return res
        self._refresh_host()
        self._state = None                      # (the host path owns L from here on)
        if native and not sharded and ntypes * m <= 16 and m <= 8:
            # overlaps (device) -> synchronise -> small algebra in C on this thread -> extrapolation enqueued: one library call
            # (pymes_diis_mix), no interpreter between the synchronisation and the next kernel
            buf = np.zeros(96)
            n0 = self.L.shape[0]
            buf[0] = n0
            pad = np.zeros((9, 9))
            pad[:n0, :n0] = self.L
            buf[1:82] = pad.ravel()
            res = [out[nt] if out is not None else ctx.pool_get(amplitude[nt].shape) for nt in range(ntypes)]
            ctx.diis_mix(buf, [self.error_list[i][nt] for nt in range(ntypes) for i in range(m)], list(error),
                         [self.amplitude_list[a][nt] for nt in range(ntypes) for a in range(m)], res, m, was_full)
            n1 = int(buf[0])
            self.L = buf[1:82].reshape(9, 9)[:n1, :n1].copy()
            self.last_coefficients = buf[82:82 + n1].copy()
            self.last_dependent = bool(buf[91] == 1.0)
            if mark is not None:
                mark("DIIS overlaps + host solve (one call)")
            if defer_log:
                self._log_pending = True
            else:
                self.log_last(force=True)
            return res
        overlaps = np.zeros(m)
        # all <e_i, e_new> of all amplitude types in one launch and one synchronisation; summed per type on the host
        # in the order of the reference's loop (diis.py:65-78)
        parts = ctx.dots([self.error_list[i][nt] for nt in range(ntypes) for i in range(m)],
                         [error[nt] for nt in range(ntypes) for _ in range(m)])
        for nt in range(ntypes):
            part = parts[nt * m:(nt + 1) * m]
            overlaps += allreduce(part) if nt in sharded else part
        if mark is not None:
            mark("DIIS overlaps (reduction + host sync)")
        dependent = False
        if m <= 8 and not os.environ.get("PYMES_NUMPY_DIIS"):
            # the small algebra in C on this thread (pymes_diis_solve: the code of pymes_diis_mix without its device part) —
            # numpy.linalg under a one-thread BLAS limit took 0.4 ms here, mostly the limit itself (threadpoolctl)
            buf = np.zeros(96)
            n0 = self.L.shape[0]
            buf[0] = n0
            pad = np.zeros((9, 9))
            pad[:n0, :n0] = self.L
            buf[1:82] = pad.ravel()
            ctx.lib.call("pymes_diis_solve", _lib.host_ptr(buf), _lib.host_ptr(np.ascontiguousarray(overlaps, dtype=np.float64)),
                         1, int(m), int(bool(was_full)))
            if buf[91] == 2.0:
                raise np.linalg.LinAlgError("DIIS: the subspace matrix is singular or not finite")
            n1 = int(buf[0])
            self.L = buf[1:82].reshape(9, 9)[:n1, :n1].copy()
            c = buf[82:82 + n1].copy()
            dependent = bool(buf[91] == 1.0)
        else:
            self._update_L(overlaps, was_full)
            c = self._solve()
        self.last_coefficients = c
        if mark is not None:
            mark("DIIS host solve")
        res = []
        for nt in range(ntypes):
            dst = out[nt] if out is not None else ctx.pool_get(amplitude[nt].shape)
            ctx.lincomb(dst, [self.amplitude_list[a][nt] for a in range(m)], c[:m])
            res.append(dst)
        out = res
        if defer_log:           # the caller prints these lines (log_last) once its next kernels are on their way
            self._log_pending, self.last_dependent = True, dependent
        else:
            if dependent:
                print_logging_info("Linear dependence found in DIIS subspace.", level=2)
            self._log(c)
        return out

    def _log(self, c):
        print_logging_info("diis.mix", level=2)
        print_logging_info("Coefficients for combining amplitudes=", level=3)
        print_logging_info(c[:-1], level=3)
        print_logging_info("Sum of coefficients = {:.8f}".format(np.sum(c[:-1])), level=3)
        print_logging_info("Lagrangian multiplier = {:.8f}".format(c[-1]), level=3)

    def log_last(self, force=False):
        """The log lines of the last ``mix(defer_log=True)`` (diis.py:86, :104-111), printed once the caller's next kernels
        are on their way (the energy read-back of the iteration)."""
        if getattr(self, "_log_pending", False) or force:
            self._log_pending = False
            self._refresh_host()
            if getattr(self, "last_dependent", False):
                print_logging_info("Linear dependence found in DIIS subspace.", level=2)
            self._log(self.last_coefficients)
