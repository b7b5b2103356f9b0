"""T1-similarity-transformed CCSD / DCSD (pymes/solver/ccsd.py) on the MI355X engine.

Drop-in for ``pymes.solver.ccsd.CCSD``: same constructor, attributes, ``solve``
signature, return dictionary and public helper methods.  Inside ``solve`` everything
(T1, T2, dressed Fock/V blocks, residuals, DIIS history) stays in HBM; the host sees
scalars only.  With ``torch.distributed`` initialised (one process per GPU) the
particle-particle ladder is sharded over the ranks on the virtual index ``a``
(pymes_amd/dist.py)."""
import ctypes as C
import os
import time

import numpy as np

from pymes_amd import _lib
from pymes_amd import dist as pdist
from pymes_amd.device import Context, DeviceArray
from pymes_amd.integral.device import DeviceIntegrals, DressedDeviceIntegrals
from pymes_amd.integral.partition import BLOCK_NAMES
from pymes_amd.log import print_logging_info
from pymes_amd.mixer import diis
from pymes_amd.solver import ccd


# blocks produced by get_T1_dressed_V (ccsd.py:322-419); the other five names stay None (:317)
DRESSED_KEYS = ("abij", "klij", "ijab", "ijka", "ijak", "iajb", "iabj", "iabc", "abic", "iajk", "abcd")
# what the T2 residual consumes (ccsd.py:449-454); ijab is a plain copy (:355-357) and is read undressed
LOOP_KEYS = ("abij", "klij", "iajb", "iabj", "abcd")


class CCSD(ccd.CCD):
    def __init__(self, no, is_diis=True, delta_e=1.e-8, is_non_canonical=False, is_dcsd=False, device=0):
        self.t_T_ai = None
        self.t_T_abij = None
        self.is_dcd = is_dcsd
        self.is_diis = is_diis
        self.is_bruekner = False
        self.is_dr_ccd = False
        self.no = no
        self.max_iter = 50
        self.delta = 1.0
        self.delta_e = delta_e
        self.debug_level = 1
        self.device = device
        if self.is_diis:
            self.mixer = diis.DIIS(dim_space=6)

    def write_logging_info(self):
        return

    # ------------------------------------------------------------------------------------
    # iteration machinery (also driven directly by bench.py)
    # ------------------------------------------------------------------------------------
    def setup(self, t_fock_pq, ints, level_shift=0., amps=None):
        """Everything before the while loop of ccsd.py:47-157.  Returns the state dict."""
        ctx = ints.ctx
        ctx.trim()           # recycled temporaries of earlier work on this context: what follows allocates engine-side
        no, nv = self.no, ctx.nv
        f = np.asarray(t_fock_pq, dtype=np.float64)
        eps_i, eps_a = f.diagonal()[:no].copy(), f.diagonal()[no:].copy()
        ctx.set_orbital_energies(eps_i, eps_a)
        rank, wsize, _ = pdist.world()
        dist_on = pdist.sharded()       # one process per GPU (or the forced one-rank rehearsal of that path)
        if dist_on:
            pdist.bind_stream(ctx)      # engine kernels and RCCL collectives ordered on one stream: no host fences
        st = {"ctx": ctx, "dist": dist_on, "f": ctx.array(f), "fd": ctx.empty(f.shape), "level_shift": level_shift,
              "eps_i": eps_i, "eps_a": eps_a, "rank": rank, "world": wsize, "first": True, "amps": amps}
        t2 = ctx.pool_get((nv, nv, no, no))
        e_dir, e_exc = ctx.mp2(t2, level_shift)                     # ccsd.py:128
        st["e_mp2"] = e_dir + e_exc
        t1 = ctx.pool_get((nv, no)).zero_()
        if amps is not None:
            t1.set(np.asarray(amps[0]))
            t2.set(np.asarray(amps[1]))
        st["t1"], st["t2"] = t1, t2
        # pair-packed ladder (1/4 of the flops) whenever T2 has the exchange symmetry T_abij = T_baji,
        # i.e. always when starting from MP2; user amplitudes are checked
        # ... and the integrals must have the electron-exchange symmetry V_pqrs = V_qpsr (every FCIDUMP-derived or
        # transcorrelated Hamiltonian has it; the reference does not assume it, so anything else takes the general path)
        st["sym"] = ctx.V_exchange_symmetric() and (amps is None or ctx.exchange_symmetric(t2))
        st["npp"] = nv * (nv + 1) // 2
        # world > 1: every rank keeps T2-sized quantities (residual, update, DIIS history) only for the virtual pairs it
        # owns ("pair-sharded tail", include/pymes_amd.h); user amplitudes keep the replicated tail (in-place contract)
        st["pairs"] = bool(dist_on and st["sym"] and amps is None and ctx.pairs_supported())
        # PYMES_OWNER_TILES=1: the rows of the ring products travel as an all-to-all of the tiles each pair owner reads
        # (0.2 GB per rank at (50,200) on 8 ranks) instead of two all-gathers of the whole matrices (1.6 GB); dist.py
        st["owner_tiles"] = bool(st["pairs"] and os.environ.get("PYMES_OWNER_TILES"))
        lo, hi = pdist.slab_rows(st["npp"], rank, wsize)
        st["cshape"] = (max(hi - lo, 1), 2, no * no)
        if self.is_diis:     # DIIS keeps dim_space (dT, T) pairs + the mixed result + residual/update scratch
            ctx.pool_reserve(st["cshape"] if st["pairs"] else t2.shape, 2 * self.mixer.dim_space + 4)
            ctx.pool_reserve(t1.shape, 2 * self.mixer.dim_space + 4)
        if dist_on:
            import torch
            dev = torch.device("cuda", ctx.device) if ctx.lib.backend.startswith("hip") else torch.device("cpu")

            def shared(rows, cols):      # exchange buffer: world equal row chunks, torch-owned for the collective
                t = torch.zeros((pdist.padded_rows(rows, wsize), cols), dtype=torch.float64, device=dev)
                return t, DeviceArray(ctx, t.data_ptr(), tuple(t.shape), owned=False, keepalive=t)
            if st["sym"]:
                st["ETd_t"], st["ETd"] = shared(no * nv, no * nv)
                st["ETx_t"], st["ETx"] = shared(no * nv, no * nv)
                st["L_t"], st["L"] = shared(st["npp"], no * no)
                st["QK_t"], st["QK"] = shared(no * nv, no * no)
                if st["pairs"]:
                    st["Tall_t"], st["Tall"] = shared(st["npp"], 2 * no * no)       # exchange buffer of the compact T2
                    st["Tc"] = self._compact(ctx, st)
                    ctx.pairs_pack(t2, st["Tc"], rank, wsize)

                def reduced(n):          # small buffer that is summed over the ranks
                    t = torch.zeros((n,), dtype=torch.float64, device=dev)
                    return t, DeviceArray(ctx, t.data_ptr(), (n,), owned=False, keepalive=t)
                st["W_t"], st["W"] = reduced(ctx.dress_fock_ws())
                st["Xvv_t"], st["Xvv"] = reduced(nv * nv)
                st["P_t"], st["P"] = reduced(ctx.slab_prepare_ws())
                st["R1_t"], st["R1"] = reduced(nv * no)
                st["R1"] = st["R1"].reshape(nv, no)
                if st["pairs"] and not (st["owner_tiles"] and os.environ.get("PYMES_PY_SEQUENCED")):
                    # the loop body as whole library steps (pymes_ccsd_sharded_residuals / _finish) that call back for their
                    # collectives (include/pymes_amd.h, pymes_collectives; pymes_amd/dist.py:Collectives fills the table with
                    # torch.distributed): the sequence a host in any language would run — since round 6 with the owner-tile
                    # all-to-all as well (pymes_set_alltoallv; PYMES_PY_SEQUENCED=1 keeps the Python-sequenced form of it).
                    # (The replicated tail of user amplitudes keeps the Python-sequenced form below.)
                    st["S_t"], st["S"] = reduced(8)
                    names = ("ETd", "ETx", "L", "QK", "Tall", "W", "Xvv", "P", "R1", "S")
                    st["coll"] = pdist.Collectives(ctx, {k: st[k + "_t"] for k in names}, rank, wsize)
                    st["bufs"] = _lib.ShardBuffers(*[st[k + "_t"].data_ptr() for k in names])
                    st["rc"] = self._compact(ctx, st)
                    st["flags"] = _lib.PYMES_DCD if self.is_dcd else 0
                    if st["owner_tiles"]:
                        st["coll"].enable_owner_tiles(dev)
                        st["flags"] |= _lib.PYMES_OWNER_TILES
            else:
                st["lad_rows"] = nv * nv
                st["lad_t"], st["lad"] = shared(nv * nv, no * no)
        if not dist_on:
            # single rank: T1, T2 and the residuals live in fixed buffers, so the residual part of the loop body (about
            # 130 kernel launches) can be recorded once as a launch graph and replayed (frees the host; measured: the device
            # executes the same latency-bound kernels either way)
            st["r1"], st["r2"] = ctx.pool_get(t1.shape), ctx.pool_get(t2.shape)
            if not self.is_diis:
                st["dt1"], st["dt2"] = ctx.pool_get(t1.shape), ctx.pool_get(t2.shape)
            st["graph"], st["eager_passes"] = None, 0
            st["eager_passes_of"] = {}
            # T1 starts at zero (MP2) unless the caller brought amplitudes
            st["t1_zero"] = amps is None and not os.environ.get("PYMES_NO_T1_SHORTCUT")
            st["graph_ok"] = ctx.graphs_supported() and not os.environ.get("PYMES_NO_GRAPH")
        return st

    def iterate(self, st):
        """One pass of the loop body ccsd.py:159-209.  Returns (e_1b, e_dir, e_ex, |T2|, |dT2|)."""
        return self._iterate_sharded(st) if st["dist"] else self._iterate_single(st)

    # ---- single rank ------------------------------------------------------------------------------------------------
    def _residuals(self, st):
        """ccsd.py:161-171: dressed Fock, dressed V blocks, R1, R2 from the fixed T1/T2 buffers into the fixed residual
        buffers.  Only enqueues kernels (no host read-back), hence replayable as a launch graph."""
        ctx, t1, t2, r1, r2 = st["ctx"], st["t1"], st["t2"], st["r1"], st["r2"]
        if st["sym"]:
            # Symmetry-reduced form = the one-rank case of the sharded form, ONE library call (pymes_ccsd_residuals: dressed
            # Fock :163, the dressed blocks the loop reads :165, R1 :167, R2 :171).  V_abcd is never dressed, its T1 dressing
            # (ccsd.py:414-419) is carried by tau = T2 + T1 T1 inside the ladders, that of V_abij by Q_kb and two small
            # products inside the finish (include/pymes_amd.h, pymes_residual_slab).  T1 = 0 exactly (the MP2 start; every
            # iteration of a momentum-conserving system such as the UEG): exp(-T1) H exp(T1) = H, :163 and :165 are
            # identities — the residuals straight from the undressed f and V
            ctx.ccsd_residuals(st["f"], t1, t2, r1, r2, is_dcd=self.is_dcd, t1_zero=st["t1_zero"])
        else:
            # general path (amplitudes or integrals without the exchange symmetry): explicitly dressed blocks
            ctx.dress_fock(st["f"], t1, st["fd"])                                     # :163
            ctx.singles_residual(st["fd"], t1, t2, r1)                                # :167
            ctx.dress_V(t1, LOOP_KEYS)                                                # :165
            ctx.doubles_residual(st["fd"], t2, r2, is_dcd=self.is_dcd, dressed=True, sym_ladder=False)   # :171

    def _launch_residuals(self, st):
        ccd.run_replayable(st["ctx"], st, lambda: self._residuals(st), key="t1=0" if st["t1_zero"] else "t1")

    def _iterate_single(self, st):
        """One pass of the loop body on one rank, software-pipelined: the reference reads the energy back (ccsd.py:189-197)
        before it builds the next residual; here the device does not wait for that read-back.  The energy reduction is only
        ENQUEUED, the residual kernels of the NEXT pass — which need the extrapolated amplitudes, not the energy — are enqueued behind
        it, and only then does the host wait for the six numbers of this pass (an event, not the stream).  A pass therefore
        finds its residuals computed already; a solve pays for this with one residual build it does not use (the one
        enqueued by its last pass), which ``st["speculate"] = False`` switches off when the caller expects the pass to be
        the last.  At (20,80) that round trip (read-back, interpreter, graph launch) was 85 us of a 2.1-ms iteration."""
        ctx, t1, t2, r1, r2 = st["ctx"], st["t1"], st["t2"], st["r1"], st["r2"]
        shift = st["level_shift"]
        mark = pdist.trace.mark          # device and host time per phase when bench.py asks for it (no-ops otherwise)
        mark("begin")
        if not st.pop("residuals_in_flight", False):
            self._launch_residuals(st)
        mark("residuals (dressing, R1, R2)")
        # update, mixer and energy reduction as ONE held phase (include/pymes_amd.h, pymes_phase_hold): the two updates share a
        # launch, and so do the extrapolations of T1 and T2 with nothing of the interpreter between them and the energy kernel;
        # the overlaps' read-back inside the mixer launches what is recorded before it waits
        with ctx.phase_hold():
            if self.is_diis:
                # the DIIS history keeps the updated amplitudes (:176-183); the extrapolation goes back into the fixed buffers
                t1n, t2n, dt1, dt2 = (ctx.pool_get(t1.shape), ctx.pool_get(t2.shape), ctx.pool_get(t1.shape),
                                      ctx.pool_get(t2.shape))
                ctx.cc_update_to(t1n, dt1, t1, r1, shift, self.delta)                     # :176-179
                ctx.cc_update_to(t2n, dt2, t2, r2, shift, self.delta)
            else:
                t1n, t2n, dt1, dt2 = t1, t2, st["dt1"], st["dt2"]
                ctx.cc_update(t1, dt1, r1, shift, self.delta)
                ctx.cc_update(t2, dt2, r2, shift, self.delta)
            if st["first"] and st["amps"] is not None:
                np.copyto(st["amps"][0], t1n.get())     # the reference updates the caller's arrays in place
                np.copyto(st["amps"][1], t2n.get())
            st["first"] = False
            mark("update")
            if self.is_diis:
                # :181-183; the mixer's log lines are printed (log_last) after the energy has been read back.  Default: overlaps
                # reduced on the device, ONE synchronisation, the 7 x 7 algebra in C on this thread, extrapolation enqueued
                # (pymes_diis_mix).  PYMES_DEVICE_DIIS=1: the whole step on the device (pymes_diis_step) — no round trip, and
                # a few microseconds while a bound on |lambda_min| proves the inverse branch of diis.py:95; but a converging
                # solve soon has overlaps below 1e-12, the reference's pseudo-inverse branch (:85-93) is then the one that is
                # MEANT, and its eigen-decomposition on one wave costs 0.4 ms against 0.13 ms for the round trip (measured at
                # (20,80), round 4).  PYMES_NUMPY_DIIS=1: numpy.linalg as the reference.
                self.mixer.mix([dt1, dt2], [t1n, t2n], release=ctx.pool_put, out=[t1, t2], mark=mark, defer_log=True,
                               on_device=bool(os.environ.get("PYMES_DEVICE_DIIS")),
                               native=not os.environ.get("PYMES_NUMPY_DIIS"))
            mark("DIIS extrapolation")
            slot = ctx.energy_norms_start(st["f"], t1, t2, dt2)                            # :189-197, one pass, enqueued
        # the next pass's residuals, behind the energy reduction: same variant as this pass's (T1 = 0 only ever changes after
        # the first pass from MP2, or never: a momentum-conserving system); recorded launch graphs only — the eager first
        # passes of a variant do their per-solve set-up work and stay where they were
        key = "t1=0" if st["t1_zero"] else "t1"
        speculate = (st.get("speculate", True) and not ctx.profiling and st.get("graphs", {}).get(key) is not None
                     and not os.environ.get("PYMES_NO_PIPELINE"))
        if speculate:
            self._launch_residuals(st)
        e1, ed, ex, nt2, nr2, n1 = ctx.energy_norms_wait(slot)
        # (pipelined: the NEXT pass's residual segment has been enqueued in front of this wait and is charged to this mark)
        mark("energy + norms (read-back) + the next pass's residuals, enqueued ahead" if speculate else "energy + norms (read-back)")
        if self.is_diis:
            self.mixer.log_last()
        was_zero = st["t1_zero"]
        st["t1_zero"] = bool(n1 == 0.0) and not os.environ.get("PYMES_NO_T1_SHORTCUT")
        # (a variant switch — T1 became non-zero — invalidates what was enqueued for the old variant: the next pass builds its
        # residuals itself; the stale kernels only wrote the residual buffers, which are overwritten)
        st["residuals_in_flight"] = bool(speculate and st["t1_zero"] == was_zero)
        return e1, ed, ex, np.sqrt(nt2), np.sqrt(nr2)

    # ---- one process per GPU ----------------------------------------------------------------------------------------
    def _hooked_residuals(self, st):
        ctx = st["ctx"]
        st["coll"].call("pymes_ccsd_sharded_residuals", ctx.handle, C.c_void_p(st["f"].ptr), C.c_void_p(st["fd"].ptr),
                        C.c_void_p(st["t1"].ptr), C.c_void_p(st["t2"].ptr), C.byref(st["bufs"]), st["flags"],
                        C.c_void_p(st["rc"].ptr))

    def _iterate_hooked(self, st):
        """One pass on one rank of many, as whole library steps with the collective table (setup): residuals — enqueued by the
        PREVIOUS pass behind its energy reduction whenever that pass did not expect to be the last, as on a single rank —
        update, mixer (its overlaps are the one host round trip of a pass), then energies all-reduced on the device and
        copied to the host on the side, the next residuals enqueued, and only then the six numbers are waited for."""
        ctx, t1, tc = st["ctx"], st["t1"], st["Tc"]
        rank, world, shift = st["rank"], st["world"], st["level_shift"]
        coll, mark = st["coll"], pdist.trace.mark
        if not st.pop("residuals_in_flight", False):
            self._hooked_residuals(st)
        dt1, dtc = ctx.pool_get(t1.shape), self._compact(ctx, st)
        ctx.cc_update(t1, dt1, st["R1"], shift, self.delta)                             # :176-179
        ctx.cc_update_pairs(tc, dtc, st["rc"], shift, self.delta, rank, world)
        st["first"] = False
        mark("update")
        if self.is_diis:
            t1, tc = self.mixer.mix([dt1, dtc], [t1, tc], release=ctx.pool_put, sharded=(1,),
                                    allreduce=pdist.allreduce_sum, mark=mark)            # :181-183
        mark("DIIS extrapolation")
        st["t1"], st["Tc"] = t1, tc
        slot = C.c_int()
        coll.call("pymes_ccsd_sharded_finish", ctx.handle, C.c_void_p(st["f"].ptr), C.c_void_p(t1.ptr), C.c_void_p(tc.ptr),
                  C.c_void_p(dtc.ptr), C.byref(st["bufs"]), C.byref(slot))
        if not self.is_diis:
            ctx.pool_put(dt1)
            ctx.pool_put(dtc)
        if st.get("speculate", True):
            self._hooked_residuals(st)
            st["residuals_in_flight"] = True
        out = (C.c_double * 6)()
        coll.call("pymes_ccsd_sharded_energy", ctx.handle, slot.value, out)
        mark("energy + norms (read-back)")
        return out[0], out[1], out[2], np.sqrt(out[3]), np.sqrt(out[4])

    def _iterate_sharded(self, st):
        if "coll" in st:
            return self._iterate_hooked(st)
        ctx, t1, t2 = st["ctx"], st["t1"], st["t2"]
        shift = st["level_shift"]
        world, rank, dist_on = st["world"], st["rank"], st["dist"]
        mark = pdist.trace.mark          # per-phase device time + exposed communication for bench.py (no-ops otherwise)
        mark("begin")
        if dist_on and st["sym"]:
            # K-sharded partial sums, all-reduced: the T1.V intermediates of the dressed Fock (:163, this rank's chunk of
            # j) and the slab's small V.T intermediates (X_ki, hole-ladder V_klcd T_cdij; this rank's chunk of c / (c,d))
            # What needs T1 only comes first: the all-gather of the new T2 that the previous iteration's tail started is
            # still in flight (0.8 GB at (50,200)) and is awaited — and unpacked into the replicated array — right
            # before the first kernel that reads T2.
            ctx.dress_fock_partial(t1, st["W"], rank, world)
            red = [pdist.allreduce_tensor_start(st["W_t"], ctx, label="fock intermediates")]
            # V~_iajb / V~_iabj only for the second-index range that this rank's column slab reads
            c0, c1 = pdist.slab_rows(ctx.no * ctx.nv, rank, world)
            if c1 > c0:      # one call: V~_klij and V~_iabj share their V_klcd t_dj intermediate       # :165
                ctx.dress_V(t1, ("klij", "iajb", "iabj"), q_range=(c0 // ctx.no, -(-c1 // ctx.no)))
            else:
                ctx.dress_V(t1, ("klij",))
            mark("T1-only: fock partial, dress V slab")
            self._await_t2(st)
            ctx.slab_prepare(t2, st["P"], rank, world, is_dcd=self.is_dcd)
            # P = [ X'_ki (o^2 doubles: read by the ring half) | pair-packed 2 V_klcd T_cdij (26 MB at (50,200): read by the
            # ladder half) ]: two all-reduces, the big one is awaited only in front of the ladders — behind the ring products
            oo = ctx.no * ctx.no
            red.append(pdist.allreduce_tensor_start(st["P_t"][:oo], ctx, label="X_ki"))
            st["J_pending"] = pdist.allreduce_tensor_start(st["P_t"][oo:], ctx, label="hole-ladder J")
            for work in red:
                work.wait()
            ctx.dress_fock_finish(st["f"], t1, st["W"], st["fd"])
            mark("await T2, slab prepare, fock finish")
        else:
            self._await_t2(st)
            ctx.dress_fock(st["f"], t1, st["fd"])                    # :163
        # the singles residual of the symmetric path is an all-reduced partial sum: it lives in its exchange buffer
        r1 = st["R1"] if st["sym"] else ctx.pool_get(t1.shape)
        if st["sym"]:
            # Symmetry-reduced, sharded form (world = 1 included): this rank's column slab of the ring products,
            # its rows of the pair-packed particle + hole ladders and of Q_kb; all-gathers; remainder.
            # V_abcd is never dressed: its T1 dressing (:165, ccsd.py:414-419) is carried by tau = T2 + T1 T1
            # inside the ladders, that of V_abij by Q_kb and two small products inside the finish (include/pymes_amd.h,
            # pymes_residual_slab).  The singles residual is enqueued after the all-gathers have been started: overlap.
            if not dist_on:  # (one process per GPU: dressed above, before T2 was needed)
                ctx.dress_V(t1, ("klij", "iajb", "iabj"))                             # :165
            # :171 in two halves: the ring products first, so that the all-gathers of their rows (ETd, ETx: 1.6 of the 1.8 GB
            # that an iteration exchanges at (50,200)) fly while the ladders — whose rows of L never leave the rank in the
            # pair-sharded tail — and the singles residual are computed
            slab = dict(is_dcd=self.is_dcd, dressed=True, t1=t1, QK=st["QK"], P=st["P"] if dist_on else None)
            pending = []
            if dist_on:
                ctx.residual_slab(st["fd"], t2, st["ETd"], st["ETx"], st["L"], rank, world, part="rings", **slab)
                mark("ring products")
                if st["owner_tiles"]:
                    pending = [pdist.exchange_pair_tiles_start([st["ETd_t"], st["ETx_t"]], ctx.no, ctx.nv, rank, world, ctx,
                                                               label="ETd+ETx owner tiles")]
                else:
                    pending = [pdist.exchange_rows_start(st[key], rank, world, ctx, label=key[:3])
                               for key in ("ETd_t", "ETx_t")]
                st.pop("J_pending").wait()
                ctx.residual_slab(st["fd"], t2, st["ETd"], st["ETx"], st["L"], rank, world, part="ladders", **slab)
                mark("ladders, Q_kb")
                pending += [pdist.exchange_rows_start(st[key], rank, world, ctx, label=key[:-2])
                            for key in (("QK_t",) if st["pairs"] else ("L_t", "QK_t"))]
                if st["pairs"]:      # X_ac (:206-221) as a partial sum over this rank's chunk of k, all-reduced below
                    ctx.xvv_partial(st["fd"], t2, st["Xvv"], rank, world, is_dcd=self.is_dcd)
                    pending.append(pdist.allreduce_tensor_start(st["Xvv_t"], ctx, label="X_ac"))
            else:
                ctx.residual_slab(st["fd"], t2, st["ETd"], st["ETx"], st["L"], rank, world, **slab)
            # :167 as a partial sum over this rank's chunk of the occupied summation index, all-reduced (80 KB)
            ctx.singles_residual_partial(st["fd"], t1, t2, r1, rank, world, reuse_layouts=True)
            pending.append(pdist.allreduce_tensor_start(st["R1_t"], ctx, label="R1"))
            for work in pending:
                work.wait()
            mark("X_ac, singles residual, waits")
            if st["pairs"]:
                return self._pair_sharded_tail(st, r1)
            r2 = ctx.pool_get(t2.shape)
            ctx.residual_finish(st["fd"], t2, st["ETd"], st["ETx"], st["L"], r2, is_dcd=self.is_dcd, dressed=True,
                                t1=t1, QK=st["QK"], reuse_layouts=True)
        else:
            # general path (user amplitudes without the exchange symmetry): explicitly dressed blocks
            r2 = ctx.pool_get(t2.shape)
            ctx.singles_residual(st["fd"], t1, t2, r1)                                # :167
            ctx.dress_V(t1, LOOP_KEYS)                                                # :165
            if not dist_on:
                ctx.doubles_residual(st["fd"], t2, r2, is_dcd=self.is_dcd, dressed=True, sym_ladder=False)   # :171
            else:       # plain ladder rows on this rank (one all-gather), everything else replicated
                lo, hi = pdist.slab_rows(st["lad_rows"], rank, world)
                if hi > lo:
                    self._ladder_rows_plain(ctx, t2, st["lad"], lo, hi)
                ctx.doubles_residual(st["fd"], t2, r2, is_dcd=self.is_dcd, dressed=True, skip_ladder=True,
                                     sym_ladder=False, sym_rings=False)
                pdist.exchange_rows(st["lad_t"], rank, world, ctx)
                full = DeviceArray(ctx, st["lad"].ptr, (r2.size,), owned=False, keepalive=st["lad"])
                r2f = r2.reshape(r2.size)
                ctx.lincomb(r2f, [r2f, full], [1.0, 1.0])
        dt1, dt2 = ctx.pool_get(t1.shape), ctx.pool_get(t2.shape)
        ctx.cc_update(t1, dt1, r1, shift, self.delta)               # :176-179
        ctx.cc_update(t2, dt2, r2, shift, self.delta)
        if not st["sym"]:
            ctx.pool_put(r1)
        ctx.pool_put(r2)
        if st["first"] and st["amps"] is not None:
            np.copyto(st["amps"][0], t1.get())      # the reference updates the caller's arrays in place
            np.copyto(st["amps"][1], t2.get())
        st["first"] = False
        if self.is_diis:
            t1, t2 = self.mixer.mix([dt1, dt2], [t1, t2], release=ctx.pool_put)           # :181-183
        e = ctx.ccsd_energy(st["f"], t1, t2)                        # :189-192
        nt, nr = np.sqrt(ctx.dots([t2, dt2], [t2, dt2]))            # :196-197
        if not self.is_diis:
            ctx.pool_put(dt1)
            ctx.pool_put(dt2)
        st["t1"], st["t2"] = t1, t2
        return e[0], e[1], e[2], nt, nr

    @staticmethod
    def _compact(ctx, st):
        """A buffer for the compact tiles of this rank's pairs; zeroed when the rank owns no pair at all."""
        lo, hi = pdist.slab_rows(st["npp"], st["rank"], st["world"])
        arr = ctx.pool_get(st["cshape"])
        return arr.zero_() if hi <= lo else arr

    def _pair_sharded_tail(self, st, r1):
        """Rest of the loop body (ccsd.py:171-197) when every rank owns a chunk of the virtual pairs: R2, the update,
        the DIIS history and extrapolation exist only for the rank's pairs (compact tiles); the new T2 is all-gathered
        (0.8 GB at (50,200)) and unpacked into the replicated full array that the next residual reads."""
        ctx, t1, t2 = st["ctx"], st["t1"], st["t2"]
        rank, world, shift = st["rank"], st["world"], st["level_shift"]
        rc = self._compact(ctx, st)
        ctx.residual_finish_pairs(st["fd"], t2, st["ETd"], st["ETx"], st["L"], rc, rank, world, t1, st["QK"],
                                  is_dcd=self.is_dcd, dressed=True, Xvv=st["Xvv"])   # :171
        dt1, dtc, tc = ctx.pool_get(t1.shape), self._compact(ctx, st), st["Tc"]
        ctx.cc_update(t1, dt1, r1, shift, self.delta)                                 # :176-179
        ctx.cc_update_pairs(tc, dtc, rc, shift, self.delta, rank, world)
        ctx.pool_put(rc)
        st["first"] = False
        pdist.trace.mark("finish + assembly (pairs), update")
        if self.is_diis:
            t1, tc = self.mixer.mix([dt1, dtc], [t1, tc], release=ctx.pool_put, sharded=(1,),
                                    allreduce=pdist.allreduce_sum, mark=pdist.trace.mark)        # :181-183
        pdist.trace.mark("DIIS extrapolation")
        lo, hi = pdist.slab_rows(st["npp"], rank, world)
        if hi > lo:
            mine = DeviceArray(ctx, st["Tall"].ptr + 8 * lo * 2 * ctx.no * ctx.no, (hi - lo, 2, ctx.no * ctx.no),
                               owned=False, keepalive=st["Tall"])
            mine.copy_from(DeviceArray(ctx, tc.ptr, mine.shape, owned=False, keepalive=tc))
        # the energy and the norms (:189-197) come from the compact tiles (partial sums, one all-reduce of six doubles), so
        # nothing below needs the replicated array: the all-gather of the new T2 is only STARTED — after that small
        # all-reduce (RCCL runs the collectives of a communicator in order: behind the 0.8-GB transfer the host would wait
        # for it) — and the next iteration, or whoever reads st["t2"], completes it (_await_t2)
        e1, ed, ex, nt2, nr2, _ = pdist.allreduce_sum(ctx.energy_norms_pairs(st["f"], t1, tc, dtc, rank, world))
        st["t2_pending"] = pdist.exchange_rows_start(st["Tall_t"], rank, world, ctx, label="new T2")
        pdist.trace.mark("energy + norms (pairs)")
        if not self.is_diis:
            ctx.pool_put(dt1)
            ctx.pool_put(dtc)
        st["t1"], st["Tc"] = t1, tc
        return e1, ed, ex, np.sqrt(nt2), np.sqrt(nr2)

    @staticmethod
    def _await_t2(st):
        """Complete the exchange of the new amplitudes that the pair-sharded tail left in flight: wait for the all-gather
        of the compact tiles and unpack them into the replicated T2."""
        if "coll" in st:
            st["coll"].call("pymes_ccsd_sharded_await", st["ctx"].handle, C.c_void_p(st["t2"].ptr), C.byref(st["bufs"]))
        work = st.pop("t2_pending", None)
        if work is not None:
            work.wait()
            st["ctx"].pairs_unpack(st["Tall"], st["t2"], st["world"])

    @staticmethod
    def _ladder_rows_plain(ctx, t2, lad, lo, hi):
        """Rows [lo,hi) of R[(a,b),(i,j)] = V[(a,b),(c,d)] T[(c,d),(i,j)] (unsymmetric amplitudes only)."""
        nv, no = ctx.nv, ctx.no
        Vab = ctx.V_block("abcd", dressed=True).reshape(nv * nv, nv * nv)
        rows = DeviceArray(ctx, Vab.ptr + 8 * lo * nv * nv, (hi - lo, nv * nv), owned=False, keepalive=Vab)
        out = DeviceArray(ctx, lad.ptr + 8 * lo * no * no, (hi - lo, no * no), owned=False, keepalive=lad)
        ctx.contract("rk,kn->rn", rows, t2.reshape(nv * nv, no * no), out=out)

    def solve(self, t_fock_pq, t_V_pqrs, level_shift=0., amps=None, sp=0, **kwargs):
        """ccsd.py:47-224."""
        algo_name = "ccsd.solve"
        time_ccsd = time.time()
        max_iter = kwargs.get("max_iter", self.max_iter)
        delta_e = kwargs.get("delta_e", self.delta_e)
        ints, own = self._integrals(t_fock_pq, t_V_pqrs)
        ctx = ints.ctx
        st = None
        collector = ccd.quiet_collector().__enter__()      # no 40-ms generation-2 collection in the middle of an iteration
        try:
            print_logging_info(algo_name)
            print_logging_info("Using dcsd: ", self.is_dcd, level=1)
            print_logging_info("Solving doubles amplitude equation", level=1)
            print_logging_info("Using data type %s" % ints.dtype, level=1)
            print_logging_info("Using DIIS mixer: ", self.is_diis, level=1)
            print_logging_info("Iteration = 0", level=1)
            st = self.setup(t_fock_pq, ints, level_shift, amps)
            self.pair_sharded = st["pairs"]          # which tail the iterations use (world > 1 only)
            self.hooked = "coll" in st               # ... and whether they run as whole library steps with the collective table
            e_mp2 = st["e_mp2"]
            dE = np.abs(e_mp2)
            iteration = 0
            e_last = e_mp2
            e_ccsd = e_1b = e_dir = e_ex = 0.
            while np.abs(dE) > delta_e and iteration <= max_iter:
                iteration += 1
                # (single rank: a pass enqueues the residuals of the next one before it reads its energy back — not when
                # this pass is expected to be the last: |dE| shrinks geometrically, a factor of 30 per pass is generous)
                st["speculate"] = bool(np.abs(dE) > 30.0 * delta_e and iteration < max_iter + 1)
                e_1b, e_dir, e_ex, nt, nr = self.iterate(st)
                e_ccsd = e_1b + e_dir + e_ex
                dE = e_ccsd - e_last
                e_last = e_ccsd
                if iteration <= max_iter:
                    print_logging_info("Iteration = ", iteration, level=1)
                    print_logging_info("Correlation Energy = {:.14f}".format(e_ccsd), level=2)
                    print_logging_info("dE = {:.12e}".format(dE), level=2)
                    print_logging_info("L1 Norm of T2 = {:.14f}".format(nt), level=2)
                    print_logging_info("Norm Residual = {:.14f}".format(nr), level=2)
                else:
                    print_logging_info("A converged solution is not found!", level=1)
            print_logging_info("Fock contribution = {:.12f}".format(e_1b), level=1)
            print_logging_info("Direct contribution = {:.12f}".format(e_dir), level=1)
            print_logging_info("Exchange contribution = {:.12f}".format(e_ex), level=1)
            print_logging_info("CCSD correlation energy = {:.12f}".format(e_ccsd), level=1)
            print_logging_info("{:.3f} seconds spent on ccsd".format((time.time() - time_ccsd)), level=1)
            self._await_t2(st)
            self.collective_calls = st["coll"]._next - 1 if self.hooked else 0
            if kwargs.get("device_amplitudes"):
                # device-resident hand-over to the callers of the solution (EOM-CCSD / FEAST: get_T1_dressed_V on the same
                # DeviceIntegrals, EOM_CCSD.solve on the result): "t1" / "t2" are DeviceArrays of the integrals' context —
                # private copies, the solver's own buffers go back to the pool of the next solve
                if own:
                    raise ValueError("device_amplitudes=True needs DeviceIntegrals (the context of a host V_pqrs dies with the call)")
                self.t_T_ai = ctx.empty(st["t1"].shape).copy_from(st["t1"])
                self.t_T_abij = ctx.empty(st["t2"].shape).copy_from(st["t2"])
            else:
                self.t_T_ai = st["t1"].get()
                self.t_T_abij = st["t2"].get()
            if amps is not None and not self.is_diis and iteration > 0:
                # without DIIS the reference keeps updating the caller's arrays in place (ccsd.py:178-179)
                np.copyto(amps[0], self.t_T_ai)
                np.copyto(amps[1], self.t_T_abij)
            self.iterations = iteration
            return {"ccsd e": e_ccsd, "t1": self.t_T_ai, "t2": self.t_T_abij, "hole e": st["eps_i"],
                    "particle e": st["eps_a"], "dE": dE}
        finally:
            collector.__exit__()
            if own:
                ctx.close()      # (a DIIS history kept in this context is parked on the host on the way: Context.on_close)
            elif ctx.handle:
                ccd.destroy_graphs(ctx, st)

    # ------------------------------------------------------------------------------------
    # public helpers with the reference's host-array call forms (used by the EOM drivers)
    # ------------------------------------------------------------------------------------
    def _ctx_from_blocks(self, dict_t_V, nv):
        ctx = Context(self.no, nv, device=self.device)
        for name, blk in dict_t_V.items():
            if blk is not None:
                ctx.set_V_block(name, np.ascontiguousarray(blk, dtype=np.float64))
        return ctx

    def get_T1_dressed_fock(self, t_fock_pq, t_T_ai, dict_t_V):
        """ccsd.py:226-288.  ``dict_t_V`` may be a ``DeviceIntegrals`` (and ``t_T_ai`` a DeviceArray of its context): the
        blocks are read where they are; the n x n result comes back as a host array either way."""
        if isinstance(dict_t_V, DeviceIntegrals):
            ctx = dict_t_V.ctx
            t1 = t_T_ai if isinstance(t_T_ai, DeviceArray) else ctx.array(t_T_ai)
            fd = ctx.empty(t_fock_pq.shape)
            ctx.dress_fock(ctx.array(np.asarray(t_fock_pq, dtype=np.float64)), t1, fd)
            return fd.get()
        ctx = self._ctx_from_blocks(dict_t_V, t_T_ai.shape[0])
        try:
            fd = ctx.empty(t_fock_pq.shape)
            ctx.dress_fock(ctx.array(np.asarray(t_fock_pq, dtype=np.float64)), ctx.array(t_T_ai), fd)
            return fd.get()
        finally:
            ctx.close()

    def get_T1_dressed_V(self, t_T_ai, dict_t_V, dict_t_V_dressed=None):
        """ccsd.py:290-421; the optional third argument selects the blocks (:316-317).  ``dict_t_V`` may be a
        ``DeviceIntegrals`` (``t_T_ai`` a host array or a DeviceArray of its context): the blocks are dressed in HBM, next to
        the undressed ones, and a ``DressedDeviceIntegrals`` — the dictionary's device-resident stand-in — is returned."""
        if isinstance(dict_t_V, DeviceIntegrals):
            ctx = dict_t_V.ctx
            t1 = t_T_ai if isinstance(t_T_ai, DeviceArray) else ctx.array(t_T_ai)
            want = [k for k in (dict_t_V_dressed or DRESSED_KEYS) if k in DRESSED_KEYS]
            ctx.dress_V(t1, want)
            return DressedDeviceIntegrals(dict_t_V, want)
        if dict_t_V_dressed is None or len(dict_t_V_dressed) == 0:
            dict_t_V_dressed = {}.fromkeys(dict_t_V, None)
        ctx = self._ctx_from_blocks(dict_t_V, t_T_ai.shape[0])
        try:
            want = [k for k in dict_t_V_dressed if k in DRESSED_KEYS]
            ctx.dress_V(ctx.array(t_T_ai), want)
            for k in want:
                dict_t_V_dressed[k] = ctx.V_block(k, dressed=True).get()
            return dict_t_V_dressed
        finally:
            ctx.close()

    def get_singles_residual(self, t_fock_pq, t_T_ai, t_T_abij, dict_t_V):
        """ccsd.py:423-438 (t_fock_pq is the dressed Fock matrix)."""
        ctx = self._ctx_from_blocks(dict_t_V, t_T_ai.shape[0])
        try:
            r1 = ctx.empty(t_T_ai.shape)
            ctx.singles_residual(ctx.array(np.asarray(t_fock_pq, dtype=np.float64)), ctx.array(t_T_ai),
                                 ctx.array(t_T_abij), r1)
            return r1.get()
        finally:
            ctx.close()

    def get_doubles_residual(self, t_fock_pq, t_T_abij, dict_t_V_dressed):
        """ccsd.py:440-456."""
        return self.get_residual(t_fock_pq, t_T_abij, dict_t_V_dressed["klij"], dict_t_V_dressed["ijab"],
                                 dict_t_V_dressed["abij"], dict_t_V_dressed["iajb"], dict_t_V_dressed["iabj"],
                                 dict_t_V_dressed["abcd"])

    def get_energy(self, t_fock_ia, t_T_ai, t_T_abij, t_V_ijab):
        """ccsd.py:458-466: [one-body, direct, exchange]."""
        no, nv = t_fock_ia.shape
        ctx = Context(no, nv, device=self.device)
        try:
            ctx.set_V_block("ijab", np.ascontiguousarray(t_V_ijab, dtype=np.float64))
            f = np.zeros((no + nv, no + nv))
            f[:no, no:] = t_fock_ia
            return list(ctx.ccsd_energy(ctx.array(f), ctx.array(t_T_ai), ctx.array(t_T_abij)))
        finally:
            ctx.close()
