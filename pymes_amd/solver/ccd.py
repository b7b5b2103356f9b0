"""CCD / DCD driver and the T2 residual (pymes/solver/ccd.py) on the MI355X engine.

Drop-in for ``pymes.solver.ccd.CCD``: same constructor, attributes, ``solve`` signature
and return dictionary.  All tensors live in HBM between iterations; per iteration the
host sees the energy, two norms and the DIIS overlaps."""
import os
import time

import numpy as np

from pymes_amd.device import Context, DeviceArray, PymesError
from pymes_amd.integral.device import DeviceIntegrals
from pymes_amd.log import print_logging_info
from pymes_amd.mixer import diis


class quiet_collector:
    """``with quiet_collector():`` — the interpreter's cyclic garbage collector stays off inside.  A full (generation-2)
    collection of a process that has numpy / torch loaded takes 35-40 ms — measured: twenty (20,80) iterations of 2 ms
    each, one of them 38 ms — and it strikes wherever the allocation counters happen to overflow, typically inside the
    mixer.  The iteration loops allocate no reference cycles worth collecting; reference counting still frees everything
    else at once.  The previous state is restored on exit."""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.disable()
        return self

    def __exit__(self, *exc):
        if self.was:
            import gc
            gc.enable()
        return False


def destroy_graphs(ctx, st):
    """Release the launch graphs a solve recorded (a caller-owned context would otherwise keep them until it closes).
    A pipelined pass may have left a replay of one of them in flight (the residual build of a pass that never came, or
    the one an exception cut short) and ``device_amplitudes=True`` hands results over without a host copy: the stream is
    drained first — a graph exec owns the kernel arguments of its queued launches."""
    graphs = (st or {}).pop("graphs", {})
    if st is not None:
        st.pop("residuals_in_flight", None)
    if graphs:
        try:
            ctx.sync()
        except PymesError:
            pass
    for g in graphs.values():
        try:
            ctx.graph_destroy(g)
        except PymesError:
            pass
    if st is not None:
        st["graph"] = None
    try:
        ctx.ccsd_release()           # the staging buffers of pymes_ccsd_residuals go back to the engine's scratch pool
    except PymesError:
        pass


def run_replayable(ctx, st, body, key="residual"):
    """Run ``body`` (a callable that only ENQUEUES kernels on fixed buffers: the residual part of a solver's loop body)
    — eagerly the first time (lazy set-up work: cached permutations, packed integrals), then recorded once as a launch
    graph and replayed.  ``key`` names the variant of the body (a solver may alternate between several, each with its own
    graph).  Per-GEMM event timing (``prof_enable``) needs the eager form.  Measured: replay does not
    shorten an iteration, the device executes the same latency-bound kernels either way — kept because it costs nothing
    and frees the host."""
    graphs, passes = st.setdefault("graphs", {}), st.setdefault("eager_passes_of", {})
    g = graphs.get(key)
    if g is not None and not ctx.profiling:
        ctx.graph_launch(g)
        return
    if st.get("graph_ok") and passes.get(key, 0) >= 1 and g is None and not ctx.profiling:
        try:
            ctx.graph_begin()
            body()
            graphs[key] = ctx.graph_end()
        except PymesError:
            ctx.graph_abort()
            st["graph_ok"] = False       # something in the body cannot be recorded: stay eager
        except BaseException:            # anything else (KeyboardInterrupt included) must not leave the stream recording
            ctx.graph_abort()
            st["graph_ok"] = False
            raise
        else:
            st["graph"] = graphs[key]
            ctx.graph_launch(graphs[key])
            return
    body()
    passes[key] = passes.get(key, 0) + 1
    st["eager_passes"] = st.get("eager_passes", 0) + 1


class CCD:
    def __init__(self, no, delta_e=1.e-8, is_dcd=False, is_diis=True, is_dr_ccd=False, is_bruekner=False,
                 device=0):
        if is_dr_ccd or is_bruekner:
            raise NotImplementedError("dr-CCD and Brueckner energies (ccd.py:95-121) are outside the HIP hot path")
        self.is_dcd = is_dcd
        self.is_diis = is_diis
        self.is_dr_ccd = is_dr_ccd
        self.is_bruekner = is_bruekner
        self.no = no
        self.delta_e = delta_e
        self.max_iter = 50
        self.device = device
        if self.is_diis:
            self.mixer = diis.DIIS(dim_space=6)

    # ---- device plumbing shared with CCSD ------------------------------------------------
    def _integrals(self, t_fock_pq, t_V_pqrs):
        if isinstance(t_V_pqrs, DeviceIntegrals):
            return t_V_pqrs, False
        return DeviceIntegrals.from_V_pqrs(self.no, t_V_pqrs, device=self.device), True

    def solve(self, t_fock_pq, t_V_pqrs, level_shift=0., sp=0, amps=None, **kwargs):
        """ccd.py:24-162."""
        algo_name = "ccd.solve"
        time_ccd = time.time()
        no = self.no
        nv = t_fock_pq.shape[0] - no
        t_epsilon_i = t_fock_pq.diagonal()[:no]
        t_epsilon_a = t_fock_pq.diagonal()[no:]
        max_iter = kwargs.get("max_iter", self.max_iter)
        delta_e = kwargs.get("delta_e", self.delta_e)
        delta = 1.0

        ints, own = self._integrals(t_fock_pq, t_V_pqrs)
        ctx = ints.ctx
        st = None
        collector = quiet_collector().__enter__()
        try:
            ctx.trim()       # recycled temporaries of earlier work on this context: the set-up below allocates engine-side
            ctx.set_orbital_energies(t_epsilon_i, t_epsilon_a)
            f_dev = ctx.array(np.asarray(t_fock_pq, dtype=np.float64))
            print_logging_info(algo_name)
            print_logging_info("Using DCD: ", self.is_dcd, level=1)
            print_logging_info("Using dr-CCD: ", self.is_dr_ccd, level=1)
            print_logging_info("Solving doubles amplitude equation", level=1)
            print_logging_info("Using data type %s" % ints.dtype, level=1)
            print_logging_info("Using DIIS mixer: ", self.is_diis, level=1)
            print_logging_info("Using Bruekner quasi-particle energy: ", self.is_bruekner, level=1)
            print_logging_info("Iteration = 0", level=1)
            t2 = ctx.pool_get((nv, nv, no, no))
            e_dir, e_exc = ctx.mp2(t2, level_shift)
            e_mp2 = e_dir + e_exc
            print("MP2 energy = ", e_mp2)
            if amps is not None:
                t2.set(np.asarray(amps))
            # the symmetry-reduced residual needs T_abij = T_baji (true for MP2, checked for user input) and
            # V_pqrs = V_qpsr (checked once; the reference assumes neither)
            sym = ctx.V_exchange_symmetric() and (amps is None or ctx.exchange_symmetric(t2))
            dE = np.abs(e_mp2)
            iteration = 0
            e_last = e_mp2
            e_ccd = e_dir_ccd = e_ex_ccd = 0.
            first = True
            shard = self._shard_setup(ctx, t2, sym and amps is None)     # one process per GPU (None for a single rank)
            self.pair_sharded = shard is not None
            self.hooked = shard is not None and "coll" in shard      # ... as whole library steps with the collective table
            if shard is None:       # fixed buffers: the residual is replayed as a launch graph (run_replayable)
                r2 = ctx.pool_get(t2.shape)
                st = {"graph": None, "eager_passes": 0,
                      "graph_ok": ctx.graphs_supported() and not os.environ.get("PYMES_NO_GRAPH")}
                dt_fixed = None if self.is_diis else ctx.pool_get(t2.shape)
            while np.abs(dE) > delta_e and iteration <= max_iter:
                iteration += 1
                if shard is not None:
                    e_dir_ccd, e_ex_ccd, nt, nr = self._sharded_iteration(ctx, shard, f_dev, t2, level_shift, delta)  # :132
                else:
                    run_replayable(ctx, st, lambda: ctx.doubles_residual(f_dev, t2, r2, is_dcd=self.is_dcd,
                                                                         sym_ladder=sym))     # ccd.py:100-102
                    if self.is_diis:
                        t2n, dt2 = ctx.pool_get(t2.shape), ctx.pool_get(t2.shape)
                        ctx.cc_update_to(t2n, dt2, t2, r2, level_shift, delta)        # :123-124
                    else:
                        t2n, dt2 = t2, dt_fixed
                        ctx.cc_update(t2, dt2, r2, level_shift, delta)
                    if first and amps is not None:
                        np.copyto(amps, t2n.get())    # the reference updates the caller's array in place (:124)
                    first = False
                    if self.is_diis:
                        self.mixer.mix([dt2], [t2n], release=ctx.pool_put, out=[t2], defer_log=True,   # :126-127
                                       on_device=bool(os.environ.get("PYMES_DEVICE_DIIS")),
                           native=not os.environ.get("PYMES_NUMPY_DIIS"))
                    _, e_dir_ccd, e_ex_ccd, nt2, nr2, _ = ctx.energy_norms(None, None, t2, dt2)   # :132 + norms, one pass
                    if self.is_diis:
                        self.mixer.log_last()
                    nt, nr = np.sqrt(nt2), np.sqrt(nr2)
                e_ccd = e_dir_ccd + e_ex_ccd
                dE = e_ccd - e_last
                e_last = e_ccd
                if iteration <= max_iter:
                    print_logging_info("Iteration = ", iteration, level=1)
                    print_logging_info("Correlation Energy = {:.12f}".format(e_ccd), level=2)
                    print_logging_info("dE = {:.12e}".format(dE), level=2)
                    print_logging_info("L1 Norm of T2 = {:.12f}".format(nt), level=2)
                    print_logging_info("Norm Residual = {:.12f}".format(nr), level=2)
                else:
                    print_logging_info("A converged solution is not found!", level=1)
            print_logging_info("Direct contribution = {:.12f}".format(e_dir_ccd), level=1)
            print_logging_info("Exchange contribution = {:.12f}".format(e_ex_ccd), level=1)
            print_logging_info("CCD correlation energy = {:.12f}".format(e_ccd), level=1)
            print_logging_info("{:.3f} seconds spent on CCD".format((time.time() - time_ccd)), level=1)
            if amps is not None and not self.is_diis and iteration > 0 and shard is None:
                np.copyto(amps, t2.get())         # without DIIS the reference updates the caller's array every iteration
            result = {"ccd e": e_ccd, "t2 amp": t2.get(), "hole e": t_epsilon_i, "particle e": t_epsilon_a, "dE": dE}
            self.iterations = iteration
            return result
        finally:
            collector.__exit__()
            if own:
                ctx.close()      # (a DIIS history kept in this context is parked on the host on the way: Context.on_close)
            elif ctx.handle:
                destroy_graphs(ctx, st)

    # ---- one process per GPU (torch.distributed): the same sharding as CCSD.iterate, without T1 ----------------
    def _shard_setup(self, ctx, t2, allowed):
        from pymes_amd import dist as pdist
        from pymes_amd.device import DeviceArray
        rank, world, _ = pdist.world()
        if not pdist.sharded() or not allowed or not ctx.pairs_supported():
            return None
        import torch
        pdist.bind_stream(ctx)          # engine kernels and RCCL collectives ordered on one stream: no host fences
        no, nv = ctx.no, ctx.nv
        dev = torch.device("cuda", ctx.device) if ctx.lib.backend.startswith("hip") else torch.device("cpu")

        def shared(rows, cols):
            t = torch.zeros((pdist.padded_rows(rows, world), cols), dtype=torch.float64, device=dev)
            return t, DeviceArray(ctx, t.data_ptr(), tuple(t.shape), owned=False, keepalive=t)
        npp = nv * (nv + 1) // 2
        lo, hi = pdist.slab_rows(npp, rank, world)
        sh = {"rank": rank, "world": world, "npp": npp, "lo": lo, "hi": hi, "cshape": (max(hi - lo, 1), 2, no * no)}
        sh["ETd_t"], sh["ETd"] = shared(no * nv, no * nv)
        sh["ETx_t"], sh["ETx"] = shared(no * nv, no * nv)
        if os.environ.get("PYMES_PY_SEQUENCED"):
            sh["L"] = ctx.zeros((pdist.padded_rows(npp, world), no * no))
        sh["Tall_t"], sh["Tall"] = shared(npp, 2 * no * no)
        sh["Tc"] = self._compact(ctx, sh)
        ctx.pairs_pack(t2, sh["Tc"], rank, world)
        if not os.environ.get("PYMES_PY_SEQUENCED"):
            # the loop body as whole library steps with the collective table (pymes_ccd_sharded_residuals, then the finish /
            # energy / await steps of CCSD with f = t1 = NULL): the sequence a torch-free host runs (include/pymes_amd.h;
            # tests/test_collective_hook.py drives it with plain ctypes callbacks).  PYMES_PY_SEQUENCED=1 keeps the
            # Python-sequenced form below.
            from pymes_amd import _lib
            sh["L_t"], sh["L"] = shared(npp, no * no)
            sh["S_t"] = torch.zeros((8,), dtype=torch.float64, device=dev)
            names = ("ETd", "ETx", "L", "Tall", "S")
            sh["coll"] = pdist.Collectives(ctx, {k: sh[k + "_t"] for k in names}, rank, world)
            ptr = {k: sh[k + "_t"].data_ptr() for k in names}
            sh["bufs"] = _lib.ShardBuffers(ptr["ETd"], ptr["ETx"], ptr["L"], None, ptr["Tall"], None, None, None, None, ptr["S"])
            sh["flags"] = _lib.PYMES_DCD if self.is_dcd else 0
            if os.environ.get("PYMES_OWNER_TILES"):
                sh["coll"].enable_owner_tiles(dev)
                sh["flags"] |= _lib.PYMES_OWNER_TILES
        return sh

    @staticmethod
    def _compact(ctx, sh):
        arr = ctx.pool_get(sh["cshape"])
        return arr.zero_() if sh["hi"] <= sh["lo"] else arr

    def _sharded_iteration(self, ctx, sh, f_dev, t2, level_shift, delta):
        """ccd.py:100-127 with the residual slab / pair-sharded tail of include/pymes_amd.h; t2 (full, replicated) is
        refreshed in place from the all-gathered compact amplitudes."""
        from pymes_amd import dist as pdist
        from pymes_amd.device import DeviceArray
        rank, world = sh["rank"], sh["world"]
        if "coll" in sh:
            return self._hooked_iteration(ctx, sh, f_dev, t2, level_shift, delta)
        # ring products first: the all-gathers of their rows fly while the ladders (rows that stay on the rank) are computed
        pdist.trace.mark("begin")
        ctx.residual_slab(f_dev, t2, sh["ETd"], sh["ETx"], sh["L"], rank, world, is_dcd=self.is_dcd, part="rings")
        pdist.trace.mark("ring products")
        if os.environ.get("PYMES_OWNER_TILES"):      # all-to-all of the tiles each pair owner reads (dist.py)
            pending = [pdist.exchange_pair_tiles_start([sh["ETd_t"], sh["ETx_t"]], ctx.no, ctx.nv, rank, world, ctx,
                                                       label="ETd+ETx owner tiles")]
        else:
            pending = [pdist.exchange_rows_start(sh[k], rank, world, ctx, label=k[:3]) for k in ("ETd_t", "ETx_t")]
        ctx.residual_slab(f_dev, t2, sh["ETd"], sh["ETx"], sh["L"], rank, world, is_dcd=self.is_dcd, part="ladders")
        for work in pending:
            work.wait()
        pdist.trace.mark("ladders, waits")
        rc, dtc, tc = self._compact(ctx, sh), self._compact(ctx, sh), sh["Tc"]
        ctx.residual_finish_pairs(f_dev, t2, sh["ETd"], sh["ETx"], sh["L"], rc, rank, world, is_dcd=self.is_dcd)
        ctx.cc_update_pairs(tc, dtc, rc, level_shift, delta, rank, world)             # :123-124
        ctx.pool_put(rc)
        if self.is_diis:
            tc = self.mixer.mix([dtc], [tc], release=ctx.pool_put, sharded=(0,), allreduce=pdist.allreduce_sum)[0]
        if sh["hi"] > sh["lo"]:
            n = sh["hi"] - sh["lo"]
            mine = DeviceArray(ctx, sh["Tall"].ptr + 8 * sh["lo"] * 2 * ctx.no * ctx.no, (n, 2, ctx.no * ctx.no),
                               owned=False, keepalive=sh["Tall"])
            mine.copy_from(DeviceArray(ctx, tc.ptr, mine.shape, owned=False, keepalive=tc))
        # energy (:132) and norms from the compact tiles of this rank's pairs: partial sums, one all-reduce of six doubles
        # (issued before the big transfer: collectives of one communicator run in order)
        _, e_dir, e_ex, nt2, nr2, _ = pdist.allreduce_sum(ctx.energy_norms_pairs(None, None, tc, dtc, rank, world))
        pending = pdist.exchange_rows_start(sh["Tall_t"], rank, world, ctx, label="new T2")
        pending.wait()
        ctx.pairs_unpack(sh["Tall"], t2, world)
        pdist.trace.mark("finish, update, DIIS, energy, T2 exchange")
        if not self.is_diis:
            ctx.pool_put(dtc)
        sh["Tc"] = tc
        return e_dir, e_ex, np.sqrt(nt2), np.sqrt(nr2)

    def _hooked_iteration(self, ctx, sh, f_dev, t2, level_shift, delta):
        """ccd.py:100-132 for one rank of many as whole library steps (include/pymes_amd.h): residuals of the rank's pairs
        (collectives called back for), update, mixer, the finish step (energies all-reduced on the device, the new compact T2
        handed to its all-gather), the energy read-back, and the replicated T2 completed for the next pass."""
        import ctypes as C
        from pymes_amd import dist as pdist
        rank, world, coll = sh["rank"], sh["world"], sh["coll"]
        rc, dtc, tc = self._compact(ctx, sh), self._compact(ctx, sh), sh["Tc"]
        coll.call("pymes_ccd_sharded_residuals", ctx.handle, C.c_void_p(f_dev.ptr), C.c_void_p(t2.ptr), C.byref(sh["bufs"]),
                  sh["flags"], C.c_void_p(rc.ptr))
        ctx.cc_update_pairs(tc, dtc, rc, level_shift, delta, rank, world)             # :123-124
        ctx.pool_put(rc)
        if self.is_diis:
            tc = self.mixer.mix([dtc], [tc], release=ctx.pool_put, sharded=(0,), allreduce=pdist.allreduce_sum)[0]
        slot = C.c_int()
        coll.call("pymes_ccsd_sharded_finish", ctx.handle, None, None, C.c_void_p(tc.ptr), C.c_void_p(dtc.ptr), C.byref(sh["bufs"]),
                  C.byref(slot))
        en = (C.c_double * 6)()
        ctx.lib.call("pymes_ccsd_sharded_energy", ctx.handle, slot.value, en)
        coll.call("pymes_ccsd_sharded_await", ctx.handle, C.c_void_p(t2.ptr), C.byref(sh["bufs"]))
        pdist.trace.mark("finish, update, DIIS, energy, T2 exchange")
        if not self.is_diis:
            ctx.pool_put(dtc)
        sh["Tc"] = tc
        return en[1], en[2], np.sqrt(en[3]), np.sqrt(en[4])

    # The reference's mixer is never reset: a second solve() on the same instance starts from the history of the first
    # (diis.py:16-112 keeps its lists, ccsd.py:42 creates the mixer once).  Device vectors cannot outlive their context:
    # the mixer registers with every context it stores vectors in and parks them on the host when that context closes —
    # the solver's own or a caller-owned ``DeviceIntegrals`` — and adopts them into the next solve's context at its
    # first call (pymes_amd/mixer/diis.py).

    def get_residual(self, t_fock_pq, t_T_abij, t_V_klij, t_V_ijab, t_V_abij, t_V_iajb, t_V_iabj, t_V_abcd):
        """ccd.py:164-254 with host arrays in and a host array out (the reference's call form)."""
        no = self.no
        nv = t_fock_pq.shape[0] - no
        ctx = Context(no, nv, device=self.device)
        try:
            for name, blk in (("klij", t_V_klij), ("ijab", t_V_ijab), ("abij", t_V_abij), ("iajb", t_V_iajb),
                              ("iabj", t_V_iabj), ("abcd", t_V_abcd)):
                ctx.set_V_block(name, np.ascontiguousarray(blk, dtype=np.float64))
            r2 = ctx.empty((nv, nv, no, no))
            ctx.doubles_residual(ctx.array(np.asarray(t_fock_pq, dtype=np.float64)),
                                 ctx.array(np.asarray(t_T_abij, dtype=np.float64)), r2, is_dcd=self.is_dcd)
            return r2.get()
        finally:
            ctx.close()

    def get_energy(self, t_T_abij, t_V_ijab):
        """ccd.py:256-262."""
        no, nv = t_V_ijab.shape[0], t_V_ijab.shape[2]
        ctx = Context(no, nv, device=self.device)
        try:
            ctx.set_V_block("ijab", np.ascontiguousarray(t_V_ijab, dtype=np.float64))
            return ctx.ccd_energy(ctx.array(np.asarray(t_T_abij, dtype=np.float64)))
        finally:
            ctx.close()
