#!/usr/bin/env python3
"""Per-rank compute time of one CCSD iteration at (50,200) for a given world size, measured on ONE GPU with the
collectives stubbed out (buffers of the other ranks stay zero: timings are valid, energies are not).

    python3 tools/rank_time.py --world 8 [--rank 0] [--nocc 50 --nvirt 200]
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--nocc", type=int, default=50)
    ap.add_argument("--nvirt", type=int, default=200)
    args = ap.parse_args()
    import torch
    from pymes_amd import dist as pdist
    from pymes_amd.integral.device import DeviceIntegrals
    from pymes_amd.model import synthetic
    from pymes_amd.solver.ccsd import CCSD

    class Done:
        def wait(self):
            return True
    pdist.world = lambda: (args.rank, args.world, 0)
    pdist.sharded = lambda: True
    pdist.bind_stream = lambda ctx: None
    pdist.exchange_rows = lambda full, rank, world, ctx=None, **kw: full
    pdist.exchange_rows_start = lambda full, rank, world, ctx=None, **kw: Done()
    pdist.allreduce_tensor_start = lambda t, ctx=None, **kw: Done()
    pdist.allreduce_sum = lambda v: np.asarray(v, dtype=np.float64)

    no, nv = args.nocc, args.nvirt
    B, eps = synthetic.factors(no, nv, seed=0)
    ints = DeviceIntegrals.from_factors(no, B, device=0, stream=torch.cuda.current_stream().cuda_stream)
    ctx = ints.ctx
    solver = CCSD(no)
    with contextlib.redirect_stdout(io.StringIO()):
        st = solver.setup(np.diag(eps), ints)
    assert st["pairs"], "pair-sharded tail not active"
    phases = {}
    calls = {name: getattr(ctx, name) for name in ("dress_fock", "dress_fock_partial", "dress_fock_finish", "xvv_partial",
                                                   "dress_V", "slab_prepare", "residual_slab", "singles_residual", "singles_residual_partial",
                                                   "residual_finish_pairs", "cc_update", "cc_update_pairs", "pairs_unpack",
                                                   "ccsd_energy", "energy_norms_pairs", "dots", "lincomb")}

    def timed(name):
        fn = calls[name]

        def wrapper(*a, **k):
            ctx.sync()
            t0 = time.perf_counter()
            out = fn(*a, **k)
            ctx.sync()
            phases[name] = phases.get(name, 0.0) + time.perf_counter() - t0
            return out
        return wrapper
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(2):
            solver.iterate(st)
        for name in calls:
            setattr(ctx, name, timed(name))
        ctx.sync()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            solver.iterate(st)
        ctx.sync()
    total = (time.perf_counter() - t0) / reps
    if os.environ.get("PYMES_GEMM_LOG"):              # one more iteration with per-GEMM events, written to that file
        for name, fn in calls.items():
            setattr(ctx, name, fn)
        ctx.prof_enable(True)
        ctx.prof_reset()
        with contextlib.redirect_stdout(io.StringIO()):
            solver.iterate(st)
        ctx.prof_query()
    out = {"world": args.world, "rank": args.rank, "no": no, "nv": nv, "iteration_compute_ms": 1e3 * total,
           "phases_ms": {k: 1e3 * v / reps for k, v in sorted(phases.items(), key=lambda kv: -kv[1])}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
