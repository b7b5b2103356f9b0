#!/usr/bin/env python3
"""GPU probe: the first context of a process runs the (20,80) iteration at 3.0-3.5 ms, later ones at 1.96 ms.  What makes
the difference?  python3 tools/probe_first_context.py <variant>"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
variant = sys.argv[1] if len(sys.argv) > 1 else "default"
from pymes_amd.device import Context
from pymes_amd.integral.device import DeviceIntegrals
from pymes_amd.model import synthetic
from pymes_amd.solver.ccsd import CCSD


def run(tag, reps=40):
    no, nv = 20, 80
    B, eps = synthetic.factors(no, nv, seed=0, scale=0.15)
    ints = DeviceIntegrals.from_factors(no, B)
    solver = CCSD(no)
    with contextlib.redirect_stdout(io.StringIO()):
        st = solver.setup(np.diag(eps), ints)
        for _ in range(8):
            solver.iterate(st)
        ints.ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            solver.iterate(st)
        ints.ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    print(f"[{variant}] {tag}: {dt*1e3:.3f} ms per iteration", flush=True)
    return ints


if variant == "dummyctx":
    c = Context(2, 3); c.zeros((10,)); c.sync(); c.close()
if variant == "prealloc":
    c = Context(2, 3); big = c.zeros((1 << 29,)); c.sync(); c.close()        # 4 GB touched, then released
if variant == "keep":
    a = run("first (kept alive)")
    b = run("second while the first lives")
    a.ctx.close(); b.ctx.close()
    sys.exit(0)
if variant == "windows":
    no, nv = 20, 80
    B, eps = synthetic.factors(no, nv, seed=0, scale=0.15)
    ints = DeviceIntegrals.from_factors(no, B)
    solver = CCSD(no)
    with contextlib.redirect_stdout(io.StringIO()):
        st = solver.setup(np.diag(eps), ints)
        for _ in range(8):
            solver.iterate(st)
        for w in range(6):
            ints.ctx.sync()
            t0 = time.perf_counter()
            for _ in range(40):
                solver.iterate(st)
            ints.ctx.sync()
            sys.stderr.write(f"[windows] iterations {9 + 40 * w}-{48 + 40 * w}: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms per iteration\n")
    sys.exit(0)
if variant == "each":
    no, nv = 20, 80
    B, eps = synthetic.factors(no, nv, seed=0, scale=0.15)
    ints = DeviceIntegrals.from_factors(no, B)
    solver = CCSD(no)
    ts = []
    with contextlib.redirect_stdout(io.StringIO()):
        st = solver.setup(np.diag(eps), ints)
        for _ in range(70):
            t0 = time.perf_counter()
            solver.iterate(st)
            ts.append((time.perf_counter() - t0) * 1e3)
    sys.stderr.write("[each] ms per iteration: " + " ".join(f"{t:.2f}" for t in ts) + "\n")
    sys.exit(0)
if variant == "long":
    a = run("first, iterations 9-48")
    with contextlib.redirect_stdout(io.StringIO()):
        pass
    a.ctx.close()
    run("second").ctx.close()
    sys.exit(0)
run("first").ctx.close()
run("second").ctx.close()
