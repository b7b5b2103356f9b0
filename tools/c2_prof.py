#!/usr/bin/env python3
"""C2 (20,80) CCSD iterations for profiling: python3 tools/c2_prof.py [no nv iters]"""
import sys, os, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymes_amd.integral.device import DeviceIntegrals
from pymes_amd.model import synthetic
from pymes_amd.solver.ccsd import CCSD
no, nv, iters = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (20, 80, 20)))
B, eps = synthetic.factors(no, nv, seed=0, scale=0.15)
ints = DeviceIntegrals.from_factors(no, B)
solver = CCSD(no)
with contextlib.redirect_stdout(io.StringIO()):
    st = solver.setup(np.diag(eps), ints)
    for _ in range(3):
        solver.iterate(st)
    ints.ctx.sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        solver.iterate(st)
    ints.ctx.sync()
dt = (time.perf_counter() - t0) / iters
print(f"({no},{nv}) {dt*1e3:.3f} ms/iteration")
