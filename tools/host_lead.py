#!/usr/bin/env python3
"""Is the interpreter ahead of the device inside a (20,80) CCSD iteration?  A busy wait of D microseconds is put in front of the
update of every pass (the first call after the residual graph has been launched): an iteration time that does not move with D
says the host runs that far ahead and the idle time in front of the update in the kernel trace is the device's own."""
import contextlib, io, sys, time
import numpy as np
sys.path.insert(0, ".")
from pymes_amd.integral.device import DeviceIntegrals
from pymes_amd.model import synthetic
from pymes_amd.solver.ccsd import CCSD

no, nv = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20, 80)
B, eps = synthetic.factors(no, nv, seed=0, scale=0.15)
ints = DeviceIntegrals.from_factors(no, B, device=0)
solver = CCSD(no, device=0)
with contextlib.redirect_stdout(io.StringIO()):
    st = solver.setup(np.diag(eps), ints)
    for _ in range(6):
        solver.iterate(st)
ctx = st["ctx"]
orig = ctx.cc_update_to
delay = [0.0]
def delayed(*a, **k):
    if delay[0] > 0 and a[0].shape == st["t1"].shape:
        t = time.perf_counter() + delay[0]
        while time.perf_counter() < t:
            pass
    return orig(*a, **k)
ctx.cc_update_to = delayed
for d in (0, 50, 100, 200, 400, 800, 0):
    delay[0] = d * 1e-6
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(10):
            solver.iterate(st)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(100):
            solver.iterate(st)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 100
    print(f"busy wait {d:4d} us in front of the update: {1e3 * dt:.4f} ms per iteration")
