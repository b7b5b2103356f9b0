#!/usr/bin/env python3
"""GPU probe: one stacked EOM-CCSD sigma build of k vectors at (30,120), exchange-symmetric form against the general form
(FEAST's complex Krylov vectors: real and imaginary part, no symmetry)."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymes_amd.integral.device import DeviceIntegrals
from pymes_amd.model import synthetic
from pymes_amd.solver.eom_ccsd import _Sigma
no, nv = int(os.environ.get("NO", 30)), int(os.environ.get("NV", 120))
B, eps = synthetic.factors(no, nv, seed=0)
ints = DeviceIntegrals.from_factors(no, B)
ctx = ints.ctx
ctx.set_orbital_energies(eps[:no], eps[no:])
t2 = ctx.empty((nv, nv, no, no)); ctx.mp2(t2, 0.0)
sig = _Sigma(ctx, np.diag(eps), t2)
rng = np.random.default_rng(0)
gc.disable()
for k, sym in ((2, True), (2, False), (1, False), (3, True), (4, False)):
    u1s = [ctx.array(rng.standard_normal((nv, no))) for _ in range(k)]
    u2s = []
    for z in range(k):
        h = rng.standard_normal((nv, nv, no, no))
        u2s.append(ctx.array(h + h.transpose(1, 0, 3, 2) if sym else h))
    for _ in range(2):
        out = sig.apply_many(u1s, u2s, [sym] * k); del out
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(3):
        out = sig.apply_many(u1s, u2s, [sym] * k); del out
    ctx.sync(); dt = (time.perf_counter() - t0) / 3
    print(f"k = {k} {'symmetric' if sym else 'general  '}: {dt*1e3:7.2f} ms per build, {dt*1e3/k:6.2f} ms per vector", flush=True)
    for a in u1s + u2s: a.free()
ctx.close()
