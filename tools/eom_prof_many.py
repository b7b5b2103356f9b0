#!/usr/bin/env python3
"""Per-GEMM log of one multi-vector EOM-CCSD sigma build (k = 4) at (30,120):
PYMES_GEMM_LOG=<file> python3 tools/eom_prof_many.py"""
import gc
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymes_amd.integral.device import DeviceIntegrals
from pymes_amd.model import synthetic
from pymes_amd.solver.eom_ccsd import _Sigma
no, nv, k = int(os.environ.get("NO", 30)), int(os.environ.get("NV", 120)), int(os.environ.get("K", 4))
B, eps = synthetic.factors(no, nv, seed=0)
ints = DeviceIntegrals.from_factors(no, B)
ctx = ints.ctx
ctx.set_orbital_energies(eps[:no], eps[no:])
t2 = ctx.empty((nv, nv, no, no)); ctx.mp2(t2, 0.0)
sig = _Sigma(ctx, np.diag(eps), t2)
rng = np.random.default_rng(0)
u1s, u2s = [], []
for z in range(k):
    u1s.append(ctx.array(rng.standard_normal((nv, no))))
    h = rng.standard_normal((nv, nv, no, no))
    u2s.append(ctx.array(h + h.transpose(1, 0, 3, 2)))
syms = [True] * k
gc.disable()
reps = int(os.environ.get("REPS", 3))
for _ in range(max(2, reps // 2)):
    out = sig.apply_many(u1s, u2s, syms)
    del out
ctx.sync()
t0 = time.perf_counter()
for _ in range(reps):
    out = sig.apply_many(u1s, u2s, syms)
    del out
ctx.sync()
print("ms per vector", 1e3 * (time.perf_counter() - t0) / reps / k)
ctx.prof_enable(True); ctx.prof_reset(); ctx.stats(reset=True)
t0 = time.perf_counter(); out = sig.apply_many(u1s, u2s, syms); ctx.sync(); dt = time.perf_counter() - t0
print("build s (events on)", dt, ctx.prof_query(), ctx.stats())
