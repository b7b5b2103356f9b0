#!/usr/bin/env python3
"""Per-GEMM log of one EOM-CCSD sigma build at (30,120): PYMES_GEMM_LOG=<file> python3 tools/eom_prof.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymes_amd.integral.device import DeviceIntegrals
from pymes_amd.model import synthetic
from pymes_amd.solver.eom_ccsd import _Sigma
no, nv = 30, 120
B, eps = synthetic.factors(no, nv, seed=0)
ints = DeviceIntegrals.from_factors(no, B)
ctx = ints.ctx
ctx.set_orbital_energies(eps[:no], eps[no:])
t2 = ctx.empty((nv, nv, no, no)); ctx.mp2(t2, 0.0)
sig = _Sigma(ctx, np.diag(eps), t2)
rng = np.random.default_rng(0)
u1 = ctx.array(rng.standard_normal((nv, no)))
u2h = rng.standard_normal((nv, nv, no, no))
u2 = ctx.array(u2h + u2h.transpose(1, 0, 3, 2))
flag = sig.exchange_symmetric(u2)
for _ in range(2):
    sig.apply(u1, u2, u2_sym=flag)
ctx.sync()
ctx.prof_enable(True); ctx.prof_reset(); ctx.stats(reset=True)
t0 = time.perf_counter(); sig.apply(u1, u2, u2_sym=flag); ctx.sync(); dt = time.perf_counter() - t0
print("sigma s", dt, ctx.prof_query(), ctx.stats())
ctx.prof_enable(False)
for _ in range(3):
    ctx.sync(); t0 = time.perf_counter(); sig.apply(u1, u2, u2_sym=flag); ctx.sync(); print("plain apply s", time.perf_counter() - t0)
