import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from pymes_amd.integral.device import DeviceIntegrals
from pymes_amd.model import synthetic
from pymes_amd.solver.eom_ccsd import _Sigma
no, nv = 30, 120
B, eps = synthetic.factors(no, nv, seed=0)
ints = DeviceIntegrals.from_factors(no, B)
ctx = ints.ctx
t2 = ctx.empty((nv, nv, no, no)); ctx.mp2(t2, 0.0)
sig = _Sigma(ctx, np.diag(eps), t2)
rng = np.random.default_rng(0)
u1 = ctx.array(rng.standard_normal((nv, no))); u2 = ctx.array(rng.standard_normal((nv, nv, no, no)))
sig.apply(u1, u2); ctx.sync()
ctx.prof_enable(True); ctx.prof_reset(); ctx.stats(reset=True)
t0 = time.perf_counter(); sig.apply(u1, u2); ctx.sync(); dt = time.perf_counter() - t0
print("sigma s", dt, ctx.prof_query(), ctx.stats())
