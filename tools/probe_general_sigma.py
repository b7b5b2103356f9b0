"""Timing of ONE sigma build for a trial vector WITHOUT exchange symmetry at (30,120) (what a FEAST operator application runs
twice); under rocprofv3 the last build's kernels are listed by tools/trace_last_build.py."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from pymes_amd.integral.device import DeviceIntegrals
from pymes_amd.model import synthetic
from pymes_amd.solver.eom_ccsd import _Sigma
no, nv = 30, 120
g = np.load(os.path.join("tests", "golden", "eom_sigma_30_120.npz"))
B, eps = synthetic.factors(no, nv, seed=0, scale=float(g["scale"]))
rng = np.random.default_rng(int(g["seed"]))
n = no + nv
fd = np.diag(eps) + 0.02 * rng.standard_normal((n, n))
t2h = rng.standard_normal((nv, nv, no, no)) * 0.02
t2h = 0.5 * (t2h + t2h.transpose(1, 0, 3, 2))
u1h = rng.standard_normal((nv, no)) * 0.3
u2h = rng.standard_normal((nv, nv, no, no)) * 0.05
ints = DeviceIntegrals.from_factors(no, B)
ctx = ints.ctx
sig = _Sigma(ctx, fd, ctx.array(t2h))
u1, u2 = ctx.array(u1h), ctx.array(u2h)
for i in range(30):
    r = sig.apply(u1, u2, u2_sym=False); del r
ctx.sync()
t0 = time.perf_counter()
for i in range(40):
    r = sig.apply(u1, u2, u2_sym=False); del r
ctx.sync(); print("general sigma, ms each", 1e3 * (time.perf_counter() - t0) / 40, flush=True)
ctx.sync(); time.sleep(0.05)
r = sig.apply(u1, u2, u2_sym=False); ctx.sync()
