import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
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
u2h = 0.5 * (u2h + u2h.transpose(1, 0, 3, 2))
ints = DeviceIntegrals.from_factors(no, B)
ctx = ints.ctx
sig = _Sigma(ctx, fd, ctx.array(t2h))
u1, u2 = ctx.array(u1h), ctx.array(u2h)
s1, s2d = sig.apply(u1, u2, u2_sym=True)
s1h, s2h = s1.get(), s2d.get()
del s1h, s2h, s1, s2d
for i in range(8):
    ctx.sync(); t0 = time.perf_counter(); r = sig.apply(u1, u2, u2_sym=True); t1 = time.perf_counter(); ctx.sync(); t2_ = time.perf_counter()
    print(i, "enqueue ms", 1e3 * (t1 - t0), "total ms", 1e3 * (t2_ - t0), flush=True)
    del r
import gc
gc.collect(); gc.disable()
ts = []
for i in range(60):
    t0 = time.perf_counter(); r = sig.apply(u1, u2, u2_sym=True); del r; ts.append(time.perf_counter() - t0)
ctx.sync()
print("enqueue-only times ms, max:", 1e3 * max(ts), "mean:", 1e3 * sum(ts) / len(ts), "argmax", ts.index(max(ts)))
print([round(1e3 * t, 2) for t in ts])
t0 = time.perf_counter(); ctx.mem_info(); print("mem_info ms", 1e3 * (time.perf_counter() - t0))
t0 = time.perf_counter()
for i in range(60):
    r = sig.apply(u1, u2, u2_sym=True); del r
ctx.sync(); print("60 builds, ms each", 1e3 * (time.perf_counter() - t0) / 60)
