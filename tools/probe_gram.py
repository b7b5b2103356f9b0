#!/usr/bin/env python3
"""GPU probe: bandwidth of the subspace kernels (pymes_gram / pymes_lincomb_multi / pymes_dots_var) on vectors of the (30,120)
Davidson size (13 M doubles)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymes_amd.device import Context

n = 3600 + 32 - 3600 % 32 + 120 * 120 * 30 * 30
ctx = Context(2, 3)
vecs = [ctx.zeros((n,)) for _ in range(20)]
for v in vecs[:4]:
    v.set(np.random.default_rng(0).standard_normal(n))

def timeit(fn, reps=5):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps

for m, k in ((12, 3), (8, 3), (4, 3), (4, 1), (16, 4), (3, 3), (1, 1)):
    dt = timeit(lambda: ctx.gram(vecs[:m], vecs[16:16 + k]))
    print(f"gram   {m:2d} x {k}: {dt*1e3:7.3f} ms  {8e-12*n*(m+k)/dt:5.2f} TB/s (unique bytes)", flush=True)
for m, k in ((12, 3), (24, 3), (3, 3)):
    C = np.ones((m, k))
    dt = timeit(lambda: ctx.lincomb_multi(vecs[16:16 + k], (vecs[:12] * 2)[:m], C))
    print(f"lincomb {m:2d} -> {k}: {dt*1e3:7.3f} ms  {8e-12*n*(m+k)/dt:5.2f} TB/s", flush=True)
for p in (6, 12, 3):
    dt = timeit(lambda: ctx.dots(vecs[:p], [vecs[16]] * p))
    print(f"dots   {p:2d} pairs against one vector: {dt*1e3:7.3f} ms  {8e-12*n*(p+1)/dt:5.2f} TB/s (unique bytes)", flush=True)
ctx.close()
