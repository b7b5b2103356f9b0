"""GPU probe: pymes_pair_layouts (t2_layouts_kernel) at (50,200) — 0.8 GB in (read twice), 3 x 0.8 GB out."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import ctypes as C
import numpy as np
from pymes_amd.device import Context
no, nv = 50, 200
ctx = Context(no, nv)
t2 = ctx.zeros((nv, nv, no, no))
outs = [ctx.zeros((no * nv, no * nv)) for _ in range(3)]
def go():
    ctx.lib.call("pymes_pair_layouts", ctx.handle, C.c_void_p(t2.ptr), C.c_void_p(outs[0].ptr), C.c_void_p(outs[1].ptr), C.c_void_p(outs[2].ptr))
for _ in range(30):
    go()
ctx.sync()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(50):
        go()
    ctx.sync()
    dt = (time.perf_counter() - t0) / 50
    print(f"t2_layouts (50,200): {1e6 * dt:8.1f} us  {5 * 0.8e9 / dt / 1e12:5.2f} TB/s (5 x 0.8 GB)", flush=True)
