import os
import sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymes_amd.device import Context
from pymes_amd.mixer.diis import DIIS
ctx = Context(2, 3)
rng = np.random.default_rng(0)
mixer = DIIS(6)
errs = [[ctx.array(rng.standard_normal((3,2))*0.5**it), ctx.array(rng.standard_normal((3,3,2,2))*0.5**it)] for it in range(40)]
amps = [[ctx.array(rng.standard_normal((3,2))), ctx.array(rng.standard_normal((3,3,2,2)))] for it in range(40)]
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    for it in range(10):
        mixer.mix(errs[it], amps[it], on_device=True)
    ctx.sync()
    t = time.perf_counter()
    for it in range(10, 40):
        mixer.mix(errs[it], amps[it], on_device=True)
    ctx.sync()
dt = (time.perf_counter() - t) / 30
print("device mix() per call (3 launches + solve kernel + 2 lincomb): %.1f us" % (dt * 1e6))
mixer2 = DIIS(6)
with contextlib.redirect_stdout(io.StringIO()):
    for it in range(10):
        mixer2.mix(errs[it], amps[it])
    ctx.sync()
    t = time.perf_counter()
    for it in range(10, 40):
        mixer2.mix(errs[it], amps[it])
    ctx.sync()
print("host mix() per call: %.1f us" % ((time.perf_counter() - t) / 30 * 1e6))
