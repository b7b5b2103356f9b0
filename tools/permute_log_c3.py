#!/usr/bin/env python3
"""Explicit permutations of ONE (50,200) CCSD iteration (eager pass), largest first: PYMES_PERMUTE_LOG through the engine."""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
LOG = "/tmp/permute_c3.log"
from pymes_amd.integral.device import DeviceIntegrals
from pymes_amd.model import synthetic
from pymes_amd.solver.ccsd import CCSD
no, nv = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (50, 200)
B, eps = synthetic.factors(no, nv, seed=0)
ints = DeviceIntegrals.from_factors(no, B)
s = CCSD(no)
os.environ["PYMES_NO_GRAPH"] = "1"
with contextlib.redirect_stdout(io.StringIO()):
    st = s.setup(np.diag(eps), ints)
    for _ in range(3):
        s.iterate(st)
    ints.ctx.sync()
    os.environ["PYMES_PERMUTE_LOG"] = LOG
    if os.path.exists(LOG):
        os.remove(LOG)
    ints.ctx.stats(reset=True)
    s.iterate(st)
    ints.ctx.sync()
os.environ.pop("PYMES_PERMUTE_LOG")
print(ints.ctx.stats())
tot = 0.0
for ln in open(LOG):
    print(ln.rstrip())
