#!/usr/bin/env python3
"""UEG N=14, rs=1: transcorrelated integral build on the device + one DCSD iteration for growing plane-wave cutoffs
(SURVEY §8(d) config 4 beyond the reference's 57 plane waves):  python3 tools/ueg_scale.py 5 7 9 11"""
import contextlib, io, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymes_amd.model.ueg import UEG
from pymes_amd.device import Context
from pymes_amd.integral.device import DeviceIntegrals
from pymes_amd.solver.ccsd import CCSD
from pymes_amd.mean_field import hf

def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)

nel, rs = 14, 1.0
for cutoff in [float(x) for x in (sys.argv[1:] or ["5", "7", "9"])]:
    m = UEG(nel, nel // 2, nel // 2, rs)
    m.init_single_basis(cutoff)
    n_p, no = len(m.basis_fns) // 2, nel // 2
    m.k_cutoff = m.L / (2 * np.pi) * 2.3225029893472993 / rs
    ctx = Context(no, n_p - no)
    t0 = time.perf_counter()
    V = quiet(m.eval_2b_integrals, correlator=m.trunc, is_only_2b=True, sp=0, on_device=True, ctx=ctx)
    ctx.sync(); t1 = time.perf_counter()
    Va = quiet(m.eval_2b_integrals, correlator=m.trunc, is_effect_2b=True, sp=0, on_device=True, ctx=ctx)
    ctx.sync(); t2 = time.perf_counter()
    out = {"cutoff": cutoff, "n_pw": n_p, "V_gb": 8e-9 * n_p**4, "only_2b_s": t1 - t0, "effect_2b_s": t2 - t1}
    print(json.dumps(out), flush=True)
    Va.free(); V.free(); ctx.close()
