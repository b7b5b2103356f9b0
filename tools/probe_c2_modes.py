#!/usr/bin/env python3
"""GPU probe: (20,80) CCSD iteration time under the combinations engine stream (own / torch-created) x DIIS step
(numpy on the host / one native call) x per-phase event marks on or off — hunting a 2x slowdown that shows in some of them."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pymes_amd import dist as pdist
from pymes_amd.integral.device import DeviceIntegrals
from pymes_amd.model import synthetic
from pymes_amd.solver.ccsd import CCSD

no, nv = 20, 80
B, eps = synthetic.factors(no, nv, seed=0, scale=0.15)
for stream in ("own", "torch"):
    for diis in ("numpy", "native"):
        for marks in (False, True):
            os.environ.pop("PYMES_NUMPY_DIIS", None)
            if diis == "numpy":
                os.environ["PYMES_NUMPY_DIIS"] = "1"
            ints = DeviceIntegrals.from_factors(no, B)
            ctx = ints.ctx
            if stream == "torch":
                ts = torch.cuda.Stream()
                ctx.set_stream(ts.cuda_stream)
                pdist.trace.stream = ts
            else:
                pdist.trace.stream = None
            solver = CCSD(no)
            with contextlib.redirect_stdout(io.StringIO()):
                st = solver.setup(np.diag(eps), ints)
                for _ in range(4):
                    solver.iterate(st)
                pdist.trace.enable(marks and stream == "torch")
                ctx.sync()
                t0 = time.perf_counter()
                for _ in range(30):
                    solver.iterate(st)
                ctx.sync()
                dt = (time.perf_counter() - t0) / 30
            pdist.trace.enable(False)
            print(f"stream={stream:5s} diis={diis:6s} marks={marks!s:5s}: {dt*1e3:.3f} ms / iteration", flush=True)
            ctx.close()
