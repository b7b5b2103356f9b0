#!/usr/bin/env python3
"""GPU probe: what the beta term of the LDS-DMA GEMM's epilogue costs at short K (whole tiles, one per CU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymes_amd.device import Context
ctx = Context(4, 4, workspace_bytes=1 << 28)
ctx.prof_enable(True)
rng = np.random.default_rng(0)
for (M, N, K) in ((2048, 2048, 1600), (2048, 2048, 3200), (1600, 1600, 1600), (2048, 2048, 10000), (4096, 2048, 3200), (4096, 4096, 3200)):
    A = ctx.array(rng.standard_normal((M, K))); B = ctx.array(rng.standard_normal((K, N))); Cm = ctx.zeros((M, N)); Ci = ctx.array(rng.standard_normal((M, N)))
    T = -(-M // 128) * -(-N // 128)
    for plan in ("%d,1,4" % T, "%d,1,8" % T, "0,2,4", "0,2,8", None):
        for beta in (0.0, 1.0):
            if plan is None: os.environ.pop("PYMES_GEMM_PLAN", None)
            else: os.environ["PYMES_GEMM_PLAN"] = plan
            for _ in range(3): ctx.dgemm(M, N, K, 1.0, A, K, 1, B, N, 1, beta, Cm, N)
            ctx.sync(); ctx.prof_reset()
            for _ in range(6): ctx.dgemm(M, N, K, 1.0, A, K, 1, B, N, 1, beta, Cm, N)
            ctx.sync(); q = ctx.prof_query()
            print(f"M={M} N={N} K={K} plan={plan} beta={beta}: {q['ms']/6*1e3:8.1f} us  {q['flops']/(q['ms']*1e-3)/1e12:6.2f} TF  (pure MFMA time of a tile {K/16*1.707:6.1f} us)", flush=True)
    for x in (A, B, Cm, Ci): x.free()
