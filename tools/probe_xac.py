#!/usr/bin/env python3
"""GPU probe: the X_ac.T product of the (50,200) finish (M = 200, N = 5e5, K = 200; 25 flop per byte) under the tile choices of
dev::gemm (PYMES_GEMM_TILE override)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymes_amd.device import Context

ctx = Context(4, 4, workspace_bytes=1 << 28)
ctx.prof_enable(True)
rng = np.random.default_rng(0)
for label, M, N, K, akc, bkc, beta in (("X_ac.T  200 x 5e5 x 200", 200, 500000, 200, True, False, 0.0),
                                       ("t.QK    200 x 5e5 x 50", 200, 500000, 50, True, False, 1.0),
                                       ("V_abic.t 2e6 x 50 x 200", 2000000, 50, 200, True, False, 1.0)):
    A = ctx.array(rng.standard_normal((M, K)))
    B = ctx.array(rng.standard_normal((K, N)))
    Cm = ctx.zeros((M, N))
    for tile in ("", "64x64", "128x64", "64x128", "128x128"):
        if tile:
            os.environ["PYMES_GEMM_TILE"] = tile
        else:
            os.environ.pop("PYMES_GEMM_TILE", None)
        ctx.dgemm(M, N, K, 1.0, A, K, 1, B, N, 1, beta, Cm, N)
        ctx.sync(); ctx.prof_reset()
        for _ in range(3):
            ctx.dgemm(M, N, K, 1.0, A, K, 1, B, N, 1, beta, Cm, N)
        ctx.sync()
        q = ctx.prof_query()
        ms = q["ms"] / 3
        gb = 8e-9 * (M * K + K * N + (2 if beta else 1) * M * N)
        print(f"{label:28s} tile {tile or 'default':8s} {ms:8.3f} ms  {q['flops']/3/ms/1e9:6.1f} TF  {gb/ms*1e3/1e3:5.2f} TB/s", flush=True)
    for x in (A, B, Cm):
        x.free()
