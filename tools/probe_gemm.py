#!/usr/bin/env python3
"""GPU probe: achieved fp64 TFLOP/s of the MFMA GEMM on the shapes of the CC path."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymes_amd.device import Context

shapes = [  # (label, M, N, K, a_kc, b_kc)
    ("C2 ladder 6400x400x6400", 6400, 400, 6400, True, False),
    ("C2 ring 1600^3 NN", 1600, 1600, 1600, True, False),
    ("C3 ring 10^4 NN", 10000, 10000, 10000, True, False),
    ("C3 ring 10^4 NT", 10000, 10000, 10000, True, True),
    ("C3 ring 10^4 TN", 10000, 10000, 10000, False, False),
    ("C3 ladder slab 5000x2500x40000", 5000, 2500, 40000, True, False),
    ("8192^3 NN", 8192, 8192, 8192, True, False),
]
ctx = Context(4, 4, workspace_bytes=1 << 28)
ctx.prof_enable(True)
rng = np.random.default_rng(0)
for label, M, N, K, akc, bkc in shapes:
    A = ctx.array(rng.standard_normal((M, K) if akc else (K, M)))
    B = ctx.array(rng.standard_normal((N, K) if bkc else (K, N)))
    Cm = ctx.zeros((M, N))
    a_sm, a_sk = (K, 1) if akc else (1, M)
    b_sk, b_sn = (1, K) if bkc else (N, 1)
    ctx.dgemm(M, N, K, 1.0, A, a_sm, a_sk, B, b_sk, b_sn, 0.0, Cm, N)   # warm-up
    ctx.sync(); ctx.prof_reset()
    reps = 3
    for _ in range(reps):
        ctx.dgemm(M, N, K, 1.0, A, a_sm, a_sk, B, b_sk, b_sn, 0.0, Cm, N)
    ctx.sync()
    q = ctx.prof_query()
    tf = q["flops"] / (q["ms"] * 1e-3) / 1e12
    print(f"{label:36s} {q['ms']/reps:9.3f} ms  {tf:7.2f} TFLOP/s  ({100*tf/78.6:5.1f}% of 78.6)", flush=True)
    for x in (A, B, Cm): x.free()
