#!/usr/bin/env python3
"""GEMM efficiency vs working-set size (is the kernel limited by memory latency?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymes_amd.device import Context
ctx = Context(4, 4, workspace_bytes=1 << 28)
ctx.prof_enable(True)
rng = np.random.default_rng(0)
for (M, N, K) in ((4096, 4096, 1024), (4096, 4096, 4096), (8192, 8192, 2048), (8192, 8192, 8192), (16384, 16384, 4096), (2048, 2048, 16384)):
    A = ctx.array(rng.standard_normal((M, K))); B = ctx.array(rng.standard_normal((K, N))); C = ctx.zeros((M, N))
    ctx.dgemm(M, N, K, 1.0, A, K, 1, B, N, 1, 0.0, C, N); ctx.sync(); ctx.prof_reset()
    for _ in range(5): ctx.dgemm(M, N, K, 1.0, A, K, 1, B, N, 1, 0.0, C, N)
    ctx.sync(); q = ctx.prof_query()
    tf = q["flops"] / (q["ms"] * 1e-3) / 1e12
    print(f"{M}x{N}x{K}: A+B {8*(M*K+K*N)/1e6:.0f} MB  {q['ms']/5:.3f} ms  {tf:.2f} TF ({100*tf/78.6:.1f}%)", flush=True)
    for x in (A, B, C): x.free()
