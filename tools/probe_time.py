#!/usr/bin/env python3
"""Time one GEMM shape through pymes_dgemm: python3 tools/probe_time.py M N K akc bkc [reps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymes_amd.device import Context
M, N, K, akc, bkc = (int(x) for x in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
ctx = Context(4, 4, workspace_bytes=1 << 28)
rng = np.random.default_rng(0)
A = ctx.array(rng.standard_normal((M, K) if akc else (K, M)))
B = ctx.array(rng.standard_normal((N, K) if bkc else (K, N)))
Cm = ctx.zeros((M, N))
a_sm, a_sk = (K, 1) if akc else (1, M)
b_sk, b_sn = (1, K) if bkc else (N, 1)
for _ in range(3):
    ctx.dgemm(M, N, K, 1.0, A, a_sm, a_sk, B, b_sk, b_sn, 0.0, Cm, N)
ctx.sync()
t0 = time.perf_counter()
for _ in range(reps):
    ctx.dgemm(M, N, K, 1.0, A, a_sm, a_sk, B, b_sk, b_sn, 0.0, Cm, N)
ctx.sync()
dt = (time.perf_counter() - t0) / reps
print(f"M={M} N={N} K={K} akc={akc} bkc={bkc} env={os.environ.get('PYMES_GEMM_NO_LDSDMA','-')}: {dt*1e3:.3f} ms  {2.0*M*N*K/dt/1e12:.1f} TF")
