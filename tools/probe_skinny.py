"""GPU probe: the short-K skinny-N streaming products of a (50,200) CCSD iteration (A [M x K] K-contiguous, B [K x N]) through
pymes_dgemm — run once per library build (PYMES_AMD_LIBRARY) for an A/B."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from pymes_amd.device import Context
ctx = Context(4, 4, workspace_bytes=1 << 28)
rng = np.random.default_rng(0)
for M, N, K, beta in [(2000000, 50, 200, 0.0), (2000000, 50, 200, 1.0), (500000, 50, 200, 0.0), (2000000, 50, 50, 1.0), (125000, 50, 200, 0.0),
                      (432000, 30, 120, 1.0), (128000, 20, 80, 0.0)]:
    A = ctx.array(rng.standard_normal((M // 8, K)).repeat(8, axis=0))
    B = ctx.array(rng.standard_normal((K, N)))
    Cm = ctx.zeros((M, N))
    def go():
        ctx.dgemm(M, N, K, 1.0, A, K, 1, B, N, 1, beta, Cm, N)
    for _ in range(10):
        go()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        go()
    ctx.sync()
    dt = (time.perf_counter() - t0) / 20
    gb = 8.0 * (M * K + M * N * (2 if beta else 1)) / 1e9
    print(f"M={M} N={N} K={K} beta={beta}: {1e6 * dt:8.1f} us  {gb / dt / 1e3:5.2f} TB/s  {2.0 * M * N * K / dt / 1e12:5.1f} TF", flush=True)
    for x in (A, B, Cm):
        x.free()
