#!/usr/bin/env python3
"""Random GEMM shapes / layouts / pitches / alpha-beta against numpy (one-off robustness run on the GPU box):
    python3 tools/fuzz_gemm.py [cases] [seed]
Covers the tile-choice rules (64x64 for short-K skinny shapes, under-filled launches, deep k-splits, LDS-DMA kernel from
K >= 384, matrix-vector kernels) with ragged extents; prints the failures and a summary."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymes_amd.device import Context

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
ctx = Context(4, 4, workspace_bytes=1 << 28)
bad = 0
sizes = [1, 2, 3, 16, 17, 31, 50, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 257, 300, 383, 384, 385, 500, 777, 1000,
         1275, 1600, 2049, 4097]
for case in range(n_cases):
    M, N, K = (int(rng.choice(sizes)) for _ in range(3))
    if rng.random() < 0.15:
        K = int(rng.choice([5000, 20000, 100001]))
        M, N = min(M, 300), min(N, 300)
    if M * N * K > 3e10 or M * K > 4e7 or K * N > 4e7:
        continue
    a_kc, b_kc = bool(rng.integers(2)), bool(rng.integers(2))
    alpha = float(rng.choice([1.0, -0.5, 2.0]))
    beta = float(rng.choice([0.0, 0.0, 1.0, 0.25]))
    pa, pb, pc = (int(rng.choice([0, 0, 1, 2, 3])) for _ in range(3))
    a_rows, a_cols = (M, K) if a_kc else (K, M)
    b_rows, b_cols = (N, K) if b_kc else (K, N)
    A = rng.standard_normal((a_rows, a_cols + pa))
    B = rng.standard_normal((b_rows, b_cols + pb))
    Cm = rng.standard_normal((M, N + pc))
    Am = A[:, :a_cols] if a_kc else A[:, :a_cols].T
    Bm = B[:, :b_cols].T if b_kc else B[:, :b_cols]
    ref = Cm.copy()
    ref[:, :N] = alpha * (Am @ Bm) + beta * Cm[:, :N]
    dA, dB, dC = ctx.array(A), ctx.array(B), ctx.array(Cm)
    a_sm, a_sk = (A.shape[1], 1) if a_kc else (1, A.shape[1])
    b_sk, b_sn = (1, B.shape[1]) if b_kc else (B.shape[1], 1)
    ctx.dgemm(M, N, K, alpha, dA, a_sm, a_sk, dB, b_sk, b_sn, beta, dC, Cm.shape[1])
    got = dC.get()
    err = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
    tol = 1e-13 * max(1, K) ** 0.5 + 1e-14
    if not err < tol:
        bad += 1
        print("FAIL", dict(M=M, N=N, K=K, a_kc=a_kc, b_kc=b_kc, alpha=alpha, beta=beta, pads=(pa, pb, pc), err=err), flush=True)
    for x in (dA, dB, dC):
        x.free()
print(f"fuzz_gemm: {n_cases} cases drawn, {bad} failures (seed {seed})")
sys.exit(1 if bad else 0)
