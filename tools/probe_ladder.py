#!/usr/bin/env python3
"""GPU probe: the pair-packed ladder GEMM of (50,200) — [20100 x 20100] (K-contiguous) times [20100 x 1275] — against
variations of its extents and pitches (which of them costs the 7 % it runs below the ring products)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymes_amd.device import Context, DeviceArray

ctx = Context(4, 4, workspace_bytes=1 << 28)
ctx.prof_enable(True)
cases = [  # label, M, N, K, lda (A row pitch), ldb (B row pitch), ldc
    ("ladder as run", 20100, 1275, 20100, 20112, 1276, 2500),
    ("ldc = 1280", 20100, 1275, 20100, 20112, 1276, 1280),
    ("ldb = 1280", 20100, 1275, 20100, 20112, 1280, 2500),
    ("N = 1280", 20100, 1280, 20100, 20112, 1280, 2500),
    ("M = 20096", 20096, 1275, 20100, 20112, 1276, 2500),
    ("M = 10000 (Q_kb)", 10000, 1275, 20100, 20112, 1276, 2500),
    ("M = 15000", 15000, 1275, 20100, 20112, 1276, 2500),
    ("K = 20096", 20100, 1275, 20096, 20112, 1276, 2500),
    ("K = 10000", 20100, 1275, 10000, 20112, 1276, 2500),
    ("N = 2560", 20100, 2560, 20100, 20112, 2560, 2560),
]
A = ctx.zeros((20100 * 20112,))
B = ctx.zeros((20100 * 2560,))
Cm = ctx.zeros((20100 * 2560,))
for label, M, N, K, lda, ldb, ldc in cases:
    ctx.dgemm(M, N, K, 1.0, A, lda, 1, B, ldb, 1, 0.0, Cm, ldc)
    ctx.sync(); ctx.prof_reset()
    for _ in range(3):
        ctx.dgemm(M, N, K, 1.0, A, lda, 1, B, ldb, 1, 0.0, Cm, ldc)
    ctx.sync()
    q = ctx.prof_query()
    print(f"{label:20s} M={M} N={N} K={K}: {q['ms']/3:8.3f} ms  {q['flops']/(q['ms']*1e-3)/1e12:6.2f} TF", flush=True)
