#!/usr/bin/env python3
"""The vendor library's fp64 GEMM (torch.matmul on ROCm: rocBLAS / hipBLASLt) on the shapes of the headline iteration, next to
the rates bench.py reports for this package's LDS-DMA kernel on the same shapes (profiles/rNN/bench_c3*.json: ring products
10000^3 in 26.9-27.1 ms = 74 TF, packed ladder halves 20100 x 1275 x 20100 in 13.6-14.5 ms = 71-72 TF).
    gpurun -- 'python3 tools/probe_library_dgemm.py'"""
import time
import torch

assert torch.cuda.is_available()
dev = torch.device("cuda:0")
shapes = [(10000, 10000, 10000, "ring product, (ov)^3 at (50,200)"),
          (20100, 1275, 20100, "pair-packed particle ladder, symmetric half"),
          (20100, 1225, 19900, "pair-packed particle ladder, antisymmetric half"),
          (3600, 3600, 3600, "ring product at (30,120)"),
          (14400, 3600, 3600, "stacked EOM sigma product, k = 4 at (30,120)"),
          (1600, 1600, 1600, "ring product at (20,80)")]
for M, N, K, what in shapes:
    torch.manual_seed(0)
    A = torch.randn(M, K, dtype=torch.float64, device=dev)
    B = torch.randn(K, N, dtype=torch.float64, device=dev)
    C = torch.empty(M, N, dtype=torch.float64, device=dev)
    for variant, (a, b) in (("A B", (A, B)), ("A B^T", (A, B.t().contiguous().t()))):
        for _ in range(3):
            torch.matmul(a, b, out=C)
        torch.cuda.synchronize()
        reps = 10 if M * N * K < 5e11 else 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            torch.matmul(a, b, out=C)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"{M:6d} x {N:6d} x {K:6d}  {variant:6s} {ms:9.3f} ms  {2.0 * M * N * K / ms / 1e9:7.2f} TF   {what}")
    del A, B, C
