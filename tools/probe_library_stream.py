#!/usr/bin/env python3
"""The vendor library (torch.matmul / bmm in fp64) on the HBM-streaming product shapes of a (50,200) iteration, next to this
package's times for the same shapes (PYMES_GEMM_LOG).  gpurun -- 'python3 tools/probe_library_stream.py'"""
import torch
dev = torch.device("cuda:0")
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for M, N, K, ours in [(2000000, 50, 200, "1.04-1.13"), (200, 500000, 200, "0.90"), (200, 50, 2000000, "1.06"), (500000, 50, 200, "0.26-0.27"), (200, 500000, 50, "0.46")]:
    A = torch.randn(M, K, dtype=torch.float64, device=dev); B = torch.randn(K, N, dtype=torch.float64, device=dev); C = torch.empty(M, N, dtype=torch.float64, device=dev)
    ms = t(lambda: torch.matmul(A, B, out=C))
    print(f"{M} x {N} x {K}: library {ms:.3f} ms ({2.0*M*N*K/ms/1e9:.1f} TF, {(M*K+K*N+M*N)*8/ms/1e9:.2f} TB/s); ours {ours} ms")
    del A, B, C
A = torch.randn(10000, 50, 200, dtype=torch.float64, device=dev); B = torch.randn(10000, 200, 200, dtype=torch.float64, device=dev); C = torch.empty(10000, 50, 200, dtype=torch.float64, device=dev)
ms = t(lambda: torch.bmm(A, B, out=C))
print(f"batch 10000 of 50 x 200 x 200: library {ms:.3f} ms; ours 1.18-1.24 ms")
