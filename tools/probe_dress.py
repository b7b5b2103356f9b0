#!/usr/bin/env python3
"""GPU probe: the bra dressing of the pair-packed V_abcd at (50,200) — both halves, random data — timed with HIP events
(stream = torch's current stream, which the context is bound to)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pymes_amd.device import Context, DeviceArray

no, nv = int(os.environ.get("NO", 50)), int(os.environ.get("NV", 200))
npp, npm = nv * (nv + 1) // 2, nv * (nv - 1) // 2
ctx = Context(no, nv, workspace_bytes=1 << 20, stream=torch.cuda.current_stream().cuda_stream)


def wrap(t):
    return DeviceArray(ctx, t.data_ptr(), tuple(t.shape), owned=False, keepalive=t)


t1 = torch.randn(nv, no, dtype=torch.float64, device="cuda") * 0.05
for label, ncol, minus in (("plus", npp, False), ("minus", npm, True)):
    ld = (ncol + 15) // 16 * 16
    V = torch.randn(npp, ld, dtype=torch.float64, device="cuda")
    Pk = torch.randn(no * nv, ld, dtype=torch.float64, device="cuda")
    W = torch.empty_like(V)
    args = (wrap(V), wrap(Pk), wrap(t1), wrap(W), ld, 0, npp)
    ctx.ladder_dress(*args, minus_half=minus)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        ctx.ladder_dress(*args, minus_half=minus)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gb = 2 * npp * ld * 8 / 1e9
    print(f"{label:6s} [{npp} x {ld}]  {ms:8.3f} ms   V in + W out {gb:5.2f} GB -> {gb/ms:6.2f} TB/s;  "
          f"MFMA flops {4.0*npp*ld*(4*((no+3)//4))/1e9/ms:7.1f} GF/ms", flush=True)
    # spot check of one row against torch
    a, b = nv - 3, 7
    r = a * (a + 1) // 2 + b
    P3 = Pk.view(nv, no, ld)
    ref = V[r] - t1[a] @ P3[b] + (1 if minus else -1) * (t1[b] @ P3[a])
    print("   max |row - torch| =", float((W[r] - ref).abs().max()))
    del V, Pk, W
ctx.close()
