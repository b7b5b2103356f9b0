"""GPU probe: what the int8 matrix cores sustain through the vendor library on the shapes of the ring products (feasibility of an
int8-split DGEMM): torch._int_mm (hipBLASLt) on [M,K] x [K,N] int8 -> int32."""
import time
import torch
dev = torch.device("cuda", 0)
for M, N, K in [(10240, 10240, 10240), (8192, 8192, 8192), (10000, 10000, 10000), (20480, 1280, 20480)]:
    a = torch.randint(-127, 127, (M, K), dtype=torch.int8, device=dev)
    b = torch.randint(-127, 127, (K, N), dtype=torch.int8, device=dev)
    try:
        for _ in range(3):
            c = torch._int_mm(a, b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            c = torch._int_mm(a, b)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"int8 {M}x{N}x{K}: {1e3 * dt:7.3f} ms  {2.0 * M * N * K / dt / 1e15:6.3f} POP/s", flush=True)
    except Exception as e:
        print("int8", M, N, K, "failed:", repr(e)[:200], flush=True)
    for dt_, name in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
        x = torch.randn((M, K), dtype=dt_, device=dev)
        y = torch.randn((K, N), dtype=dt_, device=dev)
        for _ in range(3):
            z = x @ y
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            z = x @ y
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{name} {M}x{N}x{K}: {1e3 * dt:7.3f} ms  {2.0 * M * N * K / dt / 1e15:6.3f} PFLOP/s", flush=True)
