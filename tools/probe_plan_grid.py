#!/usr/bin/env python3
"""GPU probe: which (whole tiles, cuts) plan of the LDS-DMA GEMM wins as a function of (tiles, k-tiles) — the data the rule in
kernels.hip::plan_dma is fitted to.  For every tile count T and depth: whole = floor(T / 256) * 256 ("a"), floor(T / 512) * 512
("b") and 0 ("c") with several cut counts; prints the time of each and marks the best.

    python3 tools/probe_plan_grid.py [ktiles,ktiles,...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymes_amd.device import Context

TILES = [140, 169, 200, 228, 300, 338, 400, 470, 520, 600, 700, 790, 841, 900, 1000, 1100, 1300, 1580, 1800]
KTS = [100, 225, 454, 625, 1257]
CUTS = [2, 3, 4, 5, 6, 8, 10]


def main():
    kts = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else KTS
    ctx = Context(4, 4, workspace_bytes=1 << 28)
    ctx.prof_enable(True)
    rng = np.random.default_rng(0)
    for kt in kts:
        K = kt * 16
        for T in TILES:
            tn = 10 if T % 10 == 0 else (13 if T % 13 == 0 else (29 if T % 29 == 0 else (19 if T % 19 == 0 else 0)))
            if not tn:
                continue
            tm = T // tn
            M, N = tm * 128 - 40, tn * 128 - 6           # ragged edges as in the real products
            A = ctx.array(rng.standard_normal((M, K)))
            B = ctx.array(rng.standard_normal((K, N)))
            Cm = ctx.zeros((M, N))

            def run(plan, reps=4):
                os.environ["PYMES_GEMM_PLAN"] = plan
                ctx.dgemm(M, N, K, 1.0, A, K, 1, B, N, 1, 0.0, Cm, N)
                ctx.sync()
                ctx.prof_reset()
                for _ in range(reps):
                    ctx.dgemm(M, N, K, 1.0, A, K, 1, B, N, 1, 0.0, Cm, N)
                ctx.sync()
                return ctx.prof_query()["ms"] / reps * 1e3
            run(f"{T},1", reps=6)          # clock ramp-up
            res = {}
            wa, wb = (T // 256) * 256, (T // 512) * 512
            res[f"{T}+0/1"] = run(f"{T},1")
            for w in sorted({wa, wb, 0}, reverse=True):
                if w == T:
                    continue
                for s in CUTS:
                    if kt // s < 24 or (T - w) * s > 2048:
                        continue
                    res[f"{w}+{T - w}/{s}"] = run(f"{w},{s}")
            best = min(res, key=res.get)
            ideal = T / 256.0 * kt * 1.707
            line = "  ".join(f"{k}:{v:.0f}{'*' if k == best else ''}" for k, v in res.items())
            print(f"kt={kt:5d} T={T:5d} ideal={ideal:7.0f} best={best} ({res[best]:.0f} us, {ideal / res[best]:.3f})  |  {line}", flush=True)
            for x in (A, B, Cm):
                x.free()
    ctx.close()


if __name__ == "__main__":
    main()
