#!/usr/bin/env python3
"""GPU probe: launch plans of the LDS-DMA GEMM (whole tiles + cut tail in one grid, kernels.hip::plan_dma) on the product
shapes where tile quantisation costs: the plan the model picks next to hand-set ones (PYMES_GEMM_PLAN=whole,cuts).

    python3 tools/probe_plan.py [shape-filter]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymes_amd.device import Context

PEAK = 78.6
LOG = "/tmp/probe_plan_gemm.log"
os.environ["PYMES_GEMM_LOG"] = LOG
# (label, M, N, K, batch, [plans]); A is [M,K] K-contiguous, B is [K,N] N-contiguous (the layout of every ring / ladder product)
SHAPES = [
    ("slab ring 1/8 (50,200)", 1250, 10000, 10000, 1, ["768,8", "sk,512,512", "sk,0,512", "512,7", "768,8", "sk,512,512"]),
    ("slab ring 1/4 (50,200)", 2500, 10000, 10000, 1, ["1536,5", "sk,1024,512", "sk,1536,512", "1536,5", "sk,1024,512"]),
    ("slab ring 1/2 (50,200)", 5000, 10000, 10000, 1, ["3072,8", "sk,2560,512", "sk,3072,512", "3072,8"]),
    ("ring (30,120) 3600^3", 3600, 3600, 3600, 1, ["512,3", "sk,512,512", "sk,0,512", "sk,768,256", "512,3", "sk,512,512"]),
    ("stacked sigma k=3 (30,120)", 10800, 3600, 3600, 1, ["2048,3", "2465,1", "sk,2048,512", "sk,1536,512", "2465,1"]),
    ("stacked sigma k=4 (30,120)", 14400, 3600, 3600, 1, ["3072,5", "3277,1", "sk,3072,512", "sk,2560,512", "3277,1"]),
    ("ring (20,80) 1600^3", 1600, 1600, 1600, 1, ["0,3", "sk,0,512", "sk,0,256", "sk,0,384", "0,3", "sk,0,512"]),
    ("ring pair (20,80) 2 x 1600^3", 1600, 1600, 1600, 2, ["256,3", "sk,0,512", "sk,256,256", "sk,0,256", "256,3", "sk,0,512"]),
    ("ladder half (50,200)", 20100, 1275, 20100, 1, ["1536,5", "sk,1024,512", "sk,1536,512", "1536,5"]),
    ("ladder slab 1/8 (50,200)", 2513, 1275, 20100, 1, ["0,5", "sk,0,512", "sk,0,256", "0,10", "0,5", "sk,0,512"]),
    ("ladder half (30,120)", 7260, 465, 7260, 1, ["0,1", "sk,0,512", "sk,0,256", "0,3", "0,1", "sk,0,512"]),
    ("ladder half (20,80)", 3240, 210, 3240, 1, ["0,1", "sk,0,512", "sk,0,256", "sk,0,128", "0,3"]),
    ("ring (12,48) 576^3", 576, 576, 576, 1, ["0,1", "sk,0,64", "sk,0,32"]),
    ("ring (50,200) 10^4 cube", 10000, 10000, 10000, 1, ["6144,2", "sk,5632,512", "6144,2"]),
]

def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    ctx = Context(4, 4, workspace_bytes=1 << 28)
    ctx.prof_enable(True)
    rng = np.random.default_rng(0)
    for label, M, N, K, nb, plans in SHAPES:
        if flt and flt not in label:
            continue
        ldb = N + (N & 1)
        A = ctx.array(rng.standard_normal((nb, M, K)))
        Bh = np.zeros((nb, K, ldb))
        Bh[:, :, :N] = rng.standard_normal((nb, K, N))
        B = ctx.array(Bh)
        del Bh
        Cm = ctx.zeros((nb, M, N))
        tiles = -(-M // 128) * -(-N // 128) * nb
        print(f"== {label}: M={M} N={N} K={K} batch={nb}  tiles={tiles} ({tiles / 512:.2f} rounds of 512), k-tiles={-(-K // 16)}", flush=True)

        def run(plan):
            if plan is None:
                os.environ.pop("PYMES_GEMM_PLAN", None)
            else:
                os.environ["PYMES_GEMM_PLAN"] = plan

            def go():
                if nb > 1:
                    ctx.contract("zmk,zkn->zmn", A, B, out=Cm, batch="z")
                else:
                    ctx.dgemm(M, N, K, 1.0, A, K, 1, B, ldb, 1, 0.0, Cm, N)
            go()
            ctx.sync()
            ctx.prof_reset()
            reps = 4
            for _ in range(reps):
                go()
            ctx.sync()
            if os.path.exists(LOG):
                os.remove(LOG)
            q = ctx.prof_query()
            log = open(LOG).read().splitlines()[0].split("flops=")[1].split(" ", 1)[1] if os.path.exists(LOG) else ""
            tf = q["flops"] / (q["ms"] * 1e-3) / 1e12
            print(f"   plan {str(plan):>10s}: {q['ms'] / reps * 1e3:10.1f} us  {tf:6.2f} TF ({100 * tf / PEAK:5.1f} %)  {log}", flush=True)
        run(plans[0])         # (clock ramp-up after the host-side set-up: not a measurement)
        run(None)
        for p in plans:
            run(p)
        os.environ.pop("PYMES_GEMM_PLAN", None)
        for x in (A, B, Cm):
            x.free()
    ctx.close()


if __name__ == "__main__":
    main()
