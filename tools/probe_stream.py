#!/usr/bin/env python3
"""Short-K, HBM-streaming GEMM shapes of the T1 dressing at (50,200): achieved bandwidth of the launch the library chooses.
(Rounds 2-4 swept tile shapes here through PYMES_GEMM_TILE; that knob left the library in round 5 with the choice it settled:
the numbers quoted in DESIGN 5 / 6c are from those sweeps, this script measures the shipped choice only.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymes_amd.device import Context
ctx = Context(4, 4, workspace_bytes=1 << 28)
ctx.prof_enable(True)
rng = np.random.default_rng(0)
o, v = 50, 200
# (label, spec, shape A, shape B, batch, beta)
cases = [
    ("pqrx,xs  M=2M N=50 K=200", "pqrx,xs->pqrs", (o, v, v, v), (v, o), "", 1.0),
    ("pqrx,xs  M=500k N=50 K=200", "pqrx,xs->pqrs", (o, o, v, v), (v, o), "", 1.0),
    ("pqxs,xr  batch 10000 50x200x200", "pqxs,xr->pqrs", (o, v, v, v), (v, o), "pq", 1.0),
    ("pqxs,xr  batch 2500 50x200x200", "pqxs,xr->pqrs", (o, o, v, v), (v, o), "pq", 1.0),
    ("qx,pxrs  batch 50 200x10000x50", "qx,pxrs->pqrs", (v, o), (o, o, v, o), "p", 1.0),
    ("px,xqrs  M=200 N=500k K=50", "px,xqrs->pqrs", (v, o), (o, v, o, o), "", 1.0),
    ("ac,cbij  M=200 N=500k K=200", "ac,cbij->abij", (v, v), (v, v, o, o), "", 0.0),
    ("ak,kn  M=200 N=50 K=2M (singles :433)", "ak,kn->an", (v, o * v * v), (o * v * v, o), "", 1.0),
    ("ak,kn  M=200 N=50 K=500k (singles :435)", "ak,kn->an", (v, o * o * v), (o * o * v, o), "", 1.0),
    ("ka,kn  M=50 N=50 K=2M", "ka,kn->an", (o * v * v, o), (o * v * v, o), "", 0.0),
]
if os.environ.get("PROBE_ONLY"):
    cases = [c for c in cases if os.environ["PROBE_ONLY"] in c[0]]
tiles = [""]
for (label, spec, sa, sb, batch, beta) in cases:
    A, B = ctx.empty(sa), ctx.empty(sb)
    A.zero_(); B.zero_()
    out = ctx.contract(spec, A, B, batch=batch)
    nbytes = 8.0 * (np.prod(sa) + np.prod(sb) + np.prod(out.shape) * (2 if beta else 1))
    for tile in tiles:
        ctx.contract(spec, A, B, out=out, beta=beta, batch=batch); ctx.sync(); ctx.prof_reset()
        for _ in range(5): ctx.contract(spec, A, B, out=out, beta=beta, batch=batch)
        ctx.sync(); q = ctx.prof_query()
        ms = q["ms"] / 5
        print(f"{label:36s} tile={tile or 'auto':8s} {ms:.3f} ms  {nbytes/1e9/ms:.2f} TB/s  {q['flops']/5/ms/1e9:.1f} TF", flush=True)
    for x in (A, B, out): x.free()
