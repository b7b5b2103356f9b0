#!/usr/bin/env python3
"""Random binary contractions (the einsum front end of the engine: label planning, cached transposes, batch labels,
alpha / beta) against numpy.einsum — a one-off robustness run on the GPU box:  python3 tools/fuzz_contract.py [cases] [seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymes_amd.device import Context

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
ctx = Context(4, 4, workspace_bytes=1 << 28)
ext = {}
bad = done = 0
letters = "abcdefgh"
while done < n_cases:
    nfree_a, nfree_b, nsum = int(rng.integers(0, 3)), int(rng.integers(0, 3)), int(rng.integers(1, 3))
    if nfree_a + nfree_b == 0:
        continue
    labs = list(letters[: nfree_a + nfree_b + nsum])
    rng.shuffle(labs)
    fa, fb, su = labs[:nfree_a], labs[nfree_a:nfree_a + nfree_b], labs[nfree_a + nfree_b:]
    dims = {ch: int(rng.choice([1, 2, 3, 5, 8, 13, 20, 33, 50])) for ch in labs}
    la = fa + su
    lb = su + fb
    rng.shuffle(la)
    rng.shuffle(lb)
    lc = fa + fb
    rng.shuffle(lc)
    la, lb, lc = "".join(la), "".join(lb), "".join(lc)
    if len(la) > 4 or len(lb) > 4 or len(lc) > 4 or len(lc) == 0:
        continue
    A = rng.standard_normal([dims[c] for c in la])
    B = rng.standard_normal([dims[c] for c in lb])
    C0 = rng.standard_normal([dims[c] for c in lc])
    alpha = float(rng.choice([1.0, -1.0, 0.5, 2.0]))
    beta = float(rng.choice([0.0, 0.0, 1.0, -0.5]))
    ref = alpha * np.einsum(f"{la},{lb}->{lc}", A, B) + beta * C0
    spec = f"{la},{lb}->{lc}"
    try:
        out = ctx.array(C0)
        ctx.contract(spec, ctx.array(A), ctx.array(B), out=out, alpha=alpha, beta=beta)
        got = out.get()
    except Exception as e:                          # a spec the planner refuses is reported, not counted as wrong
        print("REFUSED", spec, dims, str(e)[:100], flush=True)
        done += 1
        continue
    err = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
    if not err < 1e-12:
        bad += 1
        print("FAIL", spec, dims, alpha, beta, err, flush=True)
    done += 1
print(f"fuzz_contract: {done} cases, {bad} failures (seed {seed})")
sys.exit(1 if bad else 0)
