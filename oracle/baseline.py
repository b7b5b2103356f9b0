"""CPU baseline for bench.py: the oracle's contraction forms timed on a bounded sample.
TEST/BENCH INFRASTRUCTURE ONLY (see oracle/__init__.py).

A full (50 occ, 200 virt) iteration of the reference takes hours and > 62 GB on the host
(SURVEY §6), so the dominant terms of ``cc_oracle.doubles_residual`` are timed on an a-slab
(their cost is exactly linear in the slab) and extrapolated to the whole iteration with the
flop table of SURVEY §8(d).  "faithful" = the call forms of the reference (plain
``np.einsum`` in the T2 residual, single-threaded C loop, ccd.py:180-240); "blas" = the same
contractions with ``optimize=True`` (tensordot -> multi-threaded dgemm)."""
import os
import time
from functools import partial

import numpy as np


def algorithmic_fma(no, nv, is_dcsd=False):
    """FMA count of one doubles residual, closed form of SURVEY §8(d)."""
    o, v = float(no), float(nv)
    if is_dcsd:
        return v**4 * o**2 + 5 * o**3 * v**3 + o**4 * v**2 + 2 * o**2 * v**3 + 2 * o**3 * v**2
    return v**4 * o**2 + 10 * o**3 * v**3 + 2 * o**4 * v**2 + 3 * o**2 * v**3 + 3 * o**3 * v**2


def sample(no, nv, budget_s=20.0, seed=0):
    rng = np.random.default_rng(seed)
    T = rng.standard_normal((nv, nv, no, no)) * 0.01
    Vijab = rng.standard_normal((no, no, nv, nv)) * 0.01
    out = {}
    for mode, ein in (("faithful", np.einsum), ("blas", partial(np.einsum, optimize=True))):
        # ladder ccd.py:187 on V[a:1, b-slab] -- linear in the (a,b) slab
        bs = max(1, min(nv, int(25 * (200.0 / nv) ** 2))) if mode == "faithful" else nv
        Vs = rng.standard_normal((1, bs, nv, nv)) * 0.01
        t0 = time.perf_counter()
        ein("abcd,cdij->abij", Vs, T)
        t_lad = (time.perf_counter() - t0) * (nv * nv / bs)
        # one quadratic ring term ccd.py:190 on an a-slab of 1 -- linear in a
        na = 1 if mode == "faithful" else min(nv, 16)      # BLAS needs a fat enough slab to be representative
        t0 = time.perf_counter()
        ein("klcd,adkj->alcj", Vijab, T[:na])
        t_ring = (time.perf_counter() - t0) * nv / na
        fma = algorithmic_fma(no, nv)
        lad_fma, ring_fma = nv**4 * no**2, float(no)**3 * nv**3
        # remaining terms priced at the ring-term rate (they have the same GEMM shape class)
        t_iter = t_lad + t_ring * (fma - lad_fma) / ring_fma
        out[mode] = {"seconds_per_doubles_residual": t_iter, "ladder_s": t_lad, "ring_term_s": t_ring,
                     "gflops": 2 * fma / t_iter / 1e9}
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count()
    out["cores"] = {"faithful": 1, "blas": ncpu}
    return out
