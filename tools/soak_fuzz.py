#!/usr/bin/env python3
"""GPU soak: the seeded fuzz tests of tests/test_gpu_fuzz.py over MANY seeds (the suite runs two), one process, with a
progress line per seed.  gpurun -- 'timeout -k 10 900 python3 tools/soak_fuzz.py 2 60'"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymes_amd import _lib
from tests import test_gpu_fuzz as tf
lo, hi = int(sys.argv[1]), int(sys.argv[2])
lib = _lib.default_library()
t0 = time.time()
for seed in range(lo, hi):
    for fn in (tf.test_random_gemm_shapes, tf.test_random_contractions, tf.test_random_bra_dressing_shapes):
        fn(lib, seed)
    print(f"seed {seed} ok  ({time.time() - t0:.0f} s)", flush=True)
print("soak ok")
