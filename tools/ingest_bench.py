#!/usr/bin/env python3
"""Integral ingestion throughput: the native FCIDUMP text parser against the reference's per-line
Python loop (pymes/util/fcidump.py:124-161) on the same 66-orbital file, and the packed binary format.

    PYTHONPATH=/root/reference python tools/ingest_bench.py --reference      build container: reference vs native, host only
    python tools/ingest_bench.py --device                                    GPU box: text -> device, packed -> device

One JSON line per measurement (copy into profiles/rNN/ingest.jsonl)."""
import argparse
import contextlib
import io
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_text(path, n, n_elec, B, h, e_core):
    """(ij|kl) for i >= j, k >= l — both (ij|kl) and (kl|ij) are listed, as the reference's reader needs (fcidump.py:143-146
    restores the index-swap images but not the electron-exchange one)."""
    chem = np.einsum("Qij,Qkl->ijkl", B, B, optimize=True)
    lines = 0
    with open(path, "w") as f:
        f.write("&FCI NORB=%d,NELEC=%d,MS2=0,\n ORBSYM=%s,\n ISYM=1,\n&END\n" % (n, n_elec, ",".join(["1"] * n)))
        tri = [(i, j) for i in range(n) for j in range(i + 1)]
        for i, j in tri:
            row = chem[i, j]
            f.write("".join(" %.16e %d %d %d %d\n" % (row[k, l], i + 1, j + 1, k + 1, l + 1) for k, l in tri))
            lines += len(tri)
        for i in range(n):
            for j in range(i + 1):
                f.write(" %.16e %d %d 0 0\n" % (h[i, j], i + 1, j + 1))
        f.write(" %.16e 0 0 0 0\n" % e_core)
    return lines


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--norb", type=int, default=66)
    ap.add_argument("--nelec", type=int, default=16)
    ap.add_argument("--reference", action="store_true", help="time the imported reference reader too (build container)")
    ap.add_argument("--device", action="store_true", help="time the device paths (GPU box)")
    args = ap.parse_args()
    from oracle.io_oracle import synthetic_factors
    from pymes_amd.util import fcidump, packed
    n, ne = args.norb, args.nelec
    B, eps = synthetic_factors(ne // 2, n - ne // 2, seed=0)
    h = np.diag(eps)
    tmp = tempfile.mkdtemp(prefix="ingest_")
    txt, pk, pkf = os.path.join(tmp, "FCIDUMP"), os.path.join(tmp, "V.pk"), os.path.join(tmp, "B.pk")
    lines = write_text(txt, n, ne, B, h, 0.5)
    size = os.path.getsize(txt)
    base = {"norb": n, "two_electron_lines": lines, "text_bytes": size, "host_cores": os.cpu_count()}
    t_native = 1e30
    for _ in range(3):          # best of three (page cache, first touch of the 150-MB result)
        t0 = time.perf_counter()
        mine = quiet(fcidump.read, txt)
        t_native = min(t_native, time.perf_counter() - t0)
    print(json.dumps(dict(base, what="native text parser + host fill (pymes_fcidump_read_host)", seconds=t_native,
                          lines_per_s=lines / t_native, MB_per_s=size / t_native / 1e6)), flush=True)
    if args.reference:
        from pymes.util import fcidump as ref_fcidump
        t_ref = 1e30
        for _ in range(2):
            t0 = time.perf_counter()
            ref = quiet(ref_fcidump.read, txt)
            t_ref = min(t_ref, time.perf_counter() - t0)
        same = all(np.array_equal(a, b) for a, b in zip(mine[3:], ref[3:])) and mine[:3] == tuple(ref[:3])
        print(json.dumps(dict(base, what="reference fcidump.read (per-line Python loop, fcidump.py:124-161)", seconds=t_ref,
                              lines_per_s=lines / t_ref, MB_per_s=size / t_ref / 1e6, identical_to_native=bool(same),
                              speedup_native=t_ref / t_native)), flush=True)
    if args.device:
        t0 = time.perf_counter()
        out = quiet(fcidump.read_to_device, txt)
        out[5].ctx.sync()
        t_dev = time.perf_counter() - t0
        print(json.dumps(dict(base, what="native text parser + device fill into the 16 blocks (pymes_fcidump_load)",
                              seconds=t_dev, lines_per_s=lines / t_dev)), flush=True)
        packed.write_packed(pk, ne, 0.5, eps, h, out[5])
        for nm in ("abcd", "iajb", "klij"):
            assert np.array_equal(out[5].block(nm).get(), quiet(packed.read_packed_to_device, pk)[5].block(nm).get())
        out[5].ctx.close()
        for path, what in ((pk, "packed blocks -> device (pymes_packed_load)"),):
            t0 = time.perf_counter()
            r = quiet(packed.read_packed_to_device, path)
            r[5].ctx.sync()
            dt = time.perf_counter() - t0
            r[5].ctx.close()
            nbytes = os.path.getsize(path)
            print(json.dumps(dict(base, what=what, seconds=dt, file_bytes=nbytes, GB_per_s=nbytes / dt / 1e9)), flush=True)
        packed.write_factors(pkf, ne, 0.5, eps, h, B)
        t0 = time.perf_counter()
        r = quiet(packed.read_packed_to_device, pkf)
        r[5].ctx.sync()
        dt = time.perf_counter() - t0
        r[5].ctx.close()
        print(json.dumps(dict(base, what="packed factors -> device, V formed by the MFMA GEMM", seconds=dt,
                              file_bytes=os.path.getsize(pkf))), flush=True)
    for p in (txt, pk, pkf):
        if os.path.exists(p):
            os.remove(p)
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
