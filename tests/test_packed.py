"""Packed binary integral files (pymes_amd/util/packed.py, csrc/packed.h): round trips against the text FCIDUMP path
(the reference's only format, fcidump.py:59-163) — host logic through the host simulator on CPU, the HIP path on GPU."""
import contextlib
import io
import os

import numpy as np
import pytest

from oracle import cc_oracle as oc
from oracle import io_oracle as oio
from oracle.cases import synthetic_case
from pymes_amd import _lib

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def check(lib, monkeypatch, tmp_path):
    from pymes_amd.solver.ccsd import CCSD
    from pymes_amd.util import fcidump, packed
    monkeypatch.setattr(_lib, "_default", lib)
    # 1. a reference FCIDUMP fixture -> packed blocks -> back: bit-identical to the text reader's arrays
    src = os.path.join(GOLD, "fcidump", "FCIDUMP.LiH.321g")
    ne, n, ec, eps, h, V = quiet(fcidump.read, src)
    p1 = str(tmp_path / "lih.pk")
    packed.write_packed(p1, ne, ec, eps, h, V)
    assert packed.header(p1) == (packed.KIND_BLOCKS, ne, n, 0)
    assert os.path.getsize(p1) == 64 + 8 * (n + n * n + n ** 4)
    got = packed.read_packed(p1)
    assert got[:3] == (ne, n, ec)
    for a, b in zip(got[3:], (eps, h, V)):
        assert np.array_equal(a, b)
    ne2, n2, ec2, eps2, h2, ints = quiet(packed.read_packed_to_device, p1)
    try:
        assert (ne2, n2, ec2) == (ne, n, ec) and np.array_equal(eps2, eps) and np.array_equal(h2, h)
        for nm, blk in oc.split_blocks(ne // 2, V).items():
            assert np.array_equal(ints.block(nm).get(), blk), nm
        # written again straight from the device blocks: the same file
        p2 = str(tmp_path / "lih2.pk")
        packed.write_packed(p2, ne, ec, eps, h, ints)
        assert open(p1, "rb").read() == open(p2, "rb").read()
        # and the solver takes the device integrals as they are
        f = oio.fock_matrix(ne // 2, h, V)
        e_dev = quiet(CCSD(ne // 2, delta_e=1e-10).solve, f, ints)["ccsd e"]
    finally:
        ints.ctx.close()
    assert abs(e_dev - quiet(CCSD(ne // 2, delta_e=1e-10).solve, f, V)["ccsd e"]) < 1e-12
    # 2. density-fitting factors: n^2 naux numbers on disk, V formed on the device
    no, nv = 4, 9
    f, V, B, eps = synthetic_case(no, nv, seed=5, scale=0.3)
    p3 = str(tmp_path / "syn.pk")
    packed.write_factors(p3, 2 * no, 0.25, eps, np.diag(eps), B)
    assert packed.header(p3) == (packed.KIND_FACTORS, 2 * no, no + nv, B.shape[0])
    host = packed.read_packed(p3)
    assert host[2] == 0.25 and np.abs(host[5] - V).max() < 1e-13
    _, _, _, _, _, ints = quiet(packed.read_packed_to_device, p3)
    try:
        for nm, blk in oc.split_blocks(no, V).items():
            assert np.abs(ints.block(nm).get() - blk).max() < 1e-13, nm
    finally:
        ints.ctx.close()
    # 3. malformed files are refused
    raw = bytearray(open(p1, "rb").read())
    bad = str(tmp_path / "bad.pk")
    open(bad, "wb").write(bytes(raw[:-8]))
    with pytest.raises(ValueError, match="file length"):
        packed.header(bad)
    raw[0:8] = b"NOTPYMES"
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="not a PYMESPK1"):
        packed.read_packed(bad)
    with pytest.raises(FileNotFoundError):
        packed.header(str(tmp_path / "missing.pk"))
    with pytest.raises(ValueError, match="does not match the context"):
        from pymes_amd.device import Context
        import ctypes as C
        ctx = Context(2, 3)
        try:
            ec_ = C.c_double()
            ctx.lib.call("pymes_packed_load", ctx.handle, p1.encode(), C.byref(ec_), _lib.host_ptr(np.zeros(5)),
                         _lib.host_ptr(np.zeros((5, 5))))
        except _lib.PymesError as exc:
            raise ValueError(str(exc))
        finally:
            ctx.close()


def test_packed_host_logic(hostsim_lib, monkeypatch, tmp_path):
    check(hostsim_lib, monkeypatch, tmp_path)


@pytest.mark.gpu
def test_packed_gpu(gpu_lib, monkeypatch, tmp_path):
    check(gpu_lib, monkeypatch, tmp_path)
