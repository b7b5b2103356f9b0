"""GPU: FCIDUMP ingestion straight into the device blocks (native parser + HIP fill kernel) and a CCSD solve on them."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

from tests.test_fcidump_hf import check_h2o_shape_chain, check_read_to_device

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_read_to_device(gpu_lib, monkeypatch, tmp_path):
    check_read_to_device(gpu_lib, monkeypatch, tmp_path, 12, 6, 3000)       # NORB <= 64: host fill + upload
    check_read_to_device(gpu_lib, monkeypatch, tmp_path, 66, 8, 30000)      # NORB > 64: fcidump_fill_kernel
    check_read_to_device(gpu_lib, monkeypatch, tmp_path, 70, 10, 5_000)


@pytest.mark.parametrize("device_path", [False, True])
def test_h2o_shape_chain_gpu(gpu_lib, monkeypatch, tmp_path, device_path):
    """BASELINE config 1 as far as it exists here (README): a (5,19) FCIDUMP through the native parser, the HF
    matrix and CCSD / DCSD on the HIP path — host arrays and device-resident blocks — against the reference's run on that file."""
    check_h2o_shape_chain(gpu_lib, monkeypatch, tmp_path, device_path)


def test_ccsd_from_device_fcidump(gpu_lib, monkeypatch):
    from pymes_amd import _lib
    from pymes_amd.mean_field import hf
    from pymes_amd.solver.ccsd import CCSD
    from pymes_amd.util import fcidump
    monkeypatch.setattr(_lib, "_default", gpu_lib)
    gold = json.load(open(os.path.join(GOLD, "solves.json")))
    path = os.path.join(GOLD, "fcidump", "FCIDUMP.LiH.sto6g")
    ne, n, ec, eps, h, ints = fcidump.read_to_device(path)
    try:
        no = ne // 2
        f = hf.construct_hf_matrix(no, h, ints)            # traces over the device blocks
        assert np.abs(f - hf.construct_hf_matrix(no, h, fcidump.read(path)[5])).max() < 1e-13
        with contextlib.redirect_stdout(io.StringIO()):
            r = CCSD(no, delta_e=1e-10).solve(f, ints)
        assert abs(r["ccsd e"] - gold["LiH.sto6g"]["ccsd"]["e"]) < 1e-9
    finally:
        ints.ctx.close()
