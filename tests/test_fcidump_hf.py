"""CPU: product FCIDUMP reader / writer and HF helpers against the reference's known answers."""
import json
import os

import numpy as np

from oracle import io_oracle as oio
from pymes_amd.mean_field import hf
from pymes_amd.util import fcidump

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_reader_matches_reference_values(capsys):
    g = json.load(open(os.path.join(GOLD, "fcidump.json")))
    for tag, ref in g.items():
        path = os.path.join(GOLD, "fcidump", "FCIDUMP." + tag)
        ne, n, ec, eps, h, V = fcidump.read(path)
        o = oio.read_fcidump(path)
        assert (ne, n, ec) == (o[0], o[1], o[2]) == (ref["n_elec"], ref["n_orb"], ref["e_core"])
        assert np.array_equal(eps, o[3]) and np.array_equal(h, o[4]) and np.array_equal(V, o[5])
        no = ne // 2
        assert abs(hf.calc_hf_e(no, ec, h, V) - ref["e_hf"]) < 1e-12
        f = hf.construct_hf_matrix(no, h, V)
        assert np.abs(f.diagonal() - np.array(ref["fock_diag"])).max() < 1e-13
    Vtc = fcidump.read(os.path.join(GOLD, "fcidump", "FCIDUMP.H2.321g"), is_tc=True)[5]
    assert np.count_nonzero(Vtc) == g["H2.321g"]["V_nnz_is_tc"] and abs(Vtc.sum() - g["H2.321g"]["V_sum_is_tc"]) < 1e-12


def test_write_read_round_trip(tmp_path, capsys):
    ne, n, ec, eps, h, V = fcidump.read(os.path.join(GOLD, "fcidump", "FCIDUMP.LiH.sto6g"))
    out = str(tmp_path / "FCIDUMP.rt")
    fcidump.write(V, h, ne // 2, e_nuc=ec, file=out)
    ne2, n2, ec2, _, h2, V2 = fcidump.read(out)
    assert (ne2, n2, ec2) == (ne, n, ec) and np.array_equal(V2, V) and np.array_equal(h2, h)
