"""GPU: the distributed CCSD iteration end to end with TWO ranks sharing the one GPU of the test box
(gloo staging instead of RCCL, see pymes_amd/dist.py): slab residual, overlapped all-gathers, pair-sharded tail
(compact R2 / update / DIIS with all-reduced overlaps, all-gather of the new T2) — must reproduce the single-rank
energies and iteration counts."""
import contextlib
import io
import json
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.cases import synthetic_case
        from pymes_amd.integral.device import DeviceIntegrals
        from pymes_amd.solver.ccsd import CCSD
        res = {}
        for no, nv, dcsd in ((6, 20, False), (4, 12, True)):
            f, V, B, eps = synthetic_case(no, nv, seed=0, scale=0.3)
            ints = DeviceIntegrals.from_V_pqrs(no, V, device=0)
            s = CCSD(no, delta_e=1e-10, is_dcsd=dcsd, device=0)
            with contextlib.redirect_stdout(io.StringIO()):
                r = s.solve(f, ints)
            assert s.pair_sharded and s.hooked       # whole library steps, collectives through the table (pymes_collectives)
            ints.ctx.close()
            res[f"syn_{no}_{nv}_{'dcsd' if dcsd else 'ccsd'}"] = (float(r["ccsd e"]), int(s.iterations))
        from pymes_amd.solver.ccd import CCD
        f, V, B, eps = synthetic_case(4, 12, seed=0, scale=0.3)
        c = CCD(4, delta_e=1e-10, device=0)
        with contextlib.redirect_stdout(io.StringIO()):
            r = c.solve(f, V)
        assert c.pair_sharded
        res["syn_4_12_ccd"] = (float(r["ccd e"]), int(c.iterations))
        # unsymmetric user amplitudes take the plain-ladder sharded path
        no, nv = 4, 12
        f, V, B, eps = synthetic_case(no, nv, seed=0, scale=0.3)
        rng = np.random.default_rng(1)
        amps = [rng.standard_normal((nv, no)) * 1e-3, rng.standard_normal((nv, nv, no, no)) * 1e-3]
        with contextlib.redirect_stdout(io.StringIO()):
            r = CCSD(no, delta_e=1e-10, device=0).solve(f, V, amps=amps)
        res["syn_4_12_unsym_start"] = (float(r["ccsd e"]), 0)
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu(gpu_lib):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    gold = json.load(open(os.path.join(GOLD, "solves.json")))
    assert len(out) == 2 and out[0] == out[1]                    # ranks agree bit for bit
    e, it = out[0]["syn_6_20_ccsd"]
    assert abs(e - gold["syn_6_20"]["ccsd"]["e"]) < 1e-9 and it == gold["syn_6_20"]["ccsd"]["iterations"]
    e, it = out[0]["syn_4_12_dcsd"]
    assert abs(e - gold["syn_4_12"]["dcsd"]["e"]) < 1e-9 and it == gold["syn_4_12"]["dcsd"]["iterations"]
    e, it = out[0]["syn_4_12_ccd"]
    assert abs(e - gold["syn_4_12"]["ccd"]["e"]) < 1e-9 and it == gold["syn_4_12"]["ccd"]["iterations"]
    # unsymmetric start: the reference's own fixed-point run (= oracle) is the expectation, not the converged CCSD
    from oracle import cc_oracle as oc
    from oracle.cases import synthetic_case
    f, V, B, eps = synthetic_case(4, 12, seed=0, scale=0.3)
    rng = np.random.default_rng(1)
    amps = [rng.standard_normal((12, 4)) * 1e-3, rng.standard_normal((12, 12, 4, 4)) * 1e-3]
    ref = oc.ccsd_solve(4, f, V, delta_e=1e-10, amps=amps)
    assert abs(out[0]["syn_4_12_unsym_start"][0] - ref["e"]) < 1e-9


def _worker_rccl(rank, port, out):
    """A world of ONE rank on an RCCL communicator (PYMES_FORCE_SHARDED=1): every collective entry point of the table is
    entered with device tensors, asynchronously, ordered on the stream the engine is bound to."""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0", PYMES_FORCE_SHARDED="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        from oracle.cases import synthetic_case
        from pymes_amd import dist as pdist
        from pymes_amd.integral.device import DeviceIntegrals
        from pymes_amd.solver.ccsd import CCSD
        assert pdist.sharded()
        res = {}
        for no, nv, dcsd in ((6, 20, False), (4, 12, True)):
            f, V, B, eps = synthetic_case(no, nv, seed=0, scale=0.3)
            ints = DeviceIntegrals.from_V_pqrs(no, V, device=0)
            s = CCSD(no, delta_e=1e-10, is_dcsd=dcsd, device=0)
            with contextlib.redirect_stdout(io.StringIO()):
                r = s.solve(f, ints)
            assert s.pair_sharded and s.hooked
            res[f"syn_{no}_{nv}_{'dcsd' if dcsd else 'ccsd'}"] = (float(r["ccsd e"]), int(s.iterations), int(s.collective_calls))
            ints.ctx.close()
        # the all-gathers are IN PLACE (sendbuff = recvbuff + rank * count); from a separate send buffer (PYMES_ALLGATHER_CLONE=1),
        # with the owner-tile all-to-all (ncclSend / ncclRecv through torch's all_to_all_single) and for CCD (the whole steps
        # of round 6: pymes_ccd_sharded_residuals + finish / energy / await): the same numbers
        from pymes_amd.solver.ccd import CCD
        f, V, B, eps = synthetic_case(4, 12, seed=0, scale=0.3)
        for tag, env, kind in (("clone", {"PYMES_ALLGATHER_CLONE": "1"}, "ccsd"), ("tiles", {"PYMES_OWNER_TILES": "1"}, "ccsd"),
                               ("ccd", {}, "ccd"), ("ccd_tiles", {"PYMES_OWNER_TILES": "1"}, "ccd")):
            os.environ.update(env)
            try:
                s = CCD(4, delta_e=1e-10, device=0) if kind == "ccd" else CCSD(4, delta_e=1e-10, is_dcsd=True, device=0)
                with contextlib.redirect_stdout(io.StringIO()):
                    r = s.solve(f, V)
                assert s.pair_sharded and s.hooked
                res["x_" + tag] = (float(r["ccd e" if kind == "ccd" else "ccsd e"]), int(s.iterations))
            finally:
                for k in env:
                    del os.environ[k]
        out[0] = res
    finally:
        dist.destroy_process_group()


def test_forced_one_rank_rccl(gpu_lib):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_rccl, args=(port, out), nprocs=1, join=True)
    gold = json.load(open(os.path.join(GOLD, "solves.json")))
    e, it, calls = out[0]["syn_6_20_ccsd"]
    assert abs(e - gold["syn_6_20"]["ccsd"]["e"]) < 1e-9 and it == gold["syn_6_20"]["ccsd"]["iterations"]
    assert calls >= 10 * it                  # ten collectives per residual build + finish
    e, it, calls = out[0]["syn_4_12_dcsd"]
    assert abs(e - gold["syn_4_12"]["dcsd"]["e"]) < 1e-9 and it == gold["syn_4_12"]["dcsd"]["iterations"]
    for tag in ("clone", "tiles"):
        assert out[0]["x_" + tag] == (e, it), (tag, out[0]["x_" + tag], e, it)
    for tag in ("ccd", "ccd_tiles"):
        e2, it2 = out[0]["x_" + tag]
        assert abs(e2 - gold["syn_4_12"]["ccd"]["e"]) < 1e-9 and it2 == gold["syn_4_12"]["ccd"]["iterations"]
