"""The C-ABI from a compiled host: tests/capi/host_ccsd.c is plain C99 on include/pymes_amd.h alone (no Python, no torch,
no C++) and runs the CCSD / DCSD fixed point of ccsd.py:159-209 from a packed factor file — through pymes_ccsd_iterate, and
through the one-process-per-GPU steps with a collective table it fills itself (a world of one rank).

CPU: the header is valid pedantic C99 and the program compiles and links against the product library's symbols.
GPU: it runs; every pass agrees with the Python host on the same library and the final pass with the oracle (= the
reference's algebra); the table is called in the documented pattern."""
import os
import subprocess

import numpy as np
import pytest

from oracle.cases import synthetic_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "capi", "host_ccsd.c")
LIBDIR = os.path.join(ROOT, "pymes_amd", "lib")


ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def build(out, rccl=False):
    # (-rpath: the program finds the in-tree library; the HIP runtime it needs is found through the library's own RUNPATH /
    # the system's ROCm installation; unresolved symbols of that runtime do not concern the host program)
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), SRC,
           "-L", LIBDIR, "-lpymes_amd", "-Wl,-rpath," + LIBDIR, "-Wl,--allow-shlib-undefined", "-o", out]
    if rccl:     # the RCCL table of INTEGRATION.md 2b: HIP runtime API + RCCL from the host program itself
        # (HIP's and RCCL's own headers are not pedantic C99: system headers, not ours)
        cmd += ["-DWITH_RCCL", "-D__HIP_PLATFORM_AMD__", "-isystem", os.path.join(ROCM, "include"), "-L", os.path.join(ROCM, "lib"),
                "-lrccl", "-lamdhip64", "-Wl,-rpath," + os.path.join(ROCM, "lib")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_header_is_c99_and_host_program_links(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, "libpymes_amd.so")):
        pytest.skip("product library not built")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c",
                        os.path.join(ROOT, "include", "pymes_amd.h")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    exe = build(str(tmp_path / "host_ccsd"))
    if os.path.exists(os.path.join(ROCM, "include", "rccl", "rccl.h")):
        build(str(tmp_path / "host_ccsd_rccl"), rccl=True)              # the RCCL table compiles and links too
    # every pymes_* symbol the program uses is one the header declares and the library exports (no GPU here: not run)
    used = subprocess.run(["nm", "-u", exe], capture_output=True, text=True).stdout
    used = {ln.split()[-1].split("@")[0] for ln in used.splitlines() if " pymes_" in ln or ln.strip().startswith("U pymes_")}
    assert {"pymes_set_collectives", "pymes_ccsd_sharded_residuals", "pymes_ccsd_iterate", "pymes_packed_load"} <= used
    exported = subprocess.run(["nm", "-D", "--defined-only", os.path.join(LIBDIR, "libpymes_amd.so")], capture_output=True,
                              text=True).stdout
    exported = {ln.split()[-1] for ln in exported.splitlines() if " T pymes_" in ln}
    assert used <= exported, used - exported


def lines_of(text):
    out = {}
    for ln in text.splitlines():
        w = ln.split()
        if w[0] == "pass":
            out[int(w[1])] = tuple(float(x) for x in w[2:])
        elif w[0] in ("mp2", "t2_norm2"):
            out[w[0]] = float(w[1])
        elif w[0] == "collectives":
            out["collectives"] = (int(w[2]), int(w[4]), int(w[6]))
            out["alltoallv"] = int(w[8])
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("dcsd", [False, True])
def test_c_host_runs_the_fixed_point(gpu_lib, tmp_path, dcsd):
    from oracle import cc_oracle as oc
    from pymes_amd.device import Context
    from pymes_amd.util import packed
    no, nv, passes = 6, 20, 4
    f, V, B, eps = synthetic_case(no, nv, seed=0, scale=0.3)
    path = str(tmp_path / "factors.pk")
    packed.write_factors(path, 2 * no, 0.0, eps, np.diag(eps), B)
    exe = build(str(tmp_path / "host_ccsd_rccl"), rccl=True)
    runs = []
    for mode in (0, 1, 2, 3):
        r = subprocess.run([exe, path, str(passes), str(int(dcsd)), str(mode)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr + r.stdout
        runs.append(lines_of(r.stdout))
    plain, hooked, rccl, tiles = runs
    # mode 3: the owner-tile exchange through the table's all-to-all (ncclSend / ncclRecv in a group; one call per pass, no
    # all-gather of ETd / ETx): the same numbers
    assert tiles.pop("alltoallv") == passes and rccl.pop("alltoallv") == 0 and hooked.pop("alltoallv") == 0
    assert tiles["collectives"][1] == rccl["collectives"][1] - 2 * passes
    assert {k: v for k, v in tiles.items() if k != "collectives"} == {k: v for k, v in rccl.items() if k != "collectives"}
    # the RCCL table (real ncclAllReduce / ncclAllGather on a communicator of one rank, event-ordered against the library's
    # stream) gives the numbers of the table that exchanges nothing, bit for bit
    assert rccl == hooked
    # the Python host on the same library, same sequence
    ctx = Context(no, nv, lib=gpu_lib)
    try:
        ctx.set_V_from_factors(B)
        ctx.set_orbital_energies(eps[:no].copy(), eps[no:].copy())
        fdev = ctx.array(np.diag(eps))
        t1, t2 = ctx.zeros((nv, no)), ctx.empty((nv, nv, no, no))
        e_mp2 = sum(ctx.mp2(t2, 0.0))
        dt1, dt2 = ctx.empty(t1.shape), ctx.empty(t2.shape)
        assert abs(plain["mp2"] - e_mp2) < 1e-13 and abs(hooked["mp2"] - e_mp2) < 1e-13
        for it in range(passes):
            out = ctx.ccsd_iterate(fdev, t1, t2, dt1, dt2, is_dcd=dcsd, t1_zero=(it == 0))
            e = out[0] + out[1] + out[2]
            for run in (plain, hooked):
                got = run[it + 1]
                assert abs(got[0] - e) < 1e-12 and abs(got[1] - out[3]) < 1e-12 and abs(got[2] - out[4]) < 1e-12, (it, got, out)
        norm2 = float(np.vdot(t2.get(), t2.get()))
        assert abs(plain["t2_norm2"] - norm2) < 1e-12 and abs(hooked["t2_norm2"] - norm2) < 1e-12
        ctx.ccsd_release()
    finally:
        ctx.close()
    ref = oc.ccsd_solve(no, f, V, is_dcsd=dcsd, is_diis=False, delta_e=1e-30, max_iter=passes - 1)
    assert abs(plain[passes][0] - ref["e"]) < 1e-12 and abs(hooked[passes][0] - ref["e"]) < 1e-12
    # per pass: six all-reduces (W, X_ki, J, X_ac, R1, the six sums), four all-gathers (ETd, ETx, QK, the new T2), ten waits
    assert hooked["collectives"] == (6 * passes, 4 * passes, 10 * passes)
