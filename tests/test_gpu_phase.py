"""GPU: phase launches (include/pymes_amd.h) — the small operations of the library recorded with their address
ranges and launched level by level of their hazard graph, one union grid per level.  The whole parity suite runs with them
on (the default); here: the same solve with phases on, off and serialised (one task per level) agrees; the environment
switches do what INTEGRATION.md 4 says; hand-made hazards — column slices of one pitched array, accumulation chains that are
fused, a chain that must NOT be fused because somebody reads the array in between — against numpy."""
import contextlib
import io
import os

import numpy as np
import pytest

from oracle.cases import synthetic_case
from pymes_amd import _lib
from pymes_amd.device import Context

pytestmark = pytest.mark.gpu


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


@contextlib.contextmanager
def phase_env(ctx, **env):
    """Environment switches are read when the phase machinery is (re)armed: pymes_phase_enable(-1)."""
    old = {k: os.environ.get(k) for k in env}
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    ctx.phase_enable(-1)
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        ctx.phase_enable(-1)


def solve(lib, monkeypatch, no, nv, dcsd=False):
    from pymes_amd.solver.ccsd import CCSD
    monkeypatch.setattr(_lib, "_default", lib)
    f, V, _, _ = synthetic_case(no, nv, seed=0, scale=0.3)
    s = CCSD(no, delta_e=1e-10, is_dcsd=dcsd)
    r = quiet(s.solve, f, V)
    return r["ccsd e"], s.iterations, r["t2"]


def test_on_off_serial_agree(gpu_lib, monkeypatch):
    ctx = Context(2, 3, lib=gpu_lib)
    try:
        res = {}
        for mode, env in (("on", {"PYMES_PHASE": None}), ("off", {"PYMES_PHASE": "0"}), ("serial", {"PYMES_PHASE": "serial"}),
                          ("nofuse", {"PYMES_PHASE_FUSE_MB": "0"}), ("big", {"PYMES_PHASE_MAX_US": "100000", "PYMES_PHASE_FUSE_MB": "64"})):
            with phase_env(ctx, **env):
                before = ctx.phase_stats()
                res[mode] = solve(gpu_lib, monkeypatch, 8, 32) + (ctx.phase_stats()["tasks"] - before["tasks"],
                                                                   ctx.phase_stats()["launches"] - before["launches"])
        e, it, t2 = res["on"][:3]
        assert res["on"][3] > 100 and res["on"][4] < res["on"][3]              # recorded, and fewer grids than tasks
        assert res["off"][3] == 0                                             # PYMES_PHASE=0: nothing is recorded
        assert res["serial"][4] >= res["serial"][3] > 0                       # one grid per task
        for mode in ("off", "serial", "nofuse", "big"):
            assert abs(res[mode][0] - e) < 1e-12 and res[mode][1] == it, (mode, res[mode][:2], e, it)
            assert np.abs(res[mode][2] - t2).max() < 1e-11
        # serialised and unfused differ from immediate launches only in WHERE the launches are cut: bit-identical
        assert res["serial"][0] == res["off"][0] and np.array_equal(res["serial"][2], res["off"][2])
    finally:
        ctx.close()


def test_switches(gpu_lib, monkeypatch, capfd):
    ctx = Context(2, 3, lib=gpu_lib)
    try:
        with phase_env(ctx, PYMES_PHASE_MAX_US="0"):                          # nothing is "small": nothing recorded
            before = ctx.phase_stats()["tasks"]
            e0 = solve(gpu_lib, monkeypatch, 4, 12)[0]
            # (only the unconditional tasks of the reductions' last stages are left)
            assert ctx.phase_stats()["tasks"] - before < 200
        capfd.readouterr()
        with phase_env(ctx, PYMES_PHASE_LOG="1"):
            e1 = solve(gpu_lib, monkeypatch, 4, 12)[0]
        err = capfd.readouterr().err
        assert "[phase] flush" in err and " gemm:" in err and "L0" in err
        assert abs(e0 - e1) < 1e-12
    finally:
        ctx.close()


def test_hand_made_hazards(gpu_lib):
    """Operations of several calls in ONE held phase (pymes_phase_hold): what the hazard analysis may run side by side and
    what it must keep apart."""
    ctx = Context(2, 3, lib=gpu_lib, workspace_bytes=1 << 28)
    rng = np.random.default_rng(5)
    try:
        M, K, N1, N2 = 150, 70, 40, 24
        A, B1, B2 = rng.standard_normal((M, K)), rng.standard_normal((K, N1)), rng.standard_normal((K, N2))
        dA, dB1, dB2 = ctx.array(A), ctx.array(B1), ctx.array(B2)
        # (1) two products into disjoint column slices of the same rows (one pitched array): independent — one level
        C = ctx.zeros((M, N1 + N2))
        from pymes_amd.device import DeviceArray
        right = DeviceArray(ctx, C.ptr + 8 * N1, (M, N2), owned=False, keepalive=C)
        before = ctx.phase_stats()
        with ctx.phase_hold():
            ctx.dgemm(M, N1, K, 1.0, dA, K, 1, dB1, N1, 1, 0.0, C, N1 + N2)
            ctx.dgemm(M, N2, K, 1.0, dA, K, 1, dB2, N2, 1, 0.0, right, N1 + N2)
        after = ctx.phase_stats()
        assert after["tasks"] - before["tasks"] == 2 and after["levels"] - before["levels"] == 1
        assert np.abs(C.get() - np.hstack([A @ B1, A @ B2])).max() < 1e-12
        # (2) an accumulation chain into a small contiguous array: fused — the members side by side, one combining task
        X = [rng.standard_normal((M, K)) for _ in range(5)]
        Y = [rng.standard_normal((K, N1)) for _ in range(5)]
        dX, dY = [ctx.array(x) for x in X], [ctx.array(y) for y in Y]
        R0 = rng.standard_normal((M, N1))
        R = ctx.array(R0)
        before = ctx.phase_stats()
        with ctx.phase_hold():
            for x, y in zip(dX, dY):
                ctx.dgemm(M, N1, K, 0.5, x, K, 1, y, N1, 1, 1.0, R, N1)
        after = ctx.phase_stats()
        assert after["levels"] - before["levels"] == 2 and after["launches"] - before["launches"] == 2
        want = R0 + 0.5 * sum(x @ y for x, y in zip(X, Y))
        assert np.abs(R.get() - want).max() < 1e-12
        # (3) ... but not across a reader of the array: R2 = R @ Z must see R after the first two members only
        R = ctx.array(R0)
        Z = rng.standard_normal((N1, 30))
        dZ, R2 = ctx.array(Z), ctx.zeros((M, 30))
        with ctx.phase_hold():
            ctx.dgemm(M, N1, K, 1.0, dX[0], K, 1, dY[0], N1, 1, 1.0, R, N1)
            ctx.dgemm(M, N1, K, 1.0, dX[1], K, 1, dY[1], N1, 1, 1.0, R, N1)
            ctx.dgemm(M, 30, N1, 1.0, R, N1, 1, dZ, 30, 1, 0.0, R2, 30)
            ctx.dgemm(M, N1, K, 1.0, dX[2], K, 1, dY[2], N1, 1, 1.0, R, N1)
        mid = R0 + X[0] @ Y[0] + X[1] @ Y[1]
        assert np.abs(R2.get() - mid @ Z).max() < 1e-11
        assert np.abs(R.get() - (mid + X[2] @ Y[2])).max() < 1e-12
        # (4) a temporary that is overwritten while an earlier reader is still only recorded (write-after-read)
        T = ctx.array(A)
        out1, out2 = ctx.zeros((M, N1)), ctx.zeros((M, N1))
        with ctx.phase_hold():
            ctx.dgemm(M, N1, K, 1.0, T, K, 1, dB1, N1, 1, 0.0, out1, N1)
            T.copy_from(dX[3])                                   # (a copy is not recorded: the phase is launched first)
            ctx.dgemm(M, N1, K, 1.0, T, K, 1, dB1, N1, 1, 0.0, out2, N1)
        assert np.abs(out1.get() - A @ B1).max() < 1e-12 and np.abs(out2.get() - X[3] @ B1).max() < 1e-12
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", list(range(8)))
def test_random_overlapping_views(gpu_lib, seed):
    """Forty products C <- alpha A B + beta C in ONE held phase on random pitched 2-D views of one buffer that overlap each
    other at random (row slices, column slices of the same rows, partial overlaps; A or B may be the transposed reading of
    a view) against the same sequence in numpy, one after the other: every read-after-write, write-after-read and write-
    after-write pair the generator happens to produce must be found by the range analysis (pitch-aware boxes, accumulation
    fusion included: beta = 1 chains into contiguous outputs occur)."""
    from pymes_amd.device import DeviceArray
    ctx = Context(2, 3, lib=gpu_lib, workspace_bytes=1 << 28)
    rng = np.random.default_rng(100 + seed)
    try:
        total = 1 << 16
        host = rng.standard_normal(total)
        buf = ctx.array(host.copy())

        def view(rows, cols):
            """(offset, pitch) of a rows x cols box somewhere in the buffer; pitch == cols now and then (contiguous)."""
            pitch = cols if rng.random() < 0.4 else cols + int(rng.integers(1, 40))
            span = (rows - 1) * pitch + cols
            # a handful of anchor offsets, so that boxes collide often
            off = int(rng.choice([0, 64, 1000, 1024, 5000, 5003, 20000, 20016])) + int(rng.integers(0, 8)) * 2
            assert off + span <= total
            return off, pitch

        def np_box(off, rows, cols, pitch):
            idx = off + np.arange(rows)[:, None] * pitch + np.arange(cols)[None, :]
            return idx

        def dev(off):
            return DeviceArray(ctx, buf.ptr + 8 * off, (1,), owned=False, keepalive=buf)

        def disjoint(i1, i2):
            return not np.intersect1d(i1.ravel(), i2.ravel()).size

        ops, done = [], 0
        before = ctx.phase_stats()
        with ctx.phase_hold():
            while done < 40:
                M, N, K = int(rng.integers(3, 70)), int(rng.integers(3, 70)), int(rng.integers(3, 50))
                ta, tb = rng.random() < 0.3, rng.random() < 0.3
                (oa, pa), (ob, pb), (oc, pc) = view(*((K, M) if ta else (M, K))), view(*((N, K) if tb else (K, N))), view(M, N)
                ia = np_box(oa, *((K, M) if ta else (M, K)), pa)
                ib = np_box(ob, *((N, K) if tb else (K, N)), pb)
                ic = np_box(oc, M, N, pc)
                if not (disjoint(ia, ic) and disjoint(ib, ic)):
                    continue                                # an output aliasing its own operands is undefined in any BLAS
                A = host[ia].T if ta else host[ia]
                B = host[ib].T if tb else host[ib]
                # (scaled by the operands' size in the host model, so that forty chained products stay O(1))
                alpha = float(rng.choice([-1.0, 0.5, 1.0])) / (np.sqrt(K) * max(1.0, np.abs(A).max()) * max(1.0, np.abs(B).max()))
                beta = float(rng.choice([0.0, 1.0, 1.0, 0.5]))
                ctx.dgemm(M, N, K, alpha, dev(oa), *((1, pa) if ta else (pa, 1)), dev(ob), *((1, pb) if tb else (pb, 1)),
                          beta, dev(oc), pc)
                host[ic] = alpha * (A @ B) + beta * host[ic]
                done += 1
        after = ctx.phase_stats()
        assert after["tasks"] - before["tasks"] >= 40 and after["levels"] - before["levels"] >= 2
        got = buf.get()
        assert np.isfinite(host).all() and np.abs(got - host).max() < 1e-11 * max(1.0, np.abs(host).max()), np.abs(got - host).max()
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", list(range(6)))
def test_random_mixed_operations(gpu_lib, seed):
    """The other recorded kinds next to the products: index permutations of 3- and 4-index arrays (both permutation kernels),
    binary contractions that end in matrix-vector kernels, linear combinations, and overlaps read back in between —
    fifty operations on contiguous arrays that overlap at random inside one buffer, in ONE held phase (an overlap read-back
    launches what is recorded and the phase goes on), against the same sequence in numpy."""
    from pymes_amd.device import DeviceArray
    ctx = Context(3, 5, lib=gpu_lib, workspace_bytes=1 << 28)
    rng = np.random.default_rng(500 + seed)
    try:
        total = 1 << 15
        host = rng.standard_normal(total)
        buf = ctx.array(host.copy())
        anchors = [0, 96, 2000, 2048, 9000, 9008, 16000]

        def place(n):
            off = int(rng.choice(anchors)) + 2 * int(rng.integers(0, 6))
            assert off + n <= total
            return off

        def dev(off, shape):
            return DeviceArray(ctx, buf.ptr + 8 * off, tuple(shape), owned=False, keepalive=buf)

        def hv(off, shape):
            n = int(np.prod(shape))
            return host[off:off + n].reshape(shape)

        def apart(o1, n1, o2, n2):
            return o1 + n1 <= o2 or o2 + n2 <= o1

        done, dots_seen = 0, 0
        with ctx.phase_hold():
            while done < 50:
                kind = rng.choice(["perm3", "perm4", "gemv", "lincomb", "dots", "gemm"])
                if kind in ("perm3", "perm4"):
                    nd = 3 if kind == "perm3" else 4
                    shape = [int(rng.integers(2, 13)) for _ in range(nd)]
                    perm = list(rng.permutation(nd))
                    n = int(np.prod(shape))
                    oi, oo = place(n), place(n)
                    if not apart(oi, n, oo, n):
                        continue
                    li = "abcd"[:nd]
                    lo = "".join(li[p] for p in perm)
                    alpha, beta = float(rng.choice([1.0, -0.5])), float(rng.choice([0.0, 1.0]))
                    out_shape = [shape[p] for p in perm]
                    res = alpha * hv(oi, shape).transpose(perm) + beta * hv(oo, out_shape)
                    ctx.permute(f"{li}->{lo}", dev(oi, shape), out=dev(oo, out_shape), alpha=alpha, beta=beta)
                    host[oo:oo + n] = res.ravel()
                elif kind == "gemv":
                    M, K = int(rng.integers(4, 120)), int(rng.integers(4, 120))
                    oa, ox, oy = place(M * K), place(K), place(M)
                    if not (apart(oa, M * K, oy, M) and apart(ox, K, oy, M)):
                        continue
                    s = 1.0 / (np.sqrt(K) * max(1.0, np.abs(hv(oa, (M, K))).max()))
                    res = s * (hv(oa, (M, K)) @ hv(ox, (K,)))
                    ctx.contract("mk,k->m", dev(oa, (M, K)), dev(ox, (K,)), out=dev(oy, (M,)), alpha=s, beta=0.0)
                    host[oy:oy + M] = res
                elif kind == "lincomb":
                    n, m = int(rng.integers(10, 1500)), int(rng.integers(1, 5))
                    oo = place(n)
                    offs = [place(n) for _ in range(m)]
                    if not all(apart(o, n, oo, n) for o in offs):
                        continue
                    cs = [float(c) for c in rng.uniform(-0.7, 0.7, m)]
                    res = sum(c * hv(o, (n,)) for c, o in zip(cs, offs))
                    ctx.lincomb(dev(oo, (n,)), [dev(o, (n,)) for o in offs], cs)
                    host[oo:oo + n] = res
                elif kind == "dots":
                    n = int(rng.integers(10, 3000))
                    ox, oy = place(n), place(n)
                    got = ctx.dots([dev(ox, (n,))], [dev(oy, (n,))])[0]
                    want = float(hv(ox, (n,)) @ hv(oy, (n,)))
                    assert abs(got - want) < 1e-10 * max(1.0, abs(want)), (done, got, want)
                    dots_seen += 1
                else:
                    M, N, K = int(rng.integers(3, 60)), int(rng.integers(3, 60)), int(rng.integers(3, 40))
                    oa, ob, oc = place(M * K), place(K * N), place(M * N)
                    if not (apart(oa, M * K, oc, M * N) and apart(ob, K * N, oc, M * N)):
                        continue
                    A, B = hv(oa, (M, K)), hv(ob, (K, N))
                    alpha = 1.0 / (np.sqrt(K) * max(1.0, np.abs(A).max()) * max(1.0, np.abs(B).max()))
                    beta = float(rng.choice([0.0, 1.0]))
                    res = alpha * (A @ B) + beta * hv(oc, (M, N))
                    ctx.dgemm(M, N, K, alpha, dev(oa, (1,)), K, 1, dev(ob, (1,)), N, 1, beta, dev(oc, (1,)), N)
                    host[oc:oc + M * N] = res.ravel()
                done += 1
        got = buf.get()
        assert np.isfinite(host).all() and np.abs(got - host).max() < 1e-11 * max(1.0, np.abs(host).max()), np.abs(got - host).max()
    finally:
        ctx.close()
