"""Tall-skinny subspace algebra of the Davidson / FEAST drivers (pymes_gram, pymes_lincomb_multi: the device side of
pymes/solver/eom_ccsd.py:91-147, :512-541) against numpy — through the host simulator on the CPU (C-ABI plumbing, chunking),
on the GPU for the kernels."""
import os

import numpy as np
import pytest

from pymes_amd.device import Context, DeviceArray


def _vectors(ctx, rng, count, n, misalign=False):
    """``count`` device vectors of length n; with ``misalign`` they start 8 bytes into their buffers (no 16-byte loads)."""
    host, dev = [], []
    for _ in range(count):
        h = rng.standard_normal(n)
        if misalign:
            buf = ctx.empty((n + 1,))
            d = DeviceArray(ctx, buf.ptr + 8, (n,), owned=False, keepalive=buf)
            d.set(h)
        else:
            d = ctx.array(h)
        host.append(h)
        dev.append(d)
    return host, dev


def check_subspace_algebra(lib, sizes):
    ctx = Context(2, 3, lib=lib)
    rng = np.random.default_rng(7)
    try:
        for (m, n, length, mis) in sizes:
            xh, xd = _vectors(ctx, rng, m, length, mis)
            yh, yd = _vectors(ctx, rng, n, length, mis)
            G = ctx.gram(xd, yd)
            ref = np.array(xh) @ np.array(yh).T if length else np.zeros((m, n))
            assert G.shape == (m, n)
            assert np.abs(G - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()) * max(1, length) ** 0.5, (m, n, length, mis)
            # combinations: fresh outputs (beta = 0, never read: NaN-filled), accumulation (beta != 0)
            Cm = rng.standard_normal((m, n))
            outs = [ctx.array(np.full(length, np.nan)) for _ in range(n)]
            ctx.lincomb_multi(outs, xd, Cm)
            want = (np.array(xh).T @ Cm).T if m else np.zeros((n, length))
            for j in range(n):
                assert np.abs(outs[j].get() - want[j]).max() <= 1e-13 * max(1.0, np.abs(want).max()) if length else True
            beta = rng.standard_normal(n)
            ctx.lincomb_multi(yd, xd, Cm, beta=beta)
            for j in range(n):
                w = want[j] + beta[j] * yh[j]
                assert np.abs(yd[j].get() - w).max() <= 1e-13 * max(1.0, np.abs(w).max()) if length else True
        # in place: the outputs ARE inputs (orthonormalisation of a block by its own triangular factor)
        xh, xd = _vectors(ctx, rng, 3, 1001)
        R = np.triu(rng.standard_normal((3, 3))) + 3.0 * np.eye(3)
        ctx.lincomb_multi(xd, xd, R)
        want = (np.array(xh).T @ R).T
        for j in range(3):
            assert np.abs(xd[j].get() - want[j]).max() < 1e-12
        xh, xd = _vectors(ctx, rng, 17, 64)
        with pytest.raises(Exception):
            ctx.lincomb_multi(xd[:1], xd, np.ones((17, 1)))          # aliasing beyond one launch is refused
        # no inputs at all: a pure scaling
        yh, yd = _vectors(ctx, rng, 2, 77)
        ctx.lincomb_multi(yd, [], np.zeros((0, 2)), beta=[2.0, -1.0])
        assert np.abs(yd[0].get() - 2.0 * yh[0]).max() < 1e-14 and np.abs(yd[1].get() + yh[1]).max() < 1e-14
    finally:
        ctx.close()


SMALL = [(1, 1, 1, False), (3, 2, 17, False), (3, 2, 17, True), (12, 3, 1000, False), (3, 9, 513, False), (16, 4, 300, False),
         (17, 5, 256, False), (5, 17, 255, True), (2, 2, 0, False), (24, 3, 129, False)]


def test_subspace_algebra_host_logic(hostsim_lib):
    check_subspace_algebra(hostsim_lib, SMALL)


@pytest.mark.gpu
def test_subspace_algebra_gpu(gpu_lib):
    check_subspace_algebra(gpu_lib, SMALL + [(12, 12, 300001, False), (9, 3, 1 << 20, False), (3, 9, (1 << 20) + 3, True),
                                             (33, 7, 70001, False)])


def check_grouped_products(lib, shapes, expect_grouped):
    """Independent small products inside ``ctx.gemm_group()`` (pymes_gemm_group_begin / _end) against numpy: every operand
    layout, odd extents and pitches (8-byte loads), k-split of an under-filled group, alpha / beta, more than 16 products
    (two launches), a product too big for a group in the middle (launched on its own, order kept)."""
    ctx = Context(2, 3, lib=lib, workspace_bytes=1 << 28)
    rng = np.random.default_rng(11)
    try:
        # the launch counts asked for are those of the GROUP machinery: with phase launches on (the default) the small
        # products of a group are tasks of the open phase instead (include/pymes_amd.h; expect_grouped None: whatever is on)
        if expect_grouped is not None:
            ctx.phase_enable(0)
        tasks0 = ctx.phase_stats()["tasks"]
        jobs = []
        for (M, N, K, a_kc, b_kc, alpha, beta) in shapes:
            A = rng.standard_normal((M, K) if a_kc else (K, M))
            B = rng.standard_normal((N, K) if b_kc else (K, N))
            Cm = rng.standard_normal((M, N))
            ref = alpha * ((A if a_kc else A.T) @ (B.T if b_kc else B)) + beta * Cm
            jobs.append((M, N, K, a_kc, b_kc, alpha, beta, ctx.array(A), ctx.array(B), ctx.array(Cm), ref))
        with ctx.gemm_group() as grp:
            for (M, N, K, a_kc, b_kc, alpha, beta, dA, dB, dC, ref) in jobs:
                a_sm, a_sk = (K, 1) if a_kc else (1, M)
                b_sk, b_sn = (1, K) if b_kc else (N, 1)
                ctx.dgemm(M, N, K, alpha, dA, a_sm, a_sk, dB, b_sk, b_sn, beta, dC, N)
        for (M, N, K, a_kc, b_kc, alpha, beta, dA, dB, dC, ref) in jobs:
            err = np.abs(dC.get() - ref).max() / max(1.0, np.abs(ref).max())
            assert err < 1e-13 * max(1, K) ** 0.5 + 1e-14, (M, N, K, a_kc, b_kc, alpha, beta, err)
        if expect_grouped is not None:
            assert grp.products == expect_grouped[0] and grp.launches == expect_grouped[1], (grp.products, grp.launches)
            assert ctx.phase_stats()["tasks"] == tasks0
        # an einsum-style contraction that needs a transposed temporary inside a group: it must not wait in the queue
        A, B = rng.standard_normal((5, 6, 7)), rng.standard_normal((7, 5, 4))
        dA, dB = ctx.array(A), ctx.array(B)
        with ctx.gemm_group():
            o1 = ctx.contract("abc,cad->bd", dA, dB)
            o2 = ctx.contract("abc,cad->db", dA, dB)
        assert np.abs(o1.get() - np.einsum("abc,cad->bd", A, B)).max() < 1e-12
        assert np.abs(o2.get() - np.einsum("abc,cad->db", A, B)).max() < 1e-12
    finally:
        ctx.phase_enable(-1)
        ctx.close()


GROUP_SHAPES = ([(M, N, K, a, b, 1.0, 0.0) for (M, N, K) in ((64, 64, 64), (70, 33, 129), (1, 1, 1), (130, 50, 17), (3, 200, 40))
                 for a in (True, False) for b in (True, False)] +
                [(200, 210, 3000, True, False, -0.5, 1.0), (64, 64, 4096, False, True, 2.0, 0.25), (97, 65, 500, True, True, 1.0, 1.0)])


def test_grouped_products_host_logic(hostsim_lib):
    check_grouped_products(hostsim_lib, GROUP_SHAPES, None)


@pytest.mark.gpu
def test_grouped_products_gpu(gpu_lib):
    # the same products as tasks of a phase (the default): numerics only, and that they were recorded, not launched
    ctx = Context(2, 3, lib=gpu_lib)
    t0 = ctx.phase_stats()
    ctx.close()
    check_grouped_products(gpu_lib, GROUP_SHAPES, None)
    if os.environ.get("PYMES_PHASE", "1") != "0":
        ctx = Context(2, 3, lib=gpu_lib)
        t1 = ctx.phase_stats()
        ctx.close()
        assert t1["tasks"] - t0["tasks"] >= len(GROUP_SHAPES) and t1["launches"] > t0["launches"]
    check_grouped_products(gpu_lib, GROUP_SHAPES, (23, 2))
    # a big product between small ones keeps its own (LDS-DMA) launch; the small ones on either side are grouped
    check_grouped_products(gpu_lib, [(100, 90, 80, True, False, 1.0, 0.0), (2048, 2048, 2048, True, False, 1.0, 0.0),
                                     (90, 100, 70, False, False, 1.0, 0.0)], (2, 2))
    # mid-size products in the LDS-DMA layout (A K-contiguous, B N-contiguous, 16-byte loads): the grouped 128 x 128 launch,
    # k-split with one grouped reduction — ragged M / N edges, an odd N under an even pitch is not eligible (64 x 64 group),
    # a partial last k-tile, alpha / beta
    check_grouped_products(gpu_lib, [(3240, 210, 3240, True, False, 1.0, 0.0), (1600, 190, 3160, True, False, -1.0, 1.0),
                                     (130, 200, 1000, True, False, 2.0, 0.5), (100, 90, 80, True, False, 1.0, 0.0),
                                     (500, 129, 2050, True, False, 1.0, 0.0), (129, 66, 777 * 2, True, False, 1.0, 1.0)], (6, 2))
    # an under-filled group of deep products: k-split inside the group, one reduction launch
    check_grouped_products(gpu_lib, [(64, 64, 20000, True, True, 1.0, 0.5), (128, 64, 9999, False, False, 1.0, 0.0),
                                     (60, 60, 7, True, False, 1.0, 0.0)], (3, 1))


def check_dots(lib, lengths_sets):
    """pymes_dots_var: pairs of different lengths in one call, a shared second operand (the DIIS pattern), odd lengths,
    8-byte-aligned operands, chunk counts on both block mappings (a multiple of 8 and not)."""
    ctx = Context(2, 3, lib=lib)
    rng = np.random.default_rng(21)
    try:
        for lengths, shared, mis in lengths_sets:
            xh, xd, yh, yd = [], [], [], []
            ys_h, ys_d = None, None
            for n in lengths:
                h, d = _vectors(ctx, rng, 1, n, mis)
                xh.append(h[0]); xd.append(d[0])
                if shared and ys_h is not None and ys_h.size == n:
                    yh.append(ys_h); yd.append(ys_d)
                else:
                    h, d = _vectors(ctx, rng, 1, n, mis)
                    ys_h, ys_d = h[0], d[0]
                    yh.append(ys_h); yd.append(ys_d)
            got = ctx.dots(xd, yd)
            for g, a, b in zip(got, xh, yh):
                ref = float(np.dot(a, b))
                assert abs(g - ref) <= 1e-13 * max(1.0, np.abs(a).max() * np.abs(b).max() * max(1, a.size) ** 0.5 * 8), (lengths, g, ref)
    finally:
        ctx.close()


DOT_SETS = [((1,), False, False), ((17, 17, 17), True, False), ((1000, 1000, 33, 33), True, False), ((513, 7), False, True),
            ((4097, 4097, 4097, 4097, 4097, 4097, 65, 65, 65, 65, 65, 65), True, False), ((2049, 2049), True, True)]


def test_dots_host_logic(hostsim_lib):
    check_dots(hostsim_lib, DOT_SETS)


@pytest.mark.gpu
def test_dots_gpu(gpu_lib):
    check_dots(gpu_lib, DOT_SETS + [((300001,) * 6 + (1201,) * 6, True, False), ((1 << 20,) * 3, True, False),
                                    (((1 << 20) + 3,) * 5, True, True), ((700000, 11), False, False)])
