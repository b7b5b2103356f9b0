"""GPU: the launch forms of the LDS-DMA fp64 GEMM (kernels.hip: plan_dma) against numpy on the same products — whole tiles,
whole tiles + a k-cut tail in one grid (ks-major cut blocks) for several (whole, cuts), for every operand layout, ragged
edges, batches, alpha / beta and an in-place beta term.  The contractions these
launches carry are the ladder ccd.py:187 and the ring products ccd.py:190-240."""
import os

import numpy as np
import pytest

from pymes_amd.device import Context

pytestmark = pytest.mark.gpu

# (M, N, K): 128 x 128 tiles, K >= 384 so that the LDS-DMA kernel is taken
SHAPES = [(1250, 1100, 900), (384, 640, 2000), (777, 300, 1601), (129, 129, 4100), (2000, 1600, 400)]
PLANS = [None, "0,1", "0,2", "0,3", "8,2", "16,3", "64,2", "88,2", "200,1", "8,1"]


def run_case(ctx, rng, M, N, K, a_kc, b_kc, alpha, beta, plan, nb=1):
    A = rng.standard_normal((nb, M, K) if a_kc else (nb, K, M))
    B = rng.standard_normal((nb, N, K) if b_kc else (nb, K, N))
    C0 = rng.standard_normal((nb, M, N))
    Am = A if a_kc else A.transpose(0, 2, 1)
    Bm = B.transpose(0, 2, 1) if b_kc else B
    ref = alpha * np.matmul(Am, Bm) + beta * C0
    dA, dB, dC = ctx.array(A), ctx.array(B), ctx.array(C0)
    if plan is None:
        os.environ.pop("PYMES_GEMM_PLAN", None)
    else:
        os.environ["PYMES_GEMM_PLAN"] = plan
    try:
        la = "zmk" if a_kc else "zkm"
        lb = "znk" if b_kc else "zkn"
        if nb == 1:
            ctx.contract(f"{la[1:]},{lb[1:]}->mn", dA.reshape(*A.shape[1:]), dB.reshape(*B.shape[1:]),
                         out=dC.reshape(M, N), alpha=alpha, beta=beta)
        else:
            ctx.contract(f"{la},{lb}->zmn", dA, dB, out=dC, alpha=alpha, beta=beta, batch="z")
    finally:
        os.environ.pop("PYMES_GEMM_PLAN", None)
    got = dC.get()
    err = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
    for x in (dA, dB, dC):
        x.free()
    return err


@pytest.mark.parametrize("shape", SHAPES)
def test_launch_plans_agree_with_numpy(gpu_lib, shape):
    M, N, K = shape
    rng = np.random.default_rng(M + 7 * N + 13 * K)
    ctx = Context(4, 4, workspace_bytes=1 << 28, lib=gpu_lib)
    try:
        for plan in PLANS:
            a_kc, b_kc = bool(rng.integers(2)), bool(rng.integers(2))
            alpha = float(rng.choice([1.0, -0.5, 2.0]))
            beta = float(rng.choice([0.0, 1.0, 0.25]))
            err = run_case(ctx, rng, M, N, K, a_kc, b_kc, alpha, beta, plan)
            assert err < 1e-13 * K ** 0.5, dict(shape=shape, plan=plan, a_kc=a_kc, b_kc=b_kc, alpha=alpha, beta=beta, err=err)
    finally:
        ctx.close()


def test_mixed_launch_all_layouts_and_batches(gpu_lib):
    rng = np.random.default_rng(5)
    ctx = Context(4, 4, workspace_bytes=1 << 28, lib=gpu_lib)
    try:
        for a_kc in (False, True):
            for b_kc in (False, True):
                for plan in ("40,2", "8,2", "0,2", None):
                    err = run_case(ctx, rng, 650, 520, 1000, a_kc, b_kc, 1.0, 1.0, plan, nb=3)       # 30 tiles x 3 batches
                    assert err < 1e-11, dict(a_kc=a_kc, b_kc=b_kc, plan=plan, err=err)
    finally:
        ctx.close()


def test_plans_are_deterministic(gpu_lib):
    """The same launch twice gives the same bits (fixed summation order of the k cuts of a tile), whatever the order in which
    the blocks ran."""
    rng = np.random.default_rng(11)
    ctx = Context(4, 4, workspace_bytes=1 << 28, lib=gpu_lib)
    try:
        A, B = rng.standard_normal((900, 2600)), rng.standard_normal((2600, 1000))
        dA, dB = ctx.array(A), ctx.array(B)
        for plan in ("32,3", "0,2", "0,3", None):
            outs = []
            for _ in range(3):
                if plan is None:
                    os.environ.pop("PYMES_GEMM_PLAN", None)
                else:
                    os.environ["PYMES_GEMM_PLAN"] = plan
                try:
                    outs.append(ctx.contract("mk,kn->mn", dA, dB).get())
                finally:
                    os.environ.pop("PYMES_GEMM_PLAN", None)
            assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]), plan
            assert np.abs(outs[0] - A @ B).max() < 1e-10
    finally:
        ctx.close()
