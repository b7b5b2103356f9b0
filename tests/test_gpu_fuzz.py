"""GPU: seeded random shapes through the two front ends every contraction of the CC path goes through — the fp64 MFMA
GEMM (tile-choice rules, ragged extents, padded pitches, k-splits, the LDS-DMA kernel from K >= 384, the matrix-vector
kernels) and the einsum-style planner (label classification, kept transposes, batch labels, alpha / beta) — against
numpy.  The reference's seam for both is the module-level ``einsum`` callable (pymes/solver/ccsd.py:11)."""
import numpy as np
import pytest

from pymes_amd.device import Context

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 3, 16, 17, 31, 50, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 257, 300, 383, 384, 385, 500, 777, 1000,
         1275, 1600, 2049, 4097]


@pytest.mark.parametrize("seed", [0, 1])
def test_random_gemm_shapes(gpu_lib, seed):
    rng = np.random.default_rng(seed)
    ctx = Context(4, 4, workspace_bytes=1 << 28, lib=gpu_lib)
    done = 0
    try:
        while done < 30:
            M, N, K = (int(rng.choice(SIZES)) for _ in range(3))
            if rng.random() < 0.15:
                K = int(rng.choice([5000, 20000, 100001]))
                M, N = min(M, 300), min(N, 300)
            if M * N * K > 3e10 or M * K > 4e7 or K * N > 4e7:
                continue
            a_kc, b_kc = bool(rng.integers(2)), bool(rng.integers(2))
            alpha = float(rng.choice([1.0, -0.5, 2.0]))
            beta = float(rng.choice([0.0, 0.0, 1.0, 0.25]))
            pa, pb, pc = (int(rng.choice([0, 0, 1, 2, 3])) for _ in range(3))
            a_rows, a_cols = (M, K) if a_kc else (K, M)
            b_rows, b_cols = (N, K) if b_kc else (K, N)
            A = rng.standard_normal((a_rows, a_cols + pa))
            B = rng.standard_normal((b_rows, b_cols + pb))
            Cm = rng.standard_normal((M, N + pc))
            Am = A[:, :a_cols] if a_kc else A[:, :a_cols].T
            Bm = B[:, :b_cols].T if b_kc else B[:, :b_cols]
            ref = Cm.copy()
            ref[:, :N] = alpha * (Am @ Bm) + beta * Cm[:, :N]
            dA, dB, dC = ctx.array(A), ctx.array(B), ctx.array(Cm)
            a_sm, a_sk = (A.shape[1], 1) if a_kc else (1, A.shape[1])
            b_sk, b_sn = (1, B.shape[1]) if b_kc else (B.shape[1], 1)
            ctx.dgemm(M, N, K, alpha, dA, a_sm, a_sk, dB, b_sk, b_sn, beta, dC, Cm.shape[1])
            got = dC.get()
            err = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
            assert err < 1e-13 * max(1, K) ** 0.5 + 1e-14, dict(M=M, N=N, K=K, a_kc=a_kc, b_kc=b_kc, alpha=alpha, beta=beta,
                                                                 pads=(pa, pb, pc), err=err)
            assert np.array_equal(got[:, N:], Cm[:, N:])              # the pad columns of C are never written
            for x in (dA, dB, dC):
                x.free()
            done += 1
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", [0, 1])
def test_random_contractions(gpu_lib, seed):
    rng = np.random.default_rng(seed)
    ctx = Context(4, 4, workspace_bytes=1 << 28, lib=gpu_lib)
    letters = "abcdefgh"
    done = 0
    try:
        while done < 40:
            nfree_a, nfree_b, nsum = int(rng.integers(0, 3)), int(rng.integers(0, 3)), int(rng.integers(1, 3))
            if nfree_a + nfree_b == 0:
                continue
            labs = list(letters[: nfree_a + nfree_b + nsum])
            rng.shuffle(labs)
            fa, fb, su = labs[:nfree_a], labs[nfree_a:nfree_a + nfree_b], labs[nfree_a + nfree_b:]
            dims = {ch: int(rng.choice([1, 2, 3, 5, 8, 13, 20, 33, 50])) for ch in labs}
            la, lb, lc = fa + su, su + fb, fa + fb
            rng.shuffle(la)
            rng.shuffle(lb)
            rng.shuffle(lc)
            la, lb, lc = "".join(la), "".join(lb), "".join(lc)
            if len(la) > 4 or len(lb) > 4 or len(lc) > 4 or len(lc) == 0:
                continue
            A = rng.standard_normal([dims[c] for c in la])
            B = rng.standard_normal([dims[c] for c in lb])
            C0 = rng.standard_normal([dims[c] for c in lc])
            alpha = float(rng.choice([1.0, -1.0, 0.5, 2.0]))
            beta = float(rng.choice([0.0, 0.0, 1.0, -0.5]))
            ref = alpha * np.einsum(f"{la},{lb}->{lc}", A, B) + beta * C0
            out = ctx.array(C0)
            ctx.contract(f"{la},{lb}->{lc}", ctx.array(A), ctx.array(B), out=out, alpha=alpha, beta=beta)
            err = np.abs(out.get() - ref).max() / max(1.0, np.abs(ref).max())
            assert err < 1e-12, (f"{la},{lb}->{lc}", dims, alpha, beta, err)
            done += 1
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", [0, 1])
def test_random_bra_dressing_shapes(gpu_lib, seed):
    """ladder_dress_kernel (ccsd.py:414-419 on pair-packed rows) on seeded random extents: every MFMA step count (nocc
    1..64), ragged and single tiles, pitches of 1..20 column blocks (more than the 8 XCD shares and fewer), row ranges that
    start and end inside tiles, both halves."""
    from tests.test_gpu_kernels import check_ladder_dress
    rng = np.random.default_rng(100 + seed)
    for case in range(24):
        no = int(rng.integers(1, 65))
        nv = int(rng.choice([1, 2, 7, 15, 16, 17, 23, 31, 32, 33, 40, 48, 57]))
        npp = nv * (nv + 1) // 2
        ld = 16 * int(rng.integers(1, 21))
        r0 = int(rng.integers(0, npp))
        r1 = int(rng.integers(r0 + 1, npp + 1))
        if rng.random() < 0.4:
            r0, r1 = 0, npp
        if no * nv * ld > 4_000_000:
            ld = 16
        check_ladder_dress(gpu_lib, no, nv, ld, r0, r1, bool(rng.integers(2)), seed=1000 * seed + case)
