"""Kernel-level parity on the GPU: fp64 MFMA GEMM, permutation, element-wise and
reduction kernels against numpy, through the C-ABI."""
import numpy as np
import pytest

from pymes_amd.device import Context

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(gpu_lib):
    c = Context(4, 12, lib=gpu_lib, workspace_bytes=1 << 30)
    yield c
    c.close()


def _gemm_case(ctx, M, N, K, a_kc, b_kc, alpha, beta, rng, lda_pad=0, ldb_pad=0, ldc_pad=0):
    a_rows, a_cols = (M, K) if a_kc else (K, M)
    b_rows, b_cols = (N, K) if b_kc else (K, N)
    A = rng.standard_normal((a_rows, a_cols + lda_pad))
    B = rng.standard_normal((b_rows, b_cols + ldb_pad))
    Cm = rng.standard_normal((M, N + ldc_pad))
    Am = A[:, :a_cols] if a_kc else A[:, :a_cols].T          # logical [M,K]
    Bm = B[:, :b_cols].T if b_kc else B[:, :b_cols]          # logical [K,N]
    ref = Cm.copy()
    ref[:, :N] = alpha * (Am @ Bm) + beta * Cm[:, :N]
    dA, dB, dC = ctx.array(A), ctx.array(B), ctx.array(Cm)
    a_sm, a_sk = (A.shape[1], 1) if a_kc else (1, A.shape[1])
    b_sk, b_sn = (1, B.shape[1]) if b_kc else (B.shape[1], 1)
    ctx.dgemm(M, N, K, alpha, dA, a_sm, a_sk, dB, b_sk, b_sn, beta, dC, Cm.shape[1])
    got = dC.get()
    scale = max(1.0, np.abs(ref).max())
    err = np.abs(got - ref).max() / scale
    assert err < 1e-13 * max(1, K) ** 0.5 + 1e-14, (M, N, K, a_kc, b_kc, alpha, beta, err)


@pytest.mark.parametrize("a_kc", [True, False])
@pytest.mark.parametrize("b_kc", [True, False])
def test_dgemm_layouts_and_edges(ctx, a_kc, b_kc):
    rng = np.random.default_rng(1)
    shapes = [(128, 128, 64), (256, 384, 160), (130, 70, 50), (1, 1, 1), (3, 5, 7), (64, 64, 16), (200, 200, 33),
              (17, 300, 129), (300, 17, 2), (129, 129, 129), (50, 2500, 50), (512, 64, 1000), (1, 200, 300),
              (200, 1, 300), (96, 40, 4096)]
    for (M, N, K) in shapes:
        _gemm_case(ctx, M, N, K, a_kc, b_kc, 1.0, 0.0, rng)
    _gemm_case(ctx, 150, 90, 77, a_kc, b_kc, -0.5, 1.0, rng)
    _gemm_case(ctx, 150, 90, 78, a_kc, b_kc, 2.0, 0.25, rng, lda_pad=2, ldb_pad=4, ldc_pad=6)
    _gemm_case(ctx, 151, 91, 78, a_kc, b_kc, 2.0, 0.25, rng, lda_pad=1, ldb_pad=3, ldc_pad=5)


@pytest.mark.parametrize("a_kc", [True, False])
@pytest.mark.parametrize("b_kc", [True, False])
def test_dgemm_lds_dma_kernel(ctx, a_kc, b_kc):
    """Shapes that reach the LDS-DMA 128x128 kernel (>= 256 tiles, K >= 1024, 16-byte aligned operands): ragged M/N
    edges, a partial last k-tile, k-split remainder tiles, alpha/beta, padded pitches, and odd M / N with an even
    pitch (pair loads one element past the extent)."""
    rng = np.random.default_rng(5)
    ctx.prof_enable(True)
    ctx.prof_reset()
    _gemm_case(ctx, 2100, 2180, 1100, a_kc, b_kc, 1.0, 0.0, rng)                     # 17 x 18 tiles, k-tail of 12
    dma = ctx.prof_query(kernel_class=1)
    ctx.prof_enable(False)
    assert dma["launches"] == 1 and dma["kernel_launches"] >= 1, "the LDS-DMA kernel was not selected"
    _gemm_case(ctx, 2100, 2180, 1104, a_kc, b_kc, -0.5, 0.75, rng, lda_pad=2, ldb_pad=6, ldc_pad=3)
    _gemm_case(ctx, 2049, 2051, 1030, a_kc, b_kc, 2.0, 1.0, rng, lda_pad=2 if a_kc else 1,
               ldb_pad=2 if b_kc else 1)                                                          # odd M, N; even pitch
    _gemm_case(ctx, 4000, 1275, 2052, a_kc, b_kc, 1.0, 0.0, rng, ldb_pad=0 if b_kc else 1)        # ladder-like N


@pytest.mark.parametrize("a_kc", [True, False])
@pytest.mark.parametrize("b_kc", [True, False])
def test_matrix_vector_kernels(ctx, a_kc, b_kc):
    """M = 1 / N = 1 products run on the HBM-streaming matrix-vector kernels (weighted column sums through the
    workspace, one wave per row with a shuffle reduction): both orientations of the matrix, odd extents and pitches
    (scalar loads), alpha/beta, a strided result (N = 1 with ldc > 1)."""
    rng = np.random.default_rng(11)
    ctx.prof_enable(True)
    ctx.prof_reset()
    for (M, N, K) in ((1, 4000, 3000), (1, 3999, 3001), (1, 700, 100000), (1, 130000, 300), (3000, 1, 4000),
                      (3001, 1, 2999), (100000, 1, 700), (600, 1, 130000)):
        _gemm_case(ctx, M, N, K, a_kc, b_kc, 1.0, 0.0, rng)
        _gemm_case(ctx, M, N, K, a_kc, b_kc, -0.5, 2.0, rng, lda_pad=2, ldb_pad=1, ldc_pad=3)
    q = ctx.prof_query()
    ctx.prof_enable(False)
    assert q["launches"] == 16


def test_lds_dma_kernel_batched(ctx):
    """Batched products big enough for the LDS-DMA kernel: per-batch base pointers (moved to SGPRs in the kernel), a
    stride-0 operand shared by all batches, and k-split remainder tiles across batch boundaries."""
    rng = np.random.default_rng(9)
    nb, M, N, K = 3, 1500, 1530, 1100
    A, B = rng.standard_normal((nb, M, K)), rng.standard_normal((nb, K, N))
    ctx.prof_enable(True)
    ctx.prof_reset()
    got = ctx.contract("zmk,zkn->zmn", ctx.array(A), ctx.array(B), batch="z").get()
    dma = ctx.prof_query(kernel_class=1)
    ctx.prof_enable(False)
    assert dma["launches"] == 1, "the LDS-DMA kernel was not selected"
    assert np.abs(got - np.einsum("zmk,zkn->zmn", A, B)).max() < 1e-10
    S = rng.standard_normal((K, N))                          # one B for all batches
    C0 = rng.standard_normal((nb, M, N))
    got = ctx.contract("zmk,kn->zmn", ctx.array(A), ctx.array(S), out=ctx.array(C0), alpha=0.5, beta=-1.0, batch="z").get()
    assert np.abs(got - (0.5 * np.einsum("zmk,kn->zmn", A, S) - C0)).max() < 1e-10


def test_dgemm_tail_wave_split(ctx):
    """529 tiles of 128x128 = one full wave of 512 + 17: the remainder runs k-split (tile-local workspace)."""
    rng = np.random.default_rng(7)
    M = N = 2944
    K = 1040
    A, B = rng.standard_normal((M, K)), rng.standard_normal((K, N))
    C0 = rng.standard_normal((M, N))
    got = ctx.contract("mk,kn->mn", ctx.array(A), ctx.array(B), out=ctx.array(C0), alpha=0.5, beta=-1.0).get()
    ref = 0.5 * (A @ B) - C0
    assert np.abs(got - ref).max() < 1e-11


def test_dgemm_identity_asymmetric(ctx):
    """A = I with an asymmetric B catches a transposed C fragment map (cdna guide §3)."""
    n = 64
    B = np.arange(n * n, dtype=np.float64).reshape(n, n)
    dA, dB, dC = ctx.array(np.eye(n)), ctx.array(B), ctx.zeros((n, n))
    ctx.dgemm(n, n, n, 1.0, dA, n, 1, dB, n, 1, 0.0, dC, n)
    assert np.array_equal(dC.get(), B)


def test_contract_splitk_and_batch(ctx):
    rng = np.random.default_rng(2)
    # small output, long K -> split-K path
    A = rng.standard_normal((40, 30000)); B = rng.standard_normal((30000, 24))
    got = ctx.contract("mk,kn->mn", ctx.array(A), ctx.array(B)).get()
    assert np.abs(got - A @ B).max() < 1e-10
    # batched small GEMMs with a stride-0 operand
    X = rng.standard_normal((7, 9)); T = rng.standard_normal((11, 7, 13, 5))
    E = rng.standard_normal((11, 9, 13, 5))
    dE = ctx.array(E)
    ctx.contract("ki,akbj->aibj", ctx.array(X), ctx.array(T), out=dE, alpha=-1.0, beta=1.0, batch="a")
    assert np.abs(dE.get() - (E - np.einsum("ki,akbj->aibj", X, T))).max() < 1e-12
    # operands that need transposition (TTGT path) and a transposed output
    V = rng.standard_normal((6, 6, 10, 10)); T2 = rng.standard_normal((10, 10, 6, 6))
    got = ctx.contract("klcd,adkj->alcj", ctx.array(V), ctx.array(T2)).get()
    assert np.abs(got - np.einsum("klcd,adkj->alcj", V, T2)).max() < 1e-12
    got = ctx.contract("klij,abkl->abij", ctx.array(rng.standard_normal((6, 6, 6, 6))), ctx.array(T2))
    assert got.shape == (10, 10, 6, 6)


@pytest.mark.parametrize("spec,shape", [("abij->aibj", (12, 12, 4, 4)), ("abij->ajbi", (33, 35, 9, 7)),
                                         ("abij->baji", (40, 40, 10, 10)), ("abcd->dcba", (5, 6, 7, 8)),
                                         ("ab->ba", (1000, 333)), ("abc->abc", (9, 8, 7)), ("klcd->ckdl", (50, 50, 64, 64))])
def test_permute(ctx, spec, shape):
    rng = np.random.default_rng(3)
    A = rng.standard_normal(shape)
    li, lo = spec.split("->")
    ref = np.einsum(spec, A)
    assert np.array_equal(ctx.permute(spec, ctx.array(A)).get(), ref)
    O = rng.standard_normal(ref.shape)
    dO = ctx.array(O)
    ctx.permute(spec, ctx.array(A), out=dO, alpha=2.0, beta=-1.0)
    assert np.abs(dO.get() - (2.0 * ref - O)).max() < 1e-14


def test_elementwise_and_reductions(ctx):
    rng = np.random.default_rng(4)
    no, nv = ctx.no, ctx.nv
    eo, ev = np.sort(-1 - rng.random(no)), np.sort(1 + rng.random(nv))
    ctx.set_orbital_energies(eo, ev)
    D = eo[None, None, :, None] + eo[None, None, None, :] - ev[:, None, None, None] - ev[None, :, None, None]
    R = rng.standard_normal((nv, nv, no, no)); T = rng.standard_normal((nv, nv, no, no))
    dT, ddT = ctx.array(T), ctx.empty(T.shape)
    ctx.cc_update(dT, ddT, ctx.array(R), level_shift=0.3, delta=1.0)
    inv = 1.0 / (D + 0.3)
    assert np.array_equal(ddT.get(), R * inv)
    assert np.array_equal(dT.get(), T + R * inv)
    r1 = rng.standard_normal((nv, no)); t1 = rng.standard_normal((nv, no))
    d1, dd1 = ctx.array(t1), ctx.empty(t1.shape)
    ctx.cc_update(d1, dd1, ctx.array(r1), level_shift=0.0)
    assert np.array_equal(dd1.get(), r1 * (1.0 / (eo[None, :] - ev[:, None])))
    xs = [ctx.array(rng.standard_normal(100003)) for _ in range(5)]
    dots = ctx.dots(xs, xs[::-1])
    ref = [float(np.dot(a.get(), b.get())) for a, b in zip(xs, xs[::-1])]
    assert np.allclose(dots, ref, rtol=1e-13, atol=1e-10)
    out = ctx.empty((100003,))
    cs = rng.standard_normal(5)
    ctx.lincomb(out, xs, cs)
    assert np.abs(out.get() - sum(c * x.get() for c, x in zip(cs, xs))).max() < 1e-13


def test_exchange_asymmetry_kernel(gpu_lib):
    """max |A[p,q,r,s] - B[q,p,s,r]| and max |A| through the LDS-tiled kernel: ragged tiles, a planted difference, NaN."""
    import ctypes as C
    from pymes_amd import _lib
    rng = np.random.default_rng(3)
    for (d0, d1, d2, d3) in ((3, 5, 7, 33), (5, 5, 40, 40), (2, 9, 65, 31), (4, 4, 1, 1)):
        A = rng.standard_normal((d0, d1, d2, d3))
        B = A.transpose(1, 0, 3, 2).copy()
        ctx = Context(2, 2, lib=gpu_lib, workspace_bytes=1 << 20)
        try:
            out = (C.c_double * 2)()

            dA = ctx.array(A)

            def asym(Bh):
                dB = ctx.array(Bh)
                ctx.lib.call("pymes_exchange_asymmetry", ctx.handle, C.c_void_p(dA.ptr), C.c_void_p(dB.ptr),
                             _lib.i64_array(A.shape), out)
                return out[0], out[1]
            assert asym(B) == (0.0, np.abs(A).max())
            B2 = B.copy()
            B2[d1 - 1, d0 - 1, d3 - 1, d2 - 1] += 0.125
            assert asym(B2) == (0.125, np.abs(A).max())
            B2[0, 0, 0, 0] = np.nan
            assert asym(B2)[0] == np.inf
        finally:
            ctx.close()


def test_energy_norms_over_pairs(gpu_lib):
    """pymes_energy_norms_pairs: the partial sums of all ranks (compact tiles of their pairs) add up to the one-pass sums
    over the full arrays, for ragged pair chunks and with the T1 terms entering once."""
    from oracle.cases import random_case
    no, nv = 5, 9
    f, V, t1, t2 = random_case(no, nv, 3, symmetric=True)
    rng = np.random.default_rng(5)
    ctx = Context(no, nv, lib=gpu_lib)
    try:
        ctx.set_V_pqrs(V)
        dF, dT1, dT2, dD = ctx.array(f), ctx.array(t1), ctx.array(t2), ctx.array(rng.standard_normal(t2.shape))
        full = np.array(ctx.energy_norms(dF, dT1, dT2, dD))
        npp = nv * (nv + 1) // 2
        for world in (1, 3, 4, 7):
            chunk = -(-npp // world)
            tot = np.zeros(6)
            for rank in range(world):
                tc, dtc = ctx.zeros((chunk, 2, no * no)), ctx.zeros((chunk, 2, no * no))
                ctx.pairs_pack(dT2, tc, rank, world)
                ctx.pairs_pack(dD, dtc, rank, world)
                tot += ctx.energy_norms_pairs(dF, dT1, tc, dtc, rank, world)
            assert np.allclose(tot, full, rtol=1e-12, atol=1e-12), (world, tot, full)
        # CCD form: no T1
        full = np.array(ctx.energy_norms(None, None, dT2, None))
        tot = np.zeros(6)
        for rank in range(3):
            tc = ctx.zeros((-(-npp // 3), 2, no * no))
            ctx.pairs_pack(dT2, tc, rank, 3)
            tot += ctx.energy_norms_pairs(None, None, tc, None, rank, 3)
        assert np.allclose(tot, full, rtol=1e-12, atol=1e-12)
    finally:
        ctx.close()


def test_mixer_against_reference_golden_gpu(gpu_lib):
    """The DIIS step on the device (pymes_diis_step: overlaps, the (m+1) x (m+1) solve by one thread, extrapolation with the
    coefficients read from HBM) and the host solve, on the reference's own DIIS.mix sequence (tests/golden/diis.json)."""
    from tests.test_host_round2 import check_mixer_against_reference_golden
    check_mixer_against_reference_golden(gpu_lib)


def ladder_dress_reference(V, Pk, t1, no, nv, r0, r1, minus):
    """numpy statement of pymes_ladder_dress (include/pymes_amd.h): rows P(a,b) in [r0,r1) of one packed half."""
    W = np.empty_like(V)
    P3 = Pk.reshape(nv, no, -1)
    for a in range(nv):
        for b in range(a + 1):
            r = a * (a + 1) // 2 + b
            if r0 <= r < r1:
                s1 = t1[a] @ P3[b]
                s2 = t1[b] @ P3[a]
                W[r - r0] = 0.0 if (minus and a == b) else V[r - r0] - s1 + (s2 if minus else -s2)
    return W


def check_ladder_dress(lib, no, nv, ld, r0, r1, minus, seed=0):
    rng = np.random.default_rng(seed)
    V = rng.standard_normal((r1 - r0, ld))
    Pk = rng.standard_normal((no * nv, ld))
    t1 = rng.standard_normal((nv, no))
    ctx = Context(no, nv, lib=lib, workspace_bytes=1 << 20)
    try:
        W = ctx.zeros((r1 - r0, ld))
        ctx.ladder_dress(ctx.array(V), ctx.array(Pk), ctx.array(t1), W, ld, r0, r1, minus_half=minus)
        got = W.get()
    finally:
        ctx.close()
    ref = ladder_dress_reference(V, Pk, t1, no, nv, r0, r1, minus)
    assert np.abs(got - ref).max() < 1e-12 * max(1.0, np.abs(ref).max()) * no, (no, nv, ld, r0, r1, minus)


DRESS_CASES = [  # no, nv, ld, r0, r1
    (3, 5, 16, 0, 15), (2, 16, 144, 0, 136), (5, 17, 160, 0, 153), (7, 33, 64, 0, 561), (13, 20, 208, 37, 161),
    (50, 21, 80, 0, 231), (64, 18, 32, 5, 171), (1, 4, 16, 0, 10), (20, 40, 832, 100, 777), (4, 35, 48, 629, 630),
    (70, 19, 48, 3, 190), (80, 17, 32, 0, 153), (77, 33, 64, 200, 561),      # nocc > 64: chains of 18-20 MFMA steps (round 4)
]


@pytest.mark.parametrize("minus", [False, True])
def test_ladder_dress_kernel(gpu_lib, minus):
    """Bra dressing of the pair-packed V_abcd (MFMA rank-nocc updates turned through LDS): ragged tiles in a, b and nocc,
    diagonal tiles, row ranges that cut tiles, pitches that leave waves of the last column block without work."""
    for i, (no, nv, ld, r0, r1) in enumerate(DRESS_CASES):
        check_ladder_dress(gpu_lib, no, nv, ld, r0, r1, minus, seed=i)


def check_dressed_fock_from_blocks(lib, no, nv, seed):
    """ccsd.py:226-288 from the six blocks it reads alone (random, no permutational symmetry), against the oracle."""
    from oracle import cc_oracle as oc
    rng = np.random.default_rng(seed)
    dims = {"o": no, "v": nv}
    shape = lambda key: tuple(dims["o" if ch in "ijkl" else "v"] for ch in key)
    Vb = {key: rng.standard_normal(shape(key)) * 0.1 for key in ("iabj", "ijab", "ijak", "iabc", "iajb", "ijka")}
    n = no + nv
    f = rng.standard_normal((n, n))
    f = 0.5 * (f + f.T)
    t1 = rng.standard_normal((nv, no)) * 0.1
    ctx = Context(no, nv, lib=lib, workspace_bytes=1 << 24)
    try:
        for key, blk in Vb.items():
            ctx.set_V_block(key, blk)
        out = ctx.dress_fock(ctx.array(f), ctx.array(t1), ctx.empty((n, n))).get()
    finally:
        ctx.close()
    ref = oc.dressed_fock(no, f, t1, Vb)
    assert np.abs(out - ref).max() < 1e-11 * max(1.0, np.abs(ref).max()), (no, nv)


def test_dressed_fock_one_pass_kernel_extents(gpu_lib):
    """fock_g12_kernel (the direct / exchange pairs over the o v^3 and o^2 v^2 blocks in one pass): odd and even nvirt, up to
    eight columns per lane, fewer j than chunks, one occupied orbital."""
    for i, (no, nv) in enumerate(((3, 131), (2, 257), (5, 128), (1, 64), (7, 33), (2, 512), (4, 200))):
        check_dressed_fock_from_blocks(gpu_lib, no, nv, i)
