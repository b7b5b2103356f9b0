// Hand-written gfx950 (MI355X, CDNA4) kernels behind device_api.h.
//
//  * dgemm_kernel   fp64 GEMM on v_mfma_f64_16x16x4_f64, LDS-tiled (BK=16, double
//                   buffered, register-staged 16-byte global loads), 256 threads =
//                   4 waves (2x2), each wave FMxFN MFMA tiles; XCD-aware block
//                   remap + grouped tile order; two-level batch; optional split-K.
//                   Every O(N^5)/O(N^6) contraction of the CC path runs here
//                   (ladder ccd.py:187, rings ccd.py:190-240, dressing ccsd.py:290-421).
//  * permute kernels strided copy/transposition (32x32 LDS tile), out = a*in + b*out.
//  * element-wise / reductions for the HBM-bound steps (mp2.py:16, ccsd.py:176-179,
//    diis.py:65-103, ccsd.py:458-466).
//
// Wave = 64 lanes.  f64 MFMA operand maps (cdna guide §3): A lane l holds
// A[i=l&15][k=l>>4], B lane l holds B[k=l>>4][j=l&15], D register r of lane l is
// D[i=(l>>4)+4r][j=l&15].
#include <hip/hip_runtime.h>
#include <type_traits>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "device_api.h"
#include "diis_small.h"

#define HIP_CHECK(expr)                                                                   \
    do {                                                                                  \
        hipError_t err__ = (expr);                                                        \
        if (err__ != hipSuccess)                                                          \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(err__) + \
                                     " at " __FILE__ ":" + std::to_string(__LINE__));     \
    } while (0)

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

namespace {

constexpr int kThreads = 256;
constexpr int BK = 16;

struct GemmK {
    const double* A;
    const double* B;
    double* C;
    const double* Cin;   // beta term is read from here (== C unless the caller fuses a copy)
    long a_ld, b_ld, ldc;
    int M, N, K;
    int Mc, Nc;          // extents the unit-stride M / N loads are clamped to: M, N rounded up to even when the pitch has room
    double alpha, beta;
    int tiles_m, tiles_n;
    int nsplit, kchunk;
    long tile_begin;   // first (batch-major) tile handled by this launch
    long nb2;
    long a_b1, a_b2, b_b1, b_b2, c_b1, c_b2;
    double* ws;   // split-K partials, tile-local: [tile - tile_begin][ks][BM][BN]
    // Mixed launch of the LDS-DMA kernel (mixed != 0): the first `whole` tiles of the launch run as whole tiles (one block
    // each, written straight to C), the remaining `tail` tiles are cut nsplit ways along K with their blocks in ks-major
    // order (block whole + ks * tail + i is cut ks of tile whole + i: co-resident blocks work on the SAME k range of
    // neighbouring tiles and share their A / B panels through the L2, as the blocks of an uncut launch do); partials
    // ws[i][ks], reduced by splitk_reduce_kernel over the tail tiles.  `whole` is a multiple of 8 (XCD remap per part).
    long whole = 0, tail = 0;
    int mixed = 0;
};

// blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2): give each
// XCD a contiguous range of logical ids so neighbouring tiles share an L2 (bijective).
__device__ __forceinline__ long xcd_remap(long b, long nblk) {
    const long q = nblk >> 3, r = nblk & 7;
    const long xcd = b & 7, pos = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
}

// STREAM (64 x 64 tiles only): ONE LDS buffer instead of two — the next k-tile waits in registers while this one is
// consumed, two barriers per k-tile — so that twice as many blocks fit a CU (19 KB of LDS each) and more loads are in
// flight: for the short-K / tiny-output products that stream a 3.2-GB integral block once (T1 dressing, singles residual),
// which are bound by HBM latency x bytes in flight, not by the MFMA pipe.
// `bid`: the block's logical id inside this product (already remapped over the XCDs)
template <int BM, int BN, bool AKC, bool BKC, int VEC, bool STREAM = false>
__device__ __forceinline__ void dgemm_body(const GemmK& g, const long bid) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int WM = BM / 2, WN = BN / 2;      // wave tile (2x2 waves)
    constexpr int FM = WM / 16, FN = WN / 16;    // MFMA tiles per wave
    constexpr int A_PITCH = AKC ? (BK + 2) : (BM + 16);
    constexpr int A_ROWS = AKC ? BM : BK;
    constexpr int A_TILE = A_PITCH * A_ROWS;
    constexpr int B_PITCH = BKC ? (BK + 2) : (BN + 16);
    constexpr int B_ROWS = BKC ? BN : BK;
    constexpr int B_TILE = B_PITCH * B_ROWS;
    constexpr int LA = BM * BK / (kThreads * VEC);   // chunks per thread per tile
    constexpr int LB = BN * BK / (kThreads * VEC);
    constexpr int A_CH = (AKC ? BK : BM) / VEC;      // chunks per LDS row
    constexpr int B_CH = (BKC ? BK : BN) / VEC;
    constexpr int A_RSTEP = kThreads / A_CH;
    constexpr int B_RSTEP = kThreads / B_CH;

    double* As = smem;
    double* Bs = smem + (STREAM ? 1 : 2) * A_TILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, l4 = lane >> 4;

    // ---- block -> (batch z, k-split ks, tile tm/tn) -------------------------------
    const int tiles = g.tiles_m * g.tiles_n;
    const long lt = bid / g.nsplit;                  // launch-local tile; its k-splits are adjacent blocks
    const int ks = (int)(bid - lt * g.nsplit);
    const long gt = g.tile_begin + lt;
    const long z = gt / tiles;
    const int t = (int)(gt - z * tiles);
    constexpr int GROUP = 8;
    const int group_sz = GROUP * g.tiles_n;
    const int grp = t / group_sz;
    const int first_m = grp * GROUP;
    const int gm = min(g.tiles_m - first_m, GROUP);
    const int tin = t - grp * group_sz;
    const int tm = first_m + tin % gm;
    const int tn = tin / gm;
    const long z1 = z / g.nb2, z2 = z - z1 * g.nb2;

    const double* __restrict__ A = g.A + z1 * g.a_b1 + z2 * g.a_b2;
    const double* __restrict__ B = g.B + z1 * g.b_b1 + z2 * g.b_b2;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = ks * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nkt = (kend - kbeg + BK - 1) / BK;

    // ---- per-thread staging geometry (constant over the k loop) -------------------
    // Rows/columns beyond M or N are CLAMPED to the last valid one instead of masked: they only
    // feed rows/columns of C that the epilogue never stores.  Only the k-tail needs zero fill,
    // and only in the last k-tile, so the steady-state loads are branch-free pointer bumps.
    const int a_r = tid / A_CH, a_c = (tid % A_CH) * VEC;
    const int b_r = tid / B_CH, b_c = (tid % B_CH) * VEC;
    const double* pa[LA];
    const double* pb[LB];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int r = a_r + i * A_RSTEP;
        if (AKC) pa[i] = A + (long)min(m0 + r, g.M - 1) * g.a_ld + (kbeg + a_c);
        else pa[i] = A + (long)(kbeg + r) * g.a_ld + min(m0 + a_c, g.Mc - VEC);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int r = b_r + i * B_RSTEP;
        if (BKC) pb[i] = B + (long)min(n0 + r, g.N - 1) * g.b_ld + (kbeg + b_c);
        else pb[i] = B + (long)(kbeg + r) * g.b_ld + min(n0 + b_c, g.Nc - VEC);
    }
    const long a_step = AKC ? (long)BK : (long)BK * g.a_ld;
    const long b_step = BKC ? (long)BK : (long)BK * g.b_ld;

    double ra[LA * VEC], rb[LB * VEC];

    // steady state: whole k-tile in range
    auto load_full = [&]() {
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            if constexpr (VEC == 2) {
                const v2d v = *reinterpret_cast<const v2d*>(pa[i]);
                ra[2 * i] = v[0];
                ra[2 * i + 1] = v[1];
            } else {
                ra[i] = *pa[i];
            }
            pa[i] += a_step;
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            if constexpr (VEC == 2) {
                const v2d v = *reinterpret_cast<const v2d*>(pb[i]);
                rb[2 * i] = v[0];
                rb[2 * i + 1] = v[1];
            } else {
                rb[i] = *pb[i];
            }
            pb[i] += b_step;
        }
    };
    // last, partial k-tile: k >= kend contributes zeros
    auto load_tail = [&](int k0) {
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int kk = k0 + (AKC ? a_c : a_r + i * A_RSTEP);
            const bool ok = kk < kend;
            if constexpr (VEC == 2) {
                v2d v = {0.0, 0.0};
                if (ok) v = *reinterpret_cast<const v2d*>(pa[i]);
                ra[2 * i] = v[0];
                ra[2 * i + 1] = v[1];
            } else {
                ra[i] = ok ? *pa[i] : 0.0;
            }
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int kk = k0 + (BKC ? b_c : b_r + i * B_RSTEP);
            const bool ok = kk < kend;
            if constexpr (VEC == 2) {
                v2d v = {0.0, 0.0};
                if (ok) v = *reinterpret_cast<const v2d*>(pb[i]);
                rb[2 * i] = v[0];
                rb[2 * i + 1] = v[1];
            } else {
                rb[i] = ok ? *pb[i] : 0.0;
            }
        }
    };
    const int nfull = (kend - kbeg) / BK;          // k-tiles completely inside [kbeg, kend)
    auto load_tiles = [&](int kt) {
        if (kt < nfull) load_full();
        else load_tail(kbeg + kt * BK);
    };
    auto store_tiles = [&](int buf) {
        double* as = As + buf * A_TILE;
        double* bs = Bs + buf * B_TILE;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int r = a_r + i * A_RSTEP;
            if constexpr (VEC == 2) {
                v2d v = {ra[2 * i], ra[2 * i + 1]};
                *reinterpret_cast<v2d*>(as + r * A_PITCH + a_c) = v;
            } else {
                as[r * A_PITCH + a_c] = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int r = b_r + i * B_RSTEP;
            if constexpr (VEC == 2) {
                v2d v = {rb[2 * i], rb[2 * i + 1]};
                *reinterpret_cast<v2d*>(bs + r * B_PITCH + b_c) = v;
            } else {
                bs[r * B_PITCH + b_c] = rb[i];
            }
        }
    };

    v4d acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = v4d{0.0, 0.0, 0.0, 0.0};

    // fragment read offsets inside a tile (k-step kk adds AKC ? 4*kk : 4*kk*PITCH)
    const int a_frag = AKC ? ((wm * WM + l15) * A_PITCH + l4) : (l4 * A_PITCH + wm * WM + l15);
    const int b_frag = BKC ? ((wn * WN + l15) * B_PITCH + l4) : (l4 * B_PITCH + wn * WN + l15);
    constexpr int A_FSTEP = AKC ? 16 * A_PITCH : 16;     // next 16-row fragment
    constexpr int B_FSTEP = BKC ? 16 * B_PITCH : 16;
    constexpr int A_KSTEP = AKC ? 4 : 4 * A_PITCH;       // next k-step of 4
    constexpr int B_KSTEP = BKC ? 4 : 4 * B_PITCH;

    // Software pipeline across the per-tile barrier: the fragments of k-step 0 of tile t+1 are read right
    // AFTER the barrier and the 16 MFMAs of the LAST k-step of tile t (operands already in registers) are
    // issued behind those reads, so neither the LDS latency nor the LDS write-back of the staged tile is
    // exposed; the staged tile is written to LDS between k-steps 1 and 2, under MFMAs.
    double a[2][FM], b[2][FN];
    auto read_frags = [&](const double* as, const double* bs, int kk, double (&ra_)[FM], double (&rb_)[FN]) {
#pragma unroll
        for (int i = 0; i < FM; ++i) ra_[i] = as[kk * A_KSTEP + i * A_FSTEP];
#pragma unroll
        for (int j = 0; j < FN; ++j) rb_[j] = bs[kk * B_KSTEP + j * B_FSTEP];
    };
    auto mfma_step = [&](const double (&ra_)[FM], const double (&rb_)[FN]) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra_[i], rb_[j], acc[i][j], 0, 0, 0);
    };
    static_assert(BK == 16, "the k-step schedule below is written for 4 k-steps per tile");
    if (nkt > 0) {
        load_tiles(0);
        store_tiles(0);
    }
    __syncthreads();
    if constexpr (STREAM) {
        for (int kt = 0; kt < nkt; ++kt) {
            const bool has_next = kt + 1 < nkt;
            if (has_next) load_tiles(kt + 1);        // in flight while this tile is consumed
            const double* as = As + a_frag;
            const double* bs = Bs + b_frag;
            read_frags(as, bs, 0, a[0], b[0]);
            read_frags(as, bs, 1, a[1], b[1]);
            mfma_step(a[0], b[0]);
            read_frags(as, bs, 2, a[0], b[0]);
            mfma_step(a[1], b[1]);
            read_frags(as, bs, 3, a[1], b[1]);
            mfma_step(a[0], b[0]);
            mfma_step(a[1], b[1]);
            __syncthreads();                         // every wave has read the tile
            if (has_next) {
                store_tiles(0);
                __syncthreads();
            }
        }
    } else {
    if (nkt > 0) read_frags(As + a_frag, Bs + b_frag, 0, a[0], b[0]);
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool has_next = kt + 1 < nkt;
        if (has_next) load_tiles(kt + 1);        // global loads in flight under the MFMAs
        const double* as = As + cur * A_TILE + a_frag;
        const double* bs = Bs + cur * B_TILE + b_frag;
        read_frags(as, bs, 1, a[1], b[1]);
        mfma_step(a[0], b[0]);
        read_frags(as, bs, 2, a[0], b[0]);
        mfma_step(a[1], b[1]);
        if (has_next) store_tiles(cur ^ 1);      // buffer cur^1 was last read before the previous barrier
        read_frags(as, bs, 3, a[1], b[1]);
        mfma_step(a[0], b[0]);
        __syncthreads();                         // tile kt fully consumed into registers; tile kt+1 visible
        if (has_next) read_frags(As + (cur ^ 1) * A_TILE + a_frag, Bs + (cur ^ 1) * B_TILE + b_frag, 0, a[0], b[0]);
        mfma_step(a[1], b[1]);
    }
    }

    // ---- epilogue -------------------------------------------------------------------
    if (g.nsplit > 1) {   // k-split partial: whole tile, tile-local layout, combined by splitk_reduce_kernel
        double* __restrict__ W = g.ws + (lt * g.nsplit + ks) * (long)(BM * BN);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    W[(wm * WM + i * 16 + l4 + 4 * r) * BN + wn * WN + j * 16 + l15] = acc[i][j][r];
        return;
    }
    double* C = g.C + z1 * g.c_b1 + z2 * g.c_b2;
    const double* Cin = g.Cin + z1 * g.c_b1 + z2 * g.c_b2;
    const long ldc = g.ldc;
    const double alpha = g.alpha, beta = g.beta;
    // beta term: Cin may be C itself, so the compiler cannot move a load above an earlier store — read the FN x 4 values of
    // one row of MFMA tiles first (independent loads, all in flight together), then combine and store.  An element-wise
    // load -> store chain costs one memory latency per element: the whole time of a short-K accumulating product.
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        double cin[FN][4];
        if (beta != 0.0) {
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n = n0 + wn * WN + j * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * WM + i * 16 + l4 + 4 * r;
                    cin[j][r] = (m < g.M && n < g.N) ? Cin[(long)m * ldc + n] : 0.0;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn * WN + j * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * WM + i * 16 + l4 + 4 * r;
                if (m < g.M && n < g.N) {
                    double v = alpha * acc[i][j][r];
                    if (beta != 0.0) v += beta * cin[j][r];
                    C[(long)m * ldc + n] = v;
                }
            }
        }
    }
}

template <int BM, int BN, bool AKC, bool BKC, int VEC, bool STREAM = false>
__global__ void __launch_bounds__(kThreads, STREAM ? 4 : 2) dgemm_kernel(const GemmK g) {
    dgemm_body<BM, BN, AKC, BKC, VEC, STREAM>(g, xcd_remap(blockIdx.x, gridDim.x));
}

// Grouped launch: up to kGroupMax INDEPENDENT small products (64 x 64 tiles, any operand layout, with or without k-split)
// in ONE grid — one launch per dependency level of a term sequence instead of one per product.  The (20,80) iteration and the
// per-vector terms of an EOM sigma build are chains of 5-50 us kernels, each a latency chain (global -> LDS -> barrier -> MFMA)
// on a quarter-filled chip; grouped, the blocks of all products of a level share the chip and one launch / drain.
// Blocks are dealt to the products by a prefix table (uniform scan of <= 16 entries in SGPRs); a block then runs the body of
// its product's layout variant.  Descriptors travel as kernel arguments (3 KB): nothing to upload, graph capture keeps them.
constexpr int kGroupMax = 16;
struct GroupK {
    int n, tile;                // tile: 64 or 128 (read by the grouped reduction)
    int blk_end[kGroupMax];     // running block count
    int variant[kGroupMax];     // bit 2: A K-contiguous, bit 1: B K-contiguous, bit 0: 16-byte loads
    GemmK g[kGroupMax];
};
__global__ void __launch_bounds__(kThreads, 2) dgemm_group_kernel(const GroupK grp) {
    const long gb = xcd_remap(blockIdx.x, gridDim.x);      // an XCD runs a contiguous range of logical blocks
    int it = 0;
    while (it + 1 < grp.n && gb >= grp.blk_end[it]) ++it;
    const long bid = gb - (it ? grp.blk_end[it - 1] : 0);
    const GemmK& g = grp.g[it];
    switch (grp.variant[it]) {
        case 0: dgemm_body<64, 64, false, false, 1>(g, bid); break;
        case 1: dgemm_body<64, 64, false, false, 2>(g, bid); break;
        case 2: dgemm_body<64, 64, false, true, 1>(g, bid); break;
        case 3: dgemm_body<64, 64, false, true, 2>(g, bid); break;
        case 4: dgemm_body<64, 64, true, false, 1>(g, bid); break;
        case 5: dgemm_body<64, 64, true, false, 2>(g, bid); break;
        case 6: dgemm_body<64, 64, true, true, 1>(g, bid); break;
        default: dgemm_body<64, 64, true, true, 2>(g, bid); break;
    }
}

// ------------------------------------------------------------------------------------
// LDS-DMA variant of the 128x128 GEMM (16-byte aligned operands).  Same tiling, MFMA schedule and
// epilogue as dgemm_kernel, but the operand tiles go global -> LDS directly
// (`global_load_lds_dwordx4`, issued through glds16_su below: no staging VGPRs, no ds_write, no per-tile pointer
// arithmetic on the VALU; M0 is written by hand — the kernel has no other M0 user):
//   * an M/N-contiguous tile [16 k][128] is 16 wave-instructions of one 1-KiB k-row each, LDS pitch 144;
//   * a K-contiguous tile [128 rows][16 k] is 16 wave-instructions of eight 128-B rows each.  The LDS
//     destination of an LDS-DMA is lane-linear, so the bank-conflict fix cannot be padding: the 16-B
//     chunk c of row r is stored at chunk slot c ^ ((r >> 1) & 7) by permuting the per-lane SOURCE address,
//     and the fragment reads apply the same XOR (conflict-free ds_read_b64 for 16 consecutive rows).
// The last, partial k-tile goes through the same DMA with the lanes beyond K switched off by EXEC and their LDS slots
// zero-filled by a ds_write (stage_tail below).
// ------------------------------------------------------------------------------------
// wave-uniform values that the compiler computes on the VALU (64-bit divisions of the block id) are moved to
// SGPRs explicitly, so that loop control and the LDS-DMA base addresses stay on the scalar unit
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ const char* uni_ptr(const void* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}

// LDS-DMA with the address split the way the hardware wants it: wave-uniform 64-bit base in SGPRs + 32-bit byte
// offset per lane + wave-uniform LDS destination in M0 (written here because the compiler turns the builtin's
// address back into a 64-bit VALU add per load inside the k loop).
__device__ __forceinline__ void glds16_su(const char* base_uniform, unsigned voff, unsigned lds_addr_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(voff), "s"(base_uniform), "s"(lds_addr_uniform)
                 : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr_of(const double* p) {
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}

// One output tile over the k range [kbeg, kend) (kbeg a multiple of BK): `gt` = tile id in the batch-major, grouped order of
// the whole product; `Wpart` = destination of a partial tile [128][128] (k-split), or null: alpha / beta epilogue into C.
// (Every wave has passed the barrier behind its last LDS read when the function returns: a persistent kernel may call it
// again at once — the hybrid stream-K launch of round 5 did)
template <bool AKC, bool BKC>
__device__ __forceinline__ void dgemm_glds_tile(const GemmK& g, const long gt, const int kbeg, const int kend,
                                                double* __restrict__ Wpart) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int BM = 128, BN = 128, WM = 64, WN = 64, FM = 4, FN = 4;
    constexpr int A_PITCH = AKC ? BK : (BM + 16);
    constexpr int A_TILE = AKC ? BM * BK : BK * A_PITCH;
    constexpr int B_PITCH = BKC ? BK : (BN + 16);
    constexpr int B_TILE = BKC ? BN * BK : BK * B_PITCH;
    double* As = smem;
    double* Bs = smem + 2 * A_TILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform (LDS-DMA base = M0)
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, l4 = lane >> 4;

    const int tiles = g.tiles_m * g.tiles_n;
    const long z = gt / tiles;
    const int t = (int)(gt - z * tiles);
    constexpr int GROUP = 8;
    const int group_sz = GROUP * g.tiles_n;
    const int grp = t / group_sz;
    const int first_m = grp * GROUP;
    const int gm = min(g.tiles_m - first_m, GROUP);
    const int tin = t - grp * group_sz;
    const int tm = uni(first_m + tin % gm);
    const int tn = uni(tin / gm);
    const long z1 = z / g.nb2, z2 = z - z1 * g.nb2;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nkt = (kend - kbeg + BK - 1) / BK;
    const int nfull = (kend - kbeg) / BK;

    // ---- LDS-DMA sources: 4 wave-instructions per operand per wave per tile -------------------------
    // K-contiguous: instruction j of wave w covers rows (4w+j)*8 .. +8; lane -> (row, chunk slot)
    // M/N-contiguous: instruction j of wave w covers k-row 4w+j; lane -> columns 2*lane, 2*lane+1
    // rows / columns beyond M or N are clamped (they only feed C entries that are never stored).
    // Address = wave-uniform 64-bit base (SGPRs, advanced per k-tile on the scalar unit) + loop-invariant
    // 32-bit byte offset per lane (the host guarantees 128 * ld * 8 < 2^32): no VALU work per k-tile.
    const char* ua = uni_ptr(g.A + z1 * g.a_b1 + z2 * g.a_b2 + (AKC ? (long)m0 * g.a_ld + kbeg : (long)kbeg * g.a_ld + m0));
    const char* ub = uni_ptr(g.B + z1 * g.b_b1 + z2 * g.b_b2 + (BKC ? (long)n0 * g.b_ld + kbeg : (long)kbeg * g.b_ld + n0));
    unsigned ao[4], bo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (AKC) {
            const int row = wave * 32 + (lane >> 3) + 8 * j;
            const int cg = (lane & 7) ^ ((row >> 1) & 7);
            ao[j] = ((unsigned)(min(m0 + row, g.M - 1) - m0) * (unsigned)g.a_ld + cg * 2) * 8u;
        } else {
            ao[j] = ((unsigned)(wave * 4 + j) * (unsigned)g.a_ld + (unsigned)(min(m0 + 2 * lane, g.Mc - 2) - m0)) * 8u;
        }
        if (BKC) {
            const int row = wave * 32 + (lane >> 3) + 8 * j;
            const int cg = (lane & 7) ^ ((row >> 1) & 7);
            bo[j] = ((unsigned)(min(n0 + row, g.N - 1) - n0) * (unsigned)g.b_ld + cg * 2) * 8u;
        } else {
            bo[j] = ((unsigned)(wave * 4 + j) * (unsigned)g.b_ld + (unsigned)(min(n0 + 2 * lane, g.Nc - 2) - n0)) * 8u;
        }
    }
    const long a_kstep = 8 * (AKC ? (long)BK : (long)BK * g.a_ld);      // bytes
    const long b_kstep = 8 * (BKC ? (long)BK : (long)BK * g.b_ld);
    // wave-uniform LDS destinations of the 4 instructions
    const int a_dst = AKC ? wave * 32 * BK : wave * 4 * A_PITCH;       // + j * (8*BK | A_PITCH)
    const int b_dst = BKC ? wave * 32 * BK : wave * 4 * B_PITCH;
    constexpr int A_DSTEP = AKC ? 8 * BK : A_PITCH;
    constexpr int B_DSTEP = BKC ? 8 * BK : B_PITCH;

    const unsigned a_lds = uni((int)lds_addr_of(As + a_dst)), b_lds = uni((int)lds_addr_of(Bs + b_dst));
    auto stage_dma = [&](int buf) {      // one full k-tile, then advance the source bases
        const unsigned as = a_lds + buf * (A_TILE * 8), bs = b_lds + buf * (B_TILE * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_su(ua, ao[j], as + j * (A_DSTEP * 8));
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_su(ub, bo[j], bs + j * (B_DSTEP * 8));
        ua += a_kstep;
        ub += b_kstep;
    };
    // partial k-tile: the same DMA with the lanes (K-contiguous: 16-byte chunks; M/N-contiguous: whole k-rows) beyond
    // kend switched off; their LDS slots are zero-filled instead.  A slot is written by exactly one of the two, so no
    // ordering between the ds_write and the DMA is needed, and the tail costs three VGPRs instead of a register-staged
    // copy of both tiles (which used to push the kernel's accumulators into scratch).
    auto stage_tail = [&](int buf, int k0) {
        const unsigned as = a_lds + buf * (A_TILE * 8), bs = b_lds + buf * (B_TILE * 8);
        const v2d zero = {0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool in_a, in_b;
            if (AKC) {
                const int row = wave * 32 + (lane >> 3) + 8 * j;
                in_a = k0 + (((lane & 7) ^ ((row >> 1) & 7)) << 1) < kend;
            } else {
                in_a = k0 + wave * 4 + j < kend;
            }
            if (BKC) {
                const int row = wave * 32 + (lane >> 3) + 8 * j;
                in_b = k0 + (((lane & 7) ^ ((row >> 1) & 7)) << 1) < kend;
            } else {
                in_b = k0 + wave * 4 + j < kend;
            }
            if (in_a) glds16_su(ua, ao[j], as + j * (A_DSTEP * 8));
            else *reinterpret_cast<v2d*>(As + buf * A_TILE + a_dst + j * A_DSTEP + 2 * lane) = zero;
            if (in_b) glds16_su(ub, bo[j], bs + j * (B_DSTEP * 8));
            else *reinterpret_cast<v2d*>(Bs + buf * B_TILE + b_dst + j * B_DSTEP + 2 * lane) = zero;
        }
    };

    v4d acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = v4d{0.0, 0.0, 0.0, 0.0};

    // fragment offsets.  K-contiguous: row*16 + (((2kk + (l4>>1)) ^ f) * 2) + (l4&1), f = (row>>1)&7 = (l15>>1)&7
    const int fsw = (l15 >> 1) & 7;
    int a_koff[4], b_koff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int sw = (((2 * kk + (l4 >> 1)) ^ fsw) << 1) + (l4 & 1);
        a_koff[kk] = AKC ? (wm * WM + l15) * BK + sw : (kk * 4 + l4) * A_PITCH + wm * WM + l15;
        b_koff[kk] = BKC ? (wn * WN + l15) * BK + sw : (kk * 4 + l4) * B_PITCH + wn * WN + l15;
    }
    constexpr int A_FSTEP = AKC ? 16 * BK : 16;
    constexpr int B_FSTEP = BKC ? 16 * BK : 16;

    double a[2][FM], b[2][FN];
    auto read_frags = [&](const double* as, const double* bs, int kk, double (&ra_)[FM], double (&rb_)[FN]) {
#pragma unroll
        for (int i = 0; i < FM; ++i) ra_[i] = as[a_koff[kk] + i * A_FSTEP];
#pragma unroll
        for (int j = 0; j < FN; ++j) rb_[j] = bs[b_koff[kk] + j * B_FSTEP];
    };
    auto mfma_step = [&](const double (&ra_)[FM], const double (&rb_)[FN]) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra_[i], rb_[j], acc[i][j], 0, 0, 0);
    };
    auto landed = [&]() {     // this wave's DMA has landed and its LDS reads are done; then everyone's
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    // one k-tile out of buffer CUR (a compile-time constant: the fragment addresses are then loop-invariant
    // registers + immediate offsets); `next`: 0 = nothing follows, 1 = a full tile (DMA), 2 = the partial tile
    auto ktile = [&](auto curc, int next, int k_next) {
        constexpr int cur = decltype(curc)::value;
        if (next == 1) stage_dma(cur ^ 1);           // buffer cur^1 was last read before the previous barrier
        else if (next == 2) stage_tail(cur ^ 1, k_next);
        const double* as = As + cur * A_TILE;
        const double* bs = Bs + cur * B_TILE;
        read_frags(as, bs, 1, a[1], b[1]);
        mfma_step(a[0], b[0]);
        read_frags(as, bs, 2, a[0], b[0]);
        mfma_step(a[1], b[1]);
        read_frags(as, bs, 3, a[1], b[1]);
        mfma_step(a[0], b[0]);
        landed();
        if (next) read_frags(As + (cur ^ 1) * A_TILE, Bs + (cur ^ 1) * B_TILE, 0, a[0], b[0]);
        mfma_step(a[1], b[1]);
    };
    const std::integral_constant<int, 0> buf0;
    const std::integral_constant<int, 1> buf1;

    if (nkt > 0) {
        if (nfull > 0) stage_dma(0);
        else stage_tail(0, kbeg);
        landed();
        read_frags(As, Bs, 0, a[0], b[0]);
        int kt = 0;
        // steady state: the tile after the current one and the one after that are both full
        for (; kt + 2 < nfull; kt += 2) {
            ktile(buf0, 1, 0);
            ktile(buf1, 1, 0);
        }
        // what is left: at most two full tiles and the partial one, starting in buffer 0 (kt is even) — straight-line
        // code with compile-time buffers (a loop over the parity made the compiler spill the accumulators)
        auto next_of = [&](int t) { return t + 1 < nfull ? 1 : (t + 1 < nkt ? 2 : 0); };
        ktile(buf0, next_of(kt), kbeg + (kt + 1) * BK);
        if (kt + 1 < nkt) {
            ktile(buf1, next_of(kt + 1), kbeg + (kt + 2) * BK);
            if (kt + 2 < nkt) ktile(buf0, 0, 0);
        }
    } else {
        landed();
    }

    // ---- epilogue (as dgemm_kernel) -----------------------------------------------------------------
    if (Wpart) {
        double* __restrict__ W = Wpart;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    W[(wm * WM + i * 16 + l4 + 4 * r) * BN + wn * WN + j * 16 + l15] = acc[i][j][r];
        return;
    }
    double* C = g.C + z1 * g.c_b1 + z2 * g.c_b2;
    const double* Cin = g.Cin + z1 * g.c_b1 + z2 * g.c_b2;
    const long ldc = g.ldc;
    const double alpha = g.alpha, beta = g.beta;
    // (element-wise beta term on purpose: this kernel runs long k loops, where the epilogue does not count, and batching the
    // Cin loads as dgemm_kernel does pushes two of the four variants to the 256-VGPR limit and costs 3 % in the k loop)
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn * WN + j * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * WM + i * 16 + l4 + 4 * r;
                if (m < g.M && n < g.N) {
                    const long off = (long)m * ldc + n;
                    double v = alpha * acc[i][j][r];
                    if (beta != 0.0) v += beta * Cin[off];
                    C[off] = v;
                }
            }
        }
    }
}

// block -> (tile, k cut) of the tile-per-block launches
template <bool AKC, bool BKC>
__device__ __forceinline__ void dgemm_glds_body(const GemmK& g, const long bid) {
    long lt;             // launch-local tile
    int ks;              // its k cut handled here
    bool cut;            // this block writes a partial tile
    if (g.mixed) {
        if (bid < g.whole) { lt = bid; ks = 0; cut = false; }
        else {
            const long r = bid - g.whole;
            ks = uni((int)(r / g.tail));
            lt = g.whole + (r - (long)ks * g.tail);
            cut = true;
        }
    } else {
        lt = bid / g.nsplit;
        ks = uni((int)(bid - lt * g.nsplit));
        cut = g.nsplit > 1;
    }
    const int kbeg = cut ? ks * g.kchunk : 0;
    const int kend = cut ? min(g.K, kbeg + g.kchunk) : g.K;
    dgemm_glds_tile<AKC, BKC>(g, g.tile_begin + lt, kbeg, kend,
                              cut ? g.ws + ((lt - (g.mixed ? g.whole : 0)) * g.nsplit + ks) * 16384L : nullptr);
}

template <bool AKC, bool BKC>
__global__ void __launch_bounds__(kThreads, 2) dgemm_glds_kernel(const GemmK g) {
    // mixed launch: whole tiles and cut blocks are dealt to the XCDs separately (each XCD its share of both: an XCD has no
    // way to take over work of another one), whole % 8 == 0 keeps hardware block b on XCD b % 8 in both parts
    const long b = blockIdx.x;
    const long id = !g.mixed ? xcd_remap(b, gridDim.x)
                             : (b < g.whole ? xcd_remap(b, g.whole) : g.whole + xcd_remap(b - g.whole, (long)gridDim.x - g.whole));
    dgemm_glds_body<AKC, BKC>(g, id);
}
// Grouped launch of mid-size products on the LDS-DMA kernel (A K-contiguous, B N-contiguous — the layout of the pair-packed
// ladders and of every product with a symmetric pair matrix on the left): the four halves of the particle ladder and of
// Q_kb at (20,80) are 26-52 tiles of 128 x 128 each, K ~ 3200 — alone each under-fills the chip and falls back to 64 x 64
// register-staged tiles at 40 TF; together, k-split to ~4 blocks per CU, they run on the kernel that reaches 55-60 TF there.
__global__ void __launch_bounds__(kThreads, 2) dgemm_glds_group_kernel(const GroupK grp) {
    const long gb = xcd_remap(blockIdx.x, gridDim.x);
    int it = 0;
    while (it + 1 < grp.n && gb >= grp.blk_end[it]) ++it;
    dgemm_glds_body<true, false>(grp.g[it], gb - (it ? grp.blk_end[it - 1] : 0));
}

// C tile = alpha * sum_ks ws[tile][ks] + beta * C tile, for the tiles [tile_begin, tile_begin + ntiles)
__device__ __forceinline__ void splitk_reduce_body(const GemmK& g, const int BM, const int BN, const long lt, const int piece) {
    const int tiles = g.tiles_m * g.tiles_n;
    const long gt = g.tile_begin + lt;
    const long z = gt / tiles;
    const int t = (int)(gt - z * tiles);
    constexpr int GROUP = 8;
    const int group_sz = GROUP * g.tiles_n;
    const int grp = t / group_sz;
    const int first_m = grp * GROUP;
    const int gm = min(g.tiles_m - first_m, GROUP);
    const int tin = t - grp * group_sz;
    const int tm = first_m + tin % gm;
    const int tn = tin / gm;
    const long z1 = z / g.nb2, z2 = z - z1 * g.nb2;
    double* C = g.C + z1 * g.c_b1 + z2 * g.c_b2;
    const double* Cin = g.Cin + z1 * g.c_b1 + z2 * g.c_b2;
    const double* __restrict__ W = g.ws + lt * g.nsplit * (long)(BM * BN);
    // blockIdx.y cuts the tile into 256-element pieces: a launch with few tiles and many splits (huge K, tiny
    // output) still spreads over the chip
    const int e = piece * 256 + threadIdx.x;
    const int r = e / BN, c = e - r * BN;
    const int m = tm * BM + r, n = tn * BN + c;
    if (m >= g.M || n >= g.N) return;
    const long st = (long)BM * BN;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = 0;
    for (; k + 16 <= g.nsplit; k += 16) {       // sixteen loads in flight (a chain of single loads is a latency chain:
        double w[16];                           // 40 us for 256 partials); same summation order as the loop below
#pragma unroll
        for (int q = 0; q < 16; ++q) w[q] = W[(long)(k + q) * st + e];
#pragma unroll
        for (int q = 0; q < 16; q += 4) { s0 += w[q]; s1 += w[q + 1]; s2 += w[q + 2]; s3 += w[q + 3]; }
    }
    for (; k + 4 <= g.nsplit; k += 4) {
        s0 += W[(long)k * st + e];
        s1 += W[(long)(k + 1) * st + e];
        s2 += W[(long)(k + 2) * st + e];
        s3 += W[(long)(k + 3) * st + e];
    }
    for (; k < g.nsplit; ++k) s0 += W[(long)k * st + e];
    const long off = (long)m * g.ldc + n;
    double v = g.alpha * ((s0 + s1) + (s2 + s3));
    if (g.beta != 0.0) v += g.beta * Cin[off];
    C[off] = v;
}
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const GemmK g, int BM, int BN) {
    splitk_reduce_body(g, BM, BN, blockIdx.x, blockIdx.y);
}
// the k-split products of one grouped launch, reduced by one launch (64 x 64 tiles: 16 pieces of 256 elements per tile)
__global__ void __launch_bounds__(256) splitk_reduce_group_kernel(const GroupK grp) {
    int it = 0;
    while (it + 1 < grp.n && (int)blockIdx.x >= grp.blk_end[it]) ++it;
    const int local = blockIdx.x - (it ? grp.blk_end[it - 1] : 0);
    if (grp.tile == 64) splitk_reduce_body(grp.g[it], 64, 64, local >> 4, local & 15);
    else splitk_reduce_body(grp.g[it], 128, 128, local >> 6, local & 63);
}

__device__ __forceinline__ double wave_sum(double v);

// ------------------------------------------------------------------------------------
// Matrix-vector products (the M = 1 / N = 1 contractions of the dressed Fock matrix and the singles residual):
// HBM-streaming, no MFMA.  W is [R][C] with unit stride along C (row pitch ld).
//   gemv_cols:  part[s][c] = sum_{r in chunk s} x[r * xs] W[r][c]      (weighted column sums; grid = column blocks x row chunks)
//   gemv_finish: y[c * ys] = alpha * sum_s part[s][c] + beta * yin[c * ys]
//   gemv_rows:  y[r * ys] = alpha * sum_c W[r][c] x[c * xs] + beta * yin[r * ys]   (one wave per row, shuffle reduction)
// ------------------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(256) gemv_cols_kernel(const double* __restrict__ W, long ld, const double* __restrict__ x,
                                                        long xs, long R, long C, long rchunk, double* __restrict__ part) {
    const long c = ((long)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (c >= C) return;
    const long r0 = (long)blockIdx.y * rchunk, r1 = min(R, r0 + rchunk);
    const double* __restrict__ w = W + r0 * ld + c;
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    long r = r0;
    for (; r + 2 <= r1; r += 2) {
        const double x0 = x[r * xs], x1 = x[(r + 1) * xs];
        if constexpr (VEC == 2) {
            const v2d u = *reinterpret_cast<const v2d*>(w), v = *reinterpret_cast<const v2d*>(w + ld);
            a0 += x0 * u[0]; a1 += x0 * u[1];
            b0 += x1 * v[0]; b1 += x1 * v[1];
        } else {
            a0 += x0 * w[0];
            b0 += x1 * w[ld];
        }
        w += 2 * ld;
    }
    if (r < r1) {
        const double x0 = x[r * xs];
        if constexpr (VEC == 2) {
            const v2d u = *reinterpret_cast<const v2d*>(w);
            a0 += x0 * u[0]; a1 += x0 * u[1];
        } else {
            a0 += x0 * w[0];
        }
    }
    double* __restrict__ p = part + (long)blockIdx.y * C + c;
    p[0] = a0 + b0;
    if constexpr (VEC == 2) p[1] = a1 + b1;
}
__global__ void __launch_bounds__(256) gemv_finish_kernel(const double* __restrict__ part, int nchunk, long C, double alpha,
                                                          double beta, const double* yin, double* y, long ys) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    // four running sums: the loads of a round are independent (a single chain waits for every load in turn)
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = 0;
    for (; k + 4 <= nchunk; k += 4) {
        s0 += part[(long)k * C + c];
        s1 += part[(long)(k + 1) * C + c];
        s2 += part[(long)(k + 2) * C + c];
        s3 += part[(long)(k + 3) * C + c];
    }
    for (; k < nchunk; ++k) s0 += part[(long)k * C + c];
    const double s = (s0 + s1) + (s2 + s3);
    double v = alpha * s;
    if (beta != 0.0) v += beta * yin[c * ys];
    y[c * ys] = v;
}

// Several independent weighted-column-sum products in ONE launch (dev::gemv_batch_begin/end): the T1.V intermediates of the
// dressed Fock matrix are eight such products, each far too small at (20,80) to fill the chip or to hide its own latency.
// The table travels as a kernel argument (launch-graph friendly).
struct GemvItem {
    const double* W; const double* x; double* y;
    long ld, xs, R, C, rchunk, ys, ws_off;
    double alpha;
    int nchunk, cblocks, vec, blk0, out0;       // first block of the item in the product / in the finishing launch
};
constexpr int kGemvBatchMax = 12;
struct GemvTable {
    int n;
    GemvItem it[kGemvBatchMax];
};

__device__ __forceinline__ void gemv_cols_item(const GemvItem& g, const int local, double* __restrict__ ws) {
    const int cb = local % g.cblocks, chunk = local / g.cblocks;
    const long c = ((long)cb * 256 + threadIdx.x) * g.vec;
    if (c >= g.C) return;
    const long r0 = (long)chunk * g.rchunk, r1 = min(g.R, r0 + g.rchunk);
    const double* __restrict__ w = g.W + r0 * g.ld + c;
    const double* __restrict__ x = g.x;
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    long r = r0;
    if (g.vec == 2) {
        for (; r + 2 <= r1; r += 2) {
            const double x0 = x[r * g.xs], x1 = x[(r + 1) * g.xs];
            const v2d u = *reinterpret_cast<const v2d*>(w), v = *reinterpret_cast<const v2d*>(w + g.ld);
            a0 += x0 * u[0]; a1 += x0 * u[1];
            b0 += x1 * v[0]; b1 += x1 * v[1];
            w += 2 * g.ld;
        }
        if (r < r1) {
            const double x0 = x[r * g.xs];
            const v2d u = *reinterpret_cast<const v2d*>(w);
            a0 += x0 * u[0]; a1 += x0 * u[1];
        }
    } else {
        for (; r + 2 <= r1; r += 2) {
            a0 += x[r * g.xs] * w[0];
            b0 += x[(r + 1) * g.xs] * w[g.ld];
            w += 2 * g.ld;
        }
        if (r < r1) a0 += x[r * g.xs] * w[0];
    }
    double* __restrict__ p = ws + g.ws_off + (long)chunk * g.C + c;
    p[0] = a0 + b0;
    if (g.vec == 2) p[1] = a1 + b1;
}
__global__ void __launch_bounds__(256) gemv_multi_cols_kernel(const GemvTable tab, double* __restrict__ ws) {
    int i = 0;
    while (i + 1 < tab.n && (int)blockIdx.x >= tab.it[i + 1].blk0) ++i;
    gemv_cols_item(tab.it[i], blockIdx.x - tab.it[i].blk0, ws);
}

// (beta / yin: the phase tasks finish accumulating products too; the batch kernel passes beta = 0)
__device__ __forceinline__ void gemv_finish_item(const GemvItem& g, const int local, const double* __restrict__ ws, const double beta,
                                                 const double* yin) {
    const long c = (long)local * 256 + threadIdx.x;
    if (c >= g.C) return;
    const double* __restrict__ part = ws + g.ws_off + c;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = 0;
    for (; k + 4 <= g.nchunk; k += 4) {
        s0 += part[(long)k * g.C];
        s1 += part[(long)(k + 1) * g.C];
        s2 += part[(long)(k + 2) * g.C];
        s3 += part[(long)(k + 3) * g.C];
    }
    for (; k < g.nchunk; ++k) s0 += part[(long)k * g.C];
    double v = g.alpha * ((s0 + s1) + (s2 + s3));
    if (beta != 0.0) v += beta * yin[c * g.ys];
    g.y[c * g.ys] = v;
}
__global__ void __launch_bounds__(256) gemv_multi_finish_kernel(const GemvTable tab, const double* __restrict__ ws) {
    int i = 0;
    while (i + 1 < tab.n && (int)blockIdx.x >= tab.it[i + 1].out0) ++i;
    gemv_finish_item(tab.it[i], blockIdx.x - tab.it[i].out0, ws, 0.0, nullptr);
}
struct GemvRowsK {
    const double* W; const double* x; const double* yin; double* y;
    long ld, xs, R, C, ys;
    double alpha, beta;
};
template <int VEC>
__device__ __forceinline__ void gemv_rows_body(const GemvRowsK& k, const unsigned vb) {
    const double* __restrict__ W = k.W;
    const double* __restrict__ x = k.x;
    const double* yin = k.yin;
    double* y = k.y;
    const long ld = k.ld, xs = k.xs, R = k.R, C = k.C, ys = k.ys;
    const double alpha = k.alpha, beta = k.beta;
    const long r = (long)vb * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int lane = threadIdx.x & 63;
    const double* __restrict__ w = W + r * ld;
    double s0 = 0.0, s1 = 0.0;
    if constexpr (VEC == 2) {
        for (long c = 2 * lane; c < C; c += 128) {
            const v2d u = *reinterpret_cast<const v2d*>(w + c);
            s0 += u[0] * x[c * xs];
            s1 += u[1] * x[(c + 1) * xs];
        }
    } else {
        for (long c = lane; c < C; c += 64) s0 += w[c] * x[c * xs];
    }
    const double s = wave_sum(s0 + s1);
    if (lane == 0) {
        double v = alpha * s;
        if (beta != 0.0) v += beta * yin[r * ys];
        y[r * ys] = v;
    }
}
template <int VEC>
__global__ void __launch_bounds__(256) gemv_rows_kernel(const GemvRowsK k) { gemv_rows_body<VEC>(k, blockIdx.x); }

// ------------------------------------------------------------------------------------
// permutation / strided copy
// ------------------------------------------------------------------------------------
struct PermK {
    int rank;
    long dim[6];
    long s_in[6], s_out[6];
    double alpha, beta;
    const double* in;
    double* out;
};

// generic: one element per thread and pass of the grid-stride loop, last (canonical) dim fastest.  The digits of the element
// index advance by the digits of the grid stride with carries (a per-element decomposition is up to six 64-bit divisions:
// the 0.8-GB permutations of (50,200) ran at 3.2-4 TB/s, VALU-bound)
// (bodies take the block's id and the size of their grid as arguments: a block of a phase launch — phase_kernel below — runs
// them for its task with ids of that task's own)
__device__ __forceinline__ void permute_direct_body(const PermK& p, const long total, const unsigned vb, const unsigned vgrid) {
    const long stride = (long)vgrid * blockDim.x;
    long idx = vb * (long)blockDim.x + threadIdx.x;
    if (idx >= total) return;
    long c[6], sd[6];          // (64-bit: a merged dimension may exceed 2^31 elements)
    {
        long rem = idx, rs = stride;
#pragma unroll
        for (int d = 5; d >= 0; --d) {
            c[d] = 0; sd[d] = 0;
            if (d < p.rank) {
                if (d == 0) { c[0] = rem; sd[0] = rs; }         // the slowest digit never wraps inside the tensor
                else {
                    long q = rem / p.dim[d];
                    c[d] = rem - q * p.dim[d];
                    rem = q;
                    q = rs / p.dim[d];
                    sd[d] = rs - q * p.dim[d];
                    rs = q;
                }
            }
        }
    }
    for (; idx < total; idx += stride) {
        long oi = 0, oo = 0;
#pragma unroll
        for (int d = 0; d < 6; ++d)
            if (d < p.rank) { oi += c[d] * p.s_in[d]; oo += c[d] * p.s_out[d]; }
        double v = p.alpha * p.in[oi];
        if (p.beta != 0.0) v += p.beta * p.out[oo];
        p.out[oo] = v;
        long carry = 0;
#pragma unroll
        for (int d = 5; d >= 1; --d) {
            if (d < p.rank) {
                c[d] += sd[d] + carry;
                carry = c[d] >= p.dim[d] ? 1 : 0;
                c[d] -= carry ? p.dim[d] : 0;
            }
        }
        c[0] += sd[0] + carry;
    }
}
__global__ void permute_direct_kernel(const PermK p, long total) { permute_direct_body(p, total, blockIdx.x, gridDim.x); }

// tiled transpose: canonical dims [rest..., Q, L] where in is unit-stride along Q
// (dim index rank-2) and out is unit-stride along L (dim index rank-1).
constexpr int kPermTileDoubles = 32 * 33;
__device__ __forceinline__ void permute_tiled_body(const PermK& p, const int tiles_q, const int tiles_l, const unsigned vb,
                                                   double* __restrict__ smem) {
    double (*tile)[33] = reinterpret_cast<double (*)[33]>(smem);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    long b = vb;
    const int tq = (int)(b % tiles_q);
    b /= tiles_q;
    const int tl = (int)(b % tiles_l);
    b /= tiles_l;
    long oi = 0, oo = 0;
#pragma unroll
    for (int d = 3; d >= 0; --d) {
        if (d < p.rank - 2) {
            const long q = b / p.dim[d];
            const long c = b - q * p.dim[d];
            b = q;
            oi += c * p.s_in[d];
            oo += c * p.s_out[d];
        }
    }
    const int dq = p.rank - 2, dl = p.rank - 1;
    const long q0 = (long)tq * 32, l0 = (long)tl * 32;
    const long nq = p.dim[dq], nl = p.dim[dl];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long ql = q0 + tx, ll = l0 + ty + 8 * j;
        if (ql < nq && ll < nl) tile[ty + 8 * j][tx] = p.in[oi + ql * p.s_in[dq] + ll * p.s_in[dl]];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long ll = l0 + tx, ql = q0 + ty + 8 * j;
        if (ql < nq && ll < nl) {
            double* o = p.out + oo + ql * p.s_out[dq] + ll * p.s_out[dl];
            double v = p.alpha * tile[tx][ty + 8 * j];
            if (p.beta != 0.0) v += p.beta * (*o);
            *o = v;
        }
    }
}
__global__ void __launch_bounds__(256) permute_tiled_kernel(const PermK p, int tiles_q, int tiles_l) {
    __shared__ double tile[kPermTileDoubles];
    permute_tiled_body(p, tiles_q, tiles_l, blockIdx.x, tile);
}

// ------------------------------------------------------------------------------------
// element-wise and reductions
// ------------------------------------------------------------------------------------
__global__ void mp2_amplitudes_kernel(double* __restrict__ t, const double* __restrict__ w,
                                      const double* __restrict__ eo, const double* __restrict__ ev, double shift,
                                      int no, int nv, long total) {
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        long r = idx;
        const int j = (int)(r % no); r /= no;
        const int i = (int)(r % no); r /= no;
        const int b = (int)(r % nv);
        const int a = (int)(r / nv);
        t[idx] = w[idx] / ((eo[i] + eo[j] - ev[a] - ev[b]) + shift);
    }
}

// t_out may alias t_in (in-place update), hence no __restrict__ on those two
// (a, b, i, j) digits of a linear index over [nv][nv][no][no], advanced by the grid stride with carries: the 64-bit
// divisions of a per-element decomposition (three of them, ~100 instructions each) made the element-wise kernels VALU-bound
struct Walk4 {
    int j, i, b, a, sj, si, sb, sa, no, nv;
    __device__ Walk4(long idx, long stride, int no_, int nv_) : no(no_), nv(nv_) {
        long r = idx;
        j = (int)(r % no); r /= no;
        i = (int)(r % no); r /= no;
        b = (int)(r % nv);
        a = (int)(r / nv);
        r = stride;
        sj = (int)(r % no); r /= no;
        si = (int)(r % no); r /= no;
        sb = (int)(r % nv);
        sa = (int)(r / nv);
    }
    __device__ __forceinline__ void step() {
        j += sj;
        int c = j >= no; j -= c ? no : 0;
        i += si + c;
        c = i >= no; i -= c ? no : 0;
        b += sb + c;
        c = b >= nv; b -= c ? nv : 0;
        a += sa + c;
    }
};

struct CcUpdateK {
    double* t; double* dt; const double* t_in; const double* r; const double* eo; const double* ev;
    double shift, delta;
    int no, nv, rank;
    long total;
};
__device__ __forceinline__ void cc_update_body(const CcUpdateK& k, const unsigned vb, const unsigned vgrid) {
    double* t = k.t;
    double* __restrict__ dt = k.dt;
    const double* t_in = k.t_in;
    const double* __restrict__ r_ = k.r;
    const double* __restrict__ eo = k.eo;
    const double* __restrict__ ev = k.ev;
    const double shift = k.shift, delta = k.delta;
    const int no = k.no, nv = k.nv, rank = k.rank;
    const long total = k.total;
    const long stride = (long)vgrid * blockDim.x;
    long idx = vb * (long)blockDim.x + threadIdx.x;
    if (rank == 4) {
        Walk4 w(idx, stride, no, nv);
        for (; idx < total; idx += stride, w.step()) {
            const double d = eo[w.i] + eo[w.j] - ev[w.a] - ev[w.b];
            const double inv = 1.0 / (d + shift);
            const double x = r_[idx] * inv;
            dt[idx] = x;
            t[idx] = t_in[idx] + delta * x;
        }
    } else {
        for (; idx < total; idx += stride) {
            const int i = (int)(idx % no);
            const int a = (int)(idx / no);
            const double inv = 1.0 / (eo[i] - ev[a] + shift);
            const double x = r_[idx] * inv;
            dt[idx] = x;
            t[idx] = t_in[idx] + delta * x;
        }
    }
}
__global__ void cc_update_kernel(const CcUpdateK k) { cc_update_body(k, blockIdx.x, gridDim.x); }

constexpr int kDotBlocks = 1024;
struct DotPtrs {
    const double* x[16];
    const double* y[16];
    long n[16];
};
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double block_sum(double v, double* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) r += sh[k];
    __syncthreads();
    return r;   // valid on thread 0
}
// block -> (pair, chunk): the pairs of a DIIS / energy call mostly share one operand (the newest vector against the history), so the
// blocks that read the same chunk are neighbours in dispatch order AND on the same XCD (ids 8 apart): the shared chunk comes
// from HBM once and from that XCD's L2 for the other pairs (pair-major order re-reads it from HBM for every pair)
struct DotsK {
    DotPtrs p;
    double* partial;
    int nb, npairs;
};
template <int VEC>
__device__ __forceinline__ void dots_stage1_body(const DotsK& k, const unsigned vb, double* sh) {
    const DotPtrs& p = k.p;
    double* __restrict__ partial = k.partial;
    const int nb = k.nb, npairs = k.npairs;
    const int lin = vb;
    int pair, b;
    if ((nb & 7) == 0) { const int q = lin >> 3; pair = q % npairs; b = (q / npairs) * 8 + (lin & 7); }
    else { pair = lin / nb; b = lin - pair * nb; }
    const double* __restrict__ x = p.x[pair];
    const double* __restrict__ y = p.y[pair];
    const long n = p.n[pair];
    double s = 0.0;
    if (VEC == 2) {
        const double2* __restrict__ x2 = reinterpret_cast<const double2*>(x);
        const double2* __restrict__ y2 = reinterpret_cast<const double2*>(y);
        const long n2 = n >> 1;
        double s1 = 0.0;
#pragma unroll 4
        for (long i = b * 256L + threadIdx.x; i < n2; i += nb * 256L) {
            const double2 a = x2[i], c = y2[i];
            s += a.x * c.x;
            s1 += a.y * c.y;
        }
        s += s1;
        if ((n & 1) && b == 0 && threadIdx.x == 0) s += x[n - 1] * y[n - 1];
    } else {
#pragma unroll 4
        for (long i = b * 256L + threadIdx.x; i < n; i += nb * 256L)
            s += x[i] * y[i];       // read-only operands: the unrolled loads are issued together
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) partial[pair * kDotBlocks + b] = s;
}
template <int VEC>
__global__ void __launch_bounds__(256) dots_stage1_kernel(const DotsK k) {
    __shared__ double sh[4];
    dots_stage1_body<VEC>(k, blockIdx.x, sh);
}
struct Dots2K {
    const double* partial;
    double* out;
    int nblocks;
};
__device__ __forceinline__ void dots_stage2_body(const Dots2K& k, const unsigned vb, double* sh) {
    const int pair = vb;
    double s = 0.0;
    for (int i = threadIdx.x; i < k.nblocks; i += blockDim.x) s += k.partial[pair * kDotBlocks + i];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) k.out[pair] = s;
}
// The last stage of a reduction whose result the HOST waits for: one block per reduction sums its partial sums (the order of
// dots_stage2: bit-identical) and writes the result straight into pinned host memory; the block that arrives last — a
// counter of the launch's own, one of a ring of 64 that the last arriver leaves at zero again (launches of several streams
// may interleave) — writes, behind a system-scope fence, the sequence number the host polls for.  No copy kernel, no event, no stream synchronisation between the device's last store and
// the host's next launch (the DIIS round trip of an iteration: 40 us of idle device with hipStreamSynchronize).
struct DotsFinalK {
    const double* partial;
    double* out;                       // device address of the pinned result slot
    long* flag;                        // ... and of its sequence word
    unsigned int* arrivals;            // this launch's arrival counter (zero before, zero after)
    unsigned int pad_;
    long seq;
    int nblocks, npairs;
};
__device__ __forceinline__ void dots_final_body(const DotsFinalK& k, const unsigned vb, double* sh) {
    const int pair = vb;
    double s = 0.0;
    for (int i = threadIdx.x; i < k.nblocks; i += blockDim.x) s += k.partial[pair * kDotBlocks + i];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) {
        k.out[pair] = s;
        __threadfence_system();
        const unsigned int old = __hip_atomic_fetch_add(k.arrivals, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned int)k.npairs - 1u) {
            __hip_atomic_store(k.arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            __hip_atomic_store(k.flag, k.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ void __launch_bounds__(256) dots_final_kernel(const DotsFinalK k) {
    __shared__ double sh[4];
    dots_final_body(k, blockIdx.x, sh);
}
__global__ void __launch_bounds__(256) dots_stage2_kernel(const double* __restrict__ partial, int nblocks,
                                                          double* __restrict__ out) {
    __shared__ double sh[4];
    dots_stage2_body(Dots2K{partial, out, nblocks}, blockIdx.x, sh);
}

// the small algebra of a DIIS step on the device: one thread (the matrices are at most 9 x 9), overlaps straight from
// dots_stage2's output — no host round trip between the overlaps and the extrapolation
__global__ void __launch_bounds__(64) diis_step_kernel(double* __restrict__ state, const double* __restrict__ overlaps, int ntypes,
                                                        int m, int was_full) {
    // one wave; the matrices live in LDS.  Cyclic Jacobi with the lanes over the row / column index of a rotation (a
    // single thread with the matrices in scratch memory takes milliseconds: every element access is a memory round trip)
    __shared__ double sL[81], sA[81], sV[81], slam[9], swork[99], sOld[81];
    const int lane = threadIdx.x, n = m + 1;
    // L of this step (diis_small::build_L, diis.py:56-80) with the lanes over its elements: one thread walking the state in
    // global memory is a chain of ~250 dependent-latency accesses (0.3 ms)
    for (int e = lane; e < 81; e += 64) sOld[e] = state[1 + e];
    double snew = 0.0;
    if (lane < m)
        for (int t = 0; t < ntypes; ++t) snew += overlaps[t * m + lane];          // (the order of the reference's loop, :65-78)
    __syncthreads();
    for (int e = lane; e < 81; e += 64) {
        const int i = e / 9, j = e - 9 * i;
        double v = 0.0;
        if ((i == m && j < m) || (j == m && i < m)) v = -1.0;                     // :56-57
        if (was_full) { if (i < n - 3 && j < n - 3) v = sOld[(i + 1) * 9 + (j + 1)]; }   // :59-60 (quirk included)
        else if (i < n - 2 && j < n - 2) v = sOld[e];                             // :62
        sL[e] = v;
    }
    __syncthreads();
    if (lane < m) sL[lane * 9 + (m - 1)] += snew;
    __syncthreads();
    if (lane < n) sL[(m - 1) * 9 + lane] = sL[lane * 9 + (m - 1)];
    __syncthreads();
    if (lane == 0) state[0] = (double)n;
    for (int e = lane; e < 81; e += 64) state[1 + e] = sL[e];
    // Fast path (the usual case): L^-1 by Gauss-Jordan with partial pivoting, the lanes over the elements of [L | 1].  L is
    // symmetric, so |lambda_min| = 1 / ||L^-1||_2 >= 1 / (n max |L^-1_ij|): if that bound clears the reference's threshold
    // (|lambda| < 1e-12, diis.py:85) with a factor of two to spare, the pseudo-inverse branch is PROVABLY not taken and the
    // coefficients are L^-1 (0, ..., 0, -1) (diis.py:95) — a few microseconds.  Anything else (a pivot that vanishes, a
    // non-finite entry, a bound that does not clear) goes through the eigen-decomposition below, as before: a cyclic Jacobi
    // on one wave is a chain of LDS and fp64-divide latencies, 0.4 ms per step.
    {
        __shared__ double sG[9 * 18];
        for (int e = lane; e < 9 * 18; e += 64) {
            const int i = e / 18, j = e - 18 * i;
            sG[e] = (j < 9) ? sL[i * 9 + j] : ((j - 9 == i) ? 1.0 : 0.0);
        }
        __syncthreads();
        bool ok = true;
        for (int k = 0; k < n && ok; ++k) {
            double v = (lane >= k && lane < n) ? fabs(sG[lane * 18 + k]) : -1.0;
            int idx = lane;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                const double v2 = __shfl_xor(v, o, 64);
                const int i2 = __shfl_xor(idx, o, 64);
                if (v2 > v || (v2 == v && i2 < idx)) { v = v2; idx = i2; }
            }
            v = __shfl(v, 0, 64);
            idx = __shfl(idx, 0, 64);
            if (!(v > 0.0) || !(v <= 1.7976931348623157e308)) { ok = false; break; }       // (wave-uniform)
            if (idx != k && lane < 18) {
                const double t = sG[k * 18 + lane];
                sG[k * 18 + lane] = sG[idx * 18 + lane];
                sG[idx * 18 + lane] = t;
            }
            __syncthreads();
            const double piv = sG[k * 18 + k];
            __syncthreads();
            if (lane < 18) sG[k * 18 + lane] /= piv;
            __syncthreads();
            double fac[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int e = lane + 64 * q, i = e / 18;
                fac[q] = (e < n * 18 && i != k) ? sG[i * 18 + k] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int e = lane + 64 * q, i = e / 18, j = e - 18 * i;
                if (e < n * 18 && i != k) sG[e] -= fac[q] * sG[k * 18 + j];
            }
            __syncthreads();
        }
        if (ok) {
            double big = 0.0;
            for (int e = lane; e < n * 18; e += 64) {
                const int i = e / 18, j = e - 18 * i;
                if (j >= 9 && j - 9 < n) {
                    const double a = fabs(sG[e]);
                    big = (a > big || !(a == a)) ? a : big;             // (a NaN wins: the bound below then fails)
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double b2 = __shfl_xor(big, o, 64);
                big = (b2 > big || !(b2 == b2)) ? b2 : big;
            }
            if (big == big && (double)n * big < 0.5e12) {                // |lambda_min| > 2e-12: the inverse branch, for sure
                if (lane < 9) state[82 + lane] = lane < n ? -sG[lane * 18 + 9 + (n - 1)] : 0.0;
                if (lane == 0) { state[91] = 0.0; state[92] += 1.0; }
                return;
            }
        }
        __syncthreads();
    }
    for (int e = lane; e < 81; e += 64) {
        sA[e] = sL[e];
        sV[e] = (e / 9 == e % 9) ? 1.0 : 0.0;
    }
    __syncthreads();

    // (stopping rule of diis_small::jacobi_eigh: elements below 1e-17 of the norm are left alone, a sweep without a
    // rotation ends the iteration — a test on the sum of the off-diagonal squares never fires near convergence)
    double tot = 0.0;
    for (int e = lane; e < 81; e += 64) {
        const int i = e / 9, j = e - 9 * i;
        if (i < n && j < n) tot += sA[e] * sA[e];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
    const double tiny = 1e-17 * sqrt(tot);
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotations = 0;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = sA[p * 9 + q];
                if (!(apq > tiny || apq < -tiny) && apq == apq) continue;   // (wave-uniform: every lane reads the same word)
                ++rotations;
                const double theta = (sA[q * 9 + q] - sA[p * 9 + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                __syncthreads();                                 // everyone has read the pivot elements
                if (lane < n) {                                  // A <- A J and V <- V J: lane = row
                    const double akp = sA[lane * 9 + p], akq = sA[lane * 9 + q];
                    sA[lane * 9 + p] = c * akp - sn * akq;
                    sA[lane * 9 + q] = sn * akp + c * akq;
                    const double vkp = sV[lane * 9 + p], vkq = sV[lane * 9 + q];
                    sV[lane * 9 + p] = c * vkp - sn * vkq;
                    sV[lane * 9 + q] = sn * vkp + c * vkq;
                }
                __syncthreads();
                if (lane < n) {                                  // A <- J^T A: lane = column
                    const double apk = sA[p * 9 + lane], aqk = sA[q * 9 + lane];
                    sA[p * 9 + lane] = c * apk - sn * aqk;
                    sA[q * 9 + lane] = sn * apk + c * aqk;
                }
                __syncthreads();
            }
        if (!rotations) break;
    }
    if (lane < n) slam[lane] = sA[lane * 9 + lane];
    __syncthreads();
    if (lane == 0) diis_small::finish(state, sL, sV, slam, n, swork);
}
struct LinPtrsDev {
    const double* x[8];
};
__global__ void lincomb_dev_kernel(double* __restrict__ out, const LinPtrsDev p, const double* __restrict__ coeff, int nx, long n) {
    double c[8];
    for (int k = 0; k < 8; ++k) c[k] = k < nx ? coeff[k] : 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int k = 0; k < nx; ++k) s += p.x[k][i] * c[k];
        out[i] = s;
    }
}

// five reductions in one pass over the amplitudes (device_api.h energy_norms); partial[q * kDotBlocks + block].
// The (a, b, i, j) digits of the element index advance by the digits of the grid stride with carries — four integer
// divisions per ELEMENT made this kernel VALU-bound (0.75 ms for 3.2 GB at (50,200)); VEC = 2: 16-byte loads over j pairs
// (nocc even: a pair never leaves its row).
struct EnergyK {
    const double *f, *t1, *t2, *Edir, *Eex, *dt2;
    double* partial;
    long total;
    int no, nv;
};
template <int VEC>
__device__ __forceinline__ void energy_norms_body(const EnergyK& k, const unsigned vb, const unsigned vgrid, double* sh) {
    const double* __restrict__ f = k.f;
    const double* __restrict__ t1 = k.t1;
    const double* __restrict__ t2 = k.t2;
    const double* __restrict__ Edir = k.Edir;
    const double* __restrict__ Eex = k.Eex;
    const double* __restrict__ dt2 = k.dt2;
    double* __restrict__ partial = k.partial;
    const int no = k.no, nv = k.nv;
    const long total = k.total;
    double s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s0 = 0.0, s5 = 0.0;
    const int noV = no / VEC;                       // j digit in units of VEC elements
    const long nvecs = total / VEC, stride = (long)vgrid * blockDim.x;
    long idx = vb * (long)blockDim.x + threadIdx.x;
    int j, i, b, a, sj, si, sb, sa;
    {
        long r = idx;
        j = (int)(r % noV); r /= noV;
        i = (int)(r % no); r /= no;
        b = (int)(r % nv);
        a = (int)(r / nv);
        r = stride;
        sj = (int)(r % noV); r /= noV;
        si = (int)(r % no); r /= no;
        sb = (int)(r % nv);
        sa = (int)(r / nv);
    }
#pragma unroll 4
    for (; idx < nvecs; idx += stride) {
        if (VEC == 2) {
            const double2 x = reinterpret_cast<const double2*>(t2)[idx];
            const double2 ed = reinterpret_cast<const double2*>(Edir)[idx], ex = reinterpret_cast<const double2*>(Eex)[idx];
            double tx = x.x, ty = x.y;
            if (t1) {
                const double ta = t1[a * no + i];
                tx += ta * t1[b * no + 2 * j];
                ty += ta * t1[b * no + 2 * j + 1];
            }
            s1 += tx * ed.x + ty * ed.y;
            s2 += tx * ex.x + ty * ex.y;
            s3 += x.x * x.x + x.y * x.y;
            if (dt2) { const double2 d = reinterpret_cast<const double2*>(dt2)[idx]; s4 += d.x * d.x + d.y * d.y; }
        } else {
            const double x = t2[idx];
            double tau = x;
            if (t1) tau += t1[a * no + i] * t1[b * no + j];
            s1 += tau * Edir[idx];
            s2 += tau * Eex[idx];
            s3 += x * x;
            if (dt2) { const double d = dt2[idx]; s4 += d * d; }
        }
        j += sj;
        int c = j >= noV; j -= c ? noV : 0;
        i += si + c;
        c = i >= no; i -= c ? no : 0;
        b += sb + c;
        c = b >= nv; b -= c ? nv : 0;
        a += sa + c;
    }
    if (t1 && f) {
        const long n = no + nv, ov = (long)no * nv;
        for (long e = vb * (long)blockDim.x + threadIdx.x; e < ov; e += (long)vgrid * blockDim.x) {
            const long a = e / no, i = e - a * no;
            const double y = t1[e];
            s0 += f[i * n + no + a] * y;
            s5 += y * y;
        }
    }
    const double r0 = block_sum(s0, sh), r1 = block_sum(s1, sh), r2 = block_sum(s2, sh), r3 = block_sum(s3, sh),
                 r4 = block_sum(s4, sh), r5 = block_sum(s5, sh);
    if (threadIdx.x == 0) {
        partial[5 * kDotBlocks + vb] = r5;
        partial[0 * kDotBlocks + vb] = r0;
        partial[1 * kDotBlocks + vb] = r1;
        partial[2 * kDotBlocks + vb] = r2;
        partial[3 * kDotBlocks + vb] = r3;
        partial[4 * kDotBlocks + vb] = r4;
    }
}
template <int VEC>
__global__ void __launch_bounds__(256) energy_norms_kernel(const EnergyK k) {
    __shared__ double sh[4];
    energy_norms_body<VEC>(k, blockIdx.x, gridDim.x, sh);
}

// out[0] = max |A[p,q,r,s] - B[q,p,s,r]|, out[1] = max(|A|, |B|) as bit patterns (non-negative doubles order like integers).
// Persistent blocks walk over (p, q, 64 x 64 tile of (r,s)): the partner tile B[q,p,s0:,r0:] is read row-wise with 16-byte
// loads and transposed through LDS, the tile of A is read row-wise too, so both tensors stream with full lines; one pair of
// atomics per BLOCK (round 2 had one block and one atomic pair per 32 x 32 tile — 1.5 M blocks of 16 KB for the V_abcd
// check at (50,200): 45 ms for a 25-GB read).  A block that is its own partner (A == B: klij, ijab, abij, abcd) is
// walked over p <= q only — every element is still touched once, as A[p,q] or as B[q,p].
template <bool VEC>
__global__ void __launch_bounds__(256) exchange_asym_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                            long d0, long d1, long d2, long d3, int tr, int ts, long ntiles,
                                                            int self, unsigned long long* __restrict__ out) {
    __shared__ double tile[64][65];
    __shared__ double sh[8];
    double m1 = 0.0, m2 = 0.0;
    auto upd = [&](double a, double b) {
        const double d = fabs(a - b);
        m1 = (d > m1 || d != d) ? d : m1;          // a NaN difference must not pass as symmetric
        m2 = fmax(m2, fmax(fabs(a), fabs(b)));
    };
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        long bid = t;
        const int t_s = (int)(bid % ts); bid /= ts;
        const int t_r = (int)(bid % tr); bid /= tr;
        const long q = bid % d1, p = bid / d1;
        if (self && q < p) continue;                  // (block-uniform)
        const long r0 = (long)t_r * 64, s0 = (long)t_s * 64;
        const double* __restrict__ Apq = A + (p * d1 + q) * d2 * d3;      // [d2][d3]
        const double* __restrict__ Bqp = B + (q * d0 + p) * d3 * d2;      // [d3][d2]
        if constexpr (VEC) {       // d2, d3 even: pairs never straddle an edge
            const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
            for (int j = 0; j < 8; ++j) {       // tile[s - s0][r - r0] = B[q,p,s,r], rows of r contiguous
                const long sg = s0 + ty + 8 * j, rg = r0 + 2 * tx;
                if (sg < d3 && rg < d2) {
                    const v2d v = *reinterpret_cast<const v2d*>(Bqp + sg * d2 + rg);
                    tile[ty + 8 * j][2 * tx] = v[0];
                    tile[ty + 8 * j][2 * tx + 1] = v[1];
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long rg = r0 + ty + 8 * j, sg = s0 + 2 * tx;
                if (rg < d2 && sg < d3) {
                    const v2d a = *reinterpret_cast<const v2d*>(Apq + rg * d3 + sg);
                    upd(a[0], tile[2 * tx][ty + 8 * j]);
                    upd(a[1], tile[2 * tx + 1][ty + 8 * j]);
                }
            }
        } else {
            const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
            for (int j = 0; j < 16; ++j) {
                const long sg = s0 + ty + 4 * j, rg = r0 + tx;
                if (sg < d3 && rg < d2) tile[ty + 4 * j][tx] = Bqp[sg * d2 + rg];
            }
            __syncthreads();
            for (int j = 0; j < 16; ++j) {
                const long rg = r0 + ty + 4 * j, sg = s0 + tx;
                if (rg < d2 && sg < d3) upd(Apq[rg * d3 + sg], tile[tx][ty + 4 * j]);
            }
        }
        __syncthreads();                               // the tile is overwritten by the next pass
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o1 = __shfl_down(m1, off, 64), o2 = __shfl_down(m2, off, 64);
        m1 = (o1 > m1 || o1 != o1) ? o1 : m1;
        m2 = fmax(m2, o2);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sh[w] = m1; sh[4 + w] = m2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) { m1 = (sh[k] > m1 || sh[k] != sh[k]) ? sh[k] : m1; m2 = fmax(m2, sh[4 + k]); }
        if (m1 != m1) m1 = INFINITY;
        atomicMax(out, (unsigned long long)__double_as_longlong(m1));
        atomicMax(out + 1, (unsigned long long)__double_as_longlong(m2));
    }
}

struct LinPtrs {
    const double* x[8];
    double c[8];
};
struct LinK {
    double* out;
    LinPtrs p;
    long n;
    int nx;
};
__device__ __forceinline__ void lincomb_body(const LinK& k, const unsigned vb, const unsigned vgrid) {
    double* __restrict__ out = k.out;
    const int nx = k.nx;
    for (long i = vb * (long)blockDim.x + threadIdx.x; i < k.n; i += (long)vgrid * blockDim.x) {
        double s = 0.0;
        for (int q = 0; q < nx; ++q) s += k.p.x[q][i] * k.p.c[q];
        out[i] = s;
    }
}
__global__ void lincomb_kernel(const LinK k) { lincomb_body(k, blockIdx.x, gridDim.x); }

// ---- tall-skinny subspace algebra of the Davidson / FEAST drivers (eom_ccsd.py:91-147, :512-541) -------------------------
// The vectors are 13 M doubles at (30,120) and a pass needs their Gram blocks and a handful of linear combinations: one
// thread keeps an MM x NN block of accumulators (resp. NN outputs) in registers, so every vector of a call is read ONCE
// (HBM-bound: MM NN fused multiply-adds per 8 (MM + NN) bytes, far below the vector-ALU rate).  Unused slots of a template
// size point at slot 0 and are discarded (Gram) / carry a zero coefficient (combination).
constexpr int kGramBlocks = 1024;
template <int MM, int NN>
struct GramPtrs {
    const double* x[MM];
    const double* y[NN];
};
template <int MM, int NN, int VEC>
__global__ void __launch_bounds__(256) gram_stage1_kernel(const GramPtrs<MM, NN> p, long len, double* __restrict__ partial) {
    double acc[MM][NN];
#pragma unroll
    for (int i = 0; i < MM; ++i)
#pragma unroll
        for (int j = 0; j < NN; ++j) acc[i][j] = 0.0;
    typedef double vec_t __attribute__((ext_vector_type(VEC)));
    const long nvec = len / VEC;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < nvec; e += (long)gridDim.x * 256L) {
        vec_t xv[MM], yv[NN];
#pragma unroll
        for (int i = 0; i < MM; ++i) xv[i] = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(p.x[i]) + e);
#pragma unroll
        for (int j = 0; j < NN; ++j) yv[j] = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(p.y[j]) + e);
#pragma unroll
        for (int i = 0; i < MM; ++i)
#pragma unroll
            for (int j = 0; j < NN; ++j)
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[i][j] = fma(xv[i][q], yv[j][q], acc[i][j]);
    }
    if (VEC > 1 && blockIdx.x == 0 && threadIdx.x == 0)          // the elements past the last whole vector
        for (long e = nvec * VEC; e < len; ++e)
#pragma unroll
            for (int i = 0; i < MM; ++i)
#pragma unroll
                for (int j = 0; j < NN; ++j) acc[i][j] = fma(p.x[i][e], p.y[j][e], acc[i][j]);
    __shared__ double sh[4][MM * NN];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < MM; ++i)
#pragma unroll
        for (int j = 0; j < NN; ++j) {
            const double s = wave_sum(acc[i][j]);
            if (lane == 0) sh[w][i * NN + j] = s;
        }
    __syncthreads();
    if (threadIdx.x < MM * NN)
        partial[(long)blockIdx.x * (MM * NN) + threadIdx.x] =
            (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}
// out[idx] = sum over the blocks of partial[b][idx], in a fixed order (one block per output entry)
__global__ void __launch_bounds__(256) gram_stage2_kernel(const double* __restrict__ partial, int nblocks, int cnt,
                                                          double* __restrict__ out) {
    __shared__ double sh[4];
    const int idx = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x) s += partial[(long)b * cnt + idx];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) out[idx] = s;
}

template <int MM, int NN>
struct LinMulti {
    const double* x[MM];
    double* y[NN];
    double c[MM][NN];
    double beta[NN];
};
// y_j = sum_i c[i][j] x_i + beta_j y_j, j < n (n <= NN).  No __restrict__: an output may be one of the inputs (every element
// is read before it is written, by the same thread)
template <int MM, int NN, int VEC>
__global__ void __launch_bounds__(256) lincomb_multi_kernel(const LinMulti<MM, NN> p, int n, long len) {
    typedef double vec_t __attribute__((ext_vector_type(VEC)));
    const long nvec = len / VEC;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < nvec; e += (long)gridDim.x * 256L) {
        vec_t xv[MM], out[NN];
#pragma unroll
        for (int i = 0; i < MM; ++i) xv[i] = reinterpret_cast<const vec_t*>(p.x[i])[e];
#pragma unroll
        for (int j = 0; j < NN; ++j) {
            vec_t s = 0.0;
            if (j < n && p.beta[j] != 0.0) s = p.beta[j] * reinterpret_cast<const vec_t*>(p.y[j])[e];
            out[j] = s;
        }
#pragma unroll
        for (int i = 0; i < MM; ++i)
#pragma unroll
            for (int j = 0; j < NN; ++j) out[j] += p.c[i][j] * xv[i];
#pragma unroll
        for (int j = 0; j < NN; ++j)
            if (j < n) reinterpret_cast<vec_t*>(p.y[j])[e] = out[j];
    }
    if (VEC > 1 && blockIdx.x == 0 && threadIdx.x == 0)
        for (long e = nvec * VEC; e < len; ++e) {
            double xs[MM], out[NN];
#pragma unroll
            for (int i = 0; i < MM; ++i) xs[i] = p.x[i][e];
#pragma unroll
            for (int j = 0; j < NN; ++j) out[j] = (j < n && p.beta[j] != 0.0) ? p.beta[j] * p.y[j][e] : 0.0;
#pragma unroll
            for (int i = 0; i < MM; ++i)
#pragma unroll
                for (int j = 0; j < NN; ++j) out[j] += p.c[i][j] * xs[i];
#pragma unroll
            for (int j = 0; j < NN; ++j)
                if (j < n) p.y[j][e] = out[j];
        }
}

// diagonal preconditioner of the FEAST linear solves (feast_eom_ccsd.py:342: 1 / (z - diag + 0.01)) on a complex vector
__global__ void cmul_kernel(const double* __restrict__ mr, const double* __restrict__ mi, const double* xr, const double* xi,
                            double* yr, double* yi, long n) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const double a = xr[e], b = xi[e], c = mr[e], d = mi[e];
        yr[e] = c * a - d * b;
        yi[e] = c * b + d * a;
    }
}

// ---- particle ladder of a trial vector WITHOUT the exchange symmetry through two pair-packed ladders (eom.cpp) ------------------
// u = us + ua, us = (u + P u) / 2 (P u)_abij = u_baji; w_abij = sgn(i - j) ua_abij is exchange-symmetric again (zero for
// i == j), dg[a,b,i] = ua_abii is what w leaves out.  One thread per element; the partner read is a transposed access (once
// per build, 0.1 ms at (30,120)).
__global__ void exchange_split_kernel(const double* __restrict__ u, double* __restrict__ us, double* __restrict__ w,
                                      double* __restrict__ dg, int no, int nv, long total) {
    const long o = no, v = nv;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long j = e % o, i = (e / o) % o, b = (e / (o * o)) % v, a = e / (o * o * v);
        const double x = u[e], y = u[((b * v + a) * o + j) * o + i];
        const double ua = 0.5 * (x - y);
        us[e] = 0.5 * (x + y);
        w[e] = i > j ? ua : (i < j ? -ua : 0.0);
        if (i == j) dg[(a * v + b) * o + i] = ua;
    }
}
// D_abij += sgn(i - j) R_abij
__global__ void sgn_ij_add_kernel(double* __restrict__ D, const double* __restrict__ R, int no, long total) {
    const long o = no;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long j = e % o, i = (e / o) % o;
        if (i != j) D[e] += (i > j ? R[e] : -R[e]);
    }
}

// ---- EOM-CCSD diagonals (eom_ccsd.py:169-266), once per solve ------------------------------------------------------------------
// V = V_ijab [o,o,v,v], T [v,v,o,o].  Stage 1: the V.T sums with one to three free indices, one block per output element
// (strided loop over the summed indices + block reduction).  Layout of ws (offsets in doubles, EomDiagWs below):
//   S1[a,i]  = sum_jb (2V[j,i,b,a] - V[j,i,a,b]) (2T[b,a,j,i] - T[a,b,j,i])                      (:181-190)
//   Sa[a]    = sum_jkb (2V[j,k,b,a] - V[j,k,a,b]) T[a,b,j,k],    Si[i] = sum_jcb V[j,i,c,b] T[b,c,j,i]   (:192-196)
//   A2[a,i]  = sum_kc (2V[k,i,c,a] - 2V[k,i,a,c]) T[c,a,k,i] + (V[k,i,a,c] - 2V[k,i,c,a]) T[a,c,k,i] + sum_kcb V[k,i,c,b] T[a,c,k,i]
//   a_[a]    = sum_klc V[k,l,c,a] (T[a,c,k,l] - 2T[c,a,k,l]),    i_[i] = sum_kcd (V[k,i,d,c] - 2V[k,i,c,d]) T[c,d,k,i]
//   IJ[i,j]  = sum_cd V[i,j,c,d] T[c,d,i,j],   AJ[a,j] = sum_kc V[k,j,a,c] T[a,c,k,j],   AB[a,b] = sum_kl V[k,l,a,b] T[a,b,k,l]
//   Z1[a,i,j] = sum_kc V[k,j,a,c] T[c,a,k,i],  X1[a,b,j] = sum_k V[k,j,a,b] T[a,b,k,j]
struct EomDiagWs {
    long S1, Sa, Si, A2, a_, i_, IJ, AJ, AB, Z1, X1, total;
    __host__ __device__ EomDiagWs(long o, long v) {
        long p = 0;
        S1 = p; p += v * o;  Sa = p; p += v;  Si = p; p += o;  A2 = p; p += v * o;  a_ = p; p += v;  i_ = p; p += o;
        IJ = p; p += o * o;  AJ = p; p += v * o;  AB = p; p += v * v;  Z1 = p; p += v * o * o;  X1 = p; p += v * v * o;
        total = p;
    }
};
__global__ void __launch_bounds__(256) eom_diag_sums_kernel(const double* __restrict__ V, const double* __restrict__ T,
                                                            double* __restrict__ ws, int no, int nv) {
    __shared__ double sh[4];
    const long o = no, v = nv;
    const EomDiagWs w(o, v);
    auto Vx = [&](long k, long l, long c, long d) { return V[((k * o + l) * v + c) * v + d]; };
    auto Tx = [&](long a, long b, long i, long j) { return T[((a * v + b) * o + i) * o + j]; };
    const long e = blockIdx.x;
    const int t = threadIdx.x;
    double s = 0.0;
    if (e < w.Sa) {                                     // S1[a,i]
        const long a = e / o, i = e % o;
        for (long x = t; x < o * v; x += 256) {
            const long j = x / v, b = x % v;
            s += (2.0 * Vx(j, i, b, a) - Vx(j, i, a, b)) * (2.0 * Tx(b, a, j, i) - Tx(a, b, j, i));
        }
    } else if (e < w.Si) {                              // Sa[a]
        const long a = e - w.Sa;
        for (long x = t; x < o * o * v; x += 256) {
            const long j = x / (o * v), k = (x / v) % o, b = x % v;
            s += (2.0 * Vx(j, k, b, a) - Vx(j, k, a, b)) * Tx(a, b, j, k);
        }
    } else if (e < w.A2) {                              // Si[i]
        const long i = e - w.Si;
        for (long x = t; x < o * v * v; x += 256) {
            const long j = x / (v * v), c = (x / v) % v, b = x % v;
            s += Vx(j, i, c, b) * Tx(b, c, j, i);
        }
    } else if (e < w.a_) {                              // A2[a,i]
        const long a = (e - w.A2) / o, i = (e - w.A2) % o;
        for (long x = t; x < o * v; x += 256) {
            const long k = x / v, c = x % v;
            const double tca = Tx(c, a, k, i), tac = Tx(a, c, k, i);
            s += (2.0 * Vx(k, i, c, a) - 2.0 * Vx(k, i, a, c)) * tca + (Vx(k, i, a, c) - 2.0 * Vx(k, i, c, a)) * tac;
            double vb = 0.0;
            for (long b = 0; b < v; ++b) vb += Vx(k, i, c, b);
            s += vb * tac;
        }
    } else if (e < w.i_) {                              // a_[a]
        const long a = e - w.a_;
        for (long x = t; x < o * o * v; x += 256) {
            const long k = x / (o * v), l = (x / v) % o, c = x % v;
            s += Vx(k, l, c, a) * (Tx(a, c, k, l) - 2.0 * Tx(c, a, k, l));
        }
    } else if (e < w.IJ) {                              // i_[i]
        const long i = e - w.i_;
        for (long x = t; x < o * v * v; x += 256) {
            const long k = x / (v * v), c = (x / v) % v, d = x % v;
            s += (Vx(k, i, d, c) - 2.0 * Vx(k, i, c, d)) * Tx(c, d, k, i);
        }
    } else if (e < w.AJ) {                              // IJ[i,j]
        const long i = (e - w.IJ) / o, j = (e - w.IJ) % o;
        for (long x = t; x < v * v; x += 256) {
            const long c = x / v, d = x % v;
            s += Vx(i, j, c, d) * Tx(c, d, i, j);
        }
    } else if (e < w.AB) {                              // AJ[a,j]
        const long a = (e - w.AJ) / o, j = (e - w.AJ) % o;
        for (long x = t; x < o * v; x += 256) {
            const long k = x / v, c = x % v;
            s += Vx(k, j, a, c) * Tx(a, c, k, j);
        }
    } else if (e < w.Z1) {                              // AB[a,b]
        const long a = (e - w.AB) / v, b = (e - w.AB) % v;
        for (long x = t; x < o * o; x += 256) {
            const long k = x / o, l = x % o;
            s += Vx(k, l, a, b) * Tx(a, b, k, l);
        }
    } else if (e < w.X1) {                              // Z1[a,i,j]
        const long r = e - w.Z1, a = r / (o * o), i = (r / o) % o, j = r % o;
        for (long x = t; x < o * v; x += 256) {
            const long k = x / v, c = x % v;
            s += Vx(k, j, a, c) * Tx(c, a, k, i);
        }
    } else {                                            // X1[a,b,j]
        const long r = e - w.X1, a = r / (v * o), b = (r / o) % v, j = r % o;
        for (long k = t; k < o; k += 256) s += Vx(k, j, a, b) * Tx(a, b, k, j);
    }
    s = block_sum(s, sh);
    if (t == 0) ws[e] = s;
}
// Stage 2: d1[a,i] (:169-198) and, one thread per (a,b,i,j), d2 = e(a,b,i,j) + e(b,a,j,i) + ijij + IJ + AB + abab (:200-266) with
//   e(a,b,i,j) = ai[a,i] + a_[a] + i_[i] - 2 X1[a,b,j] - 2 IJ[i,j] + sum_k V[k,i,a,b] T[a,b,k,j] + sum_c V[i,j,c,a] T[c,b,i,j]
//                + Z1[a,i,j] + AJ[a,j],     ai = dai + iaai - 2 iaia + A2
__global__ void eom_diag_singles_kernel(const double* __restrict__ ws, const double* __restrict__ dai, const double* __restrict__ iaai,
                                        const double* __restrict__ iaia, double* __restrict__ d1, int no, int nv) {
    const long o = no, v = nv;
    const EomDiagWs w(o, v);
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= v * o) return;
    const long a = e / o, i = e % o;
    d1[e] = dai[e] + 2.0 * iaai[e] - iaia[e] + ws[w.S1 + e] - ws[w.Sa + a] - ws[w.Si + i];
}
__global__ void __launch_bounds__(256) eom_diag_doubles_kernel(const double* __restrict__ V, const double* __restrict__ T,
                                                               const double* __restrict__ ws, const double* __restrict__ dai,
                                                               const double* __restrict__ iaai, const double* __restrict__ iaia,
                                                               const double* __restrict__ ijij, const double* __restrict__ abab,
                                                               double* __restrict__ d2, int no, int nv, long total) {
    const long o = no, v = nv;
    const EomDiagWs w(o, v);
    auto Vx = [&](long k, long l, long c, long d) { return V[((k * o + l) * v + c) * v + d]; };
    auto Tx = [&](long a, long b, long i, long j) { return T[((a * v + b) * o + i) * o + j]; };
    auto half = [&](long a, long b, long i, long j) {
        const long ai = a * o + i;
        double r = dai[ai] + iaai[ai] - 2.0 * iaia[ai] + ws[w.A2 + ai] + ws[w.a_ + a] + ws[w.i_ + i] -
                   2.0 * ws[w.X1 + (a * v + b) * o + j] - 2.0 * ws[w.IJ + i * o + j] + ws[w.Z1 + (a * o + i) * o + j] +
                   ws[w.AJ + a * o + j];
        double y = 0.0;
        for (long k = 0; k < o; ++k) y += Vx(k, i, a, b) * Tx(a, b, k, j);
        for (long c = 0; c < v; ++c) y += Vx(i, j, c, a) * Tx(c, b, i, j);
        return r + y;
    };
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long j = e % o, i = (e / o) % o, b = (e / (o * o)) % v, a = e / (o * o * v);
        d2[e] = half(a, b, i, j) + half(b, a, j, i) + ijij[i * o + j] + ws[w.IJ + i * o + j] + ws[w.AB + a * v + b] + abab[a * v + b];
    }
}
// (mr + i mi)[e] = 1 / (z - hs d[e] + shift): the diagonal preconditioner of the FEAST linear solves from the device-resident
// diagonal (feast_eom_ccsd.py:276-278, :342)
__global__ void cshift_inv_kernel(const double* __restrict__ d, double zr, double zi, double hr, double hi, double shift,
                                  double* __restrict__ mr, double* __restrict__ mi, long n) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const double x = zr - hr * d[e] + shift, y = zi - hi * d[e];
        const double q = x * x + y * y;
        mr[e] = x / q;
        mi[e] = -y / q;
    }
}

struct TauK {
    double* tau; const double* t2; const double* t1;
    long total;
    int no, nv;
};
__device__ __forceinline__ void tau_body(const TauK& k, const unsigned vb, const unsigned vgrid) {
    double* __restrict__ tau = k.tau;
    const double* __restrict__ t2 = k.t2;
    const double* __restrict__ t1 = k.t1;
    const int no = k.no, nv = k.nv;
    const long total = k.total;
    for (long idx = vb * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)vgrid * blockDim.x) {
        long r = idx;
        const int j = (int)(r % no); r /= no;
        const int i = (int)(r % no); r /= no;
        const int b = (int)(r % nv);
        const int a = (int)(r / nv);
        tau[idx] = t2[idx] + t1[a * no + i] * t1[b * no + j];
    }
}
__global__ void tau_kernel(const TauK k) { tau_body(k, blockIdx.x, gridDim.x); }

// ------------------------------------------------------------------------------------
// symmetry-packed ladder helpers
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void unrank_pair(long r, int& x, int& y) {   // r = x(x+1)/2 + y, x >= y
    long xx = (long)((sqrt(8.0 * (double)r + 1.0) - 1.0) * 0.5);
    while (xx * (xx + 1) / 2 > r) --xx;
    while ((xx + 1) * (xx + 2) / 2 <= r) ++xx;
    x = (int)xx;
    y = (int)(r - xx * (xx + 1) / 2);
}

__global__ void __launch_bounds__(256) ladder_pack_V_kernel(const double* __restrict__ V, double* __restrict__ Vp,
                                                            double* __restrict__ Vm, int nr, int nv, long rp0,
                                                            int nt, long ntp, long npp, long npm) {     // npp, npm: row pitches
    __shared__ double sA[32][33], sB[32][33];
    const long bid = blockIdx.x;
    const long row = bid / ntp;               // local pair row
    const long tp = bid - row * ntp;          // tile pair (tc >= td)
    int a = 1, b = 0, tc, td;
    if (nr > 0) unrank_pair(rp0 + row, a, b);
    unrank_pair(tp, tc, td);
    const double* __restrict__ Vab = V + (nr > 0 ? (long)a * nr + b : rp0 + row) * nv * nv;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = tc * 32, d0 = td * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int y = ty + 8 * j;
        if (c0 + y < nv && d0 + tx < nv) sA[y][tx] = Vab[(long)(c0 + y) * nv + d0 + tx];     // V[a,b,c,d]
        if (d0 + y < nv && c0 + tx < nv) sB[y][tx] = Vab[(long)(d0 + y) * nv + c0 + tx];     // V[a,b,d,c]
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int y = ty + 8 * j;
        const int c = c0 + y, d = d0 + tx;
        if (c < nv && d < nv && c >= d) {
            const double x1 = sA[y][tx], x2 = sB[tx][y];
            Vp[row * npp + (long)c * (c + 1) / 2 + d] = x1 + x2;
            if (c > d) Vm[row * npm + (long)c * (c - 1) / 2 + d] = (a > b) ? x1 - x2 : 0.0;   // zero rows for a == b
        }
    }
}

// T1 dressing of the BRA of the pair-packed V_abcd (ccsd.py:414-419 restricted to what the packed ladder reads):
//   W[P(a,b)][cd] = V[P(a,b)][cd] - sum_k t[a,k] Pk[(b,k)][cd] + sgn sum_k t[b,k] Pk[(a,k)][cd],     a >= b,
// Pk = the rows (x,k) (x slow) of V_kxcd packed over (c,d) like V itself; sgn = -1 for the symmetric half (V_alcd + V_aldc =
// V_lacd + V_ladc), +1 for the antisymmetric one (V_alcd - V_aldc = -(V_lacd - V_ladc)), whose rows a == b stay zero.
// Two rank-no updates whose right factor depends on ONE index of the pair: the first is a matrix product for a fixed b
// (rows a), the second for a fixed a (rows b).  One wave (= one block) owns a 16 x 16 tile of (a,b) and 16 columns (c,d):
//   second term, per a of the tile ("chain"):  D2_a[b][cd] = sum_k (sgn t[b,k]) Pk[(a,k)][cd]      13 MFMA 16x16x4 for no = 50
//   first term,  per b of the tile:            out_b[a][cd] = V + D2 - sum_k t[a,k] Pk[(b,k)][cd]   (V and D2 enter as C)
// The MFMA result has its tile row in (lane >> 4) + 4 reg and the column (c,d) in lane & 15, so D2 comes out with b where
// the first term has a: it is turned through LDS (34 KB per wave: written with b, read with a in the lane group).  Every global access is a 128-byte row segment
// (16 lanes x 8 B), addressed as a wave-uniform base + a loop-invariant 32-bit lane offset; the right factors are read
// straight from global memory (each segment serves 16 MFMA rows).  The tile state fills the register file (one wave per
// SIMD), so the latencies are covered inside the wave: chains run in PAIRS with independent accumulators (no
// back-to-back dependent MFMAs), and the right factors and V rows of the pairs AHEAD are loaded before the MFMAs of the
// current one.  The t fragments stay in registers (the tile state, not the wave count, is what fills a SIMD: 4 waves per CU).  Blocks are dealt to the 8 XCDs round-robin: XCD x takes the column
// blocks x, x + 8, ... and runs all tile pairs of a column block back to back, so the slab of Pk they share (no nv rows
// x 128 B = 1.3 MB at (50,200)) stays in the 4-MB L2 of that XCD while V streams past it (non-temporal loads / stores).
// Pad columns up to the 16-double pitch are processed like any other (never read by the GEMMs).  NK = ceil(no / 4) MFMA
// steps; the k beyond no are zero on the t side and a clamped (finite) row on the Pk side.
// -t1 in MFMA A-operand order, one 512-byte fragment per (tile of 16 rows, step kk): [tile][kk][lane] holds
// -t1[16 tile + (lane & 15)][4 kk + (lane >> 4)], zero beyond nv / no.  Staged once per call so that the tile kernel
// fetches its 2 NK fragments with plain coalesced loads (no clamps, no selects: all in flight together).
__global__ void ladder_dress_tfrag_kernel(const double* __restrict__ t1, double* __restrict__ tf, int no, int nv, int nk, long total) {
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int L = (int)(e & 63), kk = (int)((e >> 6) % nk), tile = (int)((e >> 6) / nk);
    const int a = 16 * tile + (L & 15), k = 4 * kk + (L >> 4);
    tf[e] = (a < nv && k < no) ? -t1[(long)a * no + k] : 0.0;
}

template <int NK>
__global__ void __launch_bounds__(64) ladder_dress_kernel(const double* __restrict__ V, const double* __restrict__ Pk,
                                                          const double* __restrict__ tf, double* __restrict__ W, int no,
                                                          int nv, long ld, long row0, long row1, double sgn, long ntp) {
    constexpr int TP = 16 * 16 + 16;               // pitch of one a of the turn buffer (bank spread of the reads)
    __shared__ double turn[16 * TP];               // second term, [a][b][cd]: written with b, read with a in the lane group
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const long idx = blockIdx.x >> 3;
    const long cdb = (blockIdx.x & 7) + 8 * (idx / ntp), tp = idx % ntp;
    if (cdb * 16 >= ld) return;
    int ta_, tb_;
    unrank_pair(tp, ta_, tb_);
    const int a0 = ta_ * 16, b0 = tb_ * 16;
    const long tri0 = (long)a0 * (a0 + 1) / 2;
    if (tri0 + (long)15 * a0 + 120 + b0 + 15 < row0 || tri0 + b0 >= row1) return;     // rows of the tile: [P(a0,b0), P(a0+15,b0+15)]
    // MFMA A operand: lane holds [i = lane & 15][k = 4 kk + (lane >> 4)] of -t1, staged in that order (tf).  Both terms
    // run with -t: the second one enters the first as V -+ D2 (sgn decides) when it is read back from the turn buffer.
    double ta[NK], tb[NK];
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
        ta[kk] = tf[((long)ta_ * NK + kk) * 64 + lane];
        tb[kk] = tf[((long)tb_ * NK + kk) * 64 + lane];
    }
    // MFMA B operand, step kk, partner x: Pk[(x no + 4 kk + g) ld + col] (the no rows of a partner are consecutive: one
    // chain reads 8 MB of address space, not no pages).  All global accesses are raw buffer operations: resource base =
    // first row of the tile's partners (Pk) / of the tile (V, W), a wave-uniform soffset picks the partner and the step,
    // a loop-invariant 32-bit voffset the lane — no vector address arithmetic and no branch in the chains; a voffset at
    // or above the 2-GB record count switches a lane off (loads return 0, stores are dropped): rows that do not exist.
    // Only the last step can run past no: its lanes are clamped to row no - 1 (their t is zero).
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    constexpr unsigned kOff = 0x80000000u;          // num_records, and the "lane off" voffset
    const long col = cdb * 16 + c;
    const unsigned lo = (unsigned)(8 * ((long)g * ld + col));
    const unsigned lo_last = (unsigned)(8 * ((long)(min(4 * (NK - 1) + g, no - 1) - 4 * (NK - 1)) * ld + col));
    const unsigned ldb = (unsigned)(8 * ld);       // row pitch in bytes (host: tile extents stay below 2 GB)
    const __amdgpu_buffer_rsrc_t rPa = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Pk + (long)a0 * no * ld), 0, kOff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rPb = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Pk + (long)b0 * no * ld), 0, kOff, 0x00020000);
    // rows of V / W: P(a0 + g + 4 r, bb) - row0 = (tri0 - row0) [base] + bb [soffset] + P(a0 + g + 4 r, 0) - tri0 [voffset]
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(V + (tri0 - row0) * ld), 0, kOff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(W + (tri0 - row0) * ld, 0, kOff, 0x00020000);
    unsigned vo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long a = a0 + g + 4 * r;
        vo[r] = a < nv ? (unsigned)(8 * ((a * (a + 1) / 2 - tri0) * ld + col)) : kOff;
    }
    auto row_off = [&](int r, int x) {              // voffset of row (a0 + g + 4 r, x), or "off"
        const int a = a0 + g + 4 * r;
        const long row = (long)a * (a + 1) / 2 + x;
        return (x <= a && row >= row0 && row < row1) ? vo[r] : kOff;
    };
    // Chains 0..15: second term, partner a0 + n; 16..31: first term, partner b0 + n - 16; run in pairs.  Register sets of
    // the right factors: PA pairs ahead of the current one; of the V rows (HBM latency, non-temporal): VA pairs ahead.
    constexpr int PA = 1, NBP = 2 * (PA + 1), VA = 3, NBV = 2 * (VA + 1);
    double pf[NBP][NK], vin[NBV][4];
    auto fetch_p = [&](int n) {
        const int x0 = n < 16 ? a0 : b0, i = n < 16 ? n : n - 16;
        const unsigned so = (unsigned)(min(x0 + i, nv - 1) - x0) * (unsigned)no * ldb;
        const __amdgpu_buffer_rsrc_t rs = n < 16 ? rPa : rPb;
#pragma unroll
        for (int kk = 0; kk < NK - 1; ++kk)
            pf[n % NBP][kk] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, lo, so + 4u * kk * ldb, 0));
        pf[n % NBP][NK - 1] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, lo_last, so + 4u * (NK - 1) * ldb, 0));
    };
    auto fetch_v = [&](int n) {
        const int x = b0 + n - 16;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            vin[n % NBV][r] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rV, row_off(r, x), (unsigned)x * ldb, 2));
    };
#pragma unroll
    for (int n = 0; n < 2 * PA; ++n) fetch_p(n);
#pragma unroll
    for (int p = 0; p < 8; ++p) {                   // second term, chains 2p and 2p + 1: a = a0 + n
        const int n0 = 2 * p, n1 = n0 + 1;
        fetch_p(n0 + 2 * PA);
        fetch_p(n1 + 2 * PA);
        if (n0 + 2 * VA >= 16) { fetch_v(n0 + 2 * VA); fetch_v(n1 + 2 * VA); }
        v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(tb[kk], pf[n0 % NBP][kk], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(tb[kk], pf[n1 % NBP][kk], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {            // result rows are b = b0 + g + 4 rb
            turn[n0 * TP + (g + 4 * rb) * 16 + c] = acc0[rb];
            turn[n1 * TP + (g + 4 * rb) * 16 + c] = acc1[rb];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const double s2 = sgn > 0.0 ? -1.0 : 1.0;      // turn holds -sum_k t[b,k] Pk[(a,k)]: wanted is sgn times the sum
#pragma unroll
    for (int p = 0; p < 8; ++p) {                   // first term, partners b0 + 2p and b0 + 2p + 1; rows a = a0 + g + 4 r
        const int b = 2 * p, n0 = 16 + b, n1 = n0 + 1;
        if (n0 + 2 * PA < 32) { fetch_p(n0 + 2 * PA); fetch_p(n1 + 2 * PA); }
        if (n0 + 2 * VA < 32) { fetch_v(n0 + 2 * VA); fetch_v(n1 + 2 * VA); }
        v4d acc0, acc1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc0[r] = fma(s2, turn[(g + 4 * r) * TP + b * 16 + c], vin[n0 % NBV][r]);
            acc1[r] = fma(s2, turn[(g + 4 * r) * TP + (b + 1) * 16 + c], vin[n1 % NBV][r]);
        }
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[kk], pf[n0 % NBP][kk], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[kk], pf[n1 % NBP][kk], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int bb = b0 + b + h;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double val = (sgn > 0.0 && a0 + g + 4 * r == bb) ? 0.0 : (double)(h ? acc1[r] : acc0[r]);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, val), rW, row_off(r, bb), (unsigned)bb * ldb, 2);
            }
        }
    }
}

struct PackTK {
    const double* T; const double* t1; double* Sp; double* Am;
    long ldp, ldm, rp0;
    int no, nv, flags;
};
__device__ __forceinline__ void ladder_pack_T_body(const PackTK& k, const unsigned vb) {
    const double* __restrict__ T = k.T;
    const double* __restrict__ t1 = k.t1;
    double* __restrict__ Sp = k.Sp;
    double* __restrict__ Am = k.Am;
    const int no = k.no, nv = k.nv, flags = k.flags;
    const long ldp = k.ldp, ldm = k.ldm;
    const long row = k.rp0 + vb;   // P(c,d)
    int c, d;
    unrank_pair(row, c, d);
    const bool row_half = flags & dev::PACK_ROW_HALF, prow = flags & dev::PACK_AM_PROWS,
               col_half = flags & dev::PACK_COL_HALF, pcol = flags & dev::PACK_AM_PCOLS;
    const long o2 = (long)no * no;
    const double* __restrict__ T1 = T ? T + ((long)c * nv + d) * o2 : nullptr;
    const double* __restrict__ T2 = T ? T + ((long)d * nv + c) * o2 : nullptr;
    const double fr = (c == d && row_half) ? 0.25 : 0.5;
    const long mrow = prow ? row : (long)c * (c - 1) / 2 + d;
    const bool has_m = prow || c > d;
    if (threadIdx.x == 0) {      // pad columns of an even pitch are part of the GEMM K range when this is an A operand
        const long opp = (long)no * (no + 1) / 2, mcols = pcol ? opp : opp - no;
        if (ldp > opp) Sp[row * ldp + opp] = 0.0;
        if (has_m && ldm > mcols) Am[mrow * ldm + mcols] = 0.0;
    }
    for (int e = threadIdx.x; e < o2; e += blockDim.x) {
        const int i = e / no, j = e - i * no;
        if (i < j) continue;
        double x1 = 0.0, x2 = 0.0;
        if (T) { x1 = T1[e]; x2 = T2[e]; }
        if (t1) {
            x1 += t1[(long)c * no + i] * t1[(long)d * no + j];
            x2 += t1[(long)d * no + i] * t1[(long)c * no + j];
        }
        Sp[row * ldp + (long)i * (i + 1) / 2 + j] = ((i == j && col_half) ? 0.5 * fr : fr) * (x1 + x2);
        if (has_m && (pcol || i > j)) {
            const long mcol = pcol ? (long)i * (i + 1) / 2 + j : (long)i * (i - 1) / 2 + j;
            Am[mrow * ldm + mcol] = (c > d && i > j) ? 0.5 * (x1 - x2) : 0.0;
        }
    }
}
__global__ void ladder_pack_T_kernel(const PackTK k) { ladder_pack_T_body(k, blockIdx.x); }

// L[P(a,b)][0:opp] = LS, L[P(a,b)][opp:opp+opm] = LA (rows of diagonal pairs carry zeros there)
struct UnpackK {
    const double* L; double* R;
    double beta;
    long total;
    int no, nv;
};
__device__ __forceinline__ void ladder_unpack_body(const UnpackK& k, const unsigned vb, const unsigned vgrid) {
    const double* __restrict__ L = k.L;
    double* __restrict__ R = k.R;
    const double beta = k.beta;
    const int no = k.no, nv = k.nv;
    const long total = k.total;
    const long opp = (long)no * (no + 1) / 2, ld = (long)no * no;
    for (long idx = vb * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)vgrid * blockDim.x) {
        long r = idx;
        const int j = (int)(r % no); r /= no;
        const int i = (int)(r % no); r /= no;
        const int b = (int)(r % nv);
        const int a = (int)(r / nv);
        const int ah = max(a, b), al = min(a, b), ih = max(i, j), il = min(i, j);
        const double* __restrict__ row = L + ((long)ah * (ah + 1) / 2 + al) * ld;
        double v = row[(long)ih * (ih + 1) / 2 + il];
        if (a != b && i != j) {
            const double x = row[opp + (long)ih * (ih - 1) / 2 + il];
            v += ((a > b) == (i > j)) ? x : -x;
        }
        R[idx] = (beta != 0.0) ? beta * R[idx] + v : v;
    }
}
__global__ void ladder_unpack_kernel(const UnpackK k) { ladder_unpack_body(k, blockIdx.x, gridDim.x); }

// Right-hand operands of the ring builds (device_api.h ring_operands); one block per (k,b), chunks of CT c: the V_iajb tile
// [j][c] goes through LDS (read with c fastest, used with j fastest), the V_iabj tile [c][j] and both outputs move in the order
// of the threads (runs of nocc doubles per c)
struct RingOpK {
    const double* Wd; const double* Ud; double* M; double* N1;
    double a1, a2;
    int no, nv;
};
__device__ __forceinline__ void ring_operands_body(const RingOpK& q, const unsigned vb) {
    const double* __restrict__ Wd = q.Wd;
    const double* __restrict__ Ud = q.Ud;
    double* __restrict__ M = q.M;
    double* __restrict__ N1 = q.N1;
    const double a1 = q.a1, a2 = q.a2;
    const int no = q.no, nv = q.nv;
    extern __shared__ double tile[];          // [no][CT + 1]
    constexpr int CT = 32;                    // (16 / 32 / 64 measure the same: 0.88 ms for 3.2 GB at (50,200))
    const int k = vb / nv, b = vb - k * nv;
    const long ov = (long)no * nv;
    const double* __restrict__ wd = Wd + ((long)k * nv + b) * ((long)nv * no);       // [c][j]
    const double* __restrict__ ud = Ud + ((long)k * nv + b) * ((long)no * nv);       // [j][c]
    const long col = (long)b * no;
    for (int c0 = 0; c0 < nv; c0 += CT) {
        const int nc = min(CT, nv - c0);
        for (int e = threadIdx.x; e < no * CT; e += blockDim.x) {
            const int j = e / CT, cc = e - j * CT;
            if (cc < nc) tile[j * (CT + 1) + cc] = ud[(long)j * nv + c0 + cc];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < nc * no; e += blockDim.x) {
            const int cc = e / no, j = e - cc * no;
            const double u = tile[j * (CT + 1) + cc];
            const double w = wd[(long)(c0 + cc) * no + j];
            const long off = ((long)(c0 + cc) * no + k) * ov + col + j;
            N1[off] = -u;
            M[off] = a1 * w - a2 * u;
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) ring_operands_kernel(const RingOpK q) { ring_operands_body(q, blockIdx.x); }

// Pair layouts of exchange-symmetric-or-not amplitudes in one pass over T[a,b,i,j]; one block per (a,b):
//   Td[(a,i),(b,j)] = T_abij,  Tx[(a,j),(b,i)] = T_abij,  Ttd[(a,i),(b,j)] = ca T_abij + cb T_baij   (2, -1 in the residual)
struct LayoutsK {
    const double* T; double* Td; double* Tx; double* Ttd;
    double ca, cb;
    int no, nv;
};
template <bool RESIDUAL>     // RESIDUAL: (ca, cb) = (2, -1) as compile-time constants (the form every CCSD iteration runs)
__device__ __forceinline__ void t2_layouts_body(const LayoutsK& k, const unsigned vb) {
    const double* __restrict__ T = k.T;
    double* __restrict__ Td = k.Td;
    double* __restrict__ Tx = k.Tx;
    double* __restrict__ Ttd = k.Ttd;
    const int no = k.no, nv = k.nv;
    const double ca = k.ca, cb = k.cb;
    extern __shared__ double tile[];          // [no][no + 1]
    const int a = vb / nv, b = vb - a * nv;
    const long o2 = (long)no * no, ov = (long)no * nv;
    const double* __restrict__ Tab = T + ((long)a * nv + b) * o2;
    const double* __restrict__ Tba = T + ((long)b * nv + a) * o2;
    const long base = (long)a * no * ov + (long)b * no;       // element [(a,0),(b,0)] of a pair matrix
    for (int e = threadIdx.x; e < o2; e += blockDim.x) {
        const int i = e / no, j = e - i * no;
        const double x = Tab[e];
        tile[i * (no + 1) + j] = x;
        const long off = base + (long)i * ov + j;
        if (Td) Td[off] = x;
        Ttd[off] = RESIDUAL ? 2.0 * x - Tba[e] : ca * x + cb * Tba[e];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < o2; e += blockDim.x) {
        const int j = e / no, i = e - j * no;                   // Tx tile row j, column i
        Tx[base + (long)j * ov + i] = tile[i * (no + 1) + j];
    }
}
template <bool RESIDUAL>
__global__ void __launch_bounds__(256) t2_layouts_kernel(const LayoutsK k) { t2_layouts_body<RESIDUAL>(k, blockIdx.x); }

// R[a,b,:,:] and R[b,a,:,:] of the symmetry-reduced residual in one pass (one block per pair a >= b):
//   S[i][j] = N_ab[i][j] + N_ba[j][i] + D[(a,i),(b,j)] + D[(b,j),(a,i)] + X[(a,j),(b,i)] + X[(b,i),(a,j)]
//   R_ab = V_ab + unpack(L)_ab + S,   R_ba = V_ba + unpack(L)_ba + S^T
struct AssembleK {
    const double* V; const double* L; const double* N; const double* D; const double* X; double* R;   // V may be R
    double xd;
    int no, nv;
};
__device__ __forceinline__ void residual_assemble_body(const AssembleK& k, const unsigned vb) {
    const double* V = k.V;
    const double* __restrict__ L = k.L;
    const double* __restrict__ N = k.N;
    const double* __restrict__ D = k.D;
    const double* __restrict__ X = k.X;
    double* R = k.R;
    const int no = k.no, nv = k.nv;
    const double xd = k.xd;
    extern __shared__ double S[];             // [no][no + 1]
    int a, b;
    unrank_pair(vb, a, b);
    const int p = no + 1;
    const long o2 = (long)no * no, ov = (long)no * nv, opp = (long)no * (no + 1) / 2;
    const long ab = ((long)a * nv + b) * o2, ba = ((long)b * nv + a) * o2;
    const long tab = (long)a * no * ov + (long)b * no, tba = (long)b * no * ov + (long)a * no;
    for (int e = threadIdx.x; e < o2; e += blockDim.x) {      // sources read in (i,j) order
        const int i = e / no, j = e - i * no;
        double v = N[ab + e] + D[tab + (long)i * ov + j] + X[tba + (long)i * ov + j];
        if (xd != 0.0) v += xd * X[tab + (long)i * ov + j];      // X in the direct placement as well (see device_api.h)
        S[i * p + j] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < o2; e += blockDim.x) {      // sources read in (j,i) order
        const int j = e / no, i = e - j * no;
        double v = N[ba + e] + D[tba + (long)j * ov + i] + X[tab + (long)j * ov + i];
        if (xd != 0.0) v += xd * X[tba + (long)j * ov + i];
        S[i * p + j] += v;
    }
    __syncthreads();
    const double* __restrict__ row = L ? L + ((long)a * (a + 1) / 2 + b) * o2 : nullptr;
    for (int e = threadIdx.x; e < o2; e += blockDim.x) {
        const int i = e / no, j = e - i * no;
        const int ih = max(i, j), il = min(i, j);
        double ls = 0.0, la = 0.0;
        if (row) {
            ls = row[(long)ih * (ih + 1) / 2 + il];
            if (a != b && i != j) la = row[opp + (long)ih * (ih - 1) / 2 + il];
        }
        const double sgn = i > j ? 1.0 : -1.0;
        R[ab + e] = (V ? V[ab + e] : 0.0) + ls + sgn * la + S[i * p + j];
        if (a != b) R[ba + e] = (V ? V[ba + e] : 0.0) + ls - sgn * la + S[j * p + i];
    }
}
__global__ void __launch_bounds__(256) residual_assemble_kernel(const AssembleK k) { residual_assemble_body(k, blockIdx.x); }

// ---- pair-sharded tail of the iteration (one process per GPU): a rank owns the virtual pairs P(a,b) in [r0,r1), a >= b,
// and keeps the tiles X[a,b,:,:] and X[b,a,:,:] of every amplitude-sized quantity in the compact layout
// Xc[P - r0][2][o*o] (tile 1 is zero for a == b, so that dot products over Xc equal those over the full array) -----
// energy_norms over the compact tiles of the pairs [r0, r0 + npairs): block-stride over the tiles (pair, half)
__global__ void __launch_bounds__(256) energy_norms_pairs_kernel(const double* __restrict__ f, const double* __restrict__ t1,
                                                                 const double* __restrict__ tc, const double* __restrict__ Edir,
                                                                 const double* __restrict__ Eex, const double* __restrict__ dtc,
                                                                 int no, int nv, long r0, long npairs, int with_t1,
                                                                 double* __restrict__ partial) {
    __shared__ double sh[4];
    double s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s0 = 0.0, s5 = 0.0;
    const long o2 = (long)no * no;
    for (long tile = blockIdx.x; tile < 2 * npairs; tile += gridDim.x) {
        int a, b;
        unrank_pair(r0 + (tile >> 1), a, b);
        if (tile & 1) {
            if (a == b) continue;          // a diagonal pair has one tile only
            const int x = a; a = b; b = x;
        }
        const double* __restrict__ x = tc + tile * o2;
        const double* __restrict__ d = dtc ? dtc + tile * o2 : nullptr;
        const double* __restrict__ ed = Edir + ((long)a * nv + b) * o2;
        const double* __restrict__ ex = Eex + ((long)a * nv + b) * o2;
        for (int e = threadIdx.x; e < o2; e += blockDim.x) {
            const double v = x[e];
            double tau = v;
            if (t1) {
                const int i = e / no, j = e - i * no;
                tau += t1[a * no + i] * t1[b * no + j];
            }
            s1 += tau * ed[e];
            s2 += tau * ex[e];
            s3 += v * v;
            if (d) s4 += d[e] * d[e];
        }
    }
    if (with_t1 && t1 && f) {
        const long n = no + nv, ov = (long)no * nv;
        for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < ov; e += (long)gridDim.x * blockDim.x) {
            const long a = e / no, i = e - a * no;
            const double y = t1[e];
            s0 += f[i * n + no + a] * y;
            s5 += y * y;
        }
    }
    const double q0 = block_sum(s0, sh), q1 = block_sum(s1, sh), q2 = block_sum(s2, sh), q3 = block_sum(s3, sh),
                 q4 = block_sum(s4, sh), q5 = block_sum(s5, sh);
    if (threadIdx.x == 0) {
        partial[0 * kDotBlocks + blockIdx.x] = q0;
        partial[1 * kDotBlocks + blockIdx.x] = q1;
        partial[2 * kDotBlocks + blockIdx.x] = q2;
        partial[3 * kDotBlocks + blockIdx.x] = q3;
        partial[4 * kDotBlocks + blockIdx.x] = q4;
        partial[5 * kDotBlocks + blockIdx.x] = q5;
    }
}

__global__ void __launch_bounds__(256) pairs_pack_kernel(const double* __restrict__ full, double* __restrict__ Xc,
                                                         int no, int nv, long r0) {
    int a, b;
    unrank_pair(r0 + blockIdx.x, a, b);
    const long o2 = (long)no * no;
    const double* __restrict__ ab = full + ((long)a * nv + b) * o2;
    const double* __restrict__ ba = full + ((long)b * nv + a) * o2;
    double* __restrict__ out = Xc + (long)blockIdx.x * 2 * o2;
    for (int e = threadIdx.x; e < o2; e += blockDim.x) {
        out[e] = ab[e];
        out[o2 + e] = (a != b) ? ba[e] : 0.0;
    }
}
__global__ void __launch_bounds__(256) pairs_unpack_kernel(const double* __restrict__ Xc, double* __restrict__ full,
                                                           int no, int nv, long r0) {
    int a, b;
    unrank_pair(r0 + blockIdx.x, a, b);
    const long o2 = (long)no * no;
    const double* __restrict__ in = Xc + (long)blockIdx.x * 2 * o2;
    double* __restrict__ ab = full + ((long)a * nv + b) * o2;
    double* __restrict__ ba = full + ((long)b * nv + a) * o2;
    for (int e = threadIdx.x; e < o2; e += blockDim.x) {
        ab[e] = in[e];
        if (a != b) ba[e] = in[o2 + e];
    }
}
// ccsd.py:176-179 on the compact tiles: the denominator of T[b,a,i,j] is that of T[a,b,i,j]
__global__ void __launch_bounds__(256) cc_update_pairs_kernel(double* __restrict__ tc, double* __restrict__ dtc,
                                                              const double* __restrict__ rc, const double* __restrict__ eo,
                                                              const double* __restrict__ ev, double shift, double delta,
                                                              int no, long r0) {
    int a, b;
    unrank_pair(r0 + blockIdx.x, a, b);
    const long o2 = (long)no * no, base = (long)blockIdx.x * 2 * o2;
    const double eab = ev[a] + ev[b];
    for (int e = threadIdx.x; e < 2 * o2; e += blockDim.x) {
        const int t = e >= o2 ? e - (int)o2 : e;
        const int i = t / no, j = t - i * no;
        const double x = rc[base + e] * (1.0 / (eo[i] + eo[j] - eab + shift));
        dtc[base + e] = x;
        tc[base + e] += delta * x;
    }
}
// residual_assemble for the pairs [r0,r1) with compact output; Np[a - a0][b][o*o] already holds the complete
// X_ac / Q_kb combination N_ab + N_ba^T of the pair (its rows a are what the owner of the pair computes)
__global__ void __launch_bounds__(256) residual_assemble_pairs_kernel(const double* __restrict__ V, const double* __restrict__ L,
                                                                      const double* __restrict__ Np, const double* __restrict__ D,
                                                                      const double* __restrict__ X, double* __restrict__ Rc,
                                                                      int no, int nv, long r0, int a0, int nbp, double xd) {
    extern __shared__ double S[];             // [no][no + 1]
    int a, b;
    unrank_pair(r0 + blockIdx.x, a, b);
    const int p = no + 1;
    const long o2 = (long)no * no, ov = (long)no * nv, opp = (long)no * (no + 1) / 2;
    const long ab = ((long)a * nv + b) * o2, ba = ((long)b * nv + a) * o2;
    const long tab = (long)a * no * ov + (long)b * no, tba = (long)b * no * ov + (long)a * no;
    const double* __restrict__ n = Np + ((long)(a - a0) * nbp + b) * o2;       // Np is [a1 - a0][nbp][o*o], nbp > b
    for (int e = threadIdx.x; e < o2; e += blockDim.x) {
        const int i = e / no, j = e - i * no;
        double v = n[e] + D[tab + (long)i * ov + j] + X[tba + (long)i * ov + j];
        if (xd != 0.0) v += xd * X[tab + (long)i * ov + j];
        S[i * p + j] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < o2; e += blockDim.x) {
        const int j = e / no, i = e - j * no;
        double v = D[tba + (long)j * ov + i] + X[tab + (long)j * ov + i];
        if (xd != 0.0) v += xd * X[tba + (long)j * ov + i];
        S[i * p + j] += v;
    }
    __syncthreads();
    const double* __restrict__ row = L ? L + ((long)a * (a + 1) / 2 + b) * o2 : nullptr;
    double* __restrict__ out = Rc + (long)blockIdx.x * 2 * o2;
    for (int e = threadIdx.x; e < o2; e += blockDim.x) {
        const int i = e / no, j = e - i * no;
        const int ih = max(i, j), il = min(i, j);
        double ls = 0.0, la = 0.0;
        if (row) {
            ls = row[(long)ih * (ih + 1) / 2 + il];
            if (a != b && i != j) la = row[opp + (long)ih * (ih - 1) / 2 + il];
        }
        const double sgn = i > j ? 1.0 : -1.0;
        out[e] = V[ab + e] + ls + sgn * la + S[i * p + j];
        out[o2 + e] = (a != b) ? V[ba + e] + ls - sgn * la + S[j * p + i] : 0.0;
    }
}

// out[r][i][j] = Q[r][P(i,j)] + sgn(i-j) Q[r][opp + Q(i,j)]   (rows are plain, not pair-packed)
struct RowsUnpackK {
    const double* Q; double* out;
    long total;
    int no;
};
__device__ __forceinline__ void rows_unpack_body(const RowsUnpackK& k, const unsigned vb, const unsigned vgrid) {
    const double* __restrict__ Q = k.Q;
    double* __restrict__ out = k.out;
    const int no = k.no;
    const long total = k.total;
    const long opp = (long)no * (no + 1) / 2, ld = (long)no * no;
    for (long idx = vb * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)vgrid * blockDim.x) {
        const long r = idx / ld;
        const int e = (int)(idx - r * ld);
        const int i = e / no, j = e - i * no;
        const int ih = max(i, j), il = min(i, j);
        const double* __restrict__ row = Q + r * ld;
        double v = row[(long)ih * (ih + 1) / 2 + il];
        if (i != j) {
            const double x = row[opp + (long)ih * (ih - 1) / 2 + il];
            v += (i > j) ? x : -x;
        }
        out[idx] = v;
    }
}
__global__ void rows_unpack_kernel(const RowsUnpackK k) { rows_unpack_body(k, blockIdx.x, gridDim.x); }

// dev::fock_finish (device_api.h): two launches instead of twenty for matrices of n^2 <= 62500 elements
struct FockW {
    const double *G1, *G2, *J1, *J2, *L1, *L2, *K1, *K2;
};
__device__ __forceinline__ FockW fock_w(const double* W, int no, int nv) {
    const long vv = (long)nv * nv, ov = (long)no * nv, oo = (long)no * no;
    FockW w;
    w.G1 = W; w.G2 = w.G1 + vv; w.J1 = w.G2 + vv; w.J2 = w.J1 + ov; w.L1 = w.J2 + ov; w.L2 = w.L1 + oo;
    w.K1 = w.L2 + oo; w.K2 = w.K1 + ov;
    return w;
}
// The direct / exchange pairs of T1.V sums of the dressed Fock matrix (ccsd.py:226-288; G1, G2 and J1, J2 of
// dress_fock_partial) over a block V[j,a,:,:] with the last two indices virtual (o v^3: a virtual; o^2 v^2: a occupied):
//   G1[a][c] = sum_{j,b} t[b,j] V[j,a,b,c],     G2[a][c] = sum_{j,b} t[b,j] V[j,a,c,b]
// in ONE pass over the block in its own layout: for a tile X = V[j,a,:,:] the first is the t_j-weighted sum of its rows, the
// second the dot of every row with t_j.  One block per (a, chunk of j): a wave takes every fourth row, its lanes two
// columns each (16-byte loads), the row dots go through a wave reduction; partial results per chunk land in ws
// [chunk][2][v][v] and are summed by fock_g12_finish_kernel in a fixed order (bit-reproducible).  Replaces two
// matrix-vector passes over two transposed static copies of the block (2 x 3.2 GB at (50,200)).
struct FockG12K {
    const double* V; const double* t1; double* ws;
    int no, nv, na, j0, j1, jper;
};
template <int VEC>
__device__ __forceinline__ void fock_g12_body(const FockG12K& k, const unsigned vb) {
    const double* __restrict__ V = k.V;
    const double* __restrict__ t1 = k.t1;
    double* __restrict__ ws = k.ws;
    const int no = k.no, nv = k.nv, na = k.na, j0 = k.j0, j1 = k.j1, jper = k.jper;
    extern __shared__ double sm[];
    double* tj = sm;                 // [nv]     t[:, j]
    double* y2 = sm + nv;            // [nv]     row dots, summed over the chunk's j
    double* y1s = sm + 2 * nv;       // [4][nv]  per-wave column sums
    const int a = vb % na, chunk = vb / na;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int jb = j0 + chunk * jper, je = min(j1, jb + jper);
    constexpr int QMAX = 8;          // columns per lane: VEC * QMAX * 64 >= nv (host checks)
    double acc[QMAX][VEC];
#pragma unroll
    for (int q = 0; q < QMAX; ++q)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[q][e] = 0.0;
    for (int i = threadIdx.x; i < nv; i += 256) y2[i] = 0.0;
    for (int j = jb; j < je; ++j) {
        __syncthreads();             // tj of the previous j is no longer read; y2 zeroed
        for (int i = threadIdx.x; i < nv; i += 256) tj[i] = t1[(long)i * no + j];
        __syncthreads();
        const double* __restrict__ X = V + ((long)j * na + a) * nv * nv;
        for (int b = w; b < nv; b += 8) {          // two rows of the wave per pass: twice the loads in flight
            const int b2 = b + 4;
            const bool two = b2 < nv;
            const double* __restrict__ row = X + (long)b * nv;
            const double* __restrict__ row2 = X + (long)(two ? b2 : b) * nv;
            const double tb = tj[b], tb2 = two ? tj[b2] : 0.0;
            double d = 0.0, d2 = 0.0;
#pragma unroll
            for (int q = 0; q < QMAX; ++q) {
                const int c = (q * 64 + lane) * VEC;
                if (c < nv) {
                    if constexpr (VEC == 2) {
                        const v2d x = *reinterpret_cast<const v2d*>(row + c), z = *reinterpret_cast<const v2d*>(row2 + c);
                        const double t0 = tj[c], t1v = tj[c + 1];
                        acc[q][0] += tb * x[0] + tb2 * z[0];
                        acc[q][1] += tb * x[1] + tb2 * z[1];
                        d += x[0] * t0 + x[1] * t1v;
                        d2 += z[0] * t0 + z[1] * t1v;
                    } else {
                        const double x = row[c], z = row2[c], t0 = tj[c];
                        acc[q][0] += tb * x + tb2 * z;
                        d += x * t0;
                        d2 += z * t0;
                    }
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { d += __shfl_down(d, off, 64); d2 += __shfl_down(d2, off, 64); }
            if (lane == 0) { y2[b] += d; if (two) y2[b2] += d2; }      // rows b, b + 4 belong to this wave alone
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < QMAX; ++q) {
        const int c = (q * 64 + lane) * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            if (c + e < nv) y1s[w * nv + c + e] = acc[q][e];
    }
    __syncthreads();
    double* __restrict__ o1 = ws + ((long)chunk * 2 * na + a) * nv;
    double* __restrict__ o2 = o1 + (long)na * nv;
    for (int i = threadIdx.x; i < nv; i += 256) {
        o1[i] = (y1s[i] + y1s[nv + i]) + (y1s[2 * nv + i] + y1s[3 * nv + i]);
        o2[i] = y2[i];
    }
}
template <int VEC>
__global__ void __launch_bounds__(256) fock_g12_kernel(const FockG12K k) { fock_g12_body<VEC>(k, blockIdx.x); }

struct FockG12FinK {
    const double* ws; double* G1; double* G2;
    long vv;
    int nchunk;
};
__device__ __forceinline__ void fock_g12_finish_body(const FockG12FinK& k, const unsigned vb) {
    const double* __restrict__ ws = k.ws;
    const long vv = k.vv;
    const long i = vb * (long)blockDim.x + threadIdx.x;
    if (i >= vv) return;
    double s1 = 0.0, s2 = 0.0;
    for (int c = 0; c < k.nchunk; ++c) {
        s1 += ws[(long)c * 2 * vv + i];
        s2 += ws[(long)c * 2 * vv + vv + i];
    }
    k.G1[i] = s1;
    k.G2[i] = s2;
}
__global__ void fock_g12_finish_kernel(const FockG12FinK k) { fock_g12_finish_body(k, blockIdx.x); }

struct FockFinK {
    const double* f; const double* t1; const double* W; double* ft; double* fd;
    int no, nv;
};
// One WAVE per output element, the lanes over the summation index (one thread per element ran its 20-80 dependent steps as a
// latency chain: 10 + 17 us for the two kernels at (20,80) on 2 and 40 blocks): grids of ceil(no^2 / 4) and ceil(n^2 / 4) blocks.
__device__ __forceinline__ void fock_ft_body(const FockFinK& k, const unsigned vb) {
    const double* __restrict__ f = k.f;
    const double* __restrict__ t1 = k.t1;
    const double* __restrict__ W = k.W;
    double* __restrict__ ft = k.ft;
    const int no = k.no, nv = k.nv;
    const int lane = threadIdx.x & 63;
    const int e = vb * 4 + (threadIdx.x >> 6);
    if (e >= no * no) return;
    const int j = e / no, i = e - j * no, n = no + nv;
    const FockW w = fock_w(W, no, nv);
    double acc = 0.0;
    for (int b = lane; b < nv; b += 64)
        acc += (f[(long)j * n + no + b] + 2.0 * w.J1[(long)j * nv + b] - w.J2[(long)j * nv + b]) * t1[(long)b * no + i];
    acc = wave_sum(acc);
    if (lane == 0) ft[e] = acc + 2.0 * w.L1[e] - w.L2[e];
}
__global__ void fock_ft_kernel(const FockFinK k) { fock_ft_body(k, blockIdx.x); }
__device__ __forceinline__ void fock_finish_body(const FockFinK& k, const unsigned vb) {
    const double* __restrict__ f = k.f;
    const double* __restrict__ t1 = k.t1;
    const double* __restrict__ W = k.W;
    const double* __restrict__ ft = k.ft;
    double* __restrict__ fd = k.fd;
    const int no = k.no, nv = k.nv;
    const int n = no + nv;
    const int lane = threadIdx.x & 63;
    const long e = (long)vb * 4 + (threadIdx.x >> 6);
    if (e >= (long)n * n) return;
    const int p = (int)(e / n), q = (int)(e - (long)p * n);
    const FockW w = fock_w(W, no, nv);
    double v = 0.0, acc = 0.0;          // v: the terms without a sum (lane 0), acc: this lane's share of the sums
    if (p < no && q < no) {
        v = ft[p * no + q];
    } else if (p < no) {
        const int a = q - no;
        v = 2.0 * w.K1[(long)p * nv + a] - w.J2[(long)p * nv + a];
    } else if (q >= no) {
        const int a = p - no, b = q - no;
        v = 2.0 * w.G1[(long)a * nv + b] - w.G2[(long)a * nv + b];
        for (int i = lane; i < no; i += 64)
            acc -= t1[(long)a * no + i] * (f[(long)i * n + no + b] + 2.0 * w.J1[(long)i * nv + b] - w.J2[(long)i * nv + b]);
    } else {
        const int a = p - no, i = q;
        v = 2.0 * w.K1[(long)i * nv + a] - w.K2[(long)a * no + i];
        for (int j = lane; j < no; j += 64) acc -= t1[(long)a * no + j] * (f[(long)j * n + i] + ft[j * no + i]);
        for (int b = lane; b < nv; b += 64)
            acc += (f[(long)(no + a) * n + no + b] + 2.0 * w.G1[(long)a * nv + b] - w.G2[(long)a * nv + b]) * t1[(long)b * no + i];
    }
    acc = wave_sum(acc);
    if (lane == 0) fd[e] = f[e] + (v + acc);
}
__global__ void fock_finish_kernel(const FockFinK k) { fock_finish_body(k, blockIdx.x); }

// partial traces of a pair matrix M[(c,k)][(b,j)] (device_api.h).  Blocks [0, nv * ceil(nv/16)): one c and sixteen a
// each — a wave takes four a (their loads in flight together), its lanes the k of sum_k M[(c,k)][(a,k)] (one element per
// 128-byte line: 1/3 of the lines of M in all), summed by a shuffle tree.  The no blocks behind them: one k each — lanes
// over i, the four waves over c = w, w+4, ..., added up in wave order through LDS.  Fixed summation orders: deterministic.
// A second matrix M2 (same shape and pitch) adds alpha2 x its traces in the same pass (the two builds of the ring terms).
struct TracesK {
    const double* M; const double* M2; double* out_vv; double* out_oo;
    long ld;
    double alpha, beta, alpha2;
    int no, nv;
};
constexpr int kTracesLdsDoubles = 4 * 64;
__device__ __forceinline__ void pair_traces_body(const TracesK& q, const unsigned vb, double* __restrict__ smem) {
    const double* __restrict__ M = q.M;
    const double* __restrict__ M2 = q.M2;
    double* __restrict__ out_vv = q.out_vv;
    double* __restrict__ out_oo = q.out_oo;
    const long ld = q.ld;
    const double alpha = q.alpha, beta = q.beta, alpha2 = q.alpha2;
    const int no = q.no, nv = q.nv;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int agroups = (nv + 15) / 16, nvv = nv * agroups;
    if ((int)vb < nvv) {
        const int c = vb / agroups, a0 = (vb - c * agroups) * 16 + wave;
        const double* __restrict__ base = M + (long)c * no * ld;
        const double* __restrict__ base2 = M2 ? M2 + (long)c * no * ld : nullptr;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int a = a0 + 4 * u;
            if (a < nv)
                for (int k = lane; k < no; k += 64) {
                    const long e = (long)k * ld + (long)a * no + k;
                    double v = alpha * base[e];
                    if (base2) v += alpha2 * base2[e];
                    acc[u] += v;
                }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[u] += __shfl_down(acc[u], off, 64);
            const int a = a0 + 4 * u;
            if (lane == 0 && a < nv) {
                double* o = out_vv + (long)a * nv + c;
                *o = (beta == 0.0 ? 0.0 : beta * *o) + acc[u];
            }
        }
        return;
    }
    double (*part)[64] = reinterpret_cast<double (*)[64]>(smem);
    const int k = vb - nvv;
    for (int i0 = 0; i0 < no; i0 += 64) {
        const int i = i0 + lane;
        double acc = 0.0;
        if (i < no) {
#pragma unroll 4
            for (int c = wave; c < nv; c += 4) {
                const long e = ((long)c * no + k) * ld + (long)c * no + i;
                double v = alpha * M[e];
                if (M2) v += alpha2 * M2[e];
                acc += v;
            }
        }
        part[wave][lane] = acc;
        __syncthreads();
        if (wave == 0 && i < no) {
            const double sum = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
            double* o = out_oo + (long)k * no + i;
            *o = (beta == 0.0 ? 0.0 : beta * *o) + sum;
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) pair_traces_kernel(const TracesK q) {
    __shared__ double part[kTracesLdsDoubles];
    pair_traces_body(q, blockIdx.x, part);
}

// ------------------------------------------------------------------------------------
// explicit 3-body operator: TCDUMP scatter (tcdump.py:52-56) and its mean-field foldings (contraction.py:17-95)
// L is dense [nb]^6 in chemists' order (or|ps|qt)
// ------------------------------------------------------------------------------------
__global__ void scatter_kernel(double* __restrict__ dst, const long* __restrict__ idx, const double* __restrict__ val,
                               long n) {
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x)
        dst[idx[t]] = val[t];                        // targets are unique (the host keeps the last of duplicates)
}

__device__ __forceinline__ double L6(const double* __restrict__ L, int nb, int a, int b, int c, int d, int e, int f) {
    return L[((((long)a * nb + b) * nb + c) * nb + d) * nb * nb + (long)e * nb + f];
}

// D[p,r,q,s] = -1/3 { -3 sum_i (L[p,q,r,i,i,s] + L[r,s,p,i,i,q]) + 6 sum_i L[p,q,r,s,i,i] }     (contraction.py:17-39)
__global__ void tc_single_kernel(const double* __restrict__ L, double* __restrict__ D, int nb, int no, long total) {
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        long x = t;
        const int s = (int)(x % nb); x /= nb;
        const int q = (int)(x % nb); x /= nb;
        const int r = (int)(x % nb);
        const int p = (int)(x / nb);
        double acc = 0.0;
        for (int i = 0; i < no; ++i)
            acc += -3.0 * (L6(L, nb, p, q, r, i, i, s) + L6(L, nb, r, s, p, i, i, q)) + 6.0 * L6(L, nb, p, q, r, s, i, i);
        D[t] = -acc / 3.0;
    }
}

// S[p,q] = -1/6 sum_ij { 12 L[i,i,j,j,p,q] - 12 L[i,i,p,j,j,q] + 6 L[p,i,j,q,i,j] - 6 L[i,j,j,i,p,q] }   (contraction.py:41-65)
__global__ void tc_double_kernel(const double* __restrict__ L, double* __restrict__ S, int nb, int no) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nb * nb) return;
    const int p = t / nb, q = t - p * nb;
    double acc = 0.0;
    for (int i = 0; i < no; ++i)
        for (int j = 0; j < no; ++j)
            acc += 12.0 * L6(L, nb, i, i, j, j, p, q) - 12.0 * L6(L, nb, i, i, p, j, j, q) +
                   6.0 * L6(L, nb, p, i, j, q, i, j) - 6.0 * L6(L, nb, i, j, j, i, p, q);
    S[t] = -acc / 6.0;
}

// T0 = -1/6 sum_ijk { 8 L[i,i,j,j,k,k] - 12 L[i,j,j,i,k,k] + 4 L[i,j,j,k,k,i] }       (contraction.py:67-95)
__global__ void __launch_bounds__(256) tc_triple_kernel(const double* __restrict__ L, double* __restrict__ out, int nb, int no) {
    __shared__ double sh[256];
    double acc = 0.0;
    const long n3 = (long)no * no * no;
    for (long t = threadIdx.x; t < n3; t += blockDim.x) {
        const int k = (int)(t % no), j = (int)((t / no) % no), i = (int)(t / ((long)no * no));
        acc += 8.0 * L6(L, nb, i, i, j, j, k, k) - 12.0 * L6(L, nb, i, j, j, i, k, k) + 4.0 * L6(L, nb, i, j, j, k, k, i);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = -sh[0] / 6.0;
}

// Hartree-Fock matrix from the packed blocks (hf.py:14-18): f[p,q] = h[p,q] + sum_i (2 V[p,i,q,i] - V[p,i,i,q]), i occupied.
// dir[tp*2+tq] = block (tp, occ, tq, occ), exc[tp*2+tq] = block (tp, occ, occ, tq); tp/tq = 1 for a virtual index.
struct HfBlocks { const double* dir[4]; const double* exc[4]; };
__global__ void hf_fock_kernel(const HfBlocks B, const double* __restrict__ h, double* __restrict__ f, int no, int nv) {
    const int n = no + nv;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * n) return;
    const int p = t / n, q = t - p * n;
    const int tp = p >= no, tq = q >= no;
    const long pl = tp ? p - no : p, ql = tq ? q - no : q, nq = tq ? nv : no;
    const double* __restrict__ D = B.dir[tp * 2 + tq];
    const double* __restrict__ X = B.exc[tp * 2 + tq];
    double acc = 0.0;
    for (long i = 0; i < no; ++i)
        acc += 2.0 * D[((pl * no + i) * nq + ql) * no + i] - X[((pl * no + i) * no + i) * nq + ql];
    f[t] = h[t] + acc;
}

// FCIDUMP lines -> dense V[n]^4 (fcidump.py:140-149): one thread per line writes the symmetry images in the
// reference's order.  A second kernel counts lines whose images do not all hold the line's value afterwards, i.e.
// files whose symmetry-related entries disagree (only there does the order of the lines matter).
__global__ void fcidump_fill_kernel(double* __restrict__ V, const double* __restrict__ val, const int* __restrict__ pqrs,
                                    long count, long n, int is_tc, int verify, unsigned long long* __restrict__ bad) {
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < count; t += (long)gridDim.x * blockDim.x) {
        const long p = pqrs[4 * t], q = pqrs[4 * t + 1], r = pqrs[4 * t + 2], s = pqrs[4 * t + 3];
        const double x = val[t];
        long tg[4];
        int m;
        if (is_tc) { tg[0] = ((q * n + p) * n + s) * n + r; tg[1] = ((p * n + q) * n + r) * n + s; m = 2; }
        else {
            tg[0] = ((p * n + q) * n + r) * n + s; tg[1] = ((r * n + q) * n + p) * n + s;
            tg[2] = ((r * n + s) * n + p) * n + q; tg[3] = ((p * n + s) * n + r) * n + q; m = 4;
        }
        if (!verify) {
            for (int i = 0; i < m; ++i) V[tg[i]] = x;
        } else {
            bool ok = true;
            for (int i = 0; i < m; ++i) ok = ok && (V[tg[i]] == x);
            if (!ok) atomicAdd(bad, 1ULL);
        }
    }
}

// ------------------------------------------------------------------------------------
// uniform electron gas integrals (ueg.py:265-596)
// ------------------------------------------------------------------------------------
struct UegK {
    int n_p, n_occ, imax, m, mode, n_ele, lat;
    double L, Omega, kc2g, gamma;
    const double* tab_s;      // correlator tables over m = |n|^2 (device_api.h UegParams); null: evaluated in place
    const double* tab_a;
    int tab_len;
    int kind;                 // 0 trunc, 1 gaskell, 2 gaskell_modified, 3 coulomb, 4 yukawa, 5 stg, 6 smooth
    double p0, p1, p2;
};
// u(k^2) for k = 2 pi n / L: x is the float k^2 the reference would pass, m = |n|^2 its integer shell;
// ARR = the reference calls the correlator with an ndarray there (else with a float)
template <bool ARR>
__device__ __forceinline__ double ueg_u(double x, long m, const UegK& u) {
    if (u.tab_a) return m < u.tab_len ? (ARR ? u.tab_a[m] : u.tab_s[m]) : 0.0;
    switch (u.kind) {
        case 1:                                                         // gaskell, ueg.py:836-883 (p0 = mu, p1 = cut)
            if (ARR) return x > u.p1 ? -0.0 : (x > 1e-12 ? -(u.p0 / x) : -0.0);
            return (x < u.p1 && x > 1e-12) ? -(u.p0 / x) : -0.0;
        case 2:                                                         // gaskell_modified, ueg.py:802-834 (p0 = cut)
            if (ARR) return x >= u.p0 ? -((4.0 * M_PI) / (x * x)) : -0.0;
            return (x < u.p0 && x > 1e-12) ? -0.0 : -((4.0 * M_PI) / (x * x));
        case 3: return x > 1e-12 ? u.p0 / x : 0.0;                      // coulomb, ueg.py:905-915 (p0 = -4 pi gamma)
        case 4: { const double b = x + u.p0; return fabs(b) > u.p1 ? (-4.0 * M_PI) / b : 0.0; }      // yukawa, :740-770
        case 5: { const double t = x + u.p0, b = t * t; return fabs(b) > u.p1 ? u.p2 / b : 0.0; }    // stg, :917-935
        case 6: {                                                       // smooth, ueg.py:885-903
            if (!(x > u.p2)) return 0.0;
            return (-4.0 * M_PI * (1.0 + erf((sqrt(x) - u.p0) / u.p1)) / 2.0) / (x * x);
        }
        default: break;
    }
    if (x <= u.kc2g) x = 0.0;                                           // trunc, ueg.py:772-800
    return x > 1e-12 ? (-4.0 * M_PI / (x * x)) * u.gamma : 0.0;
}
__device__ __forceinline__ double ueg_kp(int k, double L) { return ((double)(k * 2) * M_PI) / L; }   // planewave.py:15

// u_mat[d] = sum_k' (k'.(k-k')) u(k'^2) u((k-k')^2) / Omega, one block per momentum transfer d  (ueg.py:581-596)
__global__ void __launch_bounds__(256) ueg_nabla_kernel(const UegK u, const double* __restrict__ dk, const int* __restrict__ dint,
                                                        double* __restrict__ out) {
    __shared__ double sh[4];
    const double kx = dk[3 * blockIdx.x], ky = dk[3 * blockIdx.x + 1], kz = dk[3 * blockIdx.x + 2];
    const long dx = dint[3 * blockIdx.x], dy = dint[3 * blockIdx.x + 1], dz = dint[3 * blockIdx.x + 2];
    const int w = 2 * u.lat + 1;
    const long total = (long)w * w * w;
    double s = 0.0;
    for (long idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const int c = (int)(idx % w), b = (int)((idx / w) % w), a = (int)(idx / ((long)w * w));
        const double x1 = 2.0 * M_PI * (a - u.lat) / u.L, y1 = 2.0 * M_PI * (b - u.lat) / u.L,
                     z1 = 2.0 * M_PI * (c - u.lat) / u.L;
        const double x2 = kx - x1, y2 = ky - y1, z2 = kz - z1;
        const long a1 = a - u.lat, b1 = b - u.lat, c1 = c - u.lat, a2 = dx - a1, b2 = dy - b1, c2 = dz - c1;
        s += (x1 * x2 + y1 * y2 + z1 * z2) * ueg_u<true>(x1 * x1 + y1 * y1 + z1 * z1, a1 * a1 + b1 * b1 + c1 * c1, u) *
             ueg_u<true>(x2 * x2 + y2 * y2 + z2 * z2, a2 * a2 + b2 * b2 + c2 * c2, u);
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) out[blockIdx.x] = s / u.Omega;
}

// per (p,r): the q-independent singly-contracted 3-body value (ueg.py:461-474, 518-573)
__global__ void ueg_effective_kernel(const UegK u, const int* __restrict__ kint, double* __restrict__ E) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= u.n_p * u.n_p) return;
    const int p = idx / u.n_p, r = idx - p * u.n_p;
    double kp[3], kr[3], dk[3];
    long di[3], md = 0;
    for (int c = 0; c < 3; ++c) {
        kp[c] = ueg_kp(kint[3 * p + c], u.L);
        kr[c] = ueg_kp(kint[3 * r + c], u.L);
        dk[c] = kr[c] - kp[c];
        di[c] = kint[3 * r + c] - kint[3 * p + c];
        md += di[c] * di[c];
    }
    const double dk2 = dk[0] * dk[0] + dk[1] * dk[1] + dk[2] * dk[2];
    const double udk_a = ueg_u<true>(dk2, md, u);      // inside contract_exchange_3_body: a 0-d array (ueg.py:536)
    const double udk_s = ueg_u<false>(dk2, md, u);     // in the main loop: a float (ueg.py:409, :461)
    double xr = 0.0, xp = 0.0, pk = 0.0;
    for (int n = 0; n < u.n_occ; ++n) {
        double o[3];
        for (int c = 0; c < 3; ++c) o[c] = ueg_kp(kint[3 * n + c], u.L);
        double a2 = 0, ad = 0, b2 = 0, bd = 0, v12 = 0, v11 = 0;
        long ma = 0, mb = 0, mv = 0;
        for (int c = 0; c < 3; ++c) {
            const double a = kr[c] - o[c], b = kp[c] - o[c], v1 = kr[c] - dk[c] - o[c];
            a2 += a * a; ad += a * dk[c];
            b2 += b * b; bd += b * dk[c];
            v12 += v1 * a; v11 += v1 * v1;
            const long ai = kint[3 * r + c] - kint[3 * n + c], bi = kint[3 * p + c] - kint[3 * n + c], vi = ai - di[c];
            ma += ai * ai; mb += bi * bi; mv += vi * vi;
        }
        xr += ad * udk_a * ueg_u<true>(a2, ma, u);
        xp += bd * udk_a * ueg_u<true>(b2, mb, u);
        pk += v12 * ueg_u<true>(v11, mv, u) * ueg_u<true>(a2, ma, u);
    }
    xr /= u.Omega; xp /= u.Omega; pk /= u.Omega;
    double val;
    if (fabs(dk2) > 0.0) val = -(double)u.n_ele * dk2 * udk_s * udk_s / u.Omega + 2.0 * xr - 2.0 * xp + 2.0 * pk;
    else val = 2.0 * pk;
    E[idx] = val / u.Omega;
}

// one thread per (p,q,r): s by momentum conservation through the flattened lookup (ueg.py:384-507)
__global__ void ueg_scatter_kernel(const UegK u, const int* __restrict__ kint, const int* __restrict__ map,
                                   const double* __restrict__ umat, const int* __restrict__ umat_index,
                                   const double* __restrict__ E, double* __restrict__ V) {
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long n = u.n_p;
    if (idx >= n * n * n) return;
    const int r = (int)(idx % n), q = (int)((idx / n) % n), p = (int)(idx / (n * n));
    int d[3], ks[3];
    for (int c = 0; c < 3; ++c) {
        d[c] = kint[3 * r + c] - kint[3 * p + c];
        ks[c] = kint[3 * q + c] - d[c];
    }
    const long loc = (long)u.m * u.m * (ks[0] + u.imax) + (long)u.m * (ks[1] + u.imax) + ks[2] + u.imax;
    if (loc < 0 || loc >= (long)u.m * u.m * u.m) return;     // only the flattened index is range-checked (:397)
    const int s = map[loc];
    if (s < 0 || s >= u.n_p) return;
    double dk[3], dk2 = 0.0;
    long md = 0;
    for (int c = 0; c < 3; ++c) {
        dk[c] = ueg_kp(kint[3 * r + c], u.L) - ueg_kp(kint[3 * p + c], u.L);
        dk2 += dk[c] * dk[c];
        md += (long)d[c] * d[c];
    }
    double w = 0.0;
    if (u.mode == 0) {
        if (fabs(dk2) > 0.0) w = 4.0 * M_PI / dk2 / u.Omega;
    } else if (u.mode == 3) {
        if (fabs(dk2) > 0.0) { const double x = ueg_u<false>(dk2, md, u); w = -(double)u.n_ele * dk2 * x * x / u.Omega / u.Omega; }
    } else if (u.mode == 1) {
        const int w4 = 4 * u.imax + 1;
        const double um = umat[umat_index[((long)(d[0] + 2 * u.imax) * w4 + (d[1] + 2 * u.imax)) * w4 + d[2] + 2 * u.imax]];
        if (fabs(dk2) > 0.0) {
            double rsdk = 0.0;
            for (int c = 0; c < 3; ++c) rsdk += (ueg_kp(kint[3 * r + c], u.L) - ueg_kp(kint[3 * s + c], u.L)) * dk[c];
            const double x = ueg_u<false>(dk2, md, u);
            w = (4.0 * M_PI / dk2 + um + dk2 * x - rsdk * x) / u.Omega;
        } else {
            w = um / u.Omega;
        }
    } else {
        w = E[(long)p * n + r];
    }
    V[((long)(p * n + q) * n + r) * n + s] = w;
}

// =====================================================================================================================
// Phase launches (round 6): the small kernels between two big products form a dependency graph that is mostly
// WIDE, not deep — the dressed Fock matrix, the T1 dressing of three V blocks, the pair layouts and the ladder operands of
// a (20,80) iteration are seventeen launches of 5-35 us, each on a fraction of the chip, none reading what another writes.
// A dependent launch costs 1.9 us on this chip and a dependency counter inside a persistent kernel more than that at
// every task size (profiles/r06/probe_phase_boundary_vs_counters.txt), so the seam between dependent tasks stays a kernel
// boundary and what is removed is the serialisation of INDEPENDENT ones: while a phase is open the device functions below
// do not launch, they append a task (kind, argument block, blocks, LDS, the address ranges it reads and writes); a flush
// assigns every task the earliest level its hazards allow (read-after-write, write-after-read and write-after-write on
// overlapping address ranges order two tasks, nothing else does — also across reused arena temporaries) and launches ONE
// grid per level that carries the blocks of all its tasks (phase_kernel).  A task's blocks are dealt round-robin over
// the eight XCDs (its range of hardware blocks starts at a multiple of 8), tasks in order of decreasing cost.
// Anything that is not recorded (a big product, a copy, a synchronisation, a graph boundary) flushes first: the order of
// effects is that of immediate execution.
// =====================================================================================================================
enum : unsigned short {
    PK_GEMM = 0, PK_SPLITK, PK_PERM_DIRECT, PK_PERM_TILED, PK_GEMV_COLS, PK_GEMV_FINISH, PK_GEMV_ROWS, PK_CC_UPDATE, PK_LINCOMB,
    PK_DOTS1, PK_DOTS2, PK_ENERGY, PK_TAU, PK_PACK_T, PK_LADDER_UNPACK, PK_RING_OPERANDS, PK_T2_LAYOUTS, PK_ASSEMBLE,
    PK_ROWS_UNPACK, PK_FOCK_G12, PK_FOCK_G12_FIN, PK_FOCK_FT, PK_FOCK_FIN, PK_TRACES, PK_DOTS_FINAL, PK_KINDS
};
struct SplitkTaskK { GemmK g; int BM, BN; };
struct PermTaskK { PermK p; long total; int tiles_q, tiles_l; };
struct GemvTaskK { GemvItem it; double* ws; const double* yin; double beta; };

constexpr int kPhaseMaxTasks = 24;
constexpr int kPhaseBlobWords = 440;
struct PhaseK {
    int n;
    int blk_end[kPhaseMaxTasks];       // running block count, every task's range padded to a multiple of 8
    int nblk[kPhaseMaxTasks];          // blocks the task really has
    unsigned short kind[kPhaseMaxTasks], sub[kPhaseMaxTasks], off[kPhaseMaxTasks];     // off: 8-byte words into blob
    long blob[kPhaseBlobWords];
};
static_assert(sizeof(PhaseK) <= 4096, "kernel arguments are limited to 4 KB");

// (the argument block of a task is addressed ONCE, before the switch over the kinds: a pointer into the by-value kernel
// argument formed inside every case makes the compiler merge them into a phi, lose track of the argument's constness and copy
// all 3.9 KB of it to scratch at the start of every block)
template <typename T>
__device__ __forceinline__ const T& phase_args(const long* args) { return *reinterpret_cast<const T*>(args); }
// GEMM variants of a phase: sub = shape * 8 + (A K-contiguous) * 4 + (B K-contiguous) * 2 + (16-byte loads); shape 0 = 64 x 64,
// 1 = 64 x 32, 2 = 32 x 64 (double-buffered register-staged tiles: the block count per CU is the launch's, not the variant's)
template <int BM, int BN>
__device__ __forceinline__ void phase_gemm_layouts(const GemmK& g, const int lay, const long bid) {
    switch (lay) {
        case 0: dgemm_body<BM, BN, false, false, 1>(g, bid); break;
        case 1: dgemm_body<BM, BN, false, false, 2>(g, bid); break;
        case 2: dgemm_body<BM, BN, false, true, 1>(g, bid); break;
        case 3: dgemm_body<BM, BN, false, true, 2>(g, bid); break;
        case 4: dgemm_body<BM, BN, true, false, 1>(g, bid); break;
        case 5: dgemm_body<BM, BN, true, false, 2>(g, bid); break;
        case 6: dgemm_body<BM, BN, true, true, 1>(g, bid); break;
        default: dgemm_body<BM, BN, true, true, 2>(g, bid); break;
    }
}
// HEAVY: with the register-hungry tasks — matrix-core products, the dressed-Fock kernels, the energy reduction (120
// registers: four blocks per CU); the launches of a level without any of them — layouts, packs, permutations, updates, dot
// products: bound by the loads they keep in flight — use the light variant (at most 64 registers: eight blocks per CU)
template <bool HEAVY>
__device__ __forceinline__ void phase_body(const PhaseK& ph) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double sh[4];
    const int b = blockIdx.x;
    int it = 0;
    while (it + 1 < ph.n && b >= ph.blk_end[it]) ++it;
    const int start = it ? ph.blk_end[it - 1] : 0;
    const int span = ph.blk_end[it] - start;
    const int nblk = ph.nblk[it];
    const int kind = ph.kind[it], sub = ph.sub[it];
    const long* const args = ph.blob + ph.off[it];      // (ONE address into the kernel arguments, formed here: see phase_args)
    int vb = b - start;                          // start % 8 == 0: vb % 8 is the block's XCD
    if (kind == PK_GEMM) vb = (int)xcd_remap(vb, span);     // neighbouring tiles of a product on one XCD
    if (vb >= nblk) return;
    switch (kind) {
        case PK_GEMM:
            if constexpr (HEAVY) {
                const GemmK& g = phase_args<GemmK>(args);
                const int shape = sub >> 3, lay = sub & 7;
                if (shape == 0) phase_gemm_layouts<64, 64>(g, lay, vb);
                else if (shape == 1) phase_gemm_layouts<64, 32>(g, lay, vb);
                else phase_gemm_layouts<32, 64>(g, lay, vb);
            }
            break;
        case PK_SPLITK: {
            const SplitkTaskK& k = phase_args<SplitkTaskK>(args);
            const int pieces = k.BM * k.BN / 256;
            splitk_reduce_body(k.g, k.BM, k.BN, vb / pieces, vb % pieces);
        } break;
        case PK_PERM_DIRECT: {
            const PermTaskK& k = phase_args<PermTaskK>(args);
            permute_direct_body(k.p, k.total, vb, nblk);
        } break;
        case PK_PERM_TILED: {
            const PermTaskK& k = phase_args<PermTaskK>(args);
            permute_tiled_body(k.p, k.tiles_q, k.tiles_l, vb, smem);
        } break;
        case PK_GEMV_COLS: {
            const GemvTaskK& k = phase_args<GemvTaskK>(args);
            gemv_cols_item(k.it, vb, k.ws);
        } break;
        case PK_GEMV_FINISH: {
            const GemvTaskK& k = phase_args<GemvTaskK>(args);
            gemv_finish_item(k.it, vb, k.ws, k.beta, k.yin);
        } break;
        case PK_GEMV_ROWS:
            if (sub) gemv_rows_body<2>(phase_args<GemvRowsK>(args), vb);
            else gemv_rows_body<1>(phase_args<GemvRowsK>(args), vb);
            break;
        case PK_CC_UPDATE: cc_update_body(phase_args<CcUpdateK>(args), vb, nblk); break;
        case PK_LINCOMB: lincomb_body(phase_args<LinK>(args), vb, nblk); break;
        case PK_DOTS1:
            if (sub) dots_stage1_body<2>(phase_args<DotsK>(args), vb, sh);
            else dots_stage1_body<1>(phase_args<DotsK>(args), vb, sh);
            break;
        case PK_DOTS2: dots_stage2_body(phase_args<Dots2K>(args), vb, sh); break;
        case PK_DOTS_FINAL: dots_final_body(phase_args<DotsFinalK>(args), vb, sh); break;
        case PK_ENERGY:
            if constexpr (HEAVY) {
                if (sub) energy_norms_body<2>(phase_args<EnergyK>(args), vb, nblk, sh);
                else energy_norms_body<1>(phase_args<EnergyK>(args), vb, nblk, sh);
            }
            break;
        case PK_TAU: tau_body(phase_args<TauK>(args), vb, nblk); break;
        case PK_PACK_T: ladder_pack_T_body(phase_args<PackTK>(args), vb); break;
        case PK_LADDER_UNPACK: ladder_unpack_body(phase_args<UnpackK>(args), vb, nblk); break;
        case PK_RING_OPERANDS: ring_operands_body(phase_args<RingOpK>(args), vb); break;
        case PK_T2_LAYOUTS:
            if (sub) t2_layouts_body<true>(phase_args<LayoutsK>(args), vb);
            else t2_layouts_body<false>(phase_args<LayoutsK>(args), vb);
            break;
        case PK_ASSEMBLE: residual_assemble_body(phase_args<AssembleK>(args), vb); break;
        case PK_ROWS_UNPACK: rows_unpack_body(phase_args<RowsUnpackK>(args), vb, nblk); break;
        case PK_FOCK_G12:
            if constexpr (HEAVY) {
                if (sub) fock_g12_body<2>(phase_args<FockG12K>(args), vb);
                else fock_g12_body<1>(phase_args<FockG12K>(args), vb);
            }
            break;
        case PK_FOCK_G12_FIN: fock_g12_finish_body(phase_args<FockG12FinK>(args), vb); break;
        case PK_FOCK_FT:
            if constexpr (HEAVY) fock_ft_body(phase_args<FockFinK>(args), vb);
            break;
        case PK_FOCK_FIN:
            if constexpr (HEAVY) fock_finish_body(phase_args<FockFinK>(args), vb);
            break;
        case PK_TRACES: pair_traces_body(phase_args<TracesK>(args), vb, smem); break;
        default: break;
    }
}
__global__ void __launch_bounds__(kThreads, 2) phase_kernel(const PhaseK ph) { phase_body<true>(ph); }
__global__ void __launch_bounds__(kThreads, 8) phase_light_kernel(const PhaseK ph) { phase_body<false>(ph); }
__host__ __device__ constexpr bool phase_kind_heavy(int kind) {
    return kind == PK_GEMM || kind == PK_ENERGY || kind == PK_FOCK_G12 || kind == PK_FOCK_FT || kind == PK_FOCK_FIN;
}

inline int grid_for(long total, int block = 256, int cap = 256 * 16) {
    long g = (total + block - 1) / block;
    return (int)std::max<long>(1, std::min<long>(g, cap));
}

// ---- profiling state ---------------------------------------------------------------
struct Prof {
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    std::vector<std::string> what;   // one description per recorded GEMM call
    std::vector<double> fl;          // its flops
    std::vector<int> klass;          // 1: ran on the LDS-DMA 128x128 kernel
    std::vector<int> nk;             // GEMM kernel launches of the call (2 with a k-split tail)
    double flops = 0.0;
};
thread_local Prof g_prof;        // one context per host thread (include/pymes_amd.h): the events are that thread's stream's

// per device ordinal (a process may hold contexts on several GPUs): reduction workspace [16*kDotBlocks + 16] on the
// device, pinned result buffer [16] on the host, and "attribute set" flags of the kernels with > 64 KB of dynamic LDS
constexpr int kMaxDevices = 16;
double* g_dot_ws[kMaxDevices] = {nullptr};
double* g_dot_host[kMaxDevices] = {nullptr};
// Gram blocks: partial sums [kGramBlocks][64] + finished tiles [kGramTiles][64] on the device, the tiles pinned on the host
constexpr int kGramTiles = 64;
double* g_gram_ws[kMaxDevices] = {nullptr};
double* g_gram_host[kMaxDevices] = {nullptr};
std::atomic<long> g_live_allocs{0};

// ---- phase queue (host side of phase_kernel) ----------------------------------------------------------------------------
// [lo, hi) in bytes; pitch != 0: only the rows [lo + i pitch, lo + i pitch + width) of it (a pitched 2-D box: two column
// slices of the same rows — LS | LA of the packed ladder rows, the blocks of a stacked operand — do not overlap)
struct PhaseRange { uintptr_t lo, hi; uintptr_t pitch = 0, width = 0; };
inline PhaseRange prange(const void* p, long doubles) {
    const uintptr_t lo = reinterpret_cast<uintptr_t>(p);
    return PhaseRange{lo, p && doubles > 0 ? lo + 8 * (uintptr_t)doubles : lo};
}
// bounding interval of a strided box: extents n[i] with strides st[i] (in doubles, any sign), + `slack` doubles at the top
inline PhaseRange pbox(const void* base, std::initializer_list<std::pair<long, long>> dims, long slack = 0) {
    long lo = 0, hi = 0;
    for (const auto& d : dims) {
        if (d.first <= 0) return PhaseRange{0, 0};
        const long span = (d.first - 1) * d.second;
        if (span < 0) lo += span; else hi += span;
    }
    const uintptr_t b = reinterpret_cast<uintptr_t>(base);
    PhaseRange r{b + 8 * lo, b + 8 * (hi + 1 + slack)};
    // exactly two extents above one, a unit stride and a larger positive pitch: rows of `w` doubles every `p`
    long w = 0, pch = 0;
    int nd = 0;
    for (const auto& d : dims)
        if (d.first > 1) {
            ++nd;
            if (d.second == 1) w = d.first;
            else pch = d.second;
        }
    if (nd == 2 && w > 0 && pch > w + slack) { r.pitch = 8 * (uintptr_t)pch; r.width = 8 * (uintptr_t)(w + slack); }
    return r;
}
struct PhaseRec {
    unsigned short kind = 0, sub = 0;
    int nblk = 0, lds = 0, level = 0, words = 0;
    double cost = 0.0;                 // estimated microseconds (order of the tasks inside a level: longest first)
    int nr = 0, nw = 0;
    PhaseRange r[10], w[4];
    // accumulation fusion (phase_fuse_accumulations): `box` is set when the task's only output is ONE contiguous array;
    // acc = 1: it ACCUMULATES into it (out = alpha X + out: a product with beta = 1, an accumulating permutation) and can be
    // redirected to a private buffer; acc = 0: it overwrites all of it
    PhaseRange box{0, 0};
    signed char acc = -1;
    long blob[56];
};
struct PhaseQueue {
    int enabled = -1;                  // -1: PYMES_PHASE decides at first use (default on)
    bool flushing = false;
    double max_us = 60.0;              // tasks estimated above this run as launches of their own (PYMES_PHASE_MAX_US)
    bool serial = false;               // PYMES_PHASE=serial: one task per level (hazard-analysis bisection)
    hipStream_t st = nullptr;
    std::vector<PhaseRec> q;
    long ws_cursor = 0;                // rolling sub-allocation of the split-K workspace among the tasks of a phase
    double* ws = nullptr;              // that workspace (the engine's; remembered from the products of the phase)
    long ws_doubles = 0;
    bool hold = false;                 // dev::phase_hold: the end of a C-interface call does not launch what is recorded
    bool log = false;                  // PYMES_PHASE_LOG
    uintptr_t fuse_bytes = 2u << 20;   // PYMES_PHASE_FUSE_MB
    long tasks = 0, launches = 0, levels = 0, flushes = 0;      // statistics (dev::phase_stats)
    long fused = 0;                    // accumulation chains fused (phase_fuse_accumulations)
};
thread_local PhaseQueue g_phase;
void phase_flush();
void gemm_group_flush();
bool gemm_group_pending();

inline bool phase_open(hipStream_t st) {
    PhaseQueue& P = g_phase;
    if (P.enabled < 0) {
        const char* e = getenv("PYMES_PHASE");
        P.enabled = (e && e[0] == '0') ? 0 : 1;
        P.serial = e && !strcmp(e, "serial");
        P.max_us = getenv("PYMES_PHASE_MAX_US") ? atof(getenv("PYMES_PHASE_MAX_US")) : 60.0;
        P.log = getenv("PYMES_PHASE_LOG") != nullptr;
        P.fuse_bytes = (uintptr_t)((getenv("PYMES_PHASE_FUSE_MB") ? atof(getenv("PYMES_PHASE_FUSE_MB")) : 2.0) * 1048576.0);
    }
    if (!P.enabled || P.flushing || g_prof.on) return false;
    if (!P.q.empty() && P.st != st) phase_flush();
    P.st = st;
    return true;
}
inline bool phase_small(double cost_us) { return cost_us <= g_phase.max_us; }

template <typename T>
PhaseRec& phase_push(unsigned short kind, unsigned short sub, long nblk, int lds, double cost, const T& args) {
    static_assert(sizeof(T) % 8 == 0 && sizeof(T) <= sizeof(PhaseRec::blob), "task arguments: 8-byte words, at most 448 bytes");
    static_assert(std::is_trivially_copyable<T>::value, "task arguments are copied as bytes");
    PhaseQueue& P = g_phase;
    // a product waiting in an open group (dev::gemm_group_*) keeps its place in the order of effects: it goes first
    if (gemm_group_pending()) gemm_group_flush();
    if (P.q.size() >= 160) phase_flush();
    if (nblk <= 0 || nblk > 0x3fffffffL) throw std::runtime_error("phase: bad block count");
    P.q.emplace_back();
    PhaseRec& t = P.q.back();
    t.kind = kind; t.sub = sub; t.nblk = (int)nblk; t.lds = lds; t.cost = cost;
    t.words = (int)(sizeof(T) / 8);
    memcpy(t.blob, &args, sizeof(T));
    ++P.tasks;
    return t;
}
inline void phase_reads(PhaseRec& t, std::initializer_list<PhaseRange> rs) {
    for (const auto& x : rs)
        if (x.hi > x.lo) {
            if (t.nr >= 10) throw std::runtime_error("phase: too many read ranges");
            t.r[t.nr++] = x;
        }
}
inline void phase_writes(PhaseRec& t, std::initializer_list<PhaseRange> ws) {
    for (const auto& x : ws)
        if (x.hi > x.lo) {
            if (t.nw >= 4) throw std::runtime_error("phase: too many write ranges");
            t.w[t.nw++] = x;
        }
}
inline bool phase_overlap(const PhaseRange& a, const PhaseRange& b) {
    if (!(a.lo < b.hi && b.lo < a.hi)) return false;
    if (a.pitch != 0 && a.pitch == b.pitch) {       // column slices of rows with one pitch
        const uintptr_t p = a.pitch, d = b.lo >= a.lo ? (b.lo - a.lo) % p : (p - (a.lo - b.lo) % p) % p;
        if (d >= a.width && d + b.width <= p) return false;
    }
    return true;
}
inline bool phase_conflict(const PhaseRec& a, const PhaseRec& b) {       // a earlier, b later
    for (int i = 0; i < a.nw; ++i) {
        for (int j = 0; j < b.nw; ++j) if (phase_overlap(a.w[i], b.w[j])) return true;
        for (int j = 0; j < b.nr; ++j) if (phase_overlap(a.w[i], b.r[j])) return true;
    }
    for (int i = 0; i < a.nr; ++i)
        for (int j = 0; j < b.nw; ++j) if (phase_overlap(a.r[i], b.w[j])) return true;
    return false;
}
// a slice of the split-K / matrix-vector workspace for one task of the open phase (rolling: a wrap-around is ordered by the
// hazard analysis like any other reuse)
inline double* phase_ws(double* ws, long ws_doubles, long need) {
    PhaseQueue& P = g_phase;
    if (!ws || need > ws_doubles) return nullptr;
    P.ws = ws;
    P.ws_doubles = ws_doubles;
    need = (need + 15) & ~15L;
    if (P.ws_cursor + need > ws_doubles) P.ws_cursor = 0;
    double* p = ws + P.ws_cursor;
    P.ws_cursor += need;
    return p;
}

// Accumulation fusion.  A term sequence accumulates its products into one array, C = W; C += X1; C += X2; ... — a chain in the
// hazard graph (every member reads and writes C) although the X_i are independent of each other.  Where the array is
// contiguous and no other task touches it in between, the members after the first toucher write alpha X_i into private
// buffers (slices of the split-K workspace) instead, all in the level their operands allow, and ONE element-wise task
// C = C + P_1 + P_2 + ... (fixed order: deterministic) follows the last of them: a chain of n becomes two levels.  The
// singles residual (six products into a v x o array), Q_kb + W_kb, the X_ac T / t Q_kb / V_abic t sum of the finish and the
// one-index terms of a sigma build are such chains.  Costs one write and one read of the array per redirected member, so
// only small arrays take part.
void phase_fuse_accumulations(std::vector<PhaseRec>& q) {
    PhaseQueue& P = g_phase;
    // (arrays up to 2 MB: a redirected member costs one write and one read of the array more.  Measured at (20,80), same box:
    // 1.486 ms without fusion, 1.478 with the cap at 2 MB — the six-term singles residual and its kin —, 1.512 at 8 MB (the
    // 5-MB Q_kb + W_kb sum), 1.525 at 32 MB (the amplitude-sized sums of the finish); PYMES_PHASE_FUSE_MB overrides, 0 switches the fusion off)
    const uintptr_t fuse_bytes = P.fuse_bytes;
    if (fuse_bytes == 0 || P.serial || !P.ws) return;
    struct Group { PhaseRange box; int first; std::vector<int> members; bool open; };
    std::vector<Group> groups;
    auto touches = [](const PhaseRec& t, const PhaseRange& b) {
        for (int i = 0; i < t.nr; ++i) if (phase_overlap(t.r[i], b)) return true;
        for (int i = 0; i < t.nw; ++i) if (phase_overlap(t.w[i], b)) return true;
        return false;
    };
    auto reads = [](const PhaseRec& t, const PhaseRange& b) {
        for (int i = 0; i < t.nr; ++i) if (phase_overlap(t.r[i], b)) return true;
        return false;
    };
    const int n = (int)q.size();
    for (int i = 0; i < n; ++i) {
        const PhaseRec& t = q[i];
        bool joined = false;
        for (auto& G : groups) {
            if (!G.open || !touches(t, G.box)) continue;
            if (t.acc == 1 && t.box.lo == G.box.lo && t.box.hi == G.box.hi && !reads(t, G.box) && G.members.size() < 7) {
                G.members.push_back(i);
                joined = true;
            } else {
                G.open = false;
            }
        }
        if (!joined && t.acc >= 0 && t.box.hi > t.box.lo && t.box.hi - t.box.lo <= fuse_bytes)
            groups.push_back(Group{t.box, i, {}, true});
    }
    struct Insert { int after; PhaseRec rec; };
    std::vector<Insert> ins;
    for (auto& G : groups) {
        if (G.members.empty()) continue;
        const long len = (long)((G.box.hi - G.box.lo) / 8);
        LinK lk;
        lk.out = reinterpret_cast<double*>(G.box.lo);
        lk.n = len;
        for (int k = 0; k < 8; ++k) { lk.p.x[k] = nullptr; lk.p.c[k] = 0.0; }
        lk.p.x[0] = lk.out;
        lk.p.c[0] = 1.0;
        int nx = 1, last = -1;
        PhaseRec comb;
        for (int m : G.members) {
            double* buf = phase_ws(P.ws, P.ws_doubles, len);
            if (!buf) break;
            PhaseRec& t = q[m];
            if (t.kind == PK_GEMM) {
                GemmK k;
                memcpy(&k, t.blob, sizeof k);
                k.C = buf; k.Cin = buf; k.beta = 0.0;
                memcpy(t.blob, &k, sizeof k);
            } else if (t.kind == PK_SPLITK) {
                SplitkTaskK k;
                memcpy(&k, t.blob, sizeof k);
                k.g.C = buf; k.g.Cin = buf; k.g.beta = 0.0;
                memcpy(t.blob, &k, sizeof k);
            } else {
                PermTaskK k;
                memcpy(&k, t.blob, sizeof k);
                k.p.out = buf; k.p.beta = 0.0;
                memcpy(t.blob, &k, sizeof k);
            }
            const PhaseRange rb = prange(buf, len);
            for (int w = 0; w < t.nw; ++w)
                if (t.w[w].lo == G.box.lo && t.w[w].hi == G.box.hi) t.w[w] = rb;
            t.box = PhaseRange{0, 0};
            t.acc = -1;
            lk.p.x[nx] = buf;
            lk.p.c[nx] = 1.0;
            if (comb.nr < 10) comb.r[comb.nr++] = rb;
            ++nx;
            last = m;
        }
        if (nx == 1) continue;
        lk.nx = nx;
        comb.kind = PK_LINCOMB; comb.sub = 0;
        comb.nblk = grid_for(len);
        comb.lds = 0;
        comb.cost = 8.0 * (double)(nx + 1) * (double)len / 4.0e6;
        comb.words = (int)(sizeof(LinK) / 8);
        memcpy(comb.blob, &lk, sizeof lk);
        comb.w[comb.nw++] = G.box;
        ins.push_back(Insert{last, comb});
        ++P.fused;
    }
    if (ins.empty()) return;
    std::stable_sort(ins.begin(), ins.end(), [](const Insert& a, const Insert& b) { return a.after > b.after; });
    for (const auto& x : ins) q.insert(q.begin() + x.after + 1, x.rec);
}

void phase_flush() {
    PhaseQueue& P = g_phase;
    if (P.q.empty() || P.flushing) return;
    struct Guard {
        PhaseQueue& P;
        ~Guard() { P.flushing = false; P.q.clear(); P.ws_cursor = 0; P.ws = nullptr; P.ws_doubles = 0; }
    } guard{P};
    P.flushing = true;
    ++P.flushes;
    std::vector<PhaseRec>& q = P.q;
    phase_fuse_accumulations(q);
    const int n = (int)q.size();
    int nlev = 0;
    for (int i = 0; i < n; ++i) {
        int lev = 0;
        if (P.serial) lev = i;
        else
            for (int j = i - 1; j >= 0; --j)
                if (q[j].level >= lev && phase_conflict(q[j], q[i])) lev = q[j].level + 1;
        q[i].level = lev;
        nlev = std::max(nlev, lev + 1);
    }
    if (P.log) {                      // PYMES_PHASE_LOG: one line per level, kind:blocks:cost of its tasks
        static const char* names[] = {"gemm", "splitk", "perm", "permT", "gemvC", "gemvF", "gemvR", "update", "lincomb", "dots1", "dots2",
                                      "energy", "tau", "packT", "unpackL", "ringops", "layouts", "assemble", "unpackR", "fockG", "fockGf",
                                      "fockFt", "fockFin", "traces", "dotsF"};
        fprintf(stderr, "[phase] flush %ld: %d tasks, %d levels\n", P.flushes, n, nlev);
        for (int lev = 0; lev < nlev; ++lev) {
            fprintf(stderr, "[phase]   L%-2d", lev);
            for (int i = 0; i < n; ++i)
                if (q[i].level == lev) fprintf(stderr, " %s:%d:%.1f", q[i].kind < PK_KINDS ? names[q[i].kind] : "?", q[i].nblk, q[i].cost);
            fprintf(stderr, "\n");
        }
    }
    static bool attr_set[kMaxDevices] = {false};
    int dv = 0;
    HIP_CHECK(hipGetDevice(&dv));
    if (dv >= 0 && dv < kMaxDevices && !attr_set[dv]) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(phase_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(phase_light_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        attr_set[dv] = true;
    }
    std::vector<int> idx;
    for (int lev = 0; lev < nlev; ++lev) {
        idx.clear();
        for (int i = 0; i < n; ++i)
            if (q[i].level == lev) idx.push_back(i);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return q[a].cost > q[b].cost; });
        ++P.levels;
        size_t pos = 0;
        while (pos < idx.size()) {
            PhaseK ph;
            ph.n = 0;
            int words = 0, lds = 0;
            long blocks = 0;
            bool heavy = false;
            while (pos < idx.size() && ph.n < kPhaseMaxTasks) {
                const PhaseRec& t = q[idx[pos]];
                const long padded = ((long)t.nblk + 7) & ~7L;
                if (words + t.words > kPhaseBlobWords || blocks + padded > 0x7ffffff0L) break;
                const int i = ph.n++;
                ph.kind[i] = t.kind; ph.sub[i] = t.sub; ph.off[i] = (unsigned short)words;
                ph.nblk[i] = t.nblk;
                memcpy(ph.blob + words, t.blob, 8 * (size_t)t.words);
                words += t.words;
                blocks += padded;
                ph.blk_end[i] = (int)blocks;
                lds = std::max(lds, t.lds);
                heavy = heavy || phase_kind_heavy(t.kind);
                ++pos;
            }
            if (ph.n == 0) throw std::runtime_error("phase: a task does not fit a launch");
            for (int i = ph.n; i < kPhaseMaxTasks; ++i) { ph.blk_end[i] = (int)blocks; ph.nblk[i] = 0; ph.kind[i] = PK_KINDS; ph.sub[i] = 0; ph.off[i] = 0; }
            if (heavy) hipLaunchKernelGGL(phase_kernel, dim3((unsigned)blocks), dim3(kThreads), (size_t)lds, P.st, ph);
            else hipLaunchKernelGGL(phase_light_kernel, dim3((unsigned)blocks), dim3(kThreads), (size_t)lds, P.st, ph);
            HIP_CHECK(hipGetLastError());
            ++P.launches;
        }
    }
}
// Every launch or stream operation below that is NOT recorded as a task goes through these: the open phase is launched
// first, so that the order of effects on the stream is that of immediate execution.
#define PYMES_LAUNCH(...) do { phase_flush(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
#define hipMemcpyAsync(...) (phase_flush(), hipMemcpyAsync(__VA_ARGS__))
#define hipMemsetAsync(...) (phase_flush(), hipMemsetAsync(__VA_ARGS__))
#define hipMemcpy(...) (phase_flush(), hipMemcpy(__VA_ARGS__))
#define hipStreamSynchronize(...) (phase_flush(), hipStreamSynchronize(__VA_ARGS__))
#define hipEventRecord(...) (phase_flush(), hipEventRecord(__VA_ARGS__))
#define hipStreamWaitEvent(...) (phase_flush(), hipStreamWaitEvent(__VA_ARGS__))
#define hipGraphLaunch(...) (phase_flush(), hipGraphLaunch(__VA_ARGS__))
#define hipStreamBeginCapture(...) (phase_flush(), hipStreamBeginCapture(__VA_ARGS__))
#define hipStreamEndCapture(...) (phase_flush(), hipStreamEndCapture(__VA_ARGS__))
#define hipFree(...) (phase_flush(), hipFree(__VA_ARGS__))

void wait_idle(hipStream_t st) { HIP_CHECK(hipStreamSynchronize(st)); }

// Small read-backs that do not drain the stream: a copy into a pinned slot + an event; the host later waits for THAT event
// while the stream goes on with whatever was enqueued behind it (the next iteration's residual kernels).
constexpr int kReadSlots = 16, kReadDoubles = 128;
double* g_read_host[kMaxDevices] = {nullptr};
hipEvent_t g_read_ev[kMaxDevices][kReadSlots];
int g_read_next[kMaxDevices] = {0};
// what a caller holds is a TICKET = slot | generation << 8: a slot that has been handed out again since (16 later starts on
// the device, whoever made them) or was never started is refused instead of returning somebody else's numbers.  The ring is
// shared by every context and thread of the process on that device, hence the lock.
unsigned g_read_gen[kMaxDevices][kReadSlots] = {{0}};
std::mutex g_read_mu;
// slots written by the device itself (dots_final_kernel: results + a sequence word straight into the pinned ring, no copy
// and no event): the wait polls the word.  g_read_flag: pinned, one word per slot; g_read_flagged: the slot's current
// read-back is of that kind
long* g_read_flag[kMaxDevices] = {nullptr};
bool g_read_flagged[kMaxDevices][kReadSlots] = {{false}};
// The host side of dots_final_kernel: wait for `seq` to appear in the pinned word.  A device that never delivers it (a fault
// in an earlier kernel) is found by the stream synchronisation this falls back to after two seconds.
bool poll_flag(const long* flag, long seq, hipStream_t st) {
    const auto t0 = std::chrono::steady_clock::now();
    long spins = 0;
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
            if (st) HIP_CHECK(hipStreamSynchronize(st));
            else HIP_CHECK(hipDeviceSynchronize());
            return __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq;
        }
        __builtin_ia32_pause();
    }
    return true;
}
void ensure_read_ring(int dv) {
    if (g_read_host[dv]) return;
    HIP_CHECK(hipHostMalloc((void**)&g_read_host[dv], sizeof(double) * kReadSlots * kReadDoubles, hipHostMallocMapped | hipHostMallocCoherent));
    HIP_CHECK(hipHostMalloc((void**)&g_read_flag[dv], sizeof(long) * kReadSlots, hipHostMallocMapped | hipHostMallocCoherent));
    for (int i = 0; i < kReadSlots; ++i) {
        g_read_flag[dv][i] = 0;
        HIP_CHECK(hipEventCreateWithFlags(&g_read_ev[dv][i], hipEventDisableTiming));
    }
}
int readback_start_impl(const double* dev_ptr, int n, hipStream_t st) {
    if (n < 1 || n > kReadDoubles) throw std::runtime_error("readback: 1..128 doubles");
    int dv = 0;
    HIP_CHECK(hipGetDevice(&dv));
    if (dv < 0 || dv >= kMaxDevices) throw std::runtime_error("device ordinal out of range");
    std::lock_guard<std::mutex> lock(g_read_mu);
    ensure_read_ring(dv);
    const int slot = g_read_next[dv];
    g_read_next[dv] = (slot + 1) % kReadSlots;
    const unsigned gen = (g_read_gen[dv][slot] = (g_read_gen[dv][slot] + 1) & 0x3fffffu);
    g_read_flagged[dv][slot] = false;
    HIP_CHECK(hipMemcpyAsync(g_read_host[dv] + slot * kReadDoubles, dev_ptr, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipEventRecord(g_read_ev[dv][slot], st));
    return (int)((gen << 8) | (unsigned)slot);
}
// A slot that the DEVICE fills (dots_final_kernel): returns the ticket and the arguments of that kernel's result part
int readback_reserve_flagged(int dv, double** out_dev, long** flag_dev, long* seq) {
    std::lock_guard<std::mutex> lock(g_read_mu);
    ensure_read_ring(dv);
    const int slot = g_read_next[dv];
    g_read_next[dv] = (slot + 1) % kReadSlots;
    const unsigned gen = (g_read_gen[dv][slot] = (g_read_gen[dv][slot] + 1) & 0x3fffffu);
    g_read_flagged[dv][slot] = true;
    // (the word counts on: generation numbers wrap at 2^22, and a word left over from 2^22 x 16 read-backs ago is not a worry)
    *seq = (long)gen | ((long)(slot + 1) << 32);
    void *od = nullptr, *fd = nullptr;
    HIP_CHECK(hipHostGetDevicePointer(&od, g_read_host[dv] + slot * kReadDoubles, 0));
    HIP_CHECK(hipHostGetDevicePointer(&fd, g_read_flag[dv] + slot, 0));
    *out_dev = static_cast<double*>(od);
    *flag_dev = static_cast<long*>(fd);
    return (int)((gen << 8) | (unsigned)slot);
}
void readback_wait_impl(int ticket, double* out, int n) {
    int dv = 0;
    HIP_CHECK(hipGetDevice(&dv));
    const int slot = ticket & 0xff;
    const unsigned gen = (unsigned)ticket >> 8;
    if (ticket < 0 || slot >= kReadSlots || dv < 0 || dv >= kMaxDevices || n < 1 || n > kReadDoubles)
        throw std::runtime_error("readback: bad ticket");
    hipEvent_t ev;
    bool flagged;
    {
        std::lock_guard<std::mutex> lock(g_read_mu);
        if (!g_read_host[dv] || gen == 0 || g_read_gen[dv][slot] != gen)
            throw std::runtime_error("readback: the slot was never started or has been reused since (16 later read-backs)");
        ev = g_read_ev[dv][slot];
        flagged = g_read_flagged[dv][slot];
    }
    if (flagged) {
        if (!poll_flag(g_read_flag[dv] + slot, (long)gen | ((long)(slot + 1) << 32), nullptr))
            throw std::runtime_error("readback: the device never delivered the result");
    } else {
        HIP_CHECK(hipEventSynchronize(ev));
    }
    std::lock_guard<std::mutex> lock(g_read_mu);
    if (g_read_gen[dv][slot] != gen) throw std::runtime_error("readback: the slot was reused while it was awaited");
    for (int i = 0; i < n; ++i) out[i] = g_read_host[dv][slot * kReadDoubles + i];
}

int current_device() {
    int d = 0;
    HIP_CHECK(hipGetDevice(&d));
    if (d < 0 || d >= kMaxDevices) throw std::runtime_error("device ordinal out of range");
    return d;
}
long g_dot_seq[kMaxDevices] = {0};
// arrival counters of dots_final_kernel: a ring of 64 per device, each on a line of its own; a launch takes the next one (the
// last arriving block leaves it at zero; 64 later launches of the process on that device are far behind any launch in flight)
constexpr int kArrivalSlots = 64;
unsigned int* g_arrivals[kMaxDevices] = {nullptr};
unsigned g_arrivals_next[kMaxDevices] = {0};
std::mutex g_arrivals_mu;
void ensure_dot_ws(int d) {
    if (g_dot_ws[d]) return;
    HIP_CHECK(hipMalloc((void**)&g_dot_ws[d], sizeof(double) * (16 * kDotBlocks + 16)));
    // [0,16): results, [16]: the sequence word of dots_final_kernel (device-written: mapped and coherent)
    HIP_CHECK(hipHostMalloc((void**)&g_dot_host[d], sizeof(double) * 24, hipHostMallocMapped | hipHostMallocCoherent));
    memset(g_dot_host[d], 0, sizeof(double) * 24);
    HIP_CHECK(hipMalloc((void**)&g_arrivals[d], sizeof(unsigned int) * 32 * kArrivalSlots));
    HIP_CHECK(hipMemset(g_arrivals[d], 0, sizeof(unsigned int) * 32 * kArrivalSlots));
}

template <int BM, int BN, bool AKC, bool BKC, int VEC, bool STREAM = false>
void launch_gemm(const GemmK& k, long nblocks, hipStream_t st) {
    constexpr int A_T = (AKC ? (BK + 2) * BM : (BM + 16) * BK);
    constexpr int B_T = (BKC ? (BK + 2) * BN : (BN + 16) * BK);
    constexpr size_t lds = (size_t)(STREAM ? 1 : 2) * (A_T + B_T) * sizeof(double);
    static bool attr_set[kMaxDevices] = {false};
    auto fn = dgemm_kernel<BM, BN, AKC, BKC, VEC, STREAM>;
    const int dv = current_device();
    if (!attr_set[dv]) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dv] = true;
    }
    PYMES_LAUNCH(fn, dim3((unsigned)nblocks), dim3(kThreads), lds, st, k);
    HIP_CHECK(hipGetLastError());
}

template <bool AKC, bool BKC>
void launch_gemm_glds(const GemmK& k, long nblocks, hipStream_t st) {
    constexpr int A_T = AKC ? 128 * BK : BK * (128 + 16);
    constexpr int B_T = BKC ? 128 * BK : BK * (128 + 16);
    constexpr size_t lds = (size_t)2 * (A_T + B_T) * sizeof(double);
    static bool attr_set[kMaxDevices] = {false};
    auto fn = dgemm_glds_kernel<AKC, BKC>;
    const int dv = current_device();
    if (!attr_set[dv]) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));
        attr_set[dv] = true;
    }
    PYMES_LAUNCH(fn, dim3((unsigned)nblocks), dim3(kThreads), lds, st, k);
    HIP_CHECK(hipGetLastError());
}

// smallest K range per block that still goes to the LDS-DMA kernel (measured at (20,80), DESIGN 5: below it the
// register-staged kernel wins)
inline long dma_min_k() { return 384L; }

// ---- launch plan of the LDS-DMA kernel -------------------------------------------------------------------------------
// A CU turns out one tile per tile-time tau whether it holds one block or two (two resident blocks share its MFMA pipes),
// and the hardware hands the next block of the grid to whichever CU frees a slot: a launch costs about (work of the
// busiest CU) x tau.  Whole tiles therefore go in rounds of 256 — whole = floor(tiles / 256) * 256 — and the remaining
// `tail` tiles are cut s ways along K so that their tail * s blocks fill rounds of 256 again:
//     cost(s) / tau = ceil(tail s / 256) / s  +  s * 3 us / tau  +  (5 us + 44 ns * tail (s + 1)) / tau      (the last: reduction)
// minimised over s <= 16 (64 for a handful of tiles) with every cut >= dma_min_k deep.  All of it is ONE grid (the cut blocks take the second slot of
// the CUs that run the last whole tiles; no launch boundary), cut blocks in ks-major order.  Fitted to, and checked
// against, profiles/r05/probe_gemm_plan_grid.txt (19 tile counts x 5 depths x 3 choices of `whole` x 7 cut counts): the
// formula's ranking of s is the measured one, whole = floor(tiles / 256) * 256 beats floor(tiles / 512) * 512 and 0 wherever
// they differ, e.g. 338 tiles x 100 k-tiles (the two ring builds of a (20,80) iteration): 256 + 82/3 280 us, 0 + 338/3 301,
// 338 whole 394; 790 x 625 (a rank's slab of a (50,200) ring product on eight): 768 + 22/10 3411 us, 512 + 278/6 3567,
// 790 whole 4695.
struct DmaPlan { long whole, tail; int s; };
inline DmaPlan plan_dma(long tiles, long ktiles, long ws_tiles) {
    const long whole = (tiles / 256) * 256, tail = tiles - whole;
    DmaPlan best{tiles, 0, 1};
    const long max_cuts = std::max<long>(1, ktiles / (dma_min_k() / BK));
    if (tail > 0) {
        const double tau = 1.75 * (double)ktiles;                      // us per tile
        // (up to 16 cuts; a handful of tiles over a huge K — tiny outputs contracted over o v^2 or o^2 v — up to 64: 4 tiles x
        // 6750 k-tiles ran as 64 blocks on 256 CUs with the cap at 16, 767 us of a sigma build)
        const long cap = tail * 16 >= 256 ? 16 : std::min<long>(64, (512 + tail - 1) / tail);
        const long smax = std::max<long>(1, std::min<long>(std::min<long>(cap, max_cuts), ws_tiles / tail));
        double best_cost = 1e300;
        for (long sp = 1; sp <= smax; ++sp) {
            const long kt_per = (ktiles + sp - 1) / sp, ns = (ktiles + kt_per - 1) / kt_per;       // the cuts that result
            if (ns != sp) continue;
            double c = (double)((tail * sp + 255) / 256) / (double)sp + (double)sp * 3.0 / tau;
            if (sp > 1) c += (5.0 + 0.0437 * (double)tail * (double)(sp + 1)) / tau;
            if (c < best_cost - 1e-9) { best_cost = c; best = DmaPlan{whole, tail, (int)sp}; }
        }
        if (best.s == 1) best = DmaPlan{tiles, 0, 1};
    }
    if (const char* e = getenv("PYMES_GEMM_PLAN")) {          // tuning experiments: "whole,cuts"
        long w = 0, sp = 1;
        if (sscanf(e, "%ld,%ld", &w, &sp) == 2 && w >= 0 && w <= tiles && (w % 8 == 0 || w == tiles) && sp >= 1 &&
            sp <= max_cuts && (tiles - w) * sp <= ws_tiles)
            best = DmaPlan{w, tiles - w, (int)sp};
    }
    return best;
}

template <int BM, int BN, bool AKC, bool BKC>
void launch_stream64(const GemmK& k, int vec, long nblocks, hipStream_t st) {
    if (vec == 2) launch_gemm<BM, BN, AKC, BKC, 2, true>(k, nblocks, st);
    else launch_gemm<BM, BN, AKC, BKC, 1, true>(k, nblocks, st);
}
thread_local bool g_stream64 = false;      // set by dev::gemm for the launches of one call

template <int BM, int BN>
bool dispatch_layout(const GemmK& k, bool akc, bool bkc, int vec, long nblocks, hipStream_t st) {   // (register-staged kernels)
    if constexpr (BM <= 64 && BN <= 64) {
        // (the narrow tiles 64 x 32 / 32 x 64 exist for the streaming shapes only)
        if (g_stream64 || BM < 64 || BN < 64) {
            if (akc && bkc) launch_stream64<BM, BN, true, true>(k, vec, nblocks, st);
            else if (akc) launch_stream64<BM, BN, true, false>(k, vec, nblocks, st);
            else if (bkc) launch_stream64<BM, BN, false, true>(k, vec, nblocks, st);
            else launch_stream64<BM, BN, false, false>(k, vec, nblocks, st);
            return false;
        }
    }
    if (vec == 2) {
        if (akc && bkc) launch_gemm<BM, BN, true, true, 2>(k, nblocks, st);
        else if (akc) launch_gemm<BM, BN, true, false, 2>(k, nblocks, st);
        else if (bkc) launch_gemm<BM, BN, false, true, 2>(k, nblocks, st);
        else launch_gemm<BM, BN, false, false, 2>(k, nblocks, st);
    } else {
        if (akc && bkc) launch_gemm<BM, BN, true, true, 1>(k, nblocks, st);
        else if (akc) launch_gemm<BM, BN, true, false, 1>(k, nblocks, st);
        else if (bkc) launch_gemm<BM, BN, false, true, 1>(k, nblocks, st);
        else launch_gemm<BM, BN, false, false, 1>(k, nblocks, st);
    }
    return false;
}

inline bool even(long x) { return (x & 1) == 0; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }


// M = 1 or N = 1 (no batch): matrix-vector product on the streaming kernels above.  Returns false when the shape does not
// qualify (then the MFMA GEMM handles it).  Timed under the same profiling events as a GEMM call (class 0).
bool gemv_dispatch(const dev::Gemm& g, long a_sm, long a_sk, long b_sk, long b_sn, hipStream_t st);

// dev::gemv_batch_begin/end: weighted-column-sum products with beta == 0 issued in between are collected and launched
// together; any other launch that could read their results (permute, a GEMM of another kind) flushes them first, so the
// order of effects on the stream is that of immediate execution.
struct GemvBatch {
    bool active = false;
    GemvTable tab;
    long ws_used = 0;
    double* ws = nullptr;
    hipStream_t st = nullptr;
};
thread_local GemvBatch g_gemv_batch;

void gemv_batch_flush() {
    GemvBatch& b = g_gemv_batch;
    if (b.tab.n == 0) return;
    int blocks = 0, outs = 0;
    for (int i = 0; i < b.tab.n; ++i) {
        GemvItem& it = b.tab.it[i];
        it.blk0 = blocks;
        it.out0 = outs;
        blocks += it.cblocks * it.nchunk;
        outs += (int)((it.C + 255) / 256);
    }
    PYMES_LAUNCH(gemv_multi_cols_kernel, dim3((unsigned)blocks), dim3(256), 0, b.st, b.tab, b.ws);
    HIP_CHECK(hipGetLastError());
    PYMES_LAUNCH(gemv_multi_finish_kernel, dim3((unsigned)outs), dim3(256), 0, b.st, b.tab, b.ws);
    HIP_CHECK(hipGetLastError());
    b.tab.n = 0;
    b.ws_used = 0;
}

// ---- dev::gemm_group_begin / _end: small products issued in between are queued and launched together ----------------
struct GemmGroup {
    static constexpr int kMaxDepth = 8;
    bool active = false;
    int depth = 0;
    hipStream_t outer[kMaxDepth] = {nullptr};
    hipStream_t st = nullptr;
    int n = 0;
    GemmK k[kGroupMax];
    int variant[kGroupMax];
    long tiles[kGroupMax], ktiles[kGroupMax];
    double flops[kGroupMax];
    double* ws = nullptr;
    long ws_doubles = 0;
    long launches = 0, products = 0;      // statistics since gemm_group_begin (tests, tuning)
    // mid-size products for the grouped LDS-DMA launch (128 x 128 tiles, A K-contiguous, B N-contiguous)
    int nd = 0;
    GemmK kd[kGroupMax];
    long tiles_d[kGroupMax], ktiles_d[kGroupMax];
    double flops_d[kGroupMax];
};
thread_local GemmGroup g_group;
bool gemm_group_pending() { return g_group.n > 0 || g_group.nd > 0; }

void gemm_group_flush_dma() {
    GemmGroup& q = g_group;
    if (q.nd == 0) return;
    hipStream_t st = q.st;
    long total = 0;
    for (int i = 0; i < q.nd; ++i) total += q.tiles_d[i];
    // How many ways every product is cut along K (each cut at least dma_min_k deep): the rule of plan_dma for the tiles of the
    // whole group — a CU turns out one tile per tile-time however many blocks it holds, so s cuts cost about
    // ceil(total s / 256) / s tile-times, + 3 us per cut and the reduction of the partial tiles (profiles/r05/
    // probe_gemm_plan_grid.txt).  A single cut is never chosen when a deeper one is possible: an uncut block applies beta
    // in its own epilogue, element by element.
    long want = 1;
    {
        long ktmax = 0;
        for (int i = 0; i < q.nd; ++i) ktmax = std::max(ktmax, q.ktiles_d[i]);
        const double tau = 1.75 * (double)ktmax;
        double best = 1e300;
        long smax = 1;                 // (beyond the deepest product's limit a larger s changes nothing but the divisor below)
        for (int i = 0; i < q.nd; ++i) smax = std::max(smax, q.ktiles_d[i] / (dma_min_k() / BK));
        for (long sp = 2; sp <= std::min<long>(16, smax); ++sp) {
            long blocks = 0;
            bool any_cut = false;
            for (int i = 0; i < q.nd; ++i) {
                const long si = std::min(sp, std::max<long>(1, q.ktiles_d[i] / (dma_min_k() / BK)));
                any_cut = any_cut || si > 1;
                blocks += q.tiles_d[i] * si;
            }
            if (!any_cut) break;
            if (blocks * 16384 > q.ws_doubles) break;
            const double cost = (double)((blocks + 255) / 256) / (double)sp + (double)sp * 3.0 / tau +
                                (5.0 + 0.0437 * (double)(blocks + total)) / tau;
            if (cost < best - 1e-9) { best = cost; want = sp; }
        }
    }
    GroupK grp, red;
    grp.n = q.nd; grp.tile = 128; red.n = 0; red.tile = 128;
    long blocks = 0, ws_used = 0, red_blocks = 0;
    double flops = 0.0;
    for (int i = 0; i < q.nd; ++i) {
        GemmK& k = q.kd[i];
        long sp = std::max<long>(1, std::min<long>(std::min<long>(want, 16), q.ktiles_d[i] / (dma_min_k() / BK)));
        while (sp > 1 && ws_used + q.tiles_d[i] * sp * 16384 > q.ws_doubles) --sp;
        const long kt_per = (q.ktiles_d[i] + sp - 1) / sp;
        k.kchunk = (int)std::max<long>(kt_per * BK, BK);
        k.nsplit = (int)std::max<long>(1, (q.ktiles_d[i] + kt_per - 1) / std::max<long>(kt_per, 1));
        k.tile_begin = 0;
        k.ws = nullptr;
        if (k.nsplit > 1) {
            k.ws = q.ws + ws_used;
            ws_used += q.tiles_d[i] * k.nsplit * 16384;
            red.g[red.n] = k;
            red.variant[red.n] = 0;
            red_blocks += q.tiles_d[i] * 64;
            red.blk_end[red.n] = (int)red_blocks;
            ++red.n;
        }
        blocks += q.tiles_d[i] * k.nsplit;
        grp.blk_end[i] = (int)blocks;
        grp.variant[i] = 5;
        grp.g[i] = k;
        flops += q.flops_d[i];
    }
    for (int i = q.nd; i < kGroupMax; ++i) { grp.blk_end[i] = (int)blocks; grp.variant[i] = 0; }
    for (int i = red.n; i < kGroupMax; ++i) { red.blk_end[i] = (int)red_blocks; red.variant[i] = 0; }
    const int nq = q.nd;
    q.nd = 0;
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (g_prof.on) {
        if (!g_prof.pool.empty()) { ev = g_prof.pool.back(); g_prof.pool.pop_back(); }
        else { HIP_CHECK(hipEventCreate(&ev.first)); HIP_CHECK(hipEventCreate(&ev.second)); }
        HIP_CHECK(hipEventRecord(ev.first, st));
    }
    constexpr size_t lds = (size_t)2 * (128 * BK + BK * (128 + 16)) * sizeof(double);
    static bool attr_set[kMaxDevices] = {false};
    const int dv = current_device();
    if (!attr_set[dv]) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dgemm_glds_group_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dv] = true;
    }
    PYMES_LAUNCH(dgemm_glds_group_kernel, dim3((unsigned)blocks), dim3(kThreads), lds, st, grp);
    HIP_CHECK(hipGetLastError());
    if (red.n > 0) {
        PYMES_LAUNCH(splitk_reduce_group_kernel, dim3((unsigned)red_blocks), dim3(256), 0, st, red);
        HIP_CHECK(hipGetLastError());
    }
    ++q.launches;
    if (g_prof.on) {
        HIP_CHECK(hipEventRecord(ev.second, st));
        g_prof.ev.push_back(ev);
        g_prof.flops += flops;
        g_prof.fl.push_back(flops);
        g_prof.klass.push_back(1);
        g_prof.nk.push_back(1);
        char buf[128];
        snprintf(buf, sizeof buf, "LDS-DMA group of %d products, %ld blocks, %d k-split, flops=%.4e", nq, blocks, red.n, flops);
        g_prof.what.push_back(buf);
    }
}

void gemm_group_flush() {
    GemmGroup& q = g_group;
    gemm_group_flush_dma();
    if (q.n == 0) return;
    hipStream_t st = q.st;
    // k-splitting: 64 x 64 tiles fill the chip from ~1024 blocks (four co-resident per CU); with fewer tiles in the whole
    // group every product that is deep enough is cut along K by a common factor (>= 8 k-tiles per cut, workspace permitting)
    long total = 0;
    for (int i = 0; i < q.n; ++i) total += q.tiles[i];
    long want = total < 768 ? (1024 + total - 1) / total : 1;
    GroupK grp, red;
    grp.n = q.n; grp.tile = 64; red.n = 0; red.tile = 64;
    long blocks = 0, ws_used = 0, red_blocks = 0;
    double flops = 0.0;
    for (int i = 0; i < q.n; ++i) {
        GemmK& k = q.k[i];
        long sp = std::max<long>(1, std::min<long>(std::min<long>(want, 64), q.ktiles[i] / 8));
        while (sp > 1 && ws_used + q.tiles[i] * sp * 4096 > q.ws_doubles) --sp;
        const long kt_per = (q.ktiles[i] + sp - 1) / sp;
        k.kchunk = (int)std::max<long>(kt_per * BK, BK);
        k.nsplit = (int)std::max<long>(1, (q.ktiles[i] + kt_per - 1) / std::max<long>(kt_per, 1));
        k.tile_begin = 0;
        k.ws = nullptr;
        if (k.nsplit > 1) {
            k.ws = q.ws + ws_used;
            ws_used += q.tiles[i] * k.nsplit * 4096;
            red.g[red.n] = k;
            red.variant[red.n] = 0;
            red_blocks += q.tiles[i] * 16;
            red.blk_end[red.n] = (int)red_blocks;
            ++red.n;
        }
        blocks += q.tiles[i] * k.nsplit;
        grp.blk_end[i] = (int)blocks;
        grp.variant[i] = q.variant[i];
        grp.g[i] = k;
        flops += q.flops[i];
    }
    for (int i = q.n; i < kGroupMax; ++i) { grp.blk_end[i] = (int)blocks; grp.variant[i] = 0; }
    for (int i = red.n; i < kGroupMax; ++i) { red.blk_end[i] = (int)red_blocks; red.variant[i] = 0; }
    const int nq = q.n;
    q.n = 0;                                   // (an exception below must not leave the queue half-consumed)
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (g_prof.on) {
        if (!g_prof.pool.empty()) { ev = g_prof.pool.back(); g_prof.pool.pop_back(); }
        else { HIP_CHECK(hipEventCreate(&ev.first)); HIP_CHECK(hipEventCreate(&ev.second)); }
        HIP_CHECK(hipEventRecord(ev.first, st));
    }
    constexpr size_t lds = (size_t)2 * 2 * ((64 + 16) * BK) * sizeof(double);       // the widest variant, double buffered
    static bool attr_set[kMaxDevices] = {false};
    const int dv = current_device();
    if (!attr_set[dv]) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dgemm_group_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dv] = true;
    }
    PYMES_LAUNCH(dgemm_group_kernel, dim3((unsigned)blocks), dim3(kThreads), lds, st, grp);
    HIP_CHECK(hipGetLastError());
    if (red.n > 0) {
        PYMES_LAUNCH(splitk_reduce_group_kernel, dim3((unsigned)red_blocks), dim3(256), 0, st, red);
        HIP_CHECK(hipGetLastError());
    }
    ++q.launches;
    if (g_prof.on) {
        HIP_CHECK(hipEventRecord(ev.second, st));
        g_prof.ev.push_back(ev);
        g_prof.flops += flops;
        g_prof.fl.push_back(flops);
        g_prof.klass.push_back(0);
        g_prof.nk.push_back(1);
        char buf[128];
        snprintf(buf, sizeof buf, "group of %d products, %ld blocks, %d k-split, flops=%.4e", nq, blocks, red.n, flops);
        g_prof.what.push_back(buf);
    }
}
}  // namespace

namespace dev {

const char* backend_name() { return "hip-gfx950"; }

void set_device(int ordinal) {
    int cur = -1;
    HIP_CHECK(hipGetDevice(&cur));
    if (cur != ordinal) {      // launches queued for the device that is being left (an open group / batch of this thread) go first
        gemv_batch_flush();
        gemm_group_flush();
    }
    HIP_CHECK(hipSetDevice(ordinal));
}

void* dmalloc(size_t bytes) {
    void* p = nullptr;
    HIP_CHECK(hipMalloc(&p, bytes ? bytes : 16));
    ++g_live_allocs;
    return p;
}
void* try_dmalloc(size_t bytes) {
    void* p = nullptr;
    const hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return nullptr; }
    HIP_CHECK(e);
    ++g_live_allocs;
    return p;
}
void dfree(void* p) {
    if (!p) return;
    HIP_CHECK(hipFree(p));
    --g_live_allocs;
}
int64_t live_allocations() { return g_live_allocs; }
stream_t stream_create() {
    hipStream_t s = nullptr;
    HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return (stream_t)s;
}
void stream_destroy(stream_t s) {
    if (s) HIP_CHECK(hipStreamDestroy((hipStream_t)s));
}

event_t event_create() {
    hipEvent_t e = nullptr;
    HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return (event_t)e;
}
void event_destroy(event_t e) {
    if (e) HIP_CHECK(hipEventDestroy((hipEvent_t)e));
}
void event_record(event_t e, stream_t s) { HIP_CHECK(hipEventRecord((hipEvent_t)e, (hipStream_t)s)); }
void stream_wait_event(stream_t s, event_t e) { HIP_CHECK(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)e, 0)); }

bool graphs_supported() { return true; }
void graph_begin(stream_t s) {
    if (!s) throw std::runtime_error("graph capture needs a stream of its own (not the default stream)");
    HIP_CHECK(hipStreamBeginCapture((hipStream_t)s, hipStreamCaptureModeRelaxed));
}
graph_t graph_end(stream_t s) {
    gemm_group_flush();
    hipGraph_t g = nullptr;
    HIP_CHECK(hipStreamEndCapture((hipStream_t)s, &g));
    if (!g) throw std::runtime_error("graph capture produced no graph");
    hipGraphExec_t ex = nullptr;
    const hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    HIP_CHECK(e);
    return (graph_t)ex;
}
void graph_abort(stream_t s) {
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture((hipStream_t)s, &g);
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
}
void graph_launch(graph_t g, stream_t s) { HIP_CHECK(hipGraphLaunch((hipGraphExec_t)g, (hipStream_t)s)); }
void graph_destroy(graph_t g) {
    if (g) HIP_CHECK(hipGraphExecDestroy((hipGraphExec_t)g));
}
void memcpy_h2d(void* d, const void* h, size_t bytes, stream_t s) {
    gemm_group_flush();
    HIP_CHECK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, (hipStream_t)s));
    wait_idle((hipStream_t)s);
}
void memcpy_d2h(void* h, const void* d, size_t bytes, stream_t s) {
    gemm_group_flush();
    HIP_CHECK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, (hipStream_t)s));
    wait_idle((hipStream_t)s);
}
void memcpy_d2d(void* d, const void* s_, size_t bytes, stream_t s) {
    gemm_group_flush();
    HIP_CHECK(hipMemcpyAsync(d, s_, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
}
void memset_zero(void* d, size_t bytes, stream_t s) {
    gemm_group_flush();
    HIP_CHECK(hipMemsetAsync(d, 0, bytes, (hipStream_t)s));
}
void stream_sync(stream_t s) {
    gemm_group_flush();
    wait_idle((hipStream_t)s);
}
size_t mem_free_bytes() {
    size_t f = 0, t = 0;
    HIP_CHECK(hipMemGetInfo(&f, &t));
    return f;
}
size_t mem_total_bytes() {
    size_t f = 0, t = 0;
    HIP_CHECK(hipMemGetInfo(&f, &t));
    return t;
}

void prof_enable(bool on) { g_prof.on = on; }
void prof_reset() {
    for (auto& e : g_prof.ev) g_prof.pool.push_back(e);
    g_prof.ev.clear();
    g_prof.what.clear();
    g_prof.fl.clear();
    g_prof.klass.clear();
    g_prof.nk.clear();
    g_prof.flops = 0.0;
}
void prof_query(int kernel_class, long* calls, long* kernel_launches, double* ms, double* flops) {
    double tot = 0.0, fl = 0.0;
    long nc = 0, nk = 0;
    // PYMES_GEMM_LOG=<file>: one line per GEMM call (shape, tile, time) for tuning
    const char* logp = getenv("PYMES_GEMM_LOG");
    FILE* lf = (logp && kernel_class == 0) ? fopen(logp, "a") : nullptr;
    for (size_t i = 0; i < g_prof.ev.size(); ++i) {
        auto& e = g_prof.ev[i];
        HIP_CHECK(hipEventSynchronize(e.second));
        float t = 0.f;
        HIP_CHECK(hipEventElapsedTime(&t, e.first, e.second));
        if (lf) fprintf(lf, "%s ms=%.4f\n", g_prof.what[i].c_str(), t);
        if (kernel_class == 1 && g_prof.klass[i] != 1) continue;
        tot += t;
        fl += g_prof.fl[i];
        nc += 1;
        nk += g_prof.nk[i];
    }
    if (lf) { fprintf(lf, "----\n"); fclose(lf); }
    *calls = nc;
    *kernel_launches = nk;
    *ms = tot;
    *flops = fl;
}

namespace {
// Queue a mid-size product in the LDS-DMA layout for the grouped 128 x 128 launch of the open group (tried before the open
// phase gets the product: deep products belong on the LDS-DMA kernel, not on the 64 x 64 tiles of a phase).
bool gemm_group_take_dma(const Gemm& g, bool a_kcontig, bool b_kcontig, int64_t a_sm, int64_t a_sk, int64_t b_sk, int64_t b_sn,
                         hipStream_t st) {
    GemmGroup& q = g_group;
    if (q.st != st) return false;
    const long nbatch = g.nb1 * g.nb2;
    const long t128 = ((g.M + 127) / 128) * ((g.N + 127) / 128) * nbatch;
    const long ktiles = (g.K + BK - 1) / BK;
    if (t128 >= 256 && ktiles * BK >= 256) return false;
    {
        const long a_ld = a_kcontig ? a_sm : a_sk, b_ld = b_kcontig ? b_sn : b_sk;
        const bool vec2 = even(a_ld) && even(b_ld) && even(g.K) && (even(g.N) || b_ld > g.N) && aligned16(g.A) && aligned16(g.B) &&
                          even(g.a_b1) && even(g.a_b2) && even(g.b_b1) && even(g.b_b2);
        const bool off32 = 128 * a_ld * 8 + 4096 < (1L << 32) && 16 * b_ld * 8 + 4096 < (1L << 32);
        // (enough blocks of >= dma_min_k depth for half the chip, else the 64 x 64 group fills it better)
        const bool fills = t128 * std::min<long>(16, g.K / dma_min_k()) >= 128;
        if (a_kcontig && !b_kcontig && vec2 && off32 && g.M > 64 && g.N > 64 && g.K >= 2 * dma_min_k() && g.splitk_ws && fills &&
            t128 <= 0x3fffffffL / 64) {
            if (q.nd == kGroupMax) gemm_group_flush_dma();
            GemmK k;
            k.A = g.A; k.B = g.B; k.C = g.C;
            k.Cin = g.Cin ? g.Cin : g.C;
            k.a_ld = a_ld; k.b_ld = b_ld; k.ldc = g.ldc;
            k.M = (int)g.M; k.N = (int)g.N; k.K = (int)g.K;
            k.Mc = k.M;
            k.Nc = ((g.N & 1) && b_ld > g.N) ? k.N + 1 : k.N;
            k.alpha = g.alpha; k.beta = g.beta;
            k.nb2 = g.nb2;
            k.a_b1 = g.a_b1; k.a_b2 = g.a_b2; k.b_b1 = g.b_b1; k.b_b2 = g.b_b2; k.c_b1 = g.c_b1; k.c_b2 = g.c_b2;
            k.ws = nullptr;
            k.tiles_m = (int)((g.M + 127) / 128);
            k.tiles_n = (int)((g.N + 127) / 128);
            const int i = q.nd++;
            q.kd[i] = k;
            q.tiles_d[i] = t128;
            q.ktiles_d[i] = ktiles;
            q.flops_d[i] = 2.0 * (double)g.M * (double)g.N * (double)g.K * (double)nbatch;
            q.ws = g.splitk_ws;
            q.ws_doubles = g.splitk_ws_doubles;
            ++q.products;
            return true;
        }
    }
    return false;
}
// Queue a product for the open group.  false: not a small product (it runs as its own launch, after the queue).
bool gemm_group_take(const Gemm& g, bool a_kcontig, bool b_kcontig, int64_t a_sm, int64_t a_sk, int64_t b_sk, int64_t b_sn,
                     hipStream_t st) {
    GemmGroup& q = g_group;
    if (q.st != st) return false;
    const long nbatch = g.nb1 * g.nb2;
    const long t64 = ((g.M + 63) / 64) * ((g.N + 63) / 64) * nbatch, t128 = ((g.M + 127) / 128) * ((g.N + 127) / 128) * nbatch;
    const long ktiles = (g.K + BK - 1) / BK;
    // big products keep their own launches: enough 128 x 128 tiles for the chip, or deep enough for the LDS-DMA kernel's
    // k-split; and so do the long streaming products (one pass over a multi-GB block: the single-buffer kernel moves more)
    if (t128 >= 256 && ktiles * BK >= 256) return false;
    if (gemm_group_take_dma(g, a_kcontig, b_kcontig, a_sm, a_sk, b_sk, b_sn, st)) return true;
    if (t128 < 256 && g.K / std::max<long>(1, (512 + t128 - 1) / t128) >= dma_min_k() && g.M > 64 && g.N > 64) return false;
    if (t64 * ktiles > 400000 || t64 > 0x3fffffffL / 16) return false;
    // streaming shapes (short K against a skinny side; tiny outputs over a huge K): one pass over a big operand, bound by the
    // bytes in flight — the single-buffer kernel of dev::gemm with its 4-6 blocks per CU, not the double-buffered group kernel
    // (from a few rounds of blocks on: tiny ones lose nothing in a group)
    if (t64 * ktiles >= 4096 && ((g.K <= 256 && std::min(g.M, g.N) <= 256) || (g.M <= 256 && g.N <= 256 && g.K >= 65536)))
        return false;
    GemmK k;
    k.A = g.A; k.B = g.B; k.C = g.C;
    k.Cin = g.Cin ? g.Cin : g.C;
    k.a_ld = a_kcontig ? a_sm : a_sk;
    k.b_ld = b_kcontig ? b_sn : b_sk;
    k.ldc = g.ldc;
    k.M = (int)g.M; k.N = (int)g.N; k.K = (int)g.K;
    k.alpha = g.alpha; k.beta = g.beta;
    k.nb2 = g.nb2;
    k.a_b1 = g.a_b1; k.a_b2 = g.a_b2; k.b_b1 = g.b_b1; k.b_b2 = g.b_b2; k.c_b1 = g.c_b1; k.c_b2 = g.c_b2;
    k.ws = nullptr;
    k.tiles_m = (int)((g.M + 63) / 64);
    k.tiles_n = (int)((g.N + 63) / 64);
    int vec = 2;
    k.Mc = k.M; k.Nc = k.N;
    if (!a_kcontig && (g.M & 1) && k.a_ld > g.M) k.Mc = k.M + 1;
    if (!b_kcontig && (g.N & 1) && k.b_ld > g.N) k.Nc = k.N + 1;
    const long a_contig_extent = a_kcontig ? g.K : k.Mc, b_contig_extent = b_kcontig ? g.K : k.Nc;
    if (!even(k.a_ld) || !even(k.b_ld) || !even(a_contig_extent) || !even(b_contig_extent) || !aligned16(g.A) || !aligned16(g.B) ||
        !even(g.a_b1) || !even(g.a_b2) || !even(g.b_b1) || !even(g.b_b2))
        vec = 1;
    if (q.n == kGroupMax) gemm_group_flush();
    // the grid of a group is one int: flush early when the running block count would not fit comfortably
    long queued = 0;
    for (int i = 0; i < q.n; ++i) queued += q.tiles[i];
    if (queued + t64 > 0x3fffffffL / 16) gemm_group_flush();
    const int i = q.n++;
    q.k[i] = k;
    q.variant[i] = (a_kcontig ? 4 : 0) | (b_kcontig ? 2 : 0) | (vec == 2 ? 1 : 0);
    q.tiles[i] = t64;
    q.ktiles[i] = ktiles;
    q.flops[i] = 2.0 * (double)g.M * (double)g.N * (double)g.K * (double)nbatch;
    q.ws = g.splitk_ws;
    q.ws_doubles = g.splitk_ws ? g.splitk_ws_doubles : 0;
    ++q.products;
    return true;
}
}  // namespace

namespace {
// A small product as tasks of the open phase: 64 x 64 tiles (64 x 32 / 32 x 64 when a side is at most 32), the k-split rule of
// the register-staged launches of dev::gemm, its partial tiles in a slice of the workspace and their reduction as a task
// of the next level.  false: not small (or no phase open) — the caller goes on with the launches of its own.
bool phase_gemm(const Gemm& g, bool akc, bool bkc, int64_t a_sm, int64_t a_sk, int64_t b_sk, int64_t b_sn, hipStream_t st) {
    if (!phase_open(st)) return false;
    const long nbatch = g.nb1 * g.nb2;
    const double dM = (double)g.M, dN = (double)g.N, dK = (double)g.K, dB = (double)nbatch;
    const double cost = 2.0 * dM * dN * dK * dB / 4.0e7 + 8.0 * (dM * dK + dK * dN + (g.beta != 0.0 ? 2.0 : 1.0) * dM * dN) * dB / 4.0e6;
    if (!phase_small(cost)) return false;
    if (g.splitk_ws) { g_phase.ws = g.splitk_ws; g_phase.ws_doubles = g.splitk_ws_doubles; }     // (of THIS phase's engine: cleared by the flush)
    // deep products with enough 128 x 128 tiles for the LDS-DMA kernel (k-split over the chip) belong there, small as they may be
    {
        const long t128 = ((g.M + 127) / 128) * ((g.N + 127) / 128) * nbatch;
        if (g.M > 64 && g.N > 64 && g.K >= 2 * dma_min_k() && t128 * std::min<long>(16, g.K / dma_min_k()) >= 128) return false;
    }
    int BM = 64, BN = 64;
    if (g.N <= 32) BN = 32;
    else if (g.M <= 32) BM = 32;
    GemmK k;
    k.A = g.A; k.B = g.B; k.C = g.C;
    k.Cin = g.Cin ? g.Cin : g.C;
    k.a_ld = akc ? a_sm : a_sk;
    k.b_ld = bkc ? b_sn : b_sk;
    k.ldc = g.ldc;
    k.M = (int)g.M; k.N = (int)g.N; k.K = (int)g.K;
    k.alpha = g.alpha; k.beta = g.beta;
    k.nb2 = g.nb2;
    k.a_b1 = g.a_b1; k.a_b2 = g.a_b2; k.b_b1 = g.b_b1; k.b_b2 = g.b_b2; k.c_b1 = g.c_b1; k.c_b2 = g.c_b2;
    k.ws = nullptr;
    k.tiles_m = (int)((g.M + BM - 1) / BM);
    k.tiles_n = (int)((g.N + BN - 1) / BN);
    const long tiles = (long)k.tiles_m * k.tiles_n * nbatch;
    if (tiles > 0x3fffffffL / 16) return false;
    int vec = 2;
    k.Mc = k.M; k.Nc = k.N;
    if (!akc && (g.M & 1) && k.a_ld > g.M) k.Mc = k.M + 1;
    if (!bkc && (g.N & 1) && k.b_ld > g.N) k.Nc = k.N + 1;
    {
        const long ace = akc ? g.K : k.Mc, bce = bkc ? g.K : k.Nc;
        if (!even(k.a_ld) || !even(k.b_ld) || !even(ace) || !even(bce) || !aligned16(g.A) || !aligned16(g.B) || !even(g.a_b1) ||
            !even(g.a_b2) || !even(g.b_b1) || !even(g.b_b2))
            vec = 1;
    }
    const long ktiles = (g.K + BK - 1) / BK;
    // k-split of the last, partially filled round of blocks (all of them for a small output), as dev::gemm does it
    const long slots = 1024;
    const long ws_cap = g.splitk_ws ? g.splitk_ws_doubles / ((long)BM * BN) : 0;
    long main_tiles = tiles, tail_tiles = 0;
    int tail_split = 1;
    if (ktiles >= 16) {
        const long rem = tiles % slots;
        if (rem > 0) {
            long best = 1;
            double best_cost = 1.0;
            long smax = std::min<long>(512, std::max<long>(8, 2048 / rem));
            smax = std::min<long>(smax, ktiles / 8);
            smax = std::min<long>(smax, ws_cap / rem);
            for (long sp = 2; sp <= smax; ++sp) {
                const double c = (double)((rem * sp + slots - 1) / slots) / (double)sp + 1e-5 * sp;
                if (c < best_cost - 1e-9) { best_cost = c; best = sp; }
            }
            if (best >= 2 && best_cost < 0.8) { tail_tiles = rem; main_tiles = tiles - rem; tail_split = (int)best; }
        }
    }
    const unsigned short sub = (unsigned short)((BM == 64 && BN == 64 ? 0 : (BN == 32 ? 1 : 2)) * 8 + (akc ? 4 : 0) + (bkc ? 2 : 0) + (vec == 2 ? 1 : 0));
    const int a_t = akc ? (BK + 2) * BM : (BM + 16) * BK, b_t = bkc ? (BK + 2) * BN : (BN + 16) * BK;
    const int lds = 2 * (a_t + b_t) * (int)sizeof(double);
    const PhaseRange rA = pbox(g.A, {{g.M, a_sm}, {g.K, a_sk}, {g.nb1, g.a_b1}, {g.nb2, g.a_b2}}, 1);
    const PhaseRange rB = pbox(g.B, {{g.K, b_sk}, {g.N, b_sn}, {g.nb1, g.b_b1}, {g.nb2, g.b_b2}}, 1);
    const PhaseRange rC = pbox(g.C, {{g.M, g.ldc}, {g.N, 1}, {g.nb1, g.c_b1}, {g.nb2, g.c_b2}});
    const PhaseRange rCin = (g.beta != 0.0 && k.Cin != g.C) ? pbox(k.Cin, {{g.M, g.ldc}, {g.N, 1}, {g.nb1, g.c_b1}, {g.nb2, g.c_b2}})
                                                           : PhaseRange{0, 0};
    // (accumulation fusion: the product's output is ONE contiguous array written by one task — no main / tail pair — and it
    // either overwrites it, beta = 0, or accumulates into it, beta = 1 with the beta term read from the output itself)
    const bool dense_c = g.ldc == g.N && (g.nb2 == 1 || g.c_b2 == g.M * g.N) && (g.nb1 == 1 || g.c_b1 == g.nb2 * g.M * g.N);
    const bool whole_dense = dense_c && (main_tiles == 0 || tail_tiles == 0) && (g.beta == 0.0 || (g.beta == 1.0 && k.Cin == g.C));
    const signed char acc = g.beta == 1.0 ? 1 : 0;
    auto add = [&](long tile_begin, long ntiles, int nsplit, double share) {
        const long kt_per = (ktiles + nsplit - 1) / nsplit;
        k.kchunk = (int)std::max<long>(kt_per * BK, BK);
        k.nsplit = (int)std::max<long>(1, (ktiles + kt_per - 1) / std::max<long>(kt_per, 1));
        k.tile_begin = tile_begin;
        k.ws = nullptr;
        if (k.nsplit > 1) {
            k.ws = phase_ws(g.splitk_ws, g.splitk_ws_doubles, ntiles * k.nsplit * (long)BM * BN);
            if (!k.ws) { k.nsplit = 1; k.kchunk = (int)(ktiles * BK); }
        }
        PhaseRec& t = phase_push(PK_GEMM, sub, ntiles * k.nsplit, lds, cost * share, k);
        phase_reads(t, {rA, rB});
        if (k.nsplit > 1) {
            const PhaseRange rW = prange(k.ws, ntiles * k.nsplit * (long)BM * BN);
            phase_writes(t, {rW});
            SplitkTaskK r{k, BM, BN};
            PhaseRec& u = phase_push(PK_SPLITK, 0, ntiles * (BM * BN / 256), 0, 2.0, r);
            phase_reads(u, {rW, rCin});
            phase_writes(u, {rC});
            if (whole_dense) { u.box = rC; u.acc = acc; }
        } else {
            phase_reads(t, {rCin});
            phase_writes(t, {rC});
            if (whole_dense) { t.box = rC; t.acc = acc; }
        }
    };
    const double fmain = tiles > 0 ? (double)main_tiles / (double)tiles : 1.0;
    if (main_tiles > 0) add(0, main_tiles, 1, fmain);
    if (tail_tiles > 0) add(main_tiles, tail_tiles, tail_split, 1.0 - fmain);
    return true;
}
}  // namespace

// begin / end nest (an engine-internal scope inside a caller's): the group stays open until the outermost end; an inner
// begin on another stream flushes what is queued and queues on its own stream until its end, then the outer stream is back
void gemm_group_begin(stream_t s) {
    gemm_group_flush();
    GemmGroup& q = g_group;
    if (q.depth >= GemmGroup::kMaxDepth) throw std::runtime_error("gemm_group_begin: nested too deeply");
    q.outer[q.depth++] = q.st;
    if (q.depth == 1) q.launches = q.products = 0;
    q.active = true;
    q.st = (hipStream_t)s;
}
void gemm_group_end() {
    GemmGroup& q = g_group;
    if (q.depth == 0) return;
    struct Close {          // the group is closed even when the flush throws
        GemmGroup& q;
        ~Close() {
            q.st = q.outer[--q.depth];
            q.active = q.depth > 0;
            if (!q.active) { q.n = 0; q.nd = 0; }
        }
    } close{q};
    gemm_group_flush();
}
void gemm_group_sync() { gemm_group_flush(); }
// (products queued in an open group — dev::gemm_group_begin / _end, which may span several calls of the C interface — stay
// queued: a task is never recorded while one is waiting, phase_push, so what is recorded precedes them in the order of effects)
void phase_sync() {
    gemv_batch_flush();
    phase_flush();
}
bool phase_pending() { return !g_phase.q.empty(); }
long phase_generation() { return g_phase.flushes; }
void phase_enable(int mode) {
    gemm_group_flush();
    phase_sync();
    g_phase.enabled = mode < 0 ? -1 : (mode ? 1 : 0);       // (-1: the environment is read again at the next operation)
    if (mode == 1) g_phase.serial = false;
}
void phase_hold(bool on) {
    g_phase.hold = on;
    if (!on) phase_sync();
}
void phase_call_end() {
    if (!g_phase.hold) phase_sync();
}
void phase_stats(long* tasks, long* launches, long* levels, long* flushes) {
    if (tasks) *tasks = g_phase.tasks;
    if (launches) *launches = g_phase.launches;
    if (levels) *levels = g_phase.levels;
    if (flushes) *flushes = g_phase.flushes;
}
void gemm_group_stats(long* launches, long* products) {
    if (launches) *launches = g_group.launches;
    if (products) *products = g_group.products;
}

void gemm(const Gemm& g, stream_t s) {
    hipStream_t st = (hipStream_t)s;
    if (g.M <= 0 || g.N <= 0 || g.nb1 <= 0 || g.nb2 <= 0) return;
    if (g.M > 0x7fffffffL || g.N > 0x7fffffffL || g.K > 0x7fffffffL)
        throw std::runtime_error("gemm: extent exceeds int32");
    // an extent-1 dimension has no meaningful stride: normalise it to the unit-stride role
    int64_t a_sm = g.a_sm, a_sk = g.a_sk, b_sk = g.b_sk, b_sn = g.b_sn;
    if (a_sk != 1 && a_sm != 1) { if (g.K == 1) a_sk = 1; else if (g.M == 1) a_sm = 1; }
    if (b_sk != 1 && b_sn != 1) { if (g.K == 1) b_sk = 1; else if (g.N == 1) b_sn = 1; }
    if (!(a_sk == 1 || a_sm == 1)) throw std::runtime_error("gemm: A has no unit stride");
    if (!(b_sk == 1 || b_sn == 1)) throw std::runtime_error("gemm: B has no unit stride");
    const bool a_kcontig = (a_sk == 1);
    const bool b_kcontig = (b_sk == 1);
    if (gemv_dispatch(g, a_sm, a_sk, b_sk, b_sn, st)) return;
    gemv_batch_flush();
    if (g_group.active && gemm_group_take_dma(g, a_kcontig, b_kcontig, a_sm, a_sk, b_sk, b_sn, st)) return;
    if (phase_gemm(g, a_kcontig, b_kcontig, a_sm, a_sk, b_sk, b_sn, st)) return;
    if (g_group.active && gemm_group_take(g, a_kcontig, b_kcontig, a_sm, a_sk, b_sk, b_sn, st)) return;
    gemm_group_flush();          // (a product that is launched on its own keeps its place in the order of effects)
    GemmK k;
    k.A = g.A; k.B = g.B; k.C = g.C;
    k.Cin = g.Cin ? g.Cin : g.C;
    k.a_ld = a_kcontig ? a_sm : a_sk;
    k.b_ld = b_kcontig ? b_sn : b_sk;
    k.ldc = g.ldc;
    k.M = (int)g.M; k.N = (int)g.N; k.K = (int)g.K;
    k.alpha = g.alpha; k.beta = g.beta;
    k.nb2 = g.nb2;
    k.a_b1 = g.a_b1; k.a_b2 = g.a_b2; k.b_b1 = g.b_b1; k.b_b2 = g.b_b2; k.c_b1 = g.c_b1; k.c_b2 = g.c_b2;
    k.ws = nullptr;

    // ---- tile shape: 128x128 unless a dimension is small -----------------------------
    int BM = 128, BN = 128;
    if (g.N <= 64) BN = 64;
    if (g.M <= 64) BM = 64;
    const long nbatch = g.nb1 * g.nb2;
    auto ntiles = [&](int bm, int bn) { return ((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * nbatch; };
    // under-filled chip (<1 block per CU at 128x128): 64x64 tiles give 4x the blocks — unless K is deep enough to fill
    // the chip by k-splitting with >= 1024 per split, which keeps the (faster) 128x128 LDS-DMA kernel
    if (BM == 128 && BN == 128 && ntiles(128, 128) < 256) {
        const long nt = ntiles(128, 128), want = (512 + nt - 1) / nt;
        const bool deep = g.splitk_ws && g.K / want >= dma_min_k() && nt * want * 128L * 128L <= g.splitk_ws_doubles;
        if (!deep) { BM = 64; BN = 64; }
    }
    else if (BM == 128 && BN == 64 && ntiles(128, 64) < 256) { BM = 64; }
    else if (BM == 64 && BN == 128 && ntiles(64, 128) < 256) { BN = 64; }
    // short K against a skinny side (the T1 dressing: K = nv or nocc, one side <= nv): the big operand and C are streamed
    // once from HBM and nothing is reused across tiles — 64x64 blocks (more of them resident, more loads in flight) move
    // 13-18 % more bytes per second than the wider tiles (tools/probe_stream.py)
    bool stream = false;
    // (min side up to 512: the stacked EOM build's [X_vv | u1] . [T ; A346] — M = k nv = 480, K = nv + nocc = 150, 0.4 GB written — ran
    // 128 x 128 tiles with ten k-steps each at a quarter of the HBM rate)
    if (g.K <= 256 && std::min(g.M, g.N) <= 512) { BM = 64; BN = 64; stream = true; }
    // ... and the tiny outputs contracted over a huge K (singles residual: 200 x 50 over o v^2 = 2e6), k-split over the chip
    if (g.M <= 256 && g.N <= 256 && g.K >= 65536 && BM == 64 && BN == 64) stream = true;
    // a streaming shape whose skinny side is at most 32 (nocc = 20 against 64-wide tiles: two thirds of the MFMA work on
    // padding, and these launches are MFMA-issue AND bandwidth co-limited): 64 x 32 / 32 x 64 tiles
    // (single-buffer kernel for every narrow launch: the only variant instantiated for them)
    if (BM == 64 && BN == 64) {
        if (g.N <= 32) BN = 32;
        else if (g.M <= 32) BM = 32;
    }
    k.tiles_m = (int)((g.M + BM - 1) / BM);
    k.tiles_n = (int)((g.N + BN - 1) / BN);
    const long tiles = (long)k.tiles_m * k.tiles_n * nbatch;

    // ---- 16-byte global loads need even strides/extents and aligned bases --------------
    // An odd M (N) of an M- (N-)contiguous operand is fine when the pitch has room for one more element: the
    // pair load at the edge then reads a pad element that only feeds a row (column) of C which is never stored.
    int vec = 2;
    k.Mc = k.M; k.Nc = k.N;
    if (!a_kcontig && (g.M & 1) && k.a_ld > g.M) k.Mc = k.M + 1;
    if (!b_kcontig && (g.N & 1) && k.b_ld > g.N) k.Nc = k.N + 1;
    {
        const long a_contig_extent = a_kcontig ? g.K : k.Mc;
        const long b_contig_extent = b_kcontig ? g.K : k.Nc;
        if (!even(k.a_ld) || !even(k.b_ld) || !even(a_contig_extent) || !even(b_contig_extent) ||
            !aligned16(g.A) || !aligned16(g.B) || !even(g.a_b1) || !even(g.a_b2) || !even(g.b_b1) ||
            !even(g.b_b2))
            vec = 1;
    }
    if (tiles > 0x7fffffffL) throw std::runtime_error("gemm: grid too large");

    // ---- k-splitting.  (1) whole problem when the output has too few tiles to fill the chip;
    // (2) only the LAST, partially filled wave of tiles otherwise: tiles all cost the same, so a
    // launch takes ceil(tiles/slots) tile-times; splitting the K range of the remainder tiles over
    // the idle CUs turns that last wave into a fraction of a tile-time.
    const long ktiles = (g.K + BK - 1) / BK;
    // Blocks that share a CU time-share its MFMA pipes, so what has to balance is the number of tiles per CU:
    // a launch costs about ceil(tiles / 256) tile-times (64x64 tiles need 4 co-resident blocks to fill a CU).
    const long slots = (BM <= 64 && BN <= 64) ? 1024 : 256;
    const long ws_tiles = g.splitk_ws ? g.splitk_ws_doubles / ((long)BM * BN) : 0;
    long main_tiles = tiles, tail_tiles = 0;
    int main_split = 1, tail_split = 1;
    // the LDS-DMA kernel (128 x 128, 16-byte loads, 32-bit lane offsets, K range per block >= dma_min_k) plans its own
    // launch: whole tiles + a cut tail in ONE grid (plan_dma)
    const bool dma_ok = BM == 128 && BN == 128 && vec == 2 && g.K >= dma_min_k() &&
                        (a_kcontig ? 128 : 16) * k.a_ld * 8 + 4096 < (1L << 32) &&
                        (b_kcontig ? 128 : 16) * k.b_ld * 8 + 4096 < (1L << 32);
    DmaPlan plan{tiles, 0, 1};
    if (dma_ok) plan = plan_dma(tiles, ktiles, ws_tiles);
    else if (ktiles >= 16) {
        // The last, partially filled round of tiles (all of them for a small output) is split s ways along K:
        // ceil(rem s / slots) rounds of 1/s tile-time each.  Few remainder tiles may be split finer (huge K, tiny output).
        const long rem = tiles % slots;
        if (rem > 0) {
            long best = 1;
            double best_cost = 1.0;
            long smax = std::min<long>(512, std::max<long>(8, 2048 / rem));
            const long min_kt = (BM == 128 && BN == 128) ? 16 : 8;       // per split: >= 256 (128x128) / 128 (smaller tiles) deep
            smax = std::min<long>(smax, ktiles / min_kt);
            smax = std::min<long>(smax, ws_tiles / rem);
            for (long sp = 2; sp <= smax; ++sp) {
                const double cost = (double)((rem * sp + slots - 1) / slots) / (double)sp + 1e-5 * sp;
                if (cost < best_cost - 1e-9) { best_cost = cost; best = sp; }
            }
            if (best >= 2 && best_cost < 0.8) {        // worth it below 0.8 of a tile-time
                tail_tiles = rem;
                main_tiles = tiles - rem;
                tail_split = (int)best;
            }
        }
    }

    std::pair<hipEvent_t, hipEvent_t> ev;
    if (g_prof.on) {
        if (!g_prof.pool.empty()) {
            ev = g_prof.pool.back();
            g_prof.pool.pop_back();
        } else {
            HIP_CHECK(hipEventCreate(&ev.first));
            HIP_CHECK(hipEventCreate(&ev.second));
        }
        HIP_CHECK(hipEventRecord(ev.first, st));
    }
    bool used_dma = false;
    int n_kernels = 0;
    g_stream64 = stream && BM <= 64 && BN <= 64;
    auto launch = [&](long tile_begin, long ntiles, int nsplit) {
        const long kt_per = (ktiles + nsplit - 1) / nsplit;
        k.kchunk = (int)std::max<long>(kt_per * BK, BK);
        k.nsplit = (int)std::max<long>(1, (ktiles + kt_per - 1) / std::max<long>(kt_per, 1));
        k.tile_begin = tile_begin;
        k.ws = k.nsplit > 1 ? g.splitk_ws : nullptr;
        const long nblocks = ntiles * k.nsplit;
        ++n_kernels;
        if (BM == 128 && BN == 128) dispatch_layout<128, 128>(k, a_kcontig, b_kcontig, vec, nblocks, st);
        else if (BM == 128 && BN == 64) dispatch_layout<128, 64>(k, a_kcontig, b_kcontig, vec, nblocks, st);
        else if (BM == 64 && BN == 128) dispatch_layout<64, 128>(k, a_kcontig, b_kcontig, vec, nblocks, st);
        else if (BM == 64 && BN == 32) dispatch_layout<64, 32>(k, a_kcontig, b_kcontig, vec, nblocks, st);
        else if (BM == 32 && BN == 64) dispatch_layout<32, 64>(k, a_kcontig, b_kcontig, vec, nblocks, st);
        else dispatch_layout<64, 64>(k, a_kcontig, b_kcontig, vec, nblocks, st);
        if (k.nsplit > 1) {
            PYMES_LAUNCH(splitk_reduce_kernel, dim3((unsigned)ntiles, (unsigned)(BM * BN / 256)), dim3(256), 0, st, k, BM, BN);
            HIP_CHECK(hipGetLastError());
        }
        return k.nsplit;
    };
    int nsplit = 1;
    if (dma_ok) {
        // one grid: plan.whole whole tiles, then plan.tail tiles cut plan.s ways (ks-major); one reduction over the tail tiles
        const long kt_per = (ktiles + plan.s - 1) / plan.s;
        k.kchunk = (int)std::max<long>(kt_per * BK, BK);
        k.nsplit = (int)std::max<long>(1, (ktiles + kt_per - 1) / std::max<long>(kt_per, 1));
        if (k.nsplit == 1) { plan.whole = tiles; plan.tail = 0; }
        k.tile_begin = 0;
        k.mixed = 1;
        k.whole = plan.whole;
        k.tail = plan.tail;
        k.ws = plan.tail > 0 ? g.splitk_ws : nullptr;
        const long nblocks = plan.whole + plan.tail * k.nsplit;
        if (nblocks > 0x7fffffffL) throw std::runtime_error("gemm: grid too large");
        if (a_kcontig && b_kcontig) launch_gemm_glds<true, true>(k, nblocks, st);
        else if (a_kcontig) launch_gemm_glds<true, false>(k, nblocks, st);
        else if (b_kcontig) launch_gemm_glds<false, true>(k, nblocks, st);
        else launch_gemm_glds<false, false>(k, nblocks, st);
        used_dma = true;
        n_kernels = 1;
        if (plan.tail > 0) {
            GemmK r = k;                   // the reduction sees the tail tiles as a launch of its own: ws[i][ks], i from 0
            r.tile_begin = plan.whole;
            r.mixed = 0;
            PYMES_LAUNCH(splitk_reduce_kernel, dim3((unsigned)plan.tail, 64u), dim3(256), 0, st, r, 128, 128);
            HIP_CHECK(hipGetLastError());
            nsplit = -k.nsplit;            // logged as a negative split
        }
    } else {
        nsplit = main_tiles > 0 ? launch(0, main_tiles, main_split) : 1;
        if (tail_tiles > 0) nsplit = -launch(main_tiles, tail_tiles, tail_split);   // logged as a negative split
    }
    if (g_prof.on) {
        HIP_CHECK(hipEventRecord(ev.second, st));
        g_prof.ev.push_back(ev);
        const double fl = 2.0 * (double)g.M * (double)g.N * (double)g.K * (double)nbatch;
        g_prof.flops += fl;
        g_prof.fl.push_back(fl);
        g_prof.klass.push_back(used_dma ? 1 : 0);
        g_prof.nk.push_back(n_kernels);
        char buf[256];
        int len = snprintf(buf, sizeof buf, "M=%ld N=%ld K=%ld batch=%ld tile=%dx%d%s akc=%d bkc=%d vec=%d dma=%d split=%d flops=%.4e",
                           (long)g.M, (long)g.N, (long)g.K, (long)nbatch, BM, BN, g_stream64 ? "s" : "", (int)a_kcontig,
                           (int)b_kcontig, vec, (int)used_dma, nsplit, fl);
        if (used_dma && len > 0 && len < (int)sizeof buf)
            snprintf(buf + len, sizeof buf - len, " plan=%ld+%ld/%d", plan.whole, plan.tail, plan.tail ? k.nsplit : 1);
        g_prof.what.push_back(buf);
    }
}

void gemv_batch_begin() {
    g_gemv_batch.active = true;
    g_gemv_batch.tab.n = 0;
    g_gemv_batch.ws_used = 0;
}
void gemv_batch_end() {
    gemv_batch_flush();
    g_gemv_batch.active = false;
}

void permute(const Permute& p, stream_t s) {
    gemv_batch_flush();
    gemm_group_flush();
    hipStream_t st = (hipStream_t)s;
    // canonicalise: drop extent-1 dims, sort by out-stride (descending), merge adjacent dims
    struct D { long n, si, so; };
    std::vector<D> d;
    long total = 1;
    for (int i = 0; i < p.rank; ++i) {
        total *= p.dim[i];
        if (p.dim[i] != 1) d.push_back({p.dim[i], p.s_in[i], p.s_out[i]});
    }
    if (total == 0) return;
    std::stable_sort(d.begin(), d.end(), [](const D& a, const D& b) { return a.so > b.so; });
    std::vector<D> m;
    for (auto& x : d) {
        if (!m.empty() && m.back().si == x.n * x.si && m.back().so == x.n * x.so) {
            m.back().n *= x.n;
            m.back().si = x.si;
            m.back().so = x.so;
        } else {
            m.push_back(x);
        }
    }
    if (m.empty()) m.push_back({1, 1, 1});
    PermK k;
    k.alpha = p.alpha; k.beta = p.beta; k.in = p.in; k.out = p.out;
    // as a task of the open phase when it is small (16 bytes per element, 24 with an accumulating output)
    const double cost = (p.beta != 0.0 ? 24.0 : 16.0) * (double)total / 4.0e6;
    const bool phase = phase_open(st) && phase_small(cost);
    auto record = [&](unsigned short kind, long nblk, int lds, const PermTaskK& t) {
        PhaseRange ri{reinterpret_cast<uintptr_t>(p.in), reinterpret_cast<uintptr_t>(p.in) + 8}, ro{reinterpret_cast<uintptr_t>(p.out), reinterpret_cast<uintptr_t>(p.out) + 8};
        for (const auto& x : m) {
            const long si = (x.n - 1) * x.si * 8, so = (x.n - 1) * x.so * 8;
            if (si < 0) ri.lo += si; else ri.hi += si;
            if (so < 0) ro.lo += so; else ro.hi += so;
        }
        PhaseRec& r = phase_push(kind, 0, nblk, lds, cost, t);
        phase_reads(r, {ri});
        phase_writes(r, {ro});
        // (accumulation fusion: the output is one contiguous array starting at p.out, overwritten or accumulated into)
        if (ro.hi - ro.lo == 8 * (uintptr_t)total && ro.lo == reinterpret_cast<uintptr_t>(p.out) && (p.beta == 0.0 || p.beta == 1.0)) {
            r.box = ro;
            r.acc = p.beta == 1.0 ? 1 : 0;
        }
    };
    // tiled path: out unit-stride on the last dim, in unit-stride on another dim
    int q = -1;
    const int r = (int)m.size();
    if (r >= 2 && m[r - 1].so == 1 && m[r - 1].si != 1) {
        for (int i = 0; i < r - 1; ++i)
            if (m[i].si == 1) q = i;
    }
    if (q >= 0 && m[q].n >= 8 && m[r - 1].n >= 8) {
        D dq = m[q];
        m.erase(m.begin() + q);
        m.insert(m.end() - 1, dq);          // [rest..., Q, L]
        k.rank = r;
        for (int i = 0; i < r; ++i) { k.dim[i] = m[i].n; k.s_in[i] = m[i].si; k.s_out[i] = m[i].so; }
        for (int i = r; i < 6; ++i) { k.dim[i] = 1; k.s_in[i] = 0; k.s_out[i] = 0; }
        const int tq = (int)((m[r - 2].n + 31) / 32), tl = (int)((m[r - 1].n + 31) / 32);
        long rest = 1;
        for (int i = 0; i < r - 2; ++i) rest *= m[i].n;
        const long nblk = rest * tq * tl;
        if (nblk > 0x7fffffffL) throw std::runtime_error("permute: grid too large");
        if (phase && nblk < 0x3fffffffL) { record(PK_PERM_TILED, nblk, kPermTileDoubles * (int)sizeof(double), PermTaskK{k, total, tq, tl}); return; }
        PYMES_LAUNCH(permute_tiled_kernel, dim3((unsigned)nblk), dim3(256), 0, st, k, tq, tl);
    } else {
        k.rank = r;
        for (int i = 0; i < r; ++i) { k.dim[i] = m[i].n; k.s_in[i] = m[i].si; k.s_out[i] = m[i].so; }
        for (int i = r; i < 6; ++i) { k.dim[i] = 1; k.s_in[i] = 0; k.s_out[i] = 0; }
        if (phase) { record(PK_PERM_DIRECT, grid_for(total, 256, 256 * 32), 0, PermTaskK{k, total, 0, 0}); return; }
        PYMES_LAUNCH(permute_direct_kernel, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, st, k, total);
    }
    HIP_CHECK(hipGetLastError());
}

void mp2_amplitudes(double* t, const double* w, const double* eo, const double* ev, double shift, int no, int nv,
                    stream_t s) {
    const long total = (long)nv * nv * no * no;
    if (!total) return;
    PYMES_LAUNCH(mp2_amplitudes_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, t, w, eo, ev,
                       shift, no, nv, total);
    HIP_CHECK(hipGetLastError());
}

void cc_update_to(double* t_out, double* dt, const double* t_in, const double* r, const double* eo, const double* ev,
                  double shift, double delta, int no, int nv, int rank, stream_t s) {
    const long total = rank == 4 ? (long)nv * nv * no * no : (long)nv * no;
    if (!total) return;
    const CcUpdateK k{t_out, dt, t_in, r, eo, ev, shift, delta, no, nv, rank, total};
    const double cost = 32.0 * (double)total / 4.0e6;
    if (phase_open((hipStream_t)s) && phase_small(cost)) {
        PhaseRec& t = phase_push(PK_CC_UPDATE, 0, grid_for(total), 0, cost, k);
        phase_reads(t, {prange(t_in, total), prange(r, total), prange(eo, no), prange(ev, nv)});
        phase_writes(t, {prange(t_out, total), prange(dt, total)});
        return;
    }
    PYMES_LAUNCH(cc_update_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, k);
    HIP_CHECK(hipGetLastError());
}
void cc_update(double* t, double* dt, const double* r, const double* eo, const double* ev, double shift,
               double delta, int no, int nv, int rank, stream_t s) {
    cc_update_to(t, dt, t, r, eo, ev, shift, delta, no, nv, rank, s);
}

// stage 1 of a batch of dot products: 16-byte loads when every operand allows them; returns the number of chunks per pair
static int launch_dots_stage1(int npairs, const double* const* x, const double* const* y, const int64_t* n, double* ws,
                              hipStream_t st) {
    DotPtrs p;
    long nmax = 0;
    bool vec = true;
    for (int i = 0; i < npairs; ++i) {
        p.x[i] = x[i]; p.y[i] = y[i]; p.n[i] = (long)n[i];
        nmax = std::max(nmax, p.n[i]);
        vec = vec && ((reinterpret_cast<uintptr_t>(x[i]) | reinterpret_cast<uintptr_t>(y[i])) & 15) == 0;
    }
    for (int i = npairs; i < 16; ++i) { p.x[i] = nullptr; p.y[i] = nullptr; p.n[i] = 0; }
    const long per = vec ? 512 : 256;
    const int nb = (int)std::max<long>(1, std::min<long>(kDotBlocks, (nmax + per - 1) / per));
    const DotsK k{p, ws, nb, npairs};
    double bytes = 0.0;
    for (int i = 0; i < npairs; ++i) bytes += 16.0 * (double)p.n[i];
    if (phase_open(st) && phase_small(bytes / 4.0e6)) {
        PhaseRec& t = phase_push(PK_DOTS1, vec ? 1 : 0, (long)nb * npairs, 0, bytes / 4.0e6, k);
        // (the operands: one bounding range per distinct pointer would overflow the list — the pairs of a call share operands)
        uintptr_t lo = ~(uintptr_t)0, hi = 0;
        for (int i = 0; i < npairs; ++i)
            for (const double* q : {p.x[i], p.y[i]}) {
                lo = std::min(lo, reinterpret_cast<uintptr_t>(q));
                hi = std::max(hi, reinterpret_cast<uintptr_t>(q) + 8 * (uintptr_t)p.n[i]);
            }
        phase_reads(t, {PhaseRange{lo, hi}});
        phase_writes(t, {prange(ws, 16L * kDotBlocks)});
        return nb;
    }
    if (vec) PYMES_LAUNCH(dots_stage1_kernel<2>, dim3(nb * npairs), dim3(256), 0, st, k);
    else PYMES_LAUNCH(dots_stage1_kernel<1>, dim3(nb * npairs), dim3(256), 0, st, k);
    HIP_CHECK(hipGetLastError());
    return nb;
}
// stage 2 (partial sums of `npairs` reductions -> out_dev[npairs]) as a task of the open phase or a launch
static void launch_dots_stage2(int npairs, double* ws, int nb, double* out_dev, hipStream_t st) {
    if (phase_open(st)) {
        PhaseRec& t = phase_push(PK_DOTS2, 0, npairs, 0, 1.0, Dots2K{ws, out_dev, nb});
        phase_reads(t, {prange(ws, 16L * kDotBlocks)});
        phase_writes(t, {prange(out_dev, npairs)});
        return;
    }
    PYMES_LAUNCH(dots_stage2_kernel, dim3(npairs), dim3(256), 0, st, ws, nb, out_dev);
    HIP_CHECK(hipGetLastError());
}

// the last stage of a reduction the host waits for (dots_final_kernel), as a task of the open phase or a launch
static void launch_dots_final(DotsFinalK k, double* ws, hipStream_t st) {
    {
        const int dv = current_device();
        std::lock_guard<std::mutex> lock(g_arrivals_mu);
        k.arrivals = g_arrivals[dv] + 32 * (g_arrivals_next[dv]++ % kArrivalSlots);
        k.pad_ = 0;
    }
    if (phase_open(st)) {
        PhaseRec& t = phase_push(PK_DOTS_FINAL, 0, k.npairs, 0, 1.0, k);
        phase_reads(t, {prange(ws, 16L * kDotBlocks)});
        // (the pinned destination is nobody else's: no range to declare; the task is ordered behind stage 1 through ws)
        return;
    }
    PYMES_LAUNCH(dots_final_kernel, dim3(k.npairs), dim3(256), 0, st, k);
    HIP_CHECK(hipGetLastError());
}

void dots(int npairs, const double* const* x, const double* const* y, const int64_t* n, double* out_host, stream_t s) {
    if (npairs <= 0) return;
    if (npairs > 16) throw std::runtime_error("dots: at most 16 pairs per call");
    hipStream_t st = (hipStream_t)s;
    const int dv = current_device();
    ensure_dot_ws(dv);
    const int nb = launch_dots_stage1(npairs, x, y, n, g_dot_ws[dv], st);
    // results and a sequence word straight into pinned memory, polled for (no copy, no stream synchronisation)
    void* hd = nullptr;
    HIP_CHECK(hipHostGetDevicePointer(&hd, g_dot_host[dv], 0));
    const long seq = ++g_dot_seq[dv];
    launch_dots_final(DotsFinalK{g_dot_ws[dv], static_cast<double*>(hd), reinterpret_cast<long*>(static_cast<double*>(hd) + 16), nullptr, 0u, seq, nb,
                                 npairs},
                      g_dot_ws[dv], st);
    phase_flush();
    if (!poll_flag(reinterpret_cast<const long*>(g_dot_host[dv] + 16), seq, st)) throw std::runtime_error("dots: the device never delivered the result");
    for (int i = 0; i < npairs; ++i) out_host[i] = g_dot_host[dv][i];
}

void diis_step(double* state, int npairs, const double* const* x, const double* const* y, const int64_t* n, int ntypes, int m,
               int was_full, stream_t s) {
    if (npairs != ntypes * m || npairs < 1 || npairs > 16 || m + 1 > diis_small::kMaxOrder)
        throw std::runtime_error("diis_step: bad sizes");
    hipStream_t st = (hipStream_t)s;
    const int dv = current_device();
    ensure_dot_ws(dv);
    const int nb = launch_dots_stage1(npairs, x, y, n, g_dot_ws[dv], st);
    double* out_dev = g_dot_ws[dv] + 16 * kDotBlocks;
    launch_dots_stage2(npairs, g_dot_ws[dv], nb, out_dev, st);
    PYMES_LAUNCH(diis_step_kernel, dim3(1), dim3(64), 0, st, state, out_dev, ntypes, m, was_full);
    HIP_CHECK(hipGetLastError());
}

void lincomb_dev(double* out, int nx, const double* const* x, const double* coeff_dev, int64_t n, stream_t s) {
    if (nx < 0 || nx > 8) throw std::runtime_error("lincomb_dev: at most 8 terms");
    if (n <= 0) return;
    LinPtrsDev p;
    for (int i = 0; i < 8; ++i) p.x[i] = i < nx ? x[i] : nullptr;
    PYMES_LAUNCH(lincomb_dev_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, out, p, coeff_dev, nx, (long)n);
    HIP_CHECK(hipGetLastError());
}

int energy_norms_start(const double* f, const double* t1, const double* t2, const double* Edir, const double* Eex,
                       const double* dt2, int no, int nv, stream_t s) {
    gemm_group_flush();
    hipStream_t st = (hipStream_t)s;
    const int dv = current_device();
    ensure_dot_ws(dv);
    const long total = (long)nv * nv * no * no;
    const int nb = (int)std::max<long>(1, std::min<long>(kDotBlocks, (total + 255) / 256));
    const bool vec = (no % 2 == 0) && !((reinterpret_cast<uintptr_t>(t2) | reinterpret_cast<uintptr_t>(Edir) |
                                          reinterpret_cast<uintptr_t>(Eex) | reinterpret_cast<uintptr_t>(dt2)) & 15);
    const EnergyK k{f, t1, t2, Edir, Eex, dt2, g_dot_ws[dv], total, no, nv};
    const double cost = (dt2 ? 32.0 : 24.0) * (double)total / 4.0e6;
    if (phase_open(st) && phase_small(cost)) {
        PhaseRec& t = phase_push(PK_ENERGY, vec ? 1 : 0, nb, 0, cost, k);
        const long n = no + nv;
        phase_reads(t, {prange(t2, total), prange(Edir, total), prange(Eex, total), prange(dt2, dt2 ? total : 0),
                        prange(t1, t1 ? (long)no * nv : 0), prange(f, (t1 && f) ? n * n : 0)});
        phase_writes(t, {prange(g_dot_ws[dv], 16L * kDotBlocks)});
    } else {
        if (vec) PYMES_LAUNCH(energy_norms_kernel<2>, dim3(nb), dim3(256), 0, st, k);
        else PYMES_LAUNCH(energy_norms_kernel<1>, dim3(nb), dim3(256), 0, st, k);
        HIP_CHECK(hipGetLastError());
    }
    // the six sums straight into a slot of the pinned read-back ring, with the slot's sequence word behind them
    double* out_pin = nullptr;
    long* flag = nullptr;
    long seq = 0;
    const int ticket = readback_reserve_flagged(dv, &out_pin, &flag, &seq);
    launch_dots_final(DotsFinalK{g_dot_ws[dv], out_pin, flag, nullptr, 0u, seq, nb, 6}, g_dot_ws[dv], st);
    phase_flush();
    return ticket;
}
void energy_norms(const double* f, const double* t1, const double* t2, const double* Edir, const double* Eex,
                  const double* dt2, int no, int nv, double out_host[6], stream_t s) {
    readback_wait_impl(energy_norms_start(f, t1, t2, Edir, Eex, dt2, no, nv, s), out_host, 6);
}
int readback_start(const double* dev_ptr, int n, stream_t s) {
    gemm_group_flush();
    return readback_start_impl(dev_ptr, n, (hipStream_t)s);
}
void readback_wait(int slot, double* out_host, int n) { readback_wait_impl(slot, out_host, n); }

void exchange_asymmetry(const double* A, const double* B, const int64_t d[4], double out_host[2], stream_t s) {
    hipStream_t st = (hipStream_t)s;
    const int dv = current_device();
    ensure_dot_ws(dv);
    const long total = (long)d[0] * d[1] * d[2] * d[3];
    out_host[0] = out_host[1] = 0.0;
    if (!total) return;
    unsigned long long* out_dev = reinterpret_cast<unsigned long long*>(g_dot_ws[dv] + 16 * kDotBlocks);
    HIP_CHECK(hipMemsetAsync(out_dev, 0, 2 * sizeof(unsigned long long), st));
    const long tr = (d[2] + 63) / 64, ts = (d[3] + 63) / 64, ntiles = d[0] * d[1] * tr * ts;
    const int self = (A == B && d[0] == d[1]) ? 1 : 0;
    const unsigned grid = (unsigned)std::min<long>(ntiles, 256L * 8);
    const bool vec = even(d[2]) && even(d[3]) && aligned16(A) && aligned16(B);
    if (vec)
        PYMES_LAUNCH(exchange_asym_kernel<true>, dim3(grid), dim3(256), 0, st, A, B, (long)d[0], (long)d[1], (long)d[2],
                           (long)d[3], (int)tr, (int)ts, ntiles, self, out_dev);
    else
        PYMES_LAUNCH(exchange_asym_kernel<false>, dim3(grid), dim3(256), 0, st, A, B, (long)d[0], (long)d[1], (long)d[2],
                           (long)d[3], (int)tr, (int)ts, ntiles, self, out_dev);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(g_dot_host[dv], out_dev, sizeof(double) * 2, hipMemcpyDeviceToHost, st));
    wait_idle(st);
    out_host[0] = g_dot_host[dv][0];
    out_host[1] = g_dot_host[dv][1];
}

void lincomb(double* out, int nx, const double* const* x, const double* c, int64_t n, stream_t s) {
    if (nx < 0 || nx > 8) throw std::runtime_error("lincomb: at most 8 terms");
    if (n <= 0) return;
    LinPtrs p;
    for (int i = 0; i < 8; ++i) { p.x[i] = i < nx ? x[i] : nullptr; p.c[i] = i < nx ? c[i] : 0.0; }
    const LinK k{out, p, (long)n, nx};
    const double cost = 8.0 * (double)(nx + 1) * (double)n / 4.0e6;
    if (phase_open((hipStream_t)s) && phase_small(cost)) {
        PhaseRec& t = phase_push(PK_LINCOMB, 0, grid_for(n), 0, cost, k);
        for (int i = 0; i < nx; ++i) phase_reads(t, {prange(x[i], n)});
        phase_writes(t, {prange(out, n)});
        return;
    }
    PYMES_LAUNCH(lincomb_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, k);
    HIP_CHECK(hipGetLastError());
}

namespace {
template <int MM, int NN>
void gram_tile(const double* const* x, int mm, const double* const* y, int nn, long len, bool vec, int nblk, double* ws,
               double* tile_dev, hipStream_t st) {
    GramPtrs<MM, NN> p;
    for (int i = 0; i < MM; ++i) p.x[i] = x[i < mm ? i : 0];
    for (int j = 0; j < NN; ++j) p.y[j] = y[j < nn ? j : 0];
    if (vec) PYMES_LAUNCH((gram_stage1_kernel<MM, NN, 2>), dim3(nblk), dim3(256), 0, st, p, len, ws);
    else PYMES_LAUNCH((gram_stage1_kernel<MM, NN, 1>), dim3(nblk), dim3(256), 0, st, p, len, ws);
    HIP_CHECK(hipGetLastError());
    PYMES_LAUNCH(gram_stage2_kernel, dim3(MM * NN), dim3(256), 0, st, ws, nblk, MM * NN, tile_dev);
    HIP_CHECK(hipGetLastError());
}
template <int MM>
void gram_tile_n(const double* const* x, int mm, const double* const* y, int nn, long len, bool vec, int nblk, double* ws,
                 double* tile_dev, hipStream_t st, int& NNout) {
    if (nn <= 1) { NNout = 1; gram_tile<MM, 1>(x, mm, y, nn, len, vec, nblk, ws, tile_dev, st); }
    else if (nn <= 2) { NNout = 2; gram_tile<MM, 2>(x, mm, y, nn, len, vec, nblk, ws, tile_dev, st); }
    else if (nn <= 3) { NNout = 3; gram_tile<MM, 3>(x, mm, y, nn, len, vec, nblk, ws, tile_dev, st); }
    else { NNout = 4; gram_tile<MM, 4>(x, mm, y, nn, len, vec, nblk, ws, tile_dev, st); }
}
template <int MM, int NN>
void lincomb_multi_tile(const double* const* x, int mm, const double* c, int ldc, const double* beta, double* const* y, int nn,
                        long len, bool vec, hipStream_t st) {
    LinMulti<MM, NN> p;
    for (int i = 0; i < MM; ++i) {
        p.x[i] = x[i < mm ? i : 0];
        for (int j = 0; j < NN; ++j) p.c[i][j] = (i < mm && j < nn) ? c[(long)i * ldc + j] : 0.0;
    }
    for (int j = 0; j < NN; ++j) { p.y[j] = y[j < nn ? j : 0]; p.beta[j] = j < nn ? beta[j] : 0.0; }
    const long nvec = vec ? len / 2 : len;
    const int nblk = (int)std::max<long>(1, std::min<long>(8192, (nvec + 255) / 256));
    if (vec) PYMES_LAUNCH((lincomb_multi_kernel<MM, NN, 2>), dim3(nblk), dim3(256), 0, st, p, nn, len);
    else PYMES_LAUNCH((lincomb_multi_kernel<MM, NN, 1>), dim3(nblk), dim3(256), 0, st, p, nn, len);
    HIP_CHECK(hipGetLastError());
}
template <int MM>
void lincomb_multi_tile_n(const double* const* x, int mm, const double* c, int ldc, const double* beta, double* const* y, int nn,
                          long len, bool vec, hipStream_t st) {
    if (nn <= 1) lincomb_multi_tile<MM, 1>(x, mm, c, ldc, beta, y, nn, len, vec, st);
    else if (nn <= 2) lincomb_multi_tile<MM, 2>(x, mm, c, ldc, beta, y, nn, len, vec, st);
    else if (nn <= 3) lincomb_multi_tile<MM, 3>(x, mm, c, ldc, beta, y, nn, len, vec, st);
    else lincomb_multi_tile<MM, 4>(x, mm, c, ldc, beta, y, nn, len, vec, st);
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
}  // namespace

void gram(int m, int n, const double* const* x, const double* const* y, int64_t len, double* out_host, stream_t s) {
    if (m <= 0 || n <= 0) return;
    if (m > 64 || n > 64) throw std::runtime_error("gram: at most 64 x 64 vectors per call");
    hipStream_t st = (hipStream_t)s;
    if (len <= 0) {
        for (int i = 0; i < m * n; ++i) out_host[i] = 0.0;
        return;
    }
    const int dv = current_device();
    if (!g_gram_ws[dv]) {
        HIP_CHECK(hipMalloc((void**)&g_gram_ws[dv], sizeof(double) * 64 * (kGramBlocks + kGramTiles)));
        HIP_CHECK(hipHostMalloc((void**)&g_gram_host[dv], sizeof(double) * 64 * kGramTiles));
    }
    // the longer list takes the 16-wide side of the register tile: fewer passes over the vectors
    const bool swap = n > m;
    const double* const* X = swap ? y : x;
    const double* const* Y = swap ? x : y;
    const int mx = swap ? n : m, ny = swap ? m : n;
    bool vec = true;
    for (int i = 0; i < mx; ++i) vec = vec && aligned16(X[i]);
    for (int j = 0; j < ny; ++j) vec = vec && aligned16(Y[j]);
    const long nvec = vec ? len / 2 : len;
    const int nblk = (int)std::max<long>(1, std::min<long>(kGramBlocks, (nvec + 255) / 256));
    double* ws = g_gram_ws[dv];
    double* tiles = ws + 64 * kGramBlocks;
    struct Tile { int i0, mm, j0, nn, NN; };
    std::vector<Tile> done;
    auto flush = [&]() {
        if (done.empty()) return;
        HIP_CHECK(hipMemcpyAsync(g_gram_host[dv], tiles, sizeof(double) * 64 * done.size(), hipMemcpyDeviceToHost, st));
        wait_idle(st);
        for (size_t t = 0; t < done.size(); ++t)
            for (int i = 0; i < done[t].mm; ++i)
                for (int j = 0; j < done[t].nn; ++j) {
                    const double v = g_gram_host[dv][64 * t + i * done[t].NN + j];
                    const int gi = done[t].i0 + i, gj = done[t].j0 + j;
                    if (swap) out_host[(long)gj * n + gi] = v;
                    else out_host[(long)gi * n + gj] = v;
                }
        done.clear();
    };
    for (int i0 = 0; i0 < mx; i0 += 16)
        for (int j0 = 0; j0 < ny; j0 += 4) {
            const int mm = std::min(16, mx - i0), nn = std::min(4, ny - j0);
            int NN = 4;
            double* tile = tiles + 64 * done.size();
            if (mm <= 4) gram_tile_n<4>(X + i0, mm, Y + j0, nn, (long)len, vec, nblk, ws, tile, st, NN);
            else if (mm <= 8) gram_tile_n<8>(X + i0, mm, Y + j0, nn, (long)len, vec, nblk, ws, tile, st, NN);
            else if (mm <= 12) gram_tile_n<12>(X + i0, mm, Y + j0, nn, (long)len, vec, nblk, ws, tile, st, NN);    // (the Davidson
            else gram_tile_n<16>(X + i0, mm, Y + j0, nn, (long)len, vec, nblk, ws, tile, st, NN);    //  subspace: 12 x 3, no padded slots)
            done.push_back({i0, mm, j0, nn, NN});
            if ((int)done.size() == kGramTiles) flush();
        }
    flush();
}

void lincomb_multi(int m, int n, const double* const* x, const double* c, const double* beta, double* const* y, int64_t len,
                   stream_t s) {
    if (n <= 0 || len <= 0) return;
    if (m < 0 || m > 64 || n > 64) throw std::runtime_error("lincomb_multi: at most 64 inputs and 64 outputs per call");
    if (m > 16 || n > 4)
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < m; ++i)
                if (y[j] == x[i]) throw std::runtime_error("lincomb_multi: an output may alias an input only for m <= 16, n <= 4");
    hipStream_t st = (hipStream_t)s;
    bool vec = true;
    for (int i = 0; i < m; ++i) vec = vec && aligned16(x[i]);
    for (int j = 0; j < n; ++j) vec = vec && aligned16(y[j]);
    double one[4] = {1.0, 1.0, 1.0, 1.0}, b0[4];
    for (int j0 = 0; j0 < n; j0 += 4) {
        const int nn = std::min(4, n - j0);
        for (int j = 0; j < nn; ++j) b0[j] = beta ? beta[j0 + j] : 0.0;
        if (m == 0) {       // y_j = beta_j y_j
            const double* none[1] = {y[j0]};
            const double zero[4] = {0.0, 0.0, 0.0, 0.0};
            lincomb_multi_tile_n<4>(none, 1, zero, 4, b0, y + j0, nn, (long)len, vec, st);
            continue;
        }
        for (int i0 = 0; i0 < m; i0 += 16) {
            const int mm = std::min(16, m - i0);
            const double* bb = i0 == 0 ? b0 : one;
            const double* cc = c + (long)i0 * n + j0;
            if (mm <= 4) lincomb_multi_tile_n<4>(x + i0, mm, cc, n, bb, y + j0, nn, (long)len, vec, st);
            else if (mm <= 8) lincomb_multi_tile_n<8>(x + i0, mm, cc, n, bb, y + j0, nn, (long)len, vec, st);
            else if (mm <= 12) lincomb_multi_tile_n<12>(x + i0, mm, cc, n, bb, y + j0, nn, (long)len, vec, st);
            else lincomb_multi_tile_n<16>(x + i0, mm, cc, n, bb, y + j0, nn, (long)len, vec, st);
        }
    }
}

void cmul(const double* mr, const double* mi, const double* xr, const double* xi, double* yr, double* yi, int64_t n, stream_t s) {
    if (n <= 0) return;
    PYMES_LAUNCH(cmul_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, mr, mi, xr, xi, yr, yi, (long)n);
    HIP_CHECK(hipGetLastError());
}

void exchange_split(const double* u, double* us, double* w, double* dg, int no, int nv, stream_t s) {
    const long total = (long)nv * nv * no * no;
    if (!total) return;
    PYMES_LAUNCH(exchange_split_kernel, dim3(grid_for(total, 256, 256 * 64)), dim3(256), 0, (hipStream_t)s, u, us, w, dg, no, nv,
                       total);
    HIP_CHECK(hipGetLastError());
}
void sgn_ij_add(double* D, const double* R, int no, int nv, stream_t s) {
    const long total = (long)nv * nv * no * no;
    if (!total) return;
    PYMES_LAUNCH(sgn_ij_add_kernel, dim3(grid_for(total, 256, 256 * 64)), dim3(256), 0, (hipStream_t)s, D, R, no, total);
    HIP_CHECK(hipGetLastError());
}
int64_t eom_diag_ws_doubles(int no, int nv) { return EomDiagWs(no, nv).total; }
void eom_diagonals(const double* V, const double* T, const double* dai, const double* iaai, const double* iaia, const double* ijij,
                   const double* abab, double* d1, double* d2, int no, int nv, double* ws, stream_t s) {
    hipStream_t st = (hipStream_t)s;
    const EomDiagWs w(no, nv);
    const long total = (long)nv * nv * no * no;
    if (w.total > 0x7fffffffL) throw std::runtime_error("eom_diagonals: problem too large for one grid");
    PYMES_LAUNCH(eom_diag_sums_kernel, dim3((unsigned)w.total), dim3(256), 0, st, V, T, ws, no, nv);
    HIP_CHECK(hipGetLastError());
    PYMES_LAUNCH(eom_diag_singles_kernel, dim3((unsigned)(((long)nv * no + 255) / 256)), dim3(256), 0, st, ws, dai, iaai, iaia,
                       d1, no, nv);
    HIP_CHECK(hipGetLastError());
    PYMES_LAUNCH(eom_diag_doubles_kernel, dim3(grid_for(total, 256, 256 * 64)), dim3(256), 0, st, V, T, ws, dai, iaai, iaia, ijij,
                       abab, d2, no, nv, total);
    HIP_CHECK(hipGetLastError());
}
void cshift_inv(const double* d, double zr, double zi, double hr, double hi, double shift, double* mr, double* mi, int64_t n,
                stream_t s) {
    if (n <= 0) return;
    PYMES_LAUNCH(cshift_inv_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, d, zr, zi, hr, hi, shift, mr, mi, (long)n);
    HIP_CHECK(hipGetLastError());
}

void tau_build(double* tau, const double* t2, const double* t1, int no, int nv, stream_t s) {
    const long total = (long)nv * nv * no * no;
    if (!total) return;
    const TauK k{tau, t2, t1, total, no, nv};
    const double cost = 16.0 * (double)total / 4.0e6;
    if (phase_open((hipStream_t)s) && phase_small(cost)) {
        PhaseRec& t = phase_push(PK_TAU, 0, grid_for(total), 0, cost, k);
        phase_reads(t, {prange(t2, total), prange(t1, (long)no * nv)});
        phase_writes(t, {prange(tau, total)});
        return;
    }
    PYMES_LAUNCH(tau_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, k);
    HIP_CHECK(hipGetLastError());
}

void ladder_pack_V(const double* V, double* Vp, double* Vm, int nr, int nc, int64_t rp0, int64_t rp1, stream_t s, int64_t ldvp,
                   int64_t ldvm) {
    if (rp1 <= rp0) return;
    const long npp = ldvp ? (long)ldvp : (long)nc * (nc + 1) / 2, npm = ldvm ? (long)ldvm : (long)nc * (nc - 1) / 2;
    const int nt = (nc + 31) / 32;
    const long ntp = (long)nt * (nt + 1) / 2;
    const long nblk = (rp1 - rp0) * ntp;
    if (nblk > 0x7fffffffL) throw std::runtime_error("ladder_pack_V: grid too large");
    PYMES_LAUNCH(ladder_pack_V_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)s, V, Vp, Vm, nr, nc,
                       (long)rp0, nt, ntp, npp, npm);
    HIP_CHECK(hipGetLastError());
}

// nocc within the instantiated MFMA step counts, and the byte offsets inside a tile (16 a of packed rows of the widest half)
// below the 2-GB record count of the buffer resources
bool ladder_dress_ok(int no, int nv) {
    const double ld = (double)((((long)nv * (nv + 1) / 2) + 15) & ~15L);
    return no >= 1 && no <= 80 && nv >= 1 && 8.0 * ld * (15.0 * nv + 137.0) < 2147483648.0;
}

int64_t ladder_dress_ws_doubles(int no, int nv) { return (int64_t)((nv + 15) / 16) * ((no + 3) / 4) * 64; }

void ladder_dress(const double* V, const double* Pk, const double* t1, double* W, int no, int nv, int64_t ld, int64_t row0,
                  int64_t row1, double sgn, double* ws, stream_t s) {
    if (!ladder_dress_ok(no, nv)) throw std::runtime_error("ladder_dress: nocc outside 1..80 or tile extents beyond 2 GB");
    if (ld <= 0 || (ld & 15)) throw std::runtime_error("ladder_dress: the row pitch must be a multiple of 16 doubles");
    if ((reinterpret_cast<uintptr_t>(V) | reinterpret_cast<uintptr_t>(Pk) | reinterpret_cast<uintptr_t>(W)) & 127)
        throw std::runtime_error("ladder_dress: operands must be 128-byte aligned");
    if (!ws) throw std::runtime_error("ladder_dress: workspace missing");
    const long npp = (long)nv * (nv + 1) / 2;
    if (row0 < 0 || row1 > npp || row0 > row1) throw std::runtime_error("ladder_dress: bad pair-row range");
    if (row1 == row0) return;
    const long nt = (nv + 15) / 16, ntp = nt * (nt + 1) / 2, ncdb = ld / 16, nblk = 8 * ((ncdb + 7) / 8) * ntp;
    if (nblk > 0x7fffffffL) throw std::runtime_error("ladder_dress: grid too large");
    // per-lane byte offsets inside a tile stay below the record count of the buffer resources (2 GB): the rows of 16 a of V
    if (8.0 * (double)ld * (15.0 * nv + 137.0) >= 2147483648.0) throw std::runtime_error("ladder_dress: tile extent exceeds the 2-GB buffer window");
    const dim3 grid((unsigned)nblk), block(64);
    hipStream_t st = (hipStream_t)s;
    const int nk = (no + 3) / 4;
    const long nfrag = ladder_dress_ws_doubles(no, nv);
    PYMES_LAUNCH(ladder_dress_tfrag_kernel, dim3((unsigned)((nfrag + 255) / 256)), dim3(256), 0, st, t1, ws, no, nv, nk, nfrag);
#define PYMES_DRESS(NK) case NK: PYMES_LAUNCH(ladder_dress_kernel<NK>, grid, block, 0, st, V, Pk, (const double*)ws, W, no, nv, (long)ld, (long)row0, (long)row1, sgn, ntp); break;
    switch (nk) {
        PYMES_DRESS(1) PYMES_DRESS(2) PYMES_DRESS(3) PYMES_DRESS(4) PYMES_DRESS(5) PYMES_DRESS(6) PYMES_DRESS(7) PYMES_DRESS(8)
        PYMES_DRESS(9) PYMES_DRESS(10) PYMES_DRESS(11) PYMES_DRESS(12) PYMES_DRESS(13) PYMES_DRESS(14) PYMES_DRESS(15) PYMES_DRESS(16)
        PYMES_DRESS(17) PYMES_DRESS(18) PYMES_DRESS(19) PYMES_DRESS(20)
    }
#undef PYMES_DRESS
    HIP_CHECK(hipGetLastError());
}

void ladder_pack_T(const double* X, const double* t1, double* Sp, double* Am, int nc, int nr, int flags, int64_t ldp,
                   int64_t ldm, stream_t s, int64_t rp0, int64_t rp1) {
    const long npp = (long)nr * (nr + 1) / 2, opp = (long)nc * (nc + 1) / 2, opm = (long)nc * (nc - 1) / 2;
    if (!ldp) ldp = opp;
    if (!ldm) ldm = (flags & PACK_AM_PCOLS) ? opp : opm;
    if (!X && !t1) throw std::runtime_error("ladder_pack_T: nothing to pack");
    if (rp1 < 0) { rp0 = 0; rp1 = npp; }
    if (rp0 < 0 || rp1 > npp || rp0 > rp1) throw std::runtime_error("ladder_pack_T: bad row range");
    if ((rp0 != 0 || rp1 != npp) && !(flags & PACK_AM_PROWS)) throw std::runtime_error("ladder_pack_T: a row range needs PACK_AM_PROWS");
    if (rp1 == rp0) return;
    const PackTK k{X, t1, Sp, Am, (long)ldp, (long)ldm, (long)rp0, nc, nr, flags};
    const double cost = 24.0 * (double)(rp1 - rp0) * (double)nc * (double)nc / 4.0e6;
    if (phase_open((hipStream_t)s) && phase_small(cost)) {
        PhaseRec& t = phase_push(PK_PACK_T, 0, rp1 - rp0, 0, cost, k);
        phase_reads(t, {prange(X, X ? (long)nr * nr * nc * nc : 0), prange(t1, t1 ? (long)nr * nc : 0)});
        // Sp rows [rp0, rp1); Am rows: the same pairs (PACK_AM_PROWS) or the strictly-lower pairs (all rows when the range is full)
        const long am_rows = (flags & PACK_AM_PROWS) ? npp : (long)nr * (nr - 1) / 2;
        phase_writes(t, {pbox(Sp + rp0 * ldp, {{rp1 - rp0, ldp}, {ldp, 1}}),
                         (flags & PACK_AM_PROWS) ? pbox(Am + rp0 * ldm, {{rp1 - rp0, ldm}, {ldm, 1}}) : prange(Am, am_rows * ldm)});
        return;
    }
    PYMES_LAUNCH(ladder_pack_T_kernel, dim3((unsigned)(rp1 - rp0)), dim3(256), 0, (hipStream_t)s, k);
    HIP_CHECK(hipGetLastError());
}

void ladder_unpack(const double* L, double* R, double beta, int no, int nv, stream_t s) {
    const long total = (long)nv * nv * no * no;
    if (!total) return;
    const UnpackK k{L, R, beta, total, no, nv};
    const double cost = 20.0 * (double)total / 4.0e6;
    if (phase_open((hipStream_t)s) && phase_small(cost)) {
        PhaseRec& t = phase_push(PK_LADDER_UNPACK, 0, grid_for(total), 0, cost, k);
        phase_reads(t, {prange(L, (long)nv * (nv + 1) / 2 * no * no)});
        phase_writes(t, {prange(R, total)});
        return;
    }
    PYMES_LAUNCH(ladder_unpack_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, k);
    HIP_CHECK(hipGetLastError());
}

bool fused_pair_kernels_ok(int no) { return (size_t)no * (no + 1) * sizeof(double) <= 64 * 1024; }

void ring_operands(const double* Viabj, const double* Viajb, double* M, double* N1, double a1, double a2, int no, int nv,
                   stream_t s) {
    if (no < 1 || nv < 1) return;
    const size_t lds = sizeof(double) * (size_t)no * 33;
    if (lds > 64 * 1024) throw std::runtime_error("ring_operands: nocc too large for the LDS tile");
    const RingOpK k{Viabj, Viajb, M, N1, a1, a2, no, nv};
    const long ov2 = (long)no * nv * no * nv;
    const double cost = 32.0 * (double)ov2 / 4.0e6;
    if (phase_open((hipStream_t)s) && phase_small(cost)) {
        PhaseRec& t = phase_push(PK_RING_OPERANDS, 0, (long)no * nv, (int)lds, cost, k);
        phase_reads(t, {prange(Viabj, ov2), prange(Viajb, ov2)});
        phase_writes(t, {prange(M, ov2), prange(N1, ov2)});
        return;
    }
    PYMES_LAUNCH(ring_operands_kernel, dim3((unsigned)((long)no * nv)), dim3(256), lds, (hipStream_t)s, k);
    HIP_CHECK(hipGetLastError());
}

void t2_layouts(const double* T, double* Td, double* Tx, double* Ttd, int no, int nv, stream_t s, double ca, double cb) {
    if (!fused_pair_kernels_ok(no)) throw std::runtime_error("t2_layouts: nocc too large for the LDS tile");
    const size_t lds = sizeof(double) * no * (no + 1);
    const LayoutsK k{T, Td, Tx, Ttd, ca, cb, no, nv};
    const bool res = ca == 2.0 && cb == -1.0;
    const long n4 = (long)nv * nv * no * no;
    const double cost = 40.0 * (double)n4 / 4.0e6;
    if (phase_open((hipStream_t)s) && phase_small(cost)) {
        PhaseRec& t = phase_push(PK_T2_LAYOUTS, res ? 1 : 0, (long)nv * nv, (int)lds, cost, k);
        phase_reads(t, {prange(T, n4)});
        phase_writes(t, {prange(Td, Td ? n4 : 0), prange(Tx, n4), prange(Ttd, n4)});
        return;
    }
    if (res) PYMES_LAUNCH(t2_layouts_kernel<true>, dim3((unsigned)(nv * nv)), dim3(256), lds, (hipStream_t)s, k);
    else PYMES_LAUNCH(t2_layouts_kernel<false>, dim3((unsigned)(nv * nv)), dim3(256), lds, (hipStream_t)s, k);
    HIP_CHECK(hipGetLastError());
}

void residual_assemble(const double* V, const double* L, const double* N, const double* D, const double* X, double* R,
                       int no, int nv, stream_t s, double xd) {
    if (!fused_pair_kernels_ok(no)) throw std::runtime_error("residual_assemble: nocc too large for the LDS tile");
    const size_t lds = sizeof(double) * no * (no + 1);
    const AssembleK k{V, L, N, D, X, R, xd, no, nv};
    const long n4 = (long)nv * nv * no * no;
    const double cost = 48.0 * (double)n4 / 4.0e6;
    if (phase_open((hipStream_t)s) && phase_small(cost)) {
        PhaseRec& t = phase_push(PK_ASSEMBLE, 0, (long)nv * (nv + 1) / 2, (int)lds, cost, k);
        phase_reads(t, {prange(V, V ? n4 : 0), prange(L, L ? (long)nv * (nv + 1) / 2 * no * no : 0), prange(N, n4), prange(D, n4), prange(X, n4)});
        phase_writes(t, {prange(R, n4)});
        return;
    }
    PYMES_LAUNCH(residual_assemble_kernel, dim3((unsigned)((long)nv * (nv + 1) / 2)), dim3(256), lds, (hipStream_t)s, k);
    HIP_CHECK(hipGetLastError());
}

void scatter(double* dst, const int64_t* idx_host, const double* val_host, int64_t n, stream_t s) {
    if (n <= 0) return;
    long* idx = nullptr;
    double* val = nullptr;
    HIP_CHECK(hipMalloc(&idx, sizeof(long) * n));
    if (hipMalloc(&val, sizeof(double) * n) != hipSuccess) { (void)hipFree(idx); throw std::runtime_error("scatter: out of device memory"); }
    hipStream_t st = (hipStream_t)s;
    HIP_CHECK(hipMemcpyAsync(idx, idx_host, sizeof(long) * n, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(val, val_host, sizeof(double) * n, hipMemcpyHostToDevice, st));
    PYMES_LAUNCH(scatter_kernel, dim3(grid_for(n)), dim3(256), 0, st, dst, idx, val, (long)n);
    const hipError_t e = hipGetLastError();
    (void)hipStreamSynchronize(st);
    (void)hipFree(idx);
    (void)hipFree(val);
    HIP_CHECK(e);
}

void hf_fock(const double* const dir[4], const double* const exc[4], const double* h_dev, double* f_dev, int no, int nv,
             stream_t s) {
    HfBlocks B;
    for (int i = 0; i < 4; ++i) { B.dir[i] = dir[i]; B.exc[i] = exc[i]; }
    const int n = no + nv;
    PYMES_LAUNCH(hf_fock_kernel, dim3((n * n + 255) / 256), dim3(256), 0, (hipStream_t)s, B, h_dev, f_dev, no, nv);
    HIP_CHECK(hipGetLastError());
}

int64_t fcidump_fill(double* V, const double* val_host, const int32_t* pqrs_host, int64_t count, int n, bool is_tc,
                     stream_t s) {
    if (count <= 0) return 0;
    hipStream_t st = (hipStream_t)s;
    const int64_t chunk = 1 << 22;                      // lines per upload
    double* dval = nullptr;
    int* didx = nullptr;
    unsigned long long* dbad = nullptr;
    unsigned long long bad = 0;
    HIP_CHECK(hipMalloc(&dval, sizeof(double) * std::min(count, chunk)));
    if (hipMalloc(&didx, sizeof(int) * 4 * std::min(count, chunk)) != hipSuccess || hipMalloc(&dbad, sizeof(bad)) != hipSuccess) {
        (void)hipFree(dval); (void)hipFree(didx);
        throw std::runtime_error("fcidump_fill: out of device memory");
    }
    hipError_t err = hipMemsetAsync(dbad, 0, sizeof(bad), st);
    for (int pass = 0; pass < 2 && err == hipSuccess; ++pass)          // fill everything, then verify everything
        for (int64_t b0 = 0; b0 < count && err == hipSuccess; b0 += chunk) {
            const int64_t nb = std::min(chunk, count - b0);
            err = hipMemcpyAsync(dval, val_host + b0, sizeof(double) * nb, hipMemcpyHostToDevice, st);
            if (err == hipSuccess) err = hipMemcpyAsync(didx, pqrs_host + 4 * b0, sizeof(int) * 4 * nb, hipMemcpyHostToDevice, st);
            if (err != hipSuccess) break;
            PYMES_LAUNCH(fcidump_fill_kernel, dim3(grid_for(nb)), dim3(256), 0, st, V, dval, didx, (long)nb, (long)n,
                               is_tc ? 1 : 0, pass, dbad);
            err = hipGetLastError();
            if (err == hipSuccess) err = hipStreamSynchronize(st);      // the staging buffers are reused
        }
    if (err == hipSuccess) err = hipMemcpy(&bad, dbad, sizeof(bad), hipMemcpyDeviceToHost);
    (void)hipFree(dval); (void)hipFree(didx); (void)hipFree(dbad);
    HIP_CHECK(err);
    return (int64_t)bad;
}

void tc_single_contraction(const double* L, double* D, int nb, int no, stream_t s) {
    const long total = (long)nb * nb * nb * nb;
    PYMES_LAUNCH(tc_single_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, L, D, nb, no, total);
    HIP_CHECK(hipGetLastError());
}
void tc_double_contraction(const double* L, double* S, int nb, int no, stream_t s) {
    PYMES_LAUNCH(tc_double_kernel, dim3((nb * nb + 255) / 256), dim3(256), 0, (hipStream_t)s, L, S, nb, no);
    HIP_CHECK(hipGetLastError());
}
double tc_triple_contraction(const double* L, int nb, int no, stream_t s) {
    double* out = nullptr;
    HIP_CHECK(hipMalloc(&out, sizeof(double)));
    PYMES_LAUNCH(tc_triple_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, L, out, nb, no);
    double h = 0.0;
    const hipError_t e1 = hipGetLastError();
    const hipError_t e2 = hipMemcpyAsync(&h, out, sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)s);
    (void)hipStreamSynchronize((hipStream_t)s);
    (void)hipFree(out);
    HIP_CHECK(e1);
    HIP_CHECK(e2);
    return h;
}

void pairs_pack(const double* full, double* Xc, int no, int nv, int64_t r0, int64_t r1, stream_t s) {
    if (r1 <= r0) return;
    PYMES_LAUNCH(pairs_pack_kernel, dim3((unsigned)(r1 - r0)), dim3(256), 0, (hipStream_t)s, full, Xc, no, nv, (long)r0);
    HIP_CHECK(hipGetLastError());
}
void energy_norms_pairs(const double* f, const double* t1, const double* tc, const double* Edir, const double* Eex,
                        const double* dtc, int no, int nv, int64_t r0, int64_t r1, bool with_t1, double out_host[6],
                        stream_t s) {
    hipStream_t st = (hipStream_t)s;
    const int dv = current_device();
    ensure_dot_ws(dv);
    const long npairs = std::max<long>(0, (long)(r1 - r0));
    const int nb = (int)std::max<long>(1, std::min<long>(kDotBlocks, 2 * npairs));
    PYMES_LAUNCH(energy_norms_pairs_kernel, dim3(nb), dim3(256), 0, st, f, t1, tc, Edir, Eex, dtc, no, nv, (long)r0,
                       npairs, with_t1 ? 1 : 0, g_dot_ws[dv]);
    HIP_CHECK(hipGetLastError());
    double* out_dev = g_dot_ws[dv] + 16 * kDotBlocks;
    PYMES_LAUNCH(dots_stage2_kernel, dim3(6), dim3(256), 0, st, g_dot_ws[dv], nb, out_dev);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(g_dot_host[dv], out_dev, sizeof(double) * 6, hipMemcpyDeviceToHost, st));
    wait_idle(st);
    for (int i = 0; i < 6; ++i) out_host[i] = g_dot_host[dv][i];
}
void energy_norms_pairs_dev(const double* f, const double* t1, const double* tc, const double* Edir, const double* Eex,
                            const double* dtc, int no, int nv, int64_t r0, int64_t r1, bool with_t1, double* out_dev,
                            stream_t s) {
    hipStream_t st = (hipStream_t)s;
    const int dv = current_device();
    ensure_dot_ws(dv);
    const long npairs = std::max<long>(0, (long)(r1 - r0));
    const int nb = (int)std::max<long>(1, std::min<long>(kDotBlocks, 2 * npairs));
    PYMES_LAUNCH(energy_norms_pairs_kernel, dim3(nb), dim3(256), 0, st, f, t1, tc, Edir, Eex, dtc, no, nv, (long)r0,
                       npairs, with_t1 ? 1 : 0, g_dot_ws[dv]);
    HIP_CHECK(hipGetLastError());
    PYMES_LAUNCH(dots_stage2_kernel, dim3(6), dim3(256), 0, st, g_dot_ws[dv], nb, out_dev);
    HIP_CHECK(hipGetLastError());
}
void pairs_unpack(const double* Xc, double* full, int no, int nv, int64_t r0, int64_t r1, stream_t s) {
    if (r1 <= r0) return;
    PYMES_LAUNCH(pairs_unpack_kernel, dim3((unsigned)(r1 - r0)), dim3(256), 0, (hipStream_t)s, Xc, full, no, nv, (long)r0);
    HIP_CHECK(hipGetLastError());
}
void cc_update_pairs(double* tc, double* dtc, const double* rc, const double* eo, const double* ev, double shift,
                     double delta, int no, int nv, int64_t r0, int64_t r1, stream_t s) {
    if (r1 <= r0) return;
    PYMES_LAUNCH(cc_update_pairs_kernel, dim3((unsigned)(r1 - r0)), dim3(256), 0, (hipStream_t)s, tc, dtc, rc, eo, ev,
                       shift, delta, no, (long)r0);
    HIP_CHECK(hipGetLastError());
}
void residual_assemble_pairs(const double* V, const double* L, const double* Np, const double* D, const double* X,
                             double* Rc, int no, int nv, int64_t r0, int64_t r1, int a0, int nbp, stream_t s, double xd) {
    if (r1 <= r0) return;
    if (!fused_pair_kernels_ok(no)) throw std::runtime_error("residual_assemble_pairs: nocc too large for the LDS tile");
    const size_t lds = sizeof(double) * no * (no + 1);
    PYMES_LAUNCH(residual_assemble_pairs_kernel, dim3((unsigned)(r1 - r0)), dim3(256), lds, (hipStream_t)s, V, L, Np,
                       D, X, Rc, no, nv, (long)r0, a0, nbp, xd);
    HIP_CHECK(hipGetLastError());
}

void rows_unpack(const double* Q, double* out, int64_t rows, int no, stream_t s) {
    const long total = (long)rows * no * no;
    if (!total) return;
    const RowsUnpackK k{Q, out, total, no};
    const double cost = 16.0 * (double)total / 4.0e6;
    if (phase_open((hipStream_t)s) && phase_small(cost)) {
        PhaseRec& t = phase_push(PK_ROWS_UNPACK, 0, grid_for(total), 0, cost, k);
        phase_reads(t, {prange(Q, total)});
        phase_writes(t, {prange(out, total)});
        t.box = prange(out, total);
        t.acc = 0;
        return;
    }
    PYMES_LAUNCH(rows_unpack_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, k);
    HIP_CHECK(hipGetLastError());
}

bool fock_g12_ok(int nv) { return nv >= 1 && nv <= 1024 && !((nv & 1) && nv > 512); }
static int fock_g12_chunks(int na, int nj) {          // about 1024 blocks, at least one j per chunk
    const int want = (1024 + na - 1) / na;
    return std::max(1, std::min(nj, want));
}
int64_t fock_g12_ws_doubles(int nv, int na, int nj) { return (int64_t)fock_g12_chunks(na, nj) * 2 * na * nv; }

void fock_g12(const double* V, const double* t1, double* G1, double* G2, int no, int nv, int na, int j0, int j1, double* ws,
              stream_t s) {
    if (!fock_g12_ok(nv)) throw std::runtime_error("fock_g12: nvirt above 1024 (odd: 512)");
    if (j0 < 0 || j1 > no || j0 >= j1 || na < 1) throw std::runtime_error("fock_g12: bad ranges");
    hipStream_t st = (hipStream_t)s;
    const int nj = j1 - j0, nchunk = fock_g12_chunks(na, nj), jper = (nj + nchunk - 1) / nchunk;
    const int used = (nj + jper - 1) / jper;                  // chunks that hold at least one j
    const size_t lds = sizeof(double) * 6 * (size_t)nv;
    const dim3 grid((unsigned)(na * used)), block(256);
    const bool vec2 = !(nv & 1) && !(reinterpret_cast<uintptr_t>(V) & 15);
    if (!vec2 && nv > 512) throw std::runtime_error("fock_g12: unaligned block with nvirt above 512");
    const long av = (long)na * nv;
    const FockG12K k{V, t1, ws, no, nv, na, j0, j1, jper};
    const FockG12FinK kf{ws, G1, G2, av, used};
    const double cost = 8.0 * (double)nj * (double)av * (double)nv / 4.0e6;
    if (phase_open(st) && phase_small(cost)) {
        PhaseRec& t = phase_push(PK_FOCK_G12, vec2 ? 1 : 0, (long)na * used, (int)lds, cost, k);
        phase_reads(t, {prange(V + (long)j0 * av * nv, (long)nj * av * nv), prange(t1, (long)no * nv)});
        phase_writes(t, {prange(ws, (long)used * 2 * av)});
        PhaseRec& u = phase_push(PK_FOCK_G12_FIN, 0, (av + 255) / 256, 0, 1.0, kf);
        phase_reads(u, {prange(ws, (long)used * 2 * av)});
        phase_writes(u, {prange(G1, av), prange(G2, av)});
        return;
    }
    if (vec2) PYMES_LAUNCH(fock_g12_kernel<2>, grid, block, lds, st, k);
    else PYMES_LAUNCH(fock_g12_kernel<1>, grid, block, lds, st, k);
    HIP_CHECK(hipGetLastError());
    PYMES_LAUNCH(fock_g12_finish_kernel, dim3((unsigned)((av + 255) / 256)), dim3(256), 0, st, kf);
    HIP_CHECK(hipGetLastError());
}

void fock_finish(const double* f, const double* t1, const double* W, double* fd, double* ft, int no, int nv, stream_t s) {
    const long n = no + nv;
    const FockFinK k{f, t1, W, ft, fd, no, nv};
    if (phase_open((hipStream_t)s)) {
        const long wlen = 2L * nv * nv + 4L * no * nv + 2L * no * no;        // G1 G2 J1 J2 L1 L2 K1 K2
        PhaseRec& t = phase_push(PK_FOCK_FT, 0, (no * no + 3) / 4, 0, 2.0, k);
        phase_reads(t, {prange(f, n * n), prange(t1, (long)no * nv), prange(W, wlen)});
        phase_writes(t, {prange(ft, (long)no * no)});
        PhaseRec& u = phase_push(PK_FOCK_FIN, 0, (n * n + 3) / 4, 0, 3.0, k);
        phase_reads(u, {prange(f, n * n), prange(t1, (long)no * nv), prange(W, wlen), prange(ft, (long)no * no)});
        phase_writes(u, {prange(fd, n * n)});
        return;
    }
    PYMES_LAUNCH(fock_ft_kernel, dim3((unsigned)((no * no + 3) / 4)), dim3(256), 0, (hipStream_t)s, k);
    HIP_CHECK(hipGetLastError());
    PYMES_LAUNCH(fock_finish_kernel, dim3((unsigned)((n * n + 3) / 4)), dim3(256), 0, (hipStream_t)s, k);
    HIP_CHECK(hipGetLastError());
}

void pair_traces(const double* M, int64_t ld, double alpha, double beta, double* out_vv, double* out_oo, int no, int nv,
                 stream_t s, const double* M2, double alpha2) {
    if (no <= 0 || nv <= 0) return;
    const unsigned blocks = (unsigned)nv * (unsigned)((nv + 15) / 16) + (unsigned)no;
    const TracesK k{M, M2, out_vv, out_oo, (long)ld, alpha, beta, alpha2, no, nv};
    const long ov = (long)no * nv;
    const double cost = 128.0 / 3.0 * (double)ov * (double)ov / 8.0 / 4.0e6 * (M2 ? 2.0 : 1.0) + 3.0;
    if (phase_open((hipStream_t)s) && phase_small(cost)) {
        PhaseRec& t = phase_push(PK_TRACES, 0, blocks, kTracesLdsDoubles * (int)sizeof(double), cost, k);
        phase_reads(t, {pbox(M, {{ov, ld}, {ov, 1}}), M2 ? pbox(M2, {{ov, ld}, {ov, 1}}) : PhaseRange{0, 0}});
        phase_writes(t, {prange(out_vv, (long)nv * nv), prange(out_oo, (long)no * no)});
        return;
    }
    PYMES_LAUNCH(pair_traces_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, k);
    HIP_CHECK(hipGetLastError());
}

void ueg_two_body(const UegParams& prm, const int* k_int_dev, const int* index_map_dev, double* V_dev, stream_t s) {
    hipStream_t st = (hipStream_t)s;
    UegK u;
    u.n_p = prm.n_p; u.n_occ = prm.n_ele / 2; u.imax = prm.imax; u.m = 2 * prm.imax + 1; u.mode = prm.mode;
    u.n_ele = prm.n_ele; u.lat = prm.lattice_cutoff; u.L = prm.L; u.Omega = prm.Omega; u.gamma = prm.gamma;
    const double kc = prm.k_cutoff * 2 * M_PI / prm.L;
    u.kc2g = kc * kc * (1 + 0.00001);
    u.tab_s = u.tab_a = nullptr;
    u.tab_len = 0;
    u.kind = prm.corr_kind; u.p0 = prm.corr_p[0]; u.p1 = prm.corr_p[1]; u.p2 = prm.corr_p[2];
    if (u.kind < 0 || u.kind > 6) throw std::runtime_error("ueg: unknown correlator kind");
    const long n = prm.n_p;
    double* tabs = nullptr;
    if (prm.tab_array) {
        if (!prm.tab_scalar || prm.tab_len < 1) throw std::runtime_error("ueg: both correlator tables are needed");
        tabs = (double*)dmalloc(sizeof(double) * 2 * prm.tab_len);
        HIP_CHECK(hipMemcpyAsync(tabs, prm.tab_scalar, sizeof(double) * prm.tab_len, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(tabs + prm.tab_len, prm.tab_array, sizeof(double) * prm.tab_len, hipMemcpyHostToDevice, st));
        u.tab_s = tabs; u.tab_a = tabs + prm.tab_len; u.tab_len = prm.tab_len;
    }
    HIP_CHECK(hipMemsetAsync(V_dev, 0, sizeof(double) * n * n * n * n, st));
    std::vector<int> kint(3 * n);
    HIP_CHECK(hipMemcpyAsync(kint.data(), k_int_dev, sizeof(int) * 3 * n, hipMemcpyDeviceToHost, st));
    wait_idle(st);
    double *umat = nullptr, *E = nullptr, *dk_dev = nullptr;
    int *uidx = nullptr, *dint_dev = nullptr;
    try {
        if (prm.mode == 1) {
            // distinct momentum transfers d = k_r - k_p, with the float d_k of their first (p,r) pair
            const int w4 = 4 * prm.imax + 1;
            std::vector<int> index((size_t)w4 * w4 * w4, -1);
            std::vector<double> dks;
            std::vector<int> dints;
            for (long p = 0; p < n; ++p)
                for (long r = 0; r < n; ++r) {
                    int d[3];
                    for (int c = 0; c < 3; ++c) {
                        d[c] = kint[3 * r + c] - kint[3 * p + c];
                        if (d[c] < -2 * prm.imax || d[c] > 2 * prm.imax) throw std::runtime_error("ueg: k outside the index map");
                    }
                    int& slot = index[((size_t)(d[0] + 2 * prm.imax) * w4 + (d[1] + 2 * prm.imax)) * w4 + d[2] + 2 * prm.imax];
                    if (slot < 0) {
                        slot = (int)(dks.size() / 3);
                        for (int c = 0; c < 3; ++c) dints.push_back(d[c]);
                        for (int c = 0; c < 3; ++c)
                            dks.push_back(((double)(kint[3 * r + c] * 2) * M_PI) / prm.L - ((double)(kint[3 * p + c] * 2) * M_PI) / prm.L);
                    }
                }
            const int nd = (int)(dks.size() / 3);
            umat = (double*)dmalloc(sizeof(double) * nd);
            dk_dev = (double*)dmalloc(sizeof(double) * 3 * nd);
            uidx = (int*)dmalloc(sizeof(int) * index.size());
            dint_dev = (int*)dmalloc(sizeof(int) * 3 * nd);
            HIP_CHECK(hipMemcpyAsync(dint_dev, dints.data(), sizeof(int) * 3 * nd, hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(dk_dev, dks.data(), sizeof(double) * 3 * nd, hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(uidx, index.data(), sizeof(int) * index.size(), hipMemcpyHostToDevice, st));
            PYMES_LAUNCH(ueg_nabla_kernel, dim3(nd), dim3(256), 0, st, u, dk_dev, dint_dev, umat);
            HIP_CHECK(hipGetLastError());
        } else if (prm.mode == 2) {
            E = (double*)dmalloc(sizeof(double) * n * n);
            PYMES_LAUNCH(ueg_effective_kernel, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, st, u, k_int_dev, E);
            HIP_CHECK(hipGetLastError());
        }
        const long total = n * n * n;
        PYMES_LAUNCH(ueg_scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, u, k_int_dev,
                           index_map_dev, umat, uidx, E, V_dev);
        HIP_CHECK(hipGetLastError());
        wait_idle(st);
    } catch (...) {
        dfree(umat); dfree(E); dfree(dk_dev); dfree(uidx); dfree(dint_dev); dfree(tabs);
        throw;
    }
    dfree(umat); dfree(E); dfree(dk_dev); dfree(uidx); dfree(dint_dev); dfree(tabs);
}

}  // namespace dev

namespace {

bool gemv_dispatch(const dev::Gemm& g, long a_sm, long a_sk, long b_sk, long b_sn, hipStream_t st) {
    if (g.nb1 != 1 || g.nb2 != 1 || g.K < 256) return false;
    const double* W; const double* x; long ld, xs, R, C, ys; bool cols;
    if (g.M == 1 && g.N >= 64) {                 // y[n] = sum_k A(0,k) B(k,n)
        x = g.A; xs = a_sk; W = g.B; ys = 1;
        if (b_sn == 1) { cols = true; R = g.K; C = g.N; ld = b_sk; }          // B[k][n]: weighted column sums
        else { cols = false; R = g.N; C = g.K; ld = b_sn; }                    // B[n][k]: one dot per row
    } else if (g.N == 1 && g.M >= 64) {          // y[m] = sum_k A(m,k) B(k,0)
        x = g.B; xs = b_sk; W = g.A; ys = g.ldc;
        if (a_sk == 1) { cols = false; R = g.M; C = g.K; ld = a_sm; }
        else { cols = true; R = g.K; C = g.M; ld = a_sk; }
    } else {
        return false;
    }
    if (!cols && R < 512) return false;          // too few rows to fill the chip with one wave per row
    const double* yin = g.Cin ? g.Cin : g.C;
    const int vec = (even(ld) && even(C) && aligned16(W)) ? 2 : 1;
    const long cblocks = (C + 256L * vec - 1) / (256L * vec);
    long nchunk = 1, rchunk = R;
    if (cols) {        // row chunks so that about 2048 blocks are in flight; their partial sums go through the workspace
        if (!g.splitk_ws || C > g.splitk_ws_doubles) return false;
        nchunk = std::max<long>(1, std::min<long>((2048 + cblocks - 1) / cblocks, R / 32));
        nchunk = std::min<long>(nchunk, g.splitk_ws_doubles / C);
        rchunk = (R + nchunk - 1) / nchunk;
        nchunk = (R + rchunk - 1) / rchunk;
    }
    if (phase_open(st)) {
        const double cost = 8.0 * (double)R * (double)C / 4.0e6;
        if (phase_small(cost)) {
            const PhaseRange rW = pbox(W, {{R, ld}, {C, 1}}, 1);
            if (cols) {
                double* part = phase_ws(g.splitk_ws, g.splitk_ws_doubles, nchunk * C);
                if (part) {
                    GemvTaskK t;
                    t.it.W = W; t.it.x = x; t.it.y = g.C; t.it.ld = ld; t.it.xs = xs; t.it.R = R; t.it.C = C; t.it.rchunk = rchunk;
                    t.it.ys = ys; t.it.ws_off = 0; t.it.alpha = g.alpha; t.it.nchunk = (int)nchunk; t.it.cblocks = (int)cblocks;
                    t.it.vec = vec; t.it.blk0 = t.it.out0 = 0;
                    t.ws = part; t.yin = yin; t.beta = g.beta;
                    const PhaseRange rP = prange(part, nchunk * C);
                    PhaseRec& a = phase_push(PK_GEMV_COLS, 0, cblocks * nchunk, 0, cost, t);
                    phase_reads(a, {rW, pbox(x, {{R, xs}})});
                    phase_writes(a, {rP});
                    PhaseRec& f = phase_push(PK_GEMV_FINISH, 0, (C + 255) / 256, 0, 1.0, t);
                    phase_reads(f, {rP, g.beta != 0.0 ? pbox(yin, {{C, ys}}) : PhaseRange{0, 0}});
                    phase_writes(f, {pbox(g.C, {{C, ys}})});
                    return true;
                }
            } else {
                GemvRowsK t{W, x, yin, g.C, ld, xs, R, C, ys, g.alpha, g.beta};
                PhaseRec& a = phase_push(PK_GEMV_ROWS, vec == 2 ? 1 : 0, (R + 3) / 4, 0, cost, t);
                phase_reads(a, {rW, pbox(x, {{C, xs}}), g.beta != 0.0 ? pbox(yin, {{R, ys}}) : PhaseRange{0, 0}});
                phase_writes(a, {pbox(g.C, {{R, ys}})});
                return true;
            }
        }
    }
    {
        GemvBatch& b = g_gemv_batch;
        if (b.active && !g_prof.on) {        // (per-call event timing wants every product on its own)
            const bool same = b.tab.n == 0 || (b.ws == g.splitk_ws && b.st == st);
            if (cols && g.beta == 0.0 && same && b.tab.n < kGemvBatchMax && b.ws_used + nchunk * C <= g.splitk_ws_doubles) {
                GemvItem& it = b.tab.it[b.tab.n++];
                it.W = W; it.x = x; it.y = g.C; it.ld = ld; it.xs = xs; it.R = R; it.C = C; it.rchunk = rchunk; it.ys = ys;
                it.ws_off = b.ws_used; it.alpha = g.alpha; it.nchunk = (int)nchunk; it.cblocks = (int)cblocks; it.vec = vec;
                it.blk0 = it.out0 = 0;
                b.ws_used += nchunk * C;
                b.ws = g.splitk_ws;
                b.st = st;
                return true;
            }
            gemv_batch_flush();              // this one runs on its own, behind the collected ones
        }
    }
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (g_prof.on) {
        if (!g_prof.pool.empty()) { ev = g_prof.pool.back(); g_prof.pool.pop_back(); }
        else { HIP_CHECK(hipEventCreate(&ev.first)); HIP_CHECK(hipEventCreate(&ev.second)); }
        HIP_CHECK(hipEventRecord(ev.first, st));
    }
    if (cols) {
        if (vec == 2) PYMES_LAUNCH(gemv_cols_kernel<2>, dim3((unsigned)cblocks, (unsigned)nchunk), dim3(256), 0, st, W, ld, x, xs, R, C, rchunk, g.splitk_ws);
        else PYMES_LAUNCH(gemv_cols_kernel<1>, dim3((unsigned)cblocks, (unsigned)nchunk), dim3(256), 0, st, W, ld, x, xs, R, C, rchunk, g.splitk_ws);
        HIP_CHECK(hipGetLastError());
        PYMES_LAUNCH(gemv_finish_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, st, g.splitk_ws, (int)nchunk, C,
                           g.alpha, g.beta, yin, g.C, ys);
    } else {
        const GemvRowsK rk{W, x, yin, g.C, ld, xs, R, C, ys, g.alpha, g.beta};
        if (vec == 2) PYMES_LAUNCH(gemv_rows_kernel<2>, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, rk);
        else PYMES_LAUNCH(gemv_rows_kernel<1>, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, rk);
    }
    HIP_CHECK(hipGetLastError());
    if (g_prof.on) {
        HIP_CHECK(hipEventRecord(ev.second, st));
        g_prof.ev.push_back(ev);
        const double fl = 2.0 * (double)g.M * (double)g.N * (double)g.K;
        g_prof.flops += fl;
        g_prof.fl.push_back(fl);
        g_prof.klass.push_back(0);
        g_prof.nk.push_back(1);
        char buf[256];
        snprintf(buf, sizeof buf, "M=%ld N=%ld K=%ld batch=1 gemv=%s vec=%d flops=%.4e", (long)g.M, (long)g.N, (long)g.K,
                 cols ? "cols" : "rows", vec, fl);
        g_prof.what.push_back(buf);
    }
    return true;
}

}  // namespace
