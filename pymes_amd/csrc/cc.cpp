// Coupled-cluster term sequences on top of Engine::contract / permute.
//
// Every function cites the reference lines it reproduces (nickirk/pymes @ 2024_10_08).
// The index algebra is restated for GEMM-friendly layouts; no symmetry of V or T is
// assumed anywhere (the transcorrelated Hamiltonian has only V_pqrs = V_qpsr, and the
// reference keeps V_ijab and V_abij separate, ccd.py:172).
//
// Matrix layouts used for the o^3 v^3 terms (all ov x ov matrices):
//   Td[(a,i),(b,j)] = T[a,b,i,j]   "direct" pairing      (labels "aibj")
//   Tx[(a,j),(b,i)] = T[a,b,i,j]   "crossed" pairing     (labels "ajbi")
// so that every ring/exchange term of ccd.py:190-240 is a plain product of two such
// matrices (possibly transposed), with no per-term transposition of the operands.
#include <algorithm>
#include <cstring>

#include "engine.h"

#include <cmath>

namespace pymes {

namespace {
// Row pitch of the pair-packed integrals V^+- (the K-contiguous left operand of the ladder GEMMs): a multiple of 16 doubles,
// so that every 128-byte piece the LDS-DMA fetches is one cache line (v(v+1)/2 = 20100 is not: two lines per piece).
inline int64_t lpitch(int64_t n) { return (n + 15) & ~int64_t(15); }
inline TView packed_rows(double* p, int64_t rows, int64_t cols) { return slice(make_view(p, {rows, lpitch(cols)}), 1, 0, cols); }
constexpr int P_klij = 0, P_ijka = 1, P_ijak = 2, P_ijab = 3, P_iajk = 4, P_iajb = 5, P_iabj = 6, P_iabc = 7,
              P_aibc = 11, P_abij = 12, P_abic = 13, P_abci = 14, P_abcd = 15;
}

// -----------------------------------------------------------------------------------
// static (per-solve) permutations of the undressed V_ijab block.  The T1-dressed ijab
// block is a plain copy of the undressed one (ccsd.py:355-357), so these stay valid for
// the whole solve.
// -----------------------------------------------------------------------------------
double* Engine::get_static(const std::string& key) {
    auto it = static_.find(key);
    if (it != static_.end()) return it->second;
    const int64_t o = no, v = nv;
    TView Vijab = block(P_ijab);
    double* p = static_cast<double*>(dev::dmalloc(sizeof(double) * o * o * v * v));
    static_[key] = p;
    // all keys: source labels of V_ijab are "klcd" (k,l occupied; c,d virtual)
    if (key == "Vd") permute(1.0, Vijab, "klcd", 0.0, make_view(p, {v, o, v, o}), "ckdl");        // [(c,k),(d,l)]
    else if (key == "Vx") permute(1.0, Vijab, "klcd", 0.0, make_view(p, {v, o, v, o}), "cldk");   // [(c,l),(d,k)]
    else if (key == "Ld") {                                      // 2 Vd - Vx: Ld[(c,k),(d,l)] = 2 V_klcd - V_lkcd
        permute(2.0, Vijab, "klcd", 0.0, make_view(p, {v, o, v, o}), "ckdl");
        permute(-1.0, Vijab, "lkcd", 1.0, make_view(p, {v, o, v, o}), "ckdl");
    }
    else if (key == "Ldh") {                                     // Ld / 2 = Vd - Vx / 2 (the D-term build of the paired ring products)
        permute(1.0, Vijab, "klcd", 0.0, make_view(p, {v, o, v, o}), "ckdl");
        permute(-0.5, Vijab, "lkcd", 1.0, make_view(p, {v, o, v, o}), "ckdl");
    }
    else if (key == "Vk") permute(1.0, Vijab, "lkdc", 0.0, make_view(p, {o, v, o, v}), "kdlc");   // [(k,d,l),c]
    else if (key == "Vk2") permute(1.0, Vijab, "lkdc", 0.0, make_view(p, {o, v, v, o}), "kcdl");  // [k,(c,d,l)]
    // K-major copies for the o x o results contracted over o v^2 (K = 2e6 at the benchmark size): with the long index
    // slowest both GEMM operands stream contiguously instead of gathering 128-byte pieces of 50 distant rows
    else if (key == "Vk3") permute(1.0, Vijab, "lkdc", 0.0, make_view(p, {v, v, o, o}), "cdlk");  // [(c,d,l),k]
    else if (key == "Vjbck") permute(1.0, Vijab, "kjbc", 0.0, make_view(p, {o, v, v, o}), "jbck"); // [(j,b,c),k]
    else if (key == "Edir") permute(1.0, Vijab, "ijab", 0.0, make_view(p, {v, v, o, o}), "abij"); // V[i,j,a,b]
    else if (key == "Eex") permute(1.0, Vijab, "ijba", 0.0, make_view(p, {v, v, o, o}), "abij");  // V[i,j,b,a]
    else throw Error("unknown static tensor " + key);
    return p;
}

// -----------------------------------------------------------------------------------
// mp2.py:9-22
// -----------------------------------------------------------------------------------
void Engine::mp2(double shift, double* t2, double e_out[2]) {
    const int64_t o = no, v = nv, n4 = v * v * o * o;
    need_eps("mp2");
    TView Vabij = block(P_abij);
    dev::mp2_amplitudes(t2, Vabij.p, eps_o, eps_v, shift, no, nv, stream);                 // :16-18
    ArenaScope scope(arena);
    double* e2 = arena.alloc(n4);
    permute(1.0, block(P_ijab), "jiab", 0.0, make_view(e2, {v, v, o, o}), "abij");          // 'abij,jiab' :20
    const double* x[2] = {t2, t2};
    const double* y[2] = {get_static("Edir"), e2};
    const int64_t len[2] = {n4, n4};
    double d[2];
    dev::dots(2, x, y, len, d, stream);
    e_out[0] = 2.0 * d[0];                                                                  // :19
    e_out[1] = -1.0 * d[1];                                                                 // :20
}

// -----------------------------------------------------------------------------------
// ccd.py:164-254  (the doubles residual; also the CCSD one via dressed blocks, ccsd.py:440-456)
// -----------------------------------------------------------------------------------
void Engine::doubles_residual(const double* f, const double* t2, double* r2, unsigned flags) {
    const bool dcd = flags & 1u, dressed = flags & 2u, skip_ladder = flags & 4u, sym_ladder = flags & 8u,
               sym_rings = flags & 16u;
    const bool quad = !dcd;
    const int64_t o = no, v = nv, nn = n;
    if (sym_rings) {      // symmetry-reduced evaluation = the one-rank case of the sharded form
        ArenaScope sc(arena);
        double* ETd = arena.alloc(o * v * o * v);
        double* ETx = arena.alloc(o * v * o * v);
        double* L = (sym_ladder && !skip_ladder) ? arena.alloc(v * (v + 1) / 2 * o * o) : nullptr;
        residual_slab(f, t2, ETd, ETx, L, 0, 1, flags);
        residual_finish(f, t2, ETd, ETx, L, r2, flags);
        return;
    }
    const double w = quad ? 1.0 : 0.5;                                                      // :213-220
    TView T = make_view(const_cast<double*>(t2), {v, v, o, o});
    TView R = make_view(r2, {v, v, o, o});
    TView F = make_view(const_cast<double*>(f), {nn, nn});
    TView Foo = slice(slice(F, 0, 0, o), 1, 0, o), Fvv = slice(slice(F, 0, o, nn), 1, o, nn);
    TView Vklij = block(P_klij, dressed), Vabij = block(P_abij, dressed), Viajb = block(P_iajb, dressed),
          Viabj = block(P_iabj, dressed);
    TView Vijab = block(P_ijab);   // dressed ijab == undressed ijab (ccsd.py:355-357)

    ArenaScope scope(arena);
    auto ov2 = [&]() { return make_view(arena.alloc(o * o * v * v), {v, o, v, o}); };
    TView Vd = make_view(get_static("Vd"), {v, o, v, o});
    TView Vk = make_view(get_static("Vk"), {o, v, o, v});
    TView Vk2 = make_view(get_static("Vk2"), {o, v, v, o});

    // permuted amplitudes: Td, Tx and Tt_d = direct layout of 2T - T^(ab)   (:199)
    TView Td = ov2(), Tx = ov2(), Ttd = ov2();
    permute(1.0, T, "abij", 0.0, Td, "aibj");
    permute(1.0, T, "abij", 0.0, Tx, "ajbi");
    permute(2.0, T, "abij", 0.0, Ttd, "aibj");
    permute(-1.0, T, "baij", 1.0, Ttd, "aibj");

    // ---- terms kept in the natural [a,b,i,j] layout -------------------------------------
    copy(Vabij, R);                                                                         // :185
    {
        ArenaScope s2(arena);
        TView hole = make_view(arena.alloc(o * o * o * o), {o, o, o, o});
        copy(Vklij, hole);                                                                  // :178
        if (quad) contract(1.0, Vijab, "klcd", T, "cdij", 1.0, hole, "klij");                // :180
        contract(1.0, hole, "klij", T, "abkl", 1.0, R, "abij");                             // :186
    }
    if (!skip_ladder) {                                                                     // :187
        if (sym_ladder) {
            ArenaScope s2(arena);
            double* L = arena.alloc(v * (v + 1) / 2 * o * o);
            ladder_sym(t2, L, 0, v * (v + 1) / 2, dressed, 0);
            ladder_sym_unpack(L, r2, 1.0);
        } else {
            contract(1.0, block(P_abcd, dressed), "abcd", T, "cdij", 1.0, R, "abij");
        }
    }

    // ---- X_ac, X_ki (:206-221) ------------------------------------------------------------
    TView Xvv = make_view(arena.alloc(v * v), {v, v}), Xoo = make_view(arena.alloc(o * o), {o, o});
    copy(Fvv, Xvv);
    copy(Foo, Xoo);
    // X_ac = f_ac - w sum_{dkl} Tt[a,d,k,l] V[l,k,d,c];  Tt[a,d,k,l] = Ttd[a,k,d,l]
    contract(-w, Ttd, "akdl", Vk, "kdlc", 1.0, Xvv, "ac");
    {
        // X_ki = f_ki + w sum_{cdl} Tt[c,d,i,l] V[l,k,d,c];  Tt[c,d,i,l] = Ttd[c,i,d,l]
        ArenaScope s2(arena);
        TView Tp = make_view(arena.alloc(o * o * v * v), {o, v, v, o});
        permute(1.0, Ttd, "cidl", 0.0, Tp, "icdl");
        contract(w, Vk2, "kcdl", Tp, "icdl", 1.0, Xoo, "ki");
    }

    // ---- accumulators in the pair layouts ---------------------------------------------------
    TView Exn = make_view(arena.alloc(o * o * v * v), {v, v, o, o});
    contract(1.0, Xvv, "ac", T, "cbij", 0.0, Exn, "abij");                                   // :231
    TView Exd = ov2(), Exx = ov2();
    TView Rd = ov2();
    {
        // :202-204  R += Tt . (V . Tt)   in the direct layout
        ArenaScope s2(arena);
        TView Y = ov2();
        contract(1.0, Vd, "ckdl", Ttd, "dlbj", 0.0, Y, "ckbj");
        contract(1.0, Ttd, "aick", Y, "ckbj", 0.0, Rd, "aibj");
    }
    {
        ArenaScope s2(arena);
        TView Ud = ov2(), Wd = ov2();
        permute(1.0, Viajb, "kaic", 0.0, Ud, "aick");     // Ud[(a,i),(c,k)] = V_iajb[k,a,i,c]
        permute(1.0, Viabj, "kbcj", 0.0, Wd, "ckbj");     // Wd[(c,k),(b,j)] = V_iabj[k,b,c,j]
        contract(-1.0, Ud, "aick", Td, "ckbj", 0.0, Exd, "aibj");                            // :233
        contract(1.0, Ttd, "aick", Wd, "ckbj", 1.0, Exd, "aibj");                            // :235
        contract(-1.0, Tx, "ajck", Ud, "bick", 0.0, Exx, "ajbi");                            // :234
    }
    contract(-1.0, Xoo, "ki", Td, "akbj", 1.0, Exd, "aibj", "a");                            // :232
    TView Rx;
    if (quad) {
        TView Vx = make_view(get_static("Vx"), {v, o, v, o});
        {
            // :238-240  Z[(a,i),(c,l)] = sum_{dk} T[d,a,k,i] V[k,l,c,d];  Ex += Z . (T[b,c,l,j] - T[c,b,l,j])
            ArenaScope s2(arena);
            TView Z = ov2(), Q = ov2();
            contract(1.0, Td, "dkai", Vx, "cldk", 0.0, Z, "aicl");
            permute(-1.0, T, "cblj", 0.0, Q, "clbj");
            permute(1.0, T, "bclj", 1.0, Q, "clbj");
            contract(1.0, Z, "aicl", Q, "clbj", 1.0, Exd, "aibj");
        }
        // :190-191  R += (Tx . Vx^T) . Tx   in the crossed layout
        Rx = ov2();
        TView X1 = ov2();
        contract(1.0, Tx, "ajdk", Vx, "cldk", 0.0, X1, "ajcl");
        contract(1.0, X1, "ajcl", Tx, "clbi", 0.0, Rx, "ajbi");
    }

    // ---- assemble: R += P(Rd) + P(Rx) + Ex + Ex^T(1,0,3,2)   (:249-252) ----------------------
    permute(1.0, Rd, "aibj", 1.0, R, "abij");
    if (quad) permute(1.0, Rx, "ajbi", 1.0, R, "abij");
    permute(1.0, Exd, "aibj", 1.0, R, "abij");
    permute(1.0, Exd, "bjai", 1.0, R, "abij");
    permute(1.0, Exx, "ajbi", 1.0, R, "abij");
    permute(1.0, Exx, "biaj", 1.0, R, "abij");
    permute(1.0, Exn, "abij", 1.0, R, "abij");
    permute(1.0, Exn, "baji", 1.0, R, "abij");
}

// -----------------------------------------------------------------------------------
// Symmetry-reduced, shardable T2 residual (requires T_abij = T_baji and V_pqrs = V_qpsr).
//
// Exchange-symmetric amplitudes make Td, Tx, Tt_d, Vd, Vx symmetric ov x ov matrices.  Round 1 merged the ten
// o^3v^3 products of the reference into 6 (5 -> 4 for DCSD) inside its own term structure; round 2 uses the C / D form of
// the closed-shell doubles equations, which is the same sum in 4 (3) products — see the comment at the products below:
//     Exd = 1/2 Tt_d (2 Wd - UdT + 1/2 Ld Tt_d) - 1/2 Xc,   Exx = -Xc,   Xc = Tx (UdT - 1/2 Vx Tx),   Ld = 2 Vd - Vx,
//     Wd[(c,k),(b,j)] = V_iabj[k,b,c,j],  UdT[(c,k),(b,j)] = V_iajb[k,b,j,c]
// COLUMNS (b,j) of these products never mix, so a rank needs no communication to compute its column
// slab; the slab is produced TRANSPOSED (ET[(b,j),(a,i)] = Ex[(a,i),(b,j)]) so that the slabs of all
// ranks are contiguous row blocks of ETd / ETx = one all-gather each.  Since only Ex + Ex^T enters R
// the transposition needs no undoing.  The pair-packed ladder rows of the same rank go to L.
// -----------------------------------------------------------------------------------
void Engine::ensure_xs() {
    if (!xs_oo_) xs_oo_ = static_cast<double*>(dev::dmalloc(sizeof(double) * no * no));
    if (!xs_vv_) xs_vv_ = static_cast<double*>(dev::dmalloc(sizeof(double) * nv * nv));
}

// The three pair layouts of T2 (Td, Tx, Tt_d) in the engine's persistent buffers
void Engine::pair_layouts_of(const double* t2) {
    const int64_t o = no, v = nv, ov = o * v;
    for (auto& p : lay_)
        if (!p) p = static_cast<double*>(dev::dmalloc(sizeof(double) * ov * ov));
    lay_t2_ = nullptr;
    if (dev::fused_pair_kernels_ok(no)) {
        dev::t2_layouts(t2, lay_[0], lay_[1], lay_[2], no, nv, stream);
        stats.permute_calls++;
        stats.permute_bytes += 8.0 * 5.0 * double(ov * ov);
    } else {
        TView T = make_view(const_cast<double*>(t2), {v, v, o, o});
        TView t4 = make_view(lay_[0], {v, o, v, o});
        permute(1.0, T, "abij", 0.0, t4, "aibj");
        t4.p = lay_[1];
        permute(1.0, T, "abij", 0.0, t4, "ajbi");
        t4.p = lay_[2];
        permute(2.0, T, "abij", 0.0, t4, "aibj");
        permute(-1.0, T, "baij", 1.0, t4, "aibj");
    }
    lay_t2_ = t2;
}

void Engine::residual_slab(const double* f, const double* t2, double* ETd_p, double* ETx_p, double* L, int rank,
                           int world, unsigned flags, const double* t1, double* QK, const double* P) {
    const bool dcd = flags & 1u, dressed = flags & 2u, skip_ladder = (flags & 4u) || (flags & 64u), skip_rings = flags & 128u;
    const bool quad = !dcd;
    const int64_t o = no, v = nv, nn = n, ov = o * v;
    if (world < 1 || rank < 0 || rank >= world) throw Error("residual_slab: bad rank/world");
    const double w = quad ? 1.0 : 0.5;
    TView F = make_view(const_cast<double*>(f), {nn, nn});
    auto chunk = [&](int64_t rows, int64_t& lo, int64_t& hi) {
        const int64_t c = (rows + world - 1) / world;
        lo = std::min<int64_t>(rank * c, rows);
        hi = std::min<int64_t>(lo + c, rows);
    };
    if (L && !skip_ladder) {
        int64_t r0, r1;
        chunk(v * (v + 1) / 2, r0, r1);
        if (t1) {
            if (!QK) throw Error("residual_slab: QK buffer missing");
            int64_t q0, q1;
            chunk(o * v, q0, q1);
            ladder_t1(t1, t2, L, r0, r1, QK, q0, q1, dcd, P ? P + o * o : nullptr);
        } else {
            ladder_sym(t2, L, r0, r1, dressed, quad ? 1 : 2);  // particle AND hole ladder rows of this rank
        }
    }
    if (skip_rings) return;
    int64_t c0, c1;
    chunk(ov, c0, c1);
    const int64_t nc = c1 - c0;
    if (nc <= 0) return;           // a rank without columns never touches (or needs) the dressed ov blocks
    TView Viajb = block(P_iajb, dressed), Viabj = block(P_iabj, dressed);

    ArenaScope scope(arena);
    auto pairm = [&](double* p) { return make_view(p, {ov, ov}); };
    auto slab = [&]() { return make_view(arena.alloc(ov * nc), {ov, nc}); };
    pair_layouts_of(t2);
    TView Td = pairm(lay_[0]), Tx = pairm(lay_[1]), Ttd = pairm(lay_[2]);
    TView ETd = slice(pairm(ETd_p), 0, c0, c1), ETx = slice(pairm(ETx_p), 0, c0, c1);
    auto cols = [&](const TView& m) { return slice(m, 1, c0, c1); };
    // column slabs (b,j) in [c0,c1) of the static / dressed right-hand factors, straight from the 4-index blocks:
    //   Wd[(c,k),(b,j)] = V_iabj[k,b,c,j],   UdT[(c,k),(b,j)] = V_iajb[k,b,j,c]
    const int64_t b0 = c0 / o, b1 = (c1 + o - 1) / o;
    auto load_cols = [&](double alpha, const TView& blk, const char* spec, const TView& dst) {
        TView src = slice(blk, 1, b0, b1);                       // only the b values the slab touches
        if (b0 * o == c0 && b1 * o == c1) {
            permute(alpha, src, spec, 0.0, make_view(dst.p, {v, o, b1 - b0, o}), "ckbj");
        } else {
            ArenaScope s2(arena);
            TView tmp = make_view(arena.alloc(ov * (b1 - b0) * o), {ov, (b1 - b0) * o});
            permute(alpha, src, spec, 0.0, make_view(tmp.p, {v, o, b1 - b0, o}), "ckbj");
            copy(slice(tmp, 1, c0 - b0 * o, c1 - b0 * o), dst);
        }
    };
    // ---- the o^3 v^3 terms in four products (three for DCSD).  With pair matrices on (a,i) = a*o + i, all symmetric,
    //   Wd[(c,k),(b,j)] = V~_iabj[k,b,c,j],  Ud^T[(c,k),(b,j)] = V~_iajb[k,b,j,c],  Ld = 2 Vd - Vx  (L_ldkc = 2 g_ldkc - g_lckd)
    // the ten ring products of ccd.py:190-191, :199-204, :233-240 are, for V_pqrs = V_qpsr and T_abij = T_baji, exactly
    //   D-term:  Ex_d  = 1/2 Tt_d (2 Wd - Ud^T + 1/2 Ld Tt_d)            (one build, one application)
    //   C-term:  Xc    = Tx (Ud^T - 1/2 Vx Tx),   Ex_d -= 1/2 Xc,   Ex_x = -Xc     (one build, ONE application, used in both
    //            index placements — the (1/2 + P_ij) of the closed-shell CCSD doubles equations in their C / D form)
    // (numerically identical to the reference's sequence to rounding: tests/test_host_round2.py pins the identity and every
    // golden solve pins the result).  DCSD keeps ccd.py:202-204 only: 2 Wd - Ud^T + Vd Tt_d in the D-term, no build in the
    // C-term.  Column slab [c0,c1): both builds are restricted to the rank's columns n, the applications give rows n.
    const bool row_form = nc != ov;
    const bool traces = !P && nc == ov;      // the small V.T sums as partial traces of the builds (below; one rank only)
    if (row_form) {
        // Several ranks: the slab in its TRANSPOSED form from the start.  MT[(b,j),(c,k)] / N1T hold the rank's columns as
        // ROWS; Tt_d, Tx, Ld, Vx, Vd are symmetric pair matrices, so
        //     MT += rows(Tt_d) Ld / 2,  N1T += rows(Tx) Vx / 2,  ET_x = N1T Tx,  ET_d = MT Tt_d / 2 + ET_x / 2
        // are all products with a K-contiguous left and an N-contiguous right operand — the LDS-DMA variant that runs at 93 %
        // of peak on one rank (the column form needed the <false,true> variant for its applications: 85.5 % at N = 8,
        // profiles/r03/stub_rank0_of8.json) — and their outputs ARE the contiguous row blocks of ETd / ETx that are exchanged.
        auto slabT = [&]() { return make_view(arena.alloc(nc * ov), {nc, ov}); };
        TView MT = slabT(), N1T = slabT();
        auto load_rows = [&](double alpha, const TView& blk, const char* spec, const TView& dst) {
            TView src = slice(blk, 1, b0, b1);                       // only the b values the slab touches
            if (b0 * o == c0 && b1 * o == c1) {
                permute(alpha, src, spec, 0.0, make_view(dst.p, {b1 - b0, o, v, o}), "bjck");
            } else {
                ArenaScope s2(arena);
                TView tmp = make_view(arena.alloc((b1 - b0) * o * ov), {(b1 - b0) * o, ov});
                permute(alpha, src, spec, 0.0, make_view(tmp.p, {b1 - b0, o, v, o}), "bjck");
                copy(slice(tmp, 0, c0 - b0 * o, c1 - b0 * o), dst);
            }
        };
        load_rows(2.0, Viabj, "kbcj", MT);                                                   // MT = (2 Wd)^T
        load_rows(-1.0, Viajb, "kbjc", N1T);                                                 // N1T = -(UdT)^T
        axpby(1.0, N1T, 1.0, MT);
        ring_xd_ = 0.0;
        auto rowsOf = [&](const TView& m) { return slice(m, 0, c0, c1); };
        if (quad) {
            contract(0.5, rowsOf(Ttd), "ny", pairm(get_static("Ld")), "yx", 1.0, MT, "nx");
            contract(0.5, rowsOf(Tx), "ny", pairm(get_static("Vx")), "yx", 1.0, N1T, "nx");
        } else {
            contract(1.0, rowsOf(Ttd), "ny", pairm(get_static("Vd")), "yx", 1.0, MT, "nx");
        }
        contract(1.0, N1T, "nk", Tx, "km", 0.0, ETx, "nm");                                  // (Ex_x)^T, rows = this rank's columns
        contract(0.5, MT, "nk", Ttd, "km", 0.5, ETd, "nm", "", &ETx);                        // (Ex_d)^T
    } else {
    TView M = slab(), N1 = slab();
    // One rank with all columns: the two builds are independent, and so are the two applications once the D-term no longer
    // carries the half of the C-term (Ex_d = D + Ex_x / 2: the assembly reads Ex_x in both placements, `ring_xd_`) — each pair
    // is ONE batched launch of 2 x tiles (the second operand set addressed through pointer differences as batch strides).
    // At (20,80) a product is 169 tiles of 128 x 128 on 256 CUs, 3-way k-split with its reduction; a pair is 338 tiles:
    // 256 whole ones and an 82-tile tail.  M is kept halved (M_h = Wd - UdT / 2 + Ld Tt_d / 4) so that both products of a
    // pair share alpha.
    // Only where a single product under-fills the chip (< 512 tiles of 128 x 128): a big product loses nothing on its own,
    // and the assembly's two extra reads of Ex_x (1.6 GB at (50,200)) would be paid for nothing.
    const int64_t ring_tiles = ((ov + 127) / 128) * ((ov + 127) / 128);
    const bool paired = nc == ov && ring_tiles < 512;
    ring_xd_ = paired ? 0.5 : 0.0;
    auto pair_gemm = [&](double alpha, const double* A0, const double* A1, const double* B0, const double* B1, double beta,
                         double* C0, double* C1) {
        auto diff = [](const double* x, const double* y) {
            return static_cast<int64_t>((reinterpret_cast<intptr_t>(x) - reinterpret_cast<intptr_t>(y)) / 8);
        };
        dev::Gemm g{};
        g.M = ov; g.N = ov; g.K = ov; g.alpha = alpha; g.beta = beta;
        g.A = A0; g.a_sm = ov; g.a_sk = 1;          // symmetric pair matrices on the left, K-contiguous
        g.B = B0; g.b_sk = ov; g.b_sn = 1;
        g.C = C0; g.ldc = ov;
        g.nb1 = 2; g.nb2 = 1;
        g.a_b1 = diff(A1, A0); g.b_b1 = diff(B1, B0); g.c_b1 = diff(C1, C0);
        g.a_b2 = g.b_b2 = g.c_b2 = 0;
        g.splitk_ws = splitk_ws_;
        g.splitk_ws_doubles = splitk_doubles_;
        dev::gemm(g, stream);
        stats.gemm_calls++;
        stats.gemm_flops += 4.0 * double(ov) * double(ov) * double(ov);
    };
    auto dense = [](const TView& t) {
        int64_t st = 1;
        for (int i = t.rank - 1; i >= 0; --i) { if (t.st[i] != st) return false; st *= t.dim[i]; }
        return true;
    };
    if (nc == ov && dense(Viabj) && dense(Viajb) && (size_t)no * 33 * sizeof(double) <= 64 * 1024) {
        // both operands in one pass over the two blocks (two permutations and an axpby before: 5 reads / 3 writes of (ov)^2)
        dev::ring_operands(Viabj.p, Viajb.p, M.p, N1.p, paired ? 1.0 : 2.0, paired ? 0.5 : 1.0, no, nv, stream);
        stats.permute_calls++;
        stats.permute_bytes += 8.0 * 4.0 * double(ov) * double(ov);
    } else {
        load_cols(paired ? 1.0 : 2.0, Viabj, "kbcj", M);                                      // M = 2 Wd  (paired: Wd)
        load_cols(-1.0, Viajb, "kbjc", N1);                                                   // N1 = -UdT
        axpby(paired ? 0.5 : 1.0, N1, 1.0, M);                                               // M = 2 Wd - UdT  (paired: half of it)
    }
    // The small V.T sums S_ac = sum_dkl Tt_adkl V_lkdc, S_ki = sum_cdl Tt_cdil V_lkdc (X_ac, X_ki, ccsd.py:434 / :436) are
    // partial traces of the builds: tr(Vd Tt_d) for DCSD, (3 tr(Vx Tx) + tr(Ld Tt_d)) / 4 for CCSD — read off the
    // accumulators before and after the products (all columns on this rank only)
    const double cz = (quad ? 0.5 : 1.0) * (paired ? 2.0 : 1.0), cu = 1.5;   // (tr after - tr before) x these = the contribution to S
    if (traces) {
        ensure_xs();
        xs_oo_tag_.clear();
        xs_vv_tag_.clear();
        dev::pair_traces(M.p, nc, -cz, 0.0, xs_vv_, xs_oo_, no, nv, stream, quad ? N1.p : nullptr, -cu);
    }
    if (quad && paired) {
        pair_gemm(0.5, get_static("Ldh"), get_static("Vx"), Ttd.p, Tx.p, 1.0, M.p, N1.p);    // M_h += Ld Tt_d / 4, N1 += Vx Tx / 2
    } else if (quad) {
        contract(0.5, pairm(get_static("Ld")), "xy", cols(Ttd), "yn", 1.0, M, "xn");         // M = 2 Wd - UdT + Ld Tt_d / 2
        contract(0.5, pairm(get_static("Vx")), "xy", cols(Tx), "yn", 1.0, N1, "xn");         // N1 = -(UdT - Vx Tx / 2)
    } else {
        contract(paired ? 0.5 : 1.0, pairm(get_static("Vd")), "xy", cols(Ttd), "yn", 1.0, M, "xn");   // M = 2 Wd - UdT + Vd Tt_d
    }
    if (traces) {
        dev::pair_traces(M.p, nc, cz, 1.0, xs_vv_, xs_oo_, no, nv, stream, quad ? N1.p : nullptr, cu);
        xs_oo_tag_.set(t2, 0, 1);
        xs_vv_tag_.set(t2, 0, 1);
    }
    if (paired) {
        pair_gemm(1.0, Tx.p, Ttd.p, N1.p, M.p, 0.0, ETx.p, ETd.p);                           // Ex_x = -Xc,  D = Tt_d M_h
    } else if (nc == ov) {
        // all columns on this rank: nothing is exchanged, and since only Ex + Ex^T enters R (residual_assemble) the slab
        // need not be transposed — Ex_x = Tx N1 and Ex_d = Tt_d M / 2 + Ex_x / 2 with the K-contiguous symmetric amplitudes
        // on the left: the same operand layout as the two builds (the LDS-DMA variant with the better L2 reuse: 6.4 GB of
        // fetches per launch against 13.6 GB for the transposed form, profiles/r02/bench_c3_pmc_hbm_traffic.csv)
        contract(1.0, Tx, "mk", N1, "kn", 0.0, ETx, "mn");                                   // Ex_x = -Xc
        contract(0.5, Ttd, "mk", M, "kn", 0.5, ETd, "mn", "", &ETx);                         // Ex_d = D-term - Xc / 2
    } else {
        contract(1.0, N1, "kn", Tx, "mk", 0.0, ETx, "nm");                                   // (Ex_x)^T, rows = this rank's columns
        contract(0.5, M, "kn", Ttd, "mk", 0.5, ETd, "nm", "", &ETx);                         // (Ex_d)^T
    }
    }
    {
        // :232  Ex[a,b,i,j] -= X_ki T[a,b,k,j]  ->  ET[(b,j),(a,i)] -= sum_k Td[(b,j),(a,k)] X_ki   (Td symmetric)
        ArenaScope s2(arena);
        TView Foo = slice(slice(F, 0, 0, o), 1, 0, o);
        TView Xoo = make_view(arena.alloc(o * o), {o, o});
        copy(Foo, Xoo);
        if (P) {                 // the V.T part was summed over the ranks (slab_prepare)
            axpby(1.0, make_view(const_cast<double*>(P), {o, o}), 1.0, Xoo);
        } else if (traces) {
            axpby(w, make_view(xs_oo_, {o, o}), 1.0, Xoo);                                   // :215-220
        } else {
            TView Tp = make_view(arena.alloc(o * o * v * v), {v, v, o, o});
            permute(1.0, make_view(Ttd.p, {v, o, v, o}), "cidl", 0.0, Tp, "cdli");
            contract(w, make_view(get_static("Vk3"), {v, v, o, o}), "cdlk", Tp, "cdli", 1.0, Xoo, "ki");   // :215-220
        }
        TView Trows = make_view(Td.p + c0 * ov, {nc, v, o});
        TView Erows = make_view(ETd.p, {nc, v, o});
        contract(-1.0, Trows, "nak", Xoo, "ki", 1.0, Erows, "nai");
    }
}

// Small replicated intermediates of residual_slab as K-sharded partial sums (one process per GPU; the caller all-reduces):
//   P = [ X'_ki = w sum_{cdl} Tt[c,d,i,l] V[l,k,d,c]  (o x o;  c in the rank's chunk)
//       | Jp, Jm = 2 V_klcd T_cdij pair-packed       (opp x ldp, opp x ldm;  pairs (c,d) in the rank's chunk) ]
int64_t Engine::slab_prepare_ws_doubles() const {
    const int64_t o = no, opp = o * (o + 1) / 2, opm = o * (o - 1) / 2;
    const int64_t ldp = opp + (opp & 1), ldm = std::max<int64_t>(opm + (opm & 1), 2);
    return o * o + opp * ldp + opp * ldm;
}

void Engine::slab_prepare(const double* t2, double* P, int rank, int world, unsigned flags) {
    const bool dcd = flags & 1u;
    const int64_t o = no, v = nv, npp = v * (v + 1) / 2, npm = v * (v - 1) / 2, opp = o * (o + 1) / 2, opm = o * (o - 1) / 2;
    const int64_t ldp = opp + (opp & 1), ldm = std::max<int64_t>(opm + (opm & 1), 2);
    if (world < 1 || rank < 0 || rank >= world) throw Error("slab_prepare: bad rank/world");
    dev::memset_zero(P, sizeof(double) * slab_prepare_ws_doubles(), stream);
    TView T = make_view(const_cast<double*>(t2), {v, v, o, o});
    ArenaScope scope(arena);
    ensure_xs();
    xs_oo_tag_.clear();
    {
        // X'_ki over c in [c0,c1)
        const int64_t cc = (v + world - 1) / world, c0 = std::min<int64_t>(rank * cc, v), c1 = std::min<int64_t>(c0 + cc, v);
        if (c1 > c0) {
            ArenaScope s2(arena);
            TView Tc = slice(T, 0, c0, c1);                                       // T[c,d,i,l]
            TView Tp = make_view(arena.alloc((c1 - c0) * v * o * o), {c1 - c0, v, o, o});
            permute(2.0, Tc, "cdil", 0.0, Tp, "cdli");                            // Tt[c,d,i,l] = 2 T[c,d,i,l] - T[d,c,i,l]
            permute(-1.0, slice(T, 1, c0, c1), "dcil", 1.0, Tp, "cdli");
            TView Vk3 = slice(make_view(get_static("Vk3"), {v, v, o, o}), 0, c0, c1);
            contract(1.0, Vk3, "cdlk", Tp, "cdli", 0.0, make_view(xs_oo_, {o, o}), "ki");         // :215-220
            axpby(dcd ? 0.5 : 1.0, make_view(xs_oo_, {o, o}), 0.0, make_view(P, {o, o}));
        } else {
            dev::memset_zero(xs_oo_, sizeof(double) * o * o, stream);
        }
        xs_oo_tag_.set(t2, rank, world);       // this rank's partial sum: the singles residual takes ccsd.py:434 from it
    }
    if (!static_.count("VpIjab")) {
        double* vp = static_cast<double*>(dev::dmalloc(sizeof(double) * opp * lpitch(npp)));
        double* vm = static_cast<double*>(dev::dmalloc(sizeof(double) * opp * lpitch(std::max<int64_t>(npm, 1))));
        static_["VpIjab"] = vp;
        static_["VmIjab"] = vm;
        dev::ladder_pack_V(block(P_ijab).p, vp, vm, no, nv, 0, opp, stream, lpitch(npp), lpitch(std::max<int64_t>(npm, 1)));
    }
    {
        // J over the packed pairs P(c,d) in the rank's chunk (Q(c,d) for the antisymmetric part)
        double* Sp = arena.alloc(npp * ldp);
        double* Am = arena.alloc(std::max<int64_t>(npm * ldm, 1));
        dev::ladder_pack_T(t2, nullptr, Sp, Am, no, nv, dev::PACK_ROW_HALF, ldp, ldm, stream);
        auto pitched = [&](double* p, int64_t r, int64_t c, int64_t ld) { return slice(make_view(p, {r, ld}), 1, 0, c); };
        auto cut = [&](int64_t n, int64_t& k0, int64_t& k1) {
            const int64_t c = (n + world - 1) / world;
            k0 = std::min<int64_t>(rank * c, n);
            k1 = std::min<int64_t>(k0 + c, n);
        };
        int64_t k0, k1;
        cut(npp, k0, k1);
        if (k1 > k0)
            contract(2.0, slice(packed_rows(static_["VpIjab"], opp, npp), 1, k0, k1), "rk",
                     slice(pitched(Sp, npp, opp, ldp), 0, k0, k1), "kn", 0.0, pitched(P + o * o, opp, opp, ldp), "rn");
        cut(npm, k0, k1);
        if (opm > 0 && k1 > k0)
            contract(2.0, slice(packed_rows(static_["VmIjab"], opp, npm), 1, k0, k1), "rk",
                     slice(pitched(Am, npm, opm, ldm), 0, k0, k1), "kn", 0.0, pitched(P + o * o + opp * ldp, opp, opm, ldm), "rn");
    }
}

void Engine::residual_finish(const double* f, const double* t2, const double* ETd_p, const double* ETx_p,
                             const double* L, double* r2, unsigned flags, const double* t1, const double* QK) {
    const bool dcd = flags & 1u, dressed = flags & 2u, skip_ladder = flags & 4u;
    const bool quad = !dcd;
    const int64_t o = no, v = nv, nn = n;
    const double w = quad ? 1.0 : 0.5;
    TView T = make_view(const_cast<double*>(t2), {v, v, o, o});
    TView R = make_view(r2, {v, v, o, o});
    TView F = make_view(const_cast<double*>(f), {nn, nn});
    TView Fvv = slice(slice(F, 0, o, nn), 1, o, nn);
    TView Vijab = block(P_ijab);
    ArenaScope scope(arena);
    const bool packed = L && !skip_ladder;
    const bool fused = dev::fused_pair_kernels_ok(no);
    // amplitude-side mode: V_abij is read UNDRESSED; its (reduced) T1 dressing Z + Z^T(baji) rides in Exn, see below
    const bool amp_side = t1 && packed;
    const TView Vabij_src = block(P_abij, amp_side ? false : dressed);
    if (!fused) copy(Vabij_src, R);                                                         // :185
    if (!packed) {
        ArenaScope s2(arena);
        if (fused) copy(block(P_abij, dressed), R);
        TView hole = make_view(arena.alloc(o * o * o * o), {o, o, o, o});
        copy(block(P_klij, dressed), hole);                                                 // :178
        if (quad) contract(1.0, Vijab, "klcd", T, "cdij", 1.0, hole, "klij");                // :180
        contract(1.0, hole, "klij", T, "abkl", 1.0, R, "abij");                             // :186
        if (!skip_ladder) contract(1.0, block(P_abcd, dressed), "abcd", T, "cdij", 1.0, R, "abij");   // :187
    } else if (!fused) {
        ladder_sym_unpack(L, r2, 1.0);                  // particle (:187) + hole (:175-186) ladders, pair-packed
    }
    // X_ac = f_ac - w sum Tt[a,d,k,l] V[l,k,d,c]  (:206-221);  Ex += X_ac T[c,b,i,j]  (:231)
    const bool reuse = (flags & 32u) && lay_t2_ == t2 && lay_[2];     // Tt_d of the preceding residual_slab on this t2
    const bool have_sum = (flags & 32u) && xs_vv_tag_.is(t2, 0, 1);   // ... and its S_ac (all columns on this rank)
    TView Xvv = make_view(arena.alloc(v * v), {v, v});
    copy(Fvv, Xvv);
    TView Exn;
    if (have_sum) {
        axpby(-w, make_view(xs_vv_, {v, v}), 1.0, Xvv);
        Exn = make_view(arena.alloc(o * o * v * v), {v, v, o, o});
    } else {
        TView Ttd = make_view(reuse ? lay_[2] : arena.alloc(o * o * v * v), {v, o, v, o});
        if (!reuse) {
            permute(2.0, T, "abij", 0.0, Ttd, "aibj");
            permute(-1.0, T, "baij", 1.0, Ttd, "aibj");
        }
        contract(-w, Ttd, "akdl", make_view(get_static("Vk"), {o, v, o, v}), "kdlc", 1.0, Xvv, "ac");
        // a private Tt_d is dead after X_ac and lends its storage to Exn; the kept one must survive
        Exn = make_view(reuse ? arena.alloc(o * o * v * v) : Ttd.p, {v, v, o, o});
    }
    xs_oo_tag_.clear();                                               // last reader of the iteration
    xs_vv_tag_.clear();
    contract(1.0, Xvv, "ac", T, "cbij", 0.0, Exn, "abij");
    if (amp_side) {
        if (!QK) throw Error("residual_finish: QK buffer missing");
        amplitude_side_abij(t1, QK, Exn, 0, nv, nv, false);
    }
    if (fused) {
        // R = V~_abij (or what R holds already) + ladders + Ex + Ex^T in one pass              (:185-187, :249-252)
        dev::residual_assemble(packed ? Vabij_src.p : r2, packed ? L : nullptr, Exn.p, ETd_p, ETx_p, r2, no,
                               nv, stream, ring_xd_);
        stats.permute_calls++;
        stats.permute_bytes += 8.0 * 5.5 * double(o * o * v * v);
        return;
    }
    permute(1.0, Exn, "abij", 1.0, R, "abij");
    permute(1.0, Exn, "baji", 1.0, R, "abij");
    TView ETd = make_view(const_cast<double*>(ETd_p), {v, o, v, o}), ETx = make_view(const_cast<double*>(ETx_p), {v, o, v, o});
    permute(1.0, ETd, "aibj", 1.0, R, "abij");                                              // Ex + Ex^T (:249-252)
    permute(1.0, ETd, "bjai", 1.0, R, "abij");
    permute(1.0, ETx, "ajbi", 1.0, R, "abij");
    permute(1.0, ETx, "biaj", 1.0, R, "abij");
    if (ring_xd_ != 0.0) {                                                                  // Ex_d = D + Ex_x / 2 (residual_slab)
        permute(ring_xd_, ETx, "aibj", 1.0, R, "abij");
        permute(ring_xd_, ETx, "bjai", 1.0, R, "abij");
    }
}

// hf.py:14-18 from the packed blocks: f = h + 2 V_piqi - V_piiq (i occupied)
void Engine::hf_fock_matrix(const double* h_host, double* f_host) {
    const double* dir[4];
    const double* exc[4];
    for (int tp = 0; tp < 2; ++tp)
        for (int tq = 0; tq < 2; ++tq) {
            dir[tp * 2 + tq] = block((tp << 3) | (tq << 1)).p;          // (tp, occ, tq, occ)
            exc[tp * 2 + tq] = block((tp << 3) | tq).p;                 // (tp, occ, occ, tq)
        }
    ArenaScope scope(arena);
    const int64_t nn = n;
    double* h = arena.alloc(nn * nn);
    double* f = arena.alloc(nn * nn);
    dev::memcpy_h2d(h, h_host, sizeof(double) * nn * nn, stream);
    dev::hf_fock(dir, exc, h, f, no, nv, stream);
    dev::memcpy_d2h(f_host, f, sizeof(double) * nn * nn, stream);
}

// The part of the T1 dressing of the residual that is linear in the rows a of R and enters through Ex + Ex^T(baji)
// (amplitude-side mode; exchange symmetry V_pqrs = V_qpsr).  With W_kbij = V_kbij + V_kbcj t_ci + V_kbid t_dj:
//     N[a,b,i,j] += V_abcj t_ci - t_ak (Q_kbij + W_kbij)          for a in [a0,a1)   (+ the (b,a,j,i) halves if asked)
// QK holds Q + W (ladder_t1 forms both for the (k,b) rows of a rank; all-gathered by the caller).  Q_kb carries
// the (c,d)-ket part of the bras (k,b)/(a,l); W the rest of those bras; V_abcj t_ci and its
// partner V_abid t_dj = (V_abcj t_ci)_baji are the (c,j)/(i,d) kets of the bra (a,b).  Together with the undressed
// V_abij in the assembly and the (k,l) bra inside the hole ladder this is all of V~_abij (ccsd.py:322-343).
void Engine::amplitude_side_abij(const double* t1, const double* QK, const TView& N, int64_t a0, int64_t a1,
                                 int64_t b1, bool with_partner) {
    // N is [a1 - a0][b1][o][o]: rows a in [a0,a1), columns b in [0,b1)
    const int64_t o = no, v = nv;
    TView t = make_view(const_cast<double*>(t1), {v, o});
    TView Qf = make_view(const_cast<double*>(QK), {o, v, o, o});        // Q + W, rows (k,b) plain [i][j] (ladder_t1)
    contract(-1.0, slice(t, 0, a0, a1), "ak", slice(Qf, 1, 0, b1), "kbij", 1.0, N, "abij");
    if (!with_partner) {
        // N is symmetrised by the caller (N_abij + N_baji, residual_assemble): instead of V_abcj t_ci, whose result has the
        // contracted operand's index in the middle (a transposed temporary and an accumulating permutation, 2.4 GB per
        // iteration at (50,200)), add its partner (V_bacj t_ci)_(ab)(ij swapped) = V_abic t_cj  (V_pqrs = V_qpsr), which
        // the GEMM writes in place
        contract(1.0, slice(slice(block(P_abic), 0, a0, a1), 1, 0, b1), "abic", t, "cj", 1.0, N, "abij");
        return;
    }
    contract(1.0, slice(slice(block(P_abci), 0, a0, a1), 1, 0, b1), "abcj", t, "ci", 1.0, N, "abij");
    // the (b,a,j,i) halves for the same rows a (pair-sharded tail): -t_bk (Q+W)[k,a,j,i] and
    // (V_bacᵢ t_cj =) V_abic t_cj, the latter straight from the block with the roles of the kets exchanged
    contract(-1.0, slice(t, 0, 0, b1), "bk", slice(Qf, 1, a0, a1), "kaji", 1.0, N, "abij");
    contract(1.0, slice(slice(block(P_abic), 0, a0, a1), 1, 0, b1), "abic", t, "cj", 1.0, N, "abij");
}

void Engine::pair_chunk(int rank, int world, int64_t& r0, int64_t& r1) const {
    if (world < 1 || rank < 0 || rank >= world) throw Error("pair_chunk: bad rank/world");
    const int64_t npp = static_cast<int64_t>(nv) * (nv + 1) / 2, c = (npp + world - 1) / world;
    r0 = std::min<int64_t>(rank * c, npp);
    r1 = std::min<int64_t>(r0 + c, npp);
}

static int a_of_pair_row(int64_t r) {
    int64_t a = static_cast<int64_t>((std::sqrt(8.0 * static_cast<double>(r) + 1.0) - 1.0) * 0.5);
    while (a * (a + 1) / 2 > r) --a;
    while ((a + 1) * (a + 2) / 2 <= r) ++a;
    return static_cast<int>(a);
}

// The replicated remainder of residual_finish reduced to what the pairs of one rank need: X_ac (v x v, cheap) is
// still formed by everyone; the X_ac.T / t.Q_kb combination N'_ab = N_ab + N_ba^T only for the rows a of the rank's
// pairs; the assembly only for its pairs.  Needs the fused assembly kernel (o x o tile in LDS).
void Engine::residual_finish_pairs(const double* f, const double* t2, const double* ETd_p, const double* ETx_p,
                                   const double* L, double* Rc, unsigned flags, const double* t1, const double* QK,
                                   int rank, int world, const double* Xvv_in) {
    const bool dcd = flags & 1u, dressed = flags & 2u;
    const int64_t o = no, v = nv, nn = n;
    const double w = dcd ? 0.5 : 1.0;
    if (!L) throw Error("residual_finish_pairs: L is required");
    if ((t1 == nullptr) != (QK == nullptr)) throw Error("residual_finish_pairs: t1 and QK must be given together");
    if (!dev::fused_pair_kernels_ok(no)) throw Error("residual_finish_pairs: nocc too large for the fused assembly");
    int64_t r0, r1;
    pair_chunk(rank, world, r0, r1);
    if (r1 <= r0) {
        xs_oo_tag_.clear();
        xs_vv_tag_.clear();
        return;
    }
    const int a0 = a_of_pair_row(r0), a1 = a_of_pair_row(r1 - 1) + 1;
    const int64_t na = a1 - a0;
    TView T = make_view(const_cast<double*>(t2), {v, v, o, o});
    TView F = make_view(const_cast<double*>(f), {nn, nn});
    TView Fvv = slice(slice(F, 0, o, nn), 1, o, nn);
    ArenaScope scope(arena);
    // X_ac = f_ac - w sum Tt[a,d,k,l] V[l,k,d,c]  (ccd.py:206-221): given (all-reduced xvv_partial) or formed here
    TView Xvv = Xvv_in ? make_view(const_cast<double*>(Xvv_in), {v, v}) : make_view(arena.alloc(v * v), {v, v});
    if (!Xvv_in) {
        ArenaScope s2(arena);
        TView Ttd = make_view(arena.alloc(o * o * v * v), {v, o, v, o});
        permute(2.0, T, "abij", 0.0, Ttd, "aibj");
        permute(-1.0, T, "baij", 1.0, Ttd, "aibj");
        copy(Fvv, Xvv);
        contract(-w, Ttd, "akdl", make_view(get_static("Vk"), {o, v, o, v}), "kdlc", 1.0, Xvv, "ac");
    }
    // N'[a,b,i,j] = X_ac T[c,b,i,j] + X_bc T[c,a,j,i] - t_ak Q[k,b,i,j] - t_bk Q[k,a,j,i]   for a in [a0,a1)   (:231)
    // N'[a,b,i,j] = N[a,b,i,j] + N[b,a,j,i] for a in [a0,a1), N = X_ac T[c,b,i,j] + amplitude_side_abij:
    // the (b,a,j,i) halves are written out term by term (they need row b of X / t and row a of the big operands)
    // only the columns b < a1 are needed (b <= a for every pair of the rank): [na][a1] tiles instead of [na][v]
    const int64_t nb = a1;
    TView Np = make_view(arena.alloc(na * nb * o * o), {na, nb, o, o});
    contract(1.0, slice(Xvv, 0, a0, a1), "ac", slice(T, 1, 0, nb), "cbij", 0.0, Np, "abij");
    contract(1.0, slice(Xvv, 0, 0, nb), "bc", slice(T, 1, a0, a1), "caji", 1.0, Np, "abij");
    if (t1) amplitude_side_abij(t1, QK, Np, a0, a1, nb, true);
    // V_abij is read undressed in the amplitude-side mode (CCSD); CCD/DCD have nothing to dress
    const TView Vabij = block(P_abij, t1 ? false : dressed);
    dev::residual_assemble_pairs(Vabij.p, L, Np.p, ETd_p, ETx_p, Rc, no, nv, r0, r1, a0, static_cast<int>(nb), stream, ring_xd_);
    stats.permute_calls++;
    stats.permute_bytes += 8.0 * 5.5 * double(r1 - r0) * 2.0 * double(o * o);
    xs_oo_tag_.clear();                                               // last call of the iteration
    xs_vv_tag_.clear();
}

// rows a in [a0,a1) of the T1-dressed V_abcd (ccsd.py:414-419): what a rank needs for its ladder rows.
// lower_only: only the entries with b <= a are produced (all the pair-packed ladder reads), in a-blocks of
// 32 rows with b < block end — 56 % of the traffic of the full rows at nv = 200.
void Engine::dress_abcd_rows(const double* t1, int a0, int a1, bool lower_only) {
    if (a0 < 0 || a1 > nv || a0 > a1) throw Error("dress_abcd_rows: bad range");
    if (a0 == a1) return;
    TView t = make_view(const_cast<double*>(t1), {(int64_t)nv, (int64_t)no});
    TView full = block_view(ensure_dressed(P_abcd), P_abcd);
    ArenaScope scope(arena);
    TView oth = block_view(arena.alloc(block_size(P_iabc)), P_iabc);
    const TView raw_iabc = block(P_iabc);
    contract(-1.0, t, "qy", block(P_ijab), "xyrs", 1.0, oth, "xqrs", "x", &raw_iabc);                 // dressed iabc (:385-388)
    const int step = lower_only ? 32 : (a1 - a0);
    for (int p0 = a0; p0 < a1; p0 += step) {
        const int p1 = std::min(a1, p0 + step);
        const int64_t qmax = lower_only ? p1 : nv;
        auto cut = [&](const TView& x) { return slice(slice(x, 0, p0, p1), 1, 0, qmax); };
        TView dst = cut(full);
        const TView raw = cut(block(P_abcd));
        // dst[p,q,r,s] = V[p,q,r,s] - sum_x t[q,x] V_aibc[p,x,r,s]                                   (:416, + copy)
        contract(-1.0, slice(t, 0, 0, qmax), "qx", slice(block(P_aibc), 0, p0, p1), "pxrs", 1.0, dst, "pqrs", "p", &raw);
        // dst[p,q,r,s] -= sum_x t[p,x] V~_iabc[x,q,r,s]                                              (:415, :417)
        contract(-1.0, slice(t, 0, p0, p1), "px", slice(oth, 1, 0, qmax), "xqrs", 1.0, dst, "pqrs");
    }
    if (lpack_.dressed) lpack_.valid = false;
}

void Engine::ladder(const double* t2, double* r2, int a0, int a1, bool dressed, double beta) {
    const int64_t o = no, v = nv;
    if (a0 < 0 || a1 > nv || a0 > a1) throw Error("ladder: bad a-range");
    if (a0 == a1) return;
    TView T = make_view(const_cast<double*>(t2), {v, v, o, o});
    TView R = slice(make_view(r2, {v, v, o, o}), 0, a0, a1);
    TView Vs = slice(block(P_abcd, dressed), 0, a0, a1);
    contract(1.0, Vs, "abcd", T, "cdij", beta, R, "abij");                                   // ccd.py:187
}

// Pair-packed ladder.  With V_abcd = V_badc (electron exchange) and T_cdij = T_dcji:
//   L_abij = LS_(ab)(ij) + sgn(a-b) sgn(i-j) LA_(ab)(ij),
//   LS = sum_{c>=d} (V_abcd + V_abdc) f_cd (T_cdij + T_dcij)/2,   LA = sum_{c>d} (V_abcd - V_abdc) (T_cdij - T_dcij)/2
// for a >= b, i >= j only: two GEMMs of v(v+1)/2 x v(v+-1)/2 x o(o+-1)/2 = 1/4 of the flops of ccd.py:187.
void Engine::ladder_sym(const double* t2, double* L, int64_t row0, int64_t row1, bool dressed, int hole) {
    const int64_t o = no, v = nv, npp = v * (v + 1) / 2, npm = v * (v - 1) / 2, opp = o * (o + 1) / 2,
                  opm = o * (o - 1) / 2;
    if (row0 < 0 || row1 > npp || row0 > row1) throw Error("ladder_sym: bad pair-row range");
    if (hole < 0 || hole > 2) throw Error("ladder_sym: hole must be 0, 1 (CCSD) or 2 (DCSD)");
    if (row0 == row1) return;
    const int64_t rows = row1 - row0;
    if (!(lpack_.valid && lpack_.dressed == dressed && lpack_.row0 == row0 && lpack_.row1 == row1)) {
        if (!lpack_.Vp || lpack_.row1 - lpack_.row0 != rows) {
            dev::stream_sync(stream);
            dev::dfree(lpack_.Vp);
            dev::dfree(lpack_.Vm);
            lpack_.Vp = lpack_.Vm = nullptr;
            lpack_.Vp = static_cast<double*>(dev::dmalloc(sizeof(double) * rows * lpitch(npp)));
            lpack_.Vm = static_cast<double*>(dev::dmalloc(sizeof(double) * rows * lpitch(std::max<int64_t>(npm, 1))));
        }
        dev::ladder_pack_V(block(P_abcd, dressed).p, lpack_.Vp, lpack_.Vm, nv, nv, row0, row1, stream, lpitch(npp), lpitch(std::max<int64_t>(npm, 1)));
        stats.permute_calls++;
        stats.permute_bytes += 8.0 * 2.0 * double(rows) * double(v * v);
        lpack_.row0 = row0; lpack_.row1 = row1; lpack_.dressed = dressed; lpack_.valid = true;
    }
    ArenaScope scope(arena);
    // even pitches (zero pad column / pad row where the pair index is a GEMM K index): every operand qualifies for
    // 16-byte loads and the LDS-DMA kernel also when o(o+1)/2 is odd — (30,120): 465, (50,200): 1275
    const int64_t ldp = opp + (opp & 1), ldm = std::max<int64_t>(opm + (opm & 1), 2);
    auto pitched = [&](double* p, int64_t r, int64_t c, int64_t ld) { return slice(make_view(p, {r, ld}), 1, 0, c); };
    double* Sp = arena.alloc(npp * ldp);
    double* Am = arena.alloc(std::max<int64_t>(npm * ldm, 1));
    dev::ladder_pack_T(t2, nullptr, Sp, Am, no, nv, dev::PACK_ROW_HALF, ldp, ldm, stream);
    stats.permute_calls++;
    stats.permute_bytes += 8.0 * 2.0 * double(v * v * o * o);
    // L rows [row0,row1): [ LS (opp) | LA (opm) ], row length o*o
    TView Lrows = make_view(L + row0 * o * o, {rows, o * o});
    TView LS = slice(Lrows, 1, 0, opp), LA = slice(Lrows, 1, opp, o * o);
    TView SpT = pitched(Sp, npp, opp, ldp), AmT = pitched(Am, npm, opm, ldm);
    contract(1.0, packed_rows(lpack_.Vp, rows, npp), "rk", SpT, "kn", 0.0, LS, "rn");
    if (opm > 0) {
        if (npm > 0) contract(1.0, packed_rows(lpack_.Vm, rows, npm), "rk", AmT, "kn", 0.0, LA, "rn");
        else zero(LA);
    }
    if (!hole) return;
    // ---- hole ladder (ccd.py:175-186) in the same pair-packed rows:  HL_abij = sum_kl I_klij T_abkl,
    // I = V~_klij (+ sum_cd V_klcd T_cdij for CCSD).  I_klij = I_lkji, so with S/A = (T_abkl +- T_bakl)/2:
    //   HLS[(a>=b),(i>=j)] = sum_{k>=l} g_kl S_abkl (I_klij + I_lkij),  HLA[(a>b),(i>j)] = sum_{k>l} A_abkl (I_klij - I_lkij)
    // and (I_klij +- I_lkij)/2 = pack(V~_klij) + sum_{c>=d} (V_klcd +- V_kldc) (f_cd S | A)_cdij.
    double* Ip = arena.alloc(ldp * ldp);
    double* Im = arena.alloc(ldp * ldm);
    if (ldp > opp) {
        dev::memset_zero(Ip + opp * ldp, sizeof(double) * ldp, stream);
        dev::memset_zero(Im + opp * ldm, sizeof(double) * ldm, stream);
    }
    dev::ladder_pack_T(block(P_klij, dressed).p, nullptr, Ip, Im, no, no, dev::PACK_AM_PROWS, ldp, ldm, stream);
    TView Ipv = pitched(Ip, opp, opp, ldp), Imv = pitched(Im, opp, opm, ldm);        // the defined part
    TView IpK = pitched(Ip, ldp, opp, ldp), ImK = pitched(Im, ldp, opm, ldm);        // with the zero pad row
    if (hole == 1) {
        if (!static_.count("VpIjab")) {      // static per solve: dressed ijab == undressed ijab
            double* vp = static_cast<double*>(dev::dmalloc(sizeof(double) * opp * lpitch(npp)));
            double* vm = static_cast<double*>(dev::dmalloc(sizeof(double) * opp * lpitch(std::max<int64_t>(npm, 1))));
            static_["VpIjab"] = vp;
            static_["VmIjab"] = vm;
            dev::ladder_pack_V(block(P_ijab).p, vp, vm, no, nv, 0, opp, stream, lpitch(npp), lpitch(std::max<int64_t>(npm, 1)));
        }
        contract(2.0, packed_rows(static_["VpIjab"], opp, npp), "rk", SpT, "kn", 2.0, Ipv, "rn");
        if (opm > 0 && npm > 0) contract(2.0, packed_rows(static_["VmIjab"], opp, npm), "rk", AmT, "kn", 2.0, Imv, "rn");
        else if (opm > 0) axpby(2.0, Imv, 0.0, Imv);
    } else {
        axpby(2.0, Ipv, 0.0, Ipv);
        if (opm > 0) axpby(2.0, Imv, 0.0, Imv);
    }
    double* SpR = arena.alloc(npp * ldp);
    double* AmR = arena.alloc(npp * ldp);
    dev::ladder_pack_T(t2, nullptr, SpR, AmR, no, nv, dev::PACK_COL_HALF | dev::PACK_AM_PROWS | dev::PACK_AM_PCOLS, ldp, ldp, stream,
                       row0, row1);                   // only the rows this call multiplies
    stats.permute_calls += 2;
    stats.permute_bytes += 8.0 * 2.0 * double(v * v * o * o);
    // the (k,l) pair is the GEMM K index: it runs over the padded pitch (zero pad column in the rows of T, zero pad row in I)
    contract(1.0, slice(make_view(SpR, {npp, ldp}), 0, row0, row1), "rk", IpK, "kn", 1.0, LS, "rn");
    if (opm > 0) contract(1.0, slice(make_view(AmR, {npp, ldp}), 0, row0, row1), "rk", ImK, "kn", 1.0, LA, "rn");
}

// The particle ladder (ladder_sym without the hole part) for k vectors in ONE batched launch per half: the packed
// integrals V^+- are the shared left operand (batch stride 0), the packed amplitudes of vector z the right operand of
// batch z.  EOM-CCSD builds sigma for every vector of the Davidson subspace per pass (eom_ccsd.py:95-101): at (30,120) one
// vector is 228 tiles on 256 CUs, k vectors are k x 228 tiles with a k-split tail.
void Engine::ladder_sym_multi(const double* const* xs, int k, double* L_all, bool dressed) {
    const int64_t o = no, v = nv, npp = v * (v + 1) / 2, npm = v * (v - 1) / 2, opp = o * (o + 1) / 2,
                  opm = o * (o - 1) / 2;
    if (k < 1) return;
    if (!(lpack_.valid && lpack_.dressed == dressed && lpack_.row0 == 0 && lpack_.row1 == npp)) {
        if (!lpack_.Vp || lpack_.row1 - lpack_.row0 != npp) {
            dev::stream_sync(stream);
            dev::dfree(lpack_.Vp);
            dev::dfree(lpack_.Vm);
            lpack_.Vp = lpack_.Vm = nullptr;
            lpack_.Vp = static_cast<double*>(dev::dmalloc(sizeof(double) * npp * lpitch(npp)));
            lpack_.Vm = static_cast<double*>(dev::dmalloc(sizeof(double) * npp * lpitch(std::max<int64_t>(npm, 1))));
        }
        dev::ladder_pack_V(block(P_abcd, dressed).p, lpack_.Vp, lpack_.Vm, nv, nv, 0, npp, stream, lpitch(npp), lpitch(std::max<int64_t>(npm, 1)));
        stats.permute_calls++;
        stats.permute_bytes += 8.0 * 2.0 * double(npp) * double(v * v);
        lpack_.row0 = 0; lpack_.row1 = npp; lpack_.dressed = dressed; lpack_.valid = true;
    }
    ArenaScope scope(arena);
    const int64_t ldp = opp + (opp & 1), ldm = std::max<int64_t>(opm + (opm & 1), 2);
    const int64_t sp_sz = npp * ldp, am_sz = std::max<int64_t>(npm * ldm, 2);
    double* Sp = arena.alloc(k * sp_sz);
    double* Am = arena.alloc(k * am_sz);
    for (int z = 0; z < k; ++z)
        dev::ladder_pack_T(xs[z], nullptr, Sp + z * sp_sz, Am + z * am_sz, no, nv, dev::PACK_ROW_HALF, ldp, ldm, stream);
    stats.permute_calls += k;
    stats.permute_bytes += 8.0 * 2.0 * double(k) * double(v * v * o * o);
    // operands with explicit batch strides: B_z = Sp + z sp_sz ([npp][ldp], columns 0..opp), C_z = L_all + z npp o^2
    int64_t bd[3] = {k, npp, opp}, bs[3] = {sp_sz, ldp, 1}, cd[3] = {k, npp, opp}, cs[3] = {npp * o * o, o * o, 1};
    contract(1.0, packed_rows(lpack_.Vp, npp, npp), "rk", make_view(Sp, 3, bd, bs), "zkn", 0.0, make_view(L_all, 3, cd, cs), "zrn", "z");
    if (opm > 0) {
        if (npm > 0) {
            int64_t bd2[3] = {k, npm, opm}, bs2[3] = {am_sz, ldm, 1}, cd2[3] = {k, npp, opm};
            contract(1.0, packed_rows(lpack_.Vm, npp, npm), "rk", make_view(Am, 3, bd2, bs2), "zkn", 0.0,
                     make_view(L_all + opp, 3, cd2, cs), "zrn", "z");
        } else {
            int64_t cd2[3] = {k, npp, opm};
            zero(make_view(L_all + opp, 3, cd2, cs));
        }
    }
}

// A hole-ladder-shaped term sum_kl I_klij X_abkl in the pair-packed rows of L (added to what the rows hold), for
// I_klij = I_lkji and X_abkl = X_balk: the (k,l) part of ladder_sym for a caller-supplied I (EOM-CCSD: eom_ccsd.py:380-382
// — u2 against V_klij + V_klcd T_cdij, T against V_kldc u2_dcij), 1/4 of the flops of the plain v^2 o^4 product.  With y
// (exchange-symmetric, [v,v,o,o]) the term sum_cd V_klcd y_cdij is added to I on the way, pair-packed too.
void Engine::hole_ladder_packed(const double* x, const double* I, double* L, int64_t row0, int64_t row1, const double* y) {
    const int64_t o = no, v = nv, npp = v * (v + 1) / 2, npm = v * (v - 1) / 2, opp = o * (o + 1) / 2, opm = o * (o - 1) / 2;
    if (row0 < 0 || row1 > npp || row0 > row1) throw Error("hole_ladder_packed: bad pair-row range");
    if (row0 == row1) return;
    const int64_t rows = row1 - row0;
    const int64_t ldp = opp + (opp & 1), ldm = std::max<int64_t>(opm + (opm & 1), 2);
    ArenaScope scope(arena);
    auto pitched = [&](double* p, int64_t r, int64_t c, int64_t ld) { return slice(make_view(p, {r, ld}), 1, 0, c); };
    TView Lrows = make_view(L + row0 * o * o, {rows, o * o});
    TView LS = slice(Lrows, 1, 0, opp), LA = slice(Lrows, 1, opp, o * o);
    double* Ip = arena.alloc(ldp * ldp);
    double* Im = arena.alloc(ldp * ldm);
    if (ldp > opp) {
        dev::memset_zero(Ip + opp * ldp, sizeof(double) * ldp, stream);
        dev::memset_zero(Im + opp * ldm, sizeof(double) * ldm, stream);
    }
    dev::ladder_pack_T(I, nullptr, Ip, Im, no, no, dev::PACK_AM_PROWS, ldp, ldm, stream);
    TView Ipv = pitched(Ip, opp, opp, ldp), Imv = pitched(Im, opp, opm, ldm);
    if (y) {
        // I += sum_cd V_klcd y_cdij, formed pair-packed as well (the V.T part of the CCSD hole ladder, ccd.py:180, with y
        // in the place of T): [opp x npp] . [npp x opp] instead of the o^2 x v^2 x o^2 product
        if (!static_.count("VpIjab")) {
            double* vp = static_cast<double*>(dev::dmalloc(sizeof(double) * opp * lpitch(npp)));
            double* vm = static_cast<double*>(dev::dmalloc(sizeof(double) * opp * lpitch(std::max<int64_t>(npm, 1))));
            static_["VpIjab"] = vp;
            static_["VmIjab"] = vm;
            dev::ladder_pack_V(block(P_ijab).p, vp, vm, no, nv, 0, opp, stream, lpitch(npp), lpitch(std::max<int64_t>(npm, 1)));
        }
        ArenaScope s2(arena);
        double* Sp = arena.alloc(npp * ldp);
        double* Am = arena.alloc(std::max<int64_t>(npm * ldm, 1));
        dev::ladder_pack_T(y, nullptr, Sp, Am, no, nv, dev::PACK_ROW_HALF, ldp, ldm, stream);
        contract(2.0, packed_rows(static_["VpIjab"], opp, npp), "rk", pitched(Sp, npp, opp, ldp), "kn", 2.0, Ipv, "rn");
        if (opm > 0 && npm > 0)
            contract(2.0, packed_rows(static_["VmIjab"], opp, npm), "rk", pitched(Am, npm, opm, ldm), "kn", 2.0, Imv, "rn");
        else if (opm > 0) axpby(2.0, Imv, 0.0, Imv);
    } else {
        axpby(2.0, Ipv, 0.0, Ipv);
        if (opm > 0) axpby(2.0, Imv, 0.0, Imv);
    }
    double* SpR = arena.alloc(npp * ldp);
    double* AmR = arena.alloc(npp * ldp);
    dev::ladder_pack_T(x, nullptr, SpR, AmR, no, nv, dev::PACK_COL_HALF | dev::PACK_AM_PROWS | dev::PACK_AM_PCOLS, ldp, ldp, stream,
                       row0, row1);
    stats.permute_calls++;
    stats.permute_bytes += 8.0 * 2.0 * double(rows) * 2.0 * double(o * o);
    contract(1.0, slice(make_view(SpR, {npp, ldp}), 0, row0, row1), "rk", pitched(Ip, ldp, opp, ldp), "kn", 1.0, LS, "rn");
    if (opm > 0) contract(1.0, slice(make_view(AmR, {npp, ldp}), 0, row0, row1), "rk", pitched(Im, ldp, opm, ldm), "kn", 1.0, LA, "rn");
}

// hole_ladder_packed for k vectors in batched launches: L_z += rows(x_z) . (2 pack(I_z) [+ 2 V_klcd y_z, pair-packed]), all
// rows.  Entries of xs (and of Is) may all be the same pointer — the EOM-CCSD sigma builds call it once with x_z = u2_z
// against the shared V_klij + V_klcd T_cdij (eom_ccsd.py:380, :382) and once with the shared T against I_z = B5_z +
// V_kldc u2_z (:381): the shared side is packed once and enters the batched GEMM with batch stride 0.
void Engine::hole_ladder_packed_multi(const double* const* xs, const double* const* Is, const double* const* ys, int k,
                                      double* L_all) {
    const int64_t o = no, v = nv, npp = v * (v + 1) / 2, npm = v * (v - 1) / 2, opp = o * (o + 1) / 2, opm = o * (o - 1) / 2;
    if (k < 1) return;
    const int64_t ldp = opp + (opp & 1), ldm = std::max<int64_t>(opm + (opm & 1), 2);
    bool same_x = true, same_I = true;
    for (int z = 1; z < k; ++z) { same_x = same_x && xs[z] == xs[0]; same_I = same_I && Is[z] == Is[0]; }
    if (ys) same_I = false;
    const int kx = same_x ? 1 : k, ki = same_I ? 1 : k;
    ArenaScope scope(arena);
    const int64_t ip_sz = ldp * ldp, im_sz = ldp * ldm;
    double* Ip = arena.alloc(ki * ip_sz);
    double* Im = arena.alloc(ki * im_sz);
    for (int z = 0; z < ki; ++z) {
        if (ldp > opp) {       // the zero pad row of the K range
            dev::memset_zero(Ip + z * ip_sz + opp * ldp, sizeof(double) * ldp, stream);
            dev::memset_zero(Im + z * im_sz + opp * ldm, sizeof(double) * ldm, stream);
        }
        dev::ladder_pack_T(Is[z], nullptr, Ip + z * ip_sz, Im + z * im_sz, no, no, dev::PACK_AM_PROWS, ldp, ldm, stream);
    }
    int64_t id[3] = {ki, opp, opp}, is_[3] = {ip_sz, ldp, 1}, md[3] = {ki, opp, std::max<int64_t>(opm, 1)}, ms[3] = {im_sz, ldm, 1};
    TView Ipv = make_view(Ip, 3, id, is_), Imv = make_view(Im, 3, md, ms);
    if (ys) {
        if (!static_.count("VpIjab")) {
            double* vp = static_cast<double*>(dev::dmalloc(sizeof(double) * opp * lpitch(npp)));
            double* vm = static_cast<double*>(dev::dmalloc(sizeof(double) * opp * lpitch(std::max<int64_t>(npm, 1))));
            static_["VpIjab"] = vp;
            static_["VmIjab"] = vm;
            dev::ladder_pack_V(block(P_ijab).p, vp, vm, no, nv, 0, opp, stream, lpitch(npp), lpitch(std::max<int64_t>(npm, 1)));
        }
        ArenaScope s2(arena);
        const int64_t sp_sz = npp * ldp, am_sz = std::max<int64_t>(npm * ldm, 2);
        double* Sp = arena.alloc(k * sp_sz);
        double* Am = arena.alloc(k * am_sz);
        for (int z = 0; z < k; ++z)
            dev::ladder_pack_T(ys[z], nullptr, Sp + z * sp_sz, Am + z * am_sz, no, nv, dev::PACK_ROW_HALF, ldp, ldm, stream);
        stats.permute_calls += k;
        stats.permute_bytes += 8.0 * 2.0 * double(k) * double(v * v * o * o);
        int64_t bd[3] = {k, npp, opp}, bs[3] = {sp_sz, ldp, 1};
        contract(2.0, packed_rows(static_["VpIjab"], opp, npp), "rk", make_view(Sp, 3, bd, bs), "zkn", 2.0, Ipv, "zrn", "z");
        if (opm > 0 && npm > 0) {
            int64_t bd2[3] = {k, npm, opm}, bs2[3] = {am_sz, ldm, 1};
            contract(2.0, packed_rows(static_["VmIjab"], opp, npm), "rk", make_view(Am, 3, bd2, bs2), "zkn", 2.0, Imv, "zrn", "z");
        } else if (opm > 0) {
            axpby(2.0, Imv, 0.0, Imv);
        }
    } else {
        axpby(2.0, Ipv, 0.0, Ipv);
        if (opm > 0) axpby(2.0, Imv, 0.0, Imv);
    }
    const int64_t xr_sz = npp * ldp;
    double* SpR = arena.alloc(kx * xr_sz);
    double* AmR = arena.alloc(kx * xr_sz);
    for (int z = 0; z < kx; ++z)
        dev::ladder_pack_T(xs[z], nullptr, SpR + z * xr_sz, AmR + z * xr_sz, no, nv,
                           dev::PACK_COL_HALF | dev::PACK_AM_PROWS | dev::PACK_AM_PCOLS, ldp, ldp, stream, 0, npp);
    stats.permute_calls += kx;
    stats.permute_bytes += 8.0 * 2.0 * double(kx) * double(npp) * 2.0 * double(o * o);
    // L_z[r][n] += sum_k' X_z[r][k'] I_z[k'][n], K over the padded pitch (zero pad column of X, zero pad row of I)
    int64_t xd[3] = {kx, npp, ldp}, xst[3] = {xr_sz, ldp, 1};
    int64_t ikd[3] = {ki, ldp, opp}, ikm[3] = {ki, ldp, std::max<int64_t>(opm, 1)};
    int64_t cd[3] = {k, npp, opp}, cd2[3] = {k, npp, std::max<int64_t>(opm, 1)}, cs[3] = {npp * o * o, o * o, 1};
    auto drop = [](const TView& t) { return t.dim[0] == 1 ? std::string() : std::string("z"); };
    TView XS = make_view(SpR, 3, xd, xst), XA = make_view(AmR, 3, xd, xst);
    TView IKp = make_view(Ip, 3, ikd, is_), IKm = make_view(Im, 3, ikm, ms);
    auto spec = [&](const TView& t, const char* two) { return drop(t) + two; };
    auto squeeze = [&](const TView& t) {
        if (t.dim[0] != 1) return t;
        int64_t d2[2] = {t.dim[1], t.dim[2]}, s2[2] = {t.st[1], t.st[2]};
        return make_view(t.p, 2, d2, s2);
    };
    contract(1.0, squeeze(XS), spec(XS, "rk").c_str(), squeeze(IKp), spec(IKp, "kn").c_str(), 1.0, make_view(L_all, 3, cd, cs), "zrn", "z");
    if (opm > 0)
        contract(1.0, squeeze(XA), spec(XA, "rk").c_str(), squeeze(IKm), spec(IKm, "kn").c_str(), 1.0,
                 make_view(L_all + opp, 3, cd2, cs), "zrn", "z");
}

// T1 dressing of the ladders on the amplitude side.  With X_a^p = delta_ap - t_ak delta_pk the dressed ladder and the
// (c,d)-ket part of V~_abij are  sum_pq X_a^p X_b^q sum_cd V_pqcd tau_cdij,  tau = T + t1 t1 (exchange-symmetric like T):
//   (p,q) = (a,b): the pair-packed ladder with the UNDRESSED V_abcd, packed once per solve          -> L rows
//   (p,q) = (k,b), (a,l): -t_ak Q_kbij - t_bl Q_laji,  Q_kbij = sum_cd V_kbcd tau_cdij (cd pair-packed) -> QK rows
//   (p,q) = (k,l): t_ak t_bl (V~_klij + V_klcd T_cdij), i.e. the hole ladder (ccd.py:175-186) taken with tau_abkl
//                  instead of T_abkl also covers the (k,l)-bra part of V~_abij                      -> L rows
// V~_abij must therefore be dressed in its reduced form (dress_V bit 16), V~_klij in full; V_abcd is never dressed:
// no o v^4 work and no second copy of V_abcd per iteration.
// Does the bra dressing of the packed V_abcd beat the Q_kb products?  The answer must be the same on every rank (a rank
// that dresses its rows expects QK without Q from all the others), so the model looks at the whole problem — both
// costs shrink with the number of ranks alike: the dressing streams V in and W out (16 B per element) at ~4 TB/s
// effective; the Q products run 2 ov (npp opp + npm opm) flops at ~70 TFLOP/s.  PYMES_LADDER_DRESS=0/1 overrides it.
bool Engine::bra_dress_pays() const {
    const int64_t o = no, v = nv, npp = v * (v + 1) / 2, npm = v * (v - 1) / 2, opp = o * (o + 1) / 2, opm = o * (o - 1) / 2;
    if (!dev::ladder_dress_ok(no, nv) || dress_off_) return false;
    if (const char* e = getenv("PYMES_LADDER_DRESS")) return atoi(e) != 0;
    const double t_dress = 16.0 * double(npp) * double(npp + npm) / 4.0e12 + 20e-6;
    const double t_q = 2.0 * double(o * v) * (double(npp) * double(opp) + double(npm) * double(opm)) / 70e12;
    return t_dress < 0.8 * t_q;
}

void Engine::ladder_t1(const double* t1, const double* t2, double* L, int64_t row0, int64_t row1, double* QK,
                       int64_t q0, int64_t q1, bool dcd, const double* J) {
    const int64_t o = no, v = nv, npp = v * (v + 1) / 2, npm = v * (v - 1) / 2, opp = o * (o + 1) / 2,
                  opm = o * (o - 1) / 2, ov = o * v;
    if (row0 < 0 || row1 > npp || row0 > row1) throw Error("ladder_t1: bad pair-row range");
    if (q0 < 0 || q1 > ov || q0 > q1) throw Error("ladder_t1: bad (k,b) row range");
    const int64_t rows = row1 - row0, qrows = q1 - q0;
    const int64_t ldp = opp + (opp & 1), ldm = std::max<int64_t>(opm + (opm & 1), 2);   // even pitches: 16-byte loads
    // ---- static packs (once per solve) ---------------------------------------------------------
    if (rows > 0 && !(lpack_.valid && !lpack_.dressed && lpack_.row0 == row0 && lpack_.row1 == row1)) {
        if (!lpack_.Vp || lpack_.row1 - lpack_.row0 != rows) {
            dev::stream_sync(stream);
            dev::dfree(lpack_.Vp);
            dev::dfree(lpack_.Vm);
            lpack_.Vp = lpack_.Vm = nullptr;
            lpack_.Vp = static_cast<double*>(dev::dmalloc(sizeof(double) * rows * lpitch(npp)));
            lpack_.Vm = static_cast<double*>(dev::dmalloc(sizeof(double) * rows * lpitch(std::max<int64_t>(npm, 1))));
        }
        dev::ladder_pack_V(block(P_abcd).p, lpack_.Vp, lpack_.Vm, nv, nv, row0, row1, stream, lpitch(npp), lpitch(std::max<int64_t>(npm, 1)));
        lpack_.row0 = row0; lpack_.row1 = row1; lpack_.dressed = false; lpack_.valid = true;
    }
    // Bra dressing of the rank's rows of the packed V_abcd instead of its share of the Q_kb products: the two rank-no updates of
    // dev::ladder_dress move 2 x 6.5 GB at (50,200) where Q_kbij = sum_cd V_kbcd tau_cdij costs 1.0e12 flops (13.9 ms);
    // the dressed copy W takes the place of V in the ladder product and QK carries the small brackets only.
    bool dress = bra_dress_pays();
    const std::string kkey = ":" + std::to_string(q0) + ":" + std::to_string(q1);
    const std::string rkey = ":" + std::to_string(row0) + ":" + std::to_string(row1);
    if (dress && rows > 0 && (!static_.count("VpKx") || !static_.count("VpDress" + rkey))) {
        // rows (x,k) of V_kxcd (x slow), pair-packed over (c,d): packed in the order of the block, rows transposed; and the
        // dressed copy W of this rank's rows.  Out of memory (2 x 1.6 GB + 2 x 1.6 GB of scratch + the size of the packed
        // rows at (50,200)): one rank falls back to the Q_kb form; among several ranks the choice must not diverge.
        const int64_t lp = lpitch(npp), lm = lpitch(std::max<int64_t>(npm, 1));
        const bool need_p = !static_.count("VpKx");
        double *tp = nullptr, *tm = nullptr, *px = nullptr, *mx = nullptr, *wp = nullptr, *wm = nullptr;
        bool ok = true;
        auto grab = [&](double*& p, int64_t n) { if (ok) { p = static_cast<double*>(dev::try_dmalloc(sizeof(double) * n)); ok = p != nullptr; } };
        if (!static_.count("VpDress" + rkey)) { grab(wp, rows * lp); grab(wm, rows * lm); }
        if (need_p) { grab(px, ov * lp); grab(mx, ov * lm); grab(tp, ov * lp); grab(tm, ov * lm); }
        if (!ok) {
            for (double* p : {tp, tm, px, mx, wp, wm}) dev::dfree(p);
            if (!(rows == npp && qrows == ov))
                throw Error("ladder_t1: out of device memory for the dressed copy of the packed V_abcd (PYMES_LADDER_DRESS=0 on every rank selects the Q_kb form)");
            dress_off_ = true;
            dress = false;
        } else {
            if (wp) { static_["VpDress" + rkey] = wp; static_["VmDress" + rkey] = wm; }
            if (need_p) {
                static_["VpKx"] = px;
                static_["VmKx"] = mx;
                dev::ladder_pack_V(block(P_iabc).p, tp, tm, 0, nv, 0, ov, stream, lp, lm);
                permute(1.0, make_view(tp, {o, v, lp}), "kxc", 0.0, make_view(px, {v, o, lp}), "xkc");
                permute(1.0, make_view(tm, {o, v, lm}), "kxc", 0.0, make_view(mx, {v, o, lm}), "xkc");
                dev::stream_sync(stream);
                dev::dfree(tp);
                dev::dfree(tm);
            }
        }
    }
    if (!dress && qrows > 0 && !static_.count("VpK" + kkey)) {
        double* vp = static_cast<double*>(dev::dmalloc(sizeof(double) * qrows * lpitch(npp)));
        double* vm = static_cast<double*>(dev::dmalloc(sizeof(double) * qrows * lpitch(std::max<int64_t>(npm, 1))));
        static_["VpK" + kkey] = vp;
        static_["VmK" + kkey] = vm;
        dev::ladder_pack_V(block(P_iabc).p, vp, vm, 0, nv, q0, q1, stream, lpitch(npp), lpitch(std::max<int64_t>(npm, 1)));          // rows (k,b) of V_kbcd as they are
    }
    if (!static_.count("VpIjab")) {
        double* vp = static_cast<double*>(dev::dmalloc(sizeof(double) * opp * lpitch(npp)));
        double* vm = static_cast<double*>(dev::dmalloc(sizeof(double) * opp * lpitch(std::max<int64_t>(npm, 1))));
        static_["VpIjab"] = vp;
        static_["VmIjab"] = vm;
        dev::ladder_pack_V(block(P_ijab).p, vp, vm, no, nv, 0, opp, stream, lpitch(npp), lpitch(std::max<int64_t>(npm, 1)));
    }
    ArenaScope scope(arena);
    auto pitched = [&](double* p, int64_t r, int64_t c, int64_t ld) { return slice(make_view(p, {r, ld}), 1, 0, c); };
    TView Lrows = make_view(L + row0 * o * o, {rows, o * o});
    TView LS = slice(Lrows, 1, 0, opp), LA = slice(Lrows, 1, opp, o * o);
    {
        // ---- particle ladder and Q_kb from tau ----------------------------------------------
        ArenaScope s2(arena);
        double* Sp = arena.alloc(npp * ldp);
        double* Am = arena.alloc(std::max<int64_t>(npm * ldm, 1));
        dev::ladder_pack_T(t2, t1, Sp, Am, no, nv, dev::PACK_ROW_HALF, ldp, ldm, stream);
        stats.permute_calls++;
        stats.permute_bytes += 8.0 * 2.0 * double(v * v * o * o);
        TView SpT = pitched(Sp, npp, opp, ldp), AmT = pitched(Am, npm, opm, ldm);
        const double* Ap = lpack_.Vp;
        const double* Am_ = lpack_.Vm;
        if (dress && rows > 0) {
            ArenaScope s3(arena);
            double* ws = arena.alloc(dev::ladder_dress_ws_doubles(no, nv));
            dev::ladder_dress(lpack_.Vp, static_["VpKx"], t1, static_["VpDress" + rkey], no, nv, lpitch(npp), row0, row1, -1.0, ws, stream);
            if (npm > 0)
                dev::ladder_dress(lpack_.Vm, static_["VmKx"], t1, static_["VmDress" + rkey], no, nv, lpitch(npm), row0, row1, 1.0, ws, stream);
            Ap = static_["VpDress" + rkey];
            Am_ = static_["VmDress" + rkey];
        }
        // the two halves of the particle ladder (and of Q_kb below) are independent products: small problems run them as
        // one grouped launch (dev::gemm_group_*), big ones keep their LDS-DMA launches
        TView Qp, QS, QA;
        if (qrows > 0 && !dress) {
            Qp = make_view(arena.alloc(qrows * o * o), {qrows, o * o});
            QS = slice(Qp, 1, 0, opp);
            QA = slice(Qp, 1, opp, o * o);
        }
        {
            GemmGroupScope grp(stream);
            if (rows > 0) {
                contract(1.0, packed_rows(const_cast<double*>(Ap), rows, npp), "rk", SpT, "kn", 0.0, LS, "rn");
                if (opm > 0 && npm > 0) contract(1.0, packed_rows(const_cast<double*>(Am_), rows, npm), "rk", AmT, "kn", 0.0, LA, "rn");
            }
            if (qrows > 0 && !dress) {
                contract(1.0, packed_rows(static_["VpK" + kkey], qrows, npp), "rk", SpT, "kn", 0.0, QS, "rn");
                if (opm > 0 && npm > 0) contract(1.0, packed_rows(static_["VmK" + kkey], qrows, npm), "rk", AmT, "kn", 0.0, QA, "rn");
            }
            grp.close();
        }
        if (rows > 0 && opm > 0 && npm <= 0) zero(LA);
        if (qrows > 0 && dress) {
            // QK = W_kbij alone (ccsd.py:322-343, the (i,j)-, (c,j)- and (i,d)-ket parts of the bras (k,b))
            TView t = make_view(const_cast<double*>(t1), {v, o});
            TView Qr = make_view(QK + q0 * o * o, {qrows, o, o});
            axpby(1.0, make_view(block(P_iajk).p + q0 * o * o, {qrows, o, o}), 0.0, Qr);
            contract(1.0, make_view(block(P_iabj).p + q0 * v * o, {qrows, v, o}), "qcj", t, "ci", 1.0, Qr, "qij", "q");
            contract(1.0, make_view(block(P_iajb).p + q0 * o * v, {qrows, o, v}), "qid", t, "dj", 1.0, Qr, "qij");
        } else if (qrows > 0) {
            // rows (k,b) in [q0,q1) of Q_kbij + W_kbij, plain [i][j]: the pair-packed product [ QS | QA ] is unpacked
            // into the exchange buffer and the small brackets of amplitude_side_abij are added for the same rows, so
            // that they are sharded with Q instead of being repeated by every rank
            if (opm > 0 && npm <= 0) zero(QA);
            dev::rows_unpack(Qp.p, QK + q0 * o * o, qrows, no, stream);
            TView t = make_view(const_cast<double*>(t1), {v, o});
            TView Qr = make_view(QK + q0 * o * o, {qrows, o, o});
            axpby(1.0, make_view(block(P_iajk).p + q0 * o * o, {qrows, o, o}), 1.0, Qr);
            contract(1.0, make_view(block(P_iabj).p + q0 * v * o, {qrows, v, o}), "qcj", t, "ci", 1.0, Qr, "qij", "q");
            contract(1.0, make_view(block(P_iajb).p + q0 * o * v, {qrows, o, v}), "qid", t, "dj", 1.0, Qr, "qij");
        }
    }
    if (rows == 0) return;
    // ---- hole ladder rows.  Ifull = V~_klij + V_klcd T_cdij (pair-packed, doubled as in ladder_sym) ------
    // The (k,l) pair is the GEMM K index here: it runs over the padded pitch ldp (zero pad column in the rows of
    // T / tau, zero pad row in I) so that both operands qualify for 16-byte loads.
    double* Ip = arena.alloc(ldp * ldp);
    double* Im = arena.alloc(ldp * ldm);
    if (ldp > opp) {
        dev::memset_zero(Ip + opp * ldp, sizeof(double) * ldp, stream);
        dev::memset_zero(Im + opp * ldm, sizeof(double) * ldm, stream);
    }
    dev::ladder_pack_T(block(P_klij, true).p, nullptr, Ip, Im, no, no, dev::PACK_AM_PROWS, ldp, ldm, stream);
    TView Ipv = pitched(Ip, opp, opp, ldp), Imv = pitched(Im, opp, opm, ldm);        // the defined part
    TView IpK = pitched(Ip, ldp, opp, ldp), ImK = pitched(Im, ldp, opm, ldm);        // with the zero pad row
    double* SpR = arena.alloc(npp * ldp);
    double* AmR = arena.alloc(npp * ldp);
    const int rflags = dev::PACK_COL_HALF | dev::PACK_AM_PROWS | dev::PACK_AM_PCOLS;
    auto rowsS = [&]() { return slice(make_view(SpR, {npp, ldp}), 0, row0, row1); };
    auto rowsA = [&]() { return slice(make_view(AmR, {npp, ldp}), 0, row0, row1); };
    if (dcd) {
        // DCSD keeps only V~_klij in the hole ladder proper (ccd.py:178), but the (k,l)-bra part still sees Ifull
        axpby(2.0, Ipv, 0.0, Ipv);
        if (opm > 0) axpby(2.0, Imv, 0.0, Imv);
        dev::ladder_pack_T(t2, nullptr, SpR, AmR, no, nv, rflags, ldp, ldp, stream, row0, row1);
        contract(1.0, rowsS(), "rk", IpK, "kn", 1.0, LS, "rn");
        if (opm > 0) contract(1.0, rowsA(), "rk", ImK, "kn", 1.0, LA, "rn");
    }
    const double bI = dcd ? 1.0 : 2.0;         // dcd: Ip already holds 2 pack(V~_klij)
    if (J) {       // 2 V_klcd T_cdij (pair-packed) was summed over the ranks (slab_prepare): [ Jp (opp x ldp) | Jm (opp x ldm) ]
        double* j = const_cast<double*>(J);
        axpby(1.0, pitched(j, opp, opp, ldp), bI, Ipv);
        if (opm > 0) axpby(1.0, pitched(j + opp * ldp, opp, opm, ldm), bI, Imv);
    } else {
        ArenaScope s2(arena);
        double* Sp = arena.alloc(npp * ldp);
        double* Am = arena.alloc(std::max<int64_t>(npm * ldm, 1));
        dev::ladder_pack_T(t2, nullptr, Sp, Am, no, nv, dev::PACK_ROW_HALF, ldp, ldm, stream);
        GemmGroupScope grp(stream);
        contract(2.0, packed_rows(static_["VpIjab"], opp, npp), "rk", pitched(Sp, npp, opp, ldp), "kn", bI, Ipv, "rn");
        if (opm > 0 && npm > 0)
            contract(2.0, packed_rows(static_["VmIjab"], opp, npm), "rk", pitched(Am, npm, opm, ldm), "kn", bI, Imv, "rn");
        grp.close();
        if (opm > 0 && npm <= 0 && !dcd) axpby(2.0, Imv, 0.0, Imv);
    }
    // CCSD: rows of tau against Ifull;  DCSD: rows of t1 t1 against Ifull (rows of T were taken above)
    dev::ladder_pack_T(dcd ? nullptr : t2, t1, SpR, AmR, no, nv, rflags, ldp, ldp, stream, row0, row1);   // this rank's rows
    stats.permute_calls += 3;
    stats.permute_bytes += 8.0 * 4.0 * double(v * v * o * o);
    GemmGroupScope grp(stream);
    contract(1.0, rowsS(), "rk", IpK, "kn", 1.0, LS, "rn");
    if (opm > 0) contract(1.0, rowsA(), "rk", ImK, "kn", 1.0, LA, "rn");
    grp.close();
}

void Engine::ladder_sym_unpack(const double* L, double* r2, double beta) {
    dev::ladder_unpack(L, r2, beta, no, nv, stream);
    stats.permute_calls++;
    stats.permute_bytes += 8.0 * 2.5 * double(nv) * nv * no * no;
}

// -----------------------------------------------------------------------------------
// ccsd.py:226-288   dressed Fock matrix, term by term (grouped; no symmetry assumed)
//   G = 2 G1 - G2,  G1[a,c] = t_bj V_iabc[j,a,b,c],  G2[a,c] = t_bj V_iabc[j,a,c,b]
//   Mm = 2 J1 - J2, J1[k,c] = t_bj V_ijab[j,k,b,c],  J2[k,c] = t_bj V_ijab[j,k,c,b]
//   L = 2 L1 - L2,  L1[k,i] = t_bj V_ijak[j,k,b,i],  L2[k,i] = t_bj V_ijka[j,k,i,b]
//   K1[i,a] = t_bj V_iabj[j,a,b,i],  K2[a,i] = t_bj V_iajb[j,a,i,b]
//   f~_ov = f_ov + 2 K1 - J2                                          (:257-258)
//   f~_oo = f_oo + L + f_ov t + Mm t                                  (:275-279)
//   f~_vv = f_vv + G - t f_ov - t Mm                                  (:282-286)
//   f~_vo = f_vo - t f_oo + f_vv t - t (f_ov t) + 2 K1^T - K2 - t L + G t - t Mm t   (:260-272)
// -----------------------------------------------------------------------------------
// The dressed Fock matrix is built in two stages.  Stage 1: the eight intermediates that contract T1 with a V block over
// (b,j) — linear in V, so a rank may sum over its chunk of the occupied index j only (one process per GPU: the partial
// buffers are all-reduced, 0.7 MB at (50,200)); W = [ G1 (v,v) | G2 (v,v) | J1 (o,v) | J2 (o,v) | L1 (o,o) | L2 (o,o) |
// K1 (o,v) | K2 (v,o) ], every piece written by ONE matrix-vector product (beta = 0): they are independent and go to the
// device as one batched launch (dev::gemv_batch_begin/end).  Stage 2: the combinations G = 2 G1 - G2, ... and their
// products with T1 and f (matrices of n^2 elements) in two small kernels (dev::fock_finish).
int64_t Engine::dress_fock_ws_doubles() const {
    const int64_t o = no, v = nv;
    return 2 * v * v + 4 * o * v + 2 * o * o;
}

void Engine::dress_fock_partial(const double* t1, double* W, int rank, int world) {
    const int64_t o = no, v = nv;
    if (world < 1 || rank < 0 || rank >= world) throw Error("dress_fock_partial: bad rank/world");
    const int64_t c = (o + world - 1) / world, j0 = std::min<int64_t>(rank * c, o), j1 = std::min<int64_t>(j0 + c, o);
    TView G1 = make_view(W, {v, v}), G2 = make_view(G1.p + v * v, {v, v}), J1 = make_view(G2.p + v * v, {o, v}),
          J2 = make_view(J1.p + o * v, {o, v}), L1 = make_view(J2.p + o * v, {o, o}), L2 = make_view(L1.p + o * o, {o, o}),
          K1 = make_view(L2.p + o * o, {o, v}), K2 = make_view(K1.p + o * v, {v, o});
    if (j1 <= j0) {
        dev::memset_zero(W, sizeof(double) * dress_fock_ws_doubles(), stream);
        return;
    }
    TView t = slice(make_view(const_cast<double*>(t1), {v, o}), 1, j0, j1);
    const bool all = (j0 == 0 && j1 == o);       // whole blocks: the planner may reuse its cached transposed copies
    auto js = [&](int pat) { return all ? block(pat) : slice(block(pat), 0, j0, j1); };
    // the two sums over the o v^3 block in one pass over it as stored (no transposed static copies of the block)
    const bool fused_g12 = dev::fock_g12_ok(nv);
    if (fused_g12) {
        ArenaScope s2(arena);
        const int ja = static_cast<int>(j0), jb = static_cast<int>(j1);
        double* ws = arena.alloc(dev::fock_g12_ws_doubles(nv, nv, jb - ja));
        dev::fock_g12(block(P_iabc).p, t1, G1.p, G2.p, no, nv, nv, ja, jb, ws, stream);
        double* ws2 = arena.alloc(dev::fock_g12_ws_doubles(nv, no, jb - ja));
        dev::fock_g12(block(P_ijab).p, t1, J1.p, J2.p, no, nv, no, ja, jb, ws2, stream);
    }
    struct Batch {
        Batch() { dev::gemv_batch_begin(); }
        ~Batch() { try { dev::gemv_batch_end(); } catch (...) {} }
    };
    {
        Batch batch;
        if (!fused_g12) {
            contract(1.0, t, "bj", js(P_iabc), "jabc", 0.0, G1, "ac");
            contract(1.0, t, "bj", js(P_iabc), "jacb", 0.0, G2, "ac");
        }
        if (!fused_g12) {
            contract(1.0, t, "bj", js(P_ijab), "jkbc", 0.0, J1, "kc");
            contract(1.0, t, "bj", js(P_ijab), "jkcb", 0.0, J2, "kc");
        }
        contract(1.0, t, "bj", js(P_ijak), "jkbi", 0.0, L1, "ki");
        contract(1.0, t, "bj", js(P_ijka), "jkib", 0.0, L2, "ki");
        contract(1.0, t, "bj", js(P_iabj), "jabi", 0.0, K1, "ia");
        contract(1.0, t, "bj", js(P_iajb), "jaib", 0.0, K2, "ai");
        dev::gemv_batch_end();           // (errors surface here; the guard only covers an exception on the way)
    }
}

void Engine::dress_fock_finish(const double* f, const double* t1, const double* W, double* fd) {
    ArenaScope scope(arena);
    dev::fock_finish(f, t1, W, fd, arena.alloc(static_cast<int64_t>(no) * no), no, nv, stream);
}

void Engine::dress_fock(const double* f, const double* t1, double* fd) {
    ArenaScope scope(arena);
    double* W = arena.alloc(dress_fock_ws_doubles());
    dress_fock_partial(t1, W, 0, 1);
    dress_fock_finish(f, t1, W, fd);
}

// X_ac = f_ac - w sum_{kdl} Tt[a,d,k,l] V[l,k,d,c] (ccd.py:206-221), the sum restricted to this rank's chunk of k (one
// process per GPU; f_ac enters on rank 0 only): v x v partial results to be all-reduced
void Engine::xvv_partial(const double* f, const double* t2, double* Xvv_p, int rank, int world, unsigned flags) {
    const bool dcd = flags & 1u;
    const int64_t o = no, v = nv, nn = n;
    if (world < 1 || rank < 0 || rank >= world) throw Error("xvv_partial: bad rank/world");
    const int64_t c = (o + world - 1) / world, k0 = std::min<int64_t>(rank * c, o), k1 = std::min<int64_t>(k0 + c, o);
    TView Xvv = make_view(Xvv_p, {v, v});
    TView F = make_view(const_cast<double*>(f), {nn, nn});
    if (rank == 0) copy(slice(slice(F, 0, o, nn), 1, o, nn), Xvv);
    else zero(Xvv);
    ensure_xs();
    xs_vv_tag_.clear();
    TView S = make_view(xs_vv_, {v, v});
    if (k1 <= k0) {
        zero(S);
    } else {
        ArenaScope scope(arena);
        TView T = make_view(const_cast<double*>(t2), {v, v, o, o});
        TView Tk = slice(T, 2, k0, k1);                                      // T[a,d,k,l], k in the chunk
        TView Ttd = make_view(arena.alloc(v * (k1 - k0) * v * o), {v, k1 - k0, v, o});
        permute(2.0, Tk, "adkl", 0.0, Ttd, "akdl");
        permute(-1.0, Tk, "dakl", 1.0, Ttd, "akdl");
        TView Vk = slice(make_view(get_static("Vk"), {o, v, o, v}), 0, k0, k1);
        contract(1.0, Ttd, "akdl", Vk, "kdlc", 0.0, S, "ac");
        axpby(dcd ? -0.5 : -1.0, S, 1.0, Xvv);
    }
    xs_vv_tag_.set(t2, rank, world);       // this rank's partial sum: the singles residual takes ccsd.py:436 from it
}

// -----------------------------------------------------------------------------------
// ccsd.py:290-421   exp(-T1) V exp(T1), block by block.
// A bra index that is virtual in the target block picks up  -t[a,k] x (occupied source),
// a ket index that is occupied in the target picks up  +(virtual source) x t[c,i]; sources
// are always undressed blocks (checked term by term by the oracle).
// Implemented as a recursion of one-index transforms: ket indices first (they shrink
// v -> o), bra indices last.
// -----------------------------------------------------------------------------------
// reduced (V_abij only, pos = {3,2,1,0}): leave out the terms with both kets virtual (V_pqcd t_ci t_dj) and the
// terms with both bras occupied (t_ak t_bl V~_klrs); ladder_t1 carries them through tau and the hole ladder.
void Engine::dressed_into(int pattern, const std::vector<int>& pos, int k, const TView& t1v, const TView& dst,
                          bool reduced, const int64_t* cut) {
    // cut = {p0, p1, q0, q1} (or null): only the range [p0,p1) of the first and [q0,q1) of the second index are
    // produced (an empty range = the whole index; a cut index must be virtual in the requested block); dst is already
    // cut.  Every block in which such an index is still that virtual index is read through the same cut; the
    // transform of the index itself (x == 0 / 1) restricts the rows of t1 instead.
    auto has = [&](int axis) { return cut && cut[2 * axis + 1] > cut[2 * axis]; };
    auto cutq = [&](TView v, int pat) {
        for (int axis = 0; axis < 2; ++axis)
            if (has(axis) && (pat >> (3 - axis) & 1)) v = slice(v, axis, cut[2 * axis], cut[2 * axis + 1]);
        return v;
    };
    if (k == 0) {
        copy(cutq(block(pattern), pattern), dst);
        return;
    }
    // same-type part: when it is the raw block the copy is fused into the product below (C = Cin + ...)
    const bool fuse_copy = (k - 1 == 0);
    if (!fuse_copy) dressed_into(pattern, pos, k - 1, t1v, dst, reduced, cut);
    const TView raw = fuse_copy ? cutq(block(pattern), pattern) : TView();
    const TView* cin = fuse_copy ? &raw : nullptr;
    if (cin) {      // the fused copy needs Cin laid out like C: a cut of the full block vs a compact temporary is not
        bool same = true;
        for (int i = 0; i < 4; ++i) same = same && (raw.dim[i] == 1 || raw.st[i] == dst.st[i]);
        if (!same) {
            copy(raw, dst);
            cin = nullptr;
        }
    }
    const int x = pos[k - 1];
    const int other = pattern ^ (1 << (3 - x));
    int depth = k - 1;                       // dressing depth of the other-type part
    if (reduced && k == 4) depth = 2;        // p -> k: no q -> l on top of it
    if (reduced && k == 2) depth = 0;        // r -> c: no s -> d on top of it
    ArenaScope scope(arena);
    TView oth;
    long memo_key = ((long)other << 8) | (long)depth;
    for (int i = 0; i < depth; ++i) memo_key = (memo_key << 3) | (long)(pos[i] + 1);
    if (depth == 0) {
        oth = cutq(block(other), other);
    } else if (!reduced && dress_memo_.count(memo_key)) {
        oth = dress_memo_[memo_key];        // formed once for all the blocks of this dress_V call (see there)
    } else {
        int64_t d[4];
        for (int i = 0; i < 4; ++i) d[i] = (other >> (3 - i) & 1) ? nv : no;
        for (int axis = 0; axis < 2; ++axis)
            if (has(axis) && (other >> (3 - axis) & 1)) d[axis] = cut[2 * axis + 1] - cut[2 * axis];
        oth = make_view(arena.alloc(d[0] * d[1] * d[2] * d[3]), 4, d, nullptr);
        dressed_into(other, pos, depth, t1v, oth, reduced, cut);
    }
    const TView tcut = (x < 2 && has(x)) ? slice(t1v, 0, cut[2 * x], cut[2 * x + 1]) : t1v;
    switch (x) {
        case 3: contract(1.0, oth, "pqrx", t1v, "xs", 1.0, dst, "pqrs", "", cin); break;
        case 2: contract(1.0, oth, "pqxs", t1v, "xr", 1.0, dst, "pqrs", "pq", cin); break;
        case 1: contract(-1.0, tcut, "qx", oth, "pxrs", 1.0, dst, "pqrs", "p", cin); break;
        case 0: contract(-1.0, tcut, "px", oth, "xqrs", 1.0, dst, "pqrs", "", cin); break;
        default: throw Error("bad index position");
    }
}

void Engine::dress_V(const double* t1, uint32_t mask, const int64_t* cut) {
    ++dress_generation_;
    TView t = make_view(const_cast<double*>(t1), {(int64_t)nv, (int64_t)no});
    for (int axis = 0; axis < 2 && cut; ++axis)
        if (cut[2 * axis] < 0 || cut[2 * axis + 1] > nv || cut[2 * axis] > cut[2 * axis + 1]) throw Error("dress_V: bad index range");
    ArenaScope memo_scope(arena);
    struct MemoGuard {
        std::map<long, TView>& m;
        ~MemoGuard() { m.clear(); }
    } memo_guard{dress_memo_};
    dress_memo_.clear();
    if ((mask >> P_klij & 1u) && (mask >> P_iabj & 1u)) {
        // V_klcj + V_klcd t_dj (ccsd.py:346-352) is also the occupied-bra source of V~_iabj (:378-383): the same
        // transform of the same block — one pass over V_ijab (0.8 GB at (50,200)) instead of two.  Both bra indices are
        // occupied, so an index range of the call does not touch it.
        const int pat = P_klij | 2;                                   // (o,o,v,o)
        const std::vector<int> pos = {3};
        int64_t d[4] = {no, no, nv, no};
        TView oth = make_view(arena.alloc(d[0] * d[1] * d[2] * d[3]), 4, d, nullptr);
        dressed_into(pat, pos, 1, t, oth, false, cut);
        dress_memo_[(((long)pat << 8 | 1L) << 3) | 4L] = oth;
    }
    for (int pat = 0; pat < 16; ++pat) {
        if (!(mask >> pat & 1u)) continue;
        TView dst = block_view(ensure_dressed(pat), pat);
        for (int axis = 0; axis < 2 && cut; ++axis) {
            if (cut[2 * axis + 1] <= cut[2 * axis]) continue;
            if (!(pat >> (3 - axis) & 1)) continue;                   // an occupied index has no range: the whole block
            dst = slice(dst, axis, cut[2 * axis], cut[2 * axis + 1]);
        }
        std::vector<int> pos;
        if (!(pat >> 0 & 1)) pos.push_back(3);   // ket s occupied
        if (!(pat >> 1 & 1)) pos.push_back(2);   // ket r occupied
        if (pat >> 2 & 1) pos.push_back(1);      // bra q virtual
        if (pat >> 3 & 1) pos.push_back(0);      // bra p virtual
        dressed_into(pat, pos, static_cast<int>(pos.size()), t, dst, pat == P_abij && (mask >> 16 & 1u), cut);
        if (pat == P_abcd && lpack_.dressed) lpack_.valid = false;
    }
}

// -----------------------------------------------------------------------------------
// ccsd.py:423-438   singles residual (dressed Fock, UNDRESSED V, explicit T1 factors)
// -----------------------------------------------------------------------------------
void Engine::singles_residual(const double* fd, const double* t1, const double* t2, double* r1) {
    const int64_t o = no, v = nv, nn = n;
    TView D = make_view(const_cast<double*>(fd), {nn, nn});
    TView Dov = slice(slice(D, 0, 0, o), 1, o, nn), Dvo = slice(slice(D, 0, o, nn), 1, 0, o);
    TView t = make_view(const_cast<double*>(t1), {v, o});
    TView T = make_view(const_cast<double*>(t2), {v, v, o, o});
    TView R = make_view(r1, {v, o});
    ArenaScope scope(arena);
    // Tt'[a,b,i,j] = 2 T[a,b,i,j] - T[a,b,j,i]   (:430), held as Tq[a,i,b,j] and P1[j,b,c,i] = Tt'[b,c,i,j]
    TView Tq = make_view(arena.alloc(o * o * v * v), {v, o, v, o});
    TView P1 = make_view(arena.alloc(o * o * v * v), {o, v, v, o});
    permute(2.0, T, "abij", 0.0, Tq, "aibj");
    permute(-1.0, T, "abji", 1.0, Tq, "aibj");
    permute(2.0, T, "bcij", 0.0, P1, "jbci");
    permute(-1.0, T, "bcji", 1.0, P1, "jbci");
    copy(Dvo, R);                                                                            // :431
    contract(1.0, Tq, "aibj", Dov, "jb", 1.0, R, "ai");                                      // :432
    contract(1.0, block(P_aibc), "ajbc", P1, "jbci", 1.0, R, "ai");                          // :433
    TView S2 = make_view(arena.alloc(o * o), {o, o});
    contract(1.0, make_view(get_static("Vjbck"), {o, v, v, o}), "jbck", P1, "jbci", 0.0, S2, "ki");
    contract(-1.0, t, "ak", S2, "ki", 1.0, R, "ai");                                         // :434
    contract(-1.0, Tq, "ajbk", block(P_ijka), "jkib", 1.0, R, "ai");                         // :435
    TView S4 = make_view(arena.alloc(v * v), {v, v});
    contract(1.0, Tq, "ajbk", block(P_ijab), "jkcb", 0.0, S4, "ac");
    contract(-1.0, S4, "ac", t, "ci", 1.0, R, "ai");                                         // :436
}

// The same residual as a K-sharded partial sum (one process per GPU; exchange-symmetric T2 only): every term contracts
// over one occupied index j together with virtual ones — (b,j), (j,b,c), (j,b,k) — so rank r sums over its chunk of j and
// the v x o partial results are all-reduced (80 KB at (50,200)); the dressed-Fock term enters on rank 0.  With
// T_abij = T_baji, Tt'[a,b,i,j] = 2 T_abij - T_abji is symmetric as an (a,i) x (b,j) matrix, so one permuted piece
// Tq[a,j,b,k] (j in the chunk) serves ccsd.py:432, :435 and :436; P1[j,b,c,i] = Tt'[b,c,i,j] serves :433 and :434.
void Engine::singles_residual_partial(const double* fd, const double* t1, const double* t2, double* r1, int rank, int world,
                                      bool reuse_layouts) {
    const int64_t o = no, v = nv, nn = n;
    if (world < 1 || rank < 0 || rank >= world) throw Error("singles_residual_partial: bad rank/world");
    const int64_t c = (o + world - 1) / world, j0 = std::min<int64_t>(rank * c, o), j1 = std::min<int64_t>(j0 + c, o);
    const int64_t nj = j1 - j0;
    TView D = make_view(const_cast<double*>(fd), {nn, nn});
    TView Dov = slice(slice(D, 0, 0, o), 1, o, nn), Dvo = slice(slice(D, 0, o, nn), 1, 0, o);
    TView t = make_view(const_cast<double*>(t1), {v, o});
    TView T = make_view(const_cast<double*>(t2), {v, v, o, o});
    TView R = make_view(r1, {v, o});
    if (rank == 0) copy(Dvo, R);                                                             // :431
    else zero(R);
    // ccsd.py:434 / :436 contract the same sums as X_ki / X_ac (ccd.py:213-220; exchange-symmetric V, T): when the
    // calls before this one have left them (residual_slab with all columns, or slab_prepare / xvv_partial: ANY partition
    // of the sums over the ranks will do, the partial residuals are added up) the two products are not repeated
    const bool have_oo = reuse_layouts && xs_oo_tag_.is(t2, rank, world), have_vv = reuse_layouts && xs_vv_tag_.is(t2, rank, world);
    if (have_oo) contract(-1.0, t, "ak", make_view(xs_oo_, {o, o}), "ki", 1.0, R, "ai");     // :434
    if (have_vv) contract(-1.0, make_view(xs_vv_, {v, v}), "ac", t, "ci", 1.0, R, "ai");     // :436
    if (nj <= 0) return;
    ArenaScope scope(arena);
    TView Tq, P1;
    const char* lp1 = "jbci";
    if (reuse_layouts && lay_t2_ == t2 && lay_[2]) {
        // Tt'[a,b,j,k] = Tt_d[(a,j),(b,k)] (exchange-symmetric T), and Tt_d is a symmetric matrix: both operands are
        // strided views of the layout residual_slab has just built — no transposition at all
        TView Ttd = make_view(lay_[2], {v, o, v, o});
        Tq = slice(Ttd, 1, j0, j1);                                                          // [a, j in chunk, b, k]
        P1 = slice(Ttd, 1, j0, j1);                                                          // read as P1[c,j,b,i]
        lp1 = "cjbi";
    } else {
        Tq = make_view(arena.alloc(v * nj * v * o), {v, nj, v, o});                          // Tq[a,j,b,k], j in the chunk
        P1 = make_view(arena.alloc(nj * v * v * o), {nj, v, v, o});                          // P1[j,b,c,i]
        permute(2.0, slice(T, 2, j0, j1), "abjk", 0.0, Tq, "ajbk");
        permute(-1.0, slice(T, 3, j0, j1), "abkj", 1.0, Tq, "ajbk");
        permute(2.0, slice(T, 3, j0, j1), "bcij", 0.0, P1, "jbci");
        permute(-1.0, slice(T, 2, j0, j1), "bcji", 1.0, P1, "jbci");
    }
    contract(1.0, Tq, "bjai", slice(Dov, 0, j0, j1), "jb", 1.0, R, "ai");                    // :432 (Tt' symmetric)
    // (these products stream 80-MB..3.2-GB blocks into tiny outputs over K = o v^2: each runs alone on the single-buffer
    // streaming kernel, which keeps 4-6 blocks per CU in flight — as ONE grouped launch on the double-buffered group kernel
    // they were slower, 178 against 112 us at (20,80), rocprofv3 round 4)
    contract(1.0, slice(block(P_aibc), 1, j0, j1), "ajbc", P1, lp1, 1.0, R, "ai");           // :433
    if (!have_oo) {
        TView S2 = make_view(arena.alloc(o * o), {o, o});
        contract(1.0, slice(make_view(get_static("Vjbck"), {o, v, v, o}), 0, j0, j1), "jbck", P1, lp1, 0.0, S2, "ki");
        contract(-1.0, t, "ak", S2, "ki", 1.0, R, "ai");                                     // :434
    }
    contract(-1.0, Tq, "ajbk", slice(block(P_ijka), 0, j0, j1), "jkib", 1.0, R, "ai");       // :435
    if (!have_vv) {
        TView S4 = make_view(arena.alloc(v * v), {v, v});
        contract(1.0, Tq, "ajbk", slice(block(P_ijab), 0, j0, j1), "jkcb", 0.0, S4, "ac");
        contract(-1.0, S4, "ac", t, "ci", 1.0, R, "ai");                                     // :436
    }
}

// -----------------------------------------------------------------------------------
// ccsd.py:176-179, ccd.py:123-124
// -----------------------------------------------------------------------------------
void Engine::cc_update(double* t, double* dt, const double* r, double shift, double delta, int rank) {
    if (rank != 2 && rank != 4) throw Error("cc_update: rank must be 2 (T1) or 4 (T2)");
    need_eps("cc_update");
    dev::cc_update(t, dt, r, eps_o, eps_v, shift, delta, no, nv, rank, stream);
}

// -----------------------------------------------------------------------------------
// ccsd.py:458-466 and ccd.py:256-262
// -----------------------------------------------------------------------------------
// ---- whole steps ------------------------------------------------------------------------------------------------------------
void Engine::ccsd_residuals(const double* f, const double* t1, const double* t2, unsigned flags, double* r1, double* r2) {
    const int64_t o = no, v = nv, ov = o * v, npp = v * (v + 1) / 2;
    if (!res_ETd_) {
        // all five or none: a failed allocation in the middle must not leave a half-initialised set behind, and a
        // launch graph that is being recorded cannot allocate (hipMalloc may synchronise)
        if (capturing_) throw Error("ccsd_residuals: the staging buffers must exist before a launch graph is recorded (run one eager pass first)");
        double* got[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        const int64_t want[5] = {(o + v) * (o + v), ov * ov, ov * ov, npp * o * o, ov * o * o};
        try {
            for (int i = 0; i < 5; ++i) got[i] = scratch_get(want[i]);
        } catch (...) {
            for (double* p : got)
                if (p) scratch_put(p);
            throw;
        }
        res_fd_ = got[0]; res_ETd_ = got[1]; res_ETx_ = got[2]; res_L_ = got[3]; res_QK_ = got[4];
    }
    const unsigned dcd = flags & 1u;                  // PYMES_DCD
    const unsigned sym = 8u | 16u;                    // PYMES_SYM_LADDER | PYMES_SYM_RINGS
    if (flags & kT1Zero) {
        residual_slab(f, t2, res_ETd_, res_ETx_, res_L_, 0, 1, dcd | sym);                                     // :171
        singles_residual_partial(f, t1, t2, r1, 0, 1, true);                                                   // :167
        residual_finish(f, t2, res_ETd_, res_ETx_, res_L_, r2, dcd | sym | 32u);                              // (PYMES_REUSE_LAYOUTS)
        return;
    }
    dress_fock(f, t1, res_fd_);                                                                                // :163
    // V_abcd is never dressed: its T1 dressing (:165, ccsd.py:414-419) is carried by tau = T2 + T1 T1 inside the ladders,
    // that of V_abij by Q_kb and two small products inside the finish; only V~_klij, V~_iajb, V~_iabj are formed
    dress_V(t1, (1u << pattern_of_name("klij")) | (1u << pattern_of_name("iajb")) | (1u << pattern_of_name("iabj")));
    residual_slab(res_fd_, t2, res_ETd_, res_ETx_, res_L_, 0, 1, dcd | sym | 2u, t1, res_QK_);                // :171 (PYMES_USE_DRESSED)
    singles_residual_partial(res_fd_, t1, t2, r1, 0, 1, true);                                                 // :167
    residual_finish(res_fd_, t2, res_ETd_, res_ETx_, res_L_, r2, dcd | sym | 2u | 32u, t1, res_QK_);
}

void Engine::ccsd_iterate(const double* f, double* t1, double* t2, unsigned flags, double shift, double delta, double* dt1,
                          double* dt2, double out[6]) {
    const int64_t o = no, v = nv;
    if (!res_r1_) {
        if (capturing_) throw Error("ccsd_iterate: the staging buffers must exist before a launch graph is recorded");
        double* r1 = scratch_get(v * o);
        try {
            res_r2_ = scratch_get(v * v * o * o);
        } catch (...) {
            scratch_put(r1);
            throw;
        }
        res_r1_ = r1;
    }
    ccsd_residuals(f, t1, t2, flags, res_r1_, res_r2_);                       // ccsd.py:161-171
    cc_update_to(t1, dt1, t1, res_r1_, shift, delta, 2);                      // :176-179
    cc_update_to(t2, dt2, t2, res_r2_, shift, delta, 4);
    energy_norms(f, t1, t2, dt2, out);                                        // :189-197
}

// ---- one process per GPU with the host program's collectives (engine.h; include/pymes_amd.h, pymes_collectives) -----------------
void Engine::set_collectives(const Collectives* c) {
    if (t2_in_flight_) {
        // a loop that was abandoned between finish and the next residuals: its last exchange is completed through the table
        // that started it (nobody reads the result) before that table goes
        t2_in_flight_ = false;
        dev::phase_sync();
        if (coll_.wait(coll_.user, t2_ticket_, stream) != 0) throw Error("collective hook: wait failed");
    }
    if (!c) {
        coll_ = Collectives();
        coll_set_ = false;
        return;
    }
    if (!c->allreduce_start || !c->allgather_start || !c->wait) throw Error("set_collectives: allreduce_start, allgather_start and wait are required");
    if (c->world < 1 || c->rank < 0 || c->rank >= c->world) throw Error("set_collectives: need 0 <= rank < world");
    coll_ = *c;
    coll_set_ = true;
}

namespace {
// the calls into the host program: a failure there (non-zero return) becomes an Error of this library
struct Hooks {
    const Engine::Collectives& c;
    void* stream;
    // (nothing of the library's own may still be queued when the host program orders a collective behind the stream)
    int64_t allreduce(double* buf, int64_t n) const {
        dev::gemm_group_sync();
        dev::phase_sync();
        int64_t t = 0;
        if (c.allreduce_start(c.user, buf, n, stream, &t) != 0) throw Error("collective hook: allreduce_start failed");
        return t;
    }
    int64_t allgather(double* buf, int64_t chunk) const {
        dev::gemm_group_sync();
        dev::phase_sync();
        int64_t t = 0;
        if (c.allgather_start(c.user, buf, chunk, stream, &t) != 0) throw Error("collective hook: allgather_start failed");
        return t;
    }
    void wait(int64_t ticket) const {
        dev::phase_sync();
        if (c.wait(c.user, ticket, stream) != 0) throw Error("collective hook: wait failed");
    }
    void mark(const char* phase) const {
        if (c.mark) {
            dev::phase_sync();
            c.mark(c.user, phase);
        }
    }
};
int64_t chunk_of(int64_t n, int world) { return (n + world - 1) / world; }
}  // namespace

// ---- owner tiles: what rank `to` reads of the rows of a ring-product matrix that rank `from` computed (pymes_amd/dist.py,
// owner_tile_plan: the same rectangles).  Rank q owns the pairs P(a,b), a in [a0,a1); its assembly reads the tiles
// [(a,.),(b,.)] and [(b,.),(a,.)], b <= a: rows [0, a1 o) x columns [a0 o, a1 o) and rows [a0 o, a1 o) x columns [0, a0 o).
std::vector<Engine::Rect> Engine::owner_tile_rects(int from, int to, int world) const {
    std::vector<Rect> out;
    if (from == to) return out;
    const int64_t o = no, ov = o * nv, npp = static_cast<int64_t>(nv) * (nv + 1) / 2;
    auto slab = [&](int64_t n, int r, int64_t& lo, int64_t& hi) {
        const int64_t c = (n + world - 1) / world;
        lo = std::min<int64_t>(r * c, n);
        hi = std::min<int64_t>(lo + c, n);
    };
    int64_t q0, q1, r0, r1;
    slab(npp, to, q0, q1);
    if (q1 <= q0) return out;
    const int64_t A0 = static_cast<int64_t>(a_of_pair_row(q0)) * o, A1 = static_cast<int64_t>(a_of_pair_row(q1 - 1) + 1) * o;
    slab(ov, from, r0, r1);
    if (std::min(r1, A1) > r0) out.push_back(Rect{r0, std::min(r1, A1), A0, A1});
    if (A0 > 0 && std::min(r1, A1) > std::max(r0, A0)) out.push_back(Rect{std::max(r0, A0), std::min(r1, A1), 0, A0});
    return out;
}
void Engine::owner_tile_sizes(int rank, int world, int64_t* send_doubles, int64_t* recv_doubles) const {
    if (world < 1 || rank < 0 || rank >= world) throw Error("owner_tile_sizes: bad rank/world");
    int64_t s = 0, r = 0;
    for (int q = 0; q < world; ++q) {
        for (const Rect& x : owner_tile_rects(rank, q, world)) s += 2 * (x.r1 - x.r0) * (x.c1 - x.c0);      // both matrices
        for (const Rect& x : owner_tile_rects(q, rank, world)) r += 2 * (x.r1 - x.r0) * (x.c1 - x.c0);
    }
    if (send_doubles) *send_doubles = std::max<int64_t>(s, 1);
    if (recv_doubles) *recv_doubles = std::max<int64_t>(r, 1);
}
int64_t Engine::owner_tiles_start(const ShardBuffers& b) {
    if (!alltoallv_ || !xs_ || !xr_) throw Error("sharded step: owner tiles need pymes_set_alltoallv and pymes_set_owner_tile_buffers");
    const int rank = coll_.rank, world = coll_.world;
    const int64_t ov = static_cast<int64_t>(no) * nv;
    std::vector<int64_t> ns(world, 0), nr(world, 0);
    int64_t off = 0;
    for (int q = 0; q < world; ++q)
        for (double* m : {b.ETd, b.ETx})
            for (const Rect& x : owner_tile_rects(rank, q, world)) {         // (small strided copies: tasks of ONE phase level)
                const int64_t rows = x.r1 - x.r0, cols = x.c1 - x.c0;
                copy(slice(slice(make_view(m, {x.r1, ov}), 0, x.r0, x.r1), 1, x.c0, x.c1), make_view(xs_ + off, {rows, cols}));
                off += rows * cols;
                ns[q] += rows * cols;
            }
    for (int p = 0; p < world; ++p)
        for (const Rect& x : owner_tile_rects(p, rank, world)) nr[p] += 2 * (x.r1 - x.r0) * (x.c1 - x.c0);
    dev::gemm_group_sync();
    dev::phase_sync();
    int64_t t = 0;
    if (alltoallv_(coll_.user, xs_, ns.data(), xr_, nr.data(), stream, &t) != 0) throw Error("collective hook: alltoallv_start failed");
    return t;
}
void Engine::owner_tiles_finish(const ShardBuffers& b) {
    const int rank = coll_.rank, world = coll_.world;
    const int64_t ov = static_cast<int64_t>(no) * nv;
    int64_t off = 0;
    for (int p = 0; p < world; ++p)
        for (double* m : {b.ETd, b.ETx})
            for (const Rect& x : owner_tile_rects(p, rank, world)) {
                const int64_t rows = x.r1 - x.r0, cols = x.c1 - x.c0;
                copy(make_view(xr_ + off, {rows, cols}), slice(slice(make_view(m, {x.r1, ov}), 0, x.r0, x.r1), 1, x.c0, x.c1));
                off += rows * cols;
            }
}

void Engine::ccsd_sharded_await(double* t2, const ShardBuffers& b) {
    if (!t2_in_flight_) return;
    if (!coll_set_) throw Error("sharded step: no collectives set (pymes_set_collectives)");
    const Hooks h{coll_, stream};
    h.wait(t2_ticket_);
    t2_in_flight_ = false;
    const int64_t o2 = static_cast<int64_t>(no) * no, c = chunk_of(static_cast<int64_t>(nv) * (nv + 1) / 2, coll_.world);
    for (int r = 0; r < coll_.world; ++r) {          // chunk r of the exchanged buffer: the compact tiles of rank r's pairs
        int64_t r0, r1;
        pair_chunk(r, coll_.world, r0, r1);
        dev::pairs_unpack(b.Tall + static_cast<int64_t>(r) * c * 2 * o2, t2, no, nv, r0, r1, stream);
    }
}

void Engine::ccsd_sharded_residuals(const double* f, double* fd, const double* t1, double* t2, const ShardBuffers& b,
                                    unsigned flags, double* rc) {
    if (!coll_set_) throw Error("sharded step: no collectives set (pymes_set_collectives)");
    if (!dev::fused_pair_kernels_ok(no)) throw Error("sharded step: nocc too large for the pair-sharded tail");
    const Hooks h{coll_, stream};
    const int rank = coll_.rank, world = coll_.world;
    const int64_t o = no, v = nv, ov = o * v, o2 = o * o;
    // flag bits of include/pymes_amd.h: PYMES_DCD 1, PYMES_USE_DRESSED 2, PYMES_SYM_LADDER 8, PYMES_SYM_RINGS 16,
    // PYMES_SLAB_RINGS_ONLY 64, PYMES_SLAB_LADDERS_ONLY 128
    const unsigned dcd = flags & 1u, kSlabRingsOnly = 64u, kSlabLaddersOnly = 128u;
    const unsigned slab = dcd | 2u | 8u | 16u;
    const bool owner = (flags & kOwnerTiles) != 0;
    h.mark("begin");
    // K-sharded partial sums, all-reduced: the T1.V intermediates of the dressed Fock matrix (ccsd.py:163, this rank's chunk of
    // j).  What needs T1 only comes first: the all-gather of the new T2 that the previous pass left in flight is awaited — and
    // unpacked into the replicated array — right before the first kernel that reads T2.
    dress_fock_partial(t1, b.W, rank, world);
    const int64_t tW = h.allreduce(b.W, dress_fock_ws_doubles());
    {
        // V~_iajb / V~_iabj only for the second-index range that this rank's column slab reads (:165); V~_klij rides in the
        // same call and shares its V_klcd t_dj intermediate with V~_iabj
        const int64_t cc = chunk_of(ov, world), c0 = std::min<int64_t>(rank * cc, ov), c1 = std::min<int64_t>(c0 + cc, ov);
        if (c1 > c0) {
            const int64_t cut[4] = {0, 0, c0 / o, (c1 + o - 1) / o};
            dress_V(t1, (1u << P_klij) | (1u << P_iajb) | (1u << P_iabj), cut);
        } else {
            dress_V(t1, 1u << P_klij);
        }
    }
    h.mark("T1-only: fock partial, dress V slab");
    ccsd_sharded_await(t2, b);
    slab_prepare(t2, b.P, rank, world, dcd);
    // P = [ X'_ki (o^2 doubles: read by the ring half) | pair-packed 2 V_klcd T_cdij (read by the ladder half) ]: two
    // all-reduces, the big one is awaited only in front of the ladders — behind the ring products
    const int64_t tX = h.allreduce(b.P, o2);
    const int64_t tJ = h.allreduce(b.P + o2, slab_prepare_ws_doubles() - o2);
    h.wait(tW);
    h.wait(tX);
    dress_fock_finish(f, t1, b.W, fd);
    h.mark("await T2, slab prepare, fock finish");
    // :171 in two halves: the ring products first, so that the all-gathers of their rows fly while the ladders — whose rows
    // of L never leave the rank — and the singles residual are computed
    residual_slab(fd, t2, b.ETd, b.ETx, b.L, rank, world, slab | kSlabRingsOnly, t1, b.QK, b.P);
    h.mark("ring products");
    // rows of the ring products: two all-gathers of the whole matrices, or ONE all-to-all of the tiles each pair owner reads
    int64_t tD = 0, tE = 0, tO = 0;
    if (owner) tO = owner_tiles_start(b);
    else {
        tD = h.allgather(b.ETd, chunk_of(ov, world) * ov);
        tE = h.allgather(b.ETx, chunk_of(ov, world) * ov);
    }
    h.wait(tJ);
    residual_slab(fd, t2, b.ETd, b.ETx, b.L, rank, world, slab | kSlabLaddersOnly, t1, b.QK, b.P);
    h.mark("ladders, Q_kb");
    const int64_t tQ = h.allgather(b.QK, chunk_of(ov, world) * o2);
    xvv_partial(fd, t2, b.Xvv, rank, world, dcd);                     // X_ac (ccd.py:206-221) over this rank's chunk of k
    const int64_t tV = h.allreduce(b.Xvv, v * v);
    singles_residual_partial(fd, t1, t2, b.R1, rank, world, true);    // :167 over this rank's chunk of the occupied index
    const int64_t tR = h.allreduce(b.R1, v * o);
    if (owner) {
        h.wait(tO);
        owner_tiles_finish(b);
    } else {
        h.wait(tD);
        h.wait(tE);
    }
    for (int64_t t : {tQ, tV, tR}) h.wait(t);
    h.mark("X_ac, singles residual, waits");
    int64_t r0, r1;
    pair_chunk(rank, world, r0, r1);
    if (r1 <= r0) dev::memset_zero(rc, sizeof(double) * 2 * o2, stream);       // a rank without pairs: one zero tile pair
    residual_finish_pairs(fd, t2, b.ETd, b.ETx, b.L, rc, slab, t1, b.QK, rank, world, b.Xvv);     // :171, this rank's pairs
    h.mark("finish + assembly (pairs)");
}

void Engine::ccd_sharded_residuals(const double* f, double* t2, const ShardBuffers& b, unsigned flags, double* rc) {
    if (!coll_set_) throw Error("sharded step: no collectives set (pymes_set_collectives)");
    if (!dev::fused_pair_kernels_ok(no)) throw Error("sharded step: nocc too large for the pair-sharded tail");
    const Hooks h{coll_, stream};
    const int rank = coll_.rank, world = coll_.world;
    const int64_t o = no, v = nv, ov = o * v, o2 = o * o;
    const unsigned dcd = flags & 1u, kSlabRingsOnly = 64u, kSlabLaddersOnly = 128u;
    const unsigned slab = dcd | 8u | 16u;                      // undressed blocks: CCD / DCD have no T1 (ccd.py:100-121)
    const bool owner = (flags & kOwnerTiles) != 0;
    h.mark("begin");
    ccsd_sharded_await(t2, b);                                 // the new T2 the previous pass left in flight
    residual_slab(f, t2, b.ETd, b.ETx, b.L, rank, world, slab | kSlabRingsOnly);
    h.mark("ring products");
    int64_t tD = 0, tE = 0, tO = 0;
    if (owner) tO = owner_tiles_start(b);
    else {
        tD = h.allgather(b.ETd, chunk_of(ov, world) * ov);
        tE = h.allgather(b.ETx, chunk_of(ov, world) * ov);
    }
    residual_slab(f, t2, b.ETd, b.ETx, b.L, rank, world, slab | kSlabLaddersOnly);      // the rows of L stay on the rank
    if (owner) {
        h.wait(tO);
        owner_tiles_finish(b);
    } else {
        h.wait(tD);
        h.wait(tE);
    }
    h.mark("ladders, waits");
    int64_t r0, r1;
    pair_chunk(rank, world, r0, r1);
    if (r1 <= r0) dev::memset_zero(rc, sizeof(double) * 2 * o2, stream);
    residual_finish_pairs(f, t2, b.ETd, b.ETx, b.L, rc, slab, nullptr, nullptr, rank, world, nullptr);
    h.mark("finish + assembly (pairs)");
}

int Engine::ccsd_sharded_finish(const double* f, const double* t1, const double* tc, const double* dtc, const ShardBuffers& b) {
    if (!coll_set_) throw Error("sharded step: no collectives set (pymes_set_collectives)");
    if (t2_in_flight_) throw Error("sharded step: the previous exchange of the amplitudes was never awaited");
    const Hooks h{coll_, stream};
    const int rank = coll_.rank, world = coll_.world;
    const int64_t o2 = static_cast<int64_t>(no) * no;
    int64_t r0, r1;
    pair_chunk(rank, world, r0, r1);
    const int64_t c = chunk_of(static_cast<int64_t>(nv) * (nv + 1) / 2, world);
    if (r1 > r0)
        dev::memcpy_d2d(b.Tall + static_cast<int64_t>(rank) * c * 2 * o2, tc, sizeof(double) * (r1 - r0) * 2 * o2, stream);
    // the energy and the norms (:189-197) come from the compact tiles: six partial sums, all-reduced where they are and copied
    // to the host on the side — nothing on this stream waits for the host.  The all-gather of the new T2 is only STARTED, after
    // that small all-reduce (a communicator runs its collectives in order: behind the 0.8-GB transfer the six numbers would
    // wait for it)
    dev::energy_norms_pairs_dev(f, t1, tc, get_static("Edir"), get_static("Eex"), dtc, no, nv, r0, r1, rank == 0 && t1 && f, b.S,
                                stream);
    h.wait(h.allreduce(b.S, 6));
    const int slot = dev::readback_start(b.S, 6, stream);
    t2_ticket_ = h.allgather(b.Tall, c * 2 * o2);
    t2_in_flight_ = true;
    h.mark("energy + norms (pairs)");
    return slot;
}

void Engine::ccsd_sharded_energy(int slot, double out[6]) {
    double r[6];
    dev::readback_wait(slot, r, 6);
    out[0] = 2.0 * r[0];
    out[1] = 2.0 * r[1];
    out[2] = -1.0 * r[2];
    out[3] = r[3];
    out[4] = r[4];
    out[5] = r[5];
}

void Engine::release_residual_buffers() {
    //  recorded graphs of ANY solver on this context replay into these buffers: they go back to the scratch pool —
    // where a later request of the same size would alias them — only when the last recorded graph is gone (graph_destroy)
    if (!graphs_.empty()) {
        release_wanted_ = true;
        return;
    }
    release_wanted_ = false;
    for (double** p : {&res_fd_, &res_ETd_, &res_ETx_, &res_L_, &res_QK_, &res_r1_, &res_r2_}) {
        if (*p) scratch_put(*p);
        *p = nullptr;
    }
}

void Engine::cc_update_to(double* t_out, double* dt, const double* t_in, const double* r, double shift, double delta,
                          int rank) {
    if (rank != 2 && rank != 4) throw Error("cc_update: rank must be 2 (T1) or 4 (T2)");
    need_eps("cc_update");
    dev::cc_update_to(t_out, dt, t_in, r, eps_o, eps_v, shift, delta, no, nv, rank, stream);
}

int Engine::energy_norms_start(const double* f, const double* t1, const double* t2, const double* dt2) {
    return dev::energy_norms_start(f, t1, t2, get_static("Edir"), get_static("Eex"), dt2, no, nv, stream);
}
void Engine::energy_norms_wait(int slot, double out[6]) {
    double r[6];
    dev::readback_wait(slot, r, 6);
    out[0] = 2.0 * r[0];
    out[1] = 2.0 * r[1];
    out[2] = -1.0 * r[2];
    out[3] = r[3];
    out[4] = r[4];
    out[5] = r[5];
}

void Engine::energy_norms(const double* f, const double* t1, const double* t2, const double* dt2, double out[6]) {
    double r[6];
    dev::energy_norms(f, t1, t2, get_static("Edir"), get_static("Eex"), dt2, no, nv, r, stream);
    out[0] = 2.0 * r[0];      // ccsd.py:465
    out[1] = 2.0 * r[1];      // :463 / ccd.py:260
    out[2] = -1.0 * r[2];     // :464 / ccd.py:261
    out[3] = r[3];
    out[4] = r[4];
    out[5] = r[5];
}

// the same over the compact tiles of this rank's pairs (one process per GPU): partial sums, to be all-reduced; the T1
// terms enter on rank 0
void Engine::energy_norms_pairs(const double* f, const double* t1, const double* tc, const double* dtc, int rank, int world,
                                double out[6]) {
    int64_t r0, r1;
    pair_chunk(rank, world, r0, r1);
    double r[6];
    dev::energy_norms_pairs(f, t1, tc, get_static("Edir"), get_static("Eex"), dtc, no, nv, r0, r1, rank == 0, r, stream);
    out[0] = 2.0 * r[0];
    out[1] = 2.0 * r[1];
    out[2] = -1.0 * r[2];
    out[3] = r[3];
    out[4] = r[4];
    out[5] = r[5];
}

void Engine::ccsd_energy(const double* f, const double* t1, const double* t2, double out[3]) {
    double r[6];
    energy_norms(f, t1, t2, nullptr, r);
    out[0] = r[0];
    out[1] = r[1];
    out[2] = r[2];
}

void Engine::ccd_energy(const double* t2, double out[2]) {
    double r[6];
    energy_norms(nullptr, nullptr, t2, nullptr, r);
    out[0] = r[1];
    out[1] = r[2];
}

}  // namespace pymes
