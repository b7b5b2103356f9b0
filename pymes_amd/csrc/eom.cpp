// EOM-CCSD sigma build and diagonals on the engine (see eom.h).  Term by term this is pymes/solver/eom_ccsd.py:268-385 with
// every V.T product that does not depend on the trial vector hoisted into per-solve intermediates; the comments name the
// reference lines.  Pair layouts (ov x ov matrices): Xd[(a,i),(b,j)] = X[a,b,i,j], Xx[(a,j),(b,i)] = X[a,b,i,j].
#include "eom.h"

#include <algorithm>
#include <cmath>
#include <initializer_list>
#include <string>

namespace pymes {

// ---- temporaries and hoisted arrays come from the engine's scratch pool: a second solve on the same context (the warm
// start of a Davidson run, the next FEAST solve) finds the buffers of the first instead of paying hipMalloc again ----------
double* EomSigma::get(int64_t doubles) { return e.scratch_get(doubles); }
void EomSigma::put(double* p) { e.scratch_put(p); }
void EomSigma::trim() { e.scratch_trim(); }
double* EomSigma::keep(int64_t doubles) {
    double* p = e.scratch_get(doubles);
    owned_.push_back(p);
    return p;
}
struct EomSigma::Tmp {
    EomSigma& s;
    double* p;
    Tmp(EomSigma& s_, int64_t n) : s(s_), p(s_.get(n)) {}
    ~Tmp() { s.put(p); }
    Tmp(const Tmp&) = delete;
    Tmp& operator=(const Tmp&) = delete;
    operator double*() const { return p; }
};

namespace {
struct Ops {       // the three engine calls every term is made of, with the output allocated by the caller
    Engine& e;
    void C(double al, const TView& A, const char* sa, const TView& B, const char* sb, double be, const TView& Cv, const char* sc,
           const char* batch = "") const {
        e.contract(al, A, sa, B, sb, be, Cv, sc, batch);
    }
    void P(double al, const TView& in, const char* si, double be, const TView& out, const char* so) const {
        e.permute(al, in, si, be, out, so);
    }
    void L(double* out, std::initializer_list<const double*> xs, std::initializer_list<double> cs, int64_t n) const {
        const double* x[8];
        double c[8];
        int m = 0;
        auto ci = cs.begin();
        for (auto xp : xs) { x[m] = xp; c[m] = *ci++; ++m; }
        dev::lincomb(out, m, x, c, n, e.stream);
    }
};
inline TView mv(const double* p, std::initializer_list<int64_t> d) { return make_view(p, d); }
}  // namespace

TView EomSigma::V(const char* name) const { return e.block(pattern_of_name(name), dressed); }

bool EomSigma::exchange_symmetric(const double* x, int64_t d0, int64_t d2) const {
    const int64_t d[4] = {d0, d0, d2, d2};
    double out[2] = {0.0, 0.0};
    dev::exchange_asymmetry(x, x, d, out, e.stream);
    return std::isfinite(out[1]) && out[0] <= 1e-13 * std::max(1.0, out[1]);      // inf / NaN entries: never "symmetric"
}

int EomSigma::flags() const {
    return (v_sym ? 1 : 0) | (t_sym ? 2 : 0) | (hole_sym ? 4 : 0) | (fused_ok ? 8 : 0) | (many_ok ? 16 : 0);
}

EomSigma::~EomSigma() {
    for (double* p : owned_) e.scratch_put(p);        // (stream-ordered: whoever gets them next enqueues behind our last kernel)
}

// ---- hoisting: everything of eom_ccsd.py:288-373 that does not depend on (u1, u2) -----------------------------------------
EomSigma::EomSigma(Engine& eng, const double* f_host, const double* t2, bool dressed_)
    : e(eng), no(eng.no), nv(eng.nv), dressed(dressed_), T(t2) {
    if (!f_host || !t2) throw Error("eom sigma: null Fock matrix / amplitudes");
    const int64_t o = no, v = nv, n = o + v, ov = o * v, ov2 = ov * ov;
    const Ops q{e};
    try {
        std::vector<double> h(static_cast<size_t>(std::max(v * v, o * v)));
        auto upload = [&](int64_t r0, int64_t nr, int64_t c0, int64_t nc, double sign, bool transposed = false) {
            for (int64_t r = 0; r < nr; ++r)
                for (int64_t c = 0; c < nc; ++c)
                    h[transposed ? c * nr + r : r * nc + c] = sign * f_host[(r0 + r) * n + c0 + c];
            double* d = keep(nr * nc);
            dev::memcpy_h2d(d, h.data(), sizeof(double) * nr * nc, e.stream);
            dev::stream_sync(e.stream);            // (h is reused)
            return d;
        };
        foo = upload(0, o, 0, o, 1.0);
        fov = upload(0, o, o, v, 1.0);
        fvv = upload(o, v, o, v, 1.0);
        const TView Vijab = V("ijab"), Viabj = V("iabj"), Viajb = V("iajb"), Vijka = V("ijka"), Vijak = V("ijak"),
                    Viabc = V("iabc"), Viajk = V("iajk"), Vklij = V("klij"), Vabcd = V("abcd");
        const TView T4 = mv(T, {v, v, o, o});
        Td = keep(ov2);
        Tx = keep(ov2);
        q.P(1.0, T4, "abij", 0.0, mv(Td, {v, o, v, o}), "aibj");
        q.P(1.0, T4, "abij", 0.0, mv(Tx, {v, o, v, o}), "ajbi");
        {
            Tmp Vd(*this, ov2), Vx(*this, ov2);
            q.P(1.0, Vijab, "klcd", 0.0, mv(Vd, {v, o, v, o}), "ckdl");          // [(c,k),(d,l)]
            q.P(1.0, Vijab, "klcd", 0.0, mv(Vx, {v, o, v, o}), "cldk");          // [(c,l),(d,k)]
            // ---- singles (eom_ccsd.py:288-308) ---------------------------------------------------------------------------
            // W1[(c,k),(a,i)] = sum_jb (2V[j,k,b,c]-V[j,k,c,b]) (2T[b,a,j,i]-T[a,b,j,i]) + 2V_iabj[k,a,c,i] - V_iajb[k,a,i,c]
            W1 = keep(ov2);
            {
                Tmp Vq(*this, ov2), Tq(*this, ov2);
                q.P(2.0, Vijab, "jkbc", 0.0, mv(Vq, {v, o, v, o}), "ckbj");
                q.P(-1.0, Vijab, "jkcb", 1.0, mv(Vq, {v, o, v, o}), "ckbj");
                q.P(2.0, T4, "baji", 0.0, mv(Tq, {v, o, v, o}), "bjai");
                q.P(-1.0, T4, "abji", 1.0, mv(Tq, {v, o, v, o}), "bjai");
                q.C(1.0, mv(Vq, {v, o, v, o}), "ckbj", mv(Tq, {v, o, v, o}), "bjai", 0.0, mv(W1, {v, o, v, o}), "ckai");
            }
            q.P(2.0, Viabj, "kaci", 1.0, mv(W1, {v, o, v, o}), "ckai");
            q.P(-1.0, Viajb, "kaic", 1.0, mv(W1, {v, o, v, o}), "ckai");
            // Gvv_s[a,c] = fvv + sum V[j,k,b,c] (-2T[b,a,j,k] + T[a,b,j,k]);  Goo_s[k,i] = -foo + sum (-2V[j,k,b,c]+V[j,k,c,b]) T[b,c,j,i]
            Gvv_s = keep(v * v);
            q.L(Gvv_s, {fvv}, {1.0}, v * v);
            q.C(-2.0, Vijab, "jkbc", T4, "bajk", 1.0, mv(Gvv_s, {v, v}), "ac");
            q.C(1.0, Vijab, "jkbc", T4, "abjk", 1.0, mv(Gvv_s, {v, v}), "ac");
            Goo_s = keep(o * o);
            q.L(Goo_s, {foo}, {-1.0}, o * o);
            q.C(-2.0, Vijab, "jkbc", T4, "bcji", 1.0, mv(Goo_s, {o, o}), "ki");
            q.C(1.0, Vijab, "jkcb", T4, "bcji", 1.0, mv(Goo_s, {o, o}), "ki");
            // ---- doubles: (V.T) pair matrices (eom_ccsd.py:352-372) --------------------------------------------------------
            M_C = keep(ov2);
            M_D = keep(ov2);
            M1 = keep(ov2);
            Ud = keep(ov2);
            M2 = keep(ov2);
            M12 = keep(ov2);
            MDU = keep(ov2);
            const TView Tdv = mv(Td, {v, o, v, o}), Txv = mv(Tx, {v, o, v, o});
            {
                // M1 = Wd' + 2 M_A - M_B with M_A = sum_kc V[k,l,c,d] T[c,a,k,i], M_B = sum_kc V[k,l,c,d] T[a,c,k,i]: the two share
                // their right operand, so the left ones are combined first — ONE (ov)^3 product instead of two
                Tmp TAB(*this, ov2);
                q.P(2.0, Tdv, "ckai", 0.0, mv(TAB, {v, o, v, o}), "aick");       // 2 T[c,a,k,i] - T[a,c,k,i] as [(a,i),(c,k)]
                q.P(-1.0, Txv, "aick", 1.0, mv(TAB, {v, o, v, o}), "aick");
                q.P(1.0, Viabj, "kaci", 0.0, mv(M1, {v, o, v, o}), "aick");      // Wd'[(a,i),(c,k)] = V_iabj[k,a,c,i]
                q.C(1.0, mv(TAB, {v, o, v, o}), "aick", mv(Vd, {v, o, v, o}), "ckdl", 1.0, mv(M1, {v, o, v, o}), "aidl");
                q.C(1.0, Tdv, "ckai", mv(Vx, {v, o, v, o}), "dlck", 0.0, mv(M_C, {v, o, v, o}), "aidl");   // sum_kc V[k,l,d,c] T[c,a,k,i]
                q.C(1.0, Txv, "aick", mv(Vx, {v, o, v, o}), "dlck", 0.0, mv(M_D, {v, o, v, o}), "aidl");   // sum_kc V[k,l,d,c] T[a,c,k,i]
            }
            q.P(1.0, Viajb, "kaic", 0.0, mv(Ud, {v, o, v, o}), "aick");          // Ud[(a,i),(c,k)] = V_iajb[k,a,i,c]
            q.L(M2, {M_D, M_C, Ud}, {1.0, -2.0, -1.0}, ov2);
            // exchange-symmetric trial doubles: M1.utd + M2.u2d + M_C.u2x = (2 M1 + M2).utd / 2 + (M_D - Ud).u2x / 2
            q.L(M12, {M1, M2}, {2.0, 1.0}, ov2);
            q.L(MDU, {M_D, Ud}, {1.0, -1.0}, ov2);
        }
        // V_kacd.T products of the u1 terms (eom_ccsd.py:334, :343, :345, :346), u-independent like the pair matrices above:
        //   WA[a,d,b,j] = sum_ck (2 V[k,a,c,d] - V[k,a,d,c]) T[c,b,k,j] - V[k,a,c,d] T[b,c,k,j],   W3[a,d,b,i] = sum_ck V[k,a,d,c] T[b,c,k,i]
        WA = keep(v * v * v * o);
        W3 = keep(v * v * v * o);
        {
            // (the first two terms share T[c,b,k,j]: 2 V[k,a,c,d] - V[k,a,d,c] is formed once — one o v^3 (ov) product less)
            Tmp Vt(*this, o * v * v * v);
            const TView VtV = mv(Vt, {o, v, v, v});
            q.P(2.0, Viabc, "kacd", 0.0, VtV, "kacd");
            q.P(-1.0, Viabc, "kadc", 1.0, VtV, "kacd");
            q.C(1.0, VtV, "kacd", T4, "cbkj", 0.0, mv(WA, {v, v, v, o}), "adbj");
        }
        q.C(-1.0, Viabc, "kacd", T4, "bckj", 1.0, mv(WA, {v, v, v, o}), "adbj");
        q.C(1.0, Viabc, "kadc", T4, "bcki", 0.0, mv(W3, {v, v, v, o}), "adbi");
        // ... merged: everything added to D is symmetrised by P(ijab,jiba) afterwards (:377), so a term X_abij may be replaced by
        // its partner X_baji — sum_d WA[a,d,b,j] u1[d,i] by sum_d WA[b,d,a,i] u1[d,j] — and the two terms become ONE product with
        // the free index of u1 innermost, WW[a,b,i,d] = WA[b,d,a,i] - W3[a,d,b,i]: [(a,b,i) x d] . u1[d,j] writes D in place, one
        // pass over one v^3 o array per build instead of two
        WW = keep(v * v * v * o);
        q.P(1.0, mv(WA, {v, v, v, o}), "bdai", 0.0, mv(WW, {v, v, o, v}), "abid");
        q.P(-1.0, mv(W3, {v, v, v, o}), "adbi", 1.0, mv(WW, {v, v, o, v}), "abid");
        // ... and the plain V_abic . u1 term (:349) has the same shape, [(a,b,i) x c] . u1[c,j]: it rides in the same product
        q.P(1.0, V("abic"), "abic", 1.0, mv(WW, {v, v, o, v}), "abic");
        for (double** w : {&WA, &W3}) {           // only WW is read by the builds
            owned_.erase(std::find(owned_.begin(), owned_.end(), *w));
            put(*w);
            *w = nullptr;
        }
        // The u2 parts of the one-index dressings (:353-354, :359-360) for an exchange-symmetric u2, whose crossed pair matrix
        // X[(a,j),(b,i)] = u2[a,b,i,j] is symmetric: both index placements of u2 are rows / columns of X itself,
        //   X_vv[a,c] = sum_kdl X[(a,k),(d,l)] (-2 V[l,k,c,d] + V[k,l,c,d]),   X_oo[k,i] = sum_dlc (-2 V[k,l,d,c] + V[k,l,c,d]) X[(d,l),(c,i)]
        // — ONE product each, X read in place, against two products over transposed copies before (2.2 ms of a 26-ms build of
        // four vectors at (30,120), rocprofv3 round 5)
        // (Bvv = columns [0, v) of BB[(k,d,l), v + o]; columns [v, v + o) hold the singles term :297 in the same form: with
        // Tt[(a,j),(b,k)] = 2 X[(a,k),(b,j)] - X[(a,j),(b,k)],  -sum_jbk Tt[(a,j),(b,k)] V[j,k,i,b] = sum_kbj X[(a,k),(b,j)] Bs[k,b,j,i],
        // Bs = -2 V[j,k,i,b] + V[k,j,i,b] — the stacked build gets X_vv AND that singles term from ONE pass over X)
        BB = keep(o * v * o * (v + o));
        {
            const TView BBv = mv(BB, {o, v, o, v + o});
            q.P(-2.0, Vijab, "lkcd", 0.0, slice(BBv, 3, 0, v), "kdlc");
            q.P(1.0, Vijab, "klcd", 1.0, slice(BBv, 3, 0, v), "kdlc");
            q.P(-2.0, Vijka, "jkib", 0.0, slice(BBv, 3, v, v + o), "kbji");
            q.P(1.0, Vijka, "kjib", 1.0, slice(BBv, 3, v, v + o), "kbji");
        }
        Aoo = keep(o * v * o * v);
        q.P(-2.0, Vijab, "kldc", 0.0, mv(Aoo, {o, v, o, v}), "kdlc");
        q.P(1.0, Vijab, "klcd", 1.0, mv(Aoo, {o, v, o, v}), "kdlc");
        // small hoisted V.T blocks
        A3 = keep(o * o * v * o);
        q.C(-2.0, Vijak, "klci", T4, "cbkj", 0.0, mv(A3, {o, o, v, o}), "libj");      // A_oovo
        q.C(1.0, Vijka, "klic", T4, "cbkj", 1.0, mv(A3, {o, o, v, o}), "libj");
        q.C(1.0, Vijak, "kldi", T4, "bdkj", 1.0, mv(A3, {o, o, v, o}), "libj");
        A4 = keep(o * o * v * o);
        q.C(1.0, Vijka, "klid", T4, "adkj", 0.0, mv(A4, {o, o, v, o}), "liaj");
        A6 = keep(o * v * o * o);
        q.C(1.0, Viabc, "lacd", T4, "cdji", 0.0, mv(A6, {o, v, o, o}), "laji");
        Gvv = keep(v * v);
        q.L(Gvv, {fvv}, {1.0}, v * v);
        q.C(-2.0, Vijab, "klcd", T4, "cakl", 1.0, mv(Gvv, {v, v}), "ad");
        q.C(1.0, Vijab, "klcd", T4, "ackl", 1.0, mv(Gvv, {v, v}), "ad");
        Goo = keep(o * o);
        q.L(Goo, {foo}, {-1.0}, o * o);
        q.C(-2.0, Vijab, "klcd", T4, "cdki", 1.0, mv(Goo, {o, o}), "li");
        q.C(1.0, Vijab, "kldc", T4, "cdki", 1.0, mv(Goo, {o, o}), "li");
        B2 = keep(o * o * o * o);
        q.P(1.0, Vklij, "klij", 0.0, mv(B2, {o, o, o, o}), "klij");
        q.C(1.0, Vijab, "klcd", T4, "cdij", 1.0, mv(B2, {o, o, o, o}), "klij");
        // particle ladder (:383): pair-packed form (1/4 of the flops) whenever V_abcd = V_badc and the trial doubles are
        // exchange-symmetric — true for every vector the Davidson driver generates
        v_sym = exchange_symmetric(Vabcd.p, v, v);
        // T_abij = T_baji (every CCSD solution): P(ijab,jiba)[T B5] = T (B5 + B5^(lkji)), so that term rides in the product
        // with B' of eom_ccsd.py:381 — one v^2 o^4 product less per sigma
        fused_ok = dev::fused_pair_kernels_ok(no);
        t_sym = exchange_symmetric(T, v, o);
        // eom_ccsd.py:380-382 in pair-packed rows needs B2_klij = B2_lkji and V_klcd = V_lkdc
        hole_sym = t_sym && exchange_symmetric(B2, o, o) && exchange_symmetric(Vijab.p, o, v);
        if (v_sym) L = keep(v * (v + 1) / 2 * o * o);
        many_ok = v_sym && hole_sym && fused_ok && t_sym;
        if (many_ok) {
            fovT = upload(0, o, o, v, 1.0, true);
            // the four u1 terms with one free index on u1 (A_oovo, A4, A6, V_iajk) as ONE product u1[a,l] A346[l,b,i,j]:
            // everything added to D is symmetrised by P(ijab,jiba) afterwards (:377), so a term X_abij may be replaced by its
            // partner X_baji
            A346 = keep(o * v * o * o);
            q.P(1.0, mv(A3, {o, o, v, o}), "libj", 0.0, mv(A346, {o, v, o, o}), "lbij");
            q.P(1.0, mv(A4, {o, o, v, o}), "ljbi", 1.0, mv(A346, {o, v, o, o}), "lbij");
            q.L(A346, {A346, A6, Viajk.p}, {1.0, -1.0, -1.0}, o * v * o * o);
            // ... and that product shares its shape with X_vv . T (:360-361): [z a] x (c | l) against the rows of T and of A346
            // stacked, TA[(c | l), (b,i,j)] — ONE product writes D (two passes over the k amplitude-sized accumulators before)
            TA = keep((v + o) * v * o * o);
            dev::memcpy_d2d(TA, T, sizeof(double) * v * v * o * o, e.stream);
            dev::memcpy_d2d(TA + v * v * o * o, A346, sizeof(double) * o * v * o * o, e.stream);
        }
    } catch (...) {
        for (double* p : owned_) e.scratch_put(p);
        owned_.clear();
        throw;
    }
}

// ---- eom_ccsd.py:268-310 -------------------------------------------------------------------------------------------------------
void EomSigma::singles(const double* u1, const double* u2, double* s1) {
    const int64_t o = no, v = nv;
    const Ops q{e};
    const TView U1 = mv(u1, {v, o}), U2 = mv(u2, {v, v, o, o}), S = mv(s1, {v, o});
    Tmp ut(*this, v * v * o * o);                                      // 2 u2[a,b,i,j] - u2[b,a,i,j]
    const TView Ut = mv(ut, {v, v, o, o});
    q.P(2.0, U2, "abij", 0.0, Ut, "abij");
    q.P(-1.0, U2, "baij", 1.0, Ut, "abij");
    q.C(1.0, U1, "ck", mv(W1, {v, o, v, o}), "ckai", 0.0, S, "ai");
    q.C(1.0, mv(Gvv_s, {v, v}), "ac", U1, "ci", 1.0, S, "ai");
    q.C(1.0, U1, "ak", mv(Goo_s, {o, o}), "ki", 1.0, S, "ai");
    q.C(1.0, mv(fov, {o, v}), "jb", Ut, "baji", 1.0, S, "ai");
    q.C(-1.0, V("ijka"), "jkib", Ut, "abjk", 1.0, S, "ai");
    q.C(1.0, V("iabc"), "jabc", Ut, "bcji", 1.0, S, "ai");
}

// ---- eom_ccsd.py:312-385 -------------------------------------------------------------------------------------------------------
void EomSigma::doubles(const double* u1, const double* u2, bool u2_sym, double* s2, bool defer_ladder) {
    const int64_t o = no, v = nv, ov = o * v, ov2 = ov * ov, npp = v * (v + 1) / 2;
    const Ops q{e};
    const TView U1 = mv(u1, {v, o}), U2 = mv(u2, {v, v, o, o}), T4 = mv(T, {v, v, o, o});
    const TView Vijab = V("ijab"), Vijka = V("ijka"), Vijak = V("ijak"), Viabc = V("iabc");
    auto P4 = [&](double* p) { return mv(p, {v, o, v, o}); };
    // a trial vector without exchange symmetry: the five (ov)^3 products as TWO, stacked along the summed pair index
    const bool kstack = !u2_sym && fused_ok;
    Tmp u2x(*this, kstack ? 1 : ov2), utd(*this, kstack ? 1 : ov2), Dx(*this, ov2), Dd(*this, ov2);
    Tmp u2d(*this, u2_sym || kstack ? 1 : ov2), R3(*this, kstack ? 3 * ov2 : 1);
    // the pair layouts u2x[(a,j),(b,i)] = u2[a,b,i,j], utd = 2 u2d - (u2[b,a,i,j] in the u2d layout), u2d[(a,i),(b,j)] = u2[a,b,i,j]
    if (kstack) {
        // R3 = [u2d ; u2x^T ; u2x] from one pass over u2 (u2x^T[(d,l),(b,j)] = u2[b,d,l,j] is "u2[b,a,i,j] in the u2d layout"):
        //   Dd = M1.(2 u2d - u2x^T) + M2.u2d + M_C.u2x = [2 M1 + M2 | -M1 | M_C] . R3                       (K = 3 ov)
        //   Dx = M_D.u2x - u2x.Ud^T, and since only Dx + Dx^T enters (:377, the assembly below) the second term may be
        //   transposed: Dx' = [-Ud | M_D] . [u2x^T ; u2x]                                                      (K = 2 ov)
        // same flops, but 841 tiles x 5 launches with their cut tails become two long launches (8.0 -> 6.9 ms at (30,120))
        general_operands();
        dev::t2_layouts(u2, R3.p, R3.p + 2 * ov2, R3.p + ov2, no, nv, e.stream, 0.0, 1.0);
        q.C(1.0, mv(LK3, {v, o, 3, v, o}), "aisdl", mv(R3.p, {3, v, o, v, o}), "sdlbj", 0.0, P4(Dd), "aibj");
        q.C(1.0, mv(LK2, {v, o, 2, v, o}), "ajsdl", mv(R3.p + ov2, {2, v, o, v, o}), "sdlbi", 0.0, P4(Dx), "ajbi");
    } else if (fused_ok) {             // ... in ONE pass over u2 (the kernel of the CCSD residual's layouts)
        dev::t2_layouts(u2, u2_sym ? nullptr : u2d.p, u2x, utd, no, nv, e.stream);
    } else {
        q.P(1.0, U2, "abij", 0.0, P4(u2x), "ajbi");
        q.P(2.0, U2, "abij", 0.0, P4(utd), "aibj");                     // ut[d,b,l,j] = 2u2[d,b,l,j] - u2[b,d,l,j]
        q.P(-1.0, U2, "baij", 1.0, P4(utd), "aibj");
        if (!u2_sym) q.P(1.0, U2, "abij", 0.0, P4(u2d), "aibj");
    }
    // ---- (ov)^3 products ---------------------------------------------------------------------------------------------------
    if (u2_sym) {
        // exchange-symmetric u2: utd = 2 u2d - u2x as matrices, hence M1.utd + M2.u2d + M_C.u2x = (2 M1 + M2).utd / 2 +
        // (M_D - Ud).u2x / 2, and the second product IS Dx (the C / D form of the ring terms): TWO (ov)^3 products per sigma
        q.C(1.0, P4(MDU), "ajdl", P4(u2x), "dlbi", 0.0, P4(Dx), "ajbi");                       // :372 and :364 (transposed)
        // Dd also carries Dx / 2 in the direct placement: the fused assembly reads Dx there itself (xd), else a scaled copy
        if (!fused_ok) q.P(0.5, P4(Dx), "ajbi", 0.0, P4(Dd), "ajbi");                          // same memory layout as "aibj"
        q.C(0.5, P4(M12), "aidl", P4(utd), "dlbj", fused_ok ? 0.0 : 1.0, P4(Dd), "aibj");
    } else if (!kstack) {
        q.C(1.0, P4(M1), "aidl", P4(utd), "dlbj", 0.0, P4(Dd), "aibj");
        q.C(1.0, P4(M2), "aidl", P4(u2d), "dlbj", 1.0, P4(Dd), "aibj");
        q.C(1.0, P4(M_C), "aidl", P4(u2x), "dlbj", 1.0, P4(Dd), "aibj");                      // u2x[(d,l),(b,j)] = u2[d,b,j,l]
        q.C(1.0, P4(M_D), "ajdl", P4(u2x), "dlbi", 0.0, P4(Dx), "ajbi");                      // :372  u2[d,b,i,l]
        q.C(-1.0, P4(u2x), "ajck", P4(Ud), "bick", 1.0, P4(Dx), "ajbi");                      // :364
    }
    // ---- one-index dressings ---------------------------------------------------------------------------------------------------
    Tmp Xoo(*this, o * o), Xvv(*this, v * v), B5(*this, o * o * o * o);
    const TView XooV = mv(Xoo, {o, o}), XvvV = mv(Xvv, {v, v}), FOV = mv(fov, {o, v});
    q.C(-2.0, Vijka, "klid", U1, "dl", 0.0, XooV, "ki");
    q.C(1.0, Vijak, "kldi", U1, "dl", 1.0, XooV, "ki");
    q.C(-1.0, FOV, "kd", U1, "di", 1.0, XooV, "ki");
    if (u2_sym) {
        q.C(1.0, mv(Aoo, {o, v, o, v}), "kdlc", P4(u2x), "dlci", 1.0, XooV, "ki");
    } else {
        q.C(-2.0, Vijab, "kldc", U2, "dcil", 1.0, XooV, "ki");
        q.C(1.0, Vijab, "kldc", U2, "dcli", 1.0, XooV, "ki");
    }
    q.C(1.0, XooV, "ki", P4(Td), "akbj", 1.0, P4(Dd), "aibj", "a");
    q.C(2.0, Viabc, "ladc", U1, "dl", 0.0, XvvV, "ac");
    q.C(-1.0, Viabc, "lacd", U1, "dl", 1.0, XvvV, "ac");
    q.C(-1.0, U1, "al", FOV, "lc", 1.0, XvvV, "ac");
    if (u2_sym) {
        q.C(1.0, P4(u2x), "akdl", slice(mv(BB, {o, v, o, v + o}), 3, 0, v), "kdlc", 1.0, XvvV, "ac");
    } else {
        q.C(-2.0, Vijab, "lkcd", U2, "adlk", 1.0, XvvV, "ac");
        q.C(1.0, Vijab, "lkcd", U2, "dalk", 1.0, XvvV, "ac");
    }
    const bool packed = v_sym && u2_sym && hole_sym;
    const bool fused = packed && fused_ok;
    // D goes straight into the caller's array unless the fused assembly needs it as an input
    Tmp Dtmp(*this, fused_ok ? v * v * o * o : 1);
    double* Dp = fused_ok ? Dtmp.p : s2;
    TView D = mv(Dp, {v, v, o, o});
    q.C(1.0, XvvV, "ac", T4, "cbij", 0.0, D, "abij");
    // V_kacd.T.u1 terms (:334, :343, :345, :346) through the hoisted V.T intermediates: o^2 v^3 instead of (ov)^3 each
    q.C(1.0, mv(WW, {v, v, o, v}), "abid", U1, "dj", 1.0, D, "abij");
    q.C(1.0, mv(Gvv, {v, v}), "ad", U2, "dbij", 1.0, D, "abij");
    q.C(1.0, mv(Goo, {o, o}), "li", U2, "ablj", 1.0, D, "abij", "ab");
    q.C(1.0, U1, "al", mv(A3, {o, o, v, o}), "libj", 1.0, D, "abij");
    q.C(1.0, U1, "bl", mv(A4, {o, o, v, o}), "liaj", 1.0, D, "abij");
    q.C(-1.0, U1, "bl", mv(A6, {o, v, o, o}), "laji", 1.0, D, "abij");
    const TView B5v = mv(B5, {o, o, o, o});
    q.C(1.0, Vijka, "klid", U1, "dj", 0.0, B5v, "klij");
    if (!t_sym) q.C(1.0, T4, "abkl", B5v, "klij", 1.0, D, "abij");
    q.C(-1.0, U1, "ak", V("iajk"), "kbij", 1.0, D, "abij");
    auto packed_terms = [&]() {       // the terms (:380-383) that stay outside P(ijab,jiba), in the pair-packed rows L
        Tmp B5s(*this, o * o * o * o);
        q.P(1.0, B5v, "klij", 0.0, mv(B5s, {o, o, o, o}), "klij");
        q.P(1.0, B5v, "lkji", 1.0, mv(B5s, {o, o, o, o}), "klij");
        e.ladder_sym(u2, L, 0, npp, dressed, 0);
        e.hole_ladder_packed(u2, B2, L, 0, npp, nullptr);
        e.hole_ladder_packed(T, B5s, L, 0, npp, u2);
    };
    if (fused) {
        // symmetrisation (:377) of D and of the two pair matrices and the unpacking of L in ONE pass (the assembly kernel of
        // the CCSD residual)
        packed_terms();
        dev::residual_assemble(nullptr, L, Dp, Dd, Dx, s2, no, nv, e.stream, 0.5);
        return;
    }
    // ---- P(ijab, jiba) (:377), then the unpermuted terms (:380-383) ------------------------------------------------------------
    if (fused_ok) {                    // D + P D and the two pair matrices with their transposes in one pass, into the result
        dev::residual_assemble(nullptr, nullptr, Dp, Dd, Dx, s2, no, nv, e.stream, u2_sym ? 0.5 : 0.0);
        Dp = s2;
        D = mv(Dp, {v, v, o, o});
    } else {
        q.P(1.0, P4(Dd), "aibj", 1.0, D, "abij");
        q.P(1.0, P4(Dx), "ajbi", 1.0, D, "abij");
        Tmp S(*this, v * v * o * o);
        q.P(1.0, D, "baji", 0.0, mv(S, {v, v, o, o}), "abij");
        q.L(Dp, {Dp, S}, {1.0, 1.0}, v * v * o * o);
    }
    if (packed) {
        packed_terms();
        e.ladder_sym_unpack(L, Dp, 1.0);
        return;
    }
    Tmp Bn(*this, o * o * o * o);
    const TView BnV = mv(Bn, {o, o, o, o});
    q.C(1.0, Vijab, "kldc", U2, "dcij", 0.0, BnV, "klij");
    if (t_sym) {                       // + the symmetrised u1 term that was held back above
        q.P(1.0, B5v, "klij", 1.0, BnV, "klij");
        q.P(1.0, B5v, "lkji", 1.0, BnV, "klij");
    }
    q.C(1.0, U2, "abkl", mv(B2, {o, o, o, o}), "klij", 1.0, D, "abij");               // :380, :382
    q.C(1.0, T4, "abkl", BnV, "klij", 1.0, D, "abij");                                  // :381
    if (v_sym && u2_sym) {                                                              // :383
        e.ladder_sym(u2, L, 0, npp, dressed, 0);
        e.ladder_sym_unpack(L, Dp, 1.0);
    } else if (v_sym) {
        if (!defer_ladder) {
            const double* us[1] = {u2};
            double* ss[1] = {Dp};
            general_ladders(1, us, ss);
        }
    } else {
        q.C(1.0, V("abcd"), "abcd", U2, "cdij", 1.0, D, "abij");
    }
}

// The u-independent left operands of the two stacked (ov)^3 products of doubles() for a trial vector without exchange symmetry
void EomSigma::general_operands() {
    if (LK3) return;
    const int64_t ov = static_cast<int64_t>(no) * nv;
    const Ops q{e};
    LK3 = keep(3 * ov * ov);
    LK2 = keep(2 * ov * ov);
    auto columns = [&](double* dst, int64_t width, int64_t s, double c0, const double* m0, double c1, const double* m1) {
        const int64_t dims[2] = {ov, ov}, st[2] = {width * ov, 1};
        const TView out = make_view(dst + s * ov, 2, dims, st);
        q.P(c0, mv(m0, {ov, ov}), "xy", 0.0, out, "xy");
        if (m1) q.P(c1, mv(m1, {ov, ov}), "xy", 1.0, out, "xy");
    };
    columns(LK3, 3, 0, 2.0, M1, 1.0, M2);
    columns(LK3, 3, 1, -1.0, M1, 0.0, nullptr);
    columns(LK3, 3, 2, 1.0, M_C, 0.0, nullptr);
    columns(LK2, 2, 0, -1.0, Ud, 0.0, nullptr);
    columns(LK2, 2, 1, 1.0, M_D, 0.0, nullptr);
}

// The particle ladder (:383) of g trial vectors WITHOUT exchange symmetry, added to s2[z].  It acts on (i,j) as spectators and
// commutes with the exchange P (V_abcd = V_badc): with u = us + ua (us = (u + P u) / 2) the symmetric part runs pair-packed as
// it stands, and so does the antisymmetric one after w_abij = sgn(i - j) ua_abij (exchange-symmetric, zero for i == j):
// Lad(ua)_abij = sgn(i - j) Lad(w)_abij; the o columns i == j that w leaves out are a skinny plain product.  2 g quarter-flop
// ladders in ONE batched launch per half (the packed integrals read once) + v^4 o flops per vector instead of g full v^4 o^2
// products.  (Nothing else of the general build can be split this way: the reference's symmetrised part is not P-covariant,
// DESIGN 6e.)
void EomSigma::general_ladders(int g, const double* const* u2, double* const* s2) {
    const int64_t o = no, v = nv, n2 = v * v * o * o, npp = v * (v + 1) / 2, G = g;
    const Ops q{e};
    Tmp usw(*this, 2 * G * n2), dg(*this, G * v * v * o), Lall(*this, 2 * G * npp * o * o), R2(*this, n2), Rd(*this, G * v * v * o);
    std::vector<const double*> xs(2 * g);
    for (int z = 0; z < g; ++z) {
        double *us = usw.p + 2 * z * n2, *w = us + n2;
        dev::exchange_split(u2[z], us, w, dg.p + z * v * v * o, no, nv, e.stream);
        xs[2 * z] = us;
        xs[2 * z + 1] = w;
    }
    e.ladder_sym_multi(xs.data(), 2 * g, Lall, dressed);
    // the diagonal columns of all vectors from one pass over V_abcd
    q.C(1.0, V("abcd"), "abcd", mv(dg, {G, v, v, o}), "zcdi", 0.0, mv(Rd, {G, v, v, o}), "zabi");
    for (int z = 0; z < g; ++z) {
        e.ladder_sym_unpack(Lall.p + 2 * z * npp * o * o, s2[z], 1.0);
        e.ladder_sym_unpack(Lall.p + (2 * z + 1) * npp * o * o, R2, 0.0);
        dev::sgn_ij_add(s2[z], R2, no, nv, e.stream);
        const int64_t dims[3] = {v, v, o}, st[3] = {v * o * o, o * o, o + 1};
        q.P(1.0, mv(Rd.p + z * v * v * o, {v, v, o}), "abi", 1.0, make_view(s2[z], 3, dims, st), "abi");      // s2[a,b,i,i] += Rd[a,b,i]
    }
}

int EomSigma::stack_limit() const {
    // nine (ov)^2-sized temporaries per vector (X, Tt, DxT, DdT, D, the packed ladder rows and their operands) must fit in
    // half of what the device has free right now (pooled buffers count as free)
    const double free_b = static_cast<double>(dev::mem_free_bytes()) + static_cast<double>(e.scratch_free_bytes());
    const double per_vector = 9.0 * 8.0 * static_cast<double>(no) * nv * static_cast<double>(no) * nv;
    return static_cast<int>(std::max(1.0, std::min(16.0, std::floor(free_b / 2.0 / std::max(per_vector, 1.0)))));
}

// ---- sigma for k exchange-symmetric trial vectors at once: every operand that does not depend on the vector is read ONCE ----
void EomSigma::stack(int k, const double* const* u1, const double* const* u2, double* const* s1, double* const* s2) {
    const int64_t o = no, v = nv, ov = o * v, ov2 = ov * ov, npp = v * (v + 1) / 2, K = k;
    const Ops q{e};
    const TView Vijka = V("ijka"), Vijak = V("ijak"), Viabc = V("iabc");
    Tmp U1(*this, K * v * o), X(*this, K * ov2), Tt(*this, K * ov2);
    for (int z = 0; z < k; ++z) {
        dev::memcpy_d2d(U1.p + z * v * o, u1[z], sizeof(double) * v * o, e.stream);
        // X[z,(a,j),(b,i)] = u2_z[a,b,i,j], Tt[z,(a,i),(b,j)] = 2 u2_z[a,b,i,j] - u2_z[b,a,i,j]        (symmetric matrices)
        dev::t2_layouts(u2[z], nullptr, X.p + z * ov2, Tt.p + z * ov2, no, nv, e.stream);
    }
    const TView U1v = mv(U1, {K, v, o}), Xv = mv(X, {K, v, o, v, o}), Ttv = mv(Tt, {K, v, o, v, o});
    // ---- singles (eom_ccsd.py:268-310), ut[a,b,i,j] = Tt[(a,i),(b,j)] -----------------------------------------------------------
    Tmp S1(*this, K * v * o);
    const TView S1v = mv(S1, {K, v, o});
    q.C(1.0, U1v, "zck", mv(W1, {v, o, v, o}), "ckai", 0.0, S1v, "zai");
    q.C(1.0, mv(Gvv_s, {v, v}), "ac", U1v, "zci", 1.0, S1v, "zai", "z");
    q.C(1.0, U1v, "zak", mv(Goo_s, {o, o}), "ki", 1.0, S1v, "zai");
    q.C(1.0, Ttv, "zaibj", mv(fovT, {v, o}), "bj", 1.0, S1v, "zai");
    // X_vv's u2 part (:359-360) and the singles term :297 from one pass over X: XS[z,a,:] = X[z,(a,k),(d,l)] BB[(k,d,l),:]
    Tmp XS(*this, K * v * (v + o));
    const TView XSv = mv(XS, {K, v, v + o});
    q.C(1.0, Xv, "zakdl", mv(BB, {o, v, o, v + o}), "kdln", 0.0, XSv, "zan");
    e.axpby(1.0, slice(XSv, 2, v, v + o), 1.0, S1v);
    q.C(1.0, Viabc, "jabc", Ttv, "zbjci", 1.0, S1v, "zai", "z");                       // (z as a batch: Tt is read in place)
    // ---- (ov)^3 products, transposed: only Dx + Dx^T and Dd + Dd^T enter (:377), X and Tt are symmetric matrices ---------------
    Tmp DxT(*this, K * ov2), DdT(*this, K * ov2);
    const TView DxTv = mv(DxT, {K, v, o, v, o}), DdTv = mv(DdT, {K, v, o, v, o});
    q.C(1.0, Xv, "zajdl", mv(MDU, {v, o, v, o}), "bidl", 0.0, DxTv, "zajbi");          // (MDU . u2x)^T per vector
    // (DdT also carries DxT / 2 in the direct placement: the assembly below reads DxT there itself, xd = 0.5 — no scaled copy)
    q.C(0.5, Ttv, "zaidl", mv(M12, {v, o, v, o}), "bjdl", 0.0, DdTv, "zaibj");
    // ---- one-index dressings -------------------------------------------------------------------------------------------------------
    Tmp Xoo(*this, K * o * o), Xvv(*this, K * v * v);
    const TView XooV = mv(Xoo, {K, o, o}), XvvV = mv(Xvv, {K, v, v}), FOV = mv(fov, {o, v});
    q.C(-2.0, Vijka, "klid", U1v, "zdl", 0.0, XooV, "zki");
    q.C(1.0, Vijak, "kldi", U1v, "zdl", 1.0, XooV, "zki");
    q.C(-1.0, FOV, "kd", U1v, "zdi", 1.0, XooV, "zki", "z");
    q.C(1.0, mv(Aoo, {o, v, o, v}), "kdlc", Xv, "zdlci", 1.0, XooV, "zki", "z");       // u2[d,c,i,l] = X[(d,l),(c,i)], both placements
    q.C(1.0, XooV, "zki", mv(Td, {v, o, v, o}), "akbj", 1.0, DdTv, "zaibj", "za");
    q.C(2.0, Viabc, "ladc", U1v, "zdl", 0.0, XvvV, "zac");
    q.C(-1.0, Viabc, "lacd", U1v, "zdl", 1.0, XvvV, "zac");
    q.C(-1.0, U1v, "zal", FOV, "lc", 1.0, XvvV, "zac");
    e.axpby(1.0, slice(XSv, 2, 0, v), 1.0, XvvV);                                       // u2[a,d,l,k] = X[(a,k),(d,l)], both placements
    Tmp D(*this, K * v * v * o * o);
    const TView Dv = mv(D, {K, v, v, o, o});
    {
        // D = [X_vv | u1] . [T ; A346]   (:360-361 and the four u1 terms with a free index on u1, see the hoist)
        Tmp XU(*this, K * v * (v + o));
        const TView XUv = mv(XU, {K, v, v + o});
        e.copy(XvvV, slice(XUv, 2, 0, v));
        e.copy(U1v, slice(XUv, 2, v, v + o));
        q.C(1.0, XUv, "zan", mv(TA, {v + o, v, o, o}), "nbij", 0.0, Dv, "zabij");
    }
    q.C(1.0, mv(WW, {v, v, o, v}), "abid", U1v, "zdj", 1.0, Dv, "zabij", "z");
    Tmp Lall(*this, K * npp * o * o);
    e.ladder_sym_multi(u2, k, Lall, dressed);                                           // :383, all vectors
    Tmp B5(*this, K * o * o * o * o), B5s(*this, K * o * o * o * o);
    std::vector<const double*> b2s(k, B2), ts(k, T), b5p(k);
    for (int z = 0; z < k; ++z) {
        // (G_vv . u2, G_oo . u2 per vector: a batched launch over a stacked direct layout moves the same bytes — measured)
        const TView Dz = mv(D.p + z * v * v * o * o, {v, v, o, o}), U2z = mv(u2[z], {v, v, o, o});
        q.C(1.0, mv(Gvv, {v, v}), "ad", U2z, "dbij", 1.0, Dz, "abij");
        q.C(1.0, mv(Goo, {o, o}), "li", U2z, "ablj", 1.0, Dz, "abij", "ab");
        const TView B5z = mv(B5.p + z * o * o * o * o, {o, o, o, o}), B5sz = mv(B5s.p + z * o * o * o * o, {o, o, o, o});
        q.C(1.0, Vijka, "klid", mv(U1.p + z * v * o, {v, o}), "dj", 0.0, B5z, "klij");
        q.P(1.0, B5z, "klij", 0.0, B5sz, "klij");
        q.P(1.0, B5z, "lkji", 1.0, B5sz, "klij");
        b5p[z] = B5sz.p;
    }
    // the hole-ladder-shaped terms of all vectors in batched launches (the shared side — V_klij + V_klcd T_cdij, then T —
    // packed once)
    e.hole_ladder_packed_multi(u2, b2s.data(), nullptr, k, Lall);                      // :380, :382
    e.hole_ladder_packed_multi(ts.data(), b5p.data(), u2, k, Lall);                    // :381 (+ the symmetrised u1 term)
    for (int z = 0; z < k; ++z) {
        dev::residual_assemble(nullptr, Lall.p + z * npp * o * o, D.p + z * v * v * o * o, DdT.p + z * ov2, DxT.p + z * ov2, s2[z],
                               no, nv, e.stream, 0.5);                                  // :377 + unpacking
        dev::memcpy_d2d(s1[z], S1.p + z * v * o, sizeof(double) * v * o, e.stream);
    }
}

void EomSigma::apply(int k, const double* const* u1, const double* const* u2, const int* sym, double* const* s1,
                     double* const* s2) {
    if (k < 1) return;
    const int64_t o = no, v = nv;
    std::vector<int> sy(k);
    bool all = true;
    for (int z = 0; z < k; ++z) {
        if (!u1[z] || !u2[z] || !s1[z] || !s2[z]) throw Error("eom sigma: null vector");
        sy[z] = sym ? (sym[z] != 0) : exchange_symmetric(u2[z], v, o);
        all = all && sy[z];
    }
    auto one = [&](int z, bool defer = false) {
        singles(u1[z], u2[z], s1[z]);
        doubles(u1[z], u2[z], sy[z] != 0, s2[z], defer);
    };
    if (k < 2 || !many_ok || !all) {
        // vectors without exchange symmetry: their particle ladders are held back and run together, eight vectors at a time
        std::vector<const double*> gu;
        std::vector<double*> gs;
        for (int z = 0; z < k; ++z) {
            const bool defer = k > 1 && v_sym && !sy[z];
            one(z, defer);
            if (defer) {
                gu.push_back(u2[z]);
                gs.push_back(s2[z]);
            }
            if (gu.size() == 8 || (z == k - 1 && !gu.empty())) {
                general_ladders(static_cast<int>(gu.size()), gu.data(), gs.data());
                gu.clear();
                gs.clear();
            }
        }
        return;
    }
    const int step = stack_limit();
    for (int lo = 0; lo < k; lo += step) {
        const int hi = std::min(k, lo + step);
        if (hi - lo < 2) { one(lo); continue; }
        try {
            stack(hi - lo, u1 + lo, u2 + lo, s1 + lo, s2 + lo);
        } catch (const Error& err) {
            if (std::string(err.what()).find("memory") == std::string::npos) throw;
            trim();                    // out of device memory in the middle of a stacked build: one vector at a time
            for (int z = lo; z < hi; ++z) one(z);
        }
    }
}

// ---- eom_ccsd.py:169-198 (singles) and :200-266 (doubles) on the device ---------------------------------------------------------
void eom_diagonals(Engine& e, const double* f_host, const double* t2, bool dressed, double* d1, double* d2) {
    if (!f_host || !t2 || !d1 || !d2) throw Error("eom diagonals: null argument");
    const int64_t o = e.no, v = e.nv, n = o + v;
    std::vector<double> h(static_cast<size_t>(v * o));
    for (int64_t a = 0; a < v; ++a)
        for (int64_t i = 0; i < o; ++i) h[a * o + i] = f_host[(o + a) * n + o + a] - f_host[i * n + i];
    ArenaScope scope(e.arena);
    double *dai = e.arena.alloc(v * o), *iaai = e.arena.alloc(v * o), *iaia = e.arena.alloc(v * o), *ijij = e.arena.alloc(o * o),
           *abab = e.arena.alloc(v * v), *ws = e.arena.alloc(dev::eom_diag_ws_doubles(e.no, e.nv));
    dev::memcpy_h2d(dai, h.data(), sizeof(double) * v * o, e.stream);
    dev::stream_sync(e.stream);
    // the four diagonal slices of other blocks, gathered through strided views (o v / o^2 / v^2 numbers)
    auto gather = [&](const char* name, int64_t d0, int64_t d1_, int64_t s0, int64_t s1_, double* out) {
        const TView blk = e.block(pattern_of_name(name), dressed);
        const int64_t dims[2] = {d0, d1_}, st[2] = {s0, s1_};
        e.permute(1.0, make_view(blk.p, 2, dims, st), "xy", 0.0, mv(out, {d0, d1_}), "xy");
    };
    gather("iabj", v, o, v * o + o, v * v * o + 1, iaai);          // [a,i] <- V[i,a,a,i]
    gather("iajb", v, o, o * v + 1, v * o * v + v, iaia);          // [a,i] <- V[i,a,i,a]
    gather("klij", o, o, o * o * o + o, o * o + 1, ijij);          // [i,j] <- V[i,j,i,j]
    gather("abcd", v, v, v * v * v + v, v * v + 1, abab);          // [a,b] <- V[a,b,a,b]
    dev::eom_diagonals(e.block(pattern_of_name("ijab"), dressed).p, t2, dai, iaai, iaia, ijij, abab, d1, d2, e.no, e.nv, ws,
                       e.stream);
}

}  // namespace pymes
