// extern "C" surface of libpymes_amd (see include/pymes_amd.h).
#include <chrono>
#include <cstdio>
#include <cstring>
#include <exception>
#include <algorithm>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/pymes_amd.h"
#include "diis_small.h"
#include "engine.h"
#include "eom.h"
#include "fcidump.h"
#include "packed.h"

using pymes::Engine;
using pymes::TView;

struct pymes_ctx {
    Engine* e;
};
struct pymes_eom {
    pymes::EomSigma* s;
    pymes_ctx* ctx;
};

namespace {
thread_local std::string g_err;
// EOM sigma handles alive per context: a context that is destroyed first takes the device side of its handles with
// it (their hoisted arrays live in the engine's scratch pool); a later pymes_eom_sigma_destroy then only frees the host shell
std::mutex g_eom_mu;
std::set<pymes_eom*> g_eom_live;

template <class F>
int guarded(F&& f) {
    // (what an entry point has recorded for the open phase — device_api.h, phase launches — is on the stream when it returns:
    // the host program may order work of its own behind it)
    try {
        f();
        dev::phase_call_end();
        return 0;
    } catch (const std::exception& ex) {
        g_err = ex.what();
    } catch (...) {
        g_err = "unknown error";
    }
    try {
        dev::phase_sync();
    } catch (...) {
    }
    return 1;
}
Engine& Eq(pymes_ctx* c) {             // entry points that may add to an open group of products (pymes_gemm_group_begin)
    if (!c || !c->e) throw pymes::Error("null context");
    dev::set_device(c->e->device);      // a process may hold contexts on several GPUs: every entry runs on its own
    return *c->e;
}
Engine& E(pymes_ctx* c) {               // every other entry point: what is queued in an open group goes first
    Engine& e = Eq(c);
    dev::gemm_group_sync();
    return e;
}
void need(const void* p, const char* what) {
    if (!p) throw pymes::Error(std::string("null pointer: ") + what);
}
pymes::EomSigma& S(pymes_eom* h) {
    if (!h || !h->s || !h->ctx) throw pymes::Error("null EOM handle");
    E(h->ctx);                           // the context's device; queued grouped products first
    return *h->s;
}
TView view_of(const double* p, const char* labels, const int64_t* dim, const int64_t* stride) {
    need(p, "tensor data");
    need(labels, "labels");
    need(dim, "dims");
    return pymes::make_view(const_cast<double*>(p), static_cast<int>(std::strlen(labels)), dim, stride);
}
}  // namespace

extern "C" {

const char* pymes_last_error(void) { return g_err.c_str(); }
const char* pymes_backend(void) { return dev::backend_name(); }

int pymes_ctx_create(pymes_ctx** out, int device, int no, int nv, uint64_t workspace_bytes) {
    return guarded([&] {
        need(out, "out");
        *out = nullptr;
        Engine* e = new Engine(device, no, nv, static_cast<size_t>(workspace_bytes));
        *out = new pymes_ctx{e};
    });
}
int pymes_ctx_destroy(pymes_ctx* ctx) {
    return guarded([&] {
        if (!ctx) return;
        {
            std::lock_guard<std::mutex> lock(g_eom_mu);
            for (pymes_eom* h : g_eom_live)
                if (h->ctx == ctx) {                 // the handle outlives its context: invalidate it while the engine is alive
                    if (ctx->e) dev::set_device(ctx->e->device);
                    delete h->s;
                    h->s = nullptr;
                    h->ctx = nullptr;
                }
        }
        delete ctx->e;
        delete ctx;
    });
}
int pymes_ctx_set_stream(pymes_ctx* ctx, void* s) {
    return guarded([&] {
        Engine& e = E(ctx);
        if (e.capturing()) throw pymes::Error("set_stream while a launch graph is being recorded");
        dev::stream_sync(e.stream);       // work already enqueued on the old stream is complete before the switch
        e.set_stream(s);
    });
}
int pymes_ctx_sync(pymes_ctx* ctx) {
    return guarded([&] {
        Engine& e = E(ctx);
        dev::stream_sync(e.stream);
    });
}
int pymes_ctx_workspace(pymes_ctx* ctx, uint64_t* cap, uint64_t* high) {
    return guarded([&] {
        if (cap) *cap = E(ctx).arena.capacity();
        if (high) *high = E(ctx).arena.high_water();
    });
}

int pymes_malloc(pymes_ctx* ctx, uint64_t bytes, void** p) {
    return guarded([&] {
        need(p, "dev_ptr");
        *p = E(ctx).user_malloc(bytes);
    });
}
int pymes_free(pymes_ctx* ctx, void* p) {
    return guarded([&] { E(ctx).user_free(p); });
}
int pymes_live_allocations(int64_t* n) {
    return guarded([&] {
        if (!n) throw pymes::Error("null output");
        *n = dev::live_allocations();
    });
}
int pymes_mem_info(pymes_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes) {
    return guarded([&] {
        E(ctx);
        if (free_bytes) *free_bytes = dev::mem_free_bytes();
        if (total_bytes) *total_bytes = dev::mem_total_bytes();
    });
}
int pymes_dress_generation(pymes_ctx* ctx, uint64_t* n) {
    return guarded([&] {
        if (!n) throw pymes::Error("null output");
        *n = E(ctx).dress_generation();
    });
}
int pymes_phase_enable(int mode) {
    return guarded([&] { dev::phase_enable(mode); });
}
int pymes_phase_hold(int on) {
    return guarded([&] { dev::phase_hold(on != 0); });
}
int pymes_phase_stats(int64_t* tasks, int64_t* launches, int64_t* levels, int64_t* flushes) {
    return guarded([&] {
        long t = 0, l = 0, v = 0, f = 0;
        dev::phase_stats(&t, &l, &v, &f);
        if (tasks) *tasks = t;
        if (launches) *launches = l;
        if (levels) *levels = v;
        if (flushes) *flushes = f;
    });
}
int pymes_graph_begin(pymes_ctx* ctx) {
    return guarded([&] { E(ctx).graph_begin(); });
}
int pymes_graph_end(pymes_ctx* ctx, void** graph) {
    return guarded([&] {
        need(graph, "graph");
        *graph = nullptr;
        *graph = E(ctx).graph_end();
    });
}
int pymes_graph_abort(pymes_ctx* ctx) {
    return guarded([&] { E(ctx).graph_abort(); });
}
int pymes_graph_launch(pymes_ctx* ctx, void* graph) {
    return guarded([&] { E(ctx).graph_launch(graph); });
}
int pymes_graph_destroy(pymes_ctx* ctx, void* graph) {
    return guarded([&] { E(ctx).graph_destroy(graph); });
}
int pymes_upload(pymes_ctx* ctx, void* d, const void* h, uint64_t bytes) {
    return guarded([&] {
        if (bytes) dev::memcpy_h2d(d, h, bytes, E(ctx).stream);
    });
}
int pymes_download(pymes_ctx* ctx, void* h, const void* d, uint64_t bytes) {
    return guarded([&] {
        if (bytes) dev::memcpy_d2h(h, d, bytes, E(ctx).stream);
    });
}
int pymes_copy(pymes_ctx* ctx, void* d, const void* s, uint64_t bytes) {
    return guarded([&] {
        if (bytes) dev::memcpy_d2d(d, s, bytes, E(ctx).stream);
    });
}
int pymes_memset_zero(pymes_ctx* ctx, void* d, uint64_t bytes) {
    return guarded([&] {
        if (bytes) dev::memset_zero(d, bytes, E(ctx).stream);
    });
}

int pymes_contract(pymes_ctx* ctx, double alpha, const double* A, const char* la, const int64_t* dA,
                   const int64_t* sA, const double* B, const char* lb, const int64_t* dB, const int64_t* sB,
                   double beta, double* C, const char* lc, const int64_t* dC, const int64_t* sC,
                   const char* batch) {
    return guarded([&] {
        Eq(ctx).contract(alpha, view_of(A, la, dA, sA), la, view_of(B, lb, dB, sB), lb, beta, view_of(C, lc, dC, sC),
                        lc, batch ? batch : "");
    });
}
int pymes_permute(pymes_ctx* ctx, double alpha, const double* in, const char* li, const int64_t* dim_in,
                  const int64_t* stride_in, double beta, double* out, const char* lo, const int64_t* stride_out) {
    return guarded([&] {
        need(li, "li");
        need(lo, "lo");
        TView vin = view_of(in, li, dim_in, stride_in);
        const int r = vin.rank;
        if (static_cast<int>(std::strlen(lo)) != r) throw pymes::Error("permute: label strings differ in length");
        int64_t dout[6];
        for (int i = 0; i < r; ++i) {
            const char* f = std::strchr(li, lo[i]);
            if (!f) throw pymes::Error("permute: output label missing from input");
            dout[i] = dim_in[f - li];
        }
        TView vout = pymes::make_view(out, r, dout, stride_out);
        need(out, "out");
        E(ctx).permute(alpha, vin, li, beta, vout, lo);
    });
}
int pymes_dgemm(pymes_ctx* ctx, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t a_sm,
                int64_t a_sk, const double* B, int64_t b_sk, int64_t b_sn, double beta, double* C, int64_t ldc) {
    return guarded([&] {
        Engine& e = Eq(ctx);
        dev::Gemm g{};
        g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta;
        g.A = A; g.a_sm = a_sm; g.a_sk = a_sk;
        g.B = B; g.b_sk = b_sk; g.b_sn = b_sn;
        g.C = C; g.ldc = ldc;
        g.nb1 = g.nb2 = 1;
        g.splitk_ws = e.splitk_ws();
        g.splitk_ws_doubles = e.splitk_ws_doubles();
        dev::gemm(g, e.stream);
        e.stats.gemm_calls++;
        e.stats.gemm_flops += 2.0 * double(M) * double(N) * double(K);
    });
}

int pymes_set_V_pqrs(pymes_ctx* ctx, const double* V, int on_device, const int64_t* strides) {
    return guarded([&] {
        need(V, "V");
        E(ctx).set_V_full(V, on_device != 0, strides);
    });
}
int pymes_set_V_block(pymes_ctx* ctx, const char* name, const double* data, int64_t n_elements, int on_device,
                      const int64_t* strides) {
    return guarded([&] {
        need(data, "data");
        Engine& e = E(ctx);
        int64_t want = 1;
        const int pat = pymes::pattern_of_name(name);
        for (int i = 0; i < 4; ++i) want *= (pat >> (3 - i) & 1) ? e.nv : e.no;
        if (n_elements != want)
            throw pymes::Error(std::string("block '") + name + "' has " + std::to_string(want) + " elements for this context, got " +
                               std::to_string(n_elements));
        e.set_V_block(name, data, on_device != 0, strides);
    });
}
int pymes_V_exchange_asymmetry(pymes_ctx* ctx, double* out_host) {
    return guarded([&] {
        need(out_host, "out");
        E(ctx).exchange_asymmetry_V(out_host);
    });
}
int pymes_exchange_asymmetry(pymes_ctx* ctx, const double* A, const double* B, const int64_t* dims, double* out_host) {
    return guarded([&] {
        need(A, "A"); need(B, "B"); need(dims, "dims"); need(out_host, "out");
        Engine& e = E(ctx);
        dev::exchange_asymmetry(A, B, dims, out_host, e.stream);
    });
}
int pymes_set_V_from_factors(pymes_ctx* ctx, const double* B, int naux) {
    return guarded([&] {
        need(B, "B");
        E(ctx).set_V_from_factors(B, naux);
    });
}
int pymes_V_block_ptr(pymes_ctx* ctx, const char* name, int dressed, double** p, int64_t* nel) {
    return guarded([&] {
        need(p, "dev_ptr");
        *p = nullptr;
        TView v = E(ctx).block(pymes::pattern_of_name(name), dressed != 0);
        *p = v.p;
        if (nel) *nel = v.size();
    });
}
int pymes_set_orbital_energies(pymes_ctx* ctx, const double* eo, const double* ev) {
    return guarded([&] {
        need(eo, "eps_o");
        need(ev, "eps_v");
        E(ctx).set_orbital_energies(eo, ev);
    });
}

int pymes_mp2(pymes_ctx* ctx, double shift, double* t2, double* e_out) {
    return guarded([&] {
        need(t2, "t2");
        need(e_out, "e_out");
        E(ctx).mp2(shift, t2, e_out);
    });
}
int pymes_ccsd_dress_fock(pymes_ctx* ctx, const double* f, const double* t1, double* fd) {
    return guarded([&] {
        need(f, "f"); need(t1, "t1"); need(fd, "fd");
        E(ctx).dress_fock(f, t1, fd);
    });
}
int pymes_ccsd_residuals(pymes_ctx* ctx, const double* f, const double* t1, const double* t2, uint32_t flags, double* r1,
                         double* r2) {
    return guarded([&] {
        need(f, "f"); need(t1, "t1"); need(t2, "t2"); need(r1, "r1"); need(r2, "r2");
        if (flags & ~(PYMES_DCD | PYMES_T1_ZERO)) throw pymes::Error("ccsd_residuals: flags are PYMES_DCD | PYMES_T1_ZERO");
        E(ctx).ccsd_residuals(f, t1, t2, (flags & PYMES_DCD) | ((flags & PYMES_T1_ZERO) ? Engine::kT1Zero : 0u), r1, r2);
    });
}
int pymes_ccsd_iterate(pymes_ctx* ctx, const double* f, double* t1, double* t2, uint32_t flags, double level_shift, double delta,
                       double* dt1, double* dt2, double* out6) {
    return guarded([&] {
        need(f, "f"); need(t1, "t1"); need(t2, "t2"); need(dt1, "dt1"); need(dt2, "dt2"); need(out6, "out");
        if (flags & ~(PYMES_DCD | PYMES_T1_ZERO)) throw pymes::Error("ccsd_iterate: flags are PYMES_DCD | PYMES_T1_ZERO");
        Engine& e = E(ctx);
        if (e.capturing()) throw pymes::Error("ccsd_iterate reads the energy back: not inside a launch graph");
        e.ccsd_iterate(f, t1, t2, (flags & PYMES_DCD) | ((flags & PYMES_T1_ZERO) ? Engine::kT1Zero : 0u), level_shift, delta, dt1,
                       dt2, out6);
    });
}
int pymes_ccsd_release(pymes_ctx* ctx) {
    return guarded([&] {
        Engine& e = E(ctx);
        if (e.capturing()) throw pymes::Error("ccsd_release while a launch graph is being recorded");
        e.release_residual_buffers();
    });
}
int pymes_ccsd_dress_V(pymes_ctx* ctx, const double* t1, uint32_t mask) {
    return guarded([&] {
        need(t1, "t1");
        E(ctx).dress_V(t1, mask);
    });
}
int pymes_ccsd_dress_V_slab(pymes_ctx* ctx, const double* t1, uint32_t mask, int p_begin, int p_end, int q_begin,
                            int q_end) {
    return guarded([&] {
        need(t1, "t1");
        const int64_t cut[4] = {p_begin, p_end, q_begin, q_end};
        E(ctx).dress_V(t1, mask, cut);
    });
}
int pymes_ccsd_singles_residual(pymes_ctx* ctx, const double* fd, const double* t1, const double* t2, double* r1) {
    return guarded([&] {
        need(fd, "fd"); need(t1, "t1"); need(t2, "t2"); need(r1, "r1");
        E(ctx).singles_residual(fd, t1, t2, r1);
    });
}
int pymes_ccsd_singles_residual_partial(pymes_ctx* ctx, const double* fd, const double* t1, const double* t2, double* r1,
                                        int rank, int world, uint32_t flags) {
    return guarded([&] {
        need(fd, "fd"); need(t1, "t1"); need(t2, "t2"); need(r1, "r1");
        E(ctx).singles_residual_partial(fd, t1, t2, r1, rank, world, (flags & PYMES_REUSE_LAYOUTS) != 0);
    });
}
int pymes_doubles_residual(pymes_ctx* ctx, const double* f, const double* t2, double* r2, uint32_t flags) {
    return guarded([&] {
        need(f, "f"); need(t2, "t2"); need(r2, "r2");
        E(ctx).doubles_residual(f, t2, r2, flags);
    });
}
int pymes_ladder(pymes_ctx* ctx, const double* t2, double* r2, int a0, int a1, int dressed, double beta) {
    return guarded([&] {
        need(t2, "t2"); need(r2, "r2");
        E(ctx).ladder(t2, r2, a0, a1, dressed != 0, beta);
    });
}
int pymes_ladder_sym(pymes_ctx* ctx, const double* t2, double* L, int64_t r0, int64_t r1, int dressed, int hole) {
    return guarded([&] {
        need(t2, "t2"); need(L, "L");
        E(ctx).ladder_sym(t2, L, r0, r1, dressed != 0, hole);
    });
}
int pymes_symmetrised_assemble(pymes_ctx* ctx, const double* V, const double* L, const double* N, const double* D,
                               const double* X, double* R) {
    return guarded([&] {
        need(N, "N"); need(D, "D"); need(X, "X"); need(R, "R");
        pymes::Engine& e = E(ctx);
        if (!dev::fused_pair_kernels_ok(e.no)) throw pymes::Error("symmetrised_assemble: nocc too large for the LDS tile");
        dev::residual_assemble(V, L, N, D, X, R, e.no, e.nv, e.stream);
    });
}
int pymes_hole_ladder_packed(pymes_ctx* ctx, const double* x, const double* I, double* L, int64_t r0, int64_t r1,
                             const double* y) {
    return guarded([&] {
        need(x, "x"); need(I, "I"); need(L, "L");
        E(ctx).hole_ladder_packed(x, I, L, r0, r1, y);
    });
}
int pymes_hole_ladder_packed_multi(pymes_ctx* ctx, const double* const* xs, const double* const* Is, const double* const* ys,
                                   int k, double* L_all) {
    return guarded([&] {
        need(xs, "x"); need(Is, "I"); need(L_all, "L_all");
        if (k < 1 || k > 64) throw pymes::Error("hole_ladder_packed_multi: 1 <= k <= 64");
        for (int z = 0; z < k; ++z) { need(xs[z], "x[z]"); need(Is[z], "I[z]"); if (ys) need(ys[z], "y[z]"); }
        E(ctx).hole_ladder_packed_multi(xs, Is, ys, k, L_all);
    });
}
int pymes_ladder_sym_multi(pymes_ctx* ctx, const double* const* xs, int k, double* L_all, int dressed) {
    return guarded([&] {
        need(xs, "x"); need(L_all, "L_all");
        if (k < 1 || k > 64) throw pymes::Error("ladder_sym_multi: 1 <= k <= 64");
        for (int z = 0; z < k; ++z) need(xs[z], "x[z]");
        E(ctx).ladder_sym_multi(xs, k, L_all, dressed != 0);
    });
}
int pymes_ladder_dress(pymes_ctx* ctx, const double* V, const double* Pk, const double* t1, double* W, int64_t ld,
                       int64_t r0, int64_t r1, int minus_half) {
    return guarded([&] {
        need(V, "V"); need(Pk, "Pk"); need(t1, "t1"); need(W, "W");
        pymes::Engine& e = E(ctx);
        pymes::ArenaScope scope(e.arena);
        double* ws = e.arena.alloc(dev::ladder_dress_ws_doubles(e.no, e.nv));
        dev::ladder_dress(V, Pk, t1, W, e.no, e.nv, ld, r0, r1, minus_half ? 1.0 : -1.0, ws, e.stream);
    });
}
int pymes_pair_layouts(pymes_ctx* ctx, const double* x, double* Xd, double* Xx, double* Xt) {
    return guarded([&] {
        need(x, "x"); need(Xx, "Xx"); need(Xt, "Xt");
        pymes::Engine& e = E(ctx);
        if (!dev::fused_pair_kernels_ok(e.no)) throw pymes::Error("pair_layouts: nocc too large for the LDS tile");
        dev::t2_layouts(x, Xd, Xx, Xt, e.no, e.nv, e.stream);
    });
}
int pymes_ladder_sym_unpack(pymes_ctx* ctx, const double* L, double* r2, double beta) {
    return guarded([&] {
        need(L, "L"); need(r2, "r2");
        E(ctx).ladder_sym_unpack(L, r2, beta);
    });
}
int pymes_slab_prepare_ws(pymes_ctx* ctx, int64_t* n_doubles) {
    return guarded([&] {
        if (!n_doubles) throw pymes::Error("null output");
        *n_doubles = E(ctx).slab_prepare_ws_doubles();
    });
}
int pymes_slab_prepare(pymes_ctx* ctx, const double* t2, double* P, int rank, int world, uint32_t flags) {
    return guarded([&] {
        need(t2, "t2"); need(P, "P");
        E(ctx).slab_prepare(t2, P, rank, world, flags);
    });
}
int pymes_residual_slab(pymes_ctx* ctx, const double* f, const double* t2, double* ETd, double* ETx, double* L, int rank,
                        int world, uint32_t flags, const double* t1, double* QK, const double* P) {
    return guarded([&] {
        need(f, "f"); need(t2, "t2"); need(ETd, "ETd"); need(ETx, "ETx");
        if ((t1 == nullptr) != (QK == nullptr)) throw pymes::Error("t1 and QK must be given together");
        if (P && !t1) throw pymes::Error("prepared partial sums go with the amplitude-side mode (t1, QK)");
        E(ctx).residual_slab(f, t2, ETd, ETx, L, rank, world, flags, t1, QK, P);
    });
}
int pymes_residual_finish(pymes_ctx* ctx, const double* f, const double* t2, const double* ETd, const double* ETx,
                          const double* L, double* r2, uint32_t flags, const double* t1, const double* QK) {
    return guarded([&] {
        need(f, "f"); need(t2, "t2"); need(ETd, "ETd"); need(ETx, "ETx"); need(r2, "r2");
        if ((t1 == nullptr) != (QK == nullptr)) throw pymes::Error("t1 and QK must be given together");
        E(ctx).residual_finish(f, t2, ETd, ETx, L, r2, flags, t1, QK);
    });
}
int pymes_residual_finish_pairs(pymes_ctx* ctx, const double* f, const double* t2, const double* ETd, const double* ETx,
                                const double* L, double* Rc, uint32_t flags, const double* t1, const double* QK, int rank,
                                int world, const double* Xvv) {
    return guarded([&] {
        need(f, "f"); need(t2, "t2"); need(ETd, "ETd"); need(ETx, "ETx"); need(L, "L"); need(Rc, "Rc");
        E(ctx).residual_finish_pairs(f, t2, ETd, ETx, L, Rc, flags, t1, QK, rank, world, Xvv);
    });
}
int pymes_xvv_partial(pymes_ctx* ctx, const double* f, const double* t2, double* Xvv, int rank, int world, uint32_t flags) {
    return guarded([&] {
        need(f, "f"); need(t2, "t2"); need(Xvv, "Xvv");
        E(ctx).xvv_partial(f, t2, Xvv, rank, world, flags);
    });
}
int pymes_ccsd_dress_fock_ws(pymes_ctx* ctx, int64_t* n_doubles) {
    return guarded([&] {
        if (!n_doubles) throw pymes::Error("null output");
        *n_doubles = E(ctx).dress_fock_ws_doubles();
    });
}
int pymes_ccsd_dress_fock_partial(pymes_ctx* ctx, const double* t1, double* W, int rank, int world) {
    return guarded([&] {
        need(t1, "t1"); need(W, "W");
        E(ctx).dress_fock_partial(t1, W, rank, world);
    });
}
int pymes_ccsd_dress_fock_finish(pymes_ctx* ctx, const double* f, const double* t1, const double* W, double* fd) {
    return guarded([&] {
        need(f, "f"); need(t1, "t1"); need(W, "W"); need(fd, "fd");
        E(ctx).dress_fock_finish(f, t1, W, fd);
    });
}
int pymes_pairs_supported(pymes_ctx* ctx, int* yes) {
    return guarded([&] {
        if (!yes) throw pymes::Error("null output");
        *yes = dev::fused_pair_kernels_ok(E(ctx).no) ? 1 : 0;
    });
}
int pymes_pairs_pack(pymes_ctx* ctx, const double* full, double* xc, int rank, int world) {
    return guarded([&] {
        need(full, "full"); need(xc, "xc");
        int64_t r0, r1;
        E(ctx).pair_chunk(rank, world, r0, r1);
        dev::pairs_pack(full, xc, E(ctx).no, E(ctx).nv, r0, r1, E(ctx).stream);
    });
}
int pymes_pairs_unpack(pymes_ctx* ctx, const double* xc_all, double* full, int world) {
    return guarded([&] {
        need(xc_all, "xc_all"); need(full, "full");
        pymes::Engine& e = E(ctx);
        const int64_t o2 = static_cast<int64_t>(e.no) * e.no;
        for (int r = 0; r < world; ++r) {       // chunk r of the exchanged buffer starts at r * chunk_rows
            int64_t r0, r1, c0, c1;
            e.pair_chunk(r, world, r0, r1);
            e.pair_chunk(0, world, c0, c1);
            dev::pairs_unpack(xc_all + static_cast<int64_t>(r) * (c1 - c0) * 2 * o2, full, e.no, e.nv, r0, r1, e.stream);
        }
    });
}
int pymes_cc_update_pairs(pymes_ctx* ctx, double* tc, double* dtc, const double* rc, double shift, double delta, int rank,
                          int world) {
    return guarded([&] {
        need(tc, "tc"); need(dtc, "dtc"); need(rc, "rc");
        pymes::Engine& e = E(ctx);
        int64_t r0, r1;
        e.need_eps("cc_update_pairs");
        e.pair_chunk(rank, world, r0, r1);
        dev::cc_update_pairs(tc, dtc, rc, e.eps_o, e.eps_v, shift, delta, e.no, e.nv, r0, r1, e.stream);
    });
}
int pymes_ccsd_dress_abcd_rows(pymes_ctx* ctx, const double* t1, int a0, int a1, int lower_only) {
    return guarded([&] {
        need(t1, "t1");
        E(ctx).dress_abcd_rows(t1, a0, a1, lower_only != 0);
    });
}
int pymes_cc_update(pymes_ctx* ctx, double* t, double* dt, const double* r, double shift, double delta, int rank) {
    return guarded([&] {
        need(t, "t"); need(dt, "dt"); need(r, "r");
        E(ctx).cc_update(t, dt, r, shift, delta, rank);
    });
}
int pymes_cc_update_to(pymes_ctx* ctx, double* t_out, double* dt, const double* t_in, const double* r, double shift,
                       double delta, int rank) {
    return guarded([&] {
        need(t_out, "t_out"); need(dt, "dt"); need(t_in, "t_in"); need(r, "r");
        E(ctx).cc_update_to(t_out, dt, t_in, r, shift, delta, rank);
    });
}
int pymes_energy_norms(pymes_ctx* ctx, const double* f, const double* t1, const double* t2, const double* dt2,
                       double* out) {
    return guarded([&] {
        need(t2, "t2"); need(out, "out");
        if ((f == nullptr) != (t1 == nullptr)) throw pymes::Error("f and t1 must be given together");
        E(ctx).energy_norms(f, t1, t2, dt2, out);
    });
}
int pymes_energy_norms_start(pymes_ctx* ctx, const double* f, const double* t1, const double* t2, const double* dt2, int* slot) {
    return guarded([&] {
        need(t2, "t2"); need(slot, "slot");
        if ((f == nullptr) != (t1 == nullptr)) throw pymes::Error("f and t1 must be given together");
        *slot = E(ctx).energy_norms_start(f, t1, t2, dt2);
    });
}
int pymes_energy_norms_wait(pymes_ctx* ctx, int slot, double* out) {
    return guarded([&] {
        need(out, "out");
        Eq(ctx).energy_norms_wait(slot, out);        // (nothing is enqueued: an open group stays as it is)
    });
}
int pymes_readback_start(pymes_ctx* ctx, const double* dev_ptr, int n, int* slot) {
    return guarded([&] {
        need(dev_ptr, "dev_ptr"); need(slot, "slot");
        *slot = dev::readback_start(dev_ptr, n, E(ctx).stream);
    });
}
int pymes_readback_wait(pymes_ctx* ctx, int slot, double* out, int n) {
    return guarded([&] {
        need(out, "out");
        Eq(ctx);
        dev::readback_wait(slot, out, n);
    });
}
int pymes_energy_norms_pairs(pymes_ctx* ctx, const double* f, const double* t1, const double* tc, const double* dtc,
                             int rank, int world, double* out) {
    return guarded([&] {
        need(tc, "tc"); need(out, "out");
        if ((f == nullptr) != (t1 == nullptr)) throw pymes::Error("f and t1 must be given together");
        E(ctx).energy_norms_pairs(f, t1, tc, dtc, rank, world, out);
    });
}
int pymes_set_collectives(pymes_ctx* ctx, const pymes_collectives* table) {
    return guarded([&] {
        if (!table) {
            E(ctx).set_collectives(nullptr);
            return;
        }
        pymes::Engine::Collectives c;
        c.user = table->user;
        c.rank = table->rank;
        c.world = table->world;
        c.allreduce_start = table->allreduce_start;
        c.allgather_start = table->allgather_start;
        c.wait = table->wait;
        c.mark = table->mark;
        E(ctx).set_collectives(&c);
    });
}
int pymes_shard_buffer_sizes(pymes_ctx* ctx, int world, int64_t* sizes) {
    return guarded([&] {
        if (!sizes) throw pymes::Error("null output");
        if (world < 1) throw pymes::Error("world must be >= 1");
        pymes::Engine& e = E(ctx);
        const int64_t o = e.no, v = e.nv, ov = o * v, npp = v * (v + 1) / 2;
        auto padded = [&](int64_t n) { return (n + world - 1) / world * world; };
        sizes[0] = sizes[1] = padded(ov) * ov;
        sizes[2] = padded(npp) * o * o;
        sizes[3] = padded(ov) * o * o;
        sizes[4] = padded(npp) * 2 * o * o;
        sizes[5] = e.dress_fock_ws_doubles();
        sizes[6] = v * v;
        sizes[7] = e.slab_prepare_ws_doubles();
        sizes[8] = v * o;
        sizes[9] = 8;
    });
}
namespace {
pymes::Engine::ShardBuffers shard_buffers(const pymes_shard_buffers* b, bool ccd = false) {
    if (!b) throw pymes::Error("null buffers");
    for (const double* p : {b->ETd, b->ETx, b->L, b->Tall, b->S})
        if (!p) throw pymes::Error("null exchange buffer");
    if (!ccd)
        for (const double* p : {b->QK, b->W, b->Xvv, b->P, b->R1})
            if (!p) throw pymes::Error("null exchange buffer");
    return pymes::Engine::ShardBuffers{b->ETd, b->ETx, b->L, b->QK, b->Tall, b->W, b->Xvv, b->P, b->R1, b->S};
}
}  // namespace
int pymes_ccsd_sharded_residuals(pymes_ctx* ctx, const double* f, double* fd, const double* t1, double* t2,
                                 const pymes_shard_buffers* buffers, uint32_t flags, double* rc) {
    return guarded([&] {
        need(f, "f"); need(fd, "fd"); need(t1, "t1"); need(t2, "t2"); need(rc, "rc");
        if (flags & ~(PYMES_DCD | PYMES_OWNER_TILES)) throw pymes::Error("sharded step: flags are PYMES_DCD and PYMES_OWNER_TILES");
        E(ctx).ccsd_sharded_residuals(f, fd, t1, t2, shard_buffers(buffers), flags, rc);
    });
}
int pymes_ccd_sharded_residuals(pymes_ctx* ctx, const double* f, double* t2, const pymes_shard_buffers* buffers, uint32_t flags,
                                double* rc) {
    return guarded([&] {
        need(f, "f"); need(t2, "t2"); need(rc, "rc");
        if (flags & ~(PYMES_DCD | PYMES_OWNER_TILES)) throw pymes::Error("sharded step: flags are PYMES_DCD and PYMES_OWNER_TILES");
        E(ctx).ccd_sharded_residuals(f, t2, shard_buffers(buffers, true), flags, rc);
    });
}
int pymes_set_alltoallv(pymes_ctx* ctx, pymes_alltoallv_fn fn) {
    return guarded([&] { E(ctx).set_alltoallv(fn); });
}
int pymes_owner_tile_sizes(pymes_ctx* ctx, int rank, int world, int64_t* send_doubles, int64_t* recv_doubles) {
    return guarded([&] { E(ctx).owner_tile_sizes(rank, world, send_doubles, recv_doubles); });
}
int pymes_set_owner_tile_buffers(pymes_ctx* ctx, double* send_dev, double* recv_dev) {
    return guarded([&] { E(ctx).set_owner_tile_buffers(send_dev, recv_dev); });
}
int pymes_ccsd_sharded_finish(pymes_ctx* ctx, const double* f, const double* t1, const double* tc, const double* dtc,
                              const pymes_shard_buffers* buffers, int* slot) {
    return guarded([&] {
        need(tc, "tc");                  // (f, t1 NULL: CCD / DCD — no one-body energy, no T1 norm)
        if ((f == nullptr) != (t1 == nullptr)) throw pymes::Error("sharded finish: f and t1 are given together or not at all");
        if (!slot) throw pymes::Error("null output");
        *slot = E(ctx).ccsd_sharded_finish(f, t1, tc, dtc, shard_buffers(buffers, t1 == nullptr));
    });
}
int pymes_ccsd_sharded_energy(pymes_ctx* ctx, int slot, double* out) {
    return guarded([&] {
        need(out, "out");
        E(ctx).ccsd_sharded_energy(slot, out);
    });
}
int pymes_ccsd_sharded_await(pymes_ctx* ctx, double* t2, const pymes_shard_buffers* buffers) {
    return guarded([&] {
        need(t2, "t2");
        E(ctx).ccsd_sharded_await(t2, shard_buffers(buffers, true));
    });
}
int pymes_ccsd_energy(pymes_ctx* ctx, const double* f, const double* t1, const double* t2, double* e_out) {
    return guarded([&] {
        need(f, "f"); need(t1, "t1"); need(t2, "t2"); need(e_out, "e_out");
        E(ctx).ccsd_energy(f, t1, t2, e_out);
    });
}
int pymes_ccd_energy(pymes_ctx* ctx, const double* t2, double* e_out) {
    return guarded([&] {
        need(t2, "t2"); need(e_out, "e_out");
        E(ctx).ccd_energy(t2, e_out);
    });
}

static void ueg_eval(pymes_ctx* ctx, int n_p, int n_ele, int imax, int mode, double L, double k_cutoff, double gamma,
                     int lattice_cutoff, const int32_t* k_int, const int32_t* index_map, const double* tab_scalar,
                     const double* tab_array, int64_t tab_len, double* V, int kind = 0, const double* params = nullptr) {
    Engine& e = E(ctx);
    need(k_int, "k_int"); need(index_map, "index_map"); need(V, "V");
    if (n_p < 1 || imax < 0 || mode < 0 || mode > 3) throw pymes::Error("ueg: bad arguments");
    if (tab_array) {       // largest |n|^2 any kernel looks up: lattice vector (<= lattice_cutoff) minus a momentum transfer (<= 2 imax)
        const int64_t r = static_cast<int64_t>(lattice_cutoff) + 2 * imax, need_len = 3 * r * r + 1;
        if (!tab_scalar || tab_len < need_len || tab_len > (int64_t(1) << 30)) throw pymes::Error("ueg: correlator tables must cover |n|^2 <= 3 (lattice_cutoff + 2 imax)^2");
    }
    const size_t m3 = size_t(2 * imax + 1) * (2 * imax + 1) * (2 * imax + 1);
    int* kd = static_cast<int*>(dev::dmalloc(sizeof(int) * 3 * n_p));
    int* md = static_cast<int*>(dev::dmalloc(sizeof(int) * m3));
    try {
        dev::memcpy_h2d(kd, k_int, sizeof(int) * 3 * n_p, e.stream);
        dev::memcpy_h2d(md, index_map, sizeof(int) * m3, e.stream);
        dev::UegParams p{n_p, n_ele, imax, mode, L, L * L * L, k_cutoff, gamma, lattice_cutoff};
        p.tab_scalar = tab_scalar; p.tab_array = tab_array; p.tab_len = static_cast<int>(tab_len);
        p.corr_kind = kind;
        for (int i = 0; i < 4 && params; ++i) p.corr_p[i] = params[i];
        dev::ueg_two_body(p, kd, md, V, e.stream);
    } catch (...) {
        dev::dfree(kd); dev::dfree(md);
        throw;
    }
    dev::dfree(kd); dev::dfree(md);
}
int pymes_ueg_eval_2b(pymes_ctx* ctx, int n_p, int n_ele, int imax, int mode, double L, double k_cutoff, double gamma,
                      int lattice_cutoff, const int32_t* k_int, const int32_t* index_map, double* V) {
    return guarded([&] {
        ueg_eval(ctx, n_p, n_ele, imax, mode, L, k_cutoff, gamma, lattice_cutoff, k_int, index_map, nullptr, nullptr, 0, V);
    });
}
int pymes_ueg_eval_2b_corr(pymes_ctx* ctx, int n_p, int n_ele, int imax, int mode, double L, int lattice_cutoff,
                           int correlator, const double* params, const int32_t* k_int, const int32_t* index_map, double* V) {
    return guarded([&] {
        need(params, "params");
        if (mode == 0) throw pymes::Error("ueg: mode 0 (Coulomb) has no correlator");
        if (correlator < 1 || correlator > 6) throw pymes::Error("ueg: correlator must be 1..6 (0 = trunc: pymes_ueg_eval_2b)");
        ueg_eval(ctx, n_p, n_ele, imax, mode, L, 0.0, 1.0, lattice_cutoff, k_int, index_map, nullptr, nullptr, 0, V, correlator, params);
    });
}
int pymes_ueg_eval_2b_tab(pymes_ctx* ctx, int n_p, int n_ele, int imax, int mode, double L, int lattice_cutoff,
                          const int32_t* k_int, const int32_t* index_map, const double* tab_scalar, const double* tab_array,
                          int64_t tab_len, double* V) {
    return guarded([&] {
        need(tab_scalar, "tab_scalar"); need(tab_array, "tab_array");
        if (mode == 0) throw pymes::Error("ueg: mode 0 (Coulomb) has no correlator");
        ueg_eval(ctx, n_p, n_ele, imax, mode, L, 0.0, 1.0, lattice_cutoff, k_int, index_map, tab_scalar, tab_array, tab_len, V);
    });
}

int pymes_dots(pymes_ctx* ctx, int npairs, const double* const* x, const double* const* y, int64_t n,
               double* out) {
    return guarded([&] {
        need(x, "x"); need(y, "y"); need(out, "out");
        if (npairs < 0 || npairs > 16) throw pymes::Error("dots: 0..16 pairs");
        int64_t len[16];
        for (int i = 0; i < npairs; ++i) len[i] = n;
        dev::dots(npairs, x, y, len, out, E(ctx).stream);
    });
}
int pymes_dots_var(pymes_ctx* ctx, int npairs, const double* const* x, const double* const* y, const int64_t* n,
                   double* out) {
    return guarded([&] {
        need(x, "x"); need(y, "y"); need(n, "n"); need(out, "out");
        dev::dots(npairs, x, y, n, out, E(ctx).stream);
    });
}
int pymes_lincomb(pymes_ctx* ctx, double* out, int nx, const double* const* x, const double* c, int64_t n) {
    return guarded([&] {
        need(out, "out"); need(x, "x"); need(c, "c");
        dev::lincomb(out, nx, x, c, n, E(ctx).stream);
    });
}

int pymes_gemm_group_begin(pymes_ctx* ctx) {
    return guarded([&] { dev::gemm_group_begin(Eq(ctx).stream); });
}
int pymes_gemm_group_end(pymes_ctx* ctx, int64_t* launches, int64_t* products) {
    return guarded([&] {
        Eq(ctx);
        dev::gemm_group_end();
        long l = 0, p = 0;
        dev::gemm_group_stats(&l, &p);
        if (launches) *launches = l;
        if (products) *products = p;
    });
}
int pymes_gram(pymes_ctx* ctx, int m, int n, const double* const* x, const double* const* y, int64_t len, double* out) {
    return guarded([&] {
        need(x, "x"); need(y, "y"); need(out, "out");
        if (m < 0 || n < 0 || len < 0) throw pymes::Error("gram: negative size");
        for (int i = 0; i < m; ++i) need(x[i], "x[i]");
        for (int j = 0; j < n; ++j) need(y[j], "y[j]");
        dev::gram(m, n, x, y, len, out, E(ctx).stream);
    });
}
int pymes_lincomb_multi(pymes_ctx* ctx, int m, int n, const double* const* x, const double* c, const double* beta,
                        double* const* y, int64_t len) {
    return guarded([&] {
        need(y, "y");
        if (m < 0 || n < 0 || len < 0) throw pymes::Error("lincomb_multi: negative size");
        if (m > 0) { need(x, "x"); need(c, "c"); }
        for (int i = 0; i < m; ++i) need(x[i], "x[i]");
        for (int j = 0; j < n; ++j) need(y[j], "y[j]");
        dev::lincomb_multi(m, n, x, c, beta, y, len, E(ctx).stream);
    });
}

int pymes_diis_step(pymes_ctx* ctx, double* state, int npairs, const double* const* x, const double* const* y, const int64_t* n,
                    int ntypes, int m, int was_full) {
    return guarded([&] {
        need(state, "state"); need(x, "x"); need(y, "y"); need(n, "n");
        dev::diis_step(state, npairs, x, y, n, ntypes, m, was_full, E(ctx).stream);
    });
}
int pymes_diis_solve(double* state_host, const double* overlaps_host, int ntypes, int m, int was_full) {
    return guarded([&] {
        need(state_host, "state"); need(overlaps_host, "overlaps");
        if (ntypes < 1 || m < 1 || m > 8) throw pymes::Error("diis_solve: need ntypes >= 1, 1 <= m <= 8");
        diis_small::step(state_host, overlaps_host, ntypes, m, was_full);
    });
}
int pymes_diis_mix(pymes_ctx* ctx, double* state_host, int ntypes, int m, int was_full, const double* const* err_hist,
                   const double* const* err_new, const int64_t* sizes, const double* const* amp_hist, double* const* out) {
    return guarded([&] {
        need(state_host, "state"); need(err_hist, "err_hist"); need(err_new, "err_new"); need(sizes, "sizes");
        need(amp_hist, "amp_hist"); need(out, "out");
        if (ntypes < 1 || m < 1 || m > 8 || ntypes * m > 16) throw pymes::Error("diis_mix: need ntypes * m <= 16, m <= 8");
        Engine& e = E(ctx);
        const double* x[16];
        const double* y[16];
        int64_t n[16];
        for (int t = 0; t < ntypes; ++t)
            for (int i = 0; i < m; ++i) {
                x[t * m + i] = err_hist[t * m + i];
                y[t * m + i] = err_new[t];
                n[t * m + i] = sizes[t];
            }
        double ov[16];
        dev::dots(ntypes * m, x, y, n, ov, e.stream);                      // one launch pair, one synchronisation
        diis_small::step(state_host, ov, ntypes, m, was_full);             // (m+1) x (m+1) algebra on this host thread
        if (state_host[91] == 2.0)      // singular or non-finite subspace matrix: the reference's numpy.linalg raises here too
            throw pymes::Error("DIIS: the subspace matrix is singular or not finite (numpy.linalg.LinAlgError in pymes/mixer/diis.py:85-95)");
        for (int t = 0; t < ntypes; ++t) dev::lincomb(out[t], m, amp_hist + t * m, state_host + 82, sizes[t], e.stream);
    });
}
int pymes_lincomb_dev(pymes_ctx* ctx, double* out, int nx, const double* const* x, const double* coeff_dev, int64_t n) {
    return guarded([&] {
        need(out, "out"); need(x, "x"); need(coeff_dev, "coeff");
        dev::lincomb_dev(out, nx, x, coeff_dev, n, E(ctx).stream);
    });
}
int pymes_cmul(pymes_ctx* ctx, const double* mr, const double* mi, const double* xr, const double* xi, double* yr, double* yi,
               int64_t n) {
    return guarded([&] {
        need(mr, "mr"); need(mi, "mi"); need(xr, "xr"); need(xi, "xi"); need(yr, "yr"); need(yi, "yi");
        dev::cmul(mr, mi, xr, xi, yr, yi, n, E(ctx).stream);
    });
}

int pymes_cshift_inv(pymes_ctx* ctx, const double* d, double zr, double zi, double hr, double hi, double shift, double* mr,
                     double* mi, int64_t n) {
    return guarded([&] {
        need(d, "d"); need(mr, "mr"); need(mi, "mi");
        dev::cshift_inv(d, zr, zi, hr, hi, shift, mr, mi, n, E(ctx).stream);
    });
}

// ---- EOM-CCSD sigma (eom.cpp) ---------------------------------------------------------------------------------------------------
int pymes_eom_sigma_prepare(pymes_ctx* ctx, const double* f_host, const double* t2, int dressed, pymes_eom** out) {
    return guarded([&] {
        need(out, "out");
        *out = nullptr;
        need(f_host, "f_host"); need(t2, "t2");
        Engine& e = E(ctx);
        if (e.capturing()) throw pymes::Error("eom_sigma_prepare while a launch graph is being recorded");
        pymes::EomSigma* s = new pymes::EomSigma(e, f_host, t2, dressed != 0);
        *out = new pymes_eom{s, ctx};
        std::lock_guard<std::mutex> lock(g_eom_mu);
        g_eom_live.insert(*out);
    });
}
int pymes_eom_sigma_flags(pymes_eom* h, int* flags) {
    return guarded([&] {
        need(flags, "flags");
        *flags = S(h).flags();
    });
}
int pymes_eom_sigma_apply(pymes_eom* h, int k, const double* const* u1, const double* const* u2, const int* sym,
                          double* const* s1, double* const* s2) {
    return guarded([&] {
        need(u1, "u1"); need(u2, "u2"); need(s1, "s1"); need(s2, "s2");
        if (k < 0 || k > 4096) throw pymes::Error("eom_sigma_apply: 0 <= k <= 4096");
        for (int z = 0; z < k; ++z)
            if (u2[z] == s2[z] || u1[z] == s1[z]) throw pymes::Error("eom_sigma_apply: output aliases input");
        S(h).apply(k, u1, u2, sym, s1, s2);
    });
}
int pymes_eom_diagonals(pymes_ctx* ctx, const double* f_host, const double* t2, int dressed, double* d1, double* d2) {
    return guarded([&] {
        need(f_host, "f_host"); need(t2, "t2"); need(d1, "d1"); need(d2, "d2");
        pymes::eom_diagonals(E(ctx), f_host, t2, dressed != 0, d1, d2);
    });
}
int pymes_scratch_trim(pymes_ctx* ctx) {
    return guarded([&] { E(ctx).scratch_trim(); });
}
int pymes_eom_sigma_destroy(pymes_eom* h) {
    return guarded([&] {
        if (!h) return;
        {
            std::lock_guard<std::mutex> lock(g_eom_mu);
            g_eom_live.erase(h);
        }
        if (h->ctx && h->ctx->e) dev::set_device(h->ctx->e->device);
        delete h->s;                         // (null when the context went first: pymes_ctx_destroy)
        delete h;
    });
}

int pymes_stats(pymes_ctx* ctx, int reset, int64_t* gemm_calls, double* gemm_flops, int64_t* permute_calls,
                double* permute_bytes) {
    return guarded([&] {
        Engine& e = E(ctx);
        if (gemm_calls) *gemm_calls = e.stats.gemm_calls;
        if (gemm_flops) *gemm_flops = e.stats.gemm_flops;
        if (permute_calls) *permute_calls = e.stats.permute_calls;
        if (permute_bytes) *permute_bytes = e.stats.permute_bytes;
        if (reset) e.stats = pymes::ContractStats{};
    });
}
int pymes_hf_fock_matrix(pymes_ctx* ctx, const double* h_host, double* f_host) {
    return guarded([&] {
        need(h_host, "h"); need(f_host, "f");
        E(ctx).hf_fock_matrix(h_host, f_host);
    });
}
int pymes_fcidump_header(const char* path, int* n_elec, int* n_orb) {
    return guarded([&] {
        if (!path || !n_elec || !n_orb) throw pymes::Error("null argument");
        pymes::FcidumpFile f;
        pymes::parse_fcidump(path, f, true);
        *n_elec = f.n_elec;
        *n_orb = f.n_orb;
    });
}
int pymes_fcidump_read_host(const char* path, int is_tc, double* e_core, double* eps, double* h, double* V) {
    return guarded([&] {
        if (!path || !e_core || !eps || !h || !V) throw pymes::Error("null argument");
        pymes::FcidumpFile f;
        pymes::parse_fcidump(path, f);
        const size_t n = static_cast<size_t>(f.n_orb);
        *e_core = f.e_core;
        std::copy(f.eps.begin(), f.eps.end(), eps);
        std::copy(f.h.begin(), f.h.end(), h);
        std::fill(V, V + n * n * n * n, 0.0);
        pymes::fill_V_host(f, is_tc != 0, V);
    });
}
int pymes_fcidump_load(pymes_ctx* ctx, const char* path, int is_tc, double* e_core, double* eps, double* h,
                       int64_t* n_lines) {
    return guarded([&] {
        if (!path || !e_core || !eps || !h) throw pymes::Error("null argument");
        pymes::Engine& e = E(ctx);
        pymes::FcidumpFile f;
        pymes::parse_fcidump(path, f);
        if (f.n_orb != e.n) throw pymes::Error("FCIDUMP NORB does not match the context (no + nv)");
        *e_core = f.e_core;
        std::copy(f.eps.begin(), f.eps.end(), eps);
        std::copy(f.h.begin(), f.h.end(), h);
        if (n_lines) *n_lines = static_cast<int64_t>(f.val.size());
        const int64_t nn = e.n, n4 = nn * nn * nn * nn;
        if (nn <= 64) {       // small: the reference's sequential fill, exact for any (even inconsistent) file
            std::vector<double> V(static_cast<size_t>(n4), 0.0);
            pymes::fill_V_host(f, is_tc != 0, V.data());
            e.set_V_full(V.data(), false, nullptr);
            return;
        }
        // large: V_pqrs never exists on the host
        double* Vd = static_cast<double*>(dev::dmalloc(sizeof(double) * n4));
        try {
            dev::memset_zero(Vd, sizeof(double) * n4, e.stream);
            const int64_t bad = dev::fcidump_fill(Vd, f.val.data(), f.pqrs.data(), static_cast<int64_t>(f.val.size()),
                                                  e.n, is_tc != 0, e.stream);
            if (bad)
                throw pymes::Error(std::to_string(bad) + " FCIDUMP lines disagree with their symmetry images: the result "
                                   "would depend on the order of the lines (not supported for NORB > 64)");
            e.set_V_full(Vd, true, nullptr);
            dev::stream_sync(e.stream);
        } catch (...) {
            dev::dfree(Vd);
            throw;
        }
        dev::dfree(Vd);
    });
}
// ---- packed binary integral files (packed.h) -------------------------------------------------------------------
int pymes_packed_header(const char* path, int* kind, int* n_elec, int* n_orb, int* naux) {
    return guarded([&] {
        if (!path || !kind || !n_elec || !n_orb || !naux) throw pymes::Error("null argument");
        pymes::PackedReader r(path);
        *kind = r.head.kind; *n_elec = r.head.n_elec; *n_orb = r.head.n_orb; *naux = r.head.naux;
    });
}
int pymes_packed_load(pymes_ctx* ctx, const char* path, double* e_core, double* eps, double* h) {
    return guarded([&] {
        if (!path || !e_core || !eps || !h) throw pymes::Error("null argument");
        pymes::Engine& e = E(ctx);
        pymes::PackedReader r(path);
        if (r.head.n_orb != e.n || r.head.n_occ != e.no) throw pymes::Error("packed file does not match the context (no, nv)");
        *e_core = r.head.e_core;
        std::copy(r.eps.begin(), r.eps.end(), eps);
        std::copy(r.h.begin(), r.h.end(), h);
        if (r.head.kind == pymes::kPackedFactors) {
            std::vector<double> B(static_cast<size_t>(r.head.payload_doubles));
            r.read(B.data(), r.head.payload_doubles);
            e.set_V_from_factors(B.data(), r.head.naux);
            return;
        }
        // blocks: file -> staging buffer -> device block, 32 Mi doubles at a time (V_pqrs never exists as a whole)
        const uint64_t chunk = uint64_t(1) << 25;
        std::vector<double> stage(static_cast<size_t>(std::min<uint64_t>(chunk, r.head.payload_doubles)));
        for (int pat = 0; pat < 16; ++pat) {
            double* dst = e.ensure_block(pat);
            const uint64_t total = static_cast<uint64_t>(pymes::packed_block_doubles(pat, e.no, e.nv));
            for (uint64_t off = 0; off < total; off += chunk) {
                const uint64_t nb = std::min(chunk, total - off);
                r.read(stage.data(), nb);
                dev::memcpy_h2d(dst + off, stage.data(), sizeof(double) * nb, e.stream);
            }
        }
    });
}
int pymes_packed_write(pymes_ctx* ctx, const char* path, int n_elec, double e_core, const double* eps, const double* h) {
    return guarded([&] {
        if (!path || !eps || !h) throw pymes::Error("null argument");
        pymes::Engine& e = E(ctx);
        if (n_elec != 2 * e.no) throw pymes::Error("packed write: NELEC must be twice the context's nocc");
        pymes::PackedWriter w(path, pymes::kPackedBlocks, e.n, n_elec, 0, e_core, eps, h);
        const uint64_t chunk = uint64_t(1) << 25;
        std::vector<double> stage(static_cast<size_t>(std::min<uint64_t>(chunk, static_cast<uint64_t>(e.n) * e.n * e.n * e.n)));
        for (int pat = 0; pat < 16; ++pat) {
            const double* src = e.block(pat).p;          // throws if a block has not been set
            const uint64_t total = static_cast<uint64_t>(pymes::packed_block_doubles(pat, e.no, e.nv));
            for (uint64_t off = 0; off < total; off += chunk) {
                const uint64_t nb = std::min(chunk, total - off);
                dev::memcpy_d2h(stage.data(), src + off, sizeof(double) * nb, e.stream);
                w.write(stage.data(), nb);
            }
        }
        w.close();
    });
}
int pymes_packed_write_factors(const char* path, int n_elec, int n_orb, int naux, double e_core, const double* eps,
                               const double* h, const double* B) {
    return guarded([&] {
        if (!path || !eps || !h || !B) throw pymes::Error("null argument");
        if (naux < 1) throw pymes::Error("packed write: naux must be positive");
        pymes::PackedWriter w(path, pymes::kPackedFactors, n_orb, n_elec, naux, e_core, eps, h);
        w.write(B, static_cast<uint64_t>(naux) * n_orb * n_orb);
        w.close();
    });
}
int pymes_scatter(pymes_ctx* ctx, double* dst, uint64_t dst_elements, const int64_t* index_host,
                  const double* value_host, int64_t n) {
    return guarded([&] {
        need(dst, "dst");
        if (n > 0) { need(index_host, "index"); need(value_host, "value"); }
        for (int64_t t = 0; t < n; ++t)
            if (index_host[t] < 0 || (uint64_t)index_host[t] >= dst_elements) throw pymes::Error("scatter: index out of range");
        dev::scatter(dst, index_host, value_host, n, E(ctx).stream);
    });
}
int pymes_tc_single_contraction(pymes_ctx* ctx, const double* L, int nb, int no, double* D) {
    return guarded([&] {
        need(L, "L"); need(D, "D");
        if (nb <= 0 || no < 0 || no > nb) throw pymes::Error("tc contraction: bad nb/no");
        dev::tc_single_contraction(L, D, nb, no, E(ctx).stream);
    });
}
int pymes_tc_double_contraction(pymes_ctx* ctx, const double* L, int nb, int no, double* S) {
    return guarded([&] {
        need(L, "L"); need(S, "S");
        if (nb <= 0 || no < 0 || no > nb) throw pymes::Error("tc contraction: bad nb/no");
        dev::tc_double_contraction(L, S, nb, no, E(ctx).stream);
    });
}
int pymes_tc_triple_contraction(pymes_ctx* ctx, const double* L, int nb, int no, double* t0_host) {
    return guarded([&] {
        need(L, "L"); need(t0_host, "t0");
        if (nb <= 0 || no < 0 || no > nb) throw pymes::Error("tc contraction: bad nb/no");
        *t0_host = dev::tc_triple_contraction(L, nb, no, E(ctx).stream);
    });
}

int pymes_prof_enable(pymes_ctx* ctx, int on) {
    return guarded([&] {
        E(ctx);
        dev::prof_enable(on != 0);
    });
}
int pymes_prof_reset(pymes_ctx* ctx) {
    return guarded([&] {
        E(ctx);
        dev::prof_reset();
    });
}
int pymes_prof_query(pymes_ctx* ctx, int kernel_class, int64_t* calls, int64_t* kernel_launches, double* ms,
                     double* flops) {
    return guarded([&] {
        E(ctx);
        if (kernel_class != 0 && kernel_class != 1) throw pymes::Error("prof_query: kernel_class must be 0 or 1");
        long c = 0, l = 0;
        double m = 0, f = 0;
        dev::prof_query(kernel_class, &c, &l, &m, &f);
        if (calls) *calls = c;
        if (kernel_launches) *kernel_launches = l;
        if (ms) *ms = m;
        if (flops) *flops = f;
    });
}

}  // extern "C"
