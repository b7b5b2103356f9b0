// HOST SIMULATOR of pymes_amd/csrc/device_api.h — TEST INFRASTRUCTURE ONLY.
//
// Plain CPU loops standing in for the HIP kernels so that the HOST logic of the
// engine (contraction planner, CC term sequencing, arena, C-ABI) can be checked
// against the oracle under `pytest -m "not gpu"` (and under ASan/valgrind), where no
// GPU exists.  It is built into tests/hostsim/_build/libpymes_hostsim.so by
// tests/hostsim/Makefile, is never linked into pymes_amd/lib/libpymes_amd.so, and the
// package loader (pymes_amd/_lib.py) rejects any library whose pymes_backend() is not
// "hip-gfx950".  It says nothing about the kernels themselves: those are tested on the
// GPU (tests/test_gpu_*.py).
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "../../pymes_amd/csrc/device_api.h"
#include "../../pymes_amd/csrc/diis_small.h"

namespace dev {

const char* backend_name() { return "hostsim"; }
void set_device(int) {}
static std::atomic<long> g_live{0};      // contexts of several host threads share the allocator (tests/test_collective_hook.py)
void* dmalloc(size_t bytes) {
    void* p = std::malloc(bytes ? bytes : 16);
    if (!p) throw std::runtime_error("hostsim: out of memory");
    std::memset(p, 0xA5, bytes ? bytes : 16);   // poison: uninitialised reads show up as garbage
    ++g_live;
    return p;
}
void* try_dmalloc(size_t bytes) {
    // failure injection for the out-of-memory paths of the host logic: requests above the limit "do not fit"
    if (const char* e = std::getenv("PYMES_HOSTSIM_ALLOC_LIMIT"))
        if (bytes > (size_t)std::atoll(e)) return nullptr;
    return dmalloc(bytes);
}
void dfree(void* p) {
    if (!p) return;
    std::free(p);
    --g_live;
}
int64_t live_allocations() { return g_live; }
stream_t stream_create() { return nullptr; }
void stream_destroy(stream_t) {}
event_t event_create() { return nullptr; }
void event_destroy(event_t) {}
void event_record(event_t, stream_t) {}
void stream_wait_event(stream_t, event_t) {}
// no launch graphs on the host: the solvers run their loop body eagerly
bool graphs_supported() { return false; }
void graph_begin(stream_t) { throw std::runtime_error("hostsim: no launch graphs"); }
graph_t graph_end(stream_t) { throw std::runtime_error("hostsim: no launch graphs"); }
void graph_abort(stream_t) {}
void graph_launch(graph_t, stream_t) { throw std::runtime_error("hostsim: no launch graphs"); }
void graph_destroy(graph_t) {}
void memcpy_h2d(void* d, const void* h, size_t n, stream_t) { std::memcpy(d, h, n); }
void memcpy_d2h(void* h, const void* d, size_t n, stream_t) { std::memcpy(h, d, n); }
void memcpy_d2d(void* d, const void* s, size_t n, stream_t) { std::memmove(d, s, n); }
void memset_zero(void* d, size_t n, stream_t) { std::memset(d, 0, n); }
void stream_sync(stream_t) {}
size_t mem_free_bytes() { return size_t(1) << 34; }
size_t mem_total_bytes() { return size_t(1) << 35; }

static thread_local long g_launches = 0;
static thread_local double g_flops = 0;
static thread_local bool g_prof = false;
void prof_enable(bool on) { g_prof = on; }
void prof_reset() { g_launches = 0; g_flops = 0; }
void prof_query(int, long* l, long* nk, double* ms, double* f) { *l = g_launches; *nk = g_launches; *ms = 0.0; *f = g_flops; }

void gemv_batch_begin() {}          // the simulator executes every product at once
void gemv_batch_end() {}

void gemm(const Gemm& g, stream_t) {
    if (!((g.a_sm == 1 || g.a_sk == 1 || g.M == 1 || g.K == 1) && (g.b_sk == 1 || g.b_sn == 1 || g.N == 1 || g.K == 1)))
        throw std::runtime_error("hostsim gemm: operand without unit stride");
    for (int64_t z1 = 0; z1 < g.nb1; ++z1)
        for (int64_t z2 = 0; z2 < g.nb2; ++z2) {
            const double* A = g.A + z1 * g.a_b1 + z2 * g.a_b2;
            const double* B = g.B + z1 * g.b_b1 + z2 * g.b_b2;
            double* C = g.C + z1 * g.c_b1 + z2 * g.c_b2;
            const double* Cin = (g.Cin ? g.Cin : g.C) + z1 * g.c_b1 + z2 * g.c_b2;
            for (int64_t m = 0; m < g.M; ++m)
                for (int64_t n = 0; n < g.N; ++n) {
                    double s = 0.0;
                    for (int64_t k = 0; k < g.K; ++k) s += A[m * g.a_sm + k * g.a_sk] * B[k * g.b_sk + n * g.b_sn];
                    C[m * g.ldc + n] = g.beta != 0.0 ? g.alpha * s + g.beta * Cin[m * g.ldc + n] : g.alpha * s;
                }
        }
    if (g_prof) { g_launches++; g_flops += 2.0 * g.M * g.N * g.K * g.nb1 * g.nb2; }
}

void permute(const Permute& p, stream_t) {
    int64_t total = 1;
    for (int i = 0; i < p.rank; ++i) total *= p.dim[i];
    // out may alias in only for identical index maps (axpby on itself); buffer otherwise-unsafe cases
    std::vector<double> tmp;
    const double* in = p.in;
    for (int64_t idx = 0; idx < total; ++idx) {
        int64_t rem = idx, oi = 0, oo = 0;
        for (int d = p.rank - 1; d >= 0; --d) {
            const int64_t c = rem % p.dim[d];
            rem /= p.dim[d];
            oi += c * p.s_in[d];
            oo += c * p.s_out[d];
        }
        double v = p.alpha * in[oi];
        if (p.beta != 0.0) v += p.beta * p.out[oo];
        p.out[oo] = v;
    }
}

void mp2_amplitudes(double* t, const double* w, const double* eo, const double* ev, double shift, int no, int nv,
                    stream_t) {
    int64_t idx = 0;
    for (int a = 0; a < nv; ++a)
        for (int b = 0; b < nv; ++b)
            for (int i = 0; i < no; ++i)
                for (int j = 0; j < no; ++j, ++idx) t[idx] = w[idx] / ((eo[i] + eo[j] - ev[a] - ev[b]) + shift);
}

void cc_update_to(double* t, double* dt, const double* tin, const double* r, const double* eo, const double* ev, double shift,
                  double delta, int no, int nv, int rank, stream_t) {
    int64_t idx = 0;
    if (rank == 4) {
        for (int a = 0; a < nv; ++a)
            for (int b = 0; b < nv; ++b)
                for (int i = 0; i < no; ++i)
                    for (int j = 0; j < no; ++j, ++idx) {
                        const double inv = 1.0 / ((eo[i] + eo[j] - ev[a] - ev[b]) + shift);
                        dt[idx] = r[idx] * inv;
                        t[idx] = tin[idx] + delta * dt[idx];
                    }
    } else {
        for (int a = 0; a < nv; ++a)
            for (int i = 0; i < no; ++i, ++idx) {
                const double inv = 1.0 / ((eo[i] - ev[a]) + shift);
                dt[idx] = r[idx] * inv;
                t[idx] = tin[idx] + delta * dt[idx];
            }
    }
}
void cc_update(double* t, double* dt, const double* r, const double* eo, const double* ev, double shift, double delta,
               int no, int nv, int rank, stream_t s) {
    cc_update_to(t, dt, t, r, eo, ev, shift, delta, no, nv, rank, s);
}

void dots(int npairs, const double* const* x, const double* const* y, const int64_t* n, double* out, stream_t) {
    if (npairs > 16) throw std::runtime_error("dots: at most 16 pairs per call");
    for (int p = 0; p < npairs; ++p) {
        double s = 0.0;
        for (int64_t i = 0; i < n[p]; ++i) s += x[p][i] * y[p][i];
        out[p] = s;
    }
}

void energy_norms(const double* f, const double* t1, const double* t2, const double* Edir, const double* Eex,
                  const double* dt2, int no, int nv, double out[6], stream_t) {
    for (int q = 0; q < 6; ++q) out[q] = 0.0;
    int64_t idx = 0;
    for (int a = 0; a < nv; ++a)
        for (int b = 0; b < nv; ++b)
            for (int i = 0; i < no; ++i)
                for (int j = 0; j < no; ++j, ++idx) {
                    const double tau = t2[idx] + (t1 ? t1[a * no + i] * t1[b * no + j] : 0.0);
                    out[1] += tau * Edir[idx];
                    out[2] += tau * Eex[idx];
                    out[3] += t2[idx] * t2[idx];
                    if (dt2) out[4] += dt2[idx] * dt2[idx];
                }
    if (t1 && f)
        for (int a = 0; a < nv; ++a)
            for (int i = 0; i < no; ++i) {
                out[0] += f[(int64_t)i * (no + nv) + no + a] * t1[a * no + i];
                out[5] += t1[a * no + i] * t1[a * no + i];
            }
}

void energy_norms_pairs(const double* f, const double* t1, const double* tc, const double* Edir, const double* Eex,
                        const double* dtc, int no, int nv, int64_t r0, int64_t r1, bool with_t1, double out[6], stream_t) {
    for (int q = 0; q < 6; ++q) out[q] = 0.0;
    const int64_t o2 = (int64_t)no * no;
    for (int64_t r = r0; r < r1; ++r) {
        int a = 0;
        while ((int64_t)(a + 1) * (a + 2) / 2 <= r) ++a;
        const int b = (int)(r - (int64_t)a * (a + 1) / 2);
        for (int half = 0; half < 2; ++half) {
            if (half && a == b) continue;
            const int p = half ? b : a, q = half ? a : b;
            const double* x = tc + ((r - r0) * 2 + half) * o2;
            const double* d = dtc ? dtc + ((r - r0) * 2 + half) * o2 : nullptr;
            for (int i = 0; i < no; ++i)
                for (int j = 0; j < no; ++j) {
                    const int64_t e = (int64_t)i * no + j, full = ((int64_t)p * nv + q) * o2 + e;
                    const double tau = x[e] + (t1 ? t1[p * no + i] * t1[q * no + j] : 0.0);
                    out[1] += tau * Edir[full];
                    out[2] += tau * Eex[full];
                    out[3] += x[e] * x[e];
                    if (d) out[4] += d[e] * d[e];
                }
        }
    }
    if (with_t1 && t1 && f)
        for (int a = 0; a < nv; ++a)
            for (int i = 0; i < no; ++i) {
                out[0] += f[(int64_t)i * (no + nv) + no + a] * t1[a * no + i];
                out[5] += t1[a * no + i] * t1[a * no + i];
            }
}

void energy_norms_pairs_dev(const double* f, const double* t1, const double* tc, const double* Edir, const double* Eex,
                            const double* dtc, int no, int nv, int64_t r0, int64_t r1, bool with_t1, double* out_dev,
                            stream_t s) {
    energy_norms_pairs(f, t1, tc, Edir, Eex, dtc, no, nv, r0, r1, with_t1, out_dev, s);       // "device" memory is host memory here
}

void exchange_asymmetry(const double* A, const double* B, const int64_t d[4], double out[2], stream_t) {
    out[0] = out[1] = 0.0;
    for (int64_t p = 0; p < d[0]; ++p)
        for (int64_t q = 0; q < d[1]; ++q)
            for (int64_t r = 0; r < d[2]; ++r)
                for (int64_t s = 0; s < d[3]; ++s) {
                    const double a = A[((p * d[1] + q) * d[2] + r) * d[3] + s], b = B[((q * d[0] + p) * d[3] + s) * d[2] + r];
                    const double df = std::fabs(a - b);
                    if (df > out[0] || df != df) out[0] = (df != df) ? INFINITY : df;
                    out[1] = std::fmax(out[1], std::fabs(a));
                }
}

// read-backs: the simulator is synchronous — the values are copied at start and handed over at wait
static thread_local double g_read_slots[16][128];      // (one context per host thread: a thread's tickets are its own)
static thread_local unsigned g_read_gen[16] = {0};
static thread_local int g_read_next = 0;
// (ticket = slot | generation << 8, as the HIP backend: a reused or never-started slot is refused)
int readback_start(const double* dev_ptr, int n, stream_t) {
    if (n < 1 || n > 128) throw std::runtime_error("readback: 1..128 doubles");
    const int slot = g_read_next;
    g_read_next = (slot + 1) % 16;
    const unsigned gen = (g_read_gen[slot] = (g_read_gen[slot] + 1) & 0x3fffffu);
    for (int i = 0; i < n; ++i) g_read_slots[slot][i] = dev_ptr[i];
    return (int)((gen << 8) | (unsigned)slot);
}
void readback_wait(int ticket, double* out, int n) {
    const int slot = ticket & 0xff;
    const unsigned gen = (unsigned)ticket >> 8;
    if (ticket < 0 || slot >= 16 || n < 1 || n > 128) throw std::runtime_error("readback: bad ticket");
    if (gen == 0 || g_read_gen[slot] != gen)
        throw std::runtime_error("readback: the slot was never started or has been reused since (16 later read-backs)");
    for (int i = 0; i < n; ++i) out[i] = g_read_slots[slot][i];
}
int energy_norms_start(const double* f, const double* t1, const double* t2, const double* Edir, const double* Eex,
                       const double* dt2, int no, int nv, stream_t s) {
    double out[6];
    energy_norms(f, t1, t2, Edir, Eex, dt2, no, nv, out, s);
    return readback_start(out, 6, s);
}

void lincomb(double* out, int nx, const double* const* x, const double* c, int64_t n, stream_t) {
    if (nx < 0 || nx > 8) throw std::runtime_error("lincomb: at most 8 terms");
    for (int64_t i = 0; i < n; ++i) {
        double s = 0.0;
        for (int k = 0; k < nx; ++k) s += x[k][i] * c[k];
        out[i] = s;
    }
}

// grouped launches: the simulator runs every product at once, so a group is only counted
static thread_local long g_group_launches = 0, g_group_products = 0;      // per thread, as the device backend's group state
static thread_local bool g_group_open = false;
void gemm_group_begin(stream_t) { g_group_open = true; g_group_launches = g_group_products = 0; }
void gemm_group_end() { g_group_open = false; }
void gemm_group_sync() {}
void phase_sync() {}
bool phase_pending() { return false; }
long phase_generation() { return 0; }
void phase_enable(int) {}
void phase_hold(bool) {}
void phase_call_end() {}
void phase_stats(long* tasks, long* launches, long* levels, long* flushes) {
    if (tasks) *tasks = 0;
    if (launches) *launches = 0;
    if (levels) *levels = 0;
    if (flushes) *flushes = 0;
}
void gemm_group_stats(long* launches, long* products) {
    if (launches) *launches = g_group_launches;
    if (products) *products = g_group_products;
}

void gram(int m, int n, const double* const* x, const double* const* y, int64_t len, double* out, stream_t) {
    if (m > 64 || n > 64) throw std::runtime_error("gram: at most 64 x 64 vectors per call");
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int64_t e = 0; e < len; ++e) s += x[i][e] * y[j][e];
            out[(int64_t)i * n + j] = s;
        }
}

void lincomb_multi(int m, int n, const double* const* x, const double* c, const double* beta, double* const* y, int64_t len,
                   stream_t) {
    if (m < 0 || m > 64 || n > 64) throw std::runtime_error("lincomb_multi: at most 64 inputs and 64 outputs per call");
    if (m > 16 || n > 4)
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < m; ++i)
                if (y[j] == x[i]) throw std::runtime_error("lincomb_multi: an output may alias an input only for m <= 16, n <= 4");
    std::vector<double> xs(m > 0 ? m : 1), out(n > 0 ? n : 1);
    for (int64_t e = 0; e < len; ++e) {           // element by element: every input is read before any output is written
        for (int i = 0; i < m; ++i) xs[i] = x[i][e];
        for (int j = 0; j < n; ++j) {
            double s = (beta && beta[j] != 0.0) ? beta[j] * y[j][e] : 0.0;
            for (int i = 0; i < m; ++i) s += c[(int64_t)i * n + j] * xs[i];
            out[j] = s;
        }
        for (int j = 0; j < n; ++j) y[j][e] = out[j];
    }
}

void diis_step(double* state, int npairs, const double* const* x, const double* const* y, const int64_t* n, int ntypes, int m,
               int was_full, stream_t) {
    if (npairs != ntypes * m || npairs > 16 || m + 1 > diis_small::kMaxOrder) throw std::runtime_error("diis_step: bad sizes");
    double ov[16];
    for (int p = 0; p < npairs; ++p) {
        double s = 0.0;
        for (int64_t i = 0; i < n[p]; ++i) s += x[p][i] * y[p][i];
        ov[p] = s;
    }
    diis_small::step(state, ov, ntypes, m, was_full);
}

void lincomb_dev(double* out, int nx, const double* const* x, const double* coeff, int64_t n, stream_t) {
    if (nx < 0 || nx > 8) throw std::runtime_error("lincomb_dev: at most 8 terms");
    for (int64_t i = 0; i < n; ++i) {
        double s = 0.0;
        for (int k = 0; k < nx; ++k) s += x[k][i] * coeff[k];
        out[i] = s;
    }
}

void exchange_split(const double* u, double* us, double* w, double* dg, int no, int nv, stream_t) {
    const long o = no, v = nv;
    for (long a = 0; a < v; ++a)
        for (long b = 0; b < v; ++b)
            for (long i = 0; i < o; ++i)
                for (long j = 0; j < o; ++j) {
                    const long e = ((a * v + b) * o + i) * o + j, p = ((b * v + a) * o + j) * o + i;
                    const double ua = 0.5 * (u[e] - u[p]);
                    us[e] = 0.5 * (u[e] + u[p]);
                    w[e] = i > j ? ua : (i < j ? -ua : 0.0);
                    if (i == j) dg[(a * v + b) * o + i] = ua;
                }
}
void sgn_ij_add(double* D, const double* R, int no, int nv, stream_t) {
    const long o = no, total = (long)nv * nv * o * o;
    for (long e = 0; e < total; ++e) {
        const long j = e % o, i = (e / o) % o;
        if (i != j) D[e] += (i > j ? R[e] : -R[e]);
    }
}
void cshift_inv(const double* d, double zr, double zi, double hr, double hi, double shift, double* mr, double* mi, int64_t n,
                stream_t) {
    for (int64_t e = 0; e < n; ++e) {
        const double x = zr - hr * d[e] + shift, y = zi - hi * d[e], q = x * x + y * y;
        mr[e] = x / q;
        mi[e] = -y / q;
    }
}
int64_t eom_diag_ws_doubles(int, int) { return 1; }
// eom_ccsd.py:169-266 as plain loops over the reference's einsum index lists (independent of the kernels' staging)
void eom_diagonals(const double* V, const double* T, const double* dai, const double* iaai, const double* iaia, const double* ijij,
                   const double* abab, double* d1, double* d2, int no, int nv, double*, stream_t) {
    const long o = no, v = nv;
    auto Vx = [&](long k, long l, long c, long d) { return V[((k * o + l) * v + c) * v + d]; };
    auto Tx = [&](long a, long b, long i, long j) { return T[((a * v + b) * o + i) * o + j]; };
    std::vector<double> sa(v, 0.0), si(o, 0.0), a_(v, 0.0), i_(o, 0.0), ij(o * o, 0.0), ab(v * v, 0.0), ai(v * o, 0.0),
        aj(v * o, 0.0), z1(v * o * o, 0.0), x1(v * v * o, 0.0);
    for (long a = 0; a < v; ++a)
        for (long j = 0; j < o; ++j)
            for (long k = 0; k < o; ++k)
                for (long b = 0; b < v; ++b) {
                    sa[a] += (2.0 * Vx(j, k, b, a) - Vx(j, k, a, b)) * Tx(a, b, j, k);
                    a_[a] += Vx(j, k, b, a) * (Tx(a, b, j, k) - 2.0 * Tx(b, a, j, k));
                }
    for (long i = 0; i < o; ++i)
        for (long j = 0; j < o; ++j)
            for (long c = 0; c < v; ++c)
                for (long b = 0; b < v; ++b) {
                    si[i] += Vx(j, i, c, b) * Tx(b, c, j, i);
                    i_[i] += (Vx(j, i, b, c) - 2.0 * Vx(j, i, c, b)) * Tx(c, b, j, i);
                    ij[i * o + j] += Vx(i, j, c, b) * Tx(c, b, i, j);
                }
    for (long a = 0; a < v; ++a)
        for (long b = 0; b < v; ++b)
            for (long k = 0; k < o; ++k)
                for (long l = 0; l < o; ++l) ab[a * v + b] += Vx(k, l, a, b) * Tx(a, b, k, l);
    for (long a = 0; a < v; ++a)
        for (long i = 0; i < o; ++i) {
            double s1 = 0.0, s2 = 0.0;
            for (long k = 0; k < o; ++k)
                for (long c = 0; c < v; ++c) {
                    s1 += (2.0 * Vx(k, i, c, a) - Vx(k, i, a, c)) * (2.0 * Tx(c, a, k, i) - Tx(a, c, k, i));
                    s2 += (2.0 * Vx(k, i, c, a) - 2.0 * Vx(k, i, a, c)) * Tx(c, a, k, i) +
                          (Vx(k, i, a, c) - 2.0 * Vx(k, i, c, a)) * Tx(a, c, k, i);
                    for (long b = 0; b < v; ++b) s2 += Vx(k, i, c, b) * Tx(a, c, k, i);
                    aj[a * o + i] += Vx(k, i, a, c) * Tx(a, c, k, i);
                }
            d1[a * o + i] = dai[a * o + i] + 2.0 * iaai[a * o + i] - iaia[a * o + i] + s1 - sa[a] - si[i];
            ai[a * o + i] = dai[a * o + i] + iaai[a * o + i] - 2.0 * iaia[a * o + i] + s2;
        }
    for (long a = 0; a < v; ++a)
        for (long i = 0; i < o; ++i)
            for (long j = 0; j < o; ++j)
                for (long k = 0; k < o; ++k)
                    for (long c = 0; c < v; ++c) z1[(a * o + i) * o + j] += Vx(k, j, a, c) * Tx(c, a, k, i);
    for (long a = 0; a < v; ++a)
        for (long b = 0; b < v; ++b)
            for (long j = 0; j < o; ++j)
                for (long k = 0; k < o; ++k) x1[(a * v + b) * o + j] += Vx(k, j, a, b) * Tx(a, b, k, j);
    auto half = [&](long a, long b, long i, long j) {
        double r = ai[a * o + i] + a_[a] + i_[i] - 2.0 * x1[(a * v + b) * o + j] - 2.0 * ij[i * o + j] + z1[(a * o + i) * o + j] +
                   aj[a * o + j];
        for (long k = 0; k < o; ++k) r += Vx(k, i, a, b) * Tx(a, b, k, j);
        for (long c = 0; c < v; ++c) r += Vx(i, j, c, a) * Tx(c, b, i, j);
        return r;
    };
    for (long a = 0; a < v; ++a)
        for (long b = 0; b < v; ++b)
            for (long i = 0; i < o; ++i)
                for (long j = 0; j < o; ++j)
                    d2[((a * v + b) * o + i) * o + j] = half(a, b, i, j) + half(b, a, j, i) + ijij[i * o + j] + ij[i * o + j] +
                                                        ab[a * v + b] + abab[a * v + b];
}
void cmul(const double* mr, const double* mi, const double* xr, const double* xi, double* yr, double* yi, int64_t n, stream_t) {
    for (int64_t e = 0; e < n; ++e) {
        const double a = xr[e], b = xi[e];
        yr[e] = mr[e] * a - mi[e] * b;
        yi[e] = mr[e] * b + mi[e] * a;
    }
}

void tau_build(double* tau, const double* t2, const double* t1, int no, int nv, stream_t) {
    int64_t idx = 0;
    for (int a = 0; a < nv; ++a)
        for (int b = 0; b < nv; ++b)
            for (int i = 0; i < no; ++i)
                for (int j = 0; j < no; ++j, ++idx) tau[idx] = t2[idx] + t1[a * no + i] * t1[b * no + j];
}

static inline int64_t P2(int64_t x, int64_t y) { return x * (x + 1) / 2 + y; }
static inline int64_t Q2(int64_t x, int64_t y) { return x * (x - 1) / 2 + y; }

void ladder_pack_V(const double* V, double* Vp, double* Vm, int nr, int nv, int64_t rp0, int64_t rp1, stream_t, int64_t ldvp,
                   int64_t ldvm) {
    const int64_t npp = ldvp ? ldvp : (int64_t)nv * (nv + 1) / 2, npm = ldvm ? ldvm : (int64_t)nv * (nv - 1) / 2;
    auto pack_row = [&](const double* Vab, int64_t r, bool diag) {
        for (int c = 0; c < nv; ++c)
            for (int d = 0; d <= c; ++d) {
                const double x1 = Vab[(int64_t)c * nv + d], x2 = Vab[(int64_t)d * nv + c];
                Vp[r * npp + P2(c, d)] = x1 + x2;
                if (c > d) Vm[r * npm + Q2(c, d)] = diag ? 0.0 : x1 - x2;
            }
    };
    if (nr == 0) {
        for (int64_t r = rp0; r < rp1; ++r) pack_row(V + r * nv * nv, r - rp0, false);
        return;
    }
    for (int a = 0; a < nr; ++a)
        for (int b = 0; b <= a; ++b) {
            const int64_t r = P2(a, b);
            if (r < rp0 || r >= rp1) continue;
            pack_row(V + ((int64_t)a * nr + b) * nv * nv, r - rp0, a == b);
        }
}

bool ladder_dress_ok(int no, int nv) { return no >= 1 && no <= 80 && nv >= 1; }

int64_t ladder_dress_ws_doubles(int no, int nv) { return (int64_t)((nv + 15) / 16) * ((no + 3) / 4) * 64; }

void ladder_dress(const double* V, const double* Pk, const double* t1, double* W, int no, int nv, int64_t ld, int64_t row0,
                  int64_t row1, double sgn, double*, stream_t) {
    if (!ladder_dress_ok(no, nv)) throw std::runtime_error("ladder_dress: nocc outside 1..64");
    for (int a = 0; a < nv; ++a)
        for (int b = 0; b <= a; ++b) {
            const int64_t r = P2(a, b);
            if (r < row0 || r >= row1) continue;
            const double* v = V + (r - row0) * ld;
            double* w = W + (r - row0) * ld;
            for (int64_t c = 0; c < ld; ++c) {
                double s1 = 0.0, s2 = 0.0;
                for (int k = 0; k < no; ++k) {
                    s1 += t1[(int64_t)a * no + k] * Pk[((int64_t)b * no + k) * ld + c];
                    s2 += t1[(int64_t)b * no + k] * Pk[((int64_t)a * no + k) * ld + c];
                }
                w[c] = (sgn > 0.0 && a == b) ? 0.0 : v[c] - s1 + sgn * s2;
            }
        }
}

void ladder_pack_T(const double* T, const double* t1, double* Sp, double* Am, int no, int nv, int flags, int64_t ldp,
                   int64_t ldm, stream_t, int64_t rp0, int64_t rp1) {
    if (rp1 < 0) { rp0 = 0; rp1 = (int64_t)nv * (nv + 1) / 2; }
    if ((rp0 != 0 || rp1 != (int64_t)nv * (nv + 1) / 2) && !(flags & PACK_AM_PROWS))
        throw std::runtime_error("ladder_pack_T: a row range needs PACK_AM_PROWS");
    auto skip = [&](int c, int d) { const int64_t r = P2(c, d); return r < rp0 || r >= rp1; };
    const bool row_half = flags & PACK_ROW_HALF, prow = flags & PACK_AM_PROWS, col_half = flags & PACK_COL_HALF,
               pcol = flags & PACK_AM_PCOLS;
    const int64_t o2 = (int64_t)no * no, opp = (int64_t)no * (no + 1) / 2, opm = (int64_t)no * (no - 1) / 2;
    if (!ldp) ldp = opp;
    if (!ldm) ldm = pcol ? opp : opm;
    for (int c = 0; c < nv; ++c)
        for (int d = 0; d <= c; ++d) {
            if (skip(c, d)) continue;
            if (ldp > opp) Sp[P2(c, d) * ldp + opp] = 0.0;
            if ((prow || c > d) && ldm > (pcol ? opp : opm)) Am[(prow ? P2(c, d) : Q2(c, d)) * ldm + (pcol ? opp : opm)] = 0.0;
        }
    for (int c = 0; c < nv; ++c)
        for (int d = 0; d <= c; ++d)
            for (int i = 0; i < no; ++i)
                for (int j = 0; j <= i; ++j) {
                    if (skip(c, d)) continue;
                    double x1 = 0.0, x2 = 0.0;
                    if (T) {
                        x1 = T[((int64_t)c * nv + d) * o2 + i * no + j];
                        x2 = T[((int64_t)d * nv + c) * o2 + i * no + j];
                    }
                    if (t1) {
                        x1 += t1[c * no + i] * t1[d * no + j];
                        x2 += t1[d * no + i] * t1[c * no + j];
                    }
                    double f = 0.5;
                    if (c == d && row_half) f *= 0.5;
                    if (i == j && col_half) f *= 0.5;
                    Sp[P2(c, d) * ldp + P2(i, j)] = f * (x1 + x2);
                    if ((prow || c > d) && (pcol || i > j))
                        Am[(prow ? P2(c, d) : Q2(c, d)) * ldm + (pcol ? P2(i, j) : Q2(i, j))] =
                            (c > d && i > j) ? 0.5 * (x1 - x2) : 0.0;
                }
}

bool fused_pair_kernels_ok(int no) { return no <= 3; }   // small, so that the tests reach both code paths

void t2_layouts(const double* T, double* Td, double* Tx, double* Ttd, int no, int nv, stream_t, double ca, double cb) {
    const int64_t o2 = (int64_t)no * no, ov = (int64_t)no * nv;
    for (int a = 0; a < nv; ++a)
        for (int b = 0; b < nv; ++b)
            for (int i = 0; i < no; ++i)
                for (int j = 0; j < no; ++j) {
                    const double x = T[((int64_t)a * nv + b) * o2 + i * no + j];
                    const double y = T[((int64_t)b * nv + a) * o2 + i * no + j];
                    if (Td) Td[((int64_t)a * no + i) * ov + b * no + j] = x;
                    Tx[((int64_t)a * no + j) * ov + b * no + i] = x;
                    Ttd[((int64_t)a * no + i) * ov + b * no + j] = ca * x + cb * y;
                }
}

void residual_assemble(const double* V, const double* L, const double* N, const double* D, const double* X, double* R,
                       int no, int nv, stream_t, double xd) {
    const int64_t o2 = (int64_t)no * no, ov = (int64_t)no * nv, opp = (int64_t)no * (no + 1) / 2;
    auto pm = [&](const double* M, int a, int i, int b, int j) { return M[((int64_t)a * no + i) * ov + b * no + j]; };
    for (int a = 0; a < nv; ++a)
        for (int b = 0; b < nv; ++b)
            for (int i = 0; i < no; ++i)
                for (int j = 0; j < no; ++j) {
                    const int64_t e = ((int64_t)a * nv + b) * o2 + i * no + j;
                    double v = (V ? V[e] : 0.0) + N[e] + N[((int64_t)b * nv + a) * o2 + j * no + i] + pm(D, a, i, b, j) +
                               pm(D, b, j, a, i) + pm(X, a, j, b, i) + pm(X, b, i, a, j);
                    if (xd != 0.0) v += xd * (pm(X, a, i, b, j) + pm(X, b, j, a, i));
                    if (L) {
                        const int ah = a > b ? a : b, al = a > b ? b : a, ih = i > j ? i : j, il = i > j ? j : i;
                        const double* row = L + P2(ah, al) * o2;
                        v += row[P2(ih, il)];
                        if (a != b && i != j) v += (((a > b) == (i > j)) ? 1.0 : -1.0) * row[opp + Q2(ih, il)];
                    }
                    R[e] = v;
                }
}

void hf_fock(const double* const dir[4], const double* const exc[4], const double* h, double* f, int no, int nv, stream_t) {
    const int n = no + nv;
    for (int p = 0; p < n; ++p)
        for (int q = 0; q < n; ++q) {
            const int tp = p >= no, tq = q >= no;
            const int64_t pl = tp ? p - no : p, ql = tq ? q - no : q, nq = tq ? nv : no;
            const double* D = dir[tp * 2 + tq];
            const double* X = exc[tp * 2 + tq];
            double acc = 0.0;
            for (int64_t i = 0; i < no; ++i)
                acc += 2.0 * D[((pl * no + i) * nq + ql) * no + i] - X[((pl * no + i) * no + i) * nq + ql];
            f[p * n + q] = h[p * n + q] + acc;
        }
}

int64_t fcidump_fill(double* V, const double* val, const int32_t* pqrs, int64_t count, int n_, bool is_tc, stream_t) {
    const int64_t n = n_;
    int64_t bad = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (int64_t t = 0; t < count; ++t) {
            const int64_t p = pqrs[4 * t], q = pqrs[4 * t + 1], r = pqrs[4 * t + 2], s = pqrs[4 * t + 3];
            int64_t tg[4];
            int m;
            if (is_tc) { tg[0] = ((q * n + p) * n + s) * n + r; tg[1] = ((p * n + q) * n + r) * n + s; m = 2; }
            else {
                tg[0] = ((p * n + q) * n + r) * n + s; tg[1] = ((r * n + q) * n + p) * n + s;
                tg[2] = ((r * n + s) * n + p) * n + q; tg[3] = ((p * n + s) * n + r) * n + q; m = 4;
            }
            bool ok = true;
            for (int i = 0; i < m; ++i) {
                if (pass == 0) V[tg[i]] = val[t];
                else ok = ok && V[tg[i]] == val[t];
            }
            if (!ok) ++bad;
        }
    return bad;
}

void scatter(double* dst, const int64_t* idx, const double* val, int64_t n, stream_t) {
    for (int64_t t = 0; t < n; ++t) dst[idx[t]] = val[t];
}
namespace {
inline double L6(const double* L, int nb, int a, int b, int c, int d, int e, int f) {
    return L[((((int64_t)a * nb + b) * nb + c) * nb + d) * nb * nb + (int64_t)e * nb + f];
}
}  // namespace
void tc_single_contraction(const double* L, double* D, int nb, int no, stream_t) {
    for (int p = 0; p < nb; ++p)
        for (int r = 0; r < nb; ++r)
            for (int q = 0; q < nb; ++q)
                for (int s = 0; s < nb; ++s) {
                    double acc = 0.0;
                    for (int i = 0; i < no; ++i)
                        acc += -3.0 * (L6(L, nb, p, q, r, i, i, s) + L6(L, nb, r, s, p, i, i, q)) + 6.0 * L6(L, nb, p, q, r, s, i, i);
                    D[(((int64_t)p * nb + r) * nb + q) * nb + s] = -acc / 3.0;
                }
}
void tc_double_contraction(const double* L, double* S, int nb, int no, stream_t) {
    for (int p = 0; p < nb; ++p)
        for (int q = 0; q < nb; ++q) {
            double acc = 0.0;
            for (int i = 0; i < no; ++i)
                for (int j = 0; j < no; ++j)
                    acc += 12.0 * L6(L, nb, i, i, j, j, p, q) - 12.0 * L6(L, nb, i, i, p, j, j, q) +
                           6.0 * L6(L, nb, p, i, j, q, i, j) - 6.0 * L6(L, nb, i, j, j, i, p, q);
            S[p * nb + q] = -acc / 6.0;
        }
}
double tc_triple_contraction(const double* L, int nb, int no, stream_t) {
    double acc = 0.0;
    for (int i = 0; i < no; ++i)
        for (int j = 0; j < no; ++j)
            for (int k = 0; k < no; ++k)
                acc += 8.0 * L6(L, nb, i, i, j, j, k, k) - 12.0 * L6(L, nb, i, j, j, i, k, k) + 4.0 * L6(L, nb, i, j, j, k, k, i);
    return -acc / 6.0;
}

namespace {
inline void unrank_pair_h(int64_t r, int& a, int& b) {
    a = 0;
    while ((int64_t)(a + 1) * (a + 2) / 2 <= r) ++a;
    b = (int)(r - (int64_t)a * (a + 1) / 2);
}
}  // namespace
void pairs_pack(const double* full, double* Xc, int no, int nv, int64_t r0, int64_t r1, stream_t) {
    const int64_t o2 = (int64_t)no * no;
    for (int64_t r = r0; r < r1; ++r) {
        int a, b;
        unrank_pair_h(r, a, b);
        for (int64_t e = 0; e < o2; ++e) {
            Xc[(r - r0) * 2 * o2 + e] = full[((int64_t)a * nv + b) * o2 + e];
            Xc[(r - r0) * 2 * o2 + o2 + e] = a != b ? full[((int64_t)b * nv + a) * o2 + e] : 0.0;
        }
    }
}
void pairs_unpack(const double* Xc, double* full, int no, int nv, int64_t r0, int64_t r1, stream_t) {
    const int64_t o2 = (int64_t)no * no;
    for (int64_t r = r0; r < r1; ++r) {
        int a, b;
        unrank_pair_h(r, a, b);
        for (int64_t e = 0; e < o2; ++e) {
            full[((int64_t)a * nv + b) * o2 + e] = Xc[(r - r0) * 2 * o2 + e];
            if (a != b) full[((int64_t)b * nv + a) * o2 + e] = Xc[(r - r0) * 2 * o2 + o2 + e];
        }
    }
}
void cc_update_pairs(double* tc, double* dtc, const double* rc, const double* eo, const double* ev, double shift,
                     double delta, int no, int nv, int64_t r0, int64_t r1, stream_t) {
    (void)nv;
    const int64_t o2 = (int64_t)no * no;
    for (int64_t r = r0; r < r1; ++r) {
        int a, b;
        unrank_pair_h(r, a, b);
        for (int64_t e = 0; e < 2 * o2; ++e) {
            const int64_t t = e % o2;
            const int i = (int)(t / no), j = (int)(t % no);
            const int64_t k = (r - r0) * 2 * o2 + e;
            const double x = rc[k] * (1.0 / (eo[i] + eo[j] - (ev[a] + ev[b]) + shift));
            dtc[k] = x;
            tc[k] += delta * x;
        }
    }
}
void residual_assemble_pairs(const double* V, const double* L, const double* Np, const double* D, const double* X,
                             double* Rc, int no, int nv, int64_t r0, int64_t r1, int a0, int nbp, stream_t, double xd) {
    const int64_t o2 = (int64_t)no * no, ov = (int64_t)no * nv, opp = (int64_t)no * (no + 1) / 2;
    auto pm = [&](const double* M, int a, int i, int b, int j) { return M[((int64_t)a * no + i) * ov + b * no + j]; };
    for (int64_t r = r0; r < r1; ++r) {
        int a, b;
        unrank_pair_h(r, a, b);
        const double* row = L ? L + r * o2 : nullptr;
        for (int i = 0; i < no; ++i)
            for (int j = 0; j < no; ++j) {
                auto S = [&](int x, int y) {
                    return Np[((int64_t)(a - a0) * nbp + b) * o2 + x * no + y] + pm(D, a, x, b, y) + pm(D, b, y, a, x) +
                           pm(X, a, y, b, x) + pm(X, b, x, a, y) + (xd != 0.0 ? xd * (pm(X, a, x, b, y) + pm(X, b, y, a, x)) : 0.0);
                };
                const int ih = i > j ? i : j, il = i > j ? j : i;
                double ls = 0.0, la = 0.0;
                if (row) {
                    ls = row[P2(ih, il)];
                    if (a != b && i != j) la = row[opp + Q2(ih, il)];
                }
                const double sgn = i > j ? 1.0 : -1.0;
                const int64_t e = (int64_t)i * no + j;
                Rc[(r - r0) * 2 * o2 + e] = V[((int64_t)a * nv + b) * o2 + e] + ls + sgn * la + S(i, j);
                Rc[(r - r0) * 2 * o2 + o2 + e] = a != b ? V[((int64_t)b * nv + a) * o2 + e] + ls - sgn * la + S(j, i) : 0.0;
            }
    }
}

void rows_unpack(const double* Q, double* out, int64_t rows, int no, stream_t) {
    const int64_t opp = (int64_t)no * (no + 1) / 2, ld = (int64_t)no * no;
    for (int64_t r = 0; r < rows; ++r)
        for (int i = 0; i < no; ++i)
            for (int j = 0; j < no; ++j) {
                const int ih = i > j ? i : j, il = i > j ? j : i;
                double v = Q[r * ld + P2(ih, il)];
                if (i != j) v += (i > j ? 1.0 : -1.0) * Q[r * ld + opp + Q2(ih, il)];
                out[r * ld + i * no + j] = v;
            }
}

bool fock_g12_ok(int nv) { return nv >= 1 && nv <= 1024; }
int64_t fock_g12_ws_doubles(int nv, int na, int) { return 2 * (int64_t)na * nv; }
void fock_g12(const double* V, const double* t1, double* G1, double* G2, int no, int nv, int na, int j0, int j1, double*, stream_t) {
    for (int a = 0; a < na; ++a)
        for (int c = 0; c < nv; ++c) {
            double s1 = 0.0, s2 = 0.0;
            for (int j = j0; j < j1; ++j)
                for (int b = 0; b < nv; ++b) {
                    const double t = t1[(int64_t)b * no + j];
                    s1 += t * V[(((int64_t)j * na + a) * nv + b) * nv + c];
                    s2 += t * V[(((int64_t)j * na + a) * nv + c) * nv + b];
                }
            G1[(int64_t)a * nv + c] = s1;
            G2[(int64_t)a * nv + c] = s2;
        }
}

void fock_finish(const double* f, const double* t1, const double* W, double* fd, double* ft, int no, int nv, stream_t) {
    const int64_t n = no + nv, vv = (int64_t)nv * nv, ov = (int64_t)no * nv, oo = (int64_t)no * no;
    const double *G1 = W, *G2 = G1 + vv, *J1 = G2 + vv, *J2 = J1 + ov, *L1 = J2 + ov, *L2 = L1 + oo, *K1 = L2 + oo, *K2 = K1 + ov;
    auto FM = [&](int i, int b) { return f[i * n + no + b] + 2.0 * J1[(int64_t)i * nv + b] - J2[(int64_t)i * nv + b]; };
    auto G = [&](int a, int b) { return 2.0 * G1[(int64_t)a * nv + b] - G2[(int64_t)a * nv + b]; };
    for (int j = 0; j < no; ++j)
        for (int i = 0; i < no; ++i) {
            double acc = 2.0 * L1[j * no + i] - L2[j * no + i];
            for (int b = 0; b < nv; ++b) acc += FM(j, b) * t1[(int64_t)b * no + i];
            ft[j * no + i] = acc;
        }
    for (int64_t e = 0; e < n * n; ++e) fd[e] = f[e];
    for (int i = 0; i < no; ++i)
        for (int j = 0; j < no; ++j) fd[i * n + j] += ft[i * no + j];
    for (int i = 0; i < no; ++i)
        for (int a = 0; a < nv; ++a) fd[i * n + no + a] += 2.0 * K1[(int64_t)i * nv + a] - J2[(int64_t)i * nv + a];
    for (int a = 0; a < nv; ++a)
        for (int b = 0; b < nv; ++b) {
            double acc = G(a, b);
            for (int i = 0; i < no; ++i) acc -= t1[(int64_t)a * no + i] * FM(i, b);
            fd[(no + a) * n + no + b] += acc;
        }
    for (int a = 0; a < nv; ++a)
        for (int i = 0; i < no; ++i) {
            double acc = 2.0 * K1[(int64_t)i * nv + a] - K2[(int64_t)a * no + i];
            for (int j = 0; j < no; ++j) acc -= t1[(int64_t)a * no + j] * (f[j * n + i] + ft[j * no + i]);
            for (int b = 0; b < nv; ++b) acc += (f[(no + a) * n + no + b] + G(a, b)) * t1[(int64_t)b * no + i];
            fd[(no + a) * n + i] += acc;
        }
}

void ring_operands(const double* Viabj, const double* Viajb, double* M, double* N1, double a1, double a2, int no, int nv,
                   stream_t) {
    const int64_t ov = (int64_t)no * nv;
    for (int k = 0; k < no; ++k)
        for (int b = 0; b < nv; ++b)
            for (int c = 0; c < nv; ++c)
                for (int j = 0; j < no; ++j) {
                    const double w = Viabj[(((int64_t)k * nv + b) * nv + c) * no + j];
                    const double u = Viajb[(((int64_t)k * nv + b) * no + j) * nv + c];
                    const int64_t off = ((int64_t)c * no + k) * ov + (int64_t)b * no + j;
                    N1[off] = -u;
                    M[off] = a1 * w - a2 * u;
                }
}

void pair_traces(const double* M, int64_t ld, double alpha, double beta, double* out_vv, double* out_oo, int no, int nv,
                 stream_t, const double* M2, double alpha2) {
    auto at = [&](int64_t e) { return alpha * M[e] + (M2 ? alpha2 * M2[e] : 0.0); };
    for (int a = 0; a < nv; ++a)
        for (int c = 0; c < nv; ++c) {
            double acc = 0.0;
            for (int k = 0; k < no; ++k) acc += at(((int64_t)c * no + k) * ld + (int64_t)a * no + k);
            double& o = out_vv[(int64_t)a * nv + c];
            o = (beta == 0.0 ? 0.0 : beta * o) + acc;
        }
    for (int k = 0; k < no; ++k)
        for (int i = 0; i < no; ++i) {
            double acc = 0.0;
            for (int c = 0; c < nv; ++c) acc += at(((int64_t)c * no + k) * ld + (int64_t)c * no + i);
            double& o = out_oo[(int64_t)k * no + i];
            o = (beta == 0.0 ? 0.0 : beta * o) + acc;
        }
}

void ladder_unpack(const double* L, double* R, double beta, int no, int nv, stream_t) {
    const int64_t opp = (int64_t)no * (no + 1) / 2, ld = (int64_t)no * no;
    int64_t idx = 0;
    for (int a = 0; a < nv; ++a)
        for (int b = 0; b < nv; ++b)
            for (int i = 0; i < no; ++i)
                for (int j = 0; j < no; ++j, ++idx) {
                    const int ah = a > b ? a : b, al = a > b ? b : a, ih = i > j ? i : j, il = i > j ? j : i;
                    const double* row = L + P2(ah, al) * ld;
                    double v = row[P2(ih, il)];
                    if (a != b && i != j) {
                        const double x = row[opp + Q2(ih, il)];
                        v += ((a > b) == (i > j)) ? x : -x;
                    }
                    R[idx] = beta != 0.0 ? beta * R[idx] + v : v;
                }
}

// ---- UEG integrals: plain loops with the same formulas as the HIP kernels ------------------------
namespace {
struct UegH { int n_p, n_occ, imax, m, mode, n_ele, lat; double L, Omega, kc2g, gamma; const double* ts = nullptr; const double* ta = nullptr; long tl = 0; int kind = 0; double p0 = 0, p1 = 0, p2 = 0; };
// m = |n|^2 of the integer vector behind x; arr: the reference passes an ndarray there (device_api.h UegParams)
inline double u_of(double x, long m, bool arr, const UegH& u) {
    if (u.ta) return m < u.tl ? (arr ? u.ta[m] : u.ts[m]) : 0.0;
    switch (u.kind) {       // the reference's named correlators (device_api.h UegParams::corr_kind)
        case 1: if (arr) return x > u.p1 ? -0.0 : (x > 1e-12 ? -(u.p0 / x) : -0.0);
                return (x < u.p1 && x > 1e-12) ? -(u.p0 / x) : -0.0;
        case 2: if (arr) return x >= u.p0 ? -((4.0 * M_PI) / (x * x)) : -0.0;
                return (x < u.p0 && x > 1e-12) ? -0.0 : -((4.0 * M_PI) / (x * x));
        case 3: return x > 1e-12 ? u.p0 / x : 0.0;
        case 4: { const double b = x + u.p0; return std::fabs(b) > u.p1 ? (-4.0 * M_PI) / b : 0.0; }
        case 5: { const double t = x + u.p0, b = t * t; return std::fabs(b) > u.p1 ? u.p2 / b : 0.0; }
        case 6: return x > u.p2 ? (-4.0 * M_PI * (1.0 + std::erf((std::sqrt(x) - u.p0) / u.p1)) / 2.0) / (x * x) : 0.0;
        default: break;
    }
    if (x <= u.kc2g) x = 0.0;
    return x > 1e-12 ? (-4.0 * M_PI / (x * x)) * u.gamma : 0.0;
}
inline double kp_of(int k, double L) { return ((double)(k * 2) * M_PI) / L; }
}  // namespace

void ueg_two_body(const UegParams& prm, const int* kint, const int* map, double* V, stream_t) {
    UegH u{prm.n_p, prm.n_ele / 2, prm.imax, 2 * prm.imax + 1, prm.mode, prm.n_ele, prm.lattice_cutoff, prm.L, prm.Omega, 0.0, prm.gamma};
    const double kc = prm.k_cutoff * 2 * M_PI / prm.L;
    u.kc2g = kc * kc * (1 + 0.00001);
    u.ts = prm.tab_scalar; u.ta = prm.tab_array; u.tl = prm.tab_len;
    u.kind = prm.corr_kind; u.p0 = prm.corr_p[0]; u.p1 = prm.corr_p[1]; u.p2 = prm.corr_p[2];
    const int64_t n = prm.n_p;
    std::memset(V, 0, sizeof(double) * n * n * n * n);
    std::vector<double> E(n * n, 0.0), umat(n * n, 0.0);
    for (int64_t p = 0; p < n; ++p)
        for (int64_t r = 0; r < n; ++r) {
            double kpv[3], kr[3], dk[3], dk2 = 0;
            long di[3], md = 0;
            for (int c = 0; c < 3; ++c) {
                kpv[c] = kp_of(kint[3 * p + c], u.L); kr[c] = kp_of(kint[3 * r + c], u.L); dk[c] = kr[c] - kpv[c]; dk2 += dk[c] * dk[c];
                di[c] = kint[3 * r + c] - kint[3 * p + c]; md += di[c] * di[c];
            }
            if (u.mode == 1) {
                bool done = false;   // reuse the value of an earlier pair with the same integer transfer
                for (int64_t p2 = 0; p2 <= p && !done; ++p2)
                    for (int64_t r2 = 0; r2 < n && !done; ++r2) {
                        if (p2 == p && r2 >= r) break;
                        bool same = true;
                        for (int c = 0; c < 3; ++c) same &= (kint[3 * r2 + c] - kint[3 * p2 + c]) == (kint[3 * r + c] - kint[3 * p + c]);
                        if (same) { umat[p * n + r] = umat[p2 * n + r2]; done = true; }
                    }
                if (!done) {
                    const int w = 2 * u.lat + 1;
                    double s = 0.0;
                    for (int a = 0; a < w; ++a) for (int b = 0; b < w; ++b) for (int c = 0; c < w; ++c) {
                        const double x1 = 2.0 * M_PI * (a - u.lat) / u.L, y1 = 2.0 * M_PI * (b - u.lat) / u.L, z1 = 2.0 * M_PI * (c - u.lat) / u.L;
                        const double x2 = dk[0] - x1, y2 = dk[1] - y1, z2 = dk[2] - z1;
                        const long a1 = a - u.lat, b1 = b - u.lat, c1 = c - u.lat, a2 = di[0] - a1, b2 = di[1] - b1, c2 = di[2] - c1;
                        s += (x1 * x2 + y1 * y2 + z1 * z2) * u_of(x1 * x1 + y1 * y1 + z1 * z1, a1 * a1 + b1 * b1 + c1 * c1, true, u) *
                             u_of(x2 * x2 + y2 * y2 + z2 * z2, a2 * a2 + b2 * b2 + c2 * c2, true, u);
                    }
                    umat[p * n + r] = s / u.Omega;
                }
            } else if (u.mode == 2) {
                const double udk = u_of(dk2, md, true, u), udk_s = u_of(dk2, md, false, u);
                double xr = 0, xp = 0, pk = 0;
                for (int o = 0; o < u.n_occ; ++o) {
                    double a2 = 0, ad = 0, b2 = 0, bd = 0, v12 = 0, v11 = 0;
                    long ma = 0, mb = 0, mv = 0;
                    for (int c = 0; c < 3; ++c) {
                        const double oc = kp_of(kint[3 * o + c], u.L);
                        const double a = kr[c] - oc, b = kpv[c] - oc, v1 = kr[c] - dk[c] - oc;
                        a2 += a * a; ad += a * dk[c]; b2 += b * b; bd += b * dk[c]; v12 += v1 * a; v11 += v1 * v1;
                        const long ai = kint[3 * r + c] - kint[3 * o + c], bi = kint[3 * p + c] - kint[3 * o + c], vi = ai - di[c];
                        ma += ai * ai; mb += bi * bi; mv += vi * vi;
                    }
                    xr += ad * udk * u_of(a2, ma, true, u); xp += bd * udk * u_of(b2, mb, true, u);
                    pk += v12 * u_of(v11, mv, true, u) * u_of(a2, ma, true, u);
                }
                xr /= u.Omega; xp /= u.Omega; pk /= u.Omega;
                const double val = std::fabs(dk2) > 0.0 ? -(double)u.n_ele * dk2 * udk_s * udk_s / u.Omega + 2.0 * xr - 2.0 * xp + 2.0 * pk : 2.0 * pk;
                E[p * n + r] = val / u.Omega;
            }
            for (int64_t q = 0; q < n; ++q) {
                int ks[3];
                for (int c = 0; c < 3; ++c) ks[c] = kint[3 * q + c] - (kint[3 * r + c] - kint[3 * p + c]);
                const int64_t loc = (int64_t)u.m * u.m * (ks[0] + u.imax) + (int64_t)u.m * (ks[1] + u.imax) + ks[2] + u.imax;
                if (loc < 0 || loc >= (int64_t)u.m * u.m * u.m) continue;
                const int s = map[loc];
                if (s < 0 || s >= u.n_p) continue;
                double w = 0.0;
                if (u.mode == 0) { if (std::fabs(dk2) > 0.0) w = 4.0 * M_PI / dk2 / u.Omega; }
                else if (u.mode == 3) { if (std::fabs(dk2) > 0.0) { const double x = u_of(dk2, md, false, u); w = -(double)u.n_ele * dk2 * x * x / u.Omega / u.Omega; } }
                else if (u.mode == 1) {
                    if (std::fabs(dk2) > 0.0) {
                        double rsdk = 0.0;
                        for (int c = 0; c < 3; ++c) rsdk += (kr[c] - kp_of(kint[3 * s + c], u.L)) * dk[c];
                        const double x = u_of(dk2, md, false, u);
                        w = (4.0 * M_PI / dk2 + umat[p * n + r] + dk2 * x - rsdk * x) / u.Omega;
                    } else w = umat[p * n + r] / u.Omega;
                } else w = E[p * n + r];
                V[((p * n + q) * n + r) * n + s] = w;
            }
        }
}

}  // namespace dev
