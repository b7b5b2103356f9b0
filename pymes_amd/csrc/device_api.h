// Device abstraction used by the host engine (engine.cpp).
//
// Product build: implemented by kernels.hip (hand-written gfx950 kernels).
// tests/hostsim/ implements the same functions with plain CPU loops so that the
// HOST logic (contraction planner, CC term sequencing, C-ABI plumbing) can be
// exercised under `pytest -m "not gpu"`, valgrind and ASan.  The host simulator is
// never linked into libpymes_amd.so and the Python package cannot load it.
#pragma once
#include <cstddef>
#include <cstdint>

namespace dev {

typedef void* stream_t;   // hipStream_t in the product build

const char* backend_name();
// throws std::runtime_error on failure
void  set_device(int ordinal);
void* dmalloc(size_t bytes);
void* try_dmalloc(size_t bytes);      // nullptr when the device is out of memory (the error state is cleared)
void  dfree(void* p);
void  memcpy_h2d(void* d, const void* h, size_t bytes, stream_t s);
void  memcpy_d2h(void* h, const void* d, size_t bytes, stream_t s);
void  memcpy_d2d(void* d, const void* s_, size_t bytes, stream_t s);
void  memset_zero(void* d, size_t bytes, stream_t s);
void  stream_sync(stream_t s);
size_t mem_free_bytes();
size_t mem_total_bytes();
// a stream of the engine's own (non-blocking with respect to the legacy default stream); null for the host simulator
stream_t stream_create();
void  stream_destroy(stream_t s);
// device allocations made through dmalloc and not yet released (leak accounting in the tests)
int64_t live_allocations();
// cross-stream ordering (the engine's side stream, engine.h): `stream_wait_event(s, e)` makes work enqueued on s later wait
// for what had been enqueued on the recording stream when e was recorded.  No-ops in the host simulator (one thread).
typedef void* event_t;
event_t event_create();
void  event_destroy(event_t e);
void  event_record(event_t e, stream_t s);
void  stream_wait_event(stream_t s, event_t e);

// ---- launch graphs: the launch-bound inner loop of a small problem is captured once and replayed -------------------
// graph_begin puts the stream into capture mode (work enqueued until graph_end is recorded, not executed);
// graph_end returns an executable graph.  Nothing that synchronises or frees may run while capturing.
typedef void* graph_t;
bool  graphs_supported();
void  graph_begin(stream_t s);
graph_t graph_end(stream_t s);          // throws (after leaving capture mode) if the capture was invalidated
void  graph_abort(stream_t s);          // leave capture mode, discard what was recorded
void  graph_launch(graph_t g, stream_t s);
void  graph_destroy(graph_t g);

// ---- timing of the dominant kernel (fp64 MFMA GEMM) with device events --------
void   prof_enable(bool on);
void   prof_reset();
// number of GEMM launches recorded, summed milliseconds, summed executed flops
// kernel_class 0: every fp64 GEMM call; 1: only the calls that ran on the LDS-DMA 128x128 kernel
void   prof_query(int kernel_class, long* calls, long* kernel_launches, double* ms, double* flops);

// ---- fp64 GEMM  C = alpha * A * B + beta * C ---------------------------------
// A(m,k) = A[z] + m*a_sm + k*a_sk   with exactly one of a_sm / a_sk equal to 1
// B(k,n) = B[z] + k*b_sk + n*b_sn   with exactly one of b_sk / b_sn equal to 1
// C(m,n) = C[z] + m*ldc + n
// z = (z1, z2) two-level batch, per-operand strides may be 0.
struct Gemm {
    int64_t M, N, K;
    double alpha, beta;
    const double* A; int64_t a_sm, a_sk;
    const double* B; int64_t b_sk, b_sn;
    double* C; int64_t ldc;
    const double* Cin = nullptr;      // optional: read the beta term from here (same layout/strides as C)
    int64_t nb1, nb2;                 // batch extents (>=1)
    int64_t a_b1, a_b2, b_b1, b_b2, c_b1, c_b2;
    double* splitk_ws;                // workspace for split-K partials (may be null)
    int64_t splitk_ws_doubles;
};
void gemm(const Gemm& g, stream_t s);
// Second stage of the T1-dressed Fock matrix (ccsd.py:226-288) from the eight T1.V intermediates
//   W = [ G1 (v,v) | G2 (v,v) | J1 (o,v) | J2 (o,v) | L1 (o,o) | L2 (o,o) | K1 (o,v) | K2 (v,o) ]
// (cc.cpp, dress_fock_partial; G = 2 G1 - G2, Mm = 2 J1 - J2, L = 2 L1 - L2), f and fd [n,n], t1 [v,o], ft: o*o scratch:
//   f~_ov = f_ov + 2 K1 - J2;   ft = (f_ov + Mm) t + L;   f~_oo = f_oo + ft;   f~_vv = f_vv + G - t (f_ov + Mm)
//   f~_vo = f_vo - t (f_oo + ft) + (f_vv + G) t + 2 K1^T - K2
// G1[a][c] = sum_{j in [j0,j1), b} t1[b,j] V[j,a,b,c],  G2[a][c] = sum_{j,b} t1[b,j] V[j,a,c,b]  for a block V [o,na,v,v] as
// stored (na = v: the o v^3 block; na = o: the o^2 v^2 block), in one pass over it; G1, G2 are [na,v];
// ws: fock_g12_ws_doubles(nv, na, j1 - j0) doubles.  fock_g12_ok: nv <= 1024 (odd nv <= 512).
bool fock_g12_ok(int nv);
int64_t fock_g12_ws_doubles(int nv, int na, int nj);
void fock_g12(const double* V, const double* t1, double* G1, double* G2, int no, int nv, int na, int j0, int j1, double* ws,
              stream_t s);
void fock_finish(const double* f, const double* t1, const double* W, double* fd, double* ft, int no, int nv, stream_t s);
// Matrix-vector-shaped gemm() calls (M == 1 or N == 1, beta == 0) issued between begin and end are independent of each
// other by the caller's promise and may be launched together at end (or earlier: any permute / other GEMM flushes them).
void gemv_batch_begin();
void gemv_batch_end();

// Small products (64 x 64 tiles; see kernels.hip, dgemm_group_kernel) issued between begin and end are INDEPENDENT of each
// other by the caller's promise — none reads or accumulates into what another writes — and are launched together, up to 16
// per launch: one launch per dependency level of a term sequence.  Anything else enqueued on the stream in between (a
// permutation, a copy, a big product, a synchronisation) launches the queue first, so the order of effects is that of
// immediate execution; gemm_group_sync() does the same explicitly (a level boundary).  A no-op region in the host simulator.
void gemm_group_begin(stream_t s);
void gemm_group_end();
void gemm_group_sync();
void gemm_group_stats(long* launches, long* products);    // grouped launches / queued products since the last begin

// ---- phase launches (kernels.hip, phase_kernel).  Small operations of this interface — products, permutations,
// the element-wise and reduction helpers below — are not launched one by one: they are recorded with the address ranges they
// read and write, and launched level by level of their dependency graph, one grid per level carrying the blocks of all its
// tasks.  Anything that is not recorded (a big product, a copy, a synchronisation, a graph boundary) launches the recorded
// tasks first, so the order of effects is that of immediate execution.  phase_sync(): launch what is recorded now (the end of
// every entry point of the C interface; before a collective of the host program).  phase_enable: 1 on, 0 off, -1 as the
// environment says (PYMES_PHASE=0 off, =serial one task per level; default on).  No-ops in the host simulator.
void phase_sync();
bool phase_pending();              // tasks are recorded and not yet launched (this thread)
long phase_generation();           // number of non-empty phases launched so far (this thread)
void phase_enable(int mode);
// phase_hold(true): the end of a C-interface call (phase_call_end) leaves its tasks recorded, so that the tasks of several
// calls share levels — the host's two amplitude updates and the DIIS overlaps of an iteration; phase_hold(false) launches
// them.  Synchronisations, copies and unrecorded launches launch the recorded tasks first, held or not.
void phase_hold(bool on);
void phase_call_end();
void phase_stats(long* tasks, long* launches, long* levels, long* flushes);     // counted per thread since its start

// ---- strided copy / permutation:  out = alpha * in + beta * out ----------------
// rank <= 6, both tensors described by the same extents and their own strides.
struct Permute {
    int rank;
    int64_t dim[6];
    int64_t s_in[6], s_out[6];
    double alpha, beta;
    const double* in; double* out;
};
void permute(const Permute& p, stream_t s);

// ---- element-wise / reduction helpers (HBM-bound) ------------------------------
// t[a,b,i,j] = w[a,b,i,j] / (eo[i]+eo[j]-ev[a]-ev[b]+shift)          (mp2.py:16-18)
void mp2_amplitudes(double* t, const double* w, const double* eo, const double* ev, double shift,
                    int no, int nv, stream_t s);
// dt = r * (1/(D+shift)) ; t += delta*dt   rank 4 (abij) or rank 2 (ai)   (ccsd.py:176-179)
void cc_update(double* t, double* dt, const double* r, const double* eo, const double* ev, double shift,
               double delta, int no, int nv, int rank, stream_t s);
// out[p] = sum_{i < n[p]} x_p[i]*y_p[i], p < npairs (<= 16); deterministic two-stage reduction; result on host
// (synchronises the stream)
void dots(int npairs, const double* const* x, const double* const* y, const int64_t* n, double* out_host,
          stream_t s);
// ccsd.py:458-466 and the norms of ccsd.py:196-197 in one pass over T2 (tau = t2 + t1 t1 formed on the fly):
//   out[0] = sum f_ov[i,a] t1[a,i]   (f is the [n,n] Fock matrix on the device)
//   out[1] = sum tau Edir,  out[2] = sum tau Eex   (Edir = V_ijab as [a,b,i,j], Eex = V_ijba as [a,b,i,j])
//   out[3] = sum t2^2,  out[4] = sum dt2^2  (dt2 may be null: 0),  out[5] = sum t1^2
// t1 / f may be null (CCD: out[0] = out[5] = 0, tau = t2).  Result on host (synchronises the stream).
void energy_norms(const double* f, const double* t1, const double* t2, const double* Edir, const double* Eex,
                  const double* dt2, int no, int nv, double out_host[6], stream_t s);
// the same without draining the stream: the kernels and a copy into a pinned slot are enqueued and the slot id is returned;
// readback_wait(slot, out, 6) later blocks until THAT copy has landed, while the stream runs what was enqueued behind it
int energy_norms_start(const double* f, const double* t1, const double* t2, const double* Edir, const double* Eex,
                       const double* dt2, int no, int nv, stream_t s);
// generic form: n <= 128 doubles of device memory (a ring of 16 slots per device: wait for a slot before 16 later starts)
int readback_start(const double* dev_ptr, int n, stream_t s);
void readback_wait(int slot, double* out_host, int n);
// the same sums over the virtual pairs P(a,b) in [r0,r1) only, amplitudes given as compact tiles tc / dtc
// [pair][2][o*o] (tile (a,b), tile (b,a); pairs_pack); the T1 sums out[0], out[5] only when with_t1 (one rank of many)
void energy_norms_pairs(const double* f, const double* t1, const double* tc, const double* Edir, const double* Eex,
                        const double* dtc, int no, int nv, int64_t r0, int64_t r1, bool with_t1, double out_host[6],
                        stream_t s);
// ... the six sums left in device memory (out_dev[6]; nothing is copied, nothing waits): one rank's share, all-reduced by
// the host program's collective before anybody reads it
void energy_norms_pairs_dev(const double* f, const double* t1, const double* tc, const double* Edir, const double* Eex,
                            const double* dtc, int no, int nv, int64_t r0, int64_t r1, bool with_t1, double* out_dev,
                            stream_t s);
// max |A[p,q,r,s] - B[q,p,s,r]| and max |A| for A [d0,d1,d2,d3], B [d1,d0,d3,d2] (electron-exchange partner; B may be
// A itself when d0 == d1 and d2 == d3).  Result on host (synchronises the stream).
void exchange_asymmetry(const double* A, const double* B, const int64_t d[4], double out_host[2], stream_t s);
// out-of-place form: dt = r * (1/(D+shift)); t_out = t_in + delta*dt  (t_out may alias t_in)
void cc_update_to(double* t_out, double* dt, const double* t_in, const double* r, const double* eo, const double* ev,
                  double shift, double delta, int no, int nv, int rank, stream_t s);
// out = sum_k c[k] * x_k   (k < nx <= 8)
void lincomb(double* out, int nx, const double* const* x, const double* c, int64_t n, stream_t s);
// ---- tall-skinny subspace algebra of the Davidson / FEAST drivers (eom_ccsd.py:91-147, :512-541): every vector is read ONCE
// per call, whatever the number of inner products / combinations it enters ------------------------------------------------
// out_host[i * n + j] = <x_i, y_j> for i < m, j < n over vectors of `len` doubles (m, n <= 64); deterministic two-stage
// reduction; result on the host (synchronises the stream)
void gram(int m, int n, const double* const* x, const double* const* y, int64_t len, double* out_host, stream_t s);
// y_j = sum_{i < m} c[i * n + j] x_i + beta[j] y_j  for j < n (c, beta on the host; beta null = 0; m, n <= 64).  A y_j may
// alias an x_i only when m <= 16 and n <= 4 (one launch: every element is read before it is written)
void lincomb_multi(int m, int n, const double* const* x, const double* c, const double* beta, double* const* y, int64_t len,
                   stream_t s);
// One DIIS step without a host round trip (diis_small.h): the ntypes * m overlaps <x_p, y_p> are reduced on the device, the
// (m+1) x (m+1) system is updated and solved by one thread, the coefficients land in state[82..]; nothing synchronises.
void diis_step(double* state, int npairs, const double* const* x, const double* const* y, const int64_t* n, int ntypes, int m,
               int was_full, stream_t s);
// out = sum_k coeff[k] x[k] with the coefficients read from device memory (the output of diis_step)
void lincomb_dev(double* out, int nx, const double* const* x, const double* coeff_dev, int64_t n, stream_t s);
// (yr + i yi)[e] = (mr + i mi)[e] * (xr + i xi)[e]: a complex diagonal applied to a complex vector held as two real arrays
// (y may alias x)
void cmul(const double* mr, const double* mi, const double* xr, const double* xi, double* yr, double* yi, int64_t n, stream_t s);
// u [v,v,o,o] without the exchange symmetry split for two pair-packed ladders: us = (u + P u) / 2, (P u)_abij = u_baji;
// w_abij = sgn(i - j) (u - P u)_abij / 2 (exchange-symmetric, zero for i == j); dg[a,b,i] = (u - P u)_abii / 2 [v,v,o]
void exchange_split(const double* u, double* us, double* w, double* dg, int no, int nv, stream_t s);
void sgn_ij_add(double* D, const double* R, int no, int nv, stream_t s);          // D_abij += sgn(i - j) R_abij
// (mr + i mi)[e] = 1 / ((zr + i zi) - (hr + i hi) d[e] + shift): that preconditioner from the device-resident diagonal
void cshift_inv(const double* d, double zr, double zi, double hr, double hi, double shift, double* mr, double* mi, int64_t n,
                stream_t s);
// EOM-CCSD diagonals (eom_ccsd.py:169-198 singles, :200-266 doubles): V = V_ijab [o,o,v,v], T [v,v,o,o], dai[a,i] = f_aa - f_ii,
// the four diagonal slices iaai[a,i] = V_iabj[i,a,a,i], iaia[a,i] = V_iajb[i,a,i,a], ijij[i,j] = V_klij[i,j,i,j],
// abab[a,b] = V_abcd[a,b,a,b] (all device); d1 [v,o], d2 [v,v,o,o]; ws: eom_diag_ws_doubles(no, nv) doubles
int64_t eom_diag_ws_doubles(int no, int nv);
void eom_diagonals(const double* V, const double* T, const double* dai, const double* iaai, const double* iaia, const double* ijij,
                   const double* abab, double* d1, double* d2, int no, int nv, double* ws, stream_t s);
// tau[a,b,i,j] = t2[a,b,i,j] + t1[a,i]*t1[b,j]                        (ccsd.py:462)
void tau_build(double* tau, const double* t2, const double* t1, int no, int nv, stream_t s);

// ---- symmetry-packed ladders (ccd.py:186-187 with V_pqrs = V_qpsr, T_cdij = T_dcji) -----------------------
// pair indices: P(x,y) = x(x+1)/2 + y for x >= y ("plus"), Q(x,y) = x(x-1)/2 + y for x > y ("minus").
// ladder_pack_V: V is [nr,nr,nc,nc].  For the pair rows r = P(a,b) in [rp0, rp1):
//   Vp[r - rp0][P(c,d)] = V[a,b,c,d] + V[a,b,d,c]
//   Vm[r - rp0][Q(c,d)] = V[a,b,c,d] - V[a,b,d,c]   (zero row when a == b)
// nr == 0: V is [rows,nc,nc] and the rows [rp0, rp1) are taken as they are (no zero rows in Vm).
void ladder_pack_V(const double* V, double* Vp, double* Vm, int nr, int nc, int64_t rp0, int64_t rp1, stream_t s,
                   int64_t ldvp = 0, int64_t ldvm = 0);      // row pitches of Vp / Vm (0: nc(nc+1)/2, nc(nc-1)/2)
// ladder_dress: T1 dressing of the bra of pair-packed rows (ccsd.py:414-419 as far as the packed ladder reads it).
// V, W: [row1 - row0][ld], rows r = P(a,b) in [row0,row1); Pk: [nv*no][ld], rows (x,k) = x*no + k of V_kxcd packed like V:
//   W[r] = V[r] - sum_k t1[a,k] Pk[(b,k)] + sgn sum_k t1[b,k] Pk[(a,k)]     (sgn = -1: "plus" half, +1: "minus" half, whose
//   rows a == b stay zero).  ld must be a multiple of 16 doubles (all ld columns are processed); ladder_dress_ok: no <= 80 and
//   the rows of 16 a of the packed block within 2 GB (nv up to ~320).
bool ladder_dress_ok(int no, int nv);
// ws: ladder_dress_ws_doubles(no, nv) doubles of scratch (the t1 fragments in matrix-core operand order).
int64_t ladder_dress_ws_doubles(int no, int nv);
void ladder_dress(const double* V, const double* Pk, const double* t1, double* W, int no, int nv, int64_t ld, int64_t row0,
                  int64_t row1, double sgn, double* ws, stream_t s);
// ladder_pack_T: X is [nr,nr,nc,nc]; pair (c,d) over nr, pair (i,j) over nc.
//   Sp[P(c,d)][P(i,j)] = fr fc (X[c,d,i,j] + X[d,c,i,j]) / 2,  fr = 1/2 on c == d if PACK_ROW_HALF, fc = 1/2 on
//   i == j if PACK_COL_HALF (a pair that is summed over carries the half on its diagonal);
//   Am[row][col] = (X[c,d,i,j] - X[d,c,i,j]) / 2 with row = Q(c,d) (c > d) or P(c,d) if PACK_AM_PROWS and
//   col = Q(i,j) (i > j) or P(i,j) if PACK_AM_PCOLS; entries on a diagonal pair are zero.
// If t1 ([nr,nc]) is given, t1[c,i] t1[d,j] is added to X[c,d,i,j] on the fly (tau of ccsd.py:462); X may then be
// null.  ldp / ldm are the row pitches of Sp / Am (0 = dense).  [rp0,rp1) (rp1 < 0: all): only these rows P(c,d) are
// written, at their usual place (needs PACK_AM_PROWS: Am rows are then the same pairs) — a rank that reads only its rows.
enum { PACK_ROW_HALF = 1, PACK_AM_PROWS = 2, PACK_COL_HALF = 4, PACK_AM_PCOLS = 8 };
void ladder_pack_T(const double* X, const double* t1, double* Sp, double* Am, int nc, int nr, int flags, int64_t ldp,
                   int64_t ldm, stream_t s, int64_t rp0 = 0, int64_t rp1 = -1);
// L[P(a,b)] = [ LS row (o(o+1)/2) | LA row (o(o-1)/2) ], row length o*o:
// R[a,b,i,j] = beta R + LS[P(ab)][P(ij)] + sgn(a-b) sgn(i-j) LA[P(ab)][Q(ij)]
void ladder_unpack(const double* L, double* R, double beta, int no, int nv, stream_t s);
// The pair layouts of T[a,b,i,j] in one pass: Td[(a,i),(b,j)] = Tx[(a,j),(b,i)] = T_abij, Ttd = 2 Td - (T_baij in
// the Td layout) — in general Ttd = ca Td + cb (T_baij in the Td layout).  fused_pair_kernels_ok(no): the o x o staging tile
// fits the LDS (else use permutes).
bool fused_pair_kernels_ok(int no);
void t2_layouts(const double* T, double* Td, double* Tx, double* Ttd, int no, int nv, stream_t s, double ca = 2.0,
                double cb = -1.0);
// Assembly of the symmetry-reduced residual (ccd.py:249-252) in one pass:
//   R_abij = V_abij + unpack(L)_abij + N_abij + N_baji + D[(a,i),(b,j)] + D[(b,j),(a,i)] + X[(a,j),(b,i)] + X[(b,i),(a,j)]
// L (pair-packed ladder rows, may be null), V (may be null: 0), N [v,v,o,o], D and X [ov,ov]
// xd != 0: X enters in the direct placement too, + xd (X[(a,i),(b,j)] + X[(b,j),(a,i)]) — the half of the C-term that the
// D-term carries (Ex_d = D + Ex_x / 2, cc.cpp residual_slab) read from X itself, so that the products D and X are independent
void residual_assemble(const double* V, const double* L, const double* N, const double* D, const double* X, double* R,
                       int no, int nv, stream_t s, double xd = 0.0);
// ---- pair-sharded tail: compact storage Xc[P - r0][2][o*o] of the tiles X[a,b,:,:], X[b,a,:,:] (zeros for a == b)
// of the virtual pairs P(a,b) in [r0,r1) ------------------------------------------------------------------------
void pairs_pack(const double* full, double* Xc, int no, int nv, int64_t r0, int64_t r1, stream_t s);
void pairs_unpack(const double* Xc, double* full, int no, int nv, int64_t r0, int64_t r1, stream_t s);   // Xc of these pairs
void cc_update_pairs(double* tc, double* dtc, const double* rc, const double* eo, const double* ev, double shift,
                     double delta, int no, int nv, int64_t r0, int64_t r1, stream_t s);
// residual_assemble for the pairs [r0,r1), compact output; Np[a - a0][b][o*o] (b < nbp) = N_ab + N_ba^T already combined
void residual_assemble_pairs(const double* V, const double* L, const double* Np, const double* D, const double* X,
                             double* Rc, int no, int nv, int64_t r0, int64_t r1, int a0, int nbp, stream_t s, double xd = 0.0);
// plain rows of the same [ S | A ] layout: out[r][i][j] = Q[r][P(i,j)] + sgn(i-j) Q[r][o(o+1)/2 + Q(i,j)]
void rows_unpack(const double* Q, double* out, int64_t rows, int no, stream_t s);
// Right-hand operands of the ring builds in ONE pass over the two (dressed) blocks V_iabj [k][b][c][j] and V_iajb [k][b][j][c]
// (ccd.py:190-191, :199-204 through the C / D form of cc.cpp): pair matrices on (c,k) = c*no + k, (b,j) = b*no + j,
//   N1[(c,k)][(b,j)] = -V_iajb[k,b,j,c],   M[(c,k)][(b,j)] = a1 V_iabj[k,b,c,j] - a2 V_iajb[k,b,j,c]
// — two permutations and an axpby before: 5 reads / 3 writes of (ov)^2 doubles instead of 2 / 2.  Blocks dense.
void ring_operands(const double* Viabj, const double* Viajb, double* M, double* N1, double a1, double a2, int no, int nv,
                   stream_t s);
// partial traces of a pair matrix M[(c,k)][(b,j)] (row pitch ld; (c,k) = c*no + k):
//   out_vv[a][c] = beta out_vv[a][c] + alpha sum_k M[(c,k)][(a,k)],   out_oo[k][i] = beta out_oo[k][i] + alpha sum_c M[(c,k)][(c,i)]
// With M2 (same shape and pitch): + alpha2 x the same traces of M2, in the same pass.
void pair_traces(const double* M, int64_t ld, double alpha, double beta, double* out_vv, double* out_oo, int no, int nv,
                 stream_t s, const double* M2 = nullptr, double alpha2 = 0.0);

// ---- Hartree-Fock matrix from the packed blocks (pymes/mean_field/hf.py:14-18); dir[tp*2+tq] = block (tp,o,tq,o),
// exc[tp*2+tq] = block (tp,o,o,tq), tp/tq = 1 for a virtual index; h and f are [n,n] on the device
void hf_fock(const double* const dir[4], const double* const exc[4], const double* h_dev, double* f_dev, int no, int nv,
             stream_t s);

// ---- FCIDUMP ingestion (pymes/util/fcidump.py:140-149): the two-electron lines (0-based p,q,r,s after the reference's
// renaming, file order) are written with their symmetry images into the zero-initialised dense V[n]^4 on the device.
// Returns the number of lines whose images do not all hold the line's value afterwards (0 for consistent files).
int64_t fcidump_fill(double* V, const double* val_host, const int32_t* pqrs_host, int64_t count, int n, bool is_tc,
                     stream_t s);

// ---- explicit 3-body (transcorrelated) operator: pymes/util/tcdump.py:52-56, pymes/integral/contraction.py:17-95 ----
// dst[idx[t]] = val[t] for t < n (host index/value lists, unique targets); L below is dense [nb]^6, (or|ps|qt) order
void scatter(double* dst, const int64_t* idx_host, const double* val_host, int64_t n, stream_t s);
void tc_single_contraction(const double* L, double* D, int nb, int no, stream_t s);   // D[p,r,q,s]
void tc_double_contraction(const double* L, double* S, int nb, int no, stream_t s);   // S[p,q]
double tc_triple_contraction(const double* L, int nb, int no, stream_t s);

// ---- 3D uniform electron gas two-body integrals (pymes/model/ueg.py:265-516) ------------------
// V[p,q,r,s] (dense [n_p]^4, zero where momentum is not conserved) for the plane-wave basis
// k_int[n_p][3] (sorted by kinetic energy) and its lookup table index_map[(2 imax + 1)^3].
struct UegParams {
    int n_p, n_ele, imax;
    int mode;               // 0 Coulomb, 1 TC "only_2b", 2 TC "effect_2b" (unsymmetrised), 3 RPA
    double L, Omega;
    double k_cutoff, gamma; // `trunc` correlator: u(k^2) = -4 pi gamma / k^4 for k > k_cutoff * 2 pi / L
    int lattice_cutoff;     // k' lattice of sumNablaUSquare (ueg.py:581), 30 in the reference
    // The reference's other correlators (ueg.py:740-935) evaluated in the kernels like trunc, from the same float k^2 the
    // reference forms: corr_kind 0 trunc (k_cutoff, gamma above), 1 gaskell {mu, cut}, 2 gaskell_modified {cut},
    // 3 coulomb {-4 pi gamma}, 4 yukawa {gamma, floor}, 5 stg {gamma^2, floor, -4 pi / gamma}, 6 smooth {kc, kc gamma,
    // (kc gamma)^2}.  The reference calls a correlator either with a float (ueg.py:409 `d_k_vec.dot(d_k_vec)`) or with an
    // ndarray (everything else); the two forms of gaskell / gaskell_modified differ AT the cut-off and both are kept.
    int corr_kind = 0;
    double corr_p[4] = {0.0, 0.0, 0.0, 0.0};
    // A correlator of the caller's own: every argument of u is |2 pi n / L|^2 for an integer vector n, so u is handed
    // over as HOST tables over m = |n|^2 < tab_len (tab_scalar / tab_array: the two call forms).  A jump of u exactly on
    // a lattice shell is then resolved per shell, not per rounding of the individual k^2 as in the reference.
    const double* tab_scalar = nullptr;
    const double* tab_array = nullptr;
    int tab_len = 0;
};
void ueg_two_body(const UegParams& prm, const int* k_int_dev, const int* index_map_dev, double* V_dev, stream_t s);

}  // namespace dev
