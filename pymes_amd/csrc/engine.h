// Host engine of pymes_amd: device arena, strided tensor views, the einsum-style
// binary contraction planner (TTGT: transpose only when needed, then the fp64 MFMA
// GEMM), and the coupled-cluster term sequences built on top of it.
//
// The seam this replaces in the reference is the module-level `einsum` callable
// (pymes/solver/ccsd.py:11, mp2.py:5; historically ctf.einsum, pymes/__init__.py:3)
// plus the solver methods that call it.
#pragma once
#include <cstdint>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "device_api.h"

namespace pymes {

struct TView {
    double* p = nullptr;
    int rank = 0;
    int64_t dim[6] = {1, 1, 1, 1, 1, 1};
    int64_t st[6] = {0, 0, 0, 0, 0, 0};
    int64_t size() const {
        int64_t n = 1;
        for (int i = 0; i < rank; ++i) n *= dim[i];
        return n;
    }
};

TView make_view(double* p, std::initializer_list<int64_t> dims);
TView make_view(const double* p, std::initializer_list<int64_t> dims);
TView make_view(double* p, int rank, const int64_t* dims, const int64_t* strides /* may be null */);
// slice [lo,hi) along one axis
TView slice(const TView& t, int axis, int64_t lo, int64_t hi);

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// Bump allocator over one device slab; all work is stream-ordered on one stream, so a
// region can be reused as soon as the kernels that touch it have been enqueued.
class Arena {
  public:
    void init(size_t bytes);
    void release();
    double* alloc(int64_t doubles);
    size_t mark() const { return top_; }
    // While tasks of an open phase are recorded but not launched (device_api.h, phase launches) a released region is NOT
    // handed out again: a later temporary at the same address would be ordered behind every recorded reader of the old one
    // (a write-after-read hazard that exists only because of the reuse) and the phase would lose the concurrency it is there
    // for.  The release is remembered and takes effect once those tasks have been launched — or at once when the arena runs
    // out of room (the phase is launched first).
    void reset(size_t m);
    size_t capacity() const { return cap_; }
    size_t high_water() const { return high_; }

  private:
    char* base_ = nullptr;
    size_t cap_ = 0, top_ = 0, high_ = 0;
    // a release that waits for the open phase: valid while nothing has been allocated since (stack discipline: the
    // allocations made after a release are the highest live ones; with none, everything above `want_` is dead)
    bool deferred_ = false;
    size_t want_ = 0;
    long gen_ = 0;
};

struct ArenaScope {
    Arena& a;
    size_t m;
    explicit ArenaScope(Arena& ar) : a(ar), m(ar.mark()) {}
    ~ArenaScope() { a.reset(m); }
};

// Products issued inside the scope are independent of each other (the caller's promise) and may share launches
// (dev::gemm_group_begin / _end); level() closes a dependency level.  Only contract / permute / copy / zero calls belong inside.
struct GemmGroupScope {
    explicit GemmGroupScope(dev::stream_t s) { dev::gemm_group_begin(s); }
    ~GemmGroupScope() {
        try { dev::gemm_group_end(); } catch (...) {}
    }
    void level() { dev::gemm_group_sync(); }
    void close() { dev::gemm_group_end(); }       // (errors of the launches surface here)
};

struct ContractStats {
    long gemm_calls = 0, permute_calls = 0;
    double gemm_flops = 0.0;       // executed 2*M*N*K*batch
    double permute_bytes = 0.0;    // bytes moved by explicit copies (read + write)
};

// block pattern id: bit (3-pos) set when the index at `pos` is virtual
int pattern_of_name(const char* name);          // "abcd" -> 15, "klij" -> 0; throws on bad name
std::string canonical_name(int pattern);         // the partition.py spelling for that pattern

class Engine {
  public:
    Engine(int device, int no, int nv, size_t workspace_bytes);
    ~Engine();

    int device;
    int no, nv, n;
    // all work of the engine is ordered on this stream: a stream of the engine's own unless the caller binds another
    // one (pymes_ctx_set_stream, e.g. torch's current stream so that RCCL collectives are ordered with the kernels)
    dev::stream_t stream = nullptr;
    Arena arena;
    ContractStats stats;

    void set_stream(dev::stream_t s) { stream = s; }
    // caller-visible device buffers (pymes_malloc): tracked so that closing the context releases whatever the host
    // side still holds
    void* user_malloc(size_t bytes);
    void user_free(void* p);
    // Pooled device scratch for multi-call objects (the EOM sigma handle, eom.cpp): buffers are handed out by exact size and
    // kept when returned — a second solve on the same context finds the buffers of the first (stream-ordered reuse: all
    // work is on `stream`).  scratch_trim() frees what is not in use (also tried by scratch_get before it gives up).
    double* scratch_get(int64_t doubles);
    void scratch_put(double* p);
    void scratch_trim();
    int64_t scratch_free_bytes() const;
    // launch graphs recorded on this engine's stream (see device_api.h)
    void graph_begin();
    dev::graph_t graph_end();
    void graph_abort();
    void graph_launch(dev::graph_t g);
    void graph_destroy(dev::graph_t g);
    bool capturing() const { return capturing_; }
    // Counts the dressings of the context's dressed blocks (every Engine::dress_V, wherever it is called from — also inside
    // ccsd_residuals / ccsd_sharded_residuals — and every replay of a recorded graph that contains one): holders of an
    // earlier dressing (the host's DressedDeviceIntegrals) compare (pymes_dress_generation).
    uint64_t dress_generation() const { return dress_generation_; }
    // max |V_pqrs - V_qpsr| over all blocks that are set (infinity if a block's exchange partner is missing) and max |V|
    void exchange_asymmetry_V(double out[2]);

    // ---- generic tensor ops ---------------------------------------------------------
    // C[sc] = alpha * sum A[sa] * B[sb] + beta * C[sc];  `batch` lists free labels that are
    // looped as GEMM batches instead of being merged into M/N.
    // `Cin` (optional, same shape and strides as C) supplies the beta term: C = alpha*A*B + beta*Cin, which
    // fuses a copy into the product.
    void contract(double alpha, const TView& A, const char* sa, const TView& B, const char* sb, double beta,
                  const TView& C, const char* sc, const char* batch = "", const TView* Cin = nullptr);
    // out[so] = alpha * in[si] + beta * out[so]   (labels are a permutation of each other)
    void permute(double alpha, const TView& in, const char* si, double beta, const TView& out, const char* so);
    // out = alpha * in + beta * out, same index order (shapes must match)
    void axpby(double alpha, const TView& in, double beta, const TView& out);
    void copy(const TView& in, const TView& out) { axpby(1.0, in, 0.0, out); }
    void zero(const TView& t);

    // ---- integrals ------------------------------------------------------------------
    void set_V_full(const double* V, bool on_device, const int64_t strides[4]);
    void set_V_block(const char* name, const double* data, bool on_device, const int64_t strides[4]);
    void set_V_from_factors(const double* B_host, int naux);
    TView block(int pattern, bool dressed = false);       // throws if the block is absent
    double* ensure_block(int pattern);                    // storage of an undressed block (allocated if absent); cached
                                                          // derived quantities are invalidated
    bool has_block(int pattern, bool dressed = false) const;
    void set_orbital_energies(const double* eo_host, const double* ev_host);

    // ---- CC path (cc.cpp) -----------------------------------------------------------
    void mp2(double shift, double* t2, double e_out[2]);                                      // mp2.py:9-22
    void hf_fock_matrix(const double* h_host, double* f_host);                                // hf.py:14-18
    void dress_fock(const double* f, const double* t1, double* fd);                           // ccsd.py:226-288
    // the same in two stages with a K-sharded first stage (one process per GPU), see cc.cpp
    int64_t dress_fock_ws_doubles() const;
    void dress_fock_partial(const double* t1, double* W, int rank, int world);
    void dress_fock_finish(const double* f, const double* t1, const double* W, double* fd);
    void xvv_partial(const double* f, const double* t2, double* Xvv, int rank, int world, unsigned flags);
    // ccsd.py:290-421; cut = {p0,p1,q0,q1} (optional): only these ranges of the first / second (virtual) index
    void dress_V(const double* t1, uint32_t mask, const int64_t* cut = nullptr);
    void singles_residual(const double* fd, const double* t1, const double* t2, double* r1);  // ccsd.py:423-438
    // the same as a partial sum over this rank's chunk of the occupied summation index (exchange-symmetric T2), see cc.cpp
    // reuse_layouts: the pair layouts that the preceding residual_slab built from this very t2 (content unchanged since) are
    // read in place of freshly permuted copies — the caller's promise (flag PYMES_REUSE_LAYOUTS)
    void singles_residual_partial(const double* fd, const double* t1, const double* t2, double* r1, int rank, int world,
                                  bool reuse_layouts = false);
    // ccd.py:164-254; flags: bit0 = DCD/DCSD, bit1 = use dressed blocks, bit2 = skip ladder,
    // bit3 = pair-packed ladder (T and V exchange-symmetric)
    void doubles_residual(const double* f, const double* t2, double* r2, unsigned flags);
    // R[a0:a1,:,:,:] = beta*R + V_abcd[a0:a1] . T    (ccd.py:187; the sharded term)
    void ladder(const double* t2, double* r2, int a0, int a1, bool dressed, double beta);
    // the same term at 1/4 of the flops, valid when V_abcd = V_badc and T_cdij = T_dcji:
    // rows [row0,row1) of the pair-packed result L[v(v+1)/2][o*o] (device_api.h), then R = beta R + unpack(L)
    // hole = 1 / 2 adds the hole ladder (ccd.py:175-186; CCSD / DCSD form of I_klij) to the same rows
    void ladder_sym(const double* t2, double* L, int64_t row0, int64_t row1, bool dressed, int hole = 0);
    // the particle ladder of k exchange-symmetric vectors x_z [v,v,o,o] at once: L_all[z] (k consecutive [v(v+1)/2][o*o]
    // arrays) = pair-packed V_abcd . x_z, one batched launch per half (S / A) over all k vectors
    void ladder_sym_multi(const double* const* xs, int k, double* L_all, bool dressed);
    void hole_ladder_packed_multi(const double* const* xs, const double* const* Is, const double* const* ys, int k, double* L_all);
    void hole_ladder_packed(const double* x, const double* I, double* L, int64_t row0, int64_t row1,
                            const double* y = nullptr);
    void ladder_sym_unpack(const double* L, double* r2, double beta);
    // symmetry-reduced residual in shardable form (cc.cpp): this rank's column slab of the ring products
    // (rows of ETd/ETx) and its rows of the packed ladder L; then the replicated remainder + assembly
    // t1 + QK given: V_abcd is never dressed; its T1 dressing is applied on the amplitude side (ladder_t1)
    void residual_slab(const double* f, const double* t2, double* ETd, double* ETx, double* L, int rank, int world,
                       unsigned flags, const double* t1 = nullptr, double* QK = nullptr, const double* P = nullptr);
    // K-sharded partial sums of the slab's small replicated intermediates (X'_ki, pair-packed V_klcd T_cdij), see cc.cpp
    int64_t slab_prepare_ws_doubles() const;
    void slab_prepare(const double* t2, double* P, int rank, int world, unsigned flags);
    void residual_finish(const double* f, const double* t2, const double* ETd, const double* ETx, const double* L,
                         double* r2, unsigned flags, const double* t1 = nullptr, const double* QK = nullptr);
    // Pair-sharded tail (one process per GPU): rank owns the virtual pairs P(a,b), a >= b, of its chunk of
    // v(v+1)/2 (the rows of L it computed itself) and produces R for exactly those pairs in the compact layout
    // Rc[P - r0][2][o*o] (tiles R[a,b,:,:], R[b,a,:,:]); L is read locally, ETd/ETx/QK must have been exchanged
    void residual_finish_pairs(const double* f, const double* t2, const double* ETd, const double* ETx, const double* L,
                               double* Rc, unsigned flags, const double* t1, const double* QK, int rank, int world,
                               const double* Xvv_in = nullptr);
    void pair_chunk(int rank, int world, int64_t& r0, int64_t& r1) const;
    void amplitude_side_abij(const double* t1, const double* QK, const TView& N, int64_t a0, int64_t a1, int64_t b1,
                             bool with_partner);
    // rows [row0,row1) of the pair-packed ladders and rows [q0,q1) of QK[(k,b)] = sum_cd V_kbcd tau_cdij, all
    // from UNDRESSED, statically packed integrals; tau = T + t1 t1
    void ladder_t1(const double* t1, const double* t2, double* L, int64_t row0, int64_t row1, double* QK, int64_t q0,
                   int64_t q1, bool dcd, const double* J = nullptr);
    void dress_abcd_rows(const double* t1, int a0, int a1, bool lower_only);
    // ---- whole steps (ccsd_residuals / ccsd_iterate) ----------------------------------------------------------
    // ccsd.py:161-171 in one call: dressed Fock matrix, the dressed blocks the loop needs, R1 [v,o] and R2 [v,v,o,o] from
    // (f, t1, t2) — the symmetry-reduced form (V_pqrs = V_qpsr and T_abij = T_baji: the caller's promise, tested by
    // exchange_asymmetry_V / dev::exchange_asymmetry), one rank.  flags: PYMES_DCD; bit `kT1Zero`: the caller knows that
    // t1 == 0 exactly (the MP2 start, a momentum-conserving system): exp(-T1) H exp(T1) = H, residuals from the undressed f, V.
    // Only enqueues kernels on buffers the engine holds (replayable as a launch graph).
    static constexpr unsigned kT1Zero = 1u << 20;
    void ccsd_residuals(const double* f, const double* t1, const double* t2, unsigned flags, double* r1, double* r2);
    // ... and one fixed-point pass without a mixer (ccsd.py:159-197 with is_diis = False): residuals, dT = R / (D + shift),
    // T += delta dT in place, then the energies and norms of the updated amplitudes: out = {one-body, direct, exchange,
    // |t2|^2, |dt2|^2, |t1|^2}.  A caller with a mixer calls ccsd_residuals, pymes_cc_update_to, pymes_diis_mix,
    // pymes_energy_norms instead (what pymes_amd/solver/ccsd.py does).
    void ccsd_iterate(const double* f, double* t1, double* t2, unsigned flags, double shift, double delta, double* dt1,
                      double* dt2, double out[6]);
    void release_residual_buffers();
    void cc_update(double* t, double* dt, const double* r, double shift, double delta, int rank);  // ccsd.py:176-179
    void cc_update_to(double* t_out, double* dt, const double* t_in, const double* r, double shift, double delta, int rank);
    void ccsd_energy(const double* f, const double* t1, const double* t2, double out[3]);     // ccsd.py:458-466
    void ccd_energy(const double* t2, double out[2]);                                         // ccd.py:256-262
    // energies (ccsd.py:458-466 / ccd.py:256-262 when f, t1 are null) and the squared norms of t2 and dt2
    // (ccsd.py:196-197) in one pass: out = {one-body, direct, exchange, |t2|^2, |dt2|^2, |t1|^2}
    void energy_norms(const double* f, const double* t1, const double* t2, const double* dt2, double out[6]);
    // the same in two halves: enqueue (returns a read-back slot) / wait for that read-back only — the stream goes on with what
    // was enqueued in between (the next iteration's residual kernels)
    int energy_norms_start(const double* f, const double* t1, const double* t2, const double* dt2);
    void energy_norms_wait(int slot, double out[6]);
    void energy_norms_pairs(const double* f, const double* t1, const double* tc, const double* dtc, int rank, int world,
                            double out[6]);
    // ---- one process per GPU: the loop body with the collectives of the HOST PROGRAM (include/pymes_amd.h, pymes_collectives:
    // RCCL calls on a communicator of the host's own; this repository's Python host wraps torch.distributed).  Every
    // collective is asynchronous: `*_start` orders it behind the work already enqueued on the engine's stream and returns a
    // ticket, `wait` orders the stream behind its completion — the host never blocks, the exchange of the ring products flies
    // while the ladders are computed.  Buffers that take part in a collective are the caller's (ShardBuffers, sizes in the
    // header): a host that registers memory with its communicator does so once.
    struct Collectives {
        void* user = nullptr;
        int rank = 0, world = 1;
        int (*allreduce_start)(void* user, double* buf, int64_t n, void* stream, int64_t* ticket) = nullptr;
        int (*allgather_start)(void* user, double* buf, int64_t chunk, void* stream, int64_t* ticket) = nullptr;
        int (*wait)(void* user, int64_t ticket, void* stream) = nullptr;
        void (*mark)(void* user, const char* phase) = nullptr;      // optional: phase boundaries (profiling)
    };
    struct ShardBuffers {
        double *ETd, *ETx, *L, *QK, *Tall, *W, *Xvv, *P, *R1, *S;
    };
    void set_collectives(const Collectives* c);          // nullptr: none (single rank)
    bool has_collectives() const { return coll_set_; }
    // Optional extension of the table: an all-to-all with per-peer counts (doubles), the pieces of `send` / `recv` contiguous
    // in rank order.  With it and the two staging buffers (owner_tile_sizes) the rows of the ring products travel as the
    // OWNER TILES of the pair-sharded tail — rank q reads of ETd / ETx only the tiles [(a,.),(b,.)] and [(b,.),(a,.)] of its
    // pairs P(a,b) — instead of two all-gathers of the whole matrices (flag kOwnerTiles of the sharded residual steps).
    typedef int (*alltoallv_fn)(void* user, const double* send, const int64_t* send_counts, double* recv,
                                const int64_t* recv_counts, void* stream, int64_t* ticket);
    void set_alltoallv(alltoallv_fn fn) { alltoallv_ = fn; }
    void set_owner_tile_buffers(double* send, double* recv) { xs_ = send; xr_ = recv; }
    void owner_tile_sizes(int rank, int world, int64_t* send_doubles, int64_t* recv_doubles) const;
    static constexpr unsigned kOwnerTiles = 1u << 21;
    // ccsd.py:161-171 for this rank: dressed Fock + the dressed blocks of its slab, ring products (rows exchanged), ladders,
    // Q_kb, X_ac, R1 (all-reduced, left in b.R1), R2 of its virtual pairs as compact tiles rc [max(pairs,1)][2][o*o]; t2
    // [v,v,o,o] is the replicated array (completed from the exchange the previous ccsd_sharded_finish left in flight before
    // it is read).  Writes only b.* , fd and rc: a caller may enqueue it ahead of reading the previous energy.  flags: PYMES_DCD
    void ccsd_sharded_residuals(const double* f, double* fd, const double* t1, double* t2, const ShardBuffers& b,
                                unsigned flags, double* rc);
    // ccd.py:100-121 for this rank (no T1: nothing is dressed, no singles residual): ring products (rows exchanged while the
    // ladders run), ladders, the doubles residual of the rank's pairs as compact tiles rc.  Of b only ETd, ETx, L, Tall, S
    // are used.  ccsd_sharded_finish / _energy / _await with f = t1 = nullptr complete the pass.
    void ccd_sharded_residuals(const double* f, double* t2, const ShardBuffers& b, unsigned flags, double* rc);
    // ccsd.py:189-197 + the hand-over of the new amplitudes: partial energies / norms of (t1, tc, dtc) all-reduced on the
    // device and copied to the host on the side (returns the read-back slot for ccsd_sharded_energy), tc into the exchange
    // buffer, all-gather of the new T2 started (awaited by the next ccsd_sharded_residuals or by ccsd_sharded_await)
    int ccsd_sharded_finish(const double* f, const double* t1, const double* tc, const double* dtc, const ShardBuffers& b);
    void ccsd_sharded_energy(int slot, double out[6]);
    void ccsd_sharded_await(double* t2, const ShardBuffers& b);
    void invalidate_static();

    double* eps_o = nullptr;
    double* eps_v = nullptr;
    bool eps_set = false;              // set_orbital_energies has been called (mp2 and the amplitude updates divide by them)
    void need_eps(const char* who) const;
    double* splitk_ws() const { return splitk_ws_; }
    int64_t splitk_ws_doubles() const { return splitk_doubles_; }

  private:
    // pair layouts Td, Tx, Tt_d of the amplitudes (cc.cpp), kept between residual_slab and the calls of the same iteration
    // that read them again (singles residual, residual_finish): persistent buffers, valid for the t2 pointer recorded
    double* lay_[3] = {nullptr, nullptr, nullptr};
    const double* lay_t2_ = nullptr;
    void pair_layouts_of(const double* t2);
    // S_ki = sum_cdl Tt[c,d,i,l] V[l,k,d,c] and S_ac = sum_dkl Tt[a,d,k,l] V[l,k,d,c] (ccd.py:213-220), or this rank's partial
    // sums of them: X_ki, X_ac AND the singles residual (ccsd.py:434, :436 are the same sums for exchange-symmetric V, T)
    // read them; valid for the t2 pointer recorded, from the producer (residual_slab / slab_prepare / xvv_partial) to the
    // residual_finish of the same iteration
    struct SumTag {
        const double* t2 = nullptr;
        int rank = 0, world = 0;          // whose share of the sum (world 1: the whole sum)
        bool is(const double* p, int r, int w) const { return t2 && t2 == p && rank == r && world == w; }
        void set(const double* p, int r, int w) { t2 = p; rank = r; world = w; }
        void clear() { t2 = nullptr; }
    };
    // weight of Ex_x in the direct placement that the assembly adds (residual_slab -> residual_finish[_pairs] of the same
    // iteration): 1/2 when the D-term product was computed without the half of the C-term (paired ring products), else 0
    double ring_xd_ = 0.0;
    double* xs_oo_ = nullptr;
    double* xs_vv_ = nullptr;
    SumTag xs_oo_tag_, xs_vv_tag_;
    void ensure_xs();
    dev::stream_t own_stream_ = nullptr;
    std::set<void*> user_allocs_;
    // exchange / staging buffers of ccsd_residuals (engine scratch, held from the first call to release_residual_buffers)
    Collectives coll_;
    alltoallv_fn alltoallv_ = nullptr;
    double *xs_ = nullptr, *xr_ = nullptr;
    struct Rect { int64_t r0, r1, c0, c1; };
    std::vector<Rect> owner_tile_rects(int from, int to, int world) const;
    int64_t owner_tiles_start(const ShardBuffers& b);
    void owner_tiles_finish(const ShardBuffers& b);
    bool coll_set_ = false, t2_in_flight_ = false;
    int64_t t2_ticket_ = 0;
    double *res_fd_ = nullptr, *res_ETd_ = nullptr, *res_ETx_ = nullptr, *res_L_ = nullptr, *res_QK_ = nullptr, *res_r1_ = nullptr,
           *res_r2_ = nullptr;
    std::multimap<int64_t, double*> scratch_free_;
    std::map<double*, int64_t> scratch_live_;
    std::set<dev::graph_t> graphs_;
    std::set<dev::graph_t> graphs_dressing_;     // recorded graphs whose replay dresses V again
    bool release_wanted_ = false;                // release_residual_buffers while recorded graphs may still replay into them
    bool capturing_ = false;
    uint64_t dress_generation_ = 0, capture_generation_ = 0;
    double* V_[16] = {nullptr};      // undressed blocks (owned)
    double* Vd_[16] = {nullptr};     // dressed blocks (owned, allocated on demand)
    std::map<std::string, double*> static_;   // cached permutations of static blocks (owned)
    struct LadderPack {   // V^+ / V^- rows of the pair-packed ladder
        double* Vp = nullptr;
        double* Vm = nullptr;
        int64_t row0 = 0, row1 = 0;
        bool dressed = false, valid = false;
    } lpack_;
    bool bra_dress_pays() const;
    bool dress_off_ = false;     // the dressed copy did not fit the device memory once: Q_kb form from then on
    double* splitk_ws_ = nullptr;
    int64_t splitk_doubles_ = 0;
    double* get_static(const std::string& key);
    // intermediates of one dress_V call that more than one block needs: (pattern, depth, transformed positions) -> view
    std::map<long, TView> dress_memo_;
    void dressed_into(int pattern, const std::vector<int>& pos, int k, const TView& t1v, const TView& dst,
                      bool reduced = false, const int64_t* cut = nullptr);
    int64_t block_size(int pattern) const;
    TView block_view(double* p, int pattern) const;
    double* ensure_dressed(int pattern);
};

}  // namespace pymes
