/* pymes_amd — C-ABI of the MI355X-native CCSD/DCSD amplitude-update engine.
 *
 * This is the drop-in boundary for the hot path of nickirk/pymes (reference @
 * 2024_10_08).  The reference has no FFI: its seam is the module-level `einsum`
 * callable (pymes/solver/ccsd.py:11, mp2.py:5; historically `ctf.einsum`,
 * pymes/__init__.py:3) and the solver-class methods built on it.  Each entry point
 * below names the reference function it replaces.  pymes_amd/_lib.py binds exactly
 * these symbols with ctypes; INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; pymes_last_error()
 *     then returns a thread-local message.  No exception crosses the boundary.
 *   - all tensors are fp64, C-contiguous unless explicit strides (in elements) are
 *     given; "dev" pointers are device (HBM) addresses, "host" pointers host memory.
 *   - shapes follow the reference: T2 is [a,b,i,j] = (nv,nv,no,no), T1 is [a,i],
 *     Fock is [n,n], V[p,q,r,s] = <pq|rs>, block names are partition.py's
 *     (i,j,k,l occupied; a,b,c,d virtual).
 *   - one context per host thread; all work of a context is ordered on one HIP stream.
 */
#ifndef PYMES_AMD_H
#define PYMES_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pymes_ctx pymes_ctx;

/* ---- library / context ------------------------------------------------------- */
const char* pymes_last_error(void);
const char* pymes_backend(void);                 /* "hip-gfx950" for the product library */
int pymes_ctx_create(pymes_ctx** out, int device, int no, int nv, uint64_t workspace_bytes /* 0 = auto */);
int pymes_ctx_destroy(pymes_ctx* ctx);
/* A context runs on a HIP stream of its own; pymes_ctx_set_stream binds another one (after completing what was
 * enqueued on the old one) — e.g. torch's current stream, so that RCCL collectives and engine kernels are ordered on
 * the device without host fences (pymes_amd/dist.py). */
int pymes_ctx_set_stream(pymes_ctx* ctx, void* hip_stream);
int pymes_ctx_sync(pymes_ctx* ctx);
int pymes_ctx_workspace(pymes_ctx* ctx, uint64_t* capacity_bytes, uint64_t* high_water_bytes);

/* ---- device memory ------------------------------------------------------------ */
/* buffers belong to the context: pymes_ctx_destroy releases whatever has not been passed to pymes_free */
int pymes_malloc(pymes_ctx* ctx, uint64_t bytes, void** dev_ptr);
int pymes_free(pymes_ctx* ctx, void* dev_ptr);
int pymes_live_allocations(int64_t* n);          /* device allocations of this library not yet released (leak tests) */
int pymes_mem_info(pymes_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes);   /* hipMemGetInfo of the context's device */
/* Number of T1 dressings the context's dressed blocks have seen: every get_T1_dressed_V-shaped call (pymes_ccsd_dress_V,
 * _slab), every pymes_ccsd_residuals / _iterate / _sharded_residuals with T1 != 0 and every replay of a recorded graph that
 * contains one.  A holder of dressed blocks (ccsd.py:290-421 hands out a dictionary; here the blocks stay in the context)
 * remembers the value and refuses to read blocks that have been dressed again since. */
int pymes_dress_generation(pymes_ctx* ctx, uint64_t* n);

/* ---- phase launches.  The small kernels between two big products (a (20,80) iteration has ~60 of 5-35 us) are not launched
 * one by one: inside the library they are recorded with the address ranges they read and write and launched level by level of
 * their dependency graph, ONE grid per level carrying the blocks of all its tasks (a dependent launch costs 1.9 us on
 * this chip, so the seams stay kernel boundaries and what goes is the serialisation of INDEPENDENT kernels).  Every entry point
 * of this header leaves nothing recorded behind when it returns; results are the same as with immediate launches.
 * pymes_phase_enable: 1 on, 0 off, -1 as the environment says (PYMES_PHASE=0 off, =serial one task per level; default on) — per
 * calling thread.  pymes_phase_stats: tasks recorded / grids launched / levels / flushes by the calling thread so far. */
int pymes_phase_enable(int mode);
/* pymes_phase_hold(1): the calls that follow leave their small operations recorded instead of launching them when they return,
 * so that operations of SEVERAL calls share levels (an iteration's two amplitude updates, pymes_cc_update_to, and the overlaps
 * of its mixer); pymes_phase_hold(0) launches them.  Anything that synchronises or copies launches them first, held or not. */
int pymes_phase_hold(int on);
int pymes_phase_stats(int64_t* tasks, int64_t* launches, int64_t* levels, int64_t* flushes);

/* ---- launch graphs (hipGraph): the loop body of a small, launch-bound solve is recorded once and replayed.
 * Between begin and end the context's entry points only RECORD their kernels (nothing executes, nothing may
 * synchronise: no pymes_dots / pymes_download / pymes_free); every pointer passed in is baked into the graph, so the
 * replayed segment must work on fixed buffers (pymes_amd/solver/ccsd.py keeps T1/T2 and the residuals in place).
 * The reference has no counterpart: its loop body (ccsd.py:159-209) is re-interpreted by Python every iteration. */
int pymes_graph_begin(pymes_ctx* ctx);
int pymes_graph_end(pymes_ctx* ctx, void** graph);
int pymes_graph_abort(pymes_ctx* ctx);
int pymes_graph_launch(pymes_ctx* ctx, void* graph);
int pymes_graph_destroy(pymes_ctx* ctx, void* graph);
int pymes_upload(pymes_ctx* ctx, void* dst_dev, const void* src_host, uint64_t bytes);   /* synchronous */
int pymes_download(pymes_ctx* ctx, void* dst_host, const void* src_dev, uint64_t bytes); /* synchronous */
int pymes_copy(pymes_ctx* ctx, void* dst_dev, const void* src_dev, uint64_t bytes);      /* stream-ordered */
int pymes_memset_zero(pymes_ctx* ctx, void* dst_dev, uint64_t bytes);

/* ---- generic tensor engine: the `einsum` seam (ccsd.py:11, mp2.py:5) ------------ */
/* C[lc] = alpha * sum_{contracted} A[la] * B[lb] + beta * C[lc]; one-letter labels; strides
 * may be NULL (contiguous); `batch` lists free labels to run as GEMM batches ("" = none). */
int pymes_contract(pymes_ctx* ctx, double alpha,
                   const double* A_dev, const char* la, const int64_t* dimA, const int64_t* strideA,
                   const double* B_dev, const char* lb, const int64_t* dimB, const int64_t* strideB,
                   double beta, double* C_dev, const char* lc, const int64_t* dimC, const int64_t* strideC,
                   const char* batch);
/* out[lo] = alpha * in[li] + beta * out[lo]; `lo` is a permutation of `li` */
int pymes_permute(pymes_ctx* ctx, double alpha, const double* in_dev, const char* li, const int64_t* dim_in,
                  const int64_t* stride_in, double beta, double* out_dev, const char* lo,
                  const int64_t* stride_out);
/* raw fp64 MFMA GEMM (kernel-level tests): C(m,n) = alpha*sum_k A(m,k)B(k,n) + beta*C(m,n),
 * A(m,k) = A[m*a_sm + k*a_sk], B(k,n) = B[k*b_sk + n*b_sn], C(m,n) = C[m*ldc + n] */
int pymes_dgemm(pymes_ctx* ctx, int64_t M, int64_t N, int64_t K, double alpha, const double* A_dev, int64_t a_sm,
                int64_t a_sk, const double* B_dev, int64_t b_sk, int64_t b_sn, double beta, double* C_dev,
                int64_t ldc);

/* ---- integrals: pymes/integral/partition.py:4-39 -------------------------------- */
/* V[n,n,n,n] (host: C-contiguous, strides ignored; device: element strides or NULL) -> 16 blocks */
int pymes_set_V_pqrs(pymes_ctx* ctx, const double* V, int on_device, const int64_t* strides);
/* a single block by its partition.py name ("abcd", "ijab", ...); n_elements must be the block's size for this
 * context (a mis-shaped block is refused instead of being read out of bounds) */
int pymes_set_V_block(pymes_ctx* ctx, const char* name, const double* data, int64_t n_elements, int on_device,
                      const int64_t* strides);
/* Electron-exchange symmetry V_pqrs = V_qpsr, which the symmetry-reduced residual (PYMES_SYM_LADDER / PYMES_SYM_RINGS)
 * presumes and the reference's own Ex += Ex^T (ccd.py:245-249) relies on: out = { max |V_pqrs - V_qpsr| over the blocks
 * that are set (infinity when a block's partner is absent), max |V| }.  pymes_exchange_asymmetry is the same test for
 * any pair A [d0,d1,d2,d3], B [d1,d0,d3,d2] on the device (T2 against itself: T_abij = T_baji). */
int pymes_V_exchange_asymmetry(pymes_ctx* ctx, double* out_host);
int pymes_exchange_asymmetry(pymes_ctx* ctx, const double* A_dev, const double* B_dev, const int64_t* dims,
                             double* out_host);
/* V[p,q,r,s] = sum_Q B[Q,p,r] B[Q,q,s]  (density-fitted / synthetic input), B_host is [naux,n,n] */
int pymes_set_V_from_factors(pymes_ctx* ctx, const double* B_host, int naux);
/* device address of a block ("dressed" = output of pymes_ccsd_dress_V); NULL+error if absent */
int pymes_V_block_ptr(pymes_ctx* ctx, const char* name, int dressed, double** dev_ptr, int64_t* n_elements);
/* diag(f): eps_o[no], eps_v[nv] used by the denominators (ccsd.py:149-156) */
int pymes_set_orbital_energies(pymes_ctx* ctx, const double* eps_o_host, const double* eps_v_host);

/* ---- the CC hot path ------------------------------------------------------------- */
/* pymes/solver/mp2.py:9-22: T2 = V_abij/(D+shift) into t2_dev; e_out = {direct, exchange} */
int pymes_mp2(pymes_ctx* ctx, double level_shift, double* t2_dev, double* e_out_host);
/* CCSD.get_T1_dressed_fock, ccsd.py:226-288 (f and fd are [n,n] on the device) */
int pymes_ccsd_dress_fock(pymes_ctx* ctx, const double* f_dev, const double* t1_dev, double* fd_dev);
/* CCSD.get_T1_dressed_V, ccsd.py:290-421; block_mask bit p = pattern id of the block
 * (bit 3-pos set when index `pos` is virtual: "abcd"=15, "klij"=0, "ijab"=3, "abij"=12 ...).
 * PYMES_DRESS_ABIJ_REDUCED: V~_abij without its V_pqcd t_ci t_dj and t_ak t_bl V~_klrs terms, i.e. what is left of
 * it once those terms travel with the ladders (the amplitude-side mode of pymes_residual_finish forms the same
 * quantity implicitly; the explicit block is kept for inspection and tests). */
#define PYMES_DRESS_ABIJ_REDUCED (1u << 16)
int pymes_ccsd_dress_V(pymes_ctx* ctx, const double* t1_dev, uint32_t block_mask);
/* the same for the range [p_begin,p_end) of the FIRST and [q_begin,q_end) of the SECOND index only (an empty range =
 * the whole index; a block in which the restricted index is occupied is dressed as a whole, so "klij" may ride in the same
 * call and share its V_klcd t_dj intermediate with "iabj"): "iajb" / "iabj" with a q-range is what
 * one rank's column slab of pymes_residual_slab reads, "abij" with (p,q) = (rows a of the rank's pairs, b below their
 * end) and the transposed pair of ranges is what the pair-sharded tail reads — no exchange */
int pymes_ccsd_dress_V_slab(pymes_ctx* ctx, const double* t1_dev, uint32_t block_mask, int p_begin, int p_end,
                            int q_begin, int q_end);
/* CCSD.get_singles_residual, ccsd.py:423-438 */
int pymes_ccsd_singles_residual(pymes_ctx* ctx, const double* fd_dev, const double* t1_dev, const double* t2_dev,
                                double* r1_dev);
/* The same residual as a partial sum over rank `rank`'s chunk of the occupied summation index (one process per GPU; needs
 * T_abij = T_baji): the caller all-reduces the [nv][no] results (f~_ai enters on rank 0).  world = 1 is the whole residual. */
int pymes_ccsd_singles_residual_partial(pymes_ctx* ctx, const double* fd_dev, const double* t1_dev, const double* t2_dev,
                                        double* r1_dev, int rank, int world, uint32_t flags /* PYMES_REUSE_LAYOUTS or 0 */);
/* CCD.get_residual, ccd.py:164-254 (and CCSD.get_doubles_residual, ccsd.py:440-456, with
 * PYMES_USE_DRESSED).  flags: */
#define PYMES_DCD 1u          /* is_dcd / is_dcsd */
#define PYMES_USE_DRESSED 2u  /* read the T1-dressed blocks instead of the undressed ones */
#define PYMES_SKIP_LADDER 4u  /* leave out V_abcd.T (ccd.py:187) — it is added per slab by pymes_ladder[_sym] */
#define PYMES_SYM_LADDER 8u   /* evaluate V_abcd.T in pair-packed form (1/4 of the flops); requires
                                 V_abcd = V_badc and T_cdij = T_dcji, true for every closed-shell solve that
                                 starts from MP2 or from symmetric amplitudes */
#define PYMES_SLAB_RINGS_ONLY 64u   /* pymes_residual_slab: only the ring products (rows of ETd / ETx) ... */
#define PYMES_SLAB_LADDERS_ONLY 128u /* ... only the ladders (rows of L, Q_kb).  One process per GPU calls the two halves
                                 separately: the all-gathers of ETd / ETx are started after the first and fly while the
                                 second — whose output stays on the rank — is computed */
#define PYMES_REUSE_LAYOUTS 32u /* the caller's promise that t2 has not changed since the preceding pymes_residual_slab call
                                 on it: its pair layouts (Td, Tx, 2T - T^(ab)) are read again instead of being rebuilt
                                 (pymes_residual_finish, pymes_ccsd_singles_residual_partial) */
#define PYMES_T1_ZERO (1u << 20) /* pymes_ccsd_residuals / _iterate: the caller knows that t1 == 0 exactly (the MP2 start; every
                                  * pass of a momentum-conserving system): the residuals come from the undressed f and V */
#define PYMES_SYM_RINGS 16u   /* same precondition: merge the o^3v^3 ring/exchange products through the symmetry of
                                 the pair matrices (C / D form: 4 products instead of 10; 3 instead of 5 for DCSD) */
int pymes_doubles_residual(pymes_ctx* ctx, const double* f_dev, const double* t2_dev, double* r2_dev,
                           uint32_t flags);
/* the particle-particle ladder on an a-slab (ccd.py:187): R[a0:a1] = beta*R[a0:a1] + V_abcd[a0:a1].T */
int pymes_ladder(pymes_ctx* ctx, const double* t2_dev, double* r2_dev, int a_begin, int a_end, int dressed,
                 double beta);
/* pair-packed form of the same term.  L is [v(v+1)/2][o*o] on the device; row r = P(a,b) = a(a+1)/2+b
 * (a >= b) holds [ LS (o(o+1)/2 entries, P(i,j)) | LA (o(o-1)/2 entries, Q(i,j) = i(i-1)/2+j) ].
 * pymes_ladder_sym fills rows [row_begin,row_end) (the shardable unit); pymes_ladder_sym_unpack adds
 * R[a,b,i,j] = beta R + LS + sgn(a-b) sgn(i-j) LA from a complete L.
 * hole_ladder: 0 = ccd.py:187 only; 1 / 2 = also add the hole ladder sum_kl I_klij T_abkl (ccd.py:175-186) to the
 * same rows, evaluated pair-packed as well (I_klij = I_lkji), with I = V_klij + V_klcd T_cdij (1, CCD/CCSD) or
 * I = V_klij (2, DCD/DCSD). */
int pymes_ladder_sym(pymes_ctx* ctx, const double* t2_dev, double* L_dev, int64_t row_begin, int64_t row_end,
                     int dressed, int hole_ladder);
int pymes_ladder_sym_unpack(pymes_ctx* ctx, const double* L_dev, double* r2_dev, double beta);
/* The particle ladder of pymes_ladder_sym (all rows, no hole ladder) for k exchange-symmetric vectors at once:
 * x_dev[z] are k device arrays [v,v,o,o], L_all_dev k consecutive arrays [v(v+1)/2][o*o].  One batched GEMM launch per half
 * over all vectors — the sigma builds of every vector of an EOM-CCSD Davidson pass (eom_ccsd.py:95-101, :383). */
int pymes_ladder_sym_multi(pymes_ctx* ctx, const double* const* x_dev, int k, double* L_all_dev, int dressed);
/* T1 dressing of the bra of the pair-packed V_abcd (ccsd.py:414-419, as far as the pair-packed ladder reads it):
 * V_dev, W_dev [r1 - r0][ld] hold the rows P(a,b) = a(a+1)/2 + b in [r0,r1) of one half of the packed block (columns (c,d)
 * packed, pitch ld = a multiple of 16 doubles), Pk_dev [v*o][ld] the rows x*o + k of V_kxcd packed the same way (x slow), t1_dev [v,o]:
 *   W[P(a,b)] = V[P(a,b)] - sum_k t1[a,k] Pk[(b,k)] -+ sum_k t1[b,k] Pk[(a,k)]
 * ("-" for the symmetric half V_abcd + V_abdc, "+" with minus_half = 1 for V_abcd - V_abdc, whose rows a == b stay zero).
 * nocc <= 64; the packed rows of 16 consecutive a within 2 GB (nvirt up to ~320). */
int pymes_ladder_dress(pymes_ctx* ctx, const double* V_dev, const double* Pk_dev, const double* t1_dev, double* W_dev,
                       int64_t ld, int64_t r0, int64_t r1, int minus_half);
/* The three pair layouts of an amplitude-like array X [v,v,o,o] in one pass: Xd[(a,i),(b,j)] = X_abij,
 * Xx[(a,j),(b,i)] = X_abij, Xt[(a,i),(b,j)] = 2 X_abij - X_baij (each [o*v][o*v]; Xd_dev may be NULL).  ccd.py:199 /
 * eom_ccsd.py:352-373 form these index orders implicitly inside their einsum calls. */
int pymes_pair_layouts(pymes_ctx* ctx, const double* x_dev, double* Xd_dev, double* Xx_dev, double* Xt_dev);
/* P(ijab,jiba) symmetrisation and assembly in one pass (ccd.py:249-252, eom_ccsd.py:377):
 *   R_abij = V_abij + unpack(L)_abij + N_abij + N_baji + D[(a,i),(b,j)] + D[(b,j),(a,i)] + X[(a,j),(b,i)] + X[(b,i),(a,j)]
 * with V [v,v,o,o] (NULL: 0; may be R), L the pair-packed rows of pymes_ladder_sym (NULL: none), N [v,v,o,o], D and X
 * [o*v][o*v] pair matrices ((a,i) = a*o + i).  Needs o(o+1) doubles of LDS (pymes_residual_finish uses it internally). */
int pymes_symmetrised_assemble(pymes_ctx* ctx, const double* V_dev, const double* L_dev, const double* N_dev,
                               const double* D_dev, const double* X_dev, double* R_dev);
/* Rows [row_begin,row_end) of L += pair-packed sum_kl I_klij X_abkl for a caller-supplied I [o,o,o,o] with I_klij = I_lkji
 * and X [v,v,o,o] with X_abkl = X_balk: the hole-ladder-shaped terms of the EOM-CCSD sigma, eom_ccsd.py:380-382
 * (u2 against V_klij + V_klcd T_cdij; T against V_kldc u2_dcij), at 1/4 of the flops of the plain product.  y_dev
 * (optional, [v,v,o,o], y_cdij = y_dcji): sum_cd V_klcd y_cdij is added to I on the way, formed pair-packed as well (the
 * caller then passes only the y-independent part of I). */
int pymes_hole_ladder_packed(pymes_ctx* ctx, const double* x_dev, const double* I_dev, double* L_dev, int64_t row_begin,
                             int64_t row_end, const double* y_dev);
/* pymes_hole_ladder_packed (all rows) for k vectors in batched launches: L_all_dev[z] += rows(x_dev[z]) . (2 I_dev[z]
 * [+ 2 V_klcd y_dev[z]_cdij]); entries of x_dev (or of I_dev) may all be the same array, which is then packed once —
 * eom_ccsd.py:380-382 over the vectors of a Davidson pass.  y_dev may be NULL. */
int pymes_hole_ladder_packed_multi(pymes_ctx* ctx, const double* const* x_dev, const double* const* I_dev,
                                   const double* const* y_dev, int k, double* L_all_dev);
/* ---- whole steps of the CCSD / DCSD loop body (ccsd_residuals, ccsd_iterate) ---------------------------------------
 * pymes_ccsd_residuals: ccsd.py:161-171 in ONE call — T1-dressed Fock matrix (:163), the dressed blocks the loop reads (:165),
 * R1 [v,o] (:167) and R2 [v,v,o,o] (:171) from f [n,n], t1 [v,o], t2 [v,v,o,o] (all on the device).  The symmetry-reduced
 * form on one rank: needs V_pqrs = V_qpsr (pymes_V_exchange_asymmetry) and T_abij = T_baji (pymes_exchange_asymmetry) — the
 * caller's promise; anything else goes through pymes_ccsd_doubles_residual.  Only enqueues kernels, on buffers the context
 * holds from the first call to pymes_ccsd_release (replayable as a launch graph between those two).
 * pymes_ccsd_iterate: one fixed-point pass without a mixer (ccsd.py:159-197 with is_diis = False): residuals, dT = R / (D +
 * level_shift) (pymes_set_orbital_energies first), T += delta dT in place, energies and norms of the updated amplitudes:
 * out6 = {one-body, direct, exchange, |t2|^2, |dt2|^2, |t1|^2} (host).  With a mixer: pymes_ccsd_residuals,
 * pymes_cc_update_to, pymes_diis_mix, pymes_energy_norms — the sequence of pymes_amd/solver/ccsd.py. */
int pymes_ccsd_residuals(pymes_ctx* ctx, const double* f_dev, const double* t1_dev, const double* t2_dev, uint32_t flags,
                         double* r1_dev, double* r2_dev);
int pymes_ccsd_iterate(pymes_ctx* ctx, const double* f_dev, double* t1_dev, double* t2_dev, uint32_t flags, double level_shift,
                       double delta, double* dt1_dev, double* dt2_dev, double* out6_host);
int pymes_ccsd_release(pymes_ctx* ctx);

/* The symmetry-reduced residual (PYMES_SYM_LADDER | PYMES_SYM_RINGS) in its shardable form, one process per
 * GPU.  pymes_residual_slab computes what rank `rank` of `world` owns: the rows [c0,c1) of ETd and ETx (both
 * [o*v][o*v] on the device; ET[(b,j),(a,i)] = Ex[(a,i),(b,j)], rows cut into `world` chunks of ceil(ov/world))
 * and, if L_dev is not NULL, its chunk of rows of the pair-packed ladders L (pymes_ladder_sym with the hole ladder included).  No
 * communication is needed to produce a slab.  After the slabs have been exchanged (one all-gather per buffer)
 * pymes_residual_finish adds the replicated terms and assembles R.  world = 1 reproduces
 * pymes_doubles_residual.  pymes_ccsd_dress_abcd_rows dresses only rows a in [a_begin,a_end) of V_abcd
 * (ccsd.py:414-419) — the rows a rank's ladder chunk reads; with lower_only != 0 only the entries b <= a
 * (all the pair-packed ladder touches) are defined afterwards.
 *
 * t1_dev / QK_dev (both or neither; CCSD/DCSD with PYMES_USE_DRESSED): the T1 dressing of V_abcd (ccsd.py:414-419)
 * is applied on the amplitude side instead — V~_abcd T = sum_pq X_a^p X_b^q V_pqcd T_cdij is evaluated from the
 * undressed, once-packed V_abcd / V_kbcd / V_klcd with tau = T + t1 t1.  V_abcd is then never dressed (no
 * pymes_ccsd_dress_abcd_rows, no second copy of V_abcd) and neither is V_abij: pymes_residual_finish reads it
 * undressed and adds its T1 dressing through Ex + Ex^T as V_abcj t_ci - t_ak (V_kbij + V_kbcj t_ci + V_kbid t_dj + Q_kbij)
 * (the (k,l)-bra part rides in the hole ladder taken with tau).  Only V~_klij, V~_iajb, V~_iabj have to be dressed.
 * QK is a fourth exchange buffer, [o*v][o*o] on the device cut into the same row chunks
 * as ETd: QK[(k,b)][i][j] = sum_cd V_kbcd tau_cdij + V_kbij + V_kbcj t_ci + V_kbid t_dj — the whole bracket above, formed
 * only for the rows (k,b) of the rank. */
int pymes_residual_slab(pymes_ctx* ctx, const double* f_dev, const double* t2_dev, double* ETd_dev, double* ETx_dev,
                        double* L_dev, int rank, int world, uint32_t flags, const double* t1_dev, double* QK_dev,
                        const double* P_dev);
/* P_dev (optional, amplitude-side mode only): the slab's small replicated intermediates — w Tt_cdil V_lkdc (the V.T part
 * of X_ki, ccd.py:215-220) and the pair-packed 2 V_klcd T_cdij of the hole ladder (:180) — as produced by
 * pymes_slab_prepare (every rank sums over its chunk of c / of the pairs (c,d)) and all-reduced by the caller;
 * pymes_slab_prepare_ws doubles.  NULL: the slab forms them itself. */
int pymes_slab_prepare_ws(pymes_ctx* ctx, int64_t* n_doubles);
int pymes_slab_prepare(pymes_ctx* ctx, const double* t2_dev, double* P_dev, int rank, int world, uint32_t flags);
int pymes_residual_finish(pymes_ctx* ctx, const double* f_dev, const double* t2_dev, const double* ETd_dev,
                          const double* ETx_dev, const double* L_dev, double* r2_dev, uint32_t flags,
                          const double* t1_dev, const double* QK_dev);
int pymes_ccsd_dress_abcd_rows(pymes_ctx* ctx, const double* t1_dev, int a_begin, int a_end, int lower_only);
/* Pair-sharded tail of the iteration (world > 1).  Rank r owns the virtual pairs P(a,b) = a(a+1)/2+b, a >= b, of chunk
 * r of v(v+1)/2 (ceil(npp/world) pairs per chunk, the rows of L it produced in pymes_residual_slab) and keeps every
 * amplitude-sized quantity only for them, in the compact layout Xc[P - r0][2][o*o]: tile 0 = X[a,b,:,:], tile 1 =
 * X[b,a,:,:] (zeros for a == b, so that dots over Xc summed over the ranks equal dots over the full array).
 *   pymes_residual_finish_pairs: pymes_residual_finish restricted to the rank's pairs -> Rc (L is read locally and is
 *                                NOT exchanged; ETd, ETx (and QK) must have been all-gathered); t1/QK NULL for CCD/DCD
 *   pymes_cc_update_pairs:       ccsd.py:176-179 on compact tiles (dt = r/(D+shift), t += delta dt)
 *   pymes_pairs_pack:            compact tiles of this rank from a full [v,v,o,o] array
 *   pymes_pairs_unpack:          full [v,v,o,o] array from the all-gathered compact buffer [world * chunk][2][o*o] */
int pymes_residual_finish_pairs(pymes_ctx* ctx, const double* f_dev, const double* t2_dev, const double* ETd_dev,
                                const double* ETx_dev, const double* L_dev, double* Rc_dev, uint32_t flags,
                                const double* t1_dev, const double* QK_dev, int rank, int world, const double* Xvv_dev);
/* Small replicated intermediates as K-sharded partial sums (world > 1): every rank sums over its chunk of one occupied
 * index and the v x v / o x v results are all-reduced by the caller.
 *   pymes_xvv_partial:              X_ac = f_ac - w Tt_adkl V_lkdc (ccd.py:206-221), k in the rank's chunk, f on rank 0;
 *                                   the all-reduced matrix is handed to pymes_residual_finish_pairs (Xvv_dev, else NULL)
 *   pymes_ccsd_dress_fock_partial:  the six T1.V intermediates of ccsd.py:226-288 (j in the rank's chunk) into W_dev
 *                                   (pymes_ccsd_dress_fock_ws doubles); pymes_ccsd_dress_fock_finish completes f~ from the
 *                                   all-reduced W.  pymes_ccsd_dress_fock = partial(0, 1) + finish. */
int pymes_xvv_partial(pymes_ctx* ctx, const double* f_dev, const double* t2_dev, double* Xvv_dev, int rank, int world,
                      uint32_t flags);
int pymes_ccsd_dress_fock_ws(pymes_ctx* ctx, int64_t* n_doubles);
int pymes_ccsd_dress_fock_partial(pymes_ctx* ctx, const double* t1_dev, double* W_dev, int rank, int world);
int pymes_ccsd_dress_fock_finish(pymes_ctx* ctx, const double* f_dev, const double* t1_dev, const double* W_dev,
                                 double* fd_dev);
int pymes_cc_update_pairs(pymes_ctx* ctx, double* tc_dev, double* dtc_dev, const double* rc_dev, double level_shift,
                          double delta, int rank, int world);
int pymes_pairs_supported(pymes_ctx* ctx, int* yes);   /* the o x o tile of the fused pair kernels fits the LDS */
int pymes_pairs_pack(pymes_ctx* ctx, const double* full_dev, double* xc_dev, int rank, int world);
int pymes_pairs_unpack(pymes_ctx* ctx, const double* xc_all_dev, double* full_dev, int world);
/* ccsd.py:176-179 / ccd.py:123-124: dt = r/(D+shift) (as r * (1/(D+shift))), t += delta*dt; rank 2 or 4 */
int pymes_cc_update(pymes_ctx* ctx, double* t_dev, double* dt_dev, const double* r_dev, double level_shift,
                    double delta, int rank);
/* out-of-place form of the same update: dt = r/(D+shift), t_out = t_in + delta*dt (t_out may alias t_in).  Lets the
 * residuals read T from a fixed buffer (launch graphs) while the DIIS history keeps the updated copy. */
int pymes_cc_update_to(pymes_ctx* ctx, double* t_out_dev, double* dt_dev, const double* t_in_dev, const double* r_dev,
                       double level_shift, double delta, int rank);
/* ccsd.py:458-466 (ccd.py:256-262 when f_dev and t1_dev are NULL) and the norms of ccsd.py:196-197 in ONE pass over
 * T2: out[6] = {one-body, direct, exchange, |t2|^2, |dt2|^2, |t1|^2} (dt2_dev may be NULL); one host synchronisation.
 * |t1|^2 == 0 exactly (the MP2 start; every iteration of a momentum-conserving system such as the UEG) lets the caller take
 * the T1-free form of the residual: exp(-T1) H exp(T1) = H then, no dressing at all (pymes_amd/solver/ccsd.py). */
int pymes_energy_norms(pymes_ctx* ctx, const double* f_dev, const double* t1_dev, const double* t2_dev,
                       const double* dt2_dev, double* out_host);
/* pymes_energy_norms in two halves.  The reference reads the energy back every iteration (ccsd.py:189-197) and its next
 * residual build waits for that; here _start enqueues the reduction and a copy into a pinned slot (`*slot`), the caller may
 * enqueue more work — the residual kernels of the NEXT iteration, which need the new amplitudes but not the energy — and
 * _wait blocks until that one copy has landed (an event, not the stream).  pymes_readback_start / _wait: the same for any
 * n <= 128 doubles of device memory (the DIIS coefficients for the log lines of pymes/mixer/diis.py:104-111); 16 slots per
 * device, reused in turn. */
int pymes_energy_norms_start(pymes_ctx* ctx, const double* f_dev, const double* t1_dev, const double* t2_dev,
                             const double* dt2_dev, int* slot);
int pymes_energy_norms_wait(pymes_ctx* ctx, int slot, double* out_host);
int pymes_readback_start(pymes_ctx* ctx, const double* dev_ptr, int n, int* slot);
int pymes_readback_wait(pymes_ctx* ctx, int slot, double* out_host, int n);
/* The same six sums over the virtual pairs of rank `rank` of `world` only, from the compact tiles [pairs][2][o*o] of the
 * pair-sharded tail (tc_dev, dtc_dev: pymes_pairs_pack layout): partial sums that the caller all-reduces — the energy of
 * ccsd.py:458-466 / ccd.py:256-262 then needs no replicated T2, so the all-gather of the new amplitudes can fly while the
 * next iteration's T1-only work runs.  The T1 sums out[0], out[5] are non-zero on rank 0 only. */
int pymes_energy_norms_pairs(pymes_ctx* ctx, const double* f_dev, const double* t1_dev, const double* tc_dev,
                             const double* dtc_dev, int rank, int world, double* out_host);

/* ---- one process per GPU as WHOLE STEPS, with the collectives of the host program -----------------------------------------
 * The call sequence above (slab, exchange, finish) leaves the collectives to the caller; here the caller hands the library a
 * table of them once and the loop body of ccsd.py:159-197 for one rank becomes three calls — the N > 1 form is expressible
 * from any host language, no Python and no torch involved (the Python host of this repository fills the table with
 * torch.distributed calls, pymes_amd/dist.py:Collectives; an RCCL host with ncclAllReduce / ncclAllGather on a communicator
 * and a stream of its own).
 *
 * Contract of the table.  All buffers are DEVICE memory; `stream` is the stream the library runs on (hipStream_t).
 *   allreduce_start(user, buf, n, stream, &ticket)      in-place sum over the ranks of n doubles at buf
 *   allgather_start(user, buf, chunk, stream, &ticket)  buf = world consecutive chunks of `chunk` doubles, chunk `rank` is
 *                                                       filled in: on completion every rank holds every chunk
 *   wait(user, ticket, stream)                          orders `stream` behind the completion of that collective
 * Each *_start orders its collective behind the work ALREADY enqueued on `stream` and returns without waiting (the usual
 * RCCL pattern: record an event on `stream`, make the communication stream wait for it, enqueue the collective there, record
 * its completion event; wait = hipStreamWaitEvent(stream, that event)).  Tickets are the host program's.  A blocking
 * implementation (synchronise, exchange through host memory, return) is equally valid — the test rigs do that over gloo.
 * Collectives are issued in the same order on every rank.  Return 0 for success; anything else fails the library call.
 *   mark(user, phase)   optional (may be NULL): called at the phase boundaries of a step, for profiling. */
typedef struct pymes_collectives {
    void* user;
    int rank, world;
    int (*allreduce_start)(void* user, double* buf_dev, int64_t n, void* stream, int64_t* ticket);
    int (*allgather_start)(void* user, double* buf_dev, int64_t chunk, void* stream, int64_t* ticket);
    int (*wait)(void* user, int64_t ticket, void* stream);
    void (*mark)(void* user, const char* phase);
} pymes_collectives;
/* the table is copied; NULL removes it.  One table per context (= per GPU = per process). */
int pymes_set_collectives(pymes_ctx* ctx, const pymes_collectives* table);
/* Buffers that take part in a collective belong to the caller (a host that registers memory with its communicator does so
 * once).  With chunk(n) = ceil(n / world), doubles each (pymes_shard_buffer_sizes fills sizes[10] in this order):
 *   ETd, ETx  world chunk(o v) x o v      rows of the ring products (pymes_residual_slab), all-gathered
 *   L         world chunk(v(v+1)/2) x o^2 pair-packed ladder rows (stay on the rank)
 *   QK        world chunk(o v) x o^2      Q_kb rows, all-gathered
 *   Tall      world chunk(v(v+1)/2) x 2 o^2   compact tiles of the new T2, all-gathered
 *   W         pymes_ccsd_dress_fock_ws,  Xvv  v^2,  P  pymes_slab_prepare_ws,  R1  v o   partial sums, all-reduced
 *   S         8                           the six energy / norm sums, all-reduced */
typedef struct pymes_shard_buffers {
    double *ETd, *ETx, *L, *QK, *Tall, *W, *Xvv, *P, *R1, *S;
} pymes_shard_buffers;
int pymes_shard_buffer_sizes(pymes_ctx* ctx, int world, int64_t* sizes_out);
/* ccsd.py:161-171 for this rank (symmetry-reduced form: V_pqrs = V_qpsr, T_abij = T_baji; pymes_pairs_supported): dressed
 * Fock matrix into fd_dev [n,n] + the dressed blocks of the rank's slab, ring products (their rows exchanged while the ladders
 * run), ladders, Q_kb, X_ac, the singles residual (all-reduced: complete in buffers->R1 [v,o] on every rank) and the doubles
 * residual of the rank's virtual pairs as compact tiles rc_dev [max(pairs,1)][2][o*o] (pymes_pairs_pack layout; zero on a rank
 * without pairs).  t2_dev [v,v,o,o] is the replicated array: an exchange left in flight by pymes_ccsd_sharded_finish is
 * completed into it before it is read.  Writes nothing but fd_dev, rc_dev and the buffers, and reads nothing from the host:
 * the caller may enqueue it BEFORE it reads the previous pass's energy (pymes_ccsd_sharded_energy).  The update follows with
 * pymes_cc_update (t1, R1) and pymes_cc_update_pairs (tc, rc), a mixer with pymes_dots / pymes_diis_solve / pymes_lincomb over
 * the compact tiles (overlaps summed over the ranks by the caller), as in the single-rank sequence.  flags: PYMES_DCD. */
int pymes_ccsd_sharded_residuals(pymes_ctx* ctx, const double* f_dev, double* fd_dev, const double* t1_dev, double* t2_dev,
                                 const pymes_shard_buffers* buffers, uint32_t flags, double* rc_dev);
/* ccsd.py:189-197 and the hand-over of the new amplitudes (after the caller's mixer, if any, has replaced t1 / tc): the six
 * sums of pymes_energy_norms_pairs all-reduced ON THE DEVICE and copied to the host on the side (`*slot`, read with
 * pymes_ccsd_sharded_energy — after the caller has enqueued the next pymes_ccsd_sharded_residuals if it wishes: nothing on the
 * stream waits for the host), tc into the exchange buffer, the all-gather of the new T2 started.  It is completed by the next
 * pymes_ccsd_sharded_residuals, or by pymes_ccsd_sharded_await (end of the solve: t2_dev then holds the amplitudes). */
int pymes_ccsd_sharded_finish(pymes_ctx* ctx, const double* f_dev, const double* t1_dev, const double* tc_dev,
                              const double* dtc_dev, const pymes_shard_buffers* buffers, int* slot);
int pymes_ccsd_sharded_energy(pymes_ctx* ctx, int slot, double* out_host /* [6], as pymes_energy_norms */);
int pymes_ccsd_sharded_await(pymes_ctx* ctx, double* t2_dev, const pymes_shard_buffers* buffers);
/* CCD / DCD (pymes/solver/ccd.py:93-150; the reference's UEG drivers, BASELINE config 4) as the same whole steps: the loop body
 * of ccd.py:100-121 for this rank — ring products (their rows exchanged while the ladders run), ladders, the doubles residual
 * of the rank's virtual pairs as compact tiles rc_dev.  There is no T1: nothing is dressed, no singles residual; of `buffers`
 * only ETd, ETx, L, Tall and S are used (the others may be NULL).  The pass is completed by pymes_cc_update_pairs, the mixer
 * and pymes_ccsd_sharded_finish / _energy / _await with f_dev = t1_dev = NULL.  flags: PYMES_DCD, PYMES_OWNER_TILES. */
int pymes_ccd_sharded_residuals(pymes_ctx* ctx, const double* f_dev, double* t2_dev, const pymes_shard_buffers* buffers,
                                uint32_t flags, double* rc_dev);
/* OWNER TILES (flag PYMES_OWNER_TILES of pymes_ccsd_sharded_residuals / pymes_ccd_sharded_residuals).  In the pair-sharded
 * tail rank q assembles R only for its virtual pairs P(a,b), a in [a0,a1), and reads of ETd / ETx just the tiles [(a,.),(b,.)]
 * and [(b,.),(a,.)]: rows [0, a1 o) x columns [a0 o, a1 o) and rows [a0 o, a1 o) x columns [0, a0 o).  ONE all-to-all of those
 * rectangles ((50,200) on 8 ranks: 1.06 GB per iteration on the wire instead of 2.33) replaces the two all-gathers.  The host
 * program adds an all-to-all to its table —
 *   alltoallv_start(user, send_dev, send_counts[world], recv_dev, recv_counts[world], stream, &ticket)
 * counts in doubles, the pieces for / from rank 0, 1, ... contiguous in that order in send_dev / recv_dev (ncclSend / ncclRecv
 * in a group, or MPI_Alltoallv with running offsets); ordering and tickets as for the other collectives — and hands over two
 * staging buffers of pymes_owner_tile_sizes doubles.  The library packs and unpacks the rectangles itself. */
#define PYMES_OWNER_TILES (1u << 21)
typedef int (*pymes_alltoallv_fn)(void* user, const double* send_dev, const int64_t* send_counts, double* recv_dev,
                                  const int64_t* recv_counts, void* stream, int64_t* ticket);
int pymes_set_alltoallv(pymes_ctx* ctx, pymes_alltoallv_fn fn /* NULL removes it */);
int pymes_owner_tile_sizes(pymes_ctx* ctx, int rank, int world, int64_t* send_doubles, int64_t* recv_doubles);
int pymes_set_owner_tile_buffers(pymes_ctx* ctx, double* send_dev, double* recv_dev);
/* CCSD.get_energy, ccsd.py:458-466: e_out = {one-body, direct, exchange}; f is the UNDRESSED Fock */
int pymes_ccsd_energy(pymes_ctx* ctx, const double* f_dev, const double* t1_dev, const double* t2_dev,
                      double* e_out_host);
/* CCD.get_energy, ccd.py:256-262: e_out = {direct, exchange} */
int pymes_ccd_energy(pymes_ctx* ctx, const double* t2_dev, double* e_out_host);

/* ---- uniform electron gas integrals: UEG.eval_2b_integrals, pymes/model/ueg.py:265-516 ---------------- */
/* V_dev[n_p^4] = <pq|rs> of the plane-wave basis k_int_host[n_p][3] (sorted by kinetic energy, as
 * UEG.init_single_basis builds it) with lookup table index_map_host[(2 imax + 1)^3] (basis_indices_map).
 * mode 0: Coulomb (correlator None); 1: transcorrelated is_only_2b; 2: is_effect_2b BEFORE its
 * electron-exchange symmetrisation (ueg.py:509-513: use pymes_permute); 3: is_rpa_approx.  The correlator is
 * UEG.trunc (ueg.py:772-800) with parameters k_cutoff, gamma; lattice_cutoff is sumNablaUSquare's (30). */
int pymes_ueg_eval_2b(pymes_ctx* ctx, int n_p, int n_ele, int imax, int mode, double L, double k_cutoff, double gamma,
                      int lattice_cutoff, const int32_t* k_int_host, const int32_t* index_map_host, double* V_dev);

/* The same with the reference's other correlators, evaluated inside the kernels from the same float k^2 the reference forms
 * (so that a cut-off lying exactly on a lattice shell is resolved the way the reference's roundings resolve it):
 * correlator 1 UEG.gaskell (ueg.py:836-883) params {mu, cut}, 2 gaskell_modified (:802-834) {cut}, 3 coulomb (:905-915)
 * {-4 pi gamma}, 4 yukawa (:740-770) {gamma, floor}, 5 stg (:917-935) {gamma^2, floor, -4 pi / gamma}, 6 smooth (:885-903)
 * {kc, kc gamma, (kc gamma)^2}; params[4], derived from the model by pymes_amd/model/ueg.py with the reference's expressions. */
int pymes_ueg_eval_2b_corr(pymes_ctx* ctx, int n_p, int n_ele, int imax, int mode, double L, int lattice_cutoff,
                           int correlator, const double* params_host, const int32_t* k_int_host,
                           const int32_t* index_map_host, double* V_dev);
/* The same for a correlator u(k^2) of the caller's own (any Python callable): every argument of u is |2 pi n / L|^2 for an integer vector n, so u is handed over as
 * two host tables over m = |n|^2 (tab_len >= 3 (lattice_cutoff + 2 imax)^2 + 1): tab_scalar[m] is what the reference's
 * correlator returns when called with a float (the main loop's `d_k_vec.dot(d_k_vec)`, ueg.py:409), tab_array[m] what it
 * returns when called with an ndarray (everything else: sumNablaUSquare, the 3-body contractions) — the two forms of
 * `gaskell` differ at its cut-off.  modes 1-3 as above. */
int pymes_ueg_eval_2b_tab(pymes_ctx* ctx, int n_p, int n_ele, int imax, int mode, double L, int lattice_cutoff,
                          const int32_t* k_int_host, const int32_t* index_map_host, const double* tab_scalar_host,
                          const double* tab_array_host, int64_t tab_len, double* V_dev);

/* ---- vector helpers for DIIS (pymes/mixer/diis.py:65-103) and norms ----------------- */
/* out_host[p] = sum_i x[p][i]*y[p][i], npairs <= 16, deterministic reduction (synchronises) */
int pymes_dots(pymes_ctx* ctx, int npairs, const double* const* x_dev, const double* const* y_dev, int64_t n,
               double* out_host);
/* the same with one length per pair (the DIIS overlaps of T1 and T2 in one launch / one synchronisation) */
int pymes_dots_var(pymes_ctx* ctx, int npairs, const double* const* x_dev, const double* const* y_dev,
                   const int64_t* n, double* out_host);
/* out = sum_k c[k]*x[k], nx <= 8 */
int pymes_lincomb(pymes_ctx* ctx, double* out_dev, int nx, const double* const* x_dev, const double* c_host,
                  int64_t n);
/* Grouped launches of small products.  The reference evaluates its term sequences one einsum after the other
 * (pymes/solver/ccd.py:175-240, eom_ccsd.py:288-383); on the device the small ones are 5-50 us kernels on a quarter-filled
 * chip.  Products (pymes_contract / pymes_dgemm) issued between begin and end are INDEPENDENT by the caller's promise — none
 * reads or accumulates into the output of another — and those that run on 64 x 64 tiles are launched together, up to 16
 * per launch (one launch per dependency level).  Anything else the context enqueues in between launches the queue first.
 * _end reports the grouped launches made and the products they carried (either pointer may be NULL). */
int pymes_gemm_group_begin(pymes_ctx* ctx);
int pymes_gemm_group_end(pymes_ctx* ctx, int64_t* launches, int64_t* products);
/* Tall-skinny subspace algebra of the Davidson driver (pymes/solver/eom_ccsd.py:91 `QR`, :103-109 the subspace matrix
 * B[j,l] = <u_j, w_l>, :122-147 collapse / expansion vectors; reference: numpy on host arrays, vector by vector) and of the
 * FEAST driver (feast_eom_ccsd.py:110-150).  Every vector of a call is read once.
 * pymes_gram: out_host[i*n + j] = <x_i, y_j>, i < m, j < n, vectors of `len` doubles on the device (m, n <= 64);
 * deterministic; synchronises the context's stream.
 * pymes_lincomb_multi: y_j = sum_{i<m} c_host[i*n + j] x_i + beta_host[j] y_j (beta_host NULL = 0; a y_j with beta 0 is
 * never read); an output may be one of the inputs only when m <= 16 and n <= 4. */
int pymes_gram(pymes_ctx* ctx, int m, int n, const double* const* x_dev, const double* const* y_dev, int64_t len,
               double* out_host);
int pymes_lincomb_multi(pymes_ctx* ctx, int m, int n, const double* const* x_dev, const double* c_host,
                        const double* beta_host, double* const* y_dev, int64_t len);
/* One DIIS step (pymes/mixer/diis.py:40-103) with no host round trip.  state_dev: 96 doubles on the device —
 * [0] order of L, [1..81] L (pitch 9), [82..90] coefficients of the last step, [91] 1 if the pseudo-inverse branch
 * (diis.py:85-93) was taken, [92] steps taken; a fresh mixer starts from {1, 0, ...}.  The npairs = ntypes * m overlaps
 * <x_p, y_p> (type-major: p = t * m + i) are reduced on the device, L is shifted / extended (including the reference's
 * full-subspace quirk, diis.py:59-60) and L c = (0,..,0,-1) is solved by one device thread; pymes_lincomb_dev then forms
 * sum_k c[k] x[k] with the coefficients read from state_dev + 82.  Nothing synchronises: the host reads the state
 * (pymes_download) only when it wants to log it. */
/* The same step with the small algebra on the calling host thread: overlaps reduced on the device (one synchronisation),
 * L and the coefficients in state_host (96 doubles, layout as above), the extrapolations out[t] = sum_i c[i] amp_hist[t*m+i]
 * enqueued before the call returns — what the solvers use: the device waits for the host only for the synchronisation itself,
 * not for an interpreter to run numpy.linalg on a 7 x 7 matrix (diis.py:65-103 in one call). */
int pymes_diis_mix(pymes_ctx* ctx, double* state_host, int ntypes, int m, int was_full, const double* const* err_hist_dev,
                   const double* const* err_new_dev, const int64_t* sizes, const double* const* amp_hist_dev,
                   double* const* out_dev);
int pymes_diis_step(pymes_ctx* ctx, double* state_dev, int npairs, const double* const* x_dev, const double* const* y_dev,
                    const int64_t* n, int ntypes, int m, int was_full);
/* Only the small algebra of that step (diis.py:56-103), on the calling host thread and on host arrays: overlaps_host[t * m + i]
 * = <e_i, e_new> of amplitude type t, already complete (one process per GPU: summed over the ranks by the caller).  L and
 * the coefficients in state_host as above; state_host[91] = 2 when L is singular or not finite.  No context, no device. */
int pymes_diis_solve(double* state_host, const double* overlaps_host, int ntypes, int m, int was_full);
int pymes_lincomb_dev(pymes_ctx* ctx, double* out_dev, int nx, const double* const* x_dev, const double* coeff_dev,
                      int64_t n);
/* (yr + i yi)[e] = (mr + i mi)[e] (xr + i xi)[e], e < n: a complex diagonal applied to a complex vector held as two real
 * arrays (y may alias x) — the preconditioner 1 / (z - diag + 0.01) of the FEAST linear solves, feast_eom_ccsd.py:342-343. */
int pymes_cmul(pymes_ctx* ctx, const double* mr_dev, const double* mi_dev, const double* xr_dev, const double* xi_dev,
               double* yr_dev, double* yi_dev, int64_t n);

/* pymes/mean_field/hf.py:14-18 from the context's device blocks: f = h + 2 V_piqi - V_piiq (i occupied); h, f [n,n] host */
int pymes_hf_fock_matrix(pymes_ctx* ctx, const double* h_host, double* f_host);

/* ---- FCIDUMP ingestion: pymes/util/fcidump.py:59-163 with a native text parser -------------------
 * Same semantics as the reference's reader (header by substring match, "value i j k l" -> p r q s, |value| < 1e-19
 * skipped, the three index-swap images restored but not the electron-exchange one, is_tc: only [q,p,s,r]; a body line
 * without exactly five fields is an error).
 *   pymes_fcidump_header:    NELEC, NORB (to size the context)
 *   pymes_fcidump_read_host: everything on the host, V[n,n,n,n] caller-allocated (the drop-in fcidump.read)
 *   pymes_fcidump_load:      e_core / eps[n] / h[n,n] on the host, V packed into the context's 16 device blocks like
 *                            pymes_set_V_pqrs; for NORB > 64 the lines are uploaded in chunks and V_pqrs is filled on
 *                            the device (it never exists on the host; files whose symmetry-related lines disagree are
 *                            refused there because the result would depend on the order of the lines). */
int pymes_fcidump_header(const char* path, int* n_elec, int* n_orb);
int pymes_fcidump_read_host(const char* path, int is_tc, double* e_core, double* eps_host, double* h_host,
                            double* V_host);
int pymes_fcidump_load(pymes_ctx* ctx, const char* path, int is_tc, double* e_core, double* eps_host, double* h_host,
                       int64_t* n_two_electron_lines);

/* ---- packed binary integral files: the on-disk form of the 16 blocks (or of density-fitting factors) -------------
 * The reference has only text ingestion (fcidump.py:124-161, one Python iteration per line) plus the hdf5 branch of its
 * TCDUMP reader (tcdump.py:44-48,88-92).  Format "PYMESPK1" (pymes_amd/csrc/packed.h): 64-byte header, eps[n], h[n,n],
 * then either the 16 partition.py blocks as raw little-endian fp64 (kind 1) or factors B[naux,n,n] with
 * V[p,q,r,s] = sum_Q B[Q,p,r] B[Q,q,s] (kind 2).
 *   pymes_packed_header:         kind, NELEC, NORB, naux (to size the context)
 *   pymes_packed_load:           e_core / eps[n] / h[n,n] on the host, the payload straight into the context's 16
 *                                device blocks (kind 1: streamed block by block; kind 2: formed by the MFMA GEMM)
 *   pymes_packed_write:          the context's 16 blocks + the caller's one-body data -> file (kind 1)
 *   pymes_packed_write_factors:  host factors -> file (kind 2) */
int pymes_packed_header(const char* path, int* kind, int* n_elec, int* n_orb, int* naux);
int pymes_packed_load(pymes_ctx* ctx, const char* path, double* e_core, double* eps_host, double* h_host);
int pymes_packed_write(pymes_ctx* ctx, const char* path, int n_elec, double e_core, const double* eps_host,
                       const double* h_host);
int pymes_packed_write_factors(const char* path, int n_elec, int n_orb, int naux, double e_core, const double* eps_host,
                               const double* h_host, const double* B_host);

/* ---- EOM-CCSD sigma build as ONE entry per step: pymes/solver/eom_ccsd.py:268-385 (update_singles + update_doubles) --------
 * The reference builds sigma = H-bar u term by term from 62 einsums per trial vector.  pymes_eom_sigma_prepare hoists every V.T
 * product that does not depend on the trial vector (once per solve: the (ov)^2 pair matrices, the o v^3 / o^3 v blocks,
 * the one-index dressings; ~10 amplitude-sized arrays held by the handle), pymes_eom_sigma_apply builds sigma for k trial
 * vectors — k >= 2 exchange-symmetric vectors stacked so that every shared operand is read once (the Davidson driver,
 * eom_ccsd.py:95-101), anything else vector by vector, complex trial vectors as two real ones (the operator is real).
 *   f_host      T1-dressed Fock matrix [n,n] (host; eom_ccsd.py:46: t_fock_dressed_pq)
 *   t2_dev      CCSD doubles [v,v,o,o] on the device; must stay alive and unchanged until pymes_eom_sigma_destroy
 *   dressed     != 0: read the context's T1-dressed blocks (pymes_ccsd_dress_V, all of eom_ccsd.py's ten), else the blocks as set
 *   u1_dev[z] [v,o], u2_dev[z] [v,v,o,o] in; s1_dev[z], s2_dev[z] out (device; may not alias the inputs)
 *   sym[z]      != 0: u2[z]_abij = u2[z]_baji is known to hold; sym == NULL: tested (one reduction + synchronisation per vector)
 *   flags       bit 0 V_abcd = V_badc, 1 T_abij = T_baji, 2 pair-packed hole-ladder terms, 3 fused pair kernels, 4 stacked build
 * pymes_eom_diagonals: eom_ccsd.py:169-198 (get_diag_singles) and :200-266 (get_diag_doubles) on the device, d1 [v,o],
 * d2 [v,v,o,o] (needs no handle).  A handle belongs to its context: destroy it before pymes_ctx_destroy. */
typedef struct pymes_eom pymes_eom;
int pymes_eom_sigma_prepare(pymes_ctx* ctx, const double* f_host, const double* t2_dev, int dressed, pymes_eom** out);
int pymes_eom_sigma_flags(pymes_eom* h, int* flags);
int pymes_eom_sigma_apply(pymes_eom* h, int k, const double* const* u1_dev, const double* const* u2_dev, const int* sym,
                          double* const* s1_dev, double* const* s2_dev);
int pymes_eom_diagonals(pymes_ctx* ctx, const double* f_host, const double* t2_dev, int dressed, double* d1_dev, double* d2_dev);
/* releases the engine's pooled scratch that is not in use (the temporaries / hoisted arrays of destroyed EOM handles) */
int pymes_scratch_trim(pymes_ctx* ctx);
int pymes_eom_sigma_destroy(pymes_eom* h);
/* (mr + i mi)[e] = 1 / ((zr + i zi) - (hr + i hi) d[e] + shift), e < n: the FEAST preconditioner 1 / (z - diag + 0.01)
 * (feast_eom_ccsd.py:342; hs = 1j dt for the real-time form :276-278) from the device-resident diagonal */
int pymes_cshift_inv(pymes_ctx* ctx, const double* d_dev, double zr, double zi, double hr, double hi, double shift,
                     double* mr_dev, double* mi_dev, int64_t n);

/* ---- explicit 3-body (transcorrelated) operator ---------------------------------------
 * pymes/util/tcdump.py:52-56: dense fill of L[nb]^6 from (flat index, value) pairs — the host parser hands over
 * unique targets (the last of duplicate entries, as the reference's sequential assignment keeps). */
int pymes_scatter(pymes_ctx* ctx, double* dst_dev, uint64_t dst_elements, const int64_t* index_host,
                  const double* value_host, int64_t n);
/* pymes/integral/contraction.py:17-39 get_single_contraction -> D[p,r,q,s] ([nb]^4 on the device),
 * :41-65 get_double_contraction -> S[p,q], :67-95 get_triple_contraction -> scalar.
 * L_dev is [nb]^6 in the chemists' order (or|ps|qt) that tcdump.read returns. */
int pymes_tc_single_contraction(pymes_ctx* ctx, const double* L_dev, int nb, int no, double* D_dev);
int pymes_tc_double_contraction(pymes_ctx* ctx, const double* L_dev, int nb, int no, double* S_dev);
int pymes_tc_triple_contraction(pymes_ctx* ctx, const double* L_dev, int nb, int no, double* t0_host);

/* ---- measurement --------------------------------------------------------------------- */
/* executed-work counters since the last reset: GEMM launches, executed GEMM flops
 * (2*M*N*K*batch), permutation launches, bytes moved by explicit permutations */
int pymes_stats(pymes_ctx* ctx, int reset, int64_t* gemm_calls, double* gemm_flops, int64_t* permute_calls,
                double* permute_bytes);
/* HIP-event timing of every fp64 GEMM call on the context's stream (off by default).  kernel_class 0 = all
 * calls, 1 = the calls that ran on the LDS-DMA 128x128 MFMA kernel (the o^3v^3 / ladder products);
 * kernel_launches counts GEMM kernel launches (a call whose last tiles are k-split launches twice). */
int pymes_prof_enable(pymes_ctx* ctx, int on);
int pymes_prof_reset(pymes_ctx* ctx);
int pymes_prof_query(pymes_ctx* ctx, int kernel_class, int64_t* calls, int64_t* kernel_launches, double* total_ms,
                     double* flops);

#ifdef __cplusplus
}
#endif
#endif /* PYMES_AMD_H */
