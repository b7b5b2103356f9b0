/* TEST INFRASTRUCTURE: a host program in plain C99 on include/pymes_amd.h alone — no Python, no torch, no C++.
 *
 * What a maintainer of a compiled host (or a cgo / JNI / FFI binding) does with the library: size a context from a packed
 * integral file, load it, run the CCSD / DCSD fixed point of pymes/solver/ccsd.py:159-209 (is_diis = False) pass by pass and
 * print the energies.  Two forms, same numbers:
 *   mode 0  one rank: pymes_ccsd_iterate (residuals, update, energies: one call per pass)
 *   mode 1  the one-process-per-GPU steps with a collective table (pymes_set_collectives) in a world of ONE rank whose
 *           callbacks have nothing to exchange:
 *           pymes_ccsd_sharded_residuals -> pymes_cc_update / pymes_cc_update_pairs -> pymes_ccsd_sharded_finish ->
 *           pymes_ccsd_sharded_energy, the replicated T2 completed by the next residuals call / pymes_ccsd_sharded_await
 *   mode 2  (built with -DWITH_RCCL) the same steps with the table of INTEGRATION.md 2b: ncclAllReduce / ncclAllGather on an
 *           RCCL communicator and a communication stream of the host's own, ordered against the library's stream through
 *           events — the code a multi-GPU host runs unchanged with its rank and world (here: a communicator of one rank,
 *           all a one-GPU test box can hold)
 *   mode 3  mode 2 with the OWNER-TILE exchange of the ring-product rows: the table's optional all-to-all (pymes_set_alltoallv,
 *           ncclSend / ncclRecv in a group), two staging buffers, flag PYMES_OWNER_TILES
 * tests/test_capi_host.py compiles it with gcc (CPU: compiles and links against the library's symbols; GPU: runs it on
 * tests/golden-sized synthetic factors and compares every pass with the Python host and the oracle).
 *
 *   host_ccsd <packed file (kind 2: factors)> <passes> <dcsd 0|1> <mode 0|1|2>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pymes_amd.h"
#ifdef WITH_RCCL
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#endif

#define CHECK(call)                                                                      \
    do {                                                                                 \
        if ((call) != 0) {                                                               \
            fprintf(stderr, "%s failed: %s\n", #call, pymes_last_error());               \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

/* a world of one rank: every collective is complete the moment it is "started" */
static int calls[4];
static int one_rank_allreduce(void* user, double* buf, int64_t n, void* stream, int64_t* ticket) {
    (void)user; (void)buf; (void)n; (void)stream;
    *ticket = ++calls[0];
    return 0;
}
static int one_rank_allgather(void* user, double* buf, int64_t chunk, void* stream, int64_t* ticket) {
    (void)user; (void)buf; (void)chunk; (void)stream;
    *ticket = 1000 + ++calls[1];
    return 0;
}
static int one_rank_wait(void* user, int64_t ticket, void* stream) {
    (void)user; (void)ticket; (void)stream;
    ++calls[2];
    return 0;
}

#ifdef WITH_RCCL
/* The table over RCCL.  A collective is ordered behind the work already enqueued on the library's stream by an event that the
 * communication stream waits for; its completion is another event that `wait` makes the library's stream wait for.  Tickets
 * index a ring of event pairs (at most a dozen tickets are alive at a time). */
#define RING 64
typedef struct {
    ncclComm_t comm;
    hipStream_t cs;
    hipEvent_t before[RING], done[RING];
    int64_t next;
    int rank, world;
} rccl_host;
static int rccl_order(rccl_host* h, void* stream, int64_t* ticket) {
    const int64_t t = h->next++;
    if (hipEventRecord(h->before[t % RING], (hipStream_t)stream) != hipSuccess) return 1;
    if (hipStreamWaitEvent(h->cs, h->before[t % RING], 0) != hipSuccess) return 1;
    *ticket = t;
    return 0;
}
static int rccl_allreduce(void* user, double* buf, int64_t n, void* stream, int64_t* ticket) {
    rccl_host* h = (rccl_host*)user;
    if (rccl_order(h, stream, ticket)) return 1;
    if (ncclAllReduce(buf, buf, (size_t)n, ncclDouble, ncclSum, h->comm, h->cs) != ncclSuccess) return 2;
    ++calls[0];
    return hipEventRecord(h->done[*ticket % RING], h->cs) != hipSuccess;
}
static int rccl_allgather(void* user, double* buf, int64_t chunk, void* stream, int64_t* ticket) {
    rccl_host* h = (rccl_host*)user;
    if (rccl_order(h, stream, ticket)) return 1;
    /* in place: this rank's chunk already sits at its position in the receive buffer */
    if (ncclAllGather(buf + (int64_t)h->rank * chunk, buf, (size_t)chunk, ncclDouble, h->comm, h->cs) != ncclSuccess) return 2;
    ++calls[1];
    return hipEventRecord(h->done[*ticket % RING], h->cs) != hipSuccess;
}
/* the optional all-to-all of the table (pymes_set_alltoallv: the owner tiles of the ring-product rows): counts in doubles, the
 * pieces for / from rank 0, 1, ... contiguous in that order — ncclSend / ncclRecv pairs in one group */
static int rccl_alltoallv(void* user, const double* send, const int64_t* ns, double* recv, const int64_t* nr, void* stream,
                          int64_t* ticket) {
    rccl_host* h = (rccl_host*)user;
    if (rccl_order(h, stream, ticket)) return 1;
    int64_t so = 0, ro = 0;
    if (ncclGroupStart() != ncclSuccess) return 2;
    for (int p = 0; p < h->world; ++p) {
        if (ns[p] > 0 && ncclSend(send + so, (size_t)ns[p], ncclDouble, p, h->comm, h->cs) != ncclSuccess) return 2;
        if (nr[p] > 0 && ncclRecv(recv + ro, (size_t)nr[p], ncclDouble, p, h->comm, h->cs) != ncclSuccess) return 2;
        so += ns[p];
        ro += nr[p];
    }
    if (ncclGroupEnd() != ncclSuccess) return 2;
    ++calls[3];
    return hipEventRecord(h->done[*ticket % RING], h->cs) != hipSuccess;
}
static int rccl_wait(void* user, int64_t ticket, void* stream) {
    rccl_host* h = (rccl_host*)user;
    ++calls[2];
    return hipStreamWaitEvent((hipStream_t)stream, h->done[ticket % RING], 0) != hipSuccess;
}
static int rccl_open(rccl_host* h) {
    int dev = 0;
    memset(h, 0, sizeof *h);
    h->world = 1;
    if (hipSetDevice(0) != hipSuccess) return 1;
    if (ncclCommInitAll(&h->comm, 1, &dev) != ncclSuccess) return 2;       /* a multi-GPU host: ncclCommInitRank(rank, world, id) */
    if (hipStreamCreateWithFlags(&h->cs, hipStreamNonBlocking) != hipSuccess) return 3;
    for (int i = 0; i < RING; ++i)
        if (hipEventCreateWithFlags(&h->before[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->done[i], hipEventDisableTiming) != hipSuccess)
            return 4;
    return 0;
}
static void rccl_close(rccl_host* h) {
    (void)hipStreamSynchronize(h->cs);
    for (int i = 0; i < RING; ++i) {
        (void)hipEventDestroy(h->before[i]);
        (void)hipEventDestroy(h->done[i]);
    }
    (void)hipStreamDestroy(h->cs);
    (void)ncclCommDestroy(h->comm);
}
#endif

static double* dev_doubles(pymes_ctx* ctx, int64_t n) {
    void* p = NULL;
    if (pymes_malloc(ctx, (uint64_t)n * sizeof(double), &p) != 0 || pymes_memset_zero(ctx, p, (uint64_t)n * sizeof(double)) != 0) {
        fprintf(stderr, "device allocation of %lld doubles failed: %s\n", (long long)n, pymes_last_error());
        exit(1);
    }
    return (double*)p;
}

int main(int argc, char** argv) {
    if (argc != 5) {
        fprintf(stderr, "usage: %s <packed factors file> <passes> <dcsd 0|1> <mode 0|1|2|3>\n", argv[0]);
        return 2;
    }
    const char* path = argv[1];
    const int passes = atoi(argv[2]), dcsd = atoi(argv[3]), mode = atoi(argv[4]);
    if (strcmp(pymes_backend(), "hip-gfx950") != 0) {
        fprintf(stderr, "not the product library: backend %s\n", pymes_backend());
        return 1;
    }
    int kind = 0, n_elec = 0, n = 0, naux = 0;
    CHECK(pymes_packed_header(path, &kind, &n_elec, &n, &naux));
    const int no = n_elec / 2, nv = n - no;
    const int64_t o = no, v = nv, nt1 = v * o, nt2 = v * v * o * o;
    pymes_ctx* ctx = NULL;
    CHECK(pymes_ctx_create(&ctx, 0, no, nv, 0));
    double e_core = 0.0;
    double* eps = (double*)malloc(sizeof(double) * (size_t)n);
    double* h = (double*)malloc(sizeof(double) * (size_t)n * (size_t)n);
    double* f = (double*)calloc((size_t)n * (size_t)n, sizeof(double));
    CHECK(pymes_packed_load(ctx, path, &e_core, eps, h));
    for (int p = 0; p < n; ++p) f[(size_t)p * n + p] = eps[p];          /* canonical orbitals: f = diag(eps) (oracle/cases.py) */
    CHECK(pymes_set_orbital_energies(ctx, eps, eps + no));
    double* f_dev = dev_doubles(ctx, (int64_t)n * n);
    CHECK(pymes_upload(ctx, f_dev, f, sizeof(double) * (size_t)n * (size_t)n));
    double *t1 = dev_doubles(ctx, nt1), *t2 = dev_doubles(ctx, nt2), *dt1 = dev_doubles(ctx, nt1);
    double e_mp2[2];
    CHECK(pymes_mp2(ctx, 0.0, t2, e_mp2));                                 /* ccsd.py:128 */
    printf("mp2 %.15e\n", e_mp2[0] + e_mp2[1]);
    uint32_t flags = dcsd ? PYMES_DCD : 0u;
    double out[6];
    if (mode == 0) {
        double* dt2 = dev_doubles(ctx, nt2);
        for (int it = 0; it < passes; ++it) {
            /* the first pass starts from T1 = 0 exactly: the library may take the undressed form (PYMES_T1_ZERO) */
            CHECK(pymes_ccsd_iterate(ctx, f_dev, t1, t2, flags | (it == 0 ? PYMES_T1_ZERO : 0u), 0.0, 1.0, dt1, dt2, out));
            printf("pass %d %.15e %.15e %.15e\n", it + 1, out[0] + out[1] + out[2], out[3], out[4]);
        }
        CHECK(pymes_ccsd_release(ctx));
    } else {
        pymes_collectives table;
        memset(&table, 0, sizeof table);
        table.rank = 0;
        table.world = 1;
        table.allreduce_start = one_rank_allreduce;
        table.allgather_start = one_rank_allgather;
        table.wait = one_rank_wait;
#ifdef WITH_RCCL
        rccl_host rh;
        if (mode >= 2) {
            const int rc = rccl_open(&rh);
            if (rc != 0) {
                fprintf(stderr, "RCCL set-up failed at step %d\n", rc);
                return 1;
            }
            table.user = &rh;
            table.allreduce_start = rccl_allreduce;
            table.allgather_start = rccl_allgather;
            table.wait = rccl_wait;
        }
#else
        if (mode >= 2) {
            fprintf(stderr, "modes 2 and 3 need a build with -DWITH_RCCL\n");
            return 2;
        }
#endif
        CHECK(pymes_set_collectives(ctx, &table));
#ifdef WITH_RCCL
        if (mode == 3) {
            int64_t nsend = 0, nrecv = 0;
            CHECK(pymes_owner_tile_sizes(ctx, table.rank, table.world, &nsend, &nrecv));
            CHECK(pymes_set_alltoallv(ctx, rccl_alltoallv));
            CHECK(pymes_set_owner_tile_buffers(ctx, dev_doubles(ctx, nsend), dev_doubles(ctx, nrecv)));
            flags |= PYMES_OWNER_TILES;
        }
#endif
        int64_t sz[10];
        CHECK(pymes_shard_buffer_sizes(ctx, 1, sz));
        double* b[10];
        for (int i = 0; i < 10; ++i) b[i] = dev_doubles(ctx, sz[i]);
        pymes_shard_buffers bufs = {b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8], b[9]};
        const int64_t npp = v * (v + 1) / 2, nc = npp * 2 * o * o;
        double *fd = dev_doubles(ctx, (int64_t)n * n), *rc = dev_doubles(ctx, nc), *tc = dev_doubles(ctx, nc),
               *dtc = dev_doubles(ctx, nc);
        CHECK(pymes_pairs_pack(ctx, t2, tc, 0, 1));                        /* compact tiles of this rank's (= all) pairs */
        for (int it = 0; it < passes; ++it) {
            CHECK(pymes_ccsd_sharded_residuals(ctx, f_dev, fd, t1, t2, &bufs, flags, rc));
            CHECK(pymes_cc_update(ctx, t1, dt1, bufs.R1, 0.0, 1.0, 2));       /* ccsd.py:176-179 */
            CHECK(pymes_cc_update_pairs(ctx, tc, dtc, rc, 0.0, 1.0, 0, 1));
            int slot = -1;
            CHECK(pymes_ccsd_sharded_finish(ctx, f_dev, t1, tc, dtc, &bufs, &slot));
            CHECK(pymes_ccsd_sharded_energy(ctx, slot, out));
            printf("pass %d %.15e %.15e %.15e\n", it + 1, out[0] + out[1] + out[2], out[3], out[4]);
        }
        CHECK(pymes_ccsd_sharded_await(ctx, t2, &bufs));
        printf("collectives allreduce %d allgather %d wait %d alltoallv %d\n", calls[0], calls[1], calls[2], calls[3]);
        CHECK(pymes_set_collectives(ctx, NULL));
#ifdef WITH_RCCL
        if (mode >= 2) {
            CHECK(pymes_ctx_sync(ctx));
            rccl_close(&rh);
        }
#endif
    }
    /* a checksum of the amplitudes, from the host copy */
    double* t2h = (double*)malloc(sizeof(double) * (size_t)nt2);
    CHECK(pymes_download(ctx, t2h, t2, sizeof(double) * (uint64_t)nt2));
    double s = 0.0;
    for (int64_t i = 0; i < nt2; ++i) s += t2h[i] * t2h[i];
    printf("t2_norm2 %.15e\n", s);
    free(t2h);
    free(eps);
    free(h);
    free(f);
    CHECK(pymes_ctx_destroy(ctx));
    return 0;
}
