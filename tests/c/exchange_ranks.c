/* exchange_ranks.c -- ONE RANK of a multi-rank run of the library's result exchange (include/siftmi.h: siftmi_exchange_*), plain C
 * against the C ABI, one process per rank, as bench.py --gpus N and a Swift / C host of the boundary run it.
 *
 *   exchange_ranks W H n_octaves F rank world steps scenario pipeline synchronous idfile frames.bin out.bin
 *     frames.bin  2 x F x H x W x 4 BGRA8 bytes: this rank's "small" frame set, then its "large" one (more keypoints)
 *     scenario    "plain"            every step the small set
 *                 "jump:K:R"         rank R runs its large set at step K (a > 25 % count jump on ONE rank only)
 *                 "overflow:R"       rank R's context has tiny keypoint lists (list overflow on one rank, every step)
 *                 "die:K:R"          rank R kills itself (SIGKILL) after submitting step K, before its gather: the others must come
 *                                    back with an error inside SIFTMI_EXCHANGE_TIMEOUT_S instead of hanging (exit code 3)
 *     pipeline    steps in flight (1 or 2); result sets = 2 x pipeline
 *     synchronous 1: siftmi_exchange_gather(x, 1) sizes every step from its own totals (host sync); 0: from the previous step
 *     idfile      rank 0 writes the unique id there (atomic rename), the others wait for it
 *     out.bin     per step, read ONE STEP LATE (after the next step's gather): the step's own packed results, then the gathered view
 *                 of every rank's row -- the parent test compares row r of every rank's view with rank r's own results:
 *                   int64 step; int32 nk, nd, flags, n_counts; counts; nk x 44 B; nd x 136 B              (own results)
 *                   int32 world, complete_at_first_look, complete_when_read, 0; int64 kp_records, desc_records
 *                   per rank r: int32 totals[4]; counts (n_counts); min(totals[0], kp_records) x 44 B; min(totals[1], desc_records) x 136 B
 *                 trailer: int64 -1; int64 regathered_steps, overflow_steps, gathers, bytes_last
 * Exit codes: 0 ok, 3 the exchange timed out / was aborted (a peer is gone), 77 the transport refused this layout (real RCCL with
 * two ranks on one GPU), anything else a failure. */
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "siftmi.h"

#define CHECK(expr)                                                                                         \
    do {                                                                                                    \
        int rc_ = (expr);                                                                                   \
        if (rc_ != SIFTMI_OK) {                                                                             \
            fprintf(stderr, "rank %d: %s -> %d: %s\n", g_rank, #expr, rc_, siftmi_last_error());            \
            return (strstr(siftmi_last_error(), "timed out") || strstr(siftmi_last_error(), "aborted")) ? 3 : 2; \
        }                                                                                                   \
    } while (0)

static int g_rank = 0;

static void sleep_ms(int ms) {
    struct timespec ts = {ms / 1000, (long)(ms % 1000) * 1000000L};
    nanosleep(&ts, NULL);
}

static int write_own(FILE *f, const siftmi_step_host *r, int n_counts) {
    int32_t hdr[4] = {r->n_keypoints, r->n_descriptors, r->overflow_flags, n_counts};
    if (fwrite(&r->step, 8, 1, f) != 1 || fwrite(hdr, 4, 4, f) != 4) return 1;
    if (fwrite(r->counts, 4, (size_t)n_counts, f) != (size_t)n_counts) return 1;
    if (r->n_keypoints && fwrite(r->keypoints, sizeof(siftmi_keypoint), (size_t)r->n_keypoints, f) != (size_t)r->n_keypoints) return 1;
    if (r->n_descriptors && fwrite(r->descriptors, sizeof(siftmi_descriptor), (size_t)r->n_descriptors, f) != (size_t)r->n_descriptors) return 1;
    return 0;
}

/* the gathered view `back` steps ago, every rank's row, copied out of HBM through the C ABI */
static int write_gathered(FILE *f, siftmi_exchange *x, int back, int64_t want_step, int first_look, int n_counts) {
    siftmi_gathered g;
    CHECK(siftmi_exchange_result(x, back, &g, SIFTMI_NO_STREAM, 1));
    if (g.step != want_step || !g.resolved) { fprintf(stderr, "rank %d: gathered step %lld (want %lld), resolved %d\n", g_rank, (long long)g.step, (long long)want_step, g.resolved); return 3; }
    int32_t hdr[4] = {g.world, first_look, g.complete, 0};
    int64_t rec[2] = {g.kp_records, g.desc_records};
    if (fwrite(hdr, 4, 4, f) != 4 || fwrite(rec, 8, 2, f) != 2) return 1;
    int32_t *hc = (int32_t *)malloc((size_t)n_counts * 4 * (size_t)g.world);
    CHECK(siftmi_memcpy(hc, g.counts, (size_t)n_counts * 4 * (size_t)g.world, 1));
    for (int r = 0; r < g.world; r++) {
        const int32_t *t = g.totals_host + 4 * r;
        int64_t nk = t[0] < g.kp_records ? t[0] : g.kp_records, nd = t[1] < g.desc_records ? t[1] : g.desc_records;
        size_t bk = (size_t)nk * sizeof(siftmi_keypoint), bd = (size_t)nd * sizeof(siftmi_descriptor);
        void *hk = malloc(bk + 1), *hd = malloc(bd + 1);
        CHECK(siftmi_memcpy(hk, (const unsigned char *)g.keypoints + (size_t)r * (size_t)g.kp_stride, bk, 1));
        CHECK(siftmi_memcpy(hd, (const unsigned char *)g.descriptors + (size_t)r * (size_t)g.desc_stride, bd, 1));
        int bad = fwrite(t, 4, 4, f) != 4 || fwrite(hc + (size_t)r * n_counts, 4, (size_t)n_counts, f) != (size_t)n_counts ||
                  (bk && fwrite(hk, 1, bk, f) != bk) || (bd && fwrite(hd, 1, bd, f) != bd);
        free(hk); free(hd);
        if (bad) return 1;
    }
    free(hc);
    return 0;
}

int main(int argc, char **argv) {
    if (argc != 14) { fprintf(stderr, "usage: %s W H n_octaves F rank world steps scenario pipeline synchronous idfile frames.bin out.bin\n", argv[0]); return 1; }
    const int W = atoi(argv[1]), H = atoi(argv[2]), n_oct = atoi(argv[3]), F = atoi(argv[4]), rank = atoi(argv[5]), world = atoi(argv[6]), steps = atoi(argv[7]);
    const char *scenario = argv[8];
    const int pipeline = atoi(argv[9]), synchronous = atoi(argv[10]);
    const char *idfile = argv[11];
    const size_t set_bytes = (size_t)F * H * W * 4;
    g_rank = rank;
    int jump_step = -1, jump_rank = -1, overflow_rank = -1, die_step = -1, die_rank = -1;
    if (sscanf(scenario, "jump:%d:%d", &jump_step, &jump_rank) == 2) {}
    else if (sscanf(scenario, "die:%d:%d", &die_step, &die_rank) == 2) {}
    else if (sscanf(scenario, "overflow:%d", &overflow_rank) == 1) {}
    else if (strcmp(scenario, "plain") != 0) { fprintf(stderr, "unknown scenario %s\n", scenario); return 1; }

    if (siftmi_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 78; }
    siftmi_config cfg;
    CHECK(siftmi_default_config(&cfg, W, H));
    cfg.n_octaves = n_oct;
    cfg.max_batch = F;
    if (rank == overflow_rank) cfg.max_keypoints = 8;
    siftmi_ctx *ctx = NULL;
    CHECK(siftmi_create(&cfg, 0, &ctx));                                   /* every rank on device 0: the ranks share the one GPU of the box */

    siftmi_stream_config scfg;
    CHECK(siftmi_stream_default_config(&scfg, F));
    scfg.steps_in_flight = pipeline;
    scfg.result_sets = 2 * pipeline;
    scfg.format = SIFTMI_FMT_BGRA8;
    scfg.kp_per_frame = 8192; scfg.desc_per_frame = 12288;                 /* equal on every rank: the payload sizes are clipped to them */
    siftmi_stream *st = NULL;
    CHECK(siftmi_stream_create(ctx, &scfg, &st));

    /* rank 0 makes the unique id; the others read it from the file (what bench.py does over gloo) */
    unsigned char id[SIFTMI_UNIQUE_ID_BYTES];
    if (rank == 0) {
        CHECK(siftmi_exchange_unique_id(id));
        char tmp[4096];
        snprintf(tmp, sizeof(tmp), "%s.tmp", idfile);
        FILE *fi = fopen(tmp, "wb");
        if (!fi || fwrite(id, 1, sizeof(id), fi) != sizeof(id)) { perror(tmp); return 1; }
        fclose(fi);
        if (rename(tmp, idfile) != 0) { perror("rename"); return 1; }
    } else {
        FILE *fi = NULL;
        for (int i = 0; i < 1200 && !(fi = fopen(idfile, "rb")); i++) sleep_ms(100);
        if (!fi || fread(id, 1, sizeof(id), fi) != sizeof(id)) { fprintf(stderr, "rank %d: no unique id in %s\n", rank, idfile); return 1; }
        fclose(fi);
    }
    siftmi_exchange *x = NULL;
    int rc = siftmi_exchange_create(st, id, rank, world, &x);
    if (rc != SIFTMI_OK) {
        fprintf(stderr, "rank %d: siftmi_exchange_create -> %d: %s\n", rank, rc, siftmi_last_error());
        return strstr(siftmi_last_error(), "ncclCommInitRank") ? 77 : 2;   /* the transport refused (real RCCL: duplicate GPU) */
    }
    CHECK(siftmi_exchange_set_headroom(x, 25, 16));
    {
        int32_t cr = -1, cn = -1;
        CHECK(siftmi_exchange_ranks(x, &cn, &cr));
        if (cn != world || cr != rank) { fprintf(stderr, "rank %d: communicator says rank %d of %d\n", rank, cr, cn); return 6; }
    }

    FILE *fin = fopen(argv[12], "rb");
    if (!fin) { perror(argv[12]); return 1; }
    void *d_sets[2] = {NULL, NULL};
    void *h_tmp = malloc(set_bytes);
    for (int j = 0; j < 2; j++) {
        if (fread(h_tmp, 1, set_bytes, fin) != set_bytes) { fprintf(stderr, "short read of %s\n", argv[12]); return 1; }
        CHECK(siftmi_device_alloc(0, set_bytes, &d_sets[j]));
        CHECK(siftmi_memcpy(d_sets[j], h_tmp, set_bytes, 0));
    }
    free(h_tmp);
    fclose(fin);

    FILE *fout = fopen(argv[13], "wb");
    if (!fout) { perror(argv[13]); return 1; }
    const int n_counts = 2 * F * n_oct;
    siftmi_step_host r;
    siftmi_gathered g;
    int64_t step = -1;
    int first_look_prev = 0;
    for (int k = 0; k < steps; k++) {
        const int big = (k == jump_step && rank == jump_rank);
        CHECK(siftmi_stream_submit_device(st, d_sets[big], (size_t)W * 4, (size_t)W * H * 4, SIFTMI_NO_STREAM, &step));
        if (step != k) { fprintf(stderr, "step number %lld, expected %d\n", (long long)step, k); return 4; }
        if (k == die_step && rank == die_rank) {                            /* gone without a word: no teardown, no message to the peers */
            fprintf(stderr, "rank %d: dying at step %d\n", rank, k);
            fflush(stderr);
            raise(SIGKILL);
        }
        CHECK(siftmi_exchange_gather(x, synchronous));
        if (k >= 1) {                                                      /* step k-1, while step k runs: own results, then every rank's row */
            rc = siftmi_stream_result_host(st, 1, &r);
            if (rc != SIFTMI_OK && rc != SIFTMI_E_CAPACITY) { fprintf(stderr, "rank %d: result_host -> %d: %s\n", rank, rc, siftmi_last_error()); return 2; }
            if (r.step != k - 1 || write_own(fout, &r, n_counts)) return 5;
            if ((rc = write_gathered(fout, x, 1, k - 1, first_look_prev, n_counts))) return rc;
        }
        CHECK(siftmi_exchange_result(x, 0, &g, SIFTMI_NO_STREAM, 1));      /* first look at step k: cut short if it outgrew its sizes */
        first_look_prev = g.complete;
    }
    int64_t regathered = -1, overflowed = -1, gathers = 0, bytes = 0;
    double ms = 0.0;
    CHECK(siftmi_exchange_finish(x, &regathered, &overflowed));
    rc = siftmi_stream_result_host(st, 0, &r);
    if (rc != SIFTMI_OK && rc != SIFTMI_E_CAPACITY) { fprintf(stderr, "rank %d: result_host -> %d: %s\n", rank, rc, siftmi_last_error()); return 2; }
    if (r.step != steps - 1 || write_own(fout, &r, n_counts)) return 5;
    if ((rc = write_gathered(fout, x, 0, steps - 1, first_look_prev, n_counts))) return rc;
    CHECK(siftmi_exchange_stats(x, &ms, &gathers, &bytes));
    int64_t trailer[5] = {-1, regathered, overflowed, gathers, bytes};
    if (fwrite(trailer, 8, 5, fout) != 5) return 5;
    fclose(fout);
    CHECK(siftmi_stream_synchronize(st));
    siftmi_exchange_destroy(x);
    siftmi_stream_destroy(st);
    for (int j = 0; j < 2; j++) CHECK(siftmi_device_free(d_sets[j]));
    siftmi_destroy(ctx);
    printf("rank %d ok %d steps, %lld regathered, %lld overflowed, %.3f ms in %lld gathers\n", rank, steps, (long long)regathered, (long long)overflowed, ms, (long long)gathers);
    return 0;
}
