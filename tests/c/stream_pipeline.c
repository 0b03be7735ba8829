/* stream_pipeline.c -- a plain C host of libsiftmi.so's frame stream (include/siftmi.h), no Python, no torch, no HIP headers.
 *
 * What a Swift / C caller of the drop-in boundary does for a stream of frames: one context, a stream with two steps in
 * flight, frames either resident in HBM (siftmi_device_alloc) or fed from page-locked host memory (siftmi_host_alloc), every
 * step's packed results read on the host one step late, and the RCCL result exchange (one rank) gathered every step and
 * compared byte for byte with the step's own results.
 *
 *   stream_pipeline W H n_octaves F n_sets steps mode frames.bin out.bin
 *     frames.bin  n_sets x F x H x W x 4 BGRA8 bytes; step k processes frame set k mod n_sets
 *     mode        "device" | "host" | "host+exchange" | "device+exchange"
 *     out.bin     per step: int64 step, int32 nk, nd, flags, n_counts; counts; nk x 44 keypoint bytes; nd x 136 descriptor bytes
 * tests/test_gpu_parity.py::test_c_host_stream_* runs it and compares out.bin with the per-step results of the host API. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "siftmi.h"

#define CHECK(expr)                                                                                         \
    do {                                                                                                    \
        int rc_ = (expr);                                                                                   \
        if (rc_ != SIFTMI_OK) { fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, siftmi_last_error()); return 2; } \
    } while (0)

static int write_step(FILE *f, const siftmi_step_host *r, int n_counts) {
    int32_t hdr[4] = {r->n_keypoints, r->n_descriptors, r->overflow_flags, n_counts};
    if (fwrite(&r->step, 8, 1, f) != 1 || fwrite(hdr, 4, 4, f) != 4) return 1;
    if (fwrite(r->counts, 4, (size_t)n_counts, f) != (size_t)n_counts) return 1;
    if (r->n_keypoints && fwrite(r->keypoints, sizeof(siftmi_keypoint), (size_t)r->n_keypoints, f) != (size_t)r->n_keypoints) return 1;
    if (r->n_descriptors && fwrite(r->descriptors, sizeof(siftmi_descriptor), (size_t)r->n_descriptors, f) != (size_t)r->n_descriptors) return 1;
    return 0;
}

/* the gathered copy of a step (rank 0's row of a one-rank exchange) must be the step's own packed results */
static int check_gathered(siftmi_exchange *x, int back, const siftmi_step_host *r, int n_counts) {
    siftmi_gathered g;
    CHECK(siftmi_exchange_result(x, back, &g, SIFTMI_NO_STREAM, 1));
    if (g.step != r->step || g.world != 1 || !g.resolved) { fprintf(stderr, "gathered step %lld / world %d / resolved %d\n", (long long)g.step, g.world, g.resolved); return 3; }
    if (g.totals_host[0] != r->n_keypoints || g.totals_host[1] != r->n_descriptors) { fprintf(stderr, "gathered totals differ\n"); return 3; }
    if (!g.complete) return 0;                       /* cut short: the next gather repeats it in full (checked by the caller then) */
    size_t nk = (size_t)r->n_keypoints * sizeof(siftmi_keypoint), nd = (size_t)r->n_descriptors * sizeof(siftmi_descriptor);
    void *hk = malloc(nk + 1), *hd = malloc(nd + 1);
    int32_t *hc = (int32_t *)malloc((size_t)n_counts * 4);
    CHECK(siftmi_memcpy(hk, g.keypoints, nk, 1));
    CHECK(siftmi_memcpy(hd, g.descriptors, nd, 1));
    CHECK(siftmi_memcpy(hc, g.counts, (size_t)n_counts * 4, 1));
    int bad = memcmp(hk, r->keypoints, nk) || memcmp(hd, r->descriptors, nd) || memcmp(hc, r->counts, (size_t)n_counts * 4);
    free(hk); free(hd); free(hc);
    if (bad) { fprintf(stderr, "gathered bytes of step %lld differ from the step's results\n", (long long)r->step); return 3; }
    return 0;
}

int main(int argc, char **argv) {
    if (argc != 10) { fprintf(stderr, "usage: %s W H n_octaves F n_sets steps mode frames.bin out.bin\n", argv[0]); return 1; }
    const int W = atoi(argv[1]), H = atoi(argv[2]), n_oct = atoi(argv[3]), F = atoi(argv[4]), n_sets = atoi(argv[5]), steps = atoi(argv[6]);
    const char *mode = argv[7];
    const int host_fed = strncmp(mode, "host", 4) == 0, exchange = strstr(mode, "+exchange") != NULL;
    const size_t set_bytes = (size_t)F * H * W * 4;

    if (siftmi_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 77; }
    siftmi_config cfg;
    CHECK(siftmi_default_config(&cfg, W, H));
    cfg.n_octaves = n_oct;
    cfg.max_batch = F;
    siftmi_ctx *ctx = NULL;
    CHECK(siftmi_create(&cfg, 0, &ctx));

    siftmi_stream_config scfg;
    CHECK(siftmi_stream_default_config(&scfg, F));
    scfg.steps_in_flight = 2;
    scfg.result_sets = 4;
    scfg.format = SIFTMI_FMT_BGRA8;
    scfg.kp_per_frame = 8192; scfg.desc_per_frame = 12288;
    siftmi_stream *st = NULL;
    CHECK(siftmi_stream_create(ctx, &scfg, &st));
    if (siftmi_stream_context(st, 0) != ctx || siftmi_stream_context(st, 1) == NULL || siftmi_stream_context(st, 2) != NULL) return 4;

    siftmi_exchange *x = NULL;
    if (exchange) {
        unsigned char id[SIFTMI_UNIQUE_ID_BYTES];
        CHECK(siftmi_exchange_unique_id(id));
        CHECK(siftmi_exchange_create(st, id, 0, 1, &x));
    }

    /* frames: page-locked host copies (host-fed) or HBM-resident copies */
    FILE *fin = fopen(argv[8], "rb");
    if (!fin) { perror(argv[8]); return 1; }
    void **h_sets = (void **)calloc((size_t)n_sets, sizeof(void *)), **d_sets = (void **)calloc((size_t)n_sets, sizeof(void *));
    for (int j = 0; j < n_sets; j++) {
        CHECK(siftmi_host_alloc(set_bytes, &h_sets[j]));
        if (fread(h_sets[j], 1, set_bytes, fin) != set_bytes) { fprintf(stderr, "short read of %s\n", argv[8]); return 1; }
        if (!host_fed) {
            CHECK(siftmi_device_alloc(0, set_bytes, &d_sets[j]));
            CHECK(siftmi_memcpy(d_sets[j], h_sets[j], set_bytes, 0));
        }
    }
    fclose(fin);
    /* host-fed: ONE page-locked buffer refilled for every step, as a capture loop would -- legal once wait_upload returns */
    void *h_live = NULL;
    if (host_fed) CHECK(siftmi_host_alloc(set_bytes, &h_live));

    FILE *fout = fopen(argv[9], "wb");
    if (!fout) { perror(argv[9]); return 1; }
    const int n_counts = 2 * F * n_oct;
    siftmi_step_host r;
    int64_t step = -1;
    int rc;
    for (int k = 0; k < steps; k++) {
        if (host_fed) {
            if (k > 0) CHECK(siftmi_stream_wait_upload(st, step));         /* the previous step's frames have left h_live */
            memcpy(h_live, h_sets[k % n_sets], set_bytes);
            CHECK(siftmi_stream_submit_host(st, h_live, (size_t)W * 4, (size_t)W * H * 4, &step));
        } else {
            CHECK(siftmi_stream_submit_device(st, d_sets[k % n_sets], (size_t)W * 4, (size_t)W * H * 4, SIFTMI_NO_STREAM, &step));
        }
        if (step != k) { fprintf(stderr, "step number %lld, expected %d\n", (long long)step, k); return 4; }
        if (x) CHECK(siftmi_exchange_gather(x, 0));
        if (k >= 1) {                                                      /* read step k-1 while step k runs */
            CHECK(siftmi_stream_result_host(st, 1, &r));
            if (r.step != k - 1 || write_step(fout, &r, n_counts)) return 5;
            if (x && (rc = check_gathered(x, 1, &r, n_counts))) return rc;  /* after gather(k) step k-1 is complete whatever its size */
        }
    }
    CHECK(siftmi_stream_result_host(st, 0, &r));
    if (r.step != steps - 1 || write_step(fout, &r, n_counts)) return 5;
    if (x) {
        int64_t regathered = -1, overflowed = -1;
        CHECK(siftmi_exchange_finish(x, &regathered, &overflowed));
        if ((rc = check_gathered(x, 0, &r, n_counts))) return rc;
        siftmi_gathered g;
        CHECK(siftmi_exchange_result(x, 0, &g, SIFTMI_NO_STREAM, 1));
        if (!g.complete) { fprintf(stderr, "last step still incomplete after finish\n"); return 3; }
        double ms; int64_t n, bytes;
        CHECK(siftmi_exchange_stats(x, &ms, &n, &bytes));
        fprintf(stderr, "exchange: %lld gathers, %.3f ms total, %lld bytes in the last, %lld steps gathered twice, %lld overflowed\n",
                (long long)n, ms, (long long)bytes, (long long)regathered, (long long)overflowed);
        if (overflowed != 0) return 3;
        printf("regathered %lld\n", (long long)regathered);
    }
    fclose(fout);
    CHECK(siftmi_stream_synchronize(st));
    if (x) siftmi_exchange_destroy(x);
    siftmi_stream_destroy(st);
    for (int j = 0; j < n_sets; j++) { CHECK(siftmi_host_free(h_sets[j])); if (d_sets[j]) CHECK(siftmi_device_free(d_sets[j])); }
    if (h_live) CHECK(siftmi_host_free(h_live));
    siftmi_destroy(ctx);
    printf("ok %d steps\n", steps);
    return 0;
}
