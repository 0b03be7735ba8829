/* fake_rccl.c -- a TEST transport with librccl's entry points, for running the library's result exchange
 * (siftmetal_amd/csrc/exchange_api.hip.h: siftmi_exchange_*) with several ranks on ONE GPU.
 *
 * libsiftmi.so resolves eleven RCCL symbols from the library named by SIFTMI_RCCL_LIB.  This file implements exactly those over
 * POSIX shared memory between processes that share a device (real RCCL refuses two ranks on one GPU), with the stream
 * semantics the exchange relies on: every call is enqueued on the caller's stream and returns at once; the data moves when the
 * stream gets there; ranks meet per collective.  It is test infrastructure (tests/test_gpu_parity.py::test_exchange_*_ranks_*,
 * bench.py --gpus N --share-gpu) and ships in no product path: the exchange inside libsiftmi.so is unchanged and cannot tell.
 *
 * An all-gather of n bytes per rank is cut into chunks of at most CHUNK bytes.  Per chunk, on the stream:
 *   1. hipMemcpyAsync  send chunk -> page-locked bounce buffer                              (device to host)
 *   2. hipLaunchHostFunc: wait until every rank has consumed chunk seq - DEPTH (the slot is free), copy the bounce buffer into this
 *      rank's shared slot, publish seq, wait until every rank has published seq, copy all ranks' slots into the second bounce
 *      buffer, mark seq consumed
 *   3. hipMemcpyAsync  bounce buffer row r -> recv + r * n + offset, for every rank r     (host to device)
 * All waits time out (FAKE_RCCL_TIMEOUT_S, default 120 s) and poison the communicator instead of hanging a GPU box.
 * Like RCCL, a collective whose peer never arrives does not complete by itself: the library's own deadline (exchange_api.hip.h,
 * SIFTMI_EXCHANGE_TIMEOUT_S) calls ncclCommAbort, which here makes this rank's pending host functions return at once
 * (tests/test_exchange_ranks.py::test_exchange_rank_dies_and_the_others_abort_within_the_deadline).
 * ncclGroupStart / ncclGroupEnd are accepted and ignored: every rank issues the same calls in the same order (the contract of a
 * real group too), so running them one after the other is an allowed schedule.  One stream at a time per communicator is ordered
 * by an event when the stream changes. */
#define _GNU_SOURCE
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <errno.h>
#include <fcntl.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

/* the slice of rccl.h this transport implements (values as in RCCL 2.x) */
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5,
               ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8, ncclBfloat16 = 9 } ncclDataType_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;

#define MAX_RANKS 16
#define DEPTH 2
#define CHUNK ((size_t)8 << 20)
#define MAGIC 0x46524343u /* "FRCC" */

typedef struct {
    _Atomic int32_t joined, left, poisoned;
    int32_t world;
    _Atomic uint64_t published[MAX_RANKS], consumed[MAX_RANKS];
    _Atomic uint64_t calls[MAX_RANKS], bytes[MAX_RANKS];
} Header;

struct fakeComm {
    Header *hdr;
    unsigned char *slots;        /* [DEPTH][world][CHUNK] */
    size_t map_bytes;
    int rank, world;
    unsigned char *bounce_out;   /* page-locked, CHUNK */
    unsigned char *bounce_in;    /* page-locked, world * CHUNK */
    uint64_t seq;                /* chunks enqueued so far */
    hipStream_t last_stream;
    hipEvent_t last_ev;
    int have_last;
    double timeout_s;
    _Atomic int32_t aborted;     /* ncclCommAbort on THIS rank: its pending host functions return at once */
};
typedef struct fakeComm *ncclComm_t;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static double timeout_from_env(void) {
    const char *e = getenv("FAKE_RCCL_TIMEOUT_S");
    double t = e ? atof(e) : 0.0;
    return t > 0.0 ? t : 120.0;
}

static unsigned char *slot_of(struct fakeComm *c, uint64_t seq, int rank) {
    return c->slots + ((size_t)(seq % DEPTH) * (size_t)c->world + (size_t)rank) * CHUNK;
}

/* spin until every rank's counter in `arr` has reached `want`; 0 on success */
static int wait_all(struct fakeComm *c, _Atomic uint64_t *arr, uint64_t want, const char *what) {
    const double t0 = now_s();
    unsigned spins = 0;
    for (;;) {
        int ok = 1;
        for (int r = 0; r < c->world; r++)
            if (atomic_load_explicit(&arr[r], memory_order_acquire) < want) { ok = 0; break; }
        if (ok) return 0;
        if (atomic_load_explicit(&c->hdr->poisoned, memory_order_acquire)) return 1;
        if (atomic_load_explicit(&c->aborted, memory_order_acquire)) return 1;
        if ((++spins & 1023u) == 0) {
            if (now_s() - t0 > c->timeout_s) {
                fprintf(stderr, "fake_rccl: rank %d timed out after %.0f s waiting for %s %llu\n", c->rank, c->timeout_s, what, (unsigned long long)want);
                atomic_store_explicit(&c->hdr->poisoned, 1, memory_order_release);
                return 1;
            }
            struct timespec ts = {0, 20000};
            nanosleep(&ts, NULL);
        }
    }
}

typedef struct { struct fakeComm *c; uint64_t seq; size_t n; } ChunkOp;

static void chunk_host_fn(void *p) {
    ChunkOp *op = (ChunkOp *)p;
    struct fakeComm *c = op->c;
    Header *h = c->hdr;
    if (op->seq > DEPTH && wait_all(c, h->consumed, op->seq - DEPTH, "slot release of chunk")) goto out;
    memcpy(slot_of(c, op->seq, c->rank), c->bounce_out, op->n);
    atomic_store_explicit(&h->published[c->rank], op->seq, memory_order_release);
    if (wait_all(c, h->published, op->seq, "publication of chunk")) goto out;
    for (int r = 0; r < c->world; r++) memcpy(c->bounce_in + (size_t)r * CHUNK, slot_of(c, op->seq, r), op->n);
    atomic_store_explicit(&h->consumed[c->rank], op->seq, memory_order_release);
out:
    free(op);
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled HIP error (fake_rccl)";
    case ncclSystemError: return "system error (fake_rccl: shared memory / timeout)";
    case ncclInternalError: return "internal error (fake_rccl)";
    case ncclInvalidArgument: return "invalid argument (fake_rccl)";
    case ncclInvalidUsage: return "invalid usage (fake_rccl)";
    case ncclRemoteError: return "remote error (fake_rccl)";
    case ncclInProgress: return "in progress (fake_rccl)";
    }
    return "unknown result (fake_rccl)";
}

ncclResult_t ncclGetVersion(int *v) {
    if (!v) return ncclInvalidArgument;
    *v = 0;                      /* not an RCCL */
    return ncclSuccess;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    unsigned rnd = 0;
    int fd = open("/dev/urandom", O_RDONLY);
    if (fd >= 0) { if (read(fd, &rnd, sizeof(rnd)) != (ssize_t)sizeof(rnd)) rnd = 0; close(fd); }
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    uint32_t magic = MAGIC;
    memcpy(id->internal, &magic, 4);
    snprintf(id->internal + 4, NCCL_UNIQUE_ID_BYTES - 4, "/fake_rccl_%d_%lx_%08x", (int)getpid(), (unsigned long)ts.tv_nsec, rnd);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    if (c->have_last) (void)hipStreamSynchronize(c->last_stream);
    if (c->last_ev) (void)hipEventDestroy(c->last_ev);
    if (c->bounce_out) (void)hipHostFree(c->bounce_out);
    if (c->bounce_in) (void)hipHostFree(c->bounce_in);
    if (c->hdr) {
        atomic_fetch_add(&c->hdr->left, 1);
        munmap((void *)c->hdr, c->map_bytes);
    }
    free(c);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank) {
    if (!out) return ncclInvalidArgument;
    *out = NULL;
    uint32_t magic = 0;
    memcpy(&magic, id.internal, 4);
    if (magic != MAGIC || id.internal[NCCL_UNIQUE_ID_BYTES - 1] != 0) return ncclInvalidArgument;
    if (world < 1 || world > MAX_RANKS || rank < 0 || rank >= world) return ncclInvalidArgument;
    const char *name = id.internal + 4;
    struct fakeComm *c = (struct fakeComm *)calloc(1, sizeof(*c));
    if (!c) return ncclSystemError;
    c->rank = rank; c->world = world; c->timeout_s = timeout_from_env();
    const size_t hdr_bytes = (sizeof(Header) + 4095) & ~(size_t)4095;
    c->map_bytes = hdr_bytes + (size_t)DEPTH * (size_t)world * CHUNK;
    int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { fprintf(stderr, "fake_rccl: shm_open(%s): %s\n", name, strerror(errno)); free(c); return ncclSystemError; }
    /* every rank sizes the object before mapping it (a fresh object is zero-filled: the all-zero header is the initial state) */
    if (ftruncate(fd, (off_t)c->map_bytes) != 0) { fprintf(stderr, "fake_rccl: ftruncate: %s\n", strerror(errno)); close(fd); free(c); return ncclSystemError; }
    void *base = mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (base == MAP_FAILED) { fprintf(stderr, "fake_rccl: mmap: %s\n", strerror(errno)); free(c); return ncclSystemError; }
    c->hdr = (Header *)base;
    c->slots = (unsigned char *)base + hdr_bytes;
    if (hipHostMalloc((void **)&c->bounce_out, CHUNK, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&c->bounce_in, (size_t)world * CHUNK, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&c->last_ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        ncclCommDestroy(c);
        shm_unlink(name);
        return ncclUnhandledCudaError;
    }
    /* rendezvous: every rank joins; the last one to see everybody removes the name (the mappings keep the object alive) */
    atomic_fetch_add(&c->hdr->joined, 1);
    const double t0 = now_s();
    while (atomic_load(&c->hdr->joined) < world) {
        if (now_s() - t0 > c->timeout_s) {
            fprintf(stderr, "fake_rccl: rank %d of %d: only %d ranks joined within %.0f s\n", rank, world, (int)atomic_load(&c->hdr->joined), c->timeout_s);
            ncclCommDestroy(c);
            shm_unlink(name);
            return ncclSystemError;
        }
        struct timespec ts = {0, 200000};
        nanosleep(&ts, NULL);
    }
    if (rank == 0) shm_unlink(name);
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int *count) {
    if (!c || !count) return ncclInvalidArgument;
    *count = c->world;
    return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t c, int *rank) {
    if (!c || !rank) return ncclInvalidArgument;
    *rank = c->rank;
    return ncclSuccess;
}

/* a wait that timed out inside the transport (any rank) is this communicator's asynchronous error */
ncclResult_t ncclCommGetAsyncError(ncclComm_t c, ncclResult_t *async_error) {
    if (!c || !async_error) return ncclInvalidArgument;
    *async_error = atomic_load(&c->hdr->poisoned) ? ncclSystemError : ncclSuccess;
    return ncclSuccess;
}

/* Stuck collectives give up (this rank's host functions stop waiting, the stream drains with garbage in the receive buffers), then
 * the communicator is freed.  The shared header is left alone: the other ranks find out by their own deadlines, as with RCCL. */
ncclResult_t ncclCommAbort(ncclComm_t c) {
    if (!c) return ncclSuccess;
    atomic_store_explicit(&c->aborted, 1, memory_order_release);
    return ncclCommDestroy(c);
}

ncclResult_t ncclGroupStart(void) { return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) { return ncclSuccess; }

static size_t dtype_bytes(ncclDataType_t t) {
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    }
    return 0;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dtype, ncclComm_t c, hipStream_t stream) {
    if (!c || !send || !recv) return ncclInvalidArgument;
    const size_t es = dtype_bytes(dtype);
    if (!es) return ncclInvalidArgument;
    if (atomic_load(&c->hdr->poisoned) || atomic_load(&c->aborted)) return ncclSystemError;
    const size_t n = count * es;
    if (c->have_last && c->last_stream != stream) {            /* the bounce buffers are shared: order the streams */
        if (hipStreamWaitEvent(stream, c->last_ev, 0) != hipSuccess) return ncclUnhandledCudaError;
    }
    atomic_fetch_add(&c->hdr->calls[c->rank], 1);
    atomic_fetch_add(&c->hdr->bytes[c->rank], n);
    for (size_t off = 0; off < n; off += CHUNK) {
        const size_t m = n - off < CHUNK ? n - off : CHUNK;
        ChunkOp *op = (ChunkOp *)malloc(sizeof(*op));
        if (!op) return ncclSystemError;
        op->c = c; op->seq = ++c->seq; op->n = m;
        if (hipMemcpyAsync(c->bounce_out, (const unsigned char *)send + off, m, hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipLaunchHostFunc(stream, chunk_host_fn, op) != hipSuccess) {
            (void)hipGetLastError();
            atomic_store(&c->hdr->poisoned, 1);
            return ncclUnhandledCudaError;
        }
        for (int r = 0; r < c->world; r++)
            if (hipMemcpyAsync((unsigned char *)recv + (size_t)r * n + off, c->bounce_in + (size_t)r * CHUNK, m, hipMemcpyHostToDevice, stream) != hipSuccess) {
                (void)hipGetLastError();
                atomic_store(&c->hdr->poisoned, 1);
                return ncclUnhandledCudaError;
            }
    }
    if (hipEventRecord(c->last_ev, stream) != hipSuccess) return ncclUnhandledCudaError;
    c->last_stream = stream; c->have_last = 1;
    return ncclSuccess;
}

/* test introspection: collectives and payload bytes this rank has enqueued, and whether a wait ever timed out */
ncclResult_t fakeRcclStats(ncclComm_t c, uint64_t *calls, uint64_t *bytes, int *poisoned) {
    if (!c) return ncclInvalidArgument;
    if (calls) *calls = atomic_load(&c->hdr->calls[c->rank]);
    if (bytes) *bytes = atomic_load(&c->hdr->bytes[c->rank]);
    if (poisoned) *poisoned = atomic_load(&c->hdr->poisoned);
    return ncclSuccess;
}
