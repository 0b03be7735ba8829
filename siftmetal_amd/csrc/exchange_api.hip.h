// exchange_api.hip.h -- siftmi_exchange_*: the RCCL all-gather of every rank's packed results (include/siftmi.h), the one
// exchange step of the path (frames are sharded frame-per-GPU; the reference keeps no cross-frame state, SIFT/SIFT.swift holds
// only scratch).  Included by siftmi_api.hip after stream_api.hip.h.
//
// xGMI on MI355X is a fully connected point-to-point mesh (7 links per GPU): an all-gather is per-link bound and each rank
// pushes its shard to 7 peers at once.  So per step there is ONE small all-gather (16 B of totals per rank) and ONE group of
// three padded payload all-gathers on a side stream, reading the step's result set while the next step's kernels write
// another set; payload sizes come from the previous step's totals, so the host never waits between kernels and collectives.
//
// librccl is loaded with dlopen at the first siftmi_exchange_* call: the library is 570 MB, single-GPU users never touch it,
// and inside a process that already carries an RCCL (PyTorch bundles one) that copy is used instead of loading a second.
//
// Failure path (round 5).  A collective completes only if every rank takes part; a rank that died or hangs would leave the others
// waiting inside hipEventSynchronize for ever (an 8-GPU job then ends when its lease does, with no output).  So no host wait in this
// file is unbounded: exchange_wait() polls the event (hipEventQuery) and the communicator (ncclCommGetAsyncError) until a deadline
// (SIFTMI_EXCHANGE_TIMEOUT_S, default 120 s per wait; siftmi_exchange_set_timeout), then aborts the communicator (ncclCommAbort makes
// the collective kernels that are stuck on the device exit, so the streams ordered behind them drain) and returns SIFTMI_E_HIP naming
// the rank and the step.  An aborted exchange fails every later call the same way; nothing is restarted.
// siftmi_exchange_destroy is bounded too: if an aborted exchange's stream has not drained by the deadline, its buffers, stream and
// communicator are leaked instead of freed (hipFree / ncclCommDestroy would wait for the stuck stream).
#pragma once
#include <rccl/rccl.h>
#include <chrono>
#include <thread>

struct RcclApi {
    void *handle = nullptr;
    std::string origin;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    bool ok = false;
    RcclApi() {
        const char *env = getenv("SIFTMI_RCCL_LIB");
        if (env && *env) { handle = dlopen(env, RTLD_NOW | RTLD_LOCAL); origin = env; }
        // an RCCL that is already mapped (e.g. the one PyTorch ships as "librccl.so") before a second copy
        const char *loaded[] = {"librccl.so", "librccl.so.1"};
        for (const char *n : loaded) if (!handle) { handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (handle) origin = std::string(n) + " (already loaded)"; }
        const char *fresh[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : fresh) if (!handle) { handle = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (handle) origin = n; }
        if (!handle) return;
        GetUniqueId = (decltype(GetUniqueId))dlsym(handle, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(handle, "ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
        AllGather = (decltype(AllGather))dlsym(handle, "ncclAllGather");
        GroupStart = (decltype(GroupStart))dlsym(handle, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(handle, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
        CommCount = (decltype(CommCount))dlsym(handle, "ncclCommCount");
        CommUserRank = (decltype(CommUserRank))dlsym(handle, "ncclCommUserRank");
        CommGetAsyncError = (decltype(CommGetAsyncError))dlsym(handle, "ncclCommGetAsyncError");
        CommAbort = (decltype(CommAbort))dlsym(handle, "ncclCommAbort");
        ok = GetUniqueId && CommInitRank && CommDestroy && AllGather && GroupStart && GroupEnd && GetErrorString && CommCount && CommUserRank &&
             CommGetAsyncError && CommAbort;
    }
};
static const RcclApi &rccl() { static RcclApi api; return api; }
static int rccl_ready() {
    if (!rccl().ok) {
        const char *why = dlerror();
        return set_error(SIFTMI_E_HIP, "librccl could not be loaded (%s): the result exchange needs RCCL", why ? why : "symbols missing");
    }
    return SIFTMI_OK;
}
#define RCCL_TRY(expr)                                                                                      \
    do {                                                                                                    \
        ncclResult_t r_ = (expr);                                                                           \
        if (r_ != ncclSuccess) return set_error(SIFTMI_E_HIP, "%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

static_assert(SIFTMI_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

extern "C" const char *siftmi_exchange_transport(void) { return rccl().ok ? rccl().origin.c_str() : ""; }

// ------------------------------------------------------------------------------------------------
// the sizing rule (host arithmetic only)
extern "C" int siftmi_gather_plan_init(siftmi_gather_plan *p, int64_t kp_capacity, int64_t desc_capacity) {
    if (!p || kp_capacity < 1 || desc_capacity < 1) return set_error(SIFTMI_E_BADARG, "bad argument");
    memset(p, 0, sizeof(*p));
    p->kp_capacity = kp_capacity; p->desc_capacity = desc_capacity;
    p->send_kp = p->send_desc = -1;
    p->quantum = 1024;
    p->headroom_percent = 25;
    return SIFTMI_OK;
}

static int64_t plan_round(const siftmi_gather_plan *p, int64_t n, int64_t cap) {
    const int64_t q = p->quantum > 0 ? p->quantum : 1;
    int64_t v = n + n * (int64_t)p->headroom_percent / 100 + 1;
    v = (v + q - 1) / q * q;
    return std::max<int64_t>(1, std::min(v, cap));
}

extern "C" int siftmi_gather_plan_resolve(siftmi_gather_plan *p, const int32_t *totals, int world, int64_t sent_kp, int64_t sent_desc) {
    if (!p || !totals || world < 1) return set_error(SIFTMI_E_BADARG, "bad argument");
    int64_t mk = 0, md = 0;
    bool overflow = false;
    for (int r = 0; r < world; r++) {
        mk = std::max<int64_t>(mk, totals[4 * r + 0]);
        md = std::max<int64_t>(md, totals[4 * r + 1]);
        overflow = overflow || totals[4 * r + 2] != 0;
    }
    const bool incomplete = mk > sent_kp || md > sent_desc;
    p->steps_resolved++;
    if (incomplete) p->steps_incomplete++;
    if (overflow) p->steps_overflowed++;
    p->send_kp = plan_round(p, std::max<int64_t>(mk, 1), p->kp_capacity);
    p->send_desc = plan_round(p, std::max<int64_t>(md, 1), p->desc_capacity);
    return incomplete ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
struct GatherSet {
    uint8_t *kp = nullptr, *desc = nullptr;
    size_t kp_bytes = 0, desc_bytes = 0;          // allocated
    // a block that had to grow is kept until this set is recycled the next time: a consumer stream handed the old pointers by
    // siftmi_exchange_result(consumer_stream, wait_host = 0) may still be reading them (the validity window siftmi.h states)
    std::vector<void *> retired;
    int32_t *counts = nullptr, *totals = nullptr; // device [world][2][F][n_oct], [world][4]
    int32_t *h_totals = nullptr;                  // pinned [world][4]
    hipEvent_t ev_done = nullptr, ev_totals = nullptr, t0 = nullptr, t1 = nullptr;
    int64_t step = -1, sent_kp = 0, sent_desc = 0;
    bool resolved = false, complete = false, needs_regather = false, timed = false;
};

struct siftmi_exchange {
    siftmi_stream *s = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    int comm_ranks = 0, comm_rank = -1;           // what the communicator itself reports (ncclCommCount / ncclCommUserRank)
    double timeout_s = 120.0;                     // deadline of every host wait (SIFTMI_EXCHANGE_TIMEOUT_S)
    bool failed = false;                          // a wait expired or the communicator reported an error: aborted, every call fails
    bool abort_on_destroy = false;                // creation was refused after ncclCommInitRank: the communicator is aborted, not destroyed
    std::string fail_msg;
    hipEvent_t ev_wait = nullptr;                 // siftmi_exchange_wait / finish / destroy: "everything enqueued so far"
    hipStream_t gstream = nullptr;
    GatherSet g[2];
    int cur = 1;                                  // index of the last gather's set
    siftmi_gather_plan plan;
    int64_t regathered = 0, gathers = 0, bytes_last = 0;
    double ms = 0.0;
};

// the exchange is unusable from here on: stuck collectives are made to exit, later calls repeat the message
static int exchange_fail(siftmi_exchange *x, const char *fmt, ...) {
    char buf[400];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (!x->failed) {
        x->failed = true;
        x->fail_msg = buf;
        if (x->comm) { (void)rccl().CommAbort(x->comm); x->comm = nullptr; }     // (frees the communicator: no ncclCommDestroy after it)
    }
    return set_error(SIFTMI_E_HIP, "%s", x->fail_msg.c_str());
}
static int exchange_check(siftmi_exchange *x) {
    if (!x) return set_error(SIFTMI_E_BADARG, "null exchange");
    if (x->failed) return set_error(SIFTMI_E_HIP, "exchange aborted earlier: %s", x->fail_msg.c_str());
    return SIFTMI_OK;
}
// Bounded host wait for `ev` (an event recorded on the gather stream, or one a gather-stream operation feeds).
static int exchange_wait(siftmi_exchange *x, hipEvent_t ev, const char *what, int64_t step) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned polls = 0;
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return SIFTMI_OK;
        if (e != hipErrorNotReady) {
            (void)hipGetLastError();
            return exchange_fail(x, "rank %d of %d: hipEventQuery failed while waiting for %s of step %lld: %s", x->rank, x->world, what, (long long)step, hipGetErrorString(e));
        }
        (void)hipGetLastError();
        if ((++polls & 63u) == 0 && x->comm) {
            ncclResult_t async = ncclSuccess;
            const ncclResult_t r = rccl().CommGetAsyncError(x->comm, &async);
            if (r != ncclSuccess || (async != ncclSuccess && async != ncclInProgress))
                return exchange_fail(x, "rank %d of %d: the communicator reported an asynchronous error while waiting for %s of step %lld: %s", x->rank, x->world, what,
                                     (long long)step, rccl().GetErrorString(r != ncclSuccess ? r : async));
        }
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (waited > x->timeout_s)
            return exchange_fail(x, "rank %d of %d: timed out after %.1f s waiting for %s of step %lld (a rank died or hangs; SIFTMI_EXCHANGE_TIMEOUT_S); communicator aborted",
                                 x->rank, x->world, waited, what, (long long)step);
        if (polls < 200) std::this_thread::yield();                      // the usual wait is a few microseconds
        else std::this_thread::sleep_for(std::chrono::microseconds(polls < 2000 ? 20 : 500));
    }
}
// everything enqueued on the gather stream so far
static int exchange_drain(siftmi_exchange *x, const char *what) {
    HIP_TRY(hipEventRecord(x->ev_wait, x->gstream));
    return exchange_wait(x, x->ev_wait, what, x->s ? x->s->step_no : -1);
}

extern "C" int siftmi_exchange_unique_id(void *id) {
    if (!id) return set_error(SIFTMI_E_BADARG, "null argument");
    int rc = rccl_ready();
    if (rc) return rc;
    ncclUniqueId u;
    RCCL_TRY(rccl().GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return SIFTMI_OK;
}

extern "C" void siftmi_exchange_destroy(siftmi_exchange *x) {
    if (!x) return;
    (void)hipSetDevice(x->s->device);
    // a gather that can no longer complete (a peer is gone) must not hang the teardown: bounded, then aborted
    if (x->gstream && x->ev_wait && !x->failed) (void)exchange_drain(x, "the pending gathers at destroy");
    bool drained = true;
    if (x->gstream && x->failed) {                                                // aborted collectives exit; bounded all the same
        const auto t0 = std::chrono::steady_clock::now();
        while (hipStreamQuery(x->gstream) == hipErrorNotReady &&
               std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < x->timeout_s)
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        drained = hipStreamQuery(x->gstream) != hipErrorNotReady;
        (void)hipGetLastError();
    }
    // the result sets must not wait for events of this exchange any more
    for (auto &rs : x->s->sets) rs.gather_rec = false;
    if (!drained) {
        // The abort did not make the stuck collective leave the stream within the deadline (or a peer died between the drain and here).
        // hipFree / hipStreamDestroy / ncclCommDestroy would each wait for that stream -- for ever.  The teardown of a failed exchange
        // stays bounded instead: buffers, stream and communicator are LEAKED (the process is about to end: bench.py exits
        // non-zero on a failed exchange) and the caller is told.
        set_error(SIFTMI_E_HIP, "exchange (rank %d of %d): the gather stream did not drain within %.0f s after the abort; its buffers and communicator are leaked",
                  x->rank, x->world, x->timeout_s);
        delete x;
        return;
    }
    // a communicator that failed (or whose creation was refused below) is aborted, not destroyed: ncclCommDestroy waits for the peers
    if (x->comm) (void)((x->failed || x->abort_on_destroy) ? rccl().CommAbort(x->comm) : rccl().CommDestroy(x->comm));
    if (x->ev_wait) (void)hipEventDestroy(x->ev_wait);
    for (auto &g : x->g) {
        void *ptrs[] = {g.kp, g.desc, g.counts, g.totals};
        for (void *p : ptrs) if (p) (void)hipFree(p);
        for (void *p : g.retired) (void)hipFree(p);
        if (g.h_totals) (void)hipHostFree(g.h_totals);
        hipEvent_t evs[] = {g.ev_done, g.ev_totals, g.t0, g.t1};
        for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    }
    if (x->gstream) (void)hipStreamDestroy(x->gstream);
    delete x;
}

extern "C" int siftmi_exchange_create(siftmi_stream *s, const void *unique_id, int rank, int world, siftmi_exchange **out) {
    if (!s || !unique_id || !out) return set_error(SIFTMI_E_BADARG, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return set_error(SIFTMI_E_BADARG, "rank %d / world %d invalid", rank, world);
    int rc = rccl_ready();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(s->device));
    siftmi_exchange *x = new siftmi_exchange();
    x->s = s; x->rank = rank; x->world = world;
    siftmi_gather_plan_init(&x->plan, s->kp_cap, s->desc_cap);
    if (const char *t = getenv("SIFTMI_EXCHANGE_TIMEOUT_S")) { const double v = atof(t); if (v > 0.0) x->timeout_s = v; }
    hipError_t e = hipStreamCreateWithFlags(&x->gstream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&x->ev_wait, hipEventDisableTiming);
    const size_t n_counts = 2 * (size_t)s->F * s->n_oct;
    for (auto &g : x->g) {
        if (e == hipSuccess) e = hipMalloc((void **)&g.counts, (size_t)world * n_counts * sizeof(int32_t));
        if (e == hipSuccess) e = hipMalloc((void **)&g.totals, (size_t)world * 4 * sizeof(int32_t));
        if (e == hipSuccess) e = hipHostMalloc((void **)&g.h_totals, (size_t)world * 4 * sizeof(int32_t), hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g.ev_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g.ev_totals, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreate(&g.t0);
        if (e == hipSuccess) e = hipEventCreate(&g.t1);
    }
    if (e != hipSuccess) {
        rc = set_error(e == hipErrorOutOfMemory ? SIFTMI_E_NOMEM : SIFTMI_E_HIP, "exchange allocation failed: %s", hipGetErrorString(e));
        siftmi_exchange_destroy(x);
        return rc;
    }
    ncclUniqueId u;
    memcpy(&u, unique_id, sizeof(u));
    ncclResult_t r = rccl().CommInitRank(&x->comm, world, u, rank);
    if (r != ncclSuccess) {
        rc = set_error(SIFTMI_E_HIP, "ncclCommInitRank(rank %d of %d) failed: %s [%s]", rank, world, rccl().GetErrorString(r), rccl().origin.c_str());
        x->comm = nullptr;
        siftmi_exchange_destroy(x);
        return rc;
    }
    // the communicator's own view of the job: a launcher that started fewer ranks than `world`, or two with one rank number, must not
    // go unnoticed until the first collective hangs
    ncclResult_t r1 = rccl().CommCount(x->comm, &x->comm_ranks), r2 = rccl().CommUserRank(x->comm, &x->comm_rank);
    if (r1 != ncclSuccess || r2 != ncclSuccess || x->comm_ranks != world || x->comm_rank != rank) {
        rc = set_error(SIFTMI_E_HIP, "communicator reports rank %d of %d, expected rank %d of %d [%s]", x->comm_rank, x->comm_ranks, rank, world, rccl().origin.c_str());
        x->abort_on_destroy = true;                                                   // (the peers of a mis-launched job may never call destroy)
        siftmi_exchange_destroy(x);
        return rc;
    }
    *out = x;
    return SIFTMI_OK;
}

extern "C" int siftmi_exchange_ranks(siftmi_exchange *x, int32_t *comm_ranks, int32_t *comm_rank) {
    if (!x) return set_error(SIFTMI_E_BADARG, "null exchange");
    if (comm_ranks) *comm_ranks = x->comm_ranks;
    if (comm_rank) *comm_rank = x->comm_rank;
    return SIFTMI_OK;
}

extern "C" int siftmi_exchange_set_timeout(siftmi_exchange *x, double seconds) {
    if (!x || !(seconds > 0.0)) return set_error(SIFTMI_E_BADARG, "bad argument");
    x->timeout_s = seconds;
    return SIFTMI_OK;
}

extern "C" int siftmi_exchange_wait(siftmi_exchange *x) {
    int rc = exchange_check(x);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(x->s->device));
    return exchange_drain(x, "the gathers enqueued so far");
}

// A gathered set whose buffers are too small gets NEW blocks (1.5 x the need, so a growing scene does not reallocate every
// step); the old ones are not freed here -- no hipFree (a device-wide synchronisation) between collectives that the other ranks
// have already enqueued, and a consumer stream may still read them -- but when the set is next recycled (recycle_gather_set).
static int grow_gather(siftmi_exchange *x, GatherSet &g, int64_t send_kp, int64_t send_desc) {
    const size_t need_kp = (size_t)x->world * (size_t)send_kp * sizeof(KeypointRec);
    const size_t need_desc = (size_t)x->world * (size_t)send_desc * sizeof(DescriptorRec);
    if (need_kp > g.kp_bytes) {
        uint8_t *fresh = nullptr;
        const size_t bytes = need_kp + need_kp / 2;
        hipError_t e = hipMalloc((void **)&fresh, bytes);
        if (e != hipSuccess) return set_error(e == hipErrorOutOfMemory ? SIFTMI_E_NOMEM : SIFTMI_E_HIP, "gathered keypoints (%zu bytes): %s", bytes, hipGetErrorString(e));
        if (g.kp) g.retired.push_back(g.kp);
        g.kp = fresh; g.kp_bytes = bytes;
    }
    if (need_desc > g.desc_bytes) {
        uint8_t *fresh = nullptr;
        const size_t bytes = need_desc + need_desc / 2;
        hipError_t e = hipMalloc((void **)&fresh, bytes);
        if (e != hipSuccess) return set_error(e == hipErrorOutOfMemory ? SIFTMI_E_NOMEM : SIFTMI_E_HIP, "gathered descriptors (%zu bytes): %s", bytes, hipGetErrorString(e));
        if (g.desc) g.retired.push_back(g.desc);
        g.desc = fresh; g.desc_bytes = bytes;
    }
    return SIFTMI_OK;
}

// the set is about to hold a new step: blocks it outgrew two gathers ago can go (every gather that wrote them has finished --
// ev_done of the set -- and the window in which a consumer may hold their addresses is over)
static int recycle_gather_set(siftmi_exchange *x, GatherSet &g) {
    if (g.retired.empty()) return SIFTMI_OK;
    if (g.step >= 0) { const int rc = exchange_wait(x, g.ev_done, "the gather", g.step); if (rc) return rc; }
    for (void *p : g.retired) (void)hipFree(p);
    g.retired.clear();
    return SIFTMI_OK;
}

// the three payload gathers of result set `rs` into `g`, `send_*` records per rank
static int payload_gathers(siftmi_exchange *x, StreamResultSet &rs, GatherSet &g, int64_t send_kp, int64_t send_desc, bool with_counts) {
    int rc = grow_gather(x, g, send_kp, send_desc);
    if (rc) return rc;
    const size_t n_counts = 2 * (size_t)x->s->F * x->s->n_oct;
    RCCL_TRY(rccl().GroupStart());
    ncclResult_t r = ncclSuccess;                             // a failing member must not leave the group open
    const char *what = "ncclAllGather(counts)";
    if (with_counts) r = rccl().AllGather(rs.d_counts, g.counts, n_counts, ncclInt32, x->comm, x->gstream);
    if (r == ncclSuccess) { what = "ncclAllGather(keypoints)"; r = rccl().AllGather(rs.d_kp, g.kp, (size_t)send_kp * sizeof(KeypointRec), ncclUint8, x->comm, x->gstream); }
    if (r == ncclSuccess) { what = "ncclAllGather(descriptors)"; r = rccl().AllGather(rs.d_desc, g.desc, (size_t)send_desc * sizeof(DescriptorRec), ncclUint8, x->comm, x->gstream); }
    const ncclResult_t r_end = rccl().GroupEnd();
    if (r != ncclSuccess) return set_error(SIFTMI_E_HIP, "%s failed: %s", what, rccl().GetErrorString(r));
    if (r_end != ncclSuccess) return set_error(SIFTMI_E_HIP, "ncclGroupEnd failed: %s", rccl().GetErrorString(r_end));
    g.sent_kp = send_kp; g.sent_desc = send_desc;
    x->bytes_last = (int64_t)x->world * (int64_t)(n_counts * 4 + (size_t)send_kp * sizeof(KeypointRec) + (size_t)send_desc * sizeof(DescriptorRec) + 16);
    HIP_TRY(hipEventRecord(g.ev_done, x->gstream));
    HIP_TRY(hipEventRecord(rs.ev_gather, x->gstream));
    rs.gather_rec = true;
    return SIFTMI_OK;
}

// totals of g's step are on the host: completeness, next sizes
static int resolve_gather(siftmi_exchange *x, GatherSet &g) {
    if (g.step < 0 || g.resolved) return SIFTMI_OK;
    { const int rc = exchange_wait(x, g.ev_totals, "the totals all-gather", g.step); if (rc) return rc; }
    const int inc = siftmi_gather_plan_resolve(&x->plan, g.h_totals, x->world, g.sent_kp, g.sent_desc);
    if (inc < 0) return inc;
    g.resolved = true;
    g.complete = inc == 0;
    g.needs_regather = inc != 0;
    return SIFTMI_OK;
}

// collective: a step whose payload gathers were undersized is gathered again in full while its result set is still intact
static int regather_if_needed(siftmi_exchange *x, GatherSet &g) {
    if (!g.needs_regather) return SIFTMI_OK;
    g.needs_regather = false;
    StreamResultSet &rs = x->s->sets[(size_t)(g.step % x->s->n_sets)];
    if (rs.step != g.step)
        return set_error(SIFTMI_E_STATE, "step %lld was gathered incompletely and its result set has been reused: call siftmi_exchange_gather after every submit", (long long)g.step);
    int64_t mk = 1, md = 1;
    for (int r = 0; r < x->world; r++) { mk = std::max<int64_t>(mk, g.h_totals[4 * r]); md = std::max<int64_t>(md, g.h_totals[4 * r + 1]); }
    mk = std::min(mk, x->s->kp_cap); md = std::min(md, x->s->desc_cap);
    const int rc = payload_gathers(x, rs, g, mk, md, false);
    if (rc) return rc;
    g.complete = true;
    x->regathered++;
    return SIFTMI_OK;
}

static int collect_time(siftmi_exchange *x, GatherSet &g) {
    if (!g.timed) return SIFTMI_OK;
    g.timed = false;
    const int rc = exchange_wait(x, g.t1, "the gather", g.step);
    if (rc) return rc;
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, g.t0, g.t1) == hipSuccess) { x->ms += ms; x->gathers++; }
    (void)hipGetLastError();
    return SIFTMI_OK;
}

extern "C" int siftmi_exchange_gather(siftmi_exchange *x, int synchronous) {
    { const int rc0 = exchange_check(x); if (rc0) return rc0; }
    siftmi_stream *s = x->s;
    if (s->step_no < 0) return set_error(SIFTMI_E_STATE, "no step submitted yet");
    HIP_TRY(hipSetDevice(s->device));
    StreamResultSet &rs = s->sets[(size_t)(s->step_no % s->n_sets)];
    GatherSet &prev = x->g[x->cur];
    if (prev.step == s->step_no) return set_error(SIFTMI_E_STATE, "step %lld has already been gathered", (long long)s->step_no);
    GatherSet &g = x->g[x->cur ^ 1];
    int rc;
    // the set being recycled held the gather before the previous one: long finished; it was resolved when `prev` was issued
    if ((rc = collect_time(x, g))) return rc;
    if ((rc = recycle_gather_set(x, g))) return rc;
    HIP_TRY(hipStreamWaitEvent(x->gstream, rs.ev_ready, 0));
    HIP_TRY(hipEventRecord(g.t0, x->gstream));
    RCCL_TRY(rccl().AllGather(rs.d_totals, g.totals, 4, ncclInt32, x->comm, x->gstream));
    HIP_TRY(hipMemcpyAsync(g.h_totals, g.totals, (size_t)x->world * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, x->gstream));
    HIP_TRY(hipEventRecord(g.ev_totals, x->gstream));
    // from here on the set holds this step (its totals collective is enqueued on every rank)
    g.step = s->step_no; g.resolved = false; g.complete = false; g.needs_regather = false;
    // the previous step's totals arrived while this step's kernels ran: check it, re-gather it if it was cut short, re-size
    if ((rc = resolve_gather(x, prev))) return rc;
    if ((rc = regather_if_needed(x, prev))) return rc;
    int64_t send_kp = x->plan.send_kp, send_desc = x->plan.send_desc;
    if (synchronous || send_kp < 0) {                      // first step (nothing to size from) or on request: this step's own totals
        if ((rc = exchange_wait(x, g.ev_totals, "the totals all-gather", g.step))) return rc;
        int64_t mk = 1, md = 1;
        for (int r = 0; r < x->world; r++) { mk = std::max<int64_t>(mk, g.h_totals[4 * r]); md = std::max<int64_t>(md, g.h_totals[4 * r + 1]); }
        send_kp = std::min(mk, s->kp_cap); send_desc = std::min(md, s->desc_cap);
    }
    if ((rc = payload_gathers(x, rs, g, send_kp, send_desc, true))) return rc;
    HIP_TRY(hipEventRecord(g.t1, x->gstream));
    g.timed = true;
    x->cur ^= 1;
    if (synchronous || x->plan.send_kp < 0) {
        if ((rc = resolve_gather(x, g))) return rc;          // complete by construction; sets the next sizes
    }
    return SIFTMI_OK;
}

extern "C" int siftmi_exchange_result(siftmi_exchange *x, int back, siftmi_gathered *out, void *consumer_stream, int wait_host) {
    if (!x || !out) return set_error(SIFTMI_E_BADARG, "null argument");
    { const int rc0 = exchange_check(x); if (rc0) return rc0; }
    if (back < 0 || back > 1) return set_error(SIFTMI_E_BADARG, "back must be 0 or 1 (two gathered sets)");
    GatherSet &g = x->g[x->cur ^ back];
    if (g.step < 0) return set_error(SIFTMI_E_STATE, "no such gather yet");
    HIP_TRY(hipSetDevice(x->s->device));
    if (consumer_stream != SIFTMI_NO_STREAM) HIP_TRY(hipStreamWaitEvent((hipStream_t)consumer_stream, g.ev_done, 0));
    if (wait_host) {
        int rc = exchange_wait(x, g.ev_done, "the gather", g.step);
        if (rc) return rc;
        if ((rc = resolve_gather(x, g))) return rc;
    }
    memset(out, 0, sizeof(*out));
    out->step = g.step; out->world = x->world; out->complete = g.complete ? 1 : 0; out->resolved = g.resolved ? 1 : 0;
    out->keypoints = g.kp; out->descriptors = g.desc; out->counts = g.counts; out->totals_device = g.totals; out->totals_host = g.h_totals;
    out->kp_records = g.sent_kp; out->desc_records = g.sent_desc;
    out->kp_stride = g.sent_kp * (int64_t)sizeof(KeypointRec); out->desc_stride = g.sent_desc * (int64_t)sizeof(DescriptorRec);
    return SIFTMI_OK;
}

extern "C" int siftmi_exchange_finish(siftmi_exchange *x, int64_t *regathered_steps, int64_t *overflow_steps) {
    int rc = exchange_check(x);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(x->s->device));
    for (int b = 1; b >= 0; b--) {
        GatherSet &g = x->g[x->cur ^ b];
        if ((rc = resolve_gather(x, g))) return rc;
        if ((rc = regather_if_needed(x, g))) return rc;
    }
    if ((rc = exchange_drain(x, "the last gathers (finish)"))) return rc;
    if ((rc = collect_time(x, x->g[0])) || (rc = collect_time(x, x->g[1]))) return rc;
    if (regathered_steps) *regathered_steps = x->regathered;
    if (overflow_steps) *overflow_steps = x->plan.steps_overflowed;
    return SIFTMI_OK;
}

extern "C" int siftmi_exchange_set_headroom(siftmi_exchange *x, int32_t headroom_percent, int64_t quantum) {
    if (!x || headroom_percent < 0 || headroom_percent > 10000 || quantum < 1) return set_error(SIFTMI_E_BADARG, "bad argument");
    x->plan.headroom_percent = headroom_percent;
    x->plan.quantum = quantum;
    return SIFTMI_OK;
}

extern "C" int siftmi_exchange_stats(siftmi_exchange *x, double *ms, int64_t *gathers, int64_t *bytes_last) {
    if (!x) return set_error(SIFTMI_E_BADARG, "null exchange");
    if (ms) *ms = x->ms;
    if (gathers) *gathers = x->gathers;
    if (bytes_last) *bytes_last = x->bytes_last;
    return SIFTMI_OK;
}
