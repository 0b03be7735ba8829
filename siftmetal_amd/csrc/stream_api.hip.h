// stream_api.hip.h -- siftmi_stream_*: batches of frames through detect + describe with several steps in flight, and the
// small device-memory helpers a host without its own HIP toolchain needs (include/siftmi.h).  Included by siftmi_api.hip.
//
// What it replaces: the reference drives ONE command queue and blocks after every stage (SIFT/SIFT.swift:139-175,
// SIFTOctave.swift:199-288: commit + waitUntilCompleted), so a stream of frames is strictly serial there.  Here a step is
// one asynchronous siftmi_detect_describe_batch_device call; consecutive steps alternate between contexts (each with its own
// pyramid, lists, launch stream and captured launch sequence), packed results rotate through a few device buffer sets, host
// frames are uploaded on a copy stream into rotating staging buffers and results are copied back on a third stream.  All
// ordering is by events on the device; the host blocks only in siftmi_stream_result_host and the explicit waits.
#pragma once

static_assert(sizeof(siftmi_stream_config) == 64 && sizeof(siftmi_step_device) == 56 && sizeof(siftmi_step_host) == 48 &&
              sizeof(siftmi_gathered) == 96 && sizeof(siftmi_gather_plan) == 72, "stream / exchange record layouts (siftmetal_amd/_capi.py, swift/)");

struct StreamResultSet {
    KeypointRec *d_kp = nullptr;
    DescriptorRec *d_desc = nullptr;
    int32_t *d_counts = nullptr;              // [2][F][n_oct]
    int32_t *d_totals = nullptr;              // {n_kp, n_desc, overflow flags, 0}
    int32_t *h_meta = nullptr;                // pinned: 4 totals, then the counts
    siftmi_ctx::PinnedBuf<siftmi_keypoint> h_kp;
    siftmi_ctx::PinnedBuf<siftmi_descriptor> h_desc;
    hipEvent_t ev_ready = nullptr;            // the step that wrote this set has finished (launch stream)
    hipEvent_t ev_d2h = nullptr;              // the host copy started at submit time has finished (d2h stream)
    hipEvent_t ev_gather = nullptr;           // the exchange that read this set has finished (gather stream)
    hipEvent_t ev_tot = nullptr;              // the step's totals have arrived in h_tot (d2h stream): the density hint of later steps
    int32_t *h_tot = nullptr;                 // pinned {n_kp, n_desc, flags, 0}
    bool tot_rec = false;
    bool ready_rec = false, d2h_rec = false, gather_rec = false;
    int64_t step = -1;                        // the step this set holds
    int32_t launch_flags = 0;                 // how the step was launched (siftmi_step_host.launch_flags)
    int64_t spec_kp = 0, spec_desc = 0;       // records covered by the copy started at submit time
    bool host_done = false;                   // siftmi_stream_result_host has completed for `step`
    int32_t nk = 0, nd = 0, flags = 0;
};

struct siftmi_stream {
    siftmi_stream_config scfg;
    int device = 0;
    int n_ctx = 1, n_sets = 1, n_oct = 0, F = 1;
    int64_t kp_cap = 0, desc_cap = 0;
    siftmi_ctx *ctx[4] = {};
    hipStream_t launch[4] = {};
    hipEvent_t ev_producer[4] = {};
    std::vector<StreamResultSet> sets;
    int64_t step_no = -1;
    // host-fed frames
    hipStream_t copy_stream = nullptr, d2h_stream = nullptr;
    std::vector<unsigned char *> staging;
    std::vector<hipEvent_t> ev_uploaded, ev_slot_read;
    std::vector<int64_t> slot_step;
    std::vector<char> slot_read_rec;
    size_t frame_bytes = 0, row_bytes = 0;
    // host results: a step's copy to the host is started at submit time while a consumer keeps reading results on the host
    // (a siftmi_stream_result_host call since the previous submit); a consumer of the device views pays nothing
    bool host_reader = false;
    bool dense = false;                       // the density hint, from the last totals seen (applied to the step's context at submit)
    int density_mode = 0;                     // siftmi_stream_config.density_mode
    int64_t spec_kp = 0, spec_desc = 0;
};

static int64_t round_records(int64_t n, int64_t cap) {
    int64_t v = n + n / 4 + 1;
    v = (v + 1023) / 1024 * 1024;
    return std::max<int64_t>(1, std::min(v, cap));
}

extern "C" int siftmi_device_alloc(int hip_device, size_t bytes, void **ptr) {
    if (!ptr) return set_error(SIFTMI_E_BADARG, "null argument");
    *ptr = nullptr;
    HIP_TRY(hipSetDevice(hip_device));
    hipError_t e = hipMalloc(ptr, bytes ? bytes : 16);
    if (e != hipSuccess) return set_error(e == hipErrorOutOfMemory ? SIFTMI_E_NOMEM : SIFTMI_E_HIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return SIFTMI_OK;
}
extern "C" int siftmi_device_free(void *ptr) {
    if (ptr) HIP_TRY(hipFree(ptr));
    return SIFTMI_OK;
}
extern "C" int siftmi_memcpy(void *dst, const void *src, size_t bytes, int kind) {
    if ((!dst || !src) && bytes) return set_error(SIFTMI_E_BADARG, "null argument");
    if (kind < 0 || kind > 2) return set_error(SIFTMI_E_BADARG, "kind must be 0 (host to device), 1 (device to host) or 2 (device to device)");
    const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (bytes) HIP_TRY(hipMemcpy(dst, src, bytes, k));
    return SIFTMI_OK;
}
extern "C" int siftmi_device_synchronize(int hip_device) {
    HIP_TRY(hipSetDevice(hip_device));
    HIP_TRY(hipDeviceSynchronize());
    return SIFTMI_OK;
}

extern "C" int siftmi_stream_default_config(siftmi_stream_config *scfg, int32_t frames_per_step) {
    if (!scfg) return set_error(SIFTMI_E_BADARG, "scfg is null");
    memset(scfg, 0, sizeof(*scfg));
    scfg->frames_per_step = frames_per_step;
    scfg->steps_in_flight = 2;
    scfg->result_sets = 0;
    scfg->format = SIFTMI_FMT_BGRA8;
    scfg->kp_per_frame = 32768;
    scfg->desc_per_frame = 49152;
    return SIFTMI_OK;
}

extern "C" void siftmi_stream_destroy(siftmi_stream *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    for (int i = 1; i < s->n_ctx; i++) if (s->ctx[i]) siftmi_destroy(s->ctx[i]);      // ctx[0] is borrowed
    for (auto &rs : s->sets) {
        void *ptrs[] = {rs.d_kp, rs.d_desc, rs.d_counts, rs.d_totals};
        for (void *p : ptrs) if (p) (void)hipFree(p);
        if (rs.h_meta) (void)hipHostFree(rs.h_meta);
        rs.h_kp.release(); rs.h_desc.release();
        hipEvent_t evs[] = {rs.ev_ready, rs.ev_d2h, rs.ev_gather, rs.ev_tot};
        for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
        if (rs.h_tot) (void)hipHostFree(rs.h_tot);
    }
    for (unsigned char *p : s->staging) if (p) (void)hipFree(p);
    for (hipEvent_t e : s->ev_uploaded) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : s->ev_slot_read) if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < s->n_ctx; i++) {
        if (s->ev_producer[i]) (void)hipEventDestroy(s->ev_producer[i]);
        if (s->launch[i]) (void)hipStreamDestroy(s->launch[i]);
    }
    if (s->copy_stream) (void)hipStreamDestroy(s->copy_stream);
    if (s->d2h_stream) (void)hipStreamDestroy(s->d2h_stream);
    delete s;
}

extern "C" int siftmi_stream_create(siftmi_ctx *ctx, const siftmi_stream_config *scfg, siftmi_stream **out) {
    if (!ctx || !scfg || !out) return set_error(SIFTMI_E_BADARG, "null argument");
    *out = nullptr;
    if (scfg->frames_per_step < 1 || scfg->frames_per_step > 65536) return set_error(SIFTMI_E_BADARG, "frames_per_step %d out of range", scfg->frames_per_step);
    if (scfg->steps_in_flight < 1 || scfg->steps_in_flight > 4) return set_error(SIFTMI_E_BADARG, "steps_in_flight %d out of range [1, 4]", scfg->steps_in_flight);
    if (scfg->result_sets < 0 || scfg->result_sets > 64) return set_error(SIFTMI_E_BADARG, "result_sets %d out of range [0, 64]", scfg->result_sets);
    if (scfg->format != SIFTMI_FMT_BGRA8 && scfg->format != SIFTMI_FMT_GRAY8 && scfg->format != SIFTMI_FMT_GRAYF32)
        return set_error(SIFTMI_E_BADARG, "unknown pixel format %d", scfg->format);
    if (scfg->kp_per_frame < 0 || scfg->desc_per_frame < 0 || scfg->staging_buffers < 0 || scfg->staging_buffers > 64)
        return set_error(SIFTMI_E_BADARG, "negative capacity / bad staging_buffers");
    if (scfg->density_mode < 0 || scfg->density_mode > 2) return set_error(SIFTMI_E_BADARG, "density_mode %d out of range [0, 2]", scfg->density_mode);
    HIP_TRY(hipSetDevice(ctx->device));
    siftmi_stream *s = new siftmi_stream();
    s->scfg = *scfg;
    s->device = ctx->device;
    s->n_ctx = scfg->steps_in_flight;
    s->F = scfg->frames_per_step;
    s->n_oct = ctx->n_oct;
    s->density_mode = scfg->density_mode;
    // a multiple of the number of contexts, so that a context always meets the same result sets: the library replays a
    // captured launch sequence per (input, output, stream) signature, and every new pairing would be captured afresh
    int n_sets = scfg->result_sets > 0 ? scfg->result_sets : 2 * s->n_ctx;
    n_sets = std::max(n_sets, s->n_ctx);
    s->n_sets = (n_sets + s->n_ctx - 1) / s->n_ctx * s->n_ctx;
    s->kp_cap = (scfg->kp_per_frame > 0 ? scfg->kp_per_frame : 32768) * (int64_t)s->F;
    s->desc_cap = (scfg->desc_per_frame > 0 ? scfg->desc_per_frame : 49152) * (int64_t)s->F;
    const size_t bpp = scfg->format == SIFTMI_FMT_GRAY8 ? 1 : 4;
    s->row_bytes = bpp * (size_t)ctx->cfg.width;
    s->frame_bytes = s->row_bytes * (size_t)ctx->cfg.height;
    s->ctx[0] = ctx;
    int rc = SIFTMI_OK;
    for (int i = 1; i < s->n_ctx && rc == SIFTMI_OK; i++) rc = siftmi_create(&ctx->cfg, ctx->device, &s->ctx[i]);
    hipError_t e = hipSuccess;
    for (int i = 0; i < s->n_ctx && e == hipSuccess && rc == SIFTMI_OK; i++) {
        e = hipStreamCreateWithFlags(&s->launch[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_producer[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = create_copy_stream(&s->copy_stream);          // own hardware queues: see create_copy_stream
    if (e == hipSuccess) e = create_copy_stream(&s->d2h_stream);
    s->sets.resize((size_t)s->n_sets);
    const size_t n_counts = 2 * (size_t)s->F * s->n_oct;
    for (auto &rs : s->sets) {
        if (e != hipSuccess || rc != SIFTMI_OK) break;
        e = hipMalloc((void **)&rs.d_kp, (size_t)s->kp_cap * sizeof(KeypointRec));
        if (e == hipSuccess) e = hipMalloc((void **)&rs.d_desc, (size_t)s->desc_cap * sizeof(DescriptorRec));
        if (e == hipSuccess) e = hipMalloc((void **)&rs.d_counts, n_counts * sizeof(int32_t));
        if (e == hipSuccess) e = hipMalloc((void **)&rs.d_totals, 4 * sizeof(int32_t));
        if (e == hipSuccess) e = hipMemset(rs.d_counts, 0, n_counts * sizeof(int32_t));
        if (e == hipSuccess) e = hipMemset(rs.d_totals, 0, 4 * sizeof(int32_t));
        if (e == hipSuccess) e = hipHostMalloc((void **)&rs.h_meta, (4 + n_counts) * sizeof(int32_t), hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&rs.ev_ready, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&rs.ev_d2h, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&rs.ev_gather, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&rs.ev_tot, hipEventDisableTiming);
        if (e == hipSuccess) e = hipHostMalloc((void **)&rs.h_tot, 4 * sizeof(int32_t), hipHostMallocDefault);
    }
    if (rc == SIFTMI_OK && e != hipSuccess) {
        rc = e == hipErrorOutOfMemory ? SIFTMI_E_NOMEM : SIFTMI_E_HIP;
        set_error(rc, "stream allocation failed: %s", hipGetErrorString(e));
    }
    if (rc != SIFTMI_OK) { siftmi_stream_destroy(s); return rc; }
    *out = s;
    return SIFTMI_OK;
}

extern "C" siftmi_ctx *siftmi_stream_context(siftmi_stream *s, int i) {
    if (!s || i < 0 || i >= s->n_ctx) return nullptr;
    return s->ctx[i];
}

// the copy of a step's packed results to page-locked host memory, started when the step is submitted
static int start_host_copy(siftmi_stream *s, StreamResultSet &rs) {
    const size_t n_counts = 2 * (size_t)s->F * s->n_oct;
    HIP_TRY(hipStreamWaitEvent(s->d2h_stream, rs.ev_ready, 0));
    HIP_TRY(hipMemcpyAsync(rs.h_meta, rs.d_totals, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s->d2h_stream));
    HIP_TRY(hipMemcpyAsync(rs.h_meta + 4, rs.d_counts, n_counts * sizeof(int32_t), hipMemcpyDeviceToHost, s->d2h_stream));
    rs.spec_kp = std::min<int64_t>(s->spec_kp, s->kp_cap);
    rs.spec_desc = std::min<int64_t>(s->spec_desc, s->desc_cap);
    // growing a host block frees the old one: nothing may still be copying into it (rare: sizes are rounded up generously)
    if ((size_t)rs.spec_kp > rs.h_kp.cap || (size_t)rs.spec_desc > rs.h_desc.cap) HIP_TRY(hipStreamSynchronize(s->d2h_stream));
    if (rs.spec_kp > 0) {
        HIP_TRY(rs.h_kp.resize((size_t)rs.spec_kp));
        HIP_TRY(hipMemcpyAsync(rs.h_kp.data(), rs.d_kp, (size_t)rs.spec_kp * sizeof(KeypointRec), hipMemcpyDeviceToHost, s->d2h_stream));
    }
    if (rs.spec_desc > 0) {
        HIP_TRY(rs.h_desc.resize((size_t)rs.spec_desc));
        HIP_TRY(hipMemcpyAsync(rs.h_desc.data(), rs.d_desc, (size_t)rs.spec_desc * sizeof(DescriptorRec), hipMemcpyDeviceToHost, s->d2h_stream));
    }
    HIP_TRY(hipEventRecord(rs.ev_d2h, s->d2h_stream));
    rs.d2h_rec = true;
    return SIFTMI_OK;
}

// slot >= 0: the frames sit in staging buffer `slot` (submit_host); its bookkeeping is committed with the launch, before
// anything that can still fail, so that the next upload into the slot always waits for this step
static int submit_step(siftmi_stream *s, const void *d_pixels, size_t row_stride, size_t frame_stride, void *producer_stream,
                       hipEvent_t uploaded, int64_t *step, int slot = -1) {
    const int64_t k = s->step_no + 1;
    const int ci = (int)(k % s->n_ctx);
    StreamResultSet &rs = s->sets[(size_t)(k % s->n_sets)];
    hipStream_t ls = s->launch[ci];
    if (producer_stream != SIFTMI_NO_STREAM) {
        HIP_TRY(hipEventRecord(s->ev_producer[ci], (hipStream_t)producer_stream));
        HIP_TRY(hipStreamWaitEvent(ls, s->ev_producer[ci], 0));
    }
    // this step overwrites the set: the exchange and the host copy that read it (n_sets steps ago) must have finished
    if (rs.gather_rec) HIP_TRY(hipStreamWaitEvent(ls, rs.ev_gather, 0));
    if (rs.d2h_rec) HIP_TRY(hipStreamWaitEvent(ls, rs.ev_d2h, 0));
    if (rs.tot_rec) HIP_TRY(hipStreamWaitEvent(ls, rs.ev_tot, 0));      // (the 16-byte totals copy of the step that last used this set)
    if (uploaded) HIP_TRY(hipStreamWaitEvent(ls, uploaded, 0));
    // Density hint for the launch graph (siftmi_ctx::dense_hint): the descriptor total of the most recent step whose totals have
    // reached the host -- no wait, a few steps late at most.  Above ~1e4 descriptors per 1080p frame (4.8e-3 per input pixel) the
    // one-chain sequence is the faster one (tools/fork_density_sweep.py, bench.py config.dense), below it the forked one.
    // The hint only selects between launch sequences that compute the same records (one chain without activity flags / forked with
    // them); WHICH one a step gets depends on when totals happen to arrive, so it is reported per step (siftmi_step_host.launch_flags)
    // and can be pinned (siftmi_stream_config.density_mode, siftmi_stream_set_density_mode).
    for (int back = 1; s->density_mode == 0 && back <= s->n_sets && back <= k; back++) {
        StreamResultSet &prev = s->sets[(size_t)((k - back) % s->n_sets)];
        if (prev.step != k - back || !prev.tot_rec || hipEventQuery(prev.ev_tot) != hipSuccess) continue;
        const double per_px = (double)prev.h_tot[1] / ((double)s->F * s->ctx[0]->cfg.width * s->ctx[0]->cfg.height);
        s->dense = per_px > 4.8e-3;
        break;
    }
    if (s->density_mode) s->dense = s->density_mode == 2;
    s->ctx[ci]->dense_hint = s->dense;
    (void)hipGetLastError();                                  // (hipEventQuery's hipErrorNotReady is not an error)
    const int rc = siftmi_detect_describe_batch_device(s->ctx[ci], s->F, d_pixels, s->scfg.format, row_stride, frame_stride, (siftmi_keypoint *)rs.d_kp,
                                                       s->kp_cap, (siftmi_descriptor *)rs.d_desc, s->desc_cap, rs.d_counts, rs.d_totals, ls);
    if (rc) return rc;
    // the step is launched: commit its bookkeeping first (events recorded on a healthy stream do not fail; if one does, the
    // state still says "this set / slot belongs to step k")
    rs.step = k; rs.host_done = false; rs.spec_kp = rs.spec_desc = 0;
    rs.launch_flags = (s->ctx[ci]->dense_hint ? SIFTMI_STEP_DENSE_HINT : 0) | (s->ctx[ci]->last_replayed ? SIFTMI_STEP_GRAPH_REPLAY : 0) |
                      (s->ctx[ci]->last_forked ? SIFTMI_STEP_FORKED : 0) | (s->ctx[ci]->raw_exact ? SIFTMI_STEP_RAW_EXACT : 0);
    rs.gather_rec = false; rs.d2h_rec = false;
    s->step_no = k;
    if (step) *step = k;
    if (slot >= 0) {
        s->slot_step[(size_t)slot] = k;
        if (hipEventRecord(s->ev_slot_read[(size_t)slot], ls) == hipSuccess) s->slot_read_rec[(size_t)slot] = 1;
    }
    HIP_TRY(hipEventRecord(rs.ev_ready, ls));
    rs.ready_rec = true;
    // the step's totals to the host, behind the step, on the copy-back stream: the density hint of the steps after it
    // (small steps -- a frame or two -- are bound by the host's submit rate: their totals are sampled every 8th step; round 4's first
    // form copied every step's and took 640 x 480 single frames, two in flight, from 0.163 to 0.21 ms per step)
    rs.tot_rec = false;
    const bool sample_totals = (long long)s->F * s->ctx[0]->cfg.width * s->ctx[0]->cfg.height >= 8ll * 1024 * 1024 || (k & 7) == 0;
    if (sample_totals && hipStreamWaitEvent(s->d2h_stream, rs.ev_ready, 0) == hipSuccess &&
        hipMemcpyAsync(rs.h_tot, rs.d_totals, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s->d2h_stream) == hipSuccess &&
        hipEventRecord(rs.ev_tot, s->d2h_stream) == hipSuccess) rs.tot_rec = true;
    if (s->host_reader) {
        s->host_reader = false;
        const int rc2 = start_host_copy(s, rs);
        if (rc2) return rc2;
    }
    return SIFTMI_OK;
}

extern "C" int siftmi_stream_submit_device(siftmi_stream *s, const void *d_pixels, size_t row_stride, size_t frame_stride,
                                           void *producer_stream, int64_t *step) {
    if (!s || !d_pixels) return set_error(SIFTMI_E_BADARG, "null argument");
    HIP_TRY(hipSetDevice(s->device));
    return submit_step(s, d_pixels, row_stride, frame_stride, producer_stream, nullptr, step);
}

extern "C" int siftmi_stream_submit_host(siftmi_stream *s, const void *pixels, size_t row_stride, size_t frame_stride, int64_t *step) {
    if (!s || !pixels) return set_error(SIFTMI_E_BADARG, "null argument");
    if (row_stride < s->row_bytes) return set_error(SIFTMI_E_BADARG, "row_stride %zu smaller than a row", row_stride);
    HIP_TRY(hipSetDevice(s->device));
    if (s->staging.empty()) {
        // more buffers than steps in flight, so that an upload never waits for a running step; as many as result sets when
        // that is enough: buffer, result set and context then rotate together (one launch signature per buffer)
        const int n = s->scfg.staging_buffers > 0 ? s->scfg.staging_buffers : (s->n_sets > s->n_ctx ? s->n_sets : 2 * s->n_ctx);
        s->staging.assign((size_t)n, nullptr);
        s->ev_uploaded.assign((size_t)n, nullptr);
        s->ev_slot_read.assign((size_t)n, nullptr);
        s->slot_step.assign((size_t)n, -1);
        s->slot_read_rec.assign((size_t)n, 0);
        for (int i = 0; i < n; i++) {
            hipError_t e = hipMalloc((void **)&s->staging[(size_t)i], (size_t)s->F * s->frame_bytes);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_uploaded[(size_t)i], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_slot_read[(size_t)i], hipEventDisableTiming);
            if (e != hipSuccess) {
                // leave no half-built state behind: a later call starts the allocation over instead of using null buffers
                for (unsigned char *p : s->staging) if (p) (void)hipFree(p);
                for (hipEvent_t ev : s->ev_uploaded) if (ev) (void)hipEventDestroy(ev);
                for (hipEvent_t ev : s->ev_slot_read) if (ev) (void)hipEventDestroy(ev);
                s->staging.clear(); s->ev_uploaded.clear(); s->ev_slot_read.clear(); s->slot_step.clear(); s->slot_read_rec.clear();
                (void)hipGetLastError();
                return set_error(e == hipErrorOutOfMemory ? SIFTMI_E_NOMEM : SIFTMI_E_HIP, "staging buffers: %s", hipGetErrorString(e));
            }
        }
    }
    const int64_t k = s->step_no + 1;
    const size_t slot = (size_t)(k % (int64_t)s->staging.size());
    unsigned char *dst = s->staging[slot];
    // (One copy stream: uploads of consecutive steps alternated between two streams -- two DMA engines sharing the link -- measured
    // 10.6-12.4 ms per step against 10.1-10.2, profiles/host_fed_trace_r05.log.)  The staging slot's last reader finished long ago in
    // steady state (there are more slots than steps in flight): ask before queueing a cross-stream wait in front of the copy.
    hipStream_t up = s->copy_stream;
    if (s->slot_read_rec[slot] && hipEventQuery(s->ev_slot_read[slot]) != hipSuccess) {
        (void)hipGetLastError();
        HIP_TRY(hipStreamWaitEvent(up, s->ev_slot_read[slot], 0));
    }
    if (row_stride == s->row_bytes && frame_stride == s->frame_bytes) {
        HIP_TRY(hipMemcpyAsync(dst, pixels, (size_t)s->F * s->frame_bytes, hipMemcpyHostToDevice, up));
    } else {
        for (int f = 0; f < s->F; f++)
            HIP_TRY(hipMemcpy2DAsync(dst + (size_t)f * s->frame_bytes, s->row_bytes, (const unsigned char *)pixels + (size_t)f * frame_stride, row_stride,
                                     s->row_bytes, (size_t)s->ctx[0]->cfg.height, hipMemcpyHostToDevice, up));
    }
    HIP_TRY(hipEventRecord(s->ev_uploaded[slot], up));
    return submit_step(s, dst, s->row_bytes, s->frame_bytes, SIFTMI_NO_STREAM, s->ev_uploaded[slot], step, (int)slot);
}

extern "C" int siftmi_stream_wait_upload(siftmi_stream *s, int64_t step) {
    if (!s) return set_error(SIFTMI_E_BADARG, "null stream");
    if (step < 0 || step > s->step_no) return set_error(SIFTMI_E_BADARG, "step %lld has not been submitted", (long long)step);
    if (s->staging.empty()) return SIFTMI_OK;
    const size_t slot = (size_t)(step % (int64_t)s->staging.size());
    if (s->slot_step[slot] == step) HIP_TRY(hipEventSynchronize(s->ev_uploaded[slot]));     // a later step in the slot: uploaded long ago
    return SIFTMI_OK;
}

extern "C" int siftmi_stream_wait_consumed(siftmi_stream *s, int64_t step) {
    if (!s) return set_error(SIFTMI_E_BADARG, "null stream");
    if (step < 0 || step > s->step_no) return set_error(SIFTMI_E_BADARG, "step %lld has not been submitted", (long long)step);
    StreamResultSet &rs = s->sets[(size_t)(step % s->n_sets)];
    if (rs.step == step && rs.ready_rec) HIP_TRY(hipEventSynchronize(rs.ev_ready));
    return SIFTMI_OK;
}

static int result_set_of(siftmi_stream *s, int back, StreamResultSet **out) {
    if (!s) return set_error(SIFTMI_E_BADARG, "null stream");
    if (s->step_no < 0) return set_error(SIFTMI_E_STATE, "no step submitted yet");
    if (back < 0 || back >= s->n_sets || back > s->step_no)
        return set_error(SIFTMI_E_BADARG, "back = %d: the stream keeps %d result sets and has run %lld steps", back, s->n_sets, (long long)(s->step_no + 1));
    StreamResultSet &rs = s->sets[(size_t)((s->step_no - back) % s->n_sets)];
    if (rs.step != s->step_no - back) return set_error(SIFTMI_E_STATE, "result set of step %lld is gone", (long long)(s->step_no - back));
    *out = &rs;
    return SIFTMI_OK;
}

extern "C" int siftmi_stream_result_device(siftmi_stream *s, int back, siftmi_step_device *out, void *consumer_stream) {
    StreamResultSet *rs = nullptr;
    int rc = result_set_of(s, back, &rs);
    if (rc) return rc;
    if (!out) return set_error(SIFTMI_E_BADARG, "null argument");
    HIP_TRY(hipSetDevice(s->device));
    if (consumer_stream != SIFTMI_NO_STREAM) HIP_TRY(hipStreamWaitEvent((hipStream_t)consumer_stream, rs->ev_ready, 0));
    out->step = rs->step;
    out->keypoints = (const siftmi_keypoint *)rs->d_kp; out->descriptors = (const siftmi_descriptor *)rs->d_desc;
    out->counts = rs->d_counts; out->totals = rs->d_totals;
    out->kp_capacity = s->kp_cap; out->desc_capacity = s->desc_cap;
    return SIFTMI_OK;
}

extern "C" int siftmi_stream_result_host(siftmi_stream *s, int back, siftmi_step_host *out) {
    StreamResultSet *prs = nullptr;
    int rc = result_set_of(s, back, &prs);
    if (rc) return rc;
    if (!out) return set_error(SIFTMI_E_BADARG, "null argument");
    StreamResultSet &rs = *prs;
    HIP_TRY(hipSetDevice(s->device));
    s->host_reader = true;                                    // the next submit starts its host copy itself
    if (!rs.host_done) {
        if (!rs.d2h_rec) {
            const int rc2 = start_host_copy(s, rs);             // (copies nothing but the totals and counts when no size is known yet)
            if (rc2) return rc2;
        }
        HIP_TRY(hipEventSynchronize(rs.ev_d2h));
        rs.nk = (int32_t)std::min<int64_t>(std::max(rs.h_meta[0], 0), s->kp_cap);
        rs.nd = (int32_t)std::min<int64_t>(std::max(rs.h_meta[1], 0), s->desc_cap);
        rs.flags = rs.h_meta[2];
        // the step found more than the copy started at submit time covered: fetch the rest now
        bool more = false;
        if (rs.nk > rs.spec_kp) {
            if ((size_t)rs.nk > rs.h_kp.cap) { HIP_TRY(rs.h_kp.resize((size_t)rs.nk)); rs.spec_kp = 0; }      // new block: copy everything
            HIP_TRY(hipMemcpyAsync(rs.h_kp.data() + rs.spec_kp, rs.d_kp + rs.spec_kp, (size_t)(rs.nk - rs.spec_kp) * sizeof(KeypointRec),
                                   hipMemcpyDeviceToHost, s->d2h_stream));
            more = true;
        }
        if (rs.nd > rs.spec_desc) {
            if ((size_t)rs.nd > rs.h_desc.cap) { HIP_TRY(rs.h_desc.resize((size_t)rs.nd)); rs.spec_desc = 0; }
            HIP_TRY(hipMemcpyAsync(rs.h_desc.data() + rs.spec_desc, rs.d_desc + rs.spec_desc, (size_t)(rs.nd - rs.spec_desc) * sizeof(DescriptorRec),
                                   hipMemcpyDeviceToHost, s->d2h_stream));
            more = true;
        }
        if (more) {
            HIP_TRY(hipEventRecord(rs.ev_d2h, s->d2h_stream));  // a following submit into this set waits for these copies too
            HIP_TRY(hipEventSynchronize(rs.ev_d2h));
        }
        rs.host_done = true;
        s->spec_kp = round_records(rs.nk, s->kp_cap);
        s->spec_desc = round_records(rs.nd, s->desc_cap);
    }
    out->step = rs.step;
    out->keypoints = rs.h_kp.data(); out->descriptors = rs.h_desc.data();
    out->counts = rs.h_meta + 4;
    out->n_keypoints = rs.nk; out->n_descriptors = rs.nd; out->overflow_flags = rs.flags; out->launch_flags = rs.launch_flags;
    if (rs.flags & 32) return set_error(SIFTMI_E_BADARG, "step %lld: SIFTMI_FMT_GRAYF32 frame with a value outside [0, 1] (include/siftmi.h, siftmi_format)", (long long)rs.step);
    if (rs.flags) return set_error(SIFTMI_E_CAPACITY, "list capacity exceeded in step %lld (overflow flags 0x%x): results truncated", (long long)rs.step, rs.flags);
    return SIFTMI_OK;
}

extern "C" int siftmi_stream_set_density_mode(siftmi_stream *s, int mode) {
    if (!s || mode < 0 || mode > 2) return set_error(SIFTMI_E_BADARG, "bad argument");
    s->density_mode = mode;
    return SIFTMI_OK;
}

extern "C" int siftmi_stream_synchronize(siftmi_stream *s) {
    if (!s) return set_error(SIFTMI_E_BADARG, "null stream");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->copy_stream));
    for (int i = 0; i < s->n_ctx; i++) HIP_TRY(hipStreamSynchronize(s->launch[i]));
    HIP_TRY(hipStreamSynchronize(s->d2h_stream));
    return SIFTMI_OK;
}
