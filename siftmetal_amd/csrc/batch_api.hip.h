// batch_api.hip.h -- a batched call = sub-batches of the launch sequence, captured into hipGraphs and replayed by signature;
// the entry points SIFT.getKeypoints maps to (siftmi_detect_describe_batch[_device], siftmi_detect).
// Part of the siftmi_api.hip translation unit.
#pragma once

// ------------------------------------------------------------------------------------------------
// the launch sequence of one batched call (all sub-batches); everything asynchronous on `st`
static int enqueue_batch(siftmi_ctx *c, hipStream_t st, int32_t n_frames, const void *d_pixels, int format, size_t row_stride,
                         size_t frame_stride, KeypointRec *d_kp, long long kp_cap, DescriptorRec *d_desc, long long desc_cap,
                         int32_t *d_counts, int32_t *d_totals, bool fork) {
    int rc;
    c->tstream = st;
    for (int f0 = 0; f0 < n_frames; f0 += c->B) {
        const int nf = std::min(c->B, n_frames - f0);
        const unsigned char *px = (const unsigned char *)d_pixels + (size_t)f0 * frame_stride;
        if ((rc = run_dense_detect(c, st, nf, px, format, row_stride, frame_stride, fork, f0 == 0))) return rc;
        if (!fork) {                                       // forked: the per-octave chains ran them (run_dense_detect)
            if ((rc = run_refine(c, st, nf, -1))) return rc;
            if ((rc = run_describe(c, st, nf, -1))) return rc;
        }
        // the last sub-batch's offsets kernel also writes the caller's {n_kp, n_desc, overflow flags, 0} (no copy node at the end)
        if ((rc = run_pack(c, st, nf, f0, n_frames, d_kp, kp_cap, d_desc, desc_cap, d_counts, c->d_stats, f0 + nf >= n_frames ? d_totals : nullptr))) return rc;
        c->last_sub_frames = nf;
    }
    return SIFTMI_OK;
}

// Replays the captured launch sequence of call signature `key` on `st`, capturing it first on the signature's SECOND sighting (a
// caller that passes fresh buffers with every call would otherwise pay capture + instantiation each time and fill the cache with
// graphs that are never replayed); *launched = false: nothing was enqueued (graphs off, first sighting, cache full on a runtime
// that cannot destroy graphs, capture failed) and the caller issues direct launches.  `enqueue(fork)` issues the launch sequence.
template <typename Enqueue>
static int replay_or_capture(siftmi_ctx *c, hipStream_t st, const siftmi_ctx::GraphKey &key_in, Enqueue enqueue, bool *launched) {
    siftmi_ctx::GraphKey key = key_in;
    key.fork = fork_chains(c);                               // (may change between calls: the stream's density hint)
    key.dense = c->dense_hint;
    *launched = false;
    static const bool graphs_off = getenv("SIFTMI_NO_GRAPH") != nullptr;    // (read once: no environment scan per call)
    const bool want_graph = c->cfg.use_hip_graph && !c->timing && !c->graph_failed && !graphs_off;
    if (!want_graph) return SIFTMI_OK;
    int rc = SIFTMI_OK;
    hipGraphExec_t exec = nullptr;
    bool seen = false;
    for (const auto &k : c->gseen) seen = seen || k == key;
    if (!seen) {
        // (as many candidates as graphs may be cached: a call of n sub-batches shows n signatures before the first repeats; with 16
        // entries round 4's 64-frame host-fed call at max_batch 8 evicted every signature before its second sighting -- ADVICE r4)
        if (c->gseen.size() >= graph_cache_cap(c)) c->gseen.erase(c->gseen.begin());
        c->gseen.push_back(key);
    }
    for (size_t i = 0; i < c->gcache.size(); i++)
        if (c->gcache[i].key == key) {                       // hit: move to the back (most recently used)
            const siftmi_ctx::GraphEntry hit = c->gcache[i];
            c->gcache.erase(c->gcache.begin() + (long)i);
            c->gcache.push_back(hit);
            exec = hit.exec;
            // a replay does not run the host code that sets it; later sub-batches of one call can only take it away
            c->raw_exact = key.frame_base == 0 ? hit.raw_exact : (c->raw_exact && hit.raw_exact);
            break;
        }
    // A full cache evicts its least recently used signature -- but only on a runtime whose hipGraphExecDestroy is safe
    // (graph_destroy_safe): on an older one an evicted graph can only be abandoned (a few hundred kB each), so a long-running
    // caller whose buffers keep changing would leak without bound.  There the cache simply stops growing and new signatures
    // run as direct launches (round 2's behaviour).
    bool may_capture = seen;
    if (!exec && seen && c->gcache.size() >= graph_cache_cap(c)) {
        if (graph_destroy_safe()) {
            (void)hipDeviceSynchronize();                  // it may still be running
            retire_exec(c->gcache.front().exec);
            c->gcache.erase(c->gcache.begin());
        } else {
            may_capture = false;
        }
    }
    if (!exec && may_capture) {
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            rc = enqueue(key.fork);
            e = hipStreamEndCapture(st, &graph);
            if (rc == SIFTMI_OK && e == hipSuccess && graph) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            else if (rc == SIFTMI_OK && e == hipSuccess) e = hipErrorUnknown;
            if (graph) (void)hipGraphDestroy(graph);
        }
        if (rc != SIFTMI_OK || e != hipSuccess || !exec) {
            (void)hipGetLastError();
            exec = nullptr;
            c->graph_failed = true;          // fall through to direct launches, now and later
        } else {
            // what run_dense_detect decided for THIS launch sequence: raw extrema counts are exact unless an octave's scan skips rows
            bool seq_exact = true;
            for (int o = 0; o < c->n_oct; o++) seq_exact = seq_exact && !c->act_valid[o];
            c->gcache.push_back(siftmi_ctx::GraphEntry{key, exec, seq_exact});
            c->n_graph_captures++;
        }
    }
    if (exec) {
        StageRange rg("siftmi graph replay (detect+describe batch)");
        HIP_TRY(hipGraphLaunch(exec, st));
        *launched = true;
        c->n_graph_replays++;
        c->last_replayed = true; c->last_forked = key.fork;
    }
    return SIFTMI_OK;
}

extern "C" int siftmi_detect_describe_batch_device(siftmi_ctx *c, int32_t n_frames, const void *d_pixels, int format, size_t row_stride,
                                                   size_t frame_stride, siftmi_keypoint *d_keypoints, int64_t kp_capacity,
                                                   siftmi_descriptor *d_descriptors, int64_t desc_capacity, int32_t *d_counts,
                                                   int32_t *d_totals, void *stream) {
    if (!c || !d_pixels || !d_keypoints || !d_descriptors || !d_counts) return set_error(SIFTMI_E_BADARG, "null argument");
    if (n_frames < 1) return set_error(SIFTMI_E_BADARG, "n_frames must be >= 1");
    int rc = check_format(c, format, row_stride);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    rc = ensure_stats(c, n_frames);
    if (rc) return rc;
    if ((rc = order_begin(c, st))) return rc;            // after whatever the previous call left running on another stream
    c->stats_on_device = true;
    const siftmi_ctx::GraphKey key{d_pixels, n_frames, format, row_stride, frame_stride, d_keypoints, (long long)kp_capacity, d_descriptors,
                                   (long long)desc_capacity, d_counts, d_totals, st, 0, n_frames};
    auto enqueue = [&](bool fork) {
        return enqueue_batch(c, st, n_frames, d_pixels, format, row_stride, frame_stride, (KeypointRec *)d_keypoints, kp_capacity,
                             (DescriptorRec *)d_descriptors, desc_capacity, d_counts, d_totals, fork);
    };
    bool launched = false;
    if ((rc = replay_or_capture(c, st, key, enqueue, &launched))) return rc;
    if (launched) c->last_sub_frames = std::min(c->B, n_frames - ((n_frames - 1) / c->B) * c->B);
    else {
        if ((rc = enqueue(false))) return rc;
        c->n_direct_sequences++;
        c->last_replayed = false; c->last_forked = false;
    }
    c->last_frames = n_frames;
    c->pyramid_valid = true;
    return order_end(c, st);
}

static int grow_outputs(siftmi_ctx *c, long long kp_need, long long desc_need) {
    if (kp_need > c->out_kp_cap) {
        if (c->d_out_kp) (void)hipFree(c->d_out_kp);
        c->d_out_kp = nullptr; c->out_kp_cap = 0;
        HIP_TRY(hipMalloc((void **)&c->d_out_kp, (size_t)kp_need * sizeof(KeypointRec)));
        c->out_kp_cap = kp_need;
    }
    if (desc_need > c->out_desc_cap) {
        if (c->d_out_desc) (void)hipFree(c->d_out_desc);
        c->d_out_desc = nullptr; c->out_desc_cap = 0;
        HIP_TRY(hipMalloc((void **)&c->d_out_desc, (size_t)desc_need * sizeof(DescriptorRec)));
        c->out_desc_cap = desc_need;
    }
    return SIFTMI_OK;
}

static int overflow_error(siftmi_ctx *c, int flags) {
    if (flags & 32)
        return set_error(SIFTMI_E_BADARG, "SIFTMI_FMT_GRAYF32 frame with a value outside [0, 1] (or a NaN): float input is the luma a unorm texture "
                                          "delivers; rescale it (include/siftmi.h, siftmi_format)");
    std::string what;
    if (flags & 1) what += " extrema(max_extrema)";
    if (flags & 2) what += " keypoints(max_keypoints)";
    if (flags & 4) what += " descriptors(max_descriptors)";
    if (flags & 8) what += " keypoint-output";
    if (flags & 16) what += " descriptor-output";
    int32_t mx[5] = {0, 0, 0, 0, 0};
    const size_t stride = (size_t)c->last_frames * c->n_oct;
    for (int k = 0; k < 5; k++)
        for (size_t i = 0; i < stride && (k * stride + i) < c->h_stats.size(); i++) mx[k] = std::max(mx[k], c->h_stats[k * stride + i]);
    return set_error(SIFTMI_E_CAPACITY, "list capacity exceeded:%s; largest per-(frame,octave) counts: candidates %d keypoints %d descriptors %d",
                     what.c_str(), mx[1], mx[2], mx[4]);
}

// stage frames from the host (or accept a device pointer) and return the device view
// Host frames go to one of two staging slots on the copy stream; the compute stream waits for the slot's copy, and the
// copy of the next sub-batch waits until the compute stream has consumed the slot (input_consumed).  The H2D copy of
// sub-batch i+1 therefore overlaps the kernels of sub-batch i.
static int stage_input(siftmi_ctx *c, int nf, const void *pixels, int format, size_t row_stride, size_t frame_stride, int on_device,
                       const void **d_px, size_t *d_row, size_t *d_frame) {
    if (on_device) { *d_px = pixels; *d_row = row_stride; *d_frame = frame_stride; return SIFTMI_OK; }
    const size_t bpp = format == SIFTMI_FMT_GRAY8 ? 1 : 4;
    const size_t row = bpp * (size_t)c->cfg.width;
    const int slot = c->input_slot;
    unsigned char *dst = c->d_input + (size_t)slot * c->B * c->input_bytes;
    HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->ev_consumed[slot], 0));
    if (row_stride == row && frame_stride == row * (size_t)c->cfg.height && c->input_bytes == frame_stride) {
        HIP_TRY(hipMemcpyAsync(dst, pixels, (size_t)nf * frame_stride, hipMemcpyHostToDevice, c->copy_stream));
    } else {
        for (int f = 0; f < nf; f++)
            HIP_TRY(hipMemcpy2DAsync(dst + (size_t)f * c->input_bytes, row, (const unsigned char *)pixels + (size_t)f * frame_stride, row_stride, row,
                                     c->cfg.height, hipMemcpyHostToDevice, c->copy_stream));
    }
    HIP_TRY(hipEventRecord(c->ev_copied[slot], c->copy_stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_copied[slot], 0));
    *d_px = dst; *d_row = row; *d_frame = c->input_bytes;
    return SIFTMI_OK;
}

// called after the kernels that read the staged frames have been enqueued on the compute stream
static int input_consumed(siftmi_ctx *c, int on_device) {
    if (on_device) return SIFTMI_OK;
    HIP_TRY(hipEventRecord(c->ev_consumed[c->input_slot], c->stream));
    c->input_slot ^= 1;
    return SIFTMI_OK;
}

// Pinned host memory for callers that feed frames from the host: H2D copies from it are asynchronous and run at the
// full PCIe rate (pageable memory is staged through the runtime's bounce buffers at roughly half that).
extern "C" int siftmi_host_alloc(size_t bytes, void **ptr) {
    if (!ptr) return set_error(SIFTMI_E_BADARG, "null argument");
    *ptr = nullptr;
    hipError_t e = hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return set_error(e == hipErrorOutOfMemory ? SIFTMI_E_NOMEM : SIFTMI_E_HIP, "hipHostMalloc: %s", hipGetErrorString(e));
    return SIFTMI_OK;
}

extern "C" int siftmi_host_free(void *ptr) {
    if (!ptr) return SIFTMI_OK;
    HIP_TRY(hipHostFree(ptr));
    return SIFTMI_OK;
}

extern "C" int siftmi_detect_describe_batch(siftmi_ctx *c, int32_t n_frames, const void *pixels, int format, size_t row_stride,
                                            size_t frame_stride, int on_device, const siftmi_keypoint **keypoints,
                                            const int32_t **kp_counts, const siftmi_descriptor **descriptors, const int32_t **desc_counts) {
    if (!c || !pixels) return set_error(SIFTMI_E_BADARG, "null argument");
    if (n_frames < 1) return set_error(SIFTMI_E_BADARG, "n_frames must be >= 1");
    int rc = check_format(c, format, row_stride);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    long long kp_need = 0, desc_need = 0;
    for (int o = 0; o < c->n_oct; o++) { kp_need += c->P.cap_kp[o]; desc_need += c->P.cap_desc[o]; }
    kp_need = std::min<long long>(kp_need, 1 << 17) * n_frames;
    desc_need = std::min<long long>(desc_need, 3 << 16) * n_frames;
    if ((rc = grow_outputs(c, kp_need, desc_need))) return rc;
    if ((rc = ensure_stats(c, n_frames))) return rc;
    hipStream_t st = c->stream;
    c->tstream = st;
    if ((rc = order_begin(c, st))) return rc;
    c->stats_on_device = false;
    // Host frames of a call that spans several sub-batches: the kernels of sub-batch i run under the upload of sub-batch i + 1
    // (stage_input), but nothing runs under the FIRST upload -- so the first sub-batch is a quarter of the lock-step size (its
    // upload is the only exposed one; frames are independent, so the split changes no result).  Round 4: 64 x 1080p from pinned
    // memory through 16-frame sub-batches, 4 + 16 + 16 + 16 + 12 instead of 4 x 16.
    const int first_nf = (!on_device && n_frames > c->B && c->B >= 4) ? c->B / 4 : c->B;
    const int n_sub = first_nf < c->B ? 1 + (n_frames - first_nf + c->B - 1) / c->B : (n_frames + c->B - 1) / c->B;
    // The launch sequence of sub-batch i is keyed by its staging slot, so a call must start on the same slot every time or an odd
    // sub-batch count doubles the signatures of a call shape (9 sub-batches x 2 slots cycling through a 16-entry candidate
    // list were never captured).  The previous host call has synchronised, both slots are free (and ev_consumed still orders them).
    c->input_slot = 0;
    // ... and a call of n sub-batches needs n signatures alive at once, so the cache grows with the largest call seen -- up to 256 graphs
    // (a hipGraphExec of this launch sequence holds ~40 kernel nodes with their kernargs); a call of more sub-batches than that cycles
    // through the cache (least recently used out, sequences captured again: correct, only slower) instead of pinning thousands of graphs
    // for the life of the context.
    c->graph_min_cap = std::min<size_t>(std::max(c->graph_min_cap, (size_t)n_sub + 8), 256);
    HIP_TRY(c->h_sub.resize(4 * (size_t)n_sub));
    HIP_TRY(c->h_kp.resize(1)); HIP_TRY(c->h_desc.resize(1));          // (callers get non-null pointers for empty results too)
    while ((int)c->ev_sub.size() < n_sub) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->ev_sub.push_back(e);
    }
    if (!c->d2h_stream) HIP_TRY(create_copy_stream(&c->d2h_stream));
    int sub = 0;
    for (int f0 = 0, nf = 0; f0 < n_frames; f0 += nf, sub++) {
        nf = std::min(f0 == 0 ? first_nf : c->B, n_frames - f0);
        const void *d_px; size_t d_row, d_frame;
        const unsigned char *src = (const unsigned char *)pixels + (size_t)f0 * frame_stride;
        if ((rc = stage_input(c, nf, src, format, row_stride, frame_stride, on_device, &d_px, &d_row, &d_frame))) return rc;
        // The launch sequence of a sub-batch is captured and replayed like a device-resident call's (round 4: until then this entry
        // issued ~40 direct launches per sub-batch on one chain): its signature is the staging slot (or the caller's device
        // frames), the sub-batch's place in the call and the context's own output buffers, so repeated calls of one shape
        // replay ~5 graphs with the per-octave chains forked.  The staged frames are free again when the whole sequence has run.
        auto enqueue = [&](bool fork) -> int {
            int r;
            if ((r = run_dense_detect(c, st, nf, d_px, format, d_row, d_frame, fork, f0 == 0))) return r;
            if (!fork) {
                if ((r = run_refine(c, st, nf))) return r;
                if ((r = run_describe(c, st, nf))) return r;
            }
            return run_pack(c, st, nf, f0, n_frames, c->d_out_kp, c->out_kp_cap, c->d_out_desc, c->out_desc_cap, c->d_out_counts, c->d_stats);
        };
        const siftmi_ctx::GraphKey key{d_px, nf, format, d_row, d_frame, c->d_out_kp, c->out_kp_cap, c->d_out_desc, c->out_desc_cap,
                                       c->d_out_counts, c->d_stats, st, f0, n_frames};
        bool launched = false;
        // Frames that are being uploaded: the sub-batch's sequence stays ONE chain.  A forked graph's four octave chains take all
        // four hardware queues the runtime gives a process by default, the copy stream then shares one with a chain and the upload
        // of the next sub-batch waits for this one's kernels instead of running under them (measured: 19.5-19.8 ms per 64 x 1080p
        // call forked, 14.8-16.0 one chain; with the copy streams on hardware queues of their own -- SIFTMI_COPY_STREAM_PRIORITY=1
        // -- 16.4-16.7 forked).
        bool host_fork = false;
        host_fork = exp_set("SIFTMI_EXP_HOST_FORK");
        auto enqueue_g = [&](bool fork) { return enqueue(fork && (on_device != 0 || host_fork)); };
        if ((rc = replay_or_capture(c, st, key, enqueue_g, &launched))) return rc;
        if (!launched) {
            if ((rc = enqueue(false))) return rc;
            c->n_direct_sequences++;
            c->last_replayed = false; c->last_forked = false;
        }
        if ((rc = input_consumed(c, on_device))) return rc;
        // the running totals after this sub-batch: its packed records are final from here on
        HIP_TRY(hipMemcpyAsync(c->h_sub.data() + 4 * sub, c->d_state, sizeof(PackState), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(c->ev_sub[(size_t)sub], st));
        c->last_sub_frames = nf;
    }
    const size_t ng = (size_t)n_frames * c->n_oct;
    c->h_counts.resize(2 * ng); c->h_stats.resize(5 * ng);
    HIP_TRY(hipMemcpyAsync(c->h_counts.data(), c->d_out_counts, 2 * ng * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(c->h_stats.data(), c->d_stats, 5 * ng * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    // Copy-back.  The packed records of sub-batch i go to the page-locked result buffers as soon as its totals are on the host,
    // on a third stream, while the later sub-batches still compute -- as long as the buffers (sized by earlier calls, + 50 %)
    // hold them; otherwise (first call, a much denser batch) everything is copied after the last sub-batch, as before round 4.
    size_t done_k = 0, done_d = 0;
    bool incremental = true;
    for (int i = 0; i < n_sub; i++) {
        HIP_TRY(hipEventSynchronize(c->ev_sub[(size_t)i]));
        const size_t tk = (size_t)std::max(c->h_sub.data()[4 * i], 0), td = (size_t)std::max(c->h_sub.data()[4 * i + 1], 0);
        if (!incremental || tk > c->h_kp.cap || td > c->h_desc.cap) { incremental = false; continue; }
        if (tk > done_k) HIP_TRY(hipMemcpyAsync(c->h_kp.data() + done_k, c->d_out_kp + done_k, (tk - done_k) * sizeof(KeypointRec), hipMemcpyDeviceToHost, c->d2h_stream));
        if (td > done_d) HIP_TRY(hipMemcpyAsync(c->h_desc.data() + done_d, c->d_out_desc + done_d, (td - done_d) * sizeof(DescriptorRec), hipMemcpyDeviceToHost, c->d2h_stream));
        done_k = std::max(done_k, tk); done_d = std::max(done_d, td);
    }
    PackState ps;
    memcpy(&ps, c->h_sub.data() + 4 * (n_sub - 1), sizeof(ps));
    HIP_TRY(hipStreamSynchronize(st));                       // counts and statistics
    HIP_TRY(hipStreamSynchronize(c->d2h_stream));
    if (!incremental) {
        HIP_TRY(c->h_kp.resize((size_t)std::max(ps.total_kp, 1))); HIP_TRY(c->h_desc.resize((size_t)std::max(ps.total_desc, 1)));
        if (ps.total_kp) HIP_TRY(hipMemcpyAsync(c->h_kp.data(), c->d_out_kp, (size_t)ps.total_kp * sizeof(KeypointRec), hipMemcpyDeviceToHost, st));
        if (ps.total_desc) HIP_TRY(hipMemcpyAsync(c->h_desc.data(), c->d_out_desc, (size_t)ps.total_desc * sizeof(DescriptorRec), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    t_collect(c);
    c->last_frames = n_frames;
    c->pyramid_valid = true;
    if ((rc = order_end(c, st))) return rc;
    if (keypoints) *keypoints = c->h_kp.data();
    if (kp_counts) *kp_counts = c->h_counts.data();
    if (descriptors) *descriptors = c->h_desc.data();
    if (desc_counts) *desc_counts = c->h_counts.data() + ng;
    if (ps.overflow_flags) return overflow_error(c, ps.overflow_flags);
    return SIFTMI_OK;
}

// SIFT.getKeypoints (SIFT.swift:147-152): one frame, detection only; the pyramid stays resident.
extern "C" int siftmi_detect(siftmi_ctx *c, const void *pixels, int format, size_t row_stride, int on_device,
                             const siftmi_keypoint **keypoints, int32_t *counts) {
    if (!c || !pixels || !counts) return set_error(SIFTMI_E_BADARG, "null argument");
    int rc = check_format(c, format, row_stride);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    c->tstream = st;
    if ((rc = order_begin(c, st))) return rc;
    c->stats_on_device = false;
    const void *d_px; size_t d_row, d_frame;
    if ((rc = stage_input(c, 1, pixels, format, row_stride, 0, on_device, &d_px, &d_row, &d_frame))) return rc;
    if ((rc = run_dense_detect(c, st, 1, d_px, format, d_row, d_frame, false, true))) return rc;
    if ((rc = input_consumed(c, on_device))) return rc;
    if ((rc = run_refine(c, st, 1))) return rc;
    std::vector<int32_t> h(5 * (size_t)c->B * c->n_oct + sizeof(PackState) / sizeof(int32_t));   // the counters and, behind them, the PackState
    HIP_TRY(hipMemcpyAsync(h.data(), c->d_counters, h.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const size_t cs = (size_t)c->B * c->n_oct;
    int flags = h[5 * cs + 2] & 32;                         // PackState.overflow_flags: the float-input range check
    size_t total = 0;
    c->h_stats.assign(5 * (size_t)c->n_oct, 0);
    for (int o = 0; o < c->n_oct; o++) {
        if (h[C_CAND * cs + o] > c->P.cap_ext[o]) flags |= 1;
        int nk = h[C_KP * cs + o];
        if (nk > c->P.cap_kp[o]) { flags |= 2; nk = c->P.cap_kp[o]; }
        counts[o] = nk;
        total += nk;
        for (int k = 0; k < 3; k++) c->h_stats[(size_t)k * c->n_oct + o] = h[k * cs + o];
    }
    HIP_TRY(c->h_kp.resize(std::max<size_t>(total, 1)));
    size_t pos = 0;
    for (int o = 0; o < c->n_oct; o++) {
        if (counts[o]) HIP_TRY(hipMemcpyAsync(c->h_kp.data() + pos, c->d_kp + c->P.kp_off[o], (size_t)counts[o] * sizeof(KeypointRec), hipMemcpyDeviceToHost, st));
        pos += counts[o];
    }
    HIP_TRY(hipStreamSynchronize(st));
    t_collect(c);
    c->last_frames = 1; c->last_sub_frames = 1; c->pyramid_valid = true;
    if ((rc = order_end(c, st))) return rc;
    if (keypoints) *keypoints = c->h_kp.data();
    if (flags) return overflow_error(c, flags);
    return SIFTMI_OK;
}
