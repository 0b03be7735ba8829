// inspect_api.hip.h -- read-backs of intermediate stages (what the parity tests compare with the oracle), counters, stage timings and
// the measurement entry points bench.py's roofline block uses (siftmi_time_blur / _copy / _blur_memory).
// Part of the siftmi_api.hip translation unit.
#pragma once

// ------------------------------------------------------------------------------------------------
// introspection
extern "C" int siftmi_get_stats(siftmi_ctx *c, siftmi_stats *out) {
    if (!c || !out) return set_error(SIFTMI_E_BADARG, "null argument");
    const size_t ng = (size_t)c->last_frames * c->n_oct;
    if (c->stats_on_device && ng > 0) {                    // last call was device-resident: fetch its statistics block now
        HIP_TRY(hipSetDevice(c->device));
        const int rc0 = order_sync(c);
        if (rc0) return rc0;
        c->h_stats.resize(5 * ng);
        HIP_TRY(hipMemcpy(c->h_stats.data(), c->d_stats, 5 * ng * sizeof(int32_t), hipMemcpyDeviceToHost));
        c->stats_on_device = false;
    }
    if (c->h_stats.size() < 5 * ng || ng == 0) return set_error(SIFTMI_E_STATE, "no statistics yet");
    out->raw_extrema_exact = c->raw_exact ? 1 : 0;
    out->n_frames = c->last_frames; out->n_octaves = c->n_oct;
    out->raw_extrema = c->h_stats.data(); out->candidates = c->h_stats.data() + ng; out->keypoints = c->h_stats.data() + 2 * ng;
    out->oriented = c->h_stats.data() + 3 * ng; out->descriptors = c->h_stats.data() + 4 * ng;
    return SIFTMI_OK;
}

extern "C" int siftmi_graph_stats(siftmi_ctx *c, int64_t *captures, int64_t *replays, int64_t *direct_sequences, int32_t *last_flags) {
    if (!c) return set_error(SIFTMI_E_BADARG, "null ctx");
    if (captures) *captures = c->n_graph_captures;
    if (replays) *replays = c->n_graph_replays;
    if (direct_sequences) *direct_sequences = c->n_direct_sequences;
    if (last_flags) *last_flags = (c->last_replayed ? 1 : 0) | (c->last_forked ? 2 : 0) | (c->dense_hint ? 4 : 0);
    return SIFTMI_OK;
}

extern "C" int siftmi_octave_size(siftmi_ctx *c, int o, int32_t *w, int32_t *h, float *delta) {
    if (!c || o < 0 || o >= c->n_oct) return set_error(SIFTMI_E_BADARG, "bad octave");
    if (w) *w = c->ow[o];
    if (h) *h = c->oh[o];
    if (delta) *delta = c->odelta[o];
    return SIFTMI_OK;
}

extern "C" int siftmi_get_sigma(siftmi_ctx *c, int o, int s, float *sigma) {
    if (!c || !sigma || o < 0 || o >= c->n_oct || s < 0 || s >= c->nspo + 3) return set_error(SIFTMI_E_BADARG, "bad octave/scale");
    *sigma = c->osigma[o][s];
    return SIFTMI_OK;
}

extern "C" int siftmi_get_weights(siftmi_ctx *c, int layer, float *weights, int32_t *count) {
    if (!c || !count || layer < 0 || layer > c->nspo + 2) return set_error(SIFTMI_E_BADARG, "bad layer");
    const int n = layer == 0 ? c->seed_taps : c->taps[layer - 1];
    const TapWeights &w = layer == 0 ? c->seed_w : c->layer_w[layer - 1];
    if (weights) memcpy(weights, w.w, sizeof(float) * (size_t)n);
    *count = n;
    return SIFTMI_OK;
}

extern "C" int siftmi_copy_gaussian(siftmi_ctx *c, int frame, int o, int s, float *dst) {
    if (!c || !dst || o < 0 || o >= c->n_oct || s < 0 || s >= c->nspo + 3 || frame < 0 || frame >= c->B)
        return set_error(SIFTMI_E_BADARG, "bad frame/octave/layer");
    if (!c->pyramid_valid) return set_error(SIFTMI_E_STATE, "no pyramid resident");
    HIP_TRY(hipSetDevice(c->device));
    {
        const int rc0 = order_sync(c);
        if (rc0) return rc0;
    }
    HIP_TRY(hipMemcpy(dst, gauss_ptr(c, o, s) + (size_t)frame * c->frame_stride, (size_t)c->ow[o] * c->oh[o] * sizeof(float), hipMemcpyDeviceToHost));
    return SIFTMI_OK;
}

// DoG layer s = G[s + 1] - G[s] (Subtract.metal:12-21): the pipeline never materialises it (extrema / refinement form the
// same single f32 subtraction on the fly), so this read-back subtracts the two Gaussian layers on the host.
extern "C" int siftmi_copy_dog(siftmi_ctx *c, int frame, int o, int s, float *dst) {
    if (!c || !dst || o < 0 || o >= c->n_oct || s < 0 || s >= c->nspo + 2 || frame < 0 || frame >= c->B)
        return set_error(SIFTMI_E_BADARG, "bad frame/octave/scale");
    const size_t n = (size_t)c->ow[o] * c->oh[o];
    std::vector<float> lo(n);
    int rc = siftmi_copy_gaussian(c, frame, o, s, lo.data());
    if (rc) return rc;
    if ((rc = siftmi_copy_gaussian(c, frame, o, s + 1, dst))) return rc;
    for (size_t i = 0; i < n; i++) dst[i] = dst[i] - lo[i];
    return SIFTMI_OK;
}

static int read_counter(siftmi_ctx *c, int which, int frame, int o, int32_t *v) {
    const int rc0 = order_sync(c);
    if (rc0) return rc0;
    HIP_TRY(hipMemcpy(v, cnt(c, which) + (size_t)frame * c->n_oct + o, sizeof(int32_t), hipMemcpyDeviceToHost));
    return SIFTMI_OK;
}

extern "C" int siftmi_copy_extrema(siftmi_ctx *c, int frame, int o, siftmi_extremum *dst, int32_t cap, int32_t *count) {
    if (!c || !count || o < 0 || o >= c->n_oct || frame < 0 || frame >= c->B) return set_error(SIFTMI_E_BADARG, "bad frame/octave");
    HIP_TRY(hipSetDevice(c->device));
    int32_t n = 0;
    int rc = read_counter(c, C_CAND, frame, o, &n);
    if (rc) return rc;
    *count = n;
    const int m = std::min(std::min(n, cap), c->P.cap_ext[o]);
    if (dst && m > 0) {
        HIP_TRY(hipMemcpy(dst, c->d_ext + (size_t)frame * c->P.ext_frame + c->P.ext_off[o], (size_t)m * sizeof(ExtremumRec), hipMemcpyDeviceToHost));
        std::sort(dst, dst + m, [](const siftmi_extremum &a, const siftmi_extremum &b) {
            if (a.scale != b.scale) return a.scale < b.scale;
            if (a.y != b.y) return a.y < b.y;
            return a.x < b.x;
        });
    }
    return SIFTMI_OK;
}

extern "C" int siftmi_copy_orientations(siftmi_ctx *c, int frame, int o, siftmi_orientation *dst, int32_t cap, int32_t *count) {
    if (!c || !count || o < 0 || o >= c->n_oct || frame < 0 || frame >= c->B) return set_error(SIFTMI_E_BADARG, "bad frame/octave");
    HIP_TRY(hipSetDevice(c->device));
    int32_t n = 0;
    int rc = read_counter(c, C_KP, frame, o, &n);
    if (rc) return rc;
    n = std::min(n, c->P.cap_kp[o]);
    *count = n;
    const int m = std::min(n, cap);
    if (dst && m > 0) {
        std::vector<int32_t> oc(m);
        std::vector<float> oa((size_t)m * ORI_BINS);
        const size_t base = (size_t)frame * c->P.kp_frame + c->P.kp_off[o];
        HIP_TRY(hipMemcpy(oc.data(), c->d_ori_count + base, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(oa.data(), c->d_ori_angles + base * ORI_BINS, (size_t)m * ORI_BINS * sizeof(float), hipMemcpyDeviceToHost));
        for (int k = 0; k < m; k++) {
            dst[k].keypoint = k; dst[k].count = oc[k];
            for (int t = 0; t < ORI_BINS; t++) dst[k].orientations[t] = (t < oc[k]) ? oa[(size_t)k * ORI_BINS + t] : 0.0f;
        }
    }
    return SIFTMI_OK;
}

extern "C" int siftmi_copy_descriptor_floats(siftmi_ctx *c, int frame, int o, float *dst, int32_t cap, int32_t *count) {
    if (!c || !count || o < 0 || o >= c->n_oct || frame < 0 || frame >= c->B) return set_error(SIFTMI_E_BADARG, "bad frame/octave");
    if (!c->d_desc_f32) return set_error(SIFTMI_E_STATE, "context was created without keep_descriptor_floats");
    HIP_TRY(hipSetDevice(c->device));
    int32_t n = 0;
    int rc = read_counter(c, C_DESC, frame, o, &n);
    if (rc) return rc;
    n = std::min(n, c->P.cap_desc[o]);
    *count = n;
    const int m = std::min(n, cap);
    if (dst && m > 0)
        HIP_TRY(hipMemcpy(dst, c->d_desc_f32 + ((size_t)frame * c->P.desc_frame + c->P.desc_off[o]) * DESC_N, (size_t)m * DESC_N * sizeof(float), hipMemcpyDeviceToHost));
    return SIFTMI_OK;
}

// ------------------------------------------------------------------------------------------------
// timing
extern "C" int siftmi_enable_timings(siftmi_ctx *c, int enable) {
    if (!c) return set_error(SIFTMI_E_BADARG, "null ctx");
    c->timing = enable != 0;
    return SIFTMI_OK;
}
extern "C" int siftmi_reset_timings(siftmi_ctx *c) {
    if (!c) return set_error(SIFTMI_E_BADARG, "null ctx");
    (void)hipStreamSynchronize(c->stream);
    t_collect(c);
    memset(c->t_ms, 0, sizeof(c->t_ms)); memset(c->t_launches, 0, sizeof(c->t_launches));
    memset(c->t_blur_ms, 0, sizeof(c->t_blur_ms)); memset(c->t_blur_launches, 0, sizeof(c->t_blur_launches));
    return SIFTMI_OK;
}
extern "C" int siftmi_get_timings(siftmi_ctx *c, double *ms, int64_t *launches) {
    if (!c) return set_error(SIFTMI_E_BADARG, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    t_collect(c);
    for (int i = 0; i < SIFTMI_T_COUNT; i++) { if (ms) ms[i] = c->t_ms[i]; if (launches) launches[i] = c->t_launches[i]; }
    return SIFTMI_OK;
}
extern "C" int siftmi_get_blur_layer_timings(siftmi_ctx *c, int o, int layer, double *ms, int64_t *launches, int32_t *marching) {
    if (!c || o < 0 || o >= c->n_oct || layer < 1 || layer > c->nspo + 2) return set_error(SIFTMI_E_BADARG, "bad octave/layer");
    HIP_TRY(hipSetDevice(c->device));
    t_collect(c);
    if (ms) *ms = c->t_blur_ms[o][layer];
    if (launches) *launches = c->t_blur_launches[o][layer];
    if (marching) {                                        // bit 0: blur_ring_kernel; bit 1: the launch writes activity flags; bit 2: blur_chain_kernel
        const int nf = std::min(c->B, std::max(c->last_frames, 1));
        const bool march = uses_march(c, c->ow[o], c->oh[o], nf), chain = chain_tile(c, o, nf) != 0;
        *marching = (march ? 1 : 0) | ((c->act_valid[o] && layer >= 2 && layer <= c->nspo + 1) ? 2 : 0) | (chain ? 4 : 0);
    }
    return SIFTMI_OK;
}
extern "C" int64_t siftmi_blur_algorithmic_bytes(siftmi_ctx *c, int o) {
    if (!c || o < 0 || o >= c->n_oct) return 0;
    return 8ll * c->ow[o] * c->oh[o];
}
extern "C" int siftmi_time_blur(siftmi_ctx *c, int o, int layer, int iters, double *ms_per_launch) {
    if (!c || !ms_per_launch || o < 0 || o >= c->n_oct || layer < 1 || layer > c->nspo + 2 || iters < 1)
        return set_error(SIFTMI_E_BADARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    SeedSource none; memset(&none, 0, sizeof(none));
    Decimate nodec; memset(&nodec, 0, sizeof(nodec));
    hipEvent_t a, b;
    {
        const int rc0 = order_begin(c, c->stream);
        if (rc0) return rc0;
    }
    HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
    HIP_TRY(hipEventRecord(a, c->stream));
    // the launch exactly as the pipeline issues it (run_dense_detect): the layer that feeds the next octave also writes its
    // decimated copy, layers 2 ... nspo+1 of a marching-blur octave also write the extrema activity flags
    Decimate dec = nodec;
    if (layer == c->nspo && o + 1 < c->n_oct) {
        dec.dst = gauss_ptr(c, o + 1, 0); dec.frame_stride = c->frame_stride; dec.w2 = c->ow[o + 1]; dec.h2 = c->oh[o + 1];
    }
    Activity act{nullptr, 0, 0, 0.0f};
    if (!c->cfg.count_raw_extrema && c->ow[o] >= 3 && c->oh[o] >= 3 && uses_march(c, c->ow[o], c->oh[o], c->B) && layer >= 2 && layer <= c->nspo + 1)
        act = Activity{c->d_act + c->act_off[o] + (size_t)(layer - 2) * c->oh[o] * c->act_ncell[o], c->act_frame, c->act_ncell[o], c->prm.dog_threshold * 0.8f};
    for (int i = 0; i < iters; i++)
        HIP_TRY((launch_blur<false>(c, c->stream, (c->taps[layer - 1] - 1) / 2, gauss_ptr(c, o, layer - 1), gauss_ptr(c, o, layer), c->ow[o],
                                    c->oh[o], c->B, c->layer_w[layer - 1], none, dec, act)));
    HIP_TRY(hipEventRecord(b, c->stream));
    HIP_TRY(hipEventSynchronize(b));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *ms_per_launch = (double)ms / iters;
    return SIFTMI_OK;
}
extern "C" int siftmi_time_copy(siftmi_ctx *c, int64_t bytes, int iters, double *ms_per_launch, int64_t *bytes_moved) {
    if (!c || !ms_per_launch || bytes < 4096 || iters < 1) return set_error(SIFTMI_E_BADARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t half = ((size_t)c->B * c->frame_stride * sizeof(float) / 2) & ~(size_t)4095;
    const size_t n = std::min<size_t>((size_t)bytes, half) & ~(size_t)15;
    if (n < 4096) return set_error(SIFTMI_E_STATE, "pyramid too small for a copy measurement");
    {
        const int rc0 = order_begin(c, c->stream);
        if (rc0) return rc0;
    }
    c->pyramid_valid = false;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(c->d_gauss);
    f32x4 *dst = reinterpret_cast<f32x4 *>(reinterpret_cast<unsigned char *>(c->d_gauss) + half);
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
    const dim3 grid((unsigned)((n / 16 + 255) / 256));
    float best = 0.0f;
    for (int nt = 0; nt < 2; nt++) {                          // plain and non-temporal: the ceiling is the faster of the two
        auto launch = [&]() {
            if (nt) hipLaunchKernelGGL(copy_f4_kernel<true>, grid, dim3(256), 0, c->stream, src, dst, n / 16);
            else hipLaunchKernelGGL(copy_f4_kernel<false>, grid, dim3(256), 0, c->stream, src, dst, n / 16);
        };
        launch();                                              // warm-up (clocks, TLB)
        HIP_TRY(hipEventRecord(a, c->stream));
        for (int i = 0; i < iters; i++) launch();
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(b, c->stream));
        HIP_TRY(hipEventSynchronize(b));
        float t = 0.0f;
        HIP_TRY(hipEventElapsedTime(&t, a, b));
        if (nt == 0 || t < best) best = t;
    }
    const float ms = best;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *ms_per_launch = (double)ms / iters;
    if (bytes_moved) *bytes_moved = 2 * (int64_t)n;
    return order_end(c, c->stream);
}

// the ring kernel of `layer` with its arithmetic compiled out (blur_ring_kernel's DBG = 8 | 16)
template <int R>
static hipError_t launch_ring_memory_only(siftmi_ctx *c, hipStream_t st, const float *src, float *dst, int w, int h, int nf, const TapWeights &wt) {
    using Gr = RingGeom<R>;
    SeedSource none; memset(&none, 0, sizeof(none));
    Decimate nodec; memset(&nodec, 0, sizeof(nodec));
    const int chr = march_chunk_rows(h);
    const int total = ((w + Gr::TW - 1) / Gr::TW) * ((h + chr - 1) / chr) * nf;
    hipLaunchKernelGGL((blur_ring_kernel<R, 4, 32, false, false, 24>), dim3(((total + 7) / 8) * 8), dim3(Gr::NTHR), Gr::lds_bytes, st, src, dst, w, h,
                       c->frame_stride, c->frame_stride, wt, nf, chr, nodec, Activity{nullptr, 0, 0, 0.0f}, none);
    return hipGetLastError();
}

extern "C" int siftmi_time_blur_memory(siftmi_ctx *c, int o, int layer, int iters, double *ms_per_launch) {
    if (!c || !ms_per_launch || o < 0 || o >= c->n_oct || layer < 1 || layer > c->nspo + 2 || iters < 1)
        return set_error(SIFTMI_E_BADARG, "bad argument");
    if (!uses_march(c, c->ow[o], c->oh[o], c->B)) return set_error(SIFTMI_E_STATE, "octave %d does not use the marching kernel at this batch size", o);
    HIP_TRY(hipSetDevice(c->device));
    const int R = (c->taps[layer - 1] - 1) / 2;
    {
        const int rc0 = order_begin(c, c->stream);
        if (rc0) return rc0;
    }
    c->pyramid_valid = false;
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
    HIP_TRY(hipEventRecord(a, c->stream));
    for (int i = 0; i < iters; i++) {
        hipError_t e = hipErrorInvalidValue;
        const float *src = gauss_ptr(c, o, layer - 1);
        float *dst = gauss_ptr(c, o, layer);
        switch (R) {
            case 5: e = launch_ring_memory_only<5>(c, c->stream, src, dst, c->ow[o], c->oh[o], c->B, c->layer_w[layer - 1]); break;
            case 7: e = launch_ring_memory_only<7>(c, c->stream, src, dst, c->ow[o], c->oh[o], c->B, c->layer_w[layer - 1]); break;
            case 8: e = launch_ring_memory_only<8>(c, c->stream, src, dst, c->ow[o], c->oh[o], c->B, c->layer_w[layer - 1]); break;
            case 10: e = launch_ring_memory_only<10>(c, c->stream, src, dst, c->ow[o], c->oh[o], c->B, c->layer_w[layer - 1]); break;
            case 13: e = launch_ring_memory_only<13>(c, c->stream, src, dst, c->ow[o], c->oh[o], c->B, c->layer_w[layer - 1]); break;
            default: break;
        }
        if (e != hipSuccess) {
            (void)hipEventDestroy(a); (void)hipEventDestroy(b);
            return set_error(SIFTMI_E_STATE, "no memory-only instantiation for radius %d (default schedule only)", R);
        }
    }
    HIP_TRY(hipEventRecord(b, c->stream));
    HIP_TRY(hipEventSynchronize(b));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *ms_per_launch = (double)ms / iters;
    return order_end(c, c->stream);
}

extern "C" int siftmi_synchronize(siftmi_ctx *c) {
    if (!c) return set_error(SIFTMI_E_BADARG, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    {
        const int rc0 = order_sync(c);
        if (rc0) return rc0;
    }
    t_collect(c);
    return SIFTMI_OK;
}
