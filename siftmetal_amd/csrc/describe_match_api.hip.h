// describe_match_api.hip.h -- SIFT.getDescriptors on caller-filtered keypoints (siftmi_describe) and the rows after describe
// (SURVEY.md section 8f): the brute-force matcher's plan and launches, the trie matcher, matchGeometry, descriptor index vectors.
// Part of the siftmi_api.hip translation unit.
#pragma once

// SIFT.getDescriptors (SIFT.swift:207-238): keypoints (possibly filtered by the caller) in, descriptors out
extern "C" int siftmi_describe(siftmi_ctx *c, const siftmi_keypoint *keypoints, const int32_t *counts,
                               const siftmi_descriptor **descriptors, int32_t *desc_counts) {
    if (!c || !counts || !desc_counts) return set_error(SIFTMI_E_BADARG, "null argument");
    if (!c->pyramid_valid) return set_error(SIFTMI_E_STATE, "siftmi_describe needs a preceding siftmi_detect on this context");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    c->tstream = st;
    {
        const int rc0 = order_begin(c, st);
        if (rc0) return rc0;
    }
    if (c->stats_on_device) { c->h_stats.clear(); c->stats_on_device = false; }
    const size_t cs = (size_t)c->B * c->n_oct;
    std::vector<int32_t> h(cs, 0);
    size_t pos = 0;
    for (int o = 0; o < c->n_oct; o++) {
        if (counts[o] < 0 || counts[o] > c->P.cap_kp[o])
            return set_error(SIFTMI_E_CAPACITY, "octave %d: %d keypoints exceed max_keypoints %d", o, counts[o], c->P.cap_kp[o]);
        if (counts[o] && !keypoints) return set_error(SIFTMI_E_BADARG, "keypoints is null");
        for (int k = 0; k < counts[o]; k++) {
            const siftmi_keypoint &kp = keypoints[pos + k];
            if (kp.scale < 0 || kp.scale >= c->nspo + 3 || !(kp.sigma > 0.0f))
                return set_error(SIFTMI_E_BADARG, "octave %d keypoint %d: scale %d / sigma %g invalid", o, k, kp.scale, (double)kp.sigma);
        }
        if (counts[o]) HIP_TRY(hipMemcpyAsync(c->d_kp + c->P.kp_off[o], keypoints + pos, (size_t)counts[o] * sizeof(KeypointRec), hipMemcpyHostToDevice, st));
        h[o] = counts[o];
        pos += counts[o];
    }
    HIP_TRY(hipMemcpyAsync(cnt(c, C_KP), h.data(), cs * sizeof(int32_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(zero_i32_kernel, dim3(1), dim3(256), 0, st, cnt(c, C_ORIENTED), 2 * cs);
    int rc;
    if ((rc = run_describe(c, st, 1))) return rc;
    std::vector<int32_t> hc(5 * cs);
    HIP_TRY(hipMemcpyAsync(hc.data(), c->d_counters, hc.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    int flags = 0;
    size_t total = 0;
    if (c->h_stats.size() != 5 * (size_t)c->n_oct) c->h_stats.assign(5 * (size_t)c->n_oct, 0);
    for (int o = 0; o < c->n_oct; o++) {
        int nd = hc[C_DESC * cs + o];
        if (nd > c->P.cap_desc[o]) { flags |= 4; nd = c->P.cap_desc[o]; }
        desc_counts[o] = nd;
        total += nd;
        c->h_stats[(size_t)2 * c->n_oct + o] = counts[o];
        c->h_stats[(size_t)3 * c->n_oct + o] = hc[C_ORIENTED * cs + o];
        c->h_stats[(size_t)4 * c->n_oct + o] = hc[C_DESC * cs + o];
    }
    HIP_TRY(c->h_desc.resize(std::max<size_t>(total, 1)));
    pos = 0;
    for (int o = 0; o < c->n_oct; o++) {
        if (desc_counts[o]) HIP_TRY(hipMemcpyAsync(c->h_desc.data() + pos, c->d_desc + c->P.desc_off[o], (size_t)desc_counts[o] * sizeof(DescriptorRec), hipMemcpyDeviceToHost, st));
        pos += desc_counts[o];
    }
    HIP_TRY(hipStreamSynchronize(st));
    t_collect(c);
    c->last_frames = 1;
    if ((rc = order_end(c, st))) return rc;
    if (descriptors) *descriptors = c->h_desc.data();
    if (flags) return overflow_error(c, flags);
    return SIFTMI_OK;
}

// One record per source comes back through a page-locked landing buffer; the matches (target >= 0) are kept, in source order.
static int collect_matches(siftmi_ctx *c, hipStream_t st, int64_t n_source) {
    HIP_TRY(c->h_match_all.resize((size_t)n_source));
    const siftmi_match *all = c->h_match_all.data();
    HIP_TRY(hipMemcpyAsync(c->h_match_all.data(), c->d_match_out, (size_t)n_source * sizeof(MatchRec), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    c->h_matches.resize((size_t)n_source);
    siftmi_match *out = c->h_matches.data();
    size_t k = 0;
    for (int64_t i = 0; i < n_source; i++) { out[k] = all[i]; k += all[i].target >= 0; }      // branch-free: every record is stored, matches advance
    c->h_matches.resize(k);
    return SIFTMI_OK;
}

// Launch geometry of the brute-force matcher for a problem size (also answered by siftmi_match_plan).
struct MatchPlan { long long groups, split_len, n_split; bool bounded; };
static MatchPlan match_plan(long long n_source, long long n_target) {
    MatchPlan p;
    p.groups = (n_source + MM_SRC_PER_BLOCK - 1) / MM_SRC_PER_BLOCK;
    // Target splits: each a contiguous target range (a multiple of the staging quantum).  A CU holds 2 blocks, so 512 run at a time.
    // Measured (tools/match_plan_sweep.sh, round 4: 20k ... 200k squared, 2 ... 63 splits, with and without the bound):
    //  * small problems (one round of blocks covers them with chunks of a few thousand targets): as many splits as fit ONE round,
    //    no bound -- its pre-pass is a dependent launch of ~30 us (20k: 12 splits 0.135 ms; 30k: 8 splits 0.256 ms);
    //  * otherwise chunks that start from a bound, 1300-2300 blocks, the count whose last round is fullest (50k: 15 splits 0.49 ms
    //    against 0.71 for the 26 unbounded splits round 3's rule chose; 70k: 0.83 against 1.18; 100k: 10 splits 1.55; 200k: 5).
    const long long groups = p.groups;
    const long long quanta = (n_target + MM_SPLIT_QUANTUM - 1) / MM_SPLIT_QUANTUM;
    const long long min_q = 2048 / MM_SPLIT_QUANTUM;                                    // a bounded chunk is at least 2048 targets
    long long q_best;
    bool bounded;
    if (groups * ((n_target + 2047) / 2048) <= 1024) {
        const long long k = std::max<long long>(2, 512 / groups);
        q_best = std::max<long long>(2, (quanta + k - 1) / k);
        bounded = false;
    } else {
        const long long lo = std::max<long long>(2, (1280 + groups - 1) / groups), hi = std::max<long long>(lo, 2304 / groups);
        double best_eff = -1.0;
        q_best = 0;
        for (long long k = lo; k <= hi; k++) {
            const long long q = std::max<long long>(min_q, (quanta + k - 1) / k);
            const long long ns = (quanta + q - 1) / q, blocks = ns * groups;
            const double eff = (double)blocks / (double)((blocks + 511) / 512 * 512);
            if (eff > best_eff + 1e-9) { best_eff = eff; q_best = q; }
        }
        bounded = true;
    }
    p.split_len = q_best * MM_SPLIT_QUANTUM;
    p.n_split = (n_target + p.split_len - 1) / p.split_len;
    p.bounded = bounded && p.n_split >= 2;
    return p;
}

extern "C" int siftmi_match_plan(int64_t n_source, int64_t n_target, int64_t *split_len, int64_t *n_split, int *bounded) {
    if (n_source <= 0 || n_target <= 0 || n_source > (1ll << 30) || n_target > (1ll << 30)) return set_error(SIFTMI_E_BADARG, "bad argument");
    const MatchPlan p = match_plan(n_source, n_target);
    if (split_len) *split_len = p.split_len;
    if (n_split) *n_split = p.n_split;
    if (bounded) *bounded = p.bounded ? 1 : 0;
    return SIFTMI_OK;
}

// The matcher's launch sequence on `st`: leaves one record per source (target -1 = no match) in c->d_match_out.  d_src / d_tgt: device memory.
// d_packed / d_count (the device-resident call): where the matched records and their number go; *packed_done says whether this sequence
// already wrote them (the fused single-launch form) or the caller still has to compact c->d_match_out (block_count_out).
constexpr int MM_FUSED_MAX_GROUPS = 128;
static int enqueue_match(siftmi_ctx *c, hipStream_t st, const DescriptorRec *d_src, int64_t n_source, const DescriptorRec *d_tgt, int64_t n_target,
                         float absolute_threshold, float relative_threshold, int32_t **block_count_out = nullptr, MatchRec *d_packed = nullptr,
                         int32_t *d_count = nullptr, bool *packed_done = nullptr) {
    auto grow = [&](void **p, long long *cap, long long need, size_t elem) -> int {
        if (need <= *cap) return SIFTMI_OK;
        if (*p) (void)hipFree(*p);
        *p = nullptr; *cap = 0;
        HIP_TRY(hipMalloc(p, (size_t)need * elem));
        *cap = need;
        return SIFTMI_OK;
    };
    int rc;
    const MatchPlan plan = match_plan(n_source, n_target);
    const long long groups = plan.groups, split_len = plan.split_len, n_split = plan.n_split;
    if (n_split > 65535) return set_error(SIFTMI_E_BADARG, "too many target splits");
    if (packed_done) *packed_done = false;
    // Calls of the sizes the path produces (two frames' descriptors ... ~60 k x 60 k: the unbounded plans): ONE launch (match_mfma_kernel<true>,
    // match_kernels.hip.h) -- operands straight from the descriptor records, the last block of a source group finalises it and packs its matches.
    static const bool fused_off = getenv("SIFTMI_MATCH_NO_FUSE") != nullptr;
    if (!plan.bounded && groups <= MM_FUSED_MAX_GROUPS && !fused_off) {
        // (Fewer, longer chunks -- so that the tail reads one batch of eight split records instead of three -- were slower: a block's loop is
        // one memory round trip per 64 targets with one iteration prefetched, 2.5 k x 2.3 k 28.1 against 24.9 us: profiles/match_fused_r06.log.)
        const long long n_split_f = n_split, split_len_f = split_len;
        const long long words_f = n_split_f * n_source * 4 + 64;
        if ((rc = grow((void **)&c->d_match_scratch, &c->match_scratch_cap, words_f, sizeof(int)))) return rc;
        if ((rc = grow((void **)&c->d_match_out, &c->match_out_cap, n_source, sizeof(MatchRec)))) return rc;
        if (!c->d_match_sync) {
            HIP_TRY(hipMalloc((void **)&c->d_match_sync, MM_FUSED_MAX_GROUPS * 16));
            HIP_TRY(hipMemset(c->d_match_sync, 0, MM_FUSED_MAX_GROUPS * 16));
        }
        if (++c->match_epoch == 0) c->match_epoch = 1;
        MatchTail tail;
        tail.abs_thr = absolute_threshold; tail.rel_thr = relative_threshold;
        tail.out = c->d_match_out; tail.packed = d_packed; tail.count = d_count;
        tail.status = reinterpret_cast<unsigned long long *>(c->d_match_sync);
        tail.ticket = reinterpret_cast<unsigned *>(c->d_match_sync + MM_FUSED_MAX_GROUPS * 8);
        tail.epoch = c->match_epoch;
        hipLaunchKernelGGL(match_mfma_kernel<true>, dim3((unsigned)groups, (unsigned)n_split_f), dim3(256), 0, st, reinterpret_cast<const int *>(d_src), (int)n_source,
                           reinterpret_cast<const int *>(d_tgt), (const int *)nullptr, (int)n_target, (int)split_len_f, (int4 *)c->d_match_scratch,
                           (const int4 *)nullptr, tail);
        HIP_TRY(hipGetLastError());
        if (packed_done) *packed_done = d_packed != nullptr;
        return SIFTMI_OK;
    }
    // scratch: packed int8 rows + norms for both sides, per-split partial results (one allocation)
    const long long n_blocks = (n_source + 255) / 256;
    const long long words = n_source * 33 + n_target * 33 + (n_split + 1) * n_source * 4 + n_blocks + 64;
    if ((rc = grow((void **)&c->d_match_scratch, &c->match_scratch_cap, words, sizeof(int)))) return rc;
    if ((rc = grow((void **)&c->d_match_out, &c->match_out_cap, n_source, sizeof(MatchRec)))) return rc;
    int *src_packed = c->d_match_scratch;                               // 16-byte aligned pieces first
    int *tgt_packed = src_packed + n_source * 32;
    int4 *part = (int4 *)(tgt_packed + n_target * 32);
    int4 *bound = part + n_split * n_source;                            // the pre-pass's records
    int *src_norm = (int *)(bound + n_source);
    int *tgt_norm = src_norm + n_source;
    int32_t *block_count = tgt_norm + n_target;                         // matches per 256-source block (device-resident variant)
    if (block_count_out) *block_count_out = block_count;
    {
        const unsigned sb = (unsigned)((n_source * 32 + 255) / 256), tb = (unsigned)((n_target * 32 + 255) / 256);
        hipLaunchKernelGGL(match_prep_kernel, dim3(sb + tb), dim3(256), 0, st, d_src, (int)n_source, src_packed, src_norm, (int)sb, d_tgt, (int)n_target,
                           tgt_packed, tgt_norm);
    }
    // starting bound of the chunks (match_kernels.hip.h, round 4): a pre-pass over the first 512 targets, then the chunks.  Short
    // chunks go without: the bound's set-up (a clear of `part`, a dependent launch) costs what it saves there.
    const long long pre_len = 512;
    const bool bounded = plan.bounded;
    static_assert(512 % MM_SPLIT_QUANTUM == 0, "the pre-pass is one split of its own");
    if (bounded) {
        HIP_TRY(hipMemsetAsync(part, 0x7f, (size_t)n_split * (size_t)n_source * sizeof(int4), st));   // "none" (0x7f7f7f7f) until a block publishes
        hipLaunchKernelGGL(match_mfma_kernel<false>, dim3((unsigned)groups, 1), dim3(256), 0, st, src_packed, (int)n_source, tgt_packed, tgt_norm,
                           (int)std::min<long long>(n_target, pre_len), (int)pre_len, bound, (const int4 *)nullptr, MatchTail{});
    }
    hipLaunchKernelGGL(match_mfma_kernel<false>, dim3((unsigned)groups, (unsigned)n_split), dim3(256), 0, st, src_packed, (int)n_source, tgt_packed, tgt_norm,
                       (int)n_target, (int)split_len, part, bounded ? bound : (const int4 *)nullptr, MatchTail{});
    hipLaunchKernelGGL(match_finalize_kernel, dim3((unsigned)((n_source + 255) / 256)), dim3(256), 0, st, part, (int)n_split, src_norm, (int)n_source,
                       absolute_threshold, relative_threshold, c->d_match_out, block_count_out ? block_count : (int32_t *)nullptr);
    HIP_TRY(hipGetLastError());
    return SIFTMI_OK;
}

// SIFTDescriptor.match (SIFT/SIFTDescriptor.swift:298-361) -- see match_kernels.hip.h
extern "C" int siftmi_match_descriptors(siftmi_ctx *c, const siftmi_descriptor *source, int64_t n_source, const siftmi_descriptor *target,
                                        int64_t n_target, int on_device, float absolute_threshold, float relative_threshold,
                                        const siftmi_match **matches, int64_t *count) {
    if (!c || !count || n_source < 0 || n_target < 0 || (n_source && !source) || (n_target && !target))
        return set_error(SIFTMI_E_BADARG, "bad argument");
    if (n_source > (1ll << 30) || n_target > (1ll << 30)) return set_error(SIFTMI_E_BADARG, "too many descriptors");
    *count = 0;
    c->h_matches.clear();
    if (matches) *matches = c->h_matches.data();
    if (n_source == 0 || n_target == 0) return SIFTMI_OK;                    // no target: every match is nil (:340-346)
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    {
        const int rc0 = order_begin(c, st);
        if (rc0) return rc0;
    }
    auto grow = [&](void **p, long long *cap, long long need, size_t elem) -> int {
        if (need <= *cap) return SIFTMI_OK;
        if (*p) (void)hipFree(*p);
        *p = nullptr; *cap = 0;
        HIP_TRY(hipMalloc(p, (size_t)need * elem));
        *cap = need;
        return SIFTMI_OK;
    };
    int rc;
    const DescriptorRec *d_src = (const DescriptorRec *)source, *d_tgt = (const DescriptorRec *)target;
    if (!on_device) {
        if ((rc = grow((void **)&c->d_match_src, &c->match_src_cap, n_source, sizeof(DescriptorRec)))) return rc;
        if ((rc = grow((void **)&c->d_match_tgt, &c->match_tgt_cap, n_target, sizeof(DescriptorRec)))) return rc;
        HIP_TRY(hipMemcpyAsync(c->d_match_src, source, (size_t)n_source * sizeof(DescriptorRec), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(c->d_match_tgt, target, (size_t)n_target * sizeof(DescriptorRec), hipMemcpyHostToDevice, st));
        d_src = c->d_match_src; d_tgt = c->d_match_tgt;
    }
    if ((rc = enqueue_match(c, st, d_src, n_source, d_tgt, n_target, absolute_threshold, relative_threshold))) return rc;
    if (int rcc = collect_matches(c, st, n_source)) return rcc;                        // source order (:304-314)
    *count = (int64_t)c->h_matches.size();
    if (matches) *matches = c->h_matches.data();
    return SIFTMI_OK;
}

// The same match with everything staying in HBM and no host synchronisation: descriptors in device memory, the matched records packed in
// source order into d_matches (capacity n_source), their number in *d_count; asynchronous on `stream`.
extern "C" int siftmi_match_descriptors_device(siftmi_ctx *c, const siftmi_descriptor *d_source, int64_t n_source, const siftmi_descriptor *d_target,
                                               int64_t n_target, float absolute_threshold, float relative_threshold, siftmi_match *d_matches,
                                               int32_t *d_count, void *stream) {
    if (!c || !d_count || n_source < 0 || n_target < 0 || (n_source && (!d_source || !d_matches)) || (n_target && !d_target))
        return set_error(SIFTMI_E_BADARG, "bad argument");
    if (n_source > (1ll << 30) || n_target > (1ll << 30)) return set_error(SIFTMI_E_BADARG, "too many descriptors");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    int rc;
    if ((rc = order_begin(c, st))) return rc;
    if (n_source == 0 || n_target == 0) {                                    // no target: every match is nil (:340-346)
        HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(int32_t), st));
        return order_end(c, st);
    }
    int32_t *block_count = nullptr;
    bool packed_done = false;
    if ((rc = enqueue_match(c, st, (const DescriptorRec *)d_source, n_source, (const DescriptorRec *)d_target, n_target, absolute_threshold, relative_threshold,
                            &block_count, (MatchRec *)d_matches, d_count, &packed_done))) {
        (void)order_end(c, st);                                               // whatever was enqueued before the failure stays ordered
        return rc;
    }
    if (packed_done) return order_end(c, st);                                 // the fused launch packed the matches itself
    const int n_blocks = (int)((n_source + 255) / 256);
    const int prefixed = n_blocks >= 1024 ? 1 : 0;                            // (match_compact_kernel: the in-block sum is quadratic in the block count)
    if (prefixed) hipLaunchKernelGGL(match_block_prefix_kernel, dim3(1), dim3(1024), 0, st, block_count, n_blocks);
    hipLaunchKernelGGL(match_compact_kernel, dim3((unsigned)n_blocks), dim3(256), 0, st, c->d_match_out, (int)n_source, block_count, prefixed,
                       (MatchRec *)d_matches, d_count);
    HIP_TRY(hipGetLastError());
    return order_end(c, st);
}

// SIFTDescriptor.approximateMatch (SIFT/SIFTDescriptor.swift:362-417) -- see trie_kernels.hip.h
extern "C" int siftmi_approximate_match(siftmi_ctx *c, const siftmi_descriptor *source, int64_t n_source, const siftmi_descriptor *target,
                                        int64_t n_target, int on_device, float absolute_threshold, float relative_threshold,
                                        const siftmi_match **matches, int64_t *count) {
    if (!c || !count || n_source < 0 || n_target < 0 || (n_source && !source) || (n_target && !target))
        return set_error(SIFTMI_E_BADARG, "bad argument");
    if (n_source > (1ll << 30) || n_target > (1ll << 30)) return set_error(SIFTMI_E_BADARG, "too many descriptors");
    *count = 0;
    c->h_matches.clear();
    if (matches) *matches = c->h_matches.data();
    if (n_source == 0 || n_target == 0) return SIFTMI_OK;                    // empty trie: no queue entries, every match is nil
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    {
        const int rc0 = order_begin(c, st);
        if (rc0) return rc0;
    }
    auto grow = [&](void **p, long long *cap, long long need, size_t elem) -> int {
        if (need <= *cap) return SIFTMI_OK;
        if (*p) (void)hipFree(*p);
        *p = nullptr; *cap = 0;
        HIP_TRY(hipMalloc(p, (size_t)need * elem));
        *cap = need;
        return SIFTMI_OK;
    };
    int rc;
    const DescriptorRec *d_src = (const DescriptorRec *)source, *d_tgt = (const DescriptorRec *)target;
    if (!on_device) {
        if ((rc = grow((void **)&c->d_match_src, &c->match_src_cap, n_source, sizeof(DescriptorRec)))) return rc;
        if ((rc = grow((void **)&c->d_match_tgt, &c->match_tgt_cap, n_target, sizeof(DescriptorRec)))) return rc;
        HIP_TRY(hipMemcpyAsync(c->d_match_src, source, (size_t)n_source * sizeof(DescriptorRec), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(c->d_match_tgt, target, (size_t)n_target * sizeof(DescriptorRec), hipMemcpyHostToDevice, st));
        d_src = c->d_match_src; d_tgt = c->d_match_tgt;
    }
    // scratch: codes in/out (u64), indices in/out (i32), then rocPRIM's temporary storage
    size_t sort_bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, sort_bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (int32_t *)nullptr,
                                      (int32_t *)nullptr, (size_t)n_target, 0, 48, st));
    const long long words = n_target * 6 + (long long)((sort_bytes + 3) / 4) + 64;
    if ((rc = grow((void **)&c->d_match_scratch, &c->match_scratch_cap, words, sizeof(int)))) return rc;
    if ((rc = grow((void **)&c->d_match_out, &c->match_out_cap, n_source, sizeof(MatchRec)))) return rc;
    unsigned long long *codes_in = (unsigned long long *)c->d_match_scratch, *codes = codes_in + n_target;
    int32_t *idx_in = (int32_t *)(codes + n_target), *idx = idx_in + n_target;
    void *sort_tmp = (void *)(idx + n_target + (n_target & 1));               // 8-byte aligned
    hipLaunchKernelGGL(trie_code_kernel, dim3((unsigned)((n_target + 255) / 256)), dim3(256), 0, st, d_tgt, (int)n_target, codes_in, idx_in);
    HIP_TRY(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, codes_in, codes, idx_in, idx, (size_t)n_target, 0, 48, st));   // stable
    hipLaunchKernelGGL(trie_query_kernel, dim3((unsigned)((n_source + 63) / 64)), dim3(64), 0, st, d_src, (int)n_source, d_tgt, codes, idx, (int)n_target,
                       absolute_threshold, relative_threshold, c->d_match_out);
    HIP_TRY(hipGetLastError());
    if (int rcc = collect_matches(c, st, n_source)) return rcc;                        // source order (:375-386)
    *count = (int64_t)c->h_matches.size();
    if (matches) *matches = c->h_matches.data();
    return SIFTMI_OK;
}

// SIFTDescriptor.matchGeometry (SIFT/SIFTDescriptor.swift:104-144)
extern "C" int siftmi_match_geometry(siftmi_ctx *c, const siftmi_descriptor *source, const float *source_xy, int64_t n_source,
                                     const siftmi_descriptor *target, const float *target_xy, int64_t n_target, float absolute_threshold,
                                     float relative_threshold, float *score, int64_t *n_matches) {
    if (!score || (n_source > 0 && !source_xy) || (n_target > 0 && !target_xy)) return set_error(SIFTMI_E_BADARG, "bad argument");
    const siftmi_match *m = nullptr;
    int64_t n = 0;
    const int rc = siftmi_match_descriptors(c, source, n_source, target, n_target, 0, absolute_threshold, relative_threshold, &m, &n);
    if (rc != SIFTMI_OK) return rc;
    if (n_matches) *n_matches = n;
    const int minimum_sample_size = 7, maximum_sample_size = 80;                           // :113-114
    *score = 0.0f;
    if (n >= minimum_sample_size)
        *score = compare_geometry(m, (int)(n < maximum_sample_size ? n : maximum_sample_size), source_xy, target_xy, minimum_sample_size);
    return SIFTMI_OK;
}

// SIFTDescriptor.init derived vectors (SIFT/SIFTDescriptor.swift:36-89)
extern "C" int siftmi_descriptor_index(const siftmi_descriptor *d, int64_t n, float *raw_features, float *index_value, float *index_key) {
    if (n < 0 || (n > 0 && !d)) return set_error(SIFTMI_E_BADARG, "bad argument");
    for (int64_t i = 0; i < n; i++)
        descriptor_index_vectors(d[i], raw_features ? raw_features + i * 128 : nullptr, index_value ? index_value + i * 128 : nullptr,
                                 index_key ? index_key + i * 16 : nullptr);
    return SIFTMI_OK;
}

extern "C" void siftmi_descriptor_to_reference(const siftmi_descriptor *in, int64_t n, siftmi_descriptor_reference *out) {
    for (int64_t i = 0; i < n; i++) {
        out[i].valid = 1;
        out[i].keypoint = in[i].keypoint;
        out[i].theta = in[i].theta;
        for (int k = 0; k < SIFTMI_DESCRIPTOR_FEATURES; k++) out[i].features[k] = in[i].features[k];
    }
}
