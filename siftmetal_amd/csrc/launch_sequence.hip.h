// launch_sequence.hip.h -- which kernel form each stage of the path is launched in, and the launch sequence of one sub-batch
// (seed, Gaussian layers, extrema scan, refine + sort, orientation + descriptors, pack), single-chain or forked per octave.
// Part of the siftmi_api.hip translation unit (included there; uses its context struct and helpers).
#pragma once

// ------------------------------------------------------------------------------------------------
// launches
// Rows per chunk of the marching (ring) blur: a workgroup walks its 128-column strip down one chunk in steps of 32 rows and
// pays 2R extra horizontally blurred rows for the chunk's prologue (20 % of a 128-row chunk at R = 13).  Tall octaves take
// 256-row chunks (tools/ubench/blur_variants.hip, 32 x 3840x2160: 2-4 % faster than 128 at every radius; whole-height strips
// are no faster and leave a worse tail); for 1920x1080 the shorter chunks win (more workgroups than resident slots): 128 rows
// in round 2, 160 since the octave chains of a batch run side by side (round 3: 9.82 against 9.97 ms per step, three interleaved
// pairs of runs; 192: 10.04).
static int march_chunk_rows(int h) {
    return (int)exp_knob(h >= 1600 ? "SIFTMI_EXP_CHUNK_BIG" : "SIFTMI_EXP_CHUNK_SMALL", h >= 1600 ? 256 : 160);
}

// the marching blur is used when its grid has at least this many workgroups (cfg.blur_march_min_blocks, default 800 = about one
// round of resident workgroups; 2000 until round 3: 3 ... 6 frames of 1920x1080 per step run 4-11 % faster with the ring kernel on
// octave 0, tools/batch_size_sweep.py with MARCH_MIN)
static bool uses_march(const siftmi_ctx *c, int w, int h, int nf) {
    using Gm = RingGeom<1>;
    const int chr = march_chunk_rows(h);
    const long long total = (long long)((w + Gm::TW - 1) / Gm::TW) * ((h + chr - 1) / chr) * nf;
    return total >= c->march_min_blocks;
}

// EXPERIMENT: unused dynamic LDS added to every ring launch (fewer resident ring workgroups per CU, so that another stream's
// keypoint kernels find LDS and wave slots beside them)
static size_t ring_pad_lds() {
    return (size_t)exp_knob("SIFTMI_EXP_RING_PAD_LDS", 0);
}

template <int R, bool SEED, bool DEC>
static hipError_t launch_blur_rd(siftmi_ctx *c, hipStream_t st, const float *src, float *dst, int w, int h, int nf,
                                 const TapWeights &wt, const SeedSource &seed, const Decimate &dec, const Activity &act) {
    bool march = false;
    // large launches: marching form (no vertical-halo recompute beyond a chunk's prologue, next rows prefetched under
    // the FMA phases); it needs enough strips x chunks to fill the chip, so small octaves keep the tile kernel.
    using Gr = RingGeom<R>;
    if constexpr (!SEED) {
        int chr = march_chunk_rows(h);
        // The FMA-bound radii pay their chunk prologue (2R rows loaded and blurred horizontally for nothing: 10 % of a 256-row chunk at
        // R = 13) in the resource they are short of, the memory-bound ones prefer many short chunks: per radius, where the grid still
        // covers the chip at least twice (round 4, tools/chunk_probe.py, 64 x 1080p: octave 0 R = 10 / 13 936 / 944 -> 895 / 920 us
        // with 544-row chunks, R = 5 / 7 / 8 792 / 817 / 921 -> 819 / 840 / 948; octave 1 R = 10 / 13 262 / 250 -> 244 / 241 us).
        if (R >= 9) {
            const long long total_long = (long long)((w + Gr::TW - 1) / Gr::TW) * ((h + 543) / 544) * nf;
            bool long_ok = total_long >= 1536;
            long_ok = long_ok && !exp_set("SIFTMI_EXP_CHUNK_BIG") && !exp_set("SIFTMI_EXP_CHUNK_SMALL");
            if (long_ok) chr = 544;
        }
        const int total = ((w + Gr::TW - 1) / Gr::TW) * ((h + chr - 1) / chr) * nf;
        if (uses_march(c, w, h, nf)) {
            march = true;
            dim3 grid(((total + 7) / 8) * 8, 1, 1);
            if (act.dst)
                hipLaunchKernelGGL((blur_ring_kernel<R, 4, 32, DEC, true>), grid, dim3(Gr::NTHR), Gr::lds_bytes_act + ring_pad_lds(), st, src, dst, w, h,
                                   c->frame_stride, c->frame_stride, wt, nf, chr, dec, act, seed);
            else
                hipLaunchKernelGGL((blur_ring_kernel<R, 4, 32, DEC, false>), grid, dim3(Gr::NTHR), Gr::lds_bytes + ring_pad_lds(), st, src, dst, w, h, c->frame_stride,
                                   c->frame_stride, wt, nf, chr, dec, act, seed);
        }
    } else if constexpr (R >= 4 && R <= 6) {
        // the seed layer in marching form (instantiated for the radii around the default schedule's 5; other sigma
        // settings keep the tile kernel): longer chunks, because its prologue runs the luma / upscale expansion twice
        // (round 4, 64 x 1080p: 128 / 192 / 256 / 384 / 544 / 1088 / 2176-row chunks 0.739 / 0.724 / 0.671-0.688 / 0.655 / 0.659 / 0.651 / 0.685 ms)
        int chr = h >= 1600 ? 544 : 256;
        chr = (int)exp_knob("SIFTMI_EXP_SEED_CHUNK", chr);
        const int total = ((w + Gr::TW - 1) / Gr::TW) * ((h + chr - 1) / chr) * nf;
        if (uses_march(c, w, h, nf)) {
            march = true;
            dim3 grid(((total + 7) / 8) * 8, 1, 1);
            switch (seed.format) {
                case FMT_BGRA8:
                    hipLaunchKernelGGL((blur_ring_kernel<R, 4, 32, false, false, 0, FMT_BGRA8>), grid, dim3(Gr::NTHR), Gr::lds_bytes, st, src, dst, w, h,
                                       c->frame_stride, c->frame_stride, wt, nf, chr, dec, act, seed);
                    break;
                case FMT_GRAY8:
                    hipLaunchKernelGGL((blur_ring_kernel<R, 4, 32, false, false, 0, FMT_GRAY8>), grid, dim3(Gr::NTHR), Gr::lds_bytes, st, src, dst, w, h,
                                       c->frame_stride, c->frame_stride, wt, nf, chr, dec, act, seed);
                    break;
                default:
                    hipLaunchKernelGGL((blur_ring_kernel<R, 4, 32, false, false, 0, FMT_GRAYF32>), grid, dim3(Gr::NTHR), Gr::lds_bytes, st, src, dst, w, h,
                                       c->frame_stride, c->frame_stride, wt, nf, chr, dec, act, seed);
                    break;
            }
        }
    }
    if (!march) {
        using S = BlurShip<R>;
        using Gm = typename S::G;
        const int total = ((w + Gm::TW - 1) / Gm::TW) * ((h + Gm::TH - 1) / Gm::TH) * nf;
        dim3 grid(((total + 7) / 8) * 8, 1, 1);          // XCD-aware 1-D tile order, see blur2_kernel
        if constexpr (!SEED) {
            if (act.dst) {
                hipLaunchKernelGGL((blur2_kernel<R, S::TH, S::NTHR, 4, S::RB, false, 1, 0, true, DEC, true>), grid, dim3(S::NTHR), Gm::lds_bytes_act, st, src, dst, w,
                                   h, c->frame_stride, c->frame_stride, wt, seed, nf, dec, act, ZeroJob{nullptr, 0, nullptr, 0});
                return hipGetLastError();
            }
        }
        ZeroJob zj{nullptr, 0, nullptr, 0};
        if (SEED) { zj = c->zero_job; c->zero_job = ZeroJob{nullptr, 0, nullptr, 0}; }       // the seed tile kernel clears the call's counters on its way
        hipLaunchKernelGGL((blur2_kernel<R, S::TH, S::NTHR, 4, S::RB, SEED, 1, 0, true, DEC>), grid, dim3(S::NTHR),
                           SEED ? Gm::seed_lds_bytes : Gm::lds_bytes, st, src, dst, w, h, c->frame_stride, c->frame_stride, wt, seed, nf, dec, act, zj);
    }
    return hipGetLastError();
}

template <int R, bool SEED>
static hipError_t launch_blur_r(siftmi_ctx *c, hipStream_t st, const float *src, float *dst, int w, int h, int nf,
                                const TapWeights &wt, const SeedSource &seed, const Decimate &dec, const Activity &act) {
    if constexpr (!SEED) {
        if (dec.dst) return launch_blur_rd<R, SEED, true>(c, st, src, dst, w, h, nf, wt, seed, dec, act);
    }
    return launch_blur_rd<R, SEED, false>(c, st, src, dst, w, h, nf, wt, seed, dec, act);
}

template <bool SEED>
static hipError_t launch_blur(siftmi_ctx *c, hipStream_t st, int radius, const float *src, float *dst, int w, int h, int nf,
                              const TapWeights &wt, const SeedSource &seed, const Decimate &dec, const Activity &act = Activity{nullptr, 0, 0, 0.0f}) {
    switch (radius) {
#define CASE_R(r) case r: return launch_blur_r<r, SEED>(c, st, src, dst, w, h, nf, wt, seed, dec, act);
        CASE_R(1) CASE_R(2) CASE_R(3) CASE_R(4) CASE_R(5) CASE_R(6) CASE_R(7) CASE_R(8)
        CASE_R(9) CASE_R(10) CASE_R(11) CASE_R(12) CASE_R(13) CASE_R(14) CASE_R(15)
#undef CASE_R
        default: return hipErrorInvalidValue;
    }
}

// Small launches (a frame or two): layers 1-3 and 4-5 of an octave from one launch each (blur_chain_kernel), 32 x 32 tiles and 256
// threads.  Default schedule only -- the radii are template parameters.  (64 x 64 tiles with 1024 threads, for octaves of >= 512
// such tiles, were measured on a 1080p frame's octave 0: 60 + 81 us against 53 + 56 us for the five per-layer launches.)
static int chain_tile(const siftmi_ctx *c, int o, int nf) {            // 0 = this octave goes layer by layer
    if (c->chain_max_tiles <= 0 || c->nspo != 3) return 0;
    static const int want[5] = {11, 15, 17, 21, 27};
    for (int s = 0; s < 5; s++) if (c->taps[s] != want[s]) return 0;
    const int w = c->ow[o], h = c->oh[o];
    if ((w & 3) != 0 || w < 64 || h < 64 || uses_march(c, w, h, nf)) return 0;
    const long long t64 = (long long)((w + 63) / 64) * ((h + 63) / 64) * nf;
    return t64 <= c->chain_max_tiles ? 32 : 0;
}
// The tile kernel writes the DoG activity flags too when the octave is large enough for the flagged-row scan to pay for them (one
// 1920x1080 frame: octave 0's scan is 50-70 us of full rows against the ~10 % the flags add to three of its layers)
static bool tile_flags(const siftmi_ctx *c, int o, int nf) {
    long long min_px = 1500000;
    min_px = exp_knob("SIFTMI_EXP_TILE_ACT_MIN_PX", min_px);
    return (long long)c->ow[o] * c->oh[o] * nf >= min_px;
}
template <int T, int NTHR, int RA, int RB, int RC>
static hipError_t launch_chain_t(siftmi_ctx *c, hipStream_t st, float *layer0, int w, int h, int nf, int first, int dec_layer, const Decimate &dec) {
    using G = ChainGeom<T, NTHR, RA, RB, RC>;
    static_assert(G::lds_bytes <= 64 * 1024, "default dynamic LDS limit");
    ChainWeights wts;
    const int n = (RA > 0) + (RB > 0) + (RC > 0);
    for (int s = 0; s < n; s++) wts.l[s] = c->layer_w[first + s];
    for (int s = n; s < 3; s++) wts.l[s] = c->layer_w[first];
    const int total = ((w + G::T - 1) / G::T) * ((h + G::T - 1) / G::T) * nf;
    hipLaunchKernelGGL((blur_chain_kernel<T, NTHR, RA, RB, RC>), dim3(((total + 7) / 8) * 8), dim3(G::NTHR), G::lds_bytes, st, layer0, w, h, c->frame_stride,
                       (size_t)w * h, first, wts, nf, dec_layer, dec);
    return hipGetLastError();
}
// layers first + 1 ... of octave layer0's octave; first = 0: layers 1-3, first = 3: layers 4-5
static hipError_t launch_blur_chain(siftmi_ctx *c, hipStream_t st, float *layer0, int w, int h, int nf, int first, int dec_layer, const Decimate &dec) {
    if (first == 0) return launch_chain_t<32, 256, 5, 7, 8>(c, st, layer0, w, h, nf, first, dec_layer, dec);
    return launch_chain_t<32, 256, 10, 13, 0>(c, st, layer0, w, h, nf, first, dec_layer, dec);
}

// A descriptor gets a whole workgroup on launches of at most this many octave-0 pixels ("a frame or two")
static long long small_launch_pixels() {
    return exp_knob("SIFTMI_EXP_COOP_PX", 16ll * 1024 * 1024);
}
// The captured launch sequence forks into one chain per octave (run_dense_detect) unless a single frame's first octave is larger
// than this.  Round 2 forked only "a frame or two" (<= 16 Mpixel per launch); measured in round 3 (tools/batch_size_sweep.py,
// bench.py): 3 ... 16 frames of 1920x1080 per step 8-18 % faster forked, 64 frames 10.41 -> 9.98 ms per step with two steps in
// flight (10.89 -> 10.40 one at a time: octave k's keypoint stages and scan run beside octave k+1's pyramid), the host-fed stream
// 12.0 -> 10.8 ms; one 8192 x 8192 tile (268 Mpixel first octave) 6.4 -> 6.6 ms, hence the cap per frame.
static bool fork_chains(const siftmi_ctx *c) {
    long long max_px = 48ll * 1024 * 1024;
    max_px = exp_knob("SIFTMI_EXP_FORK_PX", max_px);
    if (c->cfg.graph_fork) return c->n_oct > 1 && c->cfg.graph_fork > 0;
    return c->n_oct > 1 && (long long)c->ow[0] * c->oh[0] <= max_px && !c->dense_hint;
}

static float *gauss_ptr(siftmi_ctx *c, int o, int s) {
    return c->d_gauss + c->P.oct_offset[o] + (size_t)s * c->ow[o] * c->oh[o];
}

static int32_t *cnt(siftmi_ctx *c, int which) { return c->d_counters + (size_t)which * c->B * c->n_oct; }
enum { C_RAW = 0, C_CAND = 1, C_KP = 2, C_ORIENTED = 3, C_DESC = 4 };

static int ensure_fork(siftmi_ctx *c) {
    if (c->fork_ready) return SIFTMI_OK;
    for (int i = 0; i < c->n_oct; i++) {
        HIP_TRY(hipStreamCreateWithFlags(&c->oct_stream[i], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_fork[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
    }
    c->fork_ready = true;
    return SIFTMI_OK;
}

static int run_refine(siftmi_ctx *c, hipStream_t st, int nf, int only_octave);
static int run_describe(siftmi_ctx *c, hipStream_t st, int nf, int only_octave);

static int launch_extrema(siftmi_ctx *c, hipStream_t st, int nf, int o) {
    if (c->ow[o] < 3 || c->oh[o] < 3) return SIFTMI_OK;
    // rows per workgroup, a multiple of 3 (the row loop is unrolled 3x); with activity flags one lane per window row
    // fetches the flags, so EH + 2 <= 64, and taller blocks amortise that fetch
    int EH = c->act_valid[o] ? 60 : 33;
    if (!c->act_valid[o]) {
        // a frame or two: the scan of a small octave is a handful of workgroups, each walking its 33 rows one dependent row-load
        // latency after the other (20 us for the 480 x 270 octave of a single 1080p frame).  Fewer rows per workgroup until the
        // launch has ~1000 of them: the walk gets shorter by the same factor (the two halo rows per workgroup are L2 hits)
        const long long cols = (c->ow[o] - 2 + EXT_COLS_PER_BLOCK - 1) / EXT_COLS_PER_BLOCK;
        while (EH > 6 && cols * ((c->oh[o] - 2 + EH - 1) / EH) * nf < 1024) EH -= 3;
    }
    t_begin(c, SIFTMI_T_EXTREMA);
    int wpb = 4;
    wpb = (int)exp_knob("SIFTMI_EXP_EXT_WPB", wpb);
    const int cols_per_wg = (wpb == 1 ? 1 : 4) * EXT_COLS_PER_WAVE;
    dim3 grid((c->ow[o] - 2 + cols_per_wg - 1) / cols_per_wg, (c->oh[o] - 2 + EH - 1) / EH, nf);
    const unsigned char *actp = c->act_valid[o] ? c->d_act + c->act_off[o] : nullptr;
#define LAUNCH_EXT_W(NS, SK, W)                                                                                                           \
    hipLaunchKernelGGL((extrema_kernel<NS, SK, W>), grid, dim3(64 * W), 0, st, c->P, c->prm, o, EH, c->d_ext, cnt(c, C_CAND), cnt(c, C_RAW), actp, \
                       c->act_frame, c->act_ncell[o])
#define LAUNCH_EXT(NS)                                                                                                                    \
    do {                                                                                                                                  \
        if (actp) { if (wpb == 1) LAUNCH_EXT_W(NS, true, 1); else LAUNCH_EXT_W(NS, true, 4); }                                            \
        else { if (wpb == 1) LAUNCH_EXT_W(NS, false, 1); else LAUNCH_EXT_W(NS, false, 4); }                                               \
    } while (0)
    switch (c->nspo) {
        case 1: LAUNCH_EXT(1); break;
        case 2: LAUNCH_EXT(2); break;
        case 3: LAUNCH_EXT(3); break;
        case 4: LAUNCH_EXT(4); break;
        case 5: LAUNCH_EXT(5); break;
        case 6: LAUNCH_EXT(6); break;
        default: LAUNCH_EXT(7); break;
    }
#undef LAUNCH_EXT
#undef LAUNCH_EXT_W
    HIP_TRY(hipGetLastError());
    t_end(c);
    return SIFTMI_OK;
}

// Dense front end + extrema for nf frames (DifferenceOfGaussians.swift:346-406, SIFTOctave.swift:177-196):
// seed -> per octave {layer blurs; the one writing layer nspo also emits the next octave's layer 0} -> extrema.
// Octave o+1 depends on octave o only through that layer, so when `fork` is set (graph capture; fork_chains) the rest
// of octave o (its last layers, its extrema scan and its keypoint stages) stays on the current stream while octave o+1
// continues on another one; everything joins before the pack.  (Round 2 forked single frames only: a two-stream probe
// of the DENSE stages of a 64-frame batch showed no gain.  With the keypoint stages on the chains a batch gains 4 %.)
static int run_dense_detect(siftmi_ctx *c, hipStream_t st, int nf, const void *d_pixels, int format, size_t row_stride, size_t frame_stride,
                            bool fork, bool first_of_call) {
    const int NG = c->nspo + 3;
    int rc;
    StageRange rg("siftmi pyramid + extrema (DifferenceOfGaussians.encode, findKeypoints)");
    if (fork && (rc = ensure_fork(c))) return rc;
    SeedSource seed;
    seed.pixels = (const unsigned char *)d_pixels; seed.frame_stride = frame_stride; seed.row_stride = row_stride;
    seed.format = format; seed.in_w = c->cfg.width; seed.in_h = c->cfg.height;
    SeedSource none; memset(&none, 0, sizeof(none));
    Decimate nodec; memset(&nodec, 0, sizeof(nodec));
    // Counters are cleared by a kernel, not hipMemsetAsync: memset nodes captured into the hipGraph stopped clearing from the
    // third launch of a serial graph on (ROCm 7.2; tests/test_gpu_parity.py::test_graph_replays_stay_correct).
    // the first sub-batch of a call also clears the running totals (PackState) that sit behind the counters
    const size_t n_cnt = 5 * (size_t)c->B * c->n_oct + (first_of_call ? sizeof(PackState) / sizeof(int32_t) : 0);
    // (+ the row buckets of the per-octave refine launches of a forked call.)  The seed TILE kernel takes the job along; after a
    // marching seed launch it is still pending and gets its own launch -- nothing before the extrema scans reads these ranges
    c->zero_job = ZeroJob{c->d_counters, n_cnt, fork ? c->d_row_count : nullptr, fork ? (size_t)nf * c->P.row_frame : 0};
    t_begin(c, SIFTMI_T_SEED);
    HIP_TRY((launch_blur<true>(c, st, (c->seed_taps - 1) / 2, nullptr, gauss_ptr(c, 0, 0), c->ow[0], c->oh[0], nf, c->seed_w, seed, nodec)));
    t_end(c);
    if (c->zero_job.a) {
        if (fork) hipLaunchKernelGGL(zero2_i32_kernel, dim3(64), dim3(256), 0, st, c->d_counters, n_cnt, c->d_row_count, (size_t)nf * c->P.row_frame);
        else hipLaunchKernelGGL(zero_i32_kernel, dim3(1), dim3(256), 0, st, c->d_counters, n_cnt);
        c->zero_job = ZeroJob{nullptr, 0, nullptr, 0};
    }
    if (format == SIFTMI_FMT_GRAYF32) {                       // input contract of float frames (siftmi_format); after the counters were cleared
        const long long px = (long long)c->cfg.width * c->cfg.height * nf;
        hipLaunchKernelGGL(check_unit_range_kernel, dim3((unsigned)std::min<long long>((px + 255) / 256, 4096)), dim3(256), 0, st, (const unsigned char *)d_pixels,
                           row_stride, frame_stride, c->cfg.width, c->cfg.height, nf, &c->d_state->overflow_flags);
        HIP_TRY(hipGetLastError());
    }
    hipStream_t cur = st;
    bool joined[MAX_OCT] = {};
    int fork_width = 0;                                       // chains a forked sequence may use (0: one per octave)
    fork_width = (int)exp_knob("SIFTMI_EXP_FORK_WIDTH", fork_width);
    for (int o = 0; o < c->n_oct; o++) {
        hipStream_t next = cur;
        const bool fork_here = fork && o + 1 < c->n_oct && (fork_width == 0 || o + 1 < fork_width);
        // DoG activity flags for the extrema scan: only when every layer of this octave goes through the marching blur
        const int chain = chain_tile(c, o, nf);
        // (dense_hint, set by the frame stream from earlier steps' descriptor totals: on frames that are texture throughout every row
        // is active, the flags skip nothing and only cost the three layers that write them ~11 % each: off)
        c->act_valid[o] = !c->cfg.count_raw_extrema && !c->dense_hint && c->ow[o] >= 3 && c->oh[o] >= 3 &&
                          (uses_march(c, c->ow[o], c->oh[o], nf) || (!chain && tile_flags(c, o, nf)));
        if (o == 0 && first_of_call) c->raw_exact = true;
        if (c->act_valid[o]) c->raw_exact = false;
        if (chain) {                                           // layers 1-3 (and the next octave's layer 0), then layers 4-5: two launches
            Decimate dec = nodec;
            if (o + 1 < c->n_oct) { dec.dst = gauss_ptr(c, o + 1, 0); dec.frame_stride = c->frame_stride; dec.w2 = c->ow[o + 1]; dec.h2 = c->oh[o + 1]; }
            t_begin(c, SIFTMI_T_BLUR, o * 16 + 1);
            HIP_TRY(launch_blur_chain(c, cur, gauss_ptr(c, o, 0), c->ow[o], c->oh[o], nf, 0, dec.dst ? c->nspo : 0, dec));
            t_end(c);
            if (fork_here) {
                HIP_TRY(hipEventRecord(c->ev_fork[o], cur));
                next = c->oct_stream[o + 1];
                HIP_TRY(hipStreamWaitEvent(next, c->ev_fork[o], 0));
            }
            t_begin(c, SIFTMI_T_BLUR, o * 16 + 4);
            HIP_TRY(launch_blur_chain(c, cur, gauss_ptr(c, o, 0), c->ow[o], c->oh[o], nf, 3, 0, nodec));
            t_end(c);
        }
        for (int s = 1; s < NG && !chain; s++) {
            Decimate dec = nodec;
            if (s == c->nspo && o + 1 < c->n_oct) {
                dec.dst = gauss_ptr(c, o + 1, 0); dec.frame_stride = c->frame_stride; dec.w2 = c->ow[o + 1]; dec.h2 = c->oh[o + 1];
            }
            // layers 2 ... nspo+1 complete DoG scales 1 ... nspo, the ones that can hold a candidate
            Activity act{nullptr, 0, 0, 0.0f};
            if (c->act_valid[o] && s >= 2 && s <= c->nspo + 1)
                act = Activity{c->d_act + c->act_off[o] + (size_t)(s - 2) * c->oh[o] * c->act_ncell[o], c->act_frame, c->act_ncell[o],
                               c->prm.dog_threshold * 0.8f};
            t_begin(c, SIFTMI_T_BLUR, o * 16 + s);
            HIP_TRY((launch_blur<false>(c, cur, (c->taps[s - 1] - 1) / 2, gauss_ptr(c, o, s - 1), gauss_ptr(c, o, s), c->ow[o], c->oh[o],
                                        nf, c->layer_w[s - 1], none, dec, act)));
            t_end(c);
            if (fork_here && s == c->nspo) {                          // next octave can start now, on its own stream
                HIP_TRY(hipEventRecord(c->ev_fork[o], cur));
                next = c->oct_stream[o + 1];
                HIP_TRY(hipStreamWaitEvent(next, c->ev_fork[o], 0));
            }
        }
        if ((rc = launch_extrema(c, cur, nf, o))) return rc;
        if (fork) {
            // forked graph (a frame or two): the keypoint stages of this octave follow its extrema scan on the same chain,
            // while the next octaves' blurs run on theirs -- octave 0's descriptors no longer wait for octave 3's pyramid
            if ((rc = run_refine(c, cur, nf, o))) return rc;
            if ((rc = run_describe(c, cur, nf, o))) return rc;
        }
        if (cur != st) { HIP_TRY(hipEventRecord(c->ev_join[o], cur)); joined[o] = true; }
        cur = next;
    }
    for (int o = 0; o < c->n_oct; o++)
        if (joined[o]) HIP_TRY(hipStreamWaitEvent(st, c->ev_join[o], 0));
    return SIFTMI_OK;
}

// refine -> sort  (SIFT.swift:190-202).  only_octave >= 0: that octave's groups only (the per-octave chains of a forked graph;
// the row-bucket counters were cleared up front by run_dense_detect)
static int run_refine(siftmi_ctx *c, hipStream_t st, int nf, int only_octave = -1) {
    const int groups = only_octave < 0 ? nf * c->n_oct : nf;
    PyramidDesc P = c->P;
    P.only_octave = only_octave;
    StageRange rg("siftmi refine + sort (interpolateKeypoints)");
    t_begin(c, SIFTMI_T_REFINE);
    if (only_octave < 0) hipLaunchKernelGGL(zero_i32_kernel, dim3(256), dim3(256), 0, st, c->d_row_count, (size_t)nf * c->P.row_frame);
    hipLaunchKernelGGL(refine_kernel, dim3(64, groups), dim3(256), 0, st, P, c->prm, c->d_ext, cnt(c, C_CAND), c->d_kp_tmp, c->d_keys,
                       cnt(c, C_KP), c->d_row_count);
    HIP_TRY(hipGetLastError());
    t_end(c);
    t_begin(c, SIFTMI_T_SORT);
    const size_t rows_bytes = only_octave >= 0 ? (size_t)(c->nspo + 2) * c->oh[only_octave] * sizeof(int32_t) : 0;
    if (only_octave >= 0 && rows_bytes <= 60 * 1024) {     // a per-octave chain of a forked graph: one launch instead of three
        // LDS behind the row buckets: bucketed keys (8 B) and source indices (4 B) of up to n_lds keypoints (64 KB in all)
        const size_t rows_al = (rows_bytes + 7) & ~(size_t)7;
        const int n_lds = (int)std::min<size_t>(4096, (64 * 1024 - 256 - rows_al) / 12);
        hipLaunchKernelGGL(kp_row_sort_small_kernel, dim3(groups), dim3(1024), rows_al + (size_t)n_lds * 12, st, P, c->d_kp_tmp, c->d_keys, cnt(c, C_KP),
                           c->d_row_count, c->d_bucket_keys, c->d_bucket_src, c->d_kp, n_lds);
        HIP_TRY(hipGetLastError());
        t_end(c);
        return SIFTMI_OK;
    }
    hipLaunchKernelGGL(kp_row_scan_kernel, dim3(groups), dim3(1024), 0, st, P, c->d_row_count, c->d_row_start);
    hipLaunchKernelGGL(kp_row_scatter_kernel, dim3(32, groups), dim3(256), 0, st, P, c->d_keys, cnt(c, C_KP), c->d_row_start, c->d_row_count,
                       c->d_bucket_keys, c->d_bucket_src);
    hipLaunchKernelGGL(kp_row_rank_kernel, dim3(32, groups), dim3(256), 0, st, P, c->d_kp_tmp, c->d_bucket_keys, c->d_bucket_src, cnt(c, C_KP),
                       c->d_row_start, c->d_row_count, c->d_kp);
    HIP_TRY(hipGetLastError());
    t_end(c);
    return SIFTMI_OK;
}

// orientation -> expansion -> descriptors  (SIFT.swift:207-238)
static int run_describe(siftmi_ctx *c, hipStream_t st, int nf, int only_octave = -1) {
    const int groups = only_octave < 0 ? nf * c->n_oct : nf;
    PyramidDesc P = c->P;
    P.only_octave = only_octave;
    StageRange rg("siftmi orientation + descriptors (getDescriptors)");
    t_begin(c, SIFTMI_T_ORIENT);
    const bool coop = (long long)nf * c->ow[0] * c->oh[0] <= small_launch_pixels();   // a frame or two: a whole workgroup per keypoint / descriptor
    // Large launches: ONE wavefront per workgroup (round 4).  Four independent wavefronts per workgroup held its LDS and wave slots
    // until the slowest was done; keypoints and descriptors differ 4x in window size.  Measured, 64 x 1080p: descriptors 1.20 -> 1.04 ms
    // on the benchmark frames, 6.46 -> 6.2 ms on dense texture (tools/dense_stage_times.py); the records do not depend on it.
    int wpb_ori = 1, wpb_desc = 1, wg1 = 1024;                // workgroups per (frame, octave) group of the one-wavefront forms
    wg1 = (int)exp_knob("SIFTMI_EXP_KP_WG", wg1); wpb_ori = (int)exp_knob("SIFTMI_EXP_ORI_WPB", wpb_ori); wpb_desc = (int)exp_knob("SIFTMI_EXP_DESC_WPB", wpb_desc);
    if (coop)
        hipLaunchKernelGGL((orientation_kernel<true, 4>), dim3((unsigned)exp_knob("SIFTMI_EXP_COOP_WG", 1024), groups), dim3(256), 0, st, P, c->prm, c->d_kp, cnt(c, C_KP), c->d_ori_count,
                           c->d_ori_angles);
    else if (wpb_ori == 1)
        hipLaunchKernelGGL((orientation_kernel<false, 1>), dim3(wg1, groups), dim3(64), 0, st, P, c->prm, c->d_kp, cnt(c, C_KP), c->d_ori_count,
                           c->d_ori_angles);
    else
        hipLaunchKernelGGL((orientation_kernel<false, 4>), dim3(256, groups), dim3(256), 0, st, P, c->prm, c->d_kp, cnt(c, C_KP), c->d_ori_count,
                           c->d_ori_angles);
    HIP_TRY(hipGetLastError());
    if (coop) {
        hipLaunchKernelGGL(expand_descriptors_kernel<true>, dim3(groups), dim3(1024), 0, st, P, c->prm, c->d_kp, cnt(c, C_KP), c->d_ori_count, c->d_ori_angles,
                           c->d_desc_in, cnt(c, C_DESC), cnt(c, C_ORIENTED));
    } else {
        hipLaunchKernelGGL(expand_descriptors_kernel<false>, dim3(groups), dim3(1024), 0, st, P, c->prm, c->d_kp, cnt(c, C_KP), c->d_ori_count, c->d_ori_angles,
                           c->d_desc_in, cnt(c, C_DESC), cnt(c, C_ORIENTED));
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(desc_derive_kernel, dim3(64, groups), dim3(256), 0, st, P, c->prm, c->d_kp, cnt(c, C_DESC), c->d_desc_in);
    }
    HIP_TRY(hipGetLastError());
    t_end(c);
    t_begin(c, SIFTMI_T_DESCRIBE);
    // a frame or two: fewer descriptors than wavefront slots -> one workgroup per descriptor (see descriptor_kernel)
    if (coop)
        hipLaunchKernelGGL((descriptor_kernel<true, 4>), dim3((unsigned)exp_knob("SIFTMI_EXP_COOP_WG", 1024), groups), dim3(256), 0, st, P, c->prm, c->d_kp, c->d_desc_in, cnt(c, C_DESC), c->d_desc,
                           c->d_desc_f32, c->d_desc_flag);
    else {
        if (wpb_desc == 1 && c->cfg.descriptor_patch_lds)
            hipLaunchKernelGGL((descriptor_kernel<false, 1, true>), dim3(wg1, groups), dim3(64), 0, st, P, c->prm, c->d_kp, c->d_desc_in, cnt(c, C_DESC),
                               c->d_desc, c->d_desc_f32, c->d_desc_flag);
        else if (wpb_desc == 1)
            hipLaunchKernelGGL((descriptor_kernel<false, 1>), dim3(wg1, groups), dim3(64), 0, st, P, c->prm, c->d_kp, c->d_desc_in, cnt(c, C_DESC), c->d_desc,
                               c->d_desc_f32, c->d_desc_flag);
        else
            hipLaunchKernelGGL((descriptor_kernel<false, 4>), dim3(256, groups), dim3(256), 0, st, P, c->prm, c->d_kp, c->d_desc_in, cnt(c, C_DESC), c->d_desc,
                               c->d_desc_f32, c->d_desc_flag);
    }
    HIP_TRY(hipGetLastError());
    // second pass, only for schedules with very wide descriptor windows (siftmi_create: desc_refine; never the reference's default): the
    // descriptors the first pass flagged as low-contrast, at a finer fixed-point unit -- the same form after every first-pass form, so that
    // all of them give the same bytes (descriptor_kernel, REFINE)
    if (c->desc_refine) {
        hipLaunchKernelGGL((descriptor_kernel<false, 4, false, true>), dim3(coop ? 16 : 64, groups), dim3(256), 0, st, P, c->prm, c->d_kp, c->d_desc_in, cnt(c, C_DESC),
                           c->d_desc, c->d_desc_f32, c->d_desc_flag);
        HIP_TRY(hipGetLastError());
    }
    t_end(c);
    return SIFTMI_OK;
}

static int run_pack(siftmi_ctx *c, hipStream_t st, int nf, int frame_base, int total_frames, KeypointRec *kp_out, long long kp_cap,
                    DescriptorRec *desc_out, long long desc_cap, int32_t *d_counts, int32_t *d_stats, int32_t *d_totals = nullptr) {
    const int groups = nf * c->n_oct;
    StageRange rg("siftmi pack results");
    t_begin(c, SIFTMI_T_PACK);
    hipLaunchKernelGGL(group_offsets_kernel, dim3(1), dim3(256), 0, st, c->P, nf, frame_base, total_frames, cnt(c, C_RAW), cnt(c, C_CAND),
                       cnt(c, C_KP), cnt(c, C_ORIENTED), cnt(c, C_DESC), c->d_dst_off, c->d_dst_off + (size_t)c->B * c->n_oct, d_counts,
                       d_stats, c->d_state, kp_cap, desc_cap, d_totals);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(pack_kernel, dim3(32, groups), dim3(256), 0, st, c->P, c->d_kp, c->d_desc, c->d_dst_off,
                       c->d_dst_off + (size_t)c->B * c->n_oct, d_counts, frame_base, total_frames, kp_out, desc_out);
    HIP_TRY(hipGetLastError());
    t_end(c);
    return SIFTMI_OK;
}

// Retiring executable graphs.  With the HIP runtime that PyTorch 2.10 bundles (a ROCm 7.0 build, the one a process gets when
// torch is imported first) hipGraphExecDestroy left the runtime in a state in which a later hipGraphLaunch of ANOTHER
// executable graph dereferenced a null pointer -- about 1 in 10 runs of tools/fuzz_api.py (600 random operations) segfaulted
// inside hipGraphLaunch, none in 80 runs without the destroy calls (round 1).  The ROCm 7.2 runtime of the image passes the same
// sweep with the destroy calls (round 3, profiles/fuzz_api_r03_graph_destroy.log).  So: destroy on a runtime >= 7.2, abandon
// (leak, a few hundred kB each) on an older one; SIFTMI_GRAPH_DESTROY=0/1 overrides.
static bool graph_destroy_safe() {
    static const int safe = [] {
        if (const char *e = getenv("SIFTMI_GRAPH_DESTROY")) return atoi(e) != 0 ? 1 : 0;
        int v = 0;
        if (hipRuntimeGetVersion(&v) != hipSuccess) return 0;
        return v >= 70200000 ? 1 : 0;                         // HIP_VERSION = major * 10^7 + minor * 10^5 + patch
    }();
    return safe != 0;
}
static size_t graph_cache_max() {                             // SIFTMI_GRAPH_CACHE: smaller caches for the eviction tests
    static const size_t n = [] {
        const char *e = getenv("SIFTMI_GRAPH_CACHE");
        const int v = e ? atoi(e) : 0;
        return (size_t)((v >= 1 && v <= (int)siftmi_ctx::GCACHE_MAX) ? v : 0);
    }();
    return n;                                                 // 0: no override
}
static size_t graph_cache_cap(const siftmi_ctx *c) {
    const size_t forced = graph_cache_max();
    return forced ? forced : std::max(siftmi_ctx::GCACHE_MAX, c->graph_min_cap);
}
static void retire_exec(hipGraphExec_t exec) {
    if (exec && graph_destroy_safe()) (void)hipGraphExecDestroy(exec);
}
// captured graphs hold raw pointers into the context's buffers: drop them (device idle) before any such buffer is replaced
static void drop_graphs(siftmi_ctx *c) {
    if (c->gcache.empty()) return;
    (void)hipDeviceSynchronize();
    for (auto &g : c->gcache) retire_exec(g.exec);
    c->gcache.clear();
    c->gseen.clear();
}

static int ensure_stats(siftmi_ctx *c, int n_frames) {
    if (n_frames <= c->out_frames_cap) return SIFTMI_OK;
    drop_graphs(c);                                       // they write the per-frame statistics block that is replaced below
    if (c->d_out_counts) (void)hipFree(c->d_out_counts);
    if (c->d_stats) (void)hipFree(c->d_stats);
    c->d_out_counts = nullptr; c->d_stats = nullptr; c->out_frames_cap = 0;
    HIP_TRY(hipMalloc((void **)&c->d_out_counts, 2 * (size_t)n_frames * c->n_oct * sizeof(int32_t)));
    HIP_TRY(hipMalloc((void **)&c->d_stats, 5 * (size_t)n_frames * c->n_oct * sizeof(int32_t)));
    c->out_frames_cap = n_frames;
    return SIFTMI_OK;
}

static int check_format(siftmi_ctx *c, int format, size_t row_stride) {
    const size_t bpp = format == SIFTMI_FMT_BGRA8 ? 4 : format == SIFTMI_FMT_GRAY8 ? 1 : format == SIFTMI_FMT_GRAYF32 ? 4 : 0;
    if (!bpp) return set_error(SIFTMI_E_BADARG, "unknown pixel format %d", format);
    if (row_stride < bpp * (size_t)c->cfg.width) return set_error(SIFTMI_E_BADARG, "row_stride %zu smaller than a row", row_stride);
    if ((format == SIFTMI_FMT_BGRA8 || format == SIFTMI_FMT_GRAYF32) && (row_stride & 3))
        return set_error(SIFTMI_E_BADARG, "row_stride must be a multiple of 4 for this format");
    return SIFTMI_OK;
}
