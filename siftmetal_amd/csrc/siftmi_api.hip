// siftmi_api.hip -- C ABI (include/siftmi.h) and host orchestration of the MI355X SIFT path.
//
// Host logic restated from the reference's Swift (never its code):
//   schedule + weights   Sources/SIFTMetal/SIFT/DifferenceOfGaussians.swift:233-344, :69-147
//                        Sources/SIFTMetal/Metal Compute/GaussianKernel.swift:20-43
//   stage order          DifferenceOfGaussians.swift:346-406, SIFT/SIFT.swift:147-238,
//                        SIFT/SIFTOctave.swift:177-492
// Design differences (MI355X-first): frames are processed max_batch at a time in lock-step (one
// launch per stage covers every frame), all lists and counters stay on the device, there is one
// host synchronisation per call instead of 22 per frame, and only the Gaussian stack is
// materialised in HBM (no DoG / gradient textures).
//
// One translation unit, in parts (round 6; each part is included below, in this order):
//   this file                   context, configuration, schedule + weights, buffers, call ordering, timing helpers, experiment knobs
//   launch_sequence.hip.h       kernel form per stage and the launch sequence of one sub-batch (single chain or forked per octave)
//   batch_api.hip.h             sub-batching, hipGraph capture / replay cache, siftmi_detect_describe_batch[_device], siftmi_detect
//   describe_match_api.hip.h    siftmi_describe and the rows after describe: matchers, matchGeometry, index vectors
//   inspect_api.hip.h           stage read-backs, counters, timings, the roofline block's measurement entry points
//   stream_api.hip.h            frame streams (steps in flight, host-fed staging, result sets)
//   exchange_api.hip.h          the RCCL result exchange
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <string>
#include <vector>

#include "../../include/siftmi.h"
#include "dense_kernels.hip.h"
#include "keypoint_kernels.hip.h"
#include "match_kernels.hip.h"
#include "host_post.h"
#include "trie_kernels.hip.h"
#include <rocprim/rocprim.hpp>

using namespace siftmi;

static_assert(sizeof(KeypointRec) == sizeof(siftmi_keypoint) && sizeof(siftmi_keypoint) == 44, "keypoint layout");
static_assert(sizeof(DescriptorRec) == sizeof(siftmi_descriptor) && sizeof(siftmi_descriptor) == 136, "descriptor layout");
static_assert(sizeof(ExtremumRec) == sizeof(siftmi_extremum), "extremum layout");
static_assert(sizeof(siftmi_descriptor_reference) == 524 && sizeof(siftmi_orientation) == 152, "reference layouts");

static thread_local std::string g_last_error;
static int set_error(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return set_error(SIFTMI_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Stage markers for rocprofv3 --marker-trace (replaces the os_signpost intervals of Utilities/Performance.swift:12-20 at its
// call sites SIFT.swift:155,179,192,212,226).  librocprofiler-sdk-roctx is looked up at run time so that the library has
// no link-time dependency on the profiler; without it the ranges are no-ops.
struct RoctxApi {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    RoctxApi() {
        if (getenv("SIFTMI_NO_ROCTX")) return;
        void *h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (!push || !pop) { push = nullptr; pop = nullptr; }
    }
};
static const RoctxApi &roctx() { static RoctxApi api; return api; }
struct StageRange {                            // host-side range around the launches of one stage
    bool on;
    explicit StageRange(const char *name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
    ~StageRange() { if (on) roctx().pop(); }
};

// Streams that only carry PCIe copies (uploads of host frames, copy-back of results).  The runtime multiplexes a process's streams
// onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and a forked launch graph of four octave chains occupies all of
// them: a copy stream that shares a queue with a chain has its barrier packets queued behind that chain's kernels, and the upload
// of sub-batch i + 1 then starts when sub-batch i has FINISHED instead of under it (round 4: 64 x 1080p through 16-frame sub-batches
// 19.7 ms forked against 14.8 ms with the serial graph).  Streams of another priority get hardware queues of their own; creating the
// copy streams at the highest priority is OFF by default (measured: 16.4-16.7 ms forked with it, still behind one chain, and nothing
// for the frame stream's host-fed step), SIFTMI_COPY_STREAM_PRIORITY=1 turns it on.  What ships: host-fed sub-batches stay one chain.
static hipError_t create_copy_stream(hipStream_t *s) {
    static const int mode = [] { const char *e = getenv("SIFTMI_COPY_STREAM_PRIORITY"); return e ? atoi(e) : 0; }();
    int least = 0, greatest = 0;
    if (mode != 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
        return hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest);
    (void)hipGetLastError();
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

struct EventPair { hipEvent_t a, b; int stage; int sub; };     // sub: octave * 8 + layer for the layer blurs, else -1

struct siftmi_ctx {
    siftmi_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    int n_oct = 0, nspo = 0, B = 1;
    // schedule
    int ow[MAX_OCT], oh[MAX_OCT];
    float odelta[MAX_OCT];
    float osigma[MAX_OCT][MAX_NG];
    int seed_taps = 0, taps[MAX_NG];
    TapWeights seed_w, layer_w[MAX_NG];
    // device memory
    float *d_gauss = nullptr;
    size_t frame_stride = 0;                  // floats
    unsigned char *d_input = nullptr;         // staging for host frames: two slots of B frames, filled on copy_stream
    size_t input_bytes = 0;                   //   while the previous sub-batch computes
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_consumed[2] = {nullptr, nullptr};
    int input_slot = 0;
    ExtremumRec *d_ext = nullptr;
    KeypointRec *d_kp_tmp = nullptr, *d_kp = nullptr;
    unsigned long long *d_keys = nullptr, *d_bucket_keys = nullptr;   // refine's sort keys; the same keys in row-bucket order
    int32_t *d_bucket_src = nullptr, *d_row_count = nullptr, *d_row_start = nullptr;   // keypoint sort (kp_row_* kernels)
    unsigned char *d_act = nullptr;           // DoG activity flags [B][octave][nspo][h][ncell] written by the marching blur
    long long march_min_blocks = 800;
    long long chain_max_tiles = 0;            // blur_chain_kernel for octaves of at most this many tiles (0 = never)
    ZeroJob zero_job{nullptr, 0, nullptr, 0}; // counters to clear at the head of a call, handed to the seed tile kernel (run_dense_detect)
    size_t act_off[MAX_OCT] = {0}, act_frame = 0;
    int act_ncell[MAX_OCT] = {0};
    bool act_valid[MAX_OCT] = {false};        // this sub-batch's flags of the octave are complete (all its layers used the marching blur)
    size_t row_table_ints = 0;
    int32_t *d_ori_count = nullptr;
    float *d_ori_angles = nullptr;
    DescInput *d_desc_in = nullptr;
    int32_t *d_desc_flag = nullptr;            // per descriptor: 1 = low-contrast window, the second descriptor launch walks it again (descriptor_kernel, REFINE);
                                              // null unless the schedule has windows wide enough to need it (desc_refine)
    bool desc_refine = false;
    DescriptorRec *d_desc = nullptr;
    float *d_desc_f32 = nullptr;
    int32_t *d_counters = nullptr;            // [5][B*n_oct]: raw, cand, kp, oriented, desc
    int32_t *d_dst_off = nullptr;             // [2][B*n_oct]
    PackState *d_state = nullptr;
    // ctx-owned outputs (host-facing API)
    KeypointRec *d_out_kp = nullptr; long long out_kp_cap = 0;
    DescriptorRec *d_out_desc = nullptr; long long out_desc_cap = 0;
    int32_t *d_out_counts = nullptr, *d_stats = nullptr; int out_frames_cap = 0;
    // host mirrors
    // host result buffers: pinned, grown geometrically, never zero-filled (std::vector::resize would touch every byte)
    template <typename T> struct PinnedBuf {
        T *p = nullptr; size_t cap = 0;
        T *data() const { return p; }
        hipError_t resize(size_t n) {
            if (n <= cap) return hipSuccess;
            if (p) (void)hipHostFree(p);
            p = nullptr; cap = 0;
            const size_t want = n + n / 2;
            hipError_t e = hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault);
            if (e == hipSuccess) cap = want;
            return e;
        }
        void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    };
    PinnedBuf<siftmi_keypoint> h_kp;
    PinnedBuf<siftmi_descriptor> h_desc;
    // siftmi_detect_describe_batch: running totals after every sub-batch (pinned, 4 ints each) + the event that says they have
    // arrived, and the stream the packed records of finished sub-batches are copied back on while later sub-batches compute
    PinnedBuf<int32_t> h_sub;
    std::vector<hipEvent_t> ev_sub;
    hipStream_t d2h_stream = nullptr;
    std::vector<int32_t> h_counts, h_stats;
    std::vector<siftmi_match> h_matches;
    PinnedBuf<siftmi_match> h_match_all;               // one record per source as the kernels leave them (page-locked landing buffer)
    DescriptorRec *d_match_src = nullptr, *d_match_tgt = nullptr; long long match_src_cap = 0, match_tgt_cap = 0;
    MatchRec *d_match_out = nullptr; long long match_out_cap = 0;
    int *d_match_scratch = nullptr; long long match_scratch_cap = 0;
    unsigned char *d_match_sync = nullptr;     // the fused matcher's per-group tickets (u32) and counts (u64): zeroed once, self-resetting / epoch-tagged
    unsigned match_epoch = 0;
    int last_frames = 0;                      // frames of the last batch call
    int last_sub_frames = 0;                  // frames resident in the pyramid
    bool pyramid_valid = false;
    // Ordering between calls: the context's scratch (pyramid, lists, counters) is shared by every entry point, and the
    // device-resident batch call runs on a caller-supplied stream.  Every entry point records ev_last on the stream it
    // used when it has enqueued its work; the next one makes its stream wait for it (a no-op on the same stream) and the
    // host-reading introspection calls synchronise on it.
    hipEvent_t ev_last = nullptr;
    bool last_recorded = false;
    bool stats_on_device = false;             // the last call was device-resident: h_stats is refreshed lazily by siftmi_get_stats
    bool raw_exact = true;                    // raw_extrema of the last call counts every row (no activity-flag skipping)
    PyramidDesc P;
    DetectParams prm;
    // hipGraph cache of the batched device path (one call signature)
    struct GraphKey {
        const void *px; int n_frames, format; size_t row_stride, frame_stride; void *kp; long long kp_cap; void *desc; long long desc_cap;
        void *counts, *totals; hipStream_t st;
        int frame_base, total_frames;              // a sub-batch of the host-fed call (frames frame_base ... of total_frames); 0, n_frames otherwise
        bool fork = false;                         // the captured sequence forks into per-octave chains (fork_chains at capture time)
        bool dense = false;                        // ... and was captured under the stream's density hint (no activity flags)
        bool operator==(const GraphKey &o) const {
            return px == o.px && n_frames == o.n_frames && format == o.format && row_stride == o.row_stride && frame_stride == o.frame_stride &&
                   kp == o.kp && kp_cap == o.kp_cap && desc == o.desc && desc_cap == o.desc_cap && counts == o.counts && totals == o.totals && st == o.st &&
                   frame_base == o.frame_base && total_frames == o.total_frames && fork == o.fork && dense == o.dense;
        }
    };
    // Captured launch sequences, most recently used last; up to GCACHE_MAX call signatures per context, the least recently
    // used one is evicted for a new one (retire_exec()).
    struct GraphEntry { GraphKey key; hipGraphExec_t exec; bool raw_exact; };   // raw_exact: what run_dense_detect decided while the sequence was captured
    std::vector<GraphEntry> gcache;
    std::vector<GraphKey> gseen;               // signatures seen once (not yet captured), oldest first
    static constexpr size_t GCACHE_MAX = 64;
    size_t graph_min_cap = 0;                  // a host-fed call of n sub-batches needs n signatures alive at once (siftmi_detect_describe_batch)
    bool graph_failed = false;
    // what the batched entry points did with their launch sequences (siftmi_graph_stats): captured, replayed, issued as direct launches
    int64_t n_graph_captures = 0, n_graph_replays = 0, n_direct_sequences = 0;
    bool last_replayed = false, last_forked = false;   // the last sequence: came from a captured graph / forked into per-octave chains
    // fork/join of the octave chains inside a captured graph (small launches only, see run_dense_detect)
    hipStream_t oct_stream[MAX_OCT] = {};
    hipEvent_t ev_fork[MAX_OCT] = {}, ev_join[MAX_OCT] = {};
    bool fork_ready = false;
    // Set by the frame stream from the descriptor totals of an earlier step (siftmi_stream_*): frames dense with keypoints run 2-3 %
    // faster on ONE chain (their keypoint stages are half the step and pair better with the other context's whole pyramid than with
    // their own octaves' blurs), sparse ones 4 % faster forked.  Only consulted when cfg.graph_fork == 0 (automatic).
    bool dense_hint = false;
    hipStream_t tstream = nullptr;            // stream the timing events are recorded on
    // timings
    bool timing = false;
    std::vector<EventPair> pending, pool;
    double t_ms[SIFTMI_T_COUNT];
    int64_t t_launches[SIFTMI_T_COUNT];
    double t_blur_ms[MAX_OCT][MAX_NG];             // the SIFTMI_T_BLUR time split by (octave, layer): one kernel name and grid each
    int64_t t_blur_launches[MAX_OCT][MAX_NG];
};

// ------------------------------------------------------------------------------------------------
// GaussianKernel.swift:20-43 / GaussianSeriesKernel.swift:27-51
static int gaussian_weights(float s, TapWeights &out) {
    const int radius = (int)std::ceil(4.0f * s);
    const int size = radius * 2 + 1;
    if (size > 31) return -1;
    float t = 0.0f;
    const float ss = s * s;
    for (int k = -radius, i = 0; k <= radius; k++, i++) {
        const float kk = (float)(k * k);
        const float w = std::exp(-0.5f * (kk / ss));
        out.w[i] = w;
        t += w;
    }
    for (int i = 0; i < size; i++) out.w[i] = out.w[i] / t;
    for (int i = size; i < 32; i++) out.w[i] = 0.0f;
    // k enters only as k * k, so w[i] == w[size - 1 - i] bit for bit; blur_ring_kernel keeps radius + 1 of them (VTapsSym)
    for (int i = 0; i < size; i++) if (memcmp(&out.w[i], &out.w[size - 1 - i], sizeof(float)) != 0) return -1;
    return size;
}

extern "C" int siftmi_default_config(siftmi_config *cfg, int32_t width, int32_t height) {
    if (!cfg) return set_error(SIFTMI_E_BADARG, "cfg is null");
    memset(cfg, 0, sizeof(*cfg));
    cfg->width = width; cfg->height = height;
    cfg->n_octaves = 7;                 // DifferenceOfGaussians.swift:41
    cfg->nspo = 3;                      // :46
    cfg->sigma_min = 0.8f; cfg->delta_min = 0.5f; cfg->sigma_in = 0.5f;   // :28-37
    cfg->dog_threshold = 0.0133f;       // SIFTOctave.swift:218
    cfg->edge_threshold = 10.0f;        // :224
    cfg->max_iterations = 5;            // :219
    cfg->max_offset = 0.6f;             // :220
    cfg->image_border = 5;              // SIFTInterpolate.metal:182
    cfg->lambda_orientation = 1.5f;     // SIFTOctave.swift:298
    cfg->orientation_threshold = 0.8f;  // :299
    cfg->orientation_smoothing = 6;     // SIFTOrientation.metal:167
    cfg->descriptor_scales_per_octave = 3;   // SIFTOctave.swift:398
    cfg->full_neighbourhood = 0;
    cfg->max_batch = 1;
    cfg->use_hip_graph = 1;
    cfg->count_raw_extrema = 0;
    cfg->blur_march_min_blocks = 800;
    return SIFTMI_OK;
}

extern "C" const char *siftmi_last_error(void) { return g_last_error.c_str(); }

extern "C" int siftmi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static void retire_exec(hipGraphExec_t exec);

static void free_ctx(siftmi_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();             // nothing of this context may still be running when its memory and graphs go
    void *ptrs[] = {c->d_gauss, c->d_input, c->d_ext, c->d_kp_tmp, c->d_kp, c->d_keys, c->d_bucket_keys, c->d_bucket_src, c->d_row_count,
                    c->d_row_start, c->d_act, c->d_ori_count, c->d_ori_angles,
                    c->d_desc_in, c->d_desc_flag, c->d_desc, c->d_desc_f32, c->d_counters, c->d_dst_off, c->d_out_kp,
                    c->d_out_desc, c->d_out_counts, c->d_stats, c->d_match_src, c->d_match_tgt, c->d_match_out, c->d_match_scratch, c->d_match_sync};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (auto &e : c->pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto &e : c->pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto &g : c->gcache) retire_exec(g.exec);   // (device idle: synchronised above)
    c->gcache.clear();
    for (int i = 0; i < MAX_OCT; i++) {
        if (c->ev_fork[i]) (void)hipEventDestroy(c->ev_fork[i]);
        if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
        if (c->oct_stream[i]) (void)hipStreamDestroy(c->oct_stream[i]);
    }
    c->h_kp.release(); c->h_desc.release(); c->h_sub.release(); c->h_match_all.release();
    for (hipEvent_t e : c->ev_sub) if (e) (void)hipEventDestroy(e);
    if (c->d2h_stream) (void)hipStreamDestroy(c->d2h_stream);
    for (int i = 0; i < 2; i++) {
        if (c->ev_copied[i]) (void)hipEventDestroy(c->ev_copied[i]);
        if (c->ev_consumed[i]) (void)hipEventDestroy(c->ev_consumed[i]);
    }
    if (c->ev_last) (void)hipEventDestroy(c->ev_last);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" void siftmi_destroy(siftmi_ctx *ctx) { free_ctx(ctx); }

extern "C" int siftmi_create(const siftmi_config *cfg, int hip_device, siftmi_ctx **out) {
    if (!cfg || !out) return set_error(SIFTMI_E_BADARG, "null argument");
    *out = nullptr;
    if (cfg->width < 1 || cfg->height < 1 || cfg->width > 32768 || cfg->height > 32768)
        return set_error(SIFTMI_E_BADARG, "input size %dx%d out of range [1, 32768]", cfg->width, cfg->height);
    if (cfg->n_octaves < 1 || cfg->n_octaves > SIFTMI_MAX_OCTAVES)
        return set_error(SIFTMI_E_BADARG, "n_octaves %d out of range [1, %d]", cfg->n_octaves, SIFTMI_MAX_OCTAVES);
    if (cfg->nspo < 1 || cfg->nspo + 3 > MAX_NG) return set_error(SIFTMI_E_BADARG, "nspo %d out of range [1, %d]", cfg->nspo, MAX_NG - 3);
    if (cfg->delta_min != 0.5f) return set_error(SIFTMI_E_BADARG, "delta_min must be 0.5 (2x seed image)");
    if (cfg->max_batch < 1 || cfg->max_batch > 4096) return set_error(SIFTMI_E_BADARG, "max_batch %d out of range", cfg->max_batch);
    if (cfg->descriptor_patch_lds != 0 && cfg->descriptor_patch_lds != 1) return set_error(SIFTMI_E_BADARG, "descriptor_patch_lds must be 0 or 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return set_error(SIFTMI_E_NODEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (hip_device < 0 || hip_device >= ndev) return set_error(SIFTMI_E_BADARG, "hip_device %d out of range [0, %d)", hip_device, ndev);

    siftmi_ctx *c = new siftmi_ctx();
    c->cfg = *cfg;
    c->device = hip_device;
    c->n_oct = cfg->n_octaves; c->nspo = cfg->nspo; c->B = cfg->max_batch;
    c->march_min_blocks = cfg->blur_march_min_blocks > 0 ? cfg->blur_march_min_blocks : 800;
    c->chain_max_tiles = cfg->blur_chain_max_tiles > 0 ? cfg->blur_chain_max_tiles : (cfg->blur_chain_max_tiles == 0 ? 256 : 0);
    memset(c->t_ms, 0, sizeof(c->t_ms)); memset(c->t_launches, 0, sizeof(c->t_launches));
    memset(c->t_blur_ms, 0, sizeof(c->t_blur_ms)); memset(c->t_blur_launches, 0, sizeof(c->t_blur_launches));
    const int W = cfg->width, H = cfg->height, nspo = cfg->nspo, NG = nspo + 3;

    // --- schedule: DifferenceOfGaussians.swift:235-238, 255-262, 315-328, Octave.init :91-102
    {
        const float i = cfg->sigma_min * cfg->sigma_min, j = cfg->sigma_in * cfg->sigma_in;
        const float k = std::sqrt(i - j) / cfg->delta_min;
        c->seed_taps = gaussian_weights(k, c->seed_w);
        if (c->seed_taps < 0) { free_ctx(c); return set_error(SIFTMI_E_BADARG, "seed blur needs more than 31 taps"); }
    }
    size_t off = 0, ext_off = 0, kp_off = 0, desc_off = 0;
    memset(&c->P, 0, sizeof(c->P));
    for (int o = 0; o < c->n_oct; o++) {
        const float delta = cfg->delta_min * std::pow(2.0f, (float)o);
        c->odelta[o] = delta;
        c->ow[o] = (int)((float)W / delta);
        c->oh[o] = (int)((float)H / delta);
        if (c->ow[o] < 1 || c->oh[o] < 1) {
            free_ctx(c);
            return set_error(SIFTMI_E_BADARG, "octave %d of a %dx%d input is empty; use at most %d octaves", o, W, H, o);
        }
        for (int s = 0; s < NG; s++) {
            const float hh = delta / cfg->delta_min;
            const float ii = (float)s / (float)nspo;
            const float jj = std::pow(2.0f, ii);
            c->osigma[o][s] = hh * cfg->sigma_min * jj;
        }
        if (o == 0) {
            for (int s = 1; s < NG; s++) {
                const float sa = c->osigma[0][s - 1], sb = c->osigma[0][s];
                const float rho = std::sqrt(sb * sb - sa * sa) / delta;
                c->taps[s - 1] = gaussian_weights(rho, c->layer_w[s - 1]);
                if (c->taps[s - 1] < 0) { free_ctx(c); return set_error(SIFTMI_E_BADARG, "layer %d blur needs more than 31 taps", s); }
            }
        }
        const size_t n = (size_t)c->ow[o] * c->oh[o];
        c->P.oct_offset[o] = off;
        off += (n * NG + 3) & ~(size_t)3;
        c->P.w[o] = c->ow[o]; c->P.h[o] = c->oh[o]; c->P.delta[o] = delta;
        c->P.sigma0[o] = c->osigma[o][0]; c->P.sigma1[o] = c->osigma[o][1];
        for (int s = 0; s < NG; s++) c->P.sigmas[o][s] = c->osigma[o][s];
        auto clampi = [](long long v, long long lo, long long hi) { return (int)std::max(lo, std::min(hi, v)); };
        c->P.cap_ext[o] = cfg->max_extrema > 0 ? cfg->max_extrema : clampi((long long)n / 32, 4096, 1 << 20);
        c->P.cap_kp[o] = cfg->max_keypoints > 0 ? cfg->max_keypoints : clampi((long long)n / 64, 4096, 1 << 19);
        c->P.cap_desc[o] = cfg->max_descriptors > 0 ? cfg->max_descriptors : (c->P.cap_kp[o] + c->P.cap_kp[o] / 2);
        c->P.ext_off[o] = ext_off; ext_off += c->P.cap_ext[o];
        c->P.kp_off[o] = kp_off; kp_off += c->P.cap_kp[o];
        c->P.desc_off[o] = desc_off; desc_off += c->P.cap_desc[o];
    }
    // the keypoint sort packs (scale * h + y) * w + x of octave 0 into 32 bits (refine_kernel / kp_row_* kernels)
    if ((long long)(nspo + 2) * c->ow[0] * c->oh[0] >= (1ll << 31)) {
        free_ctx(c);
        return set_error(SIFTMI_E_BADARG, "input %dx%d too large: (nspo + 2) x octave-0 pixels must stay below 2^31", W, H);
    }
    c->frame_stride = off;
    c->P.frame_stride = off; c->P.n_octaves = c->n_oct; c->P.nspo = nspo; c->P.only_octave = -1;
    c->P.ext_frame = ext_off; c->P.kp_frame = kp_off; c->P.desc_frame = desc_off;
    {
        size_t row_off = 0;
        for (int o = 0; o < c->n_oct; o++) { c->P.row_off[o] = row_off; row_off += (size_t)(c->P.nspo + 2) * c->P.h[o]; }
        c->P.row_frame = row_off;
    }
    c->prm.dog_threshold = cfg->dog_threshold; c->prm.edge_threshold = cfg->edge_threshold;
    c->prm.max_offset = cfg->max_offset; c->prm.max_iterations = cfg->max_iterations;
    c->prm.border = cfg->image_border; c->prm.full_neighbourhood = cfg->full_neighbourhood;
    c->prm.lambda_ori = cfg->lambda_orientation; c->prm.ori_threshold = cfg->orientation_threshold;
    c->prm.ori_smoothing = cfg->orientation_smoothing; c->prm.desc_scales_per_octave = cfg->descriptor_scales_per_octave;

    hipError_t e = hipSetDevice(hip_device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = create_copy_stream(&c->copy_stream);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_last, hipEventDisableTiming);
    for (int i = 0; i < 2; i++) {
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_copied[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_consumed[i], hipEventDisableTiming);
    }
    const size_t B = (size_t)c->B, G = B * c->n_oct;
    auto alloc = [&](void **p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, std::max<size_t>(bytes, 16)); };
    // + 64 B: the quad walks of the orientation / descriptor kernels read up to 8 B past a row's last candidate through the SGPR row
    // offset, which the buffer resource's range check does not see (it bounds the VGPR offset).  A described layer is never the stack's
    // last (scale <= nspo of nspo + 3 layers), so such a read lands in the next layer -- or, for the last frame's last octave at most,
    // in this tail; the values are never used.
    alloc((void **)&c->d_gauss, B * off * sizeof(float) + 64);
    c->input_bytes = (size_t)W * H * 4;
    alloc((void **)&c->d_input, 2 * B * c->input_bytes);
    alloc((void **)&c->d_ext, B * ext_off * sizeof(ExtremumRec));
    alloc((void **)&c->d_kp_tmp, B * kp_off * sizeof(KeypointRec));
    alloc((void **)&c->d_kp, B * kp_off * sizeof(KeypointRec));
    alloc((void **)&c->d_keys, B * kp_off * sizeof(unsigned long long));
    alloc((void **)&c->d_bucket_keys, B * kp_off * sizeof(unsigned long long));
    alloc((void **)&c->d_bucket_src, B * kp_off * sizeof(int32_t));
    {
        size_t off_b = 0;
        for (int o = 0; o < c->n_oct; o++) {
            c->act_ncell[o] = (c->ow[o] + 63) / 64;
            c->act_off[o] = off_b;
            off_b += (size_t)c->nspo * c->oh[o] * c->act_ncell[o];
        }
        c->act_frame = (off_b + 15) & ~(size_t)15;
    }
    alloc((void **)&c->d_act, B * c->act_frame);
    c->row_table_ints = B * c->P.row_frame;
    alloc((void **)&c->d_row_count, c->row_table_ints * sizeof(int32_t));
    alloc((void **)&c->d_row_start, c->row_table_ints * sizeof(int32_t));
    alloc((void **)&c->d_ori_count, B * kp_off * sizeof(int32_t));
    alloc((void **)&c->d_ori_angles, B * kp_off * ORI_BINS * sizeof(float));
    alloc((void **)&c->d_desc_in, B * desc_off * sizeof(DescInput));
    {   // the widest descriptor window of this schedule: hw = 3 sigma 2^(interval / descriptor_scales_per_octave), interval < nspo + 1 (make_desc_input);
        // from hw ~ 17.4 on a window's fixed-point unit is 2^-22 and low-contrast windows get the second descriptor pass (descriptor_kernel, REFINE)
        const float hw_max = 3.0f * 1.6f * powf(2.0f, ((float)c->nspo + 1.0f) / (float)std::max(1, cfg->descriptor_scales_per_octave));
        c->desc_refine = hw_max >= 17.0f;
    }
    if (c->desc_refine) alloc((void **)&c->d_desc_flag, B * desc_off * sizeof(int32_t));
    alloc((void **)&c->d_desc, B * desc_off * sizeof(DescriptorRec));
    if (cfg->keep_descriptor_floats) alloc((void **)&c->d_desc_f32, B * desc_off * DESC_N * sizeof(float));
    alloc((void **)&c->d_counters, 5 * G * sizeof(int32_t) + sizeof(PackState));
    if (e == hipSuccess) c->d_state = (PackState *)(c->d_counters + 5 * G);
    alloc((void **)&c->d_dst_off, 2 * G * sizeof(int32_t));
    // d_state lives right behind the counters (same allocation) so that one kernel clears both at the start of a call
    if (e != hipSuccess) {
        const int code = (e == hipErrorOutOfMemory) ? SIFTMI_E_NOMEM : SIFTMI_E_HIP;
        set_error(code, "context allocation failed: %s", hipGetErrorString(e));
        free_ctx(c);
        return code;
    }
    c->P.gauss = c->d_gauss;
    *out = c;
    return SIFTMI_OK;
}

// ------------------------------------------------------------------------------------------------
// ordering of a call on stream `st` after the previous call on this context (see siftmi_ctx::ev_last)
static int order_begin(siftmi_ctx *c, hipStream_t st) {
    if (c->last_recorded) HIP_TRY(hipStreamWaitEvent(st, c->ev_last, 0));
    return SIFTMI_OK;
}
static int order_end(siftmi_ctx *c, hipStream_t st) {
    HIP_TRY(hipEventRecord(c->ev_last, st));
    c->last_recorded = true;
    return SIFTMI_OK;
}
// host-side reads of the context's device state: everything enqueued by the last call has finished
static int order_sync(siftmi_ctx *c) {
    if (c->last_recorded) HIP_TRY(hipEventSynchronize(c->ev_last));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SIFTMI_OK;
}

// ------------------------------------------------------------------------------------------------
// timing helpers
static void t_begin(siftmi_ctx *c, int stage, int sub = -1) {
    if (!c->timing) return;
    EventPair ep;
    if (!c->pool.empty()) { ep = c->pool.back(); c->pool.pop_back(); }
    else { (void)hipEventCreate(&ep.a); (void)hipEventCreate(&ep.b); }
    ep.stage = stage;
    ep.sub = sub;
    (void)hipEventRecord(ep.a, c->tstream ? c->tstream : c->stream);
    c->pending.push_back(ep);
}
static void t_end(siftmi_ctx *c) {
    if (!c->timing) return;
    (void)hipEventRecord(c->pending.back().b, c->tstream ? c->tstream : c->stream);
}
static void t_collect(siftmi_ctx *c) {
    for (auto &ep : c->pending) {
        float ms = 0.0f;
        if (hipEventSynchronize(ep.b) == hipSuccess && hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) {
            c->t_ms[ep.stage] += ms;
            c->t_launches[ep.stage] += 1;
            if (ep.stage == SIFTMI_T_BLUR && ep.sub >= 0) { c->t_blur_ms[ep.sub >> 4][ep.sub & 15] += ms; c->t_blur_launches[ep.sub >> 4][ep.sub & 15] += 1; }
        }
        c->pool.push_back(ep);
    }
    c->pending.clear();
}

// ------------------------------------------------------------------------------------------------
// Tuning knobs of the measurement tools (tools/README.md: chunk heights, fork widths, workgroup shapes ...).  They are read from the
// environment ONLY by builds with -DSIFTMI_EXPERIMENT (tools/build_variant.sh); the shipped library compiles every one of them to its
// default.  (Rounds 2-5 spelled each as its own #ifdef block; the two probes that were more than a knob -- a per-sub-batch timeline of the
// host-buffer call and a one-phase-per-call split of the step -- are tools/experiments/api_probes_r05.diff.)
#ifdef SIFTMI_EXPERIMENT
static long long exp_knob(const char *name, long long dflt) { const char *e = getenv(name); return e ? atoll(e) : dflt; }
static bool exp_set(const char *name) { return getenv(name) != nullptr; }
#else
static constexpr long long exp_knob(const char *, long long dflt) { return dflt; }
static constexpr bool exp_set(const char *) { return false; }
#endif

#include "launch_sequence.hip.h"
#include "batch_api.hip.h"
#include "describe_match_api.hip.h"
#include "inspect_api.hip.h"

// ------------------------------------------------------------------------------------------------
// frame stream (several steps in flight, host-fed staging, host result copies) and the RCCL result exchange
#include "stream_api.hip.h"
#include "exchange_api.hip.h"
