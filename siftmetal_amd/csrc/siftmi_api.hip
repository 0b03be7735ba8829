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
                    c->d_desc_in, c->d_desc, c->d_desc_f32, c->d_counters, c->d_dst_off, c->d_out_kp,
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
                           c->d_desc_f32);
    else {
        if (wpb_desc == 1 && c->cfg.descriptor_patch_lds)
            hipLaunchKernelGGL((descriptor_kernel<false, 1, true>), dim3(wg1, groups), dim3(64), 0, st, P, c->prm, c->d_kp, c->d_desc_in, cnt(c, C_DESC),
                               c->d_desc, c->d_desc_f32);
        else if (wpb_desc == 1)
            hipLaunchKernelGGL((descriptor_kernel<false, 1>), dim3(wg1, groups), dim3(64), 0, st, P, c->prm, c->d_kp, c->d_desc_in, cnt(c, C_DESC), c->d_desc,
                               c->d_desc_f32);
        else
            hipLaunchKernelGGL((descriptor_kernel<false, 4>), dim3(256, groups), dim3(256), 0, st, P, c->prm, c->d_kp, c->d_desc_in, cnt(c, C_DESC), c->d_desc,
                               c->d_desc_f32);
    }
    HIP_TRY(hipGetLastError());
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

// ------------------------------------------------------------------------------------------------
// frame stream (several steps in flight, host-fed staging, host result copies) and the RCCL result exchange
#include "stream_api.hip.h"
#include "exchange_api.hip.h"
