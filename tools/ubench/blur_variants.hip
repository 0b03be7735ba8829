// Standalone harness: times Gaussian-layer blur kernel variants on 8 x 3840x2160 f32 frames and checks
// every variant bit-for-bit against a naive kernel (same tap order, fmaf).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I../../siftmetal_amd/csrc -o blur_variants blur_variants.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "dense_kernels.hip.h"
#include "blur_ws.hip.h"
#include "blur_dma.hip.h"
using namespace siftmi;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void naive_x(const float *in, float *out, int w, int h, TapWeights wt, int n) {
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y; if (x >= w) return;
    in += (size_t)blockIdx.z * w * h; out += (size_t)blockIdx.z * w * h;
    float s = 0; int o = x - n / 2;
    for (int i = 0; i < n; i++) { int xx = symm(o + i, w); s = fmaf(wt.w[i], in[(size_t)y * w + xx], s); }
    out[(size_t)y * w + x] = s;
}
__global__ void naive_y(const float *in, float *out, int w, int h, TapWeights wt, int n) {
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y; if (x >= w) return;
    in += (size_t)blockIdx.z * w * h; out += (size_t)blockIdx.z * w * h;
    float s = 0; int o = y - n / 2;
    for (int i = 0; i < n; i++) { int yy = symm(o + i, h); s = fmaf(wt.w[i], in[(size_t)yy * w + x], s); }
    out[(size_t)y * w + x] = s;
}

static int gw(float s, TapWeights &out) {
    int radius = (int)std::ceil(4.0f * s), size = 2 * radius + 1; float t = 0, ss = s * s;
    for (int k = -radius, i = 0; k <= radius; k++, i++) { float w = std::exp(-0.5f * ((float)(k * k) / ss)); out.w[i] = w; t += w; }
    for (int i = 0; i < size; i++) out.w[i] /= t;
    for (int i = size; i < 32; i++) out.w[i] = 0;
    return size;
}

struct Ctx { float *src, *dst, *ref, *tmp; void *diag; int w, h, nf; size_t n; };

static const char *g_filter = nullptr; static int g_rfilter = 0;
template <typename F>
static void run_variant(const char *name, Ctx &c, int R, F launch) {
    if (g_filter && !strstr(name, g_filter)) return;
    if (g_rfilter && R != g_rfilter) return;
    CHECK(hipMemset(c.dst, 0xff, c.n * c.nf * 4));
    launch();
    CHECK(hipDeviceSynchronize());
    std::vector<float> a(c.n), b(c.n);
    size_t bad = 0;
    for (int f : {0, c.nf - 1}) {
        CHECK(hipMemcpy(a.data(), c.dst + (size_t)f * c.n, c.n * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(b.data(), c.ref + (size_t)f * c.n, c.n * 4, hipMemcpyDeviceToHost));
        bad += memcmp(a.data(), b.data(), c.n * 4) != 0;
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) launch();
    hipEventRecord(e0);
    const int iters = 20;
    for (int i = 0; i < iters; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    printf("  %-34s R=%2d taps=%2d  %.4f ms  %7.1f GB/s  %s\n", name, R, 2 * R + 1, ms, 8.0 * c.n * c.nf / (ms * 1e-3) / 1e9, bad ? "MISMATCH" : "exact");
    fflush(stdout);
}

template <int R>
static void bench_R(Ctx &c, float rho) {
    TapWeights wt; int n = gw(rho, wt);
    if (n != 2 * R + 1) { printf("rho %f gives %d taps, expected %d\n", rho, n, 2 * R + 1); return; }
    dim3 g((c.w + 255) / 256, c.h, c.nf);
    hipLaunchKernelGGL(naive_x, g, dim3(256), 0, 0, c.src, c.tmp, c.w, c.h, wt, n);
    hipLaunchKernelGGL(naive_y, g, dim3(256), 0, 0, c.tmp, c.ref, c.w, c.h, wt, n);
    CHECK(hipDeviceSynchronize());
    SeedSource none; memset(&none, 0, sizeof(none)); Decimate nodec; memset(&nodec, 0, sizeof(nodec));
    Activity noact{nullptr, 0, 0, 0.0f};
    // tile form (small octaves, single frames)
#define V2X(TH_, NTHR_, RB_, MINW_, KCH_) { using G = Blur2Geom<R, TH_, NTHR_, 4, RB_>; \
        const int total = ((c.w + G::TW - 1) / G::TW) * ((c.h + G::TH - 1) / G::TH) * c.nf; \
        dim3 grid(((total + 7) / 8) * 8, 1, 1); \
        run_variant("tile TH=" #TH_ " RB=" #RB_ " XCD", c, R, [&] { hipLaunchKernelGGL((blur2_kernel<R, TH_, NTHR_, 4, RB_, false, MINW_, KCH_, true>), grid, dim3(NTHR_), G::lds_bytes, 0, c.src, c.dst, c.w, c.h, c.n, c.n, wt, none, c.nf, nodec, noact, ZeroJob{nullptr, 0, nullptr, 0}); }); }
    V2X(32, 256, 4, 1, 0)
    // round 2: ring form
#define VR(S_, CHR_, MINW_, DBG_) { using G = RingGeom<R, S_>; \
        const int tx = (c.w + G::TW - 1) / G::TW, nch = (c.h + CHR_ - 1) / CHR_; \
        const int total = tx * nch * c.nf; dim3 grid(((total + 7) / 8) * 8, 1, 1); \
        Activity dbg{(unsigned char *)c.diag, 0, 0, 0.0f}; \
        if ((DBG_) & 1) CHECK(hipMemset(c.diag, 0, (size_t)grid.x * 4 * 8 * 8)); \
        run_variant("ring S=" #S_ " rows/chunk=" #CHR_ " minw=" #MINW_ " dbg=" #DBG_, c, R, [&] { hipLaunchKernelGGL((blur_ring_kernel<R, MINW_, S_, false, false, DBG_>), grid, dim3(256), G::lds_bytes, 0, c.src, c.dst, c.w, c.h, c.n, c.n, wt, c.nf, CHR_, nodec, dbg, none); }); \
        if ((DBG_) & 1) { std::vector<unsigned long long> d((size_t)total * 32); CHECK(hipMemcpy(d.data(), c.diag, d.size() * 8, hipMemcpyDeviceToHost)); \
            double sum[8] = {0}; for (size_t i = 0; i < d.size(); i++) sum[i & 7] += (double)d[i]; double tot = 0; for (int k = 0; k < 8; k++) tot += sum[k]; \
            const char *nm[8] = {"issue", "H", "B2wait", "V", "stores", "B3wait", "vmwait+ldsw", "B1wait"}; \
            printf("      stamps (cycles per wave-step, share):"); for (int k = 0; k < 8; k++) printf(" %s %.0f (%.0f%%)", nm[k], sum[k] / ((double)tx * c.nf * ((c.h + S_ - 1) / S_) * 4) , 100.0 * sum[k] / tot); printf("\n"); } }
    VR(32, 128, 4, 0) VR(32, 256, 4, 0) VR(32, 256, 4, 24)
    // round 5: 16-row steps, two wavefronts per workgroup (18 KB ring: eight workgroups per CU, cheaper barriers); R <= 8 only (ring = 2 S rows)
#define VRS(S_, CHR_, MINW_, DBG_) if constexpr (2 * R <= S_) { using G = RingGeom<R <= S_ / 2 ? R : 1, S_>; \
        const int tx = (c.w + G::TW - 1) / G::TW, nch = (c.h + CHR_ - 1) / CHR_; \
        const int total = tx * nch * c.nf; dim3 grid(((total + 7) / 8) * 8, 1, 1); \
        Activity dbg{(unsigned char *)c.diag, 0, 0, 0.0f}; \
        run_variant("ring S=" #S_ " (" "2 wavefronts) rows/chunk=" #CHR_ " minw=" #MINW_ " dbg=" #DBG_, c, R, [&] { hipLaunchKernelGGL((blur_ring_kernel<(R <= S_ / 2 ? R : 1), MINW_, S_, false, false, DBG_>), grid, dim3(8 * S_), G::lds_bytes, 0, c.src, c.dst, c.w, c.h, c.n, c.n, wt, c.nf, CHR_, nodec, dbg, none); }); }
    VRS(16, 256, 4, 0) VRS(16, 512, 4, 0) VRS(16, 256, 4, 24)
#define VRH4(S_, CHR_, MINW_) { using G = RingGeom<R, S_>; \
        const int tx = (c.w + G::TW - 1) / G::TW, nch = (c.h + CHR_ - 1) / CHR_; \
        const int total = tx * nch * c.nf; dim3 grid(((total + 7) / 8) * 8, 1, 1); \
        run_variant("ring S=" #S_ " rows/chunk=" #CHR_ " minw=" #MINW_ " H4 (4 outputs per lane, no swizzle)", c, R, [&] { hipLaunchKernelGGL((blur_ring_kernel<R, MINW_, S_, false, false, 0, -1, false>), grid, dim3(256), G::lds_bytes, 0, c.src, c.dst, c.w, c.h, c.n, c.n, wt, c.nf, CHR_, nodec, noact, none); }); }
    VRH4(32, 256, 4)
#define VRX(S_, CHR_, MINW_, H8_) { using G = RingGeom<R, S_>; \
        const int tx = (c.w + G::TW - 1) / G::TW, nch = (c.h + CHR_ - 1) / CHR_; \
        const int total = tx * nch * c.nf; dim3 grid(((total + 7) / 8) * 8, 1, 1); \
        run_variant("ring S=" #S_ " rows/chunk=" #CHR_ " minw=" #MINW_ " H8=" #H8_, c, R, [&] { hipLaunchKernelGGL((blur_ring_kernel<R, MINW_, S_, false, false, 0, -1, H8_>), grid, dim3(256), G::lds_bytes, 0, c.src, c.dst, c.w, c.h, c.n, c.n, wt, c.nf, CHR_, nodec, noact, none); }); }
    VRX(32, 256, 4, true) VRX(32, 256, 3, true) VRX(32, 256, 3, false) VRX(32, 128, 3, true)
    // round 3: the horizontal pass software-pipelined over two items (HPIPE), 3 workgroups per CU (168 VGPRs)
#define VRP(S_, CHR_, MINW_, H8_) { using G = RingGeom<R, S_>; \
        const int tx = (c.w + G::TW - 1) / G::TW, nch = (c.h + CHR_ - 1) / CHR_; \
        const int total = tx * nch * c.nf; dim3 grid(((total + 7) / 8) * 8, 1, 1); \
        run_variant("ring S=" #S_ " rows/chunk=" #CHR_ " minw=" #MINW_ " H8=" #H8_ " HPIPE", c, R, [&] { hipLaunchKernelGGL((blur_ring_kernel<R, MINW_, S_, false, false, 0, -1, H8_, true>), grid, dim3(256), G::lds_bytes, 0, c.src, c.dst, c.w, c.h, c.n, c.n, wt, c.nf, CHR_, nodec, noact, none); }); }
    VRP(32, 256, 3, true) VRP(32, 256, 2, true)
    // (round 3's vertical pass on the matrix cores needs tools/experiments/blur_mfma_vertical_r03.diff applied to dense_kernels.hip.h:
    // build with -DBLUR_VARIANTS_WITH_VM after `patch -p0 < tools/experiments/blur_mfma_vertical_r03.diff`)
#ifdef BLUR_VARIANTS_WITH_VM
    // round 3: the vertical pass on the matrix cores (VM): v_mfma_f32_4x4x1 (16 blocks) per window row against the banded tap matrix
#define VRM(S_, CHR_, MINW_, H8_, HP_, VM_) { using G = RingGeom<R, S_>; \
        const int tx = (c.w + G::TW - 1) / G::TW, nch = (c.h + CHR_ - 1) / CHR_; \
        const int total = tx * nch * c.nf; dim3 grid(((total + 7) / 8) * 8, 1, 1); \
        run_variant("ring VM=" #VM_ " (MFMA vertical pass) rows/chunk=" #CHR_ " minw=" #MINW_ " H8=" #H8_ " HPIPE=" #HP_, c, R, [&] { hipLaunchKernelGGL((blur_ring_kernel<R, MINW_, S_, false, false, 0, -1, H8_, HP_, VM_>), grid, dim3(256), G::lds_bytes, 0, c.src, c.dst, c.w, c.h, c.n, c.n, wt, c.nf, CHR_, nodec, noact, none); }); }
    VRM(32, 256, 4, (R <= 12), false, 1) VRM(32, 256, 3, true, false, 1) VRM(32, 256, 4, (R <= 12), false, 2) VRM(32, 256, 3, true, false, 2) VRM(32, 256, 3, true, true, 2) VRM(32, 128, 3, true, false, 2)
#endif
#define VRMD(S_, CHR_, MINW_, H8_, VM_) { using G = RingGeom<R, S_>; \
        const int tx = (c.w + G::TW - 1) / G::TW, nch = (c.h + CHR_ - 1) / CHR_; \
        const int total = tx * nch * c.nf; dim3 grid(((total + 7) / 8) * 8, 1, 1); \
        Activity dbg{(unsigned char *)c.diag, 0, 0, 0.0f}; \
        CHECK(hipMemset(c.diag, 0, (size_t)grid.x * 4 * 8 * 8)); \
        run_variant("stamps rows/chunk=" #CHR_ " minw=" #MINW_ " H8=" #H8_ " VM=" #VM_, c, R, [&] { hipLaunchKernelGGL((blur_ring_kernel<R, MINW_, S_, false, false, 1, -1, H8_, false>), grid, dim3(256), G::lds_bytes, 0, c.src, c.dst, c.w, c.h, c.n, c.n, wt, c.nf, CHR_, nodec, dbg, none); }); \
        { std::vector<unsigned long long> d((size_t)total * 32); CHECK(hipMemcpy(d.data(), c.diag, d.size() * 8, hipMemcpyDeviceToHost)); \
            double sum[8] = {0}; for (size_t i = 0; i < d.size(); i++) sum[i & 7] += (double)d[i]; double tot = 0; for (int k = 0; k < 8; k++) tot += sum[k]; \
            const char *nm[8] = {"issue", "H", "B2wait", "V", "stores", "B3wait", "vmwait+ldsw", "B1wait"}; \
            printf("      stamps (cycles per wave-step, share):"); for (int k = 0; k < 8; k++) printf(" %s %.0f (%.0f%%)", nm[k], sum[k] / ((double)tx * c.nf * ((c.h + S_ - 1) / S_) * 4 * 24) , 100.0 * sum[k] / tot); printf("\n"); } }
    VRMD(32, 256, 4, (R <= 12), 0)
    // occupancy sensitivity: the same kernel with 12 KB of unused dynamic LDS (3 instead of 4 workgroups per CU at R <= 8, 2 instead of 3 above)
#define VRL(S_, CHR_, MINW_) { using G = RingGeom<R, S_>; \
        const int tx = (c.w + G::TW - 1) / G::TW, nch = (c.h + CHR_ - 1) / CHR_; \
        const int total = tx * nch * c.nf; dim3 grid(((total + 7) / 8) * 8, 1, 1); \
        run_variant("ring S=" #S_ " rows/chunk=" #CHR_ " minw=" #MINW_ " +12KB LDS (one workgroup per CU fewer)", c, R, [&] { hipLaunchKernelGGL((blur_ring_kernel<R, MINW_, S_, false, false, 0>), grid, dim3(256), G::lds_bytes + 12288, 0, c.src, c.dst, c.w, c.h, c.n, c.n, wt, c.nf, CHR_, nodec, noact, none); }); }
    VRL(32, 256, 4)
    // round 3 experiment: wave-specialised ring (blur_ws.hip.h), R >= 9 only
    if constexpr (R >= 8) {
#define VWS(CHR_) { const int tx = c.w / 128, nch = (c.h + CHR_ - 1) / CHR_; \
        const int total = tx * nch * c.nf; dim3 grid(((total + 7) / 8) * 8, 1, 1); \
        run_variant("ring WS (2 H waves + 2 V waves, S=16) rows/chunk=" #CHR_, c, R, [&] { hipLaunchKernelGGL((blur_ring_wsx_kernel<R>), grid, dim3(256), 64 * 160 * 4, 0, c.src, c.dst, c.w, c.h, c.n, c.n, wt, c.nf, CHR_); }); }
        VWS(256) VWS(128) VWS(512)
#define VWSM(CHR_, MAP_) { const int tx = c.w / 128, nch = (c.h + CHR_ - 1) / CHR_; \
        const int total = tx * nch * c.nf; dim3 grid(((total + 7) / 8) * 8, 1, 1); \
        run_variant("ring WS rolemap=" #MAP_ " rows/chunk=" #CHR_, c, R, [&] { hipLaunchKernelGGL((blur_ring_wsx_kernel<R, MAP_>), grid, dim3(256), 64 * 160 * 4, 0, c.src, c.dst, c.w, c.h, c.n, c.n, wt, c.nf, CHR_); }); }
        VWSM(256, 1) VWSM(256, 2) VWSM(512, 1) VWSM(512, 2)
        // round 4 experiment: S = 64 -- 64-row steps, a 128-row ring, 512 threads (8 wavefronts), two workgroups per CU:
        // half the barriers per output row at the same wavefronts per CU
#define VR64(CHR_, MINW_, H8_) { using G = RingGeom<R, 64>; \
        const int tx = (c.w + G::TW - 1) / G::TW, nch = (c.h + CHR_ - 1) / CHR_; \
        const int total = tx * nch * c.nf; dim3 grid(((total + 7) / 8) * 8, 1, 1); \
        auto kfn = blur_ring_kernel<R, MINW_, 64, false, false, 0, -1, H8_, false>; \
        CHECK(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds_bytes)); \
        run_variant("ring S=64 (512 threads, 128-row ring) rows/chunk=" #CHR_ " minw=" #MINW_ " H8=" #H8_, c, R, [&] { hipLaunchKernelGGL(kfn, grid, dim3(512), G::lds_bytes, 0, c.src, c.dst, c.w, c.h, c.n, c.n, wt, c.nf, CHR_, nodec, noact, none); }); }
        VR64(256, 4, (R <= 12)) VR64(512, 4, (R <= 12)) VR64(256, 4, true) VR64(256, 2, (R <= 12))
        // round 4 experiment: LDS-DMA staging on a five-segment ring (blur_dma.hip.h), plain and with 8 outputs per lane in the horizontal pass
#define VDMA(CHR_, H8_) { const int tx = c.w / 128, nch = (c.h + CHR_ - 1) / CHR_; \
        const int total = tx * nch * c.nf; dim3 grid(((total + 7) / 8) * 8, 1, 1); \
        run_variant("ring DMA (LDS-DMA staging, 5 x 16-row segments, 2 H + 2 V waves) H8=" #H8_ " rows/chunk=" #CHR_, c, R, [&] { hipLaunchKernelGGL((blur_ring_dma_kernel<R, H8_>), grid, dim3(256), 80 * 160 * 4, 0, c.src, c.dst, c.w, c.h, c.n, c.n, wt, c.nf, CHR_); }); }
        VDMA(256, false) VDMA(256, true) VDMA(512, false) VDMA(512, true)
        if constexpr (R >= 9) {     // the shipping wave-specialised kernel (dense_kernels.hip.h), without and with the activity flags
#define VWSP(CHR_, ACT_) { using Gw = RingWsGeom<R>; const int tx = (c.w + 127) / 128, nch = (c.h + CHR_ - 1) / CHR_; \
        const int total = tx * nch * c.nf; dim3 grid(((total + 7) / 8) * 8, 1, 1); \
        const int ncell = (c.w + 63) / 64; Activity act{(ACT_) ? (unsigned char *)c.diag : nullptr, (size_t)c.h * ncell, ncell, 0.8f * 0.0133f}; \
        run_variant("ring WS shipping kernel ACT=" #ACT_ " rows/chunk=" #CHR_, c, R, [&] { hipLaunchKernelGGL((blur_ring_ws_kernel<R, ACT_>), grid, dim3(256), (ACT_) ? Gw::lds_bytes_act : Gw::lds_bytes, 0, c.src, c.dst, c.w, c.h, c.n, c.n, wt, c.nf, CHR_, act); }); }
            VWSP(256, false) VWSP(256, true)
        }
    }
    // with the extrema activity flags (ACT) as layers 2 ... nspo+1 of the pipeline write them
#define VRA(S_, CHR_, MINW_) { using G = RingGeom<R, S_>; \
        const int tx = (c.w + G::TW - 1) / G::TW, nch = (c.h + CHR_ - 1) / CHR_; \
        const int total = tx * nch * c.nf; dim3 grid(((total + 7) / 8) * 8, 1, 1); \
        const int ncell = (c.w + 63) / 64; Activity act{(unsigned char *)c.diag, (size_t)c.h * ncell, ncell, 0.8f * 0.0133f}; \
        run_variant("ring S=" #S_ " rows/chunk=" #CHR_ " minw=" #MINW_ " ACT", c, R, [&] { hipLaunchKernelGGL((blur_ring_kernel<R, MINW_, S_, false, true, 0>), grid, dim3(256), G::lds_bytes_act, 0, c.src, c.dst, c.w, c.h, c.n, c.n, wt, c.nf, CHR_, nodec, act, none); }); }
    VRA(32, 128, 4) VRA(32, 256, 4)
}

int main(int argc, char **argv) {
    Ctx c; c.w = argc > 1 ? atoi(argv[1]) : 3840; c.h = argc > 2 ? atoi(argv[2]) : 2160; c.nf = argc > 3 ? atoi(argv[3]) : 8;
    c.n = (size_t)c.w * c.h;
    if (argc > 4) g_rfilter = atoi(argv[4]);
    if (argc > 5) g_filter = argv[5];
    CHECK(hipMalloc(&c.src, c.n * c.nf * 4)); CHECK(hipMalloc(&c.dst, c.n * c.nf * 4));
    CHECK(hipMalloc(&c.ref, c.n * c.nf * 4)); CHECK(hipMalloc(&c.tmp, c.n * c.nf * 4)); CHECK(hipMalloc(&c.diag, 64u << 20));
    std::vector<float> h(c.n * c.nf);
    unsigned s = 12345; for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) * (1.0f / 16777216.0f); }
    CHECK(hipMemcpy(c.src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    printf("%dx%d x %d frames\n", c.w, c.h, c.nf);
    bench_R<5>(c, 1.2263f); bench_R<7>(c, 1.5450f); bench_R<8>(c, 1.9466f); bench_R<10>(c, 2.4525f); bench_R<13>(c, 3.0900f);
    return 0;
}
