// Harness of the fused two-layer marching blur (blur_fused.hip.h): times it against the two blur_ring_kernel launches
// it would replace and checks BOTH layers bit for bit against them (which tests/ hold against the oracle).
//   blur_fused [w h frames]      default 3840 2160 32
// build: make -C tools/ubench blur_fused
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "dense_kernels.hip.h"
#include "blur_fused.hip.h"
using namespace siftmi;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static int gw(float s, TapWeights &out) {
    int radius = (int)std::ceil(4.0f * s), size = 2 * radius + 1; float t = 0, ss = s * s;
    for (int k = -radius, i = 0; k <= radius; k++, i++) { float w = std::exp(-0.5f * ((float)(k * k) / ss)); out.w[i] = w; t += w; }
    for (int i = 0; i < size; i++) out.w[i] /= t;
    for (int i = size; i < 32; i++) out.w[i] = 0;
    return size;
}

struct Ctx { float *src, *ref_a, *ref_b, *dst_a, *dst_b; int w, h, nf; size_t n; };

template <typename F>
static float time_ms(F launch, int iters = 20) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) launch();
    hipEventRecord(e0);
    for (int i = 0; i < iters; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / iters;
}

static size_t mismatches(const Ctx &c, const float *a, const float *b) {
    std::vector<float> x(c.n), y(c.n);
    size_t bad = 0;
    for (int f : {0, c.nf - 1}) {
        CHECK(hipMemcpy(x.data(), a + (size_t)f * c.n, c.n * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(y.data(), b + (size_t)f * c.n, c.n * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < c.n; i++) bad += memcmp(&x[i], &y[i], 4) != 0;
    }
    return bad;
}

template <int RA, int RB>
static void bench_pair(Ctx &c, float rho_a, float rho_b) {
    TapWeights wa, wb;
    if (gw(rho_a, wa) != 2 * RA + 1 || gw(rho_b, wb) != 2 * RB + 1) { printf("tap counts do not match the template radii\n"); return; }
    SeedSource none; memset(&none, 0, sizeof(none)); Decimate nodec; memset(&nodec, 0, sizeof(nodec));
    Activity noact{nullptr, 0, 0, 0.0f};
    const double px = (double)c.n * c.nf;
    // the two launches of the pipeline (256-row chunks, as octave 0 of the library)
    auto sep_a = [&] { using Gr = RingGeom<RA>; const int total = ((c.w + Gr::TW - 1) / Gr::TW) * ((c.h + 255) / 256) * c.nf;
        hipLaunchKernelGGL((blur_ring_kernel<RA, 4, 32, false, false>), dim3(((total + 7) / 8) * 8), dim3(256), Gr::lds_bytes, 0, c.src, c.ref_a, c.w, c.h, c.n, c.n, wa, c.nf, 256, nodec, noact, none); };
    auto sep_b = [&] { using Gr = RingGeom<RB>; const int total = ((c.w + Gr::TW - 1) / Gr::TW) * ((c.h + 255) / 256) * c.nf;
        hipLaunchKernelGGL((blur_ring_kernel<RB, 4, 32, false, false>), dim3(((total + 7) / 8) * 8), dim3(256), Gr::lds_bytes, 0, c.ref_a, c.ref_b, c.w, c.h, c.n, c.n, wb, c.nf, 256, nodec, noact, none); };
    sep_a(); sep_b();
    CHECK(hipDeviceSynchronize());
    const float ta = time_ms(sep_a), tb = time_ms(sep_b), tab = time_ms([&] { sep_a(); sep_b(); });
    printf("R = %d then R = %d on %d x %dx%d\n", RA, RB, c.nf, c.w, c.h);
    printf("  two blur_ring_kernel launches              %.4f + %.4f ms alone, %.4f ms back to back   %7.1f GB/s of 16 B/px\n", ta, tb, tab, 16.0 * px / (tab * 1e-3) / 1e9);
#define FUSED(CHR_, MINW_, DBG_) FUSEDX(CHR_, MINW_, DBG_, false)
#define FUSEDX(CHR_, MINW_, DBG_, PF2_) { \
        auto kfn = blur_ring2_kernel<RA, RB, MINW_, DBG_, PF2_>; \
        CHECK(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Ring2Geom::lds_bytes)); \
        const int total = ((c.w + Ring2Geom::TWO - 1) / Ring2Geom::TWO) * ((c.h + CHR_ - 1) / CHR_) * c.nf; \
        auto launch = [&] { hipLaunchKernelGGL(kfn, dim3(((total + 7) / 8) * 8), dim3(256), Ring2Geom::lds_bytes, 0, c.src, c.dst_a, c.dst_b, c.w, c.h, c.n, wa, wb, c.nf, CHR_); }; \
        CHECK(hipMemset(c.dst_a, 0xff, c.n * c.nf * 4)); CHECK(hipMemset(c.dst_b, 0xff, c.n * c.nf * 4)); \
        launch(); CHECK(hipDeviceSynchronize()); CHECK(hipGetLastError()); \
        const size_t bad_a = (DBG_) ? 0 : mismatches(c, c.dst_a, c.ref_a), bad_b = (DBG_) ? 0 : mismatches(c, c.dst_b, c.ref_b); \
        const float tf = time_ms(launch); \
        printf("  fused rows/chunk=%-4d minw=%d dbg=%-2d prefetch %d   %.4f ms   %7.1f GB/s of 16 B/px (moves 12)   %s\n", CHR_, MINW_, DBG_, (PF2_) ? 2 : 1, tf, 16.0 * px / (tf * 1e-3) / 1e9, \
               (DBG_) ? "(ablation: no check)" : (bad_a || bad_b) ? "MISMATCH" : "both layers exact"); \
        if (bad_a || bad_b) printf("      mismatching floats: layer A %zu, layer B %zu\n", bad_a, bad_b); \
        fflush(stdout); }
    FUSED(256, 2, 0) FUSED(512, 2, 0) FUSED(128, 2, 0) FUSED(1088, 2, 0) FUSED(256, 1, 0) FUSED(256, 2, 24)
    FUSEDX(256, 2, 0, true) FUSEDX(512, 2, 0, true) FUSEDX(512, 2, 24, true)
    FUSED(512, 2, 4)
}

// seed + layer 1 as the pipeline launches them today: blur_ring_kernel<5, ..., SEEDF = BGRA8> (luma + 2x bilinear + seed blur from the w/2 x h/2
// BGRA8 frames) and blur_ring_kernel<5> on its output -- the pair a fused "seed + layer 1" launch would replace.  A fused launch's stage A
// would read 1 B per octave pixel instead of 4; the fused kernel above with NO input loads at all (dbg 4) bounds it from below.
static void bench_seed_pair(Ctx &c) {
    TapWeights ws, wl;
    if (gw(1.2490f, ws) != 11 || gw(1.2263f, wl) != 11) return;
    const int wi = c.w / 2, hi = c.h / 2;
    unsigned char *px;
    CHECK(hipMalloc(&px, (size_t)wi * hi * 4 * c.nf));
    std::vector<unsigned char> hp((size_t)wi * hi * 4 * c.nf);
    unsigned s = 777; for (auto &v : hp) { s = s * 1664525u + 1013904223u; v = (unsigned char)(s >> 24); }
    CHECK(hipMemcpy(px, hp.data(), hp.size(), hipMemcpyHostToDevice));
    SeedSource seed; seed.pixels = px; seed.frame_stride = (size_t)wi * hi * 4; seed.row_stride = (size_t)wi * 4; seed.format = FMT_BGRA8; seed.in_w = wi; seed.in_h = hi;
    SeedSource none; memset(&none, 0, sizeof(none)); Decimate nodec; memset(&nodec, 0, sizeof(nodec));
    Activity noact{nullptr, 0, 0, 0.0f};
    using Gr = RingGeom<5>;
    auto l_seed = [&] { const int total = ((c.w + Gr::TW - 1) / Gr::TW) * ((c.h + 543) / 544) * c.nf;
        hipLaunchKernelGGL((blur_ring_kernel<5, 4, 32, false, false, 0, FMT_BGRA8>), dim3(((total + 7) / 8) * 8), dim3(256), Gr::lds_bytes, 0, nullptr, c.ref_a, c.w, c.h, c.n, c.n, ws, c.nf, 544, nodec, noact, seed); };
    auto l_layer = [&] { const int total = ((c.w + Gr::TW - 1) / Gr::TW) * ((c.h + 255) / 256) * c.nf;
        hipLaunchKernelGGL((blur_ring_kernel<5, 4, 32, false, false>), dim3(((total + 7) / 8) * 8), dim3(256), Gr::lds_bytes, 0, c.ref_a, c.ref_b, c.w, c.h, c.n, c.n, wl, c.nf, 256, nodec, noact, none); };
    l_seed(); l_layer(); CHECK(hipDeviceSynchronize()); CHECK(hipGetLastError());
    const float t0 = time_ms(l_seed), t1 = time_ms(l_layer), t01 = time_ms([&] { l_seed(); l_layer(); });
    printf("seed (BGRA8 -> layer 0) + layer 1 on %d x %dx%d, the pipeline's two launches: %.4f + %.4f ms alone, %.4f ms back to back\n", c.nf, c.w, c.h, t0, t1, t01);
    CHECK(hipFree(px));
}

int main(int argc, char **argv) {
    Ctx c; c.w = argc > 1 ? atoi(argv[1]) : 3840; c.h = argc > 2 ? atoi(argv[2]) : 2160; c.nf = argc > 3 ? atoi(argv[3]) : 32;
    c.n = (size_t)c.w * c.h;
    for (float **p : {&c.src, &c.ref_a, &c.ref_b, &c.dst_a, &c.dst_b}) CHECK(hipMalloc(p, c.n * c.nf * 4));
    std::vector<float> h(c.n * c.nf);
    unsigned s = 12345; for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) * (1.0f / 16777216.0f); }
    CHECK(hipMemcpy(c.src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    bench_pair<5, 7>(c, 1.2263f, 1.5450f);      // layers 1 + 2 of an octave (the memory-bound pair)
    bench_pair<7, 8>(c, 1.5450f, 1.9466f);      // layers 2 + 3
    bench_pair<5, 5>(c, 1.2490f, 1.2263f);      // the radii of seed + layer 1 (float input here: the arithmetic and traffic of the blur part only)
    bench_seed_pair(c);
    return 0;
}
