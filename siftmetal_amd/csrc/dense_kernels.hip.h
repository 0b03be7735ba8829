// dense_kernels.hip.h -- Gaussian scale-space kernels for gfx950 (MI355X).
//
// What the reference computes (one dispatch per pass, full texture round trip each):
//   convertSRGBToGrayscale  Sources/MetalShaders/Metal/ConvertSRGBToGrayscale.metal:11-23
//   bilinearUpScale         Sources/MetalShaders/Metal/BilinearUpScale.metal:12-64
//   convolutionX/Y          Sources/MetalShaders/Metal/Convolution.metal:15-52        (seed blur)
//   convolutionSeriesX/Y    Sources/MetalShaders/Metal/ConvolutionSeries.metal:16-53  (layer blurs)
//   nearestNeighborDownScale Sources/MetalShaders/Metal/NearestNeighborDownScale.metal:15-22  (fused: Decimate)
//
// How it is built here: ONE fused separable kernel per Gaussian layer.  A 256-thread workgroup
// owns a TW x TH output tile, stages the (TH+2R) x (TW+2RP) input window in LDS (mirror
// extension resolved at load time), runs the horizontal pass in place on the LDS rows (one
// 32-lane half-wave per row, 4 adjacent outputs per lane, b128 LDS reads), then the vertical
// pass from LDS with a register sliding window (4 columns x RB rows per lane) and writes
// coalesced float4 rows.  Algorithmic traffic: 4 B read + 4 B written per octave pixel.
// The seed layer uses the same kernel with a loader that evaluates luma + 2x bilinear on the
// fly, so gray / upscaled / X-pass intermediates never touch HBM.
//
// Float policy: each output is the reference's tap loop `sum += w[i] * c` in the same tap order,
// evaluated as fmaf(w[i], c, sum); the CPU oracle does the same, so the pyramid is bit-identical
// between the two (tests/test_gpu_parity.py).  Built with -ffp-contract=off: nothing else fuses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace siftmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) volatile f32x4 lds_cv_f32x4;      // LDS-qualified, see blur phase 1

struct TapWeights { float w[32]; };            // ConvolutionParameters.weights (ConvolutionSeries.h:13-21)

enum { FMT_BGRA8 = 0, FMT_GRAY8 = 1, FMT_GRAYF32 = 2 };

// Sources/MetalShaders/Metal/Common.hpp:15-22 symmetrizedCoordinates.  (i + 2l) % 2l without the
// division for the single-wrap range; the general branch keeps C's truncating remainder so that
// far-out-of-range indices stay negative (-> the read is out of bounds -> 0), as in the reference.
__device__ __forceinline__ int symm(int i, int l) {
    const int ll = 2 * l;
    if (i >= -ll && i < 2 * ll) {
        if (i < 0) i += ll; else if (i >= ll) i -= ll;
    } else {
        i = (i + ll) % ll;
    }
    if (i > l - 1) i = ll - 1 - i;
    return i;
}

// Optional second output of a layer blur: the next octave's layer 0 = even rows / even columns of this
// layer (NearestNeighborDownScale.metal:15-22, DifferenceOfGaussians.swift:193-199), written by the lanes that
// hold those pixels so that the decimation costs no extra launch and no re-read.
struct Decimate {
    float *dst;                     // next octave layer 0, frame 0 (nullptr = off)
    size_t frame_stride;            // floats between frames
    int w2, h2;                     // next octave size
};

// Activity flags for the extrema scan (see extrema_kernel): the blur that produces Gaussian layer s+1 records, per image
// row and 64-column cell, whether |DoG[s]| = |G[s+1] - G[s]| can exceed the refinement-entry threshold there, without
// touching the input layer again.  With hb = the horizontally blurred input at the pixel,
//     DoG = (G[s+1] - hb) + (hb - G[s]) = Ev + Eh,
// Eh is known in the horizontal pass (the raw centre values are still in its registers) and Ev in the vertical pass (hb
// is the centre tap's operand).  The horizontal pass leaves max|Eh| per (window row, 8-column sub-cell) in LDS -- a lane's
// own 8 outputs, no cross-lane reduction -- and the vertical pass flags a cell when |Ev| + max|Eh| > 0.9999 thr for one of
// its pixels: a conservative bound (|DoG| <= |Ev| + |Eh|), so no candidate row is ever skipped, and tight in practice
// (benchmark frames, with the maximum still taken over the whole 64-column cell: 15.2 % of octave 0's cells flagged
// against 12.2 % with the exact test; the sub-cells took 3 % off the extrema scan and 2-4 % off the flagged layers).
struct Activity {
    unsigned char *dst;             // plane [h][ncell] of this DoG scale, frame 0 (nullptr = off)
    size_t frame_stride;            // bytes between frames
    int ncell;                      // cells per row = ceil(w / 64)
    float thr;                      // 0.8 * dog_threshold (SIFTInterpolate.metal:208)
};

// Two int ranges cleared by the seed tile kernel on its way (the per-call counters and, for the per-octave chains of a
// single-frame call, the keypoint sort's row buckets): a launch and a dependency gap fewer at the head of every small call.
struct ZeroJob { int32_t *a; size_t na; int32_t *b; size_t nb; };

struct SeedSource {                 // input frame description for the seed loader
    const unsigned char *pixels;    // frame 0
    size_t frame_stride;            // bytes between frames
    size_t row_stride;              // bytes between rows
    int format;
    int in_w, in_h;                 // input size (W, H); the seed image is 2W x 2H
};

// byte / 255 (a bgra8Unorm texel), correctly rounded like the IEEE division it replaces: q = x * fl(1/255) followed by one
// residual correction gives fl(x / 255) for every byte value (all 256 checked in tests/test_oracle_golden.py) in 3 VALU
// operations; the division sequence takes ~10.
__device__ __forceinline__ float unorm8(unsigned b) {
    const float x = (float)b, r = 0.003921568859368563f;
    const float q = x * r;
    return fmaf(fmaf(-q, 255.0f, x), r, q);
}

// luma from the raw 32 bits of a pixel: ConvertSRGBToGrayscale.metal:17-20 on bgra8Unorm texels (FMT_BGRA8), a byte / 255
// (FMT_GRAY8), or the float itself
__device__ __forceinline__ float luma_of(int format, unsigned raw) {
    if (format == FMT_BGRA8) {
        const float b = unorm8(raw & 255u), g = unorm8((raw >> 8) & 255u), r = unorm8((raw >> 16) & 255u);
        return 0.0f + (0.212639005871510f * r) + (0.715168678767756f * g) + (0.072192315360734f * b);
    } else if (format == FMT_GRAY8) {
        return unorm8(raw & 255u);
    }
    return __uint_as_float(raw);
}
__device__ __forceinline__ unsigned raw_pixel(int format, const unsigned char *frame, const SeedSource &s, int x, int y) {   // in range
    const unsigned char *row = frame + (size_t)y * s.row_stride;
    if (format == FMT_GRAY8) return row[x];
    return *reinterpret_cast<const unsigned *>(row + 4 * (size_t)x);
}

// luma of input pixel (x, y), 0 outside the image
__device__ __forceinline__ float luma_at(const unsigned char *frame, const SeedSource &s, int x, int y) {
    if (x < 0 || y < 0 || x >= s.in_w || y >= s.in_h) return 0.0f;
    return luma_of(s.format, raw_pixel(s.format, frame, s, x, y));
}

// BilinearUpScale.metal:24-61 at output pixel (i, j) of the wo x ho = 2W x 2H image
__device__ __forceinline__ float seed_sample(const unsigned char *frame, const SeedSource &s, int i, int j, int wo, int ho) {
    if (i < 0 || j < 0 || i >= wo || j >= ho) return 0.0f;
    const int wi = s.in_w, hi = s.in_h;
    const float dx = (float)wi / (float)wo, dy = (float)hi / (float)ho;
    const float x = (float)i * dx, y = (float)j * dy;
    int im = (int)x, jm = (int)y;
    int ip = im + 1, jp = jm + 1;
    if (ip >= wi) ip = 2 * wi - 1 - ip;
    if (im >= wi) im = 2 * wi - 1 - im;
    if (jp >= hi) jp = 2 * hi - 1 - jp;
    if (jm >= hi) jm = 2 * hi - 1 - jm;
    const float fx = x - floorf(x), fy = y - floorf(y);
    const float c0 = luma_at(frame, s, ip, jp), c1 = luma_at(frame, s, ip, jm);
    const float c2 = luma_at(frame, s, im, jp), c3 = luma_at(frame, s, im, jm);
    return fx * (fy * c0 + (1.0f - fy) * c1) + (1.0f - fx) * (fy * c2 + (1.0f - fy) * c3);
}

// Streaming hints of the ring kernel's global traffic (bit 0: row stores non-temporal, bit 1: row loads non-temporal).  A layer
// is written once and next read 4 GB later (64 frames), far beyond L2 + Infinity Cache, so its lines need not stay cached:
// non-temporal row stores took 2.7-3.1 % off the memory-bound R = 5 layer and nothing off R = 13 (round 4, tools/ubench/blur_variants
// built with -DSIFTMI_RING_NT=0...3, two interleaved passes on one box: 0.431 / 0.427 -> 0.418 / 0.415 ms per 32 x 3840x2160);
// non-temporal LOADS cost 6-17 % -- the 25 % halo columns a strip shares with its neighbours are L2 hits only while they stay cached.
#ifndef SIFTMI_RING_NT
#define SIFTMI_RING_NT 1
#endif
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also emits s_waitcnt vmcnt(0), which
// drains every outstanding global load and store at each barrier -- fatal for kernels that keep
// prefetches in flight across phases (the marching blur lost its whole load/compute overlap to it).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Tap weights live in VGPRs.  Measured on MI355X (tools/ubench/ubench_valu.hip): v_fmac_f32 with an
// SGPR multiplicand issues at HALF the rate of the all-VGPR form (57-67 vs 108-126 TFLOP/s), so
// the kernarg weights are copied into vector registers once per wave; the asm keeps the compiler
// from folding the copy back into an SGPR operand.
template <int NT>
struct VTaps {
    float w[NT];
    __device__ __forceinline__ explicit VTaps(const TapWeights &wt) {
#pragma unroll
        for (int i = 0; i < NT; i++) asm volatile("v_mov_b32 %0, %1" : "=v"(w[i]) : "s"(wt.w[i]));
    }
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ring_store2(float *p, f32x2 v) {
    if (SIFTMI_RING_NT & 1) __builtin_nontemporal_store(v, reinterpret_cast<f32x2 *>(p));
    else *reinterpret_cast<f32x2 *>(p) = v;
}
__device__ __forceinline__ f32x4 ring_load4(const float *p) {
    if (SIFTMI_RING_NT & 2) return __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
    return *reinterpret_cast<const f32x4 *>(p);
}
typedef __attribute__((address_space(3))) volatile f32x2 lds_cv_f32x2;
typedef __attribute__((address_space(3))) volatile float lds_cv_f32;
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Mirror extension without a branch (both blur kernels' row loads).  For w % 4 == 0 a float4 at columns gx ... gx+3
// (gx % 4 == 0) lies wholly inside or wholly outside the image, and outside it is the float4 at -gx - 4 (left) or
// 2w - 4 - gx (right) with its elements reversed (one reflection; Common.hpp:15-22).  The clamp only matters for far
// halo columns of a partial tile that no output depends on: it keeps their address inside the row.
__device__ __forceinline__ f32x4 load_quad_mirrored(const float *rowp, int gx, int w) {
    const bool mir = gx < 0 || gx >= w;
    const int g2 = min(max(gx < 0 ? -gx - 4 : (gx >= w ? 2 * w - 4 - gx : gx), 0), w - 4);
    const f32x4 v = ring_load4(rowp + g2);
    f32x4 r;
    r.x = mir ? v.w : v.x; r.y = mir ? v.z : v.y; r.z = mir ? v.y : v.z; r.w = mir ? v.x : v.w;
    return r;
}
__device__ __forceinline__ f32x2 load_pair_mirrored(const float *rowp, int gx, int w) {      // gx even, w even
    const bool mir = gx < 0 || gx >= w;
    const int g2 = min(max(gx < 0 ? -gx - 2 : (gx >= w ? 2 * w - 2 - gx : gx), 0), w - 2);
    const f32x2 v = *reinterpret_cast<const f32x2 *>(rowp + g2);
    f32x2 r;
    r.x = mir ? v.y : v.x; r.y = mir ? v.x : v.y;
    return r;
}

// ------------------------------------------------------------------------------------------------
// One Gaussian layer: dst = blur_R(src), separable, mirror extension.  SEED = true: src is ignored and
// the input is seed_sample() of the frame's pixels.  Tile height, workgroup size, vertical register
// blocking, occupancy hint and the XCD-aware tile order are template parameters so that
// tools/ubench/blur_variants.hip can time variants; BlurShip below is the shipping choice.
template <int R, int TH_, int NTHR_, int HO_, int RB_>
struct Blur2Geom {
    // staged halo of 8 or 16 columns per side: a staged row is then a whole number of float4 (+ one float2) for each of
    // the 8 lanes that load it (see the staging code)
    static constexpr int RP = R <= 8 ? 8 : 16;
    static constexpr int TW = 128, TH = TH_, NTHR = NTHR_, HO = HO_, RB = RB_;
    static constexpr int LW = TW + 2 * RP, LH = TH + 2 * R, NT = 2 * R + 1;
    static constexpr int NPF4 = LW / 32, REM = (LW - 32 * NPF4) / 8;          // per lane and staged row: float4s + floats left over
    static constexpr int NBATCH = (LH * 8 + NTHR - 1) / NTHR;                 // row batches of NTHR / 8 rows
    static_assert((REM == 0 || REM == 2) && (NPF4 * 4 + REM) * 8 == LW, "row decomposition");
    static constexpr int V_ITEMS = (TW / 4) * (TH / RB);
    static constexpr size_t lds_bytes = (size_t)LW * LH * sizeof(float);
    static constexpr int NSUB = TW / 8;                                       // activity flags: max|Eh| per (window row, 8-column sub-cell) behind the tile
    static constexpr size_t lds_bytes_act = lds_bytes + (size_t)LH * NSUB * sizeof(float);
    // seed variant: luma of the input pixels under the staged 2x window, computed once per workgroup
    static constexpr int LWI = LW / 2 + 3, LHI = LH / 2 + 3;
    static constexpr size_t seed_lds_bytes = lds_bytes + (size_t)LWI * LHI * sizeof(float);
    static_assert(HO == 4, "horizontal pass computes 4 adjacent outputs per lane");
    static_assert(TH % RB == 0, "TH must be a multiple of RB");
};

// ACT (round 3): the DoG activity flags of the marching kernel (see Activity) from the tile kernel, so that a single large
// frame's extrema scan can skip rows too: max|Eh| per (window row, 8-column sub-cell) = two lanes of the horizontal pass,
// |Ev| from the vertical pass's centre operand, one ballot per output row.
template <int R, int TH_, int NTHR_, int HO_, int RB_, bool SEED, int MINW = 1, int KCH = 0, bool XCD = false, bool DEC = false, bool ACT = false>
__global__ __launch_bounds__(NTHR_, MINW) void blur2_kernel(const float *__restrict__ src, float *__restrict__ dst, int w, int h,
                                                           size_t src_frame_stride, size_t dst_frame_stride, TapWeights wt, SeedSource seed, int n_frames,
                                                           Decimate dec, Activity act, ZeroJob zj) {
    using G = Blur2Geom<R, TH_, NTHR_, HO_, RB_>;
    constexpr int NTHR = G::NTHR;
    if (SEED && zj.a) {                                    // nothing reads these before the kernels that follow the seed
        const size_t gsz = (size_t)gridDim.x * gridDim.y * gridDim.z * NTHR;
        const size_t gtid = ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NTHR + threadIdx.x;
        for (size_t i = gtid; i < zj.na; i += gsz) zj.a[i] = 0;
        for (size_t i = gtid; i < zj.nb; i += gsz) zj.b[i] = 0;
    }
    static_assert(!ACT || (!SEED && G::V_ITEMS == NTHR && NTHR % 64 == 0 && (G::LH * (G::TW / 4)) % 64 == 0), "ACT: whole wavefronts in both passes");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    int bx = blockIdx.x, by = blockIdx.y, frame = blockIdx.z;
    if (XCD) {
        // 1-D launch; workgroups are dealt round-robin over the 8 XCDs (b % 8 = XCD group), so give
        // every XCD one contiguous run of tiles: neighbouring tiles then share an L2 and the halo
        // rows/columns are L2 hits instead of a second fetch through the fabric.
        const int tx = (w + G::TW - 1) / G::TW, ty = (h + G::TH - 1) / G::TH;
        const int total = tx * ty * n_frames;
        const int chunk = (total + 7) >> 3;
        const int t = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
        if (t >= total) return;
        frame = t / (tx * ty);
        const int r = t - frame * (tx * ty);
        by = r / tx; bx = r - by * tx;
    }
    const int x0 = bx * G::TW, y0 = by * G::TH;
    const float *__restrict__ in = SEED ? nullptr : src + (size_t)frame * src_frame_stride;
    float *__restrict__ out = dst + (size_t)frame * dst_frame_stride;
    const unsigned char *px = SEED ? seed.pixels + (size_t)frame * seed.frame_stride : nullptr;

    const bool interior = (x0 - G::RP >= 0) && (x0 + G::TW + G::RP <= w) && (y0 - R >= 0) && (y0 + G::TH + R <= h);
    // Row staging, fast form: 8 lanes per row (float4 columns q + 8 j and the float2 left over), NTHR / 8 rows per batch,
    // EVERY load of every batch issued before the first LDS store -- one memory latency per tile.  (Round 1 staged with a
    // load -> wait -> store loop: 9 latencies for an interior tile, 36 for a border tile with its per-element mirror
    // arithmetic, which put a floor of ~17 us under every small-octave launch.)  Mirrored rows and columns cost no branch
    // (symm per row, load_quad_mirrored); needs w % 4 == 0 and an image of at least 16 x 16 so that one reflection
    // reaches every input an output depends on.
    if (!SEED && (w & 3) == 0 && w >= 16 && h >= 16) {
        const int prow = tid / 8, pq = tid & 7;
        f32x4 buf[G::NBATCH][G::NPF4];
        f32x2 rem[G::NBATCH];
#pragma unroll
        for (int b = 0; b < G::NBATCH; b++) {
            const int ly = min(b * (NTHR / 8) + prow, G::LH - 1);
            const float *rowp = in + (size_t)min(max(symm(y0 - R + ly, h), 0), h - 1) * w;
#pragma unroll
            for (int j = 0; j < G::NPF4; j++) buf[b][j] = load_quad_mirrored(rowp, x0 - G::RP + 4 * pq + 32 * j, w);
            if (G::REM) rem[b] = load_pair_mirrored(rowp, x0 - G::RP + 32 * G::NPF4 + 2 * pq, w);
        }
#pragma unroll
        for (int b = 0; b < G::NBATCH; b++) {
            float *rowp = lds + min(b * (NTHR / 8) + prow, G::LH - 1) * G::LW;      // lanes past the last row repeat it
#pragma unroll
            for (int j = 0; j < G::NPF4; j++) *reinterpret_cast<f32x4 *>(rowp + 4 * pq + 32 * j) = buf[b][j];
            if (G::REM) *reinterpret_cast<f32x2 *>(rowp + 32 * G::NPF4 + 2 * pq) = rem[b];
        }
    } else if (SEED) {
        // luma (ConvertSRGBToGrayscale.metal) of every input pixel the staged window can touch, once,
        // into LDS; the 2x bilinear samples (BilinearUpScale.metal) are then formed from LDS.  Each
        // luma is reused by >= 4 upscaled samples, and the byte->float divisions are the expensive part.
        float *lum = lds + G::LH * G::LW;
        const int ix0 = ((x0 - G::RP) >> 1) - 1, iy0 = ((y0 - R) >> 1) - 1;
        {   // all pixel loads first (clamped coordinates, so none is conditional), then the lumas: one memory latency
            constexpr int NL = (G::LHI * G::LWI + NTHR - 1) / NTHR;
            unsigned raw[NL];
#pragma unroll
            for (int k = 0; k < NL; k++) {
                const int idx = min(tid + k * NTHR, G::LHI * G::LWI - 1);
                const int ly = idx / G::LWI, lx = idx - ly * G::LWI;
                raw[k] = raw_pixel(seed.format, px, seed, min(max(ix0 + lx, 0), seed.in_w - 1), min(max(iy0 + ly, 0), seed.in_h - 1));
            }
#pragma unroll
            for (int k = 0; k < NL; k++) {
                const int idx = min(tid + k * NTHR, G::LHI * G::LWI - 1);
                const int ly = idx / G::LWI, lx = idx - ly * G::LWI;
                const int x = ix0 + lx, y = iy0 + ly;
                lum[idx] = (x < 0 || y < 0 || x >= seed.in_w || y >= seed.in_h) ? 0.0f : luma_of(seed.format, raw[k]);   // 0 outside the image
            }
        }
        __syncthreads();
        const int wi = seed.in_w, hi = seed.in_h;
        const float dx = (float)wi / (float)w, dy = (float)hi / (float)h;
        if (interior && w == 2 * wi && h == 2 * hi) {
            // interior tile of the 2x seed image: no mirror, and an (even, odd) column pair shares its lumas.
            // BilinearUpScale.metal:56-59 is fx*(fy*c0 + (1-fy)*c1) + (1-fx)*(fy*c2 + (1-fy)*c3); with A(i) the
            // vertical blend of input column i this is fx*A(ip) + (1-fx)*A(im), and fx = 0 on even columns, so
            // the even output is A(im) exactly (0 * finite + 1 * A) and the odd one 0.5*A(ip) + 0.5*A(im).
            for (int idx = tid; idx < G::LH * (G::LW / 2); idx += NTHR) {
                const int ly = idx / (G::LW / 2), lp = idx - ly * (G::LW / 2);
                const int gy = y0 - R + ly, gx = x0 - G::RP + 2 * lp;       // gx even
                const float y = (float)gy * dy;
                int jm = (int)y, jp = jm + 1;
                if (jp >= hi) jp = 2 * hi - 1 - jp;
                const float fy = y - floorf(y);
                const int im = gx >> 1;
                int ip = im + 1;
                if (ip >= wi) ip = 2 * wi - 1 - ip;
                const float *rm = lum + (jm - iy0) * G::LWI - ix0, *rp = lum + (jp - iy0) * G::LWI - ix0;
                const float Am = fy * rp[im] + (1.0f - fy) * rm[im];
                const float Ap = fy * rp[ip] + (1.0f - fy) * rm[ip];
                const float fx1 = ((float)(gx + 1) * dx) - floorf((float)(gx + 1) * dx);   // 0.5
                float2 o;
                o.x = 0.0f * Ap + (1.0f - 0.0f) * Am;
                o.y = fx1 * Ap + (1.0f - fx1) * Am;
                *reinterpret_cast<float2 *>(lds + ly * G::LW + 2 * lp) = o;
            }
        } else
        for (int idx = tid; idx < G::LH * G::LW; idx += NTHR) {
            const int ly = idx / G::LW, lx = idx - ly * G::LW;
            const int sx = symm(x0 - G::RP + lx, w), sy = symm(y0 - R + ly, h);
            float v = 0.0f;
            if (sx >= 0 && sy >= 0 && sx < w && sy < h) {                 // BilinearUpScale.metal:24-61
                const float x = (float)sx * dx, y = (float)sy * dy;
                int im = (int)x, jm = (int)y;
                int ip = im + 1, jp = jm + 1;
                if (ip >= wi) ip = 2 * wi - 1 - ip;
                if (im >= wi) im = 2 * wi - 1 - im;
                if (jp >= hi) jp = 2 * hi - 1 - jp;
                if (jm >= hi) jm = 2 * hi - 1 - jm;
                const float fx = x - floorf(x), fy = y - floorf(y);
                auto L = [&](int i, int j) -> float {
                    const int a = i - ix0, b = j - iy0;
                    if (a >= 0 && b >= 0 && a < G::LWI && b < G::LHI) return lum[b * G::LWI + a];
                    return luma_at(px, seed, i, j);                       // outside the staged luma (tiny images)
                };
                const float c0 = L(ip, jp), c1 = L(ip, jm), c2 = L(im, jp), c3 = L(im, jm);
                v = fx * (fy * c0 + (1.0f - fy) * c1) + (1.0f - fx) * (fy * c2 + (1.0f - fy) * c3);
            }
            lds[idx] = v;
        }
    } else {
        for (int idx = tid; idx < G::LH * G::LW; idx += NTHR) {
            const int ly = idx / G::LW, lx = idx - ly * G::LW;
            const int sx = symm(x0 - G::RP + lx, w), sy = symm(y0 - R + ly, h);
            lds[idx] = (sx < 0 || sy < 0 || sx >= w || sy >= h) ? 0.0f : in[(size_t)sy * w + sx];
        }
    }
    const VTaps<G::NT> tw(wt);
    __syncthreads();

    for (int item = tid; item < G::LH * (G::TW / 4); item += NTHR) {
        const int row = item >> 5, c4 = (item & 31) * 4;
        float *rowp = lds + row * G::LW;
        constexpr int M0 = (G::RP - R) / 4, M1 = (G::RP + R + 3) / 4 + 1;      // float4 of the row segment that hold the taps' operands
        float v[4 * (M1 - M0)];
        const lds_cv_f32x4 *rp4 = (const lds_cv_f32x4 *)(rowp + c4);
#pragma unroll
        for (int m = M0; m < M1; m++) {
            const f32x4 t = rp4[m];
            v[4 * (m - M0) + 0] = t.x; v[4 * (m - M0) + 1] = t.y; v[4 * (m - M0) + 2] = t.z; v[4 * (m - M0) + 3] = t.w;
        }
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < G::NT; i++) {
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k] = fmaf(tw.w[i], v[(G::RP - R - 4 * M0) + k + i], acc[k]);
        }
        *reinterpret_cast<float4 *>(rowp + G::RP + c4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        if (ACT) {                                               // |Eh| = |hb - input| at the lane's 4 pixels; the neighbour lane completes the sub-cell
            float e = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; k++) e = fmaxf(e, fabsf(acc[k] - v[(G::RP - R - 4 * M0) + k + R]));
            e = fmaxf(e, __shfl_xor(e, 1));
            if ((tid & 1) == 0) lds[G::LW * G::LH + row * G::NSUB + (c4 >> 3)] = e;
        }
    }
    __syncthreads();

    for (int item = tid; item < G::V_ITEMS; item += NTHR) {
        const int cg = item & 31, rg = item >> 5;
        const float *colp = lds + (rg * G::RB) * G::LW + G::RP + cg * 4;
        float4 acc[G::RB];
        float4 cen[ACT ? G::RB : 1];                             // hb under each output (the centre tap's operand)
#pragma unroll
        for (int rr = 0; rr < G::RB; rr++) acc[rr] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int k = 0; k < G::RB + 2 * R; k++) {
            // bound the live ranges: without this fence hipcc hoists all RB+2R row reads (120+ VGPRs)
            if (KCH > 0 && k > 0 && (k % KCH) == 0) __builtin_amdgcn_sched_barrier(0);
            const float4 v = *reinterpret_cast<const float4 *>(colp + k * G::LW);
#pragma unroll
            for (int rr = 0; rr < G::RB; rr++) {
                const int i = k - rr;
                if (ACT && i == R) cen[rr] = v;
                if (i >= 0 && i < G::NT) {
                    acc[rr].x = fmaf(tw.w[i], v.x, acc[rr].x);
                    acc[rr].y = fmaf(tw.w[i], v.y, acc[rr].y);
                    acc[rr].z = fmaf(tw.w[i], v.z, acc[rr].z);
                    acc[rr].w = fmaf(tw.w[i], v.w, acc[rr].w);
                }
            }
        }
        // pin the accumulators here: otherwise LLVM sinks each row's whole FMA chain into the
        // `gy < h` store guard below, which keeps all RB+2R loaded rows live (120+ VGPRs)
#pragma unroll
        for (int rr = 0; rr < G::RB; rr++) asm volatile("" : "+v"(acc[rr].x), "+v"(acc[rr].y), "+v"(acc[rr].z), "+v"(acc[rr].w));
        const int gx = x0 + cg * 4;
#pragma unroll
        for (int rr = 0; rr < G::RB; rr++) {
            const int gy = y0 + rg * G::RB + rr;
            if (gy >= h) continue;
            float *o = out + (size_t)gy * w + gx;
            if (gx + 3 < w && (w & 3) == 0) {
                *reinterpret_cast<float4 *>(o) = acc[rr];
            } else {
                if (gx + 0 < w) o[0] = acc[rr].x;
                if (gx + 1 < w) o[1] = acc[rr].y;
                if (gx + 2 < w) o[2] = acc[rr].z;
                if (gx + 3 < w) o[3] = acc[rr].w;
            }
            if (DEC && (gy & 1) == 0 && (gy >> 1) < dec.h2) {          // gx is a multiple of 4: columns gx, gx+2 are even
                float *o2 = dec.dst + (size_t)frame * dec.frame_stride + (size_t)(gy >> 1) * dec.w2 + (gx >> 1);
                if ((gx >> 1) + 0 < dec.w2 && gx + 0 < w) o2[0] = acc[rr].x;
                if ((gx >> 1) + 1 < dec.w2 && gx + 2 < w) o2[1] = acc[rr].z;
            }
        }
        if (ACT) {                                               // 16 lanes = 64 columns = one cell of a row; a wavefront holds 2 row groups x 2 cells
            const float lim = act.thr * 0.9999f;
#pragma unroll
            for (int rr = 0; rr < G::RB; rr++) {
                const float eh = lds[G::LW * G::LH + (rg * G::RB + rr + R) * G::NSUB + (cg >> 1)];
                const bool f = (gx + 0 < w && fabsf(acc[rr].x - cen[rr].x) + eh > lim) || (gx + 1 < w && fabsf(acc[rr].y - cen[rr].y) + eh > lim) ||
                               (gx + 2 < w && fabsf(acc[rr].z - cen[rr].z) + eh > lim) || (gx + 3 < w && fabsf(acc[rr].w - cen[rr].w) + eh > lim);
                const unsigned long long b = __ballot(f);
                const int gy = y0 + rg * G::RB + rr, cell = (x0 >> 6) + (cg >> 4);
                if ((tid & 15) == 0 && gy < h && cell < act.ncell)
                    act.dst[(size_t)frame * act.frame_stride + (size_t)gy * act.ncell + cell] = (unsigned char)(((b >> (tid & 48)) & 0xffffull) != 0ull);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Marching (ring) form of the layer blur, the kernel of the large launches.  A workgroup owns a 128-column strip and walks
// down one chunk of it in steps of S rows; the horizontal pass runs once per image row and every input row is fetched once
// per strip (the tile kernel recomputes the horizontal pass (TH + 2R) / TH = 1.8x at R = 13).  The LDS window is a RING of
// NR = 2S rows addressed modulo NR, so nothing is carried between steps (round 1's marching kernel copied the last 2R
// blurred rows to the top of its window through registers after every step: 26 rows per 16 new ones at R = 13, more LDS
// store traffic than the staging itself).  A step is S = 32 rows, and the vertical pass gives every wavefront ONE group of S/4 = 8
// output rows with a lane owning 2 adjacent columns: the ring slot of every window row is then wave-uniform
// (scalar address arithmetic, one v_add per read), the reads are ds_read_b64 at consecutive lanes (full LDS
// rate at any row pitch), and a window row is read (8 + 2R)/8 times per output row instead of (2 + 2R)/2:
// 17 B of LDS reads per pixel at R = 13 against 56 B.  The tap weights are symmetric (w[i] == w[2R - i]
// bit for bit, see gaussian_weights) and held in R + 1 VGPRs.
// Row bookkeeping: u = image row - ybeg - R.  Step st stages rows u in [st S, st S + S) into ring half
// (st & 1); the chunk's prologue also stages u in [-2R, 0) (slots NR - 2R ... NR - 1).  Output row
// y0 + j of step st reads u in [st S + j - 2R, st S + j].
// Same arithmetic and tap order as blur2_kernel: bit-identical results.
template <int R, int S_ = 32>
struct RingGeom {
    // staged halo: 8 or 16 columns per side, so that a step's S new rows are a whole number of float4 (+ one float2)
    // per lane -- every lane issues the same loads, none masked (see the note on s_waitcnt at the kernel)
    static constexpr int RP = R <= 8 ? 8 : 16;
    static constexpr int TW = 128, S = S_, NTHR = 8 * S_, NR = 2 * S_, RB = 8;       // S = 32: 4 wavefronts; S = 64 (experiment): 8, half the barriers per row
    static constexpr int LW = TW + 2 * RP, NT = 2 * R + 1;
    static constexpr int V = LW / 4;                                    // float4 per staged row
    static constexpr int NPF4 = LW / 32;                                // 8 lanes per row: whole float4 per lane ...
    static constexpr int REM = (LW - 32 * NPF4) / 8;                    // ... + this many floats per lane (0 or 2)
    static_assert(S * 8 == NTHR && (REM == 0 || REM == 2) && (NPF4 * 4 + REM) * 8 == LW, "prefetch decomposition");
    static_assert(R <= 15 && S + 2 * R <= NR, "window must fit the ring");
    static_assert((NR & (NR - 1)) == 0, "ring size must be a power of two");
    // max|Eh| per (ring row, 8-column sub-cell) for the activity flags, behind the ring: 4 KB.  (At R <= 8 that makes
    // exactly 40 KB per workgroup: four still fit a CU's 160 KB.)
    static constexpr int NSUB = TW / 8;
    static constexpr size_t lds_bytes = (size_t)LW * NR * sizeof(float);
    static constexpr size_t lds_bytes_act = lds_bytes + (size_t)NR * NSUB * sizeof(float);
};

template <int R>
struct VTapsSym {                   // w[i] for i <= R; w[2R - i] beyond (bit-identical, the weights are symmetric)
    float w[R + 1];
    __device__ __forceinline__ explicit VTapsSym(const TapWeights &wt) {
#pragma unroll
        for (int i = 0; i <= R; i++) asm volatile("v_mov_b32 %0, %1" : "=v"(w[i]) : "s"(wt.w[i]));
    }
    __device__ __forceinline__ float operator()(int i) const { return w[i <= R ? i : 2 * R - i]; }
};

// Register prefetch and s_waitcnt: the S new rows of step st+1 are requested at the start of step st and written to LDS
// at its end.  hipcc counts outstanding vector-memory operations per basic block and merges conservatively at joins, so
// every branch around a store (row / column guards) between the loads and their use made it wait for vmcnt(0) there --
// draining the step's stores and serialising load latency with compute (round 1's "two-deep" prefetch was in fact
// zero-deep; long chunks ran 2x slower than short, prologue-dominated ones).  Interior steps therefore run a FAST body
// whose loads and stores are unconditional straight-line code: the compiler's own count is then exact (vmcnt(8): the
// eight row stores stay in flight) and the loads have a whole step of FMA work to land.
// DBG (tools/ubench only; 0 in the library): 1 = s_memtime stamps per phase, summed per wavefront into the buffer passed as
// act.dst ([workgroup][wave][8] u64); 2 = no global stores, 4 = no global loads, 8 = no horizontal-pass arithmetic,
// 16 = no vertical-pass arithmetic (timing ablations, wrong results).
// SEEDF >= 0: the seed layer (octave 0, layer 0).  `src` is unused; the rows a step stages are the 2x bilinear upscale
// (BilinearUpScale.metal:24-61) of the luma (ConvertSRGBToGrayscale.metal:17-20) of the input frame, pixel format SEEDF.
// A step's S new rows need at most S/2 + 2 input rows x LW/2 + 4 input columns: their raw pixels are prefetched like the
// float rows of a layer blur (6 dwords per lane instead of 18 floats), turned into luma ONCE per input pixel into a tile
// that borrows the dead rows of the current ring half, and expanded from there into the ring rows (an even output row /
// column is a luma row / column exactly, an odd one the 0.5 / 0.5 blend, in the reference's expression order).
// H8: the horizontal pass gives a lane 8 adjacent outputs (16 lanes per row) instead of 4: it reads (8 + 2R) floats for 8
// outputs where the 4-output form reads (4 + 2R) for 4 -- at R = 13, 10 float4 LDS reads per 8 outputs instead of 18 -- and
// the per-output share of its address arithmetic halves.  Lanes 32 B apart would collide on the LDS banks with the row
// below them in the same ds_read_b128 lane group, so rows at odd ring slots keep every pair of adjacent float4s swapped
// (float offset ^ 4): a lane still reads its own contiguous floats (in another order), the two rows of a lane group
// interleave on the banks, and every other access to the ring applies the same XOR.  Measured on 32 x 3840x2160: 2-3.5 %
// faster for R = 5 ... 10; at R >= 13 its 40 + 8 registers per item no longer fit 128 VGPRs beside the prefetch (spill),
// so those radii keep 4 outputs per lane.
// HPIPE (round 3): the horizontal pass takes its items two at a time and issues both items' LDS reads before the first FMA.
// Measured on 32 x 3840x2160 (profiles/blur_variants_r03_hpipe.log): R = 5 0.418 -> 0.398 ms, R = 7 0.439 -> 0.428, R = 8
// 0.454 -> 0.445; nothing at R = 10 and 13 (0.511 / 0.607 either way, also with 168 registers per lane) -- those layers sit at
// the FMA issue rate three waves per SIMD reach (tools/ubench/ubench_valu: 3.4 / 2.9 / 2.7 cycles per v_fma_f32 at 2 / 4 / 8
// waves), not at LDS latency.  On for R <= 8.
// (The vertical pass on the matrix cores -- v_mfma_f32_4x4x1 / 16x16x1 against the banded tap matrix, bit-identical, slower: the f32 matrix
// rate IS the vector rate -- was a template branch of this kernel in rounds 3-5; it now lives in tools/experiments/blur_mfma_vertical_r03.diff,
// results in profiles/blur_variants_r03_mfma_vertical.log.)
template <int R, int MINW = 4, int S_ = 32, bool DEC = false, bool ACT = false, int DBG = 0, int SEEDF = -1, bool H8 = (R <= 12),
          bool HPIPE = (R <= 7) || (R == 8 && !(DEC && ACT)) /* that one instantiation would spill 4 registers */>
__global__ __launch_bounds__(8 * S_, MINW) void blur_ring_kernel(const float *__restrict__ src, float *__restrict__ dst, int w, int h,
                                                             size_t src_frame_stride, size_t dst_frame_stride, TapWeights wt,
                                                             int n_frames, int ch_rows /* rows per chunk, a multiple of S */, Decimate dec, Activity act,
                                                             SeedSource seed) {
    using G = RingGeom<R, S_>;
    constexpr int NR = G::NR, LW = G::LW, RP = G::RP, S = G::S, RB = G::RB;
    constexpr bool SEED = SEEDF >= 0;
    constexpr int TLW = LW / 2 + 4, TLH = S / 2 + 2, NPX = (TLW * TLH + G::NTHR - 1) / G::NTHR;   // luma tile of a step (input pixels)
    static_assert(!SEED || (TLW * TLH <= (S - 2 * R) * LW && !DEC && !ACT), "the luma tile borrows the dead rows of a ring half");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // max|Eh| of (ring row, 8-column sub-cell): only with ACT
    auto x4 = [](int slot) { return H8 ? (slot & 1) << 2 : 0; };                 // XOR on a float offset inside ring row `slot` (see H8)
    auto ehm = [&](int slot, int sub) -> float & { return lds[LW * NR + slot * G::NSUB + sub]; };
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware 1-D order (see blur2_kernel): frame, chunk, strip with the strip index fastest
    const int tx = (w + G::TW - 1) / G::TW;
    const int nch = (h + ch_rows - 1) / ch_rows;
    const int total = tx * nch * n_frames;
    const int per_xcd = (total + 7) >> 3;
    const int t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (t >= total) return;
    const int frame = t / (tx * nch);
    const int rem = t - frame * (tx * nch);
    const int chunk = rem / tx, bx = rem - chunk * tx;
    const int x0 = bx * G::TW, ybeg = chunk * ch_rows;
    const int nst = (min(ch_rows, h - ybeg) + S - 1) / S;   // steps of this chunk
    const float *__restrict__ in = SEED ? nullptr : src + (size_t)frame * src_frame_stride;
    float *__restrict__ out = dst + (size_t)frame * dst_frame_stride;
    const unsigned char *px = SEED ? seed.pixels + (size_t)frame * seed.frame_stride : nullptr;
    // general staging of rows u in [u0, u1) (mirror resolved per element)
    auto stage_rows = [&](int u0, int u1) {
        for (int idx = tid; idx < (u1 - u0) * LW; idx += G::NTHR) {
            const int lu = idx / LW, lx = idx - lu * LW;
            const int slot = (u0 + lu + NR) & (NR - 1);
            const int sx = symm(x0 - RP + lx, w), sy = symm(ybeg + R + u0 + lu, h);
            float v = 0.0f;
            if (sx >= 0 && sy >= 0 && sx < w && sy < h) v = SEED ? seed_sample(px, seed, sx, sy, w, h) : in[(size_t)sy * w + sx];
            lds[slot * LW + (lx ^ x4(slot))] = v;
        }
    };
    // Row loads of the fast path.  8 lanes per row: lane q of a row takes the float4 columns q + 8 j (j < NPF4) and, when
    // the row length leaves a remainder, the float2 at float 32 NPF4 + 2 q -- one address register per side and immediate
    // offsets.  The mirror extension costs no branch: a row index is mirrored once per lane (symm) and columns by
    // load_quad_mirrored, so border strips and border rows issue exactly the loads interior ones do.
    // Needs: w a multiple of 4 and an image large enough for single reflections (a partial last strip is fine: the far halo
    // columns no output depends on are clamped into the row).
    const bool colfast = (w & 3) == 0 && w >= 64 && h >= 64;
    const int pf_row = tid >> 3, pf_q = tid & 7;
    auto load_rows = [&](int y_first, int nrows, f32x4 (&buf)[G::NPF4], f32x2 &rem) {     // image rows y_first + (0 ... nrows-1)
        const int sy = symm(y_first + min(pf_row, nrows - 1), h);
        const float *rowp = in + (size_t)sy * w;
#pragma unroll
        for (int j = 0; j < G::NPF4; j++) buf[j] = load_quad_mirrored(rowp, x0 - RP + 4 * pf_q + 32 * j, w);
        if (G::REM) rem = load_pair_mirrored(rowp, x0 - RP + 32 * G::NPF4 + 2 * pf_q, w);   // the last 16 halo columns, right of the strip
    };
    auto store_rows = [&](int u_first, int nrows, const f32x4 (&buf)[G::NPF4], const f32x2 &rem) {   // -> ring rows u_first + ...
        const int slot = (u_first + min(pf_row, nrows - 1) + NR) & (NR - 1);               // lanes past nrows repeat the last row
        float *rowp = lds + slot * LW;
        const int x = x4(slot);
#pragma unroll
        for (int j = 0; j < G::NPF4; j++) *reinterpret_cast<f32x4 *>(rowp + ((4 * pf_q + 32 * j) ^ x)) = buf[j];
        if (G::REM) *reinterpret_cast<f32x2 *>(rowp + ((32 * G::NPF4 + 2 * pf_q) ^ x)) = rem;
    };

    // Seed loader (SEED only).  Tile = luma of input rows ilo ... ilo + TLH - 1, columns clo ... clo + TLW - 1 (clamped to the
    // image: a clamped duplicate is never read, or is the mirrored neighbour BilinearUpScale.metal:33-48 asks for).
    const int wi = seed.in_w, hi = seed.in_h, clo = max(0, (x0 - RP) >> 1);
    auto tile_first_row = [&](int ya, int n) {              // lowest input row under output rows ya ... ya + n - 1 (mirrored)
        return (ya < 0 && ya + n > 0) ? 0 : min(symm(ya, h) >> 1, symm(ya + n - 1, h) >> 1);
    };
    auto load_pixels = [&](int ilo, unsigned (&raw)[NPX]) {
#pragma unroll
        for (int k = 0; k < NPX; k++) {
            const int idx = min(tid + k * G::NTHR, TLW * TLH - 1);
            const int r = idx / TLW, c = idx - r * TLW;
            raw[k] = raw_pixel(SEEDF, px, seed, min(clo + c, wi - 1), min(ilo + r, hi - 1));
        }
    };
    auto write_tile = [&](float *tile, const unsigned (&raw)[NPX]) {
#pragma unroll
        for (int k = 0; k < NPX; k++) tile[min(tid + k * G::NTHR, TLW * TLH - 1)] = luma_of(SEEDF, raw[k]);
    };
    auto expand_rows = [&](const float *tile, int ilo, int ya, int n, int u_first) {      // tile -> ring rows u_first + (0 ... n-1)
        const int lr = min(pf_row, n - 1);
        const int sy = symm(ya + lr, h);
        const int jm = sy >> 1, jp = min(jm + 1, hi - 1);   // BilinearUpScale.metal:29-48 for the exact 2x case
        const bool odd = (sy & 1) != 0;                     // fy = 0.5 (odd rows) or 0
        const float *tm = tile + (jm - ilo) * TLW - clo, *tp = tile + (jp - ilo) * TLW - clo;
        // vertical blend of input column c: fy * L(c, jp) + (1 - fy) * L(c, jm); with fy = 0 that is L(c, jm) exactly
        auto A = [&](int c) { const float cm = tm[c], cp = tp[c]; return odd ? 0.5f * cp + 0.5f * cm : cm; };
        const int slot = (u_first + lr + NR) & (NR - 1), x = x4(slot);
        float *rowp = lds + slot * LW;
#pragma unroll
        for (int j = 0; j < G::NPF4; j++) {
            const int gx = x0 - RP + 4 * pf_q + 32 * j;
            const bool mir = gx < 0 || gx >= w;
            const int g2 = gx < 0 ? -gx - 4 : (gx >= w ? 2 * w - 4 - gx : gx);
            const int c0 = g2 >> 1;
            const float a0 = A(c0), a1 = A(c0 + 1), a2 = A(min(c0 + 2, wi - 1));
            // even column: fx = 0 -> A(im); odd column: fx * A(ip) + (1 - fx) * A(im) with fx = 0.5
            const float o0 = a0, o1 = 0.5f * a1 + 0.5f * a0, o2 = a1, o3 = 0.5f * a2 + 0.5f * a1;
            f32x4 o;
            o.x = mir ? o3 : o0; o.y = mir ? o2 : o1; o.z = mir ? o1 : o2; o.w = mir ? o0 : o3;
            *reinterpret_cast<f32x4 *>(rowp + ((4 * pf_q + 32 * j) ^ x)) = o;
        }
        if (G::REM) {
            const int gx = x0 - RP + 32 * G::NPF4 + 2 * pf_q;
            const bool mir = gx >= w;
            const int c0 = (mir ? 2 * w - 2 - gx : gx) >> 1;
            const float a0 = A(c0), a1 = A(min(c0 + 1, wi - 1));
            const float o0 = a0, o1 = 0.5f * a1 + 0.5f * a0;
            f32x2 o;
            o.x = mir ? o1 : o0; o.y = mir ? o0 : o1;
            *reinterpret_cast<f32x2 *>(rowp + ((32 * G::NPF4 + 2 * pf_q) ^ x)) = o;
        }
    };

    // prologue: rows u in [-2R, S) of the first step.  Fast path: every load of both batches in flight before the first
    // LDS write (a load -> wait -> write loop cost nine memory latencies per chunk, as long as five steps).
    if (colfast && SEED) {
        float *tile = lds + S * LW;                          // rows S ... of the ring: written by neither pass
        unsigned ra[NPX], rb[NPX];
        const int ia = tile_first_row(ybeg - R, S), ib = tile_first_row(ybeg - R + S, 2 * R);
        load_pixels(ia, ra);
        load_pixels(ib, rb);
        write_tile(tile, ra);
        lds_barrier();
        expand_rows(tile, ia, ybeg - R, S, -2 * R);
        lds_barrier();
        write_tile(tile, rb);
        lds_barrier();
        expand_rows(tile, ib, ybeg - R + S, 2 * R, S - 2 * R);
    } else if (colfast) {
        f32x4 a[G::NPF4], b[G::NPF4];
        f32x2 ar, br;
        load_rows(ybeg - R, S, a, ar);
        load_rows(ybeg - R + S, 2 * R, b, br);
        store_rows(-2 * R, S, a, ar);
        store_rows(-2 * R + S, 2 * R, b, br);
    } else {
        stage_rows(-2 * R, S);
    }
    const VTapsSym<R> tw(wt);
    unsigned long long dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dlast = 0;
    auto stamp = [&](int k) {                                // DBG & 1: time since the previous stamp -> dsum[k]
        if (!(DBG & 1)) return;
        unsigned long long tnow;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        if (k >= 0) dsum[k] += tnow - dlast;
        dlast = tnow;
    };
    stamp(-1);
    auto body = [&](auto fast_tag, int st) {
        // MODE 2: FAST (prefetched rows, unguarded straight-line stores); 1: prefetched rows, guarded stores (a partial last
        // strip, a step that runs past the bottom of the image); 0: general (no next step to prefetch for, or an image
        // the fast loads do not cover)
        constexpr int MODE = decltype(fast_tag)::value;
        constexpr bool FAST = MODE >= 1, FULL = MODE == 2;
        const int y0 = ybeg + st * S;                        // first output row of this step
        const bool has_next = FAST || st + 1 < nst;
        lds_barrier();                                       // B1: this step's rows are in LDS
        stamp(7);
        f32x4 pf[G::NPF4];
        f32x2 pfr;
        unsigned raw[NPX];
        int tile_row0 = 0;
        if (FAST && (DBG & 4)) {
#pragma unroll
            for (int j = 0; j < G::NPF4; j++) pf[j] = f32x4{1.0f, 2.0f, 3.0f, (float)j};
            pfr = f32x2{1.0f, 2.0f};
        } else
        if (FAST && SEED) {                                  // the input pixels under the next step's S new rows -> registers
            tile_row0 = tile_first_row(ybeg + R + (st + 1) * S, S);
            load_pixels(tile_row0, raw);
        } else
        if (FAST) load_rows(ybeg + R + (st + 1) * S, S, pf, pfr);   // the next step's S new rows -> registers, all lanes alike

        stamp(0);
        // horizontal pass, in place (first step: the S + 2R prologue rows; later steps: the S new rows)
        const int hb = st == 0 ? -2 * R : st * S, hn = st == 0 ? S + 2 * R : S;
        if constexpr (H8) {
            constexpr int M0 = (RP - R) / 4, M1 = (RP + R + 7) / 4 + 1;
            static_assert((RP / 4) % 2 == 0, "the first output float4 of a lane must be an even one");
            // logical float4 m of a lane's segment sits at float4 m ^ (slot & 1): even m at +D, odd m at -D, D = 4 (slot & 1)
            auto h8_load = [&](int item, float (&v)[4 * (M1 - M0)]) {
                const int slot = (hb + (item >> 4) + NR) & (NR - 1), c8 = (item & 15) * 8;
                const float *rowp = lds + slot * LW + c8;
                const int D = x4(slot);
                const lds_cv_f32x4 *re = (const lds_cv_f32x4 *)(rowp + D), *ro = (const lds_cv_f32x4 *)(rowp - D);
#pragma unroll
                for (int m = M0; m < M1; m++) {
                    const f32x4 tv = (m & 1) ? ro[m] : re[m];
                    v[4 * (m - M0) + 0] = tv.x; v[4 * (m - M0) + 1] = tv.y; v[4 * (m - M0) + 2] = tv.z; v[4 * (m - M0) + 3] = tv.w;
                }
            };
            auto h8_finish = [&](int item, const float (&v)[4 * (M1 - M0)]) {
                const int slot = (hb + (item >> 4) + NR) & (NR - 1), c8 = (item & 15) * 8;
                float *rowp = lds + slot * LW + c8;
                const int D = x4(slot);
                float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int i = 0; i < G::NT; i++) {
#pragma unroll
                    for (int k = 0; k < 8; k++) acc[k] = fmaf(tw(i), v[(RP - R - 4 * M0) + k + i], acc[k]);
                }
                *reinterpret_cast<float4 *>(rowp + RP + D) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                *reinterpret_cast<float4 *>(rowp + RP + 4 - D) = make_float4(acc[4], acc[5], acc[6], acc[7]);
                if (ACT) {                                  // max |hb - raw| over this lane's 8 columns = one sub-cell
                    constexpr int C = RP - 4 * M0;
                    float e = fabsf(acc[0] - v[C + 0]);
#pragma unroll
                    for (int k = 1; k < 8; k++) e = fmaxf(e, fabsf(acc[k] - v[C + k]));
                    ehm(slot, item & 15) = e;
                }
            };
            const int n_items = (DBG & 8) ? 0 : hn * 16;
            if constexpr (HPIPE) {
                // Two items per trip, both items' LDS reads issued before the first FMA: the second item's read latency runs
                // under the first item's 8 (2R + 1) FMAs instead of in front of its own (a row is still read and written by one
                // wavefront within one trip, so the in-place update stays safe).  Two items' operands (2 x (16 + 2R) registers) fit the
                // 128-VGPR budget of four workgroups per CU up to R = 8, which is where the template default turns it on (R <= 7, and
                // R = 8 unless DEC && ACT); at R >= 9 it needs the 168 VGPRs of three workgroups per CU and measured no gain there.
#pragma unroll 1
                for (int item = tid; item < n_items; item += 2 * G::NTHR) {
                    float va[4 * (M1 - M0)], vb[4 * (M1 - M0)];
                    const int item_b = item + G::NTHR;
                    const bool has_b = item_b < n_items;             // wave-uniform (n_items is a multiple of 64)
                    h8_load(item, va);
                    if (has_b) h8_load(item_b, vb);
                    h8_finish(item, va);
                    if (has_b) h8_finish(item_b, vb);
                }
            } else {
#pragma unroll 1
                for (int item = tid; item < n_items; item += G::NTHR) {
                    float v[4 * (M1 - M0)];
                    h8_load(item, v);
                    h8_finish(item, v);
                }
            }
        } else {
#pragma unroll 1
        for (int item = tid; item < ((DBG & 8) ? 0 : hn * 32); item += G::NTHR) {
            const int slot = (hb + (item >> 5) + NR) & (NR - 1), c4 = (item & 31) * 4;
            float *rowp = lds + slot * LW;
            constexpr int M0 = (RP - R) / 4, M1 = (RP + R + 3) / 4 + 1;    // float4 of the row segment that hold the taps' operands
            float v[4 * (M1 - M0)];
            const lds_cv_f32x4 *rp4 = (const lds_cv_f32x4 *)(rowp + c4);
#pragma unroll
            for (int m = M0; m < M1; m++) {
                const f32x4 tv = rp4[m];
                v[4 * (m - M0) + 0] = tv.x; v[4 * (m - M0) + 1] = tv.y; v[4 * (m - M0) + 2] = tv.z; v[4 * (m - M0) + 3] = tv.w;
            }
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < G::NT; i++) {
#pragma unroll
                for (int k = 0; k < 4; k++) acc[k] = fmaf(tw(i), v[(RP - R - 4 * M0) + k + i], acc[k]);
            }
            *reinterpret_cast<float4 *>(rowp + RP + c4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            if (ACT) {                                      // max |hb - raw| over an 8-column sub-cell = this lane's 4 columns and its neighbour's
                constexpr int C = RP - 4 * M0;
                float e = fmaxf(__builtin_fmaxf(__builtin_fmaxf(fabsf(acc[0] - v[C + 0]), fabsf(acc[1] - v[C + 1])), fabsf(acc[2] - v[C + 2])), fabsf(acc[3] - v[C + 3]));
                e = fmaxf(e, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, e), 0x111 /* row_shr:1 */, 0xf, 0xf, true)));
                if (tid & 1) ehm(slot, (item & 31) >> 1) = e;
            }
        }
        }
        stamp(1);
        lds_barrier();                                       // B2: blurred rows complete
        stamp(2);

        // vertical pass: wavefront wv owns output rows wv RB ... wv RB + RB - 1 of the step, a lane 2 columns
        {
            const int u0 = st * S + wv * RB - 2 * R;         // first window row (wave-uniform)
            const float *colp = lds + RP + 2 * lane, *colx = lds + ((RP + 2 * lane) ^ (H8 ? 4 : 0));   // even / odd ring slots (u0 is even: slot parity = k & 1)
            f32x2 cen[ACT ? RB : 1];                        // hb under each output (the centre tap's operand), for the activity bound
            f32x2 acc[RB];
#pragma unroll
            for (int rr = 0; rr < RB; rr++) { acc[rr].x = 0.0f; acc[rr].y = 0.0f; }
#pragma unroll
            for (int k = 0; k < ((DBG & 16) ? 0 : RB + 2 * R); k++) {
                const int slot = (u0 + k + NR) & (NR - 1);
                const f32x2 v = *(const lds_cv_f32x2 *)(((k & 1) ? colx : colp) + slot * LW);
#pragma unroll
                for (int rr = 0; rr < RB; rr++) {
                    const int i = k - rr;
                    if (ACT && i == R) cen[rr] = v;
                    if (i >= 0 && i < G::NT) {
                        acc[rr].x = fmaf(tw(i), v.x, acc[rr].x);
                        acc[rr].y = fmaf(tw(i), v.y, acc[rr].y);
                    }
                }
            }
            // pin the accumulators: otherwise LLVM sinks each row's FMA chain into the store guards below
#pragma unroll
            for (int rr = 0; rr < RB; rr++) asm volatile("" : "+v"(acc[rr].x), "+v"(acc[rr].y));
            unsigned act_mask = 0;                          // wave-uniform: bit rr / RB + rr = cell 0 / 1 of output row rr is active
            float eh_row[ACT ? RB : 1];                     // max|Eh| of this lane's cell under each output row: all RB LDS reads in one batch
            if (ACT) {
#pragma unroll
                for (int rr = 0; rr < RB; rr++) eh_row[rr] = ehm((st * S + wv * RB + rr - R + NR) & (NR - 1), lane >> 2);
            }
            stamp(3);
            const int gx = x0 + 2 * lane;
#pragma unroll
            for (int rr = 0; rr < ((DBG & 2) ? 0 : RB); rr++) {
                const int gy = y0 + wv * RB + rr;            // wave-uniform
                if (!FULL && gy >= h) continue;
                float *o = out + (size_t)gy * w + gx;
                if (FULL || (gx + 1 < w && (w & 1) == 0)) {
                    ring_store2(o, acc[rr]);
                } else {
                    if (gx + 0 < w) o[0] = acc[rr].x;
                    if (gx + 1 < w) o[1] = acc[rr].y;
                }
                if (DEC && (FULL ? ((wv * RB + rr) & 1) == 0 : (((gy & 1) == 0) && (gy >> 1) < dec.h2))) {   // y0 is even on the FAST path; gx is even
                    if (FULL || ((gx >> 1) < dec.w2 && gx < w))
                        dec.dst[(size_t)frame * dec.frame_stride + (size_t)(gy >> 1) * dec.w2 + (gx >> 1)] = acc[rr].x;
                }
                if (ACT) {                                  // 32 lanes = 64 columns = one cell of this row
                    const float eh = eh_row[rr];            // ring row (st S + wv RB + rr - R) under this output row
                    const float lim = act.thr * 0.9999f;
                    const bool f = ((FULL || gx + 0 < w) && fabsf(acc[rr].x - cen[rr].x) + eh > lim) ||
                                   ((FULL || gx + 1 < w) && fabsf(acc[rr].y - cen[rr].y) + eh > lim);
                    const unsigned long long b = __ballot(f);
                    act_mask |= (((unsigned)b != 0u) ? 1u : 0u) << rr | (((unsigned)(b >> 32) != 0u) ? 1u : 0u) << (RB + rr);
                }
            }
            if (ACT) {                                      // one store for the wave's RB rows x 2 cells: lanes 0 ... RB-1 and 32 ... 32+RB-1
                const int half = lane >> 5, rr = lane & 31;
                const int gy = y0 + wv * RB + rr, cell = (x0 >> 6) + half;
                if (rr < RB && gy < h && cell < act.ncell)
                    act.dst[(size_t)frame * act.frame_stride + (size_t)gy * act.ncell + cell] = (unsigned char)((act_mask >> (half * RB + rr)) & 1u);
            }
        }
        stamp(4);
        if (!has_next) return;                               // uniform
        lds_barrier();                                       // B3: every read of the other ring half is done
        stamp(5);
        if (FAST && SEED) {
            float *tile = lds + ((st & 1) * S) * LW;         // first S - 2R rows of the current half: no later step reads them
            write_tile(tile, raw);
            lds_barrier();
            expand_rows(tile, tile_row0, ybeg + R + (st + 1) * S, S, (st + 1) * S);
        } else
        if (FAST) store_rows((st + 1) * S, S, pf, pfr);
        else stage_rows((st + 1) * S, (st + 2) * S);        // general step: mirror per element, straight to LDS
        // keeps LLVM from tail-merging the LDS writes above of the two instantiations: merged, they would be reached from
        // the general body as well and get its conservative s_waitcnt vmcnt(0)
        if (MODE == 2) asm volatile("; end of a FAST ring step" ::: "memory");
        else if (MODE == 1) asm volatile("; end of a prefetching ring step with guarded stores" ::: "memory");
        else asm volatile("; end of a general ring step" ::: "memory");
        if (DBG & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(6);
    };
    // FAST steps: a whole strip of a column-fast image, a next step to prefetch for, all S output rows inside the image (and,
    // with DEC, an even first row and a decimated row for every even output row).  A partial last strip and a step that
    // runs past the bottom still prefetch (MODE 1); only the last step of a chunk and tiny / odd-width images are general.
    const bool strip_full = colfast && x0 + G::TW <= w && (!DEC || ((ybeg & 1) == 0 && (S & 1) == 0));
    for (int st = 0; st < nst; st++) {
        const int y0 = ybeg + st * S;
#ifdef SIFTMI_RING_NO_MODE1                                  // tools/ubench experiment only
        const bool pre = false;
#else
        const bool pre = colfast && st + 1 < nst;
#endif
        const bool full = colfast && st + 1 < nst && strip_full && y0 + S <= h && (!DEC || ((y0 + S - 1) >> 1) < dec.h2);
        if (full) body(std::integral_constant<int, 2>{}, st);
        else if (pre) body(std::integral_constant<int, 1>{}, st);
        else body(std::integral_constant<int, 0>{}, st);
    }
    if ((DBG & 1) && lane == 0) {
        unsigned long long *d = reinterpret_cast<unsigned long long *>(act.dst) + ((size_t)t * 4 + wv) * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = dsum[k];
    }
}

// ------------------------------------------------------------------------------------------------
// SIFTMI_FMT_GRAYF32 input contract (include/siftmi.h): every pixel in [0, 1].  One pass over the frames of a sub-batch (4 B per
// input pixel, float input only); any value outside the range, or a NaN, sets bit 5 of the call's overflow flags.
__global__ __launch_bounds__(256) void check_unit_range_kernel(const unsigned char *__restrict__ pixels, size_t row_stride, size_t frame_stride,
                                                              int w, int h, int n_frames, int32_t *__restrict__ flags) {
    const long long total = (long long)w * h * n_frames;
    bool bad = false;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int f = (int)(i / ((long long)w * h));
        const int r = (int)(i - (long long)f * w * h);
        const int y = r / w, x = r - y * w;
        const float v = *reinterpret_cast<const float *>(pixels + (size_t)f * frame_stride + (size_t)y * row_stride + 4 * (size_t)x);
        bad = bad || !(v >= 0.0f && v <= 1.0f);
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flags, 32);
}

// ------------------------------------------------------------------------------------------------
// Plain float4 streaming copy: the measured HBM ceiling bench.py quotes the pyramid kernel against (siftmi_time_copy).
// One float4 per lane, one pass, workgroups in address order -- the fastest of the forms tools/ubench/ubench_copy.hip compares on
// this hardware (6.2 TB/s; 6.5 with non-temporal loads and stores; grid-stride loops and several float4 per lane: 4.3-5.8).
template <bool NT>
__global__ __launch_bounds__(256) void copy_f4_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
    else dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------------
// Up to three consecutive Gaussian layers of an octave in one launch (a frame or two per call).  As one launch per layer, every
// layer of a single 1920x1080 frame is a 5-30 us kernel that re-stages its input, plus a dependency gap, on the call's critical
// path (seed -> three layers each of octaves 0, 1, 2 -> octave 3's whole chain).  Here a workgroup keeps a T x T tile with a
// halo of HALO = RA + RB + RC pixels in LDS and runs the cascaded blurs on it in place, the valid margin shrinking by R per
// layer: 1.6 x (T = 64) to 2.4 x (T = 32) the arithmetic of the per-layer launches, one staging and one launch instead of three.
// Same arithmetic as blur2_kernel -- horizontal pass, then vertical pass, k-ordered fmaf chains -- so the layers are
// bit-identical.  Mirror extension (Common.hpp:15-22): every layer blurs the symmetric extension of the layer before it, and a
// reflected position's own blur would sum its taps in reverse order (other roundings), so after each layer the cells of the
// region outside the image are overwritten with the values at their mirrored positions (which the tile holds: HALO < T).
// Needs w % 4 == 0 and w, h >= 64 (one reflection reaches every input).
struct ChainWeights { TapWeights l[3]; };

template <int T_, int NTHR_, int RA, int RB, int RC>
struct ChainGeom {
    static constexpr int T = T_, NTHR = NTHR_;
    static constexpr int HALO = RA + RB + RC;
    static constexpr int HP = (HALO + 3) & ~3;                  // staged halo: whole float4s
    static constexpr int RW = T + 2 * HP;                       // region (rows and columns)
    static constexpr int PADC = 16;                             // columns either side that the horizontal pass may read (never uses)
    static constexpr int LW = RW + 2 * PADC, LH = RW + 4;       // + 4 rows the last vertical row group may read
    static constexpr size_t lds_bytes = (size_t)LW * LH * sizeof(float);
    static_assert(T % 4 == 0 && HALO < T && RA > 0, "the mirrored source of every halo cell lies in the tile");
    static_assert(RA <= 16 && RB <= 16 && RC <= 16, "PADC");
};

// one layer, in place on the region in LDS.  MP = margin (cells beyond the tile, every side) on which the input layer is valid.
template <typename G, int R, int MP, bool LAST>
__device__ __forceinline__ void chain_layer(float *lds, const TapWeights &wt, float *__restrict__ out, int w, int h, int x0, int y0,
                                            bool border, bool do_dec, float *__restrict__ dec_out, int w2, int h2) {
    constexpr int MC = MP - R, NT = 2 * R + 1;
    static_assert(MC >= 0, "halo");
    const int tid = threadIdx.x;
    const VTaps<NT> tw(wt);
    // horizontal pass, in place: the rows on which the input is valid; 4 adjacent outputs per lane, a row's lanes in one wavefront
    constexpr int HR0 = G::HP - MP, HR1 = G::HP + G::T + MP;
    constexpr int G0 = (G::HP - MC) / 4, G1 = (G::HP + G::T + MC + 3) / 4;       // float4 column groups [G0, G1)
    constexpr int LPR = (G1 - G0) <= 16 ? 16 : 32, LSH = LPR == 16 ? 4 : 5;      // lanes per row
    static_assert(G1 - G0 <= 32 && HR0 >= 0 && HR1 <= G::RW, "a row's items fit half a wavefront");
    constexpr int ML = -((R + 3) / 4), MH = (R + 3) / 4;                          // float4s around the lane's own: ML ... MH
    for (int it = tid; it < (HR1 - HR0) * LPR; it += G::NTHR) {
        const int row = HR0 + (it >> LSH), g = G0 + (it & (LPR - 1));
        if (g < G1) {
            float *rowp = lds + row * G::LW + G::PADC + 4 * g;
            const lds_cv_f32x4 *rp4 = (const lds_cv_f32x4 *)rowp;
            float v[4 * (MH - ML + 1)];
#pragma unroll
            for (int m = ML; m <= MH; m++) {
                const f32x4 t = rp4[m];
                v[4 * (m - ML) + 0] = t.x; v[4 * (m - ML) + 1] = t.y; v[4 * (m - ML) + 2] = t.z; v[4 * (m - ML) + 3] = t.w;
            }
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < NT; i++) {
#pragma unroll
                for (int k = 0; k < 4; k++) acc[k] = fmaf(tw.w[i], v[(-R - 4 * ML) + k + i], acc[k]);
            }
            *reinterpret_cast<float4 *>(rowp) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
    }
    __syncthreads();
    // vertical pass: ONE item (4 columns x RBV rows) per thread, results held in registers until every thread has read its
    // window; RBV = the fewest rows per item that still make one round (more threads busy: a small octave is one workgroup per CU)
    constexpr int VR0 = G::HP - MC, VROWS = G::T + 2 * MC;
    constexpr int RBV = VROWS * LPR <= G::NTHR ? 1 : ((VROWS + 1) / 2 * LPR <= G::NTHR ? 2 : 4), NRG = (VROWS + RBV - 1) / RBV;
    static_assert(NRG * LPR <= G::NTHR && VR0 - R >= 0 && VR0 + RBV * NRG + R <= G::LH, "one round");
    const int g = G0 + (tid & (LPR - 1)), rgi = tid >> LSH;
    const bool vact = rgi < NRG && g < G1;
    float4 acc[RBV];
    if (vact) {
        const float *colp = lds + (VR0 + RBV * rgi - R) * G::LW + G::PADC + 4 * g;
#pragma unroll
        for (int rr = 0; rr < RBV; rr++) acc[rr] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int k = 0; k < RBV + 2 * R; k++) {
            const float4 v = *reinterpret_cast<const float4 *>(colp + k * G::LW);
#pragma unroll
            for (int rr = 0; rr < RBV; rr++) {
                const int i = k - rr;
                if (i >= 0 && i < NT) {
                    acc[rr].x = fmaf(tw.w[i], v.x, acc[rr].x);
                    acc[rr].y = fmaf(tw.w[i], v.y, acc[rr].y);
                    acc[rr].z = fmaf(tw.w[i], v.z, acc[rr].z);
                    acc[rr].w = fmaf(tw.w[i], v.w, acc[rr].w);
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < RBV; rr++) asm volatile("" : "+v"(acc[rr].x), "+v"(acc[rr].y), "+v"(acc[rr].z), "+v"(acc[rr].w));
        // the tile itself goes to the layer in global memory (w % 4 == 0: a float4 is inside the image or outside it)
        const int gx = x0 - G::HP + 4 * g;
        if (4 * g >= G::HP && 4 * g < G::HP + G::T && gx < w) {
#pragma unroll
            for (int rr = 0; rr < RBV; rr++) {
                const int ry = VR0 + RBV * rgi + rr, gy = y0 - G::HP + ry;
                if (ry < G::HP || ry >= G::HP + G::T || gy >= h) continue;
                *reinterpret_cast<float4 *>(out + (size_t)gy * w + gx) = acc[rr];
                if (do_dec && (gy & 1) == 0 && (gy >> 1) < h2) {      // next octave's layer 0 (NearestNeighborDownScale.metal:15-22)
                    float *o2 = dec_out + (size_t)(gy >> 1) * w2 + (gx >> 1);
                    if ((gx >> 1) + 0 < w2) o2[0] = acc[rr].x;
                    if ((gx >> 1) + 1 < w2) o2[1] = acc[rr].z;
                }
            }
        }
    }
    if (LAST) return;
    __syncthreads();
    if (vact) {
        float *colp = lds + (VR0 + RBV * rgi) * G::LW + G::PADC + 4 * g;
#pragma unroll
        for (int rr = 0; rr < RBV; rr++) *reinterpret_cast<float4 *>(colp + rr * G::LW) = acc[rr];
    }
    __syncthreads();
    if (border && MC > 0) {                                  // uniform: the region reaches past an image border
        // Cells outside the image <- their mirrored positions: columns first (on every row), then whole rows (corners then hold
        // the value mirrored both ways).  Only the MC cells next to a border can feed an output; a cell further out (partial
        // last tiles) keeps whatever the passes left there.
        constexpr int C0 = G::HP - MC, C1 = G::HP + G::T + MC, NV = C1 - C0, MCD = MC > 0 ? MC : 1;
        const int cz = G::HP - x0, cw = w - (x0 - G::HP);   // region column of image column 0 / of the first column right of the image
        const int rz = G::HP - y0, rh = h - (y0 - G::HP);
        if (cz > C0 || cw < C1) {
            for (int idx = tid; idx < NV * MCD; idx += G::NTHR) {
                const int r = C0 + idx / MCD, j = idx % MCD;
                float *rowp = lds + r * G::LW + G::PADC;
                if (cz - 1 - j >= C0) rowp[cz - 1 - j] = rowp[cz + j];
                if (cw + j < C1) rowp[cw + j] = rowp[cw - 1 - j];
            }
            __syncthreads();
        }
        if (rz > C0 || rh < C1) {
            for (int idx = tid; idx < NV * MCD; idx += G::NTHR) {
                const int j = idx / NV, cc = C0 + idx % NV;
                float *colp = lds + G::PADC + cc;
                if (rz - 1 - j >= C0) colp[(rz - 1 - j) * G::LW] = colp[(rz + j) * G::LW];
                if (rh + j < C1) colp[(rh + j) * G::LW] = colp[(rh - 1 - j) * G::LW];
            }
            __syncthreads();
        }
    }
}

template <int T_, int NTHR_, int RA, int RB, int RC>
__global__ __launch_bounds__(NTHR_) void blur_chain_kernel(float *__restrict__ layers /* the octave's layer 0, frame 0 */, int w, int h,
                                                           size_t frame_stride, size_t layer_stride, int first /* input layer */, ChainWeights wts,
                                                           int n_frames, int dec_layer /* the layer that also emits the next octave's layer 0, or 0 */,
                                                           Decimate dec) {
    using G = ChainGeom<T_, NTHR_, RA, RB, RC>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int tx = (w + G::T - 1) / G::T, ty = (h + G::T - 1) / G::T;
    const int total = tx * ty * n_frames;
    const int per_xcd = (total + 7) >> 3;
    const int t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);      // XCD-aware 1-D order, see blur2_kernel
    if (t >= total) return;
    const int frame = t / (tx * ty);
    const int rem = t - frame * (tx * ty);
    const int by = rem / tx, bx = rem - by * tx;
    const int x0 = bx * G::T, y0 = by * G::T;
    float *__restrict__ base = layers + (size_t)frame * frame_stride + (size_t)first * layer_stride;
    float *__restrict__ dec_out = dec.dst ? dec.dst + (size_t)frame * dec.frame_stride : nullptr;
    const bool border = x0 - G::HP < 0 || y0 - G::HP < 0 || x0 + G::T + G::HP > w || y0 + G::T + G::HP > h;
    {   // input layer -> region, mirror extension by row (symm) and by float4 (load_quad_mirrored); every load before the first LDS store
        constexpr int Q = G::RW / 4, NB = (G::RW * Q + G::NTHR - 1) / G::NTHR;
        f32x4 buf[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const int idx = min(tid + b * G::NTHR, G::RW * Q - 1);
            const int row = idx / Q, q = idx - row * Q;
            const float *rowp = base + (size_t)min(max(symm(y0 - G::HP + row, h), 0), h - 1) * w;
            buf[b] = load_quad_mirrored(rowp, x0 - G::HP + 4 * q, w);
        }
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const int idx = min(tid + b * G::NTHR, G::RW * Q - 1);
            const int row = idx / Q, q = idx - row * Q;
            *reinterpret_cast<f32x4 *>(lds + row * G::LW + G::PADC + 4 * q) = buf[b];
        }
    }
    __syncthreads();
    constexpr int M0 = G::HP, M1 = M0 - RA, M2 = M1 - RB;
    chain_layer<G, RA, M0, RB == 0>(lds, wts.l[0], base + 1 * layer_stride, w, h, x0, y0, border, dec_layer == first + 1, dec_out, dec.w2, dec.h2);
    if constexpr (RB > 0)
        chain_layer<G, RB, M1, RC == 0>(lds, wts.l[1], base + 2 * layer_stride, w, h, x0, y0, border, dec_layer == first + 2, dec_out, dec.w2, dec.h2);
    if constexpr (RC > 0)
        chain_layer<G, RC, M2, true>(lds, wts.l[2], base + 3 * layer_stride, w, h, x0, y0, border, dec_layer == first + 3, dec_out, dec.w2, dec.h2);
}

// Shipping geometry (tools/ubench/blur_variants.hip on 8 x 3840x2160, MI355X): 128 x 32 tiles, 256
// threads, 4 output rows per lane in the vertical pass, XCD-aware 1-D tile order.
template <int R>
struct BlurShip {
    static constexpr int TH = 32, NTHR = 256, RB = 4;
    using G = Blur2Geom<R, TH, NTHR, 4, RB>;
};

}  // namespace siftmi
