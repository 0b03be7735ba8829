// dense_kernels.hip.h -- Gaussian scale-space kernels for gfx950 (MI355X).
//
// What the reference computes (one dispatch per pass, full texture round trip each):
//   convertSRGBToGrayscale  Sources/MetalShaders/Metal/ConvertSRGBToGrayscale.metal:11-23
//   bilinearUpScale         Sources/MetalShaders/Metal/BilinearUpScale.metal:12-64
//   convolutionX/Y          Sources/MetalShaders/Metal/Convolution.metal:15-52        (seed blur)
//   convolutionSeriesX/Y    Sources/MetalShaders/Metal/ConvolutionSeries.metal:16-53  (layer blurs)
//   nearestNeighborDownScale Sources/MetalShaders/Metal/NearestNeighborDownScale.metal:15-22
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

namespace siftmi {

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

struct SeedSource {                 // input frame description for the seed loader
    const unsigned char *pixels;    // frame 0
    size_t frame_stride;            // bytes between frames
    size_t row_stride;              // bytes between rows
    int format;
    int in_w, in_h;                 // input size (W, H); the seed image is 2W x 2H
};

// luma of input pixel (x, y): ConvertSRGBToGrayscale.metal:17-20 on bgra8Unorm texels (byte/255)
__device__ __forceinline__ float luma_at(const unsigned char *frame, const SeedSource &s, int x, int y) {
    if (x < 0 || y < 0 || x >= s.in_w || y >= s.in_h) return 0.0f;
    const unsigned char *row = frame + (size_t)y * s.row_stride;
    if (s.format == FMT_BGRA8) {
        const uchar4 p = *reinterpret_cast<const uchar4 *>(row + 4 * (size_t)x);
        const float b = (float)p.x / 255.0f, g = (float)p.y / 255.0f, r = (float)p.z / 255.0f;
        return 0.0f + (0.212639005871510f * r) + (0.715168678767756f * g) + (0.072192315360734f * b);
    } else if (s.format == FMT_GRAY8) {
        return (float)row[x] / 255.0f;
    }
    return reinterpret_cast<const float *>(row)[x];
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

template <int R>
struct BlurGeom {
    static constexpr int RP = (R + 3) & ~3;       // halo rounded up so LDS rows stay 16-B aligned
    static constexpr int TW = 128;                // tile width  (outputs)
    static constexpr int TH = 32;                 // tile height (outputs)
    static constexpr int RB = TH / 8;             // output rows per lane in the vertical pass
    static constexpr int LW = TW + 2 * RP;        // LDS row pitch (floats)
    static constexpr int LH = TH + 2 * R;         // LDS rows
    static constexpr int NT = 2 * R + 1;          // taps
    static constexpr size_t lds_bytes = (size_t)LW * LH * sizeof(float);
};

// One Gaussian layer: dst = blur_R(src), separable, mirror extension, one frame per blockIdx.z.
// SEED = true: src is ignored and the input is seed_sample() of the frame's pixels.
template <int R, bool SEED>
__global__ __launch_bounds__(256) void blur_layer_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                        int w, int h, size_t src_frame_stride, size_t dst_frame_stride,
                                                        TapWeights wt, SeedSource seed) {
    using G = BlurGeom<R>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * G::TW, y0 = blockIdx.y * G::TH;
    const int frame = blockIdx.z;
    const float *__restrict__ in = SEED ? nullptr : src + (size_t)frame * src_frame_stride;
    float *__restrict__ out = dst + (size_t)frame * dst_frame_stride;
    const unsigned char *px = SEED ? seed.pixels + (size_t)frame * seed.frame_stride : nullptr;

    // ---- phase 0: stage the input window, mirror extension resolved here ---------------------
    const bool interior = (x0 - G::RP >= 0) && (x0 + G::TW + G::RP <= w) && (y0 - R >= 0) && (y0 + G::TH + R <= h);
    if (!SEED && interior && (w & 3) == 0) {
        constexpr int V = G::LW / 4;                       // float4 per LDS row
        for (int idx = tid; idx < G::LH * V; idx += 256) {
            const int ly = idx / V, lv = idx - ly * V;
            const float4 v = *reinterpret_cast<const float4 *>(in + (size_t)(y0 - R + ly) * w + (x0 - G::RP) + 4 * lv);
            *reinterpret_cast<float4 *>(lds + ly * G::LW + 4 * lv) = v;
        }
    } else {
        for (int idx = tid; idx < G::LH * G::LW; idx += 256) {
            const int ly = idx / G::LW, lx = idx - ly * G::LW;
            const int sx = symm(x0 - G::RP + lx, w), sy = symm(y0 - R + ly, h);
            float v;
            if (SEED) v = seed_sample(px, seed, sx, sy, w, h);
            else v = (sx < 0 || sy < 0 || sx >= w || sy >= h) ? 0.0f : in[(size_t)sy * w + sx];
            lds[idx] = v;
        }
    }
    __syncthreads();

    // ---- phase 1: horizontal pass, in place.  A row's 32 items sit in one half-wave, so every
    // lane's reads of the row are issued before any lane's write of it (in-order LDS per wave).
    for (int item = tid; item < G::LH * (G::TW / 4); item += 256) {
        const int row = item >> 5, c4 = (item & 31) * 4;
        float *rowp = lds + row * G::LW;
        float v[4 + 2 * G::RP];
#pragma unroll
        for (int m = 0; m < (4 + 2 * G::RP) / 4; m++) {
            const float4 t = *reinterpret_cast<const float4 *>(rowp + c4 + 4 * m);
            v[4 * m + 0] = t.x; v[4 * m + 1] = t.y; v[4 * m + 2] = t.z; v[4 * m + 3] = t.w;
        }
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < G::NT; i++) {
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k] = fmaf(wt.w[i], v[(G::RP - R) + k + i], acc[k]);
        }
        *reinterpret_cast<float4 *>(rowp + G::RP + c4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    __syncthreads();

    // ---- phase 2: vertical pass, 4 columns x RB rows per lane, taps in increasing order --------
    {
        const int cg = tid & 31, rg = tid >> 5;
        const float *colp = lds + (rg * G::RB) * G::LW + G::RP + cg * 4;
        float4 acc[G::RB];
#pragma unroll
        for (int rr = 0; rr < G::RB; rr++) acc[rr] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int k = 0; k < G::RB + 2 * R; k++) {
            const float4 v = *reinterpret_cast<const float4 *>(colp + k * G::LW);
#pragma unroll
            for (int rr = 0; rr < G::RB; rr++) {
                const int i = k - rr;
                if (i >= 0 && i < G::NT) {
                    acc[rr].x = fmaf(wt.w[i], v.x, acc[rr].x);
                    acc[rr].y = fmaf(wt.w[i], v.y, acc[rr].y);
                    acc[rr].z = fmaf(wt.w[i], v.z, acc[rr].z);
                    acc[rr].w = fmaf(wt.w[i], v.w, acc[rr].w);
                }
            }
        }
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
        }
    }
}

// NearestNeighborDownScale.metal:15-22: out[y][x] = in[2y][2x] (previous octave's layer nspo)
__global__ __launch_bounds__(256) void downsample_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                        int sw, int sh, int dw, int dh,
                                                        size_t src_frame_stride, size_t dst_frame_stride) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    const float *in = src + (size_t)blockIdx.z * src_frame_stride;
    const int sx = 2 * x, sy = 2 * y;
    const float v = (sx < sw && sy < sh) ? in[(size_t)sy * sw + sx] : 0.0f;
    dst[(size_t)blockIdx.z * dst_frame_stride + (size_t)y * dw + x] = v;
}

}  // namespace siftmi
