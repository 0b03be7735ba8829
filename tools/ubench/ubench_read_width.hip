// Microbenchmark: HBM read bandwidth of a streaming kernel by LOAD WIDTH (dword / dwordx2 / dwordx4 per lane) at a fixed number of
// wave-loads in flight -- is the memory pipeline's depth counted in instructions (so that narrow loads cap the bandwidth) or bytes?
// Shape of the extrema scan's read path: each wavefront walks rows; per row it issues 6 independent loads (one per Gaussian layer,
// planes 8 MB apart) and keeps 3 rows (18 loads) in flight.
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_read_width ubench_read_width.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int W> struct Vec;
template <> struct Vec<1> { typedef float T; };
template <> struct Vec<2> { typedef float2 T; };
template <> struct Vec<4> { typedef float4 T; };
__device__ inline float sum(float v) { return v; }
__device__ inline float sum(float2 v) { return v.x + v.y; }
__device__ inline float sum(float4 v) { return v.x + v.y + v.z + v.w; }

// image rows of `w` floats, 6 layer planes of n = w * h floats; a block of 4 wavefronts covers 4 * 64 * W columns and walks EH rows
// MIS: a wavefront covers 62 columns and starts one column early (the extrema scan's halo lanes): unaligned 256-byte wave loads;
// LDSB: bytes of LDS per block (occupancy limiter)
template <int W, int AHEAD, bool MIS = false, int LDSB = 0>
__global__ __launch_bounds__(256) void k(const float *__restrict__ g, int w, int h, size_t n, size_t frame_stride, int EH, float *out) {
    typedef typename Vec<W>::T V;
    __shared__ float pad[LDSB / 4 + 1];
    if (LDSB && threadIdx.x == 0 && EH < 0) pad[EH & 7] = 1.f;
    int x = (blockIdx.x * 256 + threadIdx.x) * W;
    if (MIS) x = min(max((int)(blockIdx.x * 248 + (threadIdx.x >> 6) * 62 + (threadIdx.x & 63)) - 1, 0), w - 1);
    if (x >= w) return;
    const int y0 = blockIdx.y * EH, y1 = min(y0 + EH, h);
    const float *base = g + blockIdx.z * frame_stride + x;
    float acc = 0.f;
    V buf[AHEAD][6];
#pragma unroll
    for (int a = 0; a < AHEAD; a++)
#pragma unroll
        for (int l = 0; l < 6; l++) buf[a][l] = *(const V *)(base + l * n + (size_t)min(y0 + a, h - 1) * w);
    for (int y = y0; y < y1; y += AHEAD) {
#pragma unroll
        for (int a = 0; a < AHEAD; a++) {
#pragma unroll
            for (int l = 0; l < 6; l++) {
                acc += sum(buf[a][l]);
                buf[a][l] = *(const V *)(base + l * n + (size_t)min(y + AHEAD + a, h - 1) * w);
            }
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int W, int AHEAD, bool MIS = false, int LDSB = 0>
static float run(const float *g, int w, int h, int frames, float *out) {
    const int EH = 33;
    dim3 grid(MIS ? (w + 247) / 248 : (w / W + 255) / 256, (h + EH - 1) / EH, frames);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<W, AHEAD, MIS, LDSB>), grid, dim3(256), 0, 0, g, w, h, (size_t)w * h, (size_t)w * h * 6, EH, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const int w = 1920, h = 1080, frames = 64;
    const size_t bytes = (size_t)w * h * 6 * frames * 4;
    float *g, *out; CHECK(hipMalloc(&g, bytes)); CHECK(hipMemset(g, 0, bytes)); CHECK(hipMalloc(&out, 64));
    const double gb = bytes / 1e9;
    float t;
    t = run<1, 3>(g, w, h, frames, out); printf("dword   loads, 3 rows (18 loads) ahead: %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
    t = run<1, 6>(g, w, h, frames, out); printf("dword   loads, 6 rows (36 loads) ahead: %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
    t = run<2, 3>(g, w, h, frames, out); printf("dwordx2 loads, 3 rows ahead:            %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
    t = run<4, 1>(g, w, h, frames, out); printf("dwordx4 loads, 1 row  (6 loads) ahead:  %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
    t = run<4, 2>(g, w, h, frames, out); printf("dwordx4 loads, 2 rows ahead:            %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
    t = run<4, 3>(g, w, h, frames, out); printf("dwordx4 loads, 3 rows ahead:            %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
    t = run<1, 3, true>(g, w, h, frames, out); printf("dword loads, 3 rows ahead, 62-column unaligned wavefronts:           %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
    t = run<1, 3, false, 32768>(g, w, h, frames, out); printf("dword loads, 3 rows ahead, 32 KB LDS per block (5 blocks per CU):      %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
    t = run<1, 3, true, 32768>(g, w, h, frames, out); printf("dword loads, 3 rows ahead, unaligned + 5 blocks per CU:                %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
    return 0;
}
