// Microbenchmark: which plain copy reaches the highest HBM rate on this box?  (bench.py quotes the pyramid kernel against a copy
// measured in the same run; MI355X_MICROARCH.md quotes 6.29 TB/s for "float4 copy".)  2 x 2.1 GB buffers, 10 launches each.
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_copy ubench_copy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_u(const f4 *__restrict__ s, f4 *__restrict__ d, size_t n4) {
    const size_t i0 = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int k = 0; k < U; k++) { const size_t i = i0 + (size_t)k * 256; if (i < n4) v[k] = NT ? __builtin_nontemporal_load(s + i) : s[i]; }
#pragma unroll
    for (int k = 0; k < U; k++) { const size_t i = i0 + (size_t)k * 256; if (i < n4) { if (NT) __builtin_nontemporal_store(v[k], d + i); else d[i] = v[k]; } }
}
template <int U>
__global__ __launch_bounds__(256) void copy_loop(const f4 *__restrict__ s, f4 *__restrict__ d, size_t n4) {
    for (size_t b = blockIdx.x; b * 256 * U < n4; b += gridDim.x) {
        const size_t i0 = b * 256 * U + threadIdx.x;
        f4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) { const size_t i = i0 + (size_t)k * 256; if (i < n4) v[k] = s[i]; }
#pragma unroll
        for (int k = 0; k < U; k++) { const size_t i = i0 + (size_t)k * 256; if (i < n4) d[i] = v[k]; }
    }
}

// the ring blur's access shapes without the blur (tools/ubench/pmc_calib.hip): 128-column x 32-row tiles, loads 8 lanes per row x float4
// (a wave instruction = 8 rows x 128 B), an LDS round trip with one barrier, stores one row per wave instruction as float2 per lane
typedef float f2 __attribute__((ext_vector_type(2)));
template <int STORE16, bool XCD>
__global__ __launch_bounds__(256) void tile_copy(const float *__restrict__ src, float *__restrict__ dst, int w, int h, int nf) {
    __shared__ __attribute__((aligned(16))) float tile[32 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tx = w / 128, ty = h / 32, total = tx * ty * nf;
    int t = blockIdx.x;
    if (XCD) { const int per = (total + 7) >> 3; t = (blockIdx.x & 7) * per + (blockIdx.x >> 3); }
    if (t >= total) return;
    const int frame = t / (tx * ty), rem = t - frame * (tx * ty), by = rem / tx, bx = rem - by * tx;
    const float *in = src + (size_t)frame * w * h + (size_t)(by * 32) * w + bx * 128;
    float *out = dst + (size_t)frame * w * h + (size_t)(by * 32) * w + bx * 128;
    const int prow = tid >> 3, pq = tid & 7;
    f4 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = *reinterpret_cast<const f4 *>(in + (size_t)prow * w + 4 * pq + 32 * j);
#pragma unroll
    for (int j = 0; j < 4; j++) *reinterpret_cast<f4 *>(tile + prow * 128 + 4 * pq + 32 * j) = v[j];
    __syncthreads();
    if (STORE16) {
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {      // 16-B stores: two rows per wave instruction
            const int row = wv * 8 + rr * 2 + (lane >> 5);
            *reinterpret_cast<f4 *>(out + (size_t)row * w + 4 * (lane & 31)) = *reinterpret_cast<const f4 *>(tile + row * 128 + 4 * (lane & 31));
        }
    } else {
#pragma unroll
        for (int rr = 0; rr < 8; rr++) {
            const int row = wv * 8 + rr;
            *reinterpret_cast<f2 *>(out + (size_t)row * w + 2 * lane) = *reinterpret_cast<const f2 *>(tile + row * 128 + 2 * lane);
        }
    }
}

int main() {
    const size_t bytes = (size_t)2123366400;           // one octave-0 layer of 64 x 3840x2160 floats
    const size_t n4 = bytes / 16;
    f4 *s, *d; CHECK(hipMalloc(&s, bytes)); CHECK(hipMalloc(&d, bytes)); CHECK(hipMemset(s, 0x3c, bytes));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char *name, auto launch) {
        for (int rep = 0; rep < 2; rep++) {
            launch();
            hipEventRecord(a);
            for (int i = 0; i < 10; i++) launch();
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
            if (rep) printf("%-48s %.4f ms  %7.1f GB/s (read + written)\n", name, ms, 2.0 * bytes / (ms * 1e-3) / 1e9);
        }
    };
#define ONE(U, NT) run("one pass, " #U " float4 per thread, nontemporal=" #NT, [&] { hipLaunchKernelGGL((copy_u<U, NT>), dim3((unsigned)((n4 + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, s, d, n4); });
    ONE(1, false) ONE(2, false) ONE(4, false) ONE(8, false) ONE(1, true) ONE(4, true)
#define LOOP(U, G) run("grid-stride loop, " #U " float4 per trip, grid " #G, [&] { hipLaunchKernelGGL((copy_loop<U>), dim3(G), dim3(256), 0, 0, s, d, n4); });
    LOOP(4, 2048) LOOP(4, 8192) LOOP(2, 4096) LOOP(1, 8192)
    { const int w = 3840, h = 2160, nf = 64; const int total = (w / 128) * (h / 32) * nf;   // (h = 2160 -> 67 whole 32-row tiles: 2144 rows)
      const double tb = 2.0 * (double)w * (h / 32 * 32) * nf * 4;
      auto runt = [&](const char *name, auto launch) {
          for (int rep = 0; rep < 2; rep++) { launch(); hipEventRecord(a); for (int i = 0; i < 10; i++) launch(); hipEventRecord(b); hipEventSynchronize(b);
              float ms; hipEventElapsedTime(&ms, a, b); ms /= 10; if (rep) printf("%-48s %.4f ms  %7.1f GB/s (read + written)\n", name, ms, tb / (ms * 1e-3) / 1e9); } };
      runt("ring access shapes, 8-B row stores", [&] { hipLaunchKernelGGL((tile_copy<0, false>), dim3(total), dim3(256), 0, 0, (const float *)s, (float *)d, w, h, nf); });
      runt("ring access shapes, 16-B row stores", [&] { hipLaunchKernelGGL((tile_copy<1, false>), dim3(total), dim3(256), 0, 0, (const float *)s, (float *)d, w, h, nf); });
      runt("ring access shapes, 8-B stores, XCD-contiguous order", [&] { hipLaunchKernelGGL((tile_copy<0, true>), dim3((total + 7) / 8 * 8), dim3(256), 0, 0, (const float *)s, (float *)d, w, h, nf); });
      runt("ring access shapes, 16-B stores, XCD-contiguous order", [&] { hipLaunchKernelGGL((tile_copy<1, true>), dim3((total + 7) / 8 * 8), dim3(256), 0, 0, (const float *)s, (float *)d, w, h, nf); }); }
    run("hipMemcpyAsync device to device", [&] { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
