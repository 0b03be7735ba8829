// Microbenchmark: write-only and read-only HBM bandwidth beside the 50/50 copy (tools/ubench/ubench_copy.hip): the seed launch writes four
// bytes for every byte it reads, so the copy figure is not its ceiling.
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_write ubench_write.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <bool NT> __global__ __launch_bounds__(256) void k_write(float4 *dst, size_t n, float v) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const f4 q = {v, v, v, v};
    if (NT) __builtin_nontemporal_store(q, (f4 *)dst + i); else ((f4 *)dst)[i] = q;
}
__global__ __launch_bounds__(256) void k_read(const float4 *src, size_t n, float *out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 q = src[i];
    if (q.x + q.y + q.z + q.w == 12345.678f) out[0] = q.x;
}
// the seed's mix: read n/4 float4 (as bytes: one BGRA8 frame per four f32 output frames), write n float4
template <bool NT> __global__ __launch_bounds__(256) void k_mix(const float4 *src, float4 *dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 q = src[i >> 2];
    const f4 o = {q.x, q.y, q.z, q.w + (float)(i & 3)};
    if (NT) __builtin_nontemporal_store(o, (f4 *)dst + i); else ((f4 *)dst)[i] = o;
}
int main() {
    const size_t bytes = (size_t)3840 * 2160 * 64 * 4, n = bytes / 16;      // one octave-0 layer of 64 frames: 2.12 GB
    float4 *a, *b; float *out;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned grid = (unsigned)((n + 255) / 256);
    for (int mode = 0; mode < 6; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_write<false>, dim3(grid), dim3(256), 0, 0, b, n, 1.0f);
            if (mode == 1) hipLaunchKernelGGL(k_write<true>, dim3(grid), dim3(256), 0, 0, b, n, 1.0f);
            if (mode == 2) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, out);
            if (mode == 3) hipLaunchKernelGGL(k_mix<false>, dim3(grid), dim3(256), 0, 0, a, b, n);
            if (mode == 4) hipLaunchKernelGGL(k_mix<true>, dim3(grid), dim3(256), 0, 0, a, b, n);
            if (mode == 5) hipMemsetAsync(b, 0, bytes, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
        }
        const char *names[6] = {"write only, plain stores", "write only, non-temporal stores", "read only", "seed mix (read 1/4, write 1), plain", "seed mix, non-temporal stores", "hipMemsetAsync"};
        const double moved = (mode == 3 || mode == 4) ? bytes * 1.25 : (double)bytes;
        printf("%-40s %.3f ms  %.0f GB/s\n", names[mode], best, moved / best / 1e6);
    }
    return 0;
}
