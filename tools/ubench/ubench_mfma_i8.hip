// Microbenchmark: what int8 matrix rate does an MI355X sustain in a pure v_mfma_i32_32x32x32_i8 loop (no loads, no LDS), at what
// shader clock, by wavefronts per SIMD and independent accumulator chains?  The matcher's roofline quotes the data-sheet dense
// figure (MI355X_MICROARCH.md); this gives the measured ceiling beside it, the way siftmi_time_copy does for the blur.
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_mfma_i8 ubench_mfma_i8.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int CHAINS>
__global__ __launch_bounds__(256) void k(int *out, unsigned long long *stamps, int iters) {
    i32x4 a, b;
    for (int i = 0; i < 4; i++) { a[i] = threadIdx.x * 0x01010101 + i; b[i] = out[threadIdx.x & 7] + i * 0x01020304; }
    i32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; c++) for (int i = 0; i < 16; i++) acc[c][i] = 0;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[c], 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    int r = 0;
    for (int c = 0; c < CHAINS; c++) for (int i = 0; i < 16; i++) r += acc[c][i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) {
        stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0;
        stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
    }
}

int main() {
    int *d; CHECK(hipMalloc(&d, 4096 * 256 * 4)); CHECK(hipMemset(d, 0, 4096 * 256 * 4));
    unsigned long long *st; CHECK(hipMalloc(&st, 4096 * 4 * 2 * 8));
    const int iters = 20000;
    for (int wps : {1, 2, 4}) {
        const int grid = 256 * wps;
        for (int chains : {2, 4}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (chains == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, d, st, iters * 2);
                else hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, d, st, iters);
                hipEventRecord(e1);
                CHECK(hipDeviceSynchronize());
            }
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h((size_t)grid * 8);
            CHECK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> cyc, mhz;
            for (int i = 0; i < grid * 4; i++) { cyc.push_back((double)h[2 * i]); mhz.push_back((double)h[2 * i] / ((double)h[2 * i + 1] / 100.0)); }
            std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
            const double n_mfma = (double)iters * 4, ops = n_mfma * 65536.0 * grid * 4;
            printf("waves/SIMD %d, %d chains: %.3f ms, %.2f POP/s int8; %.1f cycles per MFMA per wave = %.1f per SIMD; clock %.0f MHz (s_memtime / s_memrealtime, median)\n",
                   wps, chains, ms, ops / ms / 1e12, cyc[cyc.size() / 2] / n_mfma, cyc[cyc.size() / 2] / n_mfma / wps, mhz[mhz.size() / 2]);
        }
    }
    return 0;
}
