// Microbenchmark: shader cycles (s_memtime) and wall clock (s_memrealtime, 100 MHz) per f32 vector instruction on gfx950, by operand
// form and waves per SIMD -- is a v_fma_f32 with three VGPR sources a 2-cycle instruction under load, and what clock does the chip
// hold in a dense FMA loop?  (ubench_valu.hip converts wall time at an assumed 2.4 GHz.)
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_fma_cycles ubench_fma_cycles.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *stamps, int iters) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 0.001f + i;
    float x = out[threadIdx.x & 7], y = x + 1.0f, z = x + 2.0f;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));          // 3 VGPR reads, dst = src2
            if (MODE == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));             // VOP2 form of the same
            if (MODE == 2) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(y));              // 2 VGPR reads
            if (MODE == 3) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(x));                  // 2 distinct VGPRs
            if (MODE == 4) asm volatile("v_fma_f32 %0, %1, 2.0, %0" : "+v"(a[i]) : "v"(x));                 // inline constant
            if (MODE == 5) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(x), "v"(y), "v"(z));  // dst not a source
            if (MODE == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double *)&a[i & ~1]) : "v"(*(double *)&x), "v"(*(double *)&y));
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    float r = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) r += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) {
        stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0;
        stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
    }
}

int main() {
    float *d; CHECK(hipMalloc(&d, 8192 * 256 * 4)); CHECK(hipMemset(d, 0, 8192 * 256 * 4));
    unsigned long long *st; CHECK(hipMalloc(&st, 8192 * 4 * 2 * 8));
    const int iters = 8192;
    const char *names[7] = {"v_fma_f32 d,x,y,d", "v_fmac_f32 d,x,y", "v_mul_f32 d,x,y", "v_fma_f32 d,x,x,d", "v_fma_f32 d,x,2.0,d", "v_fma_f32 d,x,y,z", "v_pk_fma_f32 (2 fma)"};
    for (int wps : {1, 2, 3, 4, 8}) {
        const int grid = 256 * wps;
        for (int mode = 0; mode < 7; mode++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d, st, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, st, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, d, st, iters);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, d, st, iters);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, d, st, iters);
                if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), 0, 0, d, st, iters);
                if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(grid), dim3(256), 0, 0, d, st, iters);
                hipEventRecord(e1);
                CHECK(hipDeviceSynchronize());
            }
            float wall_ms = 0; hipEventElapsedTime(&wall_ms, e0, e1);
            std::vector<unsigned long long> h((size_t)grid * 8);
            CHECK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> cyc, mhz;
            for (int i = 0; i < grid * 4; i++) { cyc.push_back((double)h[2 * i]); mhz.push_back((double)h[2 * i] / ((double)h[2 * i + 1] / 100.0)); }
            std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
            const double c = cyc[cyc.size() / 2], n = (double)iters * 16;
            printf("waves/SIMD %d  %-22s %6.2f cycles per instruction per wave = %5.2f per SIMD   clock %4.0f MHz (median over waves)   wall %.3f ms = %.2f ns per instruction per SIMD\n", wps, names[mode], c / n, c / n / wps,
                   mhz[mhz.size() / 2], wall_ms, wall_ms * 1e6 / (n * wps));
        }
    }
    return 0;
}
