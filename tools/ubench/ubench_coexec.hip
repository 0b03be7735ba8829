// Microbenchmark: do f32 vector FMAs and f32 MFMAs (v_mfma_f32_16x16x4_f32) of DIFFERENT wavefronts on one SIMD run side by side?
// (MI355X_MICROARCH.md: "MFMA and VALU pipes are separate"; an MFMA holds the SIMD's vector issue for 8 of its 32 cycles.)
// Round 3 tried the ring blur's vertical pass on the K = 1 multi-block MFMA forms and found no second pipe; those forms hold the
// SIMD for their whole duration (4x4x1) or were one dependent chain (16x16x1).  This measures the K = 4 form with four
// independent accumulator chains, the shape a 16-row x 64-column vertical-pass tile would use.
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_coexec ubench_coexec.hip ; run: ./ubench_coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// role: 0 = 16 independent v_fma chains, 1 = 4 independent MFMA chains, 2 = both in one instruction stream (16 fma + 1 mfma per trip)
template <int MODE>
__global__ __launch_bounds__(512) void k_mix(float *out, int iters_valu, int iters_mfma) {
    const int wv = threadIdx.x >> 6;
    int role;
    if (MODE == 0) role = 0;
    else if (MODE == 1) role = 1;
    else if (MODE == 2) role = wv >= 4 ? 1 : 0;          // SIMD partners: wave w and w + 4
    else if (MODE == 5) role = wv >= 4 ? 3 : 0;          // the fma half alone (partners exit)
    else if (MODE == 6) role = wv >= 4 ? 1 : 3;          // the MFMA half alone
    else role = 2;
    if (role == 3) return;
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 0.001f + i;
    float x = out[threadIdx.x & 7], y = x + 1.0f;
    f4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = f4{0.f, 0.f, 0.f, 0.f};
    if (role == 0) {
        for (int it = 0; it < iters_valu; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
        }
    } else if (role == 1) {
        for (int it = 0; it < iters_mfma; it++) {
#pragma unroll
            for (int q = 0; q < 4; q++) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(x), "v"(y));
        }
    } else {
        for (int it = 0; it < iters_mfma * 4; it++) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[it & 3]) : "v"(x), "v"(y));
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) r += a[i];
#pragma unroll
    for (int q = 0; q < 4; q++) r += acc[q].x + acc[q].y + acc[q].z + acc[q].w;
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

// MODE 3's loop indexes acc[] dynamically; give it a static form instead
__global__ __launch_bounds__(512) void k_interleaved(float *out, int iters) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 0.001f + i;
    float x = out[threadIdx.x & 7], y = x + 1.0f;
    f4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = f4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(x), "v"(y));
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) r += a[i];
#pragma unroll
    for (int q = 0; q < 4; q++) r += acc[q].x + acc[q].y + acc[q].z + acc[q].w;
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

int main() {
    float *d; CHECK(hipMalloc(&d, 4096 * 512 * 4)); CHECK(hipMemset(d, 0, 4096 * 512 * 4));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    // per wave: iv trips x 16 v_fma (2 cycles each at SIMD-32) and im trips x 4 MFMA (32 cycles each): equal pipe time when 32 iv = 128 im
    const int iv = 16384, im = 4096;
    for (int wg_per_cu : {1, 2}) {
        const int grid = 256 * wg_per_cu;
        float ms[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int mode = 0; mode < 7; mode++) {
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k_mix<0>, dim3(grid), dim3(512), 0, 0, d, iv, im);
                if (mode == 1) hipLaunchKernelGGL(k_mix<1>, dim3(grid), dim3(512), 0, 0, d, iv, im);
                if (mode == 2) hipLaunchKernelGGL(k_mix<2>, dim3(grid), dim3(512), 0, 0, d, iv, im);
                if (mode == 3) hipLaunchKernelGGL(k_interleaved, dim3(grid), dim3(512), 0, 0, d, im);       // per wave: im x 4 MFMA + im x 64 fma
                if (mode == 4) hipLaunchKernelGGL(k_mix<0>, dim3(grid), dim3(512), 0, 0, d, im * 4, im);   // the fma share of mode 3 alone
                if (mode == 5) hipLaunchKernelGGL(k_mix<5>, dim3(grid), dim3(512), 0, 0, d, iv, im);
                if (mode == 6) hipLaunchKernelGGL(k_mix<6>, dim3(grid), dim3(512), 0, 0, d, iv, im);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            hipEventElapsedTime(&ms[mode], a, b);
        }
        printf("%d x 512-thread workgroups per CU (2 waves per SIMD each)\n", wg_per_cu);
        printf("  all 8 waves: %d x 16 v_fma_f32                       %8.3f ms\n", iv, ms[0]);
        printf("  all 8 waves: %d x 4 v_mfma_f32_16x16x4_f32            %8.3f ms\n", im, ms[1]);
        printf("  waves 0-3 the fma loop, waves 4-7 the MFMA loop         %8.3f ms   (waves 0-3 alone %.3f, waves 4-7 alone %.3f: separate pipes ~ the larger, one pipe ~ the sum)\n",
               ms[2], ms[5], ms[6]);
        printf("  every wave: %d x (4 MFMA + 64 fma) in one stream       %8.3f ms   (the 64-fma share alone: %.3f, the MFMA share alone: %.3f)\n", im, ms[3], ms[4], ms[1]);
    }
    return 0;
}
