// Microbenchmarks: f32 FMA issue rate (scalar / packed / SGPR operand) and ds_read_b128 rate on gfx950.
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu ubench_valu.hip ; run: ./ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_fma(float *out, int iters, float s0, float s1, float s2, float s3) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 0.001f + i;
    float x = out[threadIdx.x & 7], y = x + 1.0f;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {          // v_fma_f32 vgpr operands, 16 independent chains
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
        } else if (MODE == 1) {   // v_fmac_f32 with SGPR multiplicand
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(s0), "v"(y));
        } else if (MODE == 2) {   // v_pk_fma_f32, 8 independent pairs
#pragma unroll
            for (int i = 0; i < 8; i++) {
                f2 acc = {a[2 * i], a[2 * i + 1]}; f2 xx = {x, y}, yy = {y, x};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(xx), "v"(yy));
                a[2 * i] = acc.x; a[2 * i + 1] = acc.y;
            }
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) r += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

__global__ __launch_bounds__(256) void k_lds(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i;
    __syncthreads();
    f4 acc = {0, 0, 0, 0};
    const __attribute__((address_space(3))) volatile f4 *p = (const __attribute__((address_space(3))) volatile f4 *)(lds + (threadIdx.x & 63) * 4);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) { f4 v = p[i * 64]; acc += v; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
    float *d; CHECK(hipMalloc(&d, 4096 * 256 * 4)); CHECK(hipMemset(d, 0, 4096 * 256 * 4));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4096;
    for (int wg_per_cu : {1, 2, 4, 8}) {
        const int grid = 256 * wg_per_cu;
        for (int mode = 0; mode < 4; mode++) {
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k_fma<0>, dim3(grid), dim3(256), 0, 0, d, iters, 1.f, 2.f, 3.f, 4.f);
                if (mode == 1) hipLaunchKernelGGL(k_fma<1>, dim3(grid), dim3(256), 0, 0, d, iters, 1.f, 2.f, 3.f, 4.f);
                if (mode == 2) hipLaunchKernelGGL(k_fma<2>, dim3(grid), dim3(256), 0, 0, d, iters, 1.f, 2.f, 3.f, 4.f);
                if (mode == 3) hipLaunchKernelGGL(k_lds, dim3(grid), dim3(256), 0, 0, d, iters);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            const double waves = (double)grid * 4;
            if (mode < 3) {
                const double fmas = waves * 64.0 * iters * 16;
                printf("waves/SIMD %d  %-22s %8.3f ms  %7.1f TFLOP/s  (%.2f cycles per wave-instr per SIMD @2.4GHz)\n", wg_per_cu,
                       mode == 0 ? "v_fma_f32 vvv" : mode == 1 ? "v_fmac_f32 s,v" : "v_pk_fma_f32", ms, 2 * fmas / ms / 1e9,
                       ms * 1e-3 * 2.4e9 / (iters * (mode == 2 ? 8.0 : 16.0) * wg_per_cu));
            } else {
                const double bytes = waves * 64.0 * 16 * iters * 16;
                printf("waves/SIMD %d  %-22s %8.3f ms  %7.1f TB/s  (%.1f B/clk/CU @2.4GHz)\n", wg_per_cu, "ds_read_b128", ms, bytes / ms / 1e9,
                       bytes / 256 / (ms * 1e-3 * 2.4e9));
            }
        }
    }
    return 0;
}
