// LDS atomic throughput on gfx950: ds_add_f32 / ds_add_u32 / ds_add_u64 / plain ds_write for several address patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, int pattern) {
    __shared__ unsigned long long lds64[4096];
    float *ldsf = reinterpret_cast<float *>(lds64);
    unsigned *ldsu = reinterpret_cast<unsigned *>(lds64);
    for (int i = threadIdx.x; i < 4096; i += 256) lds64[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int idx;
    if (pattern == 0) idx = lane;                 // all distinct, consecutive
    else if (pattern == 1) idx = lane >> 2;       // 4 lanes per address
    else if (pattern == 2) idx = lane >> 4;       // 16 lanes per address
    else if (pattern == 3) idx = 0;               // all same
    else idx = (lane * 17) & 127;                 // scattered
    idx += wv * 512;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int a = idx + u * 64;
            if (MODE == 0) atomicAdd(&ldsf[a], 1.0f);
            else if (MODE == 1) atomicAdd(&ldsu[a], 1u);
            else if (MODE == 2) atomicAdd(&lds64[a], 1ull);
            else ldsf[a] = (float)it;
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = ldsf[threadIdx.x];
}

int main() {
    float *d; CHECK(hipMalloc(&d, 2048 * 256 * 4));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int iters = 1024, grid = 1024;   // 4 WGs per CU
    const char *mn[] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "ds_write_b32"};
    const char *pn[] = {"distinct", "4 lanes/addr", "16 lanes/addr", "all same", "scattered"};
    for (int mode = 0; mode < 4; mode++)
        for (int p = 0; p < 5; p++) {
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(a));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d, iters, p);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, iters, p);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, d, iters, p);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, d, iters, p);
                CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            }
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            const double winstr = (double)grid * 4 * iters * 8;            // wave-instructions
            printf("%-13s %-14s %8.3f ms  %6.1f cycles per wave-instr per CU @2.2GHz\n", mn[mode], pn[p], ms, ms * 1e-3 * 2.2e9 / (winstr / 256));
        }
    return 0;
}
