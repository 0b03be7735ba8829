// Times match_mfma_kernel (and an MFMA-issue-rate loop) on random packed rows.  Variants through -D:
//   MM_VARIANT_NO_UPDATE  -- screening only, never takes the update path (upper bound of the MFMA + scan loop)
// usage: match_variants n_src n_tgt n_split [pre-pass length]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "match_kernels.hip.h"
using namespace siftmi;

__global__ __launch_bounds__(256) void mfma_rate_kernel(int *out, int iters) {
    i32x4 a = {1, 2, 3, (int)threadIdx.x}, b = {4, 5, 6, 7};
    i32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main(int argc, char **argv) {
    const int ns = argc > 1 ? atoi(argv[1]) : 100000, nt = argc > 2 ? atoi(argv[2]) : 100000;
    int n_split = argc > 3 ? atoi(argv[3]) : 6;
    const int prepass = argc > 4 ? atoi(argv[4]) : 0;          // targets [0, prepass) matched by a launch of their own; its records bound every later chunk
    int4 *bound = nullptr; hipMalloc(&bound, (size_t)2000000 * 16);
    long long split_len = (nt + n_split - 1) / n_split;
    split_len = (split_len + MM_SPLIT_QUANTUM - 1) / MM_SPLIT_QUANTUM * MM_SPLIT_QUANTUM;
    n_split = (nt + split_len - 1) / split_len;
    std::vector<int> hs((size_t)ns * 32), ht((size_t)nt * 32), hn(nt);
    srand(1);
    for (auto &v : hs) v = rand() * 7919 + rand();
    for (auto &v : ht) v = rand() * 7919 + rand();
    for (int t = 0; t < nt; t++) {
        long long n = 0;
        for (int k = 0; k < 32; k++) for (int e = 0; e < 4; e++) { int b = (signed char)(ht[(size_t)t * 32 + k] >> (8 * e)); n += b * b; }
        hn[t] = (int)n;
    }
    int *ds, *dt, *dn; int4 *part; int *out;
    hipMalloc(&ds, hs.size() * 4); hipMalloc(&dt, ht.size() * 4); hipMalloc(&dn, nt * 4); hipMalloc(&part, (size_t)n_split * ns * 16);
    hipMalloc(&out, 1024 * 256 * 4 * 8);
    hipMemcpy(ds, hs.data(), hs.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dt, ht.data(), ht.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dn, hn.data(), nt * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int groups = (ns + MM_SRC_PER_BLOCK - 1) / MM_SRC_PER_BLOCK;
    float ms;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        if (prepass) hipMemsetAsync(part, 0x7f, (size_t)n_split * ns * 16, 0);
        if (prepass) hipLaunchKernelGGL(match_mfma_kernel<false>, dim3(groups, 1), dim3(256), 0, 0, ds, ns, dt, dn, std::min(nt, prepass), prepass, bound, (const int4 *)nullptr, MatchTail{});
        hipLaunchKernelGGL(match_mfma_kernel<false>, dim3(groups, n_split), dim3(256), 0, 0, ds, ns, dt, dn, nt, (int)split_len, part, prepass ? bound : (const int4 *)nullptr, MatchTail{});
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
#ifdef MM_COUNT_SLOW
    { unsigned long long z = 0, v = 0; hipMemcpyToSymbol(HIP_SYMBOL(mm_slow_count), &z, 8);
      if (prepass) hipMemset(part, 0x7f, (size_t)n_split * ns * 16);
      if (prepass) { hipLaunchKernelGGL(match_mfma_kernel<false>, dim3(groups, 1), dim3(256), 0, 0, ds, ns, dt, dn, std::min(nt, prepass), prepass, bound, (const int4 *)nullptr, MatchTail{});
                     hipDeviceSynchronize(); hipMemcpyToSymbol(HIP_SYMBOL(mm_slow_count), &z, 8); }
      hipLaunchKernelGGL(match_mfma_kernel<false>, dim3(groups, n_split), dim3(256), 0, 0, ds, ns, dt, dn, nt, (int)split_len, part, prepass ? bound : (const int4 *)nullptr, MatchTail{});
      hipDeviceSynchronize(); hipMemcpyFromSymbol(&v, HIP_SYMBOL(mm_slow_count), 8);
      const double tiles = (double)groups * n_split * 4 /*waves*/ * MM_NB * (split_len / 2 / 16);
      printf("ordered updates taken: %llu of %.0f (wavefront, 16-target tile, 32-source tile) triples = %.1f %%\n", v, tiles, 100.0 * v / tiles); }
#endif
#ifdef MM_STAMPS
    { unsigned long long z[4] = {0, 0, 0, 0}, v[4]; hipMemcpyToSymbol(HIP_SYMBOL(mm_stamps), z, 32);
      if (prepass) hipMemset(part, 0x7f, (size_t)n_split * ns * 16);
      hipLaunchKernelGGL(match_mfma_kernel<false>, dim3(groups, n_split), dim3(256), 0, 0, ds, ns, dt, dn, nt, (int)split_len, part, (const int4 *)nullptr, MatchTail{});
      hipDeviceSynchronize(); hipMemcpyFromSymbol(v, HIP_SYMBOL(mm_stamps), 32);
      const double iters = (double)groups * n_split * 4 * (split_len / 2 / (16 * MM_TT));
      printf("stamps, cycles per wavefront-iteration (%d MFMAs = %d matrix-pipe cycles): prefetch issue %.0f, MFMA groups + screens %.0f, staging %.0f, barrier %.0f\n",
             16 * MM_TT * MM_NB / 4 * 4, 32 * 4 * MM_TT * MM_NB, v[0] / iters, v[1] / iters, v[2] / iters, v[3] / iters); }
#endif
    const double pairs = (double)ns * nt;
    printf("match_mfma %d x %d, %d groups x %d splits (len %lld), pre-pass %d: %.3f ms, %.2f Tpairs/s, %.3f PFLOP/s(i8)\n", ns, nt, groups, n_split, split_len, prepass, ms,
           pairs / ms / 1e9, pairs * 256 / ms / 1e12);
    for (int wpb = 1; wpb <= 2; wpb++) {
        const int iters = 20000, blocks = 256 * wpb;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_rate_kernel, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        const double n_mfma = (double)blocks * 4 * iters * 4;
        printf("mfma_i32_32x32x32_i8 issue loop, %d blocks: %.3f ms, %.2f PFLOP/s, %.1f ns per MFMA per SIMD\n", blocks, ms, n_mfma * 65536 / ms / 1e12,
               ms * 1e6 / (n_mfma / 1024));
    }
    return 0;
}
