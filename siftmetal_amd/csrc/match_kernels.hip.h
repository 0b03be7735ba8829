// match_kernels.hip.h -- brute-force descriptor matching with ratio test for gfx950 (MI355X).
//
// Replaces SIFTDescriptor.match(source:target:absoluteThreshold:relativeThreshold:)
// (Sources/SIFTMetal/SIFT/SIFTDescriptor.swift:298-361), the first "next" row of SURVEY.md 8f.
//
// The reference scans the targets sequentially per source descriptor with
//     if distance < best { second = best; best = distance; bestIndex = t }
// so `second` is the running minimum at the moment the final best was found, i.e. the minimum over the
// targets BEFORE the best one (FLT_MAX if the best is the first target) -- not the true second nearest.
// That order dependence decomposes into two order-independent reductions:
//     pass 1  (best, bestIndex) = argmin over all targets, smallest index on ties (strict '<' in the scan)
//     pass 2  second = min over targets with index < bestIndex
// Distances: the reference takes the f32 Euclidean distance of features/255; here the squared distance of
// the 0..255 integers is formed exactly in int32 (bytes re-biased by 128 to signed i8, v_dot4_i32_i8:
// |a-b|^2 = |a'|^2 + |b'|^2 - 2 a'.b', shift-invariant) and distance = sqrt(D) / 255 in f32; the two agree
// to f32 rounding (~1e-6 relative), which only matters on threshold knife edges.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "keypoint_kernels.hip.h"

namespace siftmi {

struct MatchRec { int32_t source, target; float distance; };          // == siftmi_match

constexpr int MATCH_SRC_PER_BLOCK = 64;     // one source descriptor per lane, the 4 waves split the targets
constexpr int MATCH_TILE = 64;              // targets staged in LDS per iteration

__device__ __forceinline__ int dot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }

// one pass over all targets for the 64 sources of this block; PASS 1: argmin, PASS 2: min over index < limit
template <int PASS>
__device__ __forceinline__ void match_pass(const DescriptorRec *__restrict__ tgt, int n_tgt, const int (&a)[32], int na, int limit,
                                           int *lds_t /* [MATCH_TILE][33] */, int &bestD, int &bestI) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    bestD = 0x7fffffff; bestI = -1;
    for (int t0 = 0; t0 < n_tgt; t0 += MATCH_TILE) {
        __syncthreads();
        // stage MATCH_TILE targets: 32 re-biased dwords + |b'|^2 each
        for (int i = threadIdx.x; i < MATCH_TILE * 32; i += 256) {
            const int tt = i >> 5, k = i & 31;
            int v = 0;
            if (t0 + tt < n_tgt) v = reinterpret_cast<const int *>(tgt[t0 + tt].features)[k] ^ (int)0x80808080;
            lds_t[tt * 33 + k] = v;
        }
        __syncthreads();
        if (threadIdx.x < MATCH_TILE) {
            int nb = 0;
#pragma unroll
            for (int k = 0; k < 32; k++) { const int v = lds_t[threadIdx.x * 33 + k]; nb = dot4(v, v, nb); }
            lds_t[MATCH_TILE * 33 + threadIdx.x] = nb;
        }
        __syncthreads();
        const int tend = min(MATCH_TILE, n_tgt - t0);
        for (int tt = wv; tt < tend; tt += 4) {                 // this wave's targets of the tile, increasing index
            const int gi = t0 + tt;
            if (PASS == 2 && gi >= limit) break;                // only targets before this source's best (per lane)
            int dot = 0;
#pragma unroll
            for (int k = 0; k < 32; k++) dot = dot4(a[k], lds_t[tt * 33 + k], dot);     // broadcast LDS reads
            const int D = na + lds_t[MATCH_TILE * 33 + tt] - 2 * dot;
            if (D < bestD) { bestD = D; bestI = gi; }           // strict: first index wins inside a wave's ordered subset
        }
        (void)lane;
    }
}

__global__ __launch_bounds__(256) void match_kernel(const DescriptorRec *__restrict__ src, int n_src, const DescriptorRec *__restrict__ tgt,
                                                   int n_tgt, float abs_thr, float rel_thr, MatchRec *__restrict__ out /* [n_src] */) {
    __shared__ int lds_t[MATCH_TILE * 33 + MATCH_TILE];
    __shared__ int red_d[4][64], red_i[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int s = blockIdx.x * MATCH_SRC_PER_BLOCK + lane;
    int a[32], na = 0;
#pragma unroll
    for (int k = 0; k < 32; k++) {
        a[k] = (s < n_src) ? (reinterpret_cast<const int *>(src[s].features)[k] ^ (int)0x80808080) : 0;
        na = dot4(a[k], a[k], na);
    }
    // ---- pass 1: global argmin (smallest index on ties)
    int bd, bi;
    match_pass<1>(tgt, n_tgt, a, na, 0, lds_t, bd, bi);
    red_d[wv][lane] = bd; red_i[wv][lane] = bi;
    __syncthreads();
    int bestD = 0x7fffffff, bestI = -1;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const int d = red_d[w][lane], i = red_i[w][lane];
        if (i >= 0 && (d < bestD || (d == bestD && i < bestI))) { bestD = d; bestI = i; }
    }
    __syncthreads();
    // ---- pass 2: minimum over the targets before the best one
    int sd, si;
    match_pass<2>(tgt, n_tgt, a, na, bestI, lds_t, sd, si);
    red_d[wv][lane] = sd;
    __syncthreads();
    int secondD = 0x7fffffff;
#pragma unroll
    for (int w = 0; w < 4; w++) secondD = min(secondD, red_d[w][lane]);
    if (wv != 0 || s >= n_src) return;
    MatchRec r; r.source = s; r.target = -1; r.distance = 0.0f;
    if (bestI >= 0) {
        const float best = sqrtf((float)bestD) / 255.0f;
        const float second = (secondD == 0x7fffffff) ? 3.402823466e+38f : sqrtf((float)secondD) / 255.0f;
        r.distance = best;
        if (best < abs_thr && best < second * rel_thr) r.target = bestI;      // SIFTDescriptor.swift:349-355
    }
    out[s] = r;
}

}  // namespace siftmi
