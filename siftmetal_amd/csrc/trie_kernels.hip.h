// trie_kernels.hip.h -- SIFTDescriptor.approximateMatch (SIFT/SIFTDescriptor.swift:362-417) over the reference's ANN
// trie (Utilities/Trie.swift:76-416), without the pointer structure.
//
// The reference inserts every target under its 16-component indexKey, 8 bins per component
// (bin = Int((v * 7).rounded())), links the leaves in depth-first child order into a ring, and answers a query by
// walking down (nearest existing child where the exact bin is missing), then scanning the leaf it lands on, 10 leaves to
// the left and 10 to the right, keeping a 2-deep queue of successive improvements.
// Here a target's path is a 48-bit code (16 x 3 bits, first component most significant).  Depth-first child order =
// ascending code order, so a stable sort of (code, target index) IS the linked leaf ring: a leaf is a run of equal codes
// (insertion order inside, because the sort is stable), the children of a node are the distinct next digits inside the
// run of codes sharing its prefix (found by binary search), and ring neighbours are the adjacent runs, wrapping at the
// ends.  One thread per query descriptor; distances are exact integers (|a|^2 + |b|^2 - 2 a.b by v_dot4 on re-biased
// bytes), compared as sqrtf((float)D) like IntVector.distance (Utilities/Vector.swift:45-59).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "match_kernels.hip.h"

namespace siftmi {

constexpr int TRIE_BINS = 8, TRIE_DEPTH = 16, TRIE_RADIUS = 10;       // SIFTDescriptor.swift:370,395

// indexKey -> path code.  Same float operations, in the same order, as descriptor_index_vectors (host_post.h).
__device__ __forceinline__ unsigned long long trie_code(const unsigned char *features) {
    const int order[16] = {5, 6, 9, 10, 0, 3, 12, 15, 1, 2, 4, 7, 8, 11, 13, 14};    // SIFTDescriptor.swift:49-73
    unsigned long long code = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const unsigned char *cell = features + 8 * order[k];
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; i++) acc += (float)cell[i] / 255.0f;
        const float mean = acc / 8.0f;
        const int bin = (int)roundf(mean * (float)(TRIE_BINS - 1));                   // Trie.swift:381-388
        code = (code << 3) | (unsigned long long)bin;
    }
    return code;
}

__global__ __launch_bounds__(256) void trie_code_kernel(const DescriptorRec *__restrict__ d, int n, unsigned long long *__restrict__ codes,
                                                       int32_t *__restrict__ idx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    codes[i] = trie_code(d[i].features);
    idx[i] = i;
}

__device__ __forceinline__ int trie_lower_bound(const unsigned long long *__restrict__ codes, int lo, int hi, unsigned long long key) {
    while (lo < hi) {                                  // first position in [lo, hi) with codes[pos] >= key
        const int mid = (lo + hi) >> 1;
        if (codes[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

struct TrieQueue {                                     // FiniteQueue<Match>(capacity: 2), Trie.swift:222-252
    int count = 0, best_idx = -1;
    float best = 3.402823466e+38f, second = 0.0f;
};

// nearestValue (Trie.swift:362-377) over the run [lo, hi) of the sorted arrays
__device__ __forceinline__ void trie_scan_leaf(const DescriptorRec *__restrict__ tgt, const int32_t *__restrict__ sorted_idx, int lo, int hi,
                                               const int (&a)[32], int na, TrieQueue &q) {
    for (int p = lo; p < hi; p++) {
        const int t = sorted_idx[p];
        const int *f = reinterpret_cast<const int *>(tgt[t].features);
        int nb = 0, dot = 0;
#pragma unroll
        for (int k = 0; k < 32; k++) { const int v = f[k] ^ (int)0x80808080; nb = dot4(v, v, nb); dot = dot4(a[k], v, dot); }
        const float distance = sqrtf((float)(na + nb - 2 * dot));
        if (distance < q.best) {                       // insert at the front, the previous front becomes the second entry
            q.second = q.best; q.best = distance; q.best_idx = t;
            q.count = min(q.count + 1, 2);
        }
    }
}

__global__ __launch_bounds__(64) void trie_query_kernel(const DescriptorRec *__restrict__ src, int n_src, const DescriptorRec *__restrict__ tgt,
                                                       const unsigned long long *__restrict__ codes, const int32_t *__restrict__ sorted_idx, int n_tgt,
                                                       float abs_thr, float rel_thr, MatchRec *__restrict__ out) {
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n_src) return;
    int a[32], na = 0;
#pragma unroll
    for (int k = 0; k < 32; k++) { a[k] = reinterpret_cast<const int *>(src[s].features)[k] ^ (int)0x80808080; na = dot4(a[k], a[k], na); }
    const unsigned long long qcode = trie_code(src[s].features);
    // nearestNode (Trie.swift:326-360): exact child if present, otherwise the present child with the smallest
    // binDifference = wrap(|j - bin|), where wrap maps 7 to 0 (wrapBinIndex with n = numberOfBins - 1), first j on ties
    int lo = 0, hi = n_tgt;
    for (int level = 0; level < TRIE_DEPTH; level++) {
        if (codes[lo] == codes[hi - 1]) break;          // one leaf left below this node: every remaining level has a single child
        const int shift = 3 * (TRIE_DEPTH - 1 - level);
        const int b = (int)((qcode >> shift) & 7);
        const unsigned long long prefix = codes[lo] >> (shift + 3);
        int l = trie_lower_bound(codes, lo, hi, ((prefix << 3) | (unsigned long long)b) << shift);
        int h = trie_lower_bound(codes, l, hi, ((prefix << 3) | (unsigned long long)(b + 1)) << shift);
        if (l == h) {
            int best_d = 0x7fffffff;
            int p = lo;
            while (p < hi) {                            // present children in ascending digit order
                const int j = (int)((codes[p] >> shift) & 7);
                const int e = trie_lower_bound(codes, p, hi, ((prefix << 3) | (unsigned long long)(j + 1)) << shift);
                int d = abs(j - b);
                if (d >= TRIE_BINS - 1) d -= TRIE_BINS - 1;
                if (d < best_d) { best_d = d; l = p; h = e; }
                p = e;
            }
        }
        lo = l; hi = h;
    }
    // nearest (Trie.swift:300-324): the leaf, then 10 leaves to the left, then 10 to the right of it on the ring
    TrieQueue q;
    trie_scan_leaf(tgt, sorted_idx, lo, hi, a, na, q);
    int l = lo, h = hi;
    for (int r = 0; r < TRIE_RADIUS; r++) {
        h = (l == 0) ? n_tgt : l;
        const unsigned long long c = codes[h - 1];
        l = h - 1;
        while (l > 0 && codes[l - 1] == c) l--;          // leaves hold a handful of values: a linear walk beats a search
        trie_scan_leaf(tgt, sorted_idx, l, h, a, na, q);
    }
    l = lo; h = hi;
    for (int r = 0; r < TRIE_RADIUS; r++) {
        l = (h == n_tgt) ? 0 : h;
        const unsigned long long c = codes[l];
        h = l + 1;
        while (h < n_tgt && codes[h] == c) h++;
        trie_scan_leaf(tgt, sorted_idx, l, h, a, na, q);
    }
    MatchRec rec; rec.source = s; rec.target = -1; rec.distance = q.best;
    if (q.count == 2 && q.best < abs_thr && q.best < q.second * rel_thr) rec.target = q.best_idx;   // SIFTDescriptor.swift:398-410
    out[s] = rec;
}

}  // namespace siftmi
