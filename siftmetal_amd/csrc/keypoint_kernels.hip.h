// keypoint_kernels.hip.h -- DoG extrema, sub-pixel refinement, orientation and descriptor
// kernels for gfx950 (MI355X).
//
// Reference kernels these replace (one thread per item, serial window loops, a blocking
// command buffer per stage and octave):
//   siftExtremaList  Sources/MetalShaders/Metal/SIFTExtrema.metal:62-110
//   siftInterpolate  Sources/MetalShaders/Metal/SIFTInterpolate.metal:193-300
//   siftGradient     Sources/MetalShaders/Metal/SIFTGradient.metal:15-39
//   siftOrientation  Sources/MetalShaders/Metal/SIFTOrientation.metal:140-175
//   siftDescriptors  Sources/MetalShaders/Metal/SIFTDescriptor.metal:120-237
// and the Swift glue between them (Sources/SIFTMetal/SIFT/SIFTOctave.swift:198-492).
//
// MI355X design: nothing but the Gaussian stack lives in HBM.  DoG values are formed on the fly
// (D[s] = G[s+1] - G[s], the same single f32 subtraction as Subtract.metal:17-19), gradients are
// evaluated on demand from the Gaussian layer (same ops as siftGradient), all lists are
// device-resident with device-side counters, every stage is launched once per sub-batch for all
// (frame, octave) groups with grid-stride loops over the device counts, and the window loops of
// orientation / descriptor are spread over the 64 lanes of a wavefront with LDS histograms.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dense_kernels.hip.h"

namespace siftmi {

#define SIFTMI_PI_F 3.14159265358979323846264338327950288f

#ifndef SIFTMI_DESC_NCOPY
#define SIFTMI_DESC_NCOPY 2                                // private histogram copies per wavefront in descriptor_kernel (tuning experiments override it)
#endif
#ifndef SIFTMI_DESC_PACK2
#define SIFTMI_DESC_PACK2 1                                // 1: a histogram slot is a pair of u32 and a corner is ONE 64-bit LDS add (descriptor_kernel)
#endif
#ifndef SIFTMI_ORI_NCOPY
#define SIFTMI_ORI_NCOPY 4                                 // same for the 36-bin histogram of orientation_kernel
#endif
constexpr int MAX_OCT = 16;
constexpr int MAX_NG = 10;            // Gaussian layers per octave = nspo + 3, nspo <= 7 (at 8 the extrema walk needs 173 VGPRs and
                                      // hipcc starts copying its in-flight load registers: tools/audit_asm_loads.py)
constexpr int ORI_BINS = 36;
constexpr int DESC_N = 128;

struct KeypointRec {            // == siftmi_keypoint (include/siftmi.h), 44 bytes
    int32_t octave, scale; float sub_scale; int32_t x, y;
    float abs_x, abs_y, norm_x, norm_y, sigma, value;
};
struct ExtremumRec { int32_t x, y, scale; };
struct DescriptorRec { int32_t keypoint; float theta; uint8_t features[DESC_N]; };   // 136 bytes
// One descriptor to compute: (keypoint, theta) plus everything that depends on them alone -- evaluated ONCE, by one lane of
// desc_derive_kernel (large launches) or expand_descriptors_kernel<true> (a frame or two).  Round 5; the descriptor kernel evaluated
// cosf / sinf / powf and five divisions per descriptor on all 64 lanes of its wavefront.  Same functions on the same arguments:
// byte-identical records (tools/dense_stage_times.py digests).  Dense frames: descriptor_kernel 5.89 -> 5.69-5.75 ms for a 46 us launch
// (it touches 270 MB of record lines) and +8 us in the expansion.  48 bytes.
struct alignas(16) DescInput {
    int32_t keypoint; float theta;
    float cosT, sinT;                  // cosf(theta), sinf(theta)
    float hw, inv_hw;                  // histogramWidth = 3 sigma 2^(interval / intervals) and its IEEE reciprocal
    float px, py;                      // (int)abs / delta (SIFTDescriptor.metal:140-141)
    int32_t radius, scale;             // window radius; Gaussian layer
    float inv_cos, inv_sin;            // 1 / cosT, 1 / sinT (the walk's row intervals)
};
__device__ __forceinline__ DescInput make_desc_input(const KeypointRec &kp, int keypoint, float theta, float delta, int scales_per_octave) {
    DescInput d;
    d.keypoint = keypoint; d.theta = theta;
    const int absoluteX = (int)kp.abs_x, absoluteY = (int)kp.abs_y;         // SIFTOctave.swift:417-418
    d.px = (float)absoluteX / delta; d.py = (float)absoluteY / delta;       // metal :140-141
    d.cosT = cosf(theta); d.sinT = sinf(theta);
    const float interval = (float)kp.scale + kp.sub_scale;
    const float intervals = (float)scales_per_octave;
    const float sigma = 1.6f;
    const float sc = sigma * powf(2.0f, interval / intervals);
    d.hw = 3.0f * sc;
    d.inv_hw = 1.0f / d.hw;
    d.radius = (int)(d.hw * sqrtf(2.0f) * ((float)4 + 1.0f) * 0.5f + 0.5f);
    d.scale = kp.scale;
    d.inv_cos = 1.0f / d.cosT; d.inv_sin = 1.0f / d.sinT;                   // (unused where the component is ~0)
    return d;
}

// Per-context tables handed to every keypoint-stage kernel by value (kernarg -> SGPRs).
struct PyramidDesc {
    const float *gauss;                // frame 0, octave 0, layer 0
    size_t frame_stride;               // floats between frames
    size_t oct_offset[MAX_OCT];        // floats from the frame base to G[o][0]
    int32_t w[MAX_OCT], h[MAX_OCT];
    float delta[MAX_OCT];
    float sigma0[MAX_OCT], sigma1[MAX_OCT];   // sigmas[0], sigmas[1] of the octave (SIFTOctave.swift:211)
    float sigmas[MAX_OCT][MAX_NG];     // sigmas[s], s < nspo+3 <= MAX_NG
    int32_t n_octaves, nspo;
    int32_t cap_ext[MAX_OCT], cap_kp[MAX_OCT], cap_desc[MAX_OCT];
    size_t ext_off[MAX_OCT], kp_off[MAX_OCT], desc_off[MAX_OCT];   // element offsets of the octave's segment inside a frame's segment
    size_t ext_frame, kp_frame, desc_frame;                        // elements per frame
    size_t row_off[MAX_OCT], row_frame;                            // keypoint sort: (nspo + 2) * h row buckets per octave
    int32_t only_octave;                                           // -1: a launch covers every (frame, octave) group; o: only octave o's (grid index = frame)
};

// (frame, octave) group of a keypoint-stage workgroup: the grid index itself, or -- when a launch covers one octave only (the
// per-octave chains of a forked single-frame graph) -- the frame index combined with that octave
__device__ __forceinline__ int group_index(const PyramidDesc &P, int idx) { return P.only_octave < 0 ? idx : idx * P.n_octaves + P.only_octave; }

struct DetectParams {              // SIFTInterpolateParameters (SIFTInterpolate.h:14-23) + literals
    float dog_threshold, edge_threshold, max_offset;
    int32_t max_iterations, border, full_neighbourhood;
    float lambda_ori, ori_threshold;
    int32_t ori_smoothing, desc_scales_per_octave;
};

__device__ __forceinline__ const float *layer_ptr(const PyramidDesc &p, int frame, int o, int s) {
    return p.gauss + (size_t)frame * p.frame_stride + p.oct_offset[o] + (size_t)s * p.w[o] * p.h[o];
}

// ------------------------------------------------------------------------------------------------
// Extrema: Sources/MetalShaders/Metal/SIFTExtrema.metal:62-110.
// One lane per column; a wavefront covers 62 output columns (lanes 0 and 63 are halo) so that the
// x-1 / x+1 neighbours come from cross-lane shuffles instead of extra loads: per row and Gaussian
// layer ONE coalesced load.  Each workgroup walks EH rows with a 3-row sliding window of DoG values
// for all nspo+2 layers in registers.  Strict test against neighbours 1..25 of the reference's
// 26-entry table (entry 0 = (-1,-1,-1) skipped unless full_neighbourhood).  The soft-threshold /
// border pre-filter that the reference applies at refinement entry (SIFTInterpolate.metal:208,
// :223) is applied at emission so that noise extrema never reach the list; raw_count still counts
// every strict extremum.
constexpr int EXT_COLS_PER_WAVE = 62, EXT_COLS_PER_BLOCK = 4 * EXT_COLS_PER_WAVE;

// SKIP: the blur kernels left per-(DoG scale, row, 64-column cell) activity flags (dense_kernels.hip.h, struct Activity):
// "some |DoG| here exceeds the refinement-entry threshold".  A candidate needs that at its own pixel, so a row without
// an active cell under this wave cannot emit one; such rows are not tested, and a row is not even loaded unless it or a
// vertical neighbour is active.  On the benchmark frames 88 % of octave 0's row segments are inactive.  The emitted
// candidate list is exactly the same; raw_count then only counts the extrema of tested rows (cfg.count_raw_extrema = 1
// turns the flags off and restores the full count).
// WPB: wavefronts per workgroup (4, or 1 -- the wavefronts of a workgroup scan independent column strips and only meet at the list
// flush; on flagged frames their row counts differ widely and a workgroup's slots stay held until the busiest is done)
template <int NS, bool SKIP = false, int WPB = 4>
__global__ __launch_bounds__(64 * WPB) void extrema_kernel(PyramidDesc P, DetectParams prm, int o, int EH,
                                                     ExtremumRec *__restrict__ lists, int32_t *__restrict__ cand_count,
                                                     int32_t *__restrict__ raw_count, const unsigned char *__restrict__ act /* octave planes, frame 0 */,
                                                     size_t act_frame_stride, int ncell) {
    constexpr int ND = NS + 2;
    // Workgroup-level staging (as the reference's threadgroup array, SIFTExtrema.metal:71-75, 101-109):
    // one global atomic per workgroup and counter.  A single device-scope counter serialises at
    // ~12 ns per atomic on MI355X, which at one atomic per extremum cost more than the whole scan.
    constexpr int STAGE = 256 * WPB;
    __shared__ ExtremumRec stage[STAGE];
    __shared__ int s_cand, s_raw, s_base;
    if (threadIdx.x == 0) { s_cand = 0; s_raw = 0; }
    __syncthreads();
    const int w = P.w[o], h = P.h[o];
    const int frame = blockIdx.z, group = frame * P.n_octaves + o;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = blockIdx.x * (WPB * EXT_COLS_PER_WAVE) + wv * EXT_COLS_PER_WAVE + lane;      // lane 0 = column x0-1 ... (first output column is 1)
    const int ya = blockIdx.y * EH + 1;                    // first output row of this block
    const int yb = min(ya + EH, h - 1);                    // one past the last output row
    if (ya >= yb || w < 3) return;                         // uniform for the workgroup
    const size_t n = (size_t)w * h;
    const float *g0 = layer_ptr(P, frame, o, 0);
    ExtremumRec *glist = lists + (size_t)frame * P.ext_frame + P.ext_off[o];
    const bool active = (lane >= 1 && lane <= 62 && x >= 1 && x <= w - 2);
    const int xc = min(x, w - 1);                          // clamped column so out-of-image lanes load safely

    // The walk.  `need` (wave-uniform, bit i = row ya - 1 + i) marks the rows whose values are read, `centre` the rows that
    // are tested.  The needed rows are visited in order, k = 0, 1, 2, ...; the k-th one lives in window slot k % 3 and its
    // raw Gaussian values arrive in load buffer k % 3, both static because the loop is unrolled three times.  A row is
    // tested when the row below it has arrived; then the three rows of its window are the last three needed rows (need
    // contains every tested row's neighbours), i.e. slots (k-2) % 3, (k-1) % 3, k % 3.
    // Loads run THREE needed rows ahead of their use whatever the gaps between needed rows (18 loads = 4.6 KB per wave in
    // flight).  hipcc cannot count that: with the loads behind wave-uniform branches its merged s_waitcnt scoreboard waits
    // for vmcnt(0), and round 1's one-row-ahead form exposed a full memory latency at the start of every cluster of flagged
    // rows (the flagged scan ran 1.7x faster than the full one while loading 3x less).  So the loads are asm statements
    // the compiler does not count, issued UNCONDITIONALLY once per visited row (past the last needed row they re-read
    // it), which makes the wait a constant: when row k is consumed exactly the loads of rows k+1 and k+2 are younger,
    // s_waitcnt vmcnt(2 (ND + 1)).  (Any other vector-memory instruction issued in between only makes that wait longer.)
    float win0[ND][3], win1[ND][3], win2[ND][3];
    float buf0[ND + 1], buf1[ND + 1], buf2[ND + 1];
    const float *lbase[ND + 1];
#pragma unroll
    for (int l = 0; l <= ND; l++) lbase[l] = g0 + (size_t)l * n;           // wave-uniform layer bases (SGPR pairs)
    auto issue = [&](int y, float (&buf)[ND + 1]) {
        const unsigned voff = (unsigned)((min(y, h - 1) * w + xc) * 4);     // byte offset inside a layer (< 2^31: siftmi_create)
        asm volatile("s_nop 4" ::: "memory");                               // SGPR bases may be fresh: VMEM reads them (ISA hazard table)
#pragma unroll
        for (int l = 0; l <= ND; l++) asm volatile("global_load_dword %0, %1, %2" : "=v"(buf[l]) : "v"(voff), "s"(lbase[l]) : "memory");
    };
    auto finish = [&](float (&buf)[ND + 1], float (&row)[ND][3]) {
        // rows k+1 and k+2 are younger: everything older than their 2 (ND + 1) loads has landed
        if constexpr (ND + 1 == 4) asm volatile("s_waitcnt vmcnt(8)" : "+v"(buf[0]), "+v"(buf[1]), "+v"(buf[2]), "+v"(buf[3])::"memory");
        if constexpr (ND + 1 == 5) asm volatile("s_waitcnt vmcnt(10)" : "+v"(buf[0]), "+v"(buf[1]), "+v"(buf[2]), "+v"(buf[3]), "+v"(buf[4])::"memory");
        if constexpr (ND + 1 == 6) asm volatile("s_waitcnt vmcnt(12)" : "+v"(buf[0]), "+v"(buf[1]), "+v"(buf[2]), "+v"(buf[3]), "+v"(buf[4]), "+v"(buf[5])::"memory");
        if constexpr (ND + 1 == 7) asm volatile("s_waitcnt vmcnt(14)" : "+v"(buf[0]), "+v"(buf[1]), "+v"(buf[2]), "+v"(buf[3]), "+v"(buf[4]), "+v"(buf[5]), "+v"(buf[6])::"memory");
        if constexpr (ND + 1 == 8) asm volatile("s_waitcnt vmcnt(16)" : "+v"(buf[0]), "+v"(buf[1]), "+v"(buf[2]), "+v"(buf[3]), "+v"(buf[4]), "+v"(buf[5]), "+v"(buf[6]), "+v"(buf[7])::"memory");
        if constexpr (ND + 1 == 9) asm volatile("s_waitcnt vmcnt(18)" : "+v"(buf[0]), "+v"(buf[1]), "+v"(buf[2]), "+v"(buf[3]), "+v"(buf[4]), "+v"(buf[5]), "+v"(buf[6]), "+v"(buf[7]), "+v"(buf[8 % (ND + 1)])::"memory");
        if constexpr (ND + 1 == 10) asm volatile("s_waitcnt vmcnt(20)" : "+v"(buf[0]), "+v"(buf[1]), "+v"(buf[2]), "+v"(buf[3]), "+v"(buf[4]), "+v"(buf[5]), "+v"(buf[6]), "+v"(buf[7]), "+v"(buf[8 % (ND + 1)]), "+v"(buf[9 % (ND + 1)])::"memory");
        static_assert(ND + 1 >= 4 && ND + 1 <= 10, "a wait statement per layer count (the registers it ties are the row's load destinations)");
#pragma unroll
        for (int l = 0; l < ND; l++) {
            const float dv = buf[l + 1] - buf[l];
            row[l][1] = dv;
            // x-1 / x+1 by DPP wavefront shifts (full-rate vector moves; a ds_bpermute per neighbour made the LDS crossbar the
            // bound of the whole scan).  Lanes 0 and 63 keep their own value, as __shfl_up / __shfl_down would: they are halo lanes.
            row[l][0] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, dv), __builtin_bit_cast(int, dv), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
            row[l][2] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, dv), __builtin_bit_cast(int, dv), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
        }
    };
    const float pre = prm.dog_threshold * 0.8f;
    const int border = prm.border;
    const bool full = prm.full_neighbourhood != 0;
    // value < min(all) / value > max(all) over the reference's neighbour list, evaluated separably:
    // per (layer, row) the max/min of the three columns, with the centre excluded in (s, mid) and
    // table entry 0 = (x-1, y-1, s-1) excluded in (s-1, top) unless full_neighbourhood.  min/max are
    // exact, so the grouping does not change the result.
    auto test_row = [&](int y, const float (&top)[ND][3], const float (&mid)[ND][3], const float (&bot)[ND][3]) {
        float M[3][ND], m[3][ND];
#pragma unroll
        for (int l = 0; l < ND; l++) {
            M[0][l] = fmaxf(fmaxf(top[l][0], top[l][1]), top[l][2]); m[0][l] = fminf(fminf(top[l][0], top[l][1]), top[l][2]);
            M[1][l] = fmaxf(fmaxf(mid[l][0], mid[l][1]), mid[l][2]); m[1][l] = fminf(fminf(mid[l][0], mid[l][1]), mid[l][2]);
            M[2][l] = fmaxf(fmaxf(bot[l][0], bot[l][1]), bot[l][2]); m[2][l] = fminf(fminf(bot[l][0], bot[l][1]), bot[l][2]);
        }
#pragma unroll
        for (int s = 1; s <= NS; s++) {
            const float v = mid[s][1];
            const float q_mx = full ? M[0][s - 1] : fmaxf(top[s - 1][1], top[s - 1][2]);
            const float q_mn = full ? m[0][s - 1] : fminf(top[s - 1][1], top[s - 1][2]);
            float mx = fmaxf(fmaxf(q_mx, M[1][s - 1]), M[2][s - 1]);
            float mn = fminf(fminf(q_mn, m[1][s - 1]), m[2][s - 1]);
            mx = fmaxf(fmaxf(mx, M[0][s]), M[2][s]);
            mn = fminf(fminf(mn, m[0][s]), m[2][s]);
            mx = fmaxf(fmaxf(mx, mid[s][0]), mid[s][2]);
            mn = fminf(fminf(mn, mid[s][0]), mid[s][2]);
            mx = fmaxf(fmaxf(mx, M[0][s + 1]), fmaxf(M[1][s + 1], M[2][s + 1]));
            mn = fminf(fminf(mn, m[0][s + 1]), fminf(m[1][s + 1], m[2][s + 1]));
            mx = fmaxf(mx, -1000.0f); mn = fminf(mn, +1000.0f);           // sentinels, SIFTExtrema.metal:81-82
            const bool ext = active && ((v < mn) || (v > mx));
            if (ext) {
                atomicAdd(&s_raw, 1);
                const bool oob = x < border || x > w - border - 1 || y < border || y > h - border - 1;
                if (fabsf(v) > pre && !oob) {
                    ExtremumRec e; e.x = x; e.y = y; e.scale = s;
                    const int li = atomicAdd(&s_cand, 1);
                    if (li < STAGE) {
                        stage[li] = e;
                    } else {                                              // staging full (very dense input): direct append
                        const int idx = atomicAdd(&cand_count[group], 1);
                        if (idx < P.cap_ext[o]) glist[idx] = e;
                    }
                }
            }
        }
    };
    // bit i of `centre`: row ya - 1 + i is an output row of this block (SKIP: with an active cell under this wave's columns);
    // bit i of `need`: that row, or the one above or below it, is such a row, so its values are read.  Wave-uniform.
    unsigned long long centre;
    if (!SKIP) {
        centre = ((1ull << (yb - ya)) - 1ull) << 1;                                       // rows ya ... yb-1 (EH + 2 <= 64)
    } else {
        const int xw = blockIdx.x * (WPB * EXT_COLS_PER_WAVE) + wv * EXT_COLS_PER_WAVE;   // column of lane 0
        const int c0 = min((xw + 1) >> 6, ncell - 1), c1 = min((xw + EXT_COLS_PER_WAVE) >> 6, ncell - 1);
        const unsigned char *ap = act + (size_t)frame * act_frame_stride;
        bool a = false;
        const int row = ya - 1 + lane;
        if (lane >= 1 && row < yb) {
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const unsigned char *q = ap + ((size_t)s * h + row) * ncell;
                a = a || q[c0] || q[c1];
            }
        }
        centre = __ballot(a);
    }
#if defined(SIFTMI_EXT_ABL) && SIFTMI_EXT_ABL == 1          // tools: the launch, the flag fetch and the epilogue alone
    const unsigned long long need = 0ull & centre;
#else
    const unsigned long long need = centre | (centre << 1) | (centre >> 1);
#endif
    if (need != 0ull) {
        unsigned long long todo = need;                     // needed rows not yet requested
        int last = ya - 1;
        auto next_row = [&]() {                             // the next needed row; past the end, the last one again
            if (todo) { last = ya - 1 + __builtin_ctzll(todo); todo &= todo - 1ull; }
            return last;
        };
        const int total = __popcll(need);
#if defined(SIFTMI_EXT_ABL) && SIFTMI_EXT_ABL == 2          // tools: rows are read and differenced, none is tested
        auto tested = [&](int y) { return y < -1000000; };
#else
        auto tested = [&](int y) { return y >= ya && ((centre >> (y - (ya - 1))) & 1ull) != 0; };
#endif
        int r0 = next_row(); issue(r0, buf0);
        int r1 = next_row(); issue(r1, buf1);
        int r2 = next_row(); issue(r2, buf2);
        for (int k = 0; k < total; k += 3) {
            finish(buf0, win0);
            if (tested(r0 - 1)) test_row(r0 - 1, win1, win2, win0);
            r0 = next_row(); issue(r0, buf0);
            if (k + 1 < total) {
                finish(buf1, win1);
                if (tested(r1 - 1)) test_row(r1 - 1, win2, win0, win1);
                r1 = next_row(); issue(r1, buf1);
            }
            if (k + 2 < total) {
                finish(buf2, win2);
                if (tested(r2 - 1)) test_row(r2 - 1, win0, win1, win2);
                r2 = next_row(); issue(r2, buf2);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(buf0[0]), "+v"(buf1[0]), "+v"(buf2[0])::"memory");   // the tail re-reads land before the registers die
    }
    __syncthreads();
    const int nloc = min(s_cand, STAGE);
    if (threadIdx.x == 0) {
        if (s_raw) atomicAdd(&raw_count[group], s_raw);
        s_base = nloc ? atomicAdd(&cand_count[group], nloc) : 0;
    }
    __syncthreads();
    const int base = s_base;
    for (int i = threadIdx.x; i < nloc; i += 64 * WPB)
        if (base + i < P.cap_ext[o]) glist[base + i] = stage[i];
}

// ------------------------------------------------------------------------------------------------
// Refinement: SIFTInterpolate.metal:17-300 + SIFTOctave.interpolateKeypoints (SIFTOctave.swift:205-288)
struct DogTex {
    const float *g; int w, h, nd; size_t n;
    // the octave's Gaussian stack as ONE range-checked buffer (round 5), valid when its nd + 1 layers stay below 4 GB: the refinement's
    // row triples then come as one 12-byte load each (load_neighbourhood)
    __amdgpu_buffer_rsrc_t rs; bool wide;
    __device__ __forceinline__ float rd(int x, int y, int s) const {
        if (x < 0 || y < 0 || s < 0 || x >= w || y >= h || s >= nd) return 0.0f;
        const float *p = g + (size_t)s * n + (size_t)y * w + x;
        return p[n] - p[0];
    }
};

__device__ __forceinline__ void cross3(const float a[3], const float b[3], float r[3]) {
    r[0] = a[1] * b[2] - a[2] * b[1];
    r[1] = a[2] * b[0] - a[0] * b[2];
    r[2] = a[0] * b[1] - a[1] * b[0];
}

// The 19 DoG values around (x, y, s) that one refinement step, the final contrast and the edge test read
// (SIFTInterpolate.metal:17-176 read them texel by texel, three times over).  Loaded ONCE per step: for an interior
// position the 28 Gaussian values behind them are fetched by unconditional, independent loads (one memory latency per
// step; round 1 read each DoG value behind its own bounds test, i.e. ~50 dependent conditional loads per step, and
// the kernel was latency-bound at 0.5 ms per launch), and a converged candidate's last step leaves everything its
// final tests need.  Each DoG value is still the single f32 subtraction G[s+1] - G[s] (Subtract.metal:17-19).
struct DogNeighbourhood {
    float zzz, pzz, nzz, zpz, znz, ppz, nnz, npz, pnz;      // scale s:   centre, x+-1, y+-1, diagonals  (p = +1, n = -1, z = 0; order x y s)
    float zzp, pzp, nzp, zpp, znp;                          // scale s+1: centre, x+-1, y+-1
    float zzn, pzn, nzn, zpn, znn;                          // scale s-1
};

typedef unsigned refine_u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void load_neighbourhood(const DogTex &t, int x, int y, int s, DogNeighbourhood &d) {
    if (t.wide && x >= 1 && y >= 1 && s >= 1 && x <= t.w - 2 && y <= t.h - 2 && s <= t.nd - 2) {
        // Round 5: the same 28 Gaussian values as the branch below from 12 loads instead of 28 -- a row's x - 1, x, x + 1 as one 12-byte
        // buffer load (4-byte aligned), the layer through the scalar offset.  The candidates of a wavefront are scattered, so every
        // dword load of the old form was its own 32-byte sector request and the kernel ran at the sector rate of the memory system
        // (dense frames: 2.4 M candidates x 1.5 steps x 28 sectors in 0.77 ms); same values, same arithmetic.
        const unsigned n4 = (unsigned)t.n * 4u, w4 = (unsigned)t.w * 4u;
        const unsigned row = ((unsigned)(s - 1) * (unsigned)t.n + (unsigned)y * (unsigned)t.w + (unsigned)(x - 1)) * 4u;   // (x - 1, y) of Gaussian layer s - 1; < 2^32 (wide)
        const unsigned up = row - w4, dn = row + w4;
        auto ld3 = [&](unsigned voff, unsigned soff, float &a, float &b, float &c) {
            const refine_u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(t.rs, (int)voff, (int)soff, 0);
            a = __uint_as_float(v.x); b = __uint_as_float(v.y); c = __uint_as_float(v.z);
        };
        auto ld1 = [&](unsigned voff, unsigned soff) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(t.rs, (int)voff, (int)soff, 0)); };
        float a_n, a_z, a_p, b_nn, b_zn, b_pn, b_nz, b_zz, b_pz, b_np, b_zp, b_pp, c_nn, c_zn, c_pn, c_nz, c_zz, c_pz, c_np, c_zp, c_pp, d_n, d_z, d_p;
        ld3(row, 0, a_n, a_z, a_p);
        const float a_zn = ld1(up + 4, 0), a_zp = ld1(dn + 4, 0);
        ld3(up, n4, b_nn, b_zn, b_pn); ld3(row, n4, b_nz, b_zz, b_pz); ld3(dn, n4, b_np, b_zp, b_pp);
        ld3(up, 2 * n4, c_nn, c_zn, c_pn); ld3(row, 2 * n4, c_nz, c_zz, c_pz); ld3(dn, 2 * n4, c_np, c_zp, c_pp);
        ld3(row, 3 * n4, d_n, d_z, d_p);
        const float d_zn = ld1(up + 4, 3 * n4), d_zp = ld1(dn + 4, 3 * n4);
        d.zzz = c_zz - b_zz; d.pzz = c_pz - b_pz; d.nzz = c_nz - b_nz; d.zpz = c_zp - b_zp; d.znz = c_zn - b_zn;
        d.ppz = c_pp - b_pp; d.nnz = c_nn - b_nn; d.npz = c_np - b_np; d.pnz = c_pn - b_pn;
        d.zzp = d_z - c_zz; d.pzp = d_p - c_pz; d.nzp = d_n - c_nz; d.zpp = d_zp - c_zp; d.znp = d_zn - c_zn;
        d.zzn = b_zz - a_z; d.pzn = b_pz - a_p; d.nzn = b_nz - a_n; d.zpn = b_zp - a_zp; d.znn = b_zn - a_zn;
    } else
    if (x >= 1 && y >= 1 && s >= 1 && x <= t.w - 2 && y <= t.h - 2 && s <= t.nd - 2) {
        const float *c = t.g + (size_t)s * t.n + (size_t)y * t.w + x;       // Gaussian layer s at (x, y)
        const int w = t.w;
        const size_t n = t.n;
        // Gaussian layers s-1 (5 values), s and s+1 (9 each), s+2 (5)
        const float a_z = c[-(ptrdiff_t)n], a_p = c[-(ptrdiff_t)n + 1], a_n = c[-(ptrdiff_t)n - 1], a_zp = c[-(ptrdiff_t)n + w], a_zn = c[-(ptrdiff_t)n - w];
        const float b_zz = c[0], b_pz = c[1], b_nz = c[-1], b_zp = c[w], b_zn = c[-w], b_pp = c[w + 1], b_nn = c[-w - 1], b_np = c[w - 1], b_pn = c[-w + 1];
        const float *e = c + n;
        const float c_zz = e[0], c_pz = e[1], c_nz = e[-1], c_zp = e[w], c_zn = e[-w], c_pp = e[w + 1], c_nn = e[-w - 1], c_np = e[w - 1], c_pn = e[-w + 1];
        const float *f = e + n;
        const float d_z = f[0], d_p = f[1], d_n = f[-1], d_zp = f[w], d_zn = f[-w];
        d.zzz = c_zz - b_zz; d.pzz = c_pz - b_pz; d.nzz = c_nz - b_nz; d.zpz = c_zp - b_zp; d.znz = c_zn - b_zn;
        d.ppz = c_pp - b_pp; d.nnz = c_nn - b_nn; d.npz = c_np - b_np; d.pnz = c_pn - b_pn;
        d.zzp = d_z - c_zz; d.pzp = d_p - c_pz; d.nzp = d_n - c_nz; d.zpp = d_zp - c_zp; d.znp = d_zn - c_zn;
        d.zzn = b_zz - a_z; d.pzn = b_pz - a_p; d.nzn = b_nz - a_n; d.zpn = b_zp - a_zp; d.znn = b_zn - a_zn;
    } else {                                                 // at the image or scale border (only with image_border = 0): reads outside are 0
        d.zzz = t.rd(x, y, s);
        d.pzz = t.rd(x + 1, y, s); d.nzz = t.rd(x - 1, y, s); d.zpz = t.rd(x, y + 1, s); d.znz = t.rd(x, y - 1, s);
        d.ppz = t.rd(x + 1, y + 1, s); d.nnz = t.rd(x - 1, y - 1, s); d.npz = t.rd(x - 1, y + 1, s); d.pnz = t.rd(x + 1, y - 1, s);
        d.zzp = t.rd(x, y, s + 1); d.pzp = t.rd(x + 1, y, s + 1); d.nzp = t.rd(x - 1, y, s + 1); d.zpp = t.rd(x, y + 1, s + 1); d.znp = t.rd(x, y - 1, s + 1);
        d.zzn = t.rd(x, y, s - 1); d.pzn = t.rd(x + 1, y, s - 1); d.nzn = t.rd(x - 1, y, s - 1); d.zpn = t.rd(x, y + 1, s - 1); d.znn = t.rd(x, y - 1, s - 1);
    }
}

__device__ __forceinline__ void derivatives3d(const DogNeighbourhood &d, float dD[3]) {   // :64-87
    dD[0] = (d.pzz - d.nzz) * 0.5f; dD[1] = (d.zpz - d.znz) * 0.5f; dD[2] = (d.zzp - d.zzn) * 0.5f;
}

__device__ __forceinline__ void interpolation_step(const DogNeighbourhood &d, float alpha[3]) {   // :90-176, Common.hpp:34-47
    const float dxx = d.pzz + d.nzz - 2.0f * d.zzz;
    const float dyy = d.zpz + d.znz - 2.0f * d.zzz;
    const float dss = d.zzp + d.zzn - 2.0f * d.zzz;
    const float dxy = (d.ppz - d.npz - d.pnz + d.nnz) * 0.25f;
    const float dxs = (d.pzp - d.nzp - d.pzn + d.nzn) * 0.25f;
    const float dys = (d.zpp - d.znp - d.zpn + d.znn) * 0.25f;
    const float x0[3] = {dxx, dxy, dxs}, x1[3] = {dxy, dyy, dys}, x2[3] = {dxs, dys, dss};
    float c12[3], c20[3], c01[3];
    cross3(x1, x2, c12); cross3(x2, x0, c20); cross3(x0, x1, c01);
    const float det = x0[0] * c12[0] + x0[1] * c12[1] + x0[2] * c12[2];
    const float inv = 1.0f / det;
    float dD[3];
    derivatives3d(d, dD);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float h0 = -1.0f * (inv * c12[k]), h1 = -1.0f * (inv * c20[k]), h2 = -1.0f * (inv * c01[k]);
        alpha[k] = h0 * dD[0] + h1 * dD[1] + h2 * dD[2];
    }
}

__device__ __forceinline__ bool is_on_edge(const DogNeighbourhood &d, float edgeThreshold) {   // :17-61
    const float v = d.zzz;
    const float zn = d.znz, zp = d.zpz;
    const float pz = d.pzz, nz = d.nzz;
    const float pp = d.ppz, np = d.npz;
    const float pn = d.pnz, nn = d.nnz;
    const float hxx = zn + zp - 2.0f * v;
    const float hyy = pz + nz - 2.0f * v;
    const float hxy = ((pp - np) - (pn - nn)) * 0.25f;
    const float trace = hxx + hyy;
    const float determinant = (hxx * hyy) - (hxy * hxy);
    if (determinant <= 0.0f) return true;
    const float threshold = ((edgeThreshold + 1.0f) * (edgeThreshold + 1.0f)) / edgeThreshold;
    const float curvature = (trace * trace) / determinant;
    return curvature >= threshold;
}

__device__ __forceinline__ bool out_of_bounds(int x, int y, int s, int w, int h, int scales, int border) {   // :179-190
    return x < border || x > w - border - 1 || y < border || y > h - border - 1 || s < 1 || s > scales;
}

// grid: (blocks, n_groups); group = frame * n_octaves + octave; grid-stride over the candidates.
// Survivors are appended (unordered) to kp_tmp with a 64-bit sort key; the kp_row_* kernels
// order them.
__global__ __launch_bounds__(256) void refine_kernel(PyramidDesc P, DetectParams prm,
                                                    const ExtremumRec *__restrict__ lists, const int32_t *__restrict__ cand_count,
                                                    KeypointRec *__restrict__ kp_tmp, unsigned long long *__restrict__ kp_keys,
                                                    int32_t *__restrict__ kp_count, int32_t *__restrict__ row_count) {
    __shared__ int s_n, s_base;
    const int group = group_index(P, blockIdx.y), frame = group / P.n_octaves, o = group - frame * P.n_octaves;
    const int n = min(cand_count[group], P.cap_ext[o]);
    const int w = P.w[o], h = P.h[o];
    DogTex t;
    t.g = layer_ptr(P, frame, o, 0); t.w = w; t.h = h; t.nd = P.nspo + 2; t.n = (size_t)w * h;
    {   // (group-uniform: every lane of the workgroup refines candidates of one (frame, octave))
        const unsigned long long bytes = (unsigned long long)(P.nspo + 3) * (unsigned long long)w * h * 4ull;
        t.wide = bytes < 0xffffff00ull;
        const unsigned long long a = (unsigned long long)t.g;
        const unsigned long long u = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
        t.rs = __builtin_amdgcn_make_buffer_rsrc((void *)u, 0, t.wide ? (int)(unsigned)bytes : 0, 0x00020000);
    }
    const float delta = P.delta[o];
    const float sigmaRatio = P.sigma1[o] / P.sigma0[o];
    const ExtremumRec *list = lists + (size_t)frame * P.ext_frame + P.ext_off[o];
    for (int k0 = blockIdx.x * 256; k0 < n; k0 += gridDim.x * 256) {      // uniform trip count per workgroup
        const int k = k0 + threadIdx.x;
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        bool keep = false;
        KeypointRec r;
        unsigned long long key = 0;
        if (k < n) {
            const ExtremumRec e = list[k];
            int x = e.x, y = e.y, s = e.scale;
            float value = t.rd(x, y, s);
            bool ok = !(fabsf(value) <= prm.dog_threshold * 0.8f) && !out_of_bounds(x, y, s, w, h, P.nspo, prm.border);
            bool converged = false;
            float alpha[3] = {0.0f, 0.0f, 0.0f};
            DogNeighbourhood nb;
            int i = 0;
            while (ok && i < prm.max_iterations) {
                load_neighbourhood(t, x, y, s, nb);
                interpolation_step(nb, alpha);
                if (fabsf(alpha[0]) < prm.max_offset && fabsf(alpha[1]) < prm.max_offset && fabsf(alpha[2]) < prm.max_offset) {
                    converged = true;
                    break;
                }
                if (alpha[0] > +prm.max_offset) x += 1;
                if (alpha[0] < -prm.max_offset) x -= 1;
                if (alpha[1] > +prm.max_offset) y += 1;
                if (alpha[1] < -prm.max_offset) y -= 1;
                if (alpha[2] > +prm.max_offset) s += 1;
                if (alpha[2] < -prm.max_offset) s -= 1;
                if (out_of_bounds(x, y, s, w, h, P.nspo, prm.border)) ok = false;
                i += 1;
            }
            ok = ok && converged;
            if (ok) {                                               // converged: nb is the neighbourhood of the final (x, y, s)
                float dD[3];
                derivatives3d(nb, dD);
                const float cx = dD[0] * alpha[0];                  // :96-99 x term only
                value = nb.zzz + cx * 0.5f;
                ok = !(fabsf(value) <= prm.dog_threshold) && !is_on_edge(nb, prm.edge_threshold);
            }
            if (ok) {
                keep = true;                                        // SIFTOctave.swift:266-284
                r.octave = o; r.scale = s; r.sub_scale = alpha[2];
                r.x = x; r.y = y;
                r.abs_x = ((float)x + alpha[0]) * delta;
                r.abs_y = ((float)y + alpha[1]) * delta;
                r.norm_x = (float)x / (float)w;
                r.norm_y = (float)y / (float)h;
                r.sigma = P.sigmas[o][s] * powf(sigmaRatio, alpha[2]);
                r.value = value;
                const unsigned int kref = (unsigned int)((s * h + y) * w + x);
                const unsigned int korg = (unsigned int)((e.scale * h + e.y) * w + e.x);
                key = ((unsigned long long)kref << 32) | korg;
            }
        }
        // one global atomic per workgroup and iteration (a single hot counter serialises at ~12 ns/atomic)
        const int li = keep ? atomicAdd(&s_n, 1) : 0;
        __syncthreads();
        if (threadIdx.x == 0) s_base = s_n ? atomicAdd(&kp_count[group], s_n) : 0;
        __syncthreads();
        if (keep && s_base + li < P.cap_kp[o]) {
            const size_t slot = (size_t)frame * P.kp_frame + P.kp_off[o] + s_base + li;
            kp_tmp[slot] = r;
            kp_keys[slot] = key;
            atomicAdd(&row_count[(size_t)frame * P.row_frame + P.row_off[o] + (size_t)(r.scale * h + r.y)], 1);   // bucket sizes for the sort
        }
        __syncthreads();
    }
}

// Keypoint sort per (frame, octave) group: deterministic order (scale, y, x, then originating extremum) for lists
// whose append order came from atomics; keys are unique.  Bucket sort by pyramid row (scale, y): refine_kernel counted
// the keypoints per row; kp_row_scan_kernel turns the counts into bucket starts; kp_row_scatter_kernel drops (key,
// source index) into the buckets in arbitrary order; kp_row_rank_kernel ranks every entry inside its bucket (a row holds
// a handful of keypoints) and moves the record.  O(n + rows) instead of the O(n^2) of a plain rank sort, which took
// 3.9 ms of a 13 ms 8192 x 8192 tile (50 k keypoints in one group).
__global__ __launch_bounds__(256) void zero_i32_kernel(int32_t *__restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0;
}
__global__ __launch_bounds__(256) void zero2_i32_kernel(int32_t *__restrict__ p, size_t n, int32_t *__restrict__ q, size_t m) {   // two ranges, one launch
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (size_t)gridDim.x * 256) q[i] = 0;
}

__global__ __launch_bounds__(1024) void kp_row_scan_kernel(PyramidDesc P, int32_t *__restrict__ row_count /* in: counts, out: zeros */,
                                                          int32_t *__restrict__ row_start) {
    // one workgroup per group: every thread sums a run of consecutive rows, ONE workgroup-wide scan of the 1024 run totals,
    // then every thread writes the starts of its run (round 1 scanned 1024 rows per trip with three barriers each: 11 trips
    // and 15 us for octave 0 of a 1080p frame, all of it latency)
    __shared__ int wsum[16];
    const int group = group_index(P, blockIdx.x), frame = group / P.n_octaves, o = group - frame * P.n_octaves;
    const int n_rows = (P.nspo + 2) * P.h[o];
    const size_t base = (size_t)frame * P.row_frame + P.row_off[o];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int per = (n_rows + 1023) / 1024;
    const int r0 = min((int)threadIdx.x * per, n_rows), r1 = min(r0 + per, n_rows);
    int sum = 0;
    for (int r = r0; r < r1; r++) sum += row_count[base + r];
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int woff = 0;
    for (int k = 0; k < wv; k++) woff += wsum[k];
    int pos = woff + incl - sum;
    for (int r = r0; r < r1; r++) {
        const int v = row_count[base + r];
        row_start[base + r] = pos;
        row_count[base + r] = 0;
        pos += v;
    }
}

__global__ __launch_bounds__(256) void kp_row_scatter_kernel(PyramidDesc P, const unsigned long long *__restrict__ kp_keys,
                                                            const int32_t *__restrict__ kp_count, const int32_t *__restrict__ row_start,
                                                            int32_t *__restrict__ row_fill, unsigned long long *__restrict__ bucket_keys,
                                                            int32_t *__restrict__ bucket_src) {
    const int group = group_index(P, blockIdx.y), frame = group / P.n_octaves, o = group - frame * P.n_octaves;
    const int n = min(kp_count[group], P.cap_kp[o]);
    const size_t base = (size_t)frame * P.kp_frame + P.kp_off[o], rbase = (size_t)frame * P.row_frame + P.row_off[o];
    const unsigned int w = (unsigned int)P.w[o];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const unsigned long long key = kp_keys[base + i];
        const unsigned int row = (unsigned int)(key >> 32) / w;
        const int slot = row_start[rbase + row] + atomicAdd(&row_fill[rbase + row], 1);
        bucket_keys[base + slot] = key;
        bucket_src[base + slot] = i;
    }
}

__global__ __launch_bounds__(256) void kp_row_rank_kernel(PyramidDesc P, const KeypointRec *__restrict__ kp_tmp,
                                                         const unsigned long long *__restrict__ bucket_keys, const int32_t *__restrict__ bucket_src,
                                                         const int32_t *__restrict__ kp_count, const int32_t *__restrict__ row_start,
                                                         const int32_t *__restrict__ row_fill, KeypointRec *__restrict__ kp_sorted) {
    const int group = group_index(P, blockIdx.y), frame = group / P.n_octaves, o = group - frame * P.n_octaves;
    const int n = min(kp_count[group], P.cap_kp[o]);
    const size_t base = (size_t)frame * P.kp_frame + P.kp_off[o], rbase = (size_t)frame * P.row_frame + P.row_off[o];
    const unsigned int w = (unsigned int)P.w[o];
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const unsigned long long key = bucket_keys[base + j];
        const unsigned int row = (unsigned int)(key >> 32) / w;
        const int b0 = row_start[rbase + row], b1 = b0 + row_fill[rbase + row];
        int rank = b0;
        for (int k = b0; k < b1; k++) rank += (bucket_keys[base + k] < key) ? 1 : 0;
        kp_sorted[base + rank] = kp_tmp[base + bucket_src[base + j]];
    }
}

// The three steps in ONE launch for small launches (a frame or two, per-octave chains of a forked graph: every launch on a
// chain costs its duration plus a dependency gap of several microseconds, and a single 1080p call is bound by exactly
// that).  One 1024-thread workgroup per group; the bucket starts live in LDS (rows[] -- the caller checks that the
// octave's (nspo + 2) h rows fit), the scatter advances them, so that afterwards bucket r is [rows[r-1], rows[r]).
// Keys and source indices still go through global memory between the steps: __syncthreads() orders a workgroup's global
// accesses, and none of those lines was read earlier in the kernel.
__global__ __launch_bounds__(1024) void kp_row_sort_small_kernel(PyramidDesc P, const KeypointRec *__restrict__ kp_tmp,
                                                                const unsigned long long *__restrict__ kp_keys, const int32_t *__restrict__ kp_count,
                                                                int32_t *__restrict__ row_count /* in: counts, out: zeros */,
                                                                unsigned long long *__restrict__ bucket_keys, int32_t *__restrict__ bucket_src,
                                                                KeypointRec *__restrict__ kp_sorted, int n_lds /* keypoints the LDS behind rows[] holds */) {
    extern __shared__ int rows[];                          // [n_rows] bucket starts, then ends; then n_lds keys (u64) and source indices
    __shared__ int wsum[16];
    const int group = group_index(P, blockIdx.x), frame = group / P.n_octaves, o = group - frame * P.n_octaves;
    const int n_rows = (P.nspo + 2) * P.h[o];
    const int n = min(kp_count[group], P.cap_kp[o]);
    const size_t rbase = (size_t)frame * P.row_frame + P.row_off[o], base = (size_t)frame * P.kp_frame + P.kp_off[o];
    const unsigned int w = (unsigned int)P.w[o];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    {   // scan (as kp_row_scan_kernel); the counts come into LDS by coalesced loads first -- a thread summing its own run of rows
        // straight from global memory walked up to 11 dependent loads on a 2160-row octave (23 us for the whole kernel)
        for (int r0 = threadIdx.x; r0 < n_rows; r0 += 4096) {
            int v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = r0 + 1024 * u < n_rows ? row_count[rbase + r0 + 1024 * u] : 0;
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (r0 + 1024 * u < n_rows) { rows[r0 + 1024 * u] = v[u]; row_count[rbase + r0 + 1024 * u] = 0; }
        }
        __syncthreads();
        const int per = (n_rows + 1023) / 1024;
        const int r0 = min((int)threadIdx.x * per, n_rows), r1 = min(r0 + per, n_rows);
        int sum = 0;
        for (int r = r0; r < r1; r++) sum += rows[r];
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int pos = incl - sum;
        for (int k = 0; k < wv; k++) pos += wsum[k];
        for (int r = r0; r < r1; r++) {
            const int v = rows[r];
            rows[r] = pos;
            pos += v;
        }
    }
    __syncthreads();
    if (n <= n_lds) {
        // the usual case of these launches (a few hundred to a few thousand keypoints): bucketed keys and source indices stay
        // in LDS -- two global write -> barrier -> read round trips fewer on a chain that is bound by exactly such latencies
        unsigned long long *lkeys = reinterpret_cast<unsigned long long *>(rows + ((n_rows + 1) & ~1));
        int *lsrc = reinterpret_cast<int *>(lkeys + n_lds);
        for (int i = threadIdx.x; i < n; i += 1024) {
            const unsigned long long key = kp_keys[base + i];
            const int slot = atomicAdd(&rows[(unsigned int)(key >> 32) / w], 1);
            lkeys[slot] = key;
            lsrc[slot] = i;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += 1024) {
            const unsigned long long key = lkeys[j];
            const unsigned int row = (unsigned int)(key >> 32) / w;
            const int b0 = row ? rows[row - 1] : 0, b1 = rows[row];
            int rank = b0;
            for (int k = b0; k < b1; k++) rank += (lkeys[k] < key) ? 1 : 0;
            kp_sorted[base + rank] = kp_tmp[base + lsrc[j]];
        }
        return;
    }
    for (int i = threadIdx.x; i < n; i += 1024) {          // scatter (as kp_row_scatter_kernel)
        const unsigned long long key = kp_keys[base + i];
        const int slot = atomicAdd(&rows[(unsigned int)(key >> 32) / w], 1);
        bucket_keys[base + slot] = key;
        bucket_src[base + slot] = i;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += 1024) {          // rank inside the bucket and move (as kp_row_rank_kernel)
        const unsigned long long key = bucket_keys[base + j];
        const unsigned int row = (unsigned int)(key >> 32) / w;
        const int b0 = row ? rows[row - 1] : 0, b1 = rows[row];
        int rank = b0;
        for (int k = b0; k < b1; k++) rank += (bucket_keys[base + k] < key) ? 1 : 0;
        kp_sorted[base + rank] = kp_tmp[base + bucket_src[base + j]];
    }
}

// ------------------------------------------------------------------------------------------------
// Histogram accumulation: order-free integer sums in LDS.  Measured on MI355X (tools/ubench/ubench_lds_atomic.hip, ubench_mix.hip):
// the native LDS float atomic ds_add_f32 costs ~177 cycles per wave-instruction, ds_add_u64 ~6, ds_add_u32 ~4.  All histogram
// contributions are non-negative, so fixed-point sums are order-independent (bit-reproducible: every launch form gives the same bytes).
// Rounds 1-5 accumulated value x 2^32 in u64 bins (fma + v_cvt_u32_f32 per contribution).  Round 6 accumulates value x 2^24 in u32
// bins and takes the integer from the BITS of a denormal-range float product -- see descriptor_kernel's header; resolution 6e-8 per
// contribution in the reference's units against its own sequential f32 sums' ~1e-6 relative.
typedef __attribute__((address_space(3))) unsigned lds_u32_t;
__device__ __forceinline__ void lds_add_bits(unsigned byte_addr, int u32_offset, float contribution) {
    lds_u32_t *p = (lds_u32_t *)(size_t)byte_addr;
    __hip_atomic_fetch_add(p + u32_offset, __float_as_uint(contribution), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Round 6: atan2(y, x) / 2 pi in (-0.5, 0.5] by the half-angle tangent.  With m = |(x, y)| (which both sample loops need anyway):
// tan(a / 2) = y / (m + x); taking |x| folds the angle into [-pi/2, pi/2] (|t| <= 1, sign of y kept, no octant swap: no min / max), and
// x < 0 reflects it, pi - a' for either sign of y.  Six-term odd polynomial (here / pi: atan(t) / pi = a' / 2 pi); 1.2e-6 rad worst
// case (tools/fit_atan.py 6 pin; the argument of atan is half the angle, so the fit's error doubles).  Rounds 2-5 reduced to an octant
// (min / max of |x|, |y|, three reflections, eight terms: 2.9e-7 rad): this form has one reflection and no v_max3 / v_min -- ~8 ns of
// ~45 per sample at the measured issue costs (profiles/ubench_mix_r06.log).  An angle error e moves e x 8 / 2 pi of a sample's weight to
// the neighbouring orientation bin of the descriptor, and one sample in ~1e5 to the neighbouring bin of the orientation histogram.
// m must be > 0 (the callers add 1e-30 under the root: a normal number -- v_sqrt_f32 flushes denormal inputs), so (0, 0) gives 0 like atan2f.
__device__ __forceinline__ float angle_turns(float y, float x, float m) {
    const float t = y * __builtin_amdgcn_rcpf(m + fabsf(x));
    const float s = t * t;
    float q = 2.410817426e-03f;                           // (the fit is pinned at t = tan(pi / 8): angle_bins36)
    q = fmaf(q, s, -1.149140950e-02f);
    q = fmaf(q, s, 2.634806186e-02f);
    q = fmaf(q, s, -4.273251072e-02f);
    q = fmaf(q, s, 6.323227286e-02f);
    q = fmaf(q, s, -1.060769856e-01f);
    const float r = t * fmaf(s, q, 0.318309886183790672f);
    return x < 0.0f ? 0.5f - r : r;
}

// |atan2(y, x)| in units of 10 degrees (the 36 bins of the orientation histogram), in [0, 18], for y >= 0: the same with the coefficients x 36.
__device__ __forceinline__ float angle_bins36(float y_abs, float x, float m) {
    const float t = y_abs * __builtin_amdgcn_rcpf(m + fabsf(x));
    const float s = t * t;
    float q = 8.678942919e-02f;
    q = fmaf(q, s, -4.136907160e-01f);
    q = fmaf(q, s, 9.485301971e-01f);
    q = fmaf(q, s, -1.538370371e+00f);
    q = fmaf(q, s, 2.276361704e+00f);
    q = fmaf(q, s, -3.818771601e+00f);
    const float r = t * fmaf(s, q, 11.4591559026164642f);
    return x < 0.0f ? 18.0f - r : r;
}

// atan2(y, x) of finite arguments by octant reduction (rounds 2-5's form of every sample; round 6: only the samples next to a 45 / 135 degree
// boundary of the orientation histogram, see orientation_kernel): quotient by v_rcp_f32, odd degree-17 polynomial (tools/fit_atan.py 8),
// 2.4 ulp / 2.9e-7 rad worst case, and an EXACT diagonal gives t = 1 exactly, i.e. one value whatever the gradient's magnitude.
__device__ __forceinline__ float atan2_octant(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float t = mn * __builtin_amdgcn_rcpf(fmaxf(mx, 1.0e-30f));
    const float s = t * t;
    float q = 2.622172935e-03f;
    q = fmaf(q, s, -1.513224095e-02f);
    q = fmaf(q, s, 4.112136364e-02f);
    q = fmaf(q, s, -7.366662472e-02f);
    q = fmaf(q, s, 1.057391018e-01f);
    q = fmaf(q, s, -1.418596953e-01f);
    q = fmaf(q, s, 1.999039650e-01f);
    q = fmaf(q, s, -3.333298564e-01f);
    float r = fmaf(t, s * q, t);
    r = ay > ax ? 1.57079632679489662f - r : r;
    r = x < 0.0f ? 3.14159265358979324f - r : r;
    return copysignf(r, y);
}

// ------------------------------------------------------------------------------------------------
// Gradient on demand: SIFTGradient.metal:15-39 (atan2(tx, ty) -- argument order as in the
// reference -- and |grad| of central differences, mirror edges; outside the image -> (0, 0)) is evaluated inside the two sample loops.
//
// The layer is addressed through a buffer resource: one keypoint per wavefront, so the layer base is wave-uniform and sits
// in SGPRs; a sample then needs one 32-bit byte offset (2 VALU) instead of four 64-bit addresses (the orientation and
// descriptor kernels are VALU-bound: PMC, dense frames).  Interior samples -- all but a handful -- take the four loads at
// constant offsets from it; samples on the image edge take the mirrored path.
struct LayerView {
    __amdgpu_buffer_rsrc_t rsrc;
    const float *g;
    int w, h, pitch;                                                   // pitch = 4 w bytes
};
__device__ __forceinline__ LayerView layer_view(const float *g, int w, int h) {
    const unsigned long long a = (unsigned long long)g;
    const unsigned long long u = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    LayerView v;
    v.g = (const float *)u; v.w = w; v.h = h; v.pitch = 4 * w;
    v.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)u, 0, 4 * w * h, 0x00020000);   // raw 32-bit elements, range-checked
    return v;
}
__device__ __forceinline__ float layer_ld(const LayerView &v, int byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(v.rsrc, byte_off, 0, 0));
}
// the same with a wave-uniform addend in an SGPR (a row pitch): no vector add for the rows above and below
__device__ __forceinline__ float layer_ld_s(const LayerView &v, int byte_off, int sgpr_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(v.rsrc, byte_off, sgpr_off, 0));
}
// ------------------------------------------------------------------------------------------------
// Orientation: SIFTOctave.getKeypointOrientations (SIFTOctave.swift:290-382) + SIFTOrientation.metal.
// One wavefront per keypoint; the (2r+1)^2 window is strided over the 64 lanes into a 36-bin LDS
// histogram (integer sums, see descriptor_kernel's header); smoothing / peak search run on lanes 0..35 with shuffles.
// ori_count[k] = -1 when the host-side border filter of the reference rejects the keypoint.
// COOP (as descriptor_kernel): the four wavefronts of a workgroup share one keypoint -- on a frame or two there are fewer keypoints
// than wavefront slots and a window of ~850 samples is 13 dependent rounds for one wavefront, 4 for a workgroup.  Same samples
// into the same integer bins: bit-identical.
// WPB: wavefronts per workgroup (COOP = false), see descriptor_kernel
// Round 6 (as the descriptor's sample loop; costs in profiles/ubench_mix_r06.log): contributions are denormal floats whose bits go to a u32
// LDS add (24 fractional bits); the Gaussian weight exp(-(i^2 + j^2) / 2 lambda^2 sigma^2) is the product of two entries of a per-keypoint
// table (zero past the window: the padding of a row's last quad needs no mask); the angle's magnitude comes in units of 10 degrees from
// the half-angle tangent (angle_bins36), is rounded to a bin by one conversion and reflected for a negative angle, slot 36 being bin 0 again.
#ifndef SIFTMI_ORI_WAVES
#define SIFTMI_ORI_WAVES 7
#endif
template <bool COOP, int WPB = 4>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(SIFTMI_ORI_WAVES))) void orientation_kernel(PyramidDesc P, DetectParams prm,
                                                         const KeypointRec *__restrict__ kps, const int32_t *__restrict__ kp_count,
                                                         int32_t *__restrict__ ori_count, float *__restrict__ ori_angles) {
    // 4 private copies of the histogram per wave (copy = lane % 4), 37 slots apart so that the copies of a bin sit on different LDS
    // banks: neighbouring lanes (neighbouring pixels) mostly share a bin, and same-address lanes of one LDS add serialise.  Slot 36 = bin 0.
    constexpr int OCOPY = SIFTMI_ORI_NCOPY, OSTRIDE = ORI_BINS + 1;
    constexpr int OTAB = 128;                               // window columns the weight table covers (+ 4 entries of padding)
    static_assert(WPB == 4 || (!COOP && WPB == 1), "COOP shares one keypoint among the four wavefronts of a workgroup");
    __shared__ unsigned hist_all[WPB][OCOPY * OSTRIDE];
    __shared__ float otab_all[WPB][OTAB + 4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned *hist0 = hist_all[COOP ? 0 : wv];
    const unsigned hist_lds = (unsigned)(size_t)(lds_u32_t *)(hist0 + (lane & (OCOPY - 1)) * OSTRIDE);   // byte address of this lane's copy
    float *otab = otab_all[wv];
    constexpr int STRIDE = COOP ? 256 : 64;                 // lanes walking one keypoint's window
    const int lidx = COOP ? (int)threadIdx.x : lane;
    const int group = group_index(P, blockIdx.y), frame = group / P.n_octaves, o = group - frame * P.n_octaves;
    const int n = min(kp_count[group], P.cap_kp[o]);
    const int w = P.w[o], h = P.h[o];
    const float delta = P.delta[o], lambda = prm.lambda_ori;
    // delta = delta_min 2^o with delta_min = 0.5 (siftmi_create accepts nothing else): a power of two, so x / delta == x * (1 / delta) bit for
    // bit -- five IEEE divisions (~10 vector instructions each, executed by all 64 lanes on wave-uniform values) less per keypoint
    const float inv_delta = 1.0f / delta;
    const size_t base = (size_t)frame * P.kp_frame + P.kp_off[o];
    const int k_first = COOP ? (int)blockIdx.x : (int)(blockIdx.x * WPB + wv), k_step = COOP ? (int)gridDim.x : (int)(gridDim.x * WPB);
    KeypointRec kp_next = kps[base + min(k_first, max(n - 1, 0))];      // the record of a wave's NEXT keypoint is requested a keypoint ahead
    for (int k = k_first; k < n; k += k_step) {
        const KeypointRec kp = kp_next;
        kp_next = kps[base + min(k + k_step, n - 1)];
        bool reject;
        {   // SIFTOctave.swift:303-329
            const float minX = 1.0f, minY = 1.0f, maxX = (float)(w - 2), maxY = (float)(h - 2);
            const float x = kp.abs_x * inv_delta, y = kp.abs_y * inv_delta;   // (/ delta: exact either way, delta is a power of two)
            const float sigma = kp.sigma * inv_delta;
            const float r = ceilf(3.0f * lambda * sigma);
            reject = (floorf(x - r) < minX) || (ceilf(x + r) > maxX) || (floorf(y - r) < minY) || (ceilf(y + r) > maxY);
        }
        if (reject) {                                                   // (uniform over the workgroup)
            if (lidx == 0) ori_count[base + k] = -1;
            continue;
        }
        const int absoluteX = (int)kp.abs_x, absoluteY = (int)kp.abs_y;     // :333-334 Int32 truncation
        const LayerView g = layer_view(layer_ptr(P, frame, o, kp.scale), w, h);
        if (COOP) __syncthreads();                                      // wave 0 is done reading the previous keypoint's bins
        for (int c = lidx; c < OCOPY * OSTRIDE; c += STRIDE) hist0[c] = 0u;
        int half_shift = 0;
        {   // SIFTOrientation.metal:87-136
            const int x = (int)roundf((float)absoluteX * inv_delta);
            const int y = (int)roundf((float)absoluteY * inv_delta);
            const float sigma = kp.sigma * inv_delta;
            const float exponentDenominator = 2.0f * lambda * lambda;
            // per keypoint: the reciprocals the sample loop multiplies by (the reference divides per sample; float note at
            // descriptor_kernel: the weight of a sample moves by 1-2 ulp, the bin it goes to does not depend on it), with
            // log2(e) folded into the exponent's factor (v_exp_f32 is 2^x)
            const float inv_sigma = 1.0f / sigma, k_exp = (-1.0f / exponentDenominator) * 1.44269504088896341f;
            const int r = (int)ceilf(3.0f * lambda * sigma);
            const int side = 2 * r + 1, total = side * side;
            const float inv_side = 1.0f / (float)side;                 // idx / side below: idx + 0.5 keeps the quotient >= 0.5 / side off every integer, far more than the float error at idx < 2^20
            // weight of a sample = gauss(i) gauss(j), gauss(k) = exp2(k_exp (k / sigma)^2) 2^-63 2^-half_shift: the product carries 2^-126,
            // which makes (|gradient| weight) a float whose bits are its value in units of 2^-24 (descriptor_kernel's header).  A bin's total
            // is below sqrt(2) / 2 (sum of gauss)^2 <= 0.7072 (2.5067 lambda sigma + 1)^2: 149 with the reference's lambda = 1.5; half_shift
            // scales the unit up when a larger lambda would take that past 2^8.
            const float sg = fmaf(2.5066283f * lambda, sigma, 1.0f), bound = 0.70710678f * sg * sg;
            while (ldexpf(bound, -2 * half_shift) >= 240.0f) half_shift++;   // (wave-uniform; no trip with the reference's parameters)
            const float gscale = ldexpf(1.0f, -63 - half_shift);
            auto gauss = [&](int kk) -> float { const float u = (float)kk * inv_sigma; return __builtin_amdgcn_exp2f((u * u) * k_exp) * gscale; };
            const bool tabled = side <= OTAB;
            if (tabled) for (int c = lane; c < side + 3; c += 64) otab[c] = c < side ? gauss(c - r) : 0.0f;    // (each wave its own copy)
            if (COOP) __syncthreads(); else __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            // The host border filter works on the float position, the window is centred on the rounded truncated one (up to two
            // pixels lower at delta = 0.5): almost every window still has all its samples and their +-1 neighbours inside the
            // image (wave-uniform test) and takes the gradient as wide loads at one offset, without per-sample range tests.
            const bool interior = x - r >= 1 && x + r <= w - 2 && y - r >= 1 && y + r <= h - 2;
            // one sample: central differences (dx, dy), not halved -- the gradient is (dx, dy) / 2 (SIFTGradient.metal:31-32): halving is
            // exact, so the angle atan2(tx, ty) is that of (dx, dy), and |gradient| = sqrt(dx^2 + dy^2) / 2 with the 1/2 in the unit
            auto accumulate = [&](float dx, float dy, float gi, float gj) {
                const float mag = __builtin_amdgcn_sqrtf(fmaf(dx, dx, fmaf(dy, dy, 1.0e-30f)));        // > 0 (angle_turns)
                // bin = round(36 atan2(tx, ty) / 2 pi), negative + 36 (SIFTOrientation.metal:122-129).  round() is half away from zero, i.e.
                // symmetric in the sign of the angle, which is the sign of tx: round the MAGNITUDE (floor(|.| + 0.5), one conversion) and
                // reflect, 36 - q, for tx < 0 -- exactly diagonal gradients (8-bit synthetic images are full of them) then fall to the same
                // side as in the reference whatever their quadrant.  q = 0 reflects to slot 36 = bin 0 again.
                const float a36 = angle_bins36(fabsf(dx), dy, mag);
                int q;
                asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(q) : "v"(a36));
                int b = dx < 0.0f ? ORI_BINS - q : q;
                // Next to the 45 / 135 degree boundaries (|a36 - 4.5| or |a36 - 13.5| below 2e-5: never in natural images, 15 % of the samples of
                // a checkerboard, 3.5 % of them |dx| == |dy| to the bit) the half-angle form is not good enough: its argument carries the errors
                // of v_sqrt and v_rcp, so EXACT diagonals of different magnitude scatter over both sides of the boundary, the histograms of an exactly
                // symmetric pattern pick up 1 % noise, and peak counts start to differ from the reference's (profiles/sweep_cases_r06.log).  Those
                // samples take rounds 2-5's octant form, which maps every exact diagonal to ONE value, and the reference's expression literally.
                if (fabsf(fabsf(a36 - 9.0f) - 4.5f) < 2.0e-5f) {
                    b = (int)roundf(atan2_octant(dx, dy) * (float)(ORI_BINS / (2.0 * 3.14159265358979323846)));
                    if (b < 0) b += ORI_BINS;
                    if (b >= ORI_BINS) b -= ORI_BINS;
                }
                const float m = (mag * gi) * gj;
                lds_add_bits(hist_lds + ((unsigned)b << 2), 0, m);
            };
            if (interior && tabled) {
                // Round 5: an interior window is walked in QUADS, as the descriptor's: four consecutive samples of a window row per lane
                // and trip (the last quad of a row runs up to three columns past the window: the table's weight there is 0).  The index
                // arithmetic, the row's weight and the address are shared, the row's texels arrive as three wide loads and one b64.
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const int qpr = (side + 3) >> 2, total_q = side * qpr;
                const float inv_qpr = 1.0f / (float)qpr;
                struct Quad { u32x4 ra, ru, rd; u32x2 rb; int jj, k4; };
                auto fetch = [&](int q, Quad &t) {
                    t.jj = (int)(((float)q + 0.5f) * inv_qpr); t.k4 = (q - t.jj * qpr) << 2;
                    const int c = __mul24(y + t.jj - r - 1, g.pitch) + ((x + t.k4 - r - 1) << 2);  // texel (x + i0 - 1, y + j - 1), i0 = k4 - r, j = jj - r
                    t.ra = __builtin_amdgcn_raw_buffer_load_b128(g.rsrc, c, g.pitch, 0);           // row y + j: columns x + i0 - 1 ... + 2
                    t.rb = __builtin_amdgcn_raw_buffer_load_b64(g.rsrc, c + 16, g.pitch, 0);       //            ... + 3, + 4
                    t.ru = __builtin_amdgcn_raw_buffer_load_b128(g.rsrc, c + 4, 0, 0);             // row y + j - 1: columns x + i0 ... + 3
                    t.rd = __builtin_amdgcn_raw_buffer_load_b128(g.rsrc, c + 4, 2 * g.pitch, 0);   // row y + j + 1
                };
                auto consume = [&](const Quad &t) {
                    const float gj = otab[t.jj];
                    const float *gq = otab + t.k4;
                    const float G[4] = {gq[0], gq[1], gq[2], gq[3]};
                    const float A[6] = {__uint_as_float(t.ra.x), __uint_as_float(t.ra.y), __uint_as_float(t.ra.z), __uint_as_float(t.ra.w),
                                        __uint_as_float(t.rb.x), __uint_as_float(t.rb.y)};
                    const float U[4] = {__uint_as_float(t.ru.x), __uint_as_float(t.ru.y), __uint_as_float(t.ru.z), __uint_as_float(t.ru.w)};
                    const float D[4] = {__uint_as_float(t.rd.x), __uint_as_float(t.rd.y), __uint_as_float(t.rd.z), __uint_as_float(t.rd.w)};
#pragma unroll
                    for (int s4 = 0; s4 < 4; s4++) accumulate(A[s4 + 2] - A[s4], D[s4] - U[s4], G[s4], gj);
                };
                // (No software pipelining: requesting the next trip's texels before this trip's are used costs 14 registers, the kernel then
                // spills, 1.26 -> 3.4-4.2 ms on dense frames: profiles/desc_variants_r06.log.)
                for (int q = lidx; q < total_q; q += STRIDE) { Quad t; fetch(q, t); consume(t); }
            } else {
                // windows that touch the image border (mirror edges; outside the image -> gradient (0, 0)), or wider than the table
                for (int idx = lidx; idx < total; idx += STRIDE) {
                    const int jj = (int)(((float)idx + 0.5f) * inv_side), ii = idx - jj * side;
                    const int gx = x + ii - r, gy = y + jj - r;
                    float dx = 0.0f, dy = 0.0f;
                    if (gx >= 0 && gy >= 0 && gx < w && gy < h) {
                        const int pxx = symm(gx + 1, w), mxx = symm(gx - 1, w), pyy = symm(gy + 1, h), myy = symm(gy - 1, h);
                        auto rd = [&](int xx, int yy) -> float { return (xx < 0 || yy < 0 || xx >= w || yy >= h) ? 0.0f : g.g[(size_t)yy * w + xx]; };
                        dx = rd(pxx, gy) - rd(mxx, gy);
                        dy = rd(gx, pyy) - rd(gx, myy);
                    }
                    accumulate(dx, dy, tabled ? otab[ii] : gauss(ii - r), tabled ? otab[jj] : gauss(jj - r));
                }
            }
        }
        if (COOP) { __syncthreads(); if (wv != 0) continue; } else __builtin_amdgcn_wave_barrier();   // COOP: wave 0 finishes the keypoint
        __threadfence_block();
        const int li = lane < ORI_BINS ? lane : 0;
        unsigned hsum = 0u;
#pragma unroll
        for (int c = 0; c < OCOPY; c++) hsum += hist0[c * OSTRIDE + li] + (li == 0 ? hist0[c * OSTRIDE + ORI_BINS] : 0u);
        float hv = (float)hsum * ldexpf(1.0f, -24 + 2 * half_shift);
        // the circular neighbours of bin `lane` (lanes 0 ... 35): a wavefront shift by one lane (DPP, a vector move) with the wrap-around lane
        // patched from a v_readlane -- round 6; rounds 1-5 used __shfl, i.e. 14 ds_bpermute round trips through the LDS crossbar per keypoint.
        // Lanes >= 36 hold values nobody reads.
        auto prev_bin = [&](float v) -> float {                         // bin (lane - 1) mod 36
            const float s = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
            const float wrap = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), ORI_BINS - 1));
            return lane == 0 ? wrap : s;
        };
        auto next_bin = [&](float v) -> float {                         // bin (lane + 1) mod 36
            const float s = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
            const float wrap = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
            return lane == ORI_BINS - 1 ? wrap : s;
        };
        for (int it = 0; it < prm.ori_smoothing; it++) {               // :67-84
            const float h0 = prev_bin(hv), h2 = next_bin(hv);
            hv = (h0 + hv + h2) / 3.0f;
        }
        float mxv = (lane < ORI_BINS) ? hv : -3.0e38f;                // :44-47 (max is order-free)
        {   // wave maximum by DPP (row_shr 1 / 2 / 4 / 8, row_bcast 15 / 31: lane 63 ends up with it), broadcast through an SGPR
#define ORI_MAX_DPP(CTRL, ROWMASK) mxv = fmaxf(mxv, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, mxv), __builtin_bit_cast(int, mxv), CTRL, ROWMASK, 0xf, false)))
            ORI_MAX_DPP(0x111, 0xf); ORI_MAX_DPP(0x112, 0xf); ORI_MAX_DPP(0x114, 0xf); ORI_MAX_DPP(0x118, 0xf);
            ORI_MAX_DPP(0x142, 0xa); ORI_MAX_DPP(0x143, 0xc);
#undef ORI_MAX_DPP
            mxv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mxv), 63));
        }
        const float threshold = prm.ori_threshold * mxv;
        const float hm = prev_bin(hv), hp = next_bin(hv);
        const bool peak = (lane < ORI_BINS) && (hv > threshold) && (hv > hm) && (hv > hp);
        const unsigned long long mask = __ballot(peak);
        if (peak) {                                                    // :16-33, :52-61
            const float offset = (hm - hp) / (2.0f * (hm + hp - 2.0f * hv));
            const float tbin = ((float)lane + offset) / (float)ORI_BINS;
            const float tau = 2.0f * SIFTMI_PI_F;
            float orientation = tbin * tau;
            if (orientation < 0.0f) orientation += tau;
            if (orientation >= tau) orientation -= tau;
            const int pos = __popcll(mask & ((1ull << lane) - 1ull));
            ori_angles[(base + k) * ORI_BINS + pos] = orientation;
        }
        if (lane == 0) ori_count[base + k] = __popcll(mask);
        __builtin_amdgcn_wave_barrier();
    }
}

// Exclusive scan of the orientation counts of one group -> descriptor inputs in keypoint order
// (SIFTOctave.swift:411-424 expansion).  One 1024-thread workgroup per group; the scan is a wave-level shuffle scan plus the 16
// wave totals (round 3; it was a 10-step Hillis-Steele scan through LDS with 20 barriers: 7 us on a single frame's chain).
// DERIVE: the (keypoint, theta)-only terms of a descriptor (DescInput) are filled in here -- a frame or two; large launches leave them to
// desc_derive_kernel (one lane per descriptor over the whole chip: here a group's ~15 k keypoints are 15 per thread of ONE workgroup).
template <bool DERIVE>
__global__ __launch_bounds__(1024) void expand_descriptors_kernel(PyramidDesc P, DetectParams prm, const KeypointRec *__restrict__ kps,
                                                                 const int32_t *__restrict__ kp_count,
                                                                 const int32_t *__restrict__ ori_count, const float *__restrict__ ori_angles,
                                                                 DescInput *__restrict__ desc_in, int32_t *__restrict__ desc_count,
                                                                 int32_t *__restrict__ oriented_count) {
    __shared__ int wsum[16], wori[16];
    const int group = group_index(P, blockIdx.x), frame = group / P.n_octaves, o = group - frame * P.n_octaves;
    const int n = min(kp_count[group], P.cap_kp[o]);
    const size_t kbase = (size_t)frame * P.kp_frame + P.kp_off[o];
    const size_t dbase = (size_t)frame * P.desc_frame + P.desc_off[o];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int per = (n + 1023) / 1024;
    const int k0 = threadIdx.x * per, k1 = min(k0 + per, n);
    int sum = 0, nori = 0;
    for (int k = k0; k < k1; k++) { const int c = ori_count[kbase + k]; sum += max(c, 0); nori += (c >= 0); }
    int incl = sum, incl2 = nori;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64), t2 = __shfl_up(incl2, d, 64);
        if (lane >= d) { incl += t; incl2 += t2; }
    }
    if (lane == 63) { wsum[wv] = incl; wori[wv] = incl2; }
    __syncthreads();
    int pos = incl - sum, total = 0, total_ori = 0;
    for (int k = 0; k < 16; k++) { if (k < wv) pos += wsum[k]; total += wsum[k]; total_ori += wori[k]; }
    const float delta = P.delta[o];
    for (int k = k0; k < k1; k++) {
        const int c = ori_count[kbase + k];
        if (c <= 0) continue;
        if (DERIVE) {
            const KeypointRec kp = kps[kbase + k];
            for (int t = 0; t < c; t++, pos++)
                if (pos < P.cap_desc[o])
                    desc_in[dbase + pos] = make_desc_input(kp, k, ori_angles[(kbase + k) * ORI_BINS + t], delta, prm.desc_scales_per_octave);
        } else {
            for (int t = 0; t < c; t++, pos++)
                if (pos < P.cap_desc[o])                                     // (keypoint, theta): the record's first 8 bytes, one store
                    *reinterpret_cast<float2 *>(&desc_in[dbase + pos]) = make_float2(__int_as_float(k), ori_angles[(kbase + k) * ORI_BINS + t]);
        }
    }
    if (threadIdx.x == 0) { desc_count[group] = total; oriented_count[group] = total_ori; }
}

// The remaining fields of the descriptor inputs expand_descriptors_kernel<false> listed: one lane per descriptor.
__global__ __launch_bounds__(256) void desc_derive_kernel(PyramidDesc P, DetectParams prm, const KeypointRec *__restrict__ kps,
                                                         const int32_t *__restrict__ desc_count, DescInput *__restrict__ desc_in) {
    const int group = group_index(P, blockIdx.y), frame = group / P.n_octaves, o = group - frame * P.n_octaves;
    const int n = min(desc_count[group], P.cap_desc[o]);
    const size_t kbase = (size_t)frame * P.kp_frame + P.kp_off[o];
    const size_t dbase = (size_t)frame * P.desc_frame + P.desc_off[o];
    const float delta = P.delta[o];
    for (int di = blockIdx.x * 256 + threadIdx.x; di < n; di += gridDim.x * 256) {
        const float2 kt = *reinterpret_cast<const float2 *>(&desc_in[dbase + di]);
        const int k = __float_as_int(kt.x);
        desc_in[dbase + di] = make_desc_input(kps[kbase + k], k, kt.y, delta, prm.desc_scales_per_octave);
    }
}

// ------------------------------------------------------------------------------------------------
// Descriptor: SIFTOctave.getDescriptors (SIFTOctave.swift:384-492) + SIFTDescriptor.metal:15-237.
// One wavefront per (keypoint, theta); the (2R+1)^2 rotated window is strided over the lanes and
// scattered trilinearly into a 4x4x8 LDS histogram (integer sums, see lds_add_bits); the two L2
// normalisations are wave reductions.  Samples whose truncated coordinate leaves the image contribute nothing (the
// reference's behaviour there is undefined).
// Float note.  Per sample the reference divides by histogramWidth twice, calls exp, sqrt and atan2.  Metal compiles those with
// fast math (reciprocal multiply, native exp / sqrt, ~1-2 ulp); here: one IEEE reciprocal per descriptor and a multiply
// per sample, v_exp_f32 (table entries) and v_sqrt_f32 -- 1-2 ulp on quantities that only weight a sample -- and angle_turns
// (1.2e-6 rad), against a stated descriptor tolerance of 1e-4 (L2 of the unit vector).  Measured against the oracle (IEEE division,
// glibc expf / sqrtf / atan2f) on 17.6 k dense descriptors (tools/desc_margin.py, profiles/desc_margin_r06.log): max L2 3.4e-6 (4.6e-6 on
// the sparse fields), 100 of 2.25 M quantised bins differ by 1.  Rounds 2-5 (2^-32 fixed point, 2.4-ulp atan2): 5.8e-7 / 11 bins, at
// 5.5 instead of 4.2 ms per 1.13 M descriptors: the round-6 forms spend 6 of the tolerance's 170x margin.
// inclusive prefix sum over the 64 lanes by DPP adds (row_shr 1 / 2 / 4 / 8 inside each row of 16, then row_bcast15 / row_bcast31 carry the
// row totals up): six full-rate vector adds.  (__shfl_up goes through the LDS crossbar: six ds_bpermute + wait + select + add.)
__device__ __forceinline__ int wave_inclusive_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118 /* row_shr:8 */, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
    return v;
}
// sum over the 64 lanes, the same on every lane: the DPP tree of wave_inclusive_scan (lane 63 ends up with the total) and a v_readlane.
// (Rounds 1-5: an xor butterfly of six __shfl_xor = six ds_bpermute round trips; the summation order differs, so the normalisation factor of a
// descriptor can differ from round 5's in its last bit -- every launch form uses this one function.)
__device__ __forceinline__ float wave_sum(float v) {
#define WSUM_DPP(CTRL, ROWMASK) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, false))
    WSUM_DPP(0x111, 0xf); WSUM_DPP(0x112, 0xf); WSUM_DPP(0x114, 0xf); WSUM_DPP(0x118, 0xf);
    WSUM_DPP(0x142, 0xa); WSUM_DPP(0x143, 0xc);
#undef WSUM_DPP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// maximum over the 64 lanes of a value >= +0, the same on every lane (the DPP tree of wave_sum; lanes without a source contribute 0)
__device__ __forceinline__ float wave_max_nonneg(float v) {
#define WMAX_DPP(CTRL, ROWMASK) v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, false)))
    WMAX_DPP(0x111, 0xf); WMAX_DPP(0x112, 0xf); WMAX_DPP(0x114, 0xf); WMAX_DPP(0x118, 0xf);
    WMAX_DPP(0x142, 0xa); WMAX_DPP(0x143, 0xc);
#undef WMAX_DPP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// COOP = false: one wavefront per descriptor (large launches: there are more descriptors than wavefronts in flight).
// COOP = true: the four wavefronts of a workgroup share one descriptor -- for a frame or two there are fewer descriptors than
// wavefront slots, and a descriptor's ~3000 samples walked by 64 lanes take 60-100 us of dependent loads and LDS atomics;
// four waves cut that to a quarter.  Same samples into the same u64 fixed-point bins, so the result is bit-identical.
// WPB = wavefronts per workgroup (COOP = false only): 4, or 1 -- a workgroup's LDS and wave slots are held until its LAST wavefront is
// done, and descriptors differ 4x in their sample count, so with four independent wavefronts per workgroup a quarter of the slots
// idles at workgroup tails on dense frames (4.35 of 6 resident, PMC round 4)
#ifndef SIFTMI_DESC_QCAP
#define SIFTMI_DESC_QCAP 832                              // entries of the walk's quad table (0: the row-advance walk of round 5)
#endif
#ifndef SIFTMI_DESC_FLAG_SS
#define SIFTMI_DESC_FLAG_SS 0.25f                           // squared norm (reference units) below which a descriptor goes to the second pass: (2^23 units)^2
#endif
#ifndef SIFTMI_DESC_WAVES
#define SIFTMI_DESC_WAVES 7                                // wavefronts per SIMD the descriptor kernel's register budget is sized for (72 VGPRs);
                                                           // the one-wavefront form of large launches: one more (64 VGPRs)
#endif
// Round 6 -- histogram contributions as DENORMAL floats.  The sample loop is bound by vector issue and its instructions are not equal
// (tools/ubench/ubench_mix.hip, profiles/ubench_mix_r06.log: add / mul / fma 1.1 ns per wavefront and SIMD, compares / conversions / floor /
// fract / shifts / min / max 1.9, v_exp / v_rcp / v_sqrt 3.5; summed over the loop's listing that model gives the measured 5.5 ms).  Rounds
// 2-5 converted every contribution to 2^-32 fixed point (fma + v_cvt_u32_f32, 3.0 ns, eight per sample).  Now the Gaussian weight carries a
// factor 2^-126 (2^-63 in either table entry), which puts every contribution c = w_xy w_bin |gradient| weight into the denormal / first
// normal binade of f32, where the BIT PATTERN of a float is its value in units of 2^-149: __float_as_uint(product) is round-to-nearest-even
// of c 2^24 (|gradient| <= sqrt(2) / 2 keeps it below 2^24, the end of the linear range; gfx950 multiplies denormals at full rate, same log),
// and goes to a u32 LDS add as it is: 1.1 ns per contribution.  Still order-free integer sums (bit-reproducible, every launch form gives the
// same bytes), now with 24 fractional bits instead of 32: rounding 3e-8 per contribution in the reference's units, ~3e-7 per bin, against bin
// totals of 0.1 ... 30 -- this is where the precision budget of the stated 1e-4 goes (measured: profiles/desc_margin_r06.log).
// A bin total stays below (hw + 1)^2 sqrt(2) / 2 (the samples of a cell's support times their bilinear weights); `unit_shift` scales the
// contributions down by a power of two for windows so large that this passes 2^8 (never with the reference's schedule: hw <= 11).
//
// Also round 6 (all byte-neutral between launch forms, all inside the tolerances):
//  * exp(-(rx^2 + ry^2) / 8) = exp(-j^2 / 8 hw^2) exp(-i^2 / 8 hw^2) (a rotation keeps j^2 + i^2): one table of 2 radius + 4 entries per
//    descriptor in LDS, built by the wavefront (two v_exp per lane), read per quad (row entry + four column entries) instead of
//    mul, fma, mul, v_exp per sample;
//  * ten orientation slots per cell -- bins 0 ... 7, slot 8 = bin 0 again, slot 9 only ever receives zeros -- so that the upper
//    orientation bin is the lower one's neighbour in memory (an immediate offset; no second address, no wrap), and the slot index is
//    formed in float (two fmas and ONE conversion for the byte address instead of three conversions and seven integer operations);
//  * floor by v_floor_f32, fraction by subtraction (exact), range tests on the bits of the floor (0 <= f <= 3 is one unsigned compare);
//  * no early exit per sample: a wavefront never skips it (64 lanes, ~90 % of them inside), and the cell tests reject what it rejected;
//  * quads run up to three candidates past a row's interval without masking them: the interval is conservative and a candidate beyond
//    the window's radius cannot pass the cell tests ((radius + 1)^2 > 12.5 hw^2 >= rx^2 + ry^2 of anything inside).
// PATCH (siftmi_config.descriptor_patch_lds; BASELINE north_star: "LDS tile staging for ... 16x16 descriptor patches"): an interior window
// is walked in 16 x 16-sample tiles of its bounding box; a tile's 18 x 18 texels (the samples and their +-1 neighbours) are copied into
// LDS once and every lane takes the four samples of one quarter row from there.  Same samples, same arithmetic, order-free bins:
// byte-identical descriptors (tests/test_gpu_parity.py::test_descriptor_patch_staging_is_byte_identical).  It is NOT the default: a
// window's texels are used ~4 times each and the L1 / L2 already serve that (profiles/pmc_descriptor_dense_r05.txt: 0.9 MB fetched from HBM
// by 2 M workgroups), while tiles of the bounding box visit the corners the compacted row walk skips and every 256 samples wait for a
// load -> LDS -> read round trip (profiles/desc_patch_lds_r05.log; the PATCH form needs 5 wavefronts' worth of registers per SIMD and says so).
// REFINE (round 6, late) -- low-contrast windows.  The 2^-24 unit is ABSOLUTE: a window whose gradients are ~1e-3 keeps ~10 bits per
// contribution, and its unit vector then differs from the oracle's by up to 2e-5 with one integer in 200 off by one (two cases of 240 in
// the sweeps; rounds 2-5, 2^-32: 4e-7 / none).  The roundings add up to ~45 units over a descriptor whatever its contrast, so the NORM of
// the accumulated bins says which descriptors are affected: the launch flags those whose norm is below 2^23 units (expected error above
// ~5e-6; `desc_flag`, one int per descriptor), and a second, small launch -- this kernel with REFINE = true -- walks the flagged ones again
// with the unit 2^-24 4^-fine_h, fine_h = 4, 3, ... (the windows the sweeps found worst are LARGE ones, hw > 17 with seven scales per octave, whose
// default unit is 2^-22: unit_shift).  A finer unit is only valid while every contribution stays in the linear range of the bit trick,
// 2 |gradient| 4^fine_h < 2, and every bin's sum below 2^32 units, so the REFINE walk keeps the largest squared gradient of the contributing
// samples and accepts the first unit that passes both; if none finer than the default does, the first launch's record stands.
// Flag and acceptance depend on integer sums and on the set of contributing samples only: every launch form gives the same bytes.  The
// default launch's loop is untouched (the flag is one compare in its epilogue).
// Every sweep case past 6e-6 had six or seven scales per octave -- the reference sizes descriptor windows with a literal 3 scales per octave
// (SIFTOctave.swift:398), so there hw reaches 24 and the default unit is 2^-22 -- and none with up to five (hw <= 17.1): the host launches
// the second pass (and passes `desc_flag`) only for schedules whose windows can reach hw >= 17 (siftmi_create: desc_refine); with the
// reference's own schedule nothing is flagged and nothing is launched.
template <bool COOP, int WPB = 4, bool PATCH = false, bool REFINE = false>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(PATCH ? 5 : (REFINE ? 4 : SIFTMI_DESC_WAVES + (WPB == 1 ? 1 : 0))))) void descriptor_kernel(PyramidDesc P, DetectParams prm,
                                                        const KeypointRec *__restrict__ kps, const DescInput *__restrict__ desc_in,
                                                        const int32_t *__restrict__ desc_count, DescriptorRec *__restrict__ desc_out,
                                                        float *__restrict__ desc_f32 /* may be null */, int32_t *__restrict__ desc_flag) {
    static_assert(!REFINE || (!COOP && !PATCH), "the second pass is the plain one-wavefront-per-descriptor form");
    // NCOPY private copies of the histogram per wave (copy = lane % NCOPY): neighbouring lanes take neighbouring samples, which mostly
    // fall into the same cell and bin, and same-address lanes of one LDS add serialise.  The copies of a slot are INTERLEAVED (u32 index
    // = slot * NCOPY + copy, round 3): four neighbouring lanes that hit the same bin then touch four neighbouring banks.
    constexpr int NCOPY = SIFTMI_DESC_NCOPY;
#if SIFTMI_DESC_PACK2
    // A slot is a PAIR of u32 -- {contributions to bin fb, contributions to bin fb + 1 from samples whose lower bin is fb} -- so that a corner
    // is ONE 64-bit LDS add (lo = w va, hi = w vb; neither half can carry: the bound above), four per sample instead of eight; bin b =
    // lo[b] + hi[b - 1].  Identical records (integer sums).  With TWO copies the 16 u64 banks of an LDS add hold (bin, copy) one to one --
    // lanes of one cell with different orientation bins are what a wavefront mostly adds at once --: 3.95 against 4.07 ms per 1.13 M
    // descriptors for the u32 form with four copies; four copies of pairs alias bin b with b + 4: 4.5 ms (profiles/desc_variants_r06.log).
    constexpr bool PACK2 = true;
    constexpr int NSLOT = 8, SLOTW = 2;
#else
    constexpr bool PACK2 = false;
    constexpr int NSLOT = 9, SLOTW = 1;                    // orientation slots per cell: bins 0 ... 7 and slot 8 = bin 0 again (the upper neighbour of 7)
#endif
    constexpr int HIST = 16 * NSLOT * NCOPY * SLOTW;       // u32 per histogram
    constexpr int MAXCOL = 128;                            // window columns handled by the compacted walk (2 per lane)
    static_assert(WPB == 4 || (!COOP && WPB == 1), "COOP shares one descriptor among the four wavefronts of a workgroup");
    // One LDS block: per wavefront {weight table, walk tables} first, the histograms after them.  The address of a contribution is that
    // of cell (fx, fy) with fx, fy >= -1 (a corner at -1 is reached through its neighbour at 0 by an immediate offset), i.e. up to 5 NSLOT
    // slots below its histogram: the tables in front keep that address non-negative for the first histogram too.
    constexpr int QCAP = PATCH ? 0 : SIFTMI_DESC_QCAP;                   // quads the walk's quad table holds (a window of the reference's schedule has <= 790)
    constexpr int TAB_BYTES = ((MAXCOL + 4) * 4 + QCAP * 2 + (MAXCOL + 2) * 2 + MAXCOL + (PATCH ? MAXCOL * 2 : 0) + 15) & ~15;   // gtab, qtab, col_start, col_lo (i8), col_len
    constexpr int PAD_NEED = 5 * NSLOT * NCOPY * 4 * SLOTW;
    constexpr int PAD_BYTES = WPB * TAB_BYTES >= PAD_NEED ? 0 : PAD_NEED - WPB * TAB_BYTES;   // (the tables usually are the pad)
    static_assert(TAB_BYTES % 16 == 0 && PAD_BYTES % 16 == 0 && HIST % 4 == 0, "16-byte aligned histograms behind the tables (cleared by 16-byte stores)");
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[WPB * TAB_BYTES + PAD_BYTES + WPB * HIST * 4];
    constexpr int TP = 20;                                 // floats per staged tile row (18 used; 80 B keeps a quad's row segment 16-byte aligned)
    static_assert(!PATCH || !COOP, "PATCH: one wavefront per descriptor");
    __shared__ __attribute__((aligned(16))) float tile_all[PATCH ? WPB : 1][PATCH ? 18 * TP : 4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int hw_ = COOP ? 0 : wv;                          // whose histogram copies: the workgroup's (COOP) or this wave's
    constexpr int STRIDE = COOP ? 256 : 64;                 // lanes walking one descriptor's samples
    const int lidx = COOP ? (int)threadIdx.x : lane;
    unsigned *hist0 = reinterpret_cast<unsigned *>(lds_raw + WPB * TAB_BYTES + PAD_BYTES) + hw_ * HIST;
    // byte address (LDS) of this lane's copy of slot 0, as a float: a contribution's address is formed in float (exact: < 2^24)
    // Which quads a lane takes and which copy it adds to (round 6; the adds' LDS conflicts were 1.0 of the loop's 4.6 ms once the vector
    // work had shrunk: profiles/desc_variants_r06.log).  An LDS add is served in two groups, lanes 0-31 and 32-63, and lanes of a group
    // that hit one bank (or worse one address) take turns.  A trip's 64 quads are ~8 window rows of ~8 quads; with quad = lane, a group
    // held four ADJACENT rows -- the same cells, mostly the same orientation bin -- and rows 8 quads long gave vertical neighbours the same
    // copy (lane & 3).  Now the lanes of a group take alternate blocks of 8 quads (position p = lane with bits 3 and 5 swapped: rows
    // 0, 2, 4, 6 against 1, 3, 5, 7), and the copy is (p & 1) -- horizontal neighbours -- with the block's index above it.
    const int qpos = (lane & ~0x28) | ((lane & 8) << 2) | ((lane & 32) >> 2);
    static_assert(NCOPY == 2 || NCOPY == 4 || NCOPY == 8, "copy = one column bit + block bits");
#ifdef SIFTMI_DESC_COPY_BLOCK_FIRST
    const int copy = ((qpos >> 4) & 1) | ((qpos & (NCOPY / 2 - 1)) << 1);     // experiment: block bit first
#else
    const int copy = (qpos & 1) | (((qpos >> 4) & (NCOPY / 2 - 1)) << 1);
#endif
    const float hist_base_f = (float)((unsigned)(size_t)(lds_u32_t *)hist0 + 4u * SLOTW * (unsigned)copy);
    unsigned char *tab = lds_raw + wv * TAB_BYTES;
    float *gtab = reinterpret_cast<float *>(tab);                                        // exp(-k^2 / 8 hw^2) 2^-63 (2^-half_shift folded in), k = -radius ... radius + 3
    unsigned short *qtab = reinterpret_cast<unsigned short *>(tab + (MAXCOL + 4) * 4);   // quad q of the walk: row | (j0 + radius) << 7
    short *col_start = reinterpret_cast<short *>(tab + (MAXCOL + 4) * 4 + QCAP * 2);     // walk index of a row's first candidate (<= 128 * 128 / 1: fits 15 bits)
    signed char *col_lo = reinterpret_cast<signed char *>(col_start + MAXCOL + 2);       // first candidate column of a row (|.| <= 63)
    short *col_len = reinterpret_cast<short *>(col_lo + MAXCOL);                         // (PATCH only)
    const int group = group_index(P, blockIdx.y), frame = group / P.n_octaves, o = group - frame * P.n_octaves;
    const int n = min(desc_count[group], P.cap_desc[o]);
    const int w = P.w[o], h = P.h[o];
    const size_t dbase = (size_t)frame * P.desc_frame + P.desc_off[o];
    // REFINE: a wavefront reads the flags of 64 descriptors at a time (one coalesced load and a ballot, not one scalar load per descriptor:
    // the flagged ones are a handful per group) and walks the flagged ones
    const int di_step = COOP ? (int)gridDim.x : (int)(gridDim.x * WPB);
    int di = (COOP ? (int)blockIdx.x : (int)(blockIdx.x * WPB + wv)) - di_step;
    int chunk = (int)(blockIdx.x * WPB + wv) * 64 - di_step * 64;
    unsigned long long pending = 0ull;
    for (;;) {
        if constexpr (REFINE) {
            while (pending == 0ull) {
                chunk += di_step * 64;
                if (chunk >= n) return;                                      // (wave-uniform; no workgroup barriers in this form)
                pending = __ballot(chunk + lane < n && desc_flag[dbase + chunk + lane] != 0);
            }
            di = chunk + __builtin_ctzll(pending);
            pending &= pending - 1ull;
        } else {
            di += di_step;
            if (di >= n) break;
        }
        const DescInput in = desc_in[dbase + di];                          // wave-uniform: scalar loads
        const float theta = in.theta;
        const LayerView g = layer_view(layer_ptr(P, frame, o, in.scale), w, h);
        const float px = in.px, py = in.py;
        const float cosT = in.cosT, sinT = in.sinT;
        const float histogramWidth = in.hw;
        const int radius = in.radius;
        // per descriptor, hoisted out of the sample loop (float note above): the rotation with 1 / histogramWidth folded in
        const float cs = cosT * in.inv_hw, sn = sinT * in.inv_hw;
        // px, py are multiples of 2^-15 well below 2^22, so px + j is exact and ushort2(px + j, ...) truncates to (int)px + j
        // wherever px + j >= 0: integer sample coordinates, no conversions in the loop
        const int ipx = (int)px, ipy = (int)py;
        // the whole window and the +-1 neighbours of its samples lie inside the image: no mirror, no range test per sample
        const bool interior = ipx - radius >= 1 && ipx + radius <= w - 2 && ipy - radius >= 1 && ipy + radius <= h - 2;
        const int side = 2 * radius + 1;
        const bool compact = side <= MAXCOL;
        // Gaussian weight of a sample = gauss(j) gauss(i) with gauss(k) = exp2(kg k^2) 2^-63 2^-(unit_shift / 2): the product carries 2^-126
        // (and 2^-unit_shift); kg = -log2(e) / (8 hw^2).  unit_shift (even): 0 while a bin's bound (hw + 1)^2 sqrt(2) / 2 stays below 2^8.
        const float kg = (in.inv_hw * in.inv_hw) * (-0.125f * 1.44269504088896341f);
        const float bound = (histogramWidth + 1.0f) * (histogramWidth + 1.0f) * 0.70710678f;
        int half_shift = 0;
        while (ldexpf(bound, -2 * half_shift) >= 240.0f) half_shift++;       // (wave-uniform; no trip with the reference's schedule)
        // REFINE: the contributions' unit is 2^-24 4^-fine_h instead of 2^-24 4^half_shift (fine_h = 4, 3, ... down to 1 - half_shift)
        int fine_h = REFINE ? 4 : -half_shift;
        float gscale = ldexpf(1.0f, -63 + fine_h);
        auto gauss = [&](int k) -> float { const float f = (float)k; return __builtin_amdgcn_exp2f(kg * (f * f)) * gscale; };
        const float theta_turns = theta * 0.159154943091895336f;              // theta in [0, 2 pi)

        auto clear_bins = [&]() {
            if (COOP) {
                __syncthreads();                                                      // wave 0 is done reading the previous descriptor's bins
                for (int c = threadIdx.x; c < HIST / 4; c += 256) reinterpret_cast<uint4 *>(hist0)[c] = make_uint4(0u, 0u, 0u, 0u);
            } else {
                for (int c = lane; c < HIST / 4; c += 64) reinterpret_cast<uint4 *>(hist0)[c] = make_uint4(0u, 0u, 0u, 0u);   // all copies (contiguous)
            }
        };
        auto build_gtab = [&]() { if (compact) for (int k = lane; k < side + 3; k += 64) gtab[k] = gauss(k - radius); };
        clear_bins();
        build_gtab();

        // The reference visits every (j, i) of the (2 radius + 1)^2 window (metal :194-195), but a sample adds
        // something only if its cell coordinates fall inside (-1, 4)^2, i.e. inside a rotated square of half
        // width 2.5 histogramWidth -- about half the window.  Per window ROW i (y offset) the j's (x offsets) that can
        // qualify form one interval; compute a conservative interval per row (+-2 px of float slack), prefix-sum the
        // lengths and walk the compacted index space, so that all 64 lanes hold candidate samples.  The exact
        // per-sample test below is unchanged, hence exactly the same samples contribute; the bins are order-free
        // (integer sums), so walking row-major instead of the reference's column-major changes nothing -- but
        // neighbouring lanes then read neighbouring pixels: the four gradient loads of a wavefront touch 3-4 cache
        // lines instead of 64 each (round 1 walked columns: every lane its own line, the texture path was the limit).
        // Round 5: an INTERIOR window is walked in QUADS -- four consecutive candidates of a row per lane and trip.  The walk's table
        // look-up and the loads' address arithmetic are then paid once per four samples, the row's texels arrive as four wide loads
        // instead of sixteen dword loads, and the terms of the rotated coordinates that depend on the row alone are shared.
        const int unit = (compact && interior) ? 4 : 1;                   // candidates per walk index
        int total;
        if (compact) {
            const float Lh = 2.5f * histogramWidth;
            const float inv_cos = in.inv_cos, inv_sin = in.inv_sin;        // (unused where the component is ~0)
            int run = 0;
            for (int c0 = 0; c0 < side; c0 += 64) {
                const int cidx = c0 + lane;
                int lo = 1, hi = 0;
                if (cidx < side) {
                    const float yf = (float)(cidx - radius);
                    // |j*cosT - yf*sinT| < Lh  and  |j*sinT + yf*cosT| < Lh
                    float a0 = -(float)radius, a1 = (float)radius;
                    if (fabsf(cosT) > 1e-6f) {
                        const float u = (yf * sinT - Lh) * inv_cos, v = (yf * sinT + Lh) * inv_cos;
                        a0 = fmaxf(a0, fminf(u, v)); a1 = fminf(a1, fmaxf(u, v));
                    } else if (fabsf(yf * sinT) >= Lh + 1.0f) { a1 = a0 - 1.0f; }
                    if (fabsf(sinT) > 1e-6f) {
                        const float u = (-Lh - yf * cosT) * inv_sin, v = (Lh - yf * cosT) * inv_sin;
                        a0 = fmaxf(a0, fminf(u, v)); a1 = fminf(a1, fmaxf(u, v));
                    } else if (fabsf(yf * cosT) >= Lh + 1.0f) { a1 = a0 - 1.0f; }
                    // a sample qualifies only for a0 < j < a1 (strict), so [floor(a0), ceil(a1)] already holds one column more
                    // than can qualify on either side -- four orders above the float error of a0 / a1.  (Rounds 1-2 added two
                    // more per side: 8 % of the walked candidates never passed the exact test below.)
                    lo = max(-radius, (int)floorf(a0));
                    hi = min(radius, (int)ceilf(a1));
                }
                const int len = max(hi - lo + 1, 0);
                const int nun = unit == 4 ? (len + 3) >> 2 : len;          // walk indices of this row
                const int incl = wave_inclusive_scan(nun);
                const int start = run + incl - nun;
                if (cidx < side) { col_start[cidx] = (short)start; col_lo[cidx] = (signed char)lo; if (PATCH) col_len[cidx] = (short)len; }
                // the quads of an interior window, one entry each: the walk then finds a quad's row and first column by ONE read (round 5
                // advanced through col_start row by row: ~8 dependent LDS reads and ~40 vector instructions per trip of 64 quads)
                if (QCAP > 0 && unit == 4 && cidx < side)
                    for (int k4 = 0; k4 < nun && start + k4 < QCAP; k4++) qtab[start + k4] = (unsigned short)(cidx | ((lo + radius + 4 * k4) << 7));
                run += __builtin_amdgcn_readlane(incl, 63);
            }
            total = run;
            if (lane == 0) col_start[side] = (short)total;
        } else {
            total = side * side;
        }
        if (COOP) __syncthreads(); else __builtin_amdgcn_wave_barrier();   // bins cleared (COOP: by all four waves); tables of this wave written
        __threadfence_block();

        // One sample (x offset fj, y offset i) of the window: SIFTDescriptor.metal:197-222.  gj, gi: gauss(j), gauss(i) (gj = 0 for a
        // candidate that is not a window sample).  INTERIOR (wave-uniform, almost every descriptor): the sample and its four neighbours
        // are inside the image, so the texels come from the walk (wide loads) and there is no per-sample range test or mirror.
#define DESC_HADD(a, off, v) lds_add_bits((a), (off), (v))
        float m2max = 0.0f;                                                 // REFINE: the largest (2 |gradient|)^2 among the contributing samples
        auto sample = [&](auto interior_tag, float fj, int i, float gj, float gi, float t_xp, float t_xm, float t_yp, float t_ym) {
            constexpr bool INTERIOR = decltype(interior_tag)::value;
            const float fi = (float)i;
            // cell coordinates (rx, ry) + d / 2 - 0.5 with (rx, ry) = the offset rotated by -theta, / histogramWidth; the reference drops a
            // sample unless both lie in (-1, 4) (addValue :66-68): the cell tests below
            const float bx = fmaf(fj, cs, fmaf(fi, -sn, 1.5f));
            const float by = fmaf(fj, sn, fmaf(fi, cs, 1.5f));
            float dx, dy;                                                  // central differences, not yet halved
            if (INTERIOR) {
                dx = t_xp - t_xm;
                dy = t_yp - t_ym;
            } else {
                // ushort2(px + j, py + i): truncation toward zero, (-1, 0) -> 0; negative: no texel
                const float fx = px + fj, fy = py + fi;
                float tx = 0.0f, ty = 0.0f;
                const int gx = (int)fx, gy = (int)fy;
                if (fx > -1.0f && fy > -1.0f && gx < w && gy < h) {        // (int) of a huge float is >= w: outside -> gradient (0, 0)
                    const int mxm = symm(gx - 1, w), mxp = symm(gx + 1, w), mym = symm(gy - 1, h), myp = symm(gy + 1, h);
                    auto rd = [&](int x, int y) -> float { return (x < 0 || y < 0 || x >= w || y >= h) ? 0.0f : g.g[(size_t)y * w + x]; };
                    tx = rd(mxp, gy) - rd(mxm, gy);
                    ty = rd(gx, myp) - rd(gx, mym);
                }
                dx = tx; dy = ty;
            }
            // gradient (tx, ty) = (dx, dy) / 2 (SIFTGradient.metal:31-32): its angle atan2(tx, ty) does not depend on the factor 1/2;
            // orientation = angle - theta wrapped into [0, 2 pi), bin = 8 orientation / 2 pi (SIFTDescriptor.metal:203-213): in turns,
            // the wrap is v_fract (which stays below 1: bin < 8)
            const float m2 = fmaf(dx, dx, fmaf(dy, dy, 1.0e-30f));
            const float mag = __builtin_amdgcn_sqrtf(m2);                                         // 2 |gradient|, > 0 (angle_turns)
            const float bin = __builtin_amdgcn_fractf(angle_turns(dx, dy, mag) - theta_turns) * 8.0f;
            // value = |gradient| exp(-(rx^2 + ry^2) / 8) 2^-63 (gj carries the table's 2^-63, the row's entry is taken without it): a NORMAL float
            // with all 24 bits, like the shares va / vb below; the other 2^-63 rides on the y weights, so only the eight final products land in
            // the denormal / first normal binade (at most sqrt(2) 2^-126: bits linear in the value, units of 2^-24 of the reference's scale) --
            // one rounding per contribution.  (The first round-6 form carried 2^-126 on `v` itself: v, its two shares and the product were each
            // rounded to the unit, 1.65e-5 instead of 6.8e-7 L2 on a low-contrast descriptor of the sweep: profiles/sweep_cases_r06.log.)
            const float v = (mag * gj) * (gi * 0x1p63f);
            {   // addFeature :82-117.  The reference calls addValue for the 8 trilinear corners, each with its own range test
                // and bin wrap (:59-79); here one test per cell corner, and the upper orientation bin is the next slot.
                // The "upper" corner is floor + 1 here, not ceil: they differ only when the coordinate is an integer, and then
                // the upper corner's weight is exactly 0 -- it adds 0 whichever cell or bin it names.
                // x - floor(x) is exact (no rounding) for the finite values here.
                const float fx = __builtin_floorf(bx), fy = __builtin_floorf(by), fb = __builtin_floorf(bin);
                const float iMax = bx - fx, iMin = 1.0f - iMax;
                const float jMax = (by - fy) * 0x1p-63f, jMin = 0x1p-63f - jMax;   // the y weights, x 2^-63 (exact: powers of two)
                // cell c in {0, 1, 2, 3} <=> bits(c) <= bits(3.0f): negative floors have the sign bit set, larger ones larger bits (the
                // floor of a value in [0, 1) is +0)
                const bool xa = __float_as_uint(fx) <= 0x40400000u, xb = __float_as_uint(fx + 1.0f) <= 0x40400000u;
                const bool ya = __float_as_uint(fy) <= 0x40400000u, yb = __float_as_uint(fy + 1.0f) <= 0x40400000u;
                if constexpr (REFINE) { if ((xa || xb) && (ya || yb)) m2max = fmaxf(m2max, m2); }
                // the value's share of either orientation bin.  Every factor below is >= +0, so is every product: a contribution's sign
                // bit is never set (vb <= v as a rounded product of v and a factor < 1, so v - vb >= +0 too)
                const float vb = (bin - fb) * v, va = v - vb;
                // byte address of slot ((fy 4 + fx) NSLOT + fb), this lane's copy: exact in float (garbage where no test passes)
                const float slot = fmaf(fy, (float)(4 * NSLOT), fmaf(fx, (float)NSLOT, fb));
                unsigned a;
                asm("v_cvt_u32_f32 %0, %1" : "=v"(a) : "v"(fmaf(slot, (float)(4 * NCOPY * SLOTW), hist_base_f)));
                auto corner = [&](float wxy, int cell_off) {                 // cell_off: slots from cell (fx, fy) to the corner's cell
                    if constexpr (PACK2) {
                        typedef __attribute__((address_space(3))) unsigned long long lds_u64_t;
                        const unsigned long long pair = ((unsigned long long)__float_as_uint(wxy * vb) << 32) | __float_as_uint(wxy * va);
                        __hip_atomic_fetch_add((lds_u64_t *)(size_t)a + cell_off * NCOPY, pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
                        DESC_HADD(a, cell_off * NCOPY, wxy * va); DESC_HADD(a, (cell_off + 1) * NCOPY, wxy * vb);
                    }
                };
                if (xa && ya) corner(iMin * jMin, 0);
                if (xb && ya) corner(iMax * jMin, NSLOT);
                if (xb && yb) corner(iMax * jMax, 5 * NSLOT);
                if (xa && yb) corner(iMin * jMax, 4 * NSLOT);
            }
        };
        auto walk = [&](auto interior_tag) {
            constexpr bool INTERIOR = decltype(interior_tag)::value;
            if constexpr (PATCH && INTERIOR) if (compact) {   // (every lane takes part: wave-level loops)
                float *tile = tile_all[wv];
                const int nt = (side + 15) >> 4;                        // tiles per side of the bounding box [-radius, radius]^2
                for (int ty = 0; ty < nt; ty++) {
                    // the columns the 16 window rows of this tile row can use (wave-uniform; lanes 16 ... 63 repeat lanes 0 ... 15)
                    const int trow = ty * 16 + (lane & 15);
                    int lo_t = 30000, hi_t = -30000;
                    if (trow < side && col_len[trow] > 0) { lo_t = col_lo[trow]; hi_t = lo_t + col_len[trow] - 1; }
#pragma unroll
                    for (int off = 1; off < 16; off <<= 1) { lo_t = min(lo_t, __shfl_xor(lo_t, off)); hi_t = max(hi_t, __shfl_xor(hi_t, off)); }
                    lo_t = __builtin_amdgcn_readfirstlane(lo_t); hi_t = __builtin_amdgcn_readfirstlane(hi_t);
                    if (hi_t < lo_t) continue;
                    const int i0 = ty * 16 - radius;                    // window row (y offset) of the tile's first sample row
                    for (int tx = (lo_t + radius) >> 4; tx <= (hi_t + radius) >> 4; tx++) {
                        const int j0 = tx * 16 - radius;                // x offset of the tile's first sample column
                        // stage texels (ipx + j0 - 1 + c, ipy + i0 - 1 + r), r, c in [0, 18): the samples and their +-1 neighbours.  A tile
                        // may reach past the window (the bounding box is rounded up to 16): such texels come from the next image row, or
                        // read 0 past the layer's end (range-checked buffer), and belong to samples outside every row interval.
                        __builtin_amdgcn_wave_barrier();                // the previous tile's reads are done (one wavefront: LDS in order)
                        for (int k = lane; k < 18 * 18; k += 64) {
                            const int r = k / 18, cc = k - r * 18;
                            tile[r * TP + cc] = layer_ld(g, __mul24(ipy + i0 - 1 + r, g.pitch) + ((ipx + j0 - 1 + cc) << 2));
                        }
                        __builtin_amdgcn_wave_barrier();
                        __threadfence_block();
                        const int r = lane >> 2, q4 = (lane & 3) << 2;  // this lane: sample row r of the tile, columns q4 ... q4 + 3
                        const int srow = ty * 16 + r;
                        int lo = 0, len = 0;
                        float gi = 0.0f;
                        if (srow < side) { lo = col_lo[srow]; len = col_len[srow]; gi = gtab[srow]; }
                        const float *tc = tile + (r + 1) * TP + q4;     // texel x - 1 of the quad's first sample, its own row
                        float A[6], U[4], D[4];
#pragma unroll
                        for (int e = 0; e < 6; e++) A[e] = tc[e];
#pragma unroll
                        for (int e = 0; e < 4; e++) { U[e] = tc[e + 1 - TP]; D[e] = tc[e + 1 + TP]; }
#pragma unroll
                        for (int s4 = 0; s4 < 4; s4++) {
                            const int j = j0 + q4 + s4;
                            const bool valid = j >= lo && j < lo + len;       // (inside a row interval => a window sample: the table holds it)
                            sample(interior_tag, valid ? (float)j : 30000.0f, i0 + r, valid ? gtab[j + radius] : 0.0f, gi, A[s4 + 2], A[s4], D[s4], U[s4]);
                        }
                    }
                }
                return;
            }
            int cur = 0;                                                   // window row of this lane's current sample
            const bool by_table = INTERIOR && QCAP > 0 && compact && total <= QCAP;   // the quad table gives the row: nothing to search
            if (compact && lidx < total && !by_table) {                    // binary search once, then only advance
                int lo_c = 0, hi_c = side - 1;                             // last row whose start <= lidx
                while (lo_c < hi_c) { const int mid = (lo_c + hi_c + 1) >> 1; if (col_start[mid] <= lidx) lo_c = mid; else hi_c = mid - 1; }
                cur = lo_c;
            }
            int cur_start = (compact && !by_table) ? col_start[cur] : 0, next_start = (compact && !by_table) ? col_start[cur + 1] : 0;
            auto locate = [&](int idx, int &j, int &i) {                   // j: x offset (inner), i: y offset (outer); idx never decreases
                if (compact) {
                    while (idx >= next_start) { cur++; cur_start = next_start; next_start = col_start[cur + 1]; }   // empty rows have equal starts
                    i = cur - radius;
                    j = (int)col_lo[cur] + (idx - cur_start);
                } else {
                    const int ii = idx / side;
                    i = ii - radius; j = idx - ii * side - radius;
                }
            };
            auto gauss_at = [&](int k) -> float { return compact ? gtab[k + radius] : gauss(k); };   // (the table holds gauss(k): same bits)
            if constexpr (INTERIOR) {
                if (compact) {
                    // quads: row y holds texels x - 1 ... x + 4 of the quad's four samples x ... x + 3 (b128 + b64), rows y - 1 and y + 1
                    // texels x ... x + 3 (b128 each).  The last quad of a row runs up to three candidates past its interval: their texels
                    // come from the image row, the next row or the next layer (a described layer is never the stack's last: scale <= nspo),
                    // or read 0 past the allocation (range-checked buffer); the cell tests reject them (header).
                    // No software pipelining here: a trip is ~300 vector instructions, the other wavefronts of the SIMD cover its one
                    // memory latency, and the 14 registers a prefetched quad would hold cost a resident wavefront
                    // (measured equal or 1-2 % behind with the prefetch, profiles/desc_variants_r05.log).
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    auto quad = [&](int row, int jidx) {                      // the four candidates jidx - radius ... + 3 of window row `row`
                        const int i = row - radius, j0 = jidx - radius;
                        const int c = __mul24(ipy + i - 1, g.pitch) + ((ipx + j0 - 1) << 2);    // texel (x - 1, y - 1); both factors < 2^24
                        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(g.rsrc, c, g.pitch, 0);
                        const u32x2 a2 = __builtin_amdgcn_raw_buffer_load_b64(g.rsrc, c + 16, g.pitch, 0);
                        const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(g.rsrc, c + 4, 0, 0);
                        const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(g.rsrc, c + 4, 2 * g.pitch, 0);
                        const float gi = gtab[row];
                        const float *gq = gtab + jidx;
                        const float G[4] = {gq[0], gq[1], gq[2], gq[3]};
                        const float fj0 = (float)j0;
                        const float A[6] = {__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w),
                                            __uint_as_float(a2.x), __uint_as_float(a2.y)};
                        const float U[4] = {__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)};
                        const float D[4] = {__uint_as_float(d.x), __uint_as_float(d.y), __uint_as_float(d.z), __uint_as_float(d.w)};
#pragma unroll
                        for (int s4 = 0; s4 < 4; s4++)
                            sample(interior_tag, fj0 + (float)s4, i, G[s4], gi, A[s4 + 2], A[s4], D[s4], U[s4]);
                    };
                    if (by_table) {
                        for (int q = COOP ? wv * 64 + qpos : qpos; q < total; q += STRIDE) { const int e = qtab[q]; quad(e & 127, e >> 7); }
                    } else {
                        for (int q = lidx; q < total; q += STRIDE) {
                            while (q >= next_start) { cur++; cur_start = next_start; next_start = col_start[cur + 1]; }
                            quad(cur, (int)col_lo[cur] + ((q - cur_start) << 2) + radius);
                        }
                    }
                    return;
                }
                // (windows wider than the walk's table -- never with the default schedule): one sample per trip, its texels requested a trip ahead
                auto fetch = [&](int j, int i, float &t_xp, float &t_xm, float &t_yp, float &t_ym) {
                    const int c = __mul24(ipy + i - 1, g.pitch) + ((ipx + j - 1) << 2);    // texel (x - 1, y - 1); both factors < 2^24
                    t_xp = layer_ld_s(g, c + 8, g.pitch); t_xm = layer_ld_s(g, c, g.pitch);
                    t_yp = layer_ld_s(g, c + 4, 2 * g.pitch); t_ym = layer_ld(g, c + 4);
                };
                if (lidx >= total) return;
                int j, i;
                float a0, a1, a2, a3;
                locate(lidx, j, i);
                fetch(j, i, a0, a1, a2, a3);
                for (int idx = lidx; idx < total; idx += STRIDE) {
                    int jn, in_;
                    float b0, b1, b2, b3;
                    locate(min(idx + STRIDE, total - 1), jn, in_);
                    fetch(jn, in_, b0, b1, b2, b3);
                    sample(interior_tag, (float)j, i, gauss_at(j), gauss_at(i), a0, a1, a2, a3);
                    j = jn; i = in_; a0 = b0; a1 = b1; a2 = b2; a3 = b3;
                }
            } else {
                for (int idx = lidx; idx < total; idx += STRIDE) {
                    int j, i;
                    locate(idx, j, i);
                    sample(interior_tag, (float)j, i, gauss_at(j), gauss_at(i), 0.0f, 0.0f, 0.0f, 0.0f);
                }
            }
        };
        bool refined = false;
        for (;;) {
            if (interior) walk(std::true_type{}); else walk(std::false_type{});
            if (COOP) __syncthreads(); else __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            if (!REFINE) break;
            // was every contribution inside the linear range at this unit, 2 |gradient| 4^fine_h < 2, and is every bin's sum below 2^32 units,
            // (hw + 1)^2 max |gradient| 4^fine_h 2^24 < 2^32 (the bound of the header, per unit of gradient)?  (with a margin for the roundings)
            const float q = 512.0f / ((histogramWidth + 1.0f) * (histogramWidth + 1.0f));
            refined = wave_max_nonneg(m2max) * ldexpf(1.0f, 4 * fine_h) < 0.8f * fminf(4.0f, q * q);
            if (refined || --fine_h == -half_shift) break;
            m2max = 0.0f;                                                        // no: the same walk at the next coarser unit
            gscale = ldexpf(1.0f, -63 + fine_h);
            clear_bins();
            build_gtab();
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
        }
#undef DESC_HADD
        if (REFINE && !refined) continue;                                        // the first launch's record stands
        if (COOP && wv != 0) continue;                                           // COOP: wave 0 finishes the descriptor
        // features lane and lane + 64 (feature = cell * 8 + bin): the NCOPY copies of their slot, and for bin 0 those of slot 8.
        // Sums stay below 2^32 (header).
        unsigned a0 = 0u, a1 = 0u;
        {
            const int cell = lane >> 3, b = lane & 7;
            if constexpr (PACK2) {
                const unsigned *s0 = hist0 + (cell * NSLOT + b) * NCOPY * 2, *s1 = s0 + 8 * NSLOT * NCOPY * 2;
                const int prev = (((b + 7) & 7) - b) * NCOPY * 2;                   // slot of bin b - 1 (same cell): its high halves belong to bin b
#pragma unroll
                for (int c = 0; c < NCOPY; c++) { a0 += s0[2 * c] + s0[prev + 2 * c + 1]; a1 += s1[2 * c] + s1[prev + 2 * c + 1]; }
            } else {
            const unsigned *s0 = hist0 + (cell * NSLOT + b) * NCOPY, *s1 = s0 + 8 * NSLOT * NCOPY;
#pragma unroll
            for (int c = 0; c < NCOPY; c++) { a0 += s0[c]; a1 += s1[c]; }
            if (b == 0) {
#pragma unroll
                for (int c = 0; c < NCOPY; c++) { a0 += s0[8 * NCOPY + c]; a1 += s1[8 * NCOPY + c]; }
            }
            }
        }
        const float unit_scale = ldexpf(1.0f, -24 - 2 * fine_h);           // back to the reference's units (the normalisation removes it again)
        float f0 = (float)a0 * unit_scale, f1 = (float)a1 * unit_scale;
        {   // normalise -> clamp 0.2 -> normalise (:15-39, :224-227)
            const float ss = wave_sum(f0 * f0 + f1 * f1);
            // norm below 2^23 units of 2^-24 (the roundings' ~45 units are then more than ~5e-6 of it): for the second launch (header)
            if (!REFINE && desc_flag != nullptr && lane == 0) desc_flag[dbase + di] = (ss > 0.0f && ss < SIFTMI_DESC_FLAG_SS * unit_scale * unit_scale * 0x1p48f) ? 1 : 0;
            float dn = 1.0f / sqrtf(ss);
            f0 *= dn; f1 *= dn;
            f0 = fminf(f0, 0.2f); f1 = fminf(f1, 0.2f);
            dn = 1.0f / sqrtf(wave_sum(f0 * f0 + f1 * f1));
            f0 *= dn; f1 *= dn;
        }
        DescriptorRec *out = desc_out + dbase + di;
        out->features[lane] = (uint8_t)(int)fminf(255.0f, f0 * 512.0f);          // :42-50 truncation
        out->features[lane + 64] = (uint8_t)(int)fminf(255.0f, f1 * 512.0f);
        if (desc_f32) {
            desc_f32[(dbase + di) * DESC_N + lane] = f0;
            desc_f32[(dbase + di) * DESC_N + lane + 64] = f1;
        }
        if (lane == 0) { out->keypoint = in.keypoint; out->theta = theta; }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------------
// Packing: dense (frame, octave)-ordered output records + counts.  group_offsets_kernel is one
// small workgroup; pack_kernel copies dwords.
struct PackState {                 // device-resident running totals of a batch call
    int32_t total_kp, total_desc, overflow_flags, pad;
};

__global__ __launch_bounds__(256) void group_offsets_kernel(PyramidDesc P, int n_frames, int frame_base, int total_frames,
                                                           const int32_t *__restrict__ raw_count, const int32_t *__restrict__ cand_count,
                                                           const int32_t *__restrict__ kp_count, const int32_t *__restrict__ oriented_count,
                                                           const int32_t *__restrict__ desc_count,
                                                           int32_t *__restrict__ kp_dst_off, int32_t *__restrict__ desc_dst_off,
                                                           int32_t *__restrict__ out_counts /* [2][total_frames][n_oct] */,
                                                           int32_t *__restrict__ stats /* [5][total_frames][n_oct] */,
                                                           PackState *__restrict__ state, long long kp_capacity, long long desc_capacity,
                                                           int32_t *__restrict__ totals_out /* the caller's {n_kp, n_desc, flags, 0} after the last sub-batch, or null */) {
    // One 256-thread workgroup; the groups are scanned 256 at a time with the running totals carried in LDS.  The clamp of
    // a group against the OUTPUT capacity depends only on the unclamped total of the groups before it (T): destination
    // offset min(T, capacity), length clamp(capacity - T, 0, n) -- what a sequential walk with a clamped running total gives.
    if (blockIdx.x != 0) return;
    __shared__ long long w_k[4], w_d[4], carry_k, carry_d;
    __shared__ int s_flags;
    const int no = P.n_octaves, ng = n_frames * no;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) { carry_k = state->total_kp; carry_d = state->total_desc; s_flags = state->overflow_flags; }
    __syncthreads();
    for (int g0 = 0; g0 < ng; g0 += 256) {
        const int g = g0 + threadIdx.x;
        int nk = 0, nd = 0, flags = 0, o = 0, f = 0;
        if (g < ng) {
            f = g / no; o = g - f * no;
            nk = kp_count[g]; nd = desc_count[g];
            if (cand_count[g] > P.cap_ext[o]) flags |= 1;
            if (nk > P.cap_kp[o]) { flags |= 2; nk = P.cap_kp[o]; }
            if (nd > P.cap_desc[o]) { flags |= 4; nd = P.cap_desc[o]; }
        }
        long long ik = nk, id = nd;                           // inclusive wave scans
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const long long tk = __shfl_up(ik, d, 64), td = __shfl_up(id, d, 64);
            if (lane >= d) { ik += tk; id += td; }
        }
        if (lane == 63) { w_k[wv] = ik; w_d[wv] = id; }
        __syncthreads();
        long long Tk = carry_k + ik - nk, Td = carry_d + id - nd;
        for (int k = 0; k < wv; k++) { Tk += w_k[k]; Td += w_d[k]; }
        if (g < ng) {
            if (Tk + nk > kp_capacity) { flags |= 8; nk = (int)max(0ll, min((long long)nk, kp_capacity - Tk)); }
            if (Td + nd > desc_capacity) { flags |= 16; nd = (int)max(0ll, min((long long)nd, desc_capacity - Td)); }
            kp_dst_off[g] = (int)min(Tk, kp_capacity); desc_dst_off[g] = (int)min(Td, desc_capacity);
            const int gi = (frame_base + f) * no + o, stride = total_frames * no;
            out_counts[gi] = nk; out_counts[stride + gi] = nd;
            stats[0 * stride + gi] = raw_count[g];
            stats[1 * stride + gi] = cand_count[g];
            stats[2 * stride + gi] = kp_count[g];
            stats[3 * stride + gi] = oriented_count[g];
            stats[4 * stride + gi] = desc_count[g];
            if (flags) atomicOr(&s_flags, flags);
        }
        __syncthreads();
        if (threadIdx.x == 0) {                               // totals of this trip: all four waves
            long long sk = 0, sd = 0;
            for (int k = 0; k < 4; k++) { sk += w_k[k]; sd += w_d[k]; }
            carry_k += sk; carry_d += sd;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        state->total_kp = (int)min(carry_k, kp_capacity); state->total_desc = (int)min(carry_d, desc_capacity); state->overflow_flags = s_flags;
        if (totals_out) { totals_out[0] = state->total_kp; totals_out[1] = state->total_desc; totals_out[2] = s_flags; totals_out[3] = 0; }
    }
}

__global__ __launch_bounds__(256) void pack_kernel(PyramidDesc P, const KeypointRec *__restrict__ kps, const DescriptorRec *__restrict__ descs,
                                                  const int32_t *__restrict__ kp_dst_off, const int32_t *__restrict__ desc_dst_off,
                                                  const int32_t *__restrict__ out_counts, int frame_base, int total_frames,
                                                  KeypointRec *__restrict__ kp_out, DescriptorRec *__restrict__ desc_out) {
    const int group = blockIdx.y, frame = group / P.n_octaves, o = group - frame * P.n_octaves;
    const int gi = (frame_base + frame) * P.n_octaves + o, stride = total_frames * P.n_octaves;
    const int nk = out_counts[gi], nd = out_counts[stride + gi];
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(kps + (size_t)frame * P.kp_frame + P.kp_off[o]);
        uint32_t *dst = reinterpret_cast<uint32_t *>(kp_out + kp_dst_off[group]);
        const int nw = nk * (int)(sizeof(KeypointRec) / 4);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < nw; i += gridDim.x * 256) dst[i] = src[i];
    }
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(descs + (size_t)frame * P.desc_frame + P.desc_off[o]);
        uint32_t *dst = reinterpret_cast<uint32_t *>(desc_out + desc_dst_off[group]);
        const long long nw = (long long)nd * (long long)(sizeof(DescriptorRec) / 4);
        for (long long i = blockIdx.x * 256 + threadIdx.x; i < nw; i += gridDim.x * 256) dst[i] = src[i];
    }
}

}  // namespace siftmi
