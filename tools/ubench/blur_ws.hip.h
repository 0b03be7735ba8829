// blur_ws.hip.h -- EXPERIMENT (tools/ubench only): the ring blur with the four wavefronts of a workgroup specialised, as the
// round-2 review proposed for the R >= 10 layers.  Waves 0-1 stage and horizontally blur the 16 rows of step k+1 while waves
// 2-3 run the vertical pass and the stores of step k; ONE workgroup barrier per 16-row step instead of three per 32 rows, and
// the LDS-heavy and the FMA-dense halves of a step run side by side inside the workgroup.  Ring of 64 rows = four 16-row
// segments: the vertical pass of step k reads segments k-2 ... k (16 + 2R <= 42 rows), the H waves write segment k+1.
// Restricted to what the 3840 x 2160 harness needs: w a multiple of 128, h and the chunk a multiple of 16, no flags / decimation.
// Same arithmetic and tap order as blur_ring_kernel: bit-identical results (the harness checks).
#pragma once
#include "dense_kernels.hip.h"

namespace siftmi {

template <int R, int ROLEMAP = 0>
__global__ __launch_bounds__(256, 4) void blur_ring_wsx_kernel(const float *__restrict__ src, float *__restrict__ dst, int w, int h,
                                                             size_t src_frame_stride, size_t dst_frame_stride, TapWeights wt, int n_frames,
                                                             int ch_rows) {
    constexpr int S = 16, NR = 64, RP = 16, TW = 128, LW = TW + 2 * RP, NT = 2 * R + 1, RB = 8;
    constexpr int NPF4 = LW / 32;                                    // 8 lanes per row: 5 float4 each
    static_assert(NPF4 * 32 == LW && S + 2 * R + S <= NR && R <= RP && 2 * R + S >= 32, "geometry (R = 8 ... 16)");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tx = w / TW, nch = (h + ch_rows - 1) / ch_rows;
    const int total = tx * nch * n_frames, per_xcd = (total + 7) >> 3;
    const int t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (t >= total) return;
    const int frame = t / (tx * nch), rem = t - frame * (tx * nch);
    const int chunk = rem / tx, bx = rem - chunk * tx;
    const int x0 = bx * TW, ybeg = chunk * ch_rows;
    const int nst = min(ch_rows, h - ybeg) / S;
    const float *__restrict__ in = src + (size_t)frame * src_frame_stride;
    float *__restrict__ out = dst + (size_t)frame * dst_frame_stride;
    const VTapsSym<R> tw(wt);

    // rows u (image row = ybeg + R + u ... as blur_ring_kernel: u = image row - ybeg - R) live in ring slot u & 63
    auto load_rows16 = [&](int y_first, int prow, int pq, f32x4 (&buf)[NPF4]) {
        const int sy = symm(y_first + prow, h);
        const float *rowp = in + (size_t)sy * w;
#pragma unroll
        for (int j = 0; j < NPF4; j++) buf[j] = load_quad_mirrored(rowp, x0 - RP + 4 * pq + 32 * j, w);
    };
    auto store_rows16 = [&](int u_first, int prow, int pq, const f32x4 (&buf)[NPF4]) {
        float *rowp = lds + ((u_first + prow + NR) & (NR - 1)) * LW;
#pragma unroll
        for (int j = 0; j < NPF4; j++) *reinterpret_cast<f32x4 *>(rowp + 4 * pq + 32 * j) = buf[j];
    };
    auto h_item = [&](int slot, int c4) {                             // 4 outputs at columns c4 ... c4 + 3 of ring row `slot`, in place
        float *rowp = lds + slot * LW;
        constexpr int M0 = (RP - R) / 4, M1 = (RP + R + 3) / 4 + 1;
        float v[4 * (M1 - M0)];
        const lds_cv_f32x4 *rp4 = (const lds_cv_f32x4 *)(rowp + c4);
#pragma unroll
        for (int m = M0; m < M1; m++) {
            const f32x4 tv = rp4[m];
            v[4 * (m - M0) + 0] = tv.x; v[4 * (m - M0) + 1] = tv.y; v[4 * (m - M0) + 2] = tv.z; v[4 * (m - M0) + 3] = tv.w;
        }
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < NT; i++) {
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k] = fmaf(tw(i), v[(RP - R - 4 * M0) + k + i], acc[k]);
        }
        *reinterpret_cast<float4 *>(rowp + RP + c4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    };

    // prologue, all four waves: rows u in [-2R, S) staged (two batches of 32 rows through the 256 threads) and blurred horizontally
    {
        const int prow = tid >> 3, pq = tid & 7;                      // 32 rows x 8 lanes
        f32x4 a[NPF4], b[NPF4];
        load_rows16(ybeg - R, prow, pq, a);                           // u = -2R ... -2R + 31
        constexpr int NB = 2 * R + S - 32;                           // u = -2R + 32 ... S - 1
        if (NB > 0) load_rows16(ybeg - R + 32, min(prow, max(NB - 1, 0)), pq, b);
        store_rows16(-2 * R, prow, pq, a);
        if (NB > 0) store_rows16(-2 * R + 32, min(prow, max(NB - 1, 0)), pq, b);
        lds_barrier();
        for (int item = tid; item < (2 * R + S) * 32; item += 256) h_item((-2 * R + (item >> 5) + NR) & (NR - 1), (item & 31) * 4);
    }
    // ROLEMAP: which two waves take the H role.  0: waves 0-1 of every workgroup (then SIMDs 0-1 of a CU only ever run H waves);
    // 1 / 2: alternate with the workgroup index (its bit 5 / bit 0 inside the XCD), so that a SIMD hosts both kinds
    const int flip = ROLEMAP == 1 ? ((blockIdx.x >> 3) >> 5) & 1 : ROLEMAP == 2 ? (blockIdx.x >> 3) & 1 : 0;
    const int rw = wv ^ (flip << 1);                                  // role index: 0, 1 = H waves, 2, 3 = V waves
    if (rw < 2) {
        // ---- H role: two waves own 8 of the 16 rows of a step each, from the global load to the horizontal pass
        const int prow = rw * 8 + (lane >> 3), pq = lane & 7;         // this lane's row of the step and its 8-lane column slice
        f32x4 pf[NPF4];
        if (nst > 1) load_rows16(ybeg + R + S, prow, pq, pf);         // rows of step 1
        for (int st = 0; st < nst; st++) {
            lds_barrier();                                            // step st's rows are blurred; segment (st + 1) & 3 is free
            if (st + 1 < nst) {
                store_rows16((st + 1) * S, prow, pq, pf);             // (the loads had a whole step to land)
                if (st + 2 < nst) load_rows16(ybeg + R + (st + 2) * S, prow, pq, pf);
                // this wave's 8 rows x 32 column groups = 256 items, 4 per lane; a row is read and written by this wave only
#pragma unroll 1
                for (int it = 0; it < 4; it++) {
                    const int item = it * 64 + lane;
                    h_item(((st + 1) * S + rw * 8 + (item >> 5) + NR) & (NR - 1), (item & 31) * 4);
                }
            }
        }
    } else {
        // ---- V role: waves 2 and 3, 8 output rows of the step each, a lane 2 columns
        const int rg = rw - 2;
        const float *colp = lds + RP + 2 * lane;
        for (int st = 0; st < nst; st++) {
            lds_barrier();
            const int u0 = st * S + rg * RB - 2 * R;
            f32x2 acc[RB];
#pragma unroll
            for (int rr = 0; rr < RB; rr++) { acc[rr].x = 0.0f; acc[rr].y = 0.0f; }
#pragma unroll
            for (int k = 0; k < RB + 2 * R; k++) {
                const int slot = (u0 + k + NR) & (NR - 1);
                const f32x2 v = *(const lds_cv_f32x2 *)(colp + slot * LW);
#pragma unroll
                for (int rr = 0; rr < RB; rr++) {
                    const int i = k - rr;
                    if (i >= 0 && i < NT) {
                        acc[rr].x = fmaf(tw(i), v.x, acc[rr].x);
                        acc[rr].y = fmaf(tw(i), v.y, acc[rr].y);
                    }
                }
            }
            const int gx = x0 + 2 * lane, y0 = ybeg + st * S + rg * RB;
#pragma unroll
            for (int rr = 0; rr < RB; rr++) *reinterpret_cast<f32x2 *>(out + (size_t)(y0 + rr) * w + gx) = acc[rr];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same with everything the pipeline's layers need (activity flags, partial strips, a partial last step): the form that
// was wired into the library for one measurement and taken out again -- see the numbers at the end of this comment.
// Wave-specialised form of the ring blur for the large radii (R >= 9: the layers bound by the vector unit, not by HBM).
// Round 3, after the round-2 review: the four wavefronts of blur_ring_kernel run load-issue, horizontal pass, vertical pass and
// stores one after another behind three barriers per 32-row step.  Here waves 0-1 (the H role) write the prefetched rows of
// step k+1 into the ring, request the rows of step k+2 and run the horizontal pass on step k+1, WHILE waves 2-3 (the V role)
// run the vertical pass and the stores of step k: one barrier per 16-row step, the LDS-heavy and the FMA-dense halves of a step
// side by side, and each role's s_waitcnt bookkeeping trivially exact (the H waves never store to memory, the V waves never
// load from it).  The ring is four 16-row segments: the vertical pass of step k reads segments k-2 ... k (16 + 2R <= 46 rows
// ending with segment k), the H waves fill segment k+1.  A wave of the H role owns 8 rows of a step from the global load to the
// horizontal pass, so no synchronisation is needed inside the role.  Per step both roles issue the same number of FMAs
// (8 (2R+1) per lane).  Same arithmetic and tap order as blur_ring_kernel and blur2_kernel: bit-identical results.
// Measured on 32 x 3840x2160 (profiles/blur_variants_r03_wave_specialised.log): R = 13 0.573-0.598 against 0.598-0.630 ms,
// R = 10 0.502-0.525 against 0.519-0.540 ms.  Needs a column-fast image (w % 4 == 0, w, h >= 64); partial strips and a partial
// last step are handled by guarded stores; no decimated output (that layer has R = 8 in the default schedule) -- the launcher
// falls back to blur_ring_kernel otherwise.
// In the pipeline (64 frames per launch, bench.py A/B on one box): R = 13 1.086 against 1.118 ms (-3 %), but R = 10 WITH the
// activity flags 1.120 against 1.038 ms (+8 %: the flag work lands on the two V waves only and unbalances the roles); blur stage
// 6.64 against 6.39 ms per step.  Not shipped.
template <int R>
struct RingWsGeom {
    static constexpr int S = 16, NR = 64, RP = 16, TW = 128, LW = TW + 2 * RP, NT = 2 * R + 1, RB = 8, NTHR = 256;
    static constexpr int NPF4 = LW / 32, NSUB = TW / 8;
    static_assert(NPF4 * 32 == LW && S + 2 * R + S <= NR && R <= RP && 2 * R + S > 32, "geometry (R = 9 ... 15)");
    static constexpr size_t lds_bytes = (size_t)LW * NR * sizeof(float);
    static constexpr size_t lds_bytes_act = lds_bytes + (size_t)NR * NSUB * sizeof(float);
};

template <int R, bool ACT>
__global__ __launch_bounds__(256, 4) void blur_ring_ws_kernel(const float *__restrict__ src, float *__restrict__ dst, int w, int h,
                                                             size_t src_frame_stride, size_t dst_frame_stride, TapWeights wt, int n_frames,
                                                             int ch_rows /* a multiple of 16 */, Activity act) {
    using G = RingWsGeom<R>;
    constexpr int S = G::S, NR = G::NR, RP = G::RP, LW = G::LW, NT = G::NT, RB = G::RB, NPF4 = G::NPF4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    auto ehm = [&](int slot, int sub) -> float & { return lds[LW * NR + slot * G::NSUB + sub]; };   // max|Eh| of (ring row, 8-column sub-cell), ACT only
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tx = (w + G::TW - 1) / G::TW, nch = (h + ch_rows - 1) / ch_rows;
    const int total = tx * nch * n_frames, per_xcd = (total + 7) >> 3;
    const int t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);      // XCD-aware 1-D order, as blur_ring_kernel
    if (t >= total) return;
    const int frame = t / (tx * nch), rem = t - frame * (tx * nch);
    const int chunk = rem / tx, bx = rem - chunk * tx;
    const int x0 = bx * G::TW, ybeg = chunk * ch_rows;
    const int nst = (min(ch_rows, h - ybeg) + S - 1) / S;
    const float *__restrict__ in = src + (size_t)frame * src_frame_stride;
    float *__restrict__ out = dst + (size_t)frame * dst_frame_stride;
    const VTapsSym<R> tw(wt);

    // ring row u = image row - ybeg - R lives in slot u & 63 (row bookkeeping of blur_ring_kernel)
    auto load_rows = [&](int y_first, int prow, int pq, f32x4 (&buf)[NPF4]) {     // mirror extension without a branch, as blur_ring_kernel
        const float *rowp = in + (size_t)symm(y_first + prow, h) * w;
#pragma unroll
        for (int j = 0; j < NPF4; j++) buf[j] = load_quad_mirrored(rowp, x0 - RP + 4 * pq + 32 * j, w);
    };
    auto store_rows = [&](int u_first, int prow, int pq, const f32x4 (&buf)[NPF4]) {
        float *rowp = lds + ((u_first + prow + NR) & (NR - 1)) * LW;
#pragma unroll
        for (int j = 0; j < NPF4; j++) *reinterpret_cast<f32x4 *>(rowp + 4 * pq + 32 * j) = buf[j];
    };
    // 4 outputs at columns c4 ... c4 + 3 of ring row `slot`, in place (the lanes of one wavefront cover whole rows: all of a
    // row's reads are issued before its first write).  ACT: max |hb - raw| of the 8-column sub-cell this lane shares with its neighbour.
    auto h_item = [&](int slot, int c4, int lane_in_row) {
        float *rowp = lds + slot * LW;
        constexpr int M0 = (RP - R) / 4, M1 = (RP + R + 3) / 4 + 1;
        float v[4 * (M1 - M0)];
        const lds_cv_f32x4 *rp4 = (const lds_cv_f32x4 *)(rowp + c4);
#pragma unroll
        for (int m = M0; m < M1; m++) {
            const f32x4 tv = rp4[m];
            v[4 * (m - M0) + 0] = tv.x; v[4 * (m - M0) + 1] = tv.y; v[4 * (m - M0) + 2] = tv.z; v[4 * (m - M0) + 3] = tv.w;
        }
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < NT; i++) {
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k] = fmaf(tw(i), v[(RP - R - 4 * M0) + k + i], acc[k]);
        }
        *reinterpret_cast<float4 *>(rowp + RP + c4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        if (ACT) {
            constexpr int C = RP - 4 * M0;
            float e = fmaxf(__builtin_fmaxf(__builtin_fmaxf(fabsf(acc[0] - v[C + 0]), fabsf(acc[1] - v[C + 1])), fabsf(acc[2] - v[C + 2])), fabsf(acc[3] - v[C + 3]));
            e = fmaxf(e, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, e), 0x111 /* row_shr:1 */, 0xf, 0xf, true)));
            if (lane_in_row & 1) ehm(slot, lane_in_row >> 1) = e;
        }
    };

    // prologue, all four waves: rows u in [-2R, S) staged (two batches of up to 32 rows, every load in flight before the first
    // LDS write) and blurred horizontally
    {
        const int prow = tid >> 3, pq = tid & 7;
        constexpr int NB = 2 * R + S - 32;                            // rows of the second batch
        f32x4 a[NPF4], b[NPF4];
        load_rows(ybeg - R, prow, pq, a);
        load_rows(ybeg - R + 32, min(prow, NB - 1), pq, b);
        store_rows(-2 * R, prow, pq, a);
        store_rows(-2 * R + 32, min(prow, NB - 1), pq, b);
        lds_barrier();
        for (int item = tid; item < (2 * R + S) * 32; item += 256) h_item((-2 * R + (item >> 5) + NR) & (NR - 1), (item & 31) * 4, item & 31);
    }
    if (wv < 2) {
        // ---- H role
        const int prow = wv * 8 + (lane >> 3), pq = lane & 7;         // this lane's row of a step and its slice of the row
        f32x4 pf[NPF4];
        if (nst > 1) load_rows(ybeg + R + S, prow, pq, pf);           // rows of step 1
        for (int st = 0; st < nst; st++) {
            lds_barrier();                                            // step st's rows are blurred; segment (st + 1) & 3 is free
            if (st + 1 < nst) {
                store_rows((st + 1) * S, prow, pq, pf);               // (requested a whole step ago)
                if (st + 2 < nst) load_rows(ybeg + R + (st + 2) * S, prow, pq, pf);
#pragma unroll 1
                for (int it = 0; it < 4; it++) {                      // this wave's 8 rows x 32 column groups: 4 items per lane
                    const int item = it * 64 + lane;
                    h_item(((st + 1) * S + wv * 8 + (item >> 5) + NR) & (NR - 1), (item & 31) * 4, item & 31);
                }
            }
        }
    } else {
        // ---- V role: 8 output rows of the step per wave, 2 columns per lane
        const int rg = wv - 2;
        const float *colp = lds + RP + 2 * lane;
        const int gx = x0 + 2 * lane;
        for (int st = 0; st < nst; st++) {
            lds_barrier();
            const int u0 = st * S + rg * RB - 2 * R;
            f32x2 cen[ACT ? RB : 1];                                  // hb under each output (the centre tap's operand), for the activity bound
            f32x2 acc[RB];
#pragma unroll
            for (int rr = 0; rr < RB; rr++) { acc[rr].x = 0.0f; acc[rr].y = 0.0f; }
#pragma unroll
            for (int k = 0; k < RB + 2 * R; k++) {
                const int slot = (u0 + k + NR) & (NR - 1);
                const f32x2 v = *(const lds_cv_f32x2 *)(colp + slot * LW);
#pragma unroll
                for (int rr = 0; rr < RB; rr++) {
                    const int i = k - rr;
                    if (ACT && i == R) cen[rr] = v;
                    if (i >= 0 && i < NT) {
                        acc[rr].x = fmaf(tw(i), v.x, acc[rr].x);
                        acc[rr].y = fmaf(tw(i), v.y, acc[rr].y);
                    }
                }
            }
#pragma unroll
            for (int rr = 0; rr < RB; rr++) asm volatile("" : "+v"(acc[rr].x), "+v"(acc[rr].y));       // keep the FMA chains out of the store guards
            unsigned act_mask = 0;
            float eh_row[ACT ? RB : 1];
            if (ACT) {
#pragma unroll
                for (int rr = 0; rr < RB; rr++) eh_row[rr] = ehm((st * S + rg * RB + rr - R + NR) & (NR - 1), lane >> 2);
            }
            const int y0 = ybeg + st * S + rg * RB;
#pragma unroll
            for (int rr = 0; rr < RB; rr++) {
                const int gy = y0 + rr;                               // wave-uniform
                if (gy >= h) continue;
                if (gx + 1 < w) *reinterpret_cast<f32x2 *>(out + (size_t)gy * w + gx) = acc[rr];      // w is even
                if (ACT) {
                    const float lim = act.thr * 0.9999f;
                    const bool f = gx + 1 < w && (fabsf(acc[rr].x - cen[rr].x) + eh_row[rr] > lim || fabsf(acc[rr].y - cen[rr].y) + eh_row[rr] > lim);
                    const unsigned long long b = __ballot(f);
                    act_mask |= (((unsigned)b != 0u) ? 1u : 0u) << rr | (((unsigned)(b >> 32) != 0u) ? 1u : 0u) << (RB + rr);
                }
            }
            if (ACT) {                                                // one store for the wave's RB rows x 2 cells
                const int half = lane >> 5, rr = lane & 31;
                const int gy = y0 + rr, cell = (x0 >> 6) + half;
                if (rr < RB && gy < h && cell < act.ncell)
                    act.dst[(size_t)frame * act.frame_stride + (size_t)gy * act.ncell + cell] = (unsigned char)((act_mask >> (half * RB + rr)) & 1u);
            }
        }
    }
}

}  // namespace siftmi
