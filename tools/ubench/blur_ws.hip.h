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

template <int R>
__global__ __launch_bounds__(256, 4) void blur_ring_ws_kernel(const float *__restrict__ src, float *__restrict__ dst, int w, int h,
                                                             size_t src_frame_stride, size_t dst_frame_stride, TapWeights wt, int n_frames,
                                                             int ch_rows) {
    constexpr int S = 16, NR = 64, RP = 16, TW = 128, LW = TW + 2 * RP, NT = 2 * R + 1, RB = 8;
    constexpr int NPF4 = LW / 32;                                    // 8 lanes per row: 5 float4 each
    static_assert(NPF4 * 32 == LW && S + 2 * R + S <= NR && R <= RP && 2 * R + S > 32, "geometry (R = 9 ... 16)");
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
        load_rows16(ybeg - R + 32, min(prow, 2 * R + S - 32 - 1), pq, b);   // u = -2R + 32 ... S - 1  (2R + S - 32 rows)
        store_rows16(-2 * R, prow, pq, a);
        store_rows16(-2 * R + 32, min(prow, 2 * R + S - 32 - 1), pq, b);
        lds_barrier();
        for (int item = tid; item < (2 * R + S) * 32; item += 256) h_item((-2 * R + (item >> 5) + NR) & (NR - 1), (item & 31) * 4);
    }
    if (wv < 2) {
        // ---- H role: waves 0 and 1 own 8 of the 16 rows of a step each, from the global load to the horizontal pass
        const int prow = wv * 8 + (lane >> 3), pq = lane & 7;         // this lane's row of the step and its 8-lane column slice
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
                    h_item(((st + 1) * S + wv * 8 + (item >> 5) + NR) & (NR - 1), (item & 31) * 4);
                }
            }
        }
    } else {
        // ---- V role: waves 2 and 3, 8 output rows of the step each, a lane 2 columns
        const int rg = wv - 2;
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

}  // namespace siftmi
