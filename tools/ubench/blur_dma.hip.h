// blur_dma.hip.h -- EXPERIMENT (tools/ubench only; round 4): the ring blur's row staging by LDS-DMA
// (global_load_lds_dwordx4: global memory -> LDS with no VGPR destination) instead of global_load -> VGPR -> ds_write_b128.
//
// Why it cannot simply replace the staging of blur_ring_kernel: that kernel's ring is two 32-row halves, and the half the NEXT step's
// rows go to still holds the 2R rows the CURRENT step's vertical pass reads; the register prefetch is exactly what lets the rows wait
// (in VGPRs) until that pass is done.  A DMA writes LDS when it lands, so it needs a destination nobody reads: a ring with one more
// segment.  This kernel therefore builds on the wave-specialised form (blur_ws.hip.h: waves 0-1 stage + horizontal pass of step k+1,
// waves 2-3 vertical pass + stores of step k, 16-row steps, one barrier per step) with FIVE 16-row segments (80 rows, 51.2 KB: three
// workgroups per CU): while the V waves read segments k-2 ... k and the H waves blur segment k+1 in place, the H waves' DMA fills
// segment k+2.  The H waves hold no prefetch registers (20 VGPRs), issue no ds_write_b128 for staging (5 per lane and step) and no
// mirror selects; a row's mirror extension is the per-lane SOURCE address (rows), and for the two border strips of a row of
// workgroups a reversed copy of 16 columns inside LDS (columns: a DMA cannot reverse the elements of a float4).
// H8 = true: 8 outputs per lane in the horizontal pass with the ring's XOR swizzle on odd slots, realised through the DMA's source
// address (a row's float4 pairs swapped) -- what the freed registers are for at R = 13.
// Same arithmetic and tap order as blur_ring_kernel: bit-identical results (the harness checks).  w % 128 == 0, h % 16 == 0.
#pragma once
#include "dense_kernels.hip.h"

namespace siftmi {

// one LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to 1 KiB of LDS at the wave-uniform byte address lds_dst
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int R, bool H8>
__global__ __launch_bounds__(256, 3) void blur_ring_dma_kernel(const float *__restrict__ src, float *__restrict__ dst, int w, int h,
                                                              size_t src_frame_stride, size_t dst_frame_stride, TapWeights wt, int n_frames,
                                                              int ch_rows) {
    constexpr int S = 16, NSEG = 5, NR = S * NSEG, RP = 16, TW = 128, LW = TW + 2 * RP, NT = 2 * R + 1, RB = 8;
    constexpr int NPF4 = LW / 32;                                    // prologue: 8 lanes per row, 5 float4 each
    static_assert(R >= 8 && R <= RP && S + 2 * R <= 3 * S && LW * 4 * 8 == 5 * 1024, "geometry: a wave's 8 rows are five 1-KiB DMA pieces");
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
    const bool left_edge = x0 == 0, right_edge = x0 + TW == w;      // strips whose halo columns lie outside the image
    auto slot_of = [](int u) { return ((u % NR) + NR) % NR; };      // ring row u = image row - ybeg - R lives in slot u mod 80 (wave-uniform u)
    auto x4 = [](int slot) { return H8 ? (slot & 1) << 2 : 0; };    // XOR on a float offset inside a row at an odd slot (H8)

    // ---- prologue, all four waves: rows u in [-2R, S) through registers (mirror by reversed float4, as blur_ring_kernel), then their
    // horizontal pass
    auto load_rows = [&](int y_first, int prow, int pq, f32x4 (&buf)[NPF4]) {
        const float *rowp = in + (size_t)symm(y_first + prow, h) * w;
#pragma unroll
        for (int j = 0; j < NPF4; j++) buf[j] = load_quad_mirrored(rowp, x0 - RP + 4 * pq + 32 * j, w);
    };
    auto store_rows = [&](int u_first, int prow, int pq, const f32x4 (&buf)[NPF4]) {
        const int slot = slot_of(u_first + prow);
        float *rowp = lds + slot * LW;
#pragma unroll
        for (int j = 0; j < NPF4; j++) *reinterpret_cast<f32x4 *>(rowp + ((4 * pq + 32 * j) ^ x4(slot))) = buf[j];
    };
    // horizontal pass of one item, in place: 4 outputs (plain) or 8 (H8, swizzled rows)
    auto h_item4 = [&](int slot, int c4) {
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
    auto h_item8 = [&](int slot, int c8) {
        constexpr int M0 = (RP - R) / 4, M1 = (RP + R + 7) / 4 + 1;
        float *rowp = lds + slot * LW + c8;
        const int D = x4(slot);
        const lds_cv_f32x4 *re = (const lds_cv_f32x4 *)(rowp + D), *ro = (const lds_cv_f32x4 *)(rowp - D);
        float v[4 * (M1 - M0)];
#pragma unroll
        for (int m = M0; m < M1; m++) {
            const f32x4 tv = (m & 1) ? ro[m] : re[m];
            v[4 * (m - M0) + 0] = tv.x; v[4 * (m - M0) + 1] = tv.y; v[4 * (m - M0) + 2] = tv.z; v[4 * (m - M0) + 3] = tv.w;
        }
        float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < NT; i++) {
#pragma unroll
            for (int k = 0; k < 8; k++) acc[k] = fmaf(tw(i), v[(RP - R - 4 * M0) + k + i], acc[k]);
        }
        *reinterpret_cast<float4 *>(rowp + RP + D) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4 *>(rowp + RP + 4 - D) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    };
    {
        const int prow = tid >> 3, pq = tid & 7;                      // 32 rows x 8 lanes per batch
        constexpr int NB = 2 * R + S - 32;                           // rows of the second batch
        f32x4 a[NPF4], b[NPF4];
        load_rows(ybeg - R, prow, pq, a);
        load_rows(ybeg - R + 32, min(prow, NB - 1), pq, b);
        store_rows(-2 * R, prow, pq, a);
        store_rows(-2 * R + 32, min(prow, NB - 1), pq, b);
        lds_barrier();
        if (H8) { for (int item = tid; item < (2 * R + S) * 16; item += 256) h_item8(slot_of(-2 * R + (item >> 4)), (item & 15) * 8); }
        else    { for (int item = tid; item < (2 * R + S) * 32; item += 256) h_item4(slot_of(-2 * R + (item >> 5)), (item & 31) * 4); }
    }
    if (wv < 2) {
        // ---- H role: wave wv owns rows wv * 8 ... + 7 of every step, from the DMA to the horizontal pass
        // DMA piece j of a step: lane's byte j * 1024 + lane * 16 of the wave's 8-row block -> (row r, float4 column c4), fixed per lane
        int prow[5], pcol[5];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const int b = j * 1024 + lane * 16;
            prow[j] = b / (LW * 4);
            const int c4 = (b - prow[j] * (LW * 4)) >> 4;
            // the swizzle of an odd slot through the SOURCE: physical float4 c4 holds logical float4 c4 ^ 1 (slot parity = row parity: the
            // first row of a wave's block sits at an even slot); columns outside the image are clamped into the row and fixed up below
            const int lc4 = c4 ^ (H8 ? (prow[j] & 1) : 0);
            pcol[j] = min(max(x0 - RP + 4 * lc4, 0), w - 4);
        }
        auto dma_step = [&](int st) {                                 // rows of step st -> segment st mod 5
            const int u0 = st * S + wv * 8;
            const unsigned lds_base = (unsigned)(slot_of(u0) * LW * 4);   // 8 consecutive slots: u0 is a multiple of 8 and 80 is one of 8
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const float *g = in + (size_t)symm(ybeg + R + u0 + prow[j], h) * w + pcol[j];
                glds16(g, lds_base + (unsigned)j * 1024u);
            }
        };
        // mirrored halo columns of a border strip, rows of step st owned by this wave: logical columns -16 ... -1 (left) are columns
        // 15 ... 0 of the strip, reversed; 128 ... 143 (right) are 127 ... 112.  8 rows x 4 float4 = 32 lanes per side.
        auto fix_halo = [&](int st) {
            const int r = lane >> 2 & 7, q = lane & 3;
            const int slot = slot_of(st * S + wv * 8 + r);
            float *rowp = lds + slot * LW;
            const int x = x4(slot);
            if (left_edge && lane < 32) {
                const f32x4 s = *reinterpret_cast<const f32x4 *>(rowp + ((RP + 12 - 4 * q) ^ x));      // strip columns 12 - 4 q ... 15 - 4 q
                *reinterpret_cast<f32x4 *>(rowp + ((4 * q) ^ x)) = f32x4{s.w, s.z, s.y, s.x};           // halo columns -16 + 4 q ...
            }
            if (right_edge && lane >= 32) {
                const f32x4 s = *reinterpret_cast<const f32x4 *>(rowp + ((RP + TW - 4 - 4 * q) ^ x));   // strip columns 124 - 4 q ...
                *reinterpret_cast<f32x4 *>(rowp + ((RP + TW + 4 * q) ^ x)) = f32x4{s.w, s.z, s.y, s.x};
            }
        };
        if (nst > 1) dma_step(1);
        for (int st = 0; st < nst; st++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the rows of step st + 1 (requested a whole step ago) have landed
            lds_barrier();                                            // step st's rows are blurred; segment (st + 2) mod 5 is free
            if (st + 2 < nst) dma_step(st + 2);
            if (st + 1 < nst) {
                if (left_edge || right_edge) { fix_halo(st + 1); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
                if (H8) {
#pragma unroll 1
                    for (int it = 0; it < 2; it++) {                  // 8 rows x 16 groups of 8 columns: 2 items per lane
                        const int item = it * 64 + lane;
                        h_item8(slot_of((st + 1) * S + wv * 8 + (item >> 4)), (item & 15) * 8);
                    }
                } else {
#pragma unroll 1
                    for (int it = 0; it < 4; it++) {                  // 8 rows x 32 groups of 4 columns: 4 items per lane
                        const int item = it * 64 + lane;
                        h_item4(slot_of((st + 1) * S + wv * 8 + (item >> 5)), (item & 31) * 4);
                    }
                }
            }
        }
    } else {
        // ---- V role: waves 2 and 3, 8 output rows of the step each, a lane 2 columns
        const int rg = wv - 2;
        const float *colp = lds + RP + 2 * lane, *colx = lds + ((RP + 2 * lane) ^ (H8 ? 4 : 0));
        for (int st = 0; st < nst; st++) {
            lds_barrier();
            const int u0 = st * S + rg * RB - 2 * R;                  // even
            f32x2 acc[RB];
#pragma unroll
            for (int rr = 0; rr < RB; rr++) { acc[rr].x = 0.0f; acc[rr].y = 0.0f; }
#pragma unroll
            for (int k = 0; k < RB + 2 * R; k++) {
                const int slot = slot_of(u0 + k);                     // parity of slot = parity of k
                const f32x2 v = *(const lds_cv_f32x2 *)(((k & 1) ? colx : colp) + slot * LW);
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
            for (int rr = 0; rr < RB; rr++) ring_store2(out + (size_t)(y0 + rr) * w + gx, acc[rr]);
        }
    }
}

}  // namespace siftmi
