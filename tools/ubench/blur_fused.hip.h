// blur_fused.hip.h -- EXPERIMENT (tools/ubench only): TWO consecutive Gaussian layers from ONE marching launch.
// Layer A = blur_RA(input), layer B = blur_RB(layer A); both are written to HBM, the input is read once: 12 B per octave pixel instead
// of the 16 B two blur_ring_kernel launches move (the memory-bound pair R = 5 + 7 of octave 0 runs at the ring skeleton's rate today).
//
// Geometry.  A workgroup (256 threads) owns a strip of 112 output columns and marches down a chunk in 32-row steps like blur_ring_kernel.
// Layer B's horizontal pass needs layer A on 8 more columns either side, so stage A works on 128 columns (= exactly one wavefront row of
// the vertical pass: 64 lanes x 2 columns) from a 144-column input window (ring A: 64 rows x 144 floats, 36 KB); its vertical pass writes
// the layer-A rows to HBM (the strip's own 112 columns) AND into ring B (64 rows x 128 floats, 32 KB), where stage B runs its horizontal
// pass in place and its vertical pass one step behind.  68 KB of LDS: two workgroups per CU.
//   macro-step m:  B1 | prefetch rows of m+1, H_A(m) | B2 | V_A(m) -> ring B + layer A | B3 | H_B(block m), ring A <- prefetched rows | B4 |
//                  [mirror copies of H_B'd rows above / below the image] | V_B(m-1) -> layer B
// Row bookkeeping: A block m = layer-A rows a in [ybeg - 8 + 32 m, +32); ring A as blur_ring_kernel with ybegA = ybeg - 8 (uA = y - ybegA - RA);
// ring B slot of layer-A row a = (a - ybeg - RB) & 63.  V_B(t) reads blocks t and t + 1 only (8 + RB <= 16 < 32): the ring holds exactly those.
// Mirror extension (Common.hpp:15-22).  Layer B blurs the SYMMETRIC EXTENSION of layer A: a layer-A value outside the image is a COPY of
// the value at the mirrored position (recomputing it from mirrored input would sum the taps in reverse order: other roundings -- the same
// point as blur_chain_kernel).  Columns: the lanes of V_A that hold the mirror sources also write the copies (border strips only).  Rows:
// the H_B'd rows are copied after B4 (first block of the top chunk, last blocks of the bottom chunk only).
// Same arithmetic and tap order as blur_ring_kernel for each layer: bit-identical results (the harness checks both layers).
// Restrictions of the experiment: w % 4 == 0, w >= 128, h >= 64, RA, RB <= 8, no decimated output, no activity flags.
#pragma once
#include "dense_kernels.hip.h"

namespace siftmi {

struct Ring2Geom {
    static constexpr int S = 32, NR = 64, NTHR = 256, TWO = 112, HA = 8, LWA = 144, LWB = 128;
    static constexpr int NPF4 = LWA / 32, REM = (LWA - 32 * NPF4) / 8;                 // 4 float4 + 1 float2 per lane and staged row
    static constexpr size_t lds_bytes = (size_t)(LWA + LWB) * NR * sizeof(float);     // 69 632
};

// DBG: 4 = no global loads (the rows are constants: a bound for ANY cheaper stage-A loader, e.g. the seed's 1 B per pixel), 8 = no
// horizontal-pass arithmetic, 16 = no vertical-pass arithmetic (timing ablations, wrong results)
// PF2: the input rows of step m + 2 are requested at step m (two register sets): 36 KB per workgroup in flight instead of 18 -- what
// four resident workgroups of blur_ring_kernel keep in flight per CU, from the two that fit here
template <int RA, int RB, int MINW = 2, int DBG = 0, bool PF2 = false>
__global__ __launch_bounds__(256, MINW) void blur_ring2_kernel(const float *__restrict__ src, float *__restrict__ dst_a, float *__restrict__ dst_b, int w, int h,
                                                              size_t frame_stride, TapWeights wt_a, TapWeights wt_b, int n_frames, int ch_rows) {
    using G = Ring2Geom;
    constexpr int S = G::S, NR = G::NR, LWA = G::LWA, LWB = G::LWB, HA = G::HA, TWO = G::TWO;
    static_assert(RA <= 8 && RB <= 8 && RA >= 1 && RB >= 1, "halo of 8 columns / rows per stage");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *ring_a = lds, *ring_b = lds + LWA * NR;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tx = (w + TWO - 1) / TWO, nch = (h + ch_rows - 1) / ch_rows;
    const int total = tx * nch * n_frames, per_xcd = (total + 7) >> 3;
    const int t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);      // XCD-aware 1-D order, as blur_ring_kernel
    if (t >= total) return;
    const int frame = t / (tx * nch), rem = t - frame * (tx * nch);
    const int chunk = rem / tx, bx = rem - chunk * tx;
    const int x0 = bx * TWO, ybeg = chunk * ch_rows;
    const int rows_here = min(ch_rows, h - ybeg);
    const int nst_b = (rows_here + S - 1) / S, nst_a = nst_b + 1;
    const int ybeg_a = ybeg - HA;
    const float *__restrict__ in = src + (size_t)frame * frame_stride;
    float *__restrict__ out_a = dst_a + (size_t)frame * frame_stride;
    float *__restrict__ out_b = dst_b + (size_t)frame * frame_stride;
    const VTapsSym<RA> twa(wt_a);
    const VTapsSym<RB> twb(wt_b);
    // Row stores go through range-checked buffer resources: a lane or row that must not store gets an offset past the layer's end, which
    // the hardware drops.  No branch around any vector-memory instruction between the prefetch and its use, so hipcc's per-basic-block
    // vmcnt bookkeeping stays exact in EVERY step, border strips and partial steps included (blur_ring_kernel needs its FAST / MODE 1 / general
    // bodies for that; the first form of this kernel had guarded stores and ran its loads and arithmetic strictly one after the other).
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto make_rsrc = [&](float *p) {
        const unsigned long long a = (unsigned long long)p;
        const unsigned long long u = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
        return __builtin_amdgcn_make_buffer_rsrc((void *)u, 0, 4 * w * h, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(out_a), rs_b = make_rsrc(out_b);
    auto row_store = [&](__amdgpu_buffer_rsrc_t rs, bool ok, int gy, int gx, f32x2 v) {
        const unsigned off = ok ? (unsigned)(gy * w + gx) * 4u : 0xfffffff0u;
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(v.x), __float_as_uint(v.y)}, rs, (int)off, 0, 2 /* nt */);
    };

    // staging of ring A: 8 lanes per row, mirror extension by row (symm) and by float4 (load_quad_mirrored), as blur_ring_kernel
    const int pf_row = tid >> 3, pf_q = tid & 7;
    auto load_rows = [&](int y_first, int nrows, f32x4 (&buf)[G::NPF4], f32x2 &rm) {
        if (DBG & 4) {
#pragma unroll
            for (int j = 0; j < G::NPF4; j++) buf[j] = f32x4{1.0f, 2.0f, 3.0f, (float)j};
            rm = f32x2{1.0f, 2.0f};
            return;
        }
        const float *rowp = in + (size_t)symm(y_first + min(pf_row, nrows - 1), h) * w;
#pragma unroll
        for (int j = 0; j < G::NPF4; j++) buf[j] = load_quad_mirrored(rowp, x0 - 2 * HA + 4 * pf_q + 32 * j, w);
        rm = load_pair_mirrored(rowp, x0 - 2 * HA + 32 * G::NPF4 + 2 * pf_q, w);
    };
    auto store_rows = [&](int u_first, int nrows, const f32x4 (&buf)[G::NPF4], const f32x2 &rm) {
        float *rowp = ring_a + ((u_first + min(pf_row, nrows - 1) + 2 * NR) & (NR - 1)) * LWA;
#pragma unroll
        for (int j = 0; j < G::NPF4; j++) *reinterpret_cast<f32x4 *>(rowp + 4 * pf_q + 32 * j) = buf[j];
        *reinterpret_cast<f32x2 *>(rowp + 32 * G::NPF4 + 2 * pf_q) = rm;
    };
    // horizontal pass of radius R, in place: 4 outputs at positions 8 + c4 ... of `rowp` (a row's 32 lanes sit in one wavefront: every read
    // of a row is issued before its first write)
    auto h_item = [&](auto tw, auto rtag, float *rowp, int c4) {
        constexpr int R = decltype(rtag)::value, NT = 2 * R + 1;
        constexpr int M0 = (HA - R) / 4, M1 = (HA + R + 3) / 4 + 1;
        float v[4 * (M1 - M0)];
        const lds_cv_f32x4 *rp4 = (const lds_cv_f32x4 *)(rowp + c4);
#pragma unroll
        for (int m = M0; m < M1; m++) {
            const f32x4 tv = rp4[m];
            v[4 * (m - M0) + 0] = tv.x; v[4 * (m - M0) + 1] = tv.y; v[4 * (m - M0) + 2] = tv.z; v[4 * (m - M0) + 3] = tv.w;
        }
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (!(DBG & 8)) {
#pragma unroll
            for (int i = 0; i < NT; i++) {
#pragma unroll
                for (int k = 0; k < 4; k++) acc[k] = fmaf(tw(i), v[(HA - R - 4 * M0) + k + i], acc[k]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k] = v[(HA - 4 * M0) + k];
        }
        *reinterpret_cast<float4 *>(rowp + HA + c4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    };
    // vertical pass of radius R for 8 rows of this wavefront: window rows from `ring` (row pitch LW, first slot-independent index u0)
    auto v_rows = [&](auto tw, auto rtag, const float *colp, int LW, int u0, f32x2 (&acc)[8]) {
        constexpr int R = decltype(rtag)::value, NT = 2 * R + 1;
#pragma unroll
        for (int rr = 0; rr < 8; rr++) { acc[rr].x = 0.0f; acc[rr].y = 0.0f; }
#pragma unroll
        for (int k = 0; k < 8 + 2 * R; k++) {
            const int slot = (u0 + k + 2 * NR) & (NR - 1);
            const f32x2 v = *(const lds_cv_f32x2 *)(colp + slot * LW);
#pragma unroll
            for (int rr = 0; rr < 8; rr++) {
                const int i = k - rr;
                if (i >= 0 && i < NT) {
                    if (!(DBG & 16)) { acc[rr].x = fmaf(tw(i), v.x, acc[rr].x); acc[rr].y = fmaf(tw(i), v.y, acc[rr].y); }
                    else if (i == R) acc[rr] = v;
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < 8; rr++) asm volatile("" : "+v"(acc[rr].x), "+v"(acc[rr].y));   // keep the FMA chains out of the store guards
    };

    // prologue: input rows uA in [-2 RA, S)
    {
        f32x4 a[G::NPF4], b[G::NPF4];
        f32x2 ar, br;
        load_rows(ybeg_a - RA, S, a, ar);
        load_rows(ybeg_a - RA + S, 2 * RA, b, br);
        store_rows(-2 * RA, S, a, ar);
        store_rows(-2 * RA + S, 2 * RA, b, br);
    }
    const int pw = w - x0 + HA;                               // ring-B position of the first column right of the image (>= 128: not in this strip)
    const bool left_border = x0 == 0, right_border = pw < LWB;
    f32x4 pf2[2][G::NPF4];                                    // PF2: rows of step m + 1 (set m & 1 ... see below) and m + 2
    f32x2 pfr2[2];
    if (PF2 && nst_a > 1) load_rows(ybeg_a + RA + S, S, pf2[1], pfr2[1]);      // step 1's rows, requested before step 0
    auto macro_step = [&](int m, auto par_tag) {
        constexpr int PAR = decltype(par_tag)::value;         // m & 1 (static register-set index)
        lds_barrier();                                        // B1: ring A rows of step m staged
        f32x4 pf[G::NPF4];
        f32x2 pfr;
        const bool has_next = m + 1 < nst_a;
        if (!PF2) { if (has_next) load_rows(ybeg_a + RA + (m + 1) * S, S, pf, pfr); }
        else if (m + 2 < nst_a) load_rows(ybeg_a + RA + (m + 2) * S, S, pf2[PAR], pfr2[PAR]);   // set PAR held step m's rows: free
        {   // H_A on the new rows (step 0: + the 2 RA prologue rows)
            const int hb = m == 0 ? -2 * RA : m * S, hn = m == 0 ? S + 2 * RA : S;
#pragma unroll 1
            for (int item = tid; item < hn * 32; item += 256)
                h_item(twa, std::integral_constant<int, RA>{}, ring_a + ((hb + (item >> 5) + 2 * NR) & (NR - 1)) * LWA, (item & 31) * 4);
        }
        lds_barrier();                                        // B2
        const int a0 = ybeg_a + m * S;                        // first layer-A row of block m
        {   // V_A: 8 rows per wavefront, 2 columns per lane -> ring B (raw) and layer A in HBM
            f32x2 acc[8];
            v_rows(twa, std::integral_constant<int, RA>{}, ring_a + HA + 2 * lane, LWA, m * S + wv * 8 - 2 * RA, acc);
            const int gx = x0 - HA + 2 * lane;
            const bool own_col = lane >= HA / 2 && lane < (HA + TWO) / 2 && gx < w;
#pragma unroll
            for (int rr = 0; rr < 8; rr++) {
                const int a = a0 + wv * 8 + rr;               // wave-uniform
                float *rowb = ring_b + ((a - ybeg - RB + 2 * NR) & (NR - 1)) * LWB;
                *reinterpret_cast<f32x2 *>(rowb + 2 * lane) = acc[rr];
                if (left_border && lane >= 4 && lane < 8) *reinterpret_cast<f32x2 *>(rowb + 14 - 2 * lane) = f32x2{acc[rr].y, acc[rr].x};   // columns -1 - j := j
                if (right_border && 2 * lane >= pw - 8 && 2 * lane < pw && 2 * pw - 2 - 2 * lane < LWB)       // columns w + j := w - 1 - j
                    *reinterpret_cast<f32x2 *>(rowb + 2 * pw - 2 - 2 * lane) = f32x2{acc[rr].y, acc[rr].x};
                row_store(rs_a, a >= ybeg && a < ybeg + rows_here && own_col, a, gx, acc[rr]);
            }
        }
        lds_barrier();                                        // B3: block m of ring B complete (raw); every read of ring A's other half done
        {   // H_B on block m, in place (28 of a row's 32 lanes: positions 8 ... 119)
#pragma unroll 1
            for (int item = tid; item < S * 32; item += 256) {
                const int c4 = (item & 31) * 4;
                if (c4 < TWO) h_item(twb, std::integral_constant<int, RB>{}, ring_b + ((a0 + (item >> 5) - ybeg - RB + 2 * NR) & (NR - 1)) * LWB, c4);
            }
        }
        if (has_next) {                                       // (requested a step or two ago; behind H_B so that a wait for V_A's stores costs nothing)
            if (!PF2) store_rows((m + 1) * S, S, pf, pfr);
            else store_rows((m + 1) * S, S, pf2[PAR ^ 1], pfr2[PAR ^ 1]);
        }
        lds_barrier();                                        // B4
        if (a0 < 0 || a0 + S > h) {                           // uniform: rows of this block above / below the image <- the H_B'd rows at their mirrored positions
            for (int idx = tid; idx < S * 32; idx += 256) {
                const int a = a0 + (idx >> 5), q = (idx & 31) * 4;
                if ((a < 0 || a >= h) && a < h + 8) {
                    const int am = a < 0 ? -a - 1 : 2 * h - 1 - a;
                    const f32x4 vv = *reinterpret_cast<const f32x4 *>(ring_b + ((am - ybeg - RB + 2 * NR) & (NR - 1)) * LWB + q);
                    *reinterpret_cast<f32x4 *>(ring_b + ((a - ybeg - RB + 2 * NR) & (NR - 1)) * LWB + q) = vv;
                }
            }
            lds_barrier();
        }
        if (m >= 1) {   // V_B of step m - 1: layer-B rows ybeg + (m - 1) S + wv 8 ...
            const int tb = m - 1;
            f32x2 acc[8];
            v_rows(twb, std::integral_constant<int, RB>{}, ring_b + HA + 2 * lane, LWB, tb * S + wv * 8 - 2 * RB, acc);
            const int gx = x0 + 2 * lane;
            const bool own_col = lane < TWO / 2 && gx < w;
#pragma unroll
            for (int rr = 0; rr < 8; rr++) {
                const int gy = ybeg + tb * S + wv * 8 + rr;   // wave-uniform
                row_store(rs_b, gy < h && own_col, gy, gx, acc[rr]);
            }
        }
    };
    for (int m = 0; m < nst_a; m += 2) {
        macro_step(m, std::integral_constant<int, 0>{});
        if (m + 1 < nst_a) macro_step(m + 1, std::integral_constant<int, 1>{});
    }
}

}  // namespace siftmi
