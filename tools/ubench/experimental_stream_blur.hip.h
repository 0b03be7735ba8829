// EXPERIMENTAL (not shipped): register-resident streaming form of the Gaussian-layer blur, kept for the
// blur_variants harness.  Bit-exact, but latency-bound at the 2 waves/SIMD its register footprint allows
// (one barrier + one LDS round trip per image row); see DESIGN.md section 6.
#pragma once
#include "dense_kernels.hip.h"

namespace siftmi {

// ------------------------------------------------------------------------------------------------
// Streaming form of the layer blur.  Each lane owns 4 adjacent columns for a whole chunk of rows and
// walks down one image row per step:
//   * the row (prefetched 3 steps ahead into registers) goes into a 3-row LDS ring, which exists only so
//     that the horizontal pass can read the 2R neighbours (9 x ds_read_b128 for 4 outputs at R = 13);
//   * the horizontal result h feeds the vertical pass IN REGISTERS: 2R+1 live output rows per lane, each a
//     float4 accumulator; row q adds w[i] * h to the accumulator of output row q - i (scatter form, taps
//     arrive in increasing i, i.e. exactly the reference's accumulation order); the accumulator that
//     received tap 2R is stored and its slot restarts with tap 0.
// The accumulator slot of output row o is o mod NTP, so the loop is unrolled NTP times and every register
// index is static.  No vertical halo is recomputed (only the 2R warm-up rows per chunk), LDS traffic is
// the horizontal pass only, one barrier per row, global loads stay 3 rows in flight.
// Same arithmetic and tap order as blur2_kernel: bit-identical to it and to the oracle.
template <int R, int NTHR_>
struct StreamGeom {
    static constexpr int RP = (R + 3) & ~3;
    static constexpr int NT = 2 * R + 1;
    static constexpr int NTP = ((NT + 2) / 3) * 3;          // slots / unroll period: multiple of 3 (prefetch + LDS ring depth)
    static constexpr int NTHR = NTHR_;
    static constexpr int SW = NTHR * 4;                     // strip width (output columns)
    static constexpr int LW = SW + 2 * RP;                  // staged row length
    static constexpr int HV = RP / 4;                       // halo float4 per side
    static constexpr size_t lds_bytes = (size_t)3 * LW * sizeof(float);
    static_assert(2 * HV <= NTHR, "halo lanes");
};

template <int R, int NTHR_>
struct StreamState {
    using G = StreamGeom<R, NTHR_>;
    f32x4 acc[G::NTP];
    f32x4 pf[3], pfh[3];
    float w[G::NT];
    const float *in; float *out; float *lds;
    int wimg, himg, x0, ybeg, nq, nout, tid;
    bool main_fast, halo_fast, halo_lane;
    int hx;                                                 // first column of this lane's halo float4 (halo lanes)
    int hoff;                                               // its float offset inside a staged row
};

template <int R, int NTHR_>
__device__ __forceinline__ f32x4 stream_fetch(const StreamState<R, NTHR_> &st, int q, int x, bool fast) {
    const int sy = symm(st.ybeg - R + q, st.himg);
    if (fast && sy >= 0 && sy < st.himg) return *reinterpret_cast<const f32x4 *>(st.in + (size_t)sy * st.wimg + x);
    f32x4 r;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int sx = symm(x + c, st.wimg);
        r[c] = (sx < 0 || sy < 0 || sx >= st.wimg || sy >= st.himg) ? 0.0f : st.in[(size_t)sy * st.wimg + sx];
    }
    return r;
}

template <int R, int NTHR_, int P>
__device__ __forceinline__ void stream_step(StreamState<R, NTHR_> &st, int q) {
    using G = StreamGeom<R, NTHR_>;
    constexpr int RD = P % 3, WR = (P + 1) % 3;             // LDS ring slot read now / slot of row q+1 (q = P mod 3)
    // 1. row q+1 (loaded 3 steps ago) -> LDS ring; refill that queue entry with row q+4
    if (q + 1 < st.nq) {
        float *rowp = st.lds + WR * G::LW;
        *reinterpret_cast<f32x4 *>(rowp + G::RP + 4 * st.tid) = st.pf[WR];
        if (st.halo_lane) *reinterpret_cast<f32x4 *>(rowp + st.hoff) = st.pfh[WR];
    }
    if (q + 4 < st.nq) {
        st.pf[WR] = stream_fetch<R, NTHR_>(st, q + 4, st.x0 + 4 * st.tid, st.main_fast);
        if (st.halo_lane) st.pfh[WR] = stream_fetch<R, NTHR_>(st, q + 4, st.hx, st.halo_fast);
    }
    // 2. horizontal pass of row q for this lane's 4 columns
    f32x4 h;
    {
        const lds_cv_f32x4 *rp4 = (const lds_cv_f32x4 *)(st.lds + RD * G::LW + 4 * st.tid);
        float v[4 + 2 * G::RP];
#pragma unroll
        for (int m = 0; m < (4 + 2 * G::RP) / 4; m++) {
            const f32x4 t = rp4[m];
            v[4 * m + 0] = t.x; v[4 * m + 1] = t.y; v[4 * m + 2] = t.z; v[4 * m + 3] = t.w;
        }
        float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < G::NT; i++) {
#pragma unroll
            for (int k = 0; k < 4; k++) a[k] = fmaf(st.w[i], v[(G::RP - R) + k + i], a[k]);
        }
        h.x = a[0]; h.y = a[1]; h.z = a[2]; h.w = a[3];
    }
    // 3. vertical pass: tap i of output row q - i (slot (P - i) mod NTP); tap 0 restarts a slot
#pragma unroll
    for (int i = 0; i < G::NT; i++) {
        constexpr int dummy = 0; (void)dummy;
        const int s = (P - i + 2 * G::NTP) % G::NTP;
        if (i == 0) {
            st.acc[s].x = fmaf(st.w[0], h.x, 0.0f); st.acc[s].y = fmaf(st.w[0], h.y, 0.0f);
            st.acc[s].z = fmaf(st.w[0], h.z, 0.0f); st.acc[s].w = fmaf(st.w[0], h.w, 0.0f);
        } else {
            st.acc[s].x = fmaf(st.w[i], h.x, st.acc[s].x); st.acc[s].y = fmaf(st.w[i], h.y, st.acc[s].y);
            st.acc[s].z = fmaf(st.w[i], h.z, st.acc[s].z); st.acc[s].w = fmaf(st.w[i], h.w, st.acc[s].w);
        }
    }
    // 4. the slot that just received tap 2R is output row o = q - 2R
    {
        constexpr int so = (P - 2 * R + 2 * G::NTP) % G::NTP;
        const int o = q - 2 * R;
        if (o >= 0 && o < st.nout) {
            const int gy = st.ybeg + o, gx = st.x0 + 4 * st.tid;
            float *op = st.out + (size_t)gy * st.wimg + gx;
            const f32x4 r = st.acc[so];
            if (st.main_fast) {
                *reinterpret_cast<f32x4 *>(op) = r;
            } else {
                if (gx + 0 < st.wimg) op[0] = r.x;
                if (gx + 1 < st.wimg) op[1] = r.y;
                if (gx + 2 < st.wimg) op[2] = r.z;
                if (gx + 3 < st.wimg) op[3] = r.w;
            }
        }
    }
    __syncthreads();
}

template <int R, int NTHR_, int P>
__device__ __forceinline__ void stream_phases(StreamState<R, NTHR_> &st, int qb) {
    if constexpr (P < StreamGeom<R, NTHR_>::NTP) {
        if (qb + P < st.nq) {
            stream_step<R, NTHR_, P>(st, qb + P);
            stream_phases<R, NTHR_, P + 1>(st, qb);
        }
    }
}

template <int R, int NTHR_, int MINW = 1>
__global__ __launch_bounds__(NTHR_, MINW) void blur_stream_kernel(const float *__restrict__ src, float *__restrict__ dst, int w, int h,
                                                                 size_t src_frame_stride, size_t dst_frame_stride, TapWeights wt,
                                                                 int n_frames, int rows_per_chunk) {
    using G = StreamGeom<R, NTHR_>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    StreamState<R, NTHR_> st;
    const int nstrips = (w + G::SW - 1) / G::SW;
    const int nch = (h + rows_per_chunk - 1) / rows_per_chunk;
    // 1-D grid: frame, chunk, strip (strip fastest)
    int t = blockIdx.x;
    const int frame = t / (nstrips * nch);
    t -= frame * (nstrips * nch);
    const int chunk = t / nstrips, strip = t - chunk * nstrips;
    if (frame >= n_frames) return;
    st.in = src + (size_t)frame * src_frame_stride;
    st.out = dst + (size_t)frame * dst_frame_stride;
    st.lds = lds;
    st.wimg = w; st.himg = h;
    st.tid = threadIdx.x;
    st.x0 = strip * G::SW;
    st.ybeg = chunk * rows_per_chunk;
    st.nout = min(rows_per_chunk, h - st.ybeg);
    st.nq = st.nout + 2 * R;
    const int gx = st.x0 + 4 * st.tid;
    st.main_fast = ((w & 3) == 0) && (gx + 4 <= w);
    st.halo_lane = st.tid < 2 * G::HV;
    const bool left = st.tid < G::HV;
    st.hx = left ? st.x0 - G::RP + 4 * st.tid : st.x0 + G::SW + 4 * (st.tid - G::HV);
    st.hoff = left ? 4 * st.tid : G::RP + G::SW + 4 * (st.tid - G::HV);
    st.halo_fast = ((w & 3) == 0) && st.hx >= 0 && st.hx + 4 <= w;
#pragma unroll
    for (int i = 0; i < G::NT; i++) asm volatile("v_mov_b32 %0, %1" : "=v"(st.w[i]) : "s"(wt.w[i]));
    // prologue: row 0 straight into ring slot 0; rows 1, 2, 3 into the prefetch queue (entry = row mod 3)
    {
        const f32x4 r0 = stream_fetch<R, NTHR_>(st, 0, gx, st.main_fast);
        *reinterpret_cast<f32x4 *>(lds + G::RP + 4 * st.tid) = r0;
        if (st.halo_lane) *reinterpret_cast<f32x4 *>(lds + st.hoff) = stream_fetch<R, NTHR_>(st, 0, st.hx, st.halo_fast);
#pragma unroll
        for (int k = 1; k <= 3; k++) {
            if (k < st.nq) {
                st.pf[k % 3] = stream_fetch<R, NTHR_>(st, k, gx, st.main_fast);
                if (st.halo_lane) st.pfh[k % 3] = stream_fetch<R, NTHR_>(st, k, st.hx, st.halo_fast);
            }
        }
    }
    __syncthreads();
    for (int qb = 0; qb < st.nq; qb += G::NTP) stream_phases<R, NTHR_, 0>(st, qb);
}


}  // namespace siftmi
