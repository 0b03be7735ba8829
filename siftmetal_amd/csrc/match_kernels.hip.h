// match_kernels.hip.h -- brute-force descriptor matching with ratio test for gfx950 (MI355X).
//
// Replaces SIFTDescriptor.match(source:target:absoluteThreshold:relativeThreshold:)
// (Sources/SIFTMetal/SIFT/SIFTDescriptor.swift:298-361), the first "next" row of SURVEY.md 8f.
//
// The reference scans the targets sequentially per source descriptor with
//     if distance < best { second = best; best = distance; bestIndex = t }
// so `second` is the running minimum at the moment the final best was found, i.e. the minimum over the
// targets BEFORE the best one (FLT_MAX if the best is the first target) -- not the true second nearest.
// Any CONTIGUOUS chunk of targets can be scanned on its own into (best, first index of best, second = minimum
// over the chunk's targets before that index), and chunks combine in target order with
//     if (b < best) { second = min(best, s); best = b; idx = i; }
// which is how the kernel below parallelises the scan without changing its result.
// Distances: the reference takes the f32 Euclidean distance of features/255; here the squared distance of
// the 0..255 integers is formed exactly in int32 (bytes re-biased to signed i8, int8 MFMA / v_dot4_i32_i8:
// |a-b|^2 = [source norm] + [target norm] + 2 a'.v with a' = 127 - a, v = b - 128, see match_prep_kernel) and distance =
// sqrt(D) / 255 in f32; the two agree
// to f32 rounding (~1e-6 relative), which only matters on threshold knife edges.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "keypoint_kernels.hip.h"

namespace siftmi {

struct MatchRec { int32_t source, target; float distance; };          // == siftmi_match

__device__ __forceinline__ int dot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }

// =====================================================================================================
// MFMA matcher.  The distance matrix is a GEMM (|a-b|^2 = |a|^2 + |b|^2 - 2 a.b, K = 128 int8), so it runs on
// the matrix cores: v_mfma_i32_32x32x32_i8, exact int32 accumulation.  A = 32 targets (LDS tile shared by the
// block's 4 waves), B = 32 sources held in registers for the whole kernel (2 source tiles per wave, 256
// sources per block).  With targets on the rows the C/D layout (col = lane&31, row = (reg&3) + 8(reg>>2) +
// 4(lane>>5)) leaves one SOURCE per lane and 16 TARGETS in its accumulator registers, so the per-source scan is
// lane-local.  The rows of a tile are assigned so that accumulator register i of lane half h is target
// chunk_h + 16*tile + i: each lane walks a CONTIGUOUS chunk of targets in index order, which is what the
// reference's order-dependent `second` needs (see the top of this file):
//     lane state: best, idx (first minimum in the chunk), second = min over the chunk's targets before idx
//     chunks combine in order:  if (b < best) { second = min(best, s); best = b; idx = i; }
// (half 0 before half 1 inside a block, target splits across blockIdx.y in match_finalize_kernel).
// The scan costs 16 mad + 8 min3 + 1 compare per 32x32 tile; the ordered 16-step update only runs when some
// lane's tile minimum beats its best (O(log n) times per source).
//
// Round 4: chunks start from a BOUND instead of from "no best".  A chunk that restarts at "no best" sets a record in about
// ln(chunk) of its tiles per lane, and a wavefront takes the 16-step ordered update when ANY of its 64 lanes does -- 40-55 % of
// all tiles at the split sizes that fill the chip, the largest single cost of the kernel.  But a target can only enter the
// reference's result (as the best, or as the `second` = the running best just before the last improvement) if it beats the running
// best at its position, and the minimum over ANY earlier targets is an upper bound on that.  So a short pre-pass (this kernel over
// the first P = 512 targets alone, one more launch) leaves every source's best key among them, and every block whose chunk does not
// start at target 0 starts with best = min(that, what earlier splits' blocks have published so far): tiles without a key below the
// bound are screened out, and
// the first improvement of a chunk records the bound as its `second`, which the ordered combine (min with the running best, which is
// <= the bound) treats exactly like "none".  A chunk that never improves reports "none".  Same results, bit for bit
// (tests/test_gpu_parity.py::test_match_*); the ordered path is then taken on ~1024 / (targets before the tile) of the tiles.
// =====================================================================================================

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
#ifdef MM_COUNT_SLOW
__device__ unsigned long long mm_slow_count;               // tools/ubench: (wavefront, tile, source tile) triples that took the ordered update
#endif
#ifdef MM_STAMPS
__device__ unsigned long long mm_stamps[4];                // tools/ubench: wavefront cycles in {prefetch issue, MFMA groups + screens, staging, barrier}
#define MM_STAMP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); mm_acc_[k] += t_ - mm_t_; mm_t_ = t_; } while (0)
#else
#define MM_STAMP(k) do {} while (0)
#endif

#ifndef MM_NB_DEF                                          // (tools/ubench/match_variants.hip builds other shapes)
#define MM_NB_DEF 4
#endif
#ifndef MM_TT_DEF
#define MM_TT_DEF 2
#endif
#ifndef MM_PIPE_DEF                                        // 1: two accumulator/fragment sets, group g + 1 issued before g is screened
#define MM_PIPE_DEF 1
#endif
#ifndef MM_CIN_DEF                                         // 1: -(|b'|^2 >> 1) enters as the MFMA chain's C operand (16 registers per set);
#define MM_CIN_DEF 1                                       // 0: chains start from 0 and the screen adds it (16 more vector adds per 4 MFMAs)
#endif
#ifndef MM_XPF_DEF                                         // 1: three LDS buffers; the A fragments / C-in of the NEXT iteration's first tile are read
#define MM_XPF_DEF 0                                       // before the barrier (needs MM_PIPE_DEF).  Measured: no gain (1.52-1.54 against 1.41-1.50 ms), off
#endif
#ifndef MM_WAVES_DEF
#define MM_WAVES_DEF 2
#endif
constexpr int MM_NB = MM_NB_DEF;                           // source tiles (of 32) per wave
constexpr int MM_TT = MM_TT_DEF;                           // target tiles (of 32) staged per barrier
constexpr int MM_SRC_PER_BLOCK = 32 * MM_NB * 4;           // 512
constexpr int MM_SPLIT_QUANTUM = 32 * MM_TT;               // a split's target range is a multiple of this
constexpr int MM_ROW = 144;                                // LDS row stride in bytes: 128 + 16 -> conflict-free ds_read_b128
constexpr int MM_NONE = 0x7fffffff;
constexpr int MM_PAD_NORM = 0x03ffffff;                    // |b|^2 of a padding row: its key can never win (and (acc << 5) still fits)
__device__ __forceinline__ int mm_thr(int best) { return (best >> 1) + (best & 1); }     // ceil(best / 2): acc < thr is necessary for key < best
constexpr int MM_PAD_LIMIT = 0x02000000;                   // real keys are below 2^23

// features -> dense rows of 128 re-biased int8 (32 dwords) + |b'|^2.  32 threads per descriptor.
// (both sides in one launch: blocks [0, src_blocks) pack the sources, the rest the targets -- a small call is launch-bound)
__global__ __launch_bounds__(256) void match_prep_kernel(const DescriptorRec *__restrict__ d_src, int n_src, int *__restrict__ packed_src,
                                                        int *__restrict__ norm_src, int src_blocks, const DescriptorRec *__restrict__ d_tgt,
                                                        int n_tgt, int *__restrict__ packed_tgt, int *__restrict__ norm_tgt) {
    const bool is_src = (int)blockIdx.x < src_blocks;
    const DescriptorRec *__restrict__ d = is_src ? d_src : d_tgt;
    const int n = is_src ? n_src : n_tgt;
    int *__restrict__ packed = is_src ? packed_src : packed_tgt;
    int *__restrict__ norm = is_src ? norm_src : norm_tgt;
    const long long gid = (long long)(is_src ? blockIdx.x : blockIdx.x - src_blocks) * 256 + threadIdx.x;
    const int i = (int)(gid >> 5), k = (int)(gid & 31);
    // Bytes re-biased to signed: v = f - 128.  Targets are packed as v; sources as 127 - f = ~v, so that the MFMA chain yields
    // MINUS the products the distance needs (round 4: key = 2 acc + parity is then one v_lshl_add in the ordered scan).  With
    // a' = 127 - f_a and v = f_b - 128:  |f_a - f_b|^2 = sum (f_a - 128)^2 + [sum (f_b - 127)^2 - 128] + 2 a'.v -- the first term is
    // the source's norm, the bracket the target's.
    int v = 0;
    if (i < n) {
        v = reinterpret_cast<const int *>(d[i].features)[k] ^ (int)0x80808080;
        packed[(long long)i * 32 + k] = is_src ? ~v : v;
    }
    int nb = dot4(v, v, 0);
    if (!is_src) nb = dot4(v, 0x02020202, nb);                 // |v|^2 + 2 sum v = sum (v + 1)^2 - 128
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) nb += __shfl_xor(nb, o, 64);
    if (i < n && k == 0) norm[i] = nb;
}

// FUSED (round 6): the whole match as ONE launch, for calls of the sizes the path produces (two frames' descriptors: a few thousand a
// side; up to 128 source groups).  Rounds 3-5 ran prep -> this kernel -> finalize (-> compaction): at 2.5 k x 2.3 k that is four
// dependent launches of 5-8 us around 2 us of matrix work.  With FUSED
//   * the operands come straight from the descriptor records (136-byte stride, features at byte 8): re-biased in registers on their
//     way into the B fragments / the LDS tile, norms summed in the same pass (match_prep_kernel's arithmetic, dword for dword);
//   * the LAST block of a source group to finish (a ticket per group) combines the group's splits in target order, applies the
//     thresholds (match_finalize_kernel's arithmetic), writes the per-source records and -- for the device-resident call -- packs the
//     matched ones in source order: every group publishes its match count under the call's epoch, a group adds up the counts of the
//     groups before it (one thread per earlier group polls; no chain) and writes its matches behind them; the last group writes the total.
// Same keys, same ordered combine, same thresholds: identical indices (tests/test_gpu_parity.py::test_match_*).
// A group's last block waits only for blocks that are already running or will be dispatched into slots other blocks free (at most one
// block per group ever waits: <= 128 of 512 resident).
struct MatchTail {
    float abs_thr, rel_thr;
    MatchRec *out;                     // one record per source (target -1: no match)
    MatchRec *packed;                  // or null: the matched records in source order
    int32_t *count;                    // their number (with `packed`)
    unsigned *ticket;                  // [groups], zero between calls (the last block of a group resets its own)
    unsigned long long *status;        // [groups]: epoch << 32 | matches of the group
    unsigned epoch;                    // of this call; never 0
};
typedef int i32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));      // a 16-byte piece of a descriptor record: dword-aligned only

// amdgpu_waves_per_eu(2): caps the kernel at 256 VGPRs, which makes the compiler keep the MFMA results in VGPRs
// (no v_accvgpr_read per element in the scan).
template <bool FUSED = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MM_WAVES_DEF, MM_WAVES_DEF)))
void match_mfma_kernel(const int *__restrict__ src_packed /* FUSED: the source DescriptorRec array */, int n_src,
                       const int *__restrict__ tgt_packed /* FUSED: the target DescriptorRec array */,
                       const int *__restrict__ tgt_norm /* FUSED: unused */, int n_tgt, int split_len /* multiple of MM_SPLIT_QUANTUM */,
                       int4 *__restrict__ part /* [gridDim.y][n_src]: best, idx, second */,
                       const int4 *__restrict__ bound /* or null: [n_src] records of a pre-pass over targets [0, P), P <= split_len */,
                       MatchTail tail = MatchTail{}) {
    constexpr bool XPF = MM_XPF_DEF && MM_PIPE_DEF && MM_TT == 2;
    constexpr int NBUF = XPF ? 3 : 2;
    __shared__ __attribute__((aligned(16))) unsigned char lds_a[NBUF][MM_TT][32 * MM_ROW];
    __shared__ __attribute__((aligned(16))) int lds_c[NBUF][MM_TT][32];    // C-in of the MFMA chain: the target's norm >> 1
    __shared__ __attribute__((aligned(16))) int lds_n[NBUF][MM_TT][32];    // parity bit of the target's norm (only read on the ordered-update path)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 31, h = lane >> 5;
    const int half_len = split_len >> 1, n_iter = half_len / (16 * MM_TT);
    const int t_lo = blockIdx.y * split_len, t_hi = min(n_tgt, t_lo + split_len);
    // sources of this wave: B operands, resident
    i32x4 b[MM_NB][4];
    int src_nrm[MM_NB];                                      // FUSED: sum (f - 128)^2 of this lane's source (the finalize step's per-source constant)
    const int s0 = blockIdx.x * MM_SRC_PER_BLOCK + wv * (32 * MM_NB);
#pragma unroll
    for (int nb = 0; nb < MM_NB; nb++) {
        const int sc = min(s0 + nb * 32 + c, n_src - 1);
        src_nrm[nb] = 0;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            if (FUSED) {
                const unsigned char *rp = reinterpret_cast<const unsigned char *>(src_packed) + (size_t)sc * sizeof(DescriptorRec) + 8 + m * 32 + h * 16;
                const i32x4 f = *reinterpret_cast<const i32x4_a4 *>(rp);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int v = f[e] ^ (int)0x80808080;                     // f - 128 per byte, signed
                    b[nb][m][e] = ~v;                                        // 127 - f (match_prep_kernel)
                    src_nrm[nb] = dot4(v, v, src_nrm[nb]);
                }
            } else {
                b[nb][m] = *reinterpret_cast<const i32x4 *>(src_packed + (long long)sc * 32 + m * 8 + h * 4);
            }
        }
        if (FUSED) src_nrm[nb] += __shfl_xor(src_nrm[nb], 32, 64);           // the other half of every 32-byte slice
    }
    // staging role of this thread: tile row r (MFMA row), 16-byte piece p, of each of the MM_TT tiles
    const int r = tid >> 3, p = tid & 7, hr = (r >> 2) & 1, pos = (r >> 3) * 4 + (r & 3);
    int st_base = t_lo + hr * half_len + pos, st_end = hr ? t_hi : min(t_hi, t_lo + half_len);
    i32x4 pre_v[MM_TT]; int pre_n[MM_TT];
    auto prefetch = [&](int it) {
#pragma unroll
        for (int j = 0; j < MM_TT; j++) {
            const int t = st_base + (it * MM_TT + j) * 16;
            const bool valid = t < st_end;
            if (FUSED) {
                const unsigned char *rp = reinterpret_cast<const unsigned char *>(tgt_packed) + (size_t)(valid ? t : 0) * sizeof(DescriptorRec) + 8 + p * 16;
                const i32x4 f = *reinterpret_cast<const i32x4_a4 *>(rp);
                int nb_ = 0;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int v = f[e] ^ (int)0x80808080;
                    pre_v[j][e] = v;
                    nb_ = dot4(v, 0x02020202, dot4(v, v, nb_));              // |v|^2 + 2 sum v = sum (v + 1)^2 - 128 over the row (match_prep_kernel)
                }
                nb_ += __shfl_xor(nb_, 1, 64); nb_ += __shfl_xor(nb_, 2, 64); nb_ += __shfl_xor(nb_, 4, 64);   // the row's 8 pieces: lanes 8 k ... 8 k + 7
                pre_n[j] = (p == 0 && valid) ? nb_ : MM_PAD_NORM;
                continue;
            }
            pre_v[j] = *reinterpret_cast<const i32x4 *>(tgt_packed + (long long)(valid ? t : 0) * 32 + p * 4);
            pre_n[j] = MM_PAD_NORM;
            if (p == 0 && valid) pre_n[j] = tgt_norm[t];
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < MM_TT; j++) {
            *reinterpret_cast<i32x4 *>(&lds_a[buf][j][r * MM_ROW + p * 16]) = pre_v[j];
            if (p == 0) {
                lds_c[buf][j][hr * 16 + pos] = pre_n[j] >> 1;     // (arithmetic: a target's norm can be as low as -128)
                lds_n[buf][j][hr * 16 + pos] = pre_n[j] & 1;
            }
        }
    };
    // key = N_b + 2 a'.v (the distance without the per-source constant; N_b = the target's norm, see match_prep_kernel).  The chain
    // starts from C = N_b >> 1, so acc = a'.v + (N_b >> 1) and key = 2 acc + (N_b & 1).  Screening test on acc alone:
    // key < best  =>  acc < ceil(best / 2) = thr  (exact up to the parity bit; the ordered path re-tests exactly).
    int best[MM_NB], idx[MM_NB], second[MM_NB], thr[MM_NB];
#pragma unroll
    for (int nb = 0; nb < MM_NB; nb++) { best[nb] = MM_NONE; idx[nb] = -1; second[nb] = MM_NONE; thr[nb] = mm_thr(MM_NONE); }

    // ---- the starting bound of a chunk that does not begin at target 0 (round 4): the best key of the pre-pass -- this same kernel
    // launched over targets [0, P) alone, P <= split_len, its records in `bound` --
    // and whatever the blocks of EARLIER splits (targets before this chunk) have already published: their chunk's best key per
    // source (`part` is cleared to "none" before the launch, a finished block stores its records write-through).  Blocks are
    // dispatched roughly in split order, so a late split usually finds most of its predecessors done.  Any value read is the key of a
    // real earlier target or "none" -- a valid bound whatever the timing, so the RESULT does not depend on it, only the number of
    // ordered updates does.
    if (bound != nullptr && blockIdx.y > 0) {
#pragma unroll
        for (int nb = 0; nb < MM_NB; nb++) {
            const int s = min(s0 + nb * 32 + c, n_src - 1);
            int pub = bound[s].x;
            // (at most 8 of the predecessors, evenly spaced: with hundreds of splits a walk over all of them is a chain of L2 latencies
            // longer than the chunk)
            const int ystep = max(1, ((int)blockIdx.y + 7) >> 3);
            for (int y = (int)blockIdx.y - 1; y >= 0; y -= ystep)
                pub = min(pub, __hip_atomic_load(reinterpret_cast<const int *>(part + (long long)y * n_src + s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (pub < best[nb] && pub > -MM_PAD_LIMIT) { best[nb] = pub; thr[nb] = mm_thr(best[nb]); }
        }
    }

    if (n_iter > 0) { prefetch(0); stage(0); }
    if (XPF && n_iter > 1) { prefetch(1); stage(1); }
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the source fragments have landed, so the loop's only VMEM waits are its prefetches
    __syncthreads();
    // Software pipeline inside an iteration (round 4).  The work of an iteration is G = MM_TT x MM_NB / 2 groups of 8 MFMAs (one
    // target tile x two source tiles) each followed by the max screen of its 2 x 16 accumulators.  Round 1 ran "8 MFMAs, then their
    // screen" group by group: a wavefront's matrix pipe idled through every screen and every wait for the last MFMA's result, and the
    // LDS reads of a tile's A fragments sat in front of its first MFMA (PMC: matrix pipe 43 % busy at two wavefronts per SIMD).  Now
    // the MFMAs of group g + 1 are ISSUED before group g is screened (two accumulator sets), and the A fragments / chain inputs of
    // tile j + 1 are read from LDS while tile j's last group runs (two fragment sets): the screen's ~40 vector instructions and the
    // LDS latency fall into the shadow of 256 cycles of MFMAs.
    constexpr int PAIRS = MM_NB / 2, G = MM_TT * PAIRS;
    static_assert(MM_NB % 2 == 0, "source tiles are processed in pairs");
#ifdef MM_STAMPS
    unsigned long long mm_t_, mm_acc_[4] = {0, 0, 0, 0};
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(mm_t_)::"memory");
#endif
    // XPF (round 4 experiment, off): with two LDS buffers every wavefront begins an iteration -- right behind the barrier -- with 16
    // ds_read_b128 for its first tile, all eight wavefronts of a CU at once.  With THREE buffers the targets of iteration it + 2 are
    // staged at the end of iteration it, so iteration it + 1's buffer is complete a whole iteration early and its first tile's
    // fragments are read during iteration it's last groups, into the fragment set the first tile has finished with.  Same results,
    // no gain: that burst is not where the matrix pipe's idle time comes from (profiles/match_variants_r04_occupancy.log).
    constexpr int NSET = MM_PIPE_DEF ? 2 : 1;
    i32x4 af[NSET][4];
    i32x16 cinf[NSET];
    auto load_frags = [&](int buf, int j, i32x4 (&a)[4], i32x16 &cin) {
#pragma unroll
        for (int m = 0; m < 4; m++) a[m] = *reinterpret_cast<const i32x4 *>(&lds_a[buf][j][c * MM_ROW + m * 32 + h * 16]);
        if (MM_CIN_DEF) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const i32x4 v = *reinterpret_cast<const i32x4 *>(&lds_c[buf][j][h * 16 + q * 4]);
                cin[q * 4 + 0] = v[0]; cin[q * 4 + 1] = v[1]; cin[q * 4 + 2] = v[2]; cin[q * 4 + 3] = v[3];
            }
        }
    };
    if (XPF && n_iter > 0) load_frags(0, 0, af[0], cinf[0]);
    int cur = 0;
    for (int it = 0; it < n_iter; it++, cur = (cur + 1 == NBUF ? 0 : cur + 1)) {
        const int nxt = cur + 1 == NBUF ? 0 : cur + 1;             // the buffer of iteration it + 1
        const int fill = XPF ? (nxt + 1 == NBUF ? 0 : nxt + 1) : nxt;   // the buffer this iteration stages (it + 2 with XPF, it + 1 without)
#ifndef MM_VARIANT_NO_LOAD
        if (it + (XPF ? 2 : 1) < n_iter) prefetch(it + (XPF ? 2 : 1));
#endif
        MM_STAMP(0);
        auto issue = [&](const i32x4 (&a)[4], const i32x16 &cin, int nb0, i32x16 (&acc)[2]) {
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (MM_CIN_DEF) acc[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], b[nb0 + u][0], cin, 0, 0, 0);
                else { const i32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; acc[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], b[nb0 + u][0], zero, 0, 0, 0); }
#pragma unroll
                for (int m = 1; m < 4; m++) acc[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b[nb0 + u][m], acc[u], 0, 0, 0);
            }
#ifdef MM_SCHED_BARRIER
            // keep the compiler from sinking this group's MFMAs below the previous group's screen (it interleaved the two groups' chains,
            // put a chain's last MFMA right in front of its screen and padded the result hazard with s_nop 6-10)
            __builtin_amdgcn_sched_barrier(0);
#endif
        };
        auto screen = [&](i32x16 (&acc)[2], int nb0, int j) {
            const int tbase = t_lo + h * half_len + (it * MM_TT + j) * 16;
            if (!MM_CIN_DEF) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const i32x4 v = *reinterpret_cast<const i32x4 *>(&lds_c[cur][j][h * 16 + q * 4]);
#pragma unroll
                    for (int e = 0; e < 4; e++) { acc[0][q * 4 + e] += v[e]; acc[1][q * 4 + e] += v[e]; }
                }
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int nb = nb0 + u;
#ifdef MM_VARIANT_NO_SCREEN                              // tools/ubench: the MFMA feed alone (one vector instruction keeps the chain alive)
                best[nb] ^= acc[u][0] & acc[u][15]; idx[nb] = 0;
                continue;
#endif
                int tmin = acc[u][0];
#pragma unroll
                for (int i = 1; i < 16; i++) tmin = min(tmin, acc[u][i]);
#ifdef MM_VARIANT_NO_UPDATE
                if (tmin < -0x7ffffff0) { best[nb] = tmin; idx[nb] = tbase; }
#else
                if (tmin < thr[nb]) {                    // some lane may improve: ordered update, exactly the reference's scan
#ifdef MM_COUNT_SLOW
                    if (lane == __builtin_ctzll(__ballot(true))) atomicAdd(&mm_slow_count, 1ull);
#endif
                    int at = -1;
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const i32x4 pv = *reinterpret_cast<const i32x4 *>(&lds_n[cur][j][h * 16 + g * 4]);
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int key = (acc[u][g * 4 + e] << 1) + pv[e];           // pv: the parity bit of the target's norm
                            if (key < best[nb]) { second[nb] = best[nb]; best[nb] = key; at = g * 4 + e; }
                        }
                    }
                    if (at >= 0) { idx[nb] = tbase + at; thr[nb] = mm_thr(best[nb]); }
                }
#endif
            }
        };
        i32x16 acc[NSET][2];
        if (MM_PIPE_DEF) {
            if (!XPF) load_frags(cur, 0, af[0], cinf[0]);
            issue(af[0], cinf[0], 0, acc[0]);
#pragma unroll
            for (int g = 0; g < G; g++) {
                const int j = g / PAIRS, pr = g % PAIRS;
                if (pr == 0 && j + 1 < MM_TT) load_frags(cur, j + 1, af[(j + 1) % NSET], cinf[(j + 1) % NSET]);   // a tile ahead of its first MFMA
                // XPF: tile 0's last group was issued at step PAIRS - 2 (or before the loop): its fragment set is free from step PAIRS on
                if (XPF && g == PAIRS && it + 1 < n_iter) load_frags(nxt, 0, af[0], cinf[0]);
                if (g + 1 < G) {
                    const int jn = (g + 1) / PAIRS, prn = (g + 1) % PAIRS;
                    issue(af[jn % NSET], cinf[jn % NSET], 2 * prn, acc[(g + 1) % NSET]);
                }
                screen(acc[g % NSET], 2 * pr, j);
            }
        } else {
#pragma unroll
            for (int g = 0; g < G; g++) {
                const int j = g / PAIRS, pr = g % PAIRS;
                if (pr == 0) load_frags(cur, j, af[0], cinf[0]);
                issue(af[0], cinf[0], 2 * pr, acc[0]);
                screen(acc[0], 2 * pr, j);
            }
        }
        MM_STAMP(1);
#ifndef MM_VARIANT_NO_LOAD
        if (it + (XPF ? 2 : 1) < n_iter) stage(fill);
#endif
        MM_STAMP(2);
#ifndef MM_VARIANT_NO_BARRIER
        __syncthreads();
#endif
        MM_STAMP(3);
    }
#ifdef MM_STAMPS
    if (lane == 0) for (int k = 0; k < 4; k++) atomicAdd(&mm_stamps[k], mm_acc_[k]);
#endif
    // padding rows and empty chunks -> none;  then half 0 (earlier chunk) with half 1
#pragma unroll
    for (int nb = 0; nb < MM_NB; nb++) {
        if (best[nb] >= MM_PAD_LIMIT || idx[nb] < 0) { best[nb] = MM_NONE; idx[nb] = -1; second[nb] = MM_NONE; }   // (idx < 0: never beat its starting bound)
        if (second[nb] >= MM_PAD_LIMIT) second[nb] = MM_NONE;
        const int ob = __shfl_xor(best[nb], 32, 64), oi = __shfl_xor(idx[nb], 32, 64), os = __shfl_xor(second[nb], 32, 64);
        if (h == 0) {
            if (ob < best[nb]) { second[nb] = min(best[nb], os); best[nb] = ob; idx[nb] = oi; }
            const int s = s0 + nb * 32 + c;
            if (s < n_src) {
                int *q = reinterpret_cast<int *>(part + (long long)blockIdx.y * n_src + s);
                if (FUSED) {                                  // every field write-through: the group's last block reads them past its own L2
                    __hip_atomic_store(q + 1, idx[nb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(q + 2, second[nb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    q[1] = idx[nb]; q[2] = second[nb]; q[3] = 0;
                }
                __hip_atomic_store(q, best[nb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // write-through: later splits' blocks read it as a bound
            }
        }
    }
    if constexpr (FUSED) {
        __shared__ int s_last, s_wave_hits[4], s_before[4];
        // This block's records were stored write-through (device-scope atomic stores go past the XCD's L2) and have been acknowledged once
        // vmcnt is 0; then the ticket.  No release fence: at device scope that is a write-back of the whole L2 (buffer_wbl2), tens of
        // microseconds per block, for three dwords per source that never sat in it.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned t = atomicAdd(&tail.ticket[blockIdx.x], 1u);
            s_last = t == gridDim.y - 1;
            if (s_last) tail.ticket[blockIdx.x] = 0u;        // (nobody else touches it before the next call)
        }
        __syncthreads();
        if (!s_last) return;
        // the group's 512 sources: lane (wave wv, column c, half 0) finishes sources s0 + nb 32 + c -- in source order that is (wv, nb, c)
        const int n_split = gridDim.y;
        MatchRec rec[MM_NB];
        unsigned hits = 0;                                   // bit nb: source nb of this lane matched
        int wave_hits = 0, lane_before[MM_NB];
#pragma unroll
        for (int nb = 0; nb < MM_NB; nb++) {
            const int s = s0 + nb * 32 + c;
            const bool live = h == 0 && s < n_src;
            int fb = MM_NONE, fi = -1, fs = MM_NONE;
            if (live) {
                // splits combine in target order (match_finalize_kernel); eight splits' records requested together (device-scope loads:
                // the other blocks' records are read past this XCD's L2, which may hold the lines of an earlier call)
                for (int k0 = 0; k0 < n_split; k0 += 8) {
                    int qb[8], qi[8], qs[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int *q = reinterpret_cast<const int *>(part + (long long)min(k0 + j, n_split - 1) * n_src + s);
                        qb[j] = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        qi[j] = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        qs[j] = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        if (k0 + j < n_split && qb[j] < fb) { fs = min(fb, qs[j]); fb = qb[j]; fi = qi[j]; }
                }
            }
            rec[nb].source = s; rec[nb].target = -1; rec[nb].distance = 0.0f;
            if (fi >= 0) {
                const float bd = sqrtf((float)(fb + src_nrm[nb])) / 255.0f;
                const float sd = (fs == MM_NONE) ? 3.402823466e+38f : sqrtf((float)(fs + src_nrm[nb])) / 255.0f;
                rec[nb].distance = bd;
                if (bd < tail.abs_thr && bd < sd * tail.rel_thr) rec[nb].target = fi;
            }
            if (live) tail.out[s] = rec[nb];
            const bool hit = live && rec[nb].target >= 0;
            const unsigned long long bal = __ballot(hit);
            lane_before[nb] = wave_hits + __popcll(bal & ((1ull << lane) - 1ull));
            wave_hits += __popcll(bal);
            hits |= hit ? (1u << nb) : 0u;
        }
        if (tail.packed == nullptr) return;
        if (lane == 0) s_wave_hits[wv] = wave_hits;
        __syncthreads();
        const int group_hits = s_wave_hits[0] + s_wave_hits[1] + s_wave_hits[2] + s_wave_hits[3];
        if (tid == 0)
            __hip_atomic_store(&tail.status[blockIdx.x], ((unsigned long long)tail.epoch << 32) | (unsigned)group_hits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // matches of the groups before this one: thread t polls group t's count of THIS call (<= 128 groups: one round)
        int before = 0;
        for (int g = tid; g < (int)blockIdx.x; g += 256) {
            unsigned long long v;
            do { v = __hip_atomic_load(&tail.status[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((unsigned)(v >> 32) != tail.epoch);
            before += (int)(unsigned)v;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) before += __shfl_xor(before, o, 64);
        if (lane == 0) s_before[wv] = before;
        __syncthreads();
        int pos = s_before[0] + s_before[1] + s_before[2] + s_before[3];
        for (int k = 0; k < wv; k++) pos += s_wave_hits[k];
#pragma unroll
        for (int nb = 0; nb < MM_NB; nb++)
            if (hits & (1u << nb)) tail.packed[pos + lane_before[nb]] = rec[nb];
        if (blockIdx.x == gridDim.x - 1 && tid == 0) *tail.count = s_before[0] + s_before[1] + s_before[2] + s_before[3] + group_hits;
    }
}

// splits combine in target order; thresholds as SIFTDescriptor.swift:349-355
// block_count (or null): the number of matched sources of every 256-source block, for match_compact_kernel
__global__ __launch_bounds__(256) void match_finalize_kernel(const int4 *__restrict__ part, int n_split, const int *__restrict__ src_norm, int n_src,
                                                            float abs_thr, float rel_thr, MatchRec *__restrict__ out, int32_t *__restrict__ block_count) {
    __shared__ int s_hits;
    if (block_count) { if (threadIdx.x == 0) s_hits = 0; __syncthreads(); }
    const int s = min((int)(blockIdx.x * 256 + threadIdx.x), n_src - 1);        // (lanes past the end repeat the last source and write nothing)
    const bool live = (int)(blockIdx.x * 256 + threadIdx.x) < n_src;
    int best = MM_NONE, idx = -1, second = MM_NONE;
    // eight splits' records requested together, combined in order (one at a time the loop is a chain of memory latencies: 26 us
    // for 20k sources x 10 splits)
    for (int k0 = 0; k0 < n_split; k0 += 8) {
        int4 q[8];
#pragma unroll
        for (int j = 0; j < 8; j++) q[j] = part[(long long)min(k0 + j, n_split - 1) * n_src + s];
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (k0 + j < n_split && q[j].x < best) { second = min(best, q[j].z); best = q[j].x; idx = q[j].y; }
    }
    MatchRec rec; rec.source = s; rec.target = -1; rec.distance = 0.0f;
    if (idx >= 0) {
        const int na = src_norm[s];
        const float bd = sqrtf((float)(best + na)) / 255.0f;
        const float sd = (second == MM_NONE) ? 3.402823466e+38f : sqrtf((float)(second + na)) / 255.0f;
        rec.distance = bd;
        if (bd < abs_thr && bd < sd * rel_thr) rec.target = idx;
    }
    if (live) out[s] = rec;
    if (block_count) {
        const unsigned long long b = __ballot(live && rec.target >= 0);
        if ((threadIdx.x & 63) == 0 && b) atomicAdd(&s_hits, __popcll(b));
        __syncthreads();
        if (threadIdx.x == 0) block_count[blockIdx.x] = s_hits;
    }
}

// Device-resident result (siftmi_match_descriptors_device): the records of the matched sources, in source order, packed into the caller's
// buffer, and their number -- the reference's output array (SIFTDescriptor.swift:304-317 appends in source order) without a host round
// trip.  A block's offset is the sum of the per-block match counts match_finalize_kernel left (n / 256 values), then a ballot scan
// inside the block; the last block writes the total.
// prefixed = 0: every block sums the counts of the blocks before it itself -- b reads for block b, which is nothing at call sizes (a
// 100 k-source call has 390 blocks) but quadratic; from 1024 blocks on the host enqueues match_block_prefix_kernel first (one workgroup,
// exclusive scan in place) and passes prefixed = 1 (the API accepts up to 2^30 sources).
__global__ __launch_bounds__(1024) void match_block_prefix_kernel(int32_t *__restrict__ block_count, int n_blocks) {
    __shared__ int wsum[16], s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n_blocks; i0 += 1024) {
        const int i = i0 + tid;
        const int v = i < n_blocks ? block_count[i] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int before = s_carry + incl - v;
        for (int k = 0; k < wv; k++) before += wsum[k];
        if (i < n_blocks) block_count[i] = before;
        __syncthreads();
        if (tid == 1023) s_carry = before + v;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void match_compact_kernel(const MatchRec *__restrict__ all, int n_src, const int32_t *__restrict__ block_count,
                                                           int prefixed, MatchRec *__restrict__ out, int32_t *__restrict__ count) {
    __shared__ int wsum[4], s_before;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (prefixed) {
        if (tid == 0) s_before = block_count[blockIdx.x];
    } else {
        int before = 0;
        for (int i = tid; i < (int)blockIdx.x; i += 256) before += block_count[i];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) before += __shfl_xor(before, o, 64);
        if (lane == 0) wsum[wv] = before;
        __syncthreads();
        if (tid == 0) s_before = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
    __syncthreads();
    const int s = blockIdx.x * 256 + tid;
    MatchRec rec; rec.source = s; rec.target = -1; rec.distance = 0.0f;
    if (s < n_src) rec = all[s];
    const bool hit = rec.target >= 0;
    const unsigned long long b = __ballot(hit);
    __syncthreads();
    if (lane == 0) wsum[wv] = __popcll(b);
    __syncthreads();
    int pos = s_before + __popcll(b & ((1ull << lane) - 1ull));
    for (int k = 0; k < wv; k++) pos += wsum[k];
    if (hit) out[pos] = rec;
    if (blockIdx.x == gridDim.x - 1 && tid == 0) *count = s_before + wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

}  // namespace siftmi
