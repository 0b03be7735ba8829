// Microbenchmark: what each instruction KIND of the descriptor / orientation sample loops costs the vector pipe of one SIMD on gfx950
// (wall time per wave-instruction per SIMD at 8 resident wavefronts, 16 independent destinations).  Round 6: the loops are bound by
// vector issue (profiles/pmc_descriptor_dense_r05_final.txt) and their instruction mix averages ~4 cycles per instruction although an
// all-VGPR v_fma_f32 issues every ~2.7 -- which operand forms and opcodes are the slow ones?
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_mix ubench_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Mode { const char *name; };
static const Mode MODES[] = {
    {"v_fma_f32 d,x,y,d (all VGPR)"},          // 0
    {"v_fmaak_f32 d,d,x,LITERAL"},             // 1
    {"v_fmamk_f32 d,d,LITERAL,x"},             // 2
    {"v_mul_f32 d,LITERAL,x"},                 // 3
    {"v_add_f32 d,LITERAL,x"},                 // 4
    {"v_mul_f32 d,sgpr,x"},                    // 5
    {"v_cndmask_b32 d,x,y,vcc"},               // 6
    {"v_cmp_gt_f32 vcc,x,d"},                  // 7
    {"v_cmp_lt_f32_e64 s[],|d|,sgpr"},         // 8
    {"v_cvt_u32_f32 d,x"},                     // 9
    {"v_cvt_flr_i32_f32 d,x"},                 // 10
    {"v_fract_f32 d,x"},                       // 11
    {"v_cvt_f32_i32 d,x"},                     // 12
    {"v_exp_f32 d,x"},                         // 13
    {"v_rcp_f32 d,x"},                         // 14
    {"v_sqrt_f32 d,x"},                        // 15
    {"v_max3_f32 d,|x|,|y|,sgpr"},             // 16
    {"v_min_f32_e64 d,|x|,|y|"},               // 17
    {"v_sub_f32 d,2.0,x (inline const)"},      // 18
    {"v_lshl_add_u32 d,x,2,y"},                // 19
    {"v_and_or_b32 d,x,sgpr,y"},               // 20
    {"v_add_u32 d,x,y"},                       // 21
    {"v_mad_i32_i24 d,x,sgpr,y"},              // 22
    {"ds_add_u64 (8 B per lane, distinct)"},   // 23
    {"v_fma + s_and_saveexec/s_or pair"},      // 24
    {"v_mul_f32 d,x,y"},                       // 25
    {"v_fma_f32 d,x,y,z (dst not a source)"},  // 26
    {"v_fmac_f32 d,x,y"},                      // 27
    {"v_fma_f32 d,|x|,-y,d (modifiers)"},      // 28
    {"v_mul_f32_e64 d,x,-y (VOP3 neg)"},       // 29
    {"v_cvt_i32_f32 d,x"},                     // 30
    {"v_floor_f32 d,x"},                       // 31
    {"v_rndne_f32 d,x"},                       // 32
    {"v_mul_u32_u24 d,x,y"},                   // 33
    {"v_bfe_u32 d,x,4,8"},                     // 34
    {"v_mov_b32 d,x"},                         // 35
    {"v_max_f32 d,x,y"},                       // 36
    {"ds_add_u32 (4 B per lane, distinct)"},   // 37
    {"v_rsq_f32 d,x"},                         // 38
    {"v_log_f32 d,x"},                         // 39
    {"v_ldexp_f32 d,x,y"},                     // 40
    {"v_med3_f32 d,x,y,z"},                    // 41
    {"v_perm_b32 d,x,y,z"},                    // 42
    {"v_cvt_pk_u16_u32? v_pack_b32_f16 d,x,y"},// 43
    {"v_mad_u32_u24 d,x,y,z"},                 // 44
    {"v_add3_u32 d,x,y,z"},                    // 45
    {"v_mul_lo_u32 d,x,y"},                    // 46
    {"v_mul_hi_u32_u24 d,x,y"},                // 47
    {"v_cmp_gt_u32 vcc,4,d"},                  // 48
    {"v_cndmask_b32_e64 d,x,y,s[]"},           // 49
    {"PAIR v_cmp_gt_f32 vcc + v_cndmask vcc"}, // 50 (time per PAIR)
    {"PAIR v_cmp_e64 s[] + v_cndmask_e64 s[]"},// 51 (time per PAIR)
    {"v_cndmask_b32_e64 d,x,y,vcc"},           // 52
    {"4 v_fma + saveexec/cbranch/s_or (per 4)"},// 53 (time per group of 4 v_fma + 3 SALU)
    {"v_fma_f32 d,x,y,0.5"},                   // 54
    {"v_lshlrev_b32 d,2,x"},                   // 55
    {"v_and_b32 d,x,y"},                       // 56
    {"v_bfi_b32 d,x,y,z"},                     // 57
    {"v_cndmask_b32 vcc, vcc written each 16"},// 58
    {"v_cndmask_b32_dpp? v_mov_b32_dpp shr1"}, // 59
    {"v_mul_f32 normal x tiny -> DENORMAL out"}, // 60
    {"v_mul_f32 DENORMAL in x normal"},        // 61
    {"v_sub_f32 tiny - tiny (denormal range)"},// 62
    {"v_fma_f32 denormal product + 0"},        // 63
};
constexpr int NMODES = sizeof(MODES) / sizeof(MODES[0]);

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float sc) {
    __shared__ unsigned long long lds[2048];
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 0.001f + i;
    float x = out[threadIdx.x & 7] + 0.5f, y = x + 1.0f, z = x + 2.0f;
    const float s = __builtin_amdgcn_readfirstlane(sc);
    const float tiny = (x + threadIdx.x * 0.01f) * 1.1754944e-38f * 0.4f, den = tiny * 0.001f;   // 0.2 ... 1.3 x 2^-126 (denormal / first binade), and far below
    unsigned long long m = 0, mm = ~0ull;
    const unsigned addr = threadIdx.x * 8;
    unsigned long long v64 = threadIdx.x;
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = 0;
    __syncthreads();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
            if (MODE == 1) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3d5674a8" : "+v"(a[i]) : "v"(x));
            if (MODE == 2) asm volatile("v_fmamk_f32 %0, %0, 0x3d5674a8, %1" : "+v"(a[i]) : "v"(x));
            if (MODE == 3) asm volatile("v_mul_f32 %0, 0xbe38aa3b, %1" : "=v"(a[i]) : "v"(x));
            if (MODE == 4) asm volatile("v_add_f32 %0, 0x3fc00001, %1" : "=v"(a[i]) : "v"(x));
            if (MODE == 5) asm volatile("v_mul_f32 %0, %2, %1" : "=v"(a[i]) : "v"(x), "s"(s));
            if (MODE == 6) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(x), "v"(y) : "vcc");
            if (MODE == 7) asm volatile("v_cmp_gt_f32 vcc, %1, %0" : "+v"(a[i]) : "v"(x) : "vcc");
            if (MODE == 8) asm volatile("v_cmp_lt_f32_e64 %0, |%1|, %2" : "=s"(m) : "v"(a[i]), "s"(s));
            if (MODE == 9) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(a[i]) : "v"(x));
            if (MODE == 10) asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(a[i]) : "v"(x));
            if (MODE == 11) asm volatile("v_fract_f32 %0, %1" : "=v"(a[i]) : "v"(x));
            if (MODE == 12) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[i]) : "v"(x));
            if (MODE == 13) asm volatile("v_exp_f32 %0, %1" : "=v"(a[i]) : "v"(x));
            if (MODE == 14) asm volatile("v_rcp_f32 %0, %1" : "=v"(a[i]) : "v"(x));
            if (MODE == 15) asm volatile("v_sqrt_f32 %0, %1" : "=v"(a[i]) : "v"(x));
            if (MODE == 16) asm volatile("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(a[i]) : "v"(x), "v"(y), "s"(s));
            if (MODE == 17) asm volatile("v_min_f32_e64 %0, |%1|, |%2|" : "=v"(a[i]) : "v"(x), "v"(y));
            if (MODE == 18) asm volatile("v_sub_f32 %0, 2.0, %1" : "=v"(a[i]) : "v"(x));
            if (MODE == 19) asm volatile("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(a[i]) : "v"(x), "v"(y));
            if (MODE == 20) asm volatile("v_and_or_b32 %0, %1, %3, %2" : "=v"(a[i]) : "v"(x), "v"(y), "s"(s));
            if (MODE == 21) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(y));
            if (MODE == 22) asm volatile("v_mad_i32_i24 %0, %1, %3, %2" : "=v"(a[i]) : "v"(x), "v"(y), "s"(s));
            if (MODE == 23) asm volatile("ds_add_u64 %0, %1 offset:0" ::"v"(addr + (i & 7) * 2048), "v"(v64) : "memory");
            if (MODE == 24) {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                if ((i & 3) == 3) asm volatile("s_and_saveexec_b64 %0, %1\n\ts_or_b64 exec, exec, %0" : "=&s"(m) : "s"(mm) : "scc");
            }
            if (MODE == 50) asm volatile("v_cmp_gt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %1, %2, vcc" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
            if (MODE == 51) asm volatile("v_cmp_gt_f32_e64 %3, %1, %0\n\tv_cndmask_b32_e64 %0, %1, %2, %3" : "+v"(a[i]) : "v"(x), "v"(y), "s"(mm));
            if (MODE == 52) asm volatile("v_cndmask_b32_e64 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(x), "v"(y) : "vcc");
            if (MODE == 53) {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                if ((i & 3) == 3) asm volatile("s_and_saveexec_b64 %0, %1\n\ts_cbranch_execz 0\n\ts_or_b64 exec, exec, %0" : "=&s"(m) : "s"(mm) : "scc");
            }
            if (MODE == 54) asm volatile("v_fma_f32 %0, %1, %2, 0.5" : "=v"(a[i]) : "v"(x), "v"(y));
            if (MODE == 55) asm volatile("v_lshlrev_b32 %0, 2, %1" : "=v"(a[i]) : "v"(x));
            if (MODE == 56) asm volatile("v_and_b32 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(y));
            if (MODE == 57) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(x), "v"(y), "v"(z));
            if (MODE == 58) {
                if (i == 0) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(x), "v"(y) : "vcc");
                asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(x), "v"(y) : "vcc");
            }
            if (MODE == 60) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(tiny));
            if (MODE == 61) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(den), "v"(x));
            if (MODE == 62) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[i]) : "v"(tiny), "v"(den));
            if (MODE == 63) asm volatile("v_fma_f32 %0, %1, %2, 0" : "=v"(a[i]) : "v"(x), "v"(tiny));
            if (MODE == 59) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(x));
            if (MODE == 25) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(y));
            if (MODE == 26) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(x), "v"(y), "v"(z));
            if (MODE == 27) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
            if (MODE == 28) asm volatile("v_fma_f32 %0, |%1|, -%2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
            if (MODE == 29) asm volatile("v_mul_f32_e64 %0, %1, -%2" : "=v"(a[i]) : "v"(x), "v"(y));
            if (MODE == 30) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(a[i]) : "v"(x));
            if (MODE == 31) asm volatile("v_floor_f32 %0, %1" : "=v"(a[i]) : "v"(x));
            if (MODE == 32) asm volatile("v_rndne_f32 %0, %1" : "=v"(a[i]) : "v"(x));
            if (MODE == 33) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(y));
            if (MODE == 34) asm volatile("v_bfe_u32 %0, %1, 4, 8" : "=v"(a[i]) : "v"(x));
            if (MODE == 35) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(x));
            if (MODE == 36) asm volatile("v_max_f32 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(y));
            if (MODE == 37) asm volatile("ds_add_u32 %0, %1 offset:0" ::"v"(addr / 2 + (i & 7) * 1024), "v"(x) : "memory");
            if (MODE == 38) asm volatile("v_rsq_f32 %0, %1" : "=v"(a[i]) : "v"(x));
            if (MODE == 39) asm volatile("v_log_f32 %0, %1" : "=v"(a[i]) : "v"(x));
            if (MODE == 40) asm volatile("v_ldexp_f32 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(y));
            if (MODE == 41) asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(x), "v"(y), "v"(z));
            if (MODE == 42) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(x), "v"(y), "v"(z));
            if (MODE == 43) asm volatile("v_pack_b32_f16 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(y));
            if (MODE == 44) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(a[i]) : "v"(x), "v"(y), "v"(z));
            if (MODE == 45) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(x), "v"(y), "v"(z));
            if (MODE == 46) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(y));
            if (MODE == 47) asm volatile("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(y));
            if (MODE == 48) asm volatile("v_cmp_gt_u32 vcc, 4, %0" : "+v"(a[i]) : : "vcc");
            if (MODE == 49) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a[i]) : "v"(x), "v"(y), "s"(mm));
        }
        if (MODE == 23 || MODE == 37) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float r = (float)(m & 1) + (float)lds[threadIdx.x];
#pragma unroll
    for (int i = 0; i < 16; i++) r += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int M>
static void launch_mode(int mode, int grid, float *d, int iters) {
    if constexpr (M < NMODES) {
        if (mode == M) { hipLaunchKernelGGL(k<M>, dim3(grid), dim3(256), 0, 0, d, iters, 1.5f); return; }
        launch_mode<M + 1>(mode, grid, d, iters);
    }
}

int main(int argc, char **argv) {
    float *d; CHECK(hipMalloc(&d, 8192 * 256 * 4)); CHECK(hipMemset(d, 0, 8192 * 256 * 4));
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int iters = 2048;
    const int m0 = argc > 1 ? atoi(argv[1]) : 0, m1 = argc > 2 ? atoi(argv[2]) : NMODES;
    for (int wps : {8, 4}) {
        const int grid = 256 * wps;
        double base = 0;
        for (int mode = m0; mode < m1; mode++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                launch_mode<0>(mode, grid, d, iters);
                hipEventRecord(e1);
                CHECK(hipDeviceSynchronize());
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                best = std::min(best, ms);
            }
            const double n = (double)iters * 16;
            const double ns = best * 1e6 / (n * wps);
            if (mode == m0) base = ns;
            printf("waves/SIMD %d  %-42s %7.3f ms  %6.2f ns per wave-instruction per SIMD  = %5.2f x v_fma\n", wps, MODES[mode].name, best, ns, ns / base);
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
    }
    return 0;
}
