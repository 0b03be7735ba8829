// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access shapes of blur_ring_kernel (MI355X_MICROARCH.md, HBM
// section: on gfx950 FETCH_SIZE counts half the bytes of a wide coalesced streaming read and WRITE_SIZE is exact for 16-B
// stores; "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// This kernel moves a KNOWN number of bytes with exactly the ring kernel's shapes and nothing else:
//   loads : 8 lanes per row, each lane float4 columns q + 8 j (global_load_dwordx4; a wave instruction = 8 rows x 128 B)
//   stores: one row per wave instruction, float2 per lane (global_store_dwordx2, 512 B per instruction)
// over 128-column strips of 32-row steps, no halo: 4 B read + 4 B written per pixel.  tools/profile_round.sh runs it under
// --pmc FETCH_SIZE and --pmc WRITE_SIZE and derives the two correction factors it then applies to the blur kernel.
// build: hipcc --offload-arch=gfx950 -O3 -o pmc_calib pmc_calib.hip ; run: ./pmc_calib [w h frames iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void calib_copy_kernel(const float *__restrict__ src, float *__restrict__ dst, int w, int h, int n_frames) {
    __shared__ __attribute__((aligned(16))) float tile[32 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tx = w / 128, ty = h / 32;
    const int t = blockIdx.x;
    if (t >= tx * ty * n_frames) return;
    const int frame = t / (tx * ty), rem = t - frame * (tx * ty), by = rem / tx, bx = rem - by * tx;
    const float *in = src + (size_t)frame * w * h + (size_t)(by * 32) * w + bx * 128;
    float *out = dst + (size_t)frame * w * h + (size_t)(by * 32) * w + bx * 128;
    const int prow = tid >> 3, pq = tid & 7;
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = *reinterpret_cast<const f32x4 *>(in + (size_t)prow * w + 4 * pq + 32 * j);
#pragma unroll
    for (int j = 0; j < 4; j++) *reinterpret_cast<f32x4 *>(tile + prow * 128 + 4 * pq + 32 * j) = v[j];
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 8; rr++) {
        const int row = wv * 8 + rr;
        *reinterpret_cast<f32x2 *>(out + (size_t)row * w + 2 * lane) = *reinterpret_cast<const f32x2 *>(tile + row * 128 + 2 * lane);
    }
}

int main(int argc, char **argv) {
    const int w = argc > 1 ? atoi(argv[1]) : 3840, h = argc > 2 ? atoi(argv[2]) : 2160, nf = argc > 3 ? atoi(argv[3]) : 8;
    const int iters = argc > 4 ? atoi(argv[4]) : 5;
    if (w % 128 || h % 32) { printf("w must be a multiple of 128, h of 32\n"); return 1; }
    const size_t n = (size_t)w * h * nf;
    float *src, *dst;
    CHECK(hipMalloc(&src, n * 4)); CHECK(hipMalloc(&dst, n * 4));
    CHECK(hipMemset(src, 0x3c, n * 4));
    const int total = (w / 128) * (h / 32) * nf;
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(calib_copy_kernel, dim3(total), dim3(256), 0, 0, src, dst, w, h, nf);
    CHECK(hipDeviceSynchronize());
    printf("calib_copy_kernel: %d launches, %zu bytes read and %zu bytes written per launch\n", iters, n * 4, n * 4);
    return 0;
}
