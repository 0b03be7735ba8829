/*
 * sift_oracle.c -- CPU restatement (plain C99) of the lukevanin/SIFTMetal detect+describe path.
 *
 * TEST INFRASTRUCTURE ONLY (see sift_oracle.h for the rules and for how the restatement is
 * pinned against the reference's own golden fixtures).
 *
 * Conventions
 *  - every image is planar float32, row-major [h][w]
 *  - a texture read outside the image returns 0 (Metal texture reads out of bounds return 0
 *    for r32Float/rg32Float; negative coordinates cast to ushort become >= 32768 and are
 *    therefore out of bounds as well)
 *  - all arithmetic is single precision, built with -ffp-contract=off; the only fused
 *    operation is the explicit fmaf() in the blur tap loop (switchable, so_config.use_fma)
 *  - MSL unsuffixed literals are single precision -> every literal here carries an f suffix
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include "sift_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SO_PI_F 3.14159265358979323846264338327950288f   /* M_PI_F */

typedef struct {
    int   w, h;
    float delta;
    float sigmas[16];
    float *gauss;    /* [nspo+3][h][w] */
    float *dog;      /* [nspo+2][h][w] */
    float *work;     /* [h][w] X-pass scratch (GaussianSeriesKernel.swift:75-87 workingTexture) */
} so_octave;

struct so_ctx {
    so_config cfg;
    int   seed_w, seed_h;
    float *gray;      /* [H][W]   luminosityTexture  */
    float *scaled;    /* [2H][2W] scaledTexture      */
    float *seed_work; /* [2H][2W]                    */
    float *seed;      /* [2H][2W] seedTexture        */
    int   seed_taps;
    float seed_weights[SO_MAX_WEIGHTS];
    int   taps[16];                       /* per layer 1..nspo+2 (index s-1)            */
    float weights[16][SO_MAX_WEIGHTS];
    so_octave *oct;
};

int so_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* n >= 1: number of OpenMP threads used by later calls (bench.py times the port on 1 thread and on all cores) */
void so_set_num_threads(int n) {
#ifdef _OPENMP
    if (n >= 1) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ---------------------------------------------------------------------------------------------
 * Sources/MetalShaders/Metal/Common.hpp:15-22  symmetrizedCoordinates
 * (C '%' truncates like MSL's, so i can stay negative for |i| > 2l: the read is then OOB -> 0) */
static inline int symmetrized(int i, int l) {
    int ll = 2 * l;
    i = (i + ll) % ll;
    if (i > l - 1) i = ll - 1 - i;
    return i;
}

static inline float tex_read(const float *img, int w, int h, int x, int y) {
    if (x < 0 || y < 0 || x >= w || y >= h) return 0.0f;
    return img[(size_t)y * w + x];
}

/* ---------------------------------------------------------------------------------------------
 * Sources/SIFTMetal/Metal Compute/GaussianKernel.swift:20-43 and GaussianSeriesKernel.swift:27-51
 * radius = ceil(4 s); w_k = exp(-0.5 (k*k / s^2)); normalised by the running f32 sum          */
static int gaussian_weights(float s, float *weights) {
    int radius = (int)ceilf(4.0f * s);
    int size = radius * 2 + 1;
    if (size > SO_MAX_WEIGHTS) return -1;
    float t = 0.0f;
    float ss = s * s;
    for (int k = -radius, i = 0; k <= radius; k++, i++) {
        float kk = (float)(k * k);
        float w = expf(-0.5f * (kk / ss));
        weights[i] = w;
        t += w;
    }
    for (int i = 0; i < size; i++) weights[i] = weights[i] / t;
    return size;
}

/* ---------------------------------------------------------------------------------------------
 * Sources/SIFTMetal/SIFT/DifferenceOfGaussians.swift:233-344 (schedule) and :69-147 (Octave.init) */
so_ctx *so_create(const so_config *cfg) {
    if (!cfg || cfg->width < 1 || cfg->height < 1 || cfg->n_octaves < 1 || cfg->n_octaves > 16 ||
        cfg->nspo < 1 || cfg->nspo > 8)
        return NULL;
    so_ctx *c = (so_ctx *)calloc(1, sizeof(so_ctx));
    c->cfg = *cfg;
    const float sigmaMinimum = 0.8f, deltaMinimum = 0.5f, sigmaInput = 0.5f;  /* :28-37 */
    const int W = cfg->width, H = cfg->height, nspo = cfg->nspo;

    c->seed_w = (int)((float)W / deltaMinimum);                 /* :235-238 */
    c->seed_h = (int)((float)H / deltaMinimum);
    {   /* :255-262 seed blur sigma */
        float i = sigmaMinimum * sigmaMinimum;
        float j = sigmaInput * sigmaInput;
        float k = sqrtf(i - j) / deltaMinimum;
        c->seed_taps = gaussian_weights(k, c->seed_weights);
    }
    c->gray      = (float *)malloc(sizeof(float) * (size_t)W * H);
    c->scaled    = (float *)malloc(sizeof(float) * (size_t)c->seed_w * c->seed_h);
    c->seed_work = (float *)malloc(sizeof(float) * (size_t)c->seed_w * c->seed_h);
    c->seed      = (float *)malloc(sizeof(float) * (size_t)c->seed_w * c->seed_h);

    c->oct = (so_octave *)calloc((size_t)cfg->n_octaves, sizeof(so_octave));
    for (int o = 0; o < cfg->n_octaves; o++) {                  /* :315-343 */
        so_octave *q = &c->oct[o];
        float delta = deltaMinimum * powf(2.0f, (float)o);
        q->delta = delta;
        q->w = (int)((float)W / delta);
        q->h = (int)((float)H / delta);
        for (int s = 0; s < nspo + 3; s++) {
            float hh = delta / deltaMinimum;
            float i = (float)s / (float)nspo;
            float j = powf(2.0f, i);
            q->sigmas[s] = hh * sigmaMinimum * j;
        }
        if (o == 0) {
            /* Octave.init :91-108 -- rho is identical in every octave (sigma and delta both scale
               by 2^o), but it is recomputed per octave in the reference; octave 0's values are
               representative to the last bit only if the float ops agree, so we keep octave 0's
               and assert equality for the others in the tests. */
            for (int s = 1; s < nspo + 3; s++) {
                float sa = q->sigmas[s - 1], sb = q->sigmas[s];
                float rho = sqrtf(sb * sb - sa * sa) / delta;
                c->taps[s - 1] = gaussian_weights(rho, c->weights[s - 1]);
            }
        }
        size_t n = (size_t)(q->w > 0 ? q->w : 0) * (size_t)(q->h > 0 ? q->h : 0);
        q->gauss = (float *)calloc(n * (size_t)(nspo + 3) + 1, sizeof(float));
        q->dog   = (float *)calloc(n * (size_t)(nspo + 2) + 1, sizeof(float));
        q->work  = (float *)calloc(n + 1, sizeof(float));
    }
    return c;
}

void so_destroy(so_ctx *c) {
    if (!c) return;
    for (int o = 0; o < c->cfg.n_octaves; o++) {
        free(c->oct[o].gauss); free(c->oct[o].dog); free(c->oct[o].work);
    }
    free(c->oct); free(c->gray); free(c->scaled); free(c->seed_work); free(c->seed);
    free(c);
}

int   so_octave_width(const so_ctx *c, int o)  { return c->oct[o].w; }
int   so_octave_height(const so_ctx *c, int o) { return c->oct[o].h; }
float so_octave_delta(const so_ctx *c, int o)  { return c->oct[o].delta; }
float so_octave_sigma(const so_ctx *c, int o, int s) { return c->oct[o].sigmas[s]; }
int so_blur_taps(const so_ctx *c, int layer, float *weights_out) {
    int n = layer == 0 ? c->seed_taps : c->taps[layer - 1];
    const float *w = layer == 0 ? c->seed_weights : c->weights[layer - 1];
    if (weights_out) memcpy(weights_out, w, sizeof(float) * (size_t)n);
    return n;
}
const float *so_gaussian(const so_ctx *c, int o, int s) {
    const so_octave *q = &c->oct[o];
    return q->gauss + (size_t)s * q->w * q->h;
}
const float *so_dog(const so_ctx *c, int o, int s) {
    const so_octave *q = &c->oct[o];
    return q->dog + (size_t)s * q->w * q->h;
}
const float *so_seed(const so_ctx *c) { return c->seed; }

/* ---------------------------------------------------------------------------------------------
 * Sources/MetalShaders/Metal/Convolution.metal:15-52 and ConvolutionSeries.metal:16-53
 * 1-D correlation, taps i = 0..n-1 in that order, o = gid - n/2, mirror extension.            */
static void convolve_x(const float *in, float *out, int w, int h, const float *wt, int n, int use_fma) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            float sum = 0.0f;
            int o = x - (n / 2);
            for (int i = 0; i < n; i++) {
                int xx = symmetrized(o + i, w);
                float cpx = tex_read(in, w, h, xx, y);
                sum = use_fma ? fmaf(wt[i], cpx, sum) : sum + wt[i] * cpx;
            }
            out[(size_t)y * w + x] = sum;
        }
    }
}

static void convolve_y(const float *in, float *out, int w, int h, const float *wt, int n, int use_fma) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        int o = y - (n / 2);
        for (int x = 0; x < w; x++) {
            float sum = 0.0f;
            for (int i = 0; i < n; i++) {
                int yy = symmetrized(o + i, h);
                float cpx = tex_read(in, w, h, x, yy);
                sum = use_fma ? fmaf(wt[i], cpx, sum) : sum + wt[i] * cpx;
            }
            out[(size_t)y * w + x] = sum;
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * Sources/MetalShaders/Metal/ConvertSRGBToGrayscale.metal:11-23 (BT.709 luma on gamma values;
 * bgra8Unorm texel -> float = byte / 255)                                                      */
static void to_gray(so_ctx *c, const void *pixels, int format, int stride) {
    const int W = c->cfg.width, H = c->cfg.height;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        const unsigned char *row = (const unsigned char *)pixels + (size_t)y * stride;
        float *dst = c->gray + (size_t)y * W;
        for (int x = 0; x < W; x++) {
            if (format == SO_FMT_BGRA8) {
                float b = (float)row[4 * x + 0] / 255.0f;
                float g = (float)row[4 * x + 1] / 255.0f;
                float r = (float)row[4 * x + 2] / 255.0f;
                float i = 0.0f + (0.212639005871510f * r) + (0.715168678767756f * g) +
                          (0.072192315360734f * b);
                dst[x] = i;
            } else if (format == SO_FMT_GRAY8) {
                dst[x] = (float)row[x] / 255.0f;     /* added entry: luma step skipped */
            } else {
                dst[x] = ((const float *)row)[x];    /* added entry: already f32 luma  */
            }
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * Sources/MetalShaders/Metal/BilinearUpScale.metal:12-64                                       */
static void bilinear_upscale(const float *in, int wi, int hi, float *out, int wo, int ho) {
    const float dx = (float)wi / (float)wo;
    const float dy = (float)hi / (float)ho;
#pragma omp parallel for schedule(static)
    for (int j = 0; j < ho; j++) {
        for (int i = 0; i < wo; i++) {
            const float x = (float)i * dx;
            const float y = (float)j * dy;
            int im = (int)x, jm = (int)y;
            int ip = im + 1, jp = jm + 1;
            if (ip >= wi) ip = 2 * wi - 1 - ip;
            if (im >= wi) im = 2 * wi - 1 - im;
            if (jp >= hi) jp = 2 * hi - 1 - jp;
            if (jm >= hi) jm = 2 * hi - 1 - jm;
            const float fx = x - floorf(x);
            const float fy = y - floorf(y);
            const float c0 = tex_read(in, wi, hi, ip, jp);
            const float c1 = tex_read(in, wi, hi, ip, jm);
            const float c2 = tex_read(in, wi, hi, im, jp);
            const float c3 = tex_read(in, wi, hi, im, jm);
            const float o = fx * (fy * c0 + (1.0f - fy) * c1) +
                            (1.0f - fx) * (fy * c2 + (1.0f - fy) * c3);
            out[(size_t)j * wo + i] = o;
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * DifferenceOfGaussians.encode (DifferenceOfGaussians.swift:346-406):
 *   seed  = blur(bilinear2x(gray))                                  (:357-389)
 *   G[o][0] = seed (o = 0)  or  G[o-1][nspo][2y][2x]                 (:159-201, NearestNeighborDownScale.metal:21)
 *   G[o][s] = blur_rho_s(G[o][s-1]), X pass then Y pass              (GaussianSeriesKernel.swift:92-119)
 *   D[o][s] = G[o][s+1] - G[o][s]                                    (Subtract.metal:12-21)            */
void so_build_pyramid(so_ctx *c, const void *pixels, int format, int stride) {
    const int nspo = c->cfg.nspo, fma = c->cfg.use_fma;
    to_gray(c, pixels, format, stride);
    bilinear_upscale(c->gray, c->cfg.width, c->cfg.height, c->scaled, c->seed_w, c->seed_h);
    convolve_x(c->scaled, c->seed_work, c->seed_w, c->seed_h, c->seed_weights, c->seed_taps, fma);
    convolve_y(c->seed_work, c->seed, c->seed_w, c->seed_h, c->seed_weights, c->seed_taps, fma);

    for (int o = 0; o < c->cfg.n_octaves; o++) {
        so_octave *q = &c->oct[o];
        const size_t n = (size_t)q->w * q->h;
        if (n == 0) continue;
        if (o == 0) {
            memcpy(q->gauss, c->seed, n * sizeof(float));
        } else {
            const so_octave *p = &c->oct[o - 1];
            const float *src = p->gauss + (size_t)nspo * p->w * p->h;
            for (int y = 0; y < q->h; y++)
                for (int x = 0; x < q->w; x++)
                    q->gauss[(size_t)y * q->w + x] = tex_read(src, p->w, p->h, 2 * x, 2 * y);
        }
        for (int s = 1; s < nspo + 3; s++) {
            convolve_x(q->gauss + (size_t)(s - 1) * n, q->work, q->w, q->h, c->weights[s - 1], c->taps[s - 1], fma);
            convolve_y(q->work, q->gauss + (size_t)s * n, q->w, q->h, c->weights[s - 1], c->taps[s - 1], fma);
        }
        for (int s = 0; s < nspo + 2; s++) {
            const float *a = q->gauss + (size_t)(s + 1) * n, *b = q->gauss + (size_t)s * n;
            float *d = q->dog + (size_t)s * n;
#pragma omp parallel for schedule(static)
            for (long i = 0; i < (long)n; i++) d[i] = a[i] - b[i];
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * Sources/MetalShaders/Metal/SIFTExtrema.metal:62-110  siftExtremaList
 * grid (w-2, h-2, D-2); g = gid.xy + 1, s = gid.z + 1; neighbours 1..25 of the 26-entry table
 * (:15-45; entry 0 = (-1,-1,-1) is never tested); strict < min or > max; sentinels +-1000.
 * Output order here is the deterministic scan order (s, y, x); the reference's is arbitrary.   */
/* one candidate test; semantically identical to the reference's min/max-over-25 formulation:
   value < min(all)  <=>  value < every neighbour;  value > max(all)  <=>  value > every one.
   (neighbour (+1,0,0) decides which of the two can still hold, then the loop exits early.)   */
static inline int is_extremum(const float *dog, size_t n, int w, int x, int y, int s, int full) {
    const float value = dog[(size_t)s * n + (size_t)y * w + x];
    const float right = dog[(size_t)s * n + (size_t)y * w + x + 1];
    if (value == right) return 0;
    const int want_min = value < right;
    if (want_min ? !(value < 1000.0f) : !(value > -1000.0f)) return 0;   /* sentinels :81-82 */
    int first = 1;
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                if (dx == 0 && dy == 0 && dz == 0) continue;
                if (first) {               /* table entry 0 = (-1,-1,-1), skipped by the loop at :84 */
                    first = 0;
                    if (!full) continue;
                }
                const float v = dog[(size_t)(s + dz) * n + (size_t)(y + dy) * w + (x + dx)];
                if (want_min ? !(value < v) : !(value > v)) return 0;
            }
    return 1;
}

int so_extrema(const so_ctx *c, int o, so_extremum *out, int cap) {
    const so_octave *q = &c->oct[o];
    const int w = q->w, h = q->h, D = c->cfg.nspo + 2;
    const size_t n = (size_t)w * h;
    const int full = c->cfg.full_neighbourhood;
    const int rows = (h - 2 > 0 && w - 2 > 0) ? (D - 2) * (h - 2) : 0;   /* (s, y) pairs in scan order */
    if (rows <= 0) return 0;
    int nthreads = so_num_threads();
    if (nthreads > rows) nthreads = rows;
    so_extremum **lists = (so_extremum **)calloc((size_t)nthreads, sizeof(*lists));
    int *counts = (int *)calloc((size_t)nthreads, sizeof(int));
#pragma omp parallel num_threads(nthreads)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num();
#else
        const int t = 0;
#endif
        const int r0 = (int)((long)rows * t / nthreads), r1 = (int)((long)rows * (t + 1) / nthreads);
        int capl = 1024, cnt = 0;
        so_extremum *l = (so_extremum *)malloc(sizeof(so_extremum) * (size_t)capl);
        for (int r = r0; r < r1; r++) {
            const int s = 1 + r / (h - 2), y = 1 + r % (h - 2);
            for (int x = 1; x <= w - 2; x++) {
                if (!is_extremum(q->dog, n, w, x, y, s, full)) continue;
                if (cnt == capl) { capl *= 2; l = (so_extremum *)realloc(l, sizeof(so_extremum) * (size_t)capl); }
                l[cnt].x = x; l[cnt].y = y; l[cnt].scale = s; cnt++;
            }
        }
        lists[t] = l; counts[t] = cnt;
    }
    int count = 0;
    for (int t = 0; t < nthreads; t++) {
        for (int i = 0; i < counts[t]; i++, count++)
            if (out && count < cap) out[count] = lists[t][i];
        free(lists[t]);
    }
    free(lists); free(counts);
    return count;
}

/* ---------------------------------------------------------------------------------------------
 * Sources/MetalShaders/Metal/SIFTInterpolate.metal                                             */
typedef struct { const float *d; int w, h, ns; size_t n; } dogtex;
static inline float dread(const dogtex *t, int x, int y, int s) {
    if (x < 0 || y < 0 || s < 0 || x >= t->w || y >= t->h || s >= t->ns) return 0.0f;
    return t->d[(size_t)s * t->n + (size_t)y * t->w + x];
}

/* :17-61 isOnEdge */
static int is_on_edge(const dogtex *t, int x, int y, int s, float edgeThreshold) {
    const float v  = dread(t, x, y, s);
    const float zn = dread(t, x, y - 1, s);
    const float zp = dread(t, x, y + 1, s);
    const float pz = dread(t, x + 1, y, s);
    const float nz = dread(t, x - 1, y, s);
    const float pp = dread(t, x + 1, y + 1, s);
    const float np = dread(t, x - 1, y + 1, s);
    const float pn = dread(t, x + 1, y - 1, s);
    const float nn = dread(t, x - 1, y - 1, s);
    const float hxx = zn + zp - 2.0f * v;
    const float hyy = pz + nz - 2.0f * v;
    const float hxy = ((pp - np) - (pn - nn)) * 0.25f;
    const float trace = hxx + hyy;
    const float determinant = (hxx * hyy) - (hxy * hxy);
    if (determinant <= 0.0f) return 1;
    const float threshold = ((edgeThreshold + 1.0f) * (edgeThreshold + 1.0f)) / edgeThreshold;
    const float curvature = (trace * trace) / determinant;
    if (curvature >= threshold) return 1;
    return 0;
}

/* :64-87 derivatives3D */
static void derivatives3d(const dogtex *t, int x, int y, int s, float dD[3]) {
    const float pzz = dread(t, x + 1, y, s), nzz = dread(t, x - 1, y, s);
    const float zpz = dread(t, x, y + 1, s), znz = dread(t, x, y - 1, s);
    const float zzp = dread(t, x, y, s + 1), zzn = dread(t, x, y, s - 1);
    dD[0] = (pzz - nzz) * 0.5f;
    dD[1] = (zpz - znz) * 0.5f;
    dD[2] = (zzp - zzn) * 0.5f;
}

/* :90-163 hessian3D + :166-176 interpolationStep + Common.hpp:34-47 invert
 * The matrix is column-major: H[c] is column c.  invert() puts cross(x1,x2), cross(x2,x0),
 * cross(x0,x1) as the COLUMNS of the adjugate (exact for the symmetric H up to rounding),
 * scales by 1/det, then by -1; alpha = Hi * dD = Hi[0]*dD.x + Hi[1]*dD.y + Hi[2]*dD.z.
 * determinant(): MSL builtin; restated as the scalar triple product x0 . (x1 x x2).            */
static void cross3(const float a[3], const float b[3], float r[3]) {
    r[0] = a[1] * b[2] - a[2] * b[1];
    r[1] = a[2] * b[0] - a[0] * b[2];
    r[2] = a[0] * b[1] - a[1] * b[0];
}
static void interpolation_step(const dogtex *t, int x, int y, int s, float alpha[3]) {
    const float zzz = dread(t, x, y, s);
    const float pzz = dread(t, x + 1, y, s), nzz = dread(t, x - 1, y, s);
    const float zpz = dread(t, x, y + 1, s), znz = dread(t, x, y - 1, s);
    const float zzp = dread(t, x, y, s + 1), zzn = dread(t, x, y, s - 1);
    const float ppz = dread(t, x + 1, y + 1, s), nnz = dread(t, x - 1, y - 1, s);
    const float npz = dread(t, x - 1, y + 1, s), pnz = dread(t, x + 1, y - 1, s);
    const float pzp = dread(t, x + 1, y, s + 1), nzp = dread(t, x - 1, y, s + 1);
    const float zpp = dread(t, x, y + 1, s + 1), znp = dread(t, x, y - 1, s + 1);
    const float pzn = dread(t, x + 1, y, s - 1), nzn = dread(t, x - 1, y, s - 1);
    const float zpn = dread(t, x, y + 1, s - 1), znn = dread(t, x, y - 1, s - 1);
    const float dxx = pzz + nzz - 2.0f * zzz;
    const float dyy = zpz + znz - 2.0f * zzz;
    const float dss = zzp + zzn - 2.0f * zzz;
    const float dxy = (ppz - npz - pnz + nnz) * 0.25f;
    const float dxs = (pzp - nzp - pzn + nzn) * 0.25f;
    const float dys = (zpp - znp - zpn + znn) * 0.25f;
    const float x0[3] = {dxx, dxy, dxs}, x1[3] = {dxy, dyy, dys}, x2[3] = {dxs, dys, dss};
    float c12[3], c20[3], c01[3];
    cross3(x1, x2, c12); cross3(x2, x0, c20); cross3(x0, x1, c01);
    const float det = x0[0] * c12[0] + x0[1] * c12[1] + x0[2] * c12[2];
    const float inv = 1.0f / det;
    float Hi[3][3];
    for (int k = 0; k < 3; k++) {
        Hi[0][k] = -1.0f * (inv * c12[k]);
        Hi[1][k] = -1.0f * (inv * c20[k]);
        Hi[2][k] = -1.0f * (inv * c01[k]);
    }
    float dD[3];
    derivatives3d(t, x, y, s, dD);
    for (int k = 0; k < 3; k++)
        alpha[k] = Hi[0][k] * dD[0] + Hi[1][k] * dD[1] + Hi[2][k] * dD[2];
}

/* :179-190 outOfBounds (border 5, scale in [1, nspo]) */
static int out_of_bounds(int x, int y, int s, int w, int h, int scales) {
    const int border = 5;
    return x < border || x > w - border - 1 || y < border || y > h - border - 1 || s < 1 || s > scales;
}

/* :193-300 siftInterpolate  +  SIFTOctave.interpolateKeypoints (SIFTOctave.swift:205-288) */
/* stages[0..3] (optional): candidates that pass the 0.8 x threshold pre-filter / whose interpolation converges inside the
 * volume / that pass the contrast test / that pass the edge test (= returned).  contrast_terms = 1: the reference's contrast
 * (SIFTInterpolate.metal:89-100, x term only); 3: all three terms, v + 0.5 (dD . alpha), as IPOL's sift_anatomy does -- only for
 * the stage-count comparison with the IPOL fixtures (tests/test_oracle_golden.py), never on a parity path.
 * rows (optional, [n][4]): per input candidate (y, x, sigma, stage reached) in IPOL's units -- stage -1 = failed the pre-filter,
 * 0 = passed it only (position = the sample), 1 = converged, 2 = passed the contrast test, 3 = returned (positions interpolated). */
static int refine_impl(const so_ctx *c, int o, const so_extremum *ext, int n, so_keypoint *out, int cap, int *stages, int contrast_terms, float *rows) {
    const so_octave *q = &c->oct[o];
    const dogtex t = {q->dog, q->w, q->h, c->cfg.nspo + 2, (size_t)q->w * q->h};
    /* SIFTOctave.swift:217-226 parameter literals */
    const float dogThreshold = 0.0133f, maxOffset = 0.6f, edgeThreshold = 10.0f;
    const int maxIterations = 5, scales = c->cfg.nspo;
    const float delta = q->delta;
    const float sigmaRatio = q->sigmas[1] / q->sigmas[0];          /* :211 */
    int count = 0;
    for (int k = 0; k < n; k++) {
        int x = ext[k].x, y = ext[k].y, s = ext[k].scale;
        float value = dread(&t, x, y, s);
        if (rows) { rows[4 * k] = (float)y * delta; rows[4 * k + 1] = (float)x * delta; rows[4 * k + 2] = q->sigmas[s]; rows[4 * k + 3] = -1.0f; }
        if (fabsf(value) <= dogThreshold * 0.8f) continue;         /* metal :208 */
        if (stages) stages[0]++;
        if (rows) rows[4 * k + 3] = 0.0f;
        if (out_of_bounds(x, y, s, q->w, q->h, scales)) continue;  /* :223 */
        int converged = 0;
        float alpha[3] = {0.0f, 0.0f, 0.0f};
        int i = 0, dropped = 0;
        while (i < maxIterations) {                                /* :231-274 */
            interpolation_step(&t, x, y, s, alpha);
            if (fabsf(alpha[0]) < maxOffset && fabsf(alpha[1]) < maxOffset && fabsf(alpha[2]) < maxOffset) {
                converged = 1;
                break;
            }
            if (alpha[0] > +maxOffset) x += 1;
            if (alpha[0] < -maxOffset) x -= 1;
            if (alpha[1] > +maxOffset) y += 1;
            if (alpha[1] < -maxOffset) y -= 1;
            if (alpha[2] > +maxOffset) s += 1;
            if (alpha[2] < -maxOffset) s -= 1;
            if (out_of_bounds(x, y, s, q->w, q->h, scales)) { dropped = 1; break; }
            i += 1;
        }
        if (dropped || !converged) continue;
        if (stages) stages[1]++;
        if (rows) {
            rows[4 * k] = ((float)y + alpha[1]) * delta; rows[4 * k + 1] = ((float)x + alpha[0]) * delta;
            rows[4 * k + 2] = q->sigmas[s] * powf(sigmaRatio, alpha[2]); rows[4 * k + 3] = 1.0f;
        }
        {   /* :89-100 interpolateContrast: v + 0.5 * dDx * alpha.x (x term only) */
            float dD[3];
            derivatives3d(&t, x, y, s, dD);
            const float cx = dD[0] * alpha[0];
            if (contrast_terms == 3) value = dread(&t, x, y, s) + (cx + dD[1] * alpha[1] + dD[2] * alpha[2]) * 0.5f;
            else value = dread(&t, x, y, s) + cx * 0.5f;
        }
        if (fabsf(value) <= dogThreshold) continue;                /* :282 */
        if (stages) stages[2]++;
        if (rows) rows[4 * k + 3] = 2.0f;
        if (is_on_edge(&t, x, y, s, edgeThreshold)) continue;      /* :287 */
        if (stages) stages[3]++;
        if (rows) rows[4 * k + 3] = 3.0f;
        if (count < cap && out) {
            so_keypoint *p = &out[count];                          /* SIFTOctave.swift:266-284 */
            p->octave = o;
            p->scale = s;
            p->subScale = alpha[2];
            p->x = x; p->y = y;
            p->absX = ((float)x + alpha[0]) * delta;
            p->absY = ((float)y + alpha[1]) * delta;
            p->normX = (float)x / (float)q->w;
            p->normY = (float)y / (float)q->h;
            p->sigma = q->sigmas[s] * powf(sigmaRatio, alpha[2]);
            p->value = value;
        }
        count++;
    }
    return count;
}

int so_refine(const so_ctx *c, int o, const so_extremum *ext, int n, so_keypoint *out, int cap) {
    return refine_impl(c, o, ext, n, out, cap, NULL, 1, NULL);
}

int so_refine_stages(const so_ctx *c, int o, const so_extremum *ext, int n, so_keypoint *out, int cap, int stages[4], int contrast_terms, float *rows) {
    stages[0] = stages[1] = stages[2] = stages[3] = 0;
    return refine_impl(c, o, ext, n, out, cap, stages, contrast_terms == 3 ? 3 : 1, rows);
}

/* ---------------------------------------------------------------------------------------------
 * Sources/MetalShaders/Metal/SIFTGradient.metal:15-39, evaluated on demand instead of being
 * materialised for all layers: (atan2(tx, ty), sqrt(tx^2+ty^2)) with mirror edges.  A read of the
 * rg32Float gradient texture outside the image returns (0,0).                                   */
static inline void gradient_at(const float *g, int w, int h, int gx, int gy, float *theta, float *mag) {
    if (gx < 0 || gy < 0 || gx >= w || gy >= h) { *theta = 0.0f; *mag = 0.0f; return; }
    const int px = symmetrized(gx + 1, w), mx = symmetrized(gx - 1, w);
    const int py = symmetrized(gy + 1, h), my = symmetrized(gy - 1, h);
    const float cpx = tex_read(g, w, h, px, gy), cmx = tex_read(g, w, h, mx, gy);
    const float cpy = tex_read(g, w, h, gx, py), cmy = tex_read(g, w, h, gx, my);
    const float tx = (cpx - cmx) * 0.5f;
    const float ty = (cpy - cmy) * 0.5f;
    *theta = atan2f(tx, ty);
    *mag = sqrtf(tx * tx + ty * ty);
}

/* ---------------------------------------------------------------------------------------------
 * SIFTOctave.getKeypointOrientations (SIFTOctave.swift:290-382: host border filter, Int32
 * truncation of the absolute coordinates) + Sources/MetalShaders/Metal/SIFTOrientation.metal   */
static void orientation_one(const so_ctx *c, int o, const so_keypoint *kpk, int k, so_orientation *res_out, float *hist_out) {
    const so_octave *q = &c->oct[o];
    const float delta = q->delta, lambda = 1.5f, orientationThreshold = 0.8f;   /* SIFTOctave.swift:296-300 */
    const size_t npx = (size_t)q->w * q->h;
    const int absoluteX = (int32_t)kpk->absX, absoluteY = (int32_t)kpk->absY;   /* :333-334 */
    const int scale = kpk->scale;
    const float *g = q->gauss + (size_t)scale * npx;

    float histogram[SO_ORI_BINS];
    for (int i = 0; i < SO_ORI_BINS; i++) histogram[i] = 0.0f;
    {   /* metal :87-136 getOrientationsHistogram */
        const int bins = SO_ORI_BINS;
        const int x = (int)roundf((float)absoluteX / delta);
        const int y = (int)roundf((float)absoluteY / delta);
        const float sigma = kpk->sigma / delta;
        const float exponentDenominator = 2.0f * lambda * lambda;
        const int r = (int)ceilf(3.0f * lambda * sigma);
        for (int j = -r; j <= r; j++) {
            for (int i = -r; i <= r; i++) {
                const float u = (float)i / sigma, v = (float)j / sigma;
                const float r2 = u * u + v * v;
                const float w = expf(-r2 / exponentDenominator);
                float orientation, magnitude;
                gradient_at(g, q->w, q->h, x + i, y + j, &orientation, &magnitude);
                const float t = orientation / (2.0f * SO_PI_F);
                int bin = (int)roundf(t * (float)bins);
                if (bin < 0) bin += bins;
                if (bin >= bins) bin -= bins;
                const float m = w * magnitude;
                histogram[bin] += m;
            }
        }
    }
    {   /* :67-84 smoothHistogram, 6 iterations (:165) */
        float temp[SO_ORI_BINS];
        const int nb = SO_ORI_BINS;
        for (int it = 0; it < 6; it++) {
            for (int i = 0; i < nb; i++) temp[i] = histogram[i];
            for (int i = 0; i < nb; i++) {
                const float h0 = temp[((i - 1) + nb) % nb], h1 = temp[i], h2 = temp[(i + 1) % nb];
                histogram[i] = (h0 + h1 + h2) / 3.0f;
            }
        }
    }
    if (hist_out) memcpy(hist_out, histogram, sizeof(histogram));      /* smoothed histogram (test hook) */
    so_orientation res;
    memset(&res, 0, sizeof(res));
    res.keypoint = k;
    {   /* :36-64 getPrincipalOrientations, :16-33 orientationFromBin, interpolatePeak */
        const int bins = SO_ORI_BINS;
        float maximum = (float)(-2147483647 - 1);
        for (int i = 0; i < bins; i++) maximum = fmaxf(maximum, histogram[i]);
        const float threshold = orientationThreshold * maximum;
        int oc = 0;
        for (int i = 0; i < bins; i++) {
            const float hm = histogram[((i - 1) + bins) % bins], h0 = histogram[i], hp = histogram[(i + 1) % bins];
            if (h0 > threshold && h0 > hm && h0 > hp) {
                const float offset = (hm - hp) / (2.0f * (hm + hp - 2.0f * h0));
                const float tbin = ((float)i + offset) / (float)bins;
                const float tau = 2.0f * SO_PI_F;
                float orientation = tbin * tau;
                if (orientation < 0.0f) orientation += tau;
                if (orientation >= tau) orientation -= tau;
                res.orientations[oc++] = orientation;
            }
        }
        res.count = oc;
    }
    *res_out = res;
}

/* test hook: the 36-bin histogram after the 6 smoothing passes, for one keypoint (no border filter) */
void so_orientation_histogram(const so_ctx *c, int o, const so_keypoint *kp, float *hist36) {
    so_orientation tmp;
    orientation_one(c, o, kp, 0, &tmp, hist36);
}

int so_orientations(const so_ctx *c, int o, const so_keypoint *kp, int n, so_orientation *out, int cap) {
    const so_octave *q = &c->oct[o];
    const float delta = q->delta, lambda = 1.5f;
    const float minX = 1.0f, minY = 1.0f;
    const float maxX = (float)(q->w - 2), maxY = (float)(q->h - 2);
    int *pass = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    int count = 0;
    for (int k = 0; k < n; k++) {
        /* SIFTOctave.swift:311-329 border filter on float coordinates */
        const float x = kp[k].absX / delta, y = kp[k].absY / delta;
        const float sigma = kp[k].sigma / delta;
        const float r = ceilf(3.0f * lambda * sigma);
        if (floorf(x - r) < minX) continue;
        if (ceilf(x + r) > maxX) continue;
        if (floorf(y - r) < minY) continue;
        if (ceilf(y + r) > maxY) continue;
        pass[count++] = k;
    }
    const int nw = (out ? (count < cap ? count : cap) : 0);
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < nw; i++) orientation_one(c, o, &kp[pass[i]], pass[i], &out[i], NULL);
    free(pass);
    return count;
}

/* ---------------------------------------------------------------------------------------------
 * Sources/MetalShaders/Metal/SIFTDescriptor.metal:53-117 offset/addValue/addFeature             */
static inline void add_value(float *patch, int x, int y, int b, float value) {
    const int side = 4, bins = 8;
    if (x < 0 || x >= side || y < 0 || y >= side) return;
    if (b < 0) b += bins;
    if (b >= bins) b -= bins;
    patch[(y * side * bins) + (x * bins) + b] += value;
}
static inline void add_feature(float *patch, float x, float y, float b, float value) {
    const int cax = (int)floorf(x), cay = (int)floorf(y);
    const int cbx = (int)ceilf(x),  cby = (int)floorf(y);
    const int ccx = (int)ceilf(x),  ccy = (int)ceilf(y);
    const int cdx = (int)floorf(x), cdy = (int)ceilf(y);
    const int ba = (int)floorf(b), bb = (int)ceilf(b);
    const float iMax = x - floorf(x), iMin = 1.0f - iMax;
    const float jMax = y - floorf(y), jMin = 1.0f - jMax;
    const float bMax = b - floorf(b), bMin = 1.0f - bMax;
    add_value(patch, cax, cay, ba, (iMin * jMin * bMin) * value);
    add_value(patch, cax, cay, bb, (iMin * jMin * bMax) * value);
    add_value(patch, cbx, cby, ba, (iMax * jMin * bMin) * value);
    add_value(patch, cbx, cby, bb, (iMax * jMin * bMax) * value);
    add_value(patch, ccx, ccy, ba, (iMax * jMax * bMin) * value);
    add_value(patch, ccx, ccy, bb, (iMax * jMax * bMax) * value);
    add_value(patch, cdx, cdy, ba, (iMin * jMax * bMin) * value);
    add_value(patch, cdx, cdy, bb, (iMin * jMax * bMax) * value);
}
/* :15-29 normalizeFeatures */
static void normalize_features(int count, float *f) {
    float magnitude = 0.0f;
    for (int i = 0; i < count; i++) magnitude += f[i] * f[i];
    const float d = 1.0f / sqrtf(magnitude);
    for (int i = 0; i < count; i++) f[i] *= d;
}

/* SIFTOctave.getDescriptors (SIFTOctave.swift:384-492: (keypoint x theta) expansion, Int32
 * truncation, no border rejection) + SIFTDescriptor.metal:120-237 siftDescriptors.
 * A sample whose truncated coordinate falls outside the image contributes nothing (the
 * reference's behaviour there is undefined: ushort cast of a negative float, OOB texture read). */
static void descriptor_one(const so_ctx *c, int o, const so_keypoint *p, int k, float theta,
                           so_descriptor *r, float *features_out) {
    const so_octave *q = &c->oct[o];
    const float delta = q->delta;
    const int scalesPerOctave = 3;            /* SIFTOctave.swift:398 literal (not nspo) */
    const size_t npx = (size_t)q->w * q->h;
    const int absoluteX = (int32_t)p->absX, absoluteY = (int32_t)p->absY;   /* :417-418 */
    const int scale = p->scale;
    const float subScale = p->subScale;
    const float *g = q->gauss + (size_t)scale * npx;

    const float px = (float)absoluteX / delta;      /* metal :140-141 */
    const float py = (float)absoluteY / delta;
    const int d = 4, bins = 8;
    const float tau = 2.0f * SO_PI_F;
    const float cosT = cosf(theta), sinT = sinf(theta);
    const float binsPerRadian = (float)bins / tau;
    const float exponentDenominator = (float)(d * d) * 0.5f;
    const float interval = (float)scale + subScale;
    const float intervals = (float)scalesPerOctave;
    const float sigma = 1.6f;
    const float sc = sigma * powf(2.0f, interval / intervals);
    const float histogramWidth = 3.0f * sc;
    const int radius = (int)(histogramWidth * sqrtf(2.0f) * ((float)d + 1.0f) * 0.5f + 0.5f);

    float features[SO_DESC_FEATURES];
    for (int i = 0; i < SO_DESC_FEATURES; i++) features[i] = 0.0f;
    for (int j = -radius; j <= +radius; j++) {
        for (int i = -radius; i <= +radius; i++) {
            const float rx = ((float)j * cosT - (float)i * sinT) / histogramWidth;
            const float ry = ((float)j * sinT + (float)i * cosT) / histogramWidth;
            const float bx = rx + (float)(d / 2) - 0.5f;
            const float by = ry + (float)(d / 2) - 0.5f;
            /* ushort2(px + j, py + i): float -> ushort truncates toward zero */
            const float fx = truncf(px + (float)j), fy = truncf(py + (float)i);
            float gth = 0.0f, gm = 0.0f;
            if (fx >= 0.0f && fy >= 0.0f && fx < (float)q->w && fy < (float)q->h)
                gradient_at(g, q->w, q->h, (int)fx, (int)fy, &gth, &gm);
            float orientation = gth - theta;
            const float magnitude = gm;
            while (orientation < 0.0f) orientation += tau;
            while (orientation >= tau) orientation -= tau;
            const float bin = orientation * binsPerRadian;
            const float exponentNumerator = rx * rx + ry * ry;
            const float w = expf(-exponentNumerator / exponentDenominator);
            const float value = magnitude * w;
            add_feature(features, bx, by, bin, value);
        }
    }
    normalize_features(SO_DESC_FEATURES, features);
    for (int i = 0; i < SO_DESC_FEATURES; i++) features[i] = fminf(features[i], 0.2f);   /* :32-39 */
    normalize_features(SO_DESC_FEATURES, features);
    if (features_out) memcpy(features_out, features, sizeof(features));
    for (int i = 0; i < SO_DESC_FEATURES; i++)                                          /* :42-50 */
        r->features[i] = (int32_t)fminf(255.0f, features[i] * 512.0f);
    r->valid = 1;
    r->keypoint = k;
    r->theta = theta;
}

int so_descriptors(const so_ctx *c, int o, const so_keypoint *kp, const so_orientation *ori, int n_ori,
                   so_descriptor *out, float *features_f32, int cap) {
    int *offs = (int *)malloc(sizeof(int) * (size_t)(n_ori + 1));
    int count = 0;
    for (int k = 0; k < n_ori; k++) { offs[k] = count; count += ori[k].count; }
    if (out) {
#pragma omp parallel for schedule(dynamic, 4)
        for (int k = 0; k < n_ori; k++) {
            for (int t = 0; t < ori[k].count; t++) {
                const int idx = offs[k] + t;
                if (idx >= cap) continue;
                descriptor_one(c, o, &kp[ori[k].keypoint], k, ori[k].orientations[t], &out[idx],
                               features_f32 ? features_f32 + (size_t)idx * SO_DESC_FEATURES : NULL);
            }
        }
    }
    free(offs);
    return count;
}

/* ---------------------------------------------------------------------------------------------
 * SIFT.getKeypoints + SIFT.getDescriptors (SIFT.swift:147-238) end to end; used for timing.   */
int so_detect_describe(so_ctx *c, const void *pixels, int format, int stride,
                       int32_t *n_extrema, int32_t *n_keypoints, int32_t *n_oriented,
                       int32_t *n_descriptors) {
    so_build_pyramid(c, pixels, format, stride);
    int total = 0;
    for (int o = 0; o < c->cfg.n_octaves; o++) {
        int ne = so_extrema(c, o, NULL, 0);
        so_extremum *ext = (so_extremum *)malloc(sizeof(so_extremum) * (size_t)(ne + 1));
        so_extrema(c, o, ext, ne);
        so_keypoint *kp = (so_keypoint *)malloc(sizeof(so_keypoint) * (size_t)(ne + 1));
        int nk = so_refine(c, o, ext, ne, kp, ne);
        so_orientation *ori = (so_orientation *)malloc(sizeof(so_orientation) * (size_t)(nk + 1));
        int no = so_orientations(c, o, kp, nk, ori, nk);
        int nd = 0;
        for (int k = 0; k < no; k++) nd += ori[k].count;
        so_descriptor *desc = (so_descriptor *)malloc(sizeof(so_descriptor) * (size_t)(nd + 1));
        so_descriptors(c, o, kp, ori, no, desc, NULL, nd);
        if (n_extrema) n_extrema[o] = ne;
        if (n_keypoints) n_keypoints[o] = nk;
        if (n_oriented) n_oriented[o] = no;
        if (n_descriptors) n_descriptors[o] = nd;
        total += nd;
        free(ext); free(kp); free(ori); free(desc);
    }
    return total;
}

/* ---------------------------------------------------------------------------------------------
 * Sources/SIFTMetal/SIFT/SIFTDescriptor.swift:298-361  SIFTDescriptor.match (brute force + ratio test)
 * distance = FloatVector.distance (Utilities/Vector.swift:226-239: sqrt(vDSP.distanceSquared)) between the
 * two descriptors' indexValue = rawFeatures (features / 255 as Float, SIFTDescriptor.swift:36-40) in a
 * permuted cell order (:42-80; a permutation does not change the distance).  vDSP's summation order is
 * not documented; restated as a sequential f32 sum.
 * The scan keeps `second` = the running best at the moment the final best was found (it is NOT updated by
 * a distance that falls between best and second, :333-337), so it equals the minimum over the targets
 * BEFORE the best one -- FLT_MAX when the best is the first target.                                   */
int so_match(const int32_t *src /* [n_src][128] */, int n_src, const int32_t *tgt /* [n_tgt][128] */, int n_tgt,
             float absoluteThreshold, float relativeThreshold, so_match_rec *out, int cap) {
    int count = 0;
    float *traw = (float *)malloc(sizeof(float) * (size_t)(n_tgt > 0 ? n_tgt : 1) * SO_DESC_FEATURES);
    for (long i = 0; i < (long)n_tgt * SO_DESC_FEATURES; i++) traw[i] = (float)tgt[i] / 255.0f;
    so_match_rec *res = (so_match_rec *)malloc(sizeof(so_match_rec) * (size_t)(n_src > 0 ? n_src : 1));
#pragma omp parallel for schedule(dynamic, 16)
    for (int s = 0; s < n_src; s++) {
        float a[SO_DESC_FEATURES];
        for (int i = 0; i < SO_DESC_FEATURES; i++) a[i] = (float)src[(size_t)s * SO_DESC_FEATURES + i] / 255.0f;
        int best = -1;
        float bestD = 3.402823466e+38f, secondD = 0.0f;
        int haveSecond = 0;
        for (int t = 0; t < n_tgt; t++) {
            const float *b = traw + (size_t)t * SO_DESC_FEATURES;
            float d2 = 0.0f;
            for (int i = 0; i < SO_DESC_FEATURES; i++) { const float d = a[i] - b[i]; d2 += d * d; }
            const float distance = sqrtf(d2);
            if (distance < bestD) { best = t; secondD = bestD; haveSecond = 1; bestD = distance; }
        }
        res[s].source = s; res[s].target = -1; res[s].distance = bestD;
        if (best < 0 || !haveSecond) continue;
        if (!(bestD < absoluteThreshold)) continue;
        if (!(bestD < secondD * relativeThreshold)) continue;
        res[s].target = best;
    }
    for (int s = 0; s < n_src; s++) {
        if (res[s].target < 0) continue;
        if (out && count < cap) out[count] = res[s];
        count++;
    }
    free(res); free(traw);
    return count;
}

/* ---------------------------------------------------------------------------------------------
 * Sources/SIFTMetal/SIFT/SIFTDescriptor.swift:36-89  SIFTDescriptor.init: rawFeatures, indexValue, indexKey
 *   rawFeatures[i] = Float(features[i]) / Float(255)                                   (:36-40)
 *   the 128 values are cut into 16 runs of 8 (one run per 4x4 cell) and re-ordered centre cells 5,6,9,10,
 *   corner cells 0,3,12,15, edge cells 1,2,4,7,8,11,13,14                              (:42-74)
 *   indexValue = the re-ordered runs joined (128), indexKey = the mean of each run (16) (:78-88)
 * mean is vDSP.mean (Utilities/Vector.swift:123-126), whose summation order is not documented; restated as
 * a sequential f32 sum divided by 8.                                                                    */
static const int so_index_cell_order[16] = {5, 6, 9, 10, 0, 3, 12, 15, 1, 2, 4, 7, 8, 11, 13, 14};

void so_descriptor_index(const int32_t *features /* [n][128] */, int n, float *raw /* [n][128] */,
                         float *indexValue /* [n][128] */, float *indexKey /* [n][16] */) {
    for (int d = 0; d < n; d++) {
        float r[SO_DESC_FEATURES];
        for (int i = 0; i < SO_DESC_FEATURES; i++) r[i] = (float)features[(size_t)d * SO_DESC_FEATURES + i] / 255.0f;
        if (raw) memcpy(raw + (size_t)d * SO_DESC_FEATURES, r, sizeof(r));
        for (int k = 0; k < 16; k++) {
            const float *run = r + so_index_cell_order[k] * 8;
            float sum = 0.0f;
            for (int i = 0; i < 8; i++) {
                if (indexValue) indexValue[(size_t)d * SO_DESC_FEATURES + k * 8 + i] = run[i];
                sum += run[i];
            }
            if (indexKey) indexKey[(size_t)d * 16 + k] = sum / 8.0f;
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * Sources/SIFTMetal/SIFT/SIFTDescriptor.swift:162-296  compareGeometry (private) on keypoint
 * absoluteCoordinate pairs (makeCoordinate :146-160).  simd_length / simd_normalize / simd_dot are Apple
 * <simd/geometry.h> inlines; restated as sqrtf(x*x + y*y), v * (1 / length), x0*y0 + x1*y1 in f32 -- their
 * exact rounding is not pinned.                                                                         */
static float so_clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }
static float so_len2(float x, float y) { return sqrtf(x * x + y * y); }
static float so_half_dot(float ax, float ay, float bx, float by) { return so_clamp01((ax * bx + ay * by) * 0.5f + 0.5f); } /* :158-160 */

float so_compare_geometry(const so_match_rec *matches, int n, const float *src_xy, const float *tgt_xy, int minimumSampleSize) {
    const float minimumLength = 2.0f;
    float sum = 0.0f;
    int count = 0;
    float *scores = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n - 3; i++) {                                                    /* :173 */
        const so_match_rec *m0 = &matches[i], *m1 = &matches[i + 1], *m2 = &matches[i + 2], *m3 = &matches[i + 3];
        const float sbx = src_xy[2 * m1->source] - src_xy[2 * m0->source], sby = src_xy[2 * m1->source + 1] - src_xy[2 * m0->source + 1];
        const float tbx = tgt_xy[2 * m1->target] - tgt_xy[2 * m0->target], tby = tgt_xy[2 * m1->target + 1] - tgt_xy[2 * m0->target + 1];
        const float sbl = so_len2(sbx, sby), tbl = so_len2(tbx, tby);
        if (!(sbl >= minimumLength)) continue;                                           /* :188-194 */
        if (!(tbl >= minimumLength)) continue;
        const float stx = src_xy[2 * m3->source] - src_xy[2 * m2->source], sty = src_xy[2 * m3->source + 1] - src_xy[2 * m2->source + 1];
        const float ttx = tgt_xy[2 * m3->target] - tgt_xy[2 * m2->target], tty = tgt_xy[2 * m3->target + 1] - tgt_xy[2 * m2->target + 1];
        const float stl = so_len2(stx, sty), ttl = so_len2(ttx, tty);
        if (!(stl >= minimumLength)) continue;                                           /* :207-213 */
        if (!(ttl >= minimumLength)) continue;
        const float isb = 1.0f / sbl, itb = 1.0f / tbl, ist = 1.0f / stl, itt = 1.0f / ttl;
        const float sourceRatio = stl / sbl, targetRatio = ttl / tbl;
        const float sd = so_half_dot(stx * ist, sty * ist, sbx * isb, sby * isb);
        const float td = so_half_dot(ttx * itt, tty * itt, tbx * itb, tby * itb);
        const float orientationSimilarity = 1.0f - fabsf(sd - td);
        const float scaleSimilarity = (sourceRatio < targetRatio) ? so_clamp01(sourceRatio / targetRatio) : so_clamp01(targetRatio / sourceRatio);
        const float similarity = orientationSimilarity * scaleSimilarity;
        const float score = similarity * similarity;
        scores[count++] = score;
        sum += score;
    }
    if (count < minimumSampleSize) { free(scores); return 0.0f; }                        /* :249-251 */
    const float mean = sum / (float)count;
    float error = 0.0f;
    for (int i = 0; i < count; i++) { const float d = scores[i] - mean; error += d * d; }
    const float variance = error / (float)(count - 1);
    const float sd = sqrtf(variance);
    float fairSum = 0.0f, fairCount = 0.0f;
    for (int i = 0; i < count; i++) {
        const float z = fabsf((scores[i] - mean) / sd);
        if (z <= 2.0f) { fairSum += scores[i]; fairCount += 1.0f; }                      /* NaN z (sd == 0) fails the test, as in the reference */
    }
    free(scores);
    return fairSum / fairCount;
}

/* Sources/SIFTMetal/SIFT/SIFTDescriptor.swift:104-144  matchGeometry: match, at least 7 matches, first 80 scored */
float so_match_geometry(const int32_t *src, const float *src_xy, int n_src, const int32_t *tgt, const float *tgt_xy, int n_tgt,
                        float absoluteThreshold, float relativeThreshold, int *n_matches) {
    const int minimumSampleSize = 7, maximumSampleSize = 80;
    so_match_rec *m = (so_match_rec *)malloc(sizeof(so_match_rec) * (size_t)(n_src > 0 ? n_src : 1));
    const int n = so_match(src, n_src, tgt, n_tgt, absoluteThreshold, relativeThreshold, m, n_src);
    if (n_matches) *n_matches = n;
    float score = 0.0f;
    if (n >= minimumSampleSize) score = so_compare_geometry(m, n < maximumSampleSize ? n : maximumSampleSize, src_xy, tgt_xy, minimumSampleSize);
    free(m);
    return score;
}

/* ---------------------------------------------------------------------------------------------
 * Sources/SIFTMetal/Utilities/Trie.swift:76-416  Trie (approximate nearest neighbour) and
 * Sources/SIFTMetal/SIFT/SIFTDescriptor.swift:362-417  SIFTDescriptor.approximateMatch.
 * A literal pointer trie: every node has numberOfBins children; a key component v in [0,1] selects child
 * Int((v * Float(numberOfBins - 1)).rounded()) (:381-388); values live in the nodes reached by a whole key;
 * link() chains the leaves (nodes without children) in depth-first child order into a ring (:118-127,141-158).
 * Values are descriptor ids; their distance is IntVector.distance = sqrt(Float(sum of squared integer
 * differences)) (Utilities/Vector.swift:45-59).                                                          */
typedef struct so_trie_node {
    struct so_trie_node **nodes;       /* [numberOfBins] */
    int hasNodes;
    int *values, n_values, cap_values;
    struct so_trie_node *left, *right;
} so_trie_node;

struct so_trie {
    int numberOfBins;
    so_trie_node *root;
    const int32_t *features;           /* [n][128], borrowed: value id -> descriptor */
};

static so_trie_node *so_trie_new_node(int nb) {
    so_trie_node *n = (so_trie_node *)calloc(1, sizeof(so_trie_node));
    n->nodes = (so_trie_node **)calloc((size_t)nb, sizeof(so_trie_node *));
    return n;
}

static void so_trie_free_node(so_trie_node *n, int nb) {
    if (!n) return;
    for (int i = 0; i < nb; i++) so_trie_free_node(n->nodes[i], nb);
    free(n->nodes); free(n->values); free(n);
}

so_trie *so_trie_create(int numberOfBins, const int32_t *features) {
    so_trie *t = (so_trie *)calloc(1, sizeof(so_trie));
    t->numberOfBins = numberOfBins; t->features = features;
    t->root = so_trie_new_node(numberOfBins);
    return t;
}

void so_trie_destroy(so_trie *t) {
    if (!t) return;
    so_trie_free_node(t->root, t->numberOfBins);
    free(t);
}

static int so_trie_bin(const so_trie *t, float value) {                      /* binIndex(for:) :381-388 */
    return (int)roundf(value * (float)(t->numberOfBins - 1));                /* .rounded(): half away from zero */
}

static int so_trie_wrap(const so_trie *t, int input) {                       /* wrapBinIndex :404-415 */
    int output = input;
    const int n = t->numberOfBins - 1;
    if (output < 0) output += n;
    else if (output >= n) output -= n;
    return output;
}

void so_trie_insert(so_trie *t, const float *key, int len, int value) {      /* insert :160-194 */
    so_trie_node *node = t->root;
    for (int i = 0; i < len; i++) {
        const int b = so_trie_bin(t, key[i]);
        if (!node->nodes[b]) { node->nodes[b] = so_trie_new_node(t->numberOfBins); node->hasNodes = 1; }
        node = node->nodes[b];
    }
    if (node->n_values == node->cap_values) {
        node->cap_values = node->cap_values ? 2 * node->cap_values : 4;
        node->values = (int *)realloc(node->values, sizeof(int) * (size_t)node->cap_values);
    }
    node->values[node->n_values++] = value;
}

int so_trie_contains(const so_trie *t, const float *key, int len) {          /* contains :196-214 */
    const so_trie_node *node = t->root;
    for (int i = 0; i < len; i++) {
        const so_trie_node *next = node->nodes[so_trie_bin(t, key[i])];
        if (!next) return 0;
        node = next;
    }
    return 1;
}

static int so_trie_capacity_node(const so_trie_node *n, int nb) {            /* capacity :108-116 */
    int total = n->n_values;
    for (int i = 0; i < nb; i++) if (n->nodes[i]) total += so_trie_capacity_node(n->nodes[i], nb);
    return total;
}
int so_trie_capacity(const so_trie *t) { return so_trie_capacity_node(t->root, t->numberOfBins); }

static void so_trie_leaves(so_trie_node *n, int nb, so_trie_node ***list, int *count, int *cap) {   /* leaves :141-158 */
    if (n->hasNodes) {
        for (int i = 0; i < nb; i++) if (n->nodes[i]) so_trie_leaves(n->nodes[i], nb, list, count, cap);
    } else {
        if (*count == *cap) { *cap = *cap ? 2 * *cap : 64; *list = (so_trie_node **)realloc(*list, sizeof(so_trie_node *) * (size_t)*cap); }
        (*list)[(*count)++] = n;
    }
}

int so_trie_link(so_trie *t) {                                               /* link :118-127; returns the number of leaves */
    so_trie_node **list = NULL;
    int count = 0, cap = 0;
    so_trie_leaves(t->root, t->numberOfBins, &list, &count, &cap);
    for (int i = 0; i < count; i++) {
        so_trie_node *thisNode = list[i], *nextNode = list[(i + 1) % count];
        thisNode->right = nextNode;
        nextNode->left = thisNode;
    }
    free(list);
    return count;
}

static so_trie_node *so_trie_closest(const so_trie *t, so_trie_node *cur, int binIndex) {   /* closestNode :343-360 */
    if (cur->nodes[binIndex]) return cur->nodes[binIndex];
    int bestDistance = 0x7fffffff;
    so_trie_node *bestNode = NULL;
    for (int j = 0; j < t->numberOfBins; j++) {
        if (!cur->nodes[j]) continue;
        const int distance = so_trie_wrap(t, abs(j - binIndex));             /* binDifference :373-379 */
        if (distance < bestDistance) { bestDistance = distance; bestNode = cur->nodes[j]; }
    }
    return bestNode;
}

static so_trie_node *so_trie_nearest_node(const so_trie *t, const float *key, int len) {    /* nearestNode :326-341 */
    so_trie_node *current = t->root;
    for (int i = 0; i < len; i++) {
        if (!current->hasNodes) return current;
        so_trie_node *node = so_trie_closest(t, current, so_trie_bin(t, key[i]));
        if (!node) return current;
        current = node;
    }
    return current;
}

/* FiniteQueue<Match> (:222-252): insert at the front, drop the last when over capacity */
typedef struct { int value[8]; float distance[8]; int count, capacity; } so_trie_queue;

static void so_trie_queue_insert(so_trie_queue *q, int value, float distance) {
    const int n = q->count < 8 ? q->count : 7;
    for (int i = n; i > 0; i--) { q->value[i] = q->value[i - 1]; q->distance[i] = q->distance[i - 1]; }
    q->value[0] = value; q->distance[0] = distance;
    q->count++;
    if (q->count > q->capacity) q->count--;
}

static void so_trie_nearest_value(const so_trie *t, const so_trie_node *node, const int32_t *query, so_trie_queue *q) {   /* :362-377 */
    float bestDistance = q->count ? q->distance[0] : 3.402823466e+38f;
    for (int v = 0; v < node->n_values; v++) {
        const int32_t *f = t->features + (size_t)node->values[v] * SO_DESC_FEATURES;
        long k = 0;
        for (int i = 0; i < SO_DESC_FEATURES; i++) { const long d = (long)f[i] - (long)query[i]; k += d * d; }
        const float distance = sqrtf((float)k);
        if (distance < bestDistance) { bestDistance = distance; so_trie_queue_insert(q, node->values[v], distance); }
    }
}

/* nearest(key:query:radius:k:) :300-324 -> number of matches (<= k <= 8); ids / distances best first */
int so_trie_nearest(const so_trie *t, const float *key, int len, const int32_t *query, int radius, int k, int *ids, float *distances) {
    so_trie_queue q; memset(&q, 0, sizeof(q)); q.capacity = k < 8 ? k : 8;
    so_trie_node *bin = so_trie_nearest_node(t, key, len);
    so_trie_nearest_value(t, bin, query, &q);
    so_trie_node *node = bin;
    for (int r = 0; r < radius; r++) { node = node->left; so_trie_nearest_value(t, node, query, &q); }
    node = bin;
    for (int r = 0; r < radius; r++) { node = node->right; so_trie_nearest_value(t, node, query, &q); }
    for (int i = 0; i < q.count; i++) { ids[i] = q.value[i]; distances[i] = q.distance[i]; }
    return q.count;
}

/* SIFTDescriptor.approximateMatch(source:target:absoluteThreshold:relativeThreshold:) :362-417: Trie(numberOfBins: 8) keyed
 * by indexKey, radius 10, k 2; a match needs two queue entries, best < absolute, best < second * relative. */
int so_approximate_match(const int32_t *src, int n_src, const int32_t *tgt, int n_tgt, float absoluteThreshold, float relativeThreshold,
                         so_match_rec *out, int cap) {
    float *tkey = (float *)malloc(sizeof(float) * 16 * (size_t)(n_tgt > 0 ? n_tgt : 1));
    float *skey = (float *)malloc(sizeof(float) * 16 * (size_t)(n_src > 0 ? n_src : 1));
    so_descriptor_index(tgt, n_tgt, NULL, NULL, tkey);
    so_descriptor_index(src, n_src, NULL, NULL, skey);
    so_trie *t = so_trie_create(8, tgt);
    for (int i = 0; i < n_tgt; i++) so_trie_insert(t, tkey + (size_t)i * 16, 16, i);
    so_trie_link(t);
    int count = 0;
    for (int s = 0; s < n_src; s++) {
        int ids[2]; float dist[2];
        const int m = so_trie_nearest(t, skey + (size_t)s * 16, 16, src + (size_t)s * SO_DESC_FEATURES, 10, 2, ids, dist);
        if (m != 2) continue;                                                /* :398-400 */
        if (!(dist[0] < absoluteThreshold)) continue;
        if (!(dist[0] < dist[1] * relativeThreshold)) continue;
        if (out && count < cap) { out[count].source = s; out[count].target = ids[0]; out[count].distance = dist[0]; }
        count++;
    }
    so_trie_destroy(t); free(tkey); free(skey);
    return count;
}
