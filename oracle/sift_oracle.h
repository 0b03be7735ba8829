/*
 * sift_oracle.h -- CPU restatement of the lukevanin/SIFTMetal detect+describe path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it, and
 * there only as the checker / the timed CPU baseline, never as the thing shipped.
 *
 * Parity pin (see DESIGN.md "Oracle"): the reference is Swift + Metal Shading Language and
 * cannot be built in this image (no swiftc, no Metal; building the .metal files would need a
 * hand-written stand-in for <metal_stdlib>, which the build rules forbid), so there is no
 * oracle/_ref.  The restatement is pinned against the golden data the reference's own tests
 * hold (Tests/SIFTMetalTests/Resources: IPOL sift_anatomy outputs on butterfly.png):
 *   - Gaussian stack      vs scalespace_butterfly_o*_s*.png          (<= 1.5/255, tight)
 *   - raw 3-D extrema     vs extra_NES_butterfly.txt (3068 rows)     (exact count with the
 *                            full 26-neighbour switch; the reference's own 25-neighbour
 *                            test is a superset)
 *   - refined keypoints   vs extra_OnEdgeResp_butterfly.txt (1304)   (>=98% within 0.01 px)
 *   - orientation hist.   vs butterfly-descriptors.txt (36 histogram columns): median
 *                            Pearson r = 0.993 bin for bin over 1289 co-located keypoints
 *   - theta / descriptor  vs butterfly-descriptors.txt: theta is the known -1/2 bin off;
 *                            features have median cosine similarity 0.971 to IPOL's (layout and
 *                            conventions pinned); magnitudes differ because the reference uses
 *                            an OpenSIFT-style descriptor, not IPOL's -- that part of the
 *                            descriptor rests on the cited restatement; see DESIGN.md.
 *
 * Every function cites the reference file:line it follows
 * (paths relative to the reference repo root).
 */
#ifndef SIFT_ORACLE_H
#define SIFT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SO_MAX_WEIGHTS 32          /* Sources/MetalShaders/include/ConvolutionSeries.h:13 */
#define SO_ORI_BINS 36             /* Sources/MetalShaders/include/SIFTOrientation.h:12  */
#define SO_DESC_FEATURES 128       /* Sources/MetalShaders/include/SIFTDescriptor.h:14-16 */

enum { SO_FMT_BGRA8 = 0, SO_FMT_GRAY8 = 1, SO_FMT_GRAYF32 = 2 };

typedef struct so_config {
    int32_t width, height;       /* input image size (SIFT.Configuration.inputSize)          */
    int32_t n_octaves;           /* reference hard-wires 7 (DifferenceOfGaussians.swift:41)   */
    int32_t nspo;                /* scales per octave, reference 3 (:46)                      */
    int32_t full_neighbourhood;  /* 0 = reference's 25-neighbour test, 1 = all 26 (IPOL)      */
    int32_t use_fma;             /* 1 = blur taps accumulate with fmaf (default), 0 = mul+add */
} so_config;

/* Record layouts = the reference's shared C structs (Sources/MetalShaders/include/ headers). */
typedef struct { int32_t x, y, scale; } so_extremum;                 /* SIFTExtrema.h:14-18 */

typedef struct {                                                      /* SIFTKeypoint.swift:11-57 */
    int32_t octave, scale;
    float   subScale;
    int32_t x, y;                 /* scaledCoordinate   */
    float   absX, absY;           /* absoluteCoordinate */
    float   normX, normY;         /* normalizedCoordinate */
    float   sigma, value;
} so_keypoint;                    /* 44 bytes */

typedef struct {                                                      /* SIFTOrientation.h:30-34 */
    int32_t keypoint, count;
    float   orientations[SO_ORI_BINS];
} so_orientation;                 /* 152 bytes */

typedef struct {                                                      /* SIFTDescriptor.h:37-42 */
    int32_t valid, keypoint;
    float   theta;
    int32_t features[SO_DESC_FEATURES];
} so_descriptor;                  /* 524 bytes */

typedef struct so_ctx so_ctx;

so_ctx *so_create(const so_config *cfg);
void    so_destroy(so_ctx *c);

/* schedule queries */
int   so_octave_width(const so_ctx *c, int o);
int   so_octave_height(const so_ctx *c, int o);
float so_octave_delta(const so_ctx *c, int o);
float so_octave_sigma(const so_ctx *c, int o, int s);
int   so_blur_taps(const so_ctx *c, int layer /*1..nspo+2, or 0 = seed*/, float *weights_out);

/* dense front end: gray -> 2x bilinear -> seed blur -> per-octave Gaussian stack + DoG */
void so_build_pyramid(so_ctx *c, const void *pixels, int format, int row_stride_bytes);
const float *so_gaussian(const so_ctx *c, int o, int s);   /* [h][w]            */
const float *so_dog(const so_ctx *c, int o, int s);        /* [h][w], s<nspo+2  */
const float *so_seed(const so_ctx *c);                     /* octave-0 layer 0  */

/* keypoint stages, per octave; return the count (which may exceed cap: nothing past cap is written) */
int so_extrema(const so_ctx *c, int o, so_extremum *out, int cap);
int so_refine(const so_ctx *c, int o, const so_extremum *ext, int n, so_keypoint *out, int cap);
/* so_refine with per-stage survivor counts (pre-filter, converged, contrast, edge) and, with contrast_terms = 3, IPOL's
   three-term contrast instead of the reference's x-term-only one: for the stage-by-stage comparison with the IPOL fixtures */
int so_refine_stages(const so_ctx *c, int o, const so_extremum *ext, int n, so_keypoint *out, int cap, int stages[4], int contrast_terms,
                     float *rows /* optional [n][4]: y, x, sigma (IPOL units), stage reached (-1 ... 3) per input candidate */);
int so_orientations(const so_ctx *c, int o, const so_keypoint *kp, int n, so_orientation *out, int cap);
/* test hook: smoothed 36-bin orientation histogram of one keypoint */
void so_orientation_histogram(const so_ctx *c, int o, const so_keypoint *kp, float *hist36);
/* expands (keypoint x theta) like SIFTOctave.getDescriptors; features_f32 (optional) receives
   the 128 pre-quantisation floats per descriptor */
int so_descriptors(const so_ctx *c, int o, const so_keypoint *kp, const so_orientation *ori, int n_ori,
                   so_descriptor *out, float *features_f32, int cap);

/* whole path in one call (used by the cpu_baseline timing leg): returns total descriptors,
   fills per-octave counts [n_octaves] (any pointer may be NULL) */
int so_detect_describe(so_ctx *c, const void *pixels, int format, int row_stride_bytes,
                       int32_t *n_extrema, int32_t *n_keypoints, int32_t *n_oriented,
                       int32_t *n_descriptors);

/* ---- rows after describe.  PARITY UNPINNED for everything below: the reference's tests hold no expected
   outputs for match / matchGeometry / the derived vectors, and Apple's vDSP / simd roundings are undocumented; these
   restatements follow the cited lines and are cross-checked by literal Python restatements in tests/. ---- */
/* SIFTDescriptor.match (SIFT/SIFTDescriptor.swift:298-361): brute force + ratio test, matches in source order */
typedef struct { int32_t source, target; float distance; } so_match_rec;
int so_match(const int32_t *src, int n_src, const int32_t *tgt, int n_tgt, float absoluteThreshold, float relativeThreshold,
             so_match_rec *out, int cap);

/* SIFTDescriptor.init (SIFT/SIFTDescriptor.swift:36-89): rawFeatures [n][128], indexValue [n][128], indexKey [n][16];
   any output may be NULL */
void so_descriptor_index(const int32_t *features, int n, float *raw, float *indexValue, float *indexKey);

/* compareGeometry (SIFT/SIFTDescriptor.swift:162-296) over matches whose source/target index rows of xy pairs, and
   matchGeometry (:104-144) = match + compareGeometry on the first 80 matches */
float so_compare_geometry(const so_match_rec *matches, int n, const float *src_xy, const float *tgt_xy, int minimumSampleSize);
float so_match_geometry(const int32_t *src, const float *src_xy, int n_src, const int32_t *tgt, const float *tgt_xy, int n_tgt,
                        float absoluteThreshold, float relativeThreshold, int *n_matches);

/* Trie (Utilities/Trie.swift:76-416) with descriptor ids as values, and SIFTDescriptor.approximateMatch
   (SIFT/SIFTDescriptor.swift:362-417).  `features` ([n][128]) is borrowed and must outlive the trie. */
typedef struct so_trie so_trie;
so_trie *so_trie_create(int numberOfBins, const int32_t *features);
void so_trie_destroy(so_trie *t);
void so_trie_insert(so_trie *t, const float *key, int len, int value);
int so_trie_contains(const so_trie *t, const float *key, int len);
int so_trie_capacity(const so_trie *t);
int so_trie_link(so_trie *t);
int so_trie_nearest(const so_trie *t, const float *key, int len, const int32_t *query, int radius, int k, int *ids, float *distances);
int so_approximate_match(const int32_t *src, int n_src, const int32_t *tgt, int n_tgt, float absoluteThreshold, float relativeThreshold,
                         so_match_rec *out, int cap);

int so_num_threads(void);
void so_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
