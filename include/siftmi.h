/*
 * siftmi.h -- C ABI of the MI355X-native SIFT detect+describe path (libsiftmi.so).
 *
 * Drop-in boundary for lukevanin/SIFTMetal's hot path.  Each entry point names the reference
 * interface it replaces (paths relative to the reference repo root).  The reference binds its
 * device code through Swift + the Clang module `MetalShaders` (Sources/MetalShaders/module.modulemap:1-4);
 * a Swift (or any FFI) host binds this header instead -- see INTEGRATION.md for the Swift stub.
 *
 * Conventions
 *  - plain pointers and sizes only; no C++/torch types
 *  - every function returns SIFTMI_OK (0) or a negative siftmi_status; siftmi_last_error()
 *    gives the message.  Nothing aborts (the reference traps via try!/precondition/fatalError).
 *  - a context is bound to one HIP device, is NOT re-entrant (like the reference's SIFT object:
 *    one command queue, shared scratch -- SIFT.swift:139), and owns all device memory.
 *  - pointers returned through `const T **` arguments are owned by the context and stay valid
 *    until the next call on the same context.
 *  - there is NO CPU fallback: without a HIP device siftmi_create fails with SIFTMI_E_NODEVICE.
 */
#ifndef SIFTMI_H
#define SIFTMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIFTMI_VERSION 100
#define SIFTMI_MAX_OCTAVES 16
#define SIFTMI_ORIENTATION_BINS 36      /* Sources/MetalShaders/include/SIFTOrientation.h:12   */
#define SIFTMI_DESCRIPTOR_FEATURES 128  /* Sources/MetalShaders/include/SIFTDescriptor.h:14-16 */

typedef enum siftmi_status {
    SIFTMI_OK          =  0,
    SIFTMI_E_BADARG    = -1,
    SIFTMI_E_CAPACITY  = -2,   /* a per-octave list overflowed its capacity; results are truncated,
                                  siftmi_last_error() names the list and the size it needed       */
    SIFTMI_E_HIP       = -3,
    SIFTMI_E_NODEVICE  = -4,
    SIFTMI_E_NOMEM     = -5,
    SIFTMI_E_STATE     = -6    /* e.g. describe before detect */
} siftmi_status;

/* Pixel formats.  The reference accepts only a .bgra8Unorm texture
   (Metal Compute/ConvertSRGBToGrayscaleKernel.swift:34); GRAY8/GRAYF32 skip the luma step.
   SIFTMI_FMT_GRAYF32 pixels are the luma itself and must lie in [0, 1], the range a unorm texture delivers: thresholds are absolute
   and the orientation / descriptor histograms are accumulated in 2^-24 fixed point (u32 bins sized for gradients of a [0, 1] image; 2^-22
   for descriptor windows wider than histogramWidth 17.4 -- six or seven scales per octave -- whose low-contrast ones are walked a second time
   at a finer unit),
   which overflows for an unnormalised (0 ... 255, HDR) image.  A frame with a value outside [0, 1] (or a NaN) is reported: SIFTMI_E_BADARG from the
   host-facing entry points, overflow_flags bit 5 on the device path; its results are not to be used. */
typedef enum siftmi_format {
    SIFTMI_FMT_BGRA8   = 0,
    SIFTMI_FMT_GRAY8   = 1,
    SIFTMI_FMT_GRAYF32 = 2
} siftmi_format;

/* Replaces SIFT.Configuration (SIFT/SIFT.swift:57-103) + DifferenceOfGaussians.Configuration
   (SIFT/DifferenceOfGaussians.swift:23-51).  Defaults (siftmi_default_config) are the literals
   the reference actually uses (SIFT/SIFTOctave.swift:217-226, 296-300, 396-401). */
typedef struct siftmi_config {
    int32_t width, height;              /* inputSize                                              */
    int32_t n_octaves;                  /* numberOfOctaves = 7                                    */
    int32_t nspo;                       /* numberOfScalesPerOctave = 3 (accepted: 1 ... 7)        */
    float   sigma_min;                  /* sigmaMinimum = 0.8                                     */
    float   delta_min;                  /* deltaMinimum = 0.5 (only 0.5 is supported: 2x seed)    */
    float   sigma_in;                   /* sigmaInput = 0.5                                       */
    float   dog_threshold;              /* 0.0133                                                 */
    float   edge_threshold;             /* 10                                                     */
    int32_t max_iterations;             /* 5                                                      */
    float   max_offset;                 /* 0.6                                                    */
    int32_t image_border;               /* 5                                                      */
    float   lambda_orientation;         /* 1.5                                                    */
    float   orientation_threshold;      /* 0.8                                                    */
    int32_t orientation_smoothing;      /* 6                                                      */
    int32_t descriptor_scales_per_octave; /* 3: literal at SIFTOctave.swift:398, NOT nspo          */
    int32_t full_neighbourhood;         /* 0 = reference's 25-neighbour extremum test
                                           (Metal/SIFTExtrema.metal:84), 1 = all 26               */
    int32_t max_batch;                  /* frames processed in lock-step per launch (>= 1)        */
    int32_t max_extrema;                /* per (frame, octave) capacities; 0 = derived from size  */
    int32_t max_keypoints;
    int32_t max_descriptors;
    int32_t keep_descriptor_floats;     /* 1 = also keep the 128 pre-quantisation floats          */
    int32_t use_hip_graph;              /* 1 (default) = siftmi_detect_describe_batch_device captures its
                                           launch sequence into a hipGraph and replays it while the
                                           caller keeps passing the same buffers                    */
    int32_t count_raw_extrema;          /* 0 (default): on octaves of >= 1.5 Mpixel per launch the extrema scan skips image rows
                                           that the blur kernels flagged as unable to hold a candidate (no |DoG| above 0.8 x
                                           dog_threshold); same candidates, keypoints and descriptors, but the raw_extrema
                                           statistic then counts tested rows only (siftmi_stats.raw_extrema_exact = 0).
                                           1 = scan every row, exact raw_extrema. */
    int32_t blur_march_min_blocks;      /* layer launches of at least this many workgroups (128-column strips x 160- or 256-row
                                           chunks x frames) use the marching ring blur (default 800; 1 = always, for tests) */
    int32_t blur_chain_max_tiles;       /* an octave of at most this many 64 x 64 tiles (frames x tiles) gets its Gaussian layers 1-3 and
                                           4-5 from ONE launch each (single frames: fewer dependent launches per call).  0 = default
                                           (256: the 960x540 and 480x270 octaves of one 1920x1080 frame), -1 = off.  Default schedule only (nspo = 3,
                                           taps 11 ... 27), octave width a multiple of 4, at least 64 x 64 */
    int32_t graph_fork;                 /* the captured launch sequence of siftmi_detect_describe_batch_device forks into one chain per
                                           octave (octave k's scan and keypoint stages beside octave k+1's pyramid): 0 = default (on,
                                           unless a frame's first octave exceeds 48 Mpixel), 1 = always, -1 = never (every kernel
                                           alone on the GPU: what a per-kernel profile wants) */
    int32_t descriptor_patch_lds;       /* 1 = the descriptor kernel of large launches stages every 16 x 16-sample tile of a window (18 x 18
                                           texels) in LDS and samples from there instead of reading the layer through the caches; the
                                           descriptors are byte-identical, the stage is slower (profiles/desc_patch_lds_r05.log): 0 = default */
    int32_t reserved[1];
} siftmi_config;

/* Replaces SIFTExtremaResult (Sources/MetalShaders/include/SIFTExtrema.h:14-18). */
typedef struct siftmi_extremum { int32_t x, y, scale; } siftmi_extremum;

/* Replaces SIFTKeypoint (SIFT/SIFTKeypoint.swift:11-57), flattened to 11 x 4 bytes. */
typedef struct siftmi_keypoint {
    int32_t octave;
    int32_t scale;
    float   sub_scale;
    int32_t x, y;                       /* scaledCoordinate     */
    float   abs_x, abs_y;               /* absoluteCoordinate   */
    float   norm_x, norm_y;             /* normalizedCoordinate */
    float   sigma;
    float   value;
} siftmi_keypoint;                      /* 44 bytes */

/* Replaces SIFTOrientationResult (Sources/MetalShaders/include/SIFTOrientation.h:30-34). */
typedef struct siftmi_orientation {
    int32_t keypoint;                   /* index into the octave's keypoint list                   */
    int32_t count;                      /* -1: rejected by the border filter (SIFTOctave.swift:311-329) */
    float   orientations[SIFTMI_ORIENTATION_BINS];
} siftmi_orientation;                   /* 152 bytes */

/* Replaces SIFTDescriptorResult (Sources/MetalShaders/include/SIFTDescriptor.h:37-42) /
   SIFTDescriptor (SIFT/SIFTDescriptor.swift:12-34).  Features are the reference's 0..255 integers
   stored as bytes; siftmi_descriptor_to_reference() widens to the 524-byte Int32 record. */
typedef struct siftmi_descriptor {
    int32_t keypoint;                   /* index into the octave's keypoint list                   */
    float   theta;
    uint8_t features[SIFTMI_DESCRIPTOR_FEATURES];
} siftmi_descriptor;                    /* 136 bytes */

typedef struct siftmi_descriptor_reference {   /* byte-for-byte SIFTDescriptorResult */
    int32_t valid;
    int32_t keypoint;
    float   theta;
    int32_t features[SIFTMI_DESCRIPTOR_FEATURES];
} siftmi_descriptor_reference;          /* 524 bytes */

/* Counters of the last call (replaces the os.Logger counts, SIFT.swift:186). [frame][octave] */
typedef struct siftmi_stats {
    int32_t n_frames, n_octaves;
    const int32_t *raw_extrema;         /* all strict 3-D extrema                                  */
    const int32_t *candidates;          /* extrema passing the 0.8*threshold / border pre-filter   */
    const int32_t *keypoints;
    const int32_t *oriented;            /* keypoints passing the orientation border filter         */
    const int32_t *descriptors;
    int32_t raw_extrema_exact;          /* 1: raw_extrema counts every strict extremum, like the reference's counter;
                                           0: the extrema scan skipped rows the blur flagged inactive (count_raw_extrema = 0
                                           on a large launch), so raw_extrema counts the tested rows only.  candidates,
                                           keypoints, oriented and descriptors are exact either way. */
} siftmi_stats;

typedef struct siftmi_ctx siftmi_ctx;

/* --- lifecycle: replaces SIFT.init(device:configuration:) (SIFT/SIFT.swift:112-143) ---------- */
int  siftmi_default_config(siftmi_config *cfg, int32_t width, int32_t height);
int  siftmi_create(const siftmi_config *cfg, int hip_device, siftmi_ctx **out);
void siftmi_destroy(siftmi_ctx *ctx);
const char *siftmi_last_error(void);
int  siftmi_device_count(void);

/* --- SIFT.getKeypoints(_:) (SIFT/SIFT.swift:147-152) ------------------------------------------
   pixels: host pointer (on_device = 0) or device pointer (on_device = 1), `format`, rows
   `row_stride` bytes apart.  Keypoints come back grouped by octave (counts[o], o < n_octaves),
   each group sorted by (scale, y, x) -- the reference's order is arbitrary (atomics).
   The pyramid stays resident in the context for a following siftmi_describe. */
int siftmi_detect(siftmi_ctx *ctx, const void *pixels, int format, size_t row_stride, int on_device,
                  const siftmi_keypoint **keypoints, int32_t *counts);

/* --- SIFT.getDescriptors(keypointOctaves:) (SIFT/SIFT.swift:207-238) ---------------------------
   keypoints grouped by octave as returned by siftmi_detect (the caller may have filtered them).
   descriptor.keypoint indexes the octave's group of the INPUT list. */
int siftmi_describe(siftmi_ctx *ctx, const siftmi_keypoint *keypoints, const int32_t *counts,
                    const siftmi_descriptor **descriptors, int32_t *desc_counts);

/* --- detect + describe for a batch of frames, no host round trip between the two -------------
   Frame f starts at pixels + f * frame_stride.  Results are dense, ordered by (frame, octave):
   kp_counts / desc_counts are [n_frames][n_octaves].  Frames are processed max_batch at a time. */
int siftmi_detect_describe_batch(siftmi_ctx *ctx, int32_t n_frames, const void *pixels, int format,
                                 size_t row_stride, size_t frame_stride, int on_device,
                                 const siftmi_keypoint **keypoints, const int32_t **kp_counts,
                                 const siftmi_descriptor **descriptors, const int32_t **desc_counts);

/* Same, everything staying in HBM (input and output are device pointers supplied by the caller,
   e.g. torch tensors that a following RCCL all-gather reads).  d_counts receives
   [2][n_frames][n_octaves] int32 (keypoints, then descriptors); d_totals receives FOUR int32:
   {n_kp, n_desc, overflow_flags, 0}.  overflow_flags != 0 means a list was truncated (bit 0 max_extrema, 1 max_keypoints,
   2 max_descriptors, 3 kp_capacity, 4 desc_capacity) -- the condition the host-facing entry points report as
   SIFTMI_E_CAPACITY; the results that fit are valid.  Bit 5: a SIFTMI_FMT_GRAYF32 frame held a value outside [0, 1]
   (SIFTMI_E_BADARG on the host-facing entry points; results unusable, see siftmi_format).
   Asynchronous on `stream` (a hipStream_t, NULL = the context's stream); no host sync.  The context's scratch is shared by
   all entry points: a following call on this context (any entry point, any stream) is ordered after this one on the
   device, and the introspection calls wait for it. */
int siftmi_detect_describe_batch_device(siftmi_ctx *ctx, int32_t n_frames, const void *d_pixels, int format,
                                        size_t row_stride, size_t frame_stride,
                                        siftmi_keypoint *d_keypoints, int64_t kp_capacity,
                                        siftmi_descriptor *d_descriptors, int64_t desc_capacity,
                                        int32_t *d_counts, int32_t *d_totals, void *stream);

/* Pinned (page-locked) host memory for frames fed from the host.  The reference hands the GPU an MTLTexture in unified
   memory; on a discrete GPU the frames cross PCIe, and copies from pinned memory are asynchronous and roughly twice as
   fast as from pageable memory.  The batch entry points overlap the copy of one sub-batch with the kernels of the
   previous one either way. */
int siftmi_host_alloc(size_t bytes, void **ptr);
int siftmi_host_free(void *ptr);

/* Device memory for callers without a HIP toolchain of their own (a Swift / C host): frames that stay resident in HBM.
   kind for siftmi_memcpy: 0 host -> device, 1 device -> host, 2 device -> device; synchronous. */
int siftmi_device_alloc(int hip_device, size_t bytes, void **ptr);
int siftmi_device_free(void *ptr);
int siftmi_memcpy(void *dst, const void *src, size_t bytes, int kind);
int siftmi_device_synchronize(int hip_device);

/* ================================================================================================================
   Frame stream: batches of frames ("steps") through detect + describe with several steps in flight.

   Replaces the reference's single synchronous queue (SIFT/SIFT.swift:139-175: one MTLCommandQueue, commit +
   waitUntilCompleted per stage) for a caller that feeds a stream of frames.  A stream borrows a context and creates
   steps_in_flight - 1 more with the same configuration (contexts are not re-entrant, independent ones run
   concurrently): consecutive steps alternate between them, each on its own launch stream, so the HBM-bound dense
   stages of step k+1 run under the VALU-bound keypoint stages of step k.  Packed results go to `result_sets` rotating
   device buffer sets (step k -> set k mod result_sets), so the results of step k stay valid until step
   k + result_sets is submitted.  Nothing here blocks the host except the *_host result call and the explicit waits.

   Stream arguments (`void *..._stream`) are hipStream_t handles: NULL is the device's legacy default stream,
   SIFTMI_NO_STREAM means "no ordering wanted".
   Like a context, a stream is not re-entrant: one host thread drives it.  The borrowed context must not be used for
   other calls while steps are in flight (its scratch is the stream's), and must outlive the stream.  Defaults hold
   32 768 keypoints + 49 152 descriptors per frame in each result set (64 frames: 0.5 GB per set) beside one pyramid
   per context; every size is in siftmi_stream_config. */
#define SIFTMI_NO_STREAM ((void *)(intptr_t)-1)

typedef struct siftmi_stream siftmi_stream;

typedef struct siftmi_stream_config {
    int32_t frames_per_step;            /* F: frames per submit (a step); processed max_batch at a time              */
    int32_t steps_in_flight;            /* 1 ... 4 contexts (default 2)                                               */
    int32_t result_sets;                /* 0 = 2 x steps_in_flight; rounded up to a multiple of steps_in_flight       */
    int32_t format;                     /* siftmi_format of every frame of the stream                                 */
    int64_t kp_per_frame;               /* capacity of the packed outputs per frame (0 = 32768 / 49152)               */
    int64_t desc_per_frame;
    int32_t staging_buffers;            /* submit_host: device staging buffers (0 = as many as result sets)          */
    int32_t density_mode;               /* which of two equivalent launch sequences a step replays: 0 = chosen per step from the
                                           descriptor totals of an earlier step (frames dense with keypoints run ONE chain without
                                           the extrema scan's activity flags, sparse ones fork into per-octave chains with them);
                                           1 = always the sparse form, 2 = always the dense form.  The records do not depend on it
                                           (tests/test_gpu_parity.py::test_stream_density_hint_flip_is_byte_identical); only
                                           siftmi_stats.raw_extrema[_exact] does, as count_raw_extrema describes */
    int32_t reserved[6];
} siftmi_stream_config;

/* packed results of one step.  Device view: everything in HBM, nothing synchronised. */
typedef struct siftmi_step_device {
    int64_t step;                       /* step number (0, 1, ... in submit order)                                    */
    const siftmi_keypoint *keypoints;   /* kp_capacity records, the first totals[0] valid, ordered (frame, octave)    */
    const siftmi_descriptor *descriptors;
    const int32_t *counts;              /* [2][frames_per_step][n_octaves]: keypoints, then descriptors               */
    const int32_t *totals;              /* {n_keypoints, n_descriptors, overflow_flags, 0}                            */
    int64_t kp_capacity, desc_capacity;
} siftmi_step_device;

/* Host view (page-locked memory owned by the stream, valid until the step's result set is reused). */
typedef struct siftmi_step_host {
    int64_t step;
    const siftmi_keypoint *keypoints;
    const siftmi_descriptor *descriptors;
    const int32_t *counts;              /* [2][frames_per_step][n_octaves]                                            */
    int32_t n_keypoints, n_descriptors;
    int32_t overflow_flags;             /* != 0: a list was truncated (bits as siftmi_detect_describe_batch_device)   */
    int32_t launch_flags;               /* how this step was launched, SIFTMI_STEP_* bits (diagnostic; the records do not depend on it) */
} siftmi_step_host;
#define SIFTMI_STEP_DENSE_HINT   1      /* launched under the density hint: one chain, no activity flags, full extrema scan        */
#define SIFTMI_STEP_GRAPH_REPLAY 2      /* the launch sequence came from a captured hipGraph (else: direct launches)                */
#define SIFTMI_STEP_FORKED       4      /* ... which forks into per-octave chains                                                  */
#define SIFTMI_STEP_RAW_EXACT    8      /* siftmi_stats.raw_extrema_exact of this step's context after the step                    */

int  siftmi_stream_default_config(siftmi_stream_config *scfg, int32_t frames_per_step);
/* ctx is borrowed (it stays the caller's and must outlive the stream); ctx->max_batch frames run in lock-step. */
int  siftmi_stream_create(siftmi_ctx *ctx, const siftmi_stream_config *scfg, siftmi_stream **out);
void siftmi_stream_destroy(siftmi_stream *s);
/* the i-th context of the stream (0 = the borrowed one) for the introspection / timing calls */
siftmi_ctx *siftmi_stream_context(siftmi_stream *s, int i);

/* One step on frames resident in HBM.  The kernels are ordered after the work already enqueued on producer_stream
   (the stream that wrote the frames; SIFTMI_NO_STREAM = frames are ready).  The frames must stay untouched until the
   step has run (siftmi_stream_result_*, or siftmi_stream_wait_consumed). */
int siftmi_stream_submit_device(siftmi_stream *s, const void *d_pixels, size_t row_stride, size_t frame_stride,
                                void *producer_stream, int64_t *step);
/* One step on frames in page-locked host memory (siftmi_host_alloc).  The upload goes to a rotating device staging
   buffer on a copy stream and is ordered only after the step that last read that buffer: the PCIe transfer of step
   k+1 runs under the kernels of step k.  `pixels` must stay untouched until siftmi_stream_wait_upload(step). */
int siftmi_stream_submit_host(siftmi_stream *s, const void *pixels, size_t row_stride, size_t frame_stride, int64_t *step);
/* host blocks until the frames of `step` have left host memory (submit_host) / have been read by the kernels (wait_consumed) */
int siftmi_stream_wait_upload(siftmi_stream *s, int64_t step);
int siftmi_stream_wait_consumed(siftmi_stream *s, int64_t step);

/* Results of the step `back` steps before the last submitted one (0 = the last).  Device view: consumer_stream is
   ordered after the step on the device (SIFTMI_NO_STREAM: no ordering); no host synchronisation. */
int siftmi_stream_result_device(siftmi_stream *s, int back, siftmi_step_device *out, void *consumer_stream);
/* Host view: blocks until the step has finished and its packed records are in page-locked host memory.  While the
   caller keeps reading results on the host (a call of this function between two submits) the copy of a step is
   started when the step is submitted (sized from the last counts read, completed here if the step found more), so a
   pipelined consumer that reads step k after submitting step k+1 ... k+steps_in_flight finds it done.
   Returns SIFTMI_E_CAPACITY (with valid truncated results in *out) if overflow_flags != 0. */
int siftmi_stream_result_host(siftmi_stream *s, int back, siftmi_step_host *out);
/* everything submitted so far has finished (host blocks) */
int siftmi_stream_synchronize(siftmi_stream *s);
/* siftmi_stream_config.density_mode of the steps submitted from now on (0 automatic, 1 sparse form, 2 dense form) */
int siftmi_stream_set_density_mode(siftmi_stream *s, int mode);

/* ================================================================================================================
   Result exchange between the GPUs of a node: RCCL all-gather of every rank's packed results (the one exchange step
   of the path: frames are sharded frame-per-GPU, nothing else crosses GPUs).  One process per GPU; librccl is loaded
   at the first siftmi_exchange_* call (no dependency for single-GPU users).

   Per step: ncclAllGather of the totals, then padded all-gathers of counts, keypoint and descriptor bytes on a side
   stream, reading the step's result set while the next step's kernels write another.  The payload sizes of step k
   come from the totals of step k-1 (+ headroom), which are read on the host while step k runs: no host
   synchronisation between a step's kernels and its collectives.  If a rank's counts outgrow what was sent (a scene
   cut), the step is re-gathered in full by the NEXT siftmi_exchange_gather / siftmi_exchange_finish call, before its
   result set can be reused; siftmi_exchange_result reports `complete`.

   Failure: a collective completes only if every rank takes part.  No host wait of the exchange is unbounded: each polls its
   event and the communicator's asynchronous error state until a deadline (SIFTMI_EXCHANGE_TIMEOUT_S seconds per wait, default
   120; siftmi_exchange_set_timeout), then aborts the communicator (ncclCommAbort: collectives stuck on the device exit, the
   streams ordered behind them drain) and returns SIFTMI_E_HIP with the rank and the step in siftmi_last_error().  An aborted
   exchange fails every later call the same way; the caller ends the job (nothing is restarted).  The reference has no
   counterpart: it drives one device (SIFT/SIFT.swift:139). */
#define SIFTMI_UNIQUE_ID_BYTES 128
typedef struct siftmi_exchange siftmi_exchange;

typedef struct siftmi_gathered {
    int64_t step;
    int32_t world, complete;            /* complete = 0: some rank held more records than were sent (see above)      */
    const uint8_t *keypoints;           /* device: [world][kp_stride] bytes; rank r's first totals[r][0] records valid */
    const uint8_t *descriptors;         /* device: [world][desc_stride] bytes                                         */
    const int32_t *counts;              /* device: [world][2][frames_per_step][n_octaves]                              */
    const int32_t *totals_device;       /* device: [world][4]                                                         */
    const int32_t *totals_host;         /* host:   [world][4], valid when `resolved` != 0                             */
    int64_t kp_stride, desc_stride;     /* bytes between ranks = records sent per rank x record size                  */
    int64_t kp_records, desc_records;   /* records sent per rank                                                       */
    int32_t resolved, reserved;
} siftmi_gathered;
/* The pointers and strides of a siftmi_gathered stay valid until the second siftmi_exchange_gather / _finish call after the
   one that produced them (two buffer sets; a set that has to grow is replaced, the old block is kept until then). */

/* An exchange belongs to one stream: destroy it before the stream.  Its calls marked "collective" must be made by every
   rank in the same order (they enqueue RCCL collectives).
   rank 0 creates the id and hands it to the other ranks out of band (a file, a socket, MPI, torch.distributed ...) */
int  siftmi_exchange_unique_id(void *id /* SIFTMI_UNIQUE_ID_BYTES */);
/* collective over the `world` ranks: ncclCommInitRank on the stream's device */
int  siftmi_exchange_create(siftmi_stream *s, const void *unique_id, int rank, int world, siftmi_exchange **out);
/* bounded like every wait of the exchange: a healthy exchange drains its pending gathers (within the timeout) and frees everything; a
   FAILED one (a wait expired, the communicator reported an error) is aborted, and if its stream still has not drained by the deadline
   its buffers, stream and communicator are leaked rather than waited for (siftmi_last_error() says so) */
void siftmi_exchange_destroy(siftmi_exchange *x);
/* what the communicator itself reports (ncclCommCount / ncclCommUserRank); siftmi_exchange_create has checked both against its
   arguments */
int  siftmi_exchange_ranks(siftmi_exchange *x, int32_t *comm_ranks, int32_t *comm_rank);
/* deadline of every host wait of this exchange, in seconds (default: SIFTMI_EXCHANGE_TIMEOUT_S, else 120) */
int  siftmi_exchange_set_timeout(siftmi_exchange *x, double seconds);
/* host blocks, bounded as above, until every collective enqueued so far has finished (call it before a device-wide
   synchronisation: that one has no deadline); not collective */
int  siftmi_exchange_wait(siftmi_exchange *x);
/* collective: all-gather the results of the last submitted step (side stream, no host synchronisation unless
   synchronous != 0, which sizes the payloads from this step's own totals) */
int  siftmi_exchange_gather(siftmi_exchange *x, int synchronous);
/* gathered results of the step `back` gathers ago (0 = the last, 1 = the one before: two buffer sets);
   consumer_stream is ordered after the gather.  wait_host != 0: block until the gather has finished and the step's
   totals are on the host (resolved = 1, complete is final). */
int  siftmi_exchange_result(siftmi_exchange *x, int back, siftmi_gathered *out, void *consumer_stream, int wait_host);
/* collective, end of stream: resolves (and if needed re-gathers) the last step; returns how many steps had to be
   re-gathered and how many reported list overflow on some rank */
int  siftmi_exchange_finish(siftmi_exchange *x, int64_t *regathered_steps, int64_t *overflow_steps);
/* sizing rule of the following gathers: next size = (1 + headroom_percent / 100) x the largest count of the last resolved
   step, rounded up to a multiple of quantum (defaults 25, 1024); the same on every rank */
int  siftmi_exchange_set_headroom(siftmi_exchange *x, int32_t headroom_percent, int64_t quantum);
/* accumulated GPU time of the gathers (side-stream hipEvents) and their number since creation; bytes received per gather */
int  siftmi_exchange_stats(siftmi_exchange *x, double *ms, int64_t *gathers, int64_t *bytes_last);
/* which library carries the collectives: the path / soname the eleven ncclXxx entry points were resolved from (SIFTMI_RCCL_LIB
   if set, else an RCCL already mapped into the process, else librccl.so[.1]); "" if none could be loaded.  Loads it. */
const char *siftmi_exchange_transport(void);

/* The sizing rule of the exchange alone (host arithmetic, no GPU, no RCCL): what siftmi_exchange_gather uses, exposed so
   that the rule can be driven over any transport (tests/test_dist_gloo.py drives it over gloo on CPU). */
typedef struct siftmi_gather_plan {
    int64_t kp_capacity, desc_capacity; /* records per rank that the result buffers can hold                          */
    int64_t send_kp, send_desc;         /* records per rank in the next payload gathers; -1 = not known yet           */
    int64_t quantum;                    /* sizes are rounded up to a multiple of this (1024)                          */
    int32_t headroom_percent;           /* 25: next size = 1.25 x the largest count seen in the last resolved step    */
    int32_t reserved;
    int64_t steps_resolved, steps_incomplete, steps_overflowed;
} siftmi_gather_plan;
int siftmi_gather_plan_init(siftmi_gather_plan *p, int64_t kp_capacity, int64_t desc_capacity);
/* feed one step's totals of all ranks ([world][4] host ints) and what was sent for it; returns 1 if the step was
   incomplete (some rank held more than was sent), 0 if not; updates send_kp / send_desc for the next step */
int siftmi_gather_plan_resolve(siftmi_gather_plan *p, const int32_t *totals, int world, int64_t sent_kp, int64_t sent_desc);

/* --- next row: SIFTDescriptor.match(source:target:absoluteThreshold:relativeThreshold:)
   (SIFT/SIFTDescriptor.swift:298-361), brute force + ratio test.  siftmi_match replaces
   SIFTCorrespondence (SIFT/SIFTCorrespondence.swift:11-16) with indices into the two input lists.
   Matches come back in source order.  Pointers are host (on_device = 0) or device (1) memory.
   The reference's defaults are absoluteThreshold 1.176, relativeThreshold 0.6. */
typedef struct siftmi_match { int32_t source, target; float distance; } siftmi_match;
int siftmi_match_descriptors(siftmi_ctx *ctx, const siftmi_descriptor *source, int64_t n_source,
                             const siftmi_descriptor *target, int64_t n_target, int on_device,
                             float absolute_threshold, float relative_threshold,
                             const siftmi_match **matches, int64_t *count);
/* The same match with everything staying in HBM: descriptors in device memory, the records of the matched sources packed in source
   order into d_matches (device, capacity n_source records) and their number into *d_count (device); asynchronous on `stream`
   (hipStream_t, NULL = the context's stream), no host synchronisation -- a consumer on the device (RANSAC, a tracker) reads them from
   there, a host reads *d_count when it needs it.  Scratch is the context's: calls on one context are ordered on the device.
   The FIRST call of a context, and any call whose sizes outgrow the scratch of the earlier ones, reallocates it (hipFree / hipMalloc:
   a device-wide synchronisation, once); size the first call for the largest sets to come and later calls enqueue without one. */
int siftmi_match_descriptors_device(siftmi_ctx *ctx, const siftmi_descriptor *d_source, int64_t n_source,
                                    const siftmi_descriptor *d_target, int64_t n_target,
                                    float absolute_threshold, float relative_threshold,
                                    siftmi_match *d_matches, int32_t *d_count, void *stream);
/* How siftmi_match_descriptors cuts a problem of this size (no reference counterpart; for tests and tuning): targets per chunk, chunks,
   and whether the chunks start from a bound (a pre-pass over the first 512 targets + the bests earlier chunks have published)
   instead of from "no best".  The results do not depend on any of it. */
int siftmi_match_plan(int64_t n_source, int64_t n_target, int64_t *split_len, int64_t *n_split, int *bounded);

/* SIFTDescriptor.approximateMatch(source:target:absoluteThreshold:relativeThreshold:) (SIFT/SIFTDescriptor.swift:362-417)
   over the reference's ANN trie (Utilities/Trie.swift:76-416): Trie(numberOfBins: 8) keyed by indexKey, radius 10, k 2.
   Same results as the reference's pointer trie (the sorted path codes are its leaf ring).  Distances here are
   IntVector.distance of the 0..255 features (not / 255): the reference's default absoluteThreshold is 300. */
int siftmi_approximate_match(siftmi_ctx *ctx, const siftmi_descriptor *source, int64_t n_source,
                             const siftmi_descriptor *target, int64_t n_target, int on_device,
                             float absolute_threshold, float relative_threshold,
                             const siftmi_match **matches, int64_t *count);

/* SIFTDescriptor.matchGeometry(source:target:absoluteThreshold:relativeThreshold:) (SIFT/SIFTDescriptor.swift:104-144;
   compareGeometry :162-296): match on the GPU, then the geometric-consistency score of the first 80 matches
   (0 with fewer than 7 matches).  *_xy: the descriptors' keypoint absoluteCoordinate, [n][2] floats (x, y). */
int siftmi_match_geometry(siftmi_ctx *ctx, const siftmi_descriptor *source, const float *source_xy, int64_t n_source,
                          const siftmi_descriptor *target, const float *target_xy, int64_t n_target,
                          float absolute_threshold, float relative_threshold, float *score, int64_t *n_matches);

/* SIFTDescriptor.init's derived vectors (SIFT/SIFTDescriptor.swift:36-89), host memory, any output may be NULL:
   raw_features [n][128] = features / 255; index_value [n][128] = the 16 cells re-ordered centre, corners, edges;
   index_key [n][16] = the mean of each re-ordered cell. */
int siftmi_descriptor_index(const siftmi_descriptor *descriptors, int64_t n, float *raw_features, float *index_value,
                            float *index_key);

/* --- record conversion (SIFTOctave.swift:470-489 host unpack) -------------------------------- */
void siftmi_descriptor_to_reference(const siftmi_descriptor *in, int64_t n, siftmi_descriptor_reference *out);

/* --- introspection / parity hooks (state of the last detect/describe call) -------------------- */
int siftmi_get_stats(siftmi_ctx *ctx, siftmi_stats *out);
/* What the batched entry points (siftmi_detect_describe_batch[_device], the stream's submits) did with their launch sequences since
   the context was created: sequences captured into a hipGraph, replayed from one, issued as direct launches (first sighting of a call
   signature, graphs off, timings on).  last_flags: bit 0 the last sequence was a replay, bit 1 it was forked into per-octave chains,
   bit 2 the context's density hint.  Any output may be NULL.  No reference counterpart (it has no launch graphs). */
int siftmi_graph_stats(siftmi_ctx *ctx, int64_t *captures, int64_t *replays, int64_t *direct_sequences, int32_t *last_flags);
int siftmi_octave_size(siftmi_ctx *ctx, int octave, int32_t *w, int32_t *h, float *delta);
int siftmi_get_sigma(siftmi_ctx *ctx, int octave, int scale, float *sigma);
int siftmi_get_weights(siftmi_ctx *ctx, int layer /*0 = seed, 1..nspo+2*/, float *weights, int32_t *count);
/* Gaussian layer G[octave][layer] of `frame` (of the last sub-batch) -> host, dense [h][w] f32 */
int siftmi_copy_gaussian(siftmi_ctx *ctx, int frame, int octave, int layer, float *dst);
/* DoG layer D[octave][scale] = G[scale + 1] - G[scale], scale < nspo + 2 -> host, dense [h][w] f32: the texture the
   reference's DifferenceOfGaussians exposes (SIFT/DifferenceOfGaussians.swift:20, 346-406; Metal/Subtract.metal:12-21)
   and DifferenceOfGaussiansTests.swift:15-270 diffs against DoG_butterfly_o*_s*.png */
int siftmi_copy_dog(siftmi_ctx *ctx, int frame, int octave, int scale, float *dst);
/* candidate list of (frame, octave), sorted by (scale, y, x) on the host */
int siftmi_copy_extrema(siftmi_ctx *ctx, int frame, int octave, siftmi_extremum *dst, int32_t cap, int32_t *count);
int siftmi_copy_orientations(siftmi_ctx *ctx, int frame, int octave, siftmi_orientation *dst, int32_t cap, int32_t *count);
/* 128 pre-quantisation floats per descriptor (needs keep_descriptor_floats) */
int siftmi_copy_descriptor_floats(siftmi_ctx *ctx, int frame, int octave, float *dst, int32_t cap, int32_t *count);

/* --- timing (replaces Utilities/Performance.swift measure(name:) signposts) -------------------
   Stage ids for siftmi_get_timings: accumulated GPU milliseconds and launch counts since the
   last siftmi_reset_timings, measured with hipEvents on the context's stream when enabled. */
enum { SIFTMI_T_SEED = 0, SIFTMI_T_BLUR = 1, SIFTMI_T_DOWNSAMPLE = 2 /* always 0: fused into the layer-nspo blur */, SIFTMI_T_EXTREMA = 3,
       SIFTMI_T_REFINE = 4, SIFTMI_T_SORT = 5, SIFTMI_T_ORIENT = 6, SIFTMI_T_DESCRIBE = 7,
       SIFTMI_T_PACK = 8, SIFTMI_T_COUNT = 9 };
int siftmi_enable_timings(siftmi_ctx *ctx, int enable);
int siftmi_reset_timings(siftmi_ctx *ctx);
int siftmi_get_timings(siftmi_ctx *ctx, double *ms /*[SIFTMI_T_COUNT]*/, int64_t *launches /*[SIFTMI_T_COUNT]*/);
/* the SIFTMI_T_BLUR time split by (octave, layer 1..nspo+2): accumulated ms and launch count of that layer's blur launches
   since the last reset -- one kernel instantiation and grid size each, so that a rocprofv3 kernel trace of the same command
   can be compared launch shape by launch shape; *marching: bit 0 = those launches use blur_ring_kernel (else blur2_kernel), bit 1 = they
   also write the extrema scan's activity flags, bit 2 = the octave's layers come from blur_chain_kernel: TWO launches produce its five layers, and their time and launch count
   are booked on layers 1 (covering layers 1-3 and the next octave's layer 0) and 4 (layers 4-5) -- layers 2, 3 and 5 of such an
   octave report 0 ms / 0 launches */
int siftmi_get_blur_layer_timings(siftmi_ctx *ctx, int octave, int layer, double *ms, int64_t *launches, int32_t *marching);
/* algorithmic bytes one blur launch of `octave` moves for ONE frame: 8 B per octave pixel */
int64_t siftmi_blur_algorithmic_bytes(siftmi_ctx *ctx, int octave);
/* runs the Gaussian-layer blur launch of the pipeline alone (layer 1..nspo+2 of `octave`, all max_batch frames, with the
   decimated / activity-flag outputs the pipeline gives that layer) `iters` times on the resident pyramid and returns the
   mean kernel time in ms (hipEvents) */
int siftmi_time_blur(siftmi_ctx *ctx, int octave, int layer, int iters, double *ms_per_launch);
/* The two measured ceilings bench.py quotes the pyramid kernel against ("verify both peaks on the box"):
   siftmi_time_copy        a plain float4 streaming copy inside the context's pyramid memory (`bytes` read and `bytes` written
                           per launch, clipped to half the pyramid; *bytes_moved = read + written): the HBM rate a kernel
                           with no arithmetic, no LDS and no halo reaches on THIS device now.  Invalidates the pyramid.
   siftmi_time_blur_memory the same launch as siftmi_time_blur with both passes' arithmetic compiled out (same loads, LDS
                           staging, barriers and stores; results are garbage): what the ring kernel's memory side alone
                           sustains.  Default schedule's radii (5, 7, 8, 10, 13) on octaves that use the marching kernel only
                           (SIFTMI_E_STATE otherwise).  Invalidates the pyramid. */
int siftmi_time_copy(siftmi_ctx *ctx, int64_t bytes, int iters, double *ms_per_launch, int64_t *bytes_moved);
int siftmi_time_blur_memory(siftmi_ctx *ctx, int octave, int layer, int iters, double *ms_per_launch);
int siftmi_synchronize(siftmi_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* SIFTMI_H */
