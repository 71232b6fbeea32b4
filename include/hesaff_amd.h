/* hesaff_amd.h -- C ABI of libhesaff_amd.so: MI355X (gfx950) Hessian-Affine + SIFT hot path.
 *
 * Drop-in boundary for perdoch/hesaff's detect+describe path.  Plain pointers and sizes
 * only; no C++/torch types.  Every entry point cites the reference interface it replaces
 * (file:line under the reference tree).  Return value: 0 = HESAFF_OK, negative = error;
 * hesaff_last_error() gives the message.  One context per device; a context is not
 * thread-safe, distinct contexts may be used concurrently.
 *
 * The library has NO CPU fallback: every function that computes needs a visible gfx950
 * device and fails with HESAFF_ERR_DEVICE otherwise.
 */
#ifndef HESAFF_AMD_H
#define HESAFF_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HESAFF_OK 0
#define HESAFF_ERR_DEVICE (-1)    /* no usable GPU / HIP runtime error            */
#define HESAFF_ERR_ARG (-2)       /* bad argument                                 */
#define HESAFF_ERR_CAPACITY (-3)  /* keypoint capacity exceeded (raise max_kpts_per_mpx) */
#define HESAFF_ERR_IO (-4)        /* file I/O                                     */
#define HESAFF_ERR_NOMEM (-5)

/* Layout version of the structs below (hesaff_params, hesaff_timings, hesaff_result, hesaff_file_status).  Neither struct
 * carries a size field, so a caller built against another header would pass shifted fields without any error: callers
 * compare hesaff_abi_version() (and, if they wish, the two sizeof functions) with the header they were compiled against
 * before the first hesaff_create - hesaff.hpp and the Python binding do.  New fields are appended at the END of a
 * struct and bump this number.   1: round 1;  2: + upscaleInputImage, fast, pack_ms (inserted mid-struct);  3: + extrema_*;
 * 4: hesaff_params.fast = 1 withdrawn, HESAFF_FILE_REJECTED, rows formatted on the device;
 * 5: + hesaff_jpeg_layout, hesaff_read_jpeg_coefficients, hesaff_stage_jpeg_pixels (JPEG pixels made on the device);
 * 6: + hesaff_host_plan_for, allocator-owned read buffers (hesaff_read_*_alloc);
 * 7: this header (+ hesaff_set_pinned_read_budget, hesaff_set_pool_priority, hesaff_stage_threads_for_pool, HESAFF_OUT_STRICT; hesaff_set_resume
 *    takes 0 / 1 / 2; no struct changed).
 * Image sizes: a side of at most 65535 pixels at the first pyramid level, and sqrt(width x height) of at most about 27900 (the largest window
 * normalizeAffine could ask for, affine.cpp:114-124, must fit a compute unit's LDS as one row): HESAFF_ERR_ARG beyond. */
#define HESAFF_ABI_VERSION 7
int hesaff_abi_version(void);
size_t hesaff_sizeof_params(void);
size_t hesaff_sizeof_timings(void);

typedef struct hesaff_ctx hesaff_ctx;

/* Parameters = the reference's compile-time structs flattened.
 * PyramidParams pyramid.h:18-41, AffineShapeParams affine.h:17-46,
 * SIFTDescriptorParams siftdesc.h:19-32, HessianAffineParams hesaff.cpp:21-36.
 * patchSize (41), smmWindowSize (19), spatialBins (4), orientationBins (8),
 * numberOfScales (3) and border (5) are fixed at the reference defaults. */
typedef struct hesaff_params {
   float threshold;            /* 16/3  pyramid.h:37, hesaff.cpp:30   */
   float edgeEigenValueRatio;  /* 10    pyramid.h:38                  */
   float initialSigma;         /* 1.6   pyramid.h:36 (also affine.h:40) */
   int maxIterations;          /* 16    affine.h:39, hesaff.cpp:31    */
   float convergenceThreshold; /* 0.05  affine.h:41                   */
   float mrSize;               /* 3*sqrt(3) affine.h:44, hesaff.cpp:32 */
   float maxBinValue;          /* 0.2   siftdesc.h:29                 */
   int upscaleInputImage;      /* 0     pyramid.h:34 (1: first octave on the 2x up-sampled image, helpers.cpp:297-329) */
   /* capacity knobs (no reference counterpart) */
   int max_batch;              /* images processed together on the device (default 64; hesaff_detect_batch pipelines chunks of this size) */
   int max_kpts_per_mpx;       /* candidate/keypoint capacity per megapixel (default 40000) */
   /* 0 (default): parity mode, results bit-identical to the reference's arithmetic.
    * 2: a different algorithm for the windows larger than the 41 x 41 patch (27 % of the keypoints, 3/4 of the patch stage's
    *    time): their samples are taken from the scale-space level whose blur matches (1681 taps) instead of warping and
    *    blurring a P x P window of the original (affine.cpp:114-135).  Detection, affine shapes, the set of described keypoints
    *    and the descriptors of the small windows stay those of parity mode; the other descriptors differ visibly, see DESIGN.md
    *    for the measured effect on descriptors and matching.
    * 1 was "the same algorithm with free summation order and approximate division" (ABI versions 2-3).  It bought 1.02x
    *    and is withdrawn: hesaff_create refuses it. */
   int fast;
} hesaff_params;

/* One detected + described region = the reference's `struct Keypoint` hesaff.cpp:41-48,
 * same field order and types (164 bytes). */
typedef struct hesaff_keypoint {
   float x, y, s;
   float a11, a12, a21, a22;
   float response;
   int32_t type;               /* HESSIAN_DARK 0 / BRIGHT 1 / SADDLE 2, pyramid.h:51-55 */
   uint8_t desc[128];
} hesaff_keypoint;

/* Per-image result of hesaff_detect_batch. keys is library-owned host memory, valid until
 * the next hesaff_detect_batch / hesaff_destroy on the same context.  Order = the
 * reference's detection order (octave, level, raster of the initial extremum). */
typedef struct hesaff_result {
   int32_t count_hessian;      /* g_numberOfPoints,       hesaff.cpp:38,68  */
   int32_t count_desc;         /* g_numberOfAffinePoints, hesaff.cpp:39,103 */
   const hesaff_keypoint *keys;
} hesaff_result;

/* Stage timings of the last device batch, milliseconds.  Every bracket is a HIP event pair recorded on the
 * stream its kernels are launched on.  pyramid / detect / pack run one after the other on the main stream; the
 * affine, patch and descriptor stages run CONCURRENTLY on their own streams over groups of images (three-deep
 * pipeline), so their three figures are per-stream busy times that overlap in wall-clock time: they add up to
 * more than total_ms - pyramid_ms - detect_ms - pack_ms. */
typedef struct hesaff_timings {
   float pyramid_ms;           /* grey + blur + Hessian + decimation (the roofline kernels) */
   float detect_ms;            /* extrema + localisation + ordering                          */
   float affine_ms;            /* Baumberg iteration, sum over groups (affine stream)        */
   float patch_ms;             /* rectify + normalizeAffine, sum over groups (main + bin streams) */
   float sift_ms;              /* descriptor kernels, sum over groups (descriptor stream)    */
   float pack_ms;              /* final stable compaction into hesaff_keypoint records       */
   float total_ms;
   float blur_hess_ms;         /* sum of the k_blur_hess launches only                       */
   int32_t blur_hess_launches;
   double blur_hess_bytes;     /* algorithmic bytes of those launches (12 N each, + 8 N with the fused R0, + 2 N with the fused decimation) */
   double pyramid_bytes;       /* algorithmic bytes B_pyr = 5 N0 + 58 sum N_k, whole batch   */
   /* appended in ABI version 3 (profiling level 2) */
   float extrema_ms;           /* sum of the k_extrema_march launches (3x3x3 extrema of the three scans of an octave) */
   int32_t extrema_launches;
   double extrema_bytes;       /* algorithmic bytes of those launches: 20 N per octave (five response planes read once), SURVEY.md 8d B_ext */
   /* appended in ABI version 4 */
   float export_ms;            /* hesaff_process_files: the export kernels (row lengths, offsets, text / sidecar rows) of the last chunk delivered */
   int32_t export_rows;        /* ... and the rows they formatted */
} hesaff_timings;

int hesaff_default_params(hesaff_params *p);

/* replaces: AffineHessianDetector ctor hesaff.cpp:56-64 (+ mask/table setup affine.h:71,
 * siftdesc.h:47-48).  device = HIP device ordinal. */
int hesaff_create(hesaff_ctx **out, const hesaff_params *p, int device);
void hesaff_destroy(hesaff_ctx *ctx);
const char *hesaff_last_error(const hesaff_ctx *ctx);   /* ctx may be NULL: last create error */

/* replaces: main()'s grey conversion hesaff.cpp:138-148 + detectPyramidKeypoints
 * hesaff.cpp:167 (pyramid.cpp:261) with the whole callback chain hesaff.cpp:66-105,
 * for n images at once.  images[i]: 8-bit, channels[i] = 1 (grey) or 3 (BGR as cv::imread
 * delivers; pass RGB bytes of a PPM in any order - the three are summed), row stride in
 * bytes.  Images of different sizes are allowed (grouped internally).  Chunks of max_batch images
 * are pipelined: host staging + H2D of the next chunk and D2H of the previous one overlap the
 * kernels of the current one.  results[i].keys point into library-owned pinned memory, valid
 * until the next call on this context. */
int hesaff_detect_batch(hesaff_ctx *ctx, int n, const uint8_t *const *images, const int *widths,
                        const int *heights, const int *strides, const int *channels, hesaff_result *results);

/* The same call with bounded host memory: results are handed to `sink` chunk by chunk (at most max_batch images at a
 * time, image_index[i] = position in the caller's arrays) and are valid only until sink returns; the library cycles
 * through three pinned result blocks however long the list is.  (hesaff_detect_batch keeps every chunk's block until
 * the next call: about 19 MB per dense UHD image, 39 GB for 2048 of them - use this form or hesaff_process_files for
 * long lists.)  A non-zero return of sink stops the run: HESAFF_ERR_IO. */
typedef int (*hesaff_chunk_sink)(void *user, int n_images, const int *image_index, const hesaff_result *results);
int hesaff_detect_batch_cb(hesaff_ctx *ctx, int n, const uint8_t *const *images, const int *widths, const int *heights,
                           const int *strides, const int *channels, hesaff_chunk_sink sink, void *user);

/* replaces: main() hesaff.cpp:133-180 for a list of image files - cv::imread (:137), grey conversion (:138-148),
 * detectPyramidKeypoints (:167), the output name <image>.hesaff.sift (:170-173) and exportKeypoints (:175) - as a
 * bounded three-stage host pipeline on one device: decode threads -> chunks of max_batch consecutive images of one
 * size through the device (copy in / kernels / copy out overlapped) -> writer threads.  Host memory stays bounded
 * (about 2 max_batch decoded images and three result blocks).  out_paths may be NULL (or hold NULLs): the reference's
 * name.  status[i].rc = HESAFF_OK, or why file i was skipped (unreadable input, unwritable output); one bad file does
 * not stop the others.  decode_threads / write_threads: 0 = auto (hesaff_host_plan_for(1): this context has the host to itself); the two counts add up to ONE pool of
 * host threads that decode when the look-ahead window has room and write otherwise.  The rows of the output files are formatted
 * on the device (hesaff.cpp:124-128 as a kernel): a writer only write()s what the copy engine delivered. */
#define HESAFF_FILE_PENDING 0   /* never reached (the run stopped on a device error before this file) */
#define HESAFF_FILE_UNREADABLE 1
#define HESAFF_FILE_DETECTED 2  /* detected and described, but the output file could not be written (rc says why) */
#define HESAFF_FILE_WRITTEN 3
#define HESAFF_FILE_REJECTED 4  /* decoded, but the device refused it: a side above 65535 pixels or a window the kernels cannot hold
                                   (rc = HESAFF_ERR_ARG), more keypoints than max_kpts_per_mpx plans for (HESAFF_ERR_CAPACITY); the other
                                   files of the list are not affected */
#define HESAFF_FILE_SKIPPED 5   /* hesaff_set_resume: the complete output of an earlier run exists; the image was not read (rc = HESAFF_OK,
                                   count_desc = the row count that output states, count_hessian = -1: not known) */
typedef struct hesaff_file_status {
   int32_t rc;                 /* HESAFF_OK only in stage HESAFF_FILE_WRITTEN */
   int32_t stage;              /* how far this file got: HESAFF_FILE_* */
   int32_t count_hessian;      /* g_numberOfPoints        hesaff.cpp:38 */
   int32_t count_desc;         /* g_numberOfAffinePoints  hesaff.cpp:39 */
} hesaff_file_status;
int hesaff_process_files(hesaff_ctx *ctx, int n, const char *const *paths, const char *const *out_paths, int decode_threads,
                         int write_threads, hesaff_file_status *status);
/* What hesaff_process_files writes for every image: HESAFF_OUT_TEXT (default) = <image>.hesaff.sift in the reference's text
 * format; HESAFF_OUT_BIN = <image>.hesaff.bin (hesaff_write_bin); HESAFF_OUT_TEXT | HESAFF_OUT_BIN = both.  With out_paths
 * given, the binary file is out_paths[i] + ".bin" when both are written, out_paths[i] itself when only the binary one is. */
#define HESAFF_OUT_TEXT 1
#define HESAFF_OUT_BIN 2
#define HESAFF_OUT_STRICT 0x100   /* hesaff_output_is_complete only: also count the rows of a text file (reads the whole file) */
int hesaff_set_output_format(hesaff_ctx *ctx, int format);
/* Resume a list that was interrupted (checkpoint / resume; no counterpart in the reference, which handles one
 * image per process): with on != 0 hesaff_process_files skips every image whose output file(s) of the selected format exist and
 * are complete - a .hesaff.sift with its two header lines, a plausible size and a final newline, a .hesaff.bin whose size matches its
 * row count (three small reads per file).  on = 2 (strict): the rows of an existing text file are counted as well - the whole file is
 * read, about 46 MB per dense 3840 x 2160 image, on the decode threads - for directories that a writer which does not rename its
 * outputs into place (the reference binary) may have left torn at a row boundary.  Every writer of this library puts its output under
 * a temporary name ("<name>.part.<pid>.<tid>") and renames it when it is complete, so a run that is killed never leaves a torn file
 * under the final name; a target that is not a regular file (/dev/stdout, a FIFO) or whose directory takes no new file is written in
 * place. */
int hesaff_set_resume(hesaff_ctx *ctx, int on);

/* Same path with inputs already resident in device memory (bench / pipelines that decode
 * on the GPU): d_gray = n contiguous height x width 8-bit grey planes (device pointer).
 * Results stay on the device; per-image counts are copied to the two host arrays.
 * d_keys_out (optional, may be NULL) receives a device pointer to the ordered
 * hesaff_keypoint array of the whole batch, total_out the number of records. */
int hesaff_detect_batch_device(hesaff_ctx *ctx, int n, const void *d_gray, int width, int height,
                               int32_t *count_hessian, int32_t *count_desc, const void **d_keys_out,
                               int64_t *total_out);
/* level 0: no events; 1: per-stage HIP events; 2: also one event pair per k_blur_hess launch */
int hesaff_set_profiling(hesaff_ctx *ctx, int level);
int hesaff_get_timings(const hesaff_ctx *ctx, hesaff_timings *t);

/* replaces: exportKeypoints hesaff.cpp:107-130 (text format README:27-44).
 * hesaff_ellipse: (a,b,c) of one region, closed form of the SVD expression hesaff.cpp:115-123. */
void hesaff_ellipse(const hesaff_keypoint *k, float mrSize, float *a, float *b, float *c);
int hesaff_write_sift(const char *path, const hesaff_keypoint *keys, int n, float mrSize);
/* the same file written by `threads` host threads (0 = auto); threads = 1 formats and writes block by block through a
 * cache-resident buffer (what the per-image workers of hesaff_write_sift_batch / hesaff_process_files do) */
int hesaff_write_sift_mt(const char *path, const hesaff_keypoint *keys, int n, float mrSize, int threads);
/* Binary sidecar of the same rows (no counterpart in the reference): the five floats unprinted and the
 * 128 descriptor bytes, 148 bytes per row instead of about 355 bytes of text.  Little-endian:
 *   char magic[8] = "HESAFFB1"; uint32 dim = 128; uint32 count; count x { float x, y, a, b, c; uint8 desc[128] } */
int hesaff_write_bin(const char *path, const hesaff_keypoint *keys, int n, float mrSize);
/* The same two files from rows that are already formatted / packed - hesaff_process_files formats them on the device
 * (kernels_export.h) so that its writer threads only write: `rows` = len bytes of "x y a b c d1 .. d128\n" lines (n of them) or
 * n rows of 148 bytes; the header lines (hesaff.cpp:109-110) / the 16-byte sidecar header are added here. */
int hesaff_write_sift_rows(const char *path, const char *rows, size_t len, int n);
int hesaff_write_bin_rows(const char *path, const char *rows, int n);
/* the row count of `path` when it is the complete output (format HESAFF_OUT_TEXT or HESAFF_OUT_BIN, optionally | HESAFF_OUT_STRICT)
 * of an earlier run, -1 otherwise: what hesaff_set_resume goes by (on = 1: as given; on = 2: with HESAFF_OUT_STRICT) */
int hesaff_output_is_complete(const char *path, int format);
/* formats into a malloc'ed buffer (*out, *len); caller frees with hesaff_free */
int hesaff_format_sift(const hesaff_keypoint *keys, int n, float mrSize, char **out, size_t *len);
/* the same bytes, rows formatted by `threads` host threads (0 = one per core, at most 64);
 * hesaff_write_sift uses this form.  SURVEY.md 8(f) rank 1: at GPU rates the text export
 * (42 MB per UHD image, hesaff.cpp:107-130) is the bottleneck of the file path. */
int hesaff_format_sift_mt(const hesaff_keypoint *keys, int n, float mrSize, int threads, char **out, size_t *len);
/* replaces: the per-image exportKeypoints + ofstream of main() (hesaff.cpp:170-176) for a whole
 * batch: results[i] -> paths[i], images spread over `threads` host threads (0 = auto) */
int hesaff_write_sift_batch(int n_images, const char *const *paths, const hesaff_result *results, float mrSize, int threads);
/* what "threads = 0" means above: the CPUs this process may run on (affinity mask), capped by the
 * container's CPU-time limit (cgroup v2 cpu.max) when there is one, at most 64, at least 1.
 * (No counterpart in the reference, which is single-threaded: hesaff.cpp:133-180.) */
int hesaff_host_threads(void);
/* The ONE rule for "host threads per device" (the CLI's --batch, hesaff_process_files' "0 = auto", bench.py and tools/ all use it).
 * devices_sharing_host: how many devices are fed from the CPUs hesaff_host_threads() counts - 8 for one rank of an 8-GPU node,
 * 1 for a process that has the host to itself.
 *   cpus           = max(1, hesaff_host_threads() / devices_sharing_host)      the CPU share of one device, everything included:
 *                                                                                the caller's thread (kernel launches; it sleeps on events),
 *                                                                                the staging threads and the decode / write pool
 *   stage_threads  = clamp(cpus / 4, 1, 4)                                       threads that copy a chunk into pinned memory
 *   pool           = max(2, cpus - stage_threads)                                decode_threads + write_threads (ONE pool, see
 *   decode_threads = max(1, pool / 4) ; write_threads = pool - decode_threads    hesaff_process_files)
 * With 2 CPUs the pool is 2 and time-slices with the caller's and the staging thread, which are idle most of the time.
 * (No counterpart in the reference, which is single-threaded: hesaff.cpp:133-180.) */
typedef struct hesaff_host_plan {
   int cpus, decode_threads, write_threads, stage_threads;
} hesaff_host_plan;
int hesaff_host_plan_for(int devices_sharing_host, hesaff_host_plan *out);
/* the plan's stage_threads for a pool of decode_threads + write_threads (what hesaff_process_files derives from the two counts it is given) */
int hesaff_stage_threads_for_pool(int pool);
/* Page-locked read buffers of hesaff_process_files (the readers fill pinned memory of the context, the copy engine takes the image
 * from where it was read): at most max_bytes are out or parked at any time (default 4 GiB; a request beyond it takes the staging
 * copy), and at most keep_bytes stay pinned when hesaff_process_files returns (default 1 GiB: the next list of a long-lived
 * context starts with warm buffers; 0 releases everything).  No counterpart in the reference. */
int hesaff_set_pinned_read_budget(hesaff_ctx *ctx, size_t max_bytes, size_t keep_bytes);
/* The decode / write pool of hesaff_process_files lowers its threads' priority (nice 10) only when the host plan is CPU-starved
 * (CPUs of this process <= pool threads + 1), so that the caller's thread - which launches the next kernels when an event fires -
 * never queues behind threads that write() flat out.  mode: 0 = never, 1 = always, -1 = that rule (default). */
int hesaff_set_pool_priority(hesaff_ctx *ctx, int mode);
/* test hook: number of inputs on which the fast "%g" formatter and snprintf disagree (must be 0) */
int hesaff_test_fmt_g(const float *v, int n);
void hesaff_free(void *p);

/* replaces: cv::imread(argv[1]) hesaff.cpp:137 for PBM / PGM / PPM files, plain and binary (P1..P6), maxval 1..65535, the way
 * OpenCV's PxM decoder delivers them at imread's default flag: binary 8-bit samples as they are, plain samples scaled by
 * 255 / maxval, 16-bit samples reduced to the high byte, bitmaps as 255 / 0.
 * *data is malloc'ed (free with hesaff_free), tightly packed, channels 1 or 3. */
int hesaff_read_pnm(const char *path, uint8_t **data, int *width, int *height, int *channels);
/* the same for PNG files (decoded with zlib): what cv::imread returns with its default flag - 8 bits per
 * channel, alpha dropped, 16-bit samples reduced to the high byte, palette / 1-2-4-bit grey expanded;
 * channels = 1 for grey files, 3 (R,G,B order) otherwise.  Adam7-interlaced files are read too. */
int hesaff_read_png(const char *path, uint8_t **data, int *width, int *height, int *channels);
/* Huffman-coded 8-bit JPEG, sequential or progressive, grey or YCbCr: the integer algorithms of libjpeg at
 * cv::imread's settings (JDCT_ISLOW inverse DCT, "fancy" chroma up-sampling, JFIF colour conversion), pixel for pixel
 * the bytes libjpeg / libjpeg-turbo return for a complete file; channels = 1 for grey files, 3 (R,G,B order)
 * otherwise.  Arithmetic-coded, lossless, 12-bit and CMYK files: HESAFF_ERR_IO. */
int hesaff_read_jpeg(const char *path, uint8_t **data, int *width, int *height, int *channels);
/* Windows bitmaps as OpenCV's BMP decoder delivers them at imread's default flag: 1 / 4 / 8 bits through the palette (uncompressed, RLE4, RLE8),
 * 16 bits (5-5-5, or 5-6-5 by bit fields; low bits left zero), 24 bits, 32 bits (fourth byte dropped), bottom-up or top-down, OS/2 core headers;
 * channels = 3 (R,G,B order), or 1 when the whole palette is grey. */
int hesaff_read_bmp(const char *path, uint8_t **data, int *width, int *height, int *channels);
/* Baseline TIFF the way cv::imread delivers it (libtiff's RGBA interface, alpha byte dropped): bilevel / 2 / 4 / 8-bit grey (MinIsBlack,
 * MinIsWhite), 8-bit palette, 8-bit RGB and RGB + alpha (unassociated alpha multiplied in: (v a + 127) / 255); strips or tiles, chunky or
 * planar, both byte orders; uncompressed, PackBits, LZW, Deflate, horizontal predictor.  16-bit / float samples, YCbCr / CMYK / Lab, JPEG-
 * or fax-compressed data and BigTIFF are refused (HESAFF_ERR_IO).  channels = 1 for grey files, 3 (R,G,B order) otherwise. */
int hesaff_read_tiff(const char *path, uint8_t **data, int *width, int *height, int *channels);
/* PBM/PGM/PPM, PNG, JPEG, BMP or TIFF by magic number */
int hesaff_read_image(const char *path, uint8_t **data, int *width, int *height, int *channels);
/* the same with the pixel buffer from the caller's allocator (only the PNM reader asks it; see hesaff_blob_alloc below):
 * hesaff_process_files reads straight into pinned host memory this way (no malloc'ed buffer, no staging copy) and recycles the
 * buffers.  A buffer the allocator handed out belongs to the caller whatever happens: when the call fails after the allocator was
 * asked, the buffer comes back through *data (NULL otherwise) and is never passed to free(). */
typedef void *(*hesaff_blob_alloc)(size_t bytes, int *zeroed, void *user);
int hesaff_read_pnm_alloc(const char *path, uint8_t **data, int *width, int *height, int *channels, hesaff_blob_alloc alloc, void *user);
int hesaff_read_image_alloc(const char *path, uint8_t **data, int *width, int *height, int *channels, hesaff_blob_alloc alloc, void *user);

/* The host half of cv::imread (hesaff.cpp:137) for a JPEG file when the pixels are made on the device (hesaff_process_files does this
 * for every JPEG of its list): markers and entropy decoding only - Huffman, sequential or progressive - which is the part of a JPEG
 * decoder that has to run in order.  The inverse DCT, the chroma up-sampling and the colour conversion of hesaff_read_jpeg (the same
 * integer algorithms, so the same bytes) run in kernels_jpeg.h over all blocks / pixels of a chunk of images at once.
 * layout: what images must share to travel in one chunk.  blob (malloc'ed, hesaff_free): HESAFF_JPEG_BLOB_HEADER bytes - the
 * components' quantisation tables, uint16[3][64] in natural order, at byte 0; int32 "YCbCr, convert to RGB" flag at byte 384 - then
 * the coefficients, int16, natural order, 64 per block, blocks row by row (bw x bh per component), component after component. */
#define HESAFF_JPEG_BLOB_HEADER 1024
typedef struct hesaff_jpeg_layout {
   int32_t width, height, channels;   /* channels: 1 (grey file) or 3 (R,G,B after conversion) */
   int32_t h[3], v[3];                /* sampling factors of the components */
   int32_t hx[3], vx[3];              /* up-sampling ratios to full resolution (hmax / h, vmax / v) */
   int32_t bw[3], bh[3];              /* blocks per row / column of a component's coefficient array (whole MCUs) */
   int32_t cw[3], chgt[3];            /* component size in samples: ceil(image size * factor / largest factor) */
} hesaff_jpeg_layout;
int hesaff_read_jpeg_coefficients(const char *path, hesaff_jpeg_layout *layout, uint8_t **blob, size_t *blob_bytes);
/* the same with the blob's memory from the caller: alloc(bytes, &zeroed, user) returns memory that free() accepts (or NULL) and says
 * whether it is already zero.  hesaff_process_files hands back the blobs of images that have gone to the device - a decoder thread
 * then writes into warm memory instead of 25 MB of fresh zero pages per photograph.  The blob's content does not depend on it. */
int hesaff_read_jpeg_coefficients_alloc(const char *path, hesaff_jpeg_layout *layout, uint8_t **blob, size_t *blob_bytes,
                                        hesaff_blob_alloc alloc, void *user);

/* ---- stage entry points (host pointers in/out; used by the parity tests and by callers
 *      that want one operator of the reference at a time) ---- */

/* gaussianBlur helpers.cpp:283-289 (cv::GaussianBlur, BORDER_REPLICATE, ksize from sigma) */
int hesaff_stage_gaussian_blur(hesaff_ctx *ctx, const float *in, int rows, int cols, float sigma, float *out);
/* HessianDetector::hessianResponse pyramid.cpp:63-114 (frame written as 0) */
int hesaff_stage_hessian_response(hesaff_ctx *ctx, const float *in, int rows, int cols, float norm, float *out);
/* halfImage helpers.cpp:331-339 */
int hesaff_stage_half_image(hesaff_ctx *ctx, const float *in, int rows, int cols, float *out);
/* Scale-space of one image: initial blur pyramid.cpp:276-280 + every octave's L0..L4 and
 * R0..R4 (pyramid.cpp:224-259).  planes receives, octave after octave, 5 blur planes then 5
 * response planes, each rows_o x cols_o tightly packed; returns the octave count in
 * *n_octaves.  Call with planes == NULL to get n_octaves and the float count in *n_floats. */
int hesaff_stage_pyramid(hesaff_ctx *ctx, const uint8_t *gray, int rows, int cols, float *planes,
                         int *n_octaves, size_t *n_floats);
/* detectPyramidKeypoints pyramid.cpp:261-292 up to the onHessianKeypointDetected callback
 * pyramid.h:46: fills f[n][5] = x,y,s,pixelDistance,response and i[n][5] =
 * type,octave,level,r0,c0 in reference order; returns n in *count (cap = array capacity). */
int hesaff_stage_hessian_keypoints(hesaff_ctx *ctx, const uint8_t *gray, int rows, int cols, int cap,
                                   float *f, int32_t *i, int *count);
/* AffineShape::findAffineShape affine.cpp:35-100 for n keypoints on one blur plane.
 * kp[n][4] = x,y,s,pixelDistance; out: converged[n], U[n][4] (a11,a12,a21,a22 as passed to
 * onAffineShapeFound affine.h:50-57), iters[n]. */
int hesaff_stage_find_affine_shape(hesaff_ctx *ctx, const float *blur, int rows, int cols, int n,
                                   const float *kp, int32_t *converged, float *U, int32_t *iters);
/* rectifyAffineTransformationUpIsUp helpers.cpp:90-97 for n matrices (in place, A[n][4]) */
int hesaff_stage_rectify(hesaff_ctx *ctx, int n, float *A);
/* AffineShape::normalizeAffine affine.cpp:102-144 for n keypoints on one image.
 * kp[n][3] = x,y,s; A[n][4] rectified; out: rejected[n] (1 = reference returned true),
 * patches[n][41*41]. */
int hesaff_stage_normalize_affine(hesaff_ctx *ctx, const float *img, int rows, int cols, int n,
                                  const float *kp, const float *A, int32_t *rejected, float *patches);
/* SIFTDescriptor::computeSiftDescriptor siftdesc.cpp:115-140 for n 41x41 patches;
 * desc[n][128] = the values of `vec` cast as at hesaff.cpp:91. */
int hesaff_stage_sift(hesaff_ctx *ctx, int n, const float *patches, uint8_t *desc);
/* exportKeypoints hesaff.cpp:107-130 on the device for n records in host memory (what hesaff_process_files runs per chunk):
 * format = HESAFF_OUT_TEXT: the bytes of the .hesaff.sift file, == hesaff_format_sift; HESAFF_OUT_BIN: the bytes of the sidecar,
 * == hesaff_write_bin's file.  *out is malloc'ed (hesaff_free). */
int hesaff_stage_export(hesaff_ctx *ctx, const hesaff_keypoint *keys, int n, float mrSize, int format, char **out, size_t *len);
/* the device's "%g" print of n floats: 16 bytes per value in text (unused bytes 0) and its length in lens; for testing that
 * it equals the host's over the whole binary32 range */
int hesaff_stage_fmt_g(hesaff_ctx *ctx, int n, const float *v, char *text, int32_t *lens);
/* the device half of the JPEG reader for n images of one layout (blobs back to back, blob_bytes each, from
 * hesaff_read_jpeg_coefficients): pixels[n][height][width][channels] == what hesaff_read_jpeg returns for the files */
int hesaff_stage_jpeg_pixels(hesaff_ctx *ctx, const hesaff_jpeg_layout *layout, int n, const uint8_t *blobs, size_t blob_bytes, uint8_t *pixels);
/* device evaluation of the pinned libm restatements (hmath.h) for testing */
int hesaff_stage_math(hesaff_ctx *ctx, int n, const float *a, const float *b, float *atan2_out, float *pow2_out);
/* the per-pixel forms of the descriptor gradient (helpers.cpp:269-280, siftdesc.cpp:123-137): orientation atan2f(gy, gx) and
 * magnitude sqrt(gx^2 + gy^2), each in the general form and in the form without range handling that the kernel uses on
 * photometrically normalised patches (operands zero or normal); for testing that the two agree bit for bit */
int hesaff_stage_math_sift(hesaff_ctx *ctx, int n, const float *gy, const float *gx, float *ori_general, float *ori_nd,
                           float *grad_general, float *grad_nd);

/* Host-side tables the kernels use (for known-answer tests): computeGaussMask
 * helpers.cpp:104, computeCircularGaussMask helpers.cpp:131, precomputeBinsAndWeights
 * siftdesc.cpp:18, OpenCV getGaussianKernel. */
int hesaff_table_gauss_mask(int size, float *mask);
int hesaff_table_circ_gauss_mask(int size, float *mask);
int hesaff_table_sift_bins(int32_t *bin0, int32_t *bin1, float *w0, float *w1);
int hesaff_table_gauss_kernel(float sigma, int cap, float *taps, int *ksize);

/* ---- multi-GPU: images of a batch shard across the devices of a node, no data-path collective ---- */
/* number of visible HIP devices (0 when there is none / no runtime) */
int hesaff_device_count(void);
/* contiguous block [*lo, *hi) of n items owned by `rank` of `world`: item i belongs to rank floor(i * world / n);
 * the same rule as hesaff_amd/shard.py:shard_range (bench.py, one process per GPU) */
int hesaff_shard_range(int n, int rank, int world, int *lo, int *hi);

const char *hesaff_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HESAFF_AMD_H */
