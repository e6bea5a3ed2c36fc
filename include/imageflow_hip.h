/*
 * imageflow_hip.h -- C ABI of libimageflow_hip.so: the MI355X (gfx950) implementation of imageflow's
 * pixel hot path.  Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * Every entry point names the reference interface it replaces (paths relative to the imageflow tree).
 * The Rust side binds these with an `extern "C"` block exactly like imageflow_core/src/ffi/c_interop.rs:115-213
 * binds c_components today (binding source: bindings/hip_interop.rs, generated from this file; call sites: INTEGRATION.md).
 *
 * Conventions (mirroring wrap_jpeg_* in c_components/lib/codec_jpeg_wrapper.c:187-233 and FlowError):
 *   - functions return an ifhip_status (0 = ok); the message of the last failure on the calling thread is
 *     available from ifhip_last_error_message();
 *   - a bitmap is (pointer, w, h, stride-in-bytes), BGRA8, rows padded as graphics/bitmaps.rs:712-740;
 *   - colours are Color32 0xAARRGGBB whose little-endian bytes are B,G,R,A (imageflow_helpers/src/colors.rs:77-117);
 *   - `*_device` variants take pointers into HBM and a hipStream_t (as void*); they enqueue and return.
 *     The non-device variants take host memory, are synchronous, and are the drop-in for the Rust callers.
 *   - there is NO CPU fallback: without a usable gfx950 device every compute entry point fails with
 *     IFHIP_GPU_UNAVAILABLE.
 */
#ifndef IMAGEFLOW_HIP_H
#define IMAGEFLOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IFHIP_API __attribute__((visibility("default")))

/* Subset of imageflow_core::ErrorKind (errors.rs:158-248) that this path can raise. */
typedef enum ifhip_status {
    IFHIP_OK = 0,
    IFHIP_INVALID_ARGUMENT = 1,        /* ErrorKind::InvalidArgument   (scaling.rs:24-29,40)   */
    IFHIP_METHOD_NOT_IMPLEMENTED = 2,  /* ErrorKind::MethodNotImplemented (scaling.rs:43-48)   */
    IFHIP_INVALID_STATE = 3,           /* ErrorKind::InvalidState      (scaling.rs:145,192)    */
    IFHIP_ALLOCATION_FAILED = 4,       /* ErrorKind::AllocationFailed                           */
    IFHIP_GPU_UNAVAILABLE = 5,         /* no gfx950 device / HIP runtime error                  */
    IFHIP_GPU_ERROR = 6
} ifhip_status;

/* graphics/weights.rs:43-78 -- same discriminants. */
typedef enum ifhip_filter {
    IFHIP_FILTER_ROBIDOUX_FAST = 1, IFHIP_FILTER_ROBIDOUX = 2, IFHIP_FILTER_ROBIDOUX_SHARP = 3,
    IFHIP_FILTER_GINSENG = 4, IFHIP_FILTER_GINSENG_SHARP = 5, IFHIP_FILTER_LANCZOS = 6,
    IFHIP_FILTER_LANCZOS_SHARP = 7, IFHIP_FILTER_LANCZOS2 = 8, IFHIP_FILTER_LANCZOS2_SHARP = 9,
    IFHIP_FILTER_CUBIC_FAST = 10, IFHIP_FILTER_CUBIC = 11, IFHIP_FILTER_CUBIC_SHARP = 12,
    IFHIP_FILTER_CATMULL_ROM = 13, IFHIP_FILTER_MITCHELL = 14, IFHIP_FILTER_CUBIC_B_SPLINE = 15,
    IFHIP_FILTER_HERMITE = 16, IFHIP_FILTER_JINC = 17, IFHIP_FILTER_RAW_LANCZOS3 = 18,
    IFHIP_FILTER_RAW_LANCZOS3_SHARP = 19, IFHIP_FILTER_RAW_LANCZOS2 = 20, IFHIP_FILTER_RAW_LANCZOS2_SHARP = 21,
    IFHIP_FILTER_TRIANGLE = 22, IFHIP_FILTER_LINEAR = 23, IFHIP_FILTER_BOX = 24,
    IFHIP_FILTER_CATMULL_ROM_FAST = 25, IFHIP_FILTER_CATMULL_ROM_FAST_SHARP = 26, IFHIP_FILTER_FASTEST = 27,
    IFHIP_FILTER_MITCHELL_FAST = 28, IFHIP_FILTER_N_CUBIC = 29, IFHIP_FILTER_N_CUBIC_SHARP = 30,
    IFHIP_FILTER_LEGACY_IDCT = 31
} ifhip_filter;

/* graphics/color.rs:4-9 WorkingFloatspace (Gamma is not reachable from scale_render.rs:293-296). */
typedef enum ifhip_working_space { IFHIP_SPACE_SRGB = 0, IFHIP_SPACE_LINEAR = 1 } ifhip_working_space;

/* ffi/mod.rs:41-47 BitmapCompositingMode. */
typedef enum ifhip_compositing {
    IFHIP_REPLACE_SELF = 0, IFHIP_BLEND_WITH_SELF = 1, IFHIP_BLEND_WITH_MATTE = 2
} ifhip_compositing;

/* weights.rs:14-40 LobeRatio. */
typedef enum ifhip_lobe_mode { IFHIP_LOBE_NATURAL = 0, IFHIP_LOBE_EXACT = 1, IFHIP_LOBE_SHARPEN_PERCENT = 2 } ifhip_lobe_mode;

/* ---- library / device -------------------------------------------------------------------------------- */
IFHIP_API const char* ifhip_last_error_message(void);
IFHIP_API const char* ifhip_version(void);
/* Development switches of tests and tools/ (kernel choice for A/B runs, rarely taken paths forced): the library reads NO
 * environment variable -- what a call launches depends on its arguments only -- and a switch exists only after this call
 * (value NULL: unset).  Not part of the drop-in surface. */
IFHIP_API int ifhip_debug_set(const char* key, const char* value);
IFHIP_API int ifhip_device_count(void);            /* number of usable gfx950 devices (0 if none)          */
/* Compute units the resample launches of this PROCESS plan for (default and 0: all 256).  The fused kernel occupies a CU per
 * workgroup (its tables fill the LDS) and cuts frames into bands so that a launch fills the chip in whole rounds; a host that
 * runs something else beside it -- the batch harness overlaps the RCCL gather of batch k, a few workgroups, with the kernel
 * of batch k + 1 -- says how many CUs that leaves, and the band count is chosen for THAT number (finer bands, a short last
 * round) instead of a grid of exactly 256 workgroups whose last few wait a whole round for a CU.  Pixels do not depend on it. */
IFHIP_API int ifhip_set_cu_budget(uint32_t compute_units);
IFHIP_API int ifhip_set_device(int ordinal);       /* one process per GPU: call once with LOCAL_RANK       */
/* The HIP stream on which THIS THREAD's create calls (plans, stages, entropy handles) upload and clear what they need;
 * default: the null stream.  A host that runs one job per thread (one imageflow Context per thread, lib.rs:20-27) sets its
 * own stream here and passes the same stream to the *_device calls, so that jobs of different threads overlap. */
IFHIP_API void ifhip_set_thread_stream(void* hip_stream);
/* The library's block cache (csrc/devmem.cpp): plans, stages, entropy handles and the frames of ABI jobs are created and
 * destroyed once per JOB, so their device and pinned blocks are recycled through size-class free lists instead of hipMalloc /
 * hipFree (whose device-wide wait would serialise every thread's jobs).  What sits in the lists is invisible to every other
 * allocator on the device (torch's caching allocator, other processes):
 *   ifhip_cache_set_limits  bytes the lists may hold, per device / pinned (defaults 8 GiB / 1 GiB; 0 = recycle nothing);
 *   ifhip_cache_trim        give listed blocks back to the driver until at most the given bytes stay (0, 0: everything) --
 *                           what a torch user calls next to torch.cuda.empty_cache() or after an out-of-memory error;
 *   ifhip_cache_stats       hits / driver calls / bytes cached and handed out, for capacity planning and the jobs bench.
 * Blocks handed out are never touched.  Scratch of the device calls is released BEHIND the launch stream (an event marks the
 * point; no host wait): such blocks count as cached from that moment, are handed out again at once for work on the same
 * stream, to anybody once the event has completed, and a trim waits for them.  Thread safe. */
typedef struct ifhip_cache_stats_t {
    uint64_t device_hits, device_driver_allocs, device_driver_frees, device_oom_flushes, device_wide_syncs;
    uint64_t device_bytes_cached, device_bytes_live, device_blocks_live, device_limit_bytes;
    uint64_t host_hits, host_driver_allocs, host_driver_frees;
    uint64_t host_bytes_cached, host_bytes_live, host_blocks_live, host_limit_bytes;
} ifhip_cache_stats_t;
IFHIP_API int ifhip_cache_set_limits(size_t device_bytes, size_t host_bytes);
IFHIP_API int ifhip_cache_trim(size_t keep_device_bytes, size_t keep_host_bytes, size_t* released_device_bytes, size_t* released_host_bytes);
IFHIP_API int ifhip_cache_stats(ifhip_cache_stats_t* out);

/* ---- host-side tables (no GPU needed) ---------------------------------------------------------------- */
/* graphics/bitmaps.rs:712-740 Bitmap::get_stride::<u8>(w, h, 4, 64); 0 when the row does not fit 32 bits. */
IFHIP_API uint32_t ifhip_stride_for_width(uint32_t w);
/* graphics/weights.rs:681-788 populate_weights + PixelWeightIndexes (:555-571).  left[u]/count[u] give the first
 * source pixel and tap count of output pixel u; weights are concatenated in output order.  Pass
 * weights_capacity = 0 to query *n_weights.  kernel_width_scale = 1.0 unless set_kernel_width_scale is wanted. */
IFHIP_API int ifhip_populate_weights(int filter, int lobe_mode, float lobe_value, double kernel_width_scale,
                                     uint32_t output_line_size, uint32_t input_line_size,
                                     uint32_t* left_pixel, uint32_t* tap_count,
                                     float* weights, uint32_t weights_capacity, uint32_t* n_weights);
/* graphics/color.rs:22-45 ColorContext::byte_to_float (space = working space), 256 floats. */
IFHIP_API int ifhip_table_srgb_to_floatspace(int working_space, float* out256);
/* graphics/lut.rs:14-271 LINEAR_TO_SRGB_LUT, 16384 bytes. */
IFHIP_API int ifhip_table_linear_to_srgb(uint8_t* out16384);
/* The same table in the form the fused kernel keeps in LDS: thr[k] = first index whose value is >= k+1
 * (65535 if none), so table[i] == number of k with thr[k] <= i. */
IFHIP_API int ifhip_table_linear_to_srgb_thresholds(uint16_t* out256);

/* ---- Inner A: resample + render ---------------------------------------------------------------------- */
/*
 * Replaces imageflow_core::graphics::scaling::scale_and_render (graphics/scaling.rs:19-90) with
 * ScaleAndRenderParams (:8-17).  Host buffers, synchronous: upload -> kernels -> download of the touched
 * canvas rect.  in_alpha_meaningful = input.info().alpha_meaningful(); `compositing`/`matte_bgra` =
 * canvas.info().compose().  Errors as the reference: rect outside the canvas -> IFHIP_INVALID_ARGUMENT.
 */
IFHIP_API int ifhip_scale_and_render(const uint8_t* in, uint32_t in_w, uint32_t in_h, uint32_t in_stride,
                                     int in_alpha_meaningful,
                                     uint8_t* canvas, uint32_t canvas_w, uint32_t canvas_h, uint32_t canvas_stride,
                                     int canvas_alpha_meaningful,
                                     uint32_t x, uint32_t y, uint32_t w, uint32_t h,
                                     int filter, float sharpen_percent_goal, int working_space,
                                     int compositing, uint32_t matte_bgra);

/*
 * Device-resident batch form: what a job that keeps frames in HBM calls.  A plan owns the per-shape tables
 * (PixelRowWeights for both axes, the vertical schedule) for (in_w, in_h) -> (w, h); its tables are immutable and it
 * may be shared by any number of launches on the device it was created on (internal lazily built caches are locked).
 * The generic two-pass kernels (up-scaling, > 8 live rows, unaligned rows) stage through stream-ordered scratch
 * (a block of the library's cache, released behind the launch stream: ifhip_cache_stats counts it), so launches of one plan
 * may run on any number of streams.
 */
typedef struct ifhip_resample_plan ifhip_resample_plan;
IFHIP_API int ifhip_resample_plan_create(ifhip_resample_plan** plan, uint32_t in_w, uint32_t in_h,
                                         uint32_t w, uint32_t h, int filter, float sharpen_percent_goal);
IFHIP_API void ifhip_resample_plan_destroy(ifhip_resample_plan* plan);
/* introspection for tests/bench: 0 = fused single-pass kernel, 1 = generic two-pass kernels */
IFHIP_API int ifhip_resample_plan_kernel_kind(const ifhip_resample_plan* plan, int in_alpha_meaningful);
/* Diagnostic: the fused kernel's fast horizontal pass for this plan -- groups of four source columns per output (0: some
 * output needs more than four groups, the general pass runs) and groups of two (0: not available, or more than two thirds
 * of the taps; used for BGRA sources without meaningful alpha). */
IFHIP_API int ifhip_resample_plan_horizontal_groups(const ifhip_resample_plan* plan, uint32_t* four_column_groups,
                                                    uint32_t* two_column_groups);

/*
 * n_images independent frames, image i at d_in + i*in_image_bytes / d_canvas + i*canvas_image_bytes.
 * d_f32_dump (nullable): [n_images][h][w][4] premultiplied working-space floats (the f32 working buffer,
 * what StreamingResize::next_output_row_f32 yields at scaling.rs:195).
 * force_kernel: -1 auto, 0 fused, 1 generic, 2 banded (the generic pair fused through LDS: up-scales and small frames);
 * tests cross-check them.
 */
IFHIP_API int ifhip_scale_and_render_batch_device(const ifhip_resample_plan* plan,
                                                  const uint8_t* d_in, size_t in_image_bytes, uint32_t in_stride,
                                                  int in_alpha_meaningful, uint32_t n_images,
                                                  uint8_t* d_canvas, size_t canvas_image_bytes,
                                                  uint32_t canvas_w, uint32_t canvas_h, uint32_t canvas_stride,
                                                  uint32_t x, uint32_t y,
                                                  int working_space, int compositing, uint32_t matte_bgra,
                                                  float* d_f32_dump, int force_kernel, void* hip_stream);

/* ---- Inner B: flatten --------------------------------------------------------------------------------- */
/* Replaces graphics::blend::apply_matte (graphics/blend.rs:6-59) / Bitmap::apply_matte (bitmaps.rs:528-541).
 * In place; no-op when !alpha_meaningful (blend.rs:11-13). */
IFHIP_API int ifhip_apply_matte(uint8_t* bgra, uint32_t w, uint32_t h, uint32_t stride, int alpha_meaningful,
                                uint32_t matte_bgra);
IFHIP_API int ifhip_apply_matte_batch_device(uint8_t* d_bgra, size_t image_bytes, uint32_t n_images,
                                             uint32_t w, uint32_t h, uint32_t stride, int alpha_meaningful,
                                             uint32_t matte_bgra, void* hip_stream);

/* ---- Inner C: JPEG pixel stage ------------------------------------------------------------------------- */
/*
 * Replaces what libjpeg does between entropy decoding and the scanline hand-over inside MzDec::read_frame
 * (codecs/mozjpeg_decoder.rs:295-420, wrap_jpeg_read_scan_lines codec_jpeg_wrapper.c:203-212): de-quantisation,
 * the islow 8x8 IDCT, "fancy" chroma up-sampling (h2v1 / h2v2) and YCbCr -> BGRA (out_color_space = JCS_EXT_BGRA,
 * :320; alpha bytes 255).  A GPU stage cannot be libjpeg's per-block IDCT callback, so the boundary is
 * "coefficient planes in, BGRA8 64-byte-stride bitmap out": the host keeps mozjpeg for the (serial) Huffman pass and
 * hands over what jpeg_read_coefficients() returns.
 *   coef[c]   int16 [blocks_h_c][blocks_w_c][64], natural (de-zigzagged) order, quantised;
 *             blocks_w_c = ceil(width / (8*hmax)) * h_samp[c], blocks_h_c likewise (MCU padded, as libjpeg's arrays)
 *   qt        uint16 [n_components][64], natural order (JQUANT_TBL.quantval)
 * scale_num / luma_spatial / luma_srgb are what MzDec::apply_downscaling (:588-618) and DecoderDownscaleHints
 * (ffi/c_interop.rs:6-15) set: libjpeg's scale_num/8 and whether the luma component goes through imageflow's
 * flow_scale_spatial[_srgb]_NxN block scalers (codec_jpeg_wrapper.c:274-343) instead of libjpeg's reduced IDCT.
 * Supported: scale_num 1..6 and 8 (7/8 is never requested, mozjpeg_decoder.rs:603-606) for grayscale, 4:4:4, 4:2:2 (h2v1),
 * 4:4:0 (h1v2) and 4:2:0 -- libjpeg's scaled IDCTs (jidctred.c 1x1/2x2/4x4, jidctint.c 3x3/5x5/6x6/10x10/12x12), its
 * per-component IDCT size rule (jdmaster.c: 4:2:0 chroma decodes at 2*scale_num, no up-sampling) and its up-sampler
 * choice (jdsample.c: triangle forms need scale_num > 1 and downsampled_width > 2, else replication).  The output
 * bitmap is ceil(width*scale_num/8) x ceil(height*scale_num/8).
 */
IFHIP_API int ifhip_jpeg_idct_color(const int16_t* coef0, const int16_t* coef1, const int16_t* coef2,
                                    const uint16_t* qt, int n_components,
                                    const uint8_t* h_samp, const uint8_t* v_samp,
                                    uint32_t width, uint32_t height,
                                    int scale_num, int luma_spatial, int luma_srgb,
                                    uint8_t* bgra, uint32_t stride);

/* Device-resident batch of equally shaped frames.  The stage object owns the component planes between the two
 * kernels.  d_coef[c]: image i at d_coef[c] + i * blocks_w_c*blocks_h_c*64; d_qt: [n_images][n_components][64]. */
typedef struct ifhip_jpeg_stage ifhip_jpeg_stage;
IFHIP_API int ifhip_jpeg_stage_create(ifhip_jpeg_stage** stage, uint32_t width, uint32_t height, int n_components,
                                      const uint8_t* h_samp, const uint8_t* v_samp,
                                      int scale_num, int luma_spatial, int luma_srgb, uint32_t max_images);
IFHIP_API int ifhip_jpeg_stage_output_size(const ifhip_jpeg_stage* stage, uint32_t* out_w, uint32_t* out_h);
IFHIP_API void ifhip_jpeg_stage_destroy(ifhip_jpeg_stage* stage);
IFHIP_API int ifhip_jpeg_stage_block_dims(const ifhip_jpeg_stage* stage, uint32_t* blocks_w3, uint32_t* blocks_h3);
IFHIP_API int ifhip_jpeg_idct_color_batch_device(ifhip_jpeg_stage* stage, const int16_t* d_coef0,
                                                 const int16_t* d_coef1, const int16_t* d_coef2,
                                                 const uint16_t* d_qt, uint32_t n_images,
                                                 uint8_t* d_bgra, size_t image_bytes, uint32_t stride, void* hip_stream);

/* Decode and resample as one device call: replaces MzDec::read_frame (codecs/mozjpeg_decoder.rs:346-362) producing the
 * bitmap that DrawImageDef::render (flow/nodes/scale_render.rs:304-313) hands to scale_and_render -- the pair every
 * `decode -> resample_2d / constrain` job runs.  `plan` must resample the stage's output size (ifhip_jpeg_stage_output_size).
 * When the component planes leave the IDCT at output resolution (4:2:0 decoded at 1/8 .. 4/8, where chroma takes the
 * twice-larger IDCT; 4:4:4 at any scale) the resampler reads the three planes and converts YCbCr -> RGB in its row fetch:
 * no decoded BGRA frame is written to or read from HBM, *fused = 1.  Every other case (fancy up-sampling, grayscale,
 * shapes the fused resampler does not take) runs ifhip_jpeg_idct_color_batch_device into a stream-ordered scratch
 * and ifhip_scale_and_render_batch_device, *fused = 0.  The canvas bytes are the same either way.  `fused` may be NULL. */
IFHIP_API int ifhip_jpeg_decode_resample_batch_device(ifhip_jpeg_stage* stage, const int16_t* d_coef0,
                                                      const int16_t* d_coef1, const int16_t* d_coef2,
                                                      const uint16_t* d_qt, uint32_t n_images,
                                                      const ifhip_resample_plan* plan,
                                                      uint8_t* d_canvas, size_t canvas_image_bytes, uint32_t canvas_w,
                                                      uint32_t canvas_h, uint32_t canvas_stride, uint32_t x, uint32_t y,
                                                      int working_space, int compositing, uint32_t matte_bgra,
                                                      int* fused, void* hip_stream);

/* Entropy stage: baseline sequential-Huffman decode of whole files on the GPU, in place of
 * the serial jpeg_read_coefficients / decode_mcu loop MzDec::read_frame drives on the host
 * (codecs/mozjpeg_decoder.rs:346-362).  ifhip_jpeg_parse_headers is host-only (no GPU): the SOF / DQT / DRI facts the
 * decoder's get_image_info needs (mozjpeg_decoder.rs:245-293).  A batch = n files of one geometry (size + sampling);
 * create() parses, un-stuffs and uploads the scans, decode_device() fills the coefficient planes the pixel stage reads
 * ([n][blocks_h][blocks_w][64] int16, natural order) and synchronises the stream.  Progressive, arithmetic-coded,
 * 12-bit, multi-scan and CMYK files return IFHIP_METHOD_NOT_IMPLEMENTED: keep those on libjpeg. */
typedef struct ifhip_jpeg_entropy ifhip_jpeg_entropy;
/* Diagnostic, host only (no GPU, nothing decoded into coefficients): the decode tables and the scan layout of one file,
 * checked against each other -- what tests/test_jpeg_headers.py asserts on every committed file without a device. */
typedef struct ifhip_jpeg_scan_report {
    uint32_t segments, sub_sequences, scan_complete;       /* restart segments; 1 024-bit sub-sequences; every MCU is covered */
    uint32_t pool_entries, pool_entries_used, prefixes_left_to_search, pair_entries;   /* second-level pool; first-level entries */
    uint32_t segments_with_wrong_block_count, segments_with_invalid_codes;           /* a serial walk with the tables */
    uint32_t pair_walk_mismatches, count_walk_mismatches;  /* sub-sequence boundaries where a pair-table walk differs from the plain one */
    uint64_t symbols, table_reads_with_pairs, blocks;
    int32_t dc_sum[3], dc_last_segment[3];                 /* sums of the DC differences per component (all segments / the last one) */
} ifhip_jpeg_scan_report;
IFHIP_API int ifhip_jpeg_debug_scan_report(const uint8_t* jpeg, size_t len, ifhip_jpeg_scan_report* out);
IFHIP_API int ifhip_jpeg_parse_headers(const uint8_t* jpeg, size_t len, uint32_t* width, uint32_t* height,
                                       int* n_components, uint8_t* h_samp3, uint8_t* v_samp3, uint32_t* blocks_w3,
                                       uint32_t* blocks_h3, uint16_t* qt3x64, uint32_t* restart_interval);
/* EXIF orientation as MozJpegDecoder::get_exif_rotation_flag reads it (codecs/mozjpeg_decoder.rs:290-292, :625-627 ->
 * mozjpeg_decoder_helpers.rs:107-202): *flag = -1 when the file carries none, else the tag's value 0..8.  Host only. */
IFHIP_API int ifhip_jpeg_exif_orientation(const uint8_t* jpeg, size_t len, int* flag);
/* The embedded colour profile as MozJpegDecoder sees it (APP2 "ICC_PROFILE" chunks, codecs/mozjpeg_decoder.rs:370-420):
 * the reference transforms a frame to sRGB whenever the file carries ANY profile (:409, SourceProfile::is_srgb is true
 * only for "no profile") unless the job told the decoder discard_color_profile (:88-95).  This library has no colour
 * management (out of scope), so callers must know when a file needs it.  *kind = 0: no profile;
 * (also: a chunk set libjpeg's reassembly rule rejects -- mozjpeg_decoder_helpers.rs:42-83 returns None -- and a GRAY
 * profile on a colour frame, which the reference maps to SourceProfile::Srgb, mozjpeg_decoder.rs:391-395);
 * 1: a profile that describes sRGB itself (RGB matrix profile, sRGB primaries within 0.003 after D50 adaptation, the sRGB
 * tone curve) -- the reference's transform is the identity up to its own rounding; 2: any other profile (Display P3,
 * Adobe RGB, CMYK, grey on a grey frame, a profile too short to parse ...) -- decoding the samples as they are gives other colours than the
 * reference.  Host only. */
IFHIP_API int ifhip_jpeg_icc_profile_kind(const uint8_t* jpeg, size_t len, int* kind);
/* Host-side entropy decoding of the Huffman JPEGs the GPU entropy stage does not take -- progressive (SOF2: what the
 * reference's own mozjpeg encoder preset writes, codecs/mozjpeg.rs:121-123) and sequential files with several / non-
 * interleaved scans -- into the same coefficient planes ([blocks_h][blocks_w][64] int16, natural order, MCU-padded), so
 * that the pixel stage behind it is the GPU path unchanged (csrc/jpeg_read.cpp; libjpeg's jdphuff.c / jdhuff.c scans).
 * ifhip_jpeg_frame_info: the frame facts of any such file (and of baseline ones), *progressive = 1 for SOF2.
 * ifhip_jpeg_read_coefficients_host: planes sized by those facts (blocks_w * blocks_h * 64 each), cleared and filled;
 * qt3x64 (optional) receives each component's quantisation table. */
IFHIP_API int ifhip_jpeg_frame_info(const uint8_t* jpeg, size_t len, uint32_t* width, uint32_t* height, int* n_components,
                                    uint8_t* h_samp3, uint8_t* v_samp3, uint32_t* blocks_w3, uint32_t* blocks_h3,
                                    uint16_t* qt3x64, int* progressive);
IFHIP_API int ifhip_jpeg_read_coefficients_host(const uint8_t* jpeg, size_t len, int16_t* coef0, int16_t* coef1,
                                                int16_t* coef2, uint16_t* qt3x64);
IFHIP_API int ifhip_jpeg_entropy_create(ifhip_jpeg_entropy** out, const uint8_t* const* files, const size_t* lengths,
                                        uint32_t n_images);
IFHIP_API void ifhip_jpeg_entropy_destroy(ifhip_jpeg_entropy* e);
/* The same batch assembled from files that were PREPARED one by one (host only, any thread: parse, un-stuff, cut at the
 * restart markers, pack into pinned memory, derive the decode tables -- everything ifhip_jpeg_entropy_create does per
 * file).  A host that runs one job per thread prepares each job's file on that job's thread and hands whatever is waiting
 * to one create + decode call; the prepared handles may be destroyed as soon as create_prepared returns.  Errors of a
 * file (ImageMalformed, MethodNotImplemented) are reported by its prepare call. */
typedef struct ifhip_jpeg_prepared ifhip_jpeg_prepared;
IFHIP_API int ifhip_jpeg_entropy_prepare(ifhip_jpeg_prepared** out, const uint8_t* jpeg, size_t len);
IFHIP_API void ifhip_jpeg_prepared_destroy(ifhip_jpeg_prepared* p);
/* Optional, on a thread with a HIP device: queue the prepared scan's copy to the device on `hip_stream` NOW (asynchronous; the
 * call does not wait).  ifhip_jpeg_entropy_create_prepared then takes the words device to device behind that copy instead
 * of uploading them itself: with one job per thread the PCIe transfer of a job's file overlaps its wait for the next
 * coalesced batch instead of lengthening that batch.  A handle uploaded on one device and handed to a batch on another is
 * uploaded again from its pinned copy.  Destroying the handle waits for the queued copy if it is still running. */
IFHIP_API int ifhip_jpeg_prepared_upload(ifhip_jpeg_prepared* p, void* hip_stream);
IFHIP_API int ifhip_jpeg_prepared_info(const ifhip_jpeg_prepared* p, uint32_t* width, uint32_t* height, int* n_components,
                                       uint8_t* h_samp3, uint8_t* v_samp3);
IFHIP_API int ifhip_jpeg_entropy_create_prepared(ifhip_jpeg_entropy** out, ifhip_jpeg_prepared* const* prepared, uint32_t n_images);
IFHIP_API int ifhip_jpeg_entropy_info(const ifhip_jpeg_entropy* e, uint32_t* width, uint32_t* height, int* n_components,
                                      uint8_t* h_samp3, uint8_t* v_samp3, uint32_t* blocks_w3, uint32_t* blocks_h3,
                                      uint32_t* n_subsequences, uint32_t* n_segments);
IFHIP_API int ifhip_jpeg_entropy_quant_tables(const ifhip_jpeg_entropy* e, uint16_t* qt_n3x64);
IFHIP_API int ifhip_jpeg_entropy_decode_device(ifhip_jpeg_entropy* e, int16_t* d_coef0, int16_t* d_coef1,
                                               int16_t* d_coef2, uint32_t* rounds, void* hip_stream);

/* Encode-side pixel stage ("next" row 1): what libjpeg runs before entropy coding when
 * MozjpegEncoder::write_frame (codecs/mozjpeg.rs:78-160, classic preset = set_fastest_defaults, input JCS_EXT_BGRA /
 * JCS_EXT_BGRX) compresses a flattened frame: rgb_ycc_convert, chroma down-sampling with libjpeg's edge expansion,
 * the islow forward DCT, quantisation (round-half-up division by 8*Q) and the dummy blocks of the last MCU
 * column/row.  Output = the quantised coefficient planes jpeg_write_coefficients / the entropy coder consume
 * (same layout as the decode stage's input); sampling factors and quantisation tables come from the host
 * (evalchroma / set_quality stay host logic).  Call ifhip_apply_matte first when alpha is meaningful (:88-94). */
typedef struct ifhip_jpeg_fwd_stage ifhip_jpeg_fwd_stage;
IFHIP_API int ifhip_jpeg_fwd_stage_create(ifhip_jpeg_fwd_stage** stage, uint32_t width, uint32_t height,
                                          const uint8_t* h_samp, const uint8_t* v_samp, uint32_t max_images);
IFHIP_API void ifhip_jpeg_fwd_stage_destroy(ifhip_jpeg_fwd_stage* stage);
IFHIP_API int ifhip_jpeg_fwd_stage_block_dims(const ifhip_jpeg_fwd_stage* stage, uint32_t* blocks_w3, uint32_t* blocks_h3);
IFHIP_API int ifhip_jpeg_forward_batch_device(ifhip_jpeg_fwd_stage* stage, const uint8_t* d_bgra, size_t image_bytes,
                                              uint32_t stride, const uint16_t* d_qt, uint32_t n_images,
                                              int16_t* d_coef0, int16_t* d_coef1, int16_t* d_coef2, void* hip_stream);
IFHIP_API int ifhip_jpeg_forward(const uint8_t* bgra, uint32_t width, uint32_t height, uint32_t stride,
                                 const uint8_t* h_samp, const uint8_t* v_samp, const uint16_t* qt,
                                 int16_t* coef0, int16_t* coef1, int16_t* coef2);
/* Host half of the same encoder: jpeg_set_quality's tables (jcparam.c: Annex K tables scaled, force_baseline; [0] luma,
 * [1] chroma, natural order) and the baseline file writer -- markers in jcmarker.c's order plus the sequential Huffman
 * coder with the Annex K tables (set_fastest_defaults: no optimised tables, not progressive).  Byte-identical to
 * libjpeg-turbo for the same pixels, quality and sampling.  coef*: host planes in the layout of ifhip_jpeg_forward.
 * out == NULL: only *len (the size needed) is written. */
IFHIP_API int ifhip_jpeg_quality_tables(int quality, uint16_t* qt2x64);
IFHIP_API int ifhip_jpeg_write_baseline(const int16_t* coef0, const int16_t* coef1, const int16_t* coef2,
                                        const uint32_t* blocks_w3, const uint32_t* blocks_h3, int n_components,
                                        const uint8_t* h_samp, const uint8_t* v_samp, uint32_t width, uint32_t height,
                                        int quality, uint8_t* out, size_t capacity, size_t* len);
/* The same writer with the classic preset's two options (codecs/mozjpeg.rs:121-129: set_progressive_mode,
 * set_optimize_coding over set_fastest_defaults).  IFHIP_JPEG_OPTIMIZE_HUFFMAN: two passes per scan -- symbol statistics,
 * then jchuff.c jpeg_gen_optimal_table's codes in the DHT segments.  IFHIP_JPEG_PROGRESSIVE: SOF2 and jcparam.c
 * jpeg_simple_progression's scan script (10 scans for YCbCr, 6 for gray; spectral selection + successive approximation,
 * jcphuff.c's end-of-band runs and correction bits), every scan with its own optimal tables as libjpeg does.
 * Byte-identical to libjpeg-turbo (Pillow optimize=True / progressive=True) for the same coefficients. */
#define IFHIP_JPEG_OPTIMIZE_HUFFMAN 1
#define IFHIP_JPEG_PROGRESSIVE 2
IFHIP_API int ifhip_jpeg_write(const int16_t* coef0, const int16_t* coef1, const int16_t* coef2,
                               const uint32_t* blocks_w3, const uint32_t* blocks_h3, int n_components,
                               const uint8_t* h_samp, const uint8_t* v_samp, uint32_t width, uint32_t height,
                               int quality, int flags, uint8_t* out, size_t capacity, size_t* len);
/* The batch form: n_images of one geometry, planes [n_images][bh_c][bw_c][64] (what ifhip_jpeg_forward_batch_device leaves,
 * downloaded), coded on up to `threads` host threads (0: one per core, at most n_images).  Files are packed into `out`
 * in image order at offsets[i], lengths[i] long; out == NULL queries *total. */
IFHIP_API int ifhip_jpeg_write_batch(const int16_t* coef0, const int16_t* coef1, const int16_t* coef2,
                                     const uint32_t* blocks_w3, const uint32_t* blocks_h3, int n_components,
                                     const uint8_t* h_samp, const uint8_t* v_samp, uint32_t width, uint32_t height,
                                     int quality, int flags, uint32_t n_images, uint32_t threads,
                                     uint8_t* out, size_t capacity, size_t* offsets, size_t* lengths, size_t* total);

/* Device entropy coder: the same baseline files without the host loop.  Replaces, for the classic preset's default
 * (codecs/mozjpeg.rs:108-129 with neither progressive nor optimize_coding), what compressor.write_scanlines / finish
 * (mozjpeg.rs:155-175) run behind the pixel stage: jchuff.c encode_mcu_huff / encode_one_block / emit_bits with byte
 * stuffing, and jcmarker.c's segments around the scan.  Coefficient planes stay in HBM ([n_images][bh_c][bw_c][64], the
 * layout ifhip_jpeg_forward_batch_device leaves); image i's file is written to d_files + i * file_pitch and its length to
 * d_lengths[i] -- 0 when the image was dropped, with the reason in d_status[i] (nullable): IFHIP_ENC_BAD_COEFFICIENT (a
 * coefficient with more magnitude bits than 8-bit JPEG codes: JERR_BAD_DCT_COEF), IFHIP_ENC_SCAN_OVERFLOW (the scan is
 * longer than the stage's scan_capacity), IFHIP_ENC_FILE_OVERFLOW (the file is longer than file_pitch).  Asynchronous on
 * hip_stream.  scan_capacity: bound of an image's entropy-coded bytes before stuffing, 0 = the most the geometry can
 * produce (208 bytes per block), with which no image is ever dropped for the scan's length.  Files are byte-identical to
 * ifhip_jpeg_write_baseline's for the same coefficients and quality.
 * A stage is bound to ONE stream at a time: it owns the scratch of a call in flight (bit counts, the word stream, the
 * marker segments of the current quality), so a second call must be ordered behind the first -- same stream, or an
 * event between them; use one stage per stream for concurrent batches. */
#define IFHIP_ENC_BAD_COEFFICIENT 1
#define IFHIP_ENC_SCAN_OVERFLOW 2
#define IFHIP_ENC_FILE_OVERFLOW 4
typedef struct ifhip_jpeg_enc_stage ifhip_jpeg_enc_stage;
IFHIP_API int ifhip_jpeg_enc_stage_create(ifhip_jpeg_enc_stage** stage, uint32_t width, uint32_t height, int n_components,
                                          const uint8_t* h_samp, const uint8_t* v_samp, const uint32_t* blocks_w3,
                                          const uint32_t* blocks_h3, uint32_t max_images, size_t scan_capacity);
IFHIP_API void ifhip_jpeg_enc_stage_destroy(ifhip_jpeg_enc_stage* stage);
/* a file_pitch with which no file overflows (marker segments + every stream byte stuffed + EOI) */
/* The files of a batch as ONE message for the job's final gather: image i's file (d_lengths[i] bytes
 * at d_files + i * file_pitch, as ifhip_jpeg_encode_batch_device leaves it; a dropped image has length 0) is copied to
 * d_out + d_offsets[i], every start rounded up to 16 bytes; d_offsets[n_files] = the bytes used.  When that exceeds
 * out_capacity the files that do not fit are not copied (the caller compares).  file_pitch, d_files and d_out 16-byte
 * aligned; a file's padding bytes are unspecified.  Asynchronous on hip_stream. */
IFHIP_API int ifhip_pack_files_device(const uint8_t* d_files, size_t file_pitch, const uint32_t* d_lengths, uint32_t n_files,
                                      uint8_t* d_out, size_t out_capacity, uint64_t* d_offsets, void* hip_stream);
IFHIP_API size_t ifhip_jpeg_enc_stage_max_file_bytes(const ifhip_jpeg_enc_stage* stage);
IFHIP_API int ifhip_jpeg_encode_batch_device(ifhip_jpeg_enc_stage* stage, const int16_t* d_coef0, const int16_t* d_coef1,
                                             const int16_t* d_coef2, int quality, uint32_t n_images, uint8_t* d_files,
                                             size_t file_pitch, uint32_t* d_lengths, uint32_t* d_status, void* hip_stream);

/* Host, for tests: what the device coder works from -- the Annex K Huffman tables in encode form (dc0, ac0, dc1, ac1; 256
 * entries `code | length << 16`) and the marker segments in front of the scan (SOI ... SOS). */
IFHIP_API int ifhip_jpeg_debug_encode_tables(uint32_t* tabs4x256, int n_components, const uint8_t* h_samp, const uint8_t* v_samp,
                                             uint32_t width, uint32_t height, int quality, uint8_t* header, size_t capacity,
                                             size_t* header_len);

/* imageflow's 8x8 -> NxN spatial block scalers for the luma plane of a scaled decode: replaces
 * flow_scale_spatial[_srgb]_{1..7}x{1..7} (c_components/lib/codecs_jpeg_idct_fast.c, .h:17-43), the functions the IDCT
 * method selector installs for component 1 (codec_jpeg_wrapper.c:274-343).  `srgb` selects the linear-light variants.
 * Plane form: blocks_w x blocks_h blocks of 8x8 bytes in, n x n bytes out per block (device pointers).
 * Block form: host array [n_blocks][64] -> [n_blocks][n][n]. */
IFHIP_API int ifhip_scale_spatial_plane_device(const uint8_t* d_in, uint32_t in_pitch, uint32_t blocks_w,
                                               uint32_t blocks_h, int n, int srgb, uint8_t* d_out, uint32_t out_pitch,
                                               void* hip_stream);
IFHIP_API int ifhip_scale_spatial_blocks(const uint8_t* blocks, uint32_t n_blocks, int n, int srgb, uint8_t* out);
/* The tables behind them (rebuilt from populate_weights(Robidoux, n, 8) as tests/integration/variation.rs does):
 * int8 weights [7][8], log2 of each output's divisor [7], lut_srgb_to_linear[256], lut_linear_to_srgb[4096]. */
IFHIP_API int ifhip_block_scaler_tables(int n, int8_t* weights_7x8, uint8_t* log2_divisors_7,
                                        uint16_t* srgb_to_linear_256, uint8_t* linear_to_srgb_4096);

/* ---- Inner D: whole-bitmap operations around the resampler ------------- */
/* Byte-exact replacements for the bitmap primitives imageflow's graphs run between decode, resample and encode, so a
 * frame batch stays in HBM for the whole chain.  In-place unless a canvas is named; BGRA8, any 4-byte aligned stride.
 * Every *_batch_device call works on n_images equally shaped frames at a fixed byte pitch. */

/* graphics::color_matrix::window_bgra32_apply_color_matrix (graphics/color_matrix.rs:5-29): matrix25 = row-major
 * [[f32; 5]; 5] (host memory) as built by flow/nodes/color.rs:86-230 (ColorFilterSrgb; watermark opacity = Alpha(a),
 * flow/nodes/watermark.rs:165-172). */
IFHIP_API int ifhip_apply_color_matrix(uint8_t* bgra, uint32_t w, uint32_t h, uint32_t stride, const float* matrix25);
IFHIP_API int ifhip_apply_color_matrix_batch_device(uint8_t* d_bgra, size_t image_bytes, uint32_t n_images, uint32_t w,
                                                    uint32_t h, uint32_t stride, const float* matrix25, void* hip_stream);

/* graphics::copy_rect::copy_rectangle (graphics/copy_rect.rs:12-119; crop / clone / expand_canvas /
 * copy_rect_to_canvas, flow/nodes/clone_crop_fill_expand.rs).  Alpha bookkeeping as the reference: a Bgr32 canvas
 * receiving a Bgra32 input gets its alpha set to 255 and *canvas_alpha_meaningful = 1 (:47-54); a Bgr32 input copied
 * into a Bgra32 canvas has its own alpha set to 255 first (:64-66, the input bitmap is modified). */
IFHIP_API int ifhip_copy_rect(uint8_t* input, uint32_t in_w, uint32_t in_h, uint32_t in_stride, int in_alpha_meaningful,
                              uint8_t* canvas, uint32_t canvas_w, uint32_t canvas_h, uint32_t canvas_stride,
                              int* canvas_alpha_meaningful, uint32_t from_x, uint32_t from_y, uint32_t to_x,
                              uint32_t to_y, uint32_t w, uint32_t h);
IFHIP_API int ifhip_copy_rect_batch_device(uint8_t* d_in, size_t in_image_bytes, uint32_t in_w, uint32_t in_h,
                                           uint32_t in_stride, int in_alpha_meaningful, uint8_t* d_canvas,
                                           size_t canvas_image_bytes, uint32_t canvas_w, uint32_t canvas_h,
                                           uint32_t canvas_stride, int* canvas_alpha_meaningful, uint32_t from_x,
                                           uint32_t from_y, uint32_t to_x, uint32_t to_y, uint32_t w, uint32_t h,
                                           uint32_t n_images, void* hip_stream);

/* BitmapWindowMut::fill_rectangle (graphics/bitmaps.rs:1504-1548): [x1, x2) x [y1, y2) := color (Color32 0xAARRGGBB);
 * empty rectangles succeed, BlendWithMatte canvases only accept the full rectangle. */
IFHIP_API int ifhip_fill_rect(uint8_t* bgra, uint32_t w, uint32_t h, uint32_t stride, int compositing, uint32_t x1,
                              uint32_t y1, uint32_t x2, uint32_t y2, uint32_t color_bgra);
IFHIP_API int ifhip_fill_rect_batch_device(uint8_t* d_bgra, size_t image_bytes, uint32_t n_images, uint32_t w, uint32_t h,
                                           uint32_t stride, int compositing, uint32_t x1, uint32_t y1, uint32_t x2,
                                           uint32_t y2, uint32_t color_bgra, void* hip_stream);

/* BitmapWindowMut::normalize_unused_alpha (graphics/bitmaps.rs:1570-1576): alpha := 255 unless it is meaningful
 * (EnableTransparency, flow/nodes/enable_transparency.rs:70-84). */
IFHIP_API int ifhip_normalize_unused_alpha_batch_device(uint8_t* d_bgra, size_t image_bytes, uint32_t n_images, uint32_t w,
                                                        uint32_t h, uint32_t stride, int alpha_meaningful, void* hip_stream);

/* graphics::flip::flow_bitmap_bgra_flip_{vertical,horizontal}_safe (graphics/flip.rs:10-38); row padding stays. */
IFHIP_API int ifhip_flip_vertical(uint8_t* bgra, uint32_t w, uint32_t h, uint32_t stride);
IFHIP_API int ifhip_flip_horizontal(uint8_t* bgra, uint32_t w, uint32_t h, uint32_t stride);
IFHIP_API int ifhip_flip_vertical_batch_device(uint8_t* d_bgra, size_t image_bytes, uint32_t n_images, uint32_t w,
                                               uint32_t h, uint32_t stride, void* hip_stream);
IFHIP_API int ifhip_flip_horizontal_batch_device(uint8_t* d_bgra, size_t image_bytes, uint32_t n_images, uint32_t w,
                                                 uint32_t h, uint32_t stride, void* hip_stream);

/* graphics::transpose::bitmap_window_transpose (graphics/transpose.rs:95-121): to(y, x) = from(x, y); needs
 * from_w == to_h and from_h == to_w, distinct bitmaps (flow/nodes/rotate_flip_transpose.rs:150-153). */
IFHIP_API int ifhip_transpose(const uint8_t* from, uint32_t from_w, uint32_t from_h, uint32_t from_stride, uint8_t* to,
                              uint32_t to_w, uint32_t to_h, uint32_t to_stride);
IFHIP_API int ifhip_transpose_batch_device(const uint8_t* d_from, size_t from_image_bytes, uint32_t from_w,
                                           uint32_t from_h, uint32_t from_stride, uint8_t* d_to, size_t to_image_bytes,
                                           uint32_t to_w, uint32_t to_h, uint32_t to_stride, uint32_t n_images,
                                           void* hip_stream);

/* ---- measurement helpers (bench.py) -------------------------------------------------------------------- */
/* Runs `launches` back-to-back launches of the batch op on `hip_stream` bracketed by hipEvents on that stream
 * and returns the average milliseconds per launch. */
IFHIP_API int ifhip_time_scale_and_render_batch_device(const ifhip_resample_plan* plan,
                                                       const uint8_t* d_in, size_t in_image_bytes, uint32_t in_stride,
                                                       int in_alpha_meaningful, uint32_t n_images,
                                                       uint8_t* d_canvas, size_t canvas_image_bytes,
                                                       uint32_t canvas_w, uint32_t canvas_h, uint32_t canvas_stride,
                                                       uint32_t x, uint32_t y,
                                                       int working_space, int compositing, uint32_t matte_bgra,
                                                       int force_kernel, void* hip_stream, int launches,
                                                       float* avg_ms_per_launch);
/* device-to-device copy bandwidth probe (bytes read + bytes written per second), for the "measured roofline" note */
IFHIP_API int ifhip_measure_copy_bandwidth(size_t bytes, int iters, double* bytes_per_second);
/* read-only streaming probe (bytes read per second; 16-byte non-temporal loads, one XOR per load): the yardstick for
 * the resample kernel, whose traffic is 99.5 % reads */
IFHIP_API int ifhip_measure_read_bandwidth(size_t bytes, int iters, double* bytes_per_second);
/* the same with writes mixed in: one 16-byte vector stored per `read_vectors_per_write` vectors read (bytes read + written
 * per second) -- the yardstick for the moderate ratios, whose canvas stores are 15 - 36 % of their bytes: reads and writes
 * together run at 5.3 - 5.5 TB/s on MI355X where reads alone reach 7 */
IFHIP_API int ifhip_measure_mixed_bandwidth(size_t read_bytes, uint32_t read_vectors_per_write, int iters, double* bytes_per_second);

#ifdef __cplusplus
}
#endif
#endif /* IMAGEFLOW_HIP_H */
