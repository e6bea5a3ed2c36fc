/*
 * imageflow_abi_subset.h -- libimageflow's C ABI v3.2 as libimageflow_hip.so re-exports it: all 25 functions of
 * bindings/headers/imageflow_default.h, so that a C caller (or any language binding generated from that header) can run
 * resize jobs on the MI355X path without the Rust host.  Same names, argument order and ownership rules as the
 * reference: imageflow_abi/src/lib.rs (line numbers per function below), ABI version imageflow_abi/src/abi_version.rs:4,7.
 * "Subset" is what a JOB may contain, not the function list.
 *
 * Endpoints of imageflow_context_send_json: v1/build, v1/execute, v1/get_image_info, v1/tell_decoder (+ v0.1/ aliases),
 * v1/get_scaled_image_info, v1/get_version_info; everything else answers 404 as json/mod.rs:158-168.
 *
 * What a job may contain (anything else answers ActionNotSupported, HTTP 400): decode (baseline JPEG, or the raw
 * BGRA container EXTENSION), create_canvas, fill_rect, expand_canvas, crop, flip_h/flip_v, transpose, rotate_90/180/270,
 * apply_orientation, color_matrix_srgb, color_filter_srgb, resample_2d, draw_image_exact and copy_rect_to_canvas (graph
 * form, with a `canvas` edge), watermark (all five fit modes), constrain (all nine modes, gravity, canvas_color),
 * command_string (ir4: width/height, mode=max), encode.
 * `encode` with the libjpeg_turbo preset writes a real JPEG (quality, matte, progressive, optimize_huffman_coding:
 * byte-identical to libjpeg-turbo's file for the same pixels at 4:2:0); every other preset writes the raw BGRA container
 * EXTENSION (this library has no deflate / GIF / WebP coder): 8 bytes "IFBGRA1\0", u32le w, h, stride,
 * alpha_meaningful, then h rows of `stride` bytes.
 * Implementation: imageflow_amd/csrc/abi_shim.cpp.
 */
#ifndef IMAGEFLOW_ABI_SUBSET_H
#define IMAGEFLOW_ABI_SUBSET_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#define IMAGEFLOW_ABI_VER_MAJOR 3
#define IMAGEFLOW_ABI_VER_MINOR 2
#define IMAGEFLOW_SHIM_API __attribute__((visibility("default")))

struct imageflow_context;
struct imageflow_json_response;

/* imageflow_abi/src/lib.rs:281-288 */
typedef enum imageflow_lifetime {
  imageflow_lifetime_lifetime_outlives_function_call = 0,
  imageflow_lifetime_lifetime_outlives_context = 1,
} imageflow_lifetime;

#ifdef __cplusplus
extern "C" {
#endif

IMAGEFLOW_SHIM_API bool imageflow_abi_compatible(uint32_t imageflow_abi_ver_major, uint32_t imageflow_abi_ver_minor);  /* :389 */
IMAGEFLOW_SHIM_API uint32_t imageflow_abi_version_major(void);                                                         /* :398 */
IMAGEFLOW_SHIM_API uint32_t imageflow_abi_version_minor(void);                                                         /* :403 */

IMAGEFLOW_SHIM_API struct imageflow_context *imageflow_context_create(uint32_t imageflow_abi_ver_major,
                                                                      uint32_t imageflow_abi_ver_minor);                /* :430 */
IMAGEFLOW_SHIM_API bool imageflow_context_begin_terminate(struct imageflow_context *context);                          /* :459 */
IMAGEFLOW_SHIM_API void imageflow_context_destroy(struct imageflow_context *context);                                  /* :492 */

IMAGEFLOW_SHIM_API bool imageflow_context_has_error(struct imageflow_context *context);                                /* :530 */
IMAGEFLOW_SHIM_API bool imageflow_context_error_recoverable(struct imageflow_context *context);                        /* :549 */
IMAGEFLOW_SHIM_API bool imageflow_context_error_try_clear(struct imageflow_context *context);                          /* :572 */
IMAGEFLOW_SHIM_API int32_t imageflow_context_error_code(struct imageflow_context *context);                            /* :600 */
IMAGEFLOW_SHIM_API int32_t imageflow_context_error_as_exit_code(struct imageflow_context *context);                    /* :620 */
IMAGEFLOW_SHIM_API int32_t imageflow_context_error_as_http_code(struct imageflow_context *context);                    /* :645 */
IMAGEFLOW_SHIM_API bool imageflow_context_error_write_to_buffer(struct imageflow_context *context, char *buffer,
                                                                size_t buffer_length, size_t *bytes_written);          /* :684 */
/* prints the error and calls exit(imageflow_context_error_as_exit_code) -- does not return when there is one */
IMAGEFLOW_SHIM_API bool imageflow_context_print_and_exit_if_error(struct imageflow_context *context);                  /* :744 */
/* the one call made from another thread while a job runs: sets the flag the job polls before every node, at every frame
 * allocation and inside decode / encode; the job then fails with category 21 (HTTP 499, exit 130) */
IMAGEFLOW_SHIM_API void imageflow_context_request_cancellation(struct imageflow_context *context);                     /* :878 */

IMAGEFLOW_SHIM_API const struct imageflow_json_response *imageflow_context_send_json(struct imageflow_context *context,
                                                                                     const char *method,
                                                                                     const uint8_t *json_buffer,
                                                                                     size_t json_buffer_size);         /* :944 */
IMAGEFLOW_SHIM_API bool imageflow_json_response_read(struct imageflow_context *context,
                                                     const struct imageflow_json_response *response_in,
                                                     int64_t *status_as_http_code_out,
                                                     const uint8_t **buffer_utf8_no_nulls_out,
                                                     size_t *buffer_size_out);                                         /* :783 */
IMAGEFLOW_SHIM_API bool imageflow_json_response_destroy(struct imageflow_context *context,
                                                        struct imageflow_json_response *response);                     /* :842 */

IMAGEFLOW_SHIM_API bool imageflow_context_add_input_buffer(struct imageflow_context *context, int32_t io_id,
                                                           const uint8_t *buffer, size_t buffer_byte_count,
                                                           imageflow_lifetime lifetime);                               /* :1137 */
IMAGEFLOW_SHIM_API bool imageflow_context_add_output_buffer(struct imageflow_context *context, int32_t io_id);         /* :1224 */
IMAGEFLOW_SHIM_API bool imageflow_context_get_output_buffer_by_id(struct imageflow_context *context, int32_t io_id,
                                                                  const uint8_t **result_buffer,
                                                                  size_t *result_buffer_length);                       /* :1272 */

/* ownership of the output bytes moves to the caller; refused once get_output_buffer_by_id lent a pointer, or twice */
IMAGEFLOW_SHIM_API bool imageflow_context_take_output_buffer(struct imageflow_context *context, int32_t io_id,
                                                             uint8_t **result_buffer, size_t *result_buffer_length);   /* :1335 */
IMAGEFLOW_SHIM_API bool imageflow_buffer_free(uint8_t *buffer, size_t length);                                         /* :1385 */

IMAGEFLOW_SHIM_API void *imageflow_context_memory_allocate(struct imageflow_context *context, size_t bytes,
                                                           const char *filename, int32_t line);                        /* :1424 */
IMAGEFLOW_SHIM_API bool imageflow_context_memory_free(struct imageflow_context *context, void *pointer,
                                                      const char *filename, int32_t line);                             /* :1484 */

/* Not part of libimageflow's header: the poll countdown debug builds of the reference carry for their own cancellation
 * test (Context::request_cancellation_after_n_polls, imageflow_core/src/context.rs:96-104,167-172). */
IMAGEFLOW_SHIM_API void ifhip_shim_request_cancellation_after_n_polls(struct imageflow_context *context, int64_t polls);
IMAGEFLOW_SHIM_API int64_t ifhip_shim_cancellation_polls_remaining(struct imageflow_context *context);
/* Diagnostic: how many decode -> resample pairs of this context's jobs ran as ONE device call without a decoded bitmap in
 * HBM (ifhip_jpeg_decode_resample_batch_device reporting fused = 1). */
IMAGEFLOW_SHIM_API int64_t ifhip_shim_fused_decode_resamples(struct imageflow_context *context);
/* Diagnostic: how many JPEG outputs of this context's jobs were entropy-coded on the device (libjpeg_turbo preset without
 * progressive / optimize_huffman_coding: ifhip_jpeg_encode_batch_device; only the file is downloaded). */
IMAGEFLOW_SHIM_API int64_t ifhip_shim_device_coded_files(struct imageflow_context *context);
/* Diagnostic: how many decodes of this context's jobs shared their entropy-decode device call with the job of another
 * thread (concurrent decodes of one geometry are coalesced into one batch; one context per thread, lib.rs:20-27). */
IMAGEFLOW_SHIM_API int64_t ifhip_shim_coalesced_decodes(struct imageflow_context *context);
/* Independent jobs over the GPUs of one node, without a process per GPU: a context may be bound to a device ordinal and
 * every job it runs (whichever thread calls) runs there -- the reference's "one Context per thread"
 * (imageflow_abi/src/lib.rs:20-27) then shards a batch of jobs by construction, with no collective (outputs are host
 * buffers).  ifhip_shim_spread_contexts(1): contexts created from now on take the usable devices round-robin;
 * _context_set_device binds one context (-1: follow the calling thread's current device, the default). */
IMAGEFLOW_SHIM_API void ifhip_shim_spread_contexts(int enable);
IMAGEFLOW_SHIM_API bool ifhip_shim_context_set_device(struct imageflow_context *context, int ordinal);
IMAGEFLOW_SHIM_API int ifhip_shim_context_device(struct imageflow_context *context);

/* The layout arithmetic behind the `constrain` and `watermark` nodes, callable on its own (no GPU, no context):
 * imageflow_riapi::ir4::process_constraint (imageflow_riapi/src/ir4/layout.rs:334-412) for a source of source_w x source_h
 * and a Constraint {mode, w, h, gravity}.  mode: "distort" | "within" | "fit" | "larger_than" | "within_crop" | "fit_crop" |
 * "aspect_crop" | "within_pad" | "fit_pad"; w / h < 0: not given; has_gravity 0: centre.  Out: crop_x1y1x2y2[4] (valid when
 * bit 0 of *flags), scale_to_wh[2], pad_ltrb[4] (valid when bit 1 of *flags), canvas_wh[2].  Returns 0, 1 for a LayoutError
 * (the reference's Err), 2 for an unknown mode or a null pointer. */
IMAGEFLOW_SHIM_API int ifhip_shim_process_constraint(const char *mode, int32_t source_w, int32_t source_h, int64_t w, int64_t h,
                                                     int has_gravity, float gravity_x, float gravity_y,
                                                     uint32_t *crop_x1y1x2y2, int32_t *scale_to_wh, uint32_t *pad_ltrb,
                                                     int32_t *canvas_wh, int *flags);

#ifdef __cplusplus
}
#endif
#endif /* IMAGEFLOW_ABI_SUBSET_H */
