/*
 * fpv_hip.h -- C ABI of the MI355X predictor/residual engine (libfpvhip.so).
 *
 * This is the boundary between the g++-compiled host codec
 * (fusion_power_video.h: fpvc::Encoder / StreamingDecoder /
 * RandomAccessDecoder / Frame) and the hipcc-compiled gfx950 kernels.  Plain
 * pointers and sizes only; no C++ or torch types.  Every entry point names the
 * loop(s) of the upstream reference it replaces ("cc:N" =
 * fusion_power_video.cc line N, "h:N" = fusion_power_video.h).
 *
 * Conventions
 *   - int results: 0 = ok, < 0 = error (text via fpvhip_last_error).  Nothing
 *     throws.  There is NO CPU fallback: without a usable GPU every call fails.
 *   - One context per (device, frame geometry).  Distinct slots of a context
 *     may be driven from distinct host threads; one slot, and the set_delta_* /
 *     *_device entry points, by one thread at a time.  "slot" selects one of the context's independent
 *     staging sets (pinned host buffers + device buffers + one HIP stream), so
 *     that slot A can be in flight while the host fills or drains slot B.
 *   - Planes are packed per frame: frame k of a batch starts at k * npix bytes
 *     (high, low), k * npix * 2 bytes (pixels) or k * npreview bytes (preview),
 *     npix = xs * ys, npreview = (xs / 4) * (ys / 4).
 *   - Flags byte per frame: bit0 USE_DELTA, bit1 USE_CG, bit2 NO_LOW_BYTES
 *     (h:68-73).
 */
#ifndef FPV_HIP_H_
#define FPV_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FPVHIP_ABI_VERSION 1

typedef struct fpvhip_ctx fpvhip_ctx;

/* Library / device discovery. */
int fpvhip_abi_version(void);
/* Number of visible HIP devices, or < 0 when the runtime cannot be used. */
int fpvhip_device_count(void);
/* Initialises the HIP runtime, the device's context and the kernels' code
 * object ahead of the first create (100s of ms once per process). */
int fpvhip_warmup(int device);
/* Last error text of ctx, or of the calling thread when ctx == NULL. */
const char *fpvhip_last_error(const fpvhip_ctx *ctx);

/*
 * Create a context for xs*ys frames on `device`.
 * shift / big_endian: as Frame's ctor (cc:370-371); shift 0..8 (0..15 when
 * !big_endian).  max_batch: frames per submit; slots: staging sets (>= 1).
 * xs and ys must be multiples of 4 (the reference reads out of bounds in its
 * preview predictor otherwise, cc:575-586).  Returns NULL on failure
 * (fpvhip_last_error(NULL)).
 */
fpvhip_ctx *fpvhip_create(int device, uint32_t xs, uint32_t ys, int shift,
                          int big_endian, uint32_t max_batch, uint32_t slots);
/*
 * Decoder-only context: any non-zero xs, ys (preview planes are (xs/4)x(ys/4),
 * files written by the CPU reference may have odd sizes).  Only the
 * set_delta_decoded / stage_plane / inverse_* entry points work on it.
 */
fpvhip_ctx *fpvhip_create_decoder(int device, uint32_t xs, uint32_t ys,
                                  uint32_t max_batch, uint32_t slots);
void fpvhip_destroy(fpvhip_ctx *ctx);

/* Geometry helpers. */
size_t fpvhip_npix(const fpvhip_ctx *ctx);
size_t fpvhip_npreview(const fpvhip_ctx *ctx);
uint32_t fpvhip_max_batch(const fpvhip_ctx *ctx);
uint32_t fpvhip_slots(const fpvhip_ctx *ctx);

/*
 * Encoder side delta frame: raw pixels as handed to Encoder::Init.  Splits them
 * on the device into the RAW high/low planes that every later frame is
 * predicted against (replaces Frame ctor cc:370-451 for delta_frame_,
 * cc:1097).  Synchronous.
 */
int fpvhip_set_delta_raw(fpvhip_ctx *ctx, const uint16_t *delta_raw);
/*
 * Decoder side delta frame: the DECODED delta image (uint16, as produced by
 * DecompressImage for the delta chunk, cc:903 / cc:984).  Synchronous.
 */
int fpvhip_set_delta_decoded(fpvhip_ctx *ctx, const uint16_t *delta_img);

/*
 * Delta frame given as RAW byte planes (Frame API: the delta Frame's high()/
 * low(), cc:534-537).  low == NULL means an absent low plane (zeros).
 */
int fpvhip_set_delta_planes(fpvhip_ctx *ctx, const uint8_t *high,
                            const uint8_t *low);

/*
 * Frame ctor alone (cc:370-451): raw split of ONE frame into its byte planes,
 * *no_low = 1 when every low byte is zero (NO_LOW_BYTES, cc:447-449).  Host
 * pointers; synchronous; uses slot 0.  low may be NULL.
 */
int fpvhip_split_frame(fpvhip_ctx *ctx, const uint16_t *img, uint8_t *high,
                       uint8_t *low, int *no_low);

/* ---- forward predictor, host-staged (Encoder pipeline) ------------------ */

/*
 * Allocate the staging memory of `slot` now (it is otherwise allocated when the slot is
 * first touched): lets a pipeline prepare its next slot on another thread while the
 * current one is being filled.  Thread-safe; a no-op for a slot that exists.
 */
int fpvhip_prepare_slot(fpvhip_ctx *ctx, uint32_t slot);

/* Bytes of pinned host staging memory the context holds right now (the same amount again
 * lives on the device). */
size_t fpvhip_staging_bytes(const fpvhip_ctx *ctx);
/* Give back every slot but the first (none may be in flight); they are allocated again when
 * next touched.  For caches of idle contexts. */
int fpvhip_trim(fpvhip_ctx *ctx);

/* Pinned host buffer for input frame k of `slot` (npix uint16). */
uint16_t *fpvhip_stage_in(fpvhip_ctx *ctx, uint32_t slot, uint32_t k);
/*
 * Queue H2D copy -> split + preview + delta/CG decisions + residuals -> D2H copy
 * for frames 0..nframes-1 of `slot` on the slot's stream and return at once.
 * Replaces Frame ctor + Frame::Predict (cc:370-451, cc:491-593, cc:777-785).
 * with_delta == 0 predicts without a delta frame (the delta frame itself,
 * cc:1099-1100).
 */
int fpvhip_predict_submit(fpvhip_ctx *ctx, uint32_t slot, uint32_t nframes,
                          int with_delta);
/*
 * Block until the slot's batch is back in pinned host memory; returns pointers
 * valid until the next submit on this slot.  low is meaningless (not written)
 * when shift == 8.
 */
int fpvhip_predict_wait(fpvhip_ctx *ctx, uint32_t slot, const uint8_t **high,
                        const uint8_t **low, const uint8_t **preview,
                        const uint8_t **flags);

/*
 * Frame::Predict on a frame that exists only as RAW byte planes (Frame ctors 2
 * and 3, h:105-107): the planes staged with fpvhip_stage_plane are the INPUT.
 * Needs a context created with shift 0 / little endian.  Wait with
 * fpvhip_predict_wait.
 */
int fpvhip_predict_planes_submit(fpvhip_ctx *ctx, uint32_t slot, uint32_t nframes,
                                 int with_delta);

/* ---- inverse predictor, host-staged (decoders) -------------------------- */

/* Pinned host buffer for plane `which` (0 high, 1 low) of frame k of `slot`:
 * the Brotli decoder writes straight into it. */
uint8_t *fpvhip_stage_plane(fpvhip_ctx *ctx, uint32_t slot, uint32_t k,
                            int which);
/*
 * Queue H2D -> inverse CG -> inverse delta + recombine -> D2H for nframes
 * frames.  flags[k] is frame k's image flags byte (bit2: low plane absent =
 * zeros).  Replaces cc:326-344 of DecompressImage.
 */
int fpvhip_inverse_submit(fpvhip_ctx *ctx, uint32_t slot, uint32_t nframes,
                          const uint8_t *flags);
/*
 * Same, with UnextractFrame (cc:850-862) applied on the device before the copy back: the
 * 16-bit words fpvhip_inverse_wait hands out are the raw file representation
 * (pixel >> unshift, stored big-endian when big_endian != 0).  unshift 0..15.
 */
int fpvhip_inverse_submit_raw(fpvhip_ctx *ctx, uint32_t slot, uint32_t nframes,
                              const uint8_t *flags, int unshift, int big_endian);
int fpvhip_inverse_wait(fpvhip_ctx *ctx, uint32_t slot,
                        const uint16_t **frames);
/*
 * Frame::Uncompress's predictor half (cc:612-641, 595-610): like inverse_submit,
 * but the result comes back as byte planes in the plane staging buffers (wait
 * with fpvhip_predict_wait; only high and low are meaningful).
 */
int fpvhip_inverse_planes_submit(fpvhip_ctx *ctx, uint32_t slot, uint32_t nframes,
                                 const uint8_t *flags);

/* ---- device-resident entry points (inputs already in HBM) --------------- */

/*
 * Same computation as predict_submit, but on caller-owned device buffers and
 * on the caller's HIP stream (hipStream_t passed as void*; NULL = default
 * stream).  d_frames: nframes*npix uint16; d_high/d_low: nframes*npix bytes
 * (d_low may be NULL when shift == 8); d_preview: nframes*npreview;
 * d_flags: nframes bytes.  nframes <= max_batch.  Asynchronous.
 */
int fpvhip_predict_device(fpvhip_ctx *ctx, const void *d_frames,
                          uint32_t nframes, int with_delta, void *d_high,
                          void *d_low, void *d_preview, void *d_flags,
                          void *stream);
/*
 * Inverse on device buffers: d_high is consumed read-only, d_low may be NULL
 * (all frames NO_LOW), d_flags nframes bytes, d_frames out nframes*npix
 * uint16.  Asynchronous on `stream`.
 */
int fpvhip_inverse_device(fpvhip_ctx *ctx, const void *d_high,
                          const void *d_low, const void *d_flags,
                          uint32_t nframes, void *d_frames, void *stream);

/*
 * Timing helper for benchmarks: runs predict_device `iters` times back to back
 * on `stream` bracketed by HIP events recorded on that same stream and returns
 * the elapsed milliseconds (< 0 on error).  kernel_ms, when not NULL, receives
 * the per-kernel average milliseconds [stats, decide, apply, preview] measured
 * with events around each launch in a second, separate pass.
 */
double fpvhip_time_predict_device(fpvhip_ctx *ctx, const void *d_frames,
                                  uint32_t nframes, int with_delta,
                                  void *d_high, void *d_low, void *d_preview,
                                  void *d_flags, void *stream, int iters,
                                  double *kernel_ms);
/* Diagnostics: the first n (<= 64) dwords of the device-entry-point workspace's
 * counter block after the last predict_device (list sizes, votes, task counter). */
int fpvhip_debug_counters(fpvhip_ctx *ctx, uint32_t *out, uint32_t n);
/* Stage timing: milliseconds the slot's stream spent on its last completed batch
 * (first transfer in -> last transfer out, HIP events on that stream); < 0 when
 * nothing has completed or a batch is still in flight.  Call after *_wait. */
double fpvhip_slot_elapsed_ms(fpvhip_ctx *ctx, uint32_t slot);
/* Diagnostics of the speculative inverse kernel (cc:326-333 replacement), four
 * dwords: out[0] = supersteps, out[1] = chain passes executed by all inverse
 * launches of this context since the last reset (passes / supersteps = average
 * passes; 1 = every guess was right, 64 = fully serial), out[2] = most passes
 * spent on one frame, out[3] = most passes spent on one superstep.
 * Synchronises the device. */
int fpvhip_inverse_pass_stats(fpvhip_ctx *ctx, uint32_t *out, int reset);
/* Diagnostics: the inverse kernel a batch of nframes takes on this context's device
 * ("k_inverse_trio", "k_inverse_spec", "k_inverse_fast", "k_inverse"). */
const char *fpvhip_inverse_kernel_name(fpvhip_ctx *ctx, uint32_t nframes);
/* Diagnostics: the three launches of one forward batch on this context, comma separated
 * ("k_tile,k_tile_decide,k_tile_finish" for the tile pass, "k_tile8,k_frame8,k_redo8" for its
 * 8-pixel variant (FPV_FORWARD=tile8), "k_stream,k_redo_decide,k_redo_stream" for geometries
 * that keep the streaming kernels): the order of kernel_ms[] of
 * fpvhip_time_predict_device. */
const char *fpvhip_forward_kernel_names(fpvhip_ctx *ctx);
/* Diagnostics: the shape of the tile pass on this context: out[0] = 4-row groups per tile,
 * out[1] = threads per workgroup, out[2] = tiles per frame, out[3] = log2 of the frames per group
 * that large frames are walked in tile by tile (0: frame by frame, the memory order).  All zero
 * when the streaming kernels are taken. */
int fpvhip_forward_tile_shape(fpvhip_ctx *ctx, uint32_t out[4]);
/* Diagnostics / recovery: the three-wave inverse kernel keeps one word per CU of "SIMDs that
 * already run some frame's chain wave" in device memory (a placement hint: results never depend
 * on it; DecompressImage's loop cc:326-344 has no such notion).  A kernel killed in mid-flight
 * leaves bits set; this entry overwrites every word of the context's device with `value` (0 =
 * all free; 0xf = all taken, which sends every workgroup through the kernel's fallback choice:
 * tests).  Synchronises the device. */
int fpvhip_debug_set_chain_mask(fpvhip_ctx *ctx, uint32_t value);
/* Bench only: `iters` plain 16-B-per-lane device copies of `bytes` bytes on
 * `stream`; returns elapsed ms.  The measured HBM ceiling kernels are compared with. */
double fpvhip_time_copy_device(void *d_dst, const void *d_src, size_t bytes,
                               void *stream, int iters);
/* The same for a named copy shape: 1 = one 16-byte element per thread, workgroups dispatched
 * in memory order (what fpvhip_time_copy_device measures: the fastest shape found on
 * MI355X), 0 = persistent workgroups copying 256 KiB chunks (round 2's reference). */
double fpvhip_time_copy_shape(void *d_dst, const void *d_src, size_t bytes,
                              void *stream, int iters, int shape);
double fpvhip_time_inverse_device(fpvhip_ctx *ctx, const void *d_high,
                                  const void *d_low, const void *d_flags,
                                  uint32_t nframes, void *d_frames,
                                  void *stream, int iters);

#ifdef __cplusplus
}
#endif
#endif /* FPV_HIP_H_ */
