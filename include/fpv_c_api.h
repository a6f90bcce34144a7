/*
 * fpv_c_api.h -- plain-C entry points over the C++ codec API of
 * fusion_power_video.h, for FFI callers (the Python tests and bench, or any
 * language binding).  Exported by libfpv_amd.so.
 */
#ifndef FPV_C_API_H_
#define FPV_C_API_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void (*fpvc_write_fn)(const uint8_t *data, size_t size, void *payload);

/* fpvc::Encoder (h:175-255) */
void *fpvc_encoder_new(size_t num_threads, int shift, int big_endian);
size_t fpvc_encoder_max_queued(const void *enc);
void fpvc_encoder_init(void *enc, const uint16_t *delta, size_t xs, size_t ys,
                       fpvc_write_fn cb, void *payload);
void fpvc_encoder_compress_frame(void *enc, const uint16_t *img, fpvc_write_fn cb,
                                 void *payload);
void fpvc_encoder_finish(void *enc, fpvc_write_fn cb, void *payload);
void fpvc_encoder_free(void *enc);
/* Encoder::LastStats (extension): stage statistics after Finish; see fusion_power_video.h. */
size_t fpvc_encoder_stats(const void *enc, double *out, size_t n);
/* Encoder::LastError (extension): the session's first GPU failure, "" while there is none.  After a
 * failure the session drops further frames and fpvc_encoder_finish emits no index. */
const char *fpvc_encoder_last_error(const void *enc);
/*
 * Benchmark helper: Encoder construction -> Finish over `total` frames cycling through
 * `distinct` resident ones (frame 0 doubles as the delta frame, as encode.cc does), timed
 * as benchmark.cc:153-180 times it, every chunk copied away by the callback.  Returns
 * the seconds; *bytes_out = stream size; stats as fpvc_encoder_stats.
 */
double fpvc_bench_encode(const uint16_t *frames, size_t distinct, size_t total, size_t xs,
                         size_t ys, int shift, int big_endian, size_t num_threads,
                         size_t *bytes_out, double *stats, size_t nstats);

/*
 * Whole-stream helpers.  encode: Init(delta) + CompressFrame x n + Finish into
 * out (capacity cap); returns the stream size or 0 when it does not fit.
 */
/* TEST HOOK: 1 when the library's arena-backed Brotli call gives exactly the bytes of libbrotli's
 * BrotliEncoderCompress(1, default window, default mode) for this input. */
int fpvc_test_brotli_equivalence(const uint8_t *in, size_t n);
size_t fpvc_encode_stream(const uint16_t *frames, size_t nframes,
                          const uint16_t *delta_raw, size_t xs, size_t ys,
                          int shift, int big_endian, size_t num_threads,
                          uint8_t *out, size_t cap);
/*
 * RandomAccessDecoder (h:143-172) over a whole stream: every frame (and, when
 * previews_out != NULL, every preview).  batched != 0 uses the DecodeFrames
 * extension.  Returns the frame count or -1.
 */
long fpvc_decode_stream(const uint8_t *data, size_t size, uint16_t *frames_out,
                        size_t cap_frames, uint8_t *previews_out, int batched,
                        size_t threads);
/* RandomAccessDecoder::DecodeFramesRaw over a whole stream: raw_out receives
 * nframes * xs * ys * 2 bytes, the file representation UnextractFrame (h:32) produces.
 * Returns the frame count or -1. */
long fpvc_decode_stream_raw(const uint8_t *data, size_t size, uint8_t *raw_out,
                            size_t cap_frames, int shift, int big_endian, size_t threads);
/* StreamingDecoder (h:37-57) fed in pieces of `block` bytes (0 = everything in one
 * call).  Returns the frame count, or -(1 + frames delivered before the failure). */
long fpvc_stream_decode(const uint8_t *data, size_t size, size_t block,
                        uint16_t *frames_out, size_t cap_frames);
/* Stream header peek: 0 on success. */
int fpvc_stream_info(const uint8_t *data, size_t size, size_t *xs, size_t *ys,
                     size_t *nframes);
void fpvc_unextract(const uint16_t *img, size_t xs, size_t ys, int shift,
                    int big_endian, uint8_t *out);


/* fpvc::Frame (h:75-139) drivers; see csrc/fpv_frame_capi.inc for semantics. */
size_t fpvc_frame_compress_full(const uint16_t *img, const uint16_t *delta_raw,
                                size_t xs, size_t ys, int shift, int big_endian,
                                uint8_t *out, size_t cap);
uint64_t fpvc_frame_cycle(const uint16_t *img, const uint16_t *delta_raw, size_t xs,
                          size_t ys, int shift, int big_endian, int parallel,
                          uint8_t *pred_high, uint8_t *pred_low, uint8_t *pred_preview,
                          uint8_t *info, uint8_t *rest_high, uint8_t *rest_low,
                          uint8_t *rest_preview);
size_t fpvc_frame8_compress_core(const uint8_t *img8, const uint8_t *delta8, size_t xs,
                                 size_t ys, uint8_t *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* FPV_C_API_H_ */
