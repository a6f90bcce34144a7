// fpv_c_api.cc -- extern "C" wrappers over the C++ codec API (fpv_c_api.h).
#include "fpv_c_api.h"

#include <string.h>

#include <chrono>
#include <vector>

#include <brotli/encode.h>

#include "fpv_host_internal.h"
#include "fusion_power_video.h"

namespace {
struct Sink {
  uint8_t* out;
  size_t cap, size;
  bool overflow;
};
void SinkWrite(const uint8_t* data, size_t size, void* payload) {
  Sink* s = static_cast<Sink*>(payload);
  if (s->size + size > s->cap) {
    s->overflow = true;
    return;
  }
  memcpy(s->out + s->size, data, size);
  s->size += size;
}
}  // namespace

extern "C" {

void* fpvc_encoder_new(size_t num_threads, int shift, int big_endian) {
  return new fpvc::Encoder(num_threads, shift, big_endian != 0);
}
size_t fpvc_encoder_max_queued(const void* enc) {
  return static_cast<const fpvc::Encoder*>(enc)->MaxQueued();
}
void fpvc_encoder_init(void* enc, const uint16_t* delta, size_t xs, size_t ys,
                       fpvc_write_fn cb, void* payload) {
  static_cast<fpvc::Encoder*>(enc)->Init(delta, xs, ys, cb, payload);
}
void fpvc_encoder_compress_frame(void* enc, const uint16_t* img, fpvc_write_fn cb,
                                 void* payload) {
  static_cast<fpvc::Encoder*>(enc)->CompressFrame(img, cb, payload);
}
void fpvc_encoder_finish(void* enc, fpvc_write_fn cb, void* payload) {
  static_cast<fpvc::Encoder*>(enc)->Finish(cb, payload);
}
void fpvc_encoder_free(void* enc) { delete static_cast<fpvc::Encoder*>(enc); }
size_t fpvc_encoder_stats(const void* enc, double* out, size_t n) {
  return static_cast<const fpvc::Encoder*>(enc)->LastStats(out, n);
}
const char* fpvc_encoder_last_error(const void* enc) { return static_cast<const fpvc::Encoder*>(enc)->LastError(); }

// Encoder construction -> Finish over `total` frames that cycle through `distinct`
// resident ones, timed like the upstream benchmark times it (benchmark.cc:153-180); the
// callbacks copy every chunk away as the upstream tools do.  Returns the seconds.
double fpvc_bench_encode(const uint16_t* frames, size_t distinct, size_t total, size_t xs, size_t ys,
                         int shift, int big_endian, size_t num_threads, size_t* bytes_out,
                         double* stats, size_t nstats) {
  struct Out { std::vector<uint8_t> last; size_t bytes; } out = {{}, 0};
  auto cb = [](const uint8_t* d, size_t n, void* p) {
    Out* o = static_cast<Out*>(p);
    o->last.assign(d, d + n);
    o->bytes += n;
  };
  const auto t0 = std::chrono::steady_clock::now();
  {
    fpvc::Encoder enc(num_threads, shift, big_endian != 0);
    enc.Init(frames, xs, ys, cb, &out);
    for (size_t i = 0; i < total; i++) enc.CompressFrame(frames + (i % distinct) * xs * ys, cb, &out);
    enc.Finish(cb, &out);
    if (stats) enc.LastStats(stats, nstats);
  }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (bytes_out) *bytes_out = out.bytes;
  return dt;
}

// TEST HOOK: 1 when the arena-backed Brotli call gives exactly what libbrotli's one-shot entry
// point gives for this input (tests/test_host_cpu.py), 0 otherwise.
int fpvc_test_brotli_equivalence(const uint8_t* in, size_t n) {
  const size_t cap = BrotliEncoderMaxCompressedSize(n) + 1024;
  std::vector<uint8_t> a(cap), b(cap);
  size_t sa = cap, sb = cap;
  const bool oa = fpvc::internal::BrotliCompress(in, n, a.data(), &sa);
  const bool ob = BrotliEncoderCompress(1, BROTLI_DEFAULT_WINDOW, BROTLI_DEFAULT_MODE, n, in, &sb, b.data()) != 0;
  return oa == ob && (!oa || (sa == sb && memcmp(a.data(), b.data(), sa) == 0));
}

size_t fpvc_encode_stream(const uint16_t* frames, size_t nframes,
                          const uint16_t* delta_raw, size_t xs, size_t ys, int shift,
                          int big_endian, size_t num_threads, uint8_t* out, size_t cap) {
  Sink sink = {out, cap, 0, false};
  fpvc::Encoder enc(num_threads, shift, big_endian != 0);
  enc.Init(delta_raw, xs, ys, SinkWrite, &sink);
  for (size_t i = 0; i < nframes; i++)
    enc.CompressFrame(frames + i * xs * ys, SinkWrite, &sink);
  enc.Finish(SinkWrite, &sink);
  return sink.overflow ? 0 : sink.size;
}

long fpvc_decode_stream(const uint8_t* data, size_t size, uint16_t* frames_out,
                        size_t cap_frames, uint8_t* previews_out, int batched,
                        size_t threads) {
  fpvc::RandomAccessDecoder dec;
  if (!dec.Init(data, size)) return -1;
  const size_t n = dec.numframes();
  if (n > cap_frames) return -1;
  const size_t npix = dec.xsize() * dec.ysize();
  const size_t npv = dec.preview_xsize() * dec.preview_ysize();
  if (batched) {
    std::vector<size_t> idx(n);
    std::vector<uint16_t*> outs(n);
    for (size_t i = 0; i < n; i++) {
      idx[i] = i;
      outs[i] = frames_out + i * npix;
    }
    if (n && !dec.DecodeFrames(idx.data(), n, outs.data(), threads)) return -1;
  } else {
    for (size_t i = 0; i < n; i++)
      if (!dec.DecodeFrame(i, frames_out + i * npix)) return -1;
  }
  if (previews_out)
    for (size_t i = 0; i < n; i++)
      if (!dec.DecodePreview(i, previews_out + i * npv)) return -1;
  return static_cast<long>(n);
}

long fpvc_decode_stream_raw(const uint8_t* data, size_t size, uint8_t* raw_out, size_t cap_frames,
                            int shift, int big_endian, size_t threads) {
  fpvc::RandomAccessDecoder dec;
  if (!dec.Init(data, size)) return -1;
  const size_t n = dec.numframes();
  if (n > cap_frames) return -1;
  const size_t fbytes = dec.xsize() * dec.ysize() * 2;
  std::vector<size_t> idx(n);
  std::vector<uint8_t*> outs(n);
  for (size_t i = 0; i < n; i++) {
    idx[i] = i;
    outs[i] = raw_out + i * fbytes;
  }
  if (n && !dec.DecodeFramesRaw(idx.data(), n, outs.data(), shift, big_endian != 0, threads)) return -1;
  return static_cast<long>(n);
}

long fpvc_stream_decode(const uint8_t* data, size_t size, size_t block,
                        uint16_t* frames_out, size_t cap_frames) {
  struct St {
    uint16_t* out;
    size_t cap, n;
    bool bad;
  } st = {frames_out, cap_frames, 0, false};
  fpvc::StreamingDecoder dec;
  if (block == 0) block = size ? size : 1;
  for (size_t pos = 0; pos < size; pos += block) {
    const size_t len = size - pos < block ? size - pos : block;
    dec.Decode(data + pos, len,
               [](bool ok, uint16_t* frame, size_t xs, size_t ys, void* p) {
                 St* s = static_cast<St*>(p);
                 if (!ok || s->n >= s->cap) {
                   s->bad = true;
                   return;
                 }
                 memcpy(s->out + s->n * xs * ys, frame, xs * ys * 2);
                 s->n++;
               },
               &st);
    if (st.bad) return -1 - static_cast<long>(st.n);
  }
  return static_cast<long>(st.n);
}

int fpvc_stream_info(const uint8_t* data, size_t size, size_t* xs, size_t* ys,
                     size_t* nframes) {
  if (size < 21) return -1;
  auto rd32 = [&](size_t p) {
    return (size_t)data[p] | (size_t)data[p + 1] << 8 | (size_t)data[p + 2] << 16 |
           (size_t)data[p + 3] << 24;
  };
  *xs = rd32(0);
  *ys = rd32(4);
  *nframes = rd32(size - 8) | (rd32(size - 4) << 32);
  return 0;
}

void fpvc_unextract(const uint16_t* img, size_t xs, size_t ys, int shift,
                    int big_endian, uint8_t* out) {
  fpvc::UnextractFrame(img, xs, ys, shift, big_endian != 0, out);
}

#define FPV_FN(name) fpvc_##name
#include "fpv_frame_capi.inc"
#undef FPV_FN

}  // extern "C"
