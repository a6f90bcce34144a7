// ref_shim.cc -- TEST INFRASTRUCTURE ONLY.
//
// A thin extern "C" driver over the REAL reference library, compiled from the
// upstream sources where they lie (/root/reference) into oracle/_ref/ by
// oracle/Makefile.  It contains no codec logic of its own: every function just
// calls the reference's public API (fusion_power_video.h) so that Python tests
// and the golden-vector generator can compare the oracle restatement and the
// HIP path with the reference itself.  Exists only in the build container;
// /root/reference is absent on the GPU box.
#include <stdint.h>
#include <string.h>

#include <chrono>
#include <vector>

#include "fusion_power_video.h"  // from /root/reference via -I

extern "C" {

// Frame ctor + Frame::Predict through the reference's own Frame class.
// delta_raw == nullptr -> no delta frame (Frame::EMPTY).
int ref_predict(const uint16_t* img, size_t xs, size_t ys, int shift,
                int big_endian, const uint16_t* delta_raw, uint8_t* high,
                uint8_t* low, uint8_t* preview, size_t* low_size,
                size_t* preview_size) {
  fpvc::Frame frame(xs, ys, img, shift, big_endian != 0);
  if (delta_raw) {
    fpvc::Frame delta(xs, ys, delta_raw, shift, big_endian != 0);
    frame.Predict(delta);
  } else {
    frame.Predict();
  }
  memcpy(high, frame.high().data(), frame.high().size());
  memcpy(low, frame.low().data(), frame.low().size());
  memcpy(preview, frame.preview().data(), frame.preview().size());
  *low_size = frame.low().size();
  *preview_size = frame.preview().size();
  return frame.flags();
}

// Raw split planes only (Frame ctor, no prediction).
int ref_split(const uint16_t* img, size_t xs, size_t ys, int shift,
              int big_endian, uint8_t* high, uint8_t* low, size_t* low_size) {
  fpvc::Frame frame(xs, ys, img, shift, big_endian != 0);
  memcpy(high, frame.high().data(), frame.high().size());
  memcpy(low, frame.low().data(), frame.low().size());
  *low_size = frame.low().size();
  return frame.flags();
}

static void Append(const uint8_t* data, size_t size, void* payload) {
  std::vector<uint8_t>* v = static_cast<std::vector<uint8_t>*>(payload);
  v->insert(v->end(), data, data + size);
}

// Encoder::Init / CompressFrame / Finish with num_threads workers.
size_t ref_encode_stream(const uint16_t* frames, size_t nframes,
                         const uint16_t* delta_raw, size_t xs, size_t ys,
                         int shift, int big_endian, size_t num_threads,
                         uint8_t* out, size_t cap) {
  std::vector<uint8_t> bytes;
  fpvc::Encoder enc(num_threads, shift, big_endian != 0);
  enc.Init(delta_raw, xs, ys, Append, &bytes);
  for (size_t i = 0; i < nframes; i++) {
    // All input frames stay alive for the whole call, so MaxQueued() buffers
    // are trivially available.
    enc.CompressFrame(frames + i * xs * ys, Append, &bytes);
  }
  enc.Finish(Append, &bytes);
  if (bytes.size() > cap) return 0;
  memcpy(out, bytes.data(), bytes.size());
  return bytes.size();
}

// The reference Encoder timed as the upstream benchmark times it (benchmark.cc:153-180:
// construction -> Finish, every chunk copied away by the callback) over `total` frames
// cycling through `distinct` resident ones.  Returns the seconds.
double ref_bench_encode(const uint16_t* frames, size_t distinct, size_t total, size_t xs, size_t ys,
                        int shift, int big_endian, size_t num_threads, size_t* bytes_out) {
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
  }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (bytes_out) *bytes_out = out.bytes;
  return dt;
}

// RandomAccessDecoder over a whole stream; returns frame count or -1.
long ref_decode_stream(const uint8_t* data, size_t size, uint16_t* frames_out,
                       size_t cap_frames, uint8_t* previews_out, size_t* xs_out,
                       size_t* ys_out) {
  fpvc::RandomAccessDecoder dec;
  if (!dec.Init(data, size)) return -1;
  if (dec.numframes() > cap_frames) return -1;
  size_t npix = dec.xsize() * dec.ysize();
  size_t np = dec.preview_xsize() * dec.preview_ysize();
  for (size_t i = 0; i < dec.numframes(); i++) {
    if (!dec.DecodeFrame(i, frames_out + i * npix)) return -1;
    if (previews_out && !dec.DecodePreview(i, previews_out + i * np)) return -1;
  }
  *xs_out = dec.xsize();
  *ys_out = dec.ysize();
  return static_cast<long>(dec.numframes());
}

// StreamingDecoder fed in blocks of block bytes; returns frame count or -1.
long ref_stream_decode(const uint8_t* data, size_t size, size_t block,
                       uint16_t* frames_out, size_t cap_frames) {
  struct State {
    uint16_t* out;
    size_t cap, n;
    bool bad;
  } st = {frames_out, cap_frames, 0, false};
  fpvc::StreamingDecoder dec;
  for (size_t pos = 0; pos < size; pos += block) {
    size_t len = size - pos < block ? size - pos : block;
    dec.Decode(data + pos, len,
               [](bool ok, uint16_t* frame, size_t xs, size_t ys, void* p) {
                 State* s = static_cast<State*>(p);
                 if (!ok || s->n >= s->cap) { s->bad = true; return; }
                 memcpy(s->out + s->n * xs * ys, frame, xs * ys * 2);
                 s->n++;
               }, &st);
    if (st.bad) return -1;
  }
  return static_cast<long>(st.n);
}

void ref_unextract(const uint16_t* img, size_t xs, size_t ys, int shift,
                   int big_endian, uint8_t* out) {
  fpvc::UnextractFrame(img, xs, ys, shift, big_endian != 0, out);
}

// The Frame-API drivers: the same text the product compiles against its own
// header (fusion-power-video_amd/csrc/fpv_frame_capi.inc), here compiled
// against the UPSTREAM header and library.
#define FPV_FN(name) ref_##name
#include "fpv_frame_capi.inc"
#undef FPV_FN

}  // extern "C"
