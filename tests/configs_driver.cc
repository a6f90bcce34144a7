// configs_driver.cc -- TEST DRIVER for the BASELINE configurations at their stated frame
// counts (tests/test_gpu_configs.py runs it on the GPU box).  `distinct` frames are read
// from a raw file and cycled to `nframes`, as SURVEY 8(d) prescribes for long streams.
//
//   configs_driver encode  file xs ys shift distinct nframes threads   stream to stdout (config 1: SHA-256 vs the reference)
//   configs_driver stream  file xs ys shift distinct nframes threads   Encoder -> StreamingDecoder, bounded memory (config 2)
//   configs_driver archive file xs ys shift distinct nframes threads   archive -> DecodeFrames with a shuffled index list (config 4)
//
// Every mode prints one line starting with "ok" on success; anything else is a failure.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "fusion_power_video.h"

static double Now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static uint64_t Load64(const uint8_t* p) {
  uint64_t v = 0;
  for (int i = 7; i >= 0; i--) v = (v << 8) | p[i];
  return v;
}

int main(int argc, char** argv) {
  if (argc < 9) {
    fprintf(stderr, "usage: %s encode|stream|archive file xs ys shift distinct nframes threads\n", argv[0]);
    return 2;
  }
  const std::string mode = argv[1];
  const size_t xs = strtoul(argv[3], nullptr, 10), ys = strtoul(argv[4], nullptr, 10);
  const int shift = atoi(argv[5]);
  const size_t distinct = strtoul(argv[6], nullptr, 10), n = strtoul(argv[7], nullptr, 10);
  const size_t threads = strtoul(argv[8], nullptr, 10);
  const size_t npix = xs * ys;
  std::vector<uint16_t> frames(distinct * npix);
  FILE* f = fopen(argv[2], "rb");
  if (!f || fread(frames.data(), 2, frames.size(), f) != frames.size()) {
    fprintf(stderr, "cannot read %zu frames from %s\n", distinct, argv[2]);
    return 1;
  }
  fclose(f);
  auto frame = [&](size_t i) { return frames.data() + (i % distinct) * npix; };

  if (mode == "encode") {
    auto write = [](const uint8_t* d, size_t sz, void*) { fwrite(d, 1, sz, stdout); };
    fpvc::Encoder enc(threads, shift, false);
    enc.Init(frame(0), xs, ys, write, nullptr);
    for (size_t i = 0; i < n; i++) enc.CompressFrame(frame(i), write, nullptr);
    enc.Finish(write, nullptr);
    fflush(stdout);
    fprintf(stderr, "ok encode %zu frames\n", n);
    return 0;
  }

  if (mode == "stream") {
    // The encoder's callbacks hand the stream to a StreamingDecoder in pieces of up to 64 MB
    // (several frame chunks per Decode call: its GPU-batch path); decoded frames are compared
    // with the input on the spot.  Nothing but the pieces in flight is ever held.
    struct St {
      fpvc::StreamingDecoder dec;
      std::vector<uint8_t> pending;
      const std::vector<uint16_t>* frames;
      size_t npix, distinct, decoded, mismatches, bytes, calls;
      int shift;
      std::vector<uint64_t> chunk_offsets;  // offset of every frame chunk from the start of the stream
      std::vector<uint8_t> footer;
      void flush() {
        if (pending.empty()) return;
        dec.Decode(pending.data(), pending.size(),
                   [](bool ok, uint16_t* px, size_t, size_t, void* p) {
                     St* s = static_cast<St*>(p);
                     if (!ok) { s->mismatches++; return; }
                     const uint16_t* want = s->frames->data() + (s->decoded % s->distinct) * s->npix;
                     for (size_t k = 0; k < s->npix; k++)
                       if (px[k] != (uint16_t)(want[k] << s->shift)) { s->mismatches++; break; }
                     s->decoded++;
                   }, this);
        pending.clear();
      }
    } st;
    st.frames = &frames; st.npix = npix; st.distinct = distinct; st.decoded = st.mismatches = st.bytes = st.calls = 0; st.shift = shift;
    auto sink = [](const uint8_t* d, size_t sz, void* p) {
      St* s = static_cast<St*>(p);
      // call 0 is the header + delta chunk (Init), then one frame chunk per call, then the index
      if (s->calls > 0 && sz >= 5 && d[4] == 0) s->chunk_offsets.push_back(s->bytes);
      if (s->calls > 0 && sz >= 5 && d[4] == 2) s->footer.assign(d, d + sz);
      s->calls++;
      s->bytes += sz;
      s->pending.insert(s->pending.end(), d, d + sz);
      if (s->pending.size() >= ((size_t)64 << 20)) s->flush();
    };
    const double t0 = Now();
    {
      fpvc::Encoder enc(threads, shift, false);
      enc.Init(frame(0), xs, ys, sink, &st);
      for (size_t i = 0; i < n; i++) enc.CompressFrame(frame(i), sink, &st);
      enc.Finish(sink, &st);
    }
    st.flush();
    const double dt = Now() - t0;
    if (st.decoded != n || st.mismatches) { printf("FAIL stream: %zu of %zu decoded, %zu mismatches\n", st.decoded, n, st.mismatches); return 1; }
    // the frame index (cc:1185-1197): n 64-bit offsets that must equal where the chunks went
    if (st.footer.size() != 5 + 8 * n + 8 || Load64(st.footer.data() + 5 + 8 * n) != n) { printf("FAIL footer size\n"); return 1; }
    for (size_t i = 0; i < n; i++)
      if (Load64(st.footer.data() + 5 + 8 * i) != st.chunk_offsets[i]) { printf("FAIL offset %zu\n", i); return 1; }
    printf("ok stream %zu frames %zux%zu, %.2f GB, last index offset %llu (%s 2^32), %.1f s, %.0f fps round trip\n", n, xs, ys,
           st.bytes / 1e9, (unsigned long long)st.chunk_offsets.back(), st.chunk_offsets.back() > 0xffffffffull ? ">" : "<=", dt, n / dt);
    return 0;
  }

  if (mode == "archive") {
    std::vector<uint8_t> arc;
    auto app = [](const uint8_t* d, size_t sz, void* p) {
      auto* v = static_cast<std::vector<uint8_t>*>(p);
      v->insert(v->end(), d, d + sz);
    };
    double t0 = Now();
    {
      fpvc::Encoder enc(threads, shift, false);
      enc.Init(frame(0), xs, ys, app, &arc);
      for (size_t i = 0; i < n; i++) enc.CompressFrame(frame(i), app, &arc);
      enc.Finish(app, &arc);
    }
    const double t_enc = Now() - t0;
    fpvc::RandomAccessDecoder dec;
    if (!dec.Init(arc.data(), arc.size()) || dec.numframes() != n) { printf("FAIL archive init\n"); return 1; }
    std::vector<size_t> idx(n);
    for (size_t i = 0; i < n; i++) idx[i] = i;
    uint32_t s = 20241003u;  // fixed shuffle
    for (size_t i = n - 1; i > 0; i--) { s = s * 1664525u + 1013904223u; std::swap(idx[i], idx[(s >> 8) % (i + 1)]); }
    const size_t group = 256;
    std::vector<uint16_t> buf(group * npix);
    std::vector<uint16_t*> outs(group);
    for (size_t k = 0; k < group; k++) outs[k] = buf.data() + k * npix;
    t0 = Now();
    for (size_t first = 0; first < n; first += group) {
      const size_t cnt = std::min(group, n - first);
      if (!dec.DecodeFrames(idx.data() + first, cnt, outs.data(), threads)) { printf("FAIL DecodeFrames\n"); return 1; }
      for (size_t k = 0; k < cnt; k++) {
        const uint16_t* want = frame(idx[first + k]);
        for (size_t q = 0; q < npix; q++)
          if (outs[k][q] != (uint16_t)(want[q] << shift)) { printf("FAIL frame %zu differs\n", idx[first + k]); return 1; }
      }
    }
    const double t_dec = Now() - t0;
    printf("ok archive %zu frames %zux%zu, %.2f GB, encode %.0f fps, shuffled DecodeFrames %.0f fps\n", n, xs, ys, arc.size() / 1e9,
           n / t_enc, n / t_dec);
    return 0;
  }
  fprintf(stderr, "unknown mode %s\n", mode.c_str());
  return 2;
}
