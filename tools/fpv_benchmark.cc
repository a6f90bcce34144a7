// fpv_benchmark -- end-to-end timing of the host pipeline on a raw file:
// Encoder construction -> Finish (like the upstream benchmark's timer), then a
// round trip through both decoders, byte-compared with the input.
//   fpv_benchmark file.raw xsize ysize big_endian shift [maxframes] [threads]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <vector>

#include "fusion_power_video.h"

static double Now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Collect {
  std::vector<uint8_t> bytes;
};
static void Append(const uint8_t* d, size_t n, void* p) {
  Collect* c = static_cast<Collect*>(p);
  c->bytes.insert(c->bytes.end(), d, d + n);
}

int main(int argc, char** argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: %s file xsize ysize big_endian shift [maxframes] [threads]\n", argv[0]);
    return 2;
  }
  const size_t xs = strtoul(argv[2], nullptr, 10), ys = strtoul(argv[3], nullptr, 10);
  const bool be = atoi(argv[4]) != 0;
  const int shift = atoi(argv[5]);
  size_t maxframes = argc > 6 ? strtoul(argv[6], nullptr, 10) : 0;
  const size_t threads = argc > 7 ? strtoul(argv[7], nullptr, 10) : 8;
  const size_t npix = xs * ys, fbytes = npix * 2;

  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  std::vector<uint8_t> raw;
  std::vector<uint8_t> buf(1 << 22);
  size_t got;
  while ((got = fread(buf.data(), 1, buf.size(), f)) > 0) {
    raw.insert(raw.end(), buf.data(), buf.data() + got);
    if (maxframes && raw.size() >= maxframes * fbytes) break;
  }
  fclose(f);
  size_t n = raw.size() / fbytes;
  if (maxframes && n > maxframes) n = maxframes;
  if (n == 0) { fprintf(stderr, "no complete frame in %s\n", argv[1]); return 1; }
  const uint16_t* px = reinterpret_cast<const uint16_t*>(raw.data());

  if (getenv("FPV_BENCH_NOCOPY")) {  // pipeline limit: the callback only counts bytes
    size_t total = 0;
    const double t0 = Now();
    {
      fpvc::Encoder enc(threads, shift, be);
      auto count = [](const uint8_t*, size_t sz, void* p) { *static_cast<size_t*>(p) += sz; };
      enc.Init(px, xs, ys, count, &total);
      for (size_t i = 0; i < n; i++) enc.CompressFrame(px + i * npix, count, &total);
      enc.Finish(count, &total);
    }
    const double dt = Now() - t0;
    fprintf(stderr, "encode (count-only callback): %zu frames in %.3f s = %.1f fps, %zu bytes, %zu threads\n",
            n, dt, n / dt, total, threads);
  }
  Collect out;
  out.bytes.reserve(n * fbytes * 3 / 5);
  for (int rep = 0; rep < 2; rep++) {  // rep 0 warms contexts / pinned memory
    out.bytes.clear();
    const double t0 = Now();
    {
      fpvc::Encoder enc(threads, shift, be);
      enc.Init(px, xs, ys, Append, &out);
      for (size_t i = 0; i < n; i++) enc.CompressFrame(px + i * npix, Append, &out);
      enc.Finish(Append, &out);
    }
    const double dt = Now() - t0;
    fprintf(stderr, "%s: %zu frames %zux%zu in %.3f s = %.1f fps, %.1f MP/s, %.3f bpp, %zu threads\n",
            rep ? "encode" : "encode (cold)", n, xs, ys, dt, n / dt, n * npix / dt / 1e6,
            out.bytes.size() * 8.0 / (n * npix), threads);
  }

  std::vector<uint8_t> back(fbytes);
  {  // streaming decoder, 64 KiB pieces
    struct St { const uint8_t* raw; size_t fbytes, n; int shift; bool be, bad; std::vector<uint8_t>* back; };
    St st = {raw.data(), fbytes, 0, shift, be, false, &back};
    fpvc::StreamingDecoder dec;
    const double t0 = Now();
    for (size_t pos = 0; pos < out.bytes.size(); pos += 65536) {
      size_t len = std::min<size_t>(65536, out.bytes.size() - pos);
      dec.Decode(out.bytes.data() + pos, len,
                 [](bool ok, uint16_t* frame, size_t xs, size_t ys, void* p) {
                   St* s = static_cast<St*>(p);
                   if (!ok) { s->bad = true; return; }
                   fpvc::UnextractFrame(frame, xs, ys, s->shift, s->be, s->back->data());
                   if (memcmp(s->back->data(), s->raw + s->n * s->fbytes, s->fbytes)) s->bad = true;
                   s->n++;
                 }, &st);
    }
    fprintf(stderr, "streaming decode: %zu frames, %s, %.1f fps\n", st.n,
            (!st.bad && st.n == n) ? "ok" : "MISMATCH", st.n / (Now() - t0));
    if (st.bad || st.n != n) return 1;
  }
  {  // random access decoder: single-frame API, then the batch extension
    fpvc::RandomAccessDecoder dec;
    if (!dec.Init(out.bytes.data(), out.bytes.size()) || dec.numframes() != n) {
      fprintf(stderr, "RandomAccessDecoder::Init failed\n");
      return 1;
    }
    std::vector<uint16_t> img(npix);
    std::vector<uint8_t> pv(dec.preview_xsize() * dec.preview_ysize());
    double t0 = Now();
    for (size_t i = 0; i < n; i++) {
      if (!dec.DecodeFrame(i, img.data()) || !dec.DecodePreview(i, pv.data())) return 1;
      fpvc::UnextractFrame(img.data(), xs, ys, shift, be, back.data());
      if (memcmp(back.data(), raw.data() + i * fbytes, fbytes)) {
        fprintf(stderr, "random access: frame %zu MISMATCH\n", i);
        return 1;
      }
    }
    fprintf(stderr, "random access decode: ok, %.1f fps (one frame per call)\n", n / (Now() - t0));
    std::vector<uint16_t> all(n * npix);
    std::vector<size_t> idx(n);
    std::vector<uint16_t*> outs(n);
    for (size_t i = 0; i < n; i++) { idx[i] = i; outs[i] = all.data() + i * npix; }
    t0 = Now();
    if (!dec.DecodeFrames(idx.data(), n, outs.data(), threads ? threads : 1)) return 1;
    const double dt = Now() - t0;
    for (size_t i = 0; i < n; i++) {
      fpvc::UnextractFrame(outs[i], xs, ys, shift, be, back.data());
      if (memcmp(back.data(), raw.data() + i * fbytes, fbytes)) {
        fprintf(stderr, "batched decode: frame %zu MISMATCH\n", i);
        return 1;
      }
    }
    fprintf(stderr, "batched random access decode: ok, %.1f fps\n", n / dt);
  }
  return 0;
}
