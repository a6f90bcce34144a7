// TEST DRIVER for the native columnar batch adapter (include/columnar_batch/*.h):
// encoder -> batches -> four decoders (FULL, FULL unshifted, MSB8, PREVIEW) plus the
// single-image Batch::ExtractImage path, checked against values computed here from the
// input pixels.  Prints "columnar ok ..." and returns 0, or a message and non-zero.
// Linked against libfpv_amd.so (GPU) or the CPU test shim.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <memory>
#include <vector>

#include "columnar_batch_decoder.h"
#include "columnar_batch_encoder.h"

using namespace fpvc::columnarbatch;

static int g_bad = 0;
size_t& npix_dummy();
#define EXPECT(c, ...)                         \
  do {                                         \
    if (!(c)) {                                \
      if (g_bad++ < 10) {                      \
        fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
        fprintf(stderr, __VA_ARGS__);          \
        fputc('\n', stderr);                   \
      }                                        \
    }                                          \
  } while (0)

struct Case {
  size_t xs, ys, n;
  int shift, per_batch;
};

static int RunCase(const Case& c) {
  const size_t npix = c.xs * c.ys, pw = c.xs / 4, ph = c.ys / 4;
  std::vector<std::vector<uint16_t>> frames(c.n, std::vector<uint16_t>(npix));
  uint32_t s = 777 + (uint32_t)c.xs;
  const uint32_t vmask = (1u << (16 - c.shift)) - 1;
  for (size_t k = 0; k < c.n; k++)
    for (size_t i = 0; i < npix; i++) {
      s = s * 1664525u + 1013904223u;
      const uint32_t smooth = (uint32_t)((i % c.xs) * 13 + (i / c.xs) * 29 + k * 301);
      frames[k][i] = (uint16_t)((k % 3 == 2 ? (s >> 12) : smooth + ((s >> 28) & 3)) & vmask);
    }
  auto shifted = [&](size_t k, size_t i) { return (uint16_t)(frames[k][i] << c.shift); };

  size_t seen[4] = {0, 0, 0, 0}, nulls = 0, batches = 0;
  std::unique_ptr<ColumnarBatchEncoder> enc;
  auto check = [&](int which, Image img) {
    const size_t k = img.timestamp() - 1000;
    EXPECT(k == seen[which], "decoder %d: image %zu out of order (expected %zu)", which, k, seen[which]);
    seen[which]++;
    if (k >= c.n) return;
    if (which <= 1) {  // FULL / FULL unshifted
      EXPECT(img.xsize() == c.xs && img.ysize() == c.ys && img.bpp() == (size_t)(16 - c.shift), "geometry");
      for (size_t i = 0; i < npix; i++) {
        const uint16_t want = which == 0 ? shifted(k, i) : frames[k][i];
        if (img.data16()[i] != want) { EXPECT(false, "decoder %d frame %zu pixel %zu: %u != %u", which, k, i, img.data16()[i], want); break; }
      }
    } else if (which == 2) {  // MSB8
      EXPECT(img.bpp() == 8 && img.xsize() == c.xs, "msb8 geometry");
      for (size_t i = 0; i < npix; i++)
        if (img.data8()[i] != (uint8_t)(shifted(k, i) >> 8)) { EXPECT(false, "msb8 frame %zu pixel %zu", k, i); break; }
    } else {  // PREVIEW: 4x4 box mean of the raw high bytes, & 0xfe (cc:500-512)
      EXPECT(img.xsize() == pw && img.ysize() == ph && img.bpp() == 8, "preview geometry");
      for (size_t py = 0; py < ph; py++)
        for (size_t px = 0; px < pw; px++) {
          uint32_t sum = 0;
          for (int j = 0; j < 4; j++)
            for (int i = 0; i < 4; i++) sum += shifted(k, (4 * py + j) * c.xs + 4 * px + i) >> 8;
          if (img.data8()[py * pw + px] != (uint8_t)((sum / 16) & 0xfe)) { EXPECT(false, "preview frame %zu (%zu,%zu)", k, px, py); py = ph; break; }
        }
    }
  };
  ColumnarBatchDecoder d_full(Image::FULL, false, [&](Image i) { check(0, std::move(i)); });
  ColumnarBatchDecoder d_unsh(Image::FULL, true, [&](Image i) { check(1, std::move(i)); });
  ColumnarBatchDecoder d_msb(Image::MSB8, false, [&](Image i) { check(2, std::move(i)); });
  ColumnarBatchDecoder d_pv(Image::PREVIEW, false, [&](Image i) { check(3, std::move(i)); });
  int64_t last_ts = -1;
  enc.reset(new ColumnarBatchEncoder(c.xs, c.ys, c.shift, false, [&](BatchPtr b) {
    if (!b) { nulls++; return; }
    batches++;
    EXPECT(b->length() >= 1 && b->length() <= (size_t)c.per_batch, "batch length %zu", b->length());
    EXPECT(reinterpret_cast<uintptr_t>(b->stream(0, 1, &npix_dummy())) % 64 == 0, "high column not 64-byte aligned");
    last_ts = b->LatestTimestamp();
    for (ColumnarBatchDecoder* d : {&d_full, &d_unsh, &d_msb, &d_pv}) {
      std::future<BatchPtr> f = d->PushBatch(b);
      EXPECT(f.valid(), "PushBatch refused");
      if (f.valid()) EXPECT(f.get().get() == b.get(), "future yields another batch");
    }
    // the single-image path must agree with the batch path
    Image one = b->ExtractImage(b->length() - 1, Image::FULL);
    const size_t k = one.timestamp() - 1000;
    for (size_t i = 0; i < npix && k < c.n; i++)
      if (one.data16()[i] != shifted(k, i)) { EXPECT(false, "ExtractImage frame %zu pixel %zu", k, i); break; }
    Image pv = b->ExtractImage(0, Image::PREVIEW);
    EXPECT(pv.xsize() == pw && pv.bpp() == 8, "ExtractImage preview geometry");
    enc->ReturnProcessedBatch(b);
  }, c.per_batch));
  std::vector<uint16_t> buf(npix);
  for (size_t k = 0; k < c.n; k++) {
    memcpy(buf.data(), frames[k].data(), npix * 2);
    std::future<void*> f = enc->PushFrame(1000 + k, buf.data(), &buf);
    EXPECT(f.valid() && f.get() == &buf, "PushFrame future");
    memset(buf.data(), 0xEE, npix * 2);  // the buffer is ours again
  }
  const int64_t end = enc->Close().get();
  EXPECT(end == (c.n ? (int64_t)(1000 + c.n - 1) : -1), "closing timestamp %lld", (long long)end);
  EXPECT(!enc->PushFrame(1, buf.data(), nullptr).valid(), "PushFrame after Close must be refused");
  EXPECT(enc->Close().get() == end, "second Close");
  const size_t want_batches = (c.n + c.per_batch - 1) / c.per_batch;
  EXPECT(batches == want_batches, "%zu batches, expected %zu", batches, want_batches);
  EXPECT(nulls == (c.n % c.per_batch == 0 ? 1u : 0u), "%zu null batches", nulls);
  for (ColumnarBatchDecoder* d : {&d_full, &d_unsh, &d_msb, &d_pv})
    EXPECT(d->Close().get() == end, "decoder closing timestamp");
  for (int w = 0; w < 4; w++) EXPECT(seen[w] == c.n, "decoder %d saw %zu of %zu images", w, seen[w], c.n);
  enc.reset();
  return g_bad;
}

// helper: a size_t lvalue for calls that need an out parameter we do not look at
size_t& npix_dummy() { static size_t v; return v; }

// FPV_COLUMNAR_BENCH=1: throughput of the adapter at the BASELINE frame size
// (1280x800 12-bit, shift 4): frames -> batches, then batches -> FULL images.
static void Bench() {
  const size_t xs = 1280, ys = 800, npix = xs * ys, distinct = 16, n = 512;
  const int per_batch = 16;
  std::vector<std::vector<uint16_t>> src(distinct, std::vector<uint16_t>(npix));
  uint32_t s = 99;
  for (size_t k = 0; k < distinct; k++)
    for (size_t i = 0; i < npix; i++) {
      s = s * 1664525u + 1013904223u;
      src[k][i] = (uint16_t)(((i % xs) * 3 + (i / xs) * 2 + k * 37 + ((s >> 29) & 7)) & 0xfff);
    }
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  std::vector<BatchPtr> batches;
  size_t bytes = 0;
  double t0 = now();
  {
    ColumnarBatchEncoder enc(xs, ys, 4, false, [&](BatchPtr b) {
      if (!b) return;
      for (size_t i = 0; i < b->length(); i++)
        for (int c = 0; c < 3; c++) { size_t z; b->stream(i, c, &z); bytes += z; }
      batches.push_back(b);
    }, per_batch);
    for (size_t k = 0; k < n; k++) enc.PushFrame(k, src[k % distinct].data(), nullptr).get();
    enc.Close().get();
  }
  const double te = now() - t0;
  size_t images = 0;
  t0 = now();
  {
    ColumnarBatchDecoder dec(Image::FULL, true, [&](Image) { images++; });
    for (BatchPtr& b : batches) dec.PushBatch(b).get();
    dec.Close().get();
  }
  const double td = now() - t0;
  printf("columnar bench: %zu frames %zux%zu, %d per batch: encode %.1f fps (%.3f bpp), decode %.1f fps (%zu images)\n",
         n, xs, ys, per_batch, n / te, bytes * 8.0 / (n * npix), images / td, images);
}

int main() {
  if (getenv("FPV_COLUMNAR_BENCH")) {
    Bench();
    return 0;
  }
  const Case cases[] = {{64, 48, 23, 4, 5}, {64, 48, 10, 0, 5}, {320, 200, 7, 4, 3}, {100, 100, 3, 0, 2}, {64, 48, 1, 4, 4}};
  for (const Case& c : cases) {
    if (RunCase(c)) {
      fprintf(stderr, "columnar FAILED for %zux%zu n=%zu shift=%d per_batch=%d\n", c.xs, c.ys, c.n, c.shift, c.per_batch);
      return 1;
    }
    printf("columnar ok: %zux%zu, %zu frames, shift %d, %d per batch\n", c.xs, c.ys, c.n, c.shift, c.per_batch);
  }
  return 0;
}
