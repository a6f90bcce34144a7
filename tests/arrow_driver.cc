// TEST DRIVER for the native ArrowEncoder (include/fpv_arrow/arrow_encoder.h): frames ->
// Arrow record batches; checks the schema and its metadata, then rebuilds columnar
// batches from the record batches' arrays (delta frame from the schema metadata) and
// decodes them with the ColumnarBatchDecoder: every pixel must come back.
// Prints "arrow ok ..." and returns 0, or a message and non-zero.
#include <arrow/api.h>
#include <arrow/io/file.h>
#include <arrow/ipc/writer.h>
#include <arrow/util/key_value_metadata.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <memory>
#include <string>
#include <vector>

#include "../include/columnar_batch/columnar_batch_decoder.h"
#include "../include/fpv_arrow/arrow_decoder.h"
#include "../include/fpv_arrow/arrow_encoder.h"

using namespace fpvc;
using namespace fpvc::columnarbatch;

#define CHECK(c, ...)                                    \
  do {                                                   \
    if (!(c)) {                                          \
      fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
      fprintf(stderr, __VA_ARGS__);                      \
      fputc('\n', stderr);                               \
      return 1;                                          \
    }                                                    \
  } while (0)

static int RunCase(size_t xs, size_t ys, size_t n, int shift, int per_batch) {
  const size_t npix = xs * ys;
  std::vector<std::vector<uint16_t>> frames(n, std::vector<uint16_t>(npix));
  uint32_t s = 4242 + (uint32_t)xs;
  const uint32_t vmask = (1u << (16 - shift)) - 1;
  for (size_t k = 0; k < n; k++)
    for (size_t i = 0; i < npix; i++) {
      s = s * 1664525u + 1013904223u;
      frames[k][i] = (uint16_t)(((i % xs) * 11 + (i / xs) * 23 + k * 257 + ((s >> 28) & 3)) & vmask);
    }
  std::vector<std::shared_ptr<arrow::RecordBatch>> batches;
  size_t nulls = 0;
  int64_t end;
  {
    ArrowEncoder enc(xs, ys, shift, false, [&](std::shared_ptr<arrow::RecordBatch> b) {
      if (b) batches.push_back(b); else nulls++;
    }, per_batch);
    std::vector<uint16_t> buf(npix);
    for (size_t k = 0; k < n; k++) {
      memcpy(buf.data(), frames[k].data(), npix * 2);
      std::future<void*> f = enc.PushFrame(5000 + k, buf.data(), &buf);
      CHECK(f.valid() && f.get() == &buf, "PushFrame future");
      memset(buf.data(), 0x77, npix * 2);
    }
    end = enc.Close().get();
  }
  CHECK(end == (int64_t)(5000 + n - 1), "closing timestamp %lld", (long long)end);
  CHECK(batches.size() == (n + per_batch - 1) / per_batch, "%zu record batches", batches.size());
  CHECK(nulls == (n % per_batch == 0 ? 1u : 0u), "%zu null batches", nulls);

  // schema and metadata (arrow/arrow_encoder.cc:80-98)
  std::shared_ptr<arrow::Schema> sc = batches[0]->schema();
  const char* names[6] = {"timestamp", "deltaPredicted", "cgPredicted", "preview", "highBytePlane", "lowBytePlane"};
  CHECK(sc->num_fields() == 6, "field count");
  for (int i = 0; i < 6; i++) CHECK(sc->field(i)->name() == names[i] && !sc->field(i)->nullable(), "field %d", i);
  CHECK(sc->field(0)->type()->Equals(arrow::timestamp(arrow::TimeUnit::NANO)), "timestamp type");
  CHECK(sc->field(1)->type()->id() == arrow::Type::BOOL && sc->field(4)->type()->id() == arrow::Type::BINARY, "types");
  auto md = sc->metadata();
  CHECK(md && md->Get("xsize").ValueOr("") == std::to_string(xs) && md->Get("ysize").ValueOr("") == std::to_string(ys) &&
            md->Get("shiftedLeft").ValueOr("") == std::to_string(shift), "geometry metadata");
  const std::string dh = md->Get("deltaFrameHighPlane").ValueOr(""), dl = md->Get("deltaFrameLowPlane").ValueOr("");
  const bool dcg = md->Get("deltaFrameCGPredicted").ValueOr("") == "true";
  CHECK(!dh.empty(), "delta frame stream missing");

  // the delta frame back from the metadata, then a columnar schema around it
  Frame delta(xs, ys, (uint8_t)((dcg ? FrameFlags::USE_CG : 0) | (dl.empty() ? FrameFlags::NO_LOW_BYTES : 0)),
              (uint8_t)(FrameState::COMPRESSED | FrameState::CG_PREDICTED | FrameState::DELTA_PREDICTED),
              std::vector<uint8_t>(dh.begin(), dh.end()), std::vector<uint8_t>(dl.begin(), dl.end()),
              std::vector<uint8_t>());
  delta.Uncompress();
  for (size_t i = 0; i < npix; i++) {
    const uint16_t want = (uint16_t)(frames[0][i] << shift);
    const uint16_t got = (uint16_t)((delta.high()[i] << 8) | (i < delta.low().size() ? delta.low()[i] : 0));
    CHECK(got == want, "delta frame from the metadata differs at pixel %zu", i);
  }
  // all the way back through the library's own ArrowDecoder (an extension): every pixel
  size_t seen = 0;
  int bad = 0;
  {
    ArrowDecoder dec(Image::FULL, true, [&](Image img) {
      const size_t k = img.timestamp() - 5000;
      if (k != seen++ || k >= n) { bad++; return; }
      for (size_t i = 0; i < npix; i++)
        if (img.data16()[i] != frames[k][i]) { bad++; return; }
    });
    for (auto& rb : batches) CHECK(dec.PushRecordBatch(rb), "PushRecordBatch");
    CHECK(dec.Close().get() == end, "decoder closing timestamp");
    // a record batch of some other schema is refused, not misread
    auto other = arrow::RecordBatch::Make(arrow::schema({arrow::field("x", arrow::int64())}), 0,
                                          {std::make_shared<arrow::Int64Array>(0, nullptr)});
    CHECK(!dec.PushRecordBatch(other), "foreign schema accepted");
  }
  CHECK(seen == n && bad == 0, "%zu of %zu images, %d bad", seen, n, bad);
  // FPV_ARROW_IPC_OUT=<path>: the record batches of this case as an Arrow IPC file, for a
  // consumer that is not this code base (tests/test_arrow.py reads it with pyarrow)
  if (const char* path = getenv("FPV_ARROW_IPC_OUT")) {
    auto sink = arrow::io::FileOutputStream::Open(path);
    CHECK(sink.ok(), "cannot open %s", path);
    auto writer = arrow::ipc::MakeFileWriter(*sink, sc);
    CHECK(writer.ok(), "MakeFileWriter");
    for (auto& rb : batches) CHECK((*writer)->WriteRecordBatch(*rb).ok(), "WriteRecordBatch");
    CHECK((*writer)->Close().ok() && (*sink)->Close().ok(), "closing the IPC file");
    unsetenv("FPV_ARROW_IPC_OUT");  // first case only
  }
  printf("arrow ok: %zux%zu, %zu frames, shift %d, %d per batch, %zu record batches\n", xs, ys, n, shift, per_batch,
         batches.size());
  return 0;
}

// FPV_ARROW_BENCH=1: throughput at the BASELINE frame size (1280x800 12-bit, shift 4).
static void Bench() {
  const size_t xs = 1280, ys = 800, npix = xs * ys, distinct = 16, n = 512;
  std::vector<std::vector<uint16_t>> src(distinct, std::vector<uint16_t>(npix));
  uint32_t s = 99;
  for (size_t k = 0; k < distinct; k++)
    for (size_t i = 0; i < npix; i++) {
      s = s * 1664525u + 1013904223u;
      src[k][i] = (uint16_t)(((i % xs) * 3 + (i / xs) * 2 + k * 37 + ((s >> 29) & 7)) & 0xfff);
    }
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  size_t rows = 0, bytes = 0;
  const double t0 = now();
  {
    ArrowEncoder enc(xs, ys, 4, false, [&](std::shared_ptr<arrow::RecordBatch> b) {
      if (!b) return;
      rows += (size_t)b->num_rows();
      for (int c = 3; c < 6; c++) bytes += (size_t)std::static_pointer_cast<arrow::BinaryArray>(b->column(c))->total_values_length();
    }, 16);
    for (size_t k = 0; k < n; k++) enc.PushFrame(k, src[k % distinct].data(), nullptr).get();
    enc.Close().get();
  }
  const double dt = now() - t0;
  printf("arrow bench: %zu frames %zux%zu, 16 per batch: %.1f fps into record batches (%zu rows, %.3f bpp)\n", n, xs, ys,
         n / dt, rows, bytes * 8.0 / (n * npix));
}

int main() {
  if (getenv("FPV_ARROW_BENCH")) {
    Bench();
    return 0;
  }
  if (RunCase(64, 48, 11, 4, 4)) return 1;
  if (RunCase(320, 200, 6, 0, 3)) return 1;
  if (RunCase(100, 100, 3, 0, 2)) return 1;
  return 0;
}
