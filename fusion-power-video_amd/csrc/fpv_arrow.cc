// fpv_arrow.cc -- ArrowEncoder (include/fpv_arrow/arrow_encoder.h): the columnar batch
// encoder of fpv_columnar.cc with an Arrow record batch as the container.  What the
// upstream adapter does per frame with Frame::Predict / CompressPredicted from std::async
// tasks (arrow/arrow_encoder.cc:33-66, 100-118) happens here per batch on the GPU; this
// file only turns finished columns into Arrow arrays (arrow/arrow_encoder.cc:120-141) and
// builds the schema with the delta frame in its metadata (arrow/arrow_encoder.cc:80-98).
#include <arrow/api.h>
#include <arrow/util/key_value_metadata.h>

#include <mutex>
#include <string>

#include "../../include/columnar_batch/columnar_batch_decoder.h"
#include "../../include/columnar_batch/columnar_batch_encoder.h"
#include "../../include/fpv_arrow/arrow_decoder.h"
#include "../../include/fpv_arrow/arrow_encoder.h"

namespace fpvc {

using columnarbatch::BatchPtr;
using columnarbatch::ColumnarBatchEncoder;

struct ArrowEncoder::Impl {
  const size_t xs, ys;
  const int shift;
  const bool big_endian;
  RecordBatchConsumer consumer;
  std::mutex m;
  std::shared_ptr<arrow::Schema> schema;  // made from the first frame
  std::unique_ptr<ColumnarBatchEncoder> columns;
  // A record batch that could not be built (the GPU side latches in `columns`): kept, printed once.
  struct Latch {
    std::mutex m;
    std::string text_;
    bool failed() { std::lock_guard<std::mutex> l(m); return !text_.empty(); }
    std::string text() { std::lock_guard<std::mutex> l(m); return text_; }
    void Fail(const char* what, const char* detail) {
      if (getenv("FPV_ABORT_ON_ERROR")) { fprintf(stderr, "fpv-mi355x: fatal: ArrowEncoder: %s: %s\n", what, detail); abort(); }
      std::lock_guard<std::mutex> l(m);
      if (!text_.empty()) return;
      text_ = std::string(what) + ": " + detail;
      fprintf(stderr, "fpv-mi355x: ArrowEncoder failed: %s (the batch is dropped, later PushFrame calls are refused)\n", text_.c_str());
    }
  } latch;

  Impl(size_t x, size_t y, int s, bool be, RecordBatchConsumer c, int per_batch)
      : xs(x), ys(y), shift(s), big_endian(be), consumer(std::move(c)) {
    columns.reset(new ColumnarBatchEncoder(xs, ys, shift, big_endian, [this](BatchPtr b) { Emit(b); }, per_batch));
  }

  // arrow/arrow_encoder.cc:80-98: the delta frame goes through Frame::Compress (it may be
  // clamped-gradient predicted; the flag says so) and travels in the schema's metadata.
  void MakeSchema(const uint16_t* first_frame) {
    Frame df(xs, ys, first_frame, shift, big_endian);
    df.Compress();
    auto bytes = [](const std::vector<uint8_t>& v) { return std::string(reinterpret_cast<const char*>(v.data()), v.size()); };
    std::shared_ptr<arrow::Schema> s = arrow::schema(
        {arrow::field("timestamp", arrow::timestamp(arrow::TimeUnit::NANO), false),
         arrow::field("deltaPredicted", arrow::boolean(), false), arrow::field("cgPredicted", arrow::boolean(), false),
         arrow::field("preview", arrow::binary(), false), arrow::field("highBytePlane", arrow::binary(), false),
         arrow::field("lowBytePlane", arrow::binary(), false)},
        arrow::key_value_metadata(
            {"xsize", "ysize", "shiftedLeft", "deltaFrameHighPlane", "deltaFrameLowPlane", "deltaFrameCGPredicted"},
            {std::to_string(xs), std::to_string(ys), std::to_string(shift), bytes(df.high()), bytes(df.low()),
             (df.flags() & FrameFlags::USE_CG) ? "true" : "false"}));
    std::lock_guard<std::mutex> l(m);
    schema = s;
  }

  // One columnar batch -> one record batch (arrow/arrow_encoder.cc:120-141).
  void Emit(BatchPtr b) {
    if (!b) {
      consumer(nullptr);
      return;
    }
    const int64_t n = (int64_t)b->length();
    arrow::TimestampBuilder ts(arrow::timestamp(arrow::TimeUnit::NANO), arrow::default_memory_pool());
    arrow::BooleanBuilder delta, cg;
    arrow::BinaryBuilder col[3];  // preview, high, low: the Batch's column order
    bool ok = ts.Reserve(n).ok() && delta.Reserve(n).ok() && cg.Reserve(n).ok();
    for (int c = 0; c < 3 && ok; c++) {
      size_t total = 0, z;
      for (int64_t i = 0; i < n; i++) { b->stream((size_t)i, c, &z); total += z; }
      ok = col[c].Reserve(n).ok() && col[c].ReserveData((int64_t)total).ok();
    }
    for (int64_t i = 0; i < n && ok; i++) {
      ts.UnsafeAppend(b->timestamp((size_t)i));
      delta.UnsafeAppend((b->flags((size_t)i) & FrameFlags::USE_DELTA) != 0);
      cg.UnsafeAppend((b->flags((size_t)i) & FrameFlags::USE_CG) != 0);
      for (int c = 0; c < 3; c++) {
        size_t z;
        const uint8_t* p = b->stream((size_t)i, c, &z);
        col[c].UnsafeAppend(p, (int32_t)z);
      }
    }
    std::shared_ptr<arrow::Array> a[6];
    ok = ok && ts.Finish(&a[0]).ok() && delta.Finish(&a[1]).ok() && cg.Finish(&a[2]).ok() &&
         col[0].Finish(&a[3]).ok() && col[1].Finish(&a[4]).ok() && col[2].Finish(&a[5]).ok();
    std::shared_ptr<arrow::Schema> s;
    {
      std::lock_guard<std::mutex> l(m);
      s = schema;
    }
    columns->ReturnProcessedBatch(b);  // the columns have been copied into Arrow buffers
    if (!ok || !s) {  // latched (ArrowEncoder::LastError), the batch is dropped, later PushFrame calls are refused
      latch.Fail("building a record batch failed", !s ? "no schema" : "Arrow builder out of memory");
      return;
    }
    consumer(arrow::RecordBatch::Make(s, n, {a[0], a[1], a[2], a[3], a[4], a[5]}));
  }
};

ArrowEncoder::ArrowEncoder(size_t xsize, size_t ysize, int shift_to_left_align, bool big_endian,
                           RecordBatchConsumer record_batch_consumer, int frames_per_batch)
    : impl_(new Impl(xsize, ysize, shift_to_left_align, big_endian, std::move(record_batch_consumer), frames_per_batch)) {}

ArrowEncoder::~ArrowEncoder() {
  Close();
  impl_->columns.reset();  // joins the workers: no Emit after this
  impl_.reset();
}

std::string ArrowEncoder::LastError() const {
  const std::string own = impl_->latch.text();
  return own.empty() ? impl_->columns->LastError() : own;
}

std::future<void*> ArrowEncoder::PushFrame(uint64_t timestamp, uint16_t* frame, void* info) {
  if (impl_->latch.failed()) return std::future<void*>();
  bool first;
  {
    std::lock_guard<std::mutex> l(impl_->m);
    first = !impl_->schema;
  }
  if (first) impl_->MakeSchema(frame);  // before the frame can reach a batch
  return impl_->columns->PushFrame(timestamp, frame, info);
}

std::shared_future<int64_t> ArrowEncoder::Close() { return impl_->columns->Close(); }

// ============================================================ ArrowDecoder ==

struct ArrowDecoder::Impl {
  columnarbatch::ColumnarBatchDecoder columns;
  columnarbatch::SchemaPtr schema;          // built from the first batch's metadata
  std::string dh, dl;                       // the delta frame streams it was built from
  size_t xs = 0, ys = 0, shift = 0;

  Impl(columnarbatch::Image::Type t, bool u, columnarbatch::ImageProcessor p) : columns(t, u, std::move(p)) {}

  static bool Fail(const char* what) {
    fprintf(stderr, "fpv-mi355x: ArrowDecoder: %s\n", what);
    return false;
  }

  bool Prepare(const arrow::Schema& sc) {
    static const char* kNames[6] = {"timestamp", "deltaPredicted", "cgPredicted", "preview", "highBytePlane", "lowBytePlane"};
    if (sc.num_fields() != 6) return Fail("not an ArrowEncoder schema (field count)");
    for (int i = 0; i < 6; i++)
      if (sc.field(i)->name() != kNames[i]) return Fail("not an ArrowEncoder schema (field names)");
    if (sc.field(0)->type()->id() != arrow::Type::TIMESTAMP || sc.field(1)->type()->id() != arrow::Type::BOOL ||
        sc.field(2)->type()->id() != arrow::Type::BOOL || sc.field(3)->type()->id() != arrow::Type::BINARY ||
        sc.field(4)->type()->id() != arrow::Type::BINARY || sc.field(5)->type()->id() != arrow::Type::BINARY)
      return Fail("not an ArrowEncoder schema (field types)");
    auto md = sc.metadata();
    if (!md) return Fail("schema metadata missing");
    const std::string h = md->Get("deltaFrameHighPlane").ValueOr(""), l = md->Get("deltaFrameLowPlane").ValueOr("");
    const size_t x = strtoul(md->Get("xsize").ValueOr("0").c_str(), nullptr, 10);
    const size_t y = strtoul(md->Get("ysize").ValueOr("0").c_str(), nullptr, 10);
    const size_t sh = strtoul(md->Get("shiftedLeft").ValueOr("0").c_str(), nullptr, 10);
    if (schema) {
      if (x != xs || y != ys || sh != shift || h != dh || l != dl) return Fail("record batch of another stream");
      return true;
    }
    if (!x || !y || sh > 15 || h.empty()) return Fail("bad geometry / delta frame in the schema metadata");
    const bool cg = md->Get("deltaFrameCGPredicted").ValueOr("") == "true";
    // the delta frame as ArrowEncoder stored it: Frame::Compress()ed without a delta frame
    Frame delta(x, y, (uint8_t)((cg ? FrameFlags::USE_CG : 0) | (l.empty() ? FrameFlags::NO_LOW_BYTES : 0)),
                (uint8_t)(FrameState::COMPRESSED | FrameState::CG_PREDICTED | FrameState::DELTA_PREDICTED),
                std::vector<uint8_t>(h.begin(), h.end()), std::vector<uint8_t>(l.begin(), l.end()), std::vector<uint8_t>());
    delta.Uncompress();
    schema = std::make_shared<columnarbatch::BatchSchema>(x, y, sh, delta);
    xs = x; ys = y; shift = sh; dh = h; dl = l;
    return true;
  }
};

ArrowDecoder::ArrowDecoder(columnarbatch::Image::Type type, bool unshift, columnarbatch::ImageProcessor image_processor)
    : impl_(new Impl(type, unshift, std::move(image_processor))) {}

ArrowDecoder::~ArrowDecoder() { Close(); }

bool ArrowDecoder::PushRecordBatch(const std::shared_ptr<arrow::RecordBatch>& rb) {
  if (!rb) return Impl::Fail("null record batch");
  if (!impl_->Prepare(*rb->schema())) return false;
  const int64_t rows = rb->num_rows();
  if (rows == 0) return true;
  auto ts = std::static_pointer_cast<arrow::TimestampArray>(rb->column(0));
  auto fd = std::static_pointer_cast<arrow::BooleanArray>(rb->column(1));
  auto fc = std::static_pointer_cast<arrow::BooleanArray>(rb->column(2));
  auto pv = std::static_pointer_cast<arrow::BinaryArray>(rb->column(3));
  auto hi = std::static_pointer_cast<arrow::BinaryArray>(rb->column(4));
  auto lo = std::static_pointer_cast<arrow::BinaryArray>(rb->column(5));
  columnarbatch::BatchPtr b = std::make_shared<columnarbatch::Batch>((size_t)rows, impl_->schema);
  for (int64_t i = 0; i < rows; i++) {
    int32_t hn, ln, pn;
    const uint8_t *h = hi->GetValue(i, &hn), *l = lo->GetValue(i, &ln), *p = pv->GetValue(i, &pn);
    const uint8_t flags = (uint8_t)((fd->Value(i) ? FrameFlags::USE_DELTA : 0) | (fc->Value(i) ? FrameFlags::USE_CG : 0) |
                                    (ln == 0 ? FrameFlags::NO_LOW_BYTES : 0));  // an empty low stream = no low plane
    if (!b->AppendCompressed(ts->Value(i), flags, h, (size_t)hn, l, (size_t)ln, p, (size_t)pn))
      return Impl::Fail("a row does not fit the geometry of the schema");
  }
  std::future<columnarbatch::BatchPtr> done = impl_->columns.PushBatch(b);
  if (!done.valid()) return Impl::Fail("decoder already closed or failed (ColumnarBatchDecoder::LastError)");
  try {
    done.get();
  } catch (const std::exception& e) {  // the columnar decoder delivers its failure through the future
    return Impl::Fail(e.what());
  }
  return true;
}

std::shared_future<int64_t> ArrowDecoder::Close() { return impl_->columns.Close(); }

}  // namespace fpvc
