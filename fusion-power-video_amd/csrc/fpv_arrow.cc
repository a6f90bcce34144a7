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

#include "../../include/columnar_batch/columnar_batch_encoder.h"
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
    if (!ok || !s) {
      fprintf(stderr, "fpv-mi355x: ArrowEncoder: building a record batch failed\n");
      abort();  // like the other encoders: no error channel, never drop data silently
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

std::future<void*> ArrowEncoder::PushFrame(uint64_t timestamp, uint16_t* frame, void* info) {
  bool first;
  {
    std::lock_guard<std::mutex> l(impl_->m);
    first = !impl_->schema;
  }
  if (first) impl_->MakeSchema(frame);  // before the frame can reach a batch
  return impl_->columns->PushFrame(timestamp, frame, info);
}

std::shared_future<int64_t> ArrowEncoder::Close() { return impl_->columns->Close(); }

}  // namespace fpvc
