// arrow_decoder.h -- EXTENSION (the upstream arrow/ adapter only encodes): the way back
// from the record batches ArrowEncoder emits to images.  The delta frame comes from the
// schema metadata; every record batch is decoded as one GPU batch by the columnar batch
// decoder underneath.  Part of libfpv_arrow.so.
#ifndef FPV_MI355X_ARROW_DECODER_H_
#define FPV_MI355X_ARROW_DECODER_H_

#include <arrow/type_fwd.h>

#include <future>
#include <memory>

#include "../columnar_batch/columnar_batch_decoder.h"
#include "../fusion_power_video.h"

namespace fpvc {

class ArrowDecoder {
 public:
  // type / unshift / image_processor: as for columnarbatch::ColumnarBatchDecoder.
  ArrowDecoder(columnarbatch::Image::Type type, bool unshift, columnarbatch::ImageProcessor image_processor);
  ~ArrowDecoder();

  // Decodes the rows of one record batch, in order; returns when its images went through
  // the image processor.  false (and one stderr line) for a batch whose schema is not the
  // ArrowEncoder's, or whose metadata differs from the first batch's.
  bool PushRecordBatch(const std::shared_ptr<arrow::RecordBatch>& batch);
  // The future yields the latest timestamp handed to the image processor.
  std::shared_future<int64_t> Close();

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

}  // namespace fpvc

#endif  // FPV_MI355X_ARROW_DECODER_H_
