// columnar_batch_decoder.h -- columnar batches in, images out (in order).  Counterpart
// of the upstream columnar_batch/columnar_batch_decoder.h (same public interface); a
// whole batch is inverted on the GPU in one launch.
#ifndef FPV_MI355X_COLUMNAR_BATCH_DECODER_H_
#define FPV_MI355X_COLUMNAR_BATCH_DECODER_H_

#include <future>
#include <memory>
#include <string>

#include "../fusion_power_video.h"
#include "columnar_batch.h"

namespace fpvc {
namespace columnarbatch {

class ColumnarBatchDecoder {
 public:
  // type: what to reconstruct from every frame; unshift: FULL images are shifted back
  // right by the schema's left shift.
  ColumnarBatchDecoder(Image::Type type, bool unshift, ImageProcessor image_processor);
  ~ColumnarBatchDecoder();

  // The future yields the batch once all of its images went through the image
  // processor.  An invalid future after Close or for a batch of another schema.
  std::future<BatchPtr> PushBatch(BatchPtr batch);
  // The future yields the latest timestamp handed to the image processor.
  std::shared_future<int64_t> Close();
  // Extension (not upstream): the decoder's first failure ("" while there is none).  The batch
  // that hit it gets it through its future (an exception), later PushBatch calls return an
  // invalid future; nothing aborts.
  std::string LastError() const;

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

}  // namespace columnarbatch
}  // namespace fpvc

#endif  // FPV_MI355X_COLUMNAR_BATCH_DECODER_H_
