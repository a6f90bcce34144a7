// columnar_batch_encoder.h -- frames in, columnar batches out.  Counterpart of the
// upstream columnar_batch/columnar_batch_encoder.h (same public interface); here a
// whole batch goes through the GPU predictor in one launch and its planes are
// Brotli-compressed by a pool of host threads.
#ifndef FPV_MI355X_COLUMNAR_BATCH_ENCODER_H_
#define FPV_MI355X_COLUMNAR_BATCH_ENCODER_H_

#include <future>
#include <memory>
#include <string>

#include "../fusion_power_video.h"
#include "columnar_batch.h"

namespace fpvc {
namespace columnarbatch {

class ColumnarBatchEncoder {
 public:
  // batch_processor receives every full batch (and the last, partial one at Close; a
  // null pointer if there is nothing left to flush).  The first frame pushed is the
  // delta frame of the stream and is also encoded as an ordinary frame.
  ColumnarBatchEncoder(size_t xsize, size_t ysize, int shift_to_left_align, bool big_endian,
                       BatchProcessor batch_processor, int frames_per_batch = 10);
  ~ColumnarBatchEncoder();

  // The returned future yields `info` as soon as `frame` has been consumed and may be
  // reused by the caller.  An invalid future after Close.
  std::future<void*> PushFrame(uint64_t timestamp, uint16_t* frame, void* info);
  // Hands a batch back for reuse once the batch processor is done with it.
  void ReturnProcessedBatch(BatchPtr processed);
  // Flushes; the future yields the timestamp of the last frame stored (-1: none).
  std::shared_future<int64_t> Close();
  // Extension (not upstream): the encoder's first GPU failure ("" while there is none).  The
  // PushFrame call that hit it gets a future carrying it as an exception, later calls an invalid
  // future (as after Close); batches in flight are dropped; nothing aborts.
  std::string LastError() const;

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

}  // namespace columnarbatch
}  // namespace fpvc

#endif  // FPV_MI355X_COLUMNAR_BATCH_ENCODER_H_
