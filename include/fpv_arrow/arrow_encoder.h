// arrow_encoder.h -- frames in, Apache Arrow record batches out: the MI355X-native
// counterpart of the upstream adapter's arrow/arrow_encoder.h (same class, same public
// member functions, same record-batch schema and metadata), built on the native columnar
// batch encoder: one GPU predictor launch per batch of frames, Brotli of the planes on
// host threads, then the columns are handed to Arrow.
//
//   schema   timestamp: timestamp[ns] not null | deltaPredicted: bool | cgPredicted: bool |
//            preview: binary | highBytePlane: binary | lowBytePlane: binary
//   metadata xsize, ysize, shiftedLeft, deltaFrameHighPlane, deltaFrameLowPlane (the delta
//            frame's Brotli streams), deltaFrameCGPredicted ("true" / "false")
//
// Lives in its own library (libfpv_arrow.so) so that the codec itself does not depend on
// Arrow; built when an Arrow C++ installation is found (here: the one pyarrow ships).
#ifndef FPV_MI355X_ARROW_ENCODER_H_
#define FPV_MI355X_ARROW_ENCODER_H_

// the same Arrow headers the upstream arrow_encoder.h pulls in: user code relies on them
#include <arrow/array/builder_binary.h>
#include <arrow/array/builder_primitive.h>
#include <arrow/type.h>

#include <functional>
#include <future>
#include <memory>
#include <string>

#include "../fusion_power_video.h"

namespace fpvc {

class ArrowEncoder {
 public:
  typedef std::function<void(std::shared_ptr<arrow::RecordBatch>)> RecordBatchConsumer;

  // The consumer receives every full batch, the last (partial) one at Close, and a null
  // pointer when Close finds nothing left to flush.  The first frame pushed is the delta
  // frame of the stream and is also encoded as an ordinary frame.
  ArrowEncoder(size_t xsize, size_t ysize, int shift_to_left_align, bool big_endian,
               RecordBatchConsumer record_batch_consumer, int frames_per_batch = 10);
  ~ArrowEncoder();

  // The future yields `info` once `frame` has been consumed and may be reused.
  std::future<void*> PushFrame(uint64_t timestamp, uint16_t* frame, void* info);
  // Flushes; the future yields the timestamp of the last frame stored (-1: none).
  std::shared_future<int64_t> Close();
  // Extension (not upstream): the first failure of this encoder -- a GPU failure of its columnar
  // encoder or a record batch that could not be built -- or "".  The PushFrame call that hits a
  // GPU failure gets it through its future (an exception), later calls an invalid future;
  // nothing aborts.
  std::string LastError() const;

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

}  // namespace fpvc

#endif  // FPV_MI355X_ARROW_ENCODER_H_
