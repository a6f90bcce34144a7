// columnar_batch.h -- batches of predicted + Brotli-compressed frames in one
// 64-byte-aligned structure-of-arrays buffer: the MI355X-native counterpart of the
// upstream adapter's columnar_batch/columnar_batch.h.  Same namespace, class names and
// public member functions, so that code written against upstream compiles unchanged;
// extensions are marked.  Everything per-pixel runs on the GPU, a whole batch per
// launch (fpv_columnar.cc).
#ifndef FPV_MI355X_COLUMNAR_BATCH_H_
#define FPV_MI355X_COLUMNAR_BATCH_H_

#include <functional>
#include <memory>
#include <vector>

#include "../fusion_power_video.h"

namespace fpvc {
namespace columnarbatch {

// ---------------------------------------------------------------------------
// One decoded image, as handed to the consumer of a ColumnarBatchDecoder.
//   FULL     16-bit pixels (data16), bpp = 16 - left shift of the stream
//   MSB8     the high byte of every pixel (data8)
//   PREVIEW  the quarter-resolution preview (data8)
// ---------------------------------------------------------------------------
class Image {
 public:
  enum Type { PREVIEW, MSB8, FULL };

  Image(int64_t timestamp = -1, size_t xsize = 0, size_t ysize = 0, uint8_t bpp = 0,
        Type type = FULL, std::vector<uint8_t>&& data = std::vector<uint8_t>());

  Type type() const { return kind_; }
  size_t timestamp() const { return static_cast<size_t>(when_); }
  size_t xsize() const { return width_; }
  size_t ysize() const { return height_; }
  size_t bpp() const { return bits_; }
  uint8_t* data8() { return bytes_.data(); }
  uint16_t* data16() { return reinterpret_cast<uint16_t*>(bytes_.data()); }

 private:
  std::vector<uint8_t> bytes_;
  int64_t when_;
  size_t width_, height_;
  Type kind_;
  uint8_t bits_;
};
using ImageProcessor = std::function<void(Image)>;

// ---------------------------------------------------------------------------
// What all batches of one stream share: geometry, the left shift the frames were
// stored with, and the delta frame -- its raw planes for prediction, and its planes
// as Brotli streams for whoever persists the schema (no preview, no clamped gradient).
// ---------------------------------------------------------------------------
class BatchSchema {
 public:
  BatchSchema(size_t xsize, size_t ysize, size_t shifted_left, Frame& uncompressed_delta_frame);

  Frame& delta_frame() { return delta_; }
  const std::vector<uint8_t>& compressedDeltaFrameHighPlane() { return delta_streams_[0]; }
  const std::vector<uint8_t>& compressedDeltaFrameLowPlane() { return delta_streams_[1]; }
  size_t shiftedLeft() const { return left_shift_; }
  size_t xsize() const { return width_; }
  size_t ysize() const { return height_; }

 private:
  Frame delta_;
  std::vector<uint8_t> delta_streams_[2];  // high, low
  size_t width_, height_, left_shift_;
};
using SchemaPtr = std::shared_ptr<BatchSchema>;

// ---------------------------------------------------------------------------
// The columns of a batch live in ONE backing buffer, each 64-byte aligned:
//   timestamps i64[n] | flags u8[n] | offsets u32[n+1] x 3 | preview streams |
//   high-plane streams | low-plane streams
// Stream k of a column occupies [offsets[k], offsets[k+1]).
// ---------------------------------------------------------------------------
class Batch {
 public:
  Batch(size_t batch_size, SchemaPtr schema);

  SchemaPtr schema() const { return schema_; }
  size_t length() const { return count_; }
  bool Empty() const { return count_ == 0; }
  bool Full() const { return count_ == rows_; }
  int64_t LatestTimestamp() const { return count_ ? stamps_[count_ - 1] : -1; }
  void Reset();

  // Compresses the planes of a predicted frame (Frame::CompressPredicted) into the columns.
  bool AppendPredicted(Frame predicted_frame);
  // One image decoded on its own (Frame::Uncompress on the GPU).  The batch decoder does
  // not use this: it inverts a whole batch per launch.
  Image ExtractImage(size_t index, Image::Type type);

  // ---- extensions ----
  // Appends planes that are compressed already (what the GPU batch encoder produces).
  bool AppendCompressed(int64_t timestamp, uint8_t flags, const uint8_t* high, size_t high_size,
                        const uint8_t* low, size_t low_size, const uint8_t* preview, size_t preview_size);
  // Column access: stream `which` (0 preview, 1 high, 2 low) of image `index`.
  const uint8_t* stream(size_t index, int which, size_t* size) const;
  uint8_t flags(size_t index) const { return flag_col_[index]; }
  int64_t timestamp(size_t index) const { return stamps_[index]; }
  const std::vector<uint8_t>& backing_buffer() const { return storage_; }

 private:
  std::vector<uint8_t> storage_;
  SchemaPtr schema_;
  size_t rows_, count_;
  int64_t* stamps_;
  uint8_t* flag_col_;
  uint32_t* offsets_[3];   // preview, high, low
  uint8_t* columns_[3];
  size_t capacity_[3];
};
using BatchPtr = std::shared_ptr<Batch>;
using BatchProcessor = std::function<void(BatchPtr)>;

}  // namespace columnarbatch
}  // namespace fpvc

#endif  // FPV_MI355X_COLUMNAR_BATCH_H_
