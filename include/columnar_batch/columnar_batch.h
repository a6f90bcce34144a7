// columnar_batch.h -- batches of predicted + Brotli-compressed frames in one
// 64-byte-aligned structure-of-arrays buffer: the MI355X-native counterpart of the
// upstream adapter's columnar_batch/columnar_batch.h.  Same namespace, classes and
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

// What every batch of one stream shares: geometry, the left shift the frames were
// stored with, and the delta frame (raw planes for prediction, Brotli streams for
// whoever persists the schema).
class BatchSchema {
 public:
  BatchSchema(size_t xsize, size_t ysize, size_t shifted_left, Frame& uncompressed_delta_frame);
  size_t xsize() const { return xsize_; }
  size_t ysize() const { return ysize_; }
  size_t shiftedLeft() const { return shifted_left_; }
  Frame& delta_frame() { return delta_frame_; }
  // The delta frame is NOT clamped-gradient predicted.
  const std::vector<uint8_t>& compressedDeltaFrameHighPlane() { return delta_high_stream_; }
  const std::vector<uint8_t>& compressedDeltaFrameLowPlane() { return delta_low_stream_; }

 private:
  size_t xsize_, ysize_, shifted_left_;
  std::vector<uint8_t> delta_high_stream_, delta_low_stream_;
  Frame delta_frame_;
};
typedef std::shared_ptr<BatchSchema> SchemaPtr;

// One decoded image handed to the consumer.
class Image {
 public:
  enum Type { PREVIEW, MSB8, FULL };
  Image(int64_t timestamp = -1, size_t xsize = 0, size_t ysize = 0, uint8_t bpp = 0,
        Type type = FULL, std::vector<uint8_t>&& data = std::vector<uint8_t>());
  size_t timestamp() const { return (size_t)timestamp_; }
  size_t xsize() const { return xsize_; }
  size_t ysize() const { return ysize_; }
  size_t bpp() const { return bpp_; }
  uint8_t* data8() { return data_.data(); }
  uint16_t* data16() { return reinterpret_cast<uint16_t*>(data_.data()); }
  Type type() const { return type_; }

 private:
  int64_t timestamp_;
  size_t xsize_, ysize_;
  uint8_t bpp_;
  std::vector<uint8_t> data_;
  Type type_;
};
typedef std::function<void(Image)> ImageProcessor;

// Columns: timestamps i64[n], flags u8[n], three offset arrays u32[n+1], then the
// preview / high / low Brotli streams back to back, every column 64-byte aligned in
// one backing buffer.
class Batch {
 public:
  Batch(size_t batch_size, SchemaPtr schema);
  void Reset();
  // Compresses the planes of a predicted frame (Frame::CompressPredicted) into the columns.
  bool AppendPredicted(Frame predicted_frame);
  bool Empty() const { return length_ == 0; }
  bool Full() const { return length_ == batch_size_; }
  int64_t LatestTimestamp() const { return length_ == 0 ? -1 : timestamps_[length_ - 1]; }
  size_t length() const { return length_; }
  // One image, decoded on its own (Frame::Uncompress on the GPU).  The batch decoder
  // does not use this: it inverts a whole batch per launch.
  Image ExtractImage(size_t index, Image::Type type);
  SchemaPtr schema() const { return schema_; }

  // ---- extensions ----
  // Appends already compressed planes (what the GPU batch encoder produces).
  bool AppendCompressed(int64_t timestamp, uint8_t flags, const uint8_t* high, size_t high_size,
                        const uint8_t* low, size_t low_size, const uint8_t* preview, size_t preview_size);
  // Column access: stream `which` (0 preview, 1 high, 2 low) of image `index`.
  const uint8_t* stream(size_t index, int which, size_t* size) const;
  uint8_t flags(size_t index) const { return flags_[index]; }
  int64_t timestamp(size_t index) const { return timestamps_[index]; }
  const std::vector<uint8_t>& backing_buffer() const { return backing_buffer_; }

 private:
  SchemaPtr schema_;
  size_t batch_size_, length_;
  size_t previews_capacity_, plane_capacity_;
  std::vector<uint8_t> backing_buffer_;
  int64_t* timestamps_;
  uint8_t* flags_;
  uint32_t* offsets_[3];   // preview, high, low
  uint8_t* columns_[3];
  size_t capacity_[3];
};
typedef std::shared_ptr<Batch> BatchPtr;
typedef std::function<void(BatchPtr)> BatchProcessor;

}  // namespace columnarbatch
}  // namespace fpvc

#endif  // FPV_MI355X_COLUMNAR_BATCH_H_
