// fusion_power_video.h -- public C++ API of the MI355X-native codec.
//
// Source-compatible with the header of google/fusion-power-video: the same
// namespace, class names, method signatures and default arguments, so the
// upstream encode.cc / decode.cc / benchmark.cc (and code written against the
// upstream header) compile unchanged.  Everything private is different: the
// per-frame predictor/residual stage runs as HIP kernels on MI355X GPUs behind
// the C ABI of fpv_hip.h; Brotli stays on host worker threads.  There is no CPU
// fallback: without a usable GPU the encoder aborts with a message and the
// decoders report failure.
//
// Interface map (upstream fusion_power_video.h line numbers):
//   UnextractFrame        h:32     StreamingDecoder     h:37-57
//   FrameState/FrameFlags h:59-73  Frame                h:75-139
//   RandomAccessDecoder   h:143-172  Encoder            h:175-255
#ifndef FUSION_POWER_VIDEO_H_
#define FUSION_POWER_VIDEO_H_

#include <stddef.h>
#include <stdint.h>

#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <queue>
#include <thread>
#include <string>
#include <vector>

namespace fpvc {

// 16-bit frame -> bytes of the original raw file (undo shift, choose endianness).
void UnextractFrame(const uint16_t* img, size_t xsize, size_t ysize, int shift,
                    bool big_endian, uint8_t* out);

// Incremental decoder: feed the stream in arbitrary pieces; `callback` fires
// once per frame that became complete, in order.  ok == false reports a
// corrupt stream (frame == nullptr then).
class StreamingDecoder {
 public:
  void Decode(const uint8_t* bytes, size_t size,
              std::function<void(bool ok, uint16_t* frame, size_t xsize,
                                 size_t ysize, void* payload)> callback,
              void* payload = nullptr);

 private:
  struct State;
  std::shared_ptr<State> state_;
};

enum FrameState {
  EMPTY = 0,
  RAW = 1,
  PREVIEW_GENERATED = 2,
  DELTA_PREDICTED = 4,
  CG_PREDICTED = 8,
  COMPRESSED = 16,
};

enum FrameFlags {
  NONE = 0,
  USE_DELTA = 1,
  USE_CG = 2,
  NO_LOW_BYTES = 4,
};

// One frame as byte planes, with the explicit predict / compress steps that the
// batch adapters drive.  Predict and Uncompress run on the GPU.
class Frame {
 public:
  static Frame EMPTY;

  Frame(size_t xsize = 0, size_t ysize = 0, const uint16_t* image = nullptr,
        int shift_to_left_align = 0, bool big_endian = false,
        int64_t timestamp = -1);
  Frame(size_t xsize, size_t ysize, const uint8_t* image, int64_t timestamp = -1);
  Frame(size_t xsize, size_t ysize, uint8_t flags, uint8_t state,
        std::vector<uint8_t>&& high, std::vector<uint8_t>&& low,
        std::vector<uint8_t>&& preview, int64_t timestamp = -1);
  // Copies synchronise with the lazy split of the source (a RAW frame shared as the delta
  // frame may be split by another thread's high() / flags() at any time); moves are the
  // owner's business, as upstream.
  Frame(const Frame& other);
  Frame& operator=(const Frame& other);
  Frame(Frame&&) = default;
  Frame& operator=(Frame&&) = default;

  size_t xsize() const { return xsize_; }
  size_t ysize() const { return ysize_; }
  uint8_t flags() const { if (PlanesPending()) SplitNow(); return flags_; }
  uint8_t state() const { return state_; }
  int64_t timestamp() const { return timestamp_; }
  const std::vector<uint8_t>& high() { if (PlanesPending()) SplitNow(); return high_; }
  const std::vector<uint8_t>& low() { if (PlanesPending()) SplitNow(); return low_; }
  const std::vector<uint8_t>& preview() { return preview_; }
  std::vector<uint8_t>&& MoveOutHigh() { Touch(); return std::move(high_); }
  std::vector<uint8_t>&& MoveOutLow() { Touch(); return std::move(low_); }
  std::vector<uint8_t>&& MoveOutPreview() { return std::move(preview_); }

  static size_t MaxCompressedPlaneSize(size_t xsize, size_t ysize);
  static size_t MaxCompressedPreviewSize(size_t xsize, size_t ysize);
  size_t MaxCompressedPlaneSize();
  size_t MaxCompressedPreviewSize();

  void Compress(Frame& delta_frame = EMPTY);
  void Uncompress(Frame& delta_frame = EMPTY);
  void Predict(Frame& delta_frame = EMPTY);
  void CompressPredicted(size_t* encoded_high_size, uint8_t* encoded_high_buffer,
                         size_t* encoded_low_size, uint8_t* encoded_low_buffer,
                         size_t* encoded_preview_size,
                         uint8_t* encoded_preview_buffer, bool parallel = true);
  void OutputCore(std::vector<uint8_t>* out);
  void OutputFull(std::vector<uint8_t>* out);

  // Extension (not in the upstream header): the first GPU failure of the Frame API in this
  // process ("" while there is none).  Every method above is void, as upstream (h:75-139), and
  // the reference never aborts: the method that hits a failure leaves its frame as it found it
  // (Predict: still RAW; Compress: not COMPRESSED; Uncompress: predictions still applied, the
  // state bits say so), and later GPU-touching methods are no-ops until ClearError().
  // FPV_ABORT_ON_ERROR=1 aborts at the failure instead.
  static std::string LastError();
  static void ClearError();

 protected:
  std::vector<uint8_t> preview_;
  std::vector<uint8_t> high_;
  std::vector<uint8_t> low_;

 private:
  friend class Encoder;
  size_t xsize_ = 0, ysize_ = 0, size_ = 0;
  uint8_t flags_ = FrameFlags::NONE;
  uint8_t state_ = FrameState::EMPTY;
  int64_t timestamp_ = -1;
  // Pixels as given to the first constructor, kept so that Predict can run
  // split + prediction as ONE pass on the GPU.  Until then the byte planes (and the
  // NO_LOW_BYTES flag) are produced on first use: planes_pending_.
  std::shared_ptr<std::vector<uint16_t>> pixels_;
  int shift_ = 0;
  bool big_endian_ = false;
  uint64_t content_id_ = 0;  // changes whenever the planes change; copies share it
  mutable int planes_pending_ = 0;  // read / written atomically: a shared delta frame may be split by any thread
  bool PlanesPending() const { return __atomic_load_n(&planes_pending_, __ATOMIC_ACQUIRE) != 0; }
  void CopyFrom(const Frame& other);
  void SplitNow() const;     // the first constructor's split (cc:370-451), on the GPU; thread-safe
  void Touch();              // SplitNow if pending + a new content id
};

// Needs the whole file; any frame or preview in any order.
class RandomAccessDecoder {
 public:
  bool Init(const uint8_t* data, size_t size);
  bool DecodeFrame(size_t index, uint16_t* frame) const;
  bool DecodePreview(size_t index, uint8_t* preview) const;

  size_t xsize() const { return xsize_; }
  size_t ysize() const { return ysize_; }
  size_t preview_xsize() const { return xsize_ / 4; }
  size_t preview_ysize() const { return ysize_ / 4; }
  size_t numframes() const { return frame_offsets_.size(); }

  // Extension (not in the upstream header): decode `count` frames in one GPU
  // batch; out[i] receives frame indices[i].  Host Brotli decoding of the
  // batch is spread over `threads` workers.
  bool DecodeFrames(const size_t* indices, size_t count, uint16_t* const* out,
                    size_t threads = 8) const;
  // Extension: the same with UnextractFrame (h:32) fused on the GPU -- out[i]
  // receives xsize*ysize*2 raw bytes of frame indices[i], exactly what
  // UnextractFrame(decoded, xsize, ysize, shift, big_endian, out[i]) would write.
  bool DecodeFramesRaw(const size_t* indices, size_t count, uint8_t* const* out,
                       int shift, bool big_endian, size_t threads = 8) const;

 private:
  struct State;
  bool DecodeSharded(const size_t* indices, size_t count, void* const* out, size_t threads,
                     int unshift, bool big_endian) const;
  size_t xsize_ = 0, ysize_ = 0;
  std::vector<size_t> frame_offsets_;
  const uint8_t* data_ = nullptr;
  size_t size_ = 0;
  std::shared_ptr<State> state_;
};

// Encoder session.  num_threads host workers run Brotli and chunk assembly;
// frames are batched, predicted on the GPU(s) on a side stream and handed to
// the workers; callbacks fire in submission order, one at a time.
// num_threads == 0: fully synchronous, one frame per GPU round trip.
class Encoder {
 public:
  Encoder(size_t num_threads = 8, int shift_to_left_align = 0,
          bool big_endian = false);
  ~Encoder();
  Encoder(const Encoder&) = delete;
  Encoder& operator=(const Encoder&) = delete;

  typedef std::function<void(const uint8_t* compressed, size_t size, void* payload)>
      Callback;

  // Emits header + compressed delta frame through `callback` before returning.
  // Geometry, encode side (differences from upstream, whose only limit is MAX_IMAGE_SIZE,
  // cc:164): xsize and ysize must be multiples of 4 -- upstream reads out of bounds for other
  // sizes (the 4x4 preview boxes and the preview's clamped gradient, cc:500-512, cc:575-586), so
  // its output is undefined there -- and xsize at most 32768 (the kernels' LDS line buffers).
  // Anything else latches LastError() ("unsupported geometry ...") and the session emits nothing.
  // Decoding has neither limit (odd sizes written by the CPU reference decode).
  void Init(const uint16_t* delta_frame, size_t xsize, size_t ysize,
            Callback callback, void* payload);
  // Queues one frame.  As upstream: `img` must stay valid until this frame's
  // callback has run (worker threads copy it into pinned staging memory), so
  // the caller needs MaxQueued() buffers.  Blocks while MaxQueued() frames are
  // in flight.  With num_threads == 0 the frame is consumed before the call returns.
  void CompressFrame(const uint16_t* img, Callback callback, void* payload);
  // Drains everything, then emits the frame index.  Idempotent.
  void Finish(Callback callback, void* payload);
  // Upper bound of frames in flight; usable right after construction.
  size_t MaxQueued() const;
  // Extension (not in the upstream header): stage statistics of the session, valid after
  // Finish.  out[0] frames, [1] wall ms since Init, [2] Brotli ms and [3] staging-copy ms
  // summed over the workers, [4] mean GPU batch latency ms (copy in + kernels + copy out),
  // [5] batches, [6] GB/s copied in and [7] out over the wall time, [8] host stage
  // (Brotli + copies, per thread) / wall, [9] ms the dispatchers waited for the GPU, [10] devices
  // of the session D, then per device d < D (at most 16): [11 + 2 d] batches it took, [12 + 2 d]
  // ms its dispatcher waited for it.  Returns the number of values written (at most n).
  size_t LastStats(double* out, size_t n) const;
  // Extension: the session's first GPU failure ("" while there is none).  The upstream API is
  // void everywhere; after a failure the session accepts no more frames (their callbacks never
  // run), Finish emits no frame index, and the process goes on.  FPV_ABORT_ON_ERROR=1 makes
  // such a failure abort() instead.
  const char* LastError() const;

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

}  // namespace fpvc

#endif  // FUSION_POWER_VIDEO_H_
