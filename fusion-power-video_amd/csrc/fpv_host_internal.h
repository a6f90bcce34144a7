// fpv_host_internal.h -- helpers shared by the host codec translation units.
#ifndef FPV_HOST_INTERNAL_H_
#define FPV_HOST_INTERNAL_H_

#include <stddef.h>
#include <stdint.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "fpv_hip.h"

namespace fpvc {
namespace internal {

bool Failure(const char* file, int line, const char* msg);
#define FPV_FAILURE(msg) ::fpvc::internal::Failure(__FILE__, __LINE__, msg)
[[noreturn]] void Fatal(const char* what, const char* detail);

// First failure of an adapter object (or of the process-wide Frame API): kept and printed once;
// the reference never aborts (cc:160-181: FAILURE returns false), so neither do the adapters.
// FPV_ABORT_ON_ERROR=1 aborts at the failure instead (rounds 1-3's behaviour).
class ErrorLatch {
 public:
  // Latches `who: what: detail` if nothing is latched yet.  Returns false (for `return latch.Fail(..)`).
  bool Fail(const char* who, const char* what, const char* detail);
  bool failed() const { return failed_.load(std::memory_order_acquire); }
  std::string text() const;
  void Reset();

 private:
  mutable std::mutex m_;
  std::atomic<bool> failed_{false};
  std::string text_;
};
ErrorLatch& FrameErrorLatch();  // the Frame API's (process-wide: frames are values, they own no session)

uint32_t Load32(const uint8_t* p);
uint64_t Load64(const uint8_t* p);
void Store32(uint32_t v, uint8_t* p);
void Store64(uint64_t v, uint8_t* p);
void Append32(uint32_t v, std::vector<uint8_t>* out);
bool OutOfRange(size_t pos, size_t width, size_t size);

// BrotliEncoderCompress(q = 1, default window, default mode) with the encoder's state in a
// per-thread arena; *size: capacity in, bytes out.  Byte-identical to the library's one-shot call.
bool BrotliCompress(const uint8_t* in, size_t n, uint8_t* out, size_t* size);
bool BrotliAppend(const uint8_t* in, size_t n, std::vector<uint8_t>* out);
bool BrotliInto(const uint8_t* in, size_t size, size_t* pos, uint8_t* dst,
                size_t cap, size_t* got);

// Runs fn(i) for every i in [0, n) on up to `threads` threads, the caller's included;
// returns when all are done.  The helpers come from a process-wide pool of parked
// threads (created on first need, never more than 64), so that per-batch fan-outs --
// Brotli of the planes of a batch, in both directions -- do not pay for thread creation.
void ParallelFor(size_t n, size_t threads, const std::function<void(size_t)>& fn);

// Device warm-up (fpv_host.cc): Request starts it on helper threads (one per visible device) for
// the directions that have not been warmed yet (object constructors), Wait also blocks until they
// are done (whoever is about to create a context).
enum { kWarmForward = 1, kWarmInverse = 2 };
void WarmUpRequest(int dirs);
void WarmUpWait(int dirs);

std::vector<int> DeviceList();
size_t EnvSize(const char* name, size_t dflt);

void BuildFrameChunk(const uint8_t* high, const uint8_t* low, const uint8_t* preview,
                     uint8_t flags, size_t npix, size_t npv, std::vector<uint8_t>* out);
void AppendCore(const uint8_t* high, const uint8_t* low, uint8_t flags, size_t npix,
                std::vector<uint8_t>* out);

// Process-wide cache of forward contexts: creating one allocates pinned
// staging memory (100s of ms), so sessions with the same geometry reuse them.
fpvhip_ctx* AcquireEncoderCtx(int device, uint32_t xs, uint32_t ys, int shift,
                              bool big_endian, uint32_t batch, uint32_t slots);
void ReleaseEncoderCtx(fpvhip_ctx* ctx);
// A context that saw a GPU error: taken out of the cache (so that no later session can be handed
// it, and a new context at the same address cannot be mistaken for it) and destroyed.
void DiscardEncoderCtx(fpvhip_ctx* ctx);
// Same cache for small inverse contexts (batch 1, one slot) used by Frame::Uncompress.
fpvhip_ctx* AcquireDecoderCtx(int device, uint32_t xs, uint32_t ys);
void ReleaseDecoderCtx(fpvhip_ctx* ctx);

// Inverse predictor front end shared by the two decoders: Brotli-decodes the
// planes of an image chunk straight into a slot's pinned staging memory and
// runs the GPU inverse (DecompressImage, cc:296-347).
class InverseEngine {
 public:
  // nullptr (after an stderr line) when no context can be created.
  static std::shared_ptr<InverseEngine> Create(size_t xs, size_t ys, uint32_t batch,
                                               uint32_t slots, int device);
  ~InverseEngine();

  size_t npix() const { return npix_; }
  uint32_t batch() const { return batch_; }
  bool SetDelta(const uint16_t* delta16);
  bool has_delta() const { return has_delta_; }

  // One image: in/size is the image chunk (flags byte, [low stream], high
  // stream).  On success *frame points at the decoded pixels inside the slot;
  // the caller must call Release(slot) when done with them.
  bool Decode(const uint8_t* in, size_t size, const uint16_t** frame, int* slot);
  void Release(int slot);

  // Many images in GPU batches, Brotli decoding of a batch spread over
  // `threads` host threads, the GPU inverting one batch while the host threads
  // unpack the next.  sink(i, pixels) is called once per image, in order, from
  // the calling thread; `pixels` is valid during the call only.  Images are
  // delivered up to (not including) the first one that fails, then false is
  // returned -- the order DecompressImage failures surface in the reference's
  // sequential loops (cc:914-936).
  typedef std::function<void(size_t index, const uint16_t* pixels)> Sink;
  // unshift >= 0: UnextractFrame (cc:850-862) runs on the device before the copy back, the
  // sink receives the raw file representation (pixel >> unshift, big- or little-endian).
  bool DecodeBatch(const uint8_t* const* ins, const size_t* sizes, size_t n,
                   const Sink& sink, size_t threads, int unshift = -1, bool big_endian = false);

  // The same for images whose planes are separate Brotli streams (columnar batches):
  // low == nullptr or NO_LOW_BYTES in flags means "no low plane".
  struct PlaneStreams {
    const uint8_t* high = nullptr;
    size_t high_size = 0;
    const uint8_t* low = nullptr;
    size_t low_size = 0;
    uint8_t flags = 0;
  };
  bool DecodePlanesBatch(const PlaneStreams* images, size_t n, const Sink& sink, size_t threads);

 private:
  InverseEngine() {}
  int AcquireSlot();
  bool Unpack(const uint8_t* in, size_t size, uint32_t slot, uint32_t k, uint8_t* flags);
  bool UnpackPlanes(const PlaneStreams& p, uint32_t slot, uint32_t k, uint8_t* flags);
  // Unpacker: (image index, slot, position in the batch, out flags) -> ok
  typedef std::function<bool(size_t, uint32_t, uint32_t, uint8_t*)> Unpacker;
  bool RunBatches(size_t n, const Unpacker& unpack, const Sink& sink, size_t threads, int unshift,
                  bool big_endian);

  fpvhip_ctx* ctx_ = nullptr;
  size_t npix_ = 0;
  uint32_t batch_ = 1, nslots_ = 1;
  bool has_delta_ = false;
  std::mutex m_;
  std::condition_variable cv_;
  std::vector<char> busy_;
  std::mutex run_m_;  // serialises batched requests (RunBatches)
};

}  // namespace internal
}  // namespace fpvc

#endif  // FPV_HOST_INTERNAL_H_
