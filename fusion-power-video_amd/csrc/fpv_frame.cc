// fpv_frame.cc -- fpvc::Frame (h:75-139): one frame as byte planes with the
// explicit Predict / Compress / Uncompress steps the batch adapters drive.
// Plane arithmetic (split, preview, predictors and their inverses) runs on the
// GPU through single-frame batches of the C ABI; Brotli and the serialisation
// stay on the host exactly like the reference's (cc:643-728, cc:820-846).
#include <brotli/encode.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <future>
#include <map>
#include <mutex>
#include <tuple>

#include "fpv_host_internal.h"
#include "fusion_power_video.h"

namespace fpvc {
using namespace internal;

Frame Frame::EMPTY(0, 0);

namespace {
// GPU failures of the Frame API are latched process-wide (Frame::LastError): the method that hit
// one leaves the frame as it found it, later GPU-touching methods are no-ops until
// Frame::ClearError().  Every signature here is void (h:75-139); the reference never aborts.
bool FrameFail(const char* what, const char* detail) { return FrameErrorLatch().Fail("Frame API", what, detail); }
bool FrameFailed() { return FrameErrorLatch().failed(); }

int FrameDevice() {  // -1 (latched) when no GPU is usable
  const std::vector<int> d = DeviceList();
  if (d.empty()) {
    FrameFail("Frame: no usable GPU", fpvhip_last_error(nullptr));
    return -1;
  }
  return d[0];
}

// Leased contexts.  A context that saw a failure is not handed back to the cache.
struct Fwd {  // forward context (batch 1, one slot)
  fpvhip_ctx* ctx = nullptr;
  bool bad = false;
  Fwd(size_t xs, size_t ys, int shift, bool be) {
    const int dev = FrameDevice();
    if (dev < 0) return;
    ctx = AcquireEncoderCtx(dev, (uint32_t)xs, (uint32_t)ys, shift, be, 1, 1);
    if (!ctx) FrameFail("Frame: fpvhip_create", fpvhip_last_error(nullptr));
  }
  ~Fwd() {
    if (ctx && bad) DiscardEncoderCtx(ctx);
    else if (ctx) ReleaseEncoderCtx(ctx);
  }
  bool Check(int rc, const char* what) {
    if (rc == 0) return true;
    bad = true;
    return FrameFail(what, fpvhip_last_error(ctx));
  }
};
struct Inv {  // inverse context
  fpvhip_ctx* ctx = nullptr;
  bool bad = false;
  Inv(size_t xs, size_t ys) {
    const int dev = FrameDevice();
    if (dev < 0) return;
    ctx = AcquireDecoderCtx(dev, (uint32_t)xs, (uint32_t)ys);
    if (!ctx) FrameFail("Frame: fpvhip_create_decoder", fpvhip_last_error(nullptr));
  }
  ~Inv() {
    if (ctx && bad) DiscardEncoderCtx(ctx);
    else if (ctx) ReleaseDecoderCtx(ctx);
  }
  bool Check(int rc, const char* what) {
    if (rc == 0) return true;
    bad = true;
    return FrameFail(what, fpvhip_last_error(ctx));
  }
};
// cc:186-214 into a vector (Frame::Uncompress does not check sizes)
void BrotliToVector(std::vector<uint8_t>* v) {
  std::vector<uint8_t> out;
  size_t cap = v->size() * 8 + 4096;
  for (;;) {
    out.resize(cap);
    size_t pos = 0, got = 0;
    const bool ok = BrotliInto(v->data(), v->size(), &pos, out.data(), cap, &got);
    if (ok || got < cap) {  // done, or a corrupt stream: keep what was produced
      out.resize(got);
      break;
    }
    cap *= 4;
  }
  v->swap(out);
}

std::atomic<uint64_t> g_content_id(1);
uint64_t NewContentId() { return g_content_id.fetch_add(1, std::memory_order_relaxed); }

// Several threads may ask for the planes of one lazily split frame (a shared delta frame).
std::mutex& SplitLock(const void* frame) {
  static std::mutex locks[16];
  return locks[(reinterpret_cast<uintptr_t>(frame) >> 6) & 15];
}

// ---- Predict calls of several threads combined into GPU batches ----
//
// The adapters above this interface (upstream's columnar and Arrow encoders) call
// Frame::Predict from a pool of threads, one frame per call.  One synchronous GPU round
// trip per call wastes the device and the host (a launch, three stream hand-offs and a
// wait per frame, all of them contending in the runtime), so calls that arrive while the
// GPU is busy with the previous batch are collected: every caller copies its pixels into
// its place of the open batch's pinned staging buffer, the first one (the leader) closes
// the batch as soon as the previous one has left the GPU, submits it, and everybody picks
// up its own planes.  A lone caller gets a batch of one at once: nothing waits for company.
class PredictCombiner {
 public:
  // Frames per GPU batch: up to 16, bounded by a byte budget of pinned staging per slot (in +
  // out = 4 bytes per pixel: 64 MB -> 16 frames up to 1 Mpixel, 4 at 4 Mpixel, 1 at 4096x4096);
  // FPV_COMBINER_MB overrides.  A combiner nobody has used for a while gives its context back.
  static constexpr uint32_t kMaxBatchCap = 16, kSlots = 2;
  static uint32_t BatchFor(size_t npix) {
    const size_t budget = EnvSize("FPV_COMBINER_MB", 64) << 20;
    return (uint32_t)std::min<size_t>(kMaxBatchCap, std::max<size_t>(1, budget / std::max<size_t>(1, npix * 4)));
  }
  typedef std::tuple<int, size_t, size_t, int, bool> Key;  // device, xs, ys, shift, big endian

  static std::shared_ptr<PredictCombiner> Get(const Key& key) {
    static std::mutex m;
    static std::map<Key, std::shared_ptr<PredictCombiner>>* all = new std::map<Key, std::shared_ptr<PredictCombiner>>();
    std::lock_guard<std::mutex> l(m);
    // A combiner nobody holds and nobody has used for a while gives its context back to the
    // pool of idle contexts, where the FPV_POOL_MB bound applies (it used to stay leased for
    // the life of the process).
    const int64_t now = NowMs(), idle_ms = (int64_t)EnvSize("FPV_COMBINER_IDLE_MS", 2000);
    for (auto d = all->begin(); d != all->end();) {
      if (d->first != key && d->second.use_count() == 1 && now - d->second->last_used_ms_.load() > idle_ms) d = all->erase(d);
      else ++d;
    }
    auto it = all->find(key);
    if (it != all->end() && (it->second->bad_ || !it->second->ctx_)) {  // its context saw a failure: start afresh
      all->erase(it);
      it = all->end();
    }
    if (it != all->end()) {
      it->second->last_used_ms_.store(now);
      return it->second;
    }
    if (all->size() >= 4)  // keep the pinned memory bounded: drop a geometry nobody is using
      for (auto d = all->begin(); d != all->end(); ++d)
        if (d->second.use_count() == 1) { all->erase(d); break; }
    std::shared_ptr<PredictCombiner> c(new PredictCombiner(key));
    c->last_used_ms_.store(now);
    (*all)[key] = c;
    return c;
  }
  static int64_t NowMs() {
    return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }

  ~PredictCombiner() {
    if (ctx_ && bad_) DiscardEncoderCtx(ctx_);
    else if (ctx_) ReleaseEncoderCtx(ctx_);
  }

  struct Result { const uint8_t *high, *low, *preview; uint8_t flags; };

  // `set_delta(ctx)` uploads the caller's delta frame (false: it failed); it runs (in the
  // leader) only when the context holds a different one.  `take(result)` copies the caller's
  // planes out.  Returns false when the batch failed on the GPU (latched by the leader; every
  // member of the batch gets false, nobody's `take` runs).
  template <class SetDelta, class Take>
  bool Run(const uint16_t* pixels, uint64_t delta_id, SetDelta set_delta, Take take) {
    if (!ctx_) return false;  // the context could not be created (latched then)
    std::unique_lock<std::mutex> l(m_);
    Batch* b = nullptr;
    for (;;) {
      if (open_ && open_->delta == delta_id && open_->joined < max_batch_) { b = open_; break; }
      if (!open_) {
        const int s = slot_free_[0] ? 0 : slot_free_[1] ? 1 : -1;
        if (s >= 0) {
          slot_free_[s] = false;
          b = open_ = new Batch();
          b->slot = s;
          b->delta = delta_id;
          break;
        }
      }
      cv_.wait(l);  // the open batch is full or for another delta frame, or both slots are busy
    }
    const uint32_t idx = b->joined++;
    l.unlock();
    if (uint16_t* dst = fpvhip_stage_in(ctx_, (uint32_t)b->slot, idx)) {
      memcpy(dst, pixels, npix_ * 2);
    } else {
      FrameFail("fpvhip_stage_in", fpvhip_last_error(ctx_));
      l.lock();
      b->failed = true;
      bad_ = true;
      l.unlock();
    }
    l.lock();
    b->staged++;
    if (idx == 0) {  // leader
      cv_.wait(l, [&] { return !inflight_; });
      open_ = nullptr;  // closed: later callers start the next batch
      inflight_ = true;
      cv_.notify_all();
      cv_.wait(l, [&] { return b->staged == b->joined; });
      const uint32_t n = b->joined;
      bool ok = !b->failed;
      l.unlock();
      if (ok && ctx_delta_ != delta_id) {  // only the leader of the batch in flight touches the context
        ctx_delta_ = 0;
        if (delta_id) ok = set_delta(ctx_);
        if (ok) ctx_delta_ = delta_id;
      }
      if (ok && fpvhip_predict_submit(ctx_, (uint32_t)b->slot, n, delta_id ? 1 : 0))
        ok = FrameFail("fpvhip_predict_submit", fpvhip_last_error(ctx_));
      if (ok && fpvhip_predict_wait(ctx_, (uint32_t)b->slot, &b->high, &b->low, &b->preview, &b->flags))
        ok = FrameFail("fpvhip_predict_wait", fpvhip_last_error(ctx_));
      l.lock();
      if (!ok) { b->failed = true; bad_ = true; }
      b->done = true;
      inflight_ = false;
      cv_.notify_all();
    } else {
      cv_.notify_all();
      cv_.wait(l, [&] { return b->done; });
    }
    const bool ok = !b->failed;
    l.unlock();
    if (ok) {
      Result r;
      r.high = b->high + (size_t)idx * npix_;
      r.low = b->low + (size_t)idx * npix_;
      r.preview = b->preview + (size_t)idx * npv_;
      r.flags = b->flags[idx];
      take(r);
    }
    l.lock();
    if (++b->left == b->joined) {
      slot_free_[b->slot] = true;
      delete b;
      cv_.notify_all();
    }
    return ok;
  }

 private:
  struct Batch {
    int slot = 0;
    uint64_t delta = 0;
    uint32_t joined = 0, staged = 0, left = 0;
    bool done = false, failed = false;
    const uint8_t *high = nullptr, *low = nullptr, *preview = nullptr, *flags = nullptr;
  };
  explicit PredictCombiner(const Key& k)
      : npix_(std::get<1>(k) * std::get<2>(k)), npv_((std::get<1>(k) / 4) * (std::get<2>(k) / 4)),
        max_batch_(BatchFor(std::get<1>(k) * std::get<2>(k))) {
    ctx_ = AcquireEncoderCtx(std::get<0>(k), (uint32_t)std::get<1>(k), (uint32_t)std::get<2>(k), std::get<3>(k),
                             std::get<4>(k), max_batch_, kSlots);
    if (!ctx_) FrameFail("Frame: fpvhip_create", fpvhip_last_error(nullptr));
  }
  std::mutex m_;
  std::condition_variable cv_;
  fpvhip_ctx* ctx_ = nullptr;
  std::atomic<bool> bad_{false};  // the context saw a failure: destroyed, not cached, when the combiner goes (written under mu_, read by Get under the registry's lock)
  size_t npix_, npv_;
  uint32_t max_batch_;
  std::atomic<int64_t> last_used_ms_{0};
  uint64_t ctx_delta_ = 0;  // content id of the delta frame the context holds (0: unknown / none)
  bool slot_free_[kSlots] = {true, true};
  bool inflight_ = false;
  Batch* open_ = nullptr;
};
constexpr uint32_t PredictCombiner::kMaxBatchCap;
}  // namespace

std::string Frame::LastError() { return FrameErrorLatch().text(); }
void Frame::ClearError() { FrameErrorLatch().Reset(); }

size_t Frame::MaxCompressedPlaneSize(size_t xsize, size_t ysize) {
  return BrotliEncoderMaxCompressedSize(xsize * ysize);
}
size_t Frame::MaxCompressedPreviewSize(size_t xsize, size_t ysize) {
  return BrotliEncoderMaxCompressedSize(xsize * ysize / 16);
}
size_t Frame::MaxCompressedPlaneSize() { return BrotliEncoderMaxCompressedSize(size_); }
size_t Frame::MaxCompressedPreviewSize() { return BrotliEncoderMaxCompressedSize(size_ / 16); }

// cc:370-451: the pixels are kept so that Predict is ONE fused pass (split + preview +
// predictors) on the GPU; the byte planes and the NO_LOW_BYTES flag of the RAW state are
// produced by SplitNow when somebody asks for them first.
Frame::Frame(size_t xsize, size_t ysize, const uint16_t* image, int shift_to_left_align,
             bool big_endian, int64_t timestamp)
    : xsize_(xsize), ysize_(ysize), size_(xsize * ysize), timestamp_(timestamp),
      shift_(shift_to_left_align), big_endian_(big_endian) {
  if (!image) return;
  state_ = FrameState::RAW;
  pixels_ = std::make_shared<std::vector<uint16_t>>(image, image + size_);
  planes_pending_ = 1;
  content_id_ = NewContentId();
}

void Frame::SplitNow() const {
  std::lock_guard<std::mutex> l(SplitLock(this));
  if (!PlanesPending()) return;
  Frame* self = const_cast<Frame*>(this);
  // (after a latched failure the planes come out as zeros of the right size: callers index them)
  self->high_.resize(size_);
  if (shift_ != 8) self->low_.resize(size_);
  int no_low = 0;
  if (!FrameFailed()) {
    Fwd f(xsize_, ysize_, shift_, big_endian_);
    if (f.ctx && f.Check(fpvhip_split_frame(f.ctx, pixels_->data(), self->high_.data(),
                                            shift_ != 8 ? self->low_.data() : nullptr, &no_low), "fpvhip_split_frame") &&
        no_low)
      self->flags_ |= FrameFlags::NO_LOW_BYTES;
  }
  __atomic_store_n(&planes_pending_, 0, __ATOMIC_RELEASE);
}

// Copying reads the planes, the flags and the pending mark: under the source's split lock, so
// that a split running on another thread is either wholly seen or not at all.
void Frame::CopyFrom(const Frame& o) {
  std::lock_guard<std::mutex> l(SplitLock(&o));
  preview_ = o.preview_;
  high_ = o.high_;
  low_ = o.low_;
  xsize_ = o.xsize_; ysize_ = o.ysize_; size_ = o.size_;
  flags_ = o.flags_;
  state_ = o.state_;
  timestamp_ = o.timestamp_;
  pixels_ = o.pixels_;
  shift_ = o.shift_;
  big_endian_ = o.big_endian_;
  content_id_ = o.content_id_;
  __atomic_store_n(&planes_pending_, __atomic_load_n(&o.planes_pending_, __ATOMIC_ACQUIRE), __ATOMIC_RELEASE);
}
Frame::Frame(const Frame& other) { CopyFrom(other); }
Frame& Frame::operator=(const Frame& other) {
  if (this != &other) CopyFrom(other);
  return *this;
}

void Frame::Touch() {
  if (PlanesPending()) SplitNow();
  content_id_ = NewContentId();
}

// cc:453-465
Frame::Frame(size_t xsize, size_t ysize, const uint8_t* image, int64_t timestamp)
    : xsize_(xsize), ysize_(ysize), size_(xsize * ysize), flags_(FrameFlags::NO_LOW_BYTES),
      timestamp_(timestamp) {
  if (image) {
    state_ = FrameState::RAW;
    high_.assign(image, image + size_);
  }
  content_id_ = NewContentId();
}

// cc:467-489, including its two no-op assignments to the parameters (cc:482, 485)
Frame::Frame(size_t xsize, size_t ysize, uint8_t flags, uint8_t state, std::vector<uint8_t>&& high,
             std::vector<uint8_t>&& low, std::vector<uint8_t>&& preview, int64_t timestamp)
    : preview_(std::move(preview)), high_(std::move(high)), low_(std::move(low)), xsize_(xsize),
      ysize_(ysize), size_(xsize * ysize), flags_(flags), state_(state), timestamp_(timestamp) {
  if (preview_.empty()) state_ &= ~FrameState::PREVIEW_GENERATED;
  if (!low_.empty()) flags_ &= ~FrameFlags::NO_LOW_BYTES;
  content_id_ = NewContentId();
}

// cc:777-785
void Frame::Predict(Frame& delta_frame) {
  const uint8_t predicted = FrameState::DELTA_PREDICTED | FrameState::CG_PREDICTED;
  if (state_ & FrameState::CG_PREDICTED) {
    const bool want_delta = delta_frame.state() > FrameState::EMPTY;
    if (!want_delta || (state_ & FrameState::DELTA_PREDICTED)) return;  // nothing left to do
  }
  if ((state_ & (predicted | FrameState::COMPRESSED)) || size_ == 0 || (!PlanesPending() && high_.size() != size_))
    Fatal("Frame::Predict", "only frames in the RAW state can be predicted on the GPU");  // API misuse, not a GPU failure
  if (FrameFailed()) return;  // latched: the frame stays as it is (Frame::LastError)
  const bool with_delta = delta_frame.state() > FrameState::EMPTY;
  const size_t npv = (xsize_ / 4) * (ysize_ / 4);
  if (pixels_) {
    // the fused pass over the pixels, in one batch with whatever other threads are asking for
    auto set_delta = [&](fpvhip_ctx* ctx) -> bool {
      if (delta_frame.pixels_ && delta_frame.shift_ == shift_ && delta_frame.big_endian_ == big_endian_) {
        if (fpvhip_set_delta_raw(ctx, delta_frame.pixels_->data())) return FrameFail("fpvhip_set_delta_raw", fpvhip_last_error(ctx));
      } else {
        if (delta_frame.PlanesPending()) delta_frame.SplitNow();
        if (fpvhip_set_delta_planes(ctx, delta_frame.high_.data(),
                                    delta_frame.low_.size() == size_ ? delta_frame.low_.data() : nullptr))
          return FrameFail("fpvhip_set_delta_planes", fpvhip_last_error(ctx));
      }
      return !FrameFailed();
    };
    auto take = [&](const PredictCombiner::Result& r) {
      high_.assign(r.high, r.high + size_);
      if (shift_ != 8) low_.assign(r.low, r.low + size_);
      preview_.assign(r.preview, r.preview + npv);
      flags_ = (uint8_t)(r.flags & (FrameFlags::USE_DELTA | FrameFlags::USE_CG | FrameFlags::NO_LOW_BYTES));
    };
    const int dev = FrameDevice();
    if (dev < 0) return;
    if (!PredictCombiner::Get(PredictCombiner::Key(dev, xsize_, ysize_, shift_, big_endian_))
             ->Run(pixels_->data(), with_delta ? delta_frame.content_id_ : 0, set_delta, take))
      return;  // latched; the frame keeps its pixels and stays RAW
    planes_pending_ = 0;
  } else {
    // planes only (8-bit constructor, or a frame that went through Uncompress)
    if (with_delta && delta_frame.PlanesPending()) delta_frame.SplitNow();
    Fwd f(xsize_, ysize_, 0, false);
    if (!f.ctx) return;
    const uint8_t *gh, *gl, *gp, *gf;
    if (with_delta &&
        !f.Check(fpvhip_set_delta_planes(f.ctx, delta_frame.high_.data(),
                                         delta_frame.low_.size() == size_ ? delta_frame.low_.data() : nullptr),
                 "fpvhip_set_delta_planes"))
      return;
    uint8_t *sh = fpvhip_stage_plane(f.ctx, 0, 0, 0), *sl = fpvhip_stage_plane(f.ctx, 0, 0, 1);
    if (!sh || !sl) { f.Check(-1, "fpvhip_stage_plane"); return; }
    memcpy(sh, high_.data(), size_);
    if (low_.size() == size_) memcpy(sl, low_.data(), size_);
    else memset(sl, 0, size_);
    if (!f.Check(fpvhip_predict_planes_submit(f.ctx, 0, 1, with_delta), "fpvhip_predict_planes_submit") ||
        !f.Check(fpvhip_predict_wait(f.ctx, 0, &gh, &gl, &gp, &gf), "fpvhip_predict_wait"))
      return;
    memcpy(high_.data(), gh, size_);
    if (low_.size() == size_) memcpy(low_.data(), gl, size_);
    preview_.assign(gp, gp + npv);
    flags_ = (uint8_t)((flags_ & FrameFlags::NO_LOW_BYTES) |
                       (gf[0] & (FrameFlags::USE_DELTA | FrameFlags::USE_CG)));
  }
  pixels_.reset();
  content_id_ = NewContentId();
  state_ &= ~FrameState::RAW;
  state_ |= FrameState::PREVIEW_GENERATED | FrameState::CG_PREDICTED;
  if (with_delta) state_ |= FrameState::DELTA_PREDICTED;
}

// cc:738-745 + cc:643-688
void Frame::Compress(Frame& delta_frame) {
  if (state_ & FrameState::COMPRESSED) return;
  Predict(delta_frame);
  if (!(state_ & FrameState::CG_PREDICTED)) return;  // a latched GPU failure: nothing to compress (Frame::LastError)
  std::vector<uint8_t> out;
  if (!BrotliAppend(high_.data(), size_, &out)) { FrameFail("Brotli", "high plane"); return; }
  high_.swap(out);
  if (flags_ & FrameFlags::NO_LOW_BYTES) {
    low_.clear();
  } else {
    out.clear();
    if (!BrotliAppend(low_.data(), size_, &out)) { FrameFail("Brotli", "low plane"); return; }
    low_.swap(out);
  }
  if (state_ & FrameState::PREVIEW_GENERATED) {
    out.clear();
    if (!BrotliAppend(preview_.data(), preview_.size(), &out)) { FrameFail("Brotli", "preview"); return; }
    preview_.swap(out);
  }
  state_ &= ~FrameState::RAW;
  state_ |= FrameState::COMPRESSED;
  content_id_ = NewContentId();
}

// cc:787-818 and cc:690-728
void Frame::CompressPredicted(size_t* encoded_high_size, uint8_t* encoded_high_buffer,
                              size_t* encoded_low_size, uint8_t* encoded_low_buffer,
                              size_t* encoded_preview_size, uint8_t* encoded_preview_buffer,
                              bool parallel) {
  auto copy_out = [](const std::vector<uint8_t>& v, size_t* sz, uint8_t* buf) {
    if (buf && *sz >= v.size()) {
      memcpy(buf, v.data(), v.size());
      *sz = v.size();
    } else {
      *sz = 0;
    }
  };
  if (state_ & FrameState::COMPRESSED) {
    copy_out(high_, encoded_high_size, encoded_high_buffer);
    copy_out(low_, encoded_low_size, encoded_low_buffer);
    copy_out(preview_, encoded_preview_size, encoded_preview_buffer);
    return;
  }
  if (PlanesPending()) SplitNow();
  auto brotli = [](const uint8_t* in, size_t n, size_t* sz, uint8_t* buf) {
    if (!BrotliCompress(in, n, buf, sz)) *sz = 0;  // (upstream ignores the result too; a failed stream has size 0)
  };
  std::future<void> low_task;
  if (!encoded_low_buffer || (flags_ & FrameFlags::NO_LOW_BYTES)) {
    *encoded_low_size = 0;
  } else if (parallel) {  // the low plane is the expensive one: overlap it (cc:697-704)
    low_task = std::async(std::launch::async, [&] { brotli(low_.data(), size_, encoded_low_size, encoded_low_buffer); });
  } else {
    brotli(low_.data(), size_, encoded_low_size, encoded_low_buffer);
  }
  if (encoded_high_buffer) brotli(high_.data(), size_, encoded_high_size, encoded_high_buffer);
  else *encoded_high_size = 0;
  if (encoded_preview_buffer && (state_ & FrameState::PREVIEW_GENERATED))
    brotli(preview_.data(), preview_.size(), encoded_preview_size, encoded_preview_buffer);
  else *encoded_preview_size = 0;
  if (low_task.valid()) low_task.wait();
}

// cc:747-775 with the two Optionally-Unapply steps (cc:612-641, cc:595-610)
void Frame::Uncompress(Frame& delta_frame) {
  if (PlanesPending()) SplitNow();
  if (delta_frame.PlanesPending()) delta_frame.SplitNow();
  content_id_ = NewContentId();
  if (state_ & FrameState::COMPRESSED) {
    if (!high_.empty()) BrotliToVector(&high_);
    if (!(low_.empty() || (flags_ & FrameFlags::NO_LOW_BYTES))) BrotliToVector(&low_);
    if ((state_ & FrameState::PREVIEW_GENERATED) && !preview_.empty()) BrotliToVector(&preview_);
    state_ &= ~FrameState::COMPRESSED;
  }
  const bool undo_cg = (state_ & FrameState::CG_PREDICTED) && (flags_ & FrameFlags::USE_CG);
  const bool undo_delta = (state_ & FrameState::DELTA_PREDICTED) && (flags_ & FrameFlags::USE_DELTA) &&
                          delta_frame.state() != FrameState::EMPTY;
  const bool planes_ok = high_.size() == size_ && size_ > 0;
  if (FrameFailed() && (undo_cg || undo_delta)) return;  // latched: the predictions stay applied, the state says so
  if (planes_ok && (undo_cg || undo_delta)) {
    Inv v(xsize_, ysize_);
    if (!v.ctx) return;
    if (undo_delta &&
        !v.Check(fpvhip_set_delta_planes(v.ctx, delta_frame.high_.data(),
                                         delta_frame.low_.size() == size_ ? delta_frame.low_.data() : nullptr),
                 "fpvhip_set_delta_planes"))
      return;
    const bool has_low = low_.size() == size_;
    uint8_t *sh = fpvhip_stage_plane(v.ctx, 0, 0, 0), *sl = fpvhip_stage_plane(v.ctx, 0, 0, 1);
    if (!sh || !sl) { v.Check(-1, "fpvhip_stage_plane"); return; }
    memcpy(sh, high_.data(), size_);
    if (has_low) memcpy(sl, low_.data(), size_);
    const uint8_t f = (uint8_t)((undo_delta ? 1 : 0) | (undo_cg ? 2 : 0) | (has_low ? 0 : 4));
    const uint8_t *gh, *gl;
    if (!v.Check(fpvhip_inverse_planes_submit(v.ctx, 0, 1, &f), "fpvhip_inverse_planes_submit") ||
        !v.Check(fpvhip_predict_wait(v.ctx, 0, &gh, &gl, nullptr, nullptr), "fpvhip_predict_wait"))
      return;
    memcpy(high_.data(), gh, size_);
    if (has_low) memcpy(low_.data(), gl, size_);
  }
  if (undo_cg) {
    const size_t pw = xsize_ / 4, ph = ysize_ / 4;
    if ((state_ & FrameState::PREVIEW_GENERATED) && preview_.size() == size_ / 16 && pw && ph &&
        pw * ph == preview_.size()) {
      Inv v(pw, ph);
      uint8_t* sp = v.ctx ? fpvhip_stage_plane(v.ctx, 0, 0, 0) : nullptr;
      const uint8_t f = 2 | 4;
      if (!sp) {
        if (v.ctx) v.Check(-1, "fpvhip_stage_plane");
      } else if ((pw * ph) % 4 == 0) {
        memcpy(sp, preview_.data(), preview_.size());
        const uint8_t* gh;
        if (v.Check(fpvhip_inverse_planes_submit(v.ctx, 0, 1, &f), "fpvhip_inverse_planes_submit") &&
            v.Check(fpvhip_predict_wait(v.ctx, 0, &gh, nullptr, nullptr, nullptr), "fpvhip_predict_wait"))
          memcpy(preview_.data(), gh, preview_.size());
      } else {  // odd preview size: pixels out, keep the high bytes
        memcpy(sp, preview_.data(), preview_.size());
        const uint16_t* px;
        if (v.Check(fpvhip_inverse_submit(v.ctx, 0, 1, &f), "fpvhip_inverse_submit") &&
            v.Check(fpvhip_inverse_wait(v.ctx, 0, &px), "fpvhip_inverse_wait"))
          for (size_t i = 0; i < preview_.size(); i++) preview_[i] = (uint8_t)(px[i] >> 8);
      }
      if (FrameFailed()) return;  // the planes are back, the preview is not: leave the state as it is
    }
    flags_ &= ~FrameFlags::USE_CG;
    state_ &= ~FrameState::CG_PREDICTED;
    if (state_ < FrameState::DELTA_PREDICTED) state_ |= FrameState::RAW;
  }
  if (undo_delta) {
    flags_ &= ~FrameFlags::USE_DELTA;
    state_ &= ~FrameState::DELTA_PREDICTED;
    if (state_ < FrameState::DELTA_PREDICTED) state_ |= FrameState::RAW;
  }
}

// cc:820-828
void Frame::OutputCore(std::vector<uint8_t>* out) {
  if (!(state_ & FrameState::COMPRESSED)) return;
  out->reserve(out->size() + 1 + high_.size() + low_.size());
  out->push_back(flags_);
  out->insert(out->end(), low_.begin(), low_.end());
  out->insert(out->end(), high_.begin(), high_.end());
}

// cc:830-846
void Frame::OutputFull(std::vector<uint8_t>* out) {
  if (!(state_ & FrameState::COMPRESSED)) return;
  const size_t total = (9 + 1 + preview_.size()) + (1 + high_.size() + low_.size());
  out->reserve(out->size() + total);
  Append32((uint32_t)total, out);
  out->push_back(0);
  Append32((uint32_t)(preview_.size() + 1), out);
  out->push_back((uint8_t)((flags_ & FrameFlags::USE_CG) | FrameFlags::NO_LOW_BYTES));
  out->insert(out->end(), preview_.begin(), preview_.end());
  OutputCore(out);
}

}  // namespace fpvc
