// fpv_columnar.cc -- the columnar batch adapter (include/columnar_batch/*.h) on the
// MI355X path.  Upstream's adapter (columnar_batch/ of the reference, cb/...) drives
// one Frame::Predict / Frame::Uncompress per frame from std::async tasks; here a batch
// is the unit of GPU work in both directions:
//
//   encoder  PushFrame stages pixels into the pinned slot of the open batch; a full
//            batch is ONE fpvhip_predict_submit (H2D, predictor kernels, D2H); a worker
//            thread then Brotli-compresses the planes of its frames in parallel and
//            appends them to the Batch columns in frame order
//            (cb/columnar_batch_encoder.cc:24-57, 93-123; cb/columnar_batch.cc:65-87)
//   decoder  the streams of a batch are Brotli-decoded in parallel straight into pinned
//            staging, ONE fpvhip_inverse_submit inverts them, images go to the image
//            processor in order (cb/columnar_batch_decoder.cc:51-95;
//            cb/columnar_batch.cc:89-122)
//
// Everything that is not per-pixel work -- the column layout, queues, futures -- is
// plain host code written for this design (no member-order race: threads start last).
#include <brotli/decode.h>
#include <brotli/encode.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "../../include/columnar_batch/columnar_batch.h"
#include "../../include/columnar_batch/columnar_batch_decoder.h"
#include "../../include/columnar_batch/columnar_batch_encoder.h"
#include "fpv_hip.h"
#include <stdexcept>

#include "fpv_host_internal.h"

namespace fpvc {
namespace columnarbatch {

using namespace fpvc::internal;

namespace {
size_t RoundUp64(size_t v) { return (v + 63) & ~(size_t)63; }

size_t WorkerThreads() {
  return EnvSize("FPV_COLUMNAR_THREADS", std::max(1u, std::min(16u, std::thread::hardware_concurrency())));
}
}  // namespace

// ============================================================= BatchSchema ==

// cb/columnar_batch.cc:6-24: keeps a copy of the raw delta frame and its two planes as
// Brotli streams (no preview, no clamped gradient).
BatchSchema::BatchSchema(size_t xsize, size_t ysize, size_t shifted_left, Frame& uncompressed_delta_frame)
    : delta_(uncompressed_delta_frame), width_(xsize), height_(ysize), left_shift_(shifted_left) {
  const size_t cap = Frame::MaxCompressedPlaneSize(xsize, ysize);
  size_t sizes[2] = {cap, cap}, no_preview = 0;
  for (auto& v : delta_streams_) v.resize(cap);
  uncompressed_delta_frame.CompressPredicted(&sizes[0], delta_streams_[0].data(), &sizes[1], delta_streams_[1].data(),
                                             &no_preview, nullptr);
  for (int k = 0; k < 2; k++) delta_streams_[k].resize(sizes[k]);
}

// =================================================================== Image ==

Image::Image(int64_t timestamp, size_t xsize, size_t ysize, uint8_t bpp, Type type, std::vector<uint8_t>&& data)
    : bytes_(std::move(data)), when_(timestamp), width_(xsize), height_(ysize), kind_(type), bits_(bpp) {}

// =================================================================== Batch ==

Batch::Batch(size_t batch_size, SchemaPtr schema) : schema_(schema), rows_(batch_size), count_(0) {
  const size_t previews_capacity_ = RoundUp64(batch_size * Frame::MaxCompressedPreviewSize(schema->xsize(), schema->ysize()));
  const size_t plane_capacity_ = RoundUp64(batch_size * Frame::MaxCompressedPlaneSize(schema->xsize(), schema->ysize()));
  const size_t ts_bytes = RoundUp64(batch_size * sizeof(int64_t));
  const size_t flag_bytes = RoundUp64(batch_size);
  const size_t off_bytes = RoundUp64((batch_size + 1) * sizeof(uint32_t));
  capacity_[0] = previews_capacity_;
  capacity_[1] = capacity_[2] = plane_capacity_;
  // 64 spare bytes: the vector's storage itself is only 16-byte aligned
  storage_.assign(ts_bytes + flag_bytes + 3 * off_bytes + previews_capacity_ + 2 * plane_capacity_ + 64, 0);
  uint8_t* p = storage_.data();
  p += (64 - reinterpret_cast<uintptr_t>(p) % 64) % 64;
  stamps_ = reinterpret_cast<int64_t*>(p);
  p += ts_bytes;
  flag_col_ = p;
  p += flag_bytes;
  for (int c = 0; c < 3; c++) {
    offsets_[c] = reinterpret_cast<uint32_t*>(p);
    p += off_bytes;
  }
  for (int c = 0; c < 3; c++) {
    columns_[c] = p;
    p += capacity_[c];
  }
  Reset();
}

void Batch::Reset() {
  count_ = 0;
  for (int c = 0; c < 3; c++) offsets_[c][0] = 0;
}

bool Batch::AppendCompressed(int64_t timestamp, uint8_t flags, const uint8_t* high, size_t high_size,
                             const uint8_t* low, size_t low_size, const uint8_t* preview, size_t preview_size) {
  if (count_ >= rows_) return false;
  const uint8_t* src[3] = {preview, high, low};
  const size_t n[3] = {preview_size, high_size, low_size};
  for (int c = 0; c < 3; c++)
    if (offsets_[c][count_] + n[c] > capacity_[c]) return false;
  stamps_[count_] = timestamp;
  flag_col_[count_] = flags;
  for (int c = 0; c < 3; c++) {
    if (n[c]) memcpy(columns_[c] + offsets_[c][count_], src[c], n[c]);
    offsets_[c][count_ + 1] = offsets_[c][count_] + (uint32_t)n[c];
  }
  count_++;
  return true;
}

// cb/columnar_batch.cc:65-87
bool Batch::AppendPredicted(Frame predicted_frame) {
  if (count_ >= rows_) return false;
  size_t n[3];
  for (int c = 0; c < 3; c++) n[c] = capacity_[c] - offsets_[c][count_];
  predicted_frame.CompressPredicted(&n[1], columns_[1] + offsets_[1][count_], &n[2], columns_[2] + offsets_[2][count_],
                                    &n[0], columns_[0] + offsets_[0][count_]);
  stamps_[count_] = predicted_frame.timestamp();
  flag_col_[count_] = predicted_frame.flags();
  for (int c = 0; c < 3; c++) offsets_[c][count_ + 1] = offsets_[c][count_] + (uint32_t)n[c];
  count_++;
  return true;
}

const uint8_t* Batch::stream(size_t index, int which, size_t* size) const {
  *size = offsets_[which][index + 1] - offsets_[which][index];
  return columns_[which] + offsets_[which][index];
}

// cb/columnar_batch.cc:89-122: one image on its own, through Frame::Uncompress.
Image Batch::ExtractImage(size_t index, Image::Type type) {
  size_t n;
  std::vector<uint8_t> high, low, preview;
  uint8_t flags = flag_col_[index];
  uint8_t state = FrameState::COMPRESSED | FrameState::DELTA_PREDICTED | FrameState::CG_PREDICTED;
  if (type == Image::PREVIEW) {
    const uint8_t* p = stream(index, 0, &n);
    preview.assign(p, p + n);
    if (!preview.empty()) state |= FrameState::PREVIEW_GENERATED;
  } else {
    const uint8_t* p = stream(index, 1, &n);
    high.assign(p, p + n);
    if (type == Image::FULL) {
      p = stream(index, 2, &n);
      low.assign(p, p + n);
    } else {
      flags |= FrameFlags::NO_LOW_BYTES;
    }
  }
  Frame frame(schema_->xsize(), schema_->ysize(), flags, state, std::move(high), std::move(low), std::move(preview),
              stamps_[index]);
  frame.Uncompress(schema_->delta_frame());
  if (type == Image::PREVIEW)
    return Image(frame.timestamp(), frame.xsize() / 4, frame.ysize() / 4, 8, type, frame.MoveOutPreview());
  if (type == Image::MSB8) return Image(frame.timestamp(), frame.xsize(), frame.ysize(), 8, type, frame.MoveOutHigh());
  const size_t npix = frame.xsize() * frame.ysize();
  std::vector<uint8_t> data(2 * npix);
  const std::vector<uint8_t>&h = frame.high(), &l = frame.low();
  for (size_t i = 0; i < npix; i++) {
    data[2 * i] = i < l.size() ? l[i] : 0;
    data[2 * i + 1] = h[i];
  }
  return Image(frame.timestamp(), frame.xsize(), frame.ysize(), (uint8_t)(16 - schema_->shiftedLeft()), type,
               std::move(data));
}

// ==================================================== ColumnarBatchEncoder ==

struct ColumnarBatchEncoder::Impl {
  struct Pending {  // a batch on the GPU
    uint32_t slot;
    std::vector<int64_t> ts;
    uint64_t seq;  // batches leave in the order they were submitted in
  };
  const size_t xs, ys, npix, npv;
  const int shift;
  const bool big_endian;
  const size_t per_batch;
  BatchProcessor processor;

  fpvhip_ctx* ctx = nullptr;
  SchemaPtr schema;
  // First GPU (or Brotli) failure of this encoder: latched.  The failing PushFrame's future
  // carries it as an exception, later PushFrame calls return an invalid future (as after
  // Close), batches in flight are dropped, Close still answers with the last timestamp that was
  // stored before the failure.  The reference's convention (cb/columnar_batch_encoder.cc, cc:160-181)
  // is to report, never to abort.
  ErrorLatch latch;

  std::mutex m;
  std::condition_variable cv;
  bool closing = false, closed = false;
  static constexpr int kSlots = 3, kWorkers = 2;  // one slot filling, two batches in the workers
  char slot_busy[kSlots] = {0, 0, 0};
  uint64_t next_seq = 0, next_emit = 0;
  int workers_left = kWorkers;
  int open_slot = -1;            // slot being filled by PushFrame
  std::vector<int64_t> open_ts;  // timestamps of the frames staged in it
  std::deque<Pending> submitted;
  std::vector<BatchPtr> spare;   // returned by ReturnProcessedBatch
  int64_t latest_stored = -1;
  std::promise<int64_t> closing_promise;
  std::shared_future<int64_t> closing_future;
  std::thread workers[kWorkers];

  Impl(size_t x, size_t y, int s, bool be, BatchProcessor p, int n)
      : xs(x), ys(y), npix(x * y), npv((x / 4) * (y / 4)), shift(s), big_endian(be),
        per_batch((size_t)std::max(1, n)), processor(std::move(p)), closing_future(closing_promise.get_future()) {
    const std::vector<int> devs = DeviceList();
    if (devs.empty()) {
      latch.Fail("ColumnarBatchEncoder", "no usable GPU", fpvhip_last_error(nullptr));
    } else {
      ctx = AcquireEncoderCtx(devs[0], (uint32_t)xs, (uint32_t)ys, shift, big_endian, (uint32_t)per_batch, kSlots);
      if (!ctx) latch.Fail("ColumnarBatchEncoder", "fpvhip_create", fpvhip_last_error(nullptr));
    }
    for (auto& w : workers) w = std::thread(&Impl::Work, this);  // last: everything they touch exists
  }

  ~Impl() {
    for (auto& w : workers)
      if (w.joinable()) w.join();
    if (ctx && latch.failed()) DiscardEncoderCtx(ctx);  // a context that saw a failure is not cached
    else if (ctx) ReleaseEncoderCtx(ctx);
  }

  bool Fail(const char* what, const char* detail) {  // m NOT held
    latch.Fail("ColumnarBatchEncoder", what, detail);
    cv.notify_all();
    return false;
  }

  bool SubmitOpenLocked() {  // m held, open_slot has frames
    if (latch.failed() || fpvhip_predict_submit(ctx, (uint32_t)open_slot, (uint32_t)open_ts.size(), 1)) {
      if (!latch.failed()) latch.Fail("ColumnarBatchEncoder", "fpvhip_predict_submit", fpvhip_last_error(ctx));
      slot_busy[open_slot] = 0;  // the frames of this batch are dropped
      open_ts.clear();
      open_slot = -1;
      cv.notify_all();
      return false;
    }
    submitted.push_back(Pending{(uint32_t)open_slot, std::move(open_ts), next_seq++});
    open_ts.clear();
    open_slot = -1;
    cv.notify_all();
    return true;
  }

  BatchPtr TakeBatch() {
    std::lock_guard<std::mutex> l(m);
    if (!spare.empty()) {
      BatchPtr b = spare.back();
      spare.pop_back();
      return b;
    }
    return std::make_shared<Batch>(per_batch, schema);
  }

  void Work() {
    const size_t threads = WorkerThreads();
    for (;;) {
      Pending p;
      {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return !submitted.empty() || closing; });
        if (submitted.empty()) break;  // closing and drained
        p = std::move(submitted.front());
        submitted.pop_front();
      }
      const uint8_t *high, *low, *pv, *flags;
      bool ok = !latch.failed();
      if (ok && fpvhip_predict_wait(ctx, p.slot, &high, &low, &pv, &flags)) ok = Fail("fpvhip_predict_wait", fpvhip_last_error(ctx));
      if (!ok) {  // dropped: free the slot, keep the emission order moving
        std::unique_lock<std::mutex> l(m);
        slot_busy[p.slot] = 0;
        cv.notify_all();
        cv.wait(l, [&] { return next_emit == p.seq; });
        next_emit++;
        cv.notify_all();
        continue;
      }
      const size_t n = p.ts.size();
      // Brotli of the 3 planes of every frame, frames in parallel (cc:643-688 per plane)
      std::vector<std::vector<uint8_t>> sh(n), sl(n), sp(n);
      // One task per plane, so that a batch of few frames still spreads over all threads;
      // frame-major order, so that low, high and preview planes are in progress at the
      // same time (sixteen threads all on low planes -- the incompressible, memory-heavy
      // ones -- measured ~20 % slower in the Encoder).
      std::atomic<bool> brotli_ok{true};
      ParallelFor(3 * n, threads, [&](size_t t) {
        const size_t k = t / 3, which = t % 3;
        if (which == 0) {
          if (!(flags[k] & NO_LOW_BYTES) && !BrotliAppend(low + k * npix, npix, &sl[k])) brotli_ok = false;
        } else if (which == 1) {
          if (!BrotliAppend(high + k * npix, npix, &sh[k])) brotli_ok = false;
        } else if (!BrotliAppend(pv + k * npv, npv, &sp[k])) {
          brotli_ok = false;
        }
      });
      if (!brotli_ok) ok = Fail("Brotli", "compressing a plane failed");
      BatchPtr batch = ok ? TakeBatch() : nullptr;
      for (size_t k = 0; ok && k < n; k++)
        if (!batch->AppendCompressed(p.ts[k], flags[k], sh[k].data(), sh[k].size(), sl[k].data(), sl[k].size(),
                                     sp[k].data(), sp[k].size()))
          ok = Fail("ColumnarBatchEncoder", "batch column overflow");
      {
        std::unique_lock<std::mutex> l(m);
        slot_busy[p.slot] = 0;  // the pinned planes have been consumed
        cv.notify_all();
        // two workers overlap the Brotli of one batch with the column copies and the
        // processor call of the one before; batches still leave in submission order
        cv.wait(l, [&] { return next_emit == p.seq; });
        // Every batch in front of this one has now been stored or has latched its failure: one that failed while this one
        // was being compressed (the other worker's) makes this one a batch BEHIND the failure -- dropped, so that what the
        // processor has seen stays a clean prefix (under load the other worker's failing wait can return after this worker's
        // successful one).
        if (ok && latch.failed()) ok = false;
        if (ok) latest_stored = batch->LatestTimestamp();
      }
      if (ok) processor(batch);  // one batch at a time, in order
      {
        std::lock_guard<std::mutex> l(m);
        next_emit++;
      }
      cv.notify_all();
    }
    // cb/columnar_batch_encoder.cc:82-91: nothing left to flush -> a null batch.  The
    // last worker to leave closes the stream.
    bool last_out, announce_empty;
    int64_t last;
    {
      std::lock_guard<std::mutex> l(m);
      last_out = --workers_left == 0;
      announce_empty = !flushed_partial;
      last = latest_stored;
    }
    if (!last_out) return;
    if (announce_empty) processor(nullptr);
    closing_promise.set_value(last);
  }
  bool flushed_partial = false;  // Close submitted a partial batch
};

ColumnarBatchEncoder::ColumnarBatchEncoder(size_t xsize, size_t ysize, int shift_to_left_align, bool big_endian,
                                           BatchProcessor batch_processor, int frames_per_batch)
    : impl_(new Impl(xsize, ysize, shift_to_left_align, big_endian, std::move(batch_processor), frames_per_batch)) {}

ColumnarBatchEncoder::~ColumnarBatchEncoder() {
  Close();
  impl_.reset();
}

std::string ColumnarBatchEncoder::LastError() const { return impl_->latch.text(); }

namespace {
std::future<void*> FailedFuture(const std::string& why) {  // the failure, delivered through the future
  std::promise<void*> p;
  p.set_exception(std::make_exception_ptr(std::runtime_error(why)));
  return p.get_future();
}
}  // namespace

std::future<void*> ColumnarBatchEncoder::PushFrame(uint64_t timestamp, uint16_t* frame, void* info) {
  Impl& s = *impl_;
  std::unique_lock<std::mutex> l(s.m);
  if (s.closing || s.latch.failed()) return std::future<void*>();
  if (!s.schema) {  // the first frame is the delta frame of the stream (and an ordinary frame too)
    l.unlock();
    Frame delta(s.xs, s.ys, frame, s.shift, s.big_endian, (int64_t)timestamp);
    SchemaPtr schema = std::make_shared<BatchSchema>(s.xs, s.ys, (size_t)s.shift, delta);
    if (fpvhip_set_delta_raw(s.ctx, frame)) {
      s.Fail("fpvhip_set_delta_raw", fpvhip_last_error(s.ctx));
      return FailedFuture(s.latch.text());
    }
    l.lock();
    s.schema = schema;
  }
  if (s.open_slot < 0) {
    s.cv.wait(l, [&] { return !s.slot_busy[0] || !s.slot_busy[1] || !s.slot_busy[2] || s.latch.failed(); });
    if (s.latch.failed()) return FailedFuture(s.latch.text());  // a batch in flight failed while this call waited for a slot
    s.open_slot = !s.slot_busy[0] ? 0 : !s.slot_busy[1] ? 1 : 2;
    s.slot_busy[s.open_slot] = 1;
  }
  const int slot = s.open_slot;
  const size_t k = s.open_ts.size();
  l.unlock();
  uint16_t* dst = fpvhip_stage_in(s.ctx, (uint32_t)slot, (uint32_t)k);
  if (!dst) {
    s.Fail("fpvhip_stage_in", fpvhip_last_error(s.ctx));
    return FailedFuture(s.latch.text());
  }
  memcpy(dst, frame, s.npix * 2);  // the caller's buffer is free from here on
  l.lock();
  s.open_ts.push_back((int64_t)timestamp);
  if (s.open_ts.size() == s.per_batch && !s.SubmitOpenLocked()) {
    l.unlock();
    return FailedFuture(s.latch.text());
  }
  l.unlock();
  std::promise<void*> done;
  done.set_value(info);
  return done.get_future();
}

void ColumnarBatchEncoder::ReturnProcessedBatch(BatchPtr processed) {
  if (!processed) return;
  processed->Reset();
  std::lock_guard<std::mutex> l(impl_->m);
  impl_->spare.push_back(processed);
}

std::shared_future<int64_t> ColumnarBatchEncoder::Close() {
  Impl& s = *impl_;
  {
    std::lock_guard<std::mutex> l(s.m);
    if (!s.closed) {
      s.closed = true;
      if (s.open_slot >= 0 && !s.open_ts.empty()) s.flushed_partial = s.SubmitOpenLocked();
      s.closing = true;
    }
  }
  s.cv.notify_all();
  return s.closing_future;
}

// ==================================================== ColumnarBatchDecoder ==

struct ColumnarBatchDecoder::Impl {
  struct Job {
    BatchPtr batch;
    std::promise<BatchPtr> done;
  };
  const Image::Type type;
  const bool unshift;
  ImageProcessor processor;

  std::mutex m;
  std::condition_variable cv;
  bool closing = false, closed = false;
  std::deque<Job> queue;
  SchemaPtr schema;
  std::shared_ptr<InverseEngine> engine;  // full-size or preview geometry, created with the first batch
  ErrorLatch latch;  // first failure: that batch's future carries it, later PushBatch calls return an invalid future
  int64_t latest = -1;
  std::promise<int64_t> closing_promise;
  std::shared_future<int64_t> closing_future;
  std::thread worker;

  Impl(Image::Type t, bool u, ImageProcessor p)
      : type(t), unshift(u), processor(std::move(p)), closing_future(closing_promise.get_future()) {
    worker = std::thread(&Impl::Work, this);  // last: everything it touches exists
  }
  ~Impl() {
    if (worker.joinable()) worker.join();
  }

  bool Prepare(const Batch& batch) {
    if (engine) return true;
    BatchSchema& sc = *batch.schema();
    const bool pv = type == Image::PREVIEW;
    const size_t xs = pv ? sc.xsize() / 4 : sc.xsize(), ys = pv ? sc.ysize() / 4 : sc.ysize();
    const uint32_t per = (uint32_t)std::max<size_t>(1, batch.length());
    const std::vector<int> devs = DeviceList();
    if (devs.empty()) return false;
    engine = InverseEngine::Create(xs, ys, std::max(per, 1u), 2, devs[0]);
    if (!engine) return false;
    capacity = per;
    if (!pv) {  // decoded delta frame = the schema's raw planes (cb/columnar_batch.cc:111)
      Frame& d = sc.delta_frame();
      const std::vector<uint8_t>&h = d.high(), &l = d.low();
      std::vector<uint16_t> delta16(xs * ys);
      for (size_t i = 0; i < delta16.size(); i++)
        delta16[i] = (uint16_t)((h[i] << 8) | (i < l.size() ? l[i] : 0));
      if (!engine->SetDelta(delta16.data())) return false;
    }
    return true;
  }
  uint32_t capacity = 0;

  bool DecodeBatch(Batch& batch) {
    const size_t n = batch.length();
    if (n == 0) return true;
    if (latch.failed()) return false;
    if (!Prepare(batch)) return latch.Fail("ColumnarBatchDecoder", "no GPU decoder", fpvhip_last_error(nullptr));
    BatchSchema& sc = *batch.schema();
    const bool pv = type == Image::PREVIEW;
    const size_t xs = pv ? sc.xsize() / 4 : sc.xsize(), ys = pv ? sc.ysize() / 4 : sc.ysize(), npix = xs * ys;
    const int shr = (unshift && type == Image::FULL) ? (int)sc.shiftedLeft() : 0;
    std::vector<InverseEngine::PlaneStreams> in(n);
    for (size_t i = 0; i < n; i++) {
      InverseEngine::PlaneStreams& p = in[i];
      const uint8_t f = batch.flags(i);
      if (pv) {  // the preview is its own little image: CG flag only, no low plane, no delta
        p.high = batch.stream(i, 0, &p.high_size);
        p.flags = (uint8_t)((f & USE_CG) | NO_LOW_BYTES);
      } else {
        p.high = batch.stream(i, 1, &p.high_size);
        p.low = batch.stream(i, 2, &p.low_size);
        p.flags = type == Image::MSB8 ? (uint8_t)(f | NO_LOW_BYTES) : f;
        if (p.flags & NO_LOW_BYTES) p.low = nullptr, p.low_size = 0;
      }
    }
    const size_t threads = WorkerThreads();
    const bool ok = engine->DecodePlanesBatch(in.data(), n, [&](size_t i, const uint16_t* px) {
      std::vector<uint8_t> data;
      uint8_t bpp = 8;
      if (type == Image::FULL) {
        data.resize(2 * npix);
        uint16_t* out = reinterpret_cast<uint16_t*>(data.data());
        if (shr > 0) for (size_t k = 0; k < npix; k++) out[k] = (uint16_t)(px[k] >> shr);
        else memcpy(out, px, 2 * npix);
        bpp = (uint8_t)(16 - sc.shiftedLeft());
      } else {  // PREVIEW, MSB8: the high bytes
        data.resize(npix);
        for (size_t k = 0; k < npix; k++) data[k] = (uint8_t)(px[k] >> 8);
      }
      processor(Image(batch.timestamp(i), xs, ys, bpp, type, std::move(data)));
    }, threads);
    if (!ok) return latch.Fail("ColumnarBatchDecoder", "decoding a batch failed", "corrupt streams or a GPU failure");
    return true;
  }

  void Work() {
    for (;;) {
      Job job;
      {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return !queue.empty() || closing; });
        if (queue.empty()) break;
        job = std::move(queue.front());
        queue.pop_front();
      }
      if (!DecodeBatch(*job.batch)) {  // through the channel the signature has: the future
        job.done.set_exception(std::make_exception_ptr(std::runtime_error(latch.text())));
        continue;
      }
      {
        std::lock_guard<std::mutex> l(m);
        if (!job.batch->Empty()) latest = job.batch->LatestTimestamp();
      }
      job.done.set_value(job.batch);
    }
    closing_promise.set_value(latest);
  }
};

ColumnarBatchDecoder::ColumnarBatchDecoder(Image::Type type, bool unshift, ImageProcessor image_processor)
    : impl_(new Impl(type, unshift, std::move(image_processor))) {}

ColumnarBatchDecoder::~ColumnarBatchDecoder() {
  Close();
  impl_.reset();
}

std::future<BatchPtr> ColumnarBatchDecoder::PushBatch(BatchPtr batch) {
  Impl& s = *impl_;
  std::future<BatchPtr> fut;
  {
    std::lock_guard<std::mutex> l(s.m);
    if (!batch || s.closing || s.latch.failed()) return std::future<BatchPtr>();
    if (!s.schema) s.schema = batch->schema();
    if (s.schema.get() != batch->schema().get()) return std::future<BatchPtr>();
    Impl::Job job;
    job.batch = batch;
    fut = job.done.get_future();
    s.queue.push_back(std::move(job));
  }
  s.cv.notify_all();
  return fut;
}

std::string ColumnarBatchDecoder::LastError() const { return impl_->latch.text(); }

std::shared_future<int64_t> ColumnarBatchDecoder::Close() {
  Impl& s = *impl_;
  {
    std::lock_guard<std::mutex> l(s.m);
    s.closing = true;
  }
  s.cv.notify_all();
  return s.closing_future;
}

}  // namespace columnarbatch
}  // namespace fpvc
