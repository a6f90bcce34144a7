// fpv_decoders.cc -- StreamingDecoder and RandomAccessDecoder (container
// parsing on the host, cc:866-1070) over the GPU inverse predictor, plus the
// process-wide forward-context cache.
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>

#include "fpv_host_internal.h"
#include "fusion_power_video.h"

namespace fpvc {
namespace internal {

// ------------------------------------------------------- context cache ------

namespace {
struct PoolEntry {
  int device;
  uint32_t xs, ys, batch, slots;
  int shift;
  bool big_endian, in_use;
  fpvhip_ctx* ctx;
};
std::mutex g_pool_m;
std::vector<PoolEntry> g_pool;
constexpr size_t kMaxIdle = 8;
}  // namespace

fpvhip_ctx* AcquireEncoderCtx(int device, uint32_t xs, uint32_t ys, int shift,
                              bool big_endian, uint32_t batch, uint32_t slots) {
  {
    std::lock_guard<std::mutex> l(g_pool_m);
    for (auto& e : g_pool) {
      if (!e.in_use && e.device == device && e.xs == xs && e.ys == ys && e.shift == shift &&
          e.big_endian == big_endian && e.batch == batch && e.slots == slots) {
        e.in_use = true;
        return e.ctx;
      }
    }
  }
  WarmUpWait(kWarmForward);
  fpvhip_ctx* c = fpvhip_create(device, xs, ys, shift, big_endian ? 1 : 0, batch, slots);
  if (!c) return nullptr;
  std::lock_guard<std::mutex> l(g_pool_m);
  g_pool.push_back(PoolEntry{device, xs, ys, batch, slots, shift, big_endian, true, c});
  return c;
}

// Idle contexts stay cached for the next session of the same geometry, bounded by count and
// by the pinned staging memory they hold (FPV_POOL_MB, default 1024): beyond the byte bound
// the idle contexts first give back all slots but one (fpvhip_trim: stream syncs and
// unregistration, so it runs OUTSIDE the pool lock on contexts marked busy meanwhile), then the
// oldest are destroyed until the total is under the bound.  Whatever a reused context remembers
// (delta planes, prediction state) is overwritten by Init / harmless.
void ReleaseEncoderCtx(fpvhip_ctx* ctx) {
  std::vector<fpvhip_ctx*> drop, trim;
  const size_t cap = EnvSize("FPV_POOL_MB", 1024) << 20;
  {
    std::lock_guard<std::mutex> l(g_pool_m);
    size_t idle = 0, bytes = 0;
    for (auto& e : g_pool) {
      if (e.ctx == ctx) e.in_use = false;
      if (!e.in_use) { idle++; bytes += fpvhip_staging_bytes(e.ctx); }
    }
    for (size_t i = 0; i < g_pool.size() && idle > kMaxIdle;) {  // count bound: the oldest go
      if (!g_pool[i].in_use && g_pool[i].ctx != ctx) {
        bytes -= fpvhip_staging_bytes(g_pool[i].ctx);
        drop.push_back(g_pool[i].ctx);
        g_pool.erase(g_pool.begin() + i);
        idle--;
      } else {
        i++;
      }
    }
    if (bytes > cap)
      for (auto& e : g_pool)
        if (!e.in_use) {
          e.in_use = true;  // nobody may acquire it while it is being trimmed
          trim.push_back(e.ctx);
        }
  }
  for (fpvhip_ctx* c : trim) fpvhip_trim(c);
  if (!trim.empty()) {
    std::lock_guard<std::mutex> l(g_pool_m);
    size_t bytes = 0;
    for (auto& e : g_pool) {
      for (fpvhip_ctx* c : trim)
        if (e.ctx == c) e.in_use = false;
      if (!e.in_use) bytes += fpvhip_staging_bytes(e.ctx);
    }
    for (size_t i = 0; i < g_pool.size() && bytes > cap;) {  // still above the bound: the oldest idle contexts go
      if (!g_pool[i].in_use) {
        bytes -= fpvhip_staging_bytes(g_pool[i].ctx);
        drop.push_back(g_pool[i].ctx);
        g_pool.erase(g_pool.begin() + i);
      } else {
        i++;
      }
    }
  }
  for (fpvhip_ctx* c : drop) fpvhip_destroy(c);
}

void DiscardEncoderCtx(fpvhip_ctx* ctx) {
  if (!ctx) return;
  {
    std::lock_guard<std::mutex> l(g_pool_m);
    for (size_t i = 0; i < g_pool.size(); i++)
      if (g_pool[i].ctx == ctx) {
        g_pool.erase(g_pool.begin() + i);
        break;
      }
  }
  fpvhip_destroy(ctx);
}

fpvhip_ctx* AcquireDecoderCtx(int device, uint32_t xs, uint32_t ys) {
  {
    std::lock_guard<std::mutex> l(g_pool_m);
    for (auto& e : g_pool) {
      if (!e.in_use && e.device == device && e.xs == xs && e.ys == ys && e.shift == -1) {
        e.in_use = true;
        return e.ctx;
      }
    }
  }
  WarmUpWait(kWarmInverse);
  fpvhip_ctx* c = fpvhip_create_decoder(device, xs, ys, 1, 1);
  if (!c) return nullptr;
  std::lock_guard<std::mutex> l(g_pool_m);
  g_pool.push_back(PoolEntry{device, xs, ys, 1, 1, -1, false, true, c});
  return c;
}

void ReleaseDecoderCtx(fpvhip_ctx* ctx) { ReleaseEncoderCtx(ctx); }

// ------------------------------------------------------- inverse engine -----

std::shared_ptr<InverseEngine> InverseEngine::Create(size_t xs, size_t ys, uint32_t batch,
                                                     uint32_t slots, int device) {
  std::shared_ptr<InverseEngine> e(new InverseEngine());
  WarmUpWait(kWarmInverse);
  e->ctx_ = fpvhip_create_decoder(device, (uint32_t)xs, (uint32_t)ys, batch, slots);
  if (!e->ctx_) {
    fprintf(stderr, "fpv-mi355x: cannot create GPU decoder context: %s\n", fpvhip_last_error(nullptr));
    return nullptr;
  }
  e->npix_ = xs * ys;
  e->batch_ = batch;
  e->nslots_ = slots;
  e->busy_.assign(slots, 0);
  return e;
}

InverseEngine::~InverseEngine() { fpvhip_destroy(ctx_); }

bool InverseEngine::SetDelta(const uint16_t* delta16) {
  if (fpvhip_set_delta_decoded(ctx_, delta16)) {
    fprintf(stderr, "fpv-mi355x: %s\n", fpvhip_last_error(ctx_));
    return false;
  }
  has_delta_ = true;
  return true;
}

int InverseEngine::AcquireSlot() {
  std::unique_lock<std::mutex> l(m_);
  for (;;) {
    for (uint32_t s = 0; s < nslots_; s++)
      if (!busy_[s]) {
        busy_[s] = 1;
        return (int)s;
      }
    cv_.wait(l);
  }
}

void InverseEngine::Release(int slot) {
  {
    std::lock_guard<std::mutex> l(m_);
    busy_[slot] = 0;
  }
  cv_.notify_one();
}

// The part of DecompressImage in front of the pixel loops (cc:299-324): flags,
// the two Brotli streams, size checks.  Planes land in pinned staging memory.
bool InverseEngine::Unpack(const uint8_t* in, size_t size, uint32_t slot, uint32_t k,
                           uint8_t* flags) {
  size_t pos = 0, got = 0;
  if (pos >= size) return FPV_FAILURE("out of bounds");
  const uint8_t f = in[pos++];
  if ((f & USE_DELTA) && !has_delta_) return FPV_FAILURE("delta frame not given");
  if (!(f & NO_LOW_BYTES)) {
    uint8_t* low = fpvhip_stage_plane(ctx_, slot, k, 1);
    if (!BrotliInto(in, size, &pos, low, npix_, &got)) return FPV_FAILURE("brotli decompression failed");
    if (got != npix_) return FPV_FAILURE("wrong decompressed plane size");
  }
  uint8_t* high = fpvhip_stage_plane(ctx_, slot, k, 0);
  if (!BrotliInto(in, size, &pos, high, npix_, &got)) return FPV_FAILURE("brotli decompression failed");
  if (got != npix_) return FPV_FAILURE("wrong decompressed plane size");
  *flags = f;
  return true;
}

bool InverseEngine::Decode(const uint8_t* in, size_t size, const uint16_t** frame, int* slot) {
  const int s = AcquireSlot();
  uint8_t flags = 0;
  if (!Unpack(in, size, (uint32_t)s, 0, &flags)) {
    Release(s);
    return false;
  }
  if (fpvhip_inverse_submit(ctx_, (uint32_t)s, 1, &flags) ||
      fpvhip_inverse_wait(ctx_, (uint32_t)s, frame)) {
    Release(s);
    return FPV_FAILURE(fpvhip_last_error(ctx_));
  }
  *slot = s;
  return true;
}

bool InverseEngine::UnpackPlanes(const PlaneStreams& p, uint32_t slot, uint32_t k, uint8_t* flags) {
  size_t pos = 0, got = 0;
  uint8_t f = p.flags;
  if ((f & USE_DELTA) && !has_delta_) return FPV_FAILURE("delta frame not given");
  if (!p.low || p.low_size == 0) f |= NO_LOW_BYTES;
  if (!(f & NO_LOW_BYTES)) {
    if (!BrotliInto(p.low, p.low_size, &pos, fpvhip_stage_plane(ctx_, slot, k, 1), npix_, &got) || got != npix_)
      return FPV_FAILURE("brotli decompression of a low plane failed");
  }
  pos = 0;
  if (!BrotliInto(p.high, p.high_size, &pos, fpvhip_stage_plane(ctx_, slot, k, 0), npix_, &got) || got != npix_)
    return FPV_FAILURE("brotli decompression of a high plane failed");
  *flags = f;
  return true;
}

bool InverseEngine::DecodeBatch(const uint8_t* const* ins, const size_t* sizes, size_t n,
                                const Sink& sink, size_t threads, int unshift, bool big_endian) {
  return RunBatches(
      n, [&](size_t i, uint32_t slot, uint32_t k, uint8_t* flags) { return Unpack(ins[i], sizes[i], slot, k, flags); },
      sink, threads, unshift, big_endian);
}

bool InverseEngine::DecodePlanesBatch(const PlaneStreams* images, size_t n, const Sink& sink, size_t threads) {
  return RunBatches(
      n, [&](size_t i, uint32_t slot, uint32_t k, uint8_t* flags) { return UnpackPlanes(images[i], slot, k, flags); },
      sink, threads, -1, false);
}

bool InverseEngine::RunBatches(size_t n, const Unpacker& unpack, const Sink& sink, size_t threads, int unshift,
                               bool big_endian) {
  if (threads == 0) threads = 1;
  // One batched request at a time per engine: a request holds the previous batch's slot
  // while it fills the next one, so two concurrent requests on an engine with two slots
  // would each end up holding one and waiting for the other's.  Single-image Decode
  // calls never hold a slot while waiting for another and stay concurrent.
  std::lock_guard<std::mutex> run_lock(run_m_);
  // Two slots alternate: while the GPU inverts batch g, the host threads
  // Brotli-decode batch g+1 into the other slot.
  struct Flight { int slot = -1; size_t first = 0, count = 0; };
  Flight prev;
  std::vector<uint8_t> flags(batch_);
  std::vector<char> good(batch_);
  bool ok = true;
  auto finish = [&](Flight& f) {
    if (f.slot < 0) return;
    const uint16_t* px = nullptr;
    if (fpvhip_inverse_wait(ctx_, (uint32_t)f.slot, &px)) {
      ok = FPV_FAILURE(fpvhip_last_error(ctx_));
    } else if (ok) {
      for (size_t i = 0; i < f.count; i++) sink(f.first + i, px + i * npix_);
    }
    Release(f.slot);
    f.slot = -1;
  };
  for (size_t first = 0; first < n && ok; first += batch_) {
    const size_t count = std::min<size_t>(batch_, n - first);
    const int s = AcquireSlot();
    // A slot's pinned memory is allocated on first touch, and calls on a context
    // must not race: touch it here, before the worker threads use it.
    if (!fpvhip_stage_plane(ctx_, (uint32_t)s, 0, 0)) {
      Release(s);
      finish(prev);
      return FPV_FAILURE(fpvhip_last_error(ctx_));
    }
    ParallelFor(count, threads, [&](size_t i) {
      good[i] = unpack(first + i, (uint32_t)s, (uint32_t)i, &flags[i]);
    });
    size_t usable = 0;  // images in front of the first failure
    while (usable < count && good[usable]) usable++;
    if (usable > 0 &&
        (unshift >= 0 ? fpvhip_inverse_submit_raw(ctx_, (uint32_t)s, (uint32_t)usable, flags.data(), unshift,
                                                  big_endian ? 1 : 0)
                      : fpvhip_inverse_submit(ctx_, (uint32_t)s, (uint32_t)usable, flags.data()))) {
      Release(s);
      finish(prev);
      return FPV_FAILURE(fpvhip_last_error(ctx_));
    }
    finish(prev);
    if (usable > 0) {
      prev.slot = s;
      prev.first = first;
      prev.count = usable;
    } else {
      Release(s);
    }
    if (usable < count) {  // deliver what precedes the failure, then stop
      finish(prev);
      return false;
    }
  }
  finish(prev);
  return ok;
}

}  // namespace internal

using namespace internal;

namespace {
constexpr size_t kMaxImageSize = 1000000000;  // cc:164

int DecoderDevice() { return DeviceList()[0]; }

uint32_t DecodeBatchFrames(size_t npix) {
  const size_t cap = EnvSize("FPV_BATCH_BYTES", (size_t)256 << 20);
  size_t b = EnvSize("FPV_DECODE_BATCH", 32);
  b = std::min(b, std::max<size_t>(1, cap / std::max<size_t>(1, 2 * npix)));
  return (uint32_t)std::max<size_t>(1, b);
}
}  // namespace

// ======================================================== StreamingDecoder ==

struct StreamingDecoder::State {
  size_t xs = 0, ys = 0;
  bool has_header = false;
  std::vector<uint8_t> carry;
  std::shared_ptr<InverseEngine> engine;  // one image per GPU call (the usual block-wise feed)
  // Calls that bring several complete frame chunks at once (whole files, large
  // blocks) go through GPU batches with threaded Brotli; created on first need.
  std::shared_ptr<InverseEngine> batch_engine;
  std::vector<uint16_t> delta16;
};

// cc:866-956
void StreamingDecoder::Decode(
    const uint8_t* bytes, size_t size,
    std::function<void(bool ok, uint16_t* frame, size_t xsize, size_t ysize, void* payload)>
        callback,
    void* payload) {
  if (!state_) state_ = std::make_shared<State>();
  State& s = *state_;
#define FPV_FAIL_CB(msg)                                   \
  {                                                        \
    callback(FPV_FAILURE(msg), nullptr, 0, 0, payload);    \
    return;                                                \
  }
  // Append only when bytes are already pending (cc:872-874).
  if (!s.carry.empty()) s.carry.insert(s.carry.end(), bytes, bytes + size);
  const uint8_t* in = s.carry.empty() ? bytes : s.carry.data();
  const size_t insize = s.carry.empty() ? size : s.carry.size();

  size_t pos = 0;
  if (!s.has_header && insize > 13) {  // header + delta chunk (cc:886-912)
    s.xs = Load32(in);
    s.ys = Load32(in + 4);
    pos = 8;
    if (s.xs == 0 || s.ys == 0) FPV_FAIL_CB("invalid image dimensions");
    if (s.xs > 65536 || s.ys > 65536 || s.xs * s.ys > kMaxImageSize) FPV_FAIL_CB("image too large");
    const size_t dsz = Load32(in + pos);
    if (dsz < 5) FPV_FAIL_CB("too small for delta frame");
    if (in[pos + 4] != 1) FPV_FAIL_CB("not a delta frame");
    if (dsz + pos <= insize) {
      s.engine = InverseEngine::Create(s.xs, s.ys, 1, 2, DecoderDevice());
      if (!s.engine) FPV_FAIL_CB("no GPU decoder");
      const uint16_t* px = nullptr;
      int slot = -1;
      if (!s.engine->Decode(in + pos + 5, dsz - 5, &px, &slot))
        FPV_FAIL_CB("decompressing delta frame failed");
      const bool set = s.engine->SetDelta(px);
      if (set) s.delta16.assign(px, px + s.xs * s.ys);
      s.engine->Release(slot);
      if (!set) FPV_FAIL_CB("decompressing delta frame failed");
      pos += dsz;
      s.has_header = true;
    } else {
      pos = 0;  // wait for the whole delta chunk
    }
  }

  if (s.has_header) {  // frames (cc:914-936)
    // Every complete frame chunk that is available now.  A malformed chunk ends
    // the scan; its error is reported after the frames in front of it, as the
    // reference's sequential loop would.
    std::vector<const uint8_t*> mains;
    std::vector<size_t> sizes;
    const char* scan_error = nullptr;
    size_t scan = pos;
    while (scan + 9 <= insize) {
      const size_t fsz = Load32(in + scan);
      const uint8_t flag = in[scan + 4];
      if (flag == 2) break;  // frame index: end of frames
      if (flag != 0) { scan_error = "not a standard frame"; break; }
      if (scan + fsz > insize) break;
      const size_t psz = Load32(in + scan + 5);
      if (fsz < 9 || psz > fsz - 9) { scan_error = "preview size too large"; break; }
      mains.push_back(in + scan + 9 + psz);
      sizes.push_back(fsz - psz - 9);
      scan += fsz;
    }
    if (mains.size() == 1) {
      const uint16_t* px = nullptr;
      int slot = -1;
      if (!s.engine->Decode(mains[0], sizes[0], &px, &slot)) FPV_FAIL_CB("decompressing frame failed");
      callback(true, const_cast<uint16_t*>(px), s.xs, s.ys, payload);
      s.engine->Release(slot);
    } else if (mains.size() > 1) {
      if (!s.batch_engine) {
        const uint32_t batch = (uint32_t)std::min<size_t>(DecodeBatchFrames(s.xs * s.ys), 16);
        s.batch_engine = InverseEngine::Create(s.xs, s.ys, batch, 2, DecoderDevice());
        if (!s.batch_engine || !s.batch_engine->SetDelta(s.delta16.data())) FPV_FAIL_CB("no GPU decoder");
      }
      const size_t threads = std::min<size_t>(
          EnvSize("FPV_DECODE_THREADS", std::max(1u, std::min(8u, std::thread::hardware_concurrency()))),
          mains.size());
      const bool ok = s.batch_engine->DecodeBatch(
          mains.data(), sizes.data(), mains.size(),
          [&](size_t, const uint16_t* px) { callback(true, const_cast<uint16_t*>(px), s.xs, s.ys, payload); },
          threads);
      if (!ok) FPV_FAIL_CB("decompressing frame failed");
    }
    pos = scan;
    if (scan_error) FPV_FAIL_CB(scan_error);
  }

  // Keep what was not consumed (cc:938-953).
  if (s.carry.empty()) {
    if (pos < size) s.carry.assign(bytes + pos, bytes + size);
  } else if (pos > 0) {
    s.carry.erase(s.carry.begin(), s.carry.begin() + pos);
  }
#undef FPV_FAIL_CB
}

// ===================================================== RandomAccessDecoder ==

struct RandomAccessDecoder::State {
  std::shared_ptr<InverseEngine> engine;   // full frames (first device)
  // DecodeFrames shards large requests over every visible GPU (FPV_DEVICES): frames
  // depend only on the delta frame, so the shards never exchange anything.  Engines of
  // the other devices are created on first need.
  std::vector<int> devices;
  std::vector<std::shared_ptr<InverseEngine>> more;   // devices[1..]
  std::vector<uint16_t> delta16;
  uint32_t batch = 1;
  std::shared_ptr<InverseEngine> preview;  // (xs/4) x (ys/4), created on first use
  std::mutex m;
};

// cc:961-1015
bool RandomAccessDecoder::Init(const uint8_t* data, size_t size) {
  if (size < 12) return FPV_FAILURE("data too small to contain header");
  data_ = data;
  size_ = size;
  xsize_ = Load32(data);
  ysize_ = Load32(data + 4);
  if (xsize_ == 0 || ysize_ == 0) return FPV_FAILURE("invalid image dimensions");
  if (xsize_ > 65536 || ysize_ > 65536 || xsize_ * ysize_ > kMaxImageSize)
    return FPV_FAILURE("image too large");

  size_t pos = 8;
  const size_t dsz = Load32(data + pos);
  if (OutOfRange(pos, dsz, size)) return FPV_FAILURE("out of bounds");
  if (dsz < 5) return FPV_FAILURE("delta frame too small");
  if (data[12] != 1) return FPV_FAILURE("must begin with delta frame");

  state_ = std::make_shared<State>();
  const uint32_t batch = DecodeBatchFrames(xsize_ * ysize_);
  state_->devices = DeviceList();
  state_->batch = batch;
  state_->engine = InverseEngine::Create(xsize_, ysize_, batch, 2, state_->devices[0]);
  if (!state_->engine) return FPV_FAILURE("no GPU decoder");
  const uint16_t* px = nullptr;
  int slot = -1;
  if (!state_->engine->Decode(data + pos + 5, dsz - 5, &px, &slot))
    return FPV_FAILURE("failed to decode delta frame");
  const bool set = state_->engine->SetDelta(px);
  if (set) state_->delta16.assign(px, px + xsize_ * ysize_);
  state_->engine->Release(slot);
  if (!set) return FPV_FAILURE("failed to decode delta frame");

  // frame index at the end of the file (cc:994-1012)
  const size_t nframes = (size_t)Load64(data + size - 8);
  if (nframes > size / 16) return FPV_FAILURE("too many frames");
  const size_t footer = 5 + 8 * nframes + 8;
  if (footer > size) return FPV_FAILURE("footer too large");
  pos = size - footer;
  if (Load32(data + pos) != footer) return FPV_FAILURE("footer size mismatch");
  if (data[pos + 4] != 2) return FPV_FAILURE("must end with frame index");
  frame_offsets_.resize(nframes);
  for (size_t i = 0; i < nframes; i++) frame_offsets_[i] = (size_t)Load64(data + pos + 5 + 8 * i);
  return true;
}

namespace {
// Locates the main image chunk of frame `index` (cc:1018-1030).
bool FindMain(const uint8_t* data, size_t size, const std::vector<size_t>& offs, size_t index,
              const uint8_t** main, size_t* main_size) {
  if (index >= offs.size()) return FPV_FAILURE("invalid frame index");
  const size_t off = offs[index];
  if (OutOfRange(off, 9, size)) return FPV_FAILURE("out of bounds");
  const uint8_t* d = data + off;
  const size_t fsz = Load32(d);
  if (fsz < 9) return FPV_FAILURE("frame too small");
  if (OutOfRange(off, fsz, size)) return FPV_FAILURE("out of bounds");
  if (d[4] != 0) return FPV_FAILURE("not a standard frame");
  const size_t psz = Load32(d + 5);
  if (psz > fsz - 9) return FPV_FAILURE("preview too large");
  *main = d + 9 + psz;
  *main_size = fsz - psz - 9;
  return true;
}
}  // namespace

// cc:1017-1036
bool RandomAccessDecoder::DecodeFrame(size_t index, uint16_t* frame) const {
  if (!state_) return FPV_FAILURE("Init not called");
  const uint8_t* main;
  size_t main_size;
  if (!FindMain(data_, size_, frame_offsets_, index, &main, &main_size)) return false;
  const uint16_t* px = nullptr;
  int slot = -1;
  if (!state_->engine->Decode(main, main_size, &px, &slot)) return FPV_FAILURE("");
  memcpy(frame, px, xsize_ * ysize_ * 2);
  state_->engine->Release(slot);
  return true;
}

bool RandomAccessDecoder::DecodeFrames(const size_t* indices, size_t count,
                                       uint16_t* const* out, size_t threads) const {
  return DecodeSharded(indices, count, reinterpret_cast<void* const*>(out), threads, -1, false);
}

bool RandomAccessDecoder::DecodeFramesRaw(const size_t* indices, size_t count, uint8_t* const* out,
                                          int shift, bool big_endian, size_t threads) const {
  if (shift < 0 || shift > 15) return FPV_FAILURE("shift must be 0..15");
  return DecodeSharded(indices, count, reinterpret_cast<void* const*>(out), threads, shift, big_endian);
}

bool RandomAccessDecoder::DecodeSharded(const size_t* indices, size_t count, void* const* out,
                                        size_t threads, int unshift, bool big_endian) const {
  if (!state_) return FPV_FAILURE("Init not called");
  std::vector<const uint8_t*> ins(count);
  std::vector<size_t> sizes(count);
  for (size_t i = 0; i < count; i++)
    if (!FindMain(data_, size_, frame_offsets_, indices[i], &ins[i], &sizes[i])) return false;
  const size_t bytes = xsize_ * ysize_ * 2;
  State& st = *state_;
  // One shard per GPU when every shard still fills a batch; otherwise the first GPU alone.
  size_t shards = std::min(st.devices.size(), count / std::max<size_t>(1, st.batch));
  if (shards < 2) {
    return st.engine->DecodeBatch(
        ins.data(), sizes.data(), count,
        [&](size_t i, const uint16_t* px) { memcpy(out[i], px, bytes); }, threads, unshift, big_endian);
  }
  std::vector<std::shared_ptr<InverseEngine>> engines;  // snapshot: other callers may append meanwhile
  {
    std::lock_guard<std::mutex> l(st.m);
    while (st.more.size() + 1 < shards) {
      auto e = InverseEngine::Create(xsize_, ysize_, st.batch, 2, st.devices[st.more.size() + 1]);
      if (!e || !e->SetDelta(st.delta16.data())) return FPV_FAILURE("no GPU decoder on a further device");
      st.more.push_back(e);
    }
    engines.push_back(st.engine);
    for (size_t g = 1; g < shards; g++) engines.push_back(st.more[g - 1]);
  }
  std::vector<char> ok(shards, 0);
  std::vector<std::thread> pool;
  const size_t per = (count + shards - 1) / shards, th = std::max<size_t>(1, threads / shards);
  for (size_t g = 0; g < shards; g++) {
    pool.emplace_back([&, g] {
      const size_t first = g * per, n = first < count ? std::min(per, count - first) : 0;
      InverseEngine* e = engines[g].get();
      ok[g] = n == 0 || e->DecodeBatch(
                            ins.data() + first, sizes.data() + first, n,
                            [&, first](size_t i, const uint16_t* px) { memcpy(out[first + i], px, bytes); }, th,
                            unshift, big_endian);
    });
  }
  for (auto& t : pool) t.join();
  for (size_t g = 0; g < shards; g++)
    if (!ok[g]) return false;
  return true;
}

// cc:1038-1070
bool RandomAccessDecoder::DecodePreview(size_t index, uint8_t* preview) const {
  if (!state_) return FPV_FAILURE("Init not called");
  if (index >= frame_offsets_.size()) return FPV_FAILURE("invalid preview index");
  const size_t off = frame_offsets_[index];
  if (OutOfRange(off, 9, size_)) return FPV_FAILURE("");
  const uint8_t* d = data_ + off;
  const size_t fsz = Load32(d);
  if (fsz < 9) return FPV_FAILURE("frame too small");
  if (OutOfRange(off, fsz, size_)) return FPV_FAILURE("out of bounds");
  if (d[4] != 0) return FPV_FAILURE("not a standard frame");
  const size_t psz = Load32(d + 5);
  if (OutOfRange(9, psz, fsz)) return FPV_FAILURE("preview too large");

  const size_t pw = preview_xsize(), ph = preview_ysize();
  if (pw == 0 || ph == 0) return FPV_FAILURE("invalid image dimensions");
  {
    std::lock_guard<std::mutex> l(state_->m);
    if (!state_->preview) state_->preview = InverseEngine::Create(pw, ph, 1, 2, DecoderDevice());
    if (!state_->preview) return FPV_FAILURE("no GPU decoder");
  }
  const uint16_t* px = nullptr;
  int slot = -1;
  if (!state_->preview->Decode(d + 9, psz, &px, &slot))
    return FPV_FAILURE("failed to decompress preview");
  for (size_t i = 0; i < pw * ph; i++) preview[i] = (uint8_t)(px[i] >> 8);  // cc:1063-1067
  state_->preview->Release(slot);
  return true;
}

}  // namespace fpvc
