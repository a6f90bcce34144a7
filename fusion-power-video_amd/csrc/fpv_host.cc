// fpv_host.cc -- host side of the MI355X codec: the fusion_power_video.h API
// (Encoder, StreamingDecoder, RandomAccessDecoder, UnextractFrame) implemented
// as a batch pipeline over the C ABI of fpv_hip.h.
//
//   CompressFrame -> pinned staging slot -> [GPU: H2D, predictor kernels, D2H on
//   the slot's stream] -> dispatcher thread -> Brotli worker pool -> in-order
//   emitter -> user callback
//
// What runs where: every per-pixel loop of the reference's Frame ctor,
// Frame::Predict and DecompressImage (cc:326-344, 370-593) runs on the GPU.
// This file keeps only what the reference also does outside those loops: the
// container format (cc:30-155), Brotli calls (cc:186-214, 643-688), queueing
// and callback ordering (cc:1076-1230).  No CPU fallback exists.
#include <brotli/decode.h>
#include <brotli/encode.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include "fpv_hip.h"
#include "fpv_host_internal.h"
#include "fusion_power_video.h"

namespace fpvc {
namespace internal {

// ------------------------------------------------------------ diagnostics --

// Decoders report failures like the reference does (cc:172-178): one stderr
// line with the place, and a false result.
bool Failure(const char* file, int line, const char* msg) {
  const char* base = strrchr(file, '/');
  fprintf(stderr, "failure at: %s:%d", base ? base + 1 : file, line);
  if (msg && *msg) fprintf(stderr, ": %s", msg);
  fputc('\n', stderr);
  return false;
}

// API misuse (Init twice, CompressFrame before Init) and, with FPV_ABORT_ON_ERROR=1, any GPU
// failure: one message, then abort().  By default a GPU failure in an Encoder is latched
// instead (Encoder::LastError): the session stops accepting frames and Finish emits no index.
void Fatal(const char* what, const char* detail) {
  fprintf(stderr, "fpv-mi355x: fatal: %s: %s\n", what, detail ? detail : "");
  fprintf(stderr, "fpv-mi355x: the predictor stage has no CPU fallback; a usable "
                  "MI355X and libfpvhip.so are required\n");
  abort();
}

bool ErrorLatch::Fail(const char* who, const char* what, const char* detail) {
  if (getenv("FPV_ABORT_ON_ERROR")) Fatal(what, detail);
  std::lock_guard<std::mutex> l(m_);
  if (!failed_.load()) {
    text_ = std::string(what) + ": " + (detail ? detail : "");
    fprintf(stderr, "fpv-mi355x: %s failed: %s (no CPU fallback exists; further calls are no-ops until the error is "
                    "cleared; FPV_ABORT_ON_ERROR=1 aborts instead)\n", who, text_.c_str());
    failed_.store(true, std::memory_order_release);
  }
  return false;
}
std::string ErrorLatch::text() const {
  std::lock_guard<std::mutex> l(m_);
  return text_;
}
void ErrorLatch::Reset() {
  std::lock_guard<std::mutex> l(m_);
  text_.clear();
  failed_.store(false, std::memory_order_release);
}
ErrorLatch& FrameErrorLatch() {
  static ErrorLatch* l = new ErrorLatch();
  return *l;
}

// -------------------------------------------------------------- LE helpers --

uint32_t Load32(const uint8_t* p) {
  return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
}
uint64_t Load64(const uint8_t* p) {
  return (uint64_t)Load32(p) | (uint64_t)Load32(p + 4) << 32;
}
void Store32(uint32_t v, uint8_t* p) {
  for (int i = 0; i < 4; i++) p[i] = (uint8_t)(v >> (8 * i));
}
void Store64(uint64_t v, uint8_t* p) {
  for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i));
}
void Append32(uint32_t v, std::vector<uint8_t>* out) {
  uint8_t b[4];
  Store32(v, b);
  out->insert(out->end(), b, b + 4);
}
bool OutOfRange(size_t pos, size_t width, size_t size) {
  return pos > size || size - pos < width;
}

// ------------------------------------------------------------------ Brotli --

// libbrotli's own allocations, out of the way of the other threads.  BrotliEncoderCompress /
// BrotliDecoder* allocate their state -- hash table, command and literal buffers, ring buffer:
// 1.5-5 MB per call -- through malloc, i.e. one mmap, its page faults and one munmap per plane and
// thread.  With many threads compressing back to back (the GPU has taken the predictor stage, so
// that is ALL the worker threads do) these calls queue on the process's address-space lock:
// measured on 8 threads, one-shot calls on predicted high planes took 24 ms each where a lone
// thread takes 6.7.  The same library entry points take an allocator: every thread hands them a
// bump arena that lives as long as the thread (the bytes produced cannot depend on where the
// state lies: tests/test_host_cpu.py::test_brotli_arena_bytes_are_the_librarys).
namespace {
struct BrotliArena {
  uint8_t* base = nullptr;
  size_t cap = 0, at = 0;
  ~BrotliArena() { free(base); }
  static void* Alloc(void* opaque, size_t n) {
    BrotliArena* a = static_cast<BrotliArena*>(opaque);
    if (!a->base) {
      a->cap = EnvSize("FPV_BROTLI_ARENA_MB", 12) << 20;
      a->base = static_cast<uint8_t*>(malloc(a->cap));  // touched as far as it is used
      if (!a->base) a->cap = 0;
    }
    const size_t p = (a->at + 63) & ~(size_t)63;
    if (p + n > a->cap) return malloc(n);  // larger states than expected: the library's own way
    a->at = p + n;
    return a->base + p;
  }
  static void Free(void* opaque, void* p) {
    BrotliArena* a = static_cast<BrotliArena*>(opaque);
    if (!p) return;
    if (a->base && p >= a->base && p < a->base + a->cap) return;  // the arena is reset per stream
    free(p);
  }
};
BrotliArena& ThreadArena() {
  static thread_local BrotliArena a;
  a.at = 0;
  return a;
}
}  // namespace

// BrotliEncoderCompress(1, BROTLI_DEFAULT_WINDOW, BROTLI_DEFAULT_MODE, n, in, size, out) with the
// encoder's state in this thread's arena: the same calls libbrotli 1.0.9's one-shot entry point
// makes (quality, lgwin, mode, size hint = n, one CompressStream with OPERATION_FINISH), and the
// library's own entry point for whatever is not the plain case (empty input, a failure, a result
// beyond BrotliEncoderMaxCompressedSize: its "store uncompressed" fall-back).  cc:653.
bool BrotliCompress(const uint8_t* in, size_t n, uint8_t* out, size_t* size) {
  const size_t cap = *size, max_out = BrotliEncoderMaxCompressedSize(n);
  static const bool plain = getenv("FPV_BROTLI_ARENA") && !strcmp(getenv("FPV_BROTLI_ARENA"), "0");
  if (n > 0 && cap > 0 && !plain) {
    BrotliArena& a = ThreadArena();
    if (BrotliEncoderState* st = BrotliEncoderCreateInstance(BrotliArena::Alloc, BrotliArena::Free, &a)) {
      BrotliEncoderSetParameter(st, BROTLI_PARAM_QUALITY, 1);
      BrotliEncoderSetParameter(st, BROTLI_PARAM_LGWIN, BROTLI_DEFAULT_WINDOW);
      BrotliEncoderSetParameter(st, BROTLI_PARAM_MODE, BROTLI_DEFAULT_MODE);
      BrotliEncoderSetParameter(st, BROTLI_PARAM_SIZE_HINT, (uint32_t)n);
      size_t avail_in = n, avail_out = cap, total = 0;
      const uint8_t* next_in = in;
      uint8_t* next_out = out;
      const bool ok = BrotliEncoderCompressStream(st, BROTLI_OPERATION_FINISH, &avail_in, &next_in, &avail_out, &next_out, &total) &&
                      BrotliEncoderIsFinished(st);
      BrotliEncoderDestroyInstance(st);
      if (ok && (max_out == 0 || total <= max_out)) {
        *size = total;
        return true;
      }
    }
  }
  *size = cap;
  return BrotliEncoderCompress(1, BROTLI_DEFAULT_WINDOW, BROTLI_DEFAULT_MODE, n, in, size, out) != 0;
}

// cc:653: quality 1, default window (22), default mode; one shot per plane.
// Appends to *out.  The reference first tries a plane-sized output buffer for
// the low plane and retries with the maximum (cc:661-673); the bytes are the
// same as this single maximum-size call.
bool BrotliAppend(const uint8_t* in, size_t n, std::vector<uint8_t>* out) {
  size_t cap = BrotliEncoderMaxCompressedSize(n);
  if (cap == 0) cap = n + 1024;
  size_t at = out->size();
  out->resize(at + cap);
  size_t sz = cap;
  if (!BrotliCompress(in, n, out->data() + at, &sz)) {
    out->resize(at);
    return false;
  }
  out->resize(at + sz);
  return true;
}

// cc:186-214: decodes ONE stream starting at in[*pos] straight into dst (cap
// bytes); *pos ends up behind the stream, *got = bytes produced.  A stream
// longer than cap stops with got == cap and returns false.
bool BrotliInto(const uint8_t* in, size_t size, size_t* pos, uint8_t* dst,
                size_t cap, size_t* got) {
  BrotliArena& arena = ThreadArena();  // the decoder's state (ring buffer, trees) in this thread's arena
  BrotliDecoderState* st = BrotliDecoderCreateInstance(BrotliArena::Alloc, BrotliArena::Free, &arena);
  if (!st) return false;
  size_t avail_in = size - *pos, avail_out = cap;
  const uint8_t* next_in = in + *pos;
  uint8_t* next_out = dst;
  BrotliDecoderResult r = BrotliDecoderDecompressStream(st, &avail_in, &next_in,
                                                        &avail_out, &next_out, nullptr);
  BrotliDecoderDestroyInstance(st);
  *pos = size - avail_in;
  *got = cap - avail_out;
  return r == BROTLI_DECODER_RESULT_SUCCESS;
}

// ------------------------------------------------------------- thread pool --

namespace {
// A job is a range of indices handed out one at a time; helpers and the caller pull
// from it until it is exhausted.  The pool's threads are detached and parked on a
// condition variable; they are never joined (the process simply ends).
struct PoolJob {
  const std::function<void(size_t)>* fn;
  size_t n;
  std::atomic<size_t> next{0}, done{0};
  size_t helpers_wanted;  // how many pool threads may still join this job
};
struct Pool {
  std::mutex m;
  std::condition_variable cv_work, cv_done;
  std::deque<std::shared_ptr<PoolJob>> jobs;  // jobs that still want helpers
  size_t threads = 0;
};
Pool& ThePool() {
  static Pool* p = new Pool();  // leaked on purpose: parked threads may outlive static destruction
  return *p;
}
void RunJob(PoolJob& j) {
  for (;;) {
    const size_t i = j.next.fetch_add(1);
    if (i >= j.n) return;
    (*j.fn)(i);
    if (j.done.fetch_add(1) + 1 == j.n) {
      Pool& p = ThePool();
      std::lock_guard<std::mutex> l(p.m);
      p.cv_done.notify_all();
    }
  }
}
void PoolThread() {
  Pool& p = ThePool();
  std::unique_lock<std::mutex> l(p.m);
  for (;;) {
    p.cv_work.wait(l, [&] { return !p.jobs.empty(); });
    std::shared_ptr<PoolJob> j = p.jobs.front();
    if (--j->helpers_wanted == 0) p.jobs.pop_front();
    l.unlock();
    RunJob(*j);
    l.lock();
  }
}
}  // namespace

void ParallelFor(size_t n, size_t threads, const std::function<void(size_t)>& fn) {
  if (n == 0) return;
  const size_t helpers = std::min<size_t>(std::min(threads, n), 64) - (threads && n ? 1 : 0);
  if (helpers == 0 || threads <= 1) {
    for (size_t i = 0; i < n; i++) fn(i);
    return;
  }
  std::shared_ptr<PoolJob> j = std::make_shared<PoolJob>();
  j->fn = &fn;
  j->n = n;
  j->helpers_wanted = helpers;
  Pool& p = ThePool();
  {
    std::lock_guard<std::mutex> l(p.m);
    while (p.threads < helpers) {  // grow the pool to what this call can use
      std::thread(PoolThread).detach();
      p.threads++;
    }
    p.jobs.push_back(j);
  }
  p.cv_work.notify_all();
  RunJob(*j);
  std::unique_lock<std::mutex> l(p.m);
  p.cv_done.wait(l, [&] { return j->done.load() == j->n; });
  // helpers that have not picked the job up yet must not: its fn dies with this frame
  for (auto it = p.jobs.begin(); it != p.jobs.end(); ++it)
    if (it->get() == j.get()) {
      p.jobs.erase(it);
      break;
    }
}

// ----------------------------------------------------------------- devices --

// GPUs used by the encoder: FPV_DEVICES="0,2,3" or every visible device.
std::vector<int> DeviceList() {
  int n = fpvhip_device_count();
  std::vector<int> devs;
  if (n <= 0) return devs;  // callers report it (no CPU fallback exists)
  if (const char* env = getenv("FPV_DEVICES")) {
    for (const char* p = env; *p;) {
      char* end;
      long d = strtol(p, &end, 10);
      if (end == p) break;
      if (d >= 0 && d < n) devs.push_back((int)d);
      p = *end ? end + 1 : end;
    }
  }
  if (devs.empty())
    for (int d = 0; d < n; d++) devs.push_back(d);
  return devs;
}

size_t EnvSize(const char* name, size_t dflt) {
  const char* e = getenv(name);
  if (!e || !*e) return dflt;
  long long v = atoll(e);
  return v > 0 ? (size_t)v : dflt;
}

// One frame chunk (OutputFull, cc:830-846) from predicted planes:
//   u32 total | 0x00 | u32 (1 + len(preview stream)) | (flags & CG) | NO_LOW |
//   brotli(preview) | flags | [brotli(low)] | brotli(high)
// Built in a scratch buffer that lives as long as its worker thread (sized once for the
// worst case) and returned as an exact-size copy: megabyte-sized vectors that grow while
// Brotli appends to them made every frame pay for several mmap / munmap calls, which many
// threads doing it at once serialise on.
void BuildFrameChunk(const uint8_t* high, const uint8_t* low, const uint8_t* preview,
                     uint8_t flags, size_t npix, size_t npv,
                     std::vector<uint8_t>* out) {
  static thread_local std::vector<uint8_t> scratch;
  const size_t cap_pv = std::max<size_t>(BrotliEncoderMaxCompressedSize(npv), npv + 1024);
  const size_t cap_pl = std::max<size_t>(BrotliEncoderMaxCompressedSize(npix), npix + 1024);
  const size_t need = 16 + cap_pv + 2 * cap_pl;
  if (scratch.size() < need) scratch.resize(need);
  uint8_t* p = scratch.data();
  size_t at = 9;
  p[at++] = (uint8_t)((flags & USE_CG) | NO_LOW_BYTES);  // cc:842
  auto brotli = [&](const uint8_t* in, size_t n, size_t cap, const char* what) {
    size_t sz = cap;  // cc:653: quality 1, default window (22), default mode; one shot per plane
    if (!BrotliCompress(in, n, p + at, &sz)) Fatal("Brotli", what);
    at += sz;
  };
  brotli(preview, npv, cap_pv, "preview plane");
  Store32((uint32_t)(at - 9), p + 5);  // preview stream + its flags byte
  p[at++] = flags;                     // cc:825
  if (!(flags & NO_LOW_BYTES)) brotli(low, npix, cap_pl, "low plane");
  brotli(high, npix, cap_pl, "high plane");
  Store32((uint32_t)at, p);
  p[4] = 0;  // chunk flag: ordinary frame (cc:840)
  out->assign(p, p + at);
}

// Image core only (OutputCore, cc:820-828), appended.
void AppendCore(const uint8_t* high, const uint8_t* low, uint8_t flags, size_t npix,
                std::vector<uint8_t>* out) {
  out->push_back(flags);
  if (!(flags & NO_LOW_BYTES) && !BrotliAppend(low, npix, out)) Fatal("Brotli", "low plane");
  if (!BrotliAppend(high, npix, out)) Fatal("Brotli", "high plane");
}

// Device warm-up.  The upstream benchmark times Encoder construction -> Finish
// (benchmark.cc:153-180): runtime and context creation (hundreds of ms once per process), the
// first launch of each kernel (1-3 ms) and the first large transfer in either direction (~7 ms:
// the copy engines' queues are created then) would land inside that timer.  A process that
// merely links the library must not touch a GPU -- and must not own a thread either (threads do
// not survive fork()) -- so nothing at all happens at load by default:
//   FPV_WARMUP=lazy  (default)  the first Encoder / decoder / Frame object starts a helper thread
//                               per visible device (FPV_DEVICES) that brings the device up --
//                               runtime, primary context, code object, stream pool, then one
//                               small frame through the kernels of the direction that object
//                               needs; whoever needs the GPU first waits for it (WarmUpWait)
//   FPV_WARMUP=eager            the warm-up (both directions) runs when the library is loaded
//                               (round 2's behaviour)
//   FPV_WARMUP=off / FPV_NO_WARMUP=1   none
// Every device of FPV_DEVICES is warmed, side by side: with eight GPUs seven of them would
// otherwise pay their first stream / copy-engine / first-launch costs inside the first session
// (the reference's analogue is the thread spawn in its constructor, cc:1076-1084).
namespace {
double WarmNowMs() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// One small frame through the ordinary entry points of the wanted directions; the common kernel
// variants (little-endian input with a low plane) are the ones this frame takes.
void WarmDevice(int dev, int dirs) {
  const bool timing = getenv("FPV_TIMING") != nullptr;
  const double t0 = WarmNowMs();
  if (fpvhip_warmup(dev)) return;
  const double t1 = WarmNowMs();
  const uint32_t wx = 512, wy = 256;  // every launch and both transfer directions once; the size does not matter
  if (fpvhip_ctx* c = (dirs & kWarmForward) ? fpvhip_create(dev, wx, wy, 4, 0, 1, 1) : fpvhip_create_decoder(dev, wx, wy, 1, 1)) {
    std::vector<uint16_t> img((size_t)wx * wy);
    for (size_t i = 0; i < img.size(); i++) img[i] = (uint16_t)((i * 37) & 0xfff);
    const uint8_t *h, *l, *pv, *fl;
    const uint16_t* px;
    uint8_t f = 3;
    bool ok = true;
    if (dirs & kWarmForward) {
      ok = fpvhip_set_delta_raw(c, img.data()) == 0;
      if (ok) {
        memcpy(fpvhip_stage_in(c, 0, 0), img.data(), img.size() * 2);
        ok = fpvhip_predict_submit(c, 0, 1, 1) == 0 && fpvhip_predict_wait(c, 0, &h, &l, &pv, &fl) == 0;
        if (ok) f = fl[0] | 2;  // through the speculative inverse as well
      }
    } else {
      ok = fpvhip_set_delta_decoded(c, img.data()) == 0;
      if (ok) {
        memset(fpvhip_stage_plane(c, 0, 0, 0), 1, img.size());
        memset(fpvhip_stage_plane(c, 0, 0, 1), 2, img.size());
      }
    }
    if (ok && (dirs & kWarmInverse) && fpvhip_inverse_submit(c, 0, 1, &f) == 0) fpvhip_inverse_wait(c, 0, &px);
    fpvhip_destroy(c);
  }
  if (timing)
    fprintf(stderr, "fpv-mi355x: warm-up of device %d: bring-up %.1f ms, first frame (%s%s) %.1f ms\n", dev, t1 - t0,
            (dirs & kWarmForward) ? "forward" : "", (dirs & kWarmInverse) ? " inverse" : "", WarmNowMs() - t1);
}

void WarmUpNow(int dirs) {
  if (fpvhip_device_count() <= 0) return;
  const std::vector<int> devs = DeviceList();
  std::vector<int> uniq;
  for (int d : devs)
    if (std::find(uniq.begin(), uniq.end(), d) == uniq.end()) uniq.push_back(d);
  std::vector<std::thread> th;
  for (size_t i = 1; i < uniq.size(); i++) th.emplace_back(WarmDevice, uniq[i], dirs);
  if (!uniq.empty()) WarmDevice(uniq[0], dirs);
  for (auto& t : th) t.join();
}

struct WarmUp {
  std::mutex m;
  std::condition_variable cv;
  int state = 0;    // 0 idle, 2 running, 3 done
  int done_dirs = 0;  // directions warmed so far
  enum Mode { kLazy, kEager, kOff } mode = kLazy;
  WarmUp() {
    const char* e = getenv("FPV_WARMUP");
    if (getenv("FPV_NO_WARMUP") || (e && !strcmp(e, "off"))) mode = kOff;
    else if (e && !strcmp(e, "eager")) mode = kEager;
    if (mode == kOff) { state = 3; done_dirs = kWarmForward | kWarmInverse; return; }
    if (mode == kEager) {  // at load, before main(): nothing of it lands in anybody's timer
      WarmUpNow(kWarmForward | kWarmInverse);
      state = 3;
      done_dirs = kWarmForward | kWarmInverse;
    }
  }
  // Starts the helper for the directions not warmed yet (m held).  The helper is created here,
  // on demand, never at load: a process forked after loading the library has no threads of the
  // parent, and a parked helper it relied on would never answer.
  void StartLocked(int dirs) {
    const int want = dirs & ~done_dirs;
    if (!want || state == 2) return;  // (a running warm-up of another direction: the next request picks it up)
    state = 2;
    std::thread([this, want] {
      WarmUpNow(want);
      std::lock_guard<std::mutex> l(m);
      done_dirs |= want;
      state = 3;
      cv.notify_all();
    }).detach();
  }
};
WarmUp& TheWarmUp() {
  static WarmUp* w = new WarmUp();  // leaked on purpose: the helper thread may outlive static destruction
  return *w;
}
// A child forked while a warm-up was running inherits "running" without the thread that would
// finish it: start over (the HIP runtime does not survive fork() either; the child's first
// object then fails with the runtime's own error instead of hanging here).
void WarmUpAfterFork() {
  WarmUp& w = TheWarmUp();
  new (&w.m) std::mutex();
  new (&w.cv) std::condition_variable();
  if (w.state == 2) w.state = w.done_dirs ? 3 : 0;
}
struct WarmUpAtLoad {
  WarmUpAtLoad() {
    TheWarmUp();  // reads FPV_WARMUP; touches no GPU and starts no thread unless FPV_WARMUP=eager
    pthread_atfork(nullptr, nullptr, WarmUpAfterFork);
  }
} g_warm_up_at_load;
}  // namespace

void WarmUpRequest(int dirs) {
  WarmUp& w = TheWarmUp();
  std::lock_guard<std::mutex> l(w.m);
  w.StartLocked(dirs);
}

void WarmUpWait(int dirs) {
  WarmUp& w = TheWarmUp();
  std::unique_lock<std::mutex> l(w.m);
  for (;;) {
    if ((w.done_dirs & dirs) == dirs) return;
    w.StartLocked(dirs);
    w.cv.wait(l, [&w] { return w.state != 2; });
  }
}

}  // namespace internal

using namespace internal;

// cc:850-862 -- file-format repacking outside the codec; plain host loop.
void UnextractFrame(const uint16_t* img, size_t xsize, size_t ysize, int shift,
                    bool big_endian, uint8_t* out) {
  const size_t n = xsize * ysize;
  const int lo = big_endian ? 1 : 0, hi = 1 - lo;
  for (size_t i = 0; i < n; i++) {
    const uint16_t u = (uint16_t)(img[i] >> shift);
    out[2 * i + lo] = (uint8_t)(u & 0xff);
    out[2 * i + hi] = (uint8_t)(u >> 8);
  }
}

// ================================================================= Encoder ==

struct Encoder::Impl {
  struct Batch {
    int dev = 0;            // index into ctxs
    uint32_t slot = 0;
    uint32_t n = 0;         // frames accepted
    uint32_t copied = 0;    // frames already copied into pinned staging
    bool closed = false;    // no more frames will be added
    size_t first_id = 0;
    std::vector<Callback> cbs;
    std::vector<void*> payloads;
    // filled by the dispatcher after the GPU batch is back
    const uint8_t *high = nullptr, *low = nullptr, *pv = nullptr, *flags = nullptr;
    std::vector<std::vector<uint8_t>> chunks;
    std::vector<char> done;
    uint32_t emitted = 0;
  };
  struct Task { Batch* b; uint32_t k; const uint16_t* src; };  // src != null: staging copy

  const size_t num_threads;
  const int shift;
  const bool big_endian;
  size_t batch_promised = 1;   // frames per batch announced through MaxQueued()
  size_t batch = 1;            // effective (after Init, bounded by bytes)
  size_t max_queued = 1;
  uint32_t slots_per_dev = 3;

  size_t xs = 0, ys = 0, npix = 0, npv = 0;
  bool initialized = false, finished = false;
  std::vector<int> devices;
  std::vector<fpvhip_ctx*> ctxs;

  std::mutex m;
  std::condition_variable cv_producer;   // slot freed / in-flight count dropped
  std::vector<std::unique_ptr<std::condition_variable>> cv_dispatch;  // [dev] batch submitted / stop
  std::condition_variable cv_work;       // tasks available / stop
  bool stop = false;
  // First GPU failure of the session (the API is void everywhere, cc:1086-1157): latched, the
  // session stops accepting frames, Finish emits no index.  FPV_ABORT_ON_ERROR=1: abort() instead.
  std::atomic<bool> failed{false};
  std::string error;

  std::vector<std::vector<char>> slot_busy;   // [dev][slot]
  size_t next_dev = 0;
  Batch* cur = nullptr;                       // batch being filled by the producer
  std::chrono::steady_clock::time_point cur_touched;
  std::vector<std::deque<Batch*>> gpu_q;      // [dev] submitted, waiting for the GPU: one dispatcher thread per device,
                                              // so that a slow device never delays the hand-off of the others' batches
  std::vector<std::deque<Batch*>> ready_q;    // [dev] complete batches waiting for room on the device
  std::vector<int> gpu_inflight;              // [dev] batches submitted and not yet back
  int gpu_inflight_max = 3;
  std::deque<Task> work_q;                    // Brotli tasks
  int working = 0;                            // workers inside a Brotli task
  std::deque<Task> copy_q;                    // staging copies (served first)
  struct Prep { int dev; uint32_t slot; };
  std::deque<Prep> prep_q;                    // slots to allocate ahead of need (served before everything)
  std::vector<std::vector<char>> slot_prepared;  // [dev][slot]: allocation done or queued
  std::deque<Batch*> emit_q;                  // submission order
  size_t inflight = 0;                        // accepted, not yet emitted
  size_t next_id = 0;

  std::vector<uint64_t> offsets;              // frame index (cc:1180)
  uint64_t bytes_written = 0;

  std::vector<std::thread> dispatchers;       // [dev]
  std::vector<std::thread> workers;

  // Stage statistics (FPV_STATS=1: one stderr line from Finish).  Times in ns.
  const bool print_stats = getenv("FPV_STATS") != nullptr;
  const bool stats = true;  // a handful of clock reads per frame
  std::atomic<uint64_t> st_brotli_ns{0}, st_copy_ns{0}, st_gpu_wait_ns{0}, st_gpu_us{0}, st_batches{0}, st_bytes_up{0}, st_bytes_down{0};
  enum : size_t { kMaxStatDevs = 16 };  // (an enumerator: never ODR-used, no out-of-class definition needed in C++14)
  std::atomic<uint64_t> st_dev_batches[kMaxStatDevs] = {}, st_dev_wait_ns[kMaxStatDevs] = {};  // per device of the session
  double last_stats[11 + 2 * kMaxStatDevs] = {};  // filled by Finish (Encoder::LastStats)
  std::chrono::steady_clock::time_point st_t0;
  static uint64_t NowNs() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
               std::chrono::steady_clock::now().time_since_epoch()).count();
  }

  Impl(size_t threads, int s, bool be) : num_threads(threads), shift(s), big_endian(be) {
    if (threads == 0) {
      batch_promised = 1;
      max_queued = 1;  // cc:1176
    } else {
      // Small batches, many slots: a slot's pinned memory is allocated when the
      // pipeline first needs it and PCIe is efficient from a few MB up.  Frames in
      // flight must cover every Brotli worker PLUS the batches that are being
      // filled, staged and transferred at the same moment: 2 x threads + 2 batches.
      size_t b = EnvSize("FPV_BATCH", 0);
      if (b == 0) b = threads >= 16 ? 16 : 8;
      batch_promised = b;
      const size_t want = (2 * threads + 2 * b + b - 1) / b;
      slots_per_dev = (uint32_t)std::min<size_t>(EnvSize("FPV_SLOTS", 20), std::max<size_t>(3, want));
      max_queued = slots_per_dev * b;
    }
  }

  // Latches the session's first failure and wakes everybody up (called without the lock).
  void Fail(const char* what, const char* detail) {
    if (getenv("FPV_ABORT_ON_ERROR")) Fatal(what, detail);
    {
      std::lock_guard<std::mutex> l(m);
      if (!failed.load()) {
        error = std::string(what) + ": " + (detail ? detail : "");
        fprintf(stderr, "fpv-mi355x: encoder session failed: %s (no CPU fallback exists; frames are dropped, "
                        "Finish will emit no index; FPV_ABORT_ON_ERROR=1 aborts instead)\n", error.c_str());
        failed.store(true);
      }
    }
    cv_producer.notify_all();
    cv_work.notify_all();
    for (auto& cv : cv_dispatch) cv->notify_all();
  }
  bool CheckHip(int rc, fpvhip_ctx* c, const char* what) {
    if (rc == 0) return true;
    Fail(what, fpvhip_last_error(c));
    return false;
  }

  // Hands a complete batch to the GPU.  Called WITHOUT the pipeline lock (the call
  // enqueues copies and launches: the Brotli workers must not queue behind it); the
  // batch is closed and fully staged, so nothing else touches it meanwhile, and only
  // one thread ever gets `true` from CloseLocked / the last staging copy for it.
  // At most kGpuInflight batches per device are queued on the GPU at a time; further complete
  // batches wait in ready_q and are submitted by the dispatcher as earlier ones come back.
  // Three keep copy-in, kernels and copy-out of consecutive batches overlapped; with more
  // queued the runtime no longer keeps the two transfer directions on separate copy engines
  // and the host link drops from 40 to 16-25 GB/s each way (tools/pcie_rate.py, 4-8 slots).
  void Submit(Batch* b) {
    {
      std::lock_guard<std::mutex> l(m);
      if (gpu_inflight[b->dev] >= gpu_inflight_max) {
        ready_q[b->dev].push_back(b);
        return;
      }
      gpu_inflight[b->dev]++;
    }
    DoSubmit(b);
  }

  void DoSubmit(Batch* b) {
    if (failed.load() || !CheckHip(fpvhip_predict_submit(ctxs[b->dev], b->slot, b->n, 1), ctxs[b->dev], "fpvhip_predict_submit"))
      return;  // the batch stays in emit_q and is freed by Finish
    st_bytes_up += (uint64_t)b->n * npix * 2;
    st_bytes_down += (uint64_t)b->n * (npix * (shift == 8 ? 1 : 2) + npv + 1);
    std::lock_guard<std::mutex> l(m);
    gpu_q[b->dev].push_back(b);
    cv_dispatch[b->dev]->notify_one();
  }

  // Closes a batch (m held).  Returns true when all of its frames are staged already:
  // the caller then submits it after dropping the lock.  Otherwise the worker that
  // finishes the last staging copy does.
  bool CloseLocked(Batch* b) {
    b->closed = true;
    return b->copied == b->n;
  }

  // Queues the allocation of up to `count` slots of device d that nobody has asked for yet
  // (m held).  Workers allocate different slots side by side.
  void QueuePrepLocked(size_t d, int count) {
    for (uint32_t s2 = 0; s2 < slots_per_dev && count > 0; s2++)
      if (!slot_prepared[d][s2]) {
        slot_prepared[d][s2] = 1;
        prep_q.push_back(Prep{(int)d, s2});
        cv_work.notify_one();
        count--;
      }
  }

  Batch* OpenBatchLocked(std::unique_lock<std::mutex>& l) {
    for (;;) {
      for (size_t t = 0; t < devices.size(); t++) {
        size_t d = (next_dev + t) % devices.size();
        for (uint32_t s = 0; s < slots_per_dev; s++) {
          if (!slot_busy[d][s]) {
            slot_busy[d][s] = 1;
            next_dev = (d + 1) % devices.size();  // batch b -> GPU b mod G
            // The staging memory of a slot is allocated when it is first needed (a few
            // ms).  Have a worker prepare the slot the NEXT batch of this device will
            // take while this one is being filled, so that the producer never waits.
            slot_prepared[d][s] = 1;
            QueuePrepLocked(d, 2);
            Batch* b = new Batch();
            b->dev = (int)d;
            b->slot = s;
            b->first_id = next_id;
            // First touch of a slot allocates its pinned memory (tens of ms):
            // do it without the lock, the slot is already marked busy.
            l.unlock();
            const bool staged = fpvhip_stage_in(ctxs[d], s, 0) != nullptr;
            if (!staged) Fail("fpvhip_stage_in", fpvhip_last_error(ctxs[d]));
            l.lock();
            emit_q.push_back(b);  // emission order = the order batches are opened in; owns the batch
            return staged ? b : nullptr;
          }
        }
      }
      if (failed.load()) return nullptr;
      cv_producer.wait(l);
    }
  }

  bool NoGpuBatchQueuedLocked() const {
    for (const auto& q : gpu_q)
      if (!q.empty()) return false;
    return true;
  }

  // --- dispatchers, one per device: wait for that device's GPU batches in submission order and
  // fan out their Brotli tasks; whichever is idle flushes a partially filled batch when the
  // producer goes quiet.  (Round 2 had ONE dispatcher blocking in fpvhip_predict_wait on the
  // head of one global queue: a late batch of one GPU delayed the hand-off of every other
  // GPU's finished batches.) ---
  void DispatchLoop(size_t d) {
    std::unique_lock<std::mutex> l(m);
    for (;;) {
      if (failed.load()) {
        gpu_q[d].clear();  // the batches stay in emit_q and are freed by Finish
        if (stop) return;
        cv_dispatch[d]->wait_until(l, std::chrono::system_clock::now() + std::chrono::milliseconds(20));
        continue;
      }
      if (gpu_q[d].empty()) {
        if (stop) return;
        // system_clock -> pthread_cond_timedwait, which ThreadSanitizer understands
        // (clockwait is not intercepted by this toolchain); only a flush timer.
        cv_dispatch[d]->wait_until(l, std::chrono::system_clock::now() + std::chrono::milliseconds(2));
        if (NoGpuBatchQueuedLocked() && cur && cur->n > 0 &&
            std::chrono::steady_clock::now() - cur_touched > std::chrono::milliseconds(2)) {
          Batch* b = cur;
          cur = nullptr;
          if (CloseLocked(b)) {
            l.unlock();
            Submit(b);
            l.lock();
          }
        }
        continue;
      }
      Batch* b = gpu_q[d].front();
      gpu_q[d].pop_front();
      l.unlock();
      const uint64_t tw = stats ? NowNs() : 0;
      const bool ok = CheckHip(fpvhip_predict_wait(ctxs[b->dev], b->slot, &b->high, &b->low, &b->pv, &b->flags),
                               ctxs[b->dev], "fpvhip_predict_wait");
      if (ok && stats) {
        const uint64_t waited = NowNs() - tw;
        st_gpu_wait_ns += waited;
        if (d < kMaxStatDevs) { st_dev_batches[d]++; st_dev_wait_ns[d] += waited; }
        const double ms = fpvhip_slot_elapsed_ms(ctxs[b->dev], b->slot);
        if (ms > 0) st_gpu_us += (uint64_t)(ms * 1000.0);
        st_batches++;
      }
      l.lock();
      if (!ok) continue;
      b->chunks.resize(b->n);
      b->done.assign(b->n, 0);
      for (uint32_t k = 0; k < b->n; k++) work_q.push_back(Task{b, k, nullptr});
      cv_work.notify_all();
      gpu_inflight[b->dev]--;
      if (!ready_q[b->dev].empty()) {  // the device has room again: hand it the next complete batch
        Batch* nb = ready_q[b->dev].front();
        ready_q[b->dev].pop_front();
        gpu_inflight[nb->dev]++;
        l.unlock();
        DoSubmit(nb);
        l.lock();
      }
    }
  }

  void CompressTask(const Task& t, std::vector<uint8_t>* chunk) {
    const Batch* b = t.b;
    BuildFrameChunk(b->high + (size_t)t.k * npix, b->low + (size_t)t.k * npix,
                    b->pv + (size_t)t.k * npv, b->flags[t.k], npix, npv, chunk);
  }

  // Emits every finished frame at the head of the order.  Called with m held;
  // the lock is dropped around each user callback, and `emitting` keeps the
  // callbacks mutually exclusive and in submission order (cc:1217-1224).
  bool emitting = false;
  void DrainLocked(std::unique_lock<std::mutex>& l) {
    if (emitting) return;
    emitting = true;
    while (!emit_q.empty()) {
      Batch* b = emit_q.front();
      if (b->done.empty()) break;  // GPU batch not back yet
      if (b->emitted < b->n) {
        const uint32_t k = b->emitted;
        if (!b->done[k]) break;
        std::vector<uint8_t> c;
        c.swap(b->chunks[k]);
        offsets.push_back(bytes_written);  // cc:1180-1182
        bytes_written += c.size();
        l.unlock();
        b->cbs[k](c.data(), c.size(), b->payloads[k]);
        l.lock();
        b->emitted++;
        inflight--;
        cv_producer.notify_all();
        continue;
      }
      emit_q.pop_front();
      slot_busy[b->dev][b->slot] = 0;
      delete b;
      cv_producer.notify_all();
    }
    emitting = false;
  }

  void WorkLoop() {
    std::unique_lock<std::mutex> l(m);
    for (;;) {
      cv_work.wait(l, [this] { return stop || failed.load() || !work_q.empty() || !copy_q.empty() || !prep_q.empty(); });
      if (failed.load()) {
        // No further batch goes to the GPU.  What is still in work_q belongs to batches that CAME
        // BACK before the failure -- their planes are valid -- and is compressed and emitted as
        // usual, so that the stream ends with a clean, deterministic prefix: every frame of every
        // batch in front of the first one that failed.
        copy_q.clear();
        prep_q.clear();
        if (work_q.empty()) {
          if (stop) return;
          cv_work.wait_until(l, std::chrono::system_clock::now() + std::chrono::milliseconds(20));
          continue;
        }
      }
      if (!prep_q.empty()) {
        const Prep pr = prep_q.front();
        prep_q.pop_front();
        l.unlock();
        fpvhip_prepare_slot(ctxs[pr.dev], pr.slot);  // a failure surfaces when the slot is used
        l.lock();
        continue;
      }
      if (!copy_q.empty()) {  // staging copy: the frame stays valid until its callback (h:197-201)
        Task t = copy_q.front();
        copy_q.pop_front();
        l.unlock();
        const uint64_t tc = stats ? NowNs() : 0;
        memcpy(fpvhip_stage_in(ctxs[t.b->dev], t.b->slot, t.k), t.src, npix * 2);
        if (stats) st_copy_ns += NowNs() - tc;
        l.lock();
        t.b->copied++;
        if (t.b->closed && t.b->copied == t.b->n) {
          l.unlock();
          Submit(t.b);
          l.lock();
        }
        continue;
      }
      if (work_q.empty()) {
        if (stop) return;
        continue;
      }
      Task t = work_q.front();
      work_q.pop_front();
      working++;
      l.unlock();
      std::vector<uint8_t> chunk;
      const uint64_t tb = stats ? NowNs() : 0;
      CompressTask(t, &chunk);
      if (stats) st_brotli_ns += NowNs() - tb;
      l.lock();
      t.b->chunks[t.k].swap(chunk);
      t.b->done[t.k] = 1;
      DrainLocked(l);
      working--;
      if (failed.load()) cv_producer.notify_all();  // Finish waits for the work that was in flight at the failure
    }
  }

  void StartThreads() {
    for (size_t d = 0; d < devices.size(); d++) dispatchers.emplace_back(&Impl::DispatchLoop, this, d);
    for (size_t i = 0; i < num_threads; i++) workers.emplace_back(&Impl::WorkLoop, this);
  }

  void StopThreads() {
    {
      std::lock_guard<std::mutex> l(m);
      stop = true;
    }
    for (auto& cv : cv_dispatch) cv->notify_all();
    cv_work.notify_all();
    for (auto& t : dispatchers) t.join();
    dispatchers.clear();
    for (auto& w : workers) w.join();
    workers.clear();
  }

  void ReleaseContexts() {
    for (size_t i = 0; i < ctxs.size(); i++) {
      if (failed.load()) DiscardEncoderCtx(ctxs[i]);  // a context that saw an error is not cached for the next session
      else ReleaseEncoderCtx(ctxs[i]);
    }
    ctxs.clear();
  }
};

Encoder::Encoder(size_t num_threads, int shift_to_left_align, bool big_endian)
    : impl_(new Impl(num_threads, shift_to_left_align, big_endian)) {
  WarmUpRequest(kWarmForward);  // FPV_WARMUP=lazy: the device warm-up starts now, on helper threads
}

const char* Encoder::LastError() const { return impl_->failed.load() ? impl_->error.c_str() : ""; }

Encoder::~Encoder() {
  if (!impl_) return;
  if (impl_->initialized && !impl_->finished) {
    // The upstream class leaks its threads here; drain quietly instead.
    Finish([](const uint8_t*, size_t, void*) {}, nullptr);
  }
}

size_t Encoder::MaxQueued() const { return impl_->max_queued; }

size_t Encoder::LastStats(double* out, size_t n) const {
  const size_t k = std::min<size_t>(n, sizeof impl_->last_stats / sizeof impl_->last_stats[0]);
  for (size_t i = 0; i < k; i++) out[i] = impl_->last_stats[i];
  return k;
}

void Encoder::Init(const uint16_t* delta_frame, size_t xsize, size_t ysize,
                   Callback callback, void* payload) {
  Impl& s = *impl_;
  if (s.initialized) Fatal("Encoder::Init", "called twice");
  WarmUpWait(kWarmForward);
  const bool timing = getenv("FPV_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_start = now();
  s.st_t0 = std::chrono::steady_clock::now();
  s.xs = xsize;
  s.ys = ysize;
  s.npix = xsize * ysize;
  s.npv = (xsize / 4) * (ysize / 4);
  // bound the pinned staging memory per slot
  const size_t cap_bytes = EnvSize("FPV_BATCH_BYTES", (size_t)256 << 20);
  s.batch = std::max<size_t>(1, std::min(s.batch_promised, cap_bytes / std::max<size_t>(1, 2 * s.npix)));
  if (s.num_threads == 0) s.slots_per_dev = 1;
  s.initialized = true;  // from here on a failure is a latched session state, not API misuse
  if (fpvhip_device_count() <= 0) { s.Fail("no usable GPU", fpvhip_last_error(nullptr)); return; }
  s.devices = DeviceList();
  if (s.num_threads == 0) s.devices.resize(1);
  {
    // One context per device, created side by side (device buffers, pinned slot 0, the delta
    // frame's upload and split: a few ms each, times eight on a full node when done in turn).
    const size_t nd = s.devices.size();
    s.ctxs.assign(nd, nullptr);
    std::vector<std::string> errs(nd);
    auto make = [&](size_t i) {
      fpvhip_ctx* c = AcquireEncoderCtx(s.devices[i], (uint32_t)xsize, (uint32_t)ysize, s.shift, s.big_endian,
                                        (uint32_t)s.batch, s.slots_per_dev);
      if (!c) { errs[i] = std::string("fpvhip_create: ") + fpvhip_last_error(nullptr); return; }
      s.ctxs[i] = c;
      if (fpvhip_set_delta_raw(c, delta_frame)) errs[i] = std::string("fpvhip_set_delta_raw: ") + fpvhip_last_error(c);
    };
    std::vector<std::thread> th;
    for (size_t i = 1; i < nd; i++) th.emplace_back(make, i);
    make(0);
    for (auto& t : th) t.join();
    for (size_t i = 0; i < nd; i++)
      if (!errs[i].empty()) {
        // contexts that exist are released by Finish / the destructor through ReleaseContexts
        s.ctxs.erase(std::remove(s.ctxs.begin(), s.ctxs.end(), nullptr), s.ctxs.end());
        s.Fail("Encoder::Init", errs[i].c_str());
        return;
      }
  }
  s.gpu_q.assign(s.devices.size(), std::deque<Impl::Batch*>());
  for (size_t d = 0; d < s.devices.size(); d++) s.cv_dispatch.emplace_back(new std::condition_variable());
  s.slot_busy.assign(s.devices.size(), std::vector<char>(s.slots_per_dev, 0));
  s.slot_prepared.assign(s.devices.size(), std::vector<char>(s.slots_per_dev, 0));
  s.ready_q.assign(s.devices.size(), std::deque<Impl::Batch*>());
  s.gpu_inflight.assign(s.devices.size(), 0);
  s.gpu_inflight_max = (int)std::max<size_t>(1, EnvSize("FPV_GPU_INFLIGHT", 3));
  for (auto& v : s.slot_prepared) v[0] = 1;  // slot 0 comes with the context
  const double t_ctx = now();

  // The delta frame itself: predicted WITHOUT a delta frame (cc:1099-1100),
  // written as header | u32 size | 0x01 | image core (cc:1092-1103).
  fpvhip_ctx* c0 = s.ctxs[0];
  uint16_t* stage0 = fpvhip_stage_in(c0, 0, 0);
  if (!stage0) { s.Fail("fpvhip_stage_in", fpvhip_last_error(c0)); return; }
  memcpy(stage0, delta_frame, s.npix * 2);
  const uint8_t *high, *low, *pv, *flags;
  if (!s.CheckHip(fpvhip_predict_submit(c0, 0, 1, 0), c0, "fpvhip_predict_submit") ||
      !s.CheckHip(fpvhip_predict_wait(c0, 0, &high, &low, &pv, &flags), c0, "fpvhip_predict_wait"))
    return;
  const double t_pred = now();
  std::vector<uint8_t> out;
  out.reserve(s.npix + 64);
  Append32((uint32_t)xsize, &out);
  Append32((uint32_t)ysize, &out);
  Append32(0, &out);
  out.push_back(1);
  if (s.num_threads > 0) {
    s.StartThreads();  // the workers spawn while the delta frame is compressed ...
    std::lock_guard<std::mutex> l(s.m);
    for (size_t d = 0; d < s.devices.size(); d++) s.QueuePrepLocked(d, 2);  // ... and prepare the next slots
  }
  {
    // image core (cc:820-828): flags | [brotli(low)] | brotli(high); the two planes are
    // compressed side by side when worker threads are wanted at all
    const uint8_t f = flags[0];
    std::vector<uint8_t> low_stream;
    std::thread low_thread;
    const bool with_low = !(f & NO_LOW_BYTES);
    if (with_low && s.num_threads > 0)
      low_thread = std::thread([&] { if (!BrotliAppend(low, s.npix, &low_stream)) Fatal("Brotli", "low plane"); });
    out.push_back(f);
    std::vector<uint8_t> high_stream;
    if (low_thread.joinable()) {
      if (!BrotliAppend(high, s.npix, &high_stream)) Fatal("Brotli", "high plane");
      low_thread.join();
      out.insert(out.end(), low_stream.begin(), low_stream.end());
      out.insert(out.end(), high_stream.begin(), high_stream.end());
    } else {
      if (with_low && !BrotliAppend(low, s.npix, &out)) Fatal("Brotli", "low plane");
      if (!BrotliAppend(high, s.npix, &out)) Fatal("Brotli", "high plane");
    }
  }
  Store32((uint32_t)(out.size() - 8), out.data() + 8);
  s.bytes_written = out.size();
  if (timing)
    fprintf(stderr, "fpv-mi355x: Init: contexts + delta upload %.1f ms, delta frame predict %.1f ms, threads + Brotli %.1f ms "
                    "(batch %zu, %u slots, %zu device(s))\n",
            t_ctx - t_start, t_pred - t_ctx, now() - t_pred, s.batch, s.slots_per_dev, s.devices.size());
  callback(out.data(), out.size(), payload);
}

void Encoder::CompressFrame(const uint16_t* img, Callback callback, void* payload) {
  Impl& s = *impl_;
  if (!s.initialized || s.finished) Fatal("Encoder::CompressFrame", "Init not called or already finished");
  if (s.failed.load()) return;  // latched failure: the frame is dropped, its callback never runs (LastError)

  if (s.num_threads == 0) {  // synchronous path (cc:1136-1141)
    fpvhip_ctx* c = s.ctxs[0];
    memcpy(fpvhip_stage_in(c, 0, 0), img, s.npix * 2);
    const uint8_t *high, *low, *pv, *flags;
    if (!s.CheckHip(fpvhip_predict_submit(c, 0, 1, 1), c, "fpvhip_predict_submit") ||
        !s.CheckHip(fpvhip_predict_wait(c, 0, &high, &low, &pv, &flags), c, "fpvhip_predict_wait"))
      return;
    std::vector<uint8_t> chunk;
    BuildFrameChunk(high, low, pv, flags[0], s.npix, s.npv, &chunk);
    s.offsets.push_back(s.bytes_written);
    s.bytes_written += chunk.size();
    s.next_id++;
    callback(chunk.data(), chunk.size(), payload);
    return;
  }

  std::unique_lock<std::mutex> l(s.m);
  if (!s.cur) s.cur = s.OpenBatchLocked(l);
  if (!s.cur) return;  // the session failed while waiting for a slot
  Impl::Batch* b = s.cur;
  // The pixels are copied into pinned staging by a worker thread: the caller
  // keeps `img` alive until this frame's callback, as the API demands.
  s.copy_q.push_back(Impl::Task{b, b->n, img});
  b->cbs.push_back(std::move(callback));
  b->payloads.push_back(payload);
  b->n++;
  s.next_id++;
  s.inflight++;
  s.cur_touched = std::chrono::steady_clock::now();
  s.cv_work.notify_one();
  // Close when full -- or when we are about to block on the in-flight limit,
  // because the frames of a partial batch would otherwise never complete.
  if (b->n == s.batch || s.inflight >= s.max_queued) {
    s.cur = nullptr;
    if (s.CloseLocked(b)) {
      l.unlock();
      s.Submit(b);
      l.lock();
    }
  }
  s.cv_producer.wait(l, [&s] { return s.inflight < s.max_queued || s.failed.load(); });  // cc:1150-1156
}

void Encoder::Finish(Callback callback, void* payload) {
  Impl& s = *impl_;
  if (s.finished) return;  // cc:1111
  if (!s.initialized) Fatal("Encoder::Finish", "Init not called");
  if (s.num_threads > 0) {
    {
      std::unique_lock<std::mutex> l(s.m);
      if (s.cur && !s.failed.load()) {
        Impl::Batch* b = s.cur;
        s.cur = nullptr;
        if (s.CloseLocked(b)) {  // never empty: a batch is opened by its first frame
          l.unlock();
          s.Submit(b);
          l.lock();
        }
      }
      // (after a failure: until the batches that came back before it have been compressed and emitted)
      s.cv_producer.wait(l, [&s] {
        return (s.inflight == 0 && s.emit_q.empty()) || (s.failed.load() && s.work_q.empty() && s.working == 0 && !s.emitting);
      });
    }
    s.StopThreads();
  }
  s.finished = true;
  if (s.failed.load()) {  // no index: the stream ends with the last chunk that was emitted
    for (Impl::Batch* b : s.emit_q) delete b;  // every open batch is in here, `cur` and the queued ones included
    s.emit_q.clear();
    s.cur = nullptr;
    for (auto& q : s.gpu_q) q.clear();
    for (auto& q : s.ready_q) q.clear();
    s.ReleaseContexts();
    return;
  }
  s.ReleaseContexts();
  // frame index (cc:1185-1197)
  const size_t n = s.offsets.size();
  if (n > 0 && s.num_threads > 0) {
    // Stage totals of the pipeline.  Brotli and staging copies are summed over the worker
    // threads; the GPU side is given as the mean latency of a batch (first transfer in ->
    // last transfer out, HIP events: batches of different slots overlap on the transfer
    // streams, so their latencies do not add up to a busy time) and as the bytes moved each
    // way per second of wall time.  With perfect overlap the wall time equals
    // (brotli + copies) / threads, the largest stage by far.
    const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - s.st_t0).count();
    const double gpu = s.st_gpu_us / 1e3, brotli = s.st_brotli_ns / 1e6, copy = s.st_copy_ns / 1e6;
    const double host_stage = (brotli + copy) / (double)s.num_threads;
    const double batches = (double)std::max<uint64_t>(1, s.st_batches.load());
    double* o = s.last_stats;
    o[0] = (double)n; o[1] = wall; o[2] = brotli; o[3] = copy; o[4] = gpu / batches; o[5] = batches;
    o[6] = s.st_bytes_up / (wall * 1e6); o[7] = s.st_bytes_down / (wall * 1e6);  // GB/s
    o[8] = host_stage / wall; o[9] = s.st_gpu_wait_ns / 1e6;
    const size_t nd = std::min<size_t>(s.devices.size(), (size_t)Impl::kMaxStatDevs);
    o[10] = (double)nd;
    for (size_t d = 0; d < nd; d++) {
      o[11 + 2 * d] = (double)s.st_dev_batches[d].load();
      o[12 + 2 * d] = s.st_dev_wait_ns[d].load() / 1e6;
    }
    if (s.print_stats)
      fprintf(stderr,
              "fpv-mi355x: stats: %zu frames %zux%zu, wall %.1f ms (%.1f fps); Brotli %.1f ms + staging copies %.1f ms over %zu "
              "threads = %.3f ms/frame/thread; GPU: %.0f batches, mean latency %.2f ms (copy in + kernels + copy out), "
              "%.1f GB/s in, %.1f GB/s out over the wall time; dispatcher blocked on the GPU %.1f ms; overlap: host stage / wall = %.2f\n",
              n, s.xs, s.ys, wall, n / wall * 1e3, brotli, copy, s.num_threads, (brotli + copy) / n, batches, gpu / batches,
              o[6], o[7], s.st_gpu_wait_ns / 1e6, host_stage / wall);
  }
  std::vector<uint8_t> out(5 + 8 * n + 8);
  Store32((uint32_t)out.size(), out.data());
  out[4] = 2;
  for (size_t i = 0; i < n; i++) Store64(s.offsets[i], out.data() + 5 + 8 * i);
  Store64(n, out.data() + 5 + 8 * n);
  callback(out.data(), out.size(), payload);
}

}  // namespace fpvc
