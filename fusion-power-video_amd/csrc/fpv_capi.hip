// fpv_capi.hip -- implementation of the C ABI in include/fpv_hip.h: contexts,
// pinned staging sets, HIP streams/events and the launches of fpv_kernels.hip.
// No CPU fallback exists: every entry point fails when HIP does.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>

#include <atomic>
#include <mutex>

#include <new>
#include <vector>

#include "fpv_hip.h"
#include "fpv_kernels.h"

using fpvhip::FwdWorkspace;
using fpvhip::Geom;

namespace {

thread_local char g_err[256] = "";

// Host staging memory of one slot: ONE region, anonymous pages (transparent huge pages
// where the system grants them) touched once and then registered with the runtime.
// On this platform that is 5-20x cheaper than hipHostMalloc of the same size
// (tools/exp/pin_bench.cc: 66 MB in 1-3 ms against 11-13 ms) and copies run at the
// same rate; hipHostMalloc is the fallback when registration is refused.
struct PinnedRegion {
  void* base = nullptr;
  size_t bytes = 0;
  bool registered = false;  // mmap + hipHostRegister (else hipHostMalloc)
};

int pinned_alloc(PinnedRegion* r, size_t bytes) {
  r->bytes = (bytes + 4095) & ~(size_t)4095;
  if (!getenv("FPV_NO_HOST_REGISTER")) {
    void* p = mmap(nullptr, r->bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p != MAP_FAILED) {
      madvise(p, r->bytes, MADV_HUGEPAGE);
      volatile char* c = static_cast<volatile char*>(p);
      for (size_t o = 0; o < r->bytes; o += 4096) c[o] = 0;  // populate before pinning: registering untouched pages is slow
      if (hipHostRegister(p, r->bytes, hipHostRegisterDefault) == hipSuccess) {
        r->base = p;
        r->registered = true;
        return 0;
      }
      (void)hipGetLastError();
      munmap(p, r->bytes);
    }
  }
  r->registered = false;
  return hipHostMalloc(&r->base, r->bytes, hipHostMallocDefault) == hipSuccess ? 0 : -1;
}

void pinned_free(PinnedRegion* r) {
  if (!r->base) return;
  if (r->registered) {
    hipHostUnregister(r->base);
    munmap(r->base, r->bytes);
  } else {
    hipHostFree(r->base);
  }
  *r = PinnedRegion();
}

struct Slot {
  hipEvent_t done = nullptr;     // last transfer out has finished
  hipEvent_t begin = nullptr;    // recorded in front of a batch's first transfer (stage timing)
  hipEvent_t up_done = nullptr;  // transfers in have finished
  hipEvent_t k_done = nullptr;   // kernels have finished
  PinnedRegion host;
  void* dev = nullptr;           // one device allocation, carved like the host region
  // pinned host
  uint16_t* h_pix = nullptr;  // [B][npix]  forward input / inverse output
  uint8_t* h_high = nullptr;  // [B][npix]  forward output / inverse input
  uint8_t* h_low = nullptr;   // [B][npix]
  uint8_t* h_pv = nullptr;    // [B][npv]
  uint8_t* h_flags = nullptr; // [B]
  uint16_t* h_delta = nullptr; // [npix]    staging of a raw delta frame (never a batch's data)
  // device
  uint16_t* d_pix = nullptr;
  uint8_t* d_high = nullptr;
  uint8_t* d_low = nullptr;
  uint8_t* d_pv = nullptr;
  uint8_t* d_flags = nullptr;
  void* d_ws = nullptr;
  FwdWorkspace ws{};
  uint32_t pending = 0;  // frames of the batch in flight
  std::atomic<bool> ready{false};  // buffers allocated (slots after the first are allocated on first use)
  std::mutex alloc_m;              // one allocator at a time per slot; different slots allocate side by side
};

// Streams are expensive to create (3-10 ms each on this platform) and do not depend on the
// frame geometry: a process-wide pool per device, filled by fpvhip_warmup, lends them to
// the contexts.
struct StreamPool {
  std::mutex m;
  std::vector<hipStream_t> idle[64];
};
StreamPool& stream_pool() {
  static StreamPool* p = new StreamPool();  // leaked on purpose (no HIP calls during static destruction)
  return *p;
}
hipStream_t take_stream(int dev) {
  StreamPool& p = stream_pool();
  {
    std::lock_guard<std::mutex> l(p.m);
    auto& v = p.idle[dev & 63];
    if (!v.empty()) {
      hipStream_t s = v.back();
      v.pop_back();
      return s;
    }
  }
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
  return s;
}
void give_stream(int dev, hipStream_t s) {
  if (!s) return;
  hipStreamSynchronize(s);
  StreamPool& p = stream_pool();
  std::lock_guard<std::mutex> l(p.m);
  if (p.idle[dev & 63].size() < 16) p.idle[dev & 63].push_back(s);
  else hipStreamDestroy(s);
}


}  // namespace

struct fpvhip_ctx {
  int device = 0;
  Geom g{};
  uint32_t max_batch = 0;
  Slot* slots = nullptr;
  uint32_t nslots = 0;
  // All transfers in share one stream and all transfers out another: the two directions of
  // the host link then run at the same time (48 GB/s each way measured against 23-29 GB/s
  // when every slot copies in and out on its own stream).  Events chain a batch's
  // copy in -> kernels (one compute stream per context) -> copy out.
  hipStream_t up = nullptr, comp = nullptr, down = nullptr;
  std::mutex xfer_m;  // enqueue order on the three streams
  // delta frame
  uint8_t* d_dhigh = nullptr;
  uint8_t* d_dlow = nullptr;
  uint16_t* d_dtmp = nullptr;  // upload buffer for set_delta_*
  uint32_t* d_inv_diag = nullptr;  // speculative inverse: supersteps, passes (diagnostics)
  // The inverse's counters as they were behind the last launch that has finished (pinned; copied behind every launch, read without
  // waiting) and what the last decision saw: a batch takes the kernel form with the memo behind the second pass when more than a
  // sixteenth of the supersteps since then ran into the long-correction path (smooth frames, a dark sensor: fpv_inverse.inc).
  volatile uint32_t* h_inv_diag = nullptr;
  uint32_t inv_seen_steps = 0, inv_seen_long = 0;
  bool inv_memo2 = false;
  bool have_delta = false;
  int timing_left = 3;  // FPV_TIMING: the first few batches are reported
  // workspace of the device-resident entry points
  void* d_ws = nullptr;
  FwdWorkspace ws{};
  hipEvent_t ev[6] = {};
  char err[256] = "";
};

namespace {

int fail(fpvhip_ctx* c, const char* fmt, ...) {
  char buf[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  snprintf(g_err, sizeof g_err, "%s", buf);
  if (c) snprintf(c->err, sizeof c->err, "%s", buf);
  return -1;
}

#define HIPOK(c, call)                                                        \
  do {                                                                        \
    hipError_t e_ = (call);                                                   \
    if (e_ != hipSuccess)                                                     \
      return fail((c), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                        \
  } while (0)

double now_ms() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
}
const bool g_timing = getenv("FPV_TIMING") != nullptr;

void free_slot(Slot* s);

int alloc_slot_impl(fpvhip_ctx* c, Slot* s) {
  const double t_begin = now_ms();
  const size_t B = c->max_batch, np = c->g.npix, nv = c->g.npv;
  auto up16 = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_pix = 0, o_high = o_pix + up16(B * np * 2), o_low = o_high + up16(B * np),
               o_pv = o_low + up16(B * np), o_flags = o_pv + up16(B * nv), o_delta = o_flags + up16(B),
               total = o_delta + up16(np * 2);
  HIPOK(c, hipEventCreate(&s->done));
  HIPOK(c, hipEventCreate(&s->begin));
  HIPOK(c, hipEventCreateWithFlags(&s->up_done, hipEventDisableTiming));
  HIPOK(c, hipEventCreateWithFlags(&s->k_done, hipEventDisableTiming));
  const double t_host = now_ms();
  if (pinned_alloc(&s->host, total)) return fail(c, "cannot allocate %zu bytes of pinned staging memory", total);
  const double t_host_done = now_ms();
  uint8_t* h = static_cast<uint8_t*>(s->host.base);
  s->h_pix = reinterpret_cast<uint16_t*>(h + o_pix);
  s->h_high = h + o_high; s->h_low = h + o_low; s->h_pv = h + o_pv; s->h_flags = h + o_flags;
  s->h_delta = reinterpret_cast<uint16_t*>(h + o_delta);
  HIPOK(c, hipMalloc(&s->dev, o_delta));
  uint8_t* d = static_cast<uint8_t*>(s->dev);
  s->d_pix = reinterpret_cast<uint16_t*>(d + o_pix);
  s->d_high = d + o_high; s->d_low = d + o_low; s->d_pv = d + o_pv; s->d_flags = d + o_flags;
  if (c->g.fwd_ok) {
    HIPOK(c, hipMalloc(&s->d_ws, fpvhip::fwd_workspace_bytes(c->g, c->max_batch)));
    fpvhip::carve_fwd_workspace(c->g, c->max_batch, s->d_ws, &s->ws);
    std::lock_guard<std::mutex> l(c->xfer_m);
    HIPOK(c, fpvhip::init_fwd_workspace(s->ws, c->comp));  // stream order: ahead of the slot's first batch
  }
  s->ready.store(true, std::memory_order_release);
  if (g_timing)
    fprintf(stderr, "fpv-mi355x: slot: events %.2f ms, %zu MB pinned (%s) %.2f ms, device + workspace %.2f ms\n",
            t_host - t_begin, total >> 20, s->host.registered ? "mmap + register" : "hipHostMalloc", t_host_done - t_host,
            now_ms() - t_host_done);
  return 0;
}

// Gives back everything a slot holds and leaves it empty (a later check_slot allocates it
// again; a slot whose allocation failed half-way must not keep what it got: the retry would
// overwrite, i.e. leak, it).
void free_slot(Slot* s) {
  pinned_free(&s->host);
  if (s->dev) hipFree(s->dev);
  if (s->d_ws) hipFree(s->d_ws);
  s->dev = nullptr;
  s->d_ws = nullptr;
  for (hipEvent_t* e : {&s->done, &s->begin, &s->up_done, &s->k_done})
    if (*e) { hipEventDestroy(*e); *e = nullptr; }
  s->h_pix = nullptr; s->h_high = s->h_low = s->h_pv = s->h_flags = nullptr; s->h_delta = nullptr;
  s->d_pix = nullptr; s->d_high = s->d_low = s->d_pv = s->d_flags = nullptr;
}

int alloc_slot(fpvhip_ctx* c, Slot* s) {
  const int rc = alloc_slot_impl(c, s);
  if (rc != 0) free_slot(s);
  return rc;
}

// Only slot 0 is allocated by create; the others when they are first touched,
// which in the encoder pipeline overlaps with the GPU and Brotli work of the
// batches before (a slot costs a few ms: see PinnedRegion).
int check_slot(fpvhip_ctx* c, uint32_t slot) {
  if (!c) return fail(nullptr, "null context");
  if (slot >= c->nslots) return fail(c, "slot %u out of range", slot);
  if (c->slots[slot].ready.load(std::memory_order_acquire)) return 0;
  std::lock_guard<std::mutex> l(c->slots[slot].alloc_m);
  if (c->slots[slot].ready.load(std::memory_order_acquire)) return 0;
  if (hipSetDevice(c->device) != hipSuccess) return fail(c, "hipSetDevice failed");
  return alloc_slot(c, &c->slots[slot]);
}

}  // namespace

extern "C" {

int fpvhip_abi_version(void) { return FPVHIP_ABI_VERSION; }

int fpvhip_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    fail(nullptr, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    return -1;
  }
  return n;
}

const char* fpvhip_last_error(const fpvhip_ctx* ctx) {
  return ctx ? ctx->err : g_err;
}

static fpvhip_ctx* create_impl(int device, const Geom& g, uint32_t max_batch,
                               uint32_t slots) {
  if (max_batch == 0 || max_batch > 65535) {
    fail(nullptr, "max_batch must be 1..65535");
    return nullptr;
  }
  int ndev = fpvhip_device_count();
  if (ndev <= 0) {
    if (ndev == 0) fail(nullptr, "no HIP device visible");
    return nullptr;
  }
  if (device < 0 || device >= ndev) {
    fail(nullptr, "device %d out of range (%d visible)", device, ndev);
    return nullptr;
  }
  fpvhip_ctx* c = new (std::nothrow) fpvhip_ctx();
  if (!c) { fail(nullptr, "out of memory"); return nullptr; }
  c->device = device;
  c->g = g;
  c->max_batch = max_batch;
  auto setup = [&]() -> int {
    const double t0 = now_ms();
    HIPOK(c, hipSetDevice(device));
    HIPOK(c, hipMalloc(reinterpret_cast<void**>(&c->d_dhigh), g.npix));
    HIPOK(c, hipMalloc(reinterpret_cast<void**>(&c->d_dlow), g.npix));
    HIPOK(c, hipMalloc(reinterpret_cast<void**>(&c->d_dtmp), (size_t)g.npix * 2));
    HIPOK(c, hipMalloc(reinterpret_cast<void**>(&c->d_inv_diag), 64));
    HIPOK(c, hipMemset(c->d_inv_diag, 0, 64));
    {
      void* hp = nullptr;
      HIPOK(c, hipHostMalloc(&hp, 64, hipHostMallocDefault));
      memset(hp, 0, 64);
      c->h_inv_diag = static_cast<volatile uint32_t*>(hp);
    }
    if (g.fwd_ok) {
      HIPOK(c, hipMalloc(&c->d_ws, fpvhip::fwd_workspace_bytes(g, max_batch)));
      fpvhip::carve_fwd_workspace(g, max_batch, c->d_ws, &c->ws);
      HIPOK(c, fpvhip::init_fwd_workspace(c->ws, nullptr));
      HIPOK(c, hipStreamSynchronize(nullptr));
    }
    for (auto& e : c->ev) HIPOK(c, hipEventCreate(&e));
    const double t1 = now_ms();
    c->up = take_stream(device); c->comp = take_stream(device); c->down = take_stream(device);
    if (!c->up || !c->comp || !c->down) return fail(c, "cannot create HIP streams");
    c->slots = new (std::nothrow) Slot[slots ? slots : 1];
    if (!c->slots) return fail(c, "out of memory");
    c->nslots = slots;
    const double t2 = now_ms();
    if (slots && alloc_slot(c, &c->slots[0])) return -1;
    if (g_timing)
      fprintf(stderr, "fpv-mi355x: create: device buffers + workspace + events %.2f ms, streams %.2f ms, slot 0 %.2f ms\n",
              t1 - t0, t2 - t1, now_ms() - t2);
    return 0;
  };
  if (setup()) {
    snprintf(g_err, sizeof g_err, "%s", c->err);
    fpvhip_destroy(c);
    return nullptr;
  }
  return c;
}

fpvhip_ctx* fpvhip_create(int device, uint32_t xs, uint32_t ys, int shift,
                          int big_endian, uint32_t max_batch, uint32_t slots) {
  Geom g;
  const char* why = "";
  if (!fpvhip::make_geom(xs, ys, shift, big_endian, &g, &why)) {
    fail(nullptr, "unsupported geometry %ux%u shift %d: %s", xs, ys, shift, why);
    return nullptr;
  }
  return create_impl(device, g, max_batch, slots);
}

fpvhip_ctx* fpvhip_create_decoder(int device, uint32_t xs, uint32_t ys,
                                  uint32_t max_batch, uint32_t slots) {
  Geom g;
  const char* why = "";
  if (!fpvhip::make_geom_decoder(xs, ys, &g, &why)) {
    fail(nullptr, "unsupported geometry %ux%u: %s", xs, ys, why);
    return nullptr;
  }
  return create_impl(device, g, max_batch, slots);
}

void fpvhip_destroy(fpvhip_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  for (hipStream_t st : {c->up, c->comp, c->down})
    if (st) hipStreamSynchronize(st);
  for (uint32_t k = 0; k < c->nslots; k++) free_slot(&c->slots[k]);
  delete[] c->slots;
  give_stream(c->device, c->up); give_stream(c->device, c->comp); give_stream(c->device, c->down);
  hipFree(c->d_dhigh); hipFree(c->d_dlow); hipFree(c->d_dtmp); hipFree(c->d_ws); hipFree(c->d_inv_diag);
  if (c->h_inv_diag) hipHostFree(const_cast<uint32_t*>(c->h_inv_diag));
  for (auto& e : c->ev) if (e) hipEventDestroy(e);
  delete c;
}

size_t fpvhip_npix(const fpvhip_ctx* c) { return c ? c->g.npix : 0; }
size_t fpvhip_npreview(const fpvhip_ctx* c) { return c ? c->g.npv : 0; }
uint32_t fpvhip_max_batch(const fpvhip_ctx* c) { return c ? c->max_batch : 0; }
uint32_t fpvhip_slots(const fpvhip_ctx* c) { return c ? c->nslots : 0; }

int fpvhip_prepare_slot(fpvhip_ctx* c, uint32_t slot) { return check_slot(c, slot); }

size_t fpvhip_staging_bytes(const fpvhip_ctx* c) {
  size_t b = 0;
  if (c)
    for (uint32_t k = 0; k < c->nslots; k++)
      if (c->slots[k].ready.load(std::memory_order_acquire)) b += c->slots[k].host.bytes;
  return b;
}

int fpvhip_trim(fpvhip_ctx* c) {
  if (!c) return fail(nullptr, "null context");
  HIPOK(c, hipSetDevice(c->device));
  for (hipStream_t st : {c->up, c->comp, c->down}) HIPOK(c, hipStreamSynchronize(st));
  for (uint32_t k = 1; k < c->nslots; k++) {
    Slot& s = c->slots[k];
    std::lock_guard<std::mutex> l(s.alloc_m);
    if (!s.ready.load(std::memory_order_acquire) || s.pending) continue;
    s.ready.store(false, std::memory_order_release);
    free_slot(&s);
    s.done = s.begin = s.up_done = s.k_done = nullptr;
    s.dev = s.d_ws = nullptr;
  }
  return 0;
}

int fpvhip_set_delta_raw(fpvhip_ctx* c, const uint16_t* delta_raw) {
  if (!c || !delta_raw) return fail(c, "null argument");
  if (!c->g.fwd_ok) return fail(c, "decoder-only context");
  const double t0 = now_ms();
  HIPOK(c, hipSetDevice(c->device));
  // through pinned staging when a slot exists: the first pageable copy of a process costs ~7 ms
  if (c->nslots && c->slots[0].ready.load()) {
    memcpy(c->slots[0].h_delta, delta_raw, (size_t)c->g.npix * 2);
    HIPOK(c, hipMemcpy(c->d_dtmp, c->slots[0].h_delta, (size_t)c->g.npix * 2, hipMemcpyHostToDevice));
  } else {
    HIPOK(c, hipMemcpy(c->d_dtmp, delta_raw, (size_t)c->g.npix * 2, hipMemcpyHostToDevice));
  }
  const double t1 = now_ms();
  HIPOK(c, fpvhip::launch_split(c->g, c->d_dtmp, c->d_dhigh, c->d_dlow, nullptr, nullptr));
  HIPOK(c, hipStreamSynchronize(nullptr));
  if (g_timing) fprintf(stderr, "fpv-mi355x: set_delta_raw: copy %.2f ms, split %.2f ms\n", t1 - t0, now_ms() - t1);
  c->have_delta = true;
  return 0;
}

int fpvhip_set_delta_decoded(fpvhip_ctx* c, const uint16_t* delta_img) {
  if (!c || !delta_img) return fail(c, "null argument");
  HIPOK(c, hipSetDevice(c->device));
  HIPOK(c, hipMemcpy(c->d_dtmp, delta_img, (size_t)c->g.npix * 2, hipMemcpyHostToDevice));
  HIPOK(c, fpvhip::launch_split_decoded(c->g, c->d_dtmp, c->d_dhigh, c->d_dlow, nullptr));
  HIPOK(c, hipStreamSynchronize(nullptr));
  c->have_delta = true;
  return 0;
}

int fpvhip_set_delta_planes(fpvhip_ctx* c, const uint8_t* high, const uint8_t* low) {
  if (!c || !high) return fail(c, "null argument");
  HIPOK(c, hipSetDevice(c->device));
  HIPOK(c, hipMemcpy(c->d_dhigh, high, c->g.npix, hipMemcpyHostToDevice));
  if (low)
    HIPOK(c, hipMemcpy(c->d_dlow, low, c->g.npix, hipMemcpyHostToDevice));
  else
    HIPOK(c, hipMemset(c->d_dlow, 0, c->g.npix));
  c->have_delta = true;
  return 0;
}

int fpvhip_split_frame(fpvhip_ctx* c, const uint16_t* img, uint8_t* high, uint8_t* low,
                       int* no_low) {
  if (!c || !img || !high || !no_low) return fail(c, "null argument");
  if (!c->g.fwd_ok || c->nslots == 0) return fail(c, "needs a forward context with a slot");
  Slot& s = c->slots[0];
  const size_t np = c->g.npix;
  HIPOK(c, hipSetDevice(c->device));
  uint32_t any = 0;
  // `any` lives on this stack frame and the copy into it is asynchronous: whatever fails, the
  // stream is drained before the frame is left.
  auto queue = [&]() -> int {
    std::lock_guard<std::mutex> l(c->xfer_m);
    HIPOK(c, hipMemcpyAsync(s.d_pix, img, np * 2, hipMemcpyHostToDevice, c->comp));
    HIPOK(c, hipMemsetAsync(s.ws.anylow, 0, 4, c->comp));
    HIPOK(c, fpvhip::launch_split(c->g, s.d_pix, s.d_high, s.d_low, s.ws.anylow, c->comp));
    HIPOK(c, hipMemcpyAsync(high, s.d_high, np, hipMemcpyDeviceToHost, c->comp));
    if (low && c->g.shift != 8)
      HIPOK(c, hipMemcpyAsync(low, s.d_low, np, hipMemcpyDeviceToHost, c->comp));
    HIPOK(c, hipMemcpyAsync(&any, s.ws.anylow, 4, hipMemcpyDeviceToHost, c->comp));
    HIPOK(c, hipMemsetAsync(s.ws.anylow, 0, 4, c->comp));  // the predictor expects its accumulators clean
    return 0;
  };
  const int rc = queue();
  const hipError_t es = hipStreamSynchronize(c->comp);
  if (rc != 0) return rc;
  if (es != hipSuccess) return fail(c, "hipStreamSynchronize failed: %s", hipGetErrorString(es));
  *no_low = any == 0;
  return 0;
}

uint16_t* fpvhip_stage_in(fpvhip_ctx* c, uint32_t slot, uint32_t k) {
  if (check_slot(c, slot) || k >= c->max_batch) return nullptr;
  return c->slots[slot].h_pix + (size_t)k * c->g.npix;
}

static int predict_submit_impl(fpvhip_ctx* c, uint32_t slot, uint32_t nframes, int with_delta,
                               bool from_planes);

int fpvhip_predict_submit(fpvhip_ctx* c, uint32_t slot, uint32_t nframes, int with_delta) {
  return predict_submit_impl(c, slot, nframes, with_delta, false);
}

int fpvhip_predict_planes_submit(fpvhip_ctx* c, uint32_t slot, uint32_t nframes, int with_delta) {
  if (c && (c->g.shift != 0 || c->g.swap)) return fail(c, "planes input needs a shift 0 / little-endian context");
  return predict_submit_impl(c, slot, nframes, with_delta, true);
}

static int predict_submit_impl(fpvhip_ctx* c, uint32_t slot, uint32_t nframes, int with_delta,
                               bool from_planes) {
  if (check_slot(c, slot)) return -1;
  if (nframes == 0 || nframes > c->max_batch) return fail(c, "bad nframes %u", nframes);
  if (with_delta && !c->have_delta) return fail(c, "delta frame not set");
  if (!c->g.fwd_ok) return fail(c, "decoder-only context");
  Slot& s = c->slots[slot];
  const size_t np = c->g.npix, nv = c->g.npv;
  HIPOK(c, hipSetDevice(c->device));
  const double t_sub = now_ms();
  std::lock_guard<std::mutex> l(c->xfer_m);
  HIPOK(c, hipEventRecord(s.begin, c->up));
  if (from_planes) {  // planes staged in h_high / h_low -> pixels on the device
    HIPOK(c, hipMemcpyAsync(s.d_high, s.h_high, nframes * np, hipMemcpyHostToDevice, c->up));
    HIPOK(c, hipMemcpyAsync(s.d_low, s.h_low, nframes * np, hipMemcpyHostToDevice, c->up));
  } else {
    HIPOK(c, hipMemcpyAsync(s.d_pix, s.h_pix, nframes * np * 2, hipMemcpyHostToDevice, c->up));
  }
  HIPOK(c, hipEventRecord(s.up_done, c->up));
  HIPOK(c, hipStreamWaitEvent(c->comp, s.up_done, 0));
  if (from_planes) HIPOK(c, fpvhip::launch_interleave(s.d_high, s.d_low, s.d_pix, nframes * np, c->comp));
  HIPOK(c, fpvhip::launch_predict(c->g, s.d_pix, nframes,
                                  with_delta ? c->d_dhigh : nullptr,
                                  with_delta ? c->d_dlow : nullptr, s.ws, s.d_high,
                                  s.d_low, s.d_pv, s.d_flags, c->comp, nullptr));
  HIPOK(c, hipEventRecord(s.k_done, c->comp));
  HIPOK(c, hipStreamWaitEvent(c->down, s.k_done, 0));
  HIPOK(c, hipMemcpyAsync(s.h_high, s.d_high, nframes * np, hipMemcpyDeviceToHost, c->down));
  if (c->g.shift != 8)
    HIPOK(c, hipMemcpyAsync(s.h_low, s.d_low, nframes * np, hipMemcpyDeviceToHost, c->down));
  HIPOK(c, hipMemcpyAsync(s.h_pv, s.d_pv, nframes * nv, hipMemcpyDeviceToHost, c->down));
  HIPOK(c, hipMemcpyAsync(s.h_flags, s.d_flags, nframes, hipMemcpyDeviceToHost, c->down));
  HIPOK(c, hipEventRecord(s.done, c->down));
  s.pending = nframes;
  if (g_timing && c->timing_left > 0) fprintf(stderr, "fpv-mi355x: predict_submit(%u frames): enqueue %.2f ms\n", nframes, now_ms() - t_sub);
  return 0;
}

int fpvhip_predict_wait(fpvhip_ctx* c, uint32_t slot, const uint8_t** high,
                        const uint8_t** low, const uint8_t** preview,
                        const uint8_t** flags) {
  if (check_slot(c, slot)) return -1;
  Slot& s = c->slots[slot];
  if (!s.pending) return fail(c, "slot %u has nothing in flight", slot);
  const double t_w = now_ms();
  HIPOK(c, hipEventSynchronize(s.done));
  if (g_timing && c->timing_left > 0) {
    c->timing_left--;
    float ms = 0;
    hipEventElapsedTime(&ms, s.begin, s.done);
    fprintf(stderr, "fpv-mi355x: predict_wait: blocked %.2f ms, device begin->done %.2f ms\n", now_ms() - t_w, ms);
  }
  s.pending = 0;
  if (high) *high = s.h_high;
  if (low) *low = s.h_low;
  if (preview) *preview = s.h_pv;
  if (flags) *flags = s.h_flags;
  return 0;
}

uint8_t* fpvhip_stage_plane(fpvhip_ctx* c, uint32_t slot, uint32_t k, int which) {
  if (check_slot(c, slot) || k >= c->max_batch) return nullptr;
  Slot& s = c->slots[slot];
  return (which ? s.h_low : s.h_high) + (size_t)k * c->g.npix;
}

static int inverse_submit_impl(fpvhip_ctx* c, uint32_t slot, uint32_t nframes,
                               const uint8_t* flags, bool to_planes, int unshift = -1,
                               int big_endian = 0);

int fpvhip_inverse_submit(fpvhip_ctx* c, uint32_t slot, uint32_t nframes, const uint8_t* flags) {
  return inverse_submit_impl(c, slot, nframes, flags, false);
}

int fpvhip_inverse_submit_raw(fpvhip_ctx* c, uint32_t slot, uint32_t nframes, const uint8_t* flags,
                              int unshift, int big_endian) {
  if (unshift < 0 || unshift > 15) return fail(c, "unshift must be 0..15");
  return inverse_submit_impl(c, slot, nframes, flags, false, unshift, big_endian);
}

int fpvhip_inverse_planes_submit(fpvhip_ctx* c, uint32_t slot, uint32_t nframes,
                                 const uint8_t* flags) {
  if (c && ((uint64_t)c->g.npix * nframes >= (1ull << 32) || (c->g.npix & 3)))
    return fail(c, "batch too large / odd size for plane output");
  return inverse_submit_impl(c, slot, nframes, flags, true);
}


// The hint for launch_inverse's memo2 form, from the counters the launches before left (never waits: what has not been copied back
// yet counts next time), and the copy behind a launch.
static bool inv_memo2_hint(fpvhip_ctx* c) {
  if (!c->h_inv_diag) return false;
  const uint32_t steps = c->h_inv_diag[0], lng = c->h_inv_diag[6];
  if (steps < c->inv_seen_steps || lng < c->inv_seen_long) { c->inv_seen_steps = steps; c->inv_seen_long = lng; }  // the counters were reset
  const uint32_t ds = steps - c->inv_seen_steps, dl = lng - c->inv_seen_long;
  if (ds >= 256) {   // enough supersteps since the last decision
    c->inv_memo2 = 16ull * dl > ds;
    c->inv_seen_steps = steps;
    c->inv_seen_long = lng;
  }
  return c->inv_memo2;
}
static hipError_t inv_diag_copy_back(fpvhip_ctx* c, hipStream_t s) {
  if (!c->h_inv_diag || getenv("FPV_TRIO_MEMO2")) return hipSuccess;   // (forced either way: nothing to decide)
  return hipMemcpyAsync(const_cast<uint32_t*>(c->h_inv_diag), c->d_inv_diag, 32, hipMemcpyDeviceToHost, s);
}

static int inverse_submit_impl(fpvhip_ctx* c, uint32_t slot, uint32_t nframes,
                               const uint8_t* flags, bool to_planes, int unshift, int big_endian) {
  if (check_slot(c, slot)) return -1;
  if (nframes == 0 || nframes > c->max_batch || !flags) return fail(c, "bad arguments");
  Slot& s = c->slots[slot];
  const size_t np = c->g.npix;
  bool any_low = false;
  for (uint32_t k = 0; k < nframes; k++) {
    s.h_flags[k] = flags[k];
    if (!(flags[k] & 4)) any_low = true;
    if ((flags[k] & 1) && !c->have_delta) return fail(c, "delta frame not given");  // cc:310
  }
  HIPOK(c, hipSetDevice(c->device));
  std::lock_guard<std::mutex> l(c->xfer_m);
  HIPOK(c, hipEventRecord(s.begin, c->up));
  HIPOK(c, hipMemcpyAsync(s.d_high, s.h_high, nframes * np, hipMemcpyHostToDevice, c->up));
  if (any_low)
    HIPOK(c, hipMemcpyAsync(s.d_low, s.h_low, nframes * np, hipMemcpyHostToDevice, c->up));
  HIPOK(c, hipMemcpyAsync(s.d_flags, s.h_flags, nframes, hipMemcpyHostToDevice, c->up));
  HIPOK(c, hipEventRecord(s.up_done, c->up));
  HIPOK(c, hipStreamWaitEvent(c->comp, s.up_done, 0));
  HIPOK(c, fpvhip::launch_inverse(c->g, s.d_high, s.d_low,
                                  c->have_delta ? c->d_dhigh : nullptr,
                                  c->have_delta ? c->d_dlow : nullptr, s.d_flags,
                                  nframes, s.d_pix, c->comp, c->d_inv_diag, inv_memo2_hint(c)));
  HIPOK(c, inv_diag_copy_back(c, c->comp));
  if (to_planes)  // (pixel >> 8, pixel & 0xff) back into the plane staging buffers
    HIPOK(c, fpvhip::launch_split_planes(s.d_pix, s.d_high, s.d_low, nframes * np, c->comp));
  else if (unshift >= 0)  // UnextractFrame (cc:850-862) fused in front of the copy back
    HIPOK(c, fpvhip::launch_unextract(s.d_pix, nframes * np, unshift, big_endian, c->comp));
  HIPOK(c, hipEventRecord(s.k_done, c->comp));
  HIPOK(c, hipStreamWaitEvent(c->down, s.k_done, 0));
  if (to_planes) {
    HIPOK(c, hipMemcpyAsync(s.h_high, s.d_high, nframes * np, hipMemcpyDeviceToHost, c->down));
    HIPOK(c, hipMemcpyAsync(s.h_low, s.d_low, nframes * np, hipMemcpyDeviceToHost, c->down));
  } else {
    HIPOK(c, hipMemcpyAsync(s.h_pix, s.d_pix, nframes * np * 2, hipMemcpyDeviceToHost, c->down));
  }
  HIPOK(c, hipEventRecord(s.done, c->down));
  s.pending = nframes;
  return 0;
}

int fpvhip_inverse_wait(fpvhip_ctx* c, uint32_t slot, const uint16_t** frames) {
  if (check_slot(c, slot)) return -1;
  Slot& s = c->slots[slot];
  if (!s.pending) return fail(c, "slot %u has nothing in flight", slot);
  HIPOK(c, hipEventSynchronize(s.done));
  s.pending = 0;
  if (frames) *frames = s.h_pix;
  return 0;
}

int fpvhip_predict_device(fpvhip_ctx* c, const void* d_frames, uint32_t nframes,
                          int with_delta, void* d_high, void* d_low,
                          void* d_preview, void* d_flags, void* stream) {
  if (!c) return fail(nullptr, "null context");
  if (!c->g.fwd_ok) return fail(c, "decoder-only context");
  if (!d_frames || !d_high || !d_preview || !d_flags || (!d_low && c->g.shift != 8))
    return fail(c, "null device pointer");
  if (nframes == 0 || nframes > c->max_batch) return fail(c, "bad nframes %u", nframes);
  if (with_delta && !c->have_delta) return fail(c, "delta frame not set");
  HIPOK(c, fpvhip::launch_predict(
               c->g, static_cast<const uint16_t*>(d_frames), nframes,
               with_delta ? c->d_dhigh : nullptr, with_delta ? c->d_dlow : nullptr,
               c->ws, static_cast<uint8_t*>(d_high), static_cast<uint8_t*>(d_low),
               static_cast<uint8_t*>(d_preview), static_cast<uint8_t*>(d_flags),
               static_cast<hipStream_t>(stream), nullptr));
  return 0;
}

int fpvhip_inverse_device(fpvhip_ctx* c, const void* d_high, const void* d_low,
                          const void* d_flags, uint32_t nframes, void* d_frames,
                          void* stream) {
  if (!c) return fail(nullptr, "null context");
  if (!d_high || !d_flags || !d_frames) return fail(c, "null device pointer");
  if (nframes == 0) return fail(c, "bad nframes");
  HIPOK(c, fpvhip::launch_inverse(
               c->g, static_cast<const uint8_t*>(d_high),
               static_cast<const uint8_t*>(d_low),
               c->have_delta ? c->d_dhigh : nullptr,
               c->have_delta ? c->d_dlow : nullptr,
               static_cast<const uint8_t*>(d_flags), nframes,
               static_cast<uint16_t*>(d_frames), static_cast<hipStream_t>(stream), c->d_inv_diag, inv_memo2_hint(c)));
  HIPOK(c, inv_diag_copy_back(c, static_cast<hipStream_t>(stream)));
  return 0;
}

double fpvhip_time_predict_device(fpvhip_ctx* c, const void* d_frames,
                                  uint32_t nframes, int with_delta, void* d_high,
                                  void* d_low, void* d_preview, void* d_flags,
                                  void* stream, int iters, double* kernel_ms) {
  if (!c || iters <= 0 || !c->g.fwd_ok) { fail(c, "bad arguments"); return -1.0; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto run = [&](hipEvent_t* ev) {
    return fpvhip::launch_predict(
        c->g, static_cast<const uint16_t*>(d_frames), nframes,
        with_delta ? c->d_dhigh : nullptr, with_delta ? c->d_dlow : nullptr, c->ws,
        static_cast<uint8_t*>(d_high), static_cast<uint8_t*>(d_low),
        static_cast<uint8_t*>(d_preview), static_cast<uint8_t*>(d_flags), st, ev);
  };
  if (with_delta && !c->have_delta) { fail(c, "delta frame not set"); return -1.0; }
  if (nframes == 0 || nframes > c->max_batch) { fail(c, "bad nframes"); return -1.0; }
  hipError_t e;
  if ((e = hipEventRecord(c->ev[0], st)) != hipSuccess) { fail(c, "event: %s", hipGetErrorString(e)); return -1.0; }
  for (int i = 0; i < iters; i++)
    if ((e = run(nullptr)) != hipSuccess) { fail(c, "launch: %s", hipGetErrorString(e)); return -1.0; }
  hipEventRecord(c->ev[5], st);
  if ((e = hipEventSynchronize(c->ev[5])) != hipSuccess) { fail(c, "sync: %s", hipGetErrorString(e)); return -1.0; }
  float ms = 0;
  hipEventElapsedTime(&ms, c->ev[0], c->ev[5]);
  if (kernel_ms) {
    for (int k = 0; k < 4; k++) kernel_ms[k] = 0;
    for (int i = 0; i < iters; i++) {
      if ((e = run(c->ev)) != hipSuccess) { fail(c, "launch: %s", hipGetErrorString(e)); return -1.0; }
      hipEventSynchronize(c->ev[4]);
      for (int k = 0; k < 4; k++) {
        float t = 0;
        hipEventElapsedTime(&t, c->ev[k], c->ev[k + 1]);
        kernel_ms[k] += t / iters;
      }
    }
  }
  return ms;
}

int fpvhip_warmup(int device) {
  const double t0 = now_ms();
  int n = fpvhip_device_count();
  if (n <= 0) return -1;
  if (device < 0 || device >= n) return fail(nullptr, "device %d out of range", device);
  const double t_count = now_ms();
  HIPOK(nullptr, hipSetDevice(device));
  HIPOK(nullptr, hipFree(nullptr));  // creates the primary context
  const double t_ctx = now_ms();
  void* p = nullptr;
  HIPOK(nullptr, hipMalloc(&p, 4096));
  const double t_malloc = now_ms();
  HIPOK(nullptr, fpvhip::launch_copy(p, p, 0, nullptr, 0));  // loads the code object
  HIPOK(nullptr, hipDeviceSynchronize());
  HIPOK(nullptr, hipFree(p));
  const double t_code = now_ms();
  // streams are slow to create (the first one of a process tens of ms, later ones 3-10 ms):
  // a context's worth of them goes into the pool; so does the first host registration
  {
    hipStream_t st[3] = {nullptr, nullptr, nullptr};
    for (auto& x : st) x = take_stream(device);
    for (auto& x : st) give_stream(device, x);
  }
  const double t_streams = now_ms();
  fpvhip::prepare_kernels();  // function attributes of the forward / inverse kernels on this device
  const double t_attr = now_ms();
  PinnedRegion r;
  if (pinned_alloc(&r, 1 << 16) == 0) pinned_free(&r);
  if (g_timing)
    fprintf(stderr, "fpv-mi355x: warmup(device %d): device count (runtime init) %.1f ms, primary context %.1f ms, first malloc %.1f ms, "
                    "code object + first launch %.1f ms, three streams %.1f ms, kernel attributes %.1f ms, first host registration %.1f ms\n",
            device, t_count - t0, t_ctx - t_count, t_malloc - t_ctx, t_code - t_malloc, t_streams - t_code, t_attr - t_streams,
            now_ms() - t_attr);
  return 0;
}

int fpvhip_debug_counters(fpvhip_ctx* c, uint32_t* out, uint32_t n) {
  if (!c || !out || n > 64 || !c->g.fwd_ok) return fail(c, "bad arguments");
  HIPOK(c, hipDeviceSynchronize());
  HIPOK(c, hipMemcpy(out, c->ws.ctr + ((c->ws.seq + 1) & 1u) * 64, n * 4, hipMemcpyDeviceToHost));  // the last batch's block
  return 0;
}

double fpvhip_slot_elapsed_ms(fpvhip_ctx* c, uint32_t slot) {
  if (!c || slot >= c->nslots || !c->slots[slot].ready.load() || c->slots[slot].pending) return -1.0;
  float ms = -1.f;
  if (hipEventElapsedTime(&ms, c->slots[slot].begin, c->slots[slot].done) != hipSuccess) return -1.0;
  return ms;
}

int fpvhip_debug_set_chain_mask(fpvhip_ctx* c, uint32_t value) {
  if (!c) return fail(nullptr, "null context");
  HIPOK(c, hipSetDevice(c->device));
  HIPOK(c, fpvhip::set_chain_simd_mask(value));
  return 0;
}

const char* fpvhip_forward_kernel_names(fpvhip_ctx* c) {
  if (!c || !c->g.fwd_ok) return "";
  if (c->g.tile8_ok) return "k_tile8,k_frame8,k_redo8";
  return c->g.tile_ok ? "k_tile,k_tile_decide,k_tile_finish" : "k_stream,k_redo_decide,k_redo_stream";
}

int fpvhip_forward_tile_shape(fpvhip_ctx* c, uint32_t out[4]) {
  if (!c || !out) return fail(c, "bad arguments");
  out[0] = out[1] = out[2] = out[3] = 0;
  if (c->g.fwd_ok && c->g.tile8_ok) { out[0] = c->g.tile8_groups; out[1] = c->g.tile8_threads; out[2] = c->g.tile8_tiles; }
  else if (c->g.fwd_ok && c->g.tile_ok) { out[0] = c->g.tile_groups; out[1] = c->g.tile_threads; out[2] = c->g.tile_tiles; out[3] = c->g.tile_lfgroup; }
  return 0;
}

const char* fpvhip_inverse_kernel_name(fpvhip_ctx* c, uint32_t nframes) {
  if (!c) return "";
  hipSetDevice(c->device);
  return fpvhip::inverse_kernel_name(c->g, nframes);
}

int fpvhip_inverse_pass_stats(fpvhip_ctx* c, uint32_t* out, int reset) {
  if (!c || !out) return fail(c, "bad arguments");
  HIPOK(c, hipSetDevice(c->device));
  HIPOK(c, hipDeviceSynchronize());
  HIPOK(c, hipMemcpy(out, c->d_inv_diag, 16, hipMemcpyDeviceToHost));
#ifdef FPV_EXP_WAITS  // experiment build: time at the barriers of the three-wave inverse (see k_inverse_trio)
  {
    uint32_t w[16];
    HIPOK(c, hipMemcpy(w, c->d_inv_diag, 64, hipMemcpyDeviceToHost));
    fprintf(stderr, "EXP_WAITS chain: at barriers %u of %u; i/o wave 1: at barriers %u, waiting for its planes %u, for residuals + stores in front of the barrier %u, of %u (x16 clocks, summed over frames)\n",
            w[4], w[5], w[8], w[9], w[10], w[7]);
  }
#endif
  if (reset) HIPOK(c, hipMemset(c->d_inv_diag, 0, 64));
  return 0;
}

double fpvhip_time_copy_shape(void* d_dst, const void* d_src, size_t bytes, void* stream,
                              int iters, int shape) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t a, b;
  if (iters <= 0 || hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1.0;
  fpvhip::launch_copy(d_src, d_dst, bytes, st, shape);  // warm
  hipEventRecord(a, st);
  for (int i = 0; i < iters; i++) fpvhip::launch_copy(d_src, d_dst, bytes, st, shape);
  hipEventRecord(b, st);
  float ms = -1.f;
  if (hipEventSynchronize(b) == hipSuccess) hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a);
  hipEventDestroy(b);
  return ms;
}

double fpvhip_time_copy_device(void* d_dst, const void* d_src, size_t bytes, void* stream,
                               int iters) {
  return fpvhip_time_copy_shape(d_dst, d_src, bytes, stream, iters, 1);
}

double fpvhip_time_inverse_device(fpvhip_ctx* c, const void* d_high,
                                  const void* d_low, const void* d_flags,
                                  uint32_t nframes, void* d_frames, void* stream,
                                  int iters) {
  if (!c || iters <= 0) { fail(c, "bad arguments"); return -1.0; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEventRecord(c->ev[0], st);
  for (int i = 0; i < iters; i++)
    if (fpvhip_inverse_device(c, d_high, d_low, d_flags, nframes, d_frames, stream)) return -1.0;
  hipEventRecord(c->ev[5], st);
  hipError_t e = hipEventSynchronize(c->ev[5]);
  if (e != hipSuccess) { fail(c, "sync: %s", hipGetErrorString(e)); return -1.0; }
  float ms = 0;
  hipEventElapsedTime(&ms, c->ev[0], c->ev[5]);
  return ms;
}

}  // extern "C"
