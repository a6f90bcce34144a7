// fpv_capi.hip -- implementation of the C ABI in include/fpv_hip.h: contexts,
// pinned staging sets, HIP streams/events and the launches of fpv_kernels.hip.
// No CPU fallback exists: every entry point fails when HIP does.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <vector>

#include "fpv_hip.h"
#include "fpv_kernels.h"

using fpvhip::FwdWorkspace;
using fpvhip::Geom;

namespace {

thread_local char g_err[256] = "";

struct Slot {
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
  hipEvent_t begin = nullptr;  // recorded in front of a batch's first transfer (stage timing)
  // pinned host
  uint16_t* h_pix = nullptr;  // [B][npix]  forward input / inverse output
  uint8_t* h_high = nullptr;  // [B][npix]  forward output / inverse input
  uint8_t* h_low = nullptr;   // [B][npix]
  uint8_t* h_pv = nullptr;    // [B][npv]
  uint8_t* h_flags = nullptr; // [B]
  // device
  uint16_t* d_pix = nullptr;
  uint8_t* d_high = nullptr;
  uint8_t* d_low = nullptr;
  uint8_t* d_pv = nullptr;
  uint8_t* d_flags = nullptr;
  void* d_ws = nullptr;
  FwdWorkspace ws{};
  uint32_t pending = 0;  // frames of the batch in flight
  bool ready = false;    // buffers allocated (slots after the first are allocated on first use)
};

}  // namespace

struct fpvhip_ctx {
  int device = 0;
  Geom g{};
  uint32_t max_batch = 0;
  std::vector<Slot> slots;
  // delta frame
  uint8_t* d_dhigh = nullptr;
  uint8_t* d_dlow = nullptr;
  uint16_t* d_dtmp = nullptr;  // upload buffer for set_delta_*
  uint32_t* d_inv_diag = nullptr;  // speculative inverse: supersteps, passes (diagnostics)
  bool have_delta = false;
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

int alloc_slot(fpvhip_ctx* c, Slot* s) {
  const size_t B = c->max_batch, np = c->g.npix, nv = c->g.npv;
  HIPOK(c, hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  HIPOK(c, hipEventCreate(&s->done));
  HIPOK(c, hipEventCreate(&s->begin));
  HIPOK(c, hipHostMalloc(reinterpret_cast<void**>(&s->h_pix), B * np * 2, hipHostMallocDefault));
  HIPOK(c, hipHostMalloc(reinterpret_cast<void**>(&s->h_high), B * np, hipHostMallocDefault));
  HIPOK(c, hipHostMalloc(reinterpret_cast<void**>(&s->h_low), B * np, hipHostMallocDefault));
  HIPOK(c, hipHostMalloc(reinterpret_cast<void**>(&s->h_pv), B * nv, hipHostMallocDefault));
  HIPOK(c, hipHostMalloc(reinterpret_cast<void**>(&s->h_flags), B, hipHostMallocDefault));
  HIPOK(c, hipMalloc(reinterpret_cast<void**>(&s->d_pix), B * np * 2));
  HIPOK(c, hipMalloc(reinterpret_cast<void**>(&s->d_high), B * np));
  HIPOK(c, hipMalloc(reinterpret_cast<void**>(&s->d_low), B * np));
  HIPOK(c, hipMalloc(reinterpret_cast<void**>(&s->d_pv), B * nv));
  HIPOK(c, hipMalloc(reinterpret_cast<void**>(&s->d_flags), B));
  if (c->g.fwd_ok) {
    HIPOK(c, hipMalloc(&s->d_ws, fpvhip::fwd_workspace_bytes(c->g, c->max_batch)));
    fpvhip::carve_fwd_workspace(c->g, c->max_batch, s->d_ws, &s->ws);
    HIPOK(c, fpvhip::init_fwd_workspace(s->ws, s->stream));
  }
  s->ready = true;
  return 0;
}

void free_slot(Slot* s) {
  if (s->stream) hipStreamSynchronize(s->stream);
  hipHostFree(s->h_pix); hipHostFree(s->h_high); hipHostFree(s->h_low);
  hipHostFree(s->h_pv); hipHostFree(s->h_flags);
  hipFree(s->d_pix); hipFree(s->d_high); hipFree(s->d_low);
  hipFree(s->d_pv); hipFree(s->d_flags); hipFree(s->d_ws);
  if (s->done) hipEventDestroy(s->done);
  if (s->begin) hipEventDestroy(s->begin);
  if (s->stream) hipStreamDestroy(s->stream);
  *s = Slot();
}

// Pinned allocations are slow (tens of ms per slot): only slot 0 is allocated
// by create; the others when they are first touched, which in the encoder
// pipeline overlaps with the GPU and Brotli work of the batches before.
int check_slot(fpvhip_ctx* c, uint32_t slot) {
  if (!c) return fail(nullptr, "null context");
  if (slot >= c->slots.size()) return fail(c, "slot %u out of range", slot);
  if (!c->slots[slot].ready) {
    if (hipSetDevice(c->device) != hipSuccess) return fail(c, "hipSetDevice failed");
    if (alloc_slot(c, &c->slots[slot])) return -1;
  }
  return 0;
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
    HIPOK(c, hipSetDevice(device));
    HIPOK(c, hipMalloc(reinterpret_cast<void**>(&c->d_dhigh), g.npix));
    HIPOK(c, hipMalloc(reinterpret_cast<void**>(&c->d_dlow), g.npix));
    HIPOK(c, hipMalloc(reinterpret_cast<void**>(&c->d_dtmp), (size_t)g.npix * 2));
    HIPOK(c, hipMalloc(reinterpret_cast<void**>(&c->d_inv_diag), 16));
    HIPOK(c, hipMemset(c->d_inv_diag, 0, 16));
    if (g.fwd_ok) {
      HIPOK(c, hipMalloc(&c->d_ws, fpvhip::fwd_workspace_bytes(g, max_batch)));
      fpvhip::carve_fwd_workspace(g, max_batch, c->d_ws, &c->ws);
      HIPOK(c, fpvhip::init_fwd_workspace(c->ws, nullptr));
      HIPOK(c, hipStreamSynchronize(nullptr));
    }
    for (auto& e : c->ev) HIPOK(c, hipEventCreate(&e));
    c->slots.resize(slots);
    if (slots && alloc_slot(c, &c->slots[0])) return -1;
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
  for (auto& s : c->slots) free_slot(&s);
  hipFree(c->d_dhigh); hipFree(c->d_dlow); hipFree(c->d_dtmp); hipFree(c->d_ws); hipFree(c->d_inv_diag);
  for (auto& e : c->ev) if (e) hipEventDestroy(e);
  delete c;
}

size_t fpvhip_npix(const fpvhip_ctx* c) { return c ? c->g.npix : 0; }
size_t fpvhip_npreview(const fpvhip_ctx* c) { return c ? c->g.npv : 0; }
uint32_t fpvhip_max_batch(const fpvhip_ctx* c) { return c ? c->max_batch : 0; }
uint32_t fpvhip_slots(const fpvhip_ctx* c) { return c ? (uint32_t)c->slots.size() : 0; }

int fpvhip_set_delta_raw(fpvhip_ctx* c, const uint16_t* delta_raw) {
  if (!c || !delta_raw) return fail(c, "null argument");
  if (!c->g.fwd_ok) return fail(c, "decoder-only context");
  HIPOK(c, hipSetDevice(c->device));
  HIPOK(c, hipMemcpy(c->d_dtmp, delta_raw, (size_t)c->g.npix * 2, hipMemcpyHostToDevice));
  HIPOK(c, fpvhip::launch_split(c->g, c->d_dtmp, c->d_dhigh, c->d_dlow, nullptr, nullptr));
  HIPOK(c, hipStreamSynchronize(nullptr));
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
  if (!c->g.fwd_ok || c->slots.empty()) return fail(c, "needs a forward context with a slot");
  Slot& s = c->slots[0];
  const size_t np = c->g.npix;
  HIPOK(c, hipSetDevice(c->device));
  HIPOK(c, hipMemcpyAsync(s.d_pix, img, np * 2, hipMemcpyHostToDevice, s.stream));
  HIPOK(c, hipMemsetAsync(s.ws.anylow, 0, 4, s.stream));
  HIPOK(c, fpvhip::launch_split(c->g, s.d_pix, s.d_high, s.d_low, s.ws.anylow, s.stream));
  HIPOK(c, hipMemcpyAsync(high, s.d_high, np, hipMemcpyDeviceToHost, s.stream));
  if (low && c->g.shift != 8)
    HIPOK(c, hipMemcpyAsync(low, s.d_low, np, hipMemcpyDeviceToHost, s.stream));
  uint32_t any = 0;
  HIPOK(c, hipMemcpyAsync(&any, s.ws.anylow, 4, hipMemcpyDeviceToHost, s.stream));
  HIPOK(c, hipMemsetAsync(s.ws.anylow, 0, 4, s.stream));  // the predictor expects its accumulators clean
  HIPOK(c, hipStreamSynchronize(s.stream));
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
  HIPOK(c, hipEventRecord(s.begin, s.stream));
  if (from_planes) {  // planes staged in h_high / h_low -> pixels on the device
    HIPOK(c, hipMemcpyAsync(s.d_high, s.h_high, nframes * np, hipMemcpyHostToDevice, s.stream));
    HIPOK(c, hipMemcpyAsync(s.d_low, s.h_low, nframes * np, hipMemcpyHostToDevice, s.stream));
    HIPOK(c, fpvhip::launch_interleave(s.d_high, s.d_low, s.d_pix, nframes * np, s.stream));
  } else {
    HIPOK(c, hipMemcpyAsync(s.d_pix, s.h_pix, nframes * np * 2, hipMemcpyHostToDevice, s.stream));
  }
  HIPOK(c, fpvhip::launch_predict(c->g, s.d_pix, nframes,
                                  with_delta ? c->d_dhigh : nullptr,
                                  with_delta ? c->d_dlow : nullptr, s.ws, s.d_high,
                                  s.d_low, s.d_pv, s.d_flags, s.stream, nullptr));
  HIPOK(c, hipMemcpyAsync(s.h_high, s.d_high, nframes * np, hipMemcpyDeviceToHost, s.stream));
  if (c->g.shift != 8)
    HIPOK(c, hipMemcpyAsync(s.h_low, s.d_low, nframes * np, hipMemcpyDeviceToHost, s.stream));
  HIPOK(c, hipMemcpyAsync(s.h_pv, s.d_pv, nframes * nv, hipMemcpyDeviceToHost, s.stream));
  HIPOK(c, hipMemcpyAsync(s.h_flags, s.d_flags, nframes, hipMemcpyDeviceToHost, s.stream));
  HIPOK(c, hipEventRecord(s.done, s.stream));
  s.pending = nframes;
  return 0;
}

int fpvhip_predict_wait(fpvhip_ctx* c, uint32_t slot, const uint8_t** high,
                        const uint8_t** low, const uint8_t** preview,
                        const uint8_t** flags) {
  if (check_slot(c, slot)) return -1;
  Slot& s = c->slots[slot];
  if (!s.pending) return fail(c, "slot %u has nothing in flight", slot);
  HIPOK(c, hipEventSynchronize(s.done));
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
  HIPOK(c, hipEventRecord(s.begin, s.stream));
  HIPOK(c, hipMemcpyAsync(s.d_high, s.h_high, nframes * np, hipMemcpyHostToDevice, s.stream));
  if (any_low)
    HIPOK(c, hipMemcpyAsync(s.d_low, s.h_low, nframes * np, hipMemcpyHostToDevice, s.stream));
  HIPOK(c, hipMemcpyAsync(s.d_flags, s.h_flags, nframes, hipMemcpyHostToDevice, s.stream));
  HIPOK(c, fpvhip::launch_inverse(c->g, s.d_high, s.d_low,
                                  c->have_delta ? c->d_dhigh : nullptr,
                                  c->have_delta ? c->d_dlow : nullptr, s.d_flags,
                                  nframes, s.d_pix, s.stream, c->d_inv_diag));
  if (to_planes) {  // (pixel >> 8, pixel & 0xff) back into the plane staging buffers
    HIPOK(c, fpvhip::launch_split_planes(s.d_pix, s.d_high, s.d_low, nframes * np, s.stream));
    HIPOK(c, hipMemcpyAsync(s.h_high, s.d_high, nframes * np, hipMemcpyDeviceToHost, s.stream));
    HIPOK(c, hipMemcpyAsync(s.h_low, s.d_low, nframes * np, hipMemcpyDeviceToHost, s.stream));
  } else {
    if (unshift >= 0)  // UnextractFrame (cc:850-862) fused in front of the copy back
      HIPOK(c, fpvhip::launch_unextract(s.d_pix, nframes * np, unshift, big_endian, s.stream));
    HIPOK(c, hipMemcpyAsync(s.h_pix, s.d_pix, nframes * np * 2, hipMemcpyDeviceToHost, s.stream));
  }
  HIPOK(c, hipEventRecord(s.done, s.stream));
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
               static_cast<uint16_t*>(d_frames), static_cast<hipStream_t>(stream), c->d_inv_diag));
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
  int n = fpvhip_device_count();
  if (n <= 0) return -1;
  if (device < 0 || device >= n) return fail(nullptr, "device %d out of range", device);
  HIPOK(nullptr, hipSetDevice(device));
  HIPOK(nullptr, hipFree(nullptr));  // creates the primary context
  void* p = nullptr;
  HIPOK(nullptr, hipMalloc(&p, 4096));
  HIPOK(nullptr, fpvhip::launch_copy(p, p, 0, nullptr));  // loads the code object
  HIPOK(nullptr, hipDeviceSynchronize());
  HIPOK(nullptr, hipFree(p));
  return 0;
}

int fpvhip_debug_counters(fpvhip_ctx* c, uint32_t* out, uint32_t n) {
  if (!c || !out || n > 64 || !c->g.fwd_ok) return fail(c, "bad arguments");
  HIPOK(c, hipDeviceSynchronize());
  HIPOK(c, hipMemcpy(out, c->ws.ctr + ((c->ws.seq + 1) & 1u) * 64, n * 4, hipMemcpyDeviceToHost));  // the last batch's block
  return 0;
}

double fpvhip_slot_elapsed_ms(fpvhip_ctx* c, uint32_t slot) {
  if (!c || slot >= c->slots.size() || !c->slots[slot].ready || c->slots[slot].pending) return -1.0;
  float ms = -1.f;
  if (hipEventElapsedTime(&ms, c->slots[slot].begin, c->slots[slot].done) != hipSuccess) return -1.0;
  return ms;
}

int fpvhip_inverse_pass_stats(fpvhip_ctx* c, uint32_t* out, int reset) {
  if (!c || !out) return fail(c, "bad arguments");
  HIPOK(c, hipSetDevice(c->device));
  HIPOK(c, hipDeviceSynchronize());
  HIPOK(c, hipMemcpy(out, c->d_inv_diag, 16, hipMemcpyDeviceToHost));
  if (reset) HIPOK(c, hipMemset(c->d_inv_diag, 0, 16));
  return 0;
}

double fpvhip_time_copy_device(void* d_dst, const void* d_src, size_t bytes, void* stream,
                               int iters) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t a, b;
  if (iters <= 0 || hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1.0;
  fpvhip::launch_copy(d_src, d_dst, bytes, st);  // warm
  hipEventRecord(a, st);
  for (int i = 0; i < iters; i++) fpvhip::launch_copy(d_src, d_dst, bytes, st);
  hipEventRecord(b, st);
  float ms = -1.f;
  if (hipEventSynchronize(b) == hipSuccess) hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a);
  hipEventDestroy(b);
  return ms;
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
