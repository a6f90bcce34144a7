/*
 * fpvhip_oracle_shim.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A stand-in for libfpvhip.so that implements the host-staged part of the C ABI
 * (include/fpv_hip.h) with the CPU oracle, so that the HOST logic of the codec
 * -- container format, batching, worker pool, callback ordering, decoders'
 * parsing -- can be exercised by `pytest -m "not gpu"` in a container without
 * a GPU.  It is linked ONLY into tests/shim/libfpv_amd_cpushim.so; the product
 * libraries never see it and have no CPU path.  The device-resident entry
 * points fail here.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "fpv_hip.h"
#include "fpv_oracle.h"

/* Test knobs (environment, read per call):
 *   FPV_SHIM_DEVICES=N        pose as N devices
 *   FPV_SHIM_DELAY=<dev>:<ms> fpvhip_predict_wait on contexts of device <dev> sleeps <ms> first
 *                             (a slow GPU); fpvhip_shim_overlapped_waits() counts the waits of
 *                             OTHER devices that returned while such a sleep was in progress
 *   FPV_SHIM_FAIL_WAIT=<k>    the k-th fpvhip_predict_wait made while the variable is set (1-based) fails
 *   FPV_SHIM_FAIL_SUBMIT=<k>  the k-th fpvhip_predict_submit / fpvhip_predict_planes_submit made while it is set fails
 *   FPV_SHIM_FAIL_INVERSE=<k> the k-th inverse submit (pixels, raw or planes) made while it is set fails
 * Bookkeeping the tests read back:
 *   fpvhip_shim_warmups(dev)      fpvhip_warmup calls seen for the device
 *   fpvhip_shim_cold_creates()    contexts created on a device that had not been warmed yet */
static int g_delayed_active, g_overlapped, g_waits;
static int g_warmups[64], g_cold_creates, g_submits, g_inverses;

int fpvhip_shim_overlapped_waits(void) { return __atomic_load_n(&g_overlapped, __ATOMIC_SEQ_CST); }
int fpvhip_shim_warmups(int device) { return __atomic_load_n(&g_warmups[device & 63], __ATOMIC_SEQ_CST); }
int fpvhip_shim_cold_creates(void) { return __atomic_load_n(&g_cold_creates, __ATOMIC_SEQ_CST); }
static void note_create(int device) {
  if (!__atomic_load_n(&g_warmups[device & 63], __ATOMIC_SEQ_CST)) __atomic_add_fetch(&g_cold_creates, 1, __ATOMIC_SEQ_CST);
}

typedef struct {
  uint16_t *pix;
  uint8_t *high, *low, *pv, *flags;
  uint32_t pending;
} slot_t;

struct fpvhip_ctx {
  uint32_t xs, ys, batch, nslots;
  size_t npix, npv;
  int shift, be, fwd_ok, have_delta, device;
  uint8_t *dhigh, *dlow;
  uint16_t *d16;
  slot_t *slots;
  char err[128];
};

static char g_err[128];
static int injected(const char *var, int *counter, fpvhip_ctx *c) {  /* counts calls made while `var` is set */
  const char *e = getenv(var);
  if (!e) {
    __atomic_store_n(counter, 0, __ATOMIC_SEQ_CST);
    return 0;
  }
  if (__atomic_add_fetch(counter, 1, __ATOMIC_SEQ_CST) == atoi(e)) {
    snprintf(c->err, sizeof c->err, "injected failure (%s)", var);
    return 1;
  }
  return 0;
}

int fpvhip_abi_version(void) { return FPVHIP_ABI_VERSION; }
int fpvhip_device_count(void) {
  const char *e = getenv("FPV_SHIM_DEVICES");
  return e ? atoi(e) : 1;
}
const char *fpvhip_last_error(const fpvhip_ctx *c) { return c ? c->err : g_err; }
int fpvhip_warmup(int device) {
  __atomic_add_fetch(&g_warmups[device & 63], 1, __ATOMIC_SEQ_CST);
  return 0;
}
int fpvhip_prepare_slot(fpvhip_ctx *c, uint32_t slot) { (void)c; (void)slot; return 0; }
size_t fpvhip_staging_bytes(const fpvhip_ctx *c) { return c ? (size_t)c->nslots * c->batch * c->npix * 4 : 0; }
int fpvhip_trim(fpvhip_ctx *c) { (void)c; return 0; }
const char *fpvhip_inverse_kernel_name(fpvhip_ctx *c, uint32_t n) { (void)c; (void)n; return "oracle"; }
const char *fpvhip_forward_kernel_names(fpvhip_ctx *c) { (void)c; return "oracle,oracle,oracle"; }
int fpvhip_debug_set_chain_mask(fpvhip_ctx *c, uint32_t v) { (void)c; (void)v; return 0; }
int fpvhip_forward_tile_shape(fpvhip_ctx *c, uint32_t out[4]) { (void)c; out[0] = out[1] = out[2] = out[3] = 0; return 0; }
int fpvhip_debug_counters(fpvhip_ctx *c, uint32_t *out, uint32_t n) { (void)c; (void)out; (void)n; return -1; }
double fpvhip_slot_elapsed_ms(fpvhip_ctx *c, uint32_t slot) { (void)c; (void)slot; return 0.0; }
double fpvhip_time_copy_device(void *d, const void *s, size_t b, void *st, int it) { (void)d; (void)s; (void)b; (void)st; (void)it; return -1.0; }
double fpvhip_time_copy_shape(void *d, const void *s, size_t b, void *st, int it, int sh) { (void)d; (void)s; (void)b; (void)st; (void)it; (void)sh; return -1.0; }

static fpvhip_ctx *mk(uint32_t xs, uint32_t ys, int shift, int be, uint32_t batch,
                      uint32_t slots, int fwd) {
  fpvhip_ctx *c;
  uint32_t s;
  if (!xs || !ys || !batch || (fwd && ((xs & 3) || (ys & 3)))) {
    snprintf(g_err, sizeof g_err, "unsupported geometry %ux%u", xs, ys);
    return NULL;
  }
  c = (fpvhip_ctx *)calloc(1, sizeof *c);
  c->xs = xs; c->ys = ys; c->batch = batch; c->nslots = slots;
  c->npix = (size_t)xs * ys; c->npv = (size_t)(xs / 4) * (ys / 4);
  c->shift = shift; c->be = be; c->fwd_ok = fwd;
  c->dhigh = (uint8_t *)calloc(c->npix, 1);
  c->dlow = (uint8_t *)calloc(c->npix, 1);
  c->d16 = (uint16_t *)calloc(c->npix, 2);
  c->slots = (slot_t *)calloc(slots ? slots : 1, sizeof(slot_t));
  for (s = 0; s < slots; s++) {
    c->slots[s].pix = (uint16_t *)calloc(c->npix * batch, 2);
    c->slots[s].high = (uint8_t *)calloc(c->npix * batch, 1);
    c->slots[s].low = (uint8_t *)calloc(c->npix * batch, 1);
    c->slots[s].pv = (uint8_t *)calloc((c->npix / 16 + c->npv + 16) * batch, 1);
    c->slots[s].flags = (uint8_t *)calloc(batch, 1);
  }
  return c;
}

fpvhip_ctx *fpvhip_create(int device, uint32_t xs, uint32_t ys, int shift, int be,
                          uint32_t batch, uint32_t slots) {
  fpvhip_ctx *c = mk(xs, ys, shift, be, batch, slots, 1);
  if (c) c->device = device;
  if (!(xs == 512 && ys == 256 && batch == 1)) note_create(device);  /* (the warm-up's own small context does not count) */
  return c;
}
fpvhip_ctx *fpvhip_create_decoder(int device, uint32_t xs, uint32_t ys, uint32_t batch,
                                  uint32_t slots) {
  if (!(xs == 512 && ys == 256 && batch == 1)) note_create(device);
  return mk(xs, ys, 0, 0, batch, slots, 0);
}
void fpvhip_destroy(fpvhip_ctx *c) {
  uint32_t s;
  if (!c) return;
  for (s = 0; s < c->nslots; s++) {
    free(c->slots[s].pix); free(c->slots[s].high); free(c->slots[s].low);
    free(c->slots[s].pv); free(c->slots[s].flags);
  }
  free(c->slots); free(c->dhigh); free(c->dlow); free(c->d16); free(c);
}
size_t fpvhip_npix(const fpvhip_ctx *c) { return c->npix; }
size_t fpvhip_npreview(const fpvhip_ctx *c) { return c->npv; }
uint32_t fpvhip_max_batch(const fpvhip_ctx *c) { return c->batch; }
uint32_t fpvhip_slots(const fpvhip_ctx *c) { return c->nslots; }

int fpvhip_set_delta_raw(fpvhip_ctx *c, const uint16_t *d) {
  fpvo_split(d, c->npix, c->shift, c->be, c->dhigh, c->dlow);
  if (c->shift == 8) memset(c->dlow, 0, c->npix);
  c->have_delta = 1;
  return 0;
}
int fpvhip_set_delta_decoded(fpvhip_ctx *c, const uint16_t *d) {
  memcpy(c->d16, d, c->npix * 2);
  c->have_delta = 1;
  return 0;
}
int fpvhip_set_delta_planes(fpvhip_ctx *c, const uint8_t *h, const uint8_t *l) {
  size_t i;
  memcpy(c->dhigh, h, c->npix);
  if (l) memcpy(c->dlow, l, c->npix); else memset(c->dlow, 0, c->npix);
  for (i = 0; i < c->npix; i++) c->d16[i] = (uint16_t)((c->dhigh[i] << 8) | c->dlow[i]);
  c->have_delta = 1;
  return 0;
}

int fpvhip_split_frame(fpvhip_ctx *c, const uint16_t *img, uint8_t *high, uint8_t *low, int *no_low) {
  uint8_t *tmp = low ? low : (uint8_t *)malloc(c->npix);
  *no_low = fpvo_split(img, c->npix, c->shift, c->be, high, tmp);
  if (!low) free(tmp);
  return 0;
}

uint16_t *fpvhip_stage_in(fpvhip_ctx *c, uint32_t slot, uint32_t k) {
  return c->slots[slot].pix + (size_t)k * c->npix;
}
int fpvhip_predict_submit(fpvhip_ctx *c, uint32_t slot, uint32_t n, int with_delta) {
  slot_t *s = &c->slots[slot];
  uint32_t k;
  uint8_t *pv;
  if (injected("FPV_SHIM_FAIL_SUBMIT", &g_submits, c)) return -1;
  pv = (uint8_t *)malloc(c->npix / 16 + c->npv + 16);
  if (!c->fwd_ok || (with_delta && !c->have_delta)) {
    snprintf(c->err, sizeof c->err, "bad predict_submit");
    free(pv);
    return -1;
  }
  for (k = 0; k < n; k++) {
    s->flags[k] = fpvo_predict(s->pix + (size_t)k * c->npix, c->xs, c->ys, c->shift, c->be,
                               with_delta ? c->dhigh : NULL, with_delta ? c->dlow : NULL,
                               s->high + (size_t)k * c->npix, s->low + (size_t)k * c->npix, pv);
    memcpy(s->pv + (size_t)k * c->npv, pv, c->npv);
  }
  free(pv);
  s->pending = n;
  return 0;
}
int fpvhip_predict_wait(fpvhip_ctx *c, uint32_t slot, const uint8_t **high,
                        const uint8_t **low, const uint8_t **pv, const uint8_t **flags) {
  slot_t *s = &c->slots[slot];
  const char *e;
  int delayed = 0;
  if (injected("FPV_SHIM_FAIL_WAIT", &g_waits, c)) return -1;
  if ((e = getenv("FPV_SHIM_DELAY")) && atoi(e) == c->device && strchr(e, ':')) {
    struct timespec ts;
    const long ms = atol(strchr(e, ':') + 1);
    ts.tv_sec = ms / 1000; ts.tv_nsec = (ms % 1000) * 1000000L;
    delayed = 1;
    __atomic_add_fetch(&g_delayed_active, 1, __ATOMIC_SEQ_CST);
    nanosleep(&ts, NULL);
    __atomic_sub_fetch(&g_delayed_active, 1, __ATOMIC_SEQ_CST);
  }
  if (!delayed && __atomic_load_n(&g_delayed_active, __ATOMIC_SEQ_CST) > 0) __atomic_add_fetch(&g_overlapped, 1, __ATOMIC_SEQ_CST);
  if (!s->pending) { snprintf(c->err, sizeof c->err, "nothing in flight"); return -1; }
  s->pending = 0;
  if (high) *high = s->high;
  if (low) *low = s->low;
  if (pv) *pv = s->pv;
  if (flags) *flags = s->flags;
  return 0;
}

int fpvhip_predict_planes_submit(fpvhip_ctx *c, uint32_t slot, uint32_t n, int with_delta) {
  slot_t *s = &c->slots[slot];
  size_t i;
  if (c->shift != 0 || c->be) { snprintf(c->err, sizeof c->err, "planes input needs shift 0"); return -1; }
  for (i = 0; i < (size_t)n * c->npix; i++) s->pix[i] = (uint16_t)((s->high[i] << 8) | s->low[i]);
  return fpvhip_predict_submit(c, slot, n, with_delta);
}

uint8_t *fpvhip_stage_plane(fpvhip_ctx *c, uint32_t slot, uint32_t k, int which) {
  slot_t *s = &c->slots[slot];
  return (which ? s->low : s->high) + (size_t)k * c->npix;
}
int fpvhip_inverse_submit(fpvhip_ctx *c, uint32_t slot, uint32_t n, const uint8_t *flags) {
  slot_t *s = &c->slots[slot];
  uint32_t k;
  if (injected("FPV_SHIM_FAIL_INVERSE", &g_inverses, c)) return -1;
  for (k = 0; k < n; k++) {
    if ((flags[k] & 1) && !c->have_delta) {
      snprintf(c->err, sizeof c->err, "delta frame not given");
      return -1;
    }
    fpvo_inverse(s->high + (size_t)k * c->npix,
                 (flags[k] & 4) ? NULL : s->low + (size_t)k * c->npix,
                 c->have_delta ? c->d16 : NULL, c->xs, c->ys, flags[k],
                 s->pix + (size_t)k * c->npix);
  }
  s->pending = n;
  return 0;
}
int fpvhip_inverse_submit_raw(fpvhip_ctx *c, uint32_t slot, uint32_t n, const uint8_t *flags,
                              int unshift, int big_endian) {
  slot_t *s = &c->slots[slot];
  uint32_t k;
  if (fpvhip_inverse_submit(c, slot, n, flags)) return -1;
  for (k = 0; k < n; k++) {  /* in place: fpvo_unextract reads pixel i before it writes bytes 2i, 2i+1 */
    uint16_t *px = s->pix + (size_t)k * c->npix;
    fpvo_unextract(px, c->npix, unshift, big_endian, (uint8_t *)px);
  }
  return 0;
}
int fpvhip_inverse_planes_submit(fpvhip_ctx *c, uint32_t slot, uint32_t n, const uint8_t *flags) {
  slot_t *s = &c->slots[slot];
  uint32_t k;
  size_t i;
  if (injected("FPV_SHIM_FAIL_INVERSE", &g_inverses, c)) return -1;
  for (k = 0; k < n; k++) {
    uint8_t *h = s->high + (size_t)k * c->npix, *l = s->low + (size_t)k * c->npix;
    if ((flags[k] & 1) && !c->have_delta) { snprintf(c->err, sizeof c->err, "delta frame not given"); return -1; }
    if (flags[k] & 2) fpvo_cg_inverse(h, c->xs, c->npix);
    if (flags[k] & 1)
      for (i = 0; i < c->npix; i++) {
        h[i] = (uint8_t)(h[i] + c->dhigh[i]);
        l[i] = (uint8_t)(((flags[k] & 4) ? 0 : l[i]) + c->dlow[i]);
      }
    else if (flags[k] & 4) memset(l, 0, c->npix);
  }
  s->pending = n;
  return 0;
}

int fpvhip_inverse_wait(fpvhip_ctx *c, uint32_t slot, const uint16_t **frames) {
  slot_t *s = &c->slots[slot];
  if (!s->pending) { snprintf(c->err, sizeof c->err, "nothing in flight"); return -1; }
  s->pending = 0;
  if (frames) *frames = s->pix;
  return 0;
}

#define NOPE(c) (snprintf((c)->err, sizeof (c)->err, "device entry points need the GPU library"), -1)
int fpvhip_predict_device(fpvhip_ctx *c, const void *a, uint32_t n, int wd, void *h, void *l,
                          void *p, void *f, void *s) {
  (void)a; (void)n; (void)wd; (void)h; (void)l; (void)p; (void)f; (void)s;
  return NOPE(c);
}
int fpvhip_inverse_device(fpvhip_ctx *c, const void *h, const void *l, const void *f, uint32_t n,
                          void *o, void *s) {
  (void)h; (void)l; (void)f; (void)n; (void)o; (void)s;
  return NOPE(c);
}
double fpvhip_time_predict_device(fpvhip_ctx *c, const void *a, uint32_t n, int wd, void *h,
                                  void *l, void *p, void *f, void *s, int it, double *km) {
  (void)a; (void)n; (void)wd; (void)h; (void)l; (void)p; (void)f; (void)s; (void)it; (void)km;
  return NOPE(c);
}
double fpvhip_time_inverse_device(fpvhip_ctx *c, const void *h, const void *l, const void *f,
                                  uint32_t n, void *o, void *s, int it) {
  (void)h; (void)l; (void)f; (void)n; (void)o; (void)s; (void)it;
  return NOPE(c);
}
