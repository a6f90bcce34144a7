/*
 * fpv_oracle.c -- TEST INFRASTRUCTURE ONLY (see fpv_oracle.h).
 *
 * Scalar CPU restatement of the fusion-power-video predictor/residual stage.
 * Written for clarity, not speed: one pixel at a time, no SIMD, no threads.
 * "cc:N" = fusion_power_video.cc line N of the upstream reference.
 */
#include "fpv_oracle.h"

#include <brotli/decode.h>
#include <brotli/encode.h>
#include <stdlib.h>
#include <string.h>

/* ---- pixel kernels ------------------------------------------------------ */

/* cc:247-252.  Equal to median(n, w, n + w - nw) over the integers. */
uint8_t fpvo_clamped_gradient(uint8_t n, uint8_t w, uint8_t nw) {
  uint8_t lo = n < w ? n : w;
  uint8_t hi = n < w ? w : n;
  if (nw > hi) return lo;
  if (nw < lo) return hi;
  return (uint8_t)(n + w - nw);
}

/* floor(log2(v)) for v > 0; cc:216-232 */
static unsigned floor_log2_u64(uint64_t v) {
  unsigned r = 0;
  while (v >>= 1) r++;
  return r;
}

/*
 * cc:235-244.  sum = total count; L = floor(log2(sum));
 * S = sum over non-empty bins of v * (L - floor(log2 v)); result 1024*S/sum,
 * truncating.  Empty bins contribute nothing (the reference multiplies its
 * undefined clz(0) by v == 0).
 */
uint64_t fpvo_estimate_entropy(const uint64_t hist[256]) {
  uint64_t sum = 0, s = 0;
  unsigned lsum;
  int b;
  for (b = 0; b < 256; b++) sum += hist[b];
  if (sum == 0) return 0;
  lsum = floor_log2_u64(sum);
  for (b = 0; b < 256; b++) {
    if (hist[b]) s += hist[b] * (uint64_t)(lsum - floor_log2_u64(hist[b]));
  }
  return 1024u * s / sum;
}

/* cc:370-451.  Host is little endian (asserted by the tests). */
int fpvo_split(const uint16_t *img, size_t npix, int shift, int big_endian,
               uint8_t *high, uint8_t *low) {
  const int swap = big_endian ? 1 : 0; /* host little endian: cc:379 */
  uint8_t any_low = 0;
  size_t i;
  for (i = 0; i < npix; i++) {
    uint16_t p = img[i];
    uint8_t h, l = 0;
    if (swap) {
      if (shift == 0) { /* cc:391-397 */
        h = (uint8_t)(p & 0xff);
        l = (uint8_t)(p >> 8);
      } else if (shift == 8) { /* cc:401-403 */
        h = (uint8_t)(p >> 8);
      } else { /* cc:407-415 */
        h = (uint8_t)(((unsigned)p << shift) | ((unsigned)p >> (16 - shift)));
        l = (uint8_t)((unsigned)p >> (8 - shift));
      }
    } else {
      if (shift == 0) { /* cc:421-427 */
        h = (uint8_t)(p >> 8);
        l = (uint8_t)(p & 0xff);
      } else if (shift == 8) { /* cc:431-433 */
        h = (uint8_t)(p & 0xff);
      } else { /* cc:437-443 */
        uint16_t q = (uint16_t)((unsigned)p << shift);
        h = (uint8_t)(q >> 8);
        l = (uint8_t)(q & 0xff);
      }
    }
    high[i] = h;
    if (shift != 8) {
      low[i] = l;
      any_low |= l;
    }
  }
  return any_low == 0; /* cc:447-449 */
}

/* cc:491-515 */
void fpvo_preview(const uint8_t *high, size_t xs, size_t ys, uint8_t *out) {
  size_t pw = xs / 4, ph = ys / 4, px, py, i, j;
  for (py = 0; py < ph; py++) {
    for (px = 0; px < pw; px++) {
      unsigned sum = 0;
      for (j = 0; j < 4; j++)
        for (i = 0; i < 4; i++) sum += high[(4 * py + j) * xs + 4 * px + i];
      out[py * pw + px] = (uint8_t)((sum / 16) & 0xfe);
    }
  }
}

/*
 * cc:517-533.  The reference histograms a = high[i] and d = a - high[i], i.e.
 * d is always 0 (cc:528), so E(countd) == 0 and the delta predictor is chosen
 * exactly when E(counta) > 0.
 */
int fpvo_delta_decision(const uint8_t *high, size_t npix) {
  uint64_t ca[256], cd[256];
  size_t i;
  memset(ca, 0, sizeof ca);
  memset(cd, 0, sizeof cd);
  for (i = 0; i < npix; i += 15) {
    ca[high[i]]++;
    cd[0]++;
  }
  return fpvo_estimate_entropy(cd) < fpvo_estimate_entropy(ca);
}

/* cc:546-564 */
int fpvo_cg_decision(const uint8_t *high, size_t xs, size_t npix) {
  uint64_t ca[256], cb[256];
  size_t i;
  memset(ca, 0, sizeof ca);
  memset(cb, 0, sizeof cb);
  for (i = xs + 1; i < npix; i += 31) {
    uint8_t a = high[i];
    uint8_t b = (uint8_t)(a - fpvo_clamped_gradient(high[i - xs], high[i - 1],
                                                    high[i - xs - 1]));
    ca[a]++;
    cb[b]++;
  }
  return fpvo_estimate_entropy(cb) < fpvo_estimate_entropy(ca);
}

/* cc:565-573 (high) and cc:575-586 (preview, xs = preview width) */
void fpvo_cg_forward(const uint8_t *in, size_t xs, size_t n, uint8_t *out) {
  size_t i, lit = xs + 1 < n ? xs + 1 : n;
  for (i = 0; i < lit; i++) out[i] = in[i];
  for (i = xs + 1; i < n; i++)
    out[i] = (uint8_t)(in[i] - fpvo_clamped_gradient(in[i - xs], in[i - 1],
                                                     in[i - xs - 1]));
}

/* cc:326-333, cc:617-622 */
void fpvo_cg_inverse(uint8_t *p, size_t xs, size_t n) {
  size_t i;
  for (i = xs + 1; i < n; i++)
    p[i] = (uint8_t)(p[i] +
                     fpvo_clamped_gradient(p[i - xs], p[i - 1], p[i - xs - 1]));
}

/* Frame ctor + Predict: cc:370-451, cc:777-785 */
uint8_t fpvo_predict(const uint16_t *img, size_t xs, size_t ys, int shift,
                     int big_endian, const uint8_t *dhigh, const uint8_t *dlow,
                     uint8_t *high, uint8_t *low, uint8_t *preview) {
  size_t npix = xs * ys, i;
  uint8_t flags = 0;
  if (fpvo_split(img, npix, shift, big_endian, high, low))
    flags |= FPVO_NO_LOW_BYTES;
  fpvo_preview(high, xs, ys, preview); /* cc:778, on the raw plane */
  if (dhigh && fpvo_delta_decision(high, npix)) { /* cc:780-782 */
    for (i = 0; i < npix; i++) high[i] = (uint8_t)(high[i] - dhigh[i]);
    if (shift != 8 && dlow) /* low_ is empty for shift 8: cc:536 is a no-op */
      for (i = 0; i < npix; i++) low[i] = (uint8_t)(low[i] - dlow[i]);
    flags |= FPVO_USE_DELTA;
  }
  if (fpvo_cg_decision(high, xs, npix)) { /* cc:784 */
    uint8_t *tmp = (uint8_t *)malloc(npix ? npix : 1);
    size_t np = npix / 16; /* cc:577: size_/16, not preview_.size() */
    fpvo_cg_forward(high, xs, npix, tmp);
    memcpy(high, tmp, npix);
    fpvo_cg_forward(preview, xs / 4, np, tmp);
    memcpy(preview, tmp, np);
    free(tmp);
    flags |= FPVO_USE_CG;
  }
  return flags;
}

/* cc:326-344 */
void fpvo_inverse(uint8_t *high, const uint8_t *low, const uint16_t *delta,
                  size_t xs, size_t ys, uint8_t flags, uint16_t *img) {
  size_t npix = xs * ys, i;
  if (flags & FPVO_USE_CG) fpvo_cg_inverse(high, xs, npix);
  if (flags & FPVO_USE_DELTA) {
    for (i = 0; i < npix; i++) {
      unsigned l = low ? low[i] : 0;
      img[i] = (uint16_t)(((high[i] + (delta[i] >> 8)) << 8) |
                          ((l + (delta[i] & 0xff)) & 0xff));
    }
  } else {
    for (i = 0; i < npix; i++)
      img[i] = (uint16_t)((high[i] << 8) | (low ? low[i] : 0));
  }
}

/* cc:850-862 */
void fpvo_unextract(const uint16_t *img, size_t npix, int shift, int big_endian,
                    uint8_t *out) {
  size_t i;
  for (i = 0; i < npix; i++) {
    uint16_t u = (uint16_t)(img[i] >> shift);
    uint8_t a = (uint8_t)(u & 255), b = (uint8_t)(u >> 8);
    out[2 * i + 0] = big_endian ? b : a;
    out[2 * i + 1] = big_endian ? a : b;
  }
}

/* ---- Brotli + container -------------------------------------------------- */

size_t fpvo_max_compressed_size(size_t n) {
  return BrotliEncoderMaxCompressedSize(n);
}

/* cc:653: quality 1, BROTLI_DEFAULT_WINDOW (22), BROTLI_DEFAULT_MODE */
static size_t brotli_plane(const uint8_t *in, size_t n, uint8_t *out,
                           size_t cap) {
  size_t sz = cap;
  if (!BrotliEncoderCompress(1, BROTLI_DEFAULT_WINDOW, BROTLI_DEFAULT_MODE, n,
                             in, &sz, out))
    return 0;
  return sz;
}

static void put32(uint8_t *p, uint32_t v) {
  p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8);
  p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}
static void put64(uint8_t *p, uint64_t v) {
  put32(p, (uint32_t)v);
  put32(p + 4, (uint32_t)(v >> 32));
}
static uint32_t get32(const uint8_t *p) {
  return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 |
         (uint32_t)p[3] << 24;
}
static uint64_t get64(const uint8_t *p) {
  return (uint64_t)get32(p) | (uint64_t)get32(p + 4) << 32;
}

/*
 * Image core (OutputCore cc:820-828): flags | [brotli low] | brotli high.
 * Returns bytes written or 0.
 */
static size_t write_core(uint8_t flags, const uint8_t *high, const uint8_t *low,
                         size_t npix, uint8_t *out, size_t cap) {
  size_t pos = 0, n;
  if (cap < 1) return 0;
  out[pos++] = flags;
  if (!(flags & FPVO_NO_LOW_BYTES)) { /* cc:658-676 */
    n = brotli_plane(low, npix, out + pos, cap - pos);
    if (!n) return 0;
    pos += n;
  }
  n = brotli_plane(high, npix, out + pos, cap - pos);
  if (!n) return 0;
  return pos + n;
}

/* Frame::Compress + OutputFull: cc:738-745, cc:830-846 */
size_t fpvo_encode_frame_chunk(const uint16_t *img, size_t xs, size_t ys,
                               int shift, int big_endian, const uint8_t *dhigh,
                               const uint8_t *dlow, uint8_t *out, size_t cap) {
  size_t npix = xs * ys, np = (xs / 4) * (ys / 4), pos, n;
  uint8_t *high = (uint8_t *)malloc(npix), *low = (uint8_t *)malloc(npix);
  uint8_t *pv = (uint8_t *)malloc(npix / 16 + np + 1);
  uint8_t flags;
  size_t total = 0;
  if (cap < 16) goto done;
  flags = fpvo_predict(img, xs, ys, shift, big_endian, dhigh, dlow, high, low, pv);
  pos = 9;
  out[pos++] = (uint8_t)((flags & FPVO_USE_CG) | FPVO_NO_LOW_BYTES); /* cc:842 */
  n = brotli_plane(pv, np, out + pos, cap - pos);
  if (!n) goto done;
  pos += n;
  put32(out + 5, (uint32_t)(n + 1)); /* cc:841 */
  n = write_core(flags, high, low, npix, out + pos, cap - pos);
  if (!n) goto done;
  pos += n;
  put32(out, (uint32_t)pos); /* cc:838 */
  out[4] = 0;                /* cc:840 */
  total = pos;
done:
  free(high); free(low); free(pv);
  return total;
}

/* Encoder::Init / CompressFrame / Finish: cc:1086-1197 */
size_t fpvo_encode_stream(const uint16_t *frames, size_t nframes,
                          const uint16_t *delta_raw, size_t xs, size_t ys,
                          int shift, int big_endian, uint8_t *out, size_t cap) {
  size_t npix = xs * ys, np = (xs / 4) * (ys / 4), pos = 0, n, f;
  uint8_t *dh = (uint8_t *)malloc(npix), *dl = (uint8_t *)malloc(npix);
  uint8_t *h = (uint8_t *)malloc(npix), *l = (uint8_t *)malloc(npix);
  uint8_t *pv = (uint8_t *)malloc(npix / 16 + np + 1);
  uint64_t *offs = (uint64_t *)malloc(sizeof(uint64_t) * (nframes + 1));
  uint8_t flags;
  size_t total = 0;

  if (cap < 13) goto done;
  put32(out, (uint32_t)xs); /* cc:1092-1095 */
  put32(out + 4, (uint32_t)ys);
  out[12] = 1;
  pos = 13;
  /* delta_frame_ keeps the RAW split planes (cc:1097); a copy is predicted
   * without delta frame and compressed (cc:1099-1101). */
  fpvo_split(delta_raw, npix, shift, big_endian, dh, dl);
  flags = fpvo_predict(delta_raw, xs, ys, shift, big_endian, NULL, NULL, h, l, pv);
  n = write_core(flags, h, l, npix, out + pos, cap - pos);
  if (!n) goto done;
  pos += n;
  put32(out + 8, (uint32_t)(pos - 8)); /* cc:1103 */

  for (f = 0; f < nframes; f++) {
    offs[f] = pos; /* cc:1180 */
    n = fpvo_encode_frame_chunk(frames + f * npix, xs, ys, shift, big_endian, dh,
                                shift == 8 ? NULL : dl, out + pos, cap - pos);
    if (!n) goto done;
    pos += n;
  }
  /* footer, cc:1185-1197 */
  n = 5 + 8 * nframes + 8;
  if (cap - pos < n) goto done;
  put32(out + pos, (uint32_t)n);
  out[pos + 4] = 2;
  for (f = 0; f < nframes; f++) put64(out + pos + 5 + 8 * f, offs[f]);
  put64(out + pos + 5 + 8 * nframes, nframes);
  total = pos + n;
done:
  free(dh); free(dl); free(h); free(l); free(pv); free(offs);
  return total;
}

/* cc:186-214: one Brotli stream starting at *pos; appends exactly what the
 * stream holds (up to cap) and advances *pos to the end of the stream. */
static int brotli_stream(const uint8_t *in, size_t size, size_t *pos,
                         uint8_t *out, size_t cap, size_t *out_len) {
  BrotliDecoderState *st = BrotliDecoderCreateInstance(NULL, NULL, NULL);
  size_t avail_in = size - *pos, avail_out = cap;
  const uint8_t *next_in = in + *pos;
  uint8_t *next_out = out;
  BrotliDecoderResult r;
  if (!st) return 0;
  r = BrotliDecoderDecompressStream(st, &avail_in, &next_in, &avail_out,
                                    &next_out, NULL);
  BrotliDecoderDestroyInstance(st);
  *pos = size - avail_in;
  *out_len = cap - avail_out;
  return r == BROTLI_DECODER_RESULT_SUCCESS;
}

/* cc:296-347 */
int fpvo_decompress_image(const uint16_t *delta, const uint8_t *in, size_t size,
                          size_t xs, size_t ys, uint16_t *img) {
  size_t npix = xs * ys, pos = 0, got = 0;
  uint8_t flags, *high, *low = NULL;
  int ok = 0;
  if (size == 0 || !xs || !ys) return 0;
  flags = in[pos++];
  if ((flags & FPVO_USE_DELTA) && !delta) return 0; /* cc:310 */
  high = (uint8_t *)malloc(npix + 1);
  if (!(flags & FPVO_NO_LOW_BYTES)) {
    low = (uint8_t *)malloc(npix + 1);
    /* capacity npix+1 so that an over-long stream is seen as a size mismatch */
    if (!brotli_stream(in, size, &pos, low, npix + 1, &got) || got != npix)
      goto done;
  }
  if (!brotli_stream(in, size, &pos, high, npix + 1, &got) || got != npix)
    goto done;
  fpvo_inverse(high, low, delta, xs, ys, flags, img);
  ok = 1;
done:
  free(high); free(low);
  return ok;
}

/* RandomAccessDecoder::Init + DecodeFrame: cc:961-1036 */
long fpvo_decode_stream(const uint8_t *data, size_t size, uint16_t *frames_out,
                        size_t cap_frames, size_t *xs_out, size_t *ys_out) {
  size_t xs, ys, npix, pos, dsz, nfr, fsz, i;
  uint16_t *delta;
  long ret = -1;
  if (size < 12) return -1;
  xs = get32(data); ys = get32(data + 4);
  if (!xs || !ys || xs > 65536 || ys > 65536) return -1;
  npix = xs * ys;
  pos = 8;
  dsz = get32(data + pos);
  if (dsz < 5 || pos > size || size - pos < dsz || data[12] != 1) return -1;
  delta = (uint16_t *)malloc(npix * 2);
  if (!fpvo_decompress_image(NULL, data + pos + 5, dsz - 5, xs, ys, delta))
    goto done;
  nfr = (size_t)get64(data + size - 8);
  if (nfr > size / 16) goto done;
  fsz = 5 + 8 * nfr + 8;
  if (fsz > size) goto done;
  pos = size - fsz;
  if (get32(data + pos) != fsz || data[pos + 4] != 2) goto done;
  if (nfr > cap_frames) goto done;
  for (i = 0; i < nfr; i++) {
    size_t off = (size_t)get64(data + pos + 5 + 8 * i), frame_size, psz;
    const uint8_t *d;
    if (off > size || size - off < 9) goto done;
    d = data + off;
    frame_size = get32(d);
    if (frame_size < 9 || size - off < frame_size || d[4] != 0) goto done;
    psz = get32(d + 5);
    if (psz > frame_size - 9) goto done;
    if (!fpvo_decompress_image(delta, d + 9 + psz, frame_size - psz - 9, xs, ys,
                               frames_out + i * npix))
      goto done;
  }
  *xs_out = xs; *ys_out = ys;
  ret = (long)nfr;
done:
  free(delta);
  return ret;
}
