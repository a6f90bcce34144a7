// fpv_kernels.hip -- gfx950 (CDNA4) kernels of the fusion-power-video
// predictor/residual stage.  Pure 8/16-bit integer filtering: no MFMA; the
// bound is HBM bandwidth (forward, and inverse with enough frames in flight) or
// one wave's issue rate (inverse of few frames).
//
// Reference loops replaced ("cc:N" = fusion_power_video.cc of the upstream
// CPU library):
//   k_stream           split + preview sums + delta + CG under speculated flags,
//                      heuristic histograms in the same pass   cc:391-443, cc:500-512,
//                                                              cc:526-537, cc:554-572
//   k_decide_spec      EstimateEntropy + the two decisions     cc:235-244, cc:533, cc:564
//   k_redo_decide / k_apply_list
//                      exact redo of mis-speculated frames     (same lines)
//   k_finish           preview CG + flags byte                 cc:575-586, cc:447-449
//   k_inverse_spec / k_inverse_trio / k_inverse(_fast)
//                      CG^-1, delta^-1, recombine              cc:326-344
//
// Neighbourhoods are FLAT 1-D (w = i-1, n = i-xs, nw = i-xs-1 on the flattened
// plane, cc:556-558): column 0's west neighbour is the previous row's last
// pixel.  The first xs+1 bytes of a predicted plane are literals (cc:572).
#include "fpv_kernels.h"

#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <type_traits>

namespace fpvhip {

namespace {

constexpr int kThreads = 256;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- helpers --

// Frame ctor arithmetic for ONE pixel (cc:388-443).  For shift == 8 the low
// plane does not exist in the reference; callers ignore l then.
template <bool SWAP>
__device__ __forceinline__ void split1(uint32_t p, int S, uint32_t* h,
                                       uint32_t* l) {
  if (SWAP) {
    *h = ((p << S) | (p >> (16 - S))) & 0xffu;  // cc:393, cc:402, cc:411
    *l = (p >> (8 - S)) & 0xffu;                // cc:394, cc:412
  } else {
    uint32_t q = (p << S) & 0xffffu;  // cc:438
    *h = q >> 8;                      // cc:423, cc:432, cc:439
    *l = q & 0xffu;
  }
}

// Same for 4 pixels held in two dwords (va = px0|px1<<16, vb = px2|px3<<16);
// returns the 4 high bytes and the 4 low bytes packed little-endian.
template <bool SWAP>
__device__ __forceinline__ void split4(uint32_t va, uint32_t vb, int S,
                                       uint32_t* h4, uint32_t* l4) {
  if (SWAP) {
    const uint32_t mA = (0xffu >> S) * 0x00010001u;
    const uint32_t mB = ((1u << S) - 1u) * 0x00010001u;
    uint32_t ha = ((va & mA) << S) | ((va >> (16 - S)) & mB);
    uint32_t hb = ((vb & mA) << S) | ((vb >> (16 - S)) & mB);
    uint32_t la = va >> (8 - S), lb = vb >> (8 - S);
    *h4 = __builtin_amdgcn_perm(hb, ha, 0x06040200u);
    *l4 = __builtin_amdgcn_perm(lb, la, 0x06040200u);
  } else {
    const uint32_t m = (0xffffu >> S) * 0x00010001u;
    uint32_t qa = (va & m) << S, qb = (vb & m) << S;
    *h4 = __builtin_amdgcn_perm(qb, qa, 0x07050301u);
    *l4 = __builtin_amdgcn_perm(qb, qa, 0x06040200u);
  }
}

// Per-byte a - b (mod 256) on 4 packed bytes (std::minus<uint8_t>, cc:534-537).
__device__ __forceinline__ uint32_t sub4(uint32_t a, uint32_t b) {
  const uint32_t H = 0x80808080u;
  return ((a | H) - (b & ~H)) ^ ((a ^ ~b) & H);
}
// Per-byte a + b (mod 256).
__device__ __forceinline__ uint32_t add4(uint32_t a, uint32_t b) {
  const uint32_t H = 0x80808080u;
  return ((a & ~H) + (b & ~H)) ^ ((a ^ b) & H);
}

// ClampedGradient (cc:247-252) == median(n, w, n + w - nw) over the integers.
__device__ __forceinline__ int cg_pred(int n, int w, int nw) {
  int g = n + w - nw;
  int lo = min(n, w), hi = max(n, w);
  return max(lo, min(hi, g));
}

__device__ __forceinline__ uint32_t byte_of(uint32_t v, int k) {
  return (v >> (8 * k)) & 0xffu;
}

// Sum over the 256 threads of a block; every thread gets the result.
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t* scratch) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();  // scratch may still be read from a previous call
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

// Streaming copy, 16 B per lane, non-temporal, one contiguous 256 KiB chunk per
// workgroup and turn: the fastest of the copy shapes tried on MI355X
// (tools/exp/bw2.hip: 5.3-5.5 TB/s read+write, grid-stride copies reach 4.3-5.0).
// It is the measured HBM ceiling the predictor kernels are compared with (bench only).
constexpr size_t kCopyChunk16 = (256u << 10) / 16;
__global__ __launch_bounds__(kThreads) void k_copy(const u32x4* __restrict__ src,
                                                  u32x4* __restrict__ dst, size_t n16) {
  for (size_t c = blockIdx.x; c * kCopyChunk16 < n16; c += gridDim.x) {
    const size_t b = c * kCopyChunk16, e = min(b + kCopyChunk16, n16);
    size_t i = b + threadIdx.x;
    for (; i + 3 * kThreads < e; i += 4 * kThreads) {
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = __builtin_nontemporal_load(src + i + u * kThreads);
#pragma unroll
      for (int u = 0; u < 4; u++) __builtin_nontemporal_store(v[u], dst + i + u * kThreads);
    }
    for (; i < e; i += kThreads) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
  }
}

// ------------------------------------------------------------- raw split ---

template <bool SWAP>
__global__ __launch_bounds__(kThreads) void k_split(
    const uint16_t* __restrict__ img, uint8_t* __restrict__ high,
    uint8_t* __restrict__ low, uint32_t npix, int S, uint32_t* __restrict__ anylow) {
  uint32_t i = (blockIdx.x * kThreads + threadIdx.x) * 4;
  uint32_t l4 = 0;
  if (i < npix) {  // npix is a multiple of 16
    const uint2 v = *reinterpret_cast<const uint2*>(img + i);
    uint32_t h4;
    split4<SWAP>(v.x, v.y, S, &h4, &l4);
    if (S == 8) l4 = 0;
    *reinterpret_cast<uint32_t*>(high + i) = h4;
    *reinterpret_cast<uint32_t*>(low + i) = l4;
  }
  if (anylow) {  // NO_LOW_BYTES of the Frame ctor (cc:447-449)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) l4 |= __shfl_xor(l4, off, 64);
    if ((threadIdx.x & 63) == 0 && l4) atomicOr(anylow, l4);
  }
}

// Byte planes -> uint16 pixels (high << 8 | low): lets frames that exist only
// as planes (Frame API) go through the same forward kernels with shift 0.
__global__ __launch_bounds__(kThreads) void k_interleave(
    const uint8_t* __restrict__ high, const uint8_t* __restrict__ low,
    uint16_t* __restrict__ img, size_t npix_total) {
  const size_t i = ((size_t)blockIdx.x * kThreads + threadIdx.x) * 4;
  if (i >= npix_total) return;
  const uint32_t h4 = *reinterpret_cast<const uint32_t*>(high + i);
  const uint32_t l4 = low ? *reinterpret_cast<const uint32_t*>(low + i) : 0u;
  uint2 o;
  o.x = __builtin_amdgcn_perm(h4, l4, 0x05010400u);
  o.y = __builtin_amdgcn_perm(h4, l4, 0x07030602u);
  *reinterpret_cast<uint2*>(img + i) = o;
}

__global__ __launch_bounds__(kThreads) void k_split_decoded(
    const uint16_t* __restrict__ img, uint8_t* __restrict__ high,
    uint8_t* __restrict__ low, uint32_t npix) {
  uint32_t i = (blockIdx.x * kThreads + threadIdx.x) * 4;
  if (i >= npix) return;
  const uint2 v = *reinterpret_cast<const uint2*>(img + i);
  uint32_t h4, l4;
  split4<false>(v.x, v.y, 0, &h4, &l4);  // delta >> 8, delta & 0xff (cc:337-338)
  *reinterpret_cast<uint32_t*>(high + i) = h4;
  *reinterpret_cast<uint32_t*>(low + i) = l4;
}

// UnextractFrame (cc:850-862) on the device, in place: decoded pixels -> the raw file
// representation (value >> shift, stored little- or big-endian).  8 pixels per thread.
__global__ __launch_bounds__(kThreads) void k_unextract(uint16_t* __restrict__ px, size_t npix_total,
                                                       int shift, int swap) {
  const size_t i = ((size_t)blockIdx.x * kThreads + threadIdx.x) * 8;
  if (i >= npix_total) return;
  const uint32_t keep = (0xffffu >> shift) * 0x00010001u;
  auto conv = [&](uint32_t v) {
    v = (v >> shift) & keep;                                  // both 16-bit lanes
    return swap ? __builtin_amdgcn_perm(0, v, 0x02030001u) : v;  // bytes of each lane exchanged
  };
  if (i + 8 <= npix_total) {
    uint4 v = *reinterpret_cast<const uint4*>(px + i);
    v.x = conv(v.x); v.y = conv(v.y); v.z = conv(v.z); v.w = conv(v.w);
    *reinterpret_cast<uint4*>(px + i) = v;
  } else {
    for (size_t k = i; k < npix_total; k++) {
      const uint32_t u = (uint32_t)px[k] >> shift;
      px[k] = (uint16_t)(swap ? ((u & 0xffu) << 8) | (u >> 8) : u);
    }
  }
}

// ------------------------------------------------------- forward predictor --
//
// The two predictor decisions of a frame depend on histograms of sampled
// pixels (cc:526-533, cc:553-564), i.e. the reference makes a pass over the
// frame before it transforms it.  A second read of every frame would cost a
// third of the HBM traffic, so the forward path SPECULATES:
//
//   k_stream       one pass per frame: split, preview, delta and CG residuals
//                  under the speculated flags, AND the exact histograms of the
//                  heuristics for that speculation, written as per-strip partials
//   k_decide_spec  reduces the partials, evaluates EstimateEntropy, compares
//                  the true decision with the speculation; mis-speculated
//                  frames go to list A (delta decision wrong: histograms of the
//                  CG decision are void) or list B (only the CG decision wrong)
//   k_redo_decide / k_apply_list
//                  exact, non-speculative redo of the listed frames (normally
//                  none: these launches exit at once)
//   k_finish       preview clamped gradient, flags byte, next speculation
//
// The speculation only ever changes speed: every frame's output is produced
// under its TRUE flags.

constexpr int kMaxIter = 8;        // passes of 256 threads x 16 pixels over a redo tile (bounds its LDS: xsize <= 8192)
constexpr int kHistFused = 3;      // partial histograms per tile: d, a, b
constexpr int kListGrid = 1024;    // workgroups of the list-driven kernels

// Tile geometry shared by the tile routines.
struct TileCtx {
  uint32_t s, npx, nseg, rows, row0;
};

__device__ __forceinline__ TileCtx tile_ctx(const Geom& g, uint32_t tile) {
  TileCtx t;
  t.row0 = tile * g.tile_rows;
  t.rows = min(g.tile_rows, g.ys - t.row0);
  t.s = t.row0 * g.xs;
  t.npx = t.rows * g.xs;
  t.nseg = t.npx >> 4;
  return t;
}

// Dynamic LDS layout of the tile routines:
//   [0, lds_h_bytes)            high line buffer: halo (xs+16 bytes in front of
//                               lds_off) + the tile's post-delta high bytes
//   [lds_h_bytes, +tile_px/2)   4-pixel sums of the raw high bytes (u16)
//   then 3 x 256 u32            histograms (fused kernel)
//   last 16 bytes               OR of the raw low bytes
struct TileLds {
  uint8_t* hbuf;
  uint16_t* psum;
  uint32_t* hist;
  uint32_t* s_or;
};

__device__ __forceinline__ TileLds tile_lds(uint8_t* smem, const Geom& g) {
  TileLds l;
  l.hbuf = smem;
  l.psum = reinterpret_cast<uint16_t*>(smem + g.lds_h_bytes);
  l.hist = reinterpret_cast<uint32_t*>(smem + g.lds_hist_off);
  l.s_or = reinterpret_cast<uint32_t*>(smem + g.lds_bytes - 16);
  return l;
}

// Halo: post-delta high bytes of the xs+16 pixels in front of the tile, 4 per
// thread and step.  dcache: this thread's cached delta-high dwords (or null).
template <bool SWAP>
__device__ __forceinline__ void load_halo(const Geom& g, const TileCtx& t, const TileLds& L,
                                          const uint16_t* img, const uint8_t* dhigh,
                                          bool use_delta) {
  const uint32_t hs = t.s - g.xs - 16, ngrp = (g.xs + 16) >> 2;
  for (uint32_t k = threadIdx.x; k < ngrp; k += kThreads) {
    const uint2 v = *reinterpret_cast<const uint2*>(img + hs + 4 * k);
    uint32_t h4, l4;
    split4<SWAP>(v.x, v.y, g.shift, &h4, &l4);
    if (use_delta) h4 = sub4(h4, *reinterpret_cast<const uint32_t*>(dhigh + hs + 4 * k));
    *reinterpret_cast<uint32_t*>(L.hbuf + g.lds_off - (g.xs + 16) + 4 * k) = h4;
  }
}

// Phase A for one 16-pixel segment: split, preview sums, NO_LOW OR, optional
// histogram sample of the raw high byte (i % 15 == 0), delta, low plane store,
// high bytes into LDS (and to global when CG is off).
template <bool SWAP, bool STATS>
__device__ __forceinline__ void seg_phase_a(const Geom& g, const TileCtx& t, const TileLds& L,
                                            uint32_t o, const uint4& pa, const uint4& pb,
                                            const uint4& dh, const uint4& dl, bool use_delta,
                                            bool use_cg, uint8_t* high_f, uint8_t* low_f,
                                            uint32_t* or_low) {
  const int S = g.shift;
  uint4 h, l;
  split4<SWAP>(pa.x, pa.y, S, &h.x, &l.x);
  split4<SWAP>(pa.z, pa.w, S, &h.y, &l.y);
  split4<SWAP>(pb.x, pb.y, S, &h.z, &l.z);
  split4<SWAP>(pb.z, pb.w, S, &h.w, &l.w);
  uint2 ps;  // cc:500-509: sums of 4 consecutive RAW high bytes
  ps.x = __builtin_amdgcn_sad_u8(h.x, 0, 0) | (__builtin_amdgcn_sad_u8(h.y, 0, 0) << 16);
  ps.y = __builtin_amdgcn_sad_u8(h.z, 0, 0) | (__builtin_amdgcn_sad_u8(h.w, 0, 0) << 16);
  *reinterpret_cast<uint2*>(L.psum + (o >> 2)) = ps;
  if (S != 8) *or_low |= l.x | l.y | l.z | l.w;  // RAW low bytes (cc:396, 414, 426, 442)
  if (STATS) {  // cc:526-531: counta of high[0::15]
    const uint32_t m = (t.s + o) % 15u;
    const uint32_t j0 = m ? 15u - m : 0u;  // 0..14: first sampled pixel of the segment
    const uint32_t d = j0 >> 2;
    uint32_t v = d == 0 ? h.x : d == 1 ? h.y : d == 2 ? h.z : h.w;
    atomicAdd(&L.hist[(v >> (8 * (j0 & 3))) & 0xffu], 1u);
    if (j0 == 0) atomicAdd(&L.hist[h.w >> 24], 1u);  // pixel 15 of the segment
  }
  if (use_delta) {  // cc:534-537
    h.x = sub4(h.x, dh.x); h.y = sub4(h.y, dh.y);
    h.z = sub4(h.z, dh.z); h.w = sub4(h.w, dh.w);
    if (S != 8) {
      l.x = sub4(l.x, dl.x); l.y = sub4(l.y, dl.y);
      l.z = sub4(l.z, dl.z); l.w = sub4(l.w, dl.w);
    }
  }
  if (S != 8) *reinterpret_cast<uint4*>(low_f + t.s + o) = l;
  *reinterpret_cast<uint4*>(L.hbuf + g.lds_off + o) = h;
  if (!use_cg) *reinterpret_cast<uint4*>(high_f + t.s + o) = h;
}

// Phase B for one segment: clamped-gradient residuals (cc:566-572) from the LDS
// line buffer and, with STATS, the CG heuristic's sample (cc:554-562).
template <bool STATS>
__device__ __forceinline__ void seg_phase_b(const Geom& g, const TileCtx& t, const TileLds& L,
                                            uint32_t o, bool use_cg, uint8_t* high_f) {
  const uint8_t* cp = L.hbuf + g.lds_off + o;
  if (use_cg) {
    const uint8_t* np = cp - g.xs;
    uint32_t c[5], n[5];
    c[0] = *reinterpret_cast<const uint32_t*>(cp - 4);
    const uint4 cv = *reinterpret_cast<const uint4*>(cp);
    c[1] = cv.x; c[2] = cv.y; c[3] = cv.z; c[4] = cv.w;
#pragma unroll
    for (int k = 0; k < 5; k++) n[k] = *reinterpret_cast<const uint32_t*>(np - 4 + 4 * k);
    uint32_t out[4];
    int wprev = (int)byte_of(c[0], 3), nwprev = (int)byte_of(n[0], 3);
    const bool lit = t.s + o < g.xs + 1;  // segment reaches into the literal prefix (cc:572)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      uint32_t r = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int x = (int)byte_of(c[k + 1], j), nn = (int)byte_of(n[k + 1], j);
        int res = (x - cg_pred(nn, wprev, nwprev)) & 0xff;
        if (lit && t.s + o + 4 * k + j < g.xs + 1) res = x;
        r |= (uint32_t)res << (8 * j);
        wprev = x;
        nwprev = nn;
      }
      out[k] = r;
    }
    *reinterpret_cast<uint4*>(high_f + t.s + o) = make_uint4(out[0], out[1], out[2], out[3]);
  }
  if (STATS) {  // samples at i = xs + 1 + 31 k
    const int q = (int)(t.s + o) - (int)(g.xs + 1);
    uint32_t j1;
    if (q >= 0) {
      const uint32_t r = (uint32_t)q % 31u;
      j1 = r ? 31u - r : 0u;
    } else {
      j1 = (uint32_t)(-q);
    }
    if (j1 < 16) {
      const uint8_t* p = cp + j1;
      const int x = p[0], w = p[-1], n = p[-(int)g.xs], nw = p[-(int)g.xs - 1];
      atomicAdd(&L.hist[256 + x], 1u);
      atomicAdd(&L.hist[512 + ((x - cg_pred(n, w, nw)) & 0xff)], 1u);
    }
  }
}

// 4x4 preview boxes of the tile from the 4-pixel sums (cc:500-512).
__device__ __forceinline__ void tile_preview(const Geom& g, const TileCtx& t, const TileLds& L,
                                             uint8_t* pv_f) {
  const uint32_t xq = g.xs >> 2;  // psum entries per image row
  uint8_t* pvo = pv_f + (size_t)(t.row0 >> 2) * g.pw;
  if ((g.pw & 3) == 0) {  // 4 preview pixels per thread: 8-byte LDS reads, one dword store
    const uint32_t pw4 = g.pw >> 2, ngrp = (t.rows >> 2) * pw4;
    for (uint32_t q = threadIdx.x; q < ngrp; q += kThreads) {
      const uint32_t pr = q / pw4, pc = (q - pr * pw4) << 2;
      const uint16_t* p = L.psum + pr * g.xs + pc;
      uint2 a = *reinterpret_cast<const uint2*>(p);
#pragma unroll
      for (int r = 1; r < 4; r++) {
        const uint2 b = *reinterpret_cast<const uint2*>(p + r * xq);
        a.x += b.x;  // 16-bit lanes cannot overflow: 16 * 255 < 65536
        a.y += b.y;
      }
      a.x = (a.x >> 4) & 0x00fe00feu;
      a.y = (a.y >> 4) & 0x00fe00feu;
      *reinterpret_cast<uint32_t*>(pvo + pr * g.pw + pc) = __builtin_amdgcn_perm(a.y, a.x, 0x06040200u);
    }
  } else {
    const uint32_t npq = (t.rows >> 2) * g.pw;
    for (uint32_t q = threadIdx.x; q < npq; q += kThreads) {
      const uint32_t pr = q / g.pw, pc = q - pr * g.pw;
      const uint16_t* p = L.psum + pr * g.xs + pc;
      pvo[q] = (uint8_t)(((p[0] + p[xq] + p[2 * xq] + p[3 * xq]) >> 4) & 0xfe);
    }
  }
}

__device__ __forceinline__ void tile_or_low(const TileLds& L, uint32_t or_low) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) or_low |= __shfl_xor(or_low, off, 64);
  if ((threadIdx.x & 63) == 0 && or_low) atomicOr(L.s_or, or_low);
}

// ---------------------------------------------------------------------------
// k_stream: the fused speculative pass as a barrier-free streaming kernel.
//
// One WAVEFRONT owns one strip (strip_rows image rows) of one frame at a time
// and pulls strips from a device-side counter.  Per step the wave handles 1024
// consecutive pixels of the flattened plane, 16 per lane:
//   - 32-byte coalesced pixel loads (and the delta bytes) come from a 3-deep
//     register prefetch pipeline: the loads of step t+3 are issued when step t
//     has consumed its registers, so each wave keeps ~6 KiB of HBM reads in
//     flight and no load is waited for right after its issue;
//   - the post-delta high bytes go into the wave's private LDS ring (the last
//     xs+16+1024 bytes of the plane): the clamped gradient reads its north /
//     north-west / west neighbours from there.  One wave executes its LDS
//     operations in order, so no barrier is needed;
//   - the 4x4 preview boxes are accumulated with LDS adds, one accumulator
//     row per 4-row group, and written when a group is complete;
//   - the heuristic samples (i % 15 == 0; i = xs+1+31k) are counted in the
//     wave's LDS histograms (16-bit bins packed in dwords).
// ---------------------------------------------------------------------------

constexpr uint32_t kStep = 1024;  // pixels per wave step

// Streaming accesses: pixels are read once and planes are written once, so they
// are marked non-temporal to keep the (re-used) delta planes resident in L2.
#ifndef FPV_NT
#define FPV_NT 1
#endif
__device__ __forceinline__ uint4 ld_stream(const void* p) {
#if FPV_NT
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
#else
  return *reinterpret_cast<const uint4*>(p);
#endif
}
__device__ __forceinline__ void st_stream(void* p, uint4 v) {
#if FPV_NT
  u32x4 t;
  t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(p));
#else
  *reinterpret_cast<uint4*>(p) = v;
#endif
}

typedef short pk16 __attribute__((ext_vector_type(2)));  // two 16-bit lanes in one VGPR

__device__ __forceinline__ pk16 as_pk(uint32_t v) { return __builtin_bit_cast(pk16, v); }
__device__ __forceinline__ uint32_t as_u32(pk16 v) { return __builtin_bit_cast(uint32_t, v); }

// x - ClampedGradient(n, w, nw) for two pixels at once (16-bit lanes holding
// 0..255): median(n, w, n + w - nw) = max(min(n,w), min(max(n,w), n+w-nw)).
// The low byte of each lane is the residual mod 256.
__device__ __forceinline__ uint32_t cg_res_pk(uint32_t x, uint32_t w, uint32_t n, uint32_t nw) {
  const pk16 pn = as_pk(n), pw = as_pk(w);
  const pk16 g = pn + pw - as_pk(nw);
  const pk16 lo = __builtin_elementwise_min(pn, pw), hi = __builtin_elementwise_max(pn, pw);
  const pk16 pred = __builtin_elementwise_max(lo, __builtin_elementwise_min(hi, g));
  return as_u32(as_pk(x) - pred);
}

// Residuals of 4 pixels (bytes of cur) given the same 4 bytes of the row above
// (north) and the previous dwords' odd bytes (for west / north-west of byte 0).
// *xo_prev / *no_prev: {b1, 0, b3, 0} of the previous cur / north dword; updated.
__device__ __forceinline__ uint32_t cg_res4(uint32_t cur, uint32_t north, uint32_t* xo_prev,
                                            uint32_t* no_prev) {
  const uint32_t xe = __builtin_amdgcn_perm(0, cur, 0x0c020c00u);    // {b0, 0, b2, 0}
  const uint32_t xo = __builtin_amdgcn_perm(0, cur, 0x0c030c01u);    // {b1, 0, b3, 0}
  const uint32_t ne = __builtin_amdgcn_perm(0, north, 0x0c020c00u);
  const uint32_t no = __builtin_amdgcn_perm(0, north, 0x0c030c01u);
  const uint32_t we = __builtin_amdgcn_alignbit(xo, *xo_prev, 16);   // {b-1, 0, b1, 0}
  const uint32_t nwe = __builtin_amdgcn_alignbit(no, *no_prev, 16);
  const uint32_t re = cg_res_pk(xe, we, ne, nwe);                    // pixels 0 and 2
  const uint32_t ro = cg_res_pk(xo, xe, no, ne);                     // pixels 1 and 3
  *xo_prev = xo;
  *no_prev = no;
  return __builtin_amdgcn_perm(ro, re, 0x06020400u);                 // low bytes, pixel order
}

struct Pre {  // prefetch registers of one step
  uint4 a, b, dh, dl;
};

#ifdef FPV_STAMPS
#define FPV_STAMP(k)                                                                      \
  do {                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    unsigned long long t_;                                                                \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    c.seg[k] += t_ - c.tlast;                                                             \
    c.tlast = t_;                                                                         \
  } while (0)
#else
#define FPV_STAMP(k)
#endif

struct StreamCtx {
#ifdef FPV_STAMPS
  unsigned long long seg[6], tlast;
#endif
  // constant per kernel
  uint32_t lane, xs, pw, ring_mask, ring_base, nacc, shift;
  uint8_t* ring;
  uint32_t* acc;     // [nacc][pw] preview accumulators
  uint32_t* hist;    // [3][128] packed 16-bit bins
  bool use_delta, use_cg;
  // per strip
  uint32_t s, npx;   // first pixel (frame-flat) and pixel count of the strip
  const uint16_t* img;
  const uint8_t *dh, *dl;
  uint8_t *high_f, *low_f, *pv_f;
  uint32_t row, col; // position of this lane's segment relative to the strip start
  uint32_t m15;      // (frame-flat index of the segment) mod 15
  int q31;           // index - (xs + 1): negative in front of the first CG sample, else mod 31
  uint32_t or_low;
  uint32_t groups_done, prow0;
};

__device__ __forceinline__ uint32_t ring_at(const StreamCtx& c, int o) {
  return ((uint32_t)(o + (int)c.ring_base)) & c.ring_mask;
}

__device__ __forceinline__ void hist_inc(uint32_t* hist, uint32_t which, uint32_t bin) {
  atomicAdd(&hist[which * 128 + (bin >> 1)], 1u << (16 * (bin & 1)));
}

// DELTA / LOW are the (wave-uniform) conditions "delta predictor speculated" and
// "a low plane exists" as template parameters, and FAST steps are always full:
// no control flow around the loads, so that the compiler can keep the three
// prefetch stages in flight with counted vmcnt waits.
template <bool FAST, bool DELTA, bool LOW>
__device__ __forceinline__ void stream_load(const StreamCtx& c, uint32_t it, uint32_t niter,
                                            Pre* p) {
  if (FAST) {
    const uint32_t o = min(it, niter - 1) * kStep + c.lane * 16;  // past the end: reload, unused
    const uint16_t* src = c.img + c.s + o;
    p->a = ld_stream(src);
    p->b = ld_stream(src + 8);
    if (DELTA) {
      p->dh = *reinterpret_cast<const uint4*>(c.dh + c.s + o);
      if (LOW) p->dl = *reinterpret_cast<const uint4*>(c.dl + c.s + o);
    }
  } else {
    const uint32_t o = it * kStep + c.lane * 16;
    if (o < c.npx) {
      const uint16_t* src = c.img + c.s + o;
      p->a = *reinterpret_cast<const uint4*>(src);
      p->b = *reinterpret_cast<const uint4*>(src + 8);
      if (DELTA) {
        p->dh = *reinterpret_cast<const uint4*>(c.dh + c.s + o);
        if (LOW) p->dl = *reinterpret_cast<const uint4*>(c.dl + c.s + o);
      }
    }
  }
}

// Writes the finished 4-row groups' preview pixels and clears their accumulators.
__device__ __forceinline__ void stream_preview_flush(StreamCtx& c, uint32_t upto_groups) {
  while (c.groups_done < upto_groups) {
    const uint32_t slot = c.nacc == 2 ? (c.groups_done & 1u) : c.groups_done % c.nacc;
    uint32_t* a = c.acc + slot * c.pw;
    uint8_t* out = c.pv_f + (size_t)(c.prow0 + c.groups_done) * c.pw;
    if ((c.pw & 3) == 0) {
      for (uint32_t k = c.lane * 4; k < c.pw; k += 256) {
        const uint4 v = *reinterpret_cast<const uint4*>(a + k);
        *reinterpret_cast<uint4*>(a + k) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint32_t*>(out + k) = ((v.x >> 4) & 0xfeu) | (((v.y >> 4) & 0xfeu) << 8) |
                                                (((v.z >> 4) & 0xfeu) << 16) | (((v.w >> 4) & 0xfeu) << 24);
      }
    } else {
      for (uint32_t k = c.lane; k < c.pw; k += 64) {
        out[k] = (uint8_t)((a[k] >> 4) & 0xfeu);
        a[k] = 0;
      }
    }
    c.groups_done++;
  }
}

// FAST: xs % 16 == 0 and xs >= 256 (a segment never straddles image rows, two
// preview accumulator rows suffice, the north bytes are one aligned 16-byte read).
template <bool SWAP, bool FAST, bool DELTA, bool LOW>
__device__ __forceinline__ void stream_step(StreamCtx& c, uint32_t it, const Pre& p) {
  const uint32_t o = it * kStep + c.lane * 16;
  const int S = (int)c.shift;
  FPV_STAMP(0);  // loop overhead + prefetch issue of the previous step
  if (FAST || o < c.npx) {
    uint4 h, l;
    split4<SWAP>(p.a.x, p.a.y, S, &h.x, &l.x);
    split4<SWAP>(p.a.z, p.a.w, S, &h.y, &l.y);
    split4<SWAP>(p.b.x, p.b.y, S, &h.z, &l.z);
    split4<SWAP>(p.b.z, p.b.w, S, &h.w, &l.w);
#ifdef FPV_STAMPS
    asm volatile("" ::"v"(h.x), "v"(h.w), "v"(l.x), "v"(l.w));
#endif
    FPV_STAMP(1);  // wait for the pixel loads + split
    // preview: 4-pixel sums of the RAW high bytes into the group accumulators (cc:500-509)
    {
      const uint32_t q[4] = {__builtin_amdgcn_sad_u8(h.x, 0, 0), __builtin_amdgcn_sad_u8(h.y, 0, 0),
                             __builtin_amdgcn_sad_u8(h.z, 0, 0), __builtin_amdgcn_sad_u8(h.w, 0, 0)};
      if (FAST) {
        uint32_t* a = c.acc + ((c.row >> 2) & 1u) * c.pw + (c.col >> 2);
#pragma unroll
        for (int j = 0; j < 4; j++) atomicAdd(a + j, q[j]);
      } else {
        uint32_t row = c.row, col = c.col;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (col >= c.xs) { col -= c.xs; row++; }
          atomicAdd(c.acc + ((row >> 2) % c.nacc) * c.pw + (col >> 2), q[j]);
          col += 4;
        }
      }
    }
    if (LOW) c.or_low |= l.x | l.y | l.z | l.w;  // RAW low bytes: NO_LOW_BYTES (cc:447)
    {  // cc:526-531: histogram of high[0::15]
      const uint32_t j0 = c.m15 ? 15u - c.m15 : 0u;
      const uint32_t d = j0 >> 2;
      const uint32_t v = d == 0 ? h.x : d == 1 ? h.y : d == 2 ? h.z : h.w;
      hist_inc(c.hist, 0, (v >> (8 * (j0 & 3))) & 0xffu);
      if (j0 == 0) hist_inc(c.hist, 0, h.w >> 24);
      c.m15 += kStep % 15u;
      if (c.m15 >= 15u) c.m15 -= 15u;
    }
    FPV_STAMP(2);  // preview adds + delta-histogram sample
    if (DELTA) {  // cc:534-537
      h.x = sub4(h.x, p.dh.x); h.y = sub4(h.y, p.dh.y);
      h.z = sub4(h.z, p.dh.z); h.w = sub4(h.w, p.dh.w);
      if (LOW) {
        l.x = sub4(l.x, p.dl.x); l.y = sub4(l.y, p.dl.y);
        l.z = sub4(l.z, p.dl.z); l.w = sub4(l.w, p.dl.w);
      }
    }
    if (LOW) st_stream(c.low_f + c.s + o, l);
    const uint32_t a0 = ring_at(c, (int)o);
    *reinterpret_cast<uint4*>(c.ring + a0) = h;
    if (!c.use_cg) st_stream(c.high_f + c.s + o, h);
    __builtin_amdgcn_wave_barrier();  // the reads below see the whole wave's ring writes
    FPV_STAMP(3);  // delta + low store + ring write
    if (c.use_cg) {  // cc:566-571
      uint32_t n0, c0;
      uint4 nv;
      c0 = *reinterpret_cast<const uint32_t*>(c.ring + ((a0 - 4) & c.ring_mask));
      if (FAST) {
        nv = *reinterpret_cast<const uint4*>(c.ring + ((a0 - c.xs) & c.ring_mask));
        n0 = *reinterpret_cast<const uint32_t*>(c.ring + ((a0 - c.xs - 4) & c.ring_mask));
      } else {
        n0 = *reinterpret_cast<const uint32_t*>(c.ring + ((a0 - c.xs - 4) & c.ring_mask));
        nv.x = *reinterpret_cast<const uint32_t*>(c.ring + ((a0 - c.xs) & c.ring_mask));
        nv.y = *reinterpret_cast<const uint32_t*>(c.ring + ((a0 - c.xs + 4) & c.ring_mask));
        nv.z = *reinterpret_cast<const uint32_t*>(c.ring + ((a0 - c.xs + 8) & c.ring_mask));
        nv.w = *reinterpret_cast<const uint32_t*>(c.ring + ((a0 - c.xs + 12) & c.ring_mask));
      }
      uint32_t xo_prev = __builtin_amdgcn_perm(0, c0, 0x0c030c01u);
      uint32_t no_prev = __builtin_amdgcn_perm(0, n0, 0x0c030c01u);
      uint4 r;
      r.x = cg_res4(h.x, nv.x, &xo_prev, &no_prev);
      r.y = cg_res4(h.y, nv.y, &xo_prev, &no_prev);
      r.z = cg_res4(h.z, nv.z, &xo_prev, &no_prev);
      r.w = cg_res4(h.w, nv.w, &xo_prev, &no_prev);
      // the first xs + 1 bytes of the plane are literals (cc:572): wave-uniform test first
      if (c.s == 0 && it * kStep < c.xs + 1) {
        const int nlit = (int)(c.xs + 1) - (int)o;  // literal pixels from the segment start
        const uint32_t rr[4] = {r.x, r.y, r.z, r.w}, hh[4] = {h.x, h.y, h.z, h.w};
        uint32_t oo[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int nl = nlit - 4 * k;
          const uint32_t mask = nl >= 4 ? 0xffffffffu : nl <= 0 ? 0u : (1u << (8 * nl)) - 1u;
          oo[k] = (hh[k] & mask) | (rr[k] & ~mask);
        }
        r = make_uint4(oo[0], oo[1], oo[2], oo[3]);
      }
      st_stream(c.high_f + c.s + o, r);
    }
    FPV_STAMP(4);  // ring reads + clamped gradient + high store
    {  // cc:554-562: the CG heuristic's sample at i = xs + 1 + 31 k, if one falls in this segment
      const uint32_t j1 = c.q31 < 0 ? (uint32_t)(-c.q31) : (c.q31 ? 31u - (uint32_t)c.q31 : 0u);
      if (j1 < 16) {
        const uint32_t aj = a0 + j1;
        const int x = c.ring[aj & c.ring_mask], w = c.ring[(aj - 1) & c.ring_mask];
        const int n = c.ring[(aj - c.xs) & c.ring_mask], nw = c.ring[(aj - c.xs - 1) & c.ring_mask];
        hist_inc(c.hist, 1, (uint32_t)x);
        hist_inc(c.hist, 2, (uint32_t)(x - cg_pred(n, w, nw)) & 0xffu);
      }
      if (c.q31 < 0) {
        c.q31 += (int)kStep;
        if (c.q31 >= 0) c.q31 %= 31;
      } else {
        c.q31 += (int)(kStep % 31u);
        if (c.q31 >= 31) c.q31 -= 31;
      }
    }
    // advance this lane's (row, col) by one step
    c.col += kStep % c.xs;
    c.row += kStep / c.xs;
    if (c.col >= c.xs) { c.col -= c.xs; c.row++; }
  }
  __builtin_amdgcn_wave_barrier();
  // preview groups completed by this step (wave-uniform)
  const uint32_t end = min((it + 1) * kStep, c.npx);
  stream_preview_flush(c, end / (4 * c.xs));
  FPV_STAMP(5);  // CG sample + bookkeeping + preview flush
}

// Kernel arguments kept small on purpose: everything else is derived.
struct StreamArgs {
  uint32_t xs, ys, npix, pw, npv;
  uint32_t strip_rows, strips, ring_fwd, ring_base, nacc, shift, nframes, strip_major;
};

template <bool SWAP, bool FAST, bool DELTA, bool LOW>
__global__ __launch_bounds__(64, 4) void k_stream(
    const uint16_t* __restrict__ frames, const uint8_t* __restrict__ dhigh,
    const uint8_t* __restrict__ dlow, const uint8_t* __restrict__ spec_ptr,
    uint8_t* __restrict__ high, uint8_t* __restrict__ low, uint8_t* __restrict__ pv_raw,
    uint32_t* __restrict__ anylow, uint16_t* __restrict__ partials,
    uint32_t* __restrict__ counters, StreamArgs g) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  StreamCtx c;
  c.lane = threadIdx.x;  // one wavefront per workgroup
  c.ring = smem;
  c.acc = reinterpret_cast<uint32_t*>(smem + g.ring_fwd);
  c.hist = reinterpret_cast<uint32_t*>(smem + g.ring_fwd + g.nacc * g.pw * 4);
  c.xs = g.xs; c.pw = g.pw; c.ring_mask = g.ring_fwd - 1; c.ring_base = g.ring_base;
  c.nacc = g.nacc; c.shift = g.shift;
  const uint32_t spec = *spec_ptr & (dhigh != nullptr ? 3u : 2u);
  if (DELTA != (bool)(spec & 1)) return;  // the host launches both variants; the other one runs
  c.use_delta = DELTA;
  c.use_cg = spec & 2;
  c.dh = dhigh; c.dl = dlow;
#ifdef FPV_STAMPS
  for (int k = 0; k < 6; k++) c.seg[k] = 0;
  c.tlast = __builtin_amdgcn_s_memtime();
#endif
  for (uint32_t k = c.lane; k < g.nacc * g.pw; k += 64) c.acc[k] = 0;
  const uint32_t row_l = (c.lane * 16) / g.xs, col_l = (c.lane * 16) % g.xs;

  const uint32_t ntasks = g.nframes * g.strips;
  // The first task of a wave is its workgroup index; only later ones come from the
  // device-wide counter (2048 waves hitting one word at launch would queue for ~25 us:
  // same-address atomics serialise at ~12 ns).
  bool first = true;
  for (;;) {
    uint32_t task = blockIdx.x;
    if (!first) {
      if (c.lane == 0) task = gridDim.x + atomicAdd(&counters[6], 1u);
      task = __builtin_amdgcn_readfirstlane(task);
    }
    first = false;
    if (task >= ntasks) {
#ifdef FPV_STAMPS
      if (c.lane == 0)  // stamps leave through a buffer nothing else reads (counters[16..27])
        for (int k = 0; k < 6; k++)
          atomicAdd(reinterpret_cast<unsigned long long*>(counters + 16) + k, c.seg[k]);
#endif
      break;
    }
    // Strip-major order: waves resident at the same time work on the SAME strip of
    // different frames, so the strip's delta bytes are shared through L1/L2.
    uint32_t frame, strip;
    if (g.strip_major) {
      strip = task / g.nframes;
      frame = task - strip * g.nframes;
    } else {
      frame = task / g.strips;
      strip = task - frame * g.strips;
    }
    const uint32_t row0 = strip * g.strip_rows;
    const uint32_t rows = min(g.strip_rows, g.ys - row0);
    const size_t fbase = (size_t)frame * g.npix;
    c.s = row0 * g.xs;
    c.npx = rows * g.xs;
    c.img = frames + fbase;
    c.high_f = high + fbase;
    c.low_f = low + fbase;
    c.pv_f = pv_raw + (size_t)frame * g.npv;
    c.row = row_l; c.col = col_l;
    c.m15 = (c.s + c.lane * 16) % 15u;
    c.q31 = (int)(c.s + c.lane * 16) - (int)(g.xs + 1);
    if (c.q31 >= 0) c.q31 %= 31;
    c.or_low = 0;
    c.groups_done = 0;
    c.prow0 = row0 >> 2;
    const uint32_t niter = (c.npx + kStep - 1) / kStep;

    for (uint32_t k = c.lane * 4; k < kHistFused * 128; k += 256)
      *reinterpret_cast<uint4*>(c.hist + k) = make_uint4(0, 0, 0, 0);
    if (c.s > 0) {  // halo: post-delta high bytes of the xs+16 pixels in front of the strip
      const uint32_t hs = c.s - g.xs - 16, ngrp = (g.xs + 16) >> 2;
      for (uint32_t k = c.lane; k < ngrp; k += 64) {
        const uint2 v = *reinterpret_cast<const uint2*>(c.img + hs + 4 * k);
        uint32_t h4, l4;
        split4<SWAP>(v.x, v.y, (int)g.shift, &h4, &l4);
        if (c.use_delta) h4 = sub4(h4, *reinterpret_cast<const uint32_t*>(dhigh + hs + 4 * k));
        *reinterpret_cast<uint32_t*>(c.ring + ring_at(c, (int)(4 * k) - (int)(g.xs + 16))) = h4;
      }
    }
    Pre p0, p1, p2;
    stream_load<FAST, DELTA, LOW>(c, 0, niter, &p0);
    stream_load<FAST, DELTA, LOW>(c, 1, niter, &p1);
    stream_load<FAST, DELTA, LOW>(c, 2, niter, &p2);
    __builtin_amdgcn_wave_barrier();

    const uint32_t n3 = niter - niter % 3;
    for (uint32_t it = 0; it < n3; it += 3) {
      stream_step<SWAP, FAST, DELTA, LOW>(c, it, p0);
      stream_load<FAST, DELTA, LOW>(c, it + 3, niter, &p0);
      stream_step<SWAP, FAST, DELTA, LOW>(c, it + 1, p1);
      stream_load<FAST, DELTA, LOW>(c, it + 4, niter, &p1);
      stream_step<SWAP, FAST, DELTA, LOW>(c, it + 2, p2);
      stream_load<FAST, DELTA, LOW>(c, it + 5, niter, &p2);
    }
    if (n3 < niter) stream_step<SWAP, FAST, DELTA, LOW>(c, n3, p0);
    if (n3 + 1 < niter) stream_step<SWAP, FAST, DELTA, LOW>(c, n3 + 1, p1);

    // strip epilogue: NO_LOW accumulator and the partial histograms
    uint32_t orl = c.or_low;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) orl |= __shfl_xor(orl, off, 64);
    if (c.lane == 0 && orl) atomicOr(&anylow[frame], orl);
    uint32_t* ph = reinterpret_cast<uint32_t*>(partials + ((size_t)frame * g.strips + strip) * (kHistFused * 256));
    for (uint32_t k = c.lane * 4; k < kHistFused * 128; k += 256)
      *reinterpret_cast<uint4*>(ph + k) = *reinterpret_cast<const uint4*>(c.hist + k);
    __builtin_amdgcn_wave_barrier();
  }
}

// EstimateEntropy (cc:235-244) of one 256-bin histogram, one bin per thread.
__device__ __forceinline__ uint64_t entropy_bins(uint32_t v, uint32_t* scratch) {
  const uint32_t sum = block_sum(v, scratch);
  if (sum == 0) return 0;
  const uint32_t lsum = 31 - __clz(sum);
  const uint32_t term = v ? v * (lsum - (31 - __clz(v))) : 0;
  const uint32_t s = block_sum(term, scratch);
  return (uint64_t)1024 * s / sum;
}

// Sum over the 64 lanes of a wave; every lane gets the result.
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// EstimateEntropy (cc:235-244) of one 256-bin histogram held by a wave, four
// bins per lane.  Same integer arithmetic as entropy_bins.
__device__ __forceinline__ uint64_t entropy_wave(const uint32_t v[4]) {
  const uint32_t sum = wave_sum(v[0] + v[1] + v[2] + v[3]);
  if (sum == 0) return 0;
  const uint32_t lsum = 31 - __clz(sum);
  uint32_t term = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) term += v[k] ? v[k] * (lsum - (31 - __clz(v[k]))) : 0;
  return (uint64_t)1024 * wave_sum(term) / sum;
}

// Workspace counters (u32): [0] |list A|, [1] |list B|, [6] k_stream's task counter.
// One WAVE per frame (four frames per workgroup): no barriers, the reductions are
// wave shuffles, so all frames of a batch decide concurrently.
__global__ __launch_bounds__(kThreads) void k_decide_spec(
    const uint16_t* __restrict__ partials, const uint8_t* __restrict__ spec_ptr,
    uint8_t* __restrict__ decision, uint32_t* __restrict__ counters,
    uint32_t* __restrict__ list_a, uint32_t* __restrict__ list_b, Geom g, int with_delta,
    uint32_t nframes) {
  const uint32_t frame = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (frame >= nframes) return;
  const uint32_t nparts = g.strips;
  const uint16_t* ph = partials + (size_t)frame * nparts * (kHistFused * 256) + lane * 4;
  uint32_t v[kHistFused][4] = {};
#pragma unroll 4
  for (uint32_t t = 0; t < nparts; t++) {
#pragma unroll
    for (int h = 0; h < kHistFused; h++) {
      const uint2 q = *reinterpret_cast<const uint2*>(ph + (t * kHistFused + h) * 256);  // 4 x u16
      v[h][0] += q.x & 0xffffu; v[h][1] += q.x >> 16;
      v[h][2] += q.y & 0xffffu; v[h][3] += q.y >> 16;
    }
  }
  const uint32_t spec = *spec_ptr & (with_delta ? 3u : 2u);
  // cc:533: E(countd) < E(counta); countd has everything in bin 0, E(countd) == 0
  const bool use_delta = with_delta && entropy_wave(v[0]) > 0;
  if (use_delta != (bool)(spec & 1)) {
    if (lane == 0) {
      decision[frame] = 0xff;
      list_a[atomicAdd(&counters[0], 1u)] = frame;
    }
    return;
  }
  const uint64_t ea = entropy_wave(v[1]);
  const uint64_t eb = entropy_wave(v[2]);
  const bool use_cg = eb < ea;  // cc:564
  if (lane == 0) {
    const uint32_t dec = (use_delta ? 1u : 0u) | (use_cg ? 2u : 0u);
    decision[frame] = (uint8_t)dec;
    if (use_cg != (bool)(spec & 2)) list_b[atomicAdd(&counters[1], 1u)] = frame;
  }
}

// ---- exact fallback for mis-speculated frames (list driven, grid stride) ----

// Exact decisions of the frames whose delta decision was mis-speculated (list A;
// normally empty, the launch exits at once): one workgroup per listed frame takes
// all five histograms (both delta hypotheses) of the heuristics' samples in LDS,
// evaluates EstimateEntropy and appends the frame to list B for the re-apply pass.
template <bool SWAP>
__global__ __launch_bounds__(kThreads) void k_redo_decide(
    const uint16_t* __restrict__ frames, const uint8_t* __restrict__ dhigh,
    uint8_t* __restrict__ decision, uint32_t* __restrict__ counters,
    const uint32_t* __restrict__ list_a, uint32_t* __restrict__ list_b, Geom g, int with_delta) {
  __shared__ uint32_t lh[kNumHists * 256];
  __shared__ uint32_t scratch[4];
  const uint32_t na = counters[0];
  for (uint32_t w = blockIdx.x; w < na; w += gridDim.x) {
    const uint32_t frame = list_a[w];
    const uint16_t* img = frames + (size_t)frame * g.npix;
    for (int b = threadIdx.x; b < kNumHists * 256; b += kThreads) lh[b] = 0;
    __syncthreads();
    uint32_t h, l;
    if (with_delta) {  // cc:526-531
      for (uint32_t k = threadIdx.x; k < g.nd; k += kThreads) {
        split1<SWAP>(img[15u * k], g.shift, &h, &l);
        atomicAdd(&lh[h], 1u);
      }
    }
    for (uint32_t k = threadIdx.x; k < g.ncg; k += kThreads) {  // cc:554-562
      const uint32_t i = g.xs + 1 + 31u * k;
      uint32_t x, n, ww, nw;
      split1<SWAP>(img[i], g.shift, &x, &l);
      split1<SWAP>(img[i - g.xs], g.shift, &n, &l);
      split1<SWAP>(img[i - 1], g.shift, &ww, &l);
      split1<SWAP>(img[i - g.xs - 1], g.shift, &nw, &l);
      atomicAdd(&lh[1 * 256 + x], 1u);
      atomicAdd(&lh[2 * 256 + ((x - cg_pred(n, ww, nw)) & 0xffu)], 1u);
      if (with_delta) {
        x = (x - dhigh[i]) & 0xffu;
        n = (n - dhigh[i - g.xs]) & 0xffu;
        ww = (ww - dhigh[i - 1]) & 0xffu;
        nw = (nw - dhigh[i - g.xs - 1]) & 0xffu;
        atomicAdd(&lh[3 * 256 + x], 1u);
        atomicAdd(&lh[4 * 256 + ((x - cg_pred(n, ww, nw)) & 0xffu)], 1u);
      }
    }
    __syncthreads();
    const bool use_delta = with_delta && entropy_bins(lh[threadIdx.x], scratch) > 0;
    const uint32_t* ha = lh + (use_delta ? 3 : 1) * 256;
    const uint64_t ea = entropy_bins(ha[threadIdx.x], scratch);
    const uint64_t eb = entropy_bins(ha[256 + threadIdx.x], scratch);
    const bool use_cg = eb < ea;
    if (threadIdx.x == 0) {
      decision[frame] = (uint8_t)((use_delta ? 1u : 0u) | (use_cg ? 2u : 0u));
      list_b[atomicAdd(&counters[1], 1u)] = frame;
    }
    __syncthreads();
  }
}

// Non-speculative tile pass under the frame's TRUE flags.
template <bool SWAP>
__global__ __launch_bounds__(kThreads) void k_apply_list(
    const uint16_t* __restrict__ frames, const uint8_t* __restrict__ dhigh,
    const uint8_t* __restrict__ dlow, const uint8_t* __restrict__ decision,
    const uint32_t* __restrict__ counters, const uint32_t* __restrict__ list_b,
    uint8_t* __restrict__ high, uint8_t* __restrict__ low, uint8_t* __restrict__ pv_raw,
    uint32_t* __restrict__ anylow, Geom g) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const TileLds L = tile_lds(smem, g);
  const uint32_t tid = threadIdx.x, items = counters[1] * g.tiles;
  const uint4 zero = make_uint4(0, 0, 0, 0);
  for (uint32_t w = blockIdx.x; w < items; w += gridDim.x) {
    const uint32_t frame = list_b[w / g.tiles];
    const TileCtx t = tile_ctx(g, w % g.tiles);
    const size_t fbase = (size_t)frame * g.npix;
    const uint16_t* img = frames + fbase;
    const uint32_t dec = decision[frame];
    const bool use_delta = dec & 1, use_cg = dec & 2;
    if (tid == 0) *L.s_or = 0;
    if (t.s > 0 && use_cg) load_halo<SWAP>(g, t, L, img, dhigh, use_delta);
    __syncthreads();
    uint32_t or_low = 0;
    for (uint32_t seg = tid; seg < t.nseg; seg += kThreads) {
      const uint32_t o = seg << 4;
      const uint4 pa = *reinterpret_cast<const uint4*>(img + t.s + o);
      const uint4 pb = *reinterpret_cast<const uint4*>(img + t.s + o + 8);
      uint4 dh = zero, dl = zero;
      if (use_delta) {
        dh = *reinterpret_cast<const uint4*>(dhigh + t.s + o);
        if (g.shift != 8) dl = *reinterpret_cast<const uint4*>(dlow + t.s + o);
      }
      seg_phase_a<SWAP, false>(g, t, L, o, pa, pb, dh, dl, use_delta, use_cg, high + fbase,
                               low + fbase, &or_low);
    }
    tile_or_low(L, or_low);
    __syncthreads();
    for (uint32_t seg = tid; seg < t.nseg; seg += kThreads)
      seg_phase_b<false>(g, t, L, seg << 4, use_cg, high + fbase);
    tile_preview(g, t, L, pv_raw + (size_t)frame * g.npv);
    if (tid == 0 && *L.s_or) atomicOr(&anylow[frame], *L.s_or);
    __syncthreads();
  }
}

// Preview clamped gradient (cc:575-586), final flags byte (cc:447-449) and the
// speculation for the next batch (majority of this batch's decisions).
// WIDE: pw % 16 == 0 -> 16 preview pixels per thread with the packed median.
template <bool WIDE>
__global__ __launch_bounds__(kThreads) void k_finish(
    const uint8_t* __restrict__ pv_raw, const uint8_t* __restrict__ decision,
    const uint32_t* __restrict__ anylow, const uint32_t* __restrict__ counters,
    uint8_t* __restrict__ preview, uint8_t* __restrict__ flags, uint8_t* __restrict__ spec_ptr,
    Geom g) {
  const uint32_t frame = blockIdx.y;
  const uint32_t dec = decision[frame];
  const uint8_t* in = pv_raw + (size_t)frame * g.npv;
  uint8_t* out = preview + (size_t)frame * g.npv;
  constexpr uint32_t kPer = WIDE ? 16 : 4;
  const uint32_t i = (blockIdx.x * kThreads + threadIdx.x) * kPer;
  if (i < g.npv) {
    if (WIDE) {
      const uint4 c4 = *reinterpret_cast<const uint4*>(in + i);
      uint4 r4 = c4;
      if ((dec & 2) && i >= g.pw) {  // rows >= 1; row 0 and pixel pw are literals
        const uint4 n4 = *reinterpret_cast<const uint4*>(in + i - g.pw);
        // west / north-west of the first pixel: the dwords in front of the two segments
        uint32_t xo_prev = __builtin_amdgcn_perm(0, *reinterpret_cast<const uint32_t*>(in + i - 4), 0x0c030c01u);
        uint32_t no_prev = i > g.pw ? __builtin_amdgcn_perm(0, *reinterpret_cast<const uint32_t*>(in + i - g.pw - 4), 0x0c030c01u) : 0u;
        r4.x = cg_res4(c4.x, n4.x, &xo_prev, &no_prev);
        r4.y = cg_res4(c4.y, n4.y, &xo_prev, &no_prev);
        r4.z = cg_res4(c4.z, n4.z, &xo_prev, &no_prev);
        r4.w = cg_res4(c4.w, n4.w, &xo_prev, &no_prev);
        if (i == g.pw) r4.x = (r4.x & 0xffffff00u) | (c4.x & 0xffu);  // pixel pw itself is a literal
      }
      *reinterpret_cast<uint4*>(out + i) = r4;
    } else if ((g.pw & 3) == 0) {  // npv and pw multiples of 4: dword path
      const uint32_t c4 = *reinterpret_cast<const uint32_t*>(in + i);
      uint32_t r4 = c4;
      if ((dec & 2) && i + 3 >= g.pw + 1) {
        const uint32_t n4 = *reinterpret_cast<const uint32_t*>(in + i - g.pw);
        int w = in[i - 1], nw = i > g.pw ? in[i - g.pw - 1] : 0;  // pixel pw is a literal
        r4 = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int x = (int)byte_of(c4, j), n = (int)byte_of(n4, j);
          int res = (x - cg_pred(n, w, nw)) & 0xff;
          if (i + j < g.pw + 1) res = x;
          r4 |= (uint32_t)res << (8 * j);
          w = x;
          nw = n;
        }
      }
      *reinterpret_cast<uint32_t*>(out + i) = r4;
    } else {
      for (uint32_t k = i; k < min(i + 4, g.npv); k++) {
        int x = in[k];
        if ((dec & 2) && k >= g.pw + 1)
          x = (x - cg_pred(in[k - g.pw], in[k - 1], in[k - g.pw - 1])) & 0xff;
        out[k] = (uint8_t)x;
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) flags[frame] = (uint8_t)(dec | (anylow[frame] ? 0 : 4));
  if (blockIdx.x == 0 && frame == 0) {
    // Next batch's speculation = the most frequent decision of this one.  All
    // decisions are final here (stream order).  Counted by this one workgroup from
    // the decision bytes: a thousand frames voting with global atomics on one word
    // cost the decision kernel ~12 us (same-address atomics serialise at ~12 ns).
    __shared__ uint32_t votes[4];
    if (threadIdx.x < 4) votes[threadIdx.x] = 0;
    __syncthreads();
    uint32_t mine[4] = {0, 0, 0, 0};
    for (uint32_t f = threadIdx.x; f < gridDim.y; f += kThreads) mine[decision[f] & 3u]++;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (mine[k]) atomicAdd(&votes[k], mine[k]);
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t best = 0;
      for (uint32_t k = 1; k < 4; k++)
        if (votes[k] > votes[best]) best = k;
      *spec_ptr = (uint8_t)best;
    }
  }
}

// ------------------------------------------------------- inverse predictor --

// One wavefront per frame.  The inverse clamped gradient is ONE serial chain of
// npix steps (pixel i needs the reconstructed i-1, also across row ends), so
// the only parallelism is across frames.  Lane 0 walks the chain through an
// LDS ring of the last `ring` reconstructed bytes; all 64 lanes stage the
// residuals in and do delta^-1 + recombine + the coalesced uint16 stores.
constexpr uint32_t kInvChunk = 1024;  // pixels per staging round (multiple of 256)

// ALIGNED: xs and npix are multiples of 4 (every real frame geometry): dword
// LDS/global accesses.  Otherwise (odd preview planes, odd-sized files from
// the CPU reference) a byte-granular path with the same structure.
template <bool ALIGNED>
__global__ __launch_bounds__(64) void k_inverse(
    const uint8_t* __restrict__ high, const uint8_t* __restrict__ low,
    const uint8_t* __restrict__ dhigh, const uint8_t* __restrict__ dlow,
    const uint8_t* __restrict__ flags, uint16_t* __restrict__ frames, Geom g) {
  extern __shared__ __attribute__((aligned(16))) uint8_t ring[];
  const uint32_t frame = blockIdx.x, lane = threadIdx.x;
  const uint32_t m = g.ring - 1;
  const size_t fbase = (size_t)frame * g.npix;
  const uint8_t* hp = high + fbase;
  const uint8_t* lp = low ? low + fbase : nullptr;
  uint16_t* outp = frames + fbase;
  const uint32_t f = flags[frame];
  const bool use_delta = (f & 1) && dhigh, use_cg = f & 2, no_low = (f & 4) || !lp;
  const uint32_t xs = g.xs;
  uint32_t w = 0;  // reconstructed value of pixel i-1 (lane 0)

  for (uint32_t c0 = 0; c0 < g.npix; c0 += kInvChunk) {
    const uint32_t cn = min(kInvChunk, g.npix - c0);
    // stage residuals, coalesced
    if (ALIGNED) {
      for (uint32_t j = lane * 4; j < cn; j += 256)
        *reinterpret_cast<uint32_t*>(ring + ((c0 + j) & m)) =
            *reinterpret_cast<const uint32_t*>(hp + c0 + j);
    } else {
      for (uint32_t j = lane; j < cn; j += 64) ring[(c0 + j) & m] = hp[c0 + j];
    }
    __syncthreads();
    if (use_cg && lane == 0) {
      uint32_t i = c0;
      // literals (cc:327 starts at xs + 1) and the unaligned head
      while (i < c0 + cn && (i < xs + 1 || (i & 3) || !ALIGNED)) {
        uint32_t x = ring[i & m];
        if (i >= xs + 1) {
          x = (x + cg_pred(ring[(i - xs) & m], (int)w, ring[(i - xs - 1) & m])) & 0xffu;
          ring[i & m] = (uint8_t)x;
        }
        w = x;
        i++;
      }
      // aligned groups of 4; xs >= 4 keeps a group's north bytes outside itself
      if (ALIGNED && i < c0 + cn) {
        uint32_t r4 = *reinterpret_cast<uint32_t*>(ring + (i & m));
        uint32_t n4 = *reinterpret_cast<uint32_t*>(ring + ((i - xs) & m));
        uint32_t nw = ring[(i - xs - 1) & m];
        for (; i < c0 + cn; i += 4) {
          uint32_t r4n = 0, n4n = 0;
          const bool more = i + 4 < c0 + cn;
          if (more && xs >= 8) {  // prefetch: next group's north is older than this group
            r4n = *reinterpret_cast<uint32_t*>(ring + ((i + 4) & m));
            n4n = *reinterpret_cast<uint32_t*>(ring + ((i + 4 - xs) & m));
          }
          uint32_t o4 = 0;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int nn = (int)byte_of(n4, j);
            w = (byte_of(r4, j) + cg_pred(nn, (int)w, (int)nw)) & 0xffu;
            nw = nn;
            o4 |= w << (8 * j);
          }
          *reinterpret_cast<uint32_t*>(ring + (i & m)) = o4;
          if (more && xs < 8) {
            r4n = *reinterpret_cast<uint32_t*>(ring + ((i + 4) & m));
            n4n = *reinterpret_cast<uint32_t*>(ring + ((i + 4 - xs) & m));
          }
          r4 = r4n;
          n4 = n4n;
        }
      }
    }
    __syncthreads();
    // delta^-1 + recombine (cc:335-344)
    if (ALIGNED) {
      for (uint32_t j = lane * 4; j < cn; j += 256) {
        uint32_t h4 = *reinterpret_cast<uint32_t*>(ring + ((c0 + j) & m));
        uint32_t l4 = no_low ? 0u : *reinterpret_cast<const uint32_t*>(lp + c0 + j);
        if (use_delta) {
          h4 = add4(h4, *reinterpret_cast<const uint32_t*>(dhigh + c0 + j));
          l4 = add4(l4, *reinterpret_cast<const uint32_t*>(dlow + c0 + j));
        }
        uint2 o;
        o.x = __builtin_amdgcn_perm(h4, l4, 0x05010400u);  // l0 h0 l1 h1
        o.y = __builtin_amdgcn_perm(h4, l4, 0x07030602u);  // l2 h2 l3 h3
        *reinterpret_cast<uint2*>(outp + c0 + j) = o;
      }
    } else {
      for (uint32_t j = lane; j < cn; j += 64) {
        uint32_t h = ring[(c0 + j) & m], l = no_low ? 0u : lp[c0 + j];
        if (use_delta) {
          h = (h + dhigh[c0 + j]) & 0xffu;
          l = (l + dlow[c0 + j]) & 0xffu;
        }
        outp[c0 + j] = (uint16_t)((h << 8) | l);
      }
    }
    // the next round overwrites ring bytes older than xs + 1 + kInvChunk only
    __syncthreads();
  }
}

// Fast inverse (xs % 4 == 0, npix % 4 == 0, xs >= 256).  Still one wavefront
// per frame and still a serial chain -- that is the algorithm (cc:327-332) --
// but the chain is cut down to three dependent VALU instructions per pixel.
// Values are kept as v << 8 in 16-bit lanes, so that "+ r mod 256" is one
// 16-bit add that wraps by itself:
//
//     T = W + D (32-bit, d = n - nw may be negative);  M = med3_i32(N, W, T);
//     X = add_u16(M, R)
//
// Everything that does not depend on the running value W is prepared for a
// whole 256-pixel chunk by all 64 lanes in parallel and left in LINEAR LDS
// arrays (one dword per pixel: N, D, R), so the chain issues only 128-bit LDS
// reads with immediate offsets, the three VALU ops and one 128-bit LDS write
// per four pixels -- no address arithmetic, no branches.  All lanes run the
// chain redundantly on wave-uniform data; the other phases are data parallel.
constexpr uint32_t kInvC = 256;
constexpr uint32_t kSpecMaxK4 = 6;  // speculative inverse: at most 24 pixels per lane and superstep
// speculative inverse: passes of a superstep during which run-of-shift-lanes proposals are made;
// afterwards the plain rule (input := previous lane's output) guarantees the end within 64 passes
constexpr uint32_t kScanAfter = 4, kScanUntil = 28;  // measured: the unrolled 4-pass loop is the fastest common case

// One-wave workgroups: LDS hand-offs between the phases only need this wave's
// own LDS operations to have completed.  __syncthreads() would also drain the
// vector-memory counter and so wait for the global prefetch issued just before.
__device__ __forceinline__ void wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t inv_px(uint32_t n, uint32_t w, uint32_t d, uint32_t r) {
  const int t = (int)w + (int)d;
  const int lo = min((int)n, (int)w), hi = max((int)n, (int)w);
  const int md = max(lo, min(hi, t));                 // one v_med3_i32
  return as_u32(as_pk((uint32_t)md) + as_pk(r));      // v_pk_add_u16: low lane wraps mod 2^16
}

__global__ __launch_bounds__(64) void k_inverse_fast(
    const uint8_t* __restrict__ high, const uint8_t* __restrict__ low,
    const uint8_t* __restrict__ dhigh, const uint8_t* __restrict__ dlow,
    const uint8_t* __restrict__ flags, uint16_t* __restrict__ frames, Geom g) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t frame = blockIdx.x, lane = threadIdx.x;
  const size_t fbase = (size_t)frame * g.npix;
  const uint8_t* hp = high + fbase;
  const uint8_t* lp = low ? low + fbase : nullptr;
  uint16_t* outp = frames + fbase;
  const uint32_t f = flags[frame];
  const bool use_delta = (f & 1) && dhigh, use_cg = f & 2, no_low = (f & 4) || !lp;
  const uint32_t xs = g.xs, npix = g.npix;

  if (!use_cg) {  // no chain: delta^-1 + recombine, 16 pixels per lane and step
    for (uint32_t o = lane * 16; o < npix; o += 1024) {
      uint4 h = *reinterpret_cast<const uint4*>(hp + o);
      uint4 l = no_low ? make_uint4(0, 0, 0, 0) : *reinterpret_cast<const uint4*>(lp + o);
      if (use_delta) {
        const uint4 dh = *reinterpret_cast<const uint4*>(dhigh + o);
        const uint4 dl = *reinterpret_cast<const uint4*>(dlow + o);
        h.x = add4(h.x, dh.x); h.y = add4(h.y, dh.y); h.z = add4(h.z, dh.z); h.w = add4(h.w, dh.w);
        l.x = add4(l.x, dl.x); l.y = add4(l.y, dl.y); l.z = add4(l.z, dl.z); l.w = add4(l.w, dl.w);
      }
      uint4 a, b;
      a.x = __builtin_amdgcn_perm(h.x, l.x, 0x05010400u); a.y = __builtin_amdgcn_perm(h.x, l.x, 0x07030602u);
      a.z = __builtin_amdgcn_perm(h.y, l.y, 0x05010400u); a.w = __builtin_amdgcn_perm(h.y, l.y, 0x07030602u);
      b.x = __builtin_amdgcn_perm(h.z, l.z, 0x05010400u); b.y = __builtin_amdgcn_perm(h.z, l.z, 0x07030602u);
      b.z = __builtin_amdgcn_perm(h.w, l.w, 0x05010400u); b.w = __builtin_amdgcn_perm(h.w, l.w, 0x07030602u);
      *reinterpret_cast<uint4*>(outp + o) = a;
      *reinterpret_cast<uint4*>(outp + o + 8) = b;
    }
    return;
  }

  // LDS: ring of reconstructed values (v << 8, one dword per pixel), then the
  // chunk's linear operand arrays and result array.
  uint32_t* xr = reinterpret_cast<uint32_t*>(smem);
  const uint32_t m = g.inv_ring - 1;  // ring size in pixels (power of two)
  uint32_t* nn = xr + g.inv_ring;
  uint32_t* dd = nn + kInvC;
  uint32_t* rr = dd + kInvC;
  uint32_t* xx = rr + kInvC;
  uint32_t w = 0;

  // Global operands of a chunk (this lane's 4 pixels): fetched ONE CHUNK AHEAD, so
  // that their latency hides behind the previous chunk's chain.
  struct Fetch { uint32_t r4, l4, dh4, dl4; };
  auto fetch = [&](uint32_t c0) {
    Fetch f = {0, 0, 0, 0};
    const uint32_t i = c0 + lane * 4;
    if (i < npix) {
      f.r4 = *reinterpret_cast<const uint32_t*>(hp + i);
      if (!no_low) f.l4 = *reinterpret_cast<const uint32_t*>(lp + i);
      if (use_delta) {
        f.dh4 = *reinterpret_cast<const uint32_t*>(dhigh + i);
        f.dl4 = *reinterpret_cast<const uint32_t*>(dlow + i);
      }
    }
    return f;
  };
  Fetch cur = fetch(0);

  for (uint32_t c0 = 0; c0 < npix; c0 += kInvC) {
    const uint32_t cn = min(kInvC, npix - c0);
    const uint32_t j = lane * 4, i = c0 + j;
    // prep (all lanes, 4 pixels each)
    if (j < cn) {
      const uint32_t r4 = cur.r4;
      *reinterpret_cast<uint4*>(rr + j) =
          make_uint4((r4 << 8) & 0xff00u, r4 & 0xff00u, (r4 >> 8) & 0xff00u, (r4 >> 16) & 0xff00u);
      if (i >= xs + 4) {  // earlier groups go through the head path below
        const uint4 n4 = *reinterpret_cast<const uint4*>(xr + ((i - xs) & m));
        const uint32_t nw0 = xr[(i - xs - 1) & m];
        *reinterpret_cast<uint4*>(nn + j) = n4;
        *reinterpret_cast<uint4*>(dd + j) = make_uint4(n4.x - nw0, n4.y - n4.x, n4.z - n4.y, n4.w - n4.z);
      }
    }
    const Fetch next = fetch(c0 + kInvC);  // in flight while the chain below runs
    wave_lds_sync();
    // chain (wave-uniform)
    uint32_t jj = 0;
    for (; jj < cn && c0 + jj < xs + 4; jj += 4) {  // head: literals (cc:327) and the first group
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t ii = c0 + jj + k;
        uint32_t x = rr[jj + k];
        if (ii >= xs + 1) {
          const uint32_t n = xr[(ii - xs) & m], nw = xr[(ii - xs - 1) & m];
          x = inv_px(n, w, n - nw, x);
        }
        xr[ii & m] = x;  // the head's own north / north-west come from the ring
        xx[jj + k] = x;
        w = x;
      }
    }
    if (jj == 0 && cn == kInvC) {  // the common case: a whole chunk, fully unrolled
      {
      // 16 pixels per block; the operands of block q+1 are read from LDS before
      // block q's chain runs (two static register sets), so no LDS latency is
      // exposed between blocks.
      uint4 n4[2][4], d4[2][4], r4[2][4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        n4[0][k] = *reinterpret_cast<const uint4*>(nn + 4 * k);
        d4[0][k] = *reinterpret_cast<const uint4*>(dd + 4 * k);
        r4[0][k] = *reinterpret_cast<const uint4*>(rr + 4 * k);
      }
#pragma unroll
      for (int q = 0; q < (int)kInvC / 16; q++) {
        const int a = q & 1, b = a ^ 1;
        if (q + 1 < (int)kInvC / 16) {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            n4[b][k] = *reinterpret_cast<const uint4*>(nn + 16 * (q + 1) + 4 * k);
            d4[b][k] = *reinterpret_cast<const uint4*>(dd + 16 * (q + 1) + 4 * k);
            r4[b][k] = *reinterpret_cast<const uint4*>(rr + 16 * (q + 1) + 4 * k);
          }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          uint4 x4;
          x4.x = inv_px(n4[a][k].x, w, d4[a][k].x, r4[a][k].x);
          x4.y = inv_px(n4[a][k].y, x4.x, d4[a][k].y, r4[a][k].y);
          x4.z = inv_px(n4[a][k].z, x4.y, d4[a][k].z, r4[a][k].z);
          x4.w = inv_px(n4[a][k].w, x4.z, d4[a][k].w, r4[a][k].w);
          w = x4.w;
          *reinterpret_cast<uint4*>(xx + 16 * q + 4 * k) = x4;
        }
      }
      }
    } else {
      for (; jj < cn; jj += 4) {
        const uint4 n4 = *reinterpret_cast<const uint4*>(nn + jj);
        const uint4 d4 = *reinterpret_cast<const uint4*>(dd + jj);
        const uint4 r4 = *reinterpret_cast<const uint4*>(rr + jj);
        uint4 x4;
        x4.x = inv_px(n4.x, w, d4.x, r4.x);
        x4.y = inv_px(n4.y, x4.x, d4.y, r4.y);
        x4.z = inv_px(n4.z, x4.y, d4.z, r4.z);
        x4.w = inv_px(n4.w, x4.z, d4.w, r4.w);
        w = x4.w;
        *reinterpret_cast<uint4*>(xx + jj) = x4;
      }
    }
    wave_lds_sync();
    // output (all lanes, 4 pixels each): ring update, delta^-1 + recombine (cc:335-344)
    if (j < cn) {
      const uint4 x4 = *reinterpret_cast<const uint4*>(xx + j);
      *reinterpret_cast<uint4*>(xr + (i & m)) = x4;
      uint32_t h4 = (x4.x >> 8) | (x4.y & 0xff00u) | ((x4.z & 0xff00u) << 8) | ((x4.w & 0xff00u) << 16);
      uint32_t l4 = cur.l4;
      if (use_delta) {
        h4 = add4(h4, cur.dh4);
        l4 = add4(l4, cur.dl4);
      }
      uint2 o;
      o.x = __builtin_amdgcn_perm(h4, l4, 0x05010400u);
      o.y = __builtin_amdgcn_perm(h4, l4, 0x07030602u);
      *reinterpret_cast<uint2*>(outp + i) = o;
    }
    cur = next;
    wave_lds_sync();
  }
}


// Speculative segmented inverse (xs % 4 == 0, npix % 4 == 0, xs >= 256).
//
// The inverse clamped gradient x[i] = r[i] + CG(x[i-xs], x[i-1], x[i-xs-1]) is a
// serial chain in i (cc:327-332), but ONLY through w = x[i-1]: for a run of up
// to xs consecutive pixels every north / north-west value belongs to pixels in
// front of the run and is already final.  So one "superstep" takes S = 64*K
// consecutive pixels (S <= xs), gives each LANE a segment of K pixels with all
// its operands in registers, and lets every lane run its segment from a GUESSED
// incoming w (the pixel above its west neighbour).  The guess is then checked
// against what the previous lane actually produced; lanes whose input was wrong
// rerun their segment from the corrected value, until no lane is wrong.
//
// This is exact, not approximate: lane 0's input is the final value of the
// previous superstep, and once lane j's input equals lane j-1's output for all
// j, every segment is the true chain by induction.  After pass p lanes 0..p-1
// are final, so at most 64 passes (= the serial chain) are ever needed; on real
// frames the chain forgets its input within a few pixels whenever w falls into
// the predictor's flat range [min(n,nw), max(n,nw)], so a wrong guess rarely
// survives a whole segment and 2-4 passes of K steps replace 64*K serial steps.
template <int K4>
__global__ __launch_bounds__(64) void k_inverse_spec(
    const uint8_t* __restrict__ high, const uint8_t* __restrict__ low,
    const uint8_t* __restrict__ dhigh, const uint8_t* __restrict__ dlow,
    const uint8_t* __restrict__ flags, uint16_t* __restrict__ frames, Geom g,
    uint32_t* __restrict__ pass_count) {
  constexpr int K = 4 * K4;
  constexpr uint32_t S = 64u * K;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* ring = smem;  // reconstructed high bytes, flat index & m
  const uint32_t frame = blockIdx.x, lane = threadIdx.x;
  const size_t fbase = (size_t)frame * g.npix;
  const uint8_t* hp = high + fbase;
  const uint8_t* lp = low ? low + fbase : nullptr;
  uint16_t* outp = frames + fbase;
  const uint32_t f = flags[frame];
  const bool use_delta = (f & 1) && dhigh, use_cg = f & 2, no_low = (f & 4) || !lp;
  const uint32_t xs = g.xs, npix = g.npix, m = g.spec_ring - 1;

  // delta^-1 + recombine (cc:335-344) of the 4 pixels at flat index i
  auto emit4 = [&](uint32_t i, uint32_t h4, uint32_t l4, uint32_t dh4, uint32_t dl4) {
    if (use_delta) {
      h4 = add4(h4, dh4);
      l4 = add4(l4, dl4);
    }
    uint2 o;
    o.x = __builtin_amdgcn_perm(h4, l4, 0x05010400u);
    o.y = __builtin_amdgcn_perm(h4, l4, 0x07030602u);
    *reinterpret_cast<uint2*>(outp + i) = o;
  };

  if (!use_cg) {
    // No chain at all: delta^-1 + recombine.  The frame still belongs to this one
    // wave, so the loads of 4 steps (4096 pixels) are issued back to back before
    // anything is consumed (16 KiB in flight per wave), which
    // keeps such a frame well below the time of a clamped-gradient frame.
    constexpr int U = 4;
    uint32_t base = 0;
    for (; base + U * 1024 <= npix; base += U * 1024) {
      uint4 h[U], l[U], dh[U], dl[U];
#pragma unroll
      for (int u = 0; u < U; u++) h[u] = *reinterpret_cast<const uint4*>(hp + base + u * 1024 + lane * 16);
      if (!no_low) {
#pragma unroll
        for (int u = 0; u < U; u++) l[u] = *reinterpret_cast<const uint4*>(lp + base + u * 1024 + lane * 16);
      } else {
#pragma unroll
        for (int u = 0; u < U; u++) l[u] = make_uint4(0, 0, 0, 0);
      }
      if (use_delta) {
#pragma unroll
        for (int u = 0; u < U; u++) {
          dh[u] = *reinterpret_cast<const uint4*>(dhigh + base + u * 1024 + lane * 16);
          dl[u] = *reinterpret_cast<const uint4*>(dlow + base + u * 1024 + lane * 16);
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; u++) dh[u] = dl[u] = make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t hh[4] = {add4(h[u].x, dh[u].x), add4(h[u].y, dh[u].y), add4(h[u].z, dh[u].z), add4(h[u].w, dh[u].w)};
        const uint32_t ll[4] = {add4(l[u].x, dl[u].x), add4(l[u].y, dl[u].y), add4(l[u].z, dl[u].z), add4(l[u].w, dl[u].w)};
        uint4 a, b2;
        a.x = __builtin_amdgcn_perm(hh[0], ll[0], 0x05010400u); a.y = __builtin_amdgcn_perm(hh[0], ll[0], 0x07030602u);
        a.z = __builtin_amdgcn_perm(hh[1], ll[1], 0x05010400u); a.w = __builtin_amdgcn_perm(hh[1], ll[1], 0x07030602u);
        b2.x = __builtin_amdgcn_perm(hh[2], ll[2], 0x05010400u); b2.y = __builtin_amdgcn_perm(hh[2], ll[2], 0x07030602u);
        b2.z = __builtin_amdgcn_perm(hh[3], ll[3], 0x05010400u); b2.w = __builtin_amdgcn_perm(hh[3], ll[3], 0x07030602u);
        uint16_t* op = outp + base + u * 1024 + lane * 16;
        *reinterpret_cast<uint4*>(op) = a;
        *reinterpret_cast<uint4*>(op + 8) = b2;
      }
    }
    for (uint32_t i = base + lane * 4; i < npix; i += 256)  // tail: fewer than 4096 pixels
      emit4(i, *reinterpret_cast<const uint32_t*>(hp + i),
            no_low ? 0u : *reinterpret_cast<const uint32_t*>(lp + i),
            use_delta ? *reinterpret_cast<const uint32_t*>(dhigh + i) : 0u,
            use_delta ? *reinterpret_cast<const uint32_t*>(dlow + i) : 0u);
    return;
  }

  // Head: the first xs + 1 pixels are literals (cc:327 starts at xs + 1); pixels
  // xs+1 .. xs+3 are walked serially so that the supersteps start 4-aligned.
  const uint32_t head = min(xs + 4, npix);
  for (uint32_t o = lane * 4; o < head; o += 256)
    *reinterpret_cast<uint32_t*>(ring + (o & m)) = *reinterpret_cast<const uint32_t*>(hp + o);
  wave_lds_sync();
  {
    int w = ring[xs & m];
    for (uint32_t i = xs + 1; i < head; i++) {
      const int x = (ring[i & m] + cg_pred(ring[(i - xs) & m], w, ring[(i - xs - 1) & m])) & 0xff;
      ring[i & m] = (uint8_t)x;  // every lane writes the same value
      w = x;
    }
  }
  wave_lds_sync();
  for (uint32_t o = lane * 4; o < head; o += 256)
    emit4(o, *reinterpret_cast<const uint32_t*>(ring + (o & m)),
          no_low ? 0u : *reinterpret_cast<const uint32_t*>(lp + o),
          use_delta ? *reinterpret_cast<const uint32_t*>(dhigh + o) : 0u,
          use_delta ? *reinterpret_cast<const uint32_t*>(dlow + o) : 0u);
  uint32_t w_prev = (uint32_t)ring[(head - 1) & m] << 8;  // wave-uniform: final x[s-1] << 8

  // Every global access of the superstep loop is unconditional (clamped index,
  // planes that do not exist replaced by a readable dummy and masked to zero), so
  // that the loads stay in flight behind counted vmcnt waits instead of being
  // drained at a branch.  Only the last, partial superstep predicates its stores.
  const uint8_t* lsrc = no_low ? hp : lp;
  const uint8_t* dhsrc = use_delta ? dhigh : hp;
  const uint8_t* dlsrc = use_delta ? dlow : hp;
  const uint32_t lmask = no_low ? 0u : ~0u, dmask = use_delta ? ~0u : 0u;
  const uint32_t last4 = npix - 4;  // npix % 4 == 0 and npix >= head >= 4

  // residual dwords of this lane's segment, fetched one superstep ahead
  uint32_t rq[K4];
  auto fetch_r = [&](uint32_t s0, uint32_t* q) {
#pragma unroll
    for (int t = 0; t < K4; t++) {
      const uint32_t i = s0 + lane * K + 4 * t;
      const uint32_t v = *reinterpret_cast<const uint32_t*>(hp + min(i, last4));
      q[t] = i < npix ? v : 0u;
    }
  };
  fetch_r(head, rq);
  uint32_t npass = 0, maxpass = 0;

  auto superstep = [&](uint32_t s0, auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    const uint32_t start = s0 + lane * K;
    const bool active = FULL || start < npix;
    // operands of the output phase (coalesced, 4 pixels per lane and round): in
    // flight while the chain passes run
    uint32_t l4[K4], dh4[K4], dl4[K4];
#pragma unroll
    for (int t = 0; t < K4; t++) {
      const uint32_t i = s0 + (t * 64 + lane) * 4;
      const uint32_t ic = FULL ? i : min(i, last4);
      l4[t] = *reinterpret_cast<const uint32_t*>(lsrc + ic) & lmask;
      dh4[t] = *reinterpret_cast<const uint32_t*>(dhsrc + ic) & dmask;
      dl4[t] = *reinterpret_cast<const uint32_t*>(dlsrc + ic) & dmask;
    }
    // segment operands into registers, everything as v << 8
    uint32_t N[K], D[K], R[K], X[K];
    uint32_t guess;
    {
      const uint32_t nb = start - xs - 4;  // dword holding the north-west of the first pixel
      uint32_t nd[K4 + 1];
#pragma unroll
      for (int t = 0; t <= K4; t++) nd[t] = *reinterpret_cast<const uint32_t*>(ring + ((nb + 4 * t) & m));
      uint32_t prev = __builtin_amdgcn_perm(0, nd[0], 0x0c0c030cu);
      guess = prev;  // x[start - 1 - xs] << 8
#pragma unroll
      for (int t = 0; t < K4; t++) {
#pragma unroll
        for (int b = 0; b < 4; b++) {
          const uint32_t sel = 0x0c0c000cu | ((uint32_t)b << 8);
          const uint32_t n = __builtin_amdgcn_perm(0, nd[t + 1], sel);
          N[4 * t + b] = n;
          D[4 * t + b] = n - prev;
          R[4 * t + b] = __builtin_amdgcn_perm(0, rq[t], sel);
          prev = n;
        }
      }
    }
    fetch_r(s0 + S, rq);  // next superstep's residuals (rq was consumed above)

    // Passes: the first one runs every lane from its guess (lane 0 from the known
    // value), the following ones only the lanes whose input turned out wrong.  ONE
    // copy of the chain code serves all passes, so X[] stays in fixed registers.
    //
    // Long correction chains come from segments whose north pixels are all equal
    // (d == 0 everywhere): there the predictor is the identity, the segment maps
    // its input w to w + sum(r) for EVERY w, and a wrong guess is never forgotten.
    // When a superstep is still wrong after kScanAfter passes, such "shift" lanes
    // are detected and the inputs of whole runs of them are proposed at once by a
    // prefix scan over the lanes (shift lanes add their sum, other lanes contribute
    // their current output).  A proposal is only ever a proposal: the loop still
    // ends on the same check (every input equals the previous lane's actual
    // output), so exactness does not depend on it.
    uint32_t in = lane == 0 ? w_prev : guess;
    bool run = true;
    const uint32_t npass0 = npass;
    bool settled = false;
    for (uint32_t round = 0; round < kScanAfter; round++) {  // the common case ends in here
      if (run) {
        uint32_t w = in;
#pragma unroll
        for (int k = 0; k < K; k++) X[k] = w = inv_px(N[k], w, D[k], R[k]);
      }
      npass++;
      uint32_t pred = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)X[K - 1], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
      if (lane == 0) pred = w_prev;
      run = active && pred != in;
      if (!__builtin_amdgcn_ballot_w64(run)) { settled = true; break; }
      in = pred;  // lanes that do not rerun have pred == in already
    }
    if (__builtin_expect(!settled, 0)) {
      // Still wrong after kScanAfter passes: a long correction chain.  Its own copy
      // of the pass loop, so that nothing of it burdens the loop above.
      uint32_t all, ord = 0, sum = 0;
      asm volatile("v_mov_b32 %0, -1" : "=v"(all));  // keeps the compiler from hoisting the 2K ops below
#pragma unroll
      for (int k = 0; k < K; k++) { ord = max(ord, D[k] & all); sum += R[k] & all; }
      const bool is_shift = ord == 0 && lane != 0;  // lane 0 is final after the first pass: a constant
      const uint32_t ssum = sum & 0xffffu;
      for (;;) {
        if (run) {
          uint32_t w = in;
#pragma unroll
          for (int k = 0; k < K; k++) X[k] = w = inv_px(N[k], w, D[k], R[k]);
        }
        npass++;
        uint32_t pred = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)X[K - 1], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        if (lane == 0) pred = w_prev;
        const bool bad = active && pred != in;
        if (!__builtin_amdgcn_ballot_w64(bad)) break;
        uint32_t next_in = pred;
        if (npass - npass0 < kScanUntil) {  // wave-uniform
          // inclusive scan of the lanes' functions: CONST(c) or SHIFT(s); lane 0 is CONST
          uint32_t isc = is_shift ? 0u : 1u, v = is_shift ? ssum : X[K - 1];
#pragma unroll
          for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const uint32_t isc_l = __shfl_up(isc, dlt, 64), v_l = __shfl_up(v, dlt, 64);
            if (lane >= (uint32_t)dlt && !isc) {
              v = (v + v_l) & 0xffffu;
              isc = isc_l;
            }
          }
          next_in = __shfl_up(v, 1, 64);  // the consistent output of lane j-1
          if (lane == 0) next_in = w_prev;
        }
        run = active && next_in != in;  // bad somewhere implies run somewhere
        in = next_in;
      }
    }
    maxpass = max(maxpass, npass - npass0);
    w_prev = (uint32_t)__builtin_amdgcn_readlane((int)X[K - 1], 63);

    // reconstructed bytes into the ring (north of the next row, source of the output phase)
#pragma unroll
    for (int t = 0; t < K4; t++) {
      const uint32_t o = __builtin_amdgcn_perm(X[4 * t + 1], X[4 * t], 0x0c0c0501u) |
                         __builtin_amdgcn_perm(X[4 * t + 3], X[4 * t + 2], 0x05010c0cu);
      *reinterpret_cast<uint32_t*>(ring + ((start + 4 * t) & m)) = o;
    }
    wave_lds_sync();
    uint32_t h4[K4];
#pragma unroll
    for (int t = 0; t < K4; t++)
      h4[t] = *reinterpret_cast<const uint32_t*>(ring + ((s0 + (t * 64 + lane) * 4) & m));
#pragma unroll
    for (int t = 0; t < K4; t++) {  // delta^-1 + recombine (cc:335-344); absent planes are zeros
      const uint32_t i = s0 + (t * 64 + lane) * 4;
      const uint32_t hh = add4(h4[t], dh4[t]), ll = add4(l4[t], dl4[t]);
      uint2 o;
      o.x = __builtin_amdgcn_perm(hh, ll, 0x05010400u);
      o.y = __builtin_amdgcn_perm(hh, ll, 0x07030602u);
      if (FULL || i < npix) *reinterpret_cast<uint2*>(outp + i) = o;
    }
    wave_lds_sync();
  };

  uint32_t s = head;
  for (; s + S <= npix; s += S) superstep(s, std::true_type());
  if (s < npix) superstep(s, std::false_type());
  if (pass_count && lane == 0) {  // diagnostics: supersteps and passes (bench / tests)
    atomicAdd(&pass_count[0], (npix - head + S - 1) / S);
    atomicAdd(&pass_count[1], npass);
    atomicMax(&pass_count[2], npass);      // the slowest frame bounds a launch with one wave per SIMD
    atomicMax(&pass_count[3], maxpass);    // longest verification chain seen in one superstep
  }
}


// k_inverse_trio: the speculative segmented inverse (see k_inverse_spec) with the
// work of a frame split over THREE waves of one workgroup, by role:
//
//   wave 0 (chain)  north bytes + residuals from LDS -> operands -> passes ->
//                   reconstructed bytes into the LDS ring.  Issues no global
//                   memory instruction inside the superstep loop: its instruction
//                   stream is the frame's critical path and nothing else.
//   wave 1, wave 2  (i/o, even / odd supersteps) move every byte the frame needs
//                   from HBM straight into LDS (global_load_lds: no registers, so
//                   nothing of a load has to be touched before it is used), two to
//                   three supersteps ahead: residuals for the chain wave, low plane
//                   and delta planes for themselves; and they run the output phase of
//                   the superstep the chain wave finished in the previous iteration:
//                   ring + staged planes -> delta^-1 + recombine -> coalesced stores.
//
// One workgroup barrier per superstep hands ring(t) to the i/o waves and the
// residuals of t+1 to the chain wave.  The barrier is a bare s_barrier behind an
// LDS-only wait, so the i/o waves' loads stay in flight across it; an i/o wave
// waits for its own loads (vmcnt) only when it is active again, two iterations
// after issuing them.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// 64 lanes x 4 bytes from per-lane global addresses into LDS at dst + 4 * lane.
__device__ __forceinline__ void dma4(const uint8_t* src, uint8_t* dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t*>(src),
                                   reinterpret_cast<uint32_t*>(dst_wave_uniform), 4, 0, 0);
}

template <int K4>
__global__ __launch_bounds__(192) void k_inverse_trio(
    const uint8_t* __restrict__ high, const uint8_t* __restrict__ low,
    const uint8_t* __restrict__ dhigh, const uint8_t* __restrict__ dlow,
    const uint8_t* __restrict__ flags, uint16_t* __restrict__ frames, Geom g,
    uint32_t* __restrict__ pass_count) {
  constexpr int K = 4 * K4;
  constexpr uint32_t S = 64u * K;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* ring = smem;                 // reconstructed high bytes, flat index & m
  uint8_t* rbuf = smem + g.spec_ring;   // residual bytes of superstep t at rbuf + (t & 3) * S
  uint8_t* lbuf = rbuf + 4 * S;         // low / delta-high / delta-low bytes of superstep t at lbuf + ((t & 1) * 3 + plane) * S
  const uint32_t role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t frame = blockIdx.x, lane = threadIdx.x & 63;
  const size_t fbase = (size_t)frame * g.npix;
  const uint8_t* hp = high + fbase;
  const uint8_t* lp = low ? low + fbase : nullptr;
  uint16_t* outp = frames + fbase;
  const uint32_t f = flags[frame];
  const bool use_delta = (f & 1) && dhigh, use_cg = f & 2, no_low = (f & 4) || !lp;
  const uint32_t xs = g.xs, npix = g.npix, m = g.spec_ring - 1;

  // delta^-1 + recombine (cc:335-344) of 4 pixels; planes that do not exist come as zeros
  auto emit4 = [&](uint32_t i, uint32_t h4, uint32_t l4, uint32_t dh4, uint32_t dl4) {
    const uint32_t hh = add4(h4, dh4), ll = add4(l4, dl4);
    uint2 o;
    o.x = __builtin_amdgcn_perm(hh, ll, 0x05010400u);
    o.y = __builtin_amdgcn_perm(hh, ll, 0x07030602u);
    *reinterpret_cast<uint2*>(outp + i) = o;
  };
  const uint8_t* lsrc = no_low ? hp : lp;  // readable dummies for planes that do not exist
  const uint8_t* dhsrc = use_delta ? dhigh : hp;
  const uint8_t* dlsrc = use_delta ? dlow : hp;
  const uint32_t lmask = no_low ? 0u : ~0u, dmask = use_delta ? ~0u : 0u;
  const uint32_t last4 = npix - 4;  // npix % 4 == 0, npix >= 4

  if (!use_cg) {  // no chain: the three waves stream the frame, 4096 pixels per wave and turn
    constexpr int U = 4;
    uint32_t base = role * (U * 1024);
    for (; base + U * 1024 <= npix; base += 3 * U * 1024) {
      uint4 h[U], l[U], dh[U], dl[U];
#pragma unroll
      for (int u = 0; u < U; u++) h[u] = *reinterpret_cast<const uint4*>(hp + base + u * 1024 + lane * 16);
      if (!no_low) {
#pragma unroll
        for (int u = 0; u < U; u++) l[u] = *reinterpret_cast<const uint4*>(lp + base + u * 1024 + lane * 16);
      } else {
#pragma unroll
        for (int u = 0; u < U; u++) l[u] = make_uint4(0, 0, 0, 0);
      }
      if (use_delta) {
#pragma unroll
        for (int u = 0; u < U; u++) {
          dh[u] = *reinterpret_cast<const uint4*>(dhigh + base + u * 1024 + lane * 16);
          dl[u] = *reinterpret_cast<const uint4*>(dlow + base + u * 1024 + lane * 16);
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; u++) dh[u] = dl[u] = make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t hh[4] = {add4(h[u].x, dh[u].x), add4(h[u].y, dh[u].y), add4(h[u].z, dh[u].z), add4(h[u].w, dh[u].w)};
        const uint32_t ll[4] = {add4(l[u].x, dl[u].x), add4(l[u].y, dl[u].y), add4(l[u].z, dl[u].z), add4(l[u].w, dl[u].w)};
        uint4 a, b2;
        a.x = __builtin_amdgcn_perm(hh[0], ll[0], 0x05010400u); a.y = __builtin_amdgcn_perm(hh[0], ll[0], 0x07030602u);
        a.z = __builtin_amdgcn_perm(hh[1], ll[1], 0x05010400u); a.w = __builtin_amdgcn_perm(hh[1], ll[1], 0x07030602u);
        b2.x = __builtin_amdgcn_perm(hh[2], ll[2], 0x05010400u); b2.y = __builtin_amdgcn_perm(hh[2], ll[2], 0x07030602u);
        b2.z = __builtin_amdgcn_perm(hh[3], ll[3], 0x05010400u); b2.w = __builtin_amdgcn_perm(hh[3], ll[3], 0x07030602u);
        uint16_t* op = outp + base + u * 1024 + lane * 16;
        *reinterpret_cast<uint4*>(op) = a;
        *reinterpret_cast<uint4*>(op + 8) = b2;
      }
    }
    // tail (fewer than 3 x 4096 pixels): the first pixel no wave has covered, split by thread
    const uint32_t done = (npix / (U * 1024)) * (U * 1024);
    for (uint32_t i = done + threadIdx.x * 4; i < npix; i += 192 * 4)
      emit4(i, *reinterpret_cast<const uint32_t*>(hp + i),
            *reinterpret_cast<const uint32_t*>(lsrc + i) & lmask,
            *reinterpret_cast<const uint32_t*>(dhsrc + i) & dmask,
            *reinterpret_cast<const uint32_t*>(dlsrc + i) & dmask);
    return;
  }

  const uint32_t head = min(xs + 4, npix);
  const uint32_t nsteps = (npix - head + S - 1) / S;
  uint32_t w_prev = 0, npass = 0, maxpass = 0;  // chain wave

  // i/o waves: stage superstep t's residuals / planes into LDS (coalesced layout:
  // dword k * 64 + lane).  Indices are clamped into the frame; what lies beyond the
  // last pixel is never emitted and never anybody's north.
  auto dma_r = [&](uint32_t t) {
#pragma unroll
    for (int k = 0; k < K4; k++)
      dma4(hp + min(head + t * S + (k * 64 + lane) * 4, last4), rbuf + (t & 3) * S + k * 256);
  };
  auto dma_ldh = [&](uint32_t t) {
    uint8_t* base = lbuf + (t & 1) * 3 * S;
#pragma unroll
    for (int k = 0; k < K4; k++) {
      const uint32_t ic = min(head + t * S + (k * 64 + lane) * 4, last4);
      dma4(lsrc + ic, base + k * 256);
      dma4(dhsrc + ic, base + S + k * 256);
      dma4(dlsrc + ic, base + 2 * S + k * 256);
    }
  };

  if (role == 0) {
    // Head: the first xs + 1 pixels are literals (cc:327 starts at xs + 1); pixels
    // xs+1 .. xs+3 are walked serially so that the supersteps start 4-aligned.
    for (uint32_t o = lane * 4; o < head; o += 256)
      *reinterpret_cast<uint32_t*>(ring + (o & m)) = *reinterpret_cast<const uint32_t*>(hp + o);
    wave_lds_sync();
    {
      int w = ring[xs & m];
      for (uint32_t i = xs + 1; i < head; i++) {
        const int x = (ring[i & m] + cg_pred(ring[(i - xs) & m], w, ring[(i - xs - 1) & m])) & 0xff;
        ring[i & m] = (uint8_t)x;  // every lane writes the same value
        w = x;
      }
    }
    wave_lds_sync();
    for (uint32_t o = lane * 4; o < head; o += 256)
      emit4(o, *reinterpret_cast<const uint32_t*>(ring + (o & m)),
            *reinterpret_cast<const uint32_t*>(lsrc + o) & lmask,
            *reinterpret_cast<const uint32_t*>(dhsrc + o) & dmask,
            *reinterpret_cast<const uint32_t*>(dlsrc + o) & dmask);
    w_prev = (uint32_t)ring[(head - 1) & m] << 8;  // wave-uniform: final x[s-1] << 8
  } else if (role == 1) {  // even supersteps: residuals of 0 and 2, planes of 0
    dma_r(0);
    dma_r(2);
    dma_ldh(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {  // odd supersteps: residuals of 1 (3 and the planes of 1 follow in iteration 0)
    dma_r(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  lds_barrier();

  for (uint32_t t = 0; t <= nsteps; t++) {
    if (role == 0) {
      if (t < nsteps) {
        const uint32_t s0 = head + t * S, start = s0 + lane * K;
        const bool active = start < npix;
        uint32_t N[K], D[K], R[K], X[K];
        uint32_t guess;
        {
          const uint32_t nb = start - xs - 4;  // dword holding the north-west of the first pixel
          uint32_t nd[K4 + 1], rq[K4];
#pragma unroll
          for (int q = 0; q <= K4; q++) nd[q] = *reinterpret_cast<const uint32_t*>(ring + ((nb + 4 * q) & m));
#pragma unroll
          for (int q = 0; q < K4; q++) rq[q] = *reinterpret_cast<const uint32_t*>(rbuf + (t & 3) * S + lane * K + 4 * q);
          uint32_t prev = __builtin_amdgcn_perm(0, nd[0], 0x0c0c030cu);
          guess = prev;  // x[start - 1 - xs] << 8
#pragma unroll
          for (int q = 0; q < K4; q++) {
#pragma unroll
            for (int b = 0; b < 4; b++) {
              const uint32_t sel = 0x0c0c000cu | ((uint32_t)b << 8);
              const uint32_t n = __builtin_amdgcn_perm(0, nd[q + 1], sel);
              N[4 * q + b] = n;
              D[4 * q + b] = n - prev;
              R[4 * q + b] = __builtin_amdgcn_perm(0, rq[q], sel);
              prev = n;
            }
          }
        }
        uint32_t in = lane == 0 ? w_prev : guess;
        bool run = true;
        const uint32_t npass0 = npass;
        bool settled = false;
        for (uint32_t round = 0; round < kScanAfter; round++) {  // the common case ends in here
          if (run) {
            uint32_t w = in;
#pragma unroll
            for (int k = 0; k < K; k++) X[k] = w = inv_px(N[k], w, D[k], R[k]);
          }
          npass++;
          uint32_t pred = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)X[K - 1], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
          if (lane == 0) pred = w_prev;
          run = active && pred != in;
          if (!__builtin_amdgcn_ballot_w64(run)) { settled = true; break; }
          in = pred;
        }
        if (__builtin_expect(!settled, 0)) {  // long correction chain: see k_inverse_spec
          uint32_t all, ord = 0, sum = 0;
          asm volatile("v_mov_b32 %0, -1" : "=v"(all));
#pragma unroll
          for (int k = 0; k < K; k++) { ord = max(ord, D[k] & all); sum += R[k] & all; }
          const bool is_shift = ord == 0 && lane != 0;
          const uint32_t ssum = sum & 0xffffu;
          for (;;) {
            if (run) {
              uint32_t w = in;
#pragma unroll
              for (int k = 0; k < K; k++) X[k] = w = inv_px(N[k], w, D[k], R[k]);
            }
            npass++;
            uint32_t pred = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)X[K - 1], 0x138, 0xf, 0xf, false);
            if (lane == 0) pred = w_prev;
            const bool bad = active && pred != in;
            if (!__builtin_amdgcn_ballot_w64(bad)) break;
            uint32_t next_in = pred;
            if (npass - npass0 < kScanUntil) {
              uint32_t isc = is_shift ? 0u : 1u, v = is_shift ? ssum : X[K - 1];
#pragma unroll
              for (int dlt = 1; dlt < 64; dlt <<= 1) {
                const uint32_t isc_l = __shfl_up(isc, dlt, 64), v_l = __shfl_up(v, dlt, 64);
                if (lane >= (uint32_t)dlt && !isc) {
                  v = (v + v_l) & 0xffffu;
                  isc = isc_l;
                }
              }
              next_in = __shfl_up(v, 1, 64);
              if (lane == 0) next_in = w_prev;
            }
            run = active && next_in != in;
            in = next_in;
          }
        }
        maxpass = max(maxpass, npass - npass0);
        w_prev = (uint32_t)__builtin_amdgcn_readlane((int)X[K - 1], 63);
#pragma unroll
        for (int q = 0; q < K4; q++) {
          const uint32_t o = __builtin_amdgcn_perm(X[4 * q + 1], X[4 * q], 0x0c0c0501u) |
                             __builtin_amdgcn_perm(X[4 * q + 3], X[4 * q + 2], 0x05010c0cu);
          *reinterpret_cast<uint32_t*>(ring + ((start + 4 * q) & m)) = o;
        }
      }
    } else if (((t + 1) & 1u) == role - 1) {  // i/o wave of the parity of superstep t - 1 (and t + 1, t + 3)
      // everything this wave issued two iterations ago has landed: planes of t - 1, residuals of t + 1
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (t >= 1) {  // output phase of superstep t - 1
        const uint32_t s0 = head + (t - 1) * S;
        const uint8_t* base = lbuf + ((t - 1) & 1) * 3 * S;
        uint32_t h4[K4], l4[K4], dh4[K4], dl4[K4];
#pragma unroll
        for (int k = 0; k < K4; k++) {
          const uint32_t o = (k * 64 + lane) * 4;
          h4[k] = *reinterpret_cast<const uint32_t*>(ring + ((s0 + o) & m));
          l4[k] = *reinterpret_cast<const uint32_t*>(base + o) & lmask;
          dh4[k] = *reinterpret_cast<const uint32_t*>(base + S + o) & dmask;
          dl4[k] = *reinterpret_cast<const uint32_t*>(base + 2 * S + o) & dmask;
        }
#pragma unroll
        for (int k = 0; k < K4; k++) {
          const uint32_t i = s0 + (k * 64 + lane) * 4;
          if (i < npix) emit4(i, h4[k], l4[k], dh4[k], dl4[k]);
        }
        wave_lds_sync();  // the reads above are done before the next planes overwrite the buffer
      }
      if (t + 1 < nsteps) dma_ldh(t + 1);
      if (t + 3 < nsteps) dma_r(t + 3);
    }
    lds_barrier();
  }
  if (pass_count && role == 0 && lane == 0) {  // diagnostics: supersteps and passes (bench / tests)
    atomicAdd(&pass_count[0], nsteps);
    atomicAdd(&pass_count[1], npass);
    atomicMax(&pass_count[2], npass);
    atomicMax(&pass_count[3], maxpass);
  }
}

}  // namespace

// ------------------------------------------------------------- host side ---

static uint32_t round_up(uint32_t v, uint32_t a) { return (v + a - 1) / a * a; }

// Function attributes (dynamic LDS above 64 KiB) are per device: runs `set` once
// for the calling thread's current device, under a lock, so that in-process
// multi-GPU sessions and concurrent first launches are safe.
template <typename F>
static void once_per_device(std::mutex& m, uint64_t& done_mask, F set) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const uint64_t bit = 1ull << (dev & 63);
  std::lock_guard<std::mutex> l(m);
  if (done_mask & bit) return;
  set();
  done_mask |= bit;
}

// Speculative inverse: the largest supported segment length K = 4 * k4 with a
// superstep of 64 * K pixels that fits into one image row (its north pixels must
// lie in front of it).
static void spec_geom(Geom* g) {
  g->spec_k4 = 0;
  g->spec_ring = 0;
  if ((g->xs & 3) || (g->npix & 3) || g->xs < 256) return;
  uint32_t k4 = std::min(kSpecMaxK4, g->xs / 256);
  if (const char* e = getenv("FPV_SPEC_K4")) k4 = std::max(1u, std::min(k4, (uint32_t)atoi(e)));
  uint32_t r = 1;
  while (r < g->xs + 256 * k4 + 16) r <<= 1;
  // ring + the three-wave kernel's staging buffers must fit the default 64 KiB of dynamic
  // LDS (frames wider than ~45 K pixels: the serial-chain kernels take over)
  if (r + 10 * 256 * k4 > 64 * 1024) return;
  g->spec_k4 = k4;
  g->spec_ring = r;
}

bool make_geom(uint32_t xs, uint32_t ys, int shift, int big_endian, Geom* g,
               const char** why) {
  static const char* kDims = "xsize and ysize must be non-zero multiples of 4";
  static const char* kShift = "shift must be 0..8 (0..15 for little-endian input)";
  static const char* kWide = "xsize too large for the LDS line buffer";
  static const char* kBig = "frame larger than 2^31 pixels";
  if (!xs || !ys || (xs & 3) || (ys & 3)) { *why = kDims; return false; }
  if (shift < 0 || shift > 15 || (big_endian && shift > 8)) { *why = kShift; return false; }
  if ((uint64_t)xs * ys >= (1ull << 31)) { *why = kBig; return false; }
  g->xs = xs; g->ys = ys; g->npix = xs * ys;
  g->pw = xs / 4; g->ph = ys / 4; g->npv = g->pw * g->ph;
  g->shift = shift; g->swap = big_endian ? 1 : 0;
  // Forward tile: whole preview rows (multiples of 4 image rows), at most
  // kMaxIter 16-pixel segments per thread, and as few idle threads in the last
  // pass over the tile as possible (nseg close to a multiple of 256).
  uint32_t rows = 4;
  {
    const uint32_t max_px = kMaxIter * kThreads * 16;
    double best = -1.0;
    for (uint32_t r = 4; r <= ys && (uint64_t)r * xs <= max_px; r += 4) {
      const uint32_t nseg = r * xs / 16, passes = (nseg + kThreads - 1) / kThreads;
      const double eff = (double)nseg / (passes * kThreads) + 1e-4 * r;  // ties: larger tile
      if (eff > best) { best = eff; rows = r; }
    }
  }
  if (const char* e = getenv("FPV_TILE_ROWS")) rows = std::max(4u, (uint32_t)atoi(e) & ~3u);
  if (rows > ys) rows = ys;
  g->tile_rows = rows;
  g->tile_px = rows * xs;
  g->tiles = (ys + rows - 1) / rows;
  if (g->tile_px > kMaxIter * kThreads * 16) { *why = kWide; return false; }
  g->lds_off = round_up(xs + 16, 16);
  g->lds_h_bytes = round_up(g->lds_off + g->tile_px, 16);
  g->lds_hist_off = round_up(g->lds_h_bytes + g->tile_px / 2, 16);
  g->lds_bytes = g->lds_hist_off + kHistFused * 256 * 4 + 16;
  if (g->lds_bytes > 160 * 1024 - 64) { *why = kWide; return false; }
  // streaming kernel: strips of about 100K pixels (whole preview rows), one wave each
  {
    // Prefer a strip height that divides ys and whose pixel count is a multiple
    // of the 1024-pixel step (the FAST kernel needs both); about 100K pixels
    // (measured on 1280x800: 80..160 rows beat 32: fewer halo rows and partials).
    const double kTarget = 102400.0;
    uint32_t srows = (((uint32_t)kTarget + xs - 1) / xs + 3) & ~3u;
    {
      uint32_t best = 0;
      double best_cost = 1e30;
      for (uint32_t r = 4; r <= ys; r += 4) {
        if (ys % r || ((uint64_t)r * xs) % kStep) continue;
        const double px = (double)r * xs;
        const double cost = px < kTarget ? kTarget / px : px / kTarget;  // distance from the target
        if (cost < best_cost) { best_cost = cost; best = r; }
      }
      if (best && best_cost < 4.0) srows = best;
    }
    if (const char* e = getenv("FPV_STRIP_ROWS")) srows = std::max(4u, (uint32_t)atoi(e) & ~3u);
    if (srows > ys) srows = ys;
    while ((uint64_t)srows * xs > 15u * 60000u && srows > 4) srows -= 4;  // 16-bit histogram bins
    g->strip_rows = srows;
    g->strips = (ys + srows - 1) / srows;
    g->ring_base = round_up(xs + 16, 16);
    uint32_t rf = 1;
    while (rf < g->ring_base + kStep) rf <<= 1;
    g->ring_fwd = rf;
    g->nacc = (kStep + 4 * xs - 1) / (4 * xs) + 1;
    g->wave_lds = round_up(rf + g->nacc * g->pw * 4 + kHistFused * 128 * 4, 16);
    if (g->wave_lds > 64 * 1024) { *why = kWide; return false; }
    uint32_t wpc = (160 * 1024 - 1024) / g->wave_lds;
    g->stream_waves_per_cu = std::max(1u, std::min(8u, wpc));  // measured: 8 beats 12 beats 16 (memory-system contention)
    if (const char* e = getenv("FPV_WAVES_PER_CU")) g->stream_waves_per_cu = std::max(1, atoi(e));
  }
  g->nd = (g->npix + 14) / 15;
  g->ncg = g->npix > xs + 1 ? (g->npix - xs - 1 + 30) / 31 : 0;
  uint32_t samples = g->nd + g->ncg;
  g->stat_blocks = (samples + kThreads * 8 - 1) / (kThreads * 8);
  if (g->stat_blocks < 1) g->stat_blocks = 1;
  uint32_t ring = 1;
  while (ring < xs + 1 + kInvChunk + 4) ring <<= 1;
  g->ring = ring;
  { uint32_t ir = 1; while (ir < xs + 8 + kInvC) ir <<= 1; g->inv_ring = ir; }
  spec_geom(g);
  g->fwd_ok = 1;
  return true;
}

bool make_geom_decoder(uint32_t xs, uint32_t ys, Geom* g, const char** why) {
  static const char* kDims = "xsize and ysize must be non-zero";
  static const char* kBig = "frame larger than 2^31 pixels";
  static const char* kWide = "xsize too large for the LDS ring";
  if (!xs || !ys) { *why = kDims; return false; }
  if ((uint64_t)xs * ys >= (1ull << 31)) { *why = kBig; return false; }
  *g = Geom();
  g->xs = xs; g->ys = ys; g->npix = xs * ys;
  g->pw = xs / 4; g->ph = ys / 4; g->npv = g->pw * g->ph;
  uint32_t ring = 1;
  while (ring < xs + 1 + kInvChunk + 4) ring <<= 1;
  if (ring > 128 * 1024) { *why = kWide; return false; }
  g->ring = ring;
  { uint32_t ir = 1; while (ir < xs + 8 + kInvC) ir <<= 1; g->inv_ring = ir; }
  spec_geom(g);
  g->fwd_ok = 0;
  return true;
}

// Workspace layout: [counters | anylow] are zeroed by ONE memset per batch; the
// rest is fully written by the kernels.  `spec` survives batches.
size_t fwd_workspace_bytes(const Geom& g, uint32_t max_frames) {
  size_t b = 256 + round_up(max_frames * 4, 256);                // counters, anylow
  b += round_up(max_frames, 256);                                // decision
  b += 2 * (size_t)round_up(max_frames * 4, 256);                // list A, list B
  b += round_up((size_t)max_frames * g.npv, 256);                // pv_raw
  b += (size_t)max_frames * g.strips * kHistFused * 256 * 2;  // partial histograms (u16)
  return b + 256 /* spec */ + 256;
}

void carve_fwd_workspace(const Geom& g, uint32_t max_frames, void* base,
                         FwdWorkspace* ws) {
  uint8_t* p = static_cast<uint8_t*>(base);
  ws->counters = reinterpret_cast<uint32_t*>(p);
  p += 256;
  ws->anylow = reinterpret_cast<uint32_t*>(p);
  p += round_up(max_frames * 4, 256);
  ws->zero_fixed = (size_t)(p - static_cast<uint8_t*>(base));
  ws->decision = p;
  p += round_up(max_frames, 256);
  ws->list_a = reinterpret_cast<uint32_t*>(p);
  p += round_up(max_frames * 4, 256);
  ws->list_b = reinterpret_cast<uint32_t*>(p);
  p += round_up(max_frames * 4, 256);
  ws->pv_raw = p;
  p += round_up((size_t)max_frames * g.npv, 256);
  ws->partials = reinterpret_cast<uint16_t*>(p);
  p += (size_t)max_frames * g.strips * kHistFused * 256 * 2;
  ws->spec = p;
}

hipError_t init_fwd_workspace(const FwdWorkspace& ws, hipStream_t s) {
  return hipMemsetAsync(ws.spec, 3, 4, s);  // first guess: delta + CG
}

hipError_t launch_copy(const void* src, void* dst, size_t bytes, hipStream_t s) {
  k_copy<<<256 * 32, kThreads, 0, s>>>(static_cast<const u32x4*>(src), static_cast<u32x4*>(dst),
                                       bytes / 16);
  return hipGetLastError();
}

hipError_t launch_split(const Geom& g, const uint16_t* d_img, uint8_t* d_high,
                        uint8_t* d_low, uint32_t* d_anylow, hipStream_t s) {
  dim3 grid((g.npix / 4 + kThreads - 1) / kThreads);
  if (g.swap)
    k_split<true><<<grid, kThreads, 0, s>>>(d_img, d_high, d_low, g.npix, g.shift, d_anylow);
  else
    k_split<false><<<grid, kThreads, 0, s>>>(d_img, d_high, d_low, g.npix, g.shift, d_anylow);
  return hipGetLastError();
}

hipError_t launch_interleave(const uint8_t* d_high, const uint8_t* d_low, uint16_t* d_img,
                             size_t npix_total, hipStream_t s) {
  k_interleave<<<(unsigned)((npix_total / 4 + kThreads - 1) / kThreads), kThreads, 0, s>>>(
      d_high, d_low, d_img, npix_total);
  return hipGetLastError();
}

hipError_t launch_split_planes(const uint16_t* d_img, uint8_t* d_high, uint8_t* d_low,
                               size_t npix_total, hipStream_t s) {
  k_split_decoded<<<(unsigned)((npix_total / 4 + kThreads - 1) / kThreads), kThreads, 0, s>>>(
      d_img, d_high, d_low, (uint32_t)npix_total);
  return hipGetLastError();
}

hipError_t launch_unextract(uint16_t* d_px, size_t npix_total, int shift, int big_endian, hipStream_t s) {
  if (npix_total == 0 || (shift == 0 && !big_endian)) return hipSuccess;  // already the raw little-endian values
  k_unextract<<<(unsigned)((npix_total / 8 + kThreads) / kThreads), kThreads, 0, s>>>(d_px, npix_total, shift,
                                                                                   big_endian ? 1 : 0);
  return hipGetLastError();
}

hipError_t launch_split_decoded(const Geom& g, const uint16_t* d_img,
                                uint8_t* d_high, uint8_t* d_low, hipStream_t s) {
  dim3 grid((g.npix / 4 + kThreads - 1) / kThreads);
  k_split_decoded<<<grid, kThreads, 0, s>>>(d_img, d_high, d_low, g.npix);
  return hipGetLastError();
}

template <bool SWAP>
static hipError_t predict_impl(const Geom& g, const uint16_t* d_frames,
                               uint32_t nframes, const uint8_t* d_dhigh,
                               const uint8_t* d_dlow, const FwdWorkspace& ws,
                               uint8_t* d_high, uint8_t* d_low,
                               uint8_t* d_preview, uint8_t* d_flags,
                               hipStream_t s, hipEvent_t* ev) {
  const int with_delta = d_dhigh != nullptr;
  hipError_t e;
  if ((e = hipMemsetAsync(ws.counters, 0, ws.zero_fixed, s)) !=
      hipSuccess)
    return e;
  if (ev) hipEventRecord(ev[0], s);
  {
    static int ncu = 0;
    if (!ncu) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
        ncu = prop.multiProcessorCount;
      if (ncu <= 0) ncu = 256;
    }
    const uint32_t ntasks = nframes * g.strips;
    const uint32_t grid = std::min<uint32_t>(ntasks, (uint32_t)ncu * g.stream_waves_per_cu);
    StreamArgs a;
    a.xs = g.xs; a.ys = g.ys; a.npix = g.npix; a.pw = g.pw; a.npv = g.npv;
    a.strip_rows = g.strip_rows; a.strips = g.strips; a.ring_fwd = g.ring_fwd;
    a.ring_base = g.ring_base; a.nacc = g.nacc; a.shift = (uint32_t)g.shift; a.nframes = nframes;
    static const int strip_major = getenv("FPV_STRIP_MAJOR") ? atoi(getenv("FPV_STRIP_MAJOR")) : 1;
    a.strip_major = (uint32_t)strip_major;
    // FAST needs every step of every strip to be full and a segment never to
    // straddle image rows.  Both delta variants are launched; the one that does
    // not match the speculation byte (device memory) exits at once.
    const bool fast = (g.xs & 15) == 0 && g.nacc == 2 && (g.strip_rows * g.xs) % kStep == 0 &&
                      g.ys % g.strip_rows == 0;
#define FPV_LAUNCH_STREAM(F, D, L)                                                            \
  k_stream<SWAP, F, D, L><<<grid, 64, g.wave_lds, s>>>(d_frames, d_dhigh, d_dlow, ws.spec,   \
                                                       d_high, d_low, ws.pv_raw, ws.anylow,  \
                                                       ws.partials, ws.counters, a)
    const bool lowp = g.shift != 8;
    if (fast) {
      if (lowp) { FPV_LAUNCH_STREAM(true, false, true); if (with_delta) FPV_LAUNCH_STREAM(true, true, true); }
      else { FPV_LAUNCH_STREAM(true, false, false); if (with_delta) FPV_LAUNCH_STREAM(true, true, false); }
    } else {
      if (lowp) { FPV_LAUNCH_STREAM(false, false, true); if (with_delta) FPV_LAUNCH_STREAM(false, true, true); }
      else { FPV_LAUNCH_STREAM(false, false, false); if (with_delta) FPV_LAUNCH_STREAM(false, true, false); }
    }
#undef FPV_LAUNCH_STREAM
  }
  if (ev) hipEventRecord(ev[1], s);
  k_decide_spec<<<(nframes + kThreads / 64 - 1) / (kThreads / 64), kThreads, 0, s>>>(
      ws.partials, ws.spec, ws.decision, ws.counters, ws.list_a, ws.list_b, g, with_delta, nframes);
  if (ev) hipEventRecord(ev[2], s);
  // exact redo of mis-speculated frames; both launches exit at once when the lists are empty
  k_redo_decide<SWAP><<<kListGrid / 4, kThreads, 0, s>>>(d_frames, d_dhigh, ws.decision, ws.counters,
                                                        ws.list_a, ws.list_b, g, with_delta);
  k_apply_list<SWAP><<<kListGrid, kThreads, g.lds_bytes, s>>>(
      d_frames, d_dhigh, d_dlow, ws.decision, ws.counters, ws.list_b, d_high, d_low, ws.pv_raw,
      ws.anylow, g);
  if (ev) hipEventRecord(ev[3], s);
  if ((g.pw & 15) == 0)
    k_finish<true><<<dim3((g.npv / 16 + kThreads - 1) / kThreads, nframes), kThreads, 0, s>>>(
        ws.pv_raw, ws.decision, ws.anylow, ws.counters, d_preview, d_flags, ws.spec, g);
  else
    k_finish<false><<<dim3((g.npv / 4 + kThreads) / kThreads, nframes), kThreads, 0, s>>>(
        ws.pv_raw, ws.decision, ws.anylow, ws.counters, d_preview, d_flags, ws.spec, g);
  if (ev) hipEventRecord(ev[4], s);
  return hipGetLastError();
}

hipError_t launch_predict(const Geom& g, const uint16_t* d_frames,
                          uint32_t nframes, const uint8_t* d_dhigh,
                          const uint8_t* d_dlow, const FwdWorkspace& ws,
                          uint8_t* d_high, uint8_t* d_low, uint8_t* d_preview,
                          uint8_t* d_flags, hipStream_t s, hipEvent_t* ev) {
  static std::mutex attr_m;
  static uint64_t attr_done = 0;
  once_per_device(attr_m, attr_done, [] {  // tiles above 64 KiB of LDS need the opt-in
#define FPV_SF(W, F, D, L) reinterpret_cast<const void*>(&k_stream<W, F, D, L>)
    const void* sf[16] = {
        FPV_SF(true, true, true, true),    FPV_SF(true, true, true, false),
        FPV_SF(true, true, false, true),   FPV_SF(true, true, false, false),
        FPV_SF(true, false, true, true),   FPV_SF(true, false, true, false),
        FPV_SF(true, false, false, true),  FPV_SF(true, false, false, false),
        FPV_SF(false, true, true, true),   FPV_SF(false, true, true, false),
        FPV_SF(false, true, false, true),  FPV_SF(false, true, false, false),
        FPV_SF(false, false, true, true),  FPV_SF(false, false, true, false),
        FPV_SF(false, false, false, true), FPV_SF(false, false, false, false)};
#undef FPV_SF
    for (const void* f : sf)
      hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    const void* fns[2] = {reinterpret_cast<const void*>(&k_apply_list<true>),
                          reinterpret_cast<const void*>(&k_apply_list<false>)};
    for (const void* f : fns)
      hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
  });
  if (g.swap)
    return predict_impl<true>(g, d_frames, nframes, d_dhigh, d_dlow, ws, d_high,
                              d_low, d_preview, d_flags, s, ev);
  return predict_impl<false>(g, d_frames, nframes, d_dhigh, d_dlow, ws, d_high,
                             d_low, d_preview, d_flags, s, ev);
}

hipError_t launch_inverse(const Geom& g, const uint8_t* d_high,
                          const uint8_t* d_low, const uint8_t* d_dhigh,
                          const uint8_t* d_dlow, const uint8_t* d_flags,
                          uint32_t nframes, uint16_t* d_frames, hipStream_t s,
                          uint32_t* d_diag) {
  static std::mutex attr_m;
  static uint64_t attr_done = 0;
  once_per_device(attr_m, attr_done, [] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_inverse<true>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_inverse<false>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_inverse_fast),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
  });
  // FPV_INVERSE = auto (default) | trio | spec | fast | slow selects the kernel (tests, A/B timing).
  // auto: three waves per frame (shorter serial path per frame) while every frame's
  // chain wave can have a SIMD to itself most of the time, one wave per frame (less
  // synchronisation, more frames resident) for larger batches.  Measured crossover
  // on MI355X (256 CUs): between 768 and 1000 frames.
  int mode = -1;  // read per launch: tests switch kernels inside one process
  if (getenv("FPV_INVERSE_SLOW")) {
    mode = 2;
  } else if (const char* e = getenv("FPV_INVERSE")) {
    mode = e[0] == 'f' ? 1 : e[0] == 's' && e[1] == 'l' ? 2 : e[0] == 's' ? 0 : e[0] == 't' ? 3 : -1;
  }
  if (mode < 0) {
    static int ncu = 0;
    if (!ncu) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
        ncu = prop.multiProcessorCount;
      if (ncu <= 0) ncu = 256;
    }
    mode = nframes <= 3u * (uint32_t)ncu ? 3 : 0;
  }
  const bool aligned = (g.xs & 3) == 0 && (g.npix & 3) == 0;
  if (mode == 3 && g.spec_k4) {
#define FPV_LAUNCH_TRIO(K4)                                                                     \
  case K4:                                                                                      \
    k_inverse_trio<K4><<<nframes, 192, g.spec_ring + 10 * 256 * K4, s>>>(                       \
        d_high, d_low, d_dhigh, d_dlow, d_flags, d_frames, g, d_diag);                          \
    break;
    switch (g.spec_k4) {
      FPV_LAUNCH_TRIO(1) FPV_LAUNCH_TRIO(2) FPV_LAUNCH_TRIO(3)
      FPV_LAUNCH_TRIO(4) FPV_LAUNCH_TRIO(5) FPV_LAUNCH_TRIO(6)
      default: return hipErrorInvalidValue;
    }
#undef FPV_LAUNCH_TRIO
  } else if (mode == 0 && g.spec_k4) {
#define FPV_LAUNCH_SPEC(K4)                                                                  \
  case K4:                                                                                   \
    k_inverse_spec<K4><<<nframes, 64, g.spec_ring, s>>>(d_high, d_low, d_dhigh, d_dlow,      \
                                                        d_flags, d_frames, g, d_diag);       \
    break;
    switch (g.spec_k4) {
      FPV_LAUNCH_SPEC(1) FPV_LAUNCH_SPEC(2) FPV_LAUNCH_SPEC(3)
      FPV_LAUNCH_SPEC(4) FPV_LAUNCH_SPEC(5) FPV_LAUNCH_SPEC(6)
      default: return hipErrorInvalidValue;
    }
#undef FPV_LAUNCH_SPEC
  } else if (mode != 2 && aligned && g.xs >= kInvC && (g.inv_ring + 4 * kInvC) * 4 <= 160 * 1024 - 64) {
    k_inverse_fast<<<nframes, 64, (g.inv_ring + 4 * kInvC) * 4, s>>>(d_high, d_low, d_dhigh, d_dlow,
                                                                     d_flags, d_frames, g);
  } else if (aligned) {
    k_inverse<true><<<nframes, 64, g.ring, s>>>(d_high, d_low, d_dhigh, d_dlow,
                                                d_flags, d_frames, g);
  } else {
    k_inverse<false><<<nframes, 64, g.ring, s>>>(d_high, d_low, d_dhigh, d_dlow,
                                                 d_flags, d_frames, g);
  }
  return hipGetLastError();
}

}  // namespace fpvhip
