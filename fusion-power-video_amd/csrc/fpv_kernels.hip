// fpv_kernels.hip -- gfx950 (CDNA4) kernels of the fusion-power-video
// predictor/residual stage.  Pure 8/16-bit integer filtering: no MFMA, the
// bound is HBM bandwidth (forward) or dependency latency (inverse).
//
// Reference loops replaced ("cc:N" = fusion_power_video.cc of the upstream
// CPU library):
//   k_stats            heuristic histograms                cc:526-531, cc:554-562
//   k_decide           EstimateEntropy + the two decisions cc:235-244, cc:533, cc:564
//   k_apply            split + preview sums + delta + CG   cc:391-443, cc:500-512,
//                                                          cc:534-537, cc:566-572
//   k_preview_finish   preview CG + flags byte             cc:575-586, cc:447-449
//   k_inverse          CG^-1, delta^-1, recombine          cc:326-344
//
// Neighbourhoods are FLAT 1-D (w = i-1, n = i-xs, nw = i-xs-1 on the flattened
// plane, cc:556-558): column 0's west neighbour is the previous row's last
// pixel.  The first xs+1 bytes of a predicted plane are literals (cc:572).
#include "fpv_kernels.h"

namespace fpvhip {

namespace {

constexpr int kThreads = 256;

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

// ------------------------------------------------------------- raw split ---

template <bool SWAP>
__global__ __launch_bounds__(kThreads) void k_split(
    const uint16_t* __restrict__ img, uint8_t* __restrict__ high,
    uint8_t* __restrict__ low, uint32_t npix, int S) {
  uint32_t i = (blockIdx.x * kThreads + threadIdx.x) * 4;
  if (i >= npix) return;  // npix is a multiple of 16
  const uint2 v = *reinterpret_cast<const uint2*>(img + i);
  uint32_t h4, l4;
  split4<SWAP>(v.x, v.y, S, &h4, &l4);
  if (S == 8) l4 = 0;
  *reinterpret_cast<uint32_t*>(high + i) = h4;
  *reinterpret_cast<uint32_t*>(low + i) = l4;
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

// ------------------------------------------------------- heuristic stats ---

// One thread per sample.  Histograms are exact (integer atomics; order does
// not matter).
template <bool SWAP>
__global__ __launch_bounds__(kThreads) void k_stats(
    const uint16_t* __restrict__ frames, const uint8_t* __restrict__ dhigh,
    uint32_t* __restrict__ hists, Geom g, int with_delta) {
  __shared__ uint32_t lh[kNumHists * 256];
  const uint32_t frame = blockIdx.y;
  const uint16_t* img = frames + (size_t)frame * g.npix;
  for (int b = threadIdx.x; b < kNumHists * 256; b += kThreads) lh[b] = 0;
  __syncthreads();

  const uint32_t stride = gridDim.x * kThreads;
  const uint32_t t0 = blockIdx.x * kThreads + threadIdx.x;
  uint32_t h, l;
  if (with_delta) {  // cc:526-531: counta of high[0::15]
    for (uint32_t k = t0; k < g.nd; k += stride) {
      split1<SWAP>(img[15u * k], g.shift, &h, &l);
      atomicAdd(&lh[h], 1u);
    }
  }
  for (uint32_t k = t0; k < g.ncg; k += stride) {  // cc:554-562
    const uint32_t i = g.xs + 1 + 31u * k;
    uint32_t x, n, w, nw;
    split1<SWAP>(img[i], g.shift, &x, &l);
    split1<SWAP>(img[i - g.xs], g.shift, &n, &l);
    split1<SWAP>(img[i - 1], g.shift, &w, &l);
    split1<SWAP>(img[i - g.xs - 1], g.shift, &nw, &l);
    atomicAdd(&lh[1 * 256 + x], 1u);
    atomicAdd(&lh[2 * 256 + ((x - cg_pred(n, w, nw)) & 0xffu)], 1u);
    if (with_delta) {
      x = (x - dhigh[i]) & 0xffu;
      n = (n - dhigh[i - g.xs]) & 0xffu;
      w = (w - dhigh[i - 1]) & 0xffu;
      nw = (nw - dhigh[i - g.xs - 1]) & 0xffu;
      atomicAdd(&lh[3 * 256 + x], 1u);
      atomicAdd(&lh[4 * 256 + ((x - cg_pred(n, w, nw)) & 0xffu)], 1u);
    }
  }
  __syncthreads();
  uint32_t* gh = hists + (size_t)frame * kNumHists * 256;
  for (int b = threadIdx.x; b < kNumHists * 256; b += kThreads) {
    uint32_t v = lh[b];
    if (v) atomicAdd(&gh[b], v);
  }
}

// EstimateEntropy (cc:235-244) of one 256-bin histogram, one bin per thread.
__device__ __forceinline__ uint64_t entropy256(const uint32_t* hist,
                                               uint32_t* scratch) {
  const uint32_t v = hist[threadIdx.x];
  const uint32_t sum = block_sum(v, scratch);
  if (sum == 0) return 0;
  const uint32_t lsum = 31 - __clz(sum);
  const uint32_t term = v ? v * (lsum - (31 - __clz(v))) : 0;
  const uint32_t s = block_sum(term, scratch);
  return (uint64_t)1024 * s / sum;
}

__global__ __launch_bounds__(kThreads) void k_decide(
    const uint32_t* __restrict__ hists, uint8_t* __restrict__ decision,
    int with_delta) {
  __shared__ uint32_t scratch[4];
  const uint32_t* gh = hists + (size_t)blockIdx.x * kNumHists * 256;
  // cc:533: E(countd) < E(counta) with countd all in bin 0, i.e. E(counta) > 0.
  const bool use_delta = with_delta && entropy256(gh, scratch) > 0;
  const uint32_t* ha = gh + (use_delta ? 3 : 1) * 256;
  const uint64_t ea = entropy256(ha, scratch);
  const uint64_t eb = entropy256(ha + 256, scratch);
  const bool use_cg = eb < ea;  // cc:564
  if (threadIdx.x == 0)
    decision[blockIdx.x] = (uint8_t)((use_delta ? 1 : 0) | (use_cg ? 2 : 0));
}

// ------------------------------------------------------- forward predictor --

// One workgroup = one tile of tile_rows rows of one frame, 16 pixels per
// thread and pass.
//   phase A: coalesced 32-byte pixel loads -> split -> (delta) -> low plane
//            stored; high plane into the LDS line buffer; 4-pixel row sums of
//            the RAW high bytes into LDS for the preview.
//   phase B: clamped-gradient residual from the LDS line buffer (west, north,
//            north-west) -> high plane stored; 4x4 preview box from the sums.
template <bool SWAP>
__global__ __launch_bounds__(kThreads) void k_apply(
    const uint16_t* __restrict__ frames, const uint8_t* __restrict__ dhigh,
    const uint8_t* __restrict__ dlow, const uint8_t* __restrict__ decision,
    uint8_t* __restrict__ high, uint8_t* __restrict__ low,
    uint8_t* __restrict__ pv_raw, uint32_t* __restrict__ anylow, Geom g) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* hbuf = smem;
  uint16_t* psum = reinterpret_cast<uint16_t*>(smem + g.lds_h_bytes);
  // last 16 bytes of the dynamic region (no static LDS in front of it: the
  // 16-byte accesses above need the carve base aligned)
  uint32_t* s_or = reinterpret_cast<uint32_t*>(smem + g.lds_bytes - 16);

  const uint32_t frame = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
  const uint32_t row0 = tile * g.tile_rows;
  const uint32_t rows = min(g.tile_rows, g.ys - row0);
  const uint32_t s = row0 * g.xs;     // first pixel of the tile (frame-flat)
  const uint32_t npx = rows * g.xs;   // multiple of 16
  const uint32_t nseg = npx >> 4;
  const size_t fbase = (size_t)frame * g.npix;
  const uint16_t* img = frames + fbase;
  const uint32_t dec = decision[frame];
  const bool use_delta = dec & 1, use_cg = dec & 2;
  const int S = g.shift;
  if (tid == 0) *s_or = 0;
  __syncthreads();

  // Halo: post-delta high bytes of the xs+1 pixels in front of the tile.
  if (use_cg && s > 0) {
    const uint32_t hs = s - g.xs - 1;
    for (uint32_t t = tid; t < g.xs + 1; t += kThreads) {
      uint32_t h, l;
      split1<SWAP>(img[hs + t], S, &h, &l);
      if (use_delta) h = (h - dhigh[hs + t]) & 0xffu;
      hbuf[g.lds_off - (g.xs + 1) + t] = (uint8_t)h;
    }
  }

  uint32_t or_low = 0;
  for (uint32_t seg = tid; seg < nseg; seg += kThreads) {
    const uint32_t o = seg << 4;
    const uint4 a = *reinterpret_cast<const uint4*>(img + s + o);
    const uint4 b = *reinterpret_cast<const uint4*>(img + s + o + 8);
    uint4 h, l;
    split4<SWAP>(a.x, a.y, S, &h.x, &l.x);
    split4<SWAP>(a.z, a.w, S, &h.y, &l.y);
    split4<SWAP>(b.x, b.y, S, &h.z, &l.z);
    split4<SWAP>(b.z, b.w, S, &h.w, &l.w);
    // preview: sums of 4 consecutive RAW high bytes (cc:500-509)
    uint2 ps;
    ps.x = __builtin_amdgcn_sad_u8(h.x, 0, 0) | (__builtin_amdgcn_sad_u8(h.y, 0, 0) << 16);
    ps.y = __builtin_amdgcn_sad_u8(h.z, 0, 0) | (__builtin_amdgcn_sad_u8(h.w, 0, 0) << 16);
    *reinterpret_cast<uint2*>(psum + (o >> 2)) = ps;
    if (S != 8) or_low |= l.x | l.y | l.z | l.w;  // cc:396 etc.: RAW low bytes
    if (use_delta) {  // cc:534-537
      const uint4 dh = *reinterpret_cast<const uint4*>(dhigh + s + o);
      h.x = sub4(h.x, dh.x); h.y = sub4(h.y, dh.y);
      h.z = sub4(h.z, dh.z); h.w = sub4(h.w, dh.w);
      if (S != 8) {
        const uint4 dl = *reinterpret_cast<const uint4*>(dlow + s + o);
        l.x = sub4(l.x, dl.x); l.y = sub4(l.y, dl.y);
        l.z = sub4(l.z, dl.z); l.w = sub4(l.w, dl.w);
      }
    }
    if (S != 8) *reinterpret_cast<uint4*>(low + fbase + s + o) = l;
    if (use_cg) {
      *reinterpret_cast<uint4*>(hbuf + g.lds_off + o) = h;
    } else {
      *reinterpret_cast<uint4*>(high + fbase + s + o) = h;
    }
  }
  // NO_LOW_BYTES needs the OR over every raw low byte of the frame (cc:447).
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) or_low |= __shfl_xor(or_low, off, 64);
  if ((tid & 63) == 0 && or_low) atomicOr(s_or, or_low);
  __syncthreads();
  if (tid == 0 && *s_or) atomicOr(&anylow[frame], *s_or);

  if (use_cg) {  // cc:566-572
    for (uint32_t seg = tid; seg < nseg; seg += kThreads) {
      const uint32_t o = seg << 4;
      const uint8_t* cp = hbuf + g.lds_off + o;
      const uint8_t* np = cp - g.xs;
      uint32_t c[5], n[5];
      c[0] = *reinterpret_cast<const uint32_t*>(cp - 4);
      const uint4 cv = *reinterpret_cast<const uint4*>(cp);
      c[1] = cv.x; c[2] = cv.y; c[3] = cv.z; c[4] = cv.w;
#pragma unroll
      for (int k = 0; k < 5; k++)
        n[k] = *reinterpret_cast<const uint32_t*>(np - 4 + 4 * k);
      uint32_t out[4];
      int wprev = (int)byte_of(c[0], 3), nwprev = (int)byte_of(n[0], 3);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        uint32_t r = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int x = (int)byte_of(c[k + 1], j), nn = (int)byte_of(n[k + 1], j);
          int res = (x - cg_pred(nn, wprev, nwprev)) & 0xff;
          if (s + o + 4 * k + j < g.xs + 1) res = x;  // literals, cc:572
          r |= (uint32_t)res << (8 * j);
          wprev = x;
          nwprev = nn;
        }
        out[k] = r;
      }
      *reinterpret_cast<uint4*>(high + fbase + s + o) =
          make_uint4(out[0], out[1], out[2], out[3]);
    }
  }

  // preview box (cc:500-512): (sum of 16 / 16) & 0xfe
  const uint32_t xq = g.xs >> 2;  // psum entries per row
  const uint32_t npq = (rows >> 2) * g.pw;
  uint8_t* pvo = pv_raw + (size_t)frame * g.npv + (size_t)(row0 >> 2) * g.pw;
  for (uint32_t q = tid; q < npq; q += kThreads) {
    const uint32_t pr = q / g.pw, pc = q - pr * g.pw;
    const uint16_t* p = psum + pr * g.xs + pc;  // 4 rows * xq = xs entries per preview row
    const uint32_t sum = p[0] + p[xq] + p[2 * xq] + p[3 * xq];
    pvo[q] = (uint8_t)((sum >> 4) & 0xfe);
  }
}

// Preview clamped gradient (cc:575-586) + final flags byte.
__global__ __launch_bounds__(kThreads) void k_preview_finish(
    const uint8_t* __restrict__ pv_raw, const uint8_t* __restrict__ decision,
    const uint32_t* __restrict__ anylow, uint8_t* __restrict__ preview,
    uint8_t* __restrict__ flags, Geom g) {
  const uint32_t frame = blockIdx.y;
  const uint32_t dec = decision[frame];
  const uint8_t* in = pv_raw + (size_t)frame * g.npv;
  uint8_t* out = preview + (size_t)frame * g.npv;
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i < g.npv) {
    int x = in[i];
    if ((dec & 2) && i >= g.pw + 1)
      x = (x - cg_pred(in[i - g.pw], in[i - 1], in[i - g.pw - 1])) & 0xff;
    out[i] = (uint8_t)x;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    flags[frame] = (uint8_t)(dec | (anylow[frame] ? 0 : 4));  // cc:447-449
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

}  // namespace

// ------------------------------------------------------------- host side ---

static uint32_t round_up(uint32_t v, uint32_t a) { return (v + a - 1) / a * a; }

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
  // forward tile: about 16K pixels, whole preview rows
  uint32_t rows = (16384 / xs) & ~3u;
  if (rows < 4) rows = 4;
  if (rows > ys) rows = ys;
  g->tile_rows = rows;
  g->tile_px = rows * xs;
  g->tiles = (ys + rows - 1) / rows;
  g->lds_off = round_up(xs + 4, 16);
  g->lds_h_bytes = round_up(g->lds_off + g->tile_px, 16);
  g->lds_bytes = round_up(g->lds_h_bytes + g->tile_px / 2, 16) + 16;
  if (g->lds_bytes > 160 * 1024 - 64) { *why = kWide; return false; }
  g->nd = (g->npix + 14) / 15;
  g->ncg = g->npix > xs + 1 ? (g->npix - xs - 1 + 30) / 31 : 0;
  uint32_t samples = g->nd + g->ncg;
  g->stat_blocks = (samples + kThreads * 8 - 1) / (kThreads * 8);
  if (g->stat_blocks < 1) g->stat_blocks = 1;
  uint32_t ring = 1;
  while (ring < xs + 1 + kInvChunk + 4) ring <<= 1;
  g->ring = ring;
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
  g->fwd_ok = 0;
  return true;
}

size_t fwd_workspace_bytes(const Geom& g, uint32_t nframes) {
  size_t b = (size_t)nframes * kNumHists * 256 * 4;  // hists
  b += round_up(nframes * 4, 256);                    // anylow
  b += round_up(nframes, 256);                        // decision
  b += (size_t)nframes * g.npv;                       // pv_raw
  return b + 256;
}

void carve_fwd_workspace(const Geom& g, uint32_t nframes, void* base,
                         FwdWorkspace* ws) {
  uint8_t* p = static_cast<uint8_t*>(base);
  ws->hists = reinterpret_cast<uint32_t*>(p);
  p += (size_t)nframes * kNumHists * 256 * 4;
  ws->anylow = reinterpret_cast<uint32_t*>(p);
  p += round_up(nframes * 4, 256);
  ws->decision = p;
  p += round_up(nframes, 256);
  ws->pv_raw = p;
}

hipError_t launch_split(const Geom& g, const uint16_t* d_img, uint8_t* d_high,
                        uint8_t* d_low, hipStream_t s) {
  dim3 grid((g.npix / 4 + kThreads - 1) / kThreads);
  if (g.swap)
    k_split<true><<<grid, kThreads, 0, s>>>(d_img, d_high, d_low, g.npix, g.shift);
  else
    k_split<false><<<grid, kThreads, 0, s>>>(d_img, d_high, d_low, g.npix, g.shift);
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
  // hists and anylow are contiguous in the workspace: one memset node
  size_t zero_bytes = (size_t)nframes * kNumHists * 256 * 4 + round_up(nframes * 4, 256);
  if ((e = hipMemsetAsync(ws.hists, 0, zero_bytes, s)) != hipSuccess) return e;
  if (ev) hipEventRecord(ev[0], s);
  k_stats<SWAP><<<dim3(g.stat_blocks, nframes), kThreads, 0, s>>>(
      d_frames, d_dhigh, ws.hists, g, with_delta);
  if (ev) hipEventRecord(ev[1], s);
  k_decide<<<nframes, kThreads, 0, s>>>(ws.hists, ws.decision, with_delta);
  if (ev) hipEventRecord(ev[2], s);
  k_apply<SWAP><<<dim3(g.tiles, nframes), kThreads, g.lds_bytes, s>>>(
      d_frames, d_dhigh, d_dlow, ws.decision, d_high, d_low, ws.pv_raw,
      ws.anylow, g);
  if (ev) hipEventRecord(ev[3], s);
  k_preview_finish<<<dim3((g.npv + kThreads - 1) / kThreads, nframes), kThreads, 0, s>>>(
      ws.pv_raw, ws.decision, ws.anylow, d_preview, d_flags, g);
  if (ev) hipEventRecord(ev[4], s);
  return hipGetLastError();
}

hipError_t launch_predict(const Geom& g, const uint16_t* d_frames,
                          uint32_t nframes, const uint8_t* d_dhigh,
                          const uint8_t* d_dlow, const FwdWorkspace& ws,
                          uint8_t* d_high, uint8_t* d_low, uint8_t* d_preview,
                          uint8_t* d_flags, hipStream_t s, hipEvent_t* ev) {
  static bool attr_set = false;
  if (!attr_set) {  // tiles above 64 KiB of LDS need the opt-in
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_apply<true>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_apply<false>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    attr_set = true;
  }
  if (g.swap)
    return predict_impl<true>(g, d_frames, nframes, d_dhigh, d_dlow, ws, d_high,
                              d_low, d_preview, d_flags, s, ev);
  return predict_impl<false>(g, d_frames, nframes, d_dhigh, d_dlow, ws, d_high,
                             d_low, d_preview, d_flags, s, ev);
}

hipError_t launch_inverse(const Geom& g, const uint8_t* d_high,
                          const uint8_t* d_low, const uint8_t* d_dhigh,
                          const uint8_t* d_dlow, const uint8_t* d_flags,
                          uint32_t nframes, uint16_t* d_frames, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_inverse<true>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_inverse<false>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    attr_set = true;
  }
  if ((g.xs & 3) == 0 && (g.npix & 3) == 0)
    k_inverse<true><<<nframes, 64, g.ring, s>>>(d_high, d_low, d_dhigh, d_dlow,
                                                d_flags, d_frames, g);
  else
    k_inverse<false><<<nframes, 64, g.ring, s>>>(d_high, d_low, d_dhigh, d_dlow,
                                                 d_flags, d_frames, g);
  return hipGetLastError();
}

}  // namespace fpvhip
