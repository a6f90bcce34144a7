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
#include <string.h>

#include <algorithm>
#include <mutex>
#include <type_traits>
#include <vector>

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
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const us2 sh = {(unsigned short)S, (unsigned short)S};  // v_pk_lshlrev_b16: (uint16_t)(p << S), two pixels per op
    const uint32_t qa = __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, va) << sh);
    const uint32_t qb = __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, vb) << sh);
    *h4 = __builtin_amdgcn_perm(qb, qa, 0x07050301u);
    *l4 = __builtin_amdgcn_perm(qb, qa, 0x06040200u);
  }
}

// Per-byte a - b (mod 256) on 4 packed bytes (std::minus<uint8_t>, cc:534-537).
// (Byte-addressed SDWA subtractions into one register would take 3 ops instead of 5,
// but partial-register writes need wait states the compiler cannot see inside inline
// asm on gfx950: measured wrong results.  Plain SWAR it stays.)
__device__ __forceinline__ uint32_t sub4(uint32_t a, uint32_t b) {
  const uint32_t H = 0x80808080u;
  // ((a | H) - (b & ~H)) ^ ((a ^ ~b) & H); the last two operations as ONE three-input
  // bit operation (v_bitop3_b32, truth table of t ^ (~x & H) over (x, H, t)): 5 VALU ops
  return (uint32_t)__builtin_amdgcn_bitop3_b32((int)(a ^ b), (int)H, (int)((a | H) - (b & ~H)), 0xa6);
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

// Device copies the predictor kernels are compared with (bench only): the measured HBM ceiling.
// k_copy_simple: one 16-byte element per thread, non-temporal, a workgroup per 4 KiB, nothing
// looping -- MI355X_MICROARCH.md's "float4 copy" (6.29 TB/s there; 6.5-6.6 TB/s on this pool,
// tools/exp/bw3.hip).  k_copy: round 2's reference, one contiguous 256 KiB chunk per workgroup and
// turn (5.5-5.9 TB/s): persistent workgroups that each stream their own region make poorer use
// of DRAM than workgroups dispatched in memory order -- the observation the tile pass is built on.
__global__ __launch_bounds__(kThreads) void k_copy_simple(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i < n16) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
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
  if (i + 4 <= npix) {
    const uint2 v = *reinterpret_cast<const uint2*>(img + i);
    uint32_t h4, l4;
    split4<false>(v.x, v.y, 0, &h4, &l4);  // delta >> 8, delta & 0xff (cc:337-338)
    *reinterpret_cast<uint32_t*>(high + i) = h4;
    *reinterpret_cast<uint32_t*>(low + i) = l4;
  } else {  // the last 1..3 pixels of a plane whose size is not a multiple of 4 (odd decoder geometries)
    for (uint32_t k = i; k < npix; k++) {
      high[k] = (uint8_t)(img[k] >> 8);
      low[k] = (uint8_t)(img[k] & 0xff);
    }
  }
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
//   k_stream       one pass per frame, strip by strip (one wavefront per strip):
//                  split, preview (with its clamped gradient), delta and CG
//                  residuals under the speculated flags, AND the exact histograms
//                  of the heuristics for that speculation.  The wave that finishes
//                  a frame's LAST strip reduces the strips' histograms, evaluates
//                  EstimateEntropy, compares the true decision with the speculation
//                  and either finalises the frame (flags byte, the preview rows on
//                  strip boundaries) or puts it on list A (delta decision wrong:
//                  the CG histograms are void) or list B (only the CG decision wrong)
//   k_redo_decide  exact decisions of list A from the heuristics' samples
//   k_redo_stream  the same strip pass under the TRUE flags for list B; also
//                  publishes the next batch's speculation (majority vote)
//
// Normally both lists are empty and the two redo launches exit at once.  The
// speculation only ever changes speed: every frame's output is produced under
// its TRUE flags.

constexpr int kHistFused = 3;      // partial histograms per strip: d, a, b
constexpr uint32_t kStep = 1024;   // pixels per wave step
#ifndef FPV_DEPTH
#define FPV_DEPTH 3
#endif
constexpr int kDepth = FPV_DEPTH;  // steps of pixel / delta loads in flight per wave

// Workspace counter block (u32 words).  Two blocks alternate between batches: a
// batch's first kernel clears the other block, so no memset launch is needed.
enum : uint32_t {
  kCtrListA = 0,     // |list A|
  kCtrListB = 1,     // |list B|
  kCtrTask = 2,      // k_stream's task counter
  kCtrRedoTask = 3,  // k_redo_stream's task counter
  kCtrVotes = 4,     // [4] decisions seen, by value
  kCtrWords = 16
};

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

// Hand-off accesses (strip -> the wave that finalises the frame, possibly on
// another XCD): agent-scope relaxed atomics = write-through `sc1` stores and
// L1-bypassing `sc1` loads.  The producer drains them (s_waitcnt vmcnt(0)) before
// its arrival atomic; the consumer runs an agent-scope acquire after its own.
__device__ __forceinline__ void st_agent64(void* p, uint64_t v) {
  __hip_atomic_store(reinterpret_cast<uint64_t*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t ld_agent64(const void* p) {
  return __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_agent32(const void* p) {
  return __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent8(uint8_t* p, uint8_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_agent8(const uint8_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint4 ld_agent128(const void* p) {
  const uint64_t a = ld_agent64(p), b = ld_agent64(static_cast<const uint8_t*>(p) + 8);
  return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
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

// Residuals of 16 pixels; wprev / nwprev: the dwords in front of cur / north.
__device__ __forceinline__ uint4 cg_res16(uint4 cur, uint4 north, uint32_t wprev, uint32_t nwprev) {
  uint32_t xo_prev = __builtin_amdgcn_perm(0, wprev, 0x0c030c01u);
  uint32_t no_prev = __builtin_amdgcn_perm(0, nwprev, 0x0c030c01u);
  uint4 r;
  r.x = cg_res4(cur.x, north.x, &xo_prev, &no_prev);
  r.y = cg_res4(cur.y, north.y, &xo_prev, &no_prev);
  r.z = cg_res4(cur.z, north.z, &xo_prev, &no_prev);
  r.w = cg_res4(cur.w, north.w, &xo_prev, &no_prev);
  return r;
}

struct Pre {  // prefetch registers of one step
  uint4 a, b, dh, dl;
};

// Diagnostic builds only (-DFPV_ABL=<bits>, never the shipped binary): leave parts of the
// step out to see what they cost.  1 CG arithmetic, 2 ring reads + CG, 4 heuristic samples,
// 8 preview adds + rows, 16 delta subtraction, 32 low-plane store, 64 high-plane store.
#ifndef FPV_ABL
#define FPV_ABL 0
#endif

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

// Kernel arguments kept small on purpose: everything else is derived.
struct StreamArgs {
  uint32_t xs, ys, npix, pw, npv;
  uint32_t strip_rows, strips, ring_fwd, ring_base, nacc, shift, nframes, strip_major;
  uint32_t pwr;         // pw rounded up to 16: bytes of one boundary preview row
  uint32_t pvb_stride;  // bytes of one (frame, strip) boundary record: first row | last row | 8
  uint32_t with_delta;  // a delta frame exists
  uint32_t wave_lds;    // LDS bytes of one wave
  uint32_t wg_waves;    // waves (= consecutive frames) per workgroup of k_stream
  uint32_t bd_stride;   // boundary words per frame
  uint32_t redo_rows[4];  // strip heights the redo pass chooses from (ascending; 0 = none)
  uint32_t in_pitch;    // pixels from one frame to the next in `frames` (npix)
  uint32_t out_pitch;   // bytes from one frame to the next in `high` / `low` (npix)
};

// Device pointers of one forward launch.
struct StreamPtrs {
  const uint16_t* frames;
  const uint8_t *dhigh, *dlow;
  uint8_t *high, *low, *preview, *flags;
  uint32_t *anylow, *frame_done;
  uint32_t* bdone;     // [frame][strip]: arrivals at the boundary above the strip
  uint16_t* partials;
  uint8_t* pvb;        // boundary preview rows (raw) of every (frame, strip)
  uint32_t* ctr;       // this batch's counter block
  uint32_t* ctr_next;  // the next batch's (cleared here)
  uint8_t* decision;
  uint32_t *list_a, *list_b;
  uint8_t* spec;       // speculation byte: read by k_stream, written by k_redo_stream
};

// The arguments of the streaming kernels as ONE by-value kernel parameter.  The kernels do
// not read the parameter but the kernarg segment it lies in, and again after every strip:
// values that are only needed between two strips (most pointers, most of the geometry)
// then do not occupy scalar registers across the strip's loop.  Held in registers, they
// pushed the loop's own scalars out: 16 spill reloads (v_readlane) per 1024-pixel step.
struct StreamLaunch {
  StreamPtrs P;
  StreamArgs g;
};
typedef const __attribute__((address_space(4))) StreamLaunch* LaunchArgs;
__device__ __forceinline__ LaunchArgs launch_args() {
  LaunchArgs p = (LaunchArgs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));  // opaque: loads through it stay behind this point
  return p;
}
__device__ __forceinline__ StreamArgs launch_g() {
  StreamArgs g;
  __builtin_memcpy(&g, &launch_args()->g, sizeof g);
  return g;
}
__device__ __forceinline__ StreamPtrs launch_p() {
  StreamPtrs P;
  __builtin_memcpy(&P, &launch_args()->P, sizeof P);
  return P;
}

struct StreamCtx {
#ifdef FPV_STAMPS
  unsigned long long seg[6], tlast;
#endif
  // constant per kernel
  uint32_t lane, xs, pw, ring_mask, ring_base, nacc, shift;
  uint8_t* ring;
  uint32_t* acc;     // preview accumulators: FAST [2][4][pw/4] (transposed), else [nacc][pw]
  uint32_t* hist;    // [3][256] bins
  uint8_t* pvprev;   // previous raw preview row (north of the next one)
  // per strip
  bool use_delta, use_cg;
  uint32_t s, npx;   // first pixel (frame-flat) and pixel count of the strip
  const uint16_t* img;
  const uint8_t *dh, *dl;
  uint8_t *high_f, *low_f, *pv_f;
  uint8_t* rec;      // this strip's boundary record
  uint32_t pwr;
  uint32_t row, col; // position of this lane's segment relative to the strip start
  uint32_t drow, dcol; // kStep / xs, kStep % xs
  uint32_t j0;       // offset of the segment's first delta-heuristic sample (index % 15 == 0): 0..14
  uint32_t j1;       // offset of the next index congruent to xs + 1 (mod 31): 0..30
  uint32_t or_low;
  uint32_t groups_done, prow0, prows;  // preview rows of the strip: written, first (absolute), count
  uint32_t wlast, wlast2;              // last raw preview pixel of the previous row / the one before
  bool first_strip, last_strip;
};

__device__ __forceinline__ uint32_t ring_at(const StreamCtx& c, int o) {
  return ((uint32_t)(o + (int)c.ring_base)) & c.ring_mask;
}

__device__ __forceinline__ void hist_inc(uint32_t* hist, uint32_t which, uint32_t bin) {
  atomicAdd(&hist[which * 256 + bin], 1u);
}

// Byte j (0..15) of 16 bytes held in four dwords, without a branch or a lane mask:
// a byte permute of each 8-byte half, merged by a bit-field insert on bit 3 of j.
struct ByteSel { uint32_t sel, mask; };
__device__ __forceinline__ ByteSel byte_sel(uint32_t j) {
  ByteSel b;
  b.sel = (j & 7u) | 0x0c0c0c00u;                                   // v_perm selector: byte j & 7, zeros above
  b.mask = (uint32_t)__builtin_amdgcn_sbfe((int)j, 3, 1);          // all ones when j >= 8
  return b;
}
__device__ __forceinline__ uint32_t byte16(const uint4& v, const ByteSel& b) {
  const uint32_t lo = __builtin_amdgcn_perm(v.y, v.x, b.sel), hi = __builtin_amdgcn_perm(v.w, v.z, b.sel);
  return (hi & b.mask) | (lo & ~b.mask);
}

// DELTA / LOW are the (wave-uniform) conditions "delta predictor applied" and
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
      const uint32_t od = (FPV_ABL & 256) ? c.lane * 16 : c.s + o;  // 256: always the same 1 KiB (L1 hits): what do the delta MISSES cost?
      p->dh = *reinterpret_cast<const uint4*>(c.dh + od);
      if (LOW) p->dl = *reinterpret_cast<const uint4*>(c.dl + od);
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

// One finished preview row (cc:500-512 then cc:575-586): (sum / 16) & 0xfe of the
// 4x4 boxes from the accumulators, the clamped gradient against the previous raw
// row (kept in LDS) when CG is applied, accumulators cleared.  Rows whose north
// row belongs to another strip (the first row of strips > 0) cannot be predicted
// here: their raw bytes go to the strip's boundary record, as do the raw bytes of
// the strip's last row, and the wave that finalises the frame predicts them.
template <bool FAST>
__device__ __forceinline__ void preview_row_out(StreamCtx& c) {
  const uint32_t r = c.prow0 + c.groups_done;  // absolute preview row
  const bool first_row = c.groups_done == 0, last_row = c.groups_done + 1 == c.prows;
  const bool defer = c.use_cg && first_row && !c.first_strip;
  const bool rec_last = c.use_cg && last_row && !c.last_strip;
  const bool predict = c.use_cg && r > 0 && !defer;
  uint8_t* out = c.pv_f + (size_t)r * c.pw;
  uint32_t newlast = 0;
  if (FAST) {  // pw % 16 == 0: 8 preview pixels per lane, transposed accumulators
    const uint32_t q4 = c.pw >> 2;
    uint32_t* a = c.acc + (c.groups_done & (c.nacc - 1)) * c.pw;
    uint32_t carry_w = c.wlast << 24, carry_nw = c.wlast2 << 24;
    for (uint32_t k0 = 0; k0 < c.pw; k0 += 512) {
      const uint32_t k = k0 + c.lane * 8;
      const bool act = k < c.pw;
      uint2 cur = make_uint2(0, 0), north = make_uint2(0, 0);
      if (act) {
        uint2 sj[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          sj[j] = *reinterpret_cast<const uint2*>(a + j * q4 + (k >> 2));
          *reinterpret_cast<uint2*>(a + j * q4 + (k >> 2)) = make_uint2(0, 0);
        }
        // dword d holds preview pixels k + 4 d .. k + 4 d + 3 = plane 0..3 of quad d
        auto pack = [](uint32_t s0, uint32_t s1, uint32_t s2, uint32_t s3) {
          const uint32_t t01 = ((s0 | (s1 << 16)) >> 4) & 0x00fe00feu;
          const uint32_t t23 = ((s2 | (s3 << 16)) >> 4) & 0x00fe00feu;
          return __builtin_amdgcn_perm(t23, t01, 0x06040200u);
        };
        cur.x = pack(sj[0].x, sj[1].x, sj[2].x, sj[3].x);
        cur.y = pack(sj[0].y, sj[1].y, sj[2].y, sj[3].y);
        if (c.use_cg) {
          north = *reinterpret_cast<const uint2*>(c.pvprev + k);
          *reinterpret_cast<uint2*>(c.pvprev + k) = cur;
        }
      }
      uint32_t wprev = (uint32_t)__shfl_up((int)cur.y, 1, 64), nwprev = (uint32_t)__shfl_up((int)north.y, 1, 64);
      if (c.lane == 0) { wprev = carry_w; nwprev = carry_nw; }
      carry_w = (uint32_t)__builtin_amdgcn_readlane((int)cur.y, 63);
      carry_nw = (uint32_t)__builtin_amdgcn_readlane((int)north.y, 63);
      if (k0 + 512 >= c.pw)  // the row's last pixel: byte 3 of the last active lane's last dword
        newlast = (uint32_t)__builtin_amdgcn_readlane((int)cur.y, (int)((c.pw - 1 - k0) >> 3)) >> 24;
      if (act) {
        uint2 res = cur;
        if (predict) {
          uint32_t xo_prev = __builtin_amdgcn_perm(0, wprev, 0x0c030c01u);
          uint32_t no_prev = __builtin_amdgcn_perm(0, nwprev, 0x0c030c01u);
          res.x = cg_res4(cur.x, north.x, &xo_prev, &no_prev);
          res.y = cg_res4(cur.y, north.y, &xo_prev, &no_prev);
          if (r == 1 && k == 0) res.x = (res.x & 0xffffff00u) | (cur.x & 0xffu);  // pixel pw is a literal (cc:575)
        }
        const uint64_t raw = (uint64_t)cur.x | ((uint64_t)cur.y << 32);
        if (defer) st_agent64(c.rec + k, raw);
        else *reinterpret_cast<uint2*>(out + k) = res;
        if (rec_last) st_agent64(c.rec + c.pwr + k, raw);
      }
    }
  } else {  // any width: one preview pixel per lane and round
    uint32_t* a = c.acc + (c.groups_done % c.nacc) * c.pw;
    uint32_t carry_w = c.wlast, carry_nw = c.wlast2;
    for (uint32_t k0 = 0; k0 < c.pw; k0 += 64) {
      const uint32_t k = k0 + c.lane;
      const bool act = k < c.pw;
      uint32_t x = 0, n = 0;
      if (act) {
        x = (a[k] >> 4) & 0xfeu;
        a[k] = 0;
        if (c.use_cg) {
          n = c.pvprev[k];
          c.pvprev[k] = (uint8_t)x;
        }
      }
      uint32_t w = (uint32_t)__shfl_up((int)x, 1, 64), nw = (uint32_t)__shfl_up((int)n, 1, 64);
      if (c.lane == 0) { w = carry_w; nw = carry_nw; }
      carry_w = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
      carry_nw = (uint32_t)__builtin_amdgcn_readlane((int)n, 63);
      if (k0 + 64 >= c.pw) newlast = (uint32_t)__builtin_amdgcn_readlane((int)x, (int)(c.pw - 1 - k0));
      if (act) {
        uint32_t res = x;
        if (predict && !(r == 1 && k == 0)) res = (x - (uint32_t)cg_pred((int)n, (int)w, (int)nw)) & 0xffu;
        if (defer) st_agent8(c.rec + k, (uint8_t)x);
        else out[k] = (uint8_t)res;
        if (rec_last) st_agent8(c.rec + c.pwr + k, (uint8_t)x);
      }
    }
  }
  // the last pixel of the row before this one: north-west of the next strip's first pixel
  if (rec_last && c.lane == 0) st_agent64(c.rec + 2 * c.pwr, (uint64_t)c.wlast);
  c.wlast2 = c.wlast;
  c.wlast = newlast;
  c.groups_done++;
}

template <bool FAST>
__device__ __forceinline__ void stream_preview_flush(StreamCtx& c, uint32_t upto_groups) {
  while (c.groups_done < upto_groups) preview_row_out<FAST>(c);
}

// FAST: xs % 64 == 0 and xs >= 256 (a segment never straddles image rows, two or four
// preview accumulator rows suffice, the north bytes are one aligned 16-byte read,
// a preview row is a whole number of 16-byte groups).
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
    if (!(FPV_ABL & 8)) {
      const uint32_t q[4] = {__builtin_amdgcn_sad_u8(h.x, 0, 0), __builtin_amdgcn_sad_u8(h.y, 0, 0),
                             __builtin_amdgcn_sad_u8(h.z, 0, 0), __builtin_amdgcn_sad_u8(h.w, 0, 0)};
      if (FAST) {
        // transposed: plane j holds quad j of every 16-pixel segment, so the 64 lanes of
        // one add hit 64 consecutive dwords (no bank conflicts)
        const uint32_t q4 = c.pw >> 2;
        const uint32_t slot = (c.row >> 2) & (c.nacc - 1);  // nacc is 2 or 4: two conditional adds, no multiply
        const uint32_t slot_off = (uint32_t)(-(int)(slot & 1u)) & c.pw;
        uint32_t* a = c.acc + slot_off + ((uint32_t)(-(int)(slot >> 1)) & (2u * c.pw)) + (c.col >> 4);
#pragma unroll
        for (int j = 0; j < 4; j++) atomicAdd(a + j * q4, q[j]);
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
    if (!(FPV_ABL & 4)) {  // cc:526-531: histogram of high[0::15]
      // j0 = offset of the segment's first sample (0..14); a second one (offset 15) iff j0 == 0
      hist_inc(c.hist, 0, byte16(h, byte_sel(c.j0)));
      atomicAdd(&c.hist[h.w >> 24], (c.j0 - 1u) >> 31);            // adds 1 iff j0 == 0
      const int t = (int)c.j0 - (int)(kStep % 15u);                // the next segment starts kStep pixels on
      c.j0 = (uint32_t)(t + ((t >> 31) & 15));
    }
    FPV_STAMP(2);  // preview adds + delta-histogram sample
    if (DELTA && !(FPV_ABL & 16)) {  // cc:534-537
      h.x = sub4(h.x, p.dh.x); h.y = sub4(h.y, p.dh.y);
      h.z = sub4(h.z, p.dh.z); h.w = sub4(h.w, p.dh.w);
      if (LOW) {
        l.x = sub4(l.x, p.dl.x); l.y = sub4(l.y, p.dl.y);
        l.z = sub4(l.z, p.dl.z); l.w = sub4(l.w, p.dl.w);
      }
    }
    if (LOW && !(FPV_ABL & 32)) st_stream(c.low_f + c.s + o, l);
    const uint32_t a0 = ring_at(c, (int)o);
    *reinterpret_cast<uint4*>(c.ring + a0) = h;
    if (!c.use_cg) st_stream(c.high_f + c.s + o, h);
    __builtin_amdgcn_wave_barrier();  // the reads below see the whole wave's ring writes
    FPV_STAMP(3);  // delta + low store + ring write
    if ((FPV_ABL & 2) && c.use_cg && !(FPV_ABL & 64)) st_stream(c.high_f + c.s + o, h);
    uint4 r;  // the clamped-gradient residuals of this segment (set when CG is applied)
    if (FPV_ABL & 2) r = h;
    if (c.use_cg && !(FPV_ABL & 2)) {  // cc:566-571
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
      r = (FPV_ABL & 1) ? make_uint4(h.x ^ nv.x, h.y ^ nv.y ^ c0, h.z ^ nv.z ^ n0, h.w ^ nv.w) : cg_res16(h, nv, c0, n0);
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
      if (!(FPV_ABL & 64)) st_stream(c.high_f + c.s + o, r);
    }
    FPV_STAMP(4);  // ring reads + clamped gradient + high store
    if (!(FPV_ABL & 4)) {  // cc:554-562: the CG heuristic's sample at i = xs + 1 + 31 k, if one falls in this segment
      // j1 = offset (0..30) of the next index congruent to xs + 1 (mod 31); a sample iff it lies
      // inside the segment and not in front of xs + 1
      const uint32_t pos = c.s + o + c.j1;
      const uint32_t take = ((c.j1 - 16u) >> 31) & ((c.xs - pos) >> 31);  // j1 < 16 && pos > xs
      if (c.use_cg) {
        // with CG applied the sample's b = a - ClampedGradient(...) (cc:559-561) IS the
        // residual just computed (samples start at xs + 1, behind the literal prefix)
        const ByteSel bs = byte_sel(c.j1);
        atomicAdd(&c.hist[256 + byte16(h, bs)], take);
        atomicAdd(&c.hist[512 + byte16(r, bs)], take);
      } else if (take) {
        const uint32_t aj = a0 + c.j1;
        const int x = c.ring[aj & c.ring_mask], w = c.ring[(aj - 1) & c.ring_mask];
        const int n = c.ring[(aj - c.xs) & c.ring_mask], nw = c.ring[(aj - c.xs - 1) & c.ring_mask];
        hist_inc(c.hist, 1, (uint32_t)x);
        hist_inc(c.hist, 2, (uint32_t)(x - cg_pred(n, w, nw)) & 0xffu);
      }
      const int t = (int)c.j1 - (int)(kStep % 31u);
      c.j1 = (uint32_t)(t + ((t >> 31) & 31));
    }
    // advance this lane's (row, col) by one step
    {
      const int t = (int)(c.col + c.dcol) - (int)c.xs;   // dcol = kStep % xs, drow = kStep / xs
      const int neg = t >> 31;
      c.col = (uint32_t)(t + ((int)c.xs & neg));
      c.row += c.drow + 1u + (uint32_t)neg;
    }
  }
  __builtin_amdgcn_wave_barrier();
  FPV_STAMP(5);  // CG sample + bookkeeping
}

// One strip of one frame under the flags (DELTA, use_cg): everything of the pass
// except the frame-level hand-off.  Returns with the strip's partial histograms
// in c.hist and the OR of its raw low bytes in every lane.
template <bool SWAP, bool FAST, bool DELTA, bool LOW>
__device__ __forceinline__ uint32_t run_strip(StreamCtx& c, const StreamArgs& g, const StreamPtrs& P,
                                              uint32_t frame, uint32_t strip, bool use_cg) {
  const uint32_t row0 = strip * g.strip_rows;
  const uint32_t rows = min(g.strip_rows, g.ys - row0);
  const size_t fbase = (size_t)frame * g.out_pitch;
  c.use_delta = DELTA;
  c.use_cg = use_cg;
  c.s = row0 * g.xs;
  c.npx = rows * g.xs;
  c.img = P.frames + (size_t)frame * g.in_pitch;
  c.dh = P.dhigh; c.dl = P.dlow;
  c.high_f = P.high + fbase;
  c.low_f = P.low + fbase;
  c.pv_f = P.preview + (size_t)frame * g.npv;
  c.rec = P.pvb + ((size_t)frame * g.strips + strip) * g.pvb_stride;
  c.row = (c.lane * 16) / g.xs; c.col = (c.lane * 16) % g.xs;
  c.j0 = (15u - (c.s + c.lane * 16) % 15u) % 15u;
  {
    const int d = ((int)(g.xs + 1) - (int)(c.s + c.lane * 16)) % 31;
    c.j1 = (uint32_t)(d < 0 ? d + 31 : d);
  }
  c.or_low = 0;
  c.groups_done = 0;
  c.prow0 = row0 >> 2;
  c.prows = rows >> 2;
  c.first_strip = strip == 0;
  c.last_strip = strip + 1 == g.strips;
  c.wlast = c.wlast2 = 0;
  const uint32_t niter = (c.npx + kStep - 1) / kStep;

  for (uint32_t k = c.lane * 4; k < kHistFused * 256; k += 256)
    *reinterpret_cast<uint4*>(c.hist + k) = make_uint4(0, 0, 0, 0);
  // kDepth steps of loads are in flight: the loads of step t + kDepth are issued when step t
  // has consumed its registers.  The loop is unrolled by kDepth so that every stage keeps
  // its own registers.  The pipeline is filled FIRST, so that the halo's loads below travel
  // together with it: a strip's prologue costs one memory round trip, not one per halo chunk.
  Pre p[kDepth];
#pragma unroll
  for (int d = 0; d < kDepth; d++) stream_load<FAST, DELTA, LOW>(c, d, niter, &p[d]);
  if (c.s > 0) {  // halo: post-delta high bytes of the xs+16 pixels in front of the strip
    const uint32_t hs = c.s - g.xs - 16, ngrp = (g.xs + 16) >> 2;
    // raw preview pixel (prow0 - 1, pw - 1): north-west of this strip's second preview
    // row's first pixel.  Its 4x4 box lies in the four image rows above the strip.
    uint2 vw = make_uint2(0, 0);
    if (use_cg && c.lane < 4)
      vw = *reinterpret_cast<const uint2*>(c.img + (size_t)(row0 - 4 + c.lane) * g.xs + g.xs - 4);
    constexpr int kHaloUnroll = 8;  // 8 x 64 lanes x 4 pixels per round: all loads first
    for (uint32_t base = 0; base < ngrp; base += 64 * kHaloUnroll) {
      uint2 v[kHaloUnroll];
      uint32_t dv[kHaloUnroll];
#pragma unroll
      for (int u = 0; u < kHaloUnroll; u++) {
        const uint32_t k = base + 64 * u + c.lane;
        v[u] = make_uint2(0, 0);
        dv[u] = 0;
        if (k < ngrp) {
          v[u] = *reinterpret_cast<const uint2*>(c.img + hs + 4 * k);
          if (DELTA) dv[u] = *reinterpret_cast<const uint32_t*>(P.dhigh + hs + 4 * k);
        }
      }
#pragma unroll
      for (int u = 0; u < kHaloUnroll; u++) {
        const uint32_t k = base + 64 * u + c.lane;
        if (k < ngrp) {
          uint32_t h4, l4;
          split4<SWAP>(v[u].x, v[u].y, (int)g.shift, &h4, &l4);
          if (DELTA) h4 = sub4(h4, dv[u]);
          *reinterpret_cast<uint32_t*>(c.ring + ring_at(c, (int)(4 * k) - (int)(g.xs + 16))) = h4;
        }
      }
    }
    if (use_cg) {
      uint32_t sum = 0;
      if (c.lane < 4) {
        uint32_t h4, l4;
        split4<SWAP>(vw.x, vw.y, (int)g.shift, &h4, &l4);
        sum = __builtin_amdgcn_sad_u8(h4, 0, 0);
      }
      sum += (uint32_t)__shfl_xor((int)sum, 1, 64);
      sum += (uint32_t)__shfl_xor((int)sum, 2, 64);
      c.wlast = ((uint32_t)__builtin_amdgcn_readfirstlane((int)sum) >> 4) & 0xfeu;
    }
  }
  __builtin_amdgcn_wave_barrier();

  const uint32_t nd = niter - niter % kDepth;
  for (uint32_t it = 0; it < nd; it += kDepth) {
#pragma unroll
    for (int d = 0; d < kDepth; d++) {
      stream_step<SWAP, FAST, DELTA, LOW>(c, it + d, p[d]);
      stream_load<FAST, DELTA, LOW>(c, it + d + kDepth, niter, &p[d]);
    }
    // preview rows completed by these steps (wave-uniform; the accumulator rows cover
    // every 4-row group that kDepth steps can touch)
    if (!(FPV_ABL & 8)) stream_preview_flush<FAST>(c, min((it + kDepth) * kStep, c.npx) / (4 * c.xs));
  }
#pragma unroll
  for (int d = 0; d < kDepth - 1; d++)
    if (nd + d < niter) stream_step<SWAP, FAST, DELTA, LOW>(c, nd + d, p[d]);
  if (!(FPV_ABL & 8)) stream_preview_flush<FAST>(c, c.prows);

  uint32_t orl = c.or_low;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) orl |= (uint32_t)__shfl_xor((int)orl, off, 64);
  return orl;
}

// Sum over the 64 lanes of a wave; every lane gets the result.
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// EstimateEntropy (cc:235-244) of one 256-bin histogram held by a wave, four
// bins per lane.  Exact integer arithmetic.
__device__ __forceinline__ uint64_t entropy_wave(const uint32_t v[4]) {
  const uint32_t sum = wave_sum(v[0] + v[1] + v[2] + v[3]);
  if (sum == 0) return 0;
  const uint32_t lsum = 31 - __clz(sum);
  uint32_t term = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) term += v[k] ? v[k] * (lsum - (31 - __clz(v[k]))) : 0;
  return (uint64_t)1024 * wave_sum(term) / sum;
}

// Clamped gradient of the preview row that opens strip s >= 1 (cc:575-586): its north
// row is the last row of the strip above, produced by another wave.  Both raw rows
// come from the strips' boundary records.  `own_up` / `own_down`: this wave is the second
// of the two neighbours to finish at the boundary above / below its strip and does that row.
template <bool FAST>
__device__ __forceinline__ void preview_boundary_rows(const StreamArgs& g, const uint8_t* pvb_frame,
                                                      uint8_t* pv_f, uint32_t lane, uint32_t strip,
                                                      bool own_up, bool own_down) {
  const uint32_t prs = g.strip_rows >> 2;  // preview rows per strip
  const uint32_t per_row = FAST ? g.pw >> 4 : g.pw;
  for (uint32_t it = lane; it < 2 * per_row; it += 64) {
    const uint32_t which = it >= per_row ? 1u : 0u;
    if (which ? !own_down : !own_up) continue;
    const uint32_t s = strip + which;
    const uint8_t* first = pvb_frame + (size_t)s * g.pvb_stride;
    const uint8_t* last = pvb_frame + (size_t)(s - 1) * g.pvb_stride + g.pwr;
    if (FAST) {
      const uint32_t k = (it - which * per_row) << 4;
      const uint4 cur = ld_agent128(first + k), north = ld_agent128(last + k);
      const uint32_t wprev = k ? ld_agent32(first + k - 4) : ld_agent32(last + g.pw - 4);
      const uint32_t nwprev = k ? ld_agent32(last + k - 4) : ld_agent32(last + g.pwr) << 24;
      uint4 res = cg_res16(cur, north, wprev, nwprev);
      if (s * prs == 1 && k == 0) res.x = (res.x & 0xffffff00u) | (cur.x & 0xffu);  // pixel pw is a literal (cc:575)
      st_stream(pv_f + (size_t)s * prs * g.pw + k, res);
    } else {
      const uint32_t k = it - which * per_row;
      const int x = (int)ld_agent8(first + k), n = (int)ld_agent8(last + k);
      const int w = (int)(k ? ld_agent8(first + k - 1) : ld_agent8(last + g.pw - 1));
      const int nw = (int)(k ? ld_agent8(last + k - 1) : ld_agent8(last + g.pwr));
      const bool literal = s * prs == 1 && k == 0;  // pixel pw (cc:575 starts at pw + 1)
      pv_f[(size_t)s * prs * g.pw + k] = (uint8_t)(literal ? x : (x - cg_pred(n, w, nw)) & 0xff);
    }
  }
}

// Strip epilogue: the strip's hand-off data leave through write-through stores, the
// wave drains them, and ONE returning atomic instruction (four lanes, four words) counts
// the strip as done, fetches the wave's next task and signs the strip off at the
// boundaries it shares with its two neighbours.  A boundary word counts arrivals and is
// never reset: each pass adds two, so an odd old value tells the second neighbour, which
// then writes the preview row that straddles the boundary (its clamped gradient needs raw
// rows of both strips).  Returns true to the wave that completed the frame; every reader of
// other strips' data goes through an agent-scope acquire.  *next_task = the task counter's
// old value.
template <bool PARTIALS, bool FAST>
__device__ __forceinline__ bool strip_arrive(const StreamCtx& c, const StreamArgs& g, const StreamPtrs& P,
                                             uint32_t frame, uint32_t strip, uint32_t orl,
                                             uint32_t* task_ctr, uint32_t* next_task) {
  if (PARTIALS) {
    uint8_t* ph = reinterpret_cast<uint8_t*>(P.partials + ((size_t)frame * g.strips + strip) * (kHistFused * 256));
#pragma unroll
    for (int i = 0; i < 3; i++) {  // 3 x 256 bins as 16-bit counts (a strip has at most 60000 samples): 192 x 8 bytes
      const uint32_t idx = c.lane + 64 * i;
      const uint4 v = *reinterpret_cast<const uint4*>(c.hist + 4 * idx);
      st_agent64(ph + 8 * idx, (uint64_t)(v.x | (v.y << 16)) | ((uint64_t)(v.z | (v.w << 16)) << 32));
    }
  }
  if (c.lane == 0 && orl) atomicOr(&P.anylow[frame], orl);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every store and atomic of this strip has left
  const bool has_up = c.use_cg && strip > 0, has_down = c.use_cg && strip + 1 < g.strips;
  uint32_t* word = nullptr;
  uint32_t* const bd = P.bdone + (size_t)frame * g.bd_stride + strip;
  if (c.lane == 0) word = &P.frame_done[frame];
  else if (c.lane == 1) word = task_ctr;
  else if (c.lane == 2 && has_up) word = bd;
  else if (c.lane == 3 && has_down) word = bd + 1;
  uint32_t old = 0;
  if (word) old = atomicAdd(word, 1u);
  *next_task = (uint32_t)__builtin_amdgcn_readlane((int)old, 1);
  const bool last = (uint32_t)__builtin_amdgcn_readlane((int)old, 0) + 1 == g.strips;
  const bool own_up = has_up && ((uint32_t)__builtin_amdgcn_readlane((int)old, 2) & 1u);
  const bool own_down = has_down && ((uint32_t)__builtin_amdgcn_readlane((int)old, 3) & 1u);
  if (last || own_up || own_down) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (own_up || own_down)
    preview_boundary_rows<FAST>(g, P.pvb + (size_t)frame * g.strips * g.pvb_stride,
                                P.preview + (size_t)frame * g.npv, c.lane, strip, own_up, own_down);
  if (last && c.lane == 0) atomicExch(&P.frame_done[frame], 0u);  // ready for the redo pass / the next batch
  return last;
}

// Flags byte (cc:447-449 + the decisions).
__device__ __forceinline__ void finalize_frame(const StreamCtx& c, const StreamPtrs& P, uint32_t frame, uint32_t dec) {
  if (c.lane == 0) P.flags[frame] = (uint8_t)(dec | (atomicExch(&P.anylow[frame], 0u) ? 0u : 4u));
}

__device__ __forceinline__ void stream_ctx_init(StreamCtx& c, const StreamArgs& g, uint8_t* smem) {
  c.lane = threadIdx.x & 63u;
  c.ring = smem;
  c.acc = reinterpret_cast<uint32_t*>(smem + g.ring_fwd);
  c.hist = reinterpret_cast<uint32_t*>(smem + g.ring_fwd + g.nacc * g.pw * 4);
  c.pvprev = smem + g.ring_fwd + g.nacc * g.pw * 4 + kHistFused * 256 * 4;
  c.xs = g.xs; c.pw = g.pw; c.ring_mask = g.ring_fwd - 1; c.ring_base = g.ring_base;
  c.nacc = g.nacc; c.shift = g.shift; c.pwr = g.pwr;
  c.drow = kStep / g.xs; c.dcol = kStep % g.xs;
  for (uint32_t k = c.lane; k < g.nacc * g.pw; k += 64) c.acc[k] = 0;
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
//     row per 4-row group; a finished row gets its clamped gradient (north =
//     the previous raw row, kept in LDS) and is written at once;
//   - the heuristic samples (i % 15 == 0; i = xs+1+31k) are counted in the
//     wave's LDS histograms (16-bit bins packed in dwords).
// ---------------------------------------------------------------------------
// A workgroup is g.wg_waves wavefronts, each with its own LDS region and its own frame:
// a task is one strip of wg_waves CONSECUTIVE frames, and the waves start it together, so
// that the strip's delta bytes, which all of them read at about the same time, come from
// the CU's L1 for all but the first.  Nothing else couples the waves of a workgroup.
#ifdef FPV_WAVES_EU  // experiments: register budget for FPV_WAVES_EU waves per SIMD
#define FPV_STREAM_BOUNDS __attribute__((amdgpu_waves_per_eu(FPV_WAVES_EU, FPV_WAVES_EU))) __launch_bounds__(512)
#else
#define FPV_STREAM_BOUNDS __launch_bounds__(512, 2)
#endif
template <bool SWAP, bool FAST, bool LOW>
__global__ FPV_STREAM_BOUNDS void k_stream(StreamLaunch) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  StreamCtx c;
  bool sp_delta, sp_cg;
  {
    const StreamArgs g = launch_g();
    const StreamPtrs P = launch_p();
    stream_ctx_init(c, g, smem + wave * g.wave_lds);
    const uint32_t spec = *P.spec & (g.with_delta ? 3u : 2u);
    sp_delta = spec & 1u;
    sp_cg = spec & 2u;
    if (blockIdx.x == 0 && threadIdx.x < kCtrWords) P.ctr_next[threadIdx.x] = 0;  // the next batch's counters
  }
#ifdef FPV_STAMPS
  for (int k = 0; k < 6; k++) c.seg[k] = 0;
  c.tlast = __builtin_amdgcn_s_memtime();
#endif
  // The first task of a workgroup is its index; later ones come from the device-wide
  // counter (thousands of waves hitting one word at launch would queue for ~25 us:
  // same-address atomics serialise at ~12 ns).
  uint32_t task = blockIdx.x;
  for (;;) {
    uint32_t frame, strip, orl = 0;
    bool active;
    {
      const StreamArgs g = launch_g();
      const StreamPtrs P = launch_p();
      const uint32_t W = g.wg_waves;
      const uint32_t ngroups = (g.nframes + W - 1) / W;
      if (task >= ngroups * g.strips) break;
      // Strip-major order: waves resident at the same time work on the SAME strip of
      // different frames, so the strip's delta bytes are shared through L1/L2.
      uint32_t group;
      if (g.strip_major) {
        strip = task / ngroups;
        group = task - strip * ngroups;
      } else {
        group = task / g.strips;
        strip = task - group * g.strips;
      }
      frame = group * W + wave;
      active = frame < g.nframes;
      if (active)
        orl = sp_delta ? run_strip<SWAP, FAST, true, LOW>(c, g, P, frame, strip, sp_cg)
                       : run_strip<SWAP, FAST, false, LOW>(c, g, P, frame, strip, sp_cg);
    }
    const StreamArgs g = launch_g();  // read again: see StreamLaunch
    const StreamPtrs P = launch_p();
    const uint32_t W = g.wg_waves;
    uint32_t next = 0;
    if (active) {
    const bool last = strip_arrive<true, FAST>(c, g, P, frame, strip, orl, wave == 0 ? &P.ctr[kCtrTask] : nullptr, &next);
    if (last) {
      // EstimateEntropy + the two decisions (cc:533, cc:564) from the strips' histograms
      const uint8_t* ph = reinterpret_cast<const uint8_t*>(P.partials + (size_t)frame * g.strips * (kHistFused * 256)) + c.lane * 8;
      uint32_t v[kHistFused][4] = {};
      for (uint32_t t = 0; t < g.strips; t++) {
#pragma unroll
        for (int h = 0; h < kHistFused; h++) {
          const uint64_t q = ld_agent64(ph + (size_t)(t * kHistFused + h) * 512);  // 4 x u16
          v[h][0] += (uint32_t)q & 0xffffu; v[h][1] += ((uint32_t)q) >> 16;
          v[h][2] += (uint32_t)(q >> 32) & 0xffffu; v[h][3] += (uint32_t)(q >> 48);
        }
      }
      // cc:533: E(countd) < E(counta); countd has everything in bin 0, E(countd) == 0
      const bool use_delta = g.with_delta && entropy_wave(v[0]) > 0;
      if (use_delta != sp_delta) {  // the CG histograms were taken under the wrong delta decision
        if (c.lane == 0) {
          P.decision[frame] = 0xff;
          P.list_a[atomicAdd(&P.ctr[kCtrListA], 1u)] = frame;
        }
      } else {
        const uint64_t ea = entropy_wave(v[1]);
        const uint64_t eb = entropy_wave(v[2]);
        const bool use_cg = eb < ea;  // cc:564
        const uint32_t dec = (use_delta ? 1u : 0u) | (use_cg ? 2u : 0u);
        if (c.lane == 0) {
          P.decision[frame] = (uint8_t)dec;
          atomicAdd(&P.ctr[kCtrVotes + dec], 1u);
          if (use_cg != sp_cg) P.list_b[atomicAdd(&P.ctr[kCtrListB], 1u)] = frame;
        }
        if (use_cg == sp_cg) finalize_frame(c, P, frame, dec);
      }
    }
    }
    if (W > 1) {  // wave 0 (always active) fetched the workgroup's next task
      uint32_t* next_slot = reinterpret_cast<uint32_t*>(smem + W * g.wave_lds);
      if (threadIdx.x == 0) *next_slot = next;
      __syncthreads();
      next = *reinterpret_cast<volatile uint32_t*>(next_slot);
      __syncthreads();
    }
    task = gridDim.x + next;
    __builtin_amdgcn_wave_barrier();
  }
#ifdef FPV_STAMPS
  if (c.lane == 0) {  // stamps leave through a buffer nothing else reads (ctr_next[32..43])
    const StreamPtrs P = launch_p();
    for (int k = 0; k < 6; k++)
      atomicAdd(reinterpret_cast<unsigned long long*>(P.ctr_next + 32) + k, c.seg[k]);
  }
#endif
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

// EstimateEntropy (cc:235-244) of one 256-bin histogram, one bin per thread.
__device__ __forceinline__ uint64_t entropy_bins(uint32_t v, uint32_t* scratch) {
  const uint32_t sum = block_sum(v, scratch);
  if (sum == 0) return 0;
  const uint32_t lsum = 31 - __clz(sum);
  const uint32_t term = v ? v * (lsum - (31 - __clz(v))) : 0;
  const uint32_t s = block_sum(term, scratch);
  return (uint64_t)1024 * s / sum;
}

// Workgroup barrier behind an LDS-only wait: vector-memory operations of the waves stay in
// flight across it (__syncthreads() would drain them).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#include "fpv_tile.inc"
#include "fpv_tile8.inc"
#ifndef FPV_TILE_FGROUP_MIN_BYTES  // delta planes beyond this many bytes: the tile pass walks groups of frames tile by tile
#define FPV_TILE_FGROUP_MIN_BYTES (8u << 20)
#endif

// ---- exact redo of mis-speculated frames (list driven) ----

constexpr int kListGrid = 256;  // workgroups of k_redo_decide

// Exact decisions of the frames whose delta decision was mis-speculated (list A;
// normally empty, the launch exits at once): one workgroup per listed frame takes
// all five histograms (both delta hypotheses) of the heuristics' samples in LDS,
// evaluates EstimateEntropy and appends the frame to list B for the redo pass.
template <bool SWAP>
__global__ __launch_bounds__(kThreads) void k_redo_decide(
    const uint16_t* __restrict__ frames, const uint8_t* __restrict__ dhigh,
    uint8_t* __restrict__ decision, uint32_t* __restrict__ ctr,
    const uint32_t* __restrict__ list_a, uint32_t* __restrict__ list_b, Geom g, int with_delta) {
  __shared__ uint32_t lh[kNumHists * 256];
  __shared__ uint32_t scratch[4];
  const uint32_t na = ctr[kCtrListA];
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
      const uint32_t dec = (use_delta ? 1u : 0u) | (use_cg ? 2u : 0u);
      decision[frame] = (uint8_t)(dec | 0x80u);  // bit 7: the low plane has to be redone as well
      atomicAdd(&ctr[kCtrVotes + dec], 1u);
      list_b[atomicAdd(&ctr[kCtrListB], 1u)] = frame;
    }
    __syncthreads();
  }
}

// The strip pass of k_stream under each listed frame's TRUE flags, and the next
// batch's speculation: the most frequent decision of this batch (all votes are
// in: they are cast by k_stream and k_redo_decide).
// Geometry of the redo pass for a list of nb frames.  Strip height: the shortest strips when
// few frames are listed (a strip is a serial task and the grid should be busy), taller ones
// while the list still gives every wave about four tasks (each strip pays a halo, a
// pipeline fill and an arrival).
__device__ __forceinline__ StreamArgs redo_args(uint32_t nb) {
  StreamArgs g = launch_g();
#pragma unroll
  for (int k = 1; k < 4; k++) {
    const uint32_t r = g.redo_rows[k];
    if (r && (uint64_t)nb * ((g.ys + r - 1) / r) >= 4ull * gridDim.x) g.strip_rows = r;
  }
  g.strips = (g.ys + g.strip_rows - 1) / g.strip_rows;
  return g;
}

template <bool SWAP, bool FAST, bool LOW>
__global__ __launch_bounds__(64, 2) void k_redo_stream(StreamLaunch) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t nb;
  {
    const StreamPtrs P = launch_p();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      uint32_t best = 0, nbest = P.ctr[kCtrVotes];
      for (uint32_t k = 1; k < 4; k++)
        if (P.ctr[kCtrVotes + k] > nbest) { best = k; nbest = P.ctr[kCtrVotes + k]; }
      if (nbest) *P.spec = (uint8_t)best;
    }
    nb = P.ctr[kCtrListB];
  }
  if (nb == 0) return;
  StreamCtx c;
  stream_ctx_init(c, redo_args(nb), smem);
#ifdef FPV_STAMPS
  for (int k = 0; k < 6; k++) c.seg[k] = 0;
  c.tlast = __builtin_amdgcn_s_memtime();
#endif
  uint32_t task = blockIdx.x;
  for (;;) {
    uint32_t frame, strip, dec, orl;
    {
      const StreamArgs g = redo_args(nb);
      const StreamPtrs P = launch_p();
      if (task >= nb * g.strips) break;
      strip = task / nb;
      frame = P.list_b[task - strip * nb];
      const uint32_t dec8 = P.decision[frame];
      dec = dec8 & 3u;
      // The low plane depends on the delta decision only: k_stream's is already right unless
      // the frame came through list A (bit 7, set by k_redo_decide).
      if (LOW && (dec8 & 0x80u))
        orl = (dec & 1u) ? run_strip<SWAP, FAST, true, LOW>(c, g, P, frame, strip, dec & 2u)
                         : run_strip<SWAP, FAST, false, LOW>(c, g, P, frame, strip, dec & 2u);
      else
        orl = (dec & 1u) ? run_strip<SWAP, FAST, true, false>(c, g, P, frame, strip, dec & 2u)
                         : run_strip<SWAP, FAST, false, false>(c, g, P, frame, strip, dec & 2u);
    }
    const StreamArgs g = redo_args(nb);  // read again: see StreamLaunch
    const StreamPtrs P = launch_p();
    uint32_t next;
    const bool last = strip_arrive<false, FAST>(c, g, P, frame, strip, orl, &P.ctr[kCtrRedoTask], &next);
    task = gridDim.x + next;
    if (last) finalize_frame(c, P, frame, dec);
    __builtin_amdgcn_wave_barrier();
  }
}

#include "fpv_inverse.inc"

}  // namespace

#ifdef FPV_EXP_PITCH
uint32_t g_exp_in_pad = 0, g_exp_out_pad = 0;
#endif

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

// Per-device multiprocessor count (the persistent grids are sized by it).
static uint32_t device_cus() {
  static std::mutex m;
  static int cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  dev &= 63;
  std::lock_guard<std::mutex> l(m);
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    int n = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 0;
    cus[dev] = n > 0 ? n : 256;
  }
  return (uint32_t)cus[dev];
}

// Workgroups per frame of k_tile_decide: one -- unless the batch is a FEW large frames (the Encoder's batches of 4096x4096 are 8
// frames), whose records (1.5 KB per tile, 795 KB per frame) would each go through ONE compute unit in six turns of loads while
// the others idle: then a power of two of them, as long as every part has records for one turn of its loads and a compute
// unit of its own (beyond that the kernel moves the same bytes in the same time: 60 frames x 8 parts 25.3 us against 24.4,
// 113 x 4096x2160 x 4 parts 22.1 against 17.6; profiles/r05_forward_ab.log).
// FPV_DECIDE_PARTS=n (tests, A/B runs) asks for n whatever the size; read per launch.
static uint32_t decide_parts(uint32_t nframes, uint32_t tiles, bool big) {
  uint32_t want = 0;
  if (const char* e = getenv("FPV_DECIDE_PARTS")) want = (uint32_t)atoi(e);
  uint32_t parts = 1;
  if (want) {
    while (parts * 2 <= want && parts < 8) parts *= 2;
  } else if (big) {
    while (parts < 8 && tiles / (parts * 2) >= 40 && (uint64_t)nframes * parts * 2 <= device_cus()) parts *= 2;
  }
  while (parts > 1 && (uint64_t)nframes * parts > kDecideSlots) parts /= 2;
  return parts;
}

constexpr uint32_t kMaxWaveLds = 160 * 1024 - 1024;  // one wave may own (nearly) the whole LDS of a CU

bool make_geom(uint32_t xs, uint32_t ys, int shift, int big_endian, Geom* g,
               const char** why) {
  static const char* kDims = "xsize and ysize must be non-zero multiples of 4";
  static const char* kShift = "shift must be 0..8 (0..15 for little-endian input)";
  static const char* kWide = "xsize too large for the LDS line buffer of the forward kernel (about 32K pixels)";
  static const char* kBig = "frame larger than 2^31 pixels";
  if (!xs || !ys || (xs & 3) || (ys & 3)) { *why = kDims; return false; }
  if (shift < 0 || shift > 15 || (big_endian && shift > 8)) { *why = kShift; return false; }
  if ((uint64_t)xs * ys >= (1ull << 31)) { *why = kBig; return false; }
  *g = Geom();
  g->xs = xs; g->ys = ys; g->npix = xs * ys;
  g->pw = xs / 4; g->ph = ys / 4; g->npv = g->pw * g->ph;
  g->shift = shift; g->swap = big_endian ? 1 : 0;
  // streaming kernel: strips of about 200K pixels (whole preview rows), one wave each
  {
    // Prefer a strip height that divides ys and whose pixel count is a multiple
    // of the 1024-pixel step (the FAST kernel needs both); about 200K pixels
    // (measured on 1280x800 with the per-strip hand-off of round 2: 160 and 200 rows
    // beat 80 and 100 by 1.5 %, 400 rows lose again to the coarser load balance).
    const double kTarget = 204800.0;
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
    if ((uint64_t)srows * xs > 15u * 65000u) { *why = kWide; return false; }
    g->strip_rows = srows;
    g->strips = (ys + srows - 1) / srows;
    g->ring_base = round_up(xs + 16, 16);
    uint32_t rf = 1;
    while (rf < g->ring_base + kStep) rf <<= 1;
    g->ring_fwd = rf;
    // preview accumulator rows: every 4-row group that kDepth steps (one flush interval) can touch
    g->nacc = (kDepth * kStep + 4 * xs - 1) / (4 * xs) + 1;
    const bool fast_shape = (xs & 63) == 0 && xs >= 256;
    if (fast_shape) g->nacc = g->nacc <= 2 ? 2 : g->nacc <= 4 ? 4 : 8;  // FAST indexes them with a mask
    g->pwr = round_up(g->pw, 16);
    g->pvb_stride = 2 * g->pwr + 16;
    // ring | preview accumulators | histograms | previous raw preview row
    g->wave_lds = round_up(rf + g->nacc * g->pw * 4 + kHistFused * 256 * 4 + g->pwr, 16);
    if (g->wave_lds > kMaxWaveLds) { *why = kWide; return false; }
    uint32_t wpc = (160 * 1024 - 1024) / g->wave_lds;
    g->stream_waves_per_cu = std::max(1u, std::min(8u, wpc));  // measured: 8 beats 12 beats 16 (memory-system contention)
    if (const char* e = getenv("FPV_WAVES_PER_CU")) g->stream_waves_per_cu = std::max(1u, std::min(wpc, (uint32_t)atoi(e)));
    // waves per workgroup: a power of two that divides the waves per CU
    g->wg_waves = 1;
    if (const char* e = getenv("FPV_WG_WAVES")) {
      const uint32_t want = (uint32_t)atoi(e);
      for (uint32_t w = 1; w <= 8 && w <= want; w <<= 1)
        if (g->stream_waves_per_cu % w == 0) g->wg_waves = w;
    }
    // FAST: every step of every strip is full, a 16-pixel segment never straddles image
    // rows, a preview row is a whole number of 16-byte groups
    g->fast = fast_shape && ((uint64_t)g->strip_rows * xs) % kStep == 0 && ys % g->strip_rows == 0;
    // The redo pass of mis-speculated frames cuts a frame into short strips (about 8K
    // pixels): usually only a few frames are listed, and a strip is a serial task.
    g->redo_strip_rows = g->strip_rows;
    for (uint32_t r = 4; r < g->strip_rows; r += 4) {
      if ((uint64_t)r * xs < 8192) continue;
      if (g->fast && (ys % r || ((uint64_t)r * xs) % kStep)) continue;
      g->redo_strip_rows = r;
      break;
    }
    g->redo_strips = (ys + g->redo_strip_rows - 1) / g->redo_strip_rows;
    // taller choices for long lists: about 2x, 4x, 8x the shortest, below the main strip height
    g->redo_rows[0] = g->redo_strip_rows;
    for (int k = 1; k < 4; k++) {
      g->redo_rows[k] = 0;
      const uint32_t prev = g->redo_rows[k - 1];
      if (!prev) continue;
      for (uint32_t r = (prev * 2 + 3) & ~3u; r < g->strip_rows; r += 4) {
        if (g->fast && (ys % r || ((uint64_t)r * xs) % kStep)) continue;
        g->redo_rows[k] = r;
        break;
      }
    }
  }
  // Tile pass (fpv_tile.inc): a workgroup per 4 * G rows, a thread per 16-pixel column segment of
  // a 4-row group (the row above a tile is recomputed: 1 / (4 G) more loads).  Widths that are
  // not multiples of 64 or beyond 16 K pixels keep the streaming kernels.
  {
    // (Column-block tiles -- half the width, twice the rows, what lies left of a block fetched from global
    // memory -- were built for wide frames, where a whole-width tile of a 512-thread workgroup is 8 rows and one
    // row in eight a halo row; bit-exact, and no faster at 4096x4096 (0.9405 against 0.9408 ms per 60 frames:
    // what those frames wait for is the delta frame, see tile_lfgroup), 3-5 % slower on narrower frames, and
    // their hooks cost the default shape 0.35 %: taken out again, commit 16feb86 has them.)
    const uint32_t ncs = xs / 16;
    // Workgroup sizes that fill the CU's wave slots evenly at this kernel's register budget (16
    // waves per CU): 512 threads where that gives at least 2 groups, else 1024 (measured at
    // 1280x800: G = 6 / 512 threads 0.72 ms, G = 4 / 320 threads 0.88, G = 8 / 640 threads 1.0;
    // at 4096x4096: G = 2 / 512 threads 0.895 ms per 60 frames, G = 4 / 1024 threads 0.965).
    uint32_t G = ncs ? 512u / ncs : 0;
    if (G < 2 && ncs) G = 1024u / ncs;
    if (const char* e = getenv("FPV_TILE_GROUPS")) G = std::max(1u, (uint32_t)atoi(e));
    if (ncs && G * ncs > 1024) G = 1024 / ncs;
    G = std::min(G, ys / 4);
    const char* fwd = getenv("FPV_FORWARD");  // "stream": round 2's kernels (A/B runs, tests)
    g->tile_ok = 0;
    if ((xs & 63) == 0 && ncs <= 1024 && G >= 1 && !(fwd && fwd[0] == 's')) {
      const uint32_t T = std::max(256u, round_up(G * ncs, 64));  // the finishing path wants 256 threads
      const uint32_t tiles = (ys + 4 * G - 1) / (4 * G);
      if (T <= 1024 && tiles <= 65535) {
        g->tile_ok = 1;
        g->tile_groups = G;
        g->tile_threads = T;
        g->tile_tiles = tiles;
        // Large frames: the batch is walked in groups of 64 frames, TILE BY TILE inside a group (the frame of the
        // group fastest), so that a tile's delta bytes are fetched once per group.  Walked frame by frame -- the
        // memory order, best while the delta planes (2 bytes per pixel) stay in the L2s -- a 4096x4096 batch
        // re-reads its 33.5 MB of delta planes for every frame.  Same process (profiles/r04_forward_sizes.log):
        // 4096x4096 0.939 -> 0.774 ms per 60 frames, 1.929 -> 1.594 per 125; 4096x2160 0.930 -> 0.835; 2560x2048
        // (10.5 MB) 0.908 -> 0.802; but 2048x1536 (6.3 MB) 0.756 -> 0.788 and 1920x1080 0.761 -> 0.778: the
        // threshold lies between.  Groups of 32 ... 256 frames are within 3 % of each other.
        g->tile_lfgroup = 2ull * xs * ys > FPV_TILE_FGROUP_MIN_BYTES ? 6u : 0u;
        if (const char* e = getenv("FPV_TILE_FGROUP")) g->tile_lfgroup = std::min(10u, (uint32_t)atoi(e));
        g->tile_lfgroup_fixed = getenv("FPV_TILE_FGROUP") != nullptr;
        g->tile_lds = tile_lds_bytes(G, ncs);
        g->tile_decide_big = getenv("FPV_DECIDE_BIG") != nullptr;
      }
    }
  }
  // Round 4's tile pass (fpv_tile8.inc): a thread per 8-pixel column segment of an R-row group,
  // a workgroup per R * G rows.  R = 4 (62 VGPRs: eight waves per SIMD) and workgroups of 1024
  // threads, two per CU (measured at 1280x800, same process: R = 4 / G = 6 0.749 ms, R = 4 / G = 3
  // 0.782 -- twice the halo rows and records --, R = 8 / G = 3 0.788, R = 8 / G = 6 0.966: one
  // workgroup per CU; profiles/r04_forward_ab.log).
  {
    const uint32_t ncs = xs / 8;
    uint32_t R = 4;
    if (const char* e = getenv("FPV_TILE8_ROWS")) R = atoi(e) == 8 ? 8u : 4u;
    uint32_t G = ncs ? 1024u / ncs : 0;
    if (const char* e = getenv("FPV_TILE8_GROUPS")) G = std::max(1u, (uint32_t)atoi(e));
    if (ncs && G * ncs > 1024) G = 1024 / ncs;
    G = std::min(G, (ys + R - 1) / R);
    // FPV_FORWARD=tile8 selects this pass (A/B runs, tests); the default stays round 3's tile pass:
    // measured in one process on six geometries the two are within 2 % of each other, each ahead
    // somewhere (profiles/r04_forward_sizes.log) -- every pixel line was requested once before, too.
    const char* fwd = getenv("FPV_FORWARD");
    g->tile8_ok = 0;
    if (g->tile_ok && G >= 1 && fwd && !strcmp(fwd, "tile8")) {
      const uint32_t T = std::max(256u, round_up(G * ncs, 256));
      const uint32_t tiles = (ys + R * G - 1) / (R * G);
      const uint32_t lds = tile8_lds_bytes(R, G, ncs, T);
      if (T <= 1024 && tiles <= 65535 && lds <= 64 * 1024) {
        g->tile8_ok = 1;
        g->tile8_rows = R;
        g->tile8_groups = G;
        g->tile8_threads = T;
        g->tile8_tiles = tiles;
        g->tile8_lds = lds;
        g->tile8_frame_big = getenv("FPV_FRAME8_BIG") != nullptr;  // experiments / tests: 1024 threads whatever the size
      }
    }
  }
  g->nd = (g->npix + 14) / 15;
  g->ncg = g->npix > xs + 1 ? (g->npix - xs - 1 + 30) / 31 : 0;
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

// Workspace layout: [counter blocks | anylow | frame_done | bdone] are cleared ONCE when
// the workspace is created; afterwards the kernels leave them usable (the counter blocks
// alternate between batches, the wave that finalises a frame resets its two words, the
// boundary words only ever matter by their parity and every pass adds two).
// The rest is fully written by the kernels.  `spec` survives batches.
size_t fwd_workspace_bytes(const Geom& g, uint32_t max_frames) {
  size_t b = 512 + 2 * (size_t)round_up(max_frames * 4, 256);    // counter blocks, anylow, frame_done
  b += round_up((size_t)max_frames * g.redo_strips * 4, 256);    // boundary words
  if (g.tile_ok) {
    b += round_up((size_t)max_frames * std::max(g.tile_tiles, g.tile8_ok ? g.tile8_tiles : 0u) * kTileRecBytes, 256);  // tile records
    b += round_up((size_t)max_frames * g.npv, 256);              // raw previews
    b += round_up(max_frames * 4, 256);                          // per-position prediction
    b += round_up((size_t)kDecideSlots * kDecidePartWords * 4, 256);  // k_tile_decide's partial sums
  }
  b += round_up(max_frames, 256);                                // decision
  b += 2 * (size_t)round_up(max_frames * 4, 256);                // list A, list B
  b += round_up((size_t)max_frames * g.redo_strips * g.pvb_stride, 256);  // boundary preview rows (redo_strips >= strips)
  b += (size_t)max_frames * g.strips * kHistFused * 256 * 2;     // partial histograms (u16)
  return b + 256 /* spec */ + 256;
}

void carve_fwd_workspace(const Geom& g, uint32_t max_frames, void* base,
                         FwdWorkspace* ws) {
  uint8_t* p = static_cast<uint8_t*>(base);
  ws->ctr = reinterpret_cast<uint32_t*>(p);
  p += 512;
  ws->anylow = reinterpret_cast<uint32_t*>(p);
  p += round_up(max_frames * 4, 256);
  ws->frame_done = reinterpret_cast<uint32_t*>(p);
  p += round_up(max_frames * 4, 256);
  ws->bdone = reinterpret_cast<uint32_t*>(p);
  p += round_up((size_t)max_frames * g.redo_strips * 4, 256);
  ws->trec = nullptr; ws->pvraw = nullptr; ws->spec_pos = nullptr; ws->dpart = nullptr;
  ws->nframes = max_frames;
  ws->zero_bytes = (size_t)(p - static_cast<uint8_t*>(base));
  if (g.tile_ok) {
    ws->trec = p;
    p += round_up((size_t)max_frames * std::max(g.tile_tiles, g.tile8_ok ? g.tile8_tiles : 0u) * kTileRecBytes, 256);
    ws->pvraw = p;
    p += round_up((size_t)max_frames * g.npv, 256);
    ws->spec_pos = reinterpret_cast<uint32_t*>(p);
    p += round_up(max_frames * 4, 256);
    ws->dpart = reinterpret_cast<uint32_t*>(p);
    p += round_up((size_t)kDecideSlots * kDecidePartWords * 4, 256);
  }
  ws->decision = p;
  p += round_up(max_frames, 256);
  ws->list_a = reinterpret_cast<uint32_t*>(p);
  p += round_up(max_frames * 4, 256);
  ws->list_b = reinterpret_cast<uint32_t*>(p);
  p += round_up(max_frames * 4, 256);
  ws->pvb = p;
  p += round_up((size_t)max_frames * g.redo_strips * g.pvb_stride, 256);
  ws->partials = reinterpret_cast<uint16_t*>(p);
  p += (size_t)max_frames * g.strips * kHistFused * 256 * 2;
  ws->spec = p;
  ws->seq = 0;
}

hipError_t init_fwd_workspace(const FwdWorkspace& ws, hipStream_t s) {
  hipError_t e = hipMemsetAsync(ws.ctr, 0, ws.zero_bytes, s);
  if (e != hipSuccess) return e;
  if (ws.spec_pos) {  // no position has a history yet: the majority guess applies
    e = hipMemsetAsync(ws.spec_pos, 3, (size_t)ws.nframes * 4, s);  // dwords 0x03030303: decision 3, "repeated" clear
    if (e != hipSuccess) return e;
  }
  return hipMemsetAsync(ws.spec, 3, 4, s);  // first guess: delta + CG
}

hipError_t launch_copy(const void* src, void* dst, size_t bytes, hipStream_t s, int shape) {
  const size_t n16 = bytes / 16;
  if (shape == 1) {
    if (n16) k_copy_simple<<<(unsigned)((n16 + kThreads - 1) / kThreads), kThreads, 0, s>>>(static_cast<const u32x4*>(src), static_cast<u32x4*>(dst), n16);
  } else {
    k_copy<<<256 * 32, kThreads, 0, s>>>(static_cast<const u32x4*>(src), static_cast<u32x4*>(dst), n16);
  }
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
  dim3 grid(((g.npix + 3) / 4 + kThreads - 1) / kThreads);  // a thread per started group of 4 pixels
  k_split_decoded<<<grid, kThreads, 0, s>>>(d_img, d_high, d_low, g.npix);
  return hipGetLastError();
}

// Function attributes (dynamic LDS above 64 KiB) of every kernel that needs them, once per
// device.  Called by the launch functions and by fpvhip_warmup (so that a first session
// does not pay for it).
void prepare_kernels() {
  static std::mutex attr_m;
  static uint64_t attr_done = 0;
  once_per_device(attr_m, attr_done, [] {  // wide frames need more than 64 KiB of LDS per wave
#define FPV_SF(K, W, F, L) reinterpret_cast<const void*>(&K<W, F, L>)
    const void* sf[16] = {
        FPV_SF(k_stream, true, true, true),        FPV_SF(k_stream, true, true, false),
        FPV_SF(k_stream, true, false, true),       FPV_SF(k_stream, true, false, false),
        FPV_SF(k_stream, false, true, true),       FPV_SF(k_stream, false, true, false),
        FPV_SF(k_stream, false, false, true),      FPV_SF(k_stream, false, false, false),
        FPV_SF(k_redo_stream, true, true, true),   FPV_SF(k_redo_stream, true, true, false),
        FPV_SF(k_redo_stream, true, false, true),  FPV_SF(k_redo_stream, true, false, false),
        FPV_SF(k_redo_stream, false, true, true),  FPV_SF(k_redo_stream, false, true, false),
        FPV_SF(k_redo_stream, false, false, true), FPV_SF(k_redo_stream, false, false, false)};
#undef FPV_SF
    for (const void* f : sf)
      hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxWaveLds);
  
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_inverse<true>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_inverse<false>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_inverse_fast),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    });
}

template <bool SWAP>
static hipError_t predict_impl(const Geom& g, const uint16_t* d_frames,
                               uint32_t nframes, const uint8_t* d_dhigh,
                               const uint8_t* d_dlow, FwdWorkspace& ws,
                               uint8_t* d_high, uint8_t* d_low,
                               uint8_t* d_preview, uint8_t* d_flags,
                               hipStream_t s, hipEvent_t* ev) {
  const int with_delta = d_dhigh != nullptr;
  if (g.tile8_ok && ws.trec) {  // dispatch-ordered tiles, 8 pixels per lane (fpv_tile8.inc)
    if ((uint64_t)nframes * g.tile8_tiles > 0x7fffffffull) return hipErrorInvalidValue;  // one workgroup per tile
    TileLaunch L;
    TileArgs& a = L.g;
    a.xs = g.xs; a.ys = g.ys; a.npix = g.npix; a.pw = g.pw; a.npv = g.npv;
    a.lfgroup = 0;
    a.ncs = g.xs / 8; a.groups = g.tile8_groups; a.tiles = g.tile8_tiles; a.nitems = a.groups * a.ncs;
    a.ncs_rcp = ((1u << 20) + a.ncs - 1u) / a.ncs;
    a.shift = (uint32_t)g.shift; a.nframes = nframes; a.with_delta = (uint32_t)with_delta;
    a.in_pitch = a.out_pitch = g.npix;
    a.xs15 = g.xs % 15u; a.xs31 = g.xs % 31u; a.c31 = (g.xs + 1u) % 31u; a.ncg = g.ncg;
    a.dparts = 1u; a.xband = 0u;
    TilePtrs& P = L.P;
    P.frames = d_frames; P.dhigh = d_dhigh; P.dlow = d_dlow;
    P.high = d_high; P.low = d_low; P.preview = d_preview; P.flags = d_flags;
    P.trec = ws.trec; P.pvraw = ws.pvraw;
    P.ctr = ws.ctr + (ws.seq & 1u) * 64;
    P.ctr_next = ws.ctr + ((ws.seq + 1) & 1u) * 64;
    ws.seq++;
    P.decision = ws.decision; P.list_a = ws.list_a; P.list_b = ws.list_b; P.spec = reinterpret_cast<uint32_t*>(ws.spec); P.spec_pos = ws.spec_pos;
    const bool lowp = g.shift != 8;
    const uint32_t T = g.tile8_threads, lds = g.tile8_lds, grid = nframes * g.tile8_tiles;
    const uint32_t rgrid = std::min<uint32_t>(grid, device_cus() * std::max(1u, 2048u / T));
    if (ev) hipEventRecord(ev[0], s);
#define FPV_T8(K, GRID) \
    do { \
      if (g.tile8_rows == 8) { if (lowp) K<SWAP, true, 8><<<GRID, T, lds, s>>>(L); else K<SWAP, false, 8><<<GRID, T, lds, s>>>(L); } \
      else { if (lowp) K<SWAP, true, 4><<<GRID, T, lds, s>>>(L); else K<SWAP, false, 4><<<GRID, T, lds, s>>>(L); } \
    } while (0)
    FPV_T8(k_tile8, dim3(g.tile8_tiles, nframes));  // x fastest = memory order
    if (ev) hipEventRecord(ev[1], s);
    const bool frame_big = g.tile8_frame_big;  // FPV_FRAME8_BIG (experiments / tests): 1024 threads whatever the size
    if (g.tile8_tiles > 64 || frame_big) k_frame8<SWAP, true><<<nframes, 1024, 0, s>>>(L);   // many records per frame (large frames)
    else k_frame8<SWAP, false><<<nframes, 256, 0, s>>>(L);
    if (ev) hipEventRecord(ev[2], s);
    FPV_T8(k_redo8, rgrid);
#undef FPV_T8
    if (ev) hipEventRecord(ev[3], s);
    if (ev) hipEventRecord(ev[4], s);
    return hipGetLastError();
  }
  if (g.tile_ok && ws.trec) {  // round 3's tiles, 16 pixels per lane (fpv_tile.inc)
    if ((uint64_t)nframes * g.tile_tiles > 0x7fffffffull) return hipErrorInvalidValue;  // one workgroup per tile
    TileLaunch L;
    TileArgs& a = L.g;
    a.xs = g.xs; a.ys = g.ys; a.npix = g.npix; a.pw = g.pw; a.npv = g.npv;
    a.lfgroup = g.tile_lfgroup;
    if (a.lfgroup && !g.tile_lfgroup_fixed)  // no larger than the batch needs (a group's unused frame slots are workgroups that leave at once)
      while (a.lfgroup > 1 && (1u << (a.lfgroup - 1)) >= nframes) a.lfgroup--;  // (the Encoder's batches of such frames are 8 frames)
    a.ncs = g.xs / 16; a.groups = g.tile_groups; a.tiles = g.tile_tiles; a.nitems = a.groups * a.ncs;
    a.ncs_rcp = ((1u << 20) + a.ncs - 1u) / a.ncs;
    a.shift = (uint32_t)g.shift; a.nframes = nframes; a.with_delta = (uint32_t)with_delta;
    a.in_pitch = a.out_pitch = g.npix;
    a.xs15 = g.xs % 15u; a.xs31 = g.xs % 31u; a.c31 = (g.xs + 1u) % 31u; a.ncg = g.ncg;
    TilePtrs& P = L.P;
    P.frames = d_frames; P.dhigh = d_dhigh; P.dlow = d_dlow;
    P.high = d_high; P.low = d_low; P.preview = d_preview; P.flags = d_flags;
    P.trec = ws.trec; P.pvraw = ws.pvraw;
    P.ctr = ws.ctr + (ws.seq & 1u) * 64;
    P.ctr_next = ws.ctr + ((ws.seq + 1) & 1u) * 64;
    ws.seq++;
    P.decision = ws.decision; P.list_a = ws.list_a; P.list_b = ws.list_b; P.spec = reinterpret_cast<uint32_t*>(ws.spec); P.spec_pos = ws.spec_pos;
    P.dpart = ws.dpart; P.dtick = ws.frame_done;  // (frame_done: zero since the workspace was made, unused by the tile pass otherwise)
    const bool dbig = g.tile_tiles > 64 || g.tile_decide_big;
    a.dparts = decide_parts(nframes, g.tile_tiles, dbig);
    const bool lowp = g.shift != 8;
    const uint32_t T = g.tile_threads, lds = g.tile_lds;
    if (ev) hipEventRecord(ev[0], s);
#if FPV_TILE_XGROUP > 1  // experiment: a flat grid
    const dim3 tgrid(nframes * g.tile_tiles);
#else
    // Small batches of large frames (the Encoder hands over 8 frames of 4096x4096 at a time): with the frames of a group spread
    // over the eight XCDs, every XCD fetches every tile's delta bytes -- as many bytes again as the frames' own.  Then XCD c (the
    // workgroups go to them round robin) walks its own eighth of the tiles, for all frames of the group: each delta byte is
    // fetched once.  4096x4096: 1 frame -11 %, 4 frames -25 %, 8 frames -8 ... -21 % by size, 16 frames -2.5 %; above that it
    // depends on the batch size and the box (32 frames +6 %, 60 frames -3.5 / +2.6 %, 125 frames +1.4 ... +2.5 %): not used there
    // (profiles/r05_forward_ab.log).  FPV_TILE_XBAND=0 / 1 (read per launch): never / whenever the frames go in groups.
    a.xband = a.lfgroup && nframes <= 16u ? 1u : 0u;
    if (const char* e = getenv("FPV_TILE_XBAND")) a.xband = a.lfgroup && atoi(e) ? 1u : 0u;
    const dim3 tgrid((a.xband ? 8u * ((g.tile_tiles + 7u) / 8u) : g.tile_tiles) << a.lfgroup, (nframes + (1u << a.lfgroup) - 1u) >> a.lfgroup);  // x fastest; nframes <= 65535 (create_impl)
#endif
    if (lowp) k_tile<SWAP, true><<<tgrid, T, lds, s>>>(L);
    else k_tile<SWAP, false><<<tgrid, T, lds, s>>>(L);
    if (ev) hipEventRecord(ev[1], s);
    if (dbig) k_tile_decide<SWAP, true><<<nframes * a.dparts, 1024, 0, s>>>(L);   // many records per frame (large frames)
    else k_tile_decide<SWAP, false><<<nframes * a.dparts, 256, 0, s>>>(L);
    if (ev) hipEventRecord(ev[2], s);
    // as many workgroups as are resident at once (two of 512 threads per CU at this kernel's registers): each takes its
    // share of the frames' previews in one turn, then its share of the redo list
    const uint32_t rgrid = std::min<uint32_t>(nframes * g.tile_tiles, device_cus() * std::max(1u, 1024u / T));
    if (lowp) k_tile_finish<SWAP, true><<<rgrid, T, lds, s>>>(L);
    else k_tile_finish<SWAP, false><<<rgrid, T, lds, s>>>(L);
    if (ev) hipEventRecord(ev[3], s);
    if (ev) hipEventRecord(ev[4], s);
    return hipGetLastError();
  }
  StreamArgs a;
  a.xs = g.xs; a.ys = g.ys; a.npix = g.npix; a.pw = g.pw; a.npv = g.npv;
  a.in_pitch = a.out_pitch = g.npix;
#ifdef FPV_EXP_PITCH  // experiments only (tools/exp/pitch.py): frames / planes at a padded pitch
  a.in_pitch += g_exp_in_pad;
  a.out_pitch += g_exp_out_pad;
#endif
  a.strip_rows = g.strip_rows; a.strips = g.strips; a.ring_fwd = g.ring_fwd;
  a.ring_base = g.ring_base; a.nacc = g.nacc; a.shift = (uint32_t)g.shift; a.nframes = nframes;
  static const int strip_major = getenv("FPV_STRIP_MAJOR") ? atoi(getenv("FPV_STRIP_MAJOR")) : 1;
  a.strip_major = (uint32_t)strip_major;
  a.pwr = g.pwr; a.pvb_stride = g.pvb_stride; a.bd_stride = g.redo_strips; a.with_delta = (uint32_t)with_delta;
  StreamPtrs P;
  P.frames = d_frames; P.dhigh = d_dhigh; P.dlow = d_dlow;
  P.high = d_high; P.low = d_low; P.preview = d_preview; P.flags = d_flags;
  P.anylow = ws.anylow; P.frame_done = ws.frame_done; P.bdone = ws.bdone; P.partials = ws.partials; P.pvb = ws.pvb;
  P.ctr = ws.ctr + (ws.seq & 1u) * 64;
  P.ctr_next = ws.ctr + ((ws.seq + 1) & 1u) * 64;
  ws.seq++;
  P.decision = ws.decision; P.list_a = ws.list_a; P.list_b = ws.list_b; P.spec = ws.spec;
  const uint32_t waves = device_cus() * g.stream_waves_per_cu;
  const uint32_t W = g.wg_waves;
  a.wave_lds = g.wave_lds; a.wg_waves = W;
  const uint32_t grid = std::min<uint32_t>((nframes + W - 1) / W * g.strips, waves / W);
  const bool lowp = g.shift != 8;
  if (ev) hipEventRecord(ev[0], s);
#define FPV_LAUNCH(K, GRID, WAVES)                                                                    \
  do {                                                                                                \
    const uint32_t lds_ = (WAVES) * g.wave_lds + 16;                                                  \
    if (g.fast) {                                                                                     \
      if (lowp) K<SWAP, true, true><<<GRID, 64 * (WAVES), lds_, s>>>(StreamLaunch{P, a});                           \
      else K<SWAP, true, false><<<GRID, 64 * (WAVES), lds_, s>>>(StreamLaunch{P, a});                               \
    } else {                                                                                          \
      if (lowp) K<SWAP, false, true><<<GRID, 64 * (WAVES), lds_, s>>>(StreamLaunch{P, a});                          \
      else K<SWAP, false, false><<<GRID, 64 * (WAVES), lds_, s>>>(StreamLaunch{P, a});                              \
    }                                                                                                 \
  } while (0)
  FPV_LAUNCH(k_stream, grid, W);
  if (ev) hipEventRecord(ev[1], s);
  // exact redo of mis-speculated frames; both launches exit at once when the lists are empty
  k_redo_decide<SWAP><<<kListGrid, kThreads, 0, s>>>(d_frames, d_dhigh, ws.decision, P.ctr,
                                                    ws.list_a, ws.list_b, g, with_delta);
  if (ev) hipEventRecord(ev[2], s);
  a.strip_rows = g.redo_strip_rows;  // k_redo_stream may pick a taller entry of redo_rows
  a.strips = g.redo_strips;
  for (int k = 0; k < 4; k++) a.redo_rows[k] = g.redo_rows[k];
  a.wg_waves = 1;
  FPV_LAUNCH(k_redo_stream, waves, 1u);
#undef FPV_LAUNCH
  if (ev) hipEventRecord(ev[3], s);
  if (ev) hipEventRecord(ev[4], s);
  return hipGetLastError();
}

#ifdef FPV_EXP_PITCH
}  // namespace fpvhip
extern "C" void fpvhip_exp_set_pad(uint32_t in_pad_px, uint32_t out_pad_bytes) {
  fpvhip::g_exp_in_pad = in_pad_px;
  fpvhip::g_exp_out_pad = out_pad_bytes;
}
namespace fpvhip {
#endif

hipError_t launch_predict(const Geom& g, const uint16_t* d_frames,
                          uint32_t nframes, const uint8_t* d_dhigh,
                          const uint8_t* d_dlow, FwdWorkspace& ws,
                          uint8_t* d_high, uint8_t* d_low, uint8_t* d_preview,
                          uint8_t* d_flags, hipStream_t s, hipEvent_t* ev) {
  prepare_kernels();
  if (g.swap)
    return predict_impl<true>(g, d_frames, nframes, d_dhigh, d_dlow, ws, d_high,
                              d_low, d_preview, d_flags, s, ev);
  return predict_impl<false>(g, d_frames, nframes, d_dhigh, d_dlow, ws, d_high,
                             d_low, d_preview, d_flags, s, ev);
}

// k_inverse_trio<K4, M>: supersteps per barrier (the waves of a frame meet once per block of M
// supersteps), ring and LDS bytes.  The ring holds the block being written, the block being
// emitted and the row above the first superstep of either.
static uint32_t trio_block_max() {  // FPV_TRIO_BLOCK caps it; read per launch: A/B runs switch it inside one process
  const char* e = getenv("FPV_TRIO_BLOCK");
  const int v = e ? atoi(e) : 4;
  return v >= 4 ? 4u : v >= 2 ? 2u : 1u;
}
// ip (k_inverse_trio<K4, M, true>): residuals in place, two blocks ahead.  The ring is a whole number of supersteps: the
// block being emitted -- or, where a row is longer than a block, the supersteps that hold the row above the current block --,
// the current block and two blocks of residuals ahead of it.
static bool trio_ip_ok(const Geom& g, uint32_t k4) {  // a lane's north bytes must not straddle the ring's end
  const char* e = getenv("FPV_TRIO_INPLACE");  // read per launch: A/B runs and tests switch inside one process
  return !(e && atoi(e) == 0) && k4 && g.xs % (4 * k4) == 0;
}
// The register-north form (k_inverse_trio<K4, M, true, m>, m >= 1): the row is m supersteps long, so the bytes above a lane's
// segment are pixels the same lane made m supersteps earlier -- they stay in registers (m + 1 sets taking turns).  Taken for
// batches of up to 2.25 frames per CU -- where the chain wave is what a frame waits for: 1280 wide 32 ... 512 frames -7 ... -10 %,
// 576 frames -3 %; from 640 frames on the batch is bound by HBM and the form costs 0 ... 1 % (profiles/r05_inverse_ab.log).
// FPV_TRIO_RES (read per launch: A/B runs, tests): 0 never, 1 whatever the batch size.
// trio_res_rows: supersteps per row where the form applies (0: it does not) -- rows of one superstep in blocks of an even
// length; rows of 2 ... 4 supersteps at segments of 20 or 24 pixels in blocks of four (100 registers of sets at most); and the
// segment lengths beyond the other forms' limit (32 / 40 / 48 pixels: trio_wide_rows) in blocks of two.
static uint32_t trio_res_rows(const Geom& g, uint32_t k4, uint32_t mblk) {
  const uint32_t S = 256 * k4;
  if (g.xs % S) return 0;
  const uint32_t m = g.xs / S;
  if (k4 > kSpecMaxK4) return m <= trio_wide_rows(k4) && mblk == 2 ? m : 0u;   // 32 / 40 / 48-pixel segments: this form only, blocks of two
  if (m == 1) return (mblk & 1) ? 0u : 1u;
  return m <= 4 && k4 >= 5 && mblk == 4 && (m + 1) * 4 * k4 <= 100 ? m : 0u;
}
static uint32_t trio_res_ok(const Geom& g, uint32_t k4, uint32_t mblk, uint32_t nframes) {
  const char* e = getenv("FPV_TRIO_RES");
  if (e && atoi(e) == 0) return 0;
  const uint32_t m = trio_res_rows(g, k4, mblk);
  if (!m) return 0;
  return e || 4ull * nframes <= 9ull * device_cus() ? m : 0u;
}
// The instantiation for (in place?, register-north form: supersteps per row, 0 = not)
template <int K4, int MB>
static auto trio_kernel(bool ip, uint32_t res, bool m2 = false) -> decltype(&k_inverse_trio<K4, MB, true, 1>) {
  // the memo behind the second pass (M2: its own instantiations, fpv_inverse.inc): the register-north forms of one superstep per row
  // in blocks of four, and the long segments' -- the batches the chain wave bounds
  if constexpr (K4 >= 4 && K4 <= (int)kSpecMaxK4 && (MB == 2 || MB == 4)) {   // ... and the ring form of the usual segment lengths (full batches, 1920 wide)
    if (m2 && ip && res == 0) return &k_inverse_trio<K4, MB, true, 0, true>;
  }
  if constexpr ((K4 > (int)kSpecMaxK4 && MB == 2) || (K4 <= (int)kSpecMaxK4 && MB == 4)) {
    if (m2 && ip && res == 1) return &k_inverse_trio<K4, MB, true, 1, true>;
    if constexpr (K4 > (int)kSpecMaxK4 && trio_wide_rows(K4) >= 2) {
      if (m2 && ip && res == 2) return &k_inverse_trio<K4, MB, true, 2, true>;
    }
  }
  if constexpr (K4 > (int)kSpecMaxK4) {   // the register-north forms only
    if constexpr (trio_wide_rows(K4) >= 2) {
      if (res == 2) return &k_inverse_trio<K4, MB, true, 2>;
    }
    if constexpr (trio_wide_rows(K4) >= 3) {
      if (res == 3) return &k_inverse_trio<K4, MB, true, 3>;
    }
    return &k_inverse_trio<K4, MB, true, 1>;
  } else {
    if (ip) {
      if constexpr (MB % 2 == 0) {
        if (res == 1) return &k_inverse_trio<K4, MB, true, 1>;
      }
      if constexpr (MB == 4 && K4 >= 5) {
        if (res == 2) return &k_inverse_trio<K4, MB, true, 2>;
        if constexpr (4 * 4 * K4 <= 100) {
          if (res == 3) return &k_inverse_trio<K4, MB, true, 3>;
        }
        if constexpr (5 * 4 * K4 <= 100) {
          if (res == 4) return &k_inverse_trio<K4, MB, true, 4>;
        }
      }
      return &k_inverse_trio<K4, MB, true, 0>;
    }
    return &k_inverse_trio<K4, MB, false, 0>;
  }
}
static uint32_t trio_ring(const Geom& g, uint32_t k4, uint32_t mblk, bool ip = false) {
  const uint32_t S = 256 * k4;
  if (ip) return (3 * mblk + std::max(mblk, (g.xs + 16 + S - 1) / S)) * S;
  uint32_t r = 1;
  while (r < 2 * mblk * S + g.xs + 16) r <<= 1;
  return r;
}
static size_t trio_lds(const Geom& g, uint32_t k4, uint32_t mblk, bool ip = false) {  // ring, 2 M residual + 3 M plane slots (ip: 4 M plane slots), control words
  return trio_ring(g, k4, mblk, ip) + (ip ? 4 : 5) * mblk * 256 * k4 + 64;
}

// Frames whose three-wave workgroups (k_inverse_trio<k4, mblk>) are all resident at the same
// time on the current device: the runtime's occupancy answer for the instantiation times the CUs.
static uint32_t trio_capacity(const Geom& g, uint32_t k4, uint32_t mblk, bool ip = false) {
  // The answer depends on the LDS bytes of the launch, i.e. on the frame width (the ring), not on
  // (k4, mblk) alone: a second context with another width must not reuse the first one's figure.
  struct Slot { uint32_t lds, cap; };
  static std::mutex m;
  static Slot capm[64][2][kTrioWideMaxK4 + 1][5][4] = {};  // a few widths per instantiation; beyond that the oldest entry is replaced
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  dev &= 63;
  if (k4 == 0 || (k4 > kSpecMaxK4 && !(trio_wide_rows(k4) && mblk == 2 && ip)) || (mblk != 1 && mblk != 2 && mblk != 4)) return 0;
  if (ip && !trio_ip_ok(g, k4)) return 0;
  const size_t lds = trio_lds(g, k4, mblk, ip);
  if (lds > 64 * 1024) return 0;
  std::lock_guard<std::mutex> l(m);
  Slot* slots = capm[dev][ip ? 1 : 0][k4][mblk];
  uint32_t* cap_slot = nullptr;
  for (int i = 0; i < 4 && !cap_slot; i++)
    if (slots[i].cap && slots[i].lds == (uint32_t)lds) cap_slot = &slots[i].cap;
  if (!cap_slot) {
    static uint32_t next_victim = 0;
    Slot* sl = nullptr;
    for (int i = 0; i < 4 && !sl; i++)
      if (!slots[i].cap) sl = &slots[i];
    if (!sl) sl = &slots[next_victim++ & 3];
    sl->lds = (uint32_t)lds;
    sl->cap = 0;
    cap_slot = &sl->cap;
  }
  if (!*cap_slot) {
    int blocks = 0;
    hipError_t e = hipErrorInvalidValue;
#define FPV_OCC(K4, MB) if (k4 == K4 && mblk == MB) { const auto kfn = trio_kernel<K4, MB>(ip, K4 > 6 ? trio_res_rows(g, K4, MB) : 0u); e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kfn, 192, lds); }
#define FPV_OCC3(K4) FPV_OCC(K4, 1) FPV_OCC(K4, 2) FPV_OCC(K4, 4)
    FPV_OCC3(1) FPV_OCC3(2) FPV_OCC3(3) FPV_OCC3(4) FPV_OCC3(5) FPV_OCC3(6) FPV_OCC(8, 2) FPV_OCC(10, 2) FPV_OCC(12, 2)
#undef FPV_OCC3
#undef FPV_OCC
    if (e != hipSuccess || blocks <= 0) blocks = 3;
    *cap_slot = (uint32_t)blocks * device_cus();
  }
  return *cap_slot;
}

// Which inverse kernel a batch of nframes takes: 3 three waves per frame, 0 one wave per
// frame (speculative), 1 the serial-chain kernel, 2 the byte-granular one; *k4 = the segment
// length (dwords per lane and superstep) the speculative kernels run with, *mblk = supersteps
// per barrier of the three-wave kernel.
// Three waves per frame as long as EVERY frame's workgroup is resident at the same time.  In
// order of preference (1280x800 on MI355X; profiles/r05_inverse_ab.log):
//   in place (round 5), blocks of 2 supersteps at the geometry's own k4 = 5   20 KB of LDS, 4 workgroups per CU (registers): 1024 frames
//   in place, blocks of 2 at k4 - 1 (fewer registers)                         5 per CU:                                      1280 frames
// then, for widths the in-place form does not take (xs % (4 k4) != 0) or with FPV_TRIO_INPLACE=0, round 3's order
// (profiles/r03_inverse_sizes.log, r03_inverse_ab.log):
//   blocks of 4 supersteps at k4 = 5                      41 KB of LDS, 3 workgroups per CU:   768 frames
//   blocks of 4 at k4 - 1 (shorter supersteps)            37 KB, 4 per CU:                    1024 frames
//   blocks of 2 at k4                                     21 KB, 4 per CU (registers):        1024 frames
//   blocks of 2 at k4 - 1 (fewer registers)               5 per CU:                           1280 frames
// and one wave per frame beyond (a second round of workgroups would run nearly alone).
static int inverse_mode(const Geom& g, uint32_t nframes, uint32_t* k4, uint32_t* mblk, bool* ip) {
  int mode = -1;  // the environment is read per launch: tests switch kernels inside one process
  *k4 = g.spec_k4;
  *mblk = 1;
  *ip = false;
  if (getenv("FPV_INVERSE_SLOW")) {
    mode = 2;
  } else if (const char* e = getenv("FPV_INVERSE")) {
    mode = e[0] == 'f' ? 1 : e[0] == 's' && e[1] == 'l' ? 2 : e[0] == 's' ? 0 : e[0] == 't' ? 3 : -1;
  }
  if (mode < 0 || mode == 3) {
    const bool forced = mode == 3;
    const uint32_t mmax = trio_block_max();
    mode = 0;
    // The in-place form first (residuals two blocks ahead, round 5) where the width allows it and its LDS admits the
    // batch (FPV_TRIO_INPLACE=0: never), in blocks of TWO supersteps unless FPV_TRIO_BLOCK says otherwise: with the
    // residuals that far ahead the shorter block is 1-2 % faster at every batch size (profiles/r05_inverse_ab.log), and it
    // leaves the geometry's own segment length within four workgroups per CU (1280 wide: k4 = 5 up to 1024 frames).
    // (where the north operands stay in registers -- trio_res_ok: the chain wave bounds the batch -- and at most two frames share a
    // CU, blocks of FOUR again: fewer barriers on the critical wave, 3-4 % at 32 ... 512 frames; at 576 frames blocks of two are 2 % ahead.  Loading the next superstep's residuals while a superstep's
    // passes run was built on top and is 3-15 % SLOWER: the backend waits for such loads in front of the pass loop.)
    const bool blk_env = getenv("FPV_TRIO_BLOCK") != nullptr;
    const uint32_t ipm = blk_env ? mmax : std::min(mmax, 2u);
    // Rows that are 1 ... 3 times 2048 pixels (2048, 4096, 6144 wide), 1 or 2 times 2560, or 3072 pixels: segments of 32 / 40 / 48
    // pixels per lane (k4 = 8 / 10 / 12, this form only, blocks of two: 35 ... 49 KB of LDS): the longer the segment, the fewer
    // supersteps and the fewer passes each (4096x4096: 7.68 -> 6.13 ms per batch).  FPV_TRIO_WIDE=0: never.
    if (!blk_env && nframes <= 2 * device_cus() && !(getenv("FPV_TRIO_WIDE") && atoi(getenv("FPV_TRIO_WIDE")) == 0))
      for (uint32_t kk = kTrioWideMaxK4; kk > kSpecMaxK4 && mode == 0; kk -= 2)   // 48, 40, 32 pixels per lane: the longest that fits the row
        if (trio_res_ok(g, kk, 2, nframes) && nframes <= trio_capacity(g, kk, 2, true)) { mode = 3; *k4 = kk; *mblk = 2; *ip = true; }
    // (Rows of 2 ... 4 supersteps take the register-north form as well where one of these two segment lengths divides the row
    // -- 2560 wide at k4 = 5, 3072 at 6: -4.5 % --; going down to a length that does -- 4096 and 2048 wide at k4 = 4 -- costs more
    // than the form returns: 8.17 against 7.67 ms per batch of 4096x4096, profiles/r05_inverse_ab.log.)
    for (uint32_t kk = g.spec_k4; kk + 1 >= g.spec_k4 && kk >= 1 && mode == 0; kk--) {
      if (!blk_env && mmax >= 4 && nframes <= 2 * device_cus() && trio_res_ok(g, kk, 4, nframes) && nframes <= trio_capacity(g, kk, 4, true)) { mode = 3; *k4 = kk; *mblk = 4; *ip = true; }
      else if (nframes <= trio_capacity(g, kk, ipm, true)) { mode = 3; *k4 = kk; *mblk = ipm; *ip = true; }
    }
    for (uint32_t mb = mmax; mb >= 1 && mode == 0; mb >>= 1)
      for (uint32_t kk = g.spec_k4; kk + 1 >= g.spec_k4 && kk >= 1 && mode == 0; kk--)
        if (nframes <= trio_capacity(g, kk, mb)) { mode = 3; *k4 = kk; *mblk = mb; }
    if (mode == 0 && forced && g.spec_k4) {  // FPV_INVERSE=trio beyond the capacity: the longest block that fits
      if (trio_capacity(g, g.spec_k4, ipm, true)) { mode = 3; *mblk = ipm; *ip = true; }
      for (uint32_t mb = mmax; mb >= 1 && mode == 0; mb >>= 1)
        if (trio_capacity(g, g.spec_k4, mb)) { mode = 3; *mblk = mb; }
    }
  }
  return mode;
}

hipError_t set_chain_simd_mask(uint32_t value) {
  std::vector<uint32_t> host(kChainSimdWords, value);
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return e;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_chain_simd_mask), host.data(), host.size() * sizeof(uint32_t));
}

const char* inverse_kernel_name(const Geom& g, uint32_t nframes) {
  uint32_t k4, mblk;
  bool ip;
  const int mode = inverse_mode(g, nframes, &k4, &mblk, &ip);
  const bool aligned = (g.xs & 3) == 0 && (g.npix & 3) == 0;
  if (mode == 3 && g.spec_k4) return "k_inverse_trio";
  if (mode == 0 && g.spec_k4) return "k_inverse_spec";
  if (mode != 2 && aligned && g.xs >= kInvC && (g.inv_ring + 4 * kInvC) * 4 <= 160 * 1024 - 64) return "k_inverse_fast";
  return "k_inverse";
}

hipError_t launch_inverse(const Geom& g, const uint8_t* d_high,
                          const uint8_t* d_low, const uint8_t* d_dhigh,
                          const uint8_t* d_dlow, const uint8_t* d_flags,
                          uint32_t nframes, uint16_t* d_frames, hipStream_t s,
                          uint32_t* d_diag, bool memo2_hint) {
  prepare_kernels();
  // The memo behind the second pass (k_inverse_trio<..., M2>): the caller's hint (fpv_capi.hip: the share of supersteps that ran
  // into the long-correction path in the batches before); FPV_TRIO_MEMO2 = 0 / 1 (read per launch: tests, A/B runs) never / always.
  bool memo2 = memo2_hint;
  if (const char* e = getenv("FPV_TRIO_MEMO2")) memo2 = atoi(e) != 0;
  // FPV_INVERSE = auto (default) | trio | spec | fast | slow selects the kernel (tests, A/B timing).
  // auto: three waves per frame (shorter serial path per frame: 0.62 ms for a lone 1280x800
  // frame, 0.90 ms for 1000 of them) as long as every frame's workgroup is resident at once,
  // i.e. up to trio_capacity() frames (1024 at 1280x800 on MI355X: the register budget admits
  // four such workgroups per CU); beyond that a second round of workgroups would run nearly
  // alone, and one wave per frame (more frames resident, less synchronisation) wins:
  // 1100 frames 1.35 ms against 1.56, 2000 frames 1.65 against 2.28 (profiles/r02_inverse_ab.log).
  uint32_t k4, mblk;
  bool ip;
  const int mode = inverse_mode(g, nframes, &k4, &mblk, &ip);
  const bool aligned = (g.xs & 3) == 0 && (g.npix & 3) == 0;
  if (mode == 3 && g.spec_k4) {
#ifdef FPV_EXP_PITCH
    Geom gp = g;
    gp.strip_rows = g_exp_in_pad;   // plane pitch = npix + pad bytes
    gp.strips = g_exp_out_pad;      // pixel pitch = npix + pad pixels
#define FPV_TRIO_GEOM gp
#else
#define FPV_TRIO_GEOM g
#endif
#define FPV_LAUNCH_TRIO(K4, MB)                                                                       \
  if (k4 == K4 && mblk == MB) {                                                                       \
    const auto kfn = trio_kernel<K4, MB>(ip, K4 > 6 ? trio_res_rows(g, K4, MB) : ip ? trio_res_ok(g, K4, MB, nframes) : 0u, memo2); \
    kfn<<<nframes, 192, trio_lds(g, K4, MB, ip), s>>>(d_high, d_low, d_dhigh, d_dlow, d_flags, d_frames, FPV_TRIO_GEOM, \
                                                      trio_ring(g, K4, MB, ip), d_diag);             \
  }
#define FPV_LAUNCH_TRIO3(K4) FPV_LAUNCH_TRIO(K4, 1) FPV_LAUNCH_TRIO(K4, 2) FPV_LAUNCH_TRIO(K4, 4)
    if (k4 < 1 || (k4 > kSpecMaxK4 && !(mblk == 2 && ip && trio_res_rows(g, k4, mblk)))) return hipErrorInvalidValue;
    FPV_LAUNCH_TRIO3(1) FPV_LAUNCH_TRIO3(2) FPV_LAUNCH_TRIO3(3)
    FPV_LAUNCH_TRIO3(4) FPV_LAUNCH_TRIO3(5) FPV_LAUNCH_TRIO3(6) FPV_LAUNCH_TRIO(8, 2) FPV_LAUNCH_TRIO(10, 2) FPV_LAUNCH_TRIO(12, 2)
#undef FPV_LAUNCH_TRIO3
#undef FPV_LAUNCH_TRIO
  } else if (mode == 0 && g.spec_k4) {
#define FPV_LAUNCH_SPEC(K4)                                                                  \
  case K4:                                                                                   \
    k_inverse_spec<K4><<<nframes, 64, g.spec_ring, s>>>(d_high, d_low, d_dhigh, d_dlow,      \
                                                        d_flags, d_frames, g, d_diag);       \
    break;
    switch (k4) {
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
