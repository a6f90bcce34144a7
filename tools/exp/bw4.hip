// Bandwidth probe 4 (experiments only, round 3): which DECOMPOSITION of the forward pass
// reaches the simple copy's rate (6.5 TB/s, bw3) on the memory side?
//
// bw3 showed: a dispatch-ordered one-float4-per-thread copy runs at 6.5-6.6 TB/s, round 2's
// chunked copy at 5.6-5.9, and persistent waves each streaming its own 400 KB strip (k_stream's
// shape) at 4.7-4.9 TB/s WITHOUT any predictor arithmetic.  Hypothesis: thousands of scattered
// streams make every DRAM access a row miss; a compact window of addresses swept by all resident
// waves together does not.  Shapes compared here (all: 32 B pixels in, 16 + 16 B delta in (L2),
// 16 + 16 B planes out per 16 pixels, split by byte permutes, SWAR subtraction):
//   P  persistent waves, flat 1024-pixel steps, strip of R rows per task, FRAME-major task order
//   T  tiles: one workgroup per R rows of one frame, dispatched in memory order, flat segments
//      strided over the threads, every load of the tile issued before the first store
//   C  column strips: one workgroup per R rows, thread = 16-pixel column segment, walks down the
//      rows (north neighbour = its own previous segment), depth-D prefetch
// X = XCD-aware order: workgroup b works on tile (b % 8) * (ntiles / 8) + b / 8, so that each XCD
// sweeps one contiguous eighth of the batch (neighbouring tiles share an L2).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned sub4(unsigned a, unsigned b) {
  const unsigned H = 0x80808080u;
  return ((a | H) - (b & ~H)) ^ ((a ^ ~b) & H);
}
__device__ __forceinline__ u32x4 ldnt(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
__device__ __forceinline__ void stnt(void* p, u32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p)); }

__device__ __forceinline__ void split_sub(u32x4 A, u32x4 B, u32x4 dh, u32x4 dl, bool delta, u32x4* h, u32x4* l) {
  h->x = __builtin_amdgcn_perm(A.y, A.x, 0x07050301u); l->x = __builtin_amdgcn_perm(A.y, A.x, 0x06040200u);
  h->y = __builtin_amdgcn_perm(A.w, A.z, 0x07050301u); l->y = __builtin_amdgcn_perm(A.w, A.z, 0x06040200u);
  h->z = __builtin_amdgcn_perm(B.y, B.x, 0x07050301u); l->z = __builtin_amdgcn_perm(B.y, B.x, 0x06040200u);
  h->w = __builtin_amdgcn_perm(B.w, B.z, 0x07050301u); l->w = __builtin_amdgcn_perm(B.w, B.z, 0x06040200u);
  if (delta) {
    h->x = sub4(h->x, dh.x); h->y = sub4(h->y, dh.y); h->z = sub4(h->z, dh.z); h->w = sub4(h->w, dh.w);
    l->x = sub4(l->x, dl.x); l->y = sub4(l->y, dl.y); l->z = sub4(l->z, dl.z); l->w = sub4(l->w, dl.w);
  }
}

struct SArgs {
  const unsigned short* frames;
  const unsigned char *dh, *dl;
  unsigned char *high, *low;
  unsigned* ctr;
  unsigned xs, npix, rows, tiles_per_frame, nframes, xcd;
};

__device__ __forceinline__ unsigned remap(unsigned b, unsigned ntiles, unsigned xcd) {
  if (!xcd) return b;
  const unsigned per = ntiles / 8;  // ntiles % 8 == 0 in these runs
  return (b & 7u) * per + (b >> 3);
}

// ---- P: persistent waves, frame-major small strips, halo re-read ----
template <int DEPTH, bool DELTA>
__global__ __launch_bounds__(64) void k_P(SArgs g) {
  const unsigned lane = threadIdx.x;
  const unsigned ntasks = g.nframes * g.tiles_per_frame;
  const unsigned strip_px = g.rows * g.xs, niter = strip_px / 1024;
  unsigned task = blockIdx.x;
  unsigned sink = 0;
  while (task < ntasks) {
    const unsigned t = remap(task, ntasks, g.xcd);
    const unsigned frame = t / g.tiles_per_frame, strip = t - frame * g.tiles_per_frame;
    const unsigned s = strip * strip_px;
    const unsigned short* img = g.frames + (size_t)frame * g.npix + s;
    unsigned char* hi = g.high + (size_t)frame * g.npix + s;
    unsigned char* lo = g.low + (size_t)frame * g.npix + s;
    u32x4 a[DEPTH], b[DEPTH], dh[DEPTH], dl[DEPTH];
    auto load = [&](int d, unsigned it) {
      const unsigned o = min(it, niter - 1) * 1024 + lane * 16;
      a[d] = ldnt(img + o); b[d] = ldnt(img + o + 8);
      if (DELTA) { dh[d] = *reinterpret_cast<const u32x4*>(g.dh + s + o); dl[d] = *reinterpret_cast<const u32x4*>(g.dl + s + o); }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++) load(d, d);
    if (s > 0) {  // halo: xs + 16 pixels in front of the strip, 4 pixels per lane and round
      const unsigned ngrp = (g.xs + 16) / 4;
      for (unsigned k = lane; k < ngrp; k += 64) {
        const uint2 v = *reinterpret_cast<const uint2*>(img - (g.xs + 16) + 4 * k);
        unsigned dv = DELTA ? *reinterpret_cast<const unsigned*>(g.dh + s - (g.xs + 16) + 4 * k) : 0u;
        sink += sub4(__builtin_amdgcn_perm(v.y, v.x, 0x07050301u), dv);
      }
    }
    auto step = [&](int d, unsigned it) {
      const unsigned o = it * 1024 + lane * 16;
      u32x4 h, l;
      split_sub(a[d], b[d], dh[d], dl[d], DELTA, &h, &l);
      stnt(lo + o, l); stnt(hi + o, h);
    };
    const unsigned nd = niter - niter % DEPTH;
    for (unsigned it = 0; it < nd; it += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; d++) { step(d, it + d); load(d, it + d + DEPTH); }
    }
#pragma unroll
    for (int d = 0; d < DEPTH - 1; d++) if (nd + d < niter) step(d, nd + d);
    unsigned next = 0;
    if (lane == 0) next = atomicAdd(g.ctr, 1u);
    task = gridDim.x + (unsigned)__builtin_amdgcn_readfirstlane((int)next);
  }
  if (sink == 0x12345) g.ctr[1] = sink;
}

// ---- T: one workgroup per tile, flat segments strided over threads, loads first ----
template <int T, int MAXSEG, bool DELTA>
__global__ __launch_bounds__(T) void k_T(SArgs g) {
  const unsigned ntiles = g.nframes * g.tiles_per_frame;
  const unsigned t = remap(blockIdx.x, ntiles, g.xcd);
  const unsigned frame = t / g.tiles_per_frame, tile = t - frame * g.tiles_per_frame;
  const unsigned tile_px = g.rows * g.xs, s = tile * tile_px;
  const unsigned short* img = g.frames + (size_t)frame * g.npix + s;
  unsigned char* hi = g.high + (size_t)frame * g.npix + s;
  unsigned char* lo = g.low + (size_t)frame * g.npix + s;
  const unsigned nseg = tile_px / 16;
  u32x4 a[MAXSEG], b[MAXSEG], dh[MAXSEG], dl[MAXSEG];
  unsigned sink = 0;
#pragma unroll
  for (int k = 0; k < MAXSEG; k++) {
    const unsigned seg = min(threadIdx.x + k * T, nseg - 1);
    a[k] = ldnt(img + seg * 16); b[k] = ldnt(img + seg * 16 + 8);
    if (DELTA) { dh[k] = *reinterpret_cast<const u32x4*>(g.dh + s + seg * 16); dl[k] = *reinterpret_cast<const u32x4*>(g.dl + s + seg * 16); }
  }
  if (s > 0) {  // halo row
    const unsigned ngrp = (g.xs + 16) / 4;
    for (unsigned k = threadIdx.x; k < ngrp; k += T) {
      const uint2 v = *reinterpret_cast<const uint2*>(img - (g.xs + 16) + 4 * k);
      unsigned dv = DELTA ? *reinterpret_cast<const unsigned*>(g.dh + s - (g.xs + 16) + 4 * k) : 0u;
      sink += sub4(__builtin_amdgcn_perm(v.y, v.x, 0x07050301u), dv);
    }
  }
#pragma unroll
  for (int k = 0; k < MAXSEG; k++) {
    const unsigned seg = threadIdx.x + k * T;
    if (seg < nseg) {
      u32x4 h, l;
      split_sub(a[k], b[k], dh[k], dl[k], DELTA, &h, &l);
      stnt(lo + seg * 16, l); stnt(hi + seg * 16, h);
    }
  }
  if (sink == 0x12345) g.ctr[1] = sink;
}

// ---- C: column strips: thread = 16-pixel column segment, walks down R rows ----
// workgroup = T threads covering whole rows: item i -> colseg = i % ncs handled for rows r0 + (i / ncs) * RPT ... no:
// here simply: the tile's rows are split into T / ncs_pad row bands; thread (band, colseg) walks its band's rows.
template <int T, int DEPTH, bool DELTA>
__global__ __launch_bounds__(T) void k_C(SArgs g, unsigned bands, unsigned rows_per_band) {
  const unsigned ntiles = g.nframes * g.tiles_per_frame;
  const unsigned t = remap(blockIdx.x, ntiles, g.xcd);
  const unsigned frame = t / g.tiles_per_frame, tile = t - frame * g.tiles_per_frame;
  const unsigned ncs = g.xs / 16;
  const unsigned item = threadIdx.x;
  const unsigned band = item / ncs, cs = item - band * ncs;
  if (band >= bands) return;
  const unsigned row0 = tile * g.rows + band * rows_per_band;
  const size_t o0 = (size_t)row0 * g.xs + cs * 16;
  const unsigned short* img = g.frames + (size_t)frame * g.npix + o0;
  unsigned char* hi = g.high + (size_t)frame * g.npix + o0;
  unsigned char* lo = g.low + (size_t)frame * g.npix + o0;
  const unsigned char* pdh = g.dh + o0;
  const unsigned char* pdl = g.dl + o0;
  const unsigned n = rows_per_band;
  u32x4 a[DEPTH], b[DEPTH], dh[DEPTH], dl[DEPTH];
  auto load = [&](int d, unsigned r) {
    const size_t o = (size_t)min(r, n - 1) * g.xs;
    a[d] = ldnt(img + o); b[d] = ldnt(img + o + 8);
    if (DELTA) { dh[d] = *reinterpret_cast<const u32x4*>(pdh + o); dl[d] = *reinterpret_cast<const u32x4*>(pdl + o); }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; d++) load(d, d);
  unsigned sink = 0;
  if (row0 > 0) {  // halo: the row above, own column segment (+ the left neighbour's last pixel)
    const u32x4 ha = ldnt(img - g.xs), hb = ldnt(img - g.xs + 8);
    sink += ha.x + hb.w;
  }
  auto step = [&](int d, unsigned r) {
    const size_t o = (size_t)r * g.xs;
    u32x4 h, l;
    split_sub(a[d], b[d], dh[d], dl[d], DELTA, &h, &l);
    stnt(lo + o, l); stnt(hi + o, h);
  };
  const unsigned nd = n - n % DEPTH;
  for (unsigned r = 0; r < nd; r += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) { step(d, r + d); load(d, r + d + DEPTH); }
  }
#pragma unroll
  for (int d = 0; d < DEPTH - 1; d++) if (nd + d < n) step(d, nd + d);
  if (sink == 0x12345) g.ctr[1] = sink;
}

static hipEvent_t e0, e1;
template <typename L>
static double time_it(const char* name, double bytes, int reps, L launch) {
  launch();
  CK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0;
  for (int r = 0; r < 3; r++) {
    CK(hipEventRecord(e0));
    for (int k = 0; k < reps; k++) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    best = ms < best ? ms : best;
    sum += ms;
  }
  printf("%-72s %8.4f ms (mean %8.4f) %7.0f GB/s\n", name, best, sum / 3, bytes / best / 1e6);
  fflush(stdout);
  return bytes / best / 1e6;
}

int main(int argc, char** argv) {
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const unsigned xs = argc > 1 ? atoi(argv[1]) : 1280, ys = argc > 2 ? atoi(argv[2]) : 800, N = argc > 3 ? atoi(argv[3]) : 1000;
  const unsigned npix = xs * ys;
  SArgs g;
  unsigned short* frames; unsigned char *dh, *dl, *high, *low; unsigned* ctr;
  CK(hipMalloc(&frames, (size_t)N * npix * 2)); CK(hipMalloc(&high, (size_t)N * npix)); CK(hipMalloc(&low, (size_t)N * npix));
  CK(hipMalloc(&dh, npix)); CK(hipMalloc(&dl, npix)); CK(hipMalloc(&ctr, 256));
  CK(hipMemset(frames, 3, (size_t)N * npix * 2)); CK(hipMemset(dh, 1, npix)); CK(hipMemset(dl, 2, npix));
  g.frames = frames; g.dh = dh; g.dl = dl; g.high = high; g.low = low; g.ctr = ctr;
  g.xs = xs; g.npix = npix; g.nframes = N;
  const double bytes = (double)N * npix * 4.0;
  printf("device %d CUs; %u x %ux%u, %.3f GB algorithmic per pass\n", cus, N, xs, ys, bytes / 1e9);
  char nm[200];
  // reference: the simple copy on the same buffers (frames -> high|low is not contiguous; copy frames -> frames-sized? use high+low halves)
  for (unsigned xcd : {0u, 1u}) {
    g.xcd = xcd;
    for (unsigned rows : {4u, 8u, 16u, 32u, 160u}) {
      if (ys % rows || (rows * xs) % 1024) continue;
      g.rows = rows; g.tiles_per_frame = ys / rows;
      if ((N * g.tiles_per_frame) % 8) continue;
#define RUN_P(D, DELTA, WPC)                                                                                   \
  snprintf(nm, sizeof nm, "P persistent frame-major rows=%u depth=%d delta=%d waves/CU=%d xcd=%u", rows, D, DELTA, WPC, xcd); \
  time_it(nm, bytes, 5, [&] { hipMemsetAsync(ctr, 0, 4, 0); k_P<D, DELTA><<<cus * WPC, 64>>>(g); });
      RUN_P(3, true, 8) RUN_P(2, true, 16) RUN_P(3, false, 8)
      if (rows <= 32) { RUN_P(2, true, 32) }
    }
    for (unsigned rows : {4u, 8u, 16u, 32u}) {
      if (ys % rows) continue;
      g.rows = rows; g.tiles_per_frame = ys / rows;
      const unsigned ntiles = N * g.tiles_per_frame;
      if (ntiles % 8) continue;
      const unsigned nseg = rows * xs / 16;
#define RUN_T(T, MAXSEG, DELTA)                                                                      \
  if ((nseg + T - 1) / T == MAXSEG) {                                                                \
    snprintf(nm, sizeof nm, "T tile rows=%u T=%d seg/thread=%d delta=%d xcd=%u", rows, T, MAXSEG, DELTA, xcd); \
    time_it(nm, bytes, 5, [&] { k_T<T, MAXSEG, DELTA><<<ntiles, T>>>(g); });                         \
  }
      RUN_T(256, 2, true) RUN_T(256, 3, true) RUN_T(256, 5, true) RUN_T(256, 10, true) RUN_T(512, 1, true) RUN_T(512, 2, true)
      RUN_T(512, 3, true) RUN_T(512, 5, true) RUN_T(1024, 1, true) RUN_T(1024, 2, true) RUN_T(1024, 3, true)
      RUN_T(256, 2, false) RUN_T(256, 3, false) RUN_T(256, 5, false) RUN_T(512, 3, false) RUN_T(320, 1, true) RUN_T(320, 2, true) RUN_T(320, 4, true)
      RUN_T(320, 1, false) RUN_T(320, 2, false)
    }
    // column strips: T threads = bands x ncs (ncs = xs / 16)
    const unsigned ncs = xs / 16;
    for (unsigned rows : {8u, 16u, 32u, 64u}) {
      if (ys % rows) continue;
      g.rows = rows; g.tiles_per_frame = ys / rows;
      const unsigned ntiles = N * g.tiles_per_frame;
      if (ntiles % 8) continue;
#define RUN_C(T, D, DELTA, BANDS)                                                                                  \
  if ((BANDS) * ncs <= T && rows % (BANDS) == 0) {                                                                 \
    snprintf(nm, sizeof nm, "C colstrip rows=%u T=%d bands=%d depth=%d delta=%d xcd=%u", rows, T, BANDS, D, DELTA, xcd); \
    time_it(nm, bytes, 5, [&] { k_C<T, D, DELTA><<<ntiles, T>>>(g, BANDS, rows / (BANDS)); });                     \
  }
      RUN_C(128, 2, true, 1) RUN_C(128, 3, true, 1) RUN_C(128, 4, true, 1) RUN_C(256, 2, true, 2) RUN_C(256, 3, true, 2)
      RUN_C(256, 4, true, 2) RUN_C(256, 3, false, 2) RUN_C(320, 2, true, 4) RUN_C(320, 3, true, 4) RUN_C(320, 4, true, 4) RUN_C(640, 2, true, 8)
      RUN_C(320, 3, false, 4)
    }
  }
  return 0;
}
