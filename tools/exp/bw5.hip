// Bandwidth probe 5 (experiments only, round 3): what in the predictor's I/O SHAPE costs
// bandwidth against the simple dispatch-ordered copy (6.5 TB/s, bw3)?  All kernels here are
// dispatch-ordered (no persistent waves, no loops over tasks), one unit of 16 pixels per thread
// (or K units strided by the block size), so only the shape differs:
//   copy1      16 B in, 16 B out per thread (the reference point)
//   copy_comb  2 x 16 B adjacent in (lane stride 32 B: each load instruction touches every
//              128-B line of a 2 KiB span, half of it), 2 x 16 B adjacent out
//   split      2 x 16 B adjacent in, 16 B to plane H + 16 B to plane L (the predictor's shape)
//   split8     16 B in (contiguous instruction), 8 B to H + 8 B to L
//   splitx     2 contiguous 16 B loads (1 KiB apart), halves exchanged between lane pairs, so that
//              loads AND stores are contiguous 16 B per lane
//   split_dma  pixels through LDS (global_load_lds_dwordx4: contiguous 1 KiB per instruction),
//              2 x ds_read_b128 per lane, 16 B stores
//   +delta     16 + 16 B of delta planes (2 MB in all: L2 / MALL hits) and SWAR subtraction
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned sub4(unsigned a, unsigned b) {
  const unsigned H = 0x80808080u;
  return ((a | H) - (b & ~H)) ^ ((a ^ ~b) & H);
}
template <bool NT> __device__ __forceinline__ u32x4 ld(const void* p) {
  return NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)) : *reinterpret_cast<const u32x4*>(p);
}
template <bool NT> __device__ __forceinline__ void st(void* p, u32x4 v) {
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
  else *reinterpret_cast<u32x4*>(p) = v;
}
__device__ __forceinline__ void split16(u32x4 A, u32x4 B, u32x4* h, u32x4* l) {
  h->x = __builtin_amdgcn_perm(A.y, A.x, 0x07050301u); l->x = __builtin_amdgcn_perm(A.y, A.x, 0x06040200u);
  h->y = __builtin_amdgcn_perm(A.w, A.z, 0x07050301u); l->y = __builtin_amdgcn_perm(A.w, A.z, 0x06040200u);
  h->z = __builtin_amdgcn_perm(B.y, B.x, 0x07050301u); l->z = __builtin_amdgcn_perm(B.y, B.x, 0x06040200u);
  h->w = __builtin_amdgcn_perm(B.w, B.z, 0x07050301u); l->w = __builtin_amdgcn_perm(B.w, B.z, 0x06040200u);
}

template <int T, bool NTL, bool NTS>
__global__ __launch_bounds__(T) void k_copy1(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, size_t nunits) {
  const size_t u = (size_t)blockIdx.x * T + threadIdx.x;  // unit = 32 B here: two threads' worth -> use 2 passes
  if (u < nunits) {
    const size_t blk = (size_t)blockIdx.x * T * 32;
    const u32x4 a = ld<NTL>(src + blk + threadIdx.x * 16), b = ld<NTL>(src + blk + T * 16 + threadIdx.x * 16);
    st<NTS>(dst + blk + threadIdx.x * 16, a);
    st<NTS>(dst + blk + T * 16 + threadIdx.x * 16, b);
  }
}
template <int T, bool NTL, bool NTS>
__global__ __launch_bounds__(T) void k_copy_comb(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, size_t nunits) {
  const size_t u = (size_t)blockIdx.x * T + threadIdx.x;
  if (u < nunits) {
    const u32x4 a = ld<NTL>(src + u * 32), b = ld<NTL>(src + u * 32 + 16);
    st<NTS>(dst + u * 32, a);
    st<NTS>(dst + u * 32 + 16, b);
  }
}
// K units per thread, strided by T; DELTA: delta planes indexed modulo dmod units
template <int T, int K, bool NTL, bool NTS, bool DELTA>
__global__ __launch_bounds__(T) void k_split(const unsigned char* __restrict__ src, unsigned char* __restrict__ hi,
                                            unsigned char* __restrict__ lo, const unsigned char* __restrict__ dh,
                                            const unsigned char* __restrict__ dl, size_t nunits, unsigned dmod) {
  const size_t u0 = (size_t)blockIdx.x * T * K + threadIdx.x;
  u32x4 a[K], b[K], vh[K], vl[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const size_t u = u0 + (size_t)k * T < nunits ? u0 + (size_t)k * T : nunits - 1;
    a[k] = ld<NTL>(src + u * 32); b[k] = ld<NTL>(src + u * 32 + 16);
    if (DELTA) {
      const unsigned du = (unsigned)(u % dmod);
      vh[k] = *reinterpret_cast<const u32x4*>(dh + (size_t)du * 16);
      vl[k] = *reinterpret_cast<const u32x4*>(dl + (size_t)du * 16);
    }
  }
#pragma unroll
  for (int k = 0; k < K; k++) {
    const size_t u = u0 + (size_t)k * T;
    if (u < nunits) {
      u32x4 h, l;
      split16(a[k], b[k], &h, &l);
      if (DELTA) {
        h.x = sub4(h.x, vh[k].x); h.y = sub4(h.y, vh[k].y); h.z = sub4(h.z, vh[k].z); h.w = sub4(h.w, vh[k].w);
        l.x = sub4(l.x, vl[k].x); l.y = sub4(l.y, vl[k].y); l.z = sub4(l.z, vl[k].z); l.w = sub4(l.w, vl[k].w);
      }
      st<NTS>(lo + u * 16, l);
      st<NTS>(hi + u * 16, h);
    }
  }
}
// 16 B in per thread (8 pixels), 8 B out to each plane
template <int T, bool NTL>
__global__ __launch_bounds__(T) void k_split8(const unsigned char* __restrict__ src, unsigned char* __restrict__ hi,
                                             unsigned char* __restrict__ lo, size_t nhalf) {
  const size_t u = (size_t)blockIdx.x * T + threadIdx.x;
  if (u < nhalf) {
    const u32x4 A = ld<NTL>(src + u * 16);
    u32x2 h, l;
    h.x = __builtin_amdgcn_perm(A.y, A.x, 0x07050301u); l.x = __builtin_amdgcn_perm(A.y, A.x, 0x06040200u);
    h.y = __builtin_amdgcn_perm(A.w, A.z, 0x07050301u); l.y = __builtin_amdgcn_perm(A.w, A.z, 0x06040200u);
    __builtin_nontemporal_store(l, reinterpret_cast<u32x2*>(lo + u * 8));
    __builtin_nontemporal_store(h, reinterpret_cast<u32x2*>(hi + u * 8));
  }
}
// contiguous loads (instruction 0: first KiB of the wave's 2 KiB, instruction 1: second KiB), lane pairs exchange
// halves: even lane ends up with 16 consecutive pixels of the first KiB, odd lane of the second
template <int T, bool NTL, bool DELTA>
__global__ __launch_bounds__(T) void k_splitx(const unsigned char* __restrict__ src, unsigned char* __restrict__ hi,
                                             unsigned char* __restrict__ lo, const unsigned char* __restrict__ dh,
                                             const unsigned char* __restrict__ dl, size_t nunits, unsigned dmod) {
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const size_t w0 = ((size_t)blockIdx.x * (T / 64) + wave) * 64;  // first unit of this wave
  if (w0 >= nunits) return;
  const unsigned char* p = src + w0 * 32;
  const u32x4 a = ld<NTL>(p + lane * 16), b = ld<NTL>(p + 1024 + lane * 16);
  // pixels: a = px [8 lane, 8 lane + 8) of KiB 0, b = same of KiB 1
  u32x2 ha, la, hb, lb;
  ha.x = __builtin_amdgcn_perm(a.y, a.x, 0x07050301u); la.x = __builtin_amdgcn_perm(a.y, a.x, 0x06040200u);
  ha.y = __builtin_amdgcn_perm(a.w, a.z, 0x07050301u); la.y = __builtin_amdgcn_perm(a.w, a.z, 0x06040200u);
  hb.x = __builtin_amdgcn_perm(b.y, b.x, 0x07050301u); lb.x = __builtin_amdgcn_perm(b.y, b.x, 0x06040200u);
  hb.y = __builtin_amdgcn_perm(b.w, b.z, 0x07050301u); lb.y = __builtin_amdgcn_perm(b.w, b.z, 0x06040200u);
  // even lane keeps its A halves and takes the odd neighbour's A halves; odd lane keeps B and takes the even neighbour's B
  const bool odd = lane & 1u;
  const u32x2 give_h = odd ? ha : hb, give_l = odd ? la : lb, keep_h = odd ? hb : ha, keep_l = odd ? lb : la;
  u32x2 got_h, got_l;
  got_h.x = (unsigned)__builtin_amdgcn_mov_dpp((int)give_h.x, 0xb1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true);
  got_h.y = (unsigned)__builtin_amdgcn_mov_dpp((int)give_h.y, 0xb1, 0xf, 0xf, true);
  got_l.x = (unsigned)__builtin_amdgcn_mov_dpp((int)give_l.x, 0xb1, 0xf, 0xf, true);
  got_l.y = (unsigned)__builtin_amdgcn_mov_dpp((int)give_l.y, 0xb1, 0xf, 0xf, true);
  u32x4 h, l;
  if (odd) { h.x = got_h.x; h.y = got_h.y; h.z = keep_h.x; h.w = keep_h.y; l.x = got_l.x; l.y = got_l.y; l.z = keep_l.x; l.w = keep_l.y; }
  else     { h.x = keep_h.x; h.y = keep_h.y; h.z = got_h.x; h.w = got_h.y; l.x = keep_l.x; l.y = keep_l.y; l.z = got_l.x; l.w = got_l.y; }
  // even lane 2k: pixels [16k, 16k+16) of KiB 0 -> plane offset 16 k; odd lane 2k+1: plane offset 512 + 16 k
  const size_t off = w0 * 16 + (odd ? 512u : 0u) + (lane >> 1) * 16;
  if (DELTA) {
    const unsigned dbase = (unsigned)((w0 * 16 + (odd ? 512u : 0u) + (lane >> 1) * 16) % ((size_t)dmod * 16));
    const u32x4 vh = *reinterpret_cast<const u32x4*>(dh + dbase), vl = *reinterpret_cast<const u32x4*>(dl + dbase);
    h.x = sub4(h.x, vh.x); h.y = sub4(h.y, vh.y); h.z = sub4(h.z, vh.z); h.w = sub4(h.w, vh.w);
    l.x = sub4(l.x, vl.x); l.y = sub4(l.y, vl.y); l.z = sub4(l.z, vl.z); l.w = sub4(l.w, vl.w);
  }
  __builtin_nontemporal_store(l, reinterpret_cast<u32x4*>(lo + off));
  __builtin_nontemporal_store(h, reinterpret_cast<u32x4*>(hi + off));
}
// pixels through LDS-DMA
template <int T, bool DELTA>
__global__ __launch_bounds__(T) void k_split_dma(const unsigned char* __restrict__ src, unsigned char* __restrict__ hi,
                                                unsigned char* __restrict__ lo, const unsigned char* __restrict__ dh,
                                                const unsigned char* __restrict__ dl, size_t nunits, unsigned dmod) {
  __shared__ __attribute__((aligned(16))) unsigned char buf[T * 32];
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const size_t w0 = ((size_t)blockIdx.x * (T / 64) + wave) * 64;
  if (w0 >= nunits) return;
  const unsigned char* p = src + w0 * 32;
  unsigned char* mine = buf + wave * 2048;
  __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned*>(p + lane * 16), reinterpret_cast<unsigned*>(mine), 16, 0, 0);
  __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned*>(p + 1024 + lane * 16), reinterpret_cast<unsigned*>(mine + 1024), 16, 0, 0);
  u32x4 vh, vl;
  const size_t u = w0 + lane;
  if (DELTA) {
    const unsigned du = (unsigned)(u % dmod);
    vh = *reinterpret_cast<const u32x4*>(dh + (size_t)du * 16);
    vl = *reinterpret_cast<const u32x4*>(dl + (size_t)du * 16);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const u32x4 a = *reinterpret_cast<const u32x4*>(mine + lane * 32), b = *reinterpret_cast<const u32x4*>(mine + lane * 32 + 16);
  u32x4 h, l;
  split16(a, b, &h, &l);
  if (DELTA) {
    h.x = sub4(h.x, vh.x); h.y = sub4(h.y, vh.y); h.z = sub4(h.z, vh.z); h.w = sub4(h.w, vh.w);
    l.x = sub4(l.x, vl.x); l.y = sub4(l.y, vl.y); l.z = sub4(l.z, vl.z); l.w = sub4(l.w, vl.w);
  }
  __builtin_nontemporal_store(l, reinterpret_cast<u32x4*>(lo + u * 16));
  __builtin_nontemporal_store(h, reinterpret_cast<u32x4*>(hi + u * 16));
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
  printf("%-64s %8.4f ms (mean %8.4f) %7.0f GB/s\n", name, best, sum / 3, bytes / best / 1e6);
  fflush(stdout);
  return bytes / best / 1e6;
}

int main() {
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t npix = 1280 * 800, N = 1000;
  const size_t in_bytes = N * npix * 2, nunits = N * npix / 16;
  const unsigned dmod = (unsigned)(npix / 16);
  unsigned char *src, *dst, *dh, *dl;
  CK(hipMalloc(&src, in_bytes)); CK(hipMalloc(&dst, in_bytes)); CK(hipMalloc(&dh, npix)); CK(hipMalloc(&dl, npix));
  CK(hipMemset(src, 3, in_bytes)); CK(hipMemset(dst, 0, in_bytes)); CK(hipMemset(dh, 1, npix)); CK(hipMemset(dl, 2, npix));
  unsigned char *hi = dst, *lo = dst + in_bytes / 2;
  const double bytes = 2.0 * in_bytes;
  const int R = 10;
  printf("2.048 GB in, 2.048 GB out per pass\n");
  for (int rep = 0; rep < 2; rep++) {
    time_it("copy1 T=256 nt/nt (2 contiguous float4 per thread)", bytes, R, [&] { k_copy1<256, true, true><<<(unsigned)((nunits + 255) / 256), 256>>>(src, dst, nunits); });
    time_it("copy1 T=256 plain/plain", bytes, R, [&] { k_copy1<256, false, false><<<(unsigned)((nunits + 255) / 256), 256>>>(src, dst, nunits); });
    time_it("copy_comb T=256 nt/nt", bytes, R, [&] { k_copy_comb<256, true, true><<<(unsigned)((nunits + 255) / 256), 256>>>(src, dst, nunits); });
    time_it("copy_comb T=256 plain loads / nt stores", bytes, R, [&] { k_copy_comb<256, false, true><<<(unsigned)((nunits + 255) / 256), 256>>>(src, dst, nunits); });
    time_it("copy_comb T=256 plain/plain", bytes, R, [&] { k_copy_comb<256, false, false><<<(unsigned)((nunits + 255) / 256), 256>>>(src, dst, nunits); });
#define SPLIT(T, K, NTL, NTS, DELTA)                                                                          \
  {                                                                                                           \
    char nm[160];                                                                                             \
    snprintf(nm, sizeof nm, "split T=%d K=%d loads %s stores %s delta=%d", T, K, NTL ? "nt" : "plain", NTS ? "nt" : "plain", DELTA); \
    time_it(nm, bytes, R, [&] {                                                                               \
      k_split<T, K, NTL, NTS, DELTA><<<(unsigned)((nunits + (size_t)T * K - 1) / ((size_t)T * K)), T>>>(src, hi, lo, dh, dl, nunits, dmod); }); \
  }
    SPLIT(256, 1, true, true, false) SPLIT(256, 1, false, true, false) SPLIT(256, 1, false, false, false) SPLIT(64, 1, true, true, false)
    SPLIT(512, 1, true, true, false) SPLIT(256, 2, true, true, false) SPLIT(256, 4, true, true, false) SPLIT(1024, 1, true, true, false)
    SPLIT(256, 1, true, true, true) SPLIT(256, 1, false, true, true) SPLIT(256, 2, true, true, true) SPLIT(256, 4, true, true, true)
    SPLIT(512, 1, true, true, true) SPLIT(64, 1, true, true, true) SPLIT(64, 4, true, true, true)
    time_it("split8 T=256 nt (16 B in, 8 + 8 B out)", bytes, R, [&] { k_split8<256, true><<<(unsigned)((2 * nunits + 255) / 256), 256>>>(src, hi, lo, 2 * nunits); });
    time_it("splitx T=256 nt (contiguous loads, lane-pair exchange)", bytes, R, [&] { k_splitx<256, true, false><<<(unsigned)((nunits + 255) / 256), 256>>>(src, hi, lo, dh, dl, nunits, dmod); });
    time_it("splitx T=256 nt + delta", bytes, R, [&] { k_splitx<256, true, true><<<(unsigned)((nunits + 255) / 256), 256>>>(src, hi, lo, dh, dl, nunits, dmod); });
    time_it("splitx T=64 nt + delta", bytes, R, [&] { k_splitx<64, true, true><<<(unsigned)((nunits + 63) / 64), 64>>>(src, hi, lo, dh, dl, nunits, dmod); });
    time_it("split_dma T=256 (pixels via LDS-DMA)", bytes, R, [&] { k_split_dma<256, false><<<(unsigned)((nunits + 255) / 256), 256>>>(src, hi, lo, dh, dl, nunits, dmod); });
    time_it("split_dma T=256 + delta", bytes, R, [&] { k_split_dma<256, true><<<(unsigned)((nunits + 255) / 256), 256>>>(src, hi, lo, dh, dl, nunits, dmod); });
    time_it("split_dma T=64 + delta", bytes, R, [&] { k_split_dma<64, true><<<(unsigned)((nunits + 63) / 64), 64>>>(src, hi, lo, dh, dl, nunits, dmod); });
  }
  return 0;
}
