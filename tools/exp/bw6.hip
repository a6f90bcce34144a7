// Bandwidth probe 6 (experiments only, round 3): the memory side of the INVERSE predictor alone.
// One workgroup per frame (as k_inverse_trio: every frame a sequential stream, 1000 of them at
// once), W waves per frame, each taking every W-th chunk of V bytes per plane: residual-high and low
// planes of the frame (read once), the two delta planes (shared by all frames: L2 / MALL hits),
// 2 V bytes of pixels out.  No chain, no LDS, no pacing: what does the SHAPE allow?
//   V   bytes per plane and visit (the kernel's superstep is 1280)
//   BL  bytes per lane and load instruction (4 = the kernel's LDS-DMA dword; stores are 2 BL)
//   D   chunks in flight per wave (register double buffering)
// Reference: the same bytes in dispatch order (one 256-thread workgroup per 4096-byte chunk).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int BL> struct Vec;
template <> struct Vec<4> { typedef uint32_t T; };
template <> struct Vec<8> { typedef uint2 T; };
template <> struct Vec<16> { typedef uint4 T; };

__device__ __forceinline__ uint32_t add4(uint32_t a, uint32_t b) {
  return ((a & 0x7f7f7f7fu) + (b & 0x7f7f7f7fu)) ^ ((a ^ b) & 0x80808080u);
}
__device__ __forceinline__ void emit(uint16_t* out, uint32_t h, uint32_t l, uint32_t dh, uint32_t dl) {
  const uint32_t hh = add4(h, dh), ll = add4(l, dl);
  uint2 o;
  o.x = __builtin_amdgcn_perm(hh, ll, 0x05010400u);
  o.y = __builtin_amdgcn_perm(hh, ll, 0x07030602u);
  *reinterpret_cast<uint2*>(out) = o;
}
template <int BL> __device__ __forceinline__ void emit_v(uint16_t* out, const typename Vec<BL>::T& h, const typename Vec<BL>::T& l,
                                                         const typename Vec<BL>::T& dh, const typename Vec<BL>::T& dl);
template <> __device__ __forceinline__ void emit_v<4>(uint16_t* out, const uint32_t& h, const uint32_t& l, const uint32_t& dh, const uint32_t& dl) {
  emit(out, h, l, dh, dl);
}
template <> __device__ __forceinline__ void emit_v<8>(uint16_t* out, const uint2& h, const uint2& l, const uint2& dh, const uint2& dl) {
  uint4 o;
  uint32_t a = add4(h.x, dh.x), b = add4(l.x, dl.x);
  o.x = __builtin_amdgcn_perm(a, b, 0x05010400u); o.y = __builtin_amdgcn_perm(a, b, 0x07030602u);
  a = add4(h.y, dh.y); b = add4(l.y, dl.y);
  o.z = __builtin_amdgcn_perm(a, b, 0x05010400u); o.w = __builtin_amdgcn_perm(a, b, 0x07030602u);
  *reinterpret_cast<uint4*>(out) = o;
}
template <> __device__ __forceinline__ void emit_v<16>(uint16_t* out, const uint4& h, const uint4& l, const uint4& dh, const uint4& dl) {
  emit_v<8>(out, make_uint2(h.x, h.y), make_uint2(l.x, l.y), make_uint2(dh.x, dh.y), make_uint2(dl.x, dl.y));
  emit_v<8>(out + 8, make_uint2(h.z, h.w), make_uint2(l.z, l.w), make_uint2(dh.z, dh.w), make_uint2(dl.z, dl.w));
}

template <int V, int BL, int W>
__global__ __launch_bounds__(64 * W) void k_frames(const uint8_t* __restrict__ high, const uint8_t* __restrict__ low,
                                                   const uint8_t* __restrict__ dhigh, const uint8_t* __restrict__ dlow,
                                                   uint16_t* __restrict__ out, uint32_t npix) {
  typedef typename Vec<BL>::T T;
  constexpr int NI = V / (64 * BL);
  const uint32_t frame = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint8_t* hp = high + (size_t)frame * npix;
  const uint8_t* lp = low + (size_t)frame * npix;
  uint16_t* op = out + (size_t)frame * npix;
  const uint32_t nchunks = npix / V;
  T cur[4][NI], nxt[4][NI];
  auto load = [&](uint32_t c, T (&b)[4][NI]) {
#pragma unroll
    for (int k = 0; k < NI; k++) {
      const uint32_t o = c * V + (k * 64 + lane) * BL;
      b[0][k] = *reinterpret_cast<const T*>(hp + o);
      b[1][k] = *reinterpret_cast<const T*>(lp + o);
      b[2][k] = *reinterpret_cast<const T*>(dhigh + o);
      b[3][k] = *reinterpret_cast<const T*>(dlow + o);
    }
  };
  uint32_t c = wave;
  if (c < nchunks) load(c, cur);
  for (; c < nchunks; c += W) {
    const bool more = c + W < nchunks;
    if (more) load(c + W, nxt);
#pragma unroll
    for (int k = 0; k < NI; k++) emit_v<BL>(op + c * V + (k * 64 + lane) * BL, cur[0][k], cur[1][k], cur[2][k], cur[3][k]);
    if (more) {
#pragma unroll
      for (int p = 0; p < 4; p++)
#pragma unroll
        for (int k = 0; k < NI; k++) cur[p][k] = nxt[p][k];
    }
  }
}

// The same i/o waves beside a DUMMY chain wave (wave 0, issue priority 3): per chunk a dependent
// chain of STEPS x 3 VALU operations on registers, no memory, no data from or for the i/o waves.
// Whatever this loses against k_frames is the price of sharing the SIMDs' issue slots with the
// chain waves, not of any hand-off.
template <int V, int BL, int W, int STEPS>
__global__ __launch_bounds__(64 * (W + 1)) void k_frames_chain(const uint8_t* __restrict__ high, const uint8_t* __restrict__ low,
                                                             const uint8_t* __restrict__ dhigh, const uint8_t* __restrict__ dlow,
                                                             uint16_t* __restrict__ out, uint32_t npix, uint32_t* __restrict__ sink) {
  typedef typename Vec<BL>::T T;
  constexpr int NI = V / (64 * BL);
  const uint32_t frame = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t nchunks = npix / V;
  if (wave == 0) {
    __builtin_amdgcn_s_setprio(3);
    uint32_t x = lane << 8, n[20], d[20], r[20];
#pragma unroll
    for (int k = 0; k < 20; k++) { n[k] = (lane * 7 + k * 13) << 8 & 0xff00; d[k] = (k * 5 + lane) & 0x1f00; r[k] = (k * 11 + lane * 3) << 8 & 0xff00; }
    for (uint32_t c = 0; c < nchunks; c++) {
      for (int rep = 0; rep < STEPS / 20; rep++) {
#pragma unroll
        for (int k = 0; k < 20; k++) {  // the chain step of the inverse kernels (4 issue slots with the compiler's s_nop)
          uint32_t t;
          asm volatile("v_add_u32_e32 %1, %0, %2\n\tv_med3_i32 %0, %3, %0, %1\n\tv_add_u16_e32 %0, %0, %4"
                       : "+v"(x), "=&v"(t) : "v"(d[k]), "v"(n[k]), "v"(r[k]));
        }
      }
    }
    if (x == 0x12345u) sink[frame] = x;
    return;
  }
  const uint8_t* hp = high + (size_t)frame * npix;
  const uint8_t* lp = low + (size_t)frame * npix;
  uint16_t* op = out + (size_t)frame * npix;
  T cur[4][NI], nxt[4][NI];
  auto load = [&](uint32_t c, T (&b)[4][NI]) {
#pragma unroll
    for (int k = 0; k < NI; k++) {
      const uint32_t o = c * V + (k * 64 + lane) * BL;
      b[0][k] = *reinterpret_cast<const T*>(hp + o);
      b[1][k] = *reinterpret_cast<const T*>(lp + o);
      b[2][k] = *reinterpret_cast<const T*>(dhigh + o);
      b[3][k] = *reinterpret_cast<const T*>(dlow + o);
    }
  };
  uint32_t c = wave - 1;
  if (c < nchunks) load(c, cur);
  for (; c < nchunks; c += W) {
    const bool more = c + W < nchunks;
    if (more) load(c + W, nxt);
#pragma unroll
    for (int k = 0; k < NI; k++) emit_v<BL>(op + c * V + (k * 64 + lane) * BL, cur[0][k], cur[1][k], cur[2][k], cur[3][k]);
    if (more) {
#pragma unroll
      for (int p = 0; p < 4; p++)
#pragma unroll
        for (int k = 0; k < NI; k++) cur[p][k] = nxt[p][k];
    }
  }
}

// k_frames with every plane byte going through LDS by LDS-DMA (global_load_lds_dword: 64 lanes x
// 4 bytes per instruction, as k_inverse_trio stages them), two chunks in flight per wave, then
// ds_read_b32 and the same stores.  W waves per frame, no chain wave.
template <int V, int W>
__global__ __launch_bounds__(64 * W) void k_frames_dma(const uint8_t* __restrict__ high, const uint8_t* __restrict__ low,
                                                       const uint8_t* __restrict__ dhigh, const uint8_t* __restrict__ dlow,
                                                       uint16_t* __restrict__ out, uint32_t npix) {
  constexpr int NI = V / 256;
  __shared__ __attribute__((aligned(16))) uint8_t lds[W][2][4][V];
  const uint32_t frame = blockIdx.x, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint8_t* hp = high + (size_t)frame * npix;
  const uint8_t* lp = low + (size_t)frame * npix;
  uint16_t* op = out + (size_t)frame * npix;
  const uint32_t nchunks = npix / V;
  auto dma = [&](uint32_t c, uint32_t b) {
#pragma unroll
    for (int k = 0; k < NI; k++) {
      const uint32_t o = c * V + (k * 64 + lane) * 4;
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t*>(hp + o), reinterpret_cast<uint32_t*>(&lds[wave][b][0][k * 256]), 4, 0, 0);
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t*>(lp + o), reinterpret_cast<uint32_t*>(&lds[wave][b][1][k * 256]), 4, 0, 0);
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t*>(dhigh + o), reinterpret_cast<uint32_t*>(&lds[wave][b][2][k * 256]), 4, 0, 0);
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t*>(dlow + o), reinterpret_cast<uint32_t*>(&lds[wave][b][3][k * 256]), 4, 0, 0);
    }
  };
  uint32_t c = wave, b = 0;
  if (c < nchunks) dma(c, 0);
  for (; c < nchunks; c += W, b ^= 1) {
    const bool more = c + W < nchunks;
    if (more) { dma(c + W, b ^ 1); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NI + NI) : "memory"); }   // this chunk's loads (and older stores) done
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t h[NI], l[NI], dh[NI], dl[NI];
#pragma unroll
    for (int k = 0; k < NI; k++) {
      const uint32_t a = (uint32_t)(uintptr_t)&lds[wave][b][0][(k * 64 + lane) * 4];
      asm volatile("ds_read_b32 %0, %1" : "=v"(h[k]) : "v"(a) : "memory");
      asm volatile("ds_read_b32 %0, %1" : "=v"(l[k]) : "v"(a + V) : "memory");
      asm volatile("ds_read_b32 %0, %1" : "=v"(dh[k]) : "v"(a + 2 * V) : "memory");
      asm volatile("ds_read_b32 %0, %1" : "=v"(dl[k]) : "v"(a + 3 * V) : "memory");
    }
#pragma unroll
    for (int k = 0; k < NI; k++) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h[k]), "+v"(l[k]), "+v"(dh[k]), "+v"(dl[k])::"memory");
#pragma unroll
    for (int k = 0; k < NI; k++) emit(op + c * V + (k * 64 + lane) * 4, h[k], l[k], dh[k], dl[k]);
  }
}

// dispatch order: one workgroup of 256 threads per 4096-byte chunk of some frame
__global__ __launch_bounds__(256) void k_tiles(const uint8_t* __restrict__ high, const uint8_t* __restrict__ low,
                                               const uint8_t* __restrict__ dhigh, const uint8_t* __restrict__ dlow,
                                               uint16_t* __restrict__ out, uint32_t npix, uint32_t chunks_per_frame) {
  const uint32_t frame = blockIdx.x / chunks_per_frame, c = blockIdx.x - frame * chunks_per_frame;
  const uint32_t o = c * 4096 + threadIdx.x * 16;
  const size_t fo = (size_t)frame * npix + o;
  const uint4 h = *reinterpret_cast<const uint4*>(high + fo), l = *reinterpret_cast<const uint4*>(low + fo);
  const uint4 dh = *reinterpret_cast<const uint4*>(dhigh + o), dl = *reinterpret_cast<const uint4*>(dlow + o);
  emit_v<16>(out + fo, h, l, dh, dl);
}

int main(int argc, char** argv) {
  const uint32_t XS = 1280, YS = 800, NP = XS * YS, N = argc > 1 ? atoi(argv[1]) : 1000;
  uint8_t *h, *l, *dh, *dl; uint16_t* out;
  CK(hipMalloc(&h, (size_t)N * NP)); CK(hipMalloc(&l, (size_t)N * NP));
  CK(hipMalloc(&dh, NP)); CK(hipMalloc(&dl, NP)); CK(hipMalloc(&out, (size_t)N * NP * 2));
  CK(hipMemset(h, 1, (size_t)N * NP)); CK(hipMemset(l, 2, (size_t)N * NP)); CK(hipMemset(dh, 3, NP)); CK(hipMemset(dl, 4, NP));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const double bytes = 4.0 * N * NP;
  auto time = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; i++) launch();
    CK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0;
    for (int r = 0; r < 5; r++) {
      CK(hipEventRecord(a));
      for (int i = 0; i < 10; i++) launch();
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10; best = ms < best ? ms : best; sum += ms;
    }
    CK(hipGetLastError());
    printf("%-64s %8.4f ms (mean %8.4f)  %6.0f GB/s algorithmic\n", name, best, sum / 5, bytes / best / 1e6);
    fflush(stdout);
  };
#define RUN(V, BL, W) time("frames V=" #V " BL=" #BL " W=" #W, [&] { k_frames<V, BL, W><<<N, 64 * W>>>(h, l, dh, dl, out, NP); });
  printf("%u frames of %ux%u: %.3f GB per pass (reads of the delta planes from L2 included)\n", N, XS, YS, bytes / 1e9);
  time("tiles (dispatch order, 4096 B per workgroup, 16 B per lane)", [&] { k_tiles<<<N * (NP / 4096), 256>>>(h, l, dh, dl, out, NP, NP / 4096); });
  RUN(1280, 4, 2) RUN(2560, 4, 2) RUN(5120, 4, 2) RUN(10240, 4, 2)
  RUN(1280, 4, 3) RUN(2560, 4, 3) RUN(5120, 4, 3)
  RUN(2560, 8, 2) RUN(5120, 8, 2) RUN(10240, 8, 2) RUN(5120, 8, 3) RUN(5120, 8, 4)
  RUN(5120, 16, 2) RUN(10240, 16, 2) RUN(10240, 16, 3) RUN(20480, 16, 2)
  time("frames through LDS-DMA V=1280 W=2", [&] { k_frames_dma<1280, 2><<<N, 128>>>(h, l, dh, dl, out, NP); });
  time("frames through LDS-DMA V=1280 W=3", [&] { k_frames_dma<1280, 3><<<N, 192>>>(h, l, dh, dl, out, NP); });
  time("frames through LDS-DMA V=2560 W=2", [&] { k_frames_dma<2560, 2><<<N, 128>>>(h, l, dh, dl, out, NP); });
  uint32_t* sink; CK(hipMalloc(&sink, N * 4));
#define RUNC(V, BL, W, ST) time("frames + dummy chain wave V=" #V " BL=" #BL " W=" #W " steps/chunk=" #ST, [&] { k_frames_chain<V, BL, W, ST><<<N, 64 * (W + 1)>>>(h, l, dh, dl, out, NP, sink); });
  RUNC(1280, 4, 2, 20) RUNC(1280, 4, 2, 60) RUNC(1280, 4, 2, 80) RUNC(1280, 4, 2, 100) RUNC(1280, 4, 2, 120) RUNC(1280, 4, 2, 140)
  RUNC(1280, 4, 2, 200)
  RUNC(2560, 8, 2, 200)
  time("tiles again", [&] { k_tiles<<<N * (NP / 4096), 256>>>(h, l, dh, dl, out, NP, NP / 4096); });
  return 0;
}
