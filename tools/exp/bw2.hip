// Bandwidth probe 2 (experiments only): which access shape reaches the device's
// read+write ceiling?  Compares grid-stride vs contiguous-chunk copies, 1->2
// stream "split" copies (the predictor's shape: 2P in, P + P out), store policies,
// workgroup sizes, and the runtime's own memcpy/memset.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

enum { PLAIN = 0, NT = 1 };

template <int POL>
__device__ __forceinline__ u32x4 ld(const u32x4* p) {
  return POL == NT ? __builtin_nontemporal_load(p) : *p;
}
template <int POL>
__device__ __forceinline__ void st(u32x4* p, u32x4 v) {
  if (POL == NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// grid-stride copy, UNROLL loads in flight
template <int LP, int SP, int UNROLL, int T>
__global__ __launch_bounds__(T) void k_gs(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * T;
  size_t i = (size_t)blockIdx.x * T + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) v[u] = ld<LP>(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < UNROLL; u++) st<SP>(dst + i + u * stride, v[u]);
  }
}

// contiguous chunk per workgroup: block b copies [b*chunk, (b+1)*chunk) in T*16-byte rows
template <int LP, int SP, int UNROLL, int T>
__global__ __launch_bounds__(T) void k_chunk(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16,
                                            size_t chunk16) {
  for (size_t c = blockIdx.x; c * chunk16 < n16; c += gridDim.x) {
    const size_t b = c * chunk16, e = b + chunk16 < n16 ? b + chunk16 : n16;
    size_t i = b + threadIdx.x;
    for (; i + (UNROLL - 1) * T < e; i += UNROLL * T) {
      u32x4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; u++) v[u] = ld<LP>(src + i + u * T);
#pragma unroll
      for (int u = 0; u < UNROLL; u++) st<SP>(dst + i + u * T, v[u]);
    }
  }
}

// predictor-shaped: each lane reads 32 B (two dwordx4) and writes 16 B to each of two planes
template <int LP, int SP, int T>
__global__ __launch_bounds__(T) void k_split2(const u32x4* __restrict__ src, u32x4* __restrict__ hi,
                                             u32x4* __restrict__ lo, size_t nseg, size_t chunk) {
  for (size_t c = blockIdx.x; c * chunk < nseg; c += gridDim.x) {
    const size_t b = c * chunk, e = b + chunk < nseg ? b + chunk : nseg;
    for (size_t i = b + threadIdx.x; i < e; i += T) {
      const u32x4 a = ld<LP>(src + 2 * i), bb = ld<LP>(src + 2 * i + 1);
      u32x4 h, l;
      h.x = __builtin_amdgcn_perm(a.y, a.x, 0x07050301u); l.x = __builtin_amdgcn_perm(a.y, a.x, 0x06040200u);
      h.y = __builtin_amdgcn_perm(a.w, a.z, 0x07050301u); l.y = __builtin_amdgcn_perm(a.w, a.z, 0x06040200u);
      h.z = __builtin_amdgcn_perm(bb.y, bb.x, 0x07050301u); l.z = __builtin_amdgcn_perm(bb.y, bb.x, 0x06040200u);
      h.w = __builtin_amdgcn_perm(bb.w, bb.z, 0x07050301u); l.w = __builtin_amdgcn_perm(bb.w, bb.z, 0x06040200u);
      st<SP>(hi + i, h);
      st<SP>(lo + i, l);
    }
  }
}

// write-only
template <int SP, int T>
__global__ __launch_bounds__(T) void k_wr(u32x4* __restrict__ dst, size_t n16, size_t chunk16) {
  for (size_t c = blockIdx.x; c * chunk16 < n16; c += gridDim.x) {
    const size_t b = c * chunk16, e = b + chunk16 < n16 ? b + chunk16 : n16;
    for (size_t i = b + threadIdx.x; i < e; i += T) {
      u32x4 v; v.x = (unsigned)i; v.y = 1; v.z = 2; v.w = 3;
      st<SP>(dst + i, v);
    }
  }
}
// read-only
template <int LP, int T>
__global__ __launch_bounds__(T) void k_rd(const u32x4* __restrict__ src, size_t n16, size_t chunk16, unsigned* sink) {
  u32x4 acc = {0, 0, 0, 0};
  for (size_t c = blockIdx.x; c * chunk16 < n16; c += gridDim.x) {
    const size_t b = c * chunk16, e = b + chunk16 < n16 ? b + chunk16 : n16;
    size_t i = b + threadIdx.x;
    for (; i + 3 * T < e; i += 4 * T) {
      u32x4 v0 = ld<LP>(src + i), v1 = ld<LP>(src + i + T), v2 = ld<LP>(src + i + 2 * T), v3 = ld<LP>(src + i + 3 * T);
      acc += v0 + v1 + v2 + v3;
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 0x12345678) *sink = 1;
}

static hipEvent_t e0, e1;
#define TIME(name, bytes_moved, reps, launch)                                     \
  do {                                                                            \
    launch;                                                                       \
    hipEventRecord(e0);                                                           \
    for (int r_ = 0; r_ < (reps); r_++) { launch; }                               \
    hipEventRecord(e1);                                                           \
    hipEventSynchronize(e1);                                                      \
    float ms_;                                                                    \
    hipEventElapsedTime(&ms_, e0, e1);                                            \
    ms_ /= (reps);                                                                \
    printf("%-52s %8.3f ms %7.0f GB/s\n", name, ms_, (double)(bytes_moved) / ms_ / 1e6); \
    fflush(stdout);                                                               \
  } while (0)

int main(int argc, char** argv) {
  size_t bytes = (size_t)2 << 30;
  if (argc > 1) bytes = (size_t)atoll(argv[1]) << 20;
  u32x4 *a, *b;
  unsigned* sink;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 4);
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t n16 = bytes / 16;
  const int R = 8;
  char nm[128];
  printf("buffer %zu MiB each\n", bytes >> 20);
  TIME("hipMemcpyDtoD", 2.0 * bytes, R, hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0));
  TIME("hipMemset", 1.0 * bytes, R, hipMemsetAsync(b, 5, bytes, 0));
  for (int blocks : {1024, 2048, 4096, 16384}) {
    snprintf(nm, sizeof nm, "gs copy plain u4 T256 blocks=%d", blocks);
    TIME(nm, 2.0 * bytes, R, (k_gs<PLAIN, PLAIN, 4, 256><<<blocks, 256>>>(a, b, n16)));
    snprintf(nm, sizeof nm, "gs copy nt u4 T256 blocks=%d", blocks);
    TIME(nm, 2.0 * bytes, R, (k_gs<NT, NT, 4, 256><<<blocks, 256>>>(a, b, n16)));
    snprintf(nm, sizeof nm, "gs copy ntld/plainst u4 T256 blocks=%d", blocks);
    TIME(nm, 2.0 * bytes, R, (k_gs<NT, PLAIN, 4, 256><<<blocks, 256>>>(a, b, n16)));
    snprintf(nm, sizeof nm, "gs copy nt u8 T256 blocks=%d", blocks);
    TIME(nm, 2.0 * bytes, R, (k_gs<NT, NT, 8, 256><<<blocks, 256>>>(a, b, n16)));
  }
  for (size_t chunkKB : {16, 64, 256, 1024}) {
    const size_t c16 = chunkKB * 1024 / 16;
    for (int blocks : {2048, 8192}) {
      snprintf(nm, sizeof nm, "chunk copy nt u4 T256 chunk=%zuK blocks=%d", chunkKB, blocks);
      TIME(nm, 2.0 * bytes, R, (k_chunk<NT, NT, 4, 256><<<blocks, 256>>>(a, b, n16, c16)));
      snprintf(nm, sizeof nm, "chunk copy plain u4 T256 chunk=%zuK blocks=%d", chunkKB, blocks);
      TIME(nm, 2.0 * bytes, R, (k_chunk<PLAIN, PLAIN, 4, 256><<<blocks, 256>>>(a, b, n16, c16)));
      snprintf(nm, sizeof nm, "chunk copy nt u4 T64 chunk=%zuK blocks=%d", chunkKB, 4 * blocks);
      TIME(nm, 2.0 * bytes, R, (k_chunk<NT, NT, 4, 64><<<4 * blocks, 64>>>(a, b, n16, c16)));
    }
  }
  // predictor shape: src = a (bytes), hi = b[0 : bytes/2), lo = b[bytes/2 : bytes)
  {
    const size_t nseg = n16 / 2;
    u32x4* hi = b; u32x4* lo = b + nseg;
    for (size_t chunkKB : {16, 64, 1024}) {
      const size_t chunk = chunkKB * 1024 / 32;
      for (int blocks : {2048, 8192}) {
        snprintf(nm, sizeof nm, "split2 nt T256 chunk(in)=%zuK blocks=%d", chunkKB, blocks);
        TIME(nm, 2.0 * bytes, R, (k_split2<NT, NT, 256><<<blocks, 256>>>(a, hi, lo, nseg, chunk)));
        snprintf(nm, sizeof nm, "split2 plain T256 chunk(in)=%zuK blocks=%d", chunkKB, blocks);
        TIME(nm, 2.0 * bytes, R, (k_split2<PLAIN, PLAIN, 256><<<blocks, 256>>>(a, hi, lo, nseg, chunk)));
        snprintf(nm, sizeof nm, "split2 nt T64 chunk(in)=%zuK blocks=%d", chunkKB, 4 * blocks);
        TIME(nm, 2.0 * bytes, R, (k_split2<NT, NT, 64><<<4 * blocks, 64>>>(a, hi, lo, nseg, chunk)));
      }
    }
  }
  for (size_t chunkKB : {64, 1024}) {
    const size_t c16 = chunkKB * 1024 / 16;
    snprintf(nm, sizeof nm, "write plain chunk=%zuK blocks=4096", chunkKB);
    TIME(nm, 1.0 * bytes, R, (k_wr<PLAIN, 256><<<4096, 256>>>(b, n16, c16)));
    snprintf(nm, sizeof nm, "write nt chunk=%zuK blocks=4096", chunkKB);
    TIME(nm, 1.0 * bytes, R, (k_wr<NT, 256><<<4096, 256>>>(b, n16, c16)));
    snprintf(nm, sizeof nm, "read plain chunk=%zuK blocks=4096", chunkKB);
    TIME(nm, 1.0 * bytes, R, (k_rd<PLAIN, 256><<<4096, 256>>>(a, n16, c16, sink)));
    snprintf(nm, sizeof nm, "read nt chunk=%zuK blocks=4096", chunkKB);
    TIME(nm, 1.0 * bytes, R, (k_rd<NT, 256><<<4096, 256>>>(a, n16, c16, sink)));
  }
  return 0;
}
