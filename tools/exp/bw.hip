// Standalone bandwidth probe (experiments only): read-only, write-only, copy.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MODE, bool NT, int UNROLL>
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16, unsigned* sink) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      if (MODE != 1) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
      else { v[u].x = (unsigned)i; v[u].y = u; v[u].z = 3; v[u].w = 4; }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      if (MODE == 0) acc += v[u];
      else if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride);
      else dst[i + u * stride] = v[u];
    }
  }
  if (MODE == 0 && acc.x + acc.y + acc.z + acc.w == 0x12345678) *sink = 1;
}
template <int MODE, bool NT, int UNROLL>
void run(const char* name, u32x4* a, u32x4* b, size_t bytes, int blocks, unsigned* sink) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, NT, UNROLL><<<blocks, 256>>>(a, b, bytes / 16, sink);
  hipEventRecord(e0);
  for (int r = 0; r < 10; r++) k<MODE, NT, UNROLL><<<blocks, 256>>>(a, b, bytes / 16, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  double moved = (MODE == 2 ? 2.0 : 1.0) * bytes;
  printf("%-28s blocks=%5d unroll=%d  %.3f ms  %.0f GB/s\n", name, blocks, UNROLL, ms, moved / ms / 1e6);
}
int main() {
  size_t bytes = (size_t)2 << 30;
  u32x4 *a, *b; unsigned* sink;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 4);
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  for (int blocks : {2048, 4096, 8192}) {
    run<0, false, 4>("read", a, b, bytes, blocks, sink);
    run<0, true, 4>("read nt", a, b, bytes, blocks, sink);
    run<1, false, 4>("write", a, b, bytes, blocks, sink);
    run<1, true, 4>("write nt", a, b, bytes, blocks, sink);
    run<2, false, 4>("copy", a, b, bytes, blocks, sink);
    run<2, true, 4>("copy nt", a, b, bytes, blocks, sink);
    run<2, false, 1>("copy", a, b, bytes, blocks, sink);
    run<2, true, 8>("copy nt", a, b, bytes, blocks, sink);
  }
  return 0;
}
