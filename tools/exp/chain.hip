// Dependent-chain latency probe for the inverse predictor's inner loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short pk16 __attribute__((ext_vector_type(2)));
template <int V>
__global__ void k(const int* __restrict__ a, int* o, int iters, unsigned long long* cyc) {
  int n = a[threadIdx.x & 3], d = a[4] , r = a[5];
  unsigned w = a[6];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      int t = (int)w + d;
      int lo = min(n, (int)w), hi = max(n, (int)w);
      int m = max(lo, min(hi, t));
      if (V == 0) { pk16 s = __builtin_bit_cast(pk16, (unsigned)m) + __builtin_bit_cast(pk16, (unsigned)r); w = __builtin_bit_cast(unsigned, s); }
      else if (V == 1) { w = (unsigned)(m + r) & 0xffffu; }
      else if (V == 2) { unsigned x; asm volatile("v_add_u16 %0, %1, %2" : "=v"(x) : "v"(m), "v"(r)); w = x; }
      else if (V == 3) { unsigned x; asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(x) : "v"(m), "v"(r)); w = x; }
      else if (V == 4) { w = (unsigned)m; }                    // no add at all (lower bound)
      else if (V == 5) { w = (unsigned)(m + r); }              // unmasked (wrong, but 3 plain ops)
      n ^= u;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; }
  o[blockIdx.x * 64 + threadIdx.x] = (int)w;
}
template <int V> void run(const char* name, int blocks, int* a, int* o, unsigned long long* c) {
  int iters = 4096;
  k<V><<<blocks, 64>>>(a, o, iters, c);
  hipDeviceSynchronize();
  unsigned long long h[4]; hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost);
  printf("%-28s blocks=%5d  %.2f s_memtime ticks / pixel\n", name, blocks, (double)h[0] / (iters * 16.0));
}
int main() {
  int ha[8] = {100 << 8, 120 << 8, 90 << 8, 130 << 8, -(3 << 8), 17 << 8, 110 << 8, 0};
  int *a, *o; unsigned long long* c;
  hipMalloc(&a, sizeof ha); hipMalloc(&o, 4 * 64 * 8192); hipMalloc(&c, 8 * 8192);
  hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice);
  for (int blocks : {1, 1024, 4096}) {
    run<0>("pk_add_u16", blocks, a, o, c);
    run<1>("add_u32 + and", blocks, a, o, c);
    run<2>("asm v_add_u16", blocks, a, o, c);
    run<3>("asm add_u32_sdwa", blocks, a, o, c);
    run<4>("no add (bound)", blocks, a, o, c);
    run<5>("add_u32 unmasked", blocks, a, o, c);
  }
}
