// chain_ilp.hip -- round 4: does the inverse predictor's dependent chain (v_add_u32 -> v_med3_i32 ->
// v_add_u16 per pixel) issue back to back, or does every dependent instruction wait for its
// predecessor's result?  One wave per SIMD (256 threads per CU, one workgroup per CU), C independent
// chains interleaved per lane (C = 1, 2, 3, 4): if C = 2 takes the time of C = 1, half of the lone
// chain's time was pipeline latency that a second chain in the same wave would fill.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int C>
__global__ void k(const uint32_t* __restrict__ a, uint32_t* o, int iters, unsigned long long* cyc) {
  uint32_t N[C][4], D[C][4], R[C][4], w[C];
#pragma unroll
  for (int c = 0; c < C; c++) {
    w[c] = a[8 + c] + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; k++) { N[c][k] = a[k] + c * 256; D[c][k] = a[4 + k]; R[c][k] = a[12 + k] + c; }
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int c = 0; c < C; c++) {
          uint32_t t, p, x;
          asm volatile("v_add_u32_e32 %0, %3, %4\n\tv_med3_i32 %1, %5, %4, %0\n\tv_add_u16_e32 %2, %1, %6"
                       : "=&v"(t), "=&v"(p), "=&v"(x) : "v"(D[c][k]), "v"(w[c]), "v"(N[c][k]), "v"(R[c][k]));
          w[c] = x;
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  uint32_t s = 0;
#pragma unroll
  for (int c = 0; c < C; c++) s += w[c];
  o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the same with the C chains issued as SEPARATE asm statements per op kind (the compiler may interleave):
template <int C>
__global__ void k2(const uint32_t* __restrict__ a, uint32_t* o, int iters, unsigned long long* cyc) {
  uint32_t N[C][4], D[C][4], R[C][4], w[C];
#pragma unroll
  for (int c = 0; c < C; c++) {
    w[c] = a[8 + c] + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; k++) { N[c][k] = a[k] + c * 256; D[c][k] = a[4 + k]; R[c][k] = a[12 + k] + c; }
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        uint32_t t[C], p[C];
#pragma unroll
        for (int c = 0; c < C; c++) asm volatile("v_add_u32_e32 %0, %1, %2" : "=v"(t[c]) : "v"(D[c][k]), "v"(w[c]));
#pragma unroll
        for (int c = 0; c < C; c++) asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(p[c]) : "v"(N[c][k]), "v"(w[c]), "v"(t[c]));
#pragma unroll
        for (int c = 0; c < C; c++) asm volatile("v_add_u16_e32 %0, %1, %2" : "=v"(w[c]) : "v"(p[c]), "v"(R[c][k]));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  uint32_t s = 0;
#pragma unroll
  for (int c = 0; c < C; c++) s += w[c];
  o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int C, bool INTER> void run(int waves_per_simd, uint32_t* a, uint32_t* o, unsigned long long* c) {
  const int iters = 20000, threads = 256 * waves_per_simd;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  if (INTER) k2<C><<<256, threads>>>(a, o, 100, c); else k<C><<<256, threads>>>(a, o, 100, c);
  hipEventRecord(e0);
  if (INTER) k2<C><<<256, threads>>>(a, o, iters, c); else k<C><<<256, threads>>>(a, o, iters, c);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double steps = (double)iters * 16;  // chain steps per chain
  printf("%s chains per lane %d, %d wave(s) per SIMD: %7.2f ns per chain step of ONE chain (%.2f ns per step and chain), 3 instructions per step\n",
         INTER ? "interleaved by op " : "step after step   ", C, waves_per_simd, ms * 1e6 / steps, ms * 1e6 / steps / C);
}
int main() {
  uint32_t ha[16] = {100 << 8, 120 << 8, 90 << 8, 130 << 8, (uint32_t)-(3 << 8), 2 << 8, (uint32_t)-(1 << 8), 5 << 8, 110 << 8, 95 << 8, 140 << 8, 99 << 8, 17 << 8, 3 << 8, 250 << 8, 9 << 8};
  uint32_t *a, *o; unsigned long long* c;
  hipMalloc(&a, sizeof ha); hipMalloc(&o, 4 * 256 * 1024); hipMalloc(&c, 8 * 256);
  hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice);
  for (int w : {1, 2}) {
    run<1, false>(w, a, o, c); run<2, false>(w, a, o, c); run<1, true>(w, a, o, c); run<2, true>(w, a, o, c); run<3, true>(w, a, o, c); run<4, true>(w, a, o, c);
  }
  return 0;
}
