// lds_atomic.hip -- what an LDS atomic add costs on gfx950 (round 4: the heuristics' histogram
// samples of the tile pass are 10 % of its time).  One workgroup per CU, W waves, each issuing N
// ds_add_u32 (no return) with `active` lanes enabled and one of several address patterns; prints
// clocks per wave instruction as seen by one wave (issue + completion, s_memtime around the loop)
// for 1 wave and for W waves per CU competing for the LDS pipeline.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__global__ void k(uint32_t* out, int n, int active, int pattern, uint32_t seed) {
  __shared__ uint32_t h[4096];
  const uint32_t t = threadIdx.x, lane = t & 63;
  for (int i = t; i < 4096; i += blockDim.x) h[i] = 0;
  __syncthreads();
  uint32_t x = seed * 2654435761u + t * 40503u;
  uint32_t addr[8];
  for (int k = 0; k < 8; k++) {
    x = x * 1664525u + 1013904223u;
    uint32_t a;
    if (pattern == 0) a = lane + 64 * k;              // distinct banks
    else if (pattern == 1) a = 7 + k;                 // all lanes the same address
    else if (pattern == 2) a = (x >> 10) & 255;       // random over 256 bins (noise)
    else if (pattern == 3) a = 100 + ((lane >> 4) & 3);  // four values per wave (smooth content)
    else a = (lane & 31) * 2 + 64 * k;                // two lanes per bank, different addresses
    addr[k] = a;
  }
  const bool on = (int)lane < active;
  __syncthreads();
  const uint64_t t0 = clock64();
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (on) atomicAdd(&h[addr[k]], 1u);
  }
  __syncthreads();
  const uint64_t t1 = clock64();
  if (t == 0 && blockIdx.x == 0) out[0] = (uint32_t)(t1 - t0);
  if (h[t & 4095] == 0xffffffffu) out[1] = 1;
}
int main() {
  uint32_t* d; hipMalloc(&d, 64);
  const char* names[5] = {"distinct banks", "one address", "random 256 bins", "4 values per wave", "2 lanes per bank"};
  for (int waves : {1, 4, 16}) {
    for (int pattern = 0; pattern < 5; pattern++) {
      for (int active : {64, 32, 16, 4}) {
        const int n = 2000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<<<256, waves * 64>>>(d, n, active, pattern, 1);
        hipEventRecord(e0);
        k<<<256, waves * 64>>>(d, n, active, pattern, 2);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        uint32_t c; hipMemcpy(&c, d, 4, hipMemcpyDeviceToHost);
        // per CU: waves * n * 8 wave instructions in `ms`
        printf("%2d waves/CU  %-18s active %2d : %8.1f ns per wave instruction per CU (LDS pipeline), %7.1f s_memtime ticks per instruction of one wave\n",
               waves, names[pattern], active, ms * 1e6 / ((double)waves * n * 8), (double)c / (n * 8));
        hipEventDestroy(e0); hipEventDestroy(e1);
      }
    }
  }
  return 0;
}
