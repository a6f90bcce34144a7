// Where do the waves of three-wave workgroups land?  1024 workgroups of 192 threads with the
// LDS footprint of k_inverse_trio, all resident at once; every wave records its XCC / SE / CU /
// SIMD (HW_ID registers) and then idles for a while so that the grid stays co-resident.
// Prints how many "wave 0" (chain role) waves each SIMD hosts.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <map>
#include <vector>
__global__ __launch_bounds__(192) void k(uint32_t* out, int spin) {
  extern __shared__ uint8_t smem[];
  smem[threadIdx.x] = 1;
  const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
  const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 3 + (threadIdx.x >> 6)) * 2] = hw;
    out[(blockIdx.x * 3 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
  }
  for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(127);
  if (smem[threadIdx.x] == 0) out[0] = 0;
}
int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 1024, lds = argc > 2 ? atoi(argv[2]) : 17 * 1024;
  uint32_t* d; hipMalloc(&d, wgs * 3 * 2 * 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  k<<<wgs, 192, lds>>>(d, 2000);
  hipDeviceSynchronize();
  std::vector<uint32_t> h(wgs * 3 * 2);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  std::map<uint32_t, int> chain, all;
  for (int w = 0; w < wgs * 3; w++) {
    const uint32_t hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
    const uint32_t simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const uint32_t key = (xcc << 16) | (se << 12) | (sh << 8) | (cu << 4) | simd;
    all[key]++;
    if (w % 3 == 0) chain[key]++;
  }
  int hist_chain[16] = {}, hist_all[16] = {};
  for (auto& kv : all) { hist_all[kv.second < 15 ? kv.second : 15]++; hist_chain[chain.count(kv.first) ? chain[kv.first] : 0]++; }
  printf("SIMDs seen: %zu\nwaves per SIMD (count of SIMDs):", all.size());
  for (int i = 0; i < 10; i++) printf(" %d:%d", i, hist_all[i]);
  printf("\nchain-role waves per SIMD (count of SIMDs):");
  for (int i = 0; i < 10; i++) printf(" %d:%d", i, hist_chain[i]);
  printf("\nfirst workgroups (xcc se cu simd per wave):\n");
  for (int b = 0; b < 8; b++) {
    for (int r = 0; r < 3; r++) {
      const uint32_t hw = h[2 * (3 * b + r)], xcc = h[2 * (3 * b + r) + 1] & 0xf;
      printf("  [%u %u %u %u]", xcc, (hw >> 13) & 7, (hw >> 8) & 0xf, (hw >> 4) & 3);
    }
    printf("\n");
  }
  return 0;
}
