#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ __forceinline__ uint32_t inv_px4b(const uint32_t* n, uint32_t w, const uint32_t* d, uint32_t rq, uint32_t* x) {
  uint32_t x0, x1, x2, x3;
#define FPV_ADDB(X, R, SEL) "v_add_u32_sdwa " X ", " X ", " R " dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" SEL "\n\t"
  asm("v_add_u32_e32 %0, %5, %4\n\tv_med3_i32 %0, %6, %4, %0\n\t" FPV_ADDB("%0", "%13", "BYTE_0")
      "v_add_u32_e32 %1, %7, %0\n\tv_med3_i32 %1, %8, %0, %1\n\t" FPV_ADDB("%1", "%13", "BYTE_1")
      "v_add_u32_e32 %2, %9, %1\n\tv_med3_i32 %2, %10, %1, %2\n\t" FPV_ADDB("%2", "%13", "BYTE_2")
      "v_add_u32_e32 %3, %11, %2\n\tv_med3_i32 %3, %12, %2, %3\n\t" FPV_ADDB("%3", "%13", "BYTE_3")
      : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)
      : "v"(w), "v"(d[0]), "v"(n[0]), "v"(d[1]), "v"(n[1]), "v"(d[2]), "v"(n[2]), "v"(d[3]), "v"(n[3]), "v"(rq));
  x[0] = x0; x[1] = x1; x[2] = x2; x[3] = x3;
  return x3;
}
__global__ void k(const uint32_t* in, uint32_t* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* p = in + i * 10;  // w, n0..3, nw0..3 (plain), rq
  uint32_t N[4] = {p[1], p[2], p[3], p[4]}, D[4] = {p[1] - p[5], p[2] - p[6], p[3] - p[7], p[4] - p[8]}, X[4];
  inv_px4b(N, p[0], D, p[9], X);
  for (int k = 0; k < 4; k++) out[i * 4 + k] = X[k];
}
static int cg(int n, int w, int nw) { int i = n < w ? n : w, a = n < w ? w : n; int g = (n + w - nw) & 255; return nw < i ? a : nw > a ? i : g; }
int main() {
  const int n = 1 << 16;
  std::vector<uint32_t> h(n * 10), o(n * 4);
  srand(1);
  for (auto& v : h) v = rand() & 255;
  for (int i = 0; i < n; i++) h[i * 10 + 9] = (uint32_t)rand() * 2654435761u;
  uint32_t *di, *dout;
  hipMalloc(&di, h.size() * 4); hipMalloc(&dout, o.size() * 4);
  hipMemcpy(di, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(di, dout, n);
  hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n && bad < 5; i++) {
    const uint32_t* p = &h[i * 10];
    int w = p[0];
    for (int k = 0; k < 4; k++) {
      int r = (p[9] >> (8 * k)) & 255;
      int x = (r + cg(p[1 + k], w, p[5 + k])) & 255;
      if ((uint32_t)x != o[i * 4 + k]) { printf("MISMATCH i %d k %d: got %u want %d (n %u w %d nw %u r %d)\n", i, k, o[i * 4 + k], x, p[1 + k], w, p[5 + k], r); bad++; break; }
      w = x;
    }
  }
  printf(bad ? "FAILED\n" : "inv_px4b ok\n");
  return bad != 0;
}
