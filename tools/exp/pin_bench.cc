// pin_bench.cc -- what does host staging memory cost on this box?  (experiment; hipcc)
//   hipHostMalloc of several sizes, malloc/mmap + (parallel) first touch + hipHostRegister,
//   hipMalloc, stream/event creation, and the H2D / D2H rate from each kind of memory.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

#include <chrono>
#include <thread>
#include <vector>

static double now() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static void touch_parallel(char* p, size_t n, int threads) {
  std::vector<std::thread> t;
  for (int i = 0; i < threads; i++)
    t.emplace_back([=] { for (size_t o = (size_t)i * 4096; o < n; o += (size_t)threads * 4096) p[o] = 1; });
  for (auto& x : t) x.join();
}

static double rate(void* dst, const void* src, size_t n, hipMemcpyKind k, hipStream_t s) {
  CK(hipMemcpyAsync(dst, src, n, k, s)); CK(hipStreamSynchronize(s));
  const double t0 = now();
  for (int i = 0; i < 4; i++) CK(hipMemcpyAsync(dst, src, n, k, s));
  CK(hipStreamSynchronize(s));
  return 4.0 * n / ((now() - t0) * 1e6);
}

int main() {
  CK(hipSetDevice(0)); CK(hipFree(nullptr));
  hipStream_t s; double t0 = now(); CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); printf("stream create %.3f ms\n", now() - t0);
  hipEvent_t ev; t0 = now(); CK(hipEventCreate(&ev)); printf("event create %.3f ms\n", now() - t0);
  void* d = nullptr; t0 = now(); CK(hipMalloc(&d, (size_t)512 << 20)); printf("hipMalloc 512 MB %.3f ms\n", now() - t0);
  for (size_t mb : {1, 8, 33, 66, 131, 262}) {
    void* dd; t0 = now(); CK(hipMalloc(&dd, mb << 20)); const double a = now() - t0; t0 = now(); CK(hipFree(dd));
    printf("hipMalloc %zu MB %.3f ms, free %.3f ms\n", mb, a, now() - t0);
  }
  for (int rep = 0; rep < 2; rep++)
    for (size_t mb : {4, 33, 66, 131, 262}) {
      void* h; t0 = now(); CK(hipHostMalloc(&h, mb << 20, hipHostMallocDefault)); const double a = now() - t0;
      const double up = rate(d, h, mb << 20, hipMemcpyHostToDevice, s), dn = rate(h, d, mb << 20, hipMemcpyDeviceToHost, s);
      t0 = now(); CK(hipHostFree(h));
      printf("hipHostMalloc %zu MB: %.2f ms (%.2f GB/s), H2D %.1f GB/s, D2H %.1f GB/s, free %.2f ms\n", mb, a, mb / 1024.0 / (a / 1e3), up, dn, now() - t0);
    }
  for (unsigned flags : {(unsigned)hipHostMallocNonCoherent, (unsigned)hipHostMallocWriteCombined, (unsigned)hipHostMallocNumaUser}) {
    void* h; t0 = now(); hipError_t e = hipHostMalloc(&h, (size_t)66 << 20, flags);
    if (e != hipSuccess) { printf("hipHostMalloc flags %u: %s\n", flags, hipGetErrorString(e)); continue; }
    printf("hipHostMalloc 66 MB flags %#x: %.2f ms, H2D %.1f, D2H %.1f GB/s\n", flags, now() - t0,
           rate(d, h, (size_t)66 << 20, hipMemcpyHostToDevice, s), rate(h, d, (size_t)66 << 20, hipMemcpyDeviceToHost, s));
    CK(hipHostFree(h));
  }
  for (int huge = 0; huge < 2; huge++)
    for (int threads : {1, 16})
      for (size_t mb : {33, 131}) {
        const size_t n = mb << 20;
        t0 = now();
        char* p = (char*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (huge) madvise(p, n, MADV_HUGEPAGE);
        const double tm = now() - t0; t0 = now();
        touch_parallel(p, n, threads);
        const double tt = now() - t0; t0 = now();
        CK(hipHostRegister(p, n, hipHostRegisterDefault));
        const double tr = now() - t0;
        const double up = rate(d, p, n, hipMemcpyHostToDevice, s), dn = rate(p, d, n, hipMemcpyDeviceToHost, s);
        t0 = now(); CK(hipHostUnregister(p)); const double tu = now() - t0;
        munmap(p, n);
        printf("mmap%s %zu MB: map %.2f, touch x%d %.2f, register %.2f (total %.2f ms = %.2f GB/s), H2D %.1f D2H %.1f GB/s, unregister %.2f\n",
               huge ? "+THP" : "", mb, tm, threads, tt, tr, tm + tt + tr, mb / 1024.0 / ((tm + tt + tr) / 1e3), up, dn, tu);
      }
  {  // register WITHOUT touching first
    const size_t n = (size_t)66 << 20;
    char* p = (char*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    t0 = now(); CK(hipHostRegister(p, n, hipHostRegisterDefault)); printf("register untouched 66 MB %.2f ms\n", now() - t0);
    CK(hipHostUnregister(p)); munmap(p, n);
  }
  {  // per-frame registration of caller memory (2 MB pieces of a touched 200 MB buffer)
    const size_t n = (size_t)200 << 20, f = 2048000;
    char* p = (char*)malloc(n); memset(p, 1, n);
    t0 = now(); int k = 0;
    for (size_t o = 0; o + f <= n && k < 32; o += f, k++) CK(hipHostRegister(p + o, f, hipHostRegisterDefault));
    printf("register %d x 2 MB pieces of touched malloc memory: %.3f ms each\n", k, (now() - t0) / k);
    const double up = rate(d, p, f, hipMemcpyHostToDevice, s);
    printf("  H2D from a registered 2 MB piece %.1f GB/s; pageable H2D %.1f GB/s, pageable D2H %.1f GB/s\n", up,
           rate(d, p + (size_t)50 * f, 16 * f, hipMemcpyHostToDevice, s), rate(p + (size_t)50 * f, d, 16 * f, hipMemcpyDeviceToHost, s));
    t0 = now(); k = 0;
    for (size_t o = 0; o + f <= n && k < 32; o += f, k++) CK(hipHostUnregister(p + o));
    printf("unregister: %.3f ms each\n", (now() - t0) / k);
    free(p);
  }
  for (int kind = 0; kind < 2; kind++) {  // FIRST copies out of / into fresh staging memory
    const size_t n = (size_t)64 << 20;
    void* h = nullptr;
    if (kind == 0) { CK(hipHostMalloc(&h, n, hipHostMallocDefault)); }
    else {
      h = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      madvise(h, n, MADV_HUGEPAGE); touch_parallel((char*)h, n, 1);
      CK(hipHostRegister(h, n, hipHostRegisterDefault));
    }
    void* dnew; CK(hipMalloc(&dnew, n));
    double t[5];
    t0 = now(); CK(hipMemcpyAsync(dnew, h, 2 << 20, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); t[0] = now() - t0;
    t0 = now(); CK(hipMemcpyAsync(dnew, h, 2 << 20, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); t[1] = now() - t0;
    t0 = now(); CK(hipMemcpyAsync(dnew, h, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); t[2] = now() - t0;
    t0 = now(); CK(hipMemcpyAsync(h, dnew, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); t[3] = now() - t0;
    t0 = now(); CK(hipMemcpyAsync(h, dnew, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); t[4] = now() - t0;
    printf("fresh %s 64 MB + fresh device buffer: first 2 MB H2D %.3f ms, again %.3f ms, 64 MB H2D %.3f ms, first 64 MB D2H %.3f ms, again %.3f ms\n",
           kind ? "mmap+register" : "hipHostMalloc", t[0], t[1], t[2], t[3], t[4]);
    std::vector<char> pg((size_t)4 << 20, 1);
    t0 = now(); CK(hipMemcpy(dnew, pg.data(), 2 << 20, hipMemcpyHostToDevice)); t[0] = now() - t0;
    t0 = now(); CK(hipMemcpy(dnew, pg.data(), 2 << 20, hipMemcpyHostToDevice)); t[1] = now() - t0;
    printf("   pageable 2 MB hipMemcpy: first %.3f ms, again %.3f ms\n", t[0], t[1]);
  }
  {  // full duplex: H2D on one stream while D2H runs on another
    const size_t n = (size_t)64 << 20;
    void *h1, *h2, *d1, *d2;
    CK(hipHostMalloc(&h1, n, hipHostMallocDefault)); CK(hipHostMalloc(&h2, n, hipHostMallocDefault));
    CK(hipMalloc(&d1, n)); CK(hipMalloc(&d2, n));
    hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipMemcpyAsync(d1, h1, n, hipMemcpyHostToDevice, s)); CK(hipMemcpyAsync(h2, d2, n, hipMemcpyDeviceToHost, s2));
    CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
    t0 = now();
    for (int i = 0; i < 8; i++) { CK(hipMemcpyAsync(d1, h1, n, hipMemcpyHostToDevice, s)); CK(hipMemcpyAsync(h2, d2, n, hipMemcpyDeviceToHost, s2)); }
    CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
    double dt = now() - t0;
    printf("duplex: 8 x 64 MB each way on two streams: %.2f ms -> %.1f GB/s each way\n", dt, 8.0 * n / (dt * 1e6));
    t0 = now();
    for (int i = 0; i < 8; i++) { CK(hipMemcpyAsync(d1, h1, n, hipMemcpyHostToDevice, s)); CK(hipMemcpyAsync(h2, d2, n, hipMemcpyDeviceToHost, s)); }
    CK(hipStreamSynchronize(s));
    dt = now() - t0;
    printf("same stream, alternating directions: %.2f ms -> %.1f GB/s each way\n", dt, 8.0 * n / (dt * 1e6));
    // three streams, each doing H2D then D2H of its own buffers (the slots' pattern)
    hipStream_t st[4]; void *hh[4], *dd[4];
    for (int i = 0; i < 4; i++) { CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking)); CK(hipHostMalloc(&hh[i], n, hipHostMallocDefault)); CK(hipMalloc(&dd[i], n)); }
    for (int nst : {2, 4}) {
      t0 = now();
      for (int r = 0; r < 6; r++)
        for (int i = 0; i < nst; i++) {
          CK(hipMemcpyAsync(dd[i], hh[i], n / 2, hipMemcpyHostToDevice, st[i]));
          CK(hipMemcpyAsync((char*)hh[i] + n / 2, (char*)dd[i] + n / 2, n / 2, hipMemcpyDeviceToHost, st[i]));
        }
      for (int i = 0; i < nst; i++) CK(hipStreamSynchronize(st[i]));
      dt = now() - t0;
      printf("%d streams x (32 MB up, 32 MB down) x 6: %.2f ms -> %.1f GB/s each way\n", nst, dt, 6.0 * nst * (n / 2) / (dt * 1e6));
    }
  }
  {  // concurrent hipHostMalloc from 4 threads
    t0 = now();
    std::vector<std::thread> t; void* hp[4];
    for (int i = 0; i < 4; i++) t.emplace_back([&, i] { hipSetDevice(0); hipHostMalloc(&hp[i], (size_t)33 << 20, hipHostMallocDefault); });
    for (auto& x : t) x.join();
    printf("4 x hipHostMalloc 33 MB concurrently: %.2f ms\n", now() - t0);
    for (int i = 0; i < 4; i++) hipHostFree(hp[i]);
  }
  return 0;
}
