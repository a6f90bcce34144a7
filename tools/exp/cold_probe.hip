// cold_probe.hip -- where a fresh process's first GPU milliseconds go (round 4, cold start of
// the upstream benchmark.cc): runtime init, primary context, first launch (code object), stream
// creation, first copies in either direction; sequentially, or with the independent parts on
// threads of their own ("par").   usage: cold_probe [par]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
#include <thread>
#include <vector>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec / 1e6; }
__global__ void k_nop(int* p) { if (p) *p = 1; }
int main(int argc, char** argv) {
  const bool par = argc > 1 && !strcmp(argv[1], "par");
  const double t0 = now();
  int n = 0;
  hipGetDeviceCount(&n);
  const double t1 = now();
  hipSetDevice(0); hipFree(nullptr);
  const double t2 = now();
  int* d = nullptr; hipMalloc(&d, 1 << 22);
  const double t3 = now();
  std::vector<char> host(1 << 22, 1);
  void* pin = nullptr;
  hipStream_t st[3] = {};
  double t_launch = 0, t_streams = 0, t_h2d = 0, t_d2h = 0, t_pin = 0;
  auto launch = [&] { const double a = now(); k_nop<<<1, 64>>>(d); hipDeviceSynchronize(); t_launch = now() - a; };
  auto streams = [&] { const double a = now(); for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking); t_streams = now() - a; };
  auto pinit = [&] { const double a = now(); hipHostMalloc(&pin, 1 << 22, 0); t_pin = now() - a; };
  if (par) {
    std::thread a(launch), b(streams), c(pinit);
    a.join(); b.join(); c.join();
  } else {
    launch(); streams(); pinit();
  }
  const double t4 = now();
  auto h2d = [&] { const double a = now(); hipMemcpyAsync(d, pin, 1 << 22, hipMemcpyHostToDevice, st[0]); hipStreamSynchronize(st[0]); t_h2d = now() - a; };
  auto d2h = [&] { const double a = now(); hipMemcpyAsync((char*)pin + (1 << 21), d + (1 << 19), 1 << 21, hipMemcpyDeviceToHost, st[2]); hipStreamSynchronize(st[2]); t_d2h = now() - a; };
  if (par) { std::thread a(h2d), b(d2h); a.join(); b.join(); } else { h2d(); d2h(); }
  const double t5 = now();
  printf("%s: devices %d | runtime init (hipGetDeviceCount) %.1f | primary context %.1f | first malloc %.1f | first launch %.1f, 3 streams %.1f, "
         "first hipHostMalloc %.1f (wall %.1f) | first H2D %.1f, first D2H %.1f (wall %.1f) | total %.1f ms\n",
         par ? "parallel" : "sequential", n, t1 - t0, t2 - t1, t3 - t2, t_launch, t_streams, t_pin, t4 - t3, t_h2d, t_d2h, t5 - t4, t5 - t0);
  return 0;
}
