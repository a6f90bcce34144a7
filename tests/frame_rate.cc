// frame_rate.cc -- TEST INFRASTRUCTURE: throughput of the public `Frame` API (h:75-139), the
// interface the upstream columnar / Arrow adapters drive: per frame a `Frame` is constructed
// from 16-bit pixels and predicted (and, second figure, compressed) against a delta frame.
// T threads work through disjoint frames, like the adapters' thread pools.
// The same source is compiled twice, against the public header only:
//   -I/root/reference + the upstream library (oracle/Makefile -> oracle/_ref/frame_rate);
//   -Iinclude + libfpv_amd.so (build/bin/frame_rate).
//
//   frame_rate xs ys shift nframes threads
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "fusion_power_video.h"

static double Now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: %s xs ys shift nframes threads\n", argv[0]);
    return 2;
  }
  const size_t xs = strtoul(argv[1], nullptr, 10), ys = strtoul(argv[2], nullptr, 10);
  const int shift = atoi(argv[3]);
  const size_t n = strtoul(argv[4], nullptr, 10), threads = strtoul(argv[5], nullptr, 10);
  const size_t npix = xs * ys, distinct = 8;
  // smooth 12-bit content with some noise (the predictors have something to decide on)
  std::vector<uint16_t> px(distinct * npix);
  uint32_t seed = 12345;
  for (size_t f = 0; f < distinct; f++)
    for (size_t y = 0; y < ys; y++)
      for (size_t x = 0; x < xs; x++) {
        seed = seed * 1664525u + 1013904223u;
        const uint32_t v = 2048 + (uint32_t)((x * 3 + y * 5 + f * 17) & 1023) + (seed >> 28);
        px[f * npix + y * xs + x] = (uint16_t)((v & 4095u) << (16 - 12 - shift));
      }
  fpvc::Frame delta(xs, ys, px.data(), shift, false, 0);
  for (int mode = 0; mode < 2; mode++) {  // 0: ctor + Predict, 1: ctor + Compress
    for (int rep = 0; rep < 2; rep++) {   // rep 0 warms up
      const size_t count = rep ? n : threads;
      std::atomic<size_t> next(0), bytes(0);
      const double t0 = Now();
      std::vector<std::thread> pool;
      for (size_t t = 0; t < threads; t++)
        pool.emplace_back([&] {
          for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= count) break;
            fpvc::Frame fr(xs, ys, px.data() + (i % distinct) * npix, shift, false, (int64_t)i);
            if (mode == 0) fr.Predict(delta);
            else fr.Compress(delta);
            bytes += fr.high().size() + fr.low().size() + fr.preview().size();
          }
        });
      for (auto& th : pool) th.join();
      const double dt = Now() - t0;
      if (rep)
        printf("{\"mode\": \"%s\", \"frames\": %zu, \"threads\": %zu, \"seconds\": %.3f, \"fps\": %.1f, \"plane_bytes\": %zu}\n",
               mode ? "ctor+Compress" : "ctor+Predict", count, threads, dt, count / dt, (size_t)bytes);
    }
  }
  return 0;
}
