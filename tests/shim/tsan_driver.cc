// TEST INFRASTRUCTURE ONLY: drives the host codec's threaded paths (encoder
// batch pipeline, batched random-access decode) over the CPU shim so that the
// whole thing can run under ThreadSanitizer (`make tsan`).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "fusion_power_video.h"

static void Append(const uint8_t* d, size_t n, void* p) {
  std::vector<uint8_t>* v = static_cast<std::vector<uint8_t>*>(p);
  v->insert(v->end(), d, d + n);
}

static int RunAll(const char* tag, bool quiet) {
  const size_t xs = 64, ys = 48, n = 300;
  std::vector<uint16_t> fr(n * xs * ys);
  uint32_t s = 12345;
  for (auto& v : fr) { s = s * 1664525u + 1013904223u; v = (s >> 16) & 0xfff; }
  const int ndev_of[4] = {1, 1, 3, 2};  // the shim poses as several devices: sharded encode / decode
  int round = 0;
  if (!quiet) setenv("FPV_DECODE_BATCH", "16", 1);  // the concurrent runs leave the environment alone
  for (int threads : {0, 1, 3, 8}) {
    char nd[8];
    snprintf(nd, sizeof nd, "%d", ndev_of[round++]);
    if (!quiet) setenv("FPV_SHIM_DEVICES", nd, 1);
    std::vector<uint8_t> out;
    {
      fpvc::Encoder enc(threads, 4, false);
      enc.Init(fr.data(), xs, ys, Append, &out);
      for (size_t i = 0; i < n; i++) enc.CompressFrame(&fr[i * xs * ys], Append, &out);
      enc.Finish(Append, &out);
    }
    fpvc::RandomAccessDecoder dec;
    if (!dec.Init(out.data(), out.size()) || dec.numframes() != n) return 1;
    std::vector<size_t> idx(n);
    for (size_t i = 0; i < n; i++) idx[i] = n - 1 - i;
    std::vector<uint16_t> back(n * xs * ys);
    std::vector<uint16_t*> outp(n);
    for (size_t i = 0; i < n; i++) outp[i] = &back[i * xs * ys];
    if (!dec.DecodeFrames(idx.data(), n, outp.data(), threads)) return 2;
    for (size_t i = 0; i < n; i++)
      for (size_t k = 0; k < xs * ys; k++)
        if (back[i * xs * ys + k] != (uint16_t)(fr[(n - 1 - i) * xs * ys + k] << 4)) return 3;
    size_t got = 0;
    fpvc::StreamingDecoder sd;
    for (size_t pos = 0; pos < out.size(); pos += 1000)
      sd.Decode(out.data() + pos, out.size() - pos < 1000 ? out.size() - pos : 1000,
                [](bool ok, uint16_t*, size_t, size_t, void* p) { if (ok) ++*static_cast<size_t*>(p); }, &got);
    if (got != n) return 4;
    got = 0;  // the whole file in one call: GPU batches + threaded Brotli inside Decode
    fpvc::StreamingDecoder whole;
    whole.Decode(out.data(), out.size(),
                 [](bool ok, uint16_t*, size_t, size_t, void* p) { if (ok) ++*static_cast<size_t*>(p); }, &got);
    if (got != n) return 5;
    if (!quiet) printf("threads %d ok, %zu bytes\n", threads, out.size());
  }
  (void)tag;
  return 0;
}

int main() {
  int rc = RunAll("main", false);
  if (rc) return rc;
  // two complete sessions side by side: the context pool, the thread pool and the
  // device list are process-wide
  int r1 = -1, r2 = -1;
  std::thread a([&] { r1 = RunAll("a", true); }), b([&] { r2 = RunAll("b", true); });
  a.join();
  b.join();
  if (r1 || r2) return 10 + r1 + r2;
  printf("concurrent sessions ok\n");
  {
    // Two threads call DecodeFrames on ONE decoder, each request spanning several GPU
    // batches: with two slots per engine each caller used to end up holding one slot and
    // waiting for the other's.  A third thread decodes single frames meanwhile.
    const size_t xs = 64, ys = 48, n = 120;
    std::vector<uint16_t> fr(n * xs * ys);
    uint32_t s = 777;
    for (auto& v : fr) { s = s * 1664525u + 1013904223u; v = (s >> 16) & 0xfff; }
    setenv("FPV_DECODE_BATCH", "8", 1);
    setenv("FPV_SHIM_DEVICES", "2", 1);
    std::vector<uint8_t> out;
    {
      fpvc::Encoder enc(4, 4, false);
      enc.Init(fr.data(), xs, ys, Append, &out);
      for (size_t i = 0; i < n; i++) enc.CompressFrame(&fr[i * xs * ys], Append, &out);
      enc.Finish(Append, &out);
    }
    fpvc::RandomAccessDecoder dec;
    if (!dec.Init(out.data(), out.size())) return 20;
    auto many = [&](size_t first, size_t count, int* rc) {
      std::vector<size_t> idx(count);
      std::vector<uint16_t> back(count * xs * ys);
      std::vector<uint16_t*> outp(count);
      for (size_t i = 0; i < count; i++) { idx[i] = (first + i) % n; outp[i] = &back[i * xs * ys]; }
      for (int rep = 0; rep < 3; rep++) {
        if (!dec.DecodeFrames(idx.data(), count, outp.data(), 3)) { *rc = 1; return; }
        for (size_t i = 0; i < count; i++)
          for (size_t k = 0; k < xs * ys; k++)
            if (back[i * xs * ys + k] != (uint16_t)(fr[idx[i] * xs * ys + k] << 4)) { *rc = 2; return; }
      }
      *rc = 0;
    };
    int ra = -1, rb = -1, rc = 0;
    std::thread ta([&] { many(0, 50, &ra); }), tb([&] { many(30, 70, &rb); });
    std::thread tc([&] {
      std::vector<uint16_t> one(xs * ys);
      for (size_t i = 0; i < 40; i++) {
        if (!dec.DecodeFrame((i * 7) % n, one.data())) { rc = 1; return; }
        if (one[5] != (uint16_t)(fr[((i * 7) % n) * xs * ys + 5] << 4)) { rc = 2; return; }
      }
    });
    ta.join(); tb.join(); tc.join();
    if (ra || rb || rc) return 30 + ra + rb + rc;
    printf("concurrent DecodeFrames on one decoder ok\n");
  }
  {
    // Frame::Predict from a pool of threads (what upstream's columnar / Arrow encoders do):
    // the calls are combined into GPU batches; two delta frames alternate, both split
    // lazily while the threads are already running.  Results = the single-threaded ones.
    const size_t xs = 64, ys = 48, n = 96, npix = xs * ys;
    std::vector<uint16_t> fr(n * npix);
    uint32_t s = 4242;
    for (size_t i = 0; i < fr.size(); i++) { s = s * 1664525u + 1013904223u; fr[i] = (uint16_t)(((i % xs) * 9 + (s >> 29)) & 0xfff); }
    fpvc::Frame d0(xs, ys, &fr[0], 4, false, 0), d1(xs, ys, &fr[npix], 4, false, 1);
    std::vector<std::vector<uint8_t>> want_high(n), want_prev(n);
    std::vector<uint8_t> want_flags(n);
    for (size_t i = 0; i < n; i++) {
      fpvc::Frame f(xs, ys, &fr[i * npix], 4, false, (int64_t)i);
      f.Predict(i % 3 ? d0 : d1);
      want_high[i] = f.high(); want_prev[i] = f.preview(); want_flags[i] = f.flags();
    }
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    std::vector<std::thread> pool;
    for (int t = 0; t < 6; t++)
      pool.emplace_back([&] {
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= n) break;
          fpvc::Frame f(xs, ys, &fr[i * npix], 4, false, (int64_t)i);
          f.Predict(i % 3 ? d0 : d1);
          if (f.high() != want_high[i] || f.preview() != want_prev[i] || f.flags() != want_flags[i]) bad++;
          if (i % 7 == 0 && d0.high().size() != npix) bad++;  // a reader of the shared delta frame's planes
        }
      });
    for (auto& th : pool) th.join();
    if (bad) return 40;
    printf("concurrent Frame::Predict ok\n");
  }
  {
    // Eight devices with one dispatcher thread each, device 3 slow: the other seven hand their
    // batches to the workers while its wait sleeps; emission stays in order, bytes unchanged.
    const size_t xs = 64, ys = 48, n = 128;
    std::vector<uint16_t> fr(n * xs * ys);
    uint32_t s = 991;
    for (auto& v : fr) { s = s * 1664525u + 1013904223u; v = (s >> 16) & 0xfff; }
    setenv("FPV_BATCH", "4", 1);
    auto encode = [&](std::vector<uint8_t>* out, std::string* err) {
      fpvc::Encoder enc(4, 4, false);
      enc.Init(fr.data(), xs, ys, Append, out);
      for (size_t i = 0; i < n; i++) enc.CompressFrame(&fr[i * xs * ys], Append, out);
      enc.Finish(Append, out);
      *err = enc.LastError();
    };
    std::vector<uint8_t> one, eight, failed;
    std::string e1, e8, ef;
    setenv("FPV_SHIM_DEVICES", "1", 1);
    encode(&one, &e1);
    setenv("FPV_SHIM_DEVICES", "8", 1);
    setenv("FPV_SHIM_DELAY", "3:15", 1);
    encode(&eight, &e8);
    unsetenv("FPV_SHIM_DELAY");
    if (one != eight || !e1.empty() || !e8.empty()) return 50;
    printf("eight devices, one slow: ok\n");
    // a failing wait in the middle of a session: latched, no index, nothing hangs
    setenv("FPV_SHIM_DEVICES", "2", 1);
    setenv("FPV_SHIM_FAIL_WAIT", "4", 1);   // counts the waits from here on: the knob is read per call
    encode(&failed, &ef);
    unsetenv("FPV_SHIM_FAIL_WAIT");
    unsetenv("FPV_BATCH");
    if (ef.empty() || failed.size() >= one.size() || failed.empty()) return 51;
    if (memcmp(failed.data(), one.data(), failed.size()) != 0) return 52;  // a clean prefix of the stream
    printf("latched GPU failure: ok\n");
  }
  return 0;
}
