// TEST DRIVER: GPU failures inside the adapters (Frame API, native columnar batch encoder /
// decoder) must be latched and reported, never abort().  Built over the CPU test shim, whose
// FPV_SHIM_FAIL_* knobs make the k-th submit / wait / inverse submit fail (tests/shim).  The
// reference's convention is cc:160-181 (FAILURE returns false; nothing aborts) and the futures
// of cb/columnar_batch_encoder.cc / cb/columnar_batch_decoder.cc.
// Prints "adapter failures ok" and returns 0, or a message and non-zero.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "columnar_batch_decoder.h"
#include "columnar_batch_encoder.h"

using namespace fpvc;
using namespace fpvc::columnarbatch;

static int g_bad = 0;
#define EXPECT(c, ...)                                       \
  do {                                                       \
    if (!(c)) {                                              \
      g_bad++;                                               \
      fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__);   \
      fprintf(stderr, __VA_ARGS__);                          \
      fputc('\n', stderr);                                   \
    }                                                        \
  } while (0)

static std::vector<uint16_t> MakeFrame(size_t xs, size_t ys, uint32_t seed) {
  std::vector<uint16_t> f(xs * ys);
  uint32_t s = seed;
  for (size_t i = 0; i < f.size(); i++) {
    s = s * 1664525u + 1013904223u;
    f[i] = (uint16_t)(((i % xs) * 11 + (i / xs) * 23 + seed * 7 + ((s >> 29) & 3)) & 0xfff);
  }
  return f;
}

static void FrameApi() {
  const size_t xs = 64, ys = 48;
  const std::vector<uint16_t> d = MakeFrame(xs, ys, 1), a = MakeFrame(xs, ys, 2);
  Frame delta(xs, ys, d.data(), 4, false);
  // what a healthy Predict gives
  Frame good(xs, ys, a.data(), 4, false);
  good.Predict(delta);
  EXPECT(Frame::LastError().empty(), "unexpected error: %s", Frame::LastError().c_str());
  EXPECT(good.state() & FrameState::CG_PREDICTED, "healthy Predict did not predict");

  setenv("FPV_SHIM_FAIL_SUBMIT", "1", 1);
  Frame f(xs, ys, a.data(), 4, false);
  f.Predict(delta);  // the submit fails: latched, the frame stays RAW
  unsetenv("FPV_SHIM_FAIL_SUBMIT");
  EXPECT(!Frame::LastError().empty() && Frame::LastError().find("injected") != std::string::npos, "no latched error: '%s'",
         Frame::LastError().c_str());
  EXPECT(f.state() == FrameState::RAW, "failed Predict changed the state to %d", (int)f.state());
  Frame g(xs, ys, a.data(), 4, false);
  g.Predict(delta);  // latched: a no-op
  EXPECT(g.state() == FrameState::RAW, "Predict ran although an error is latched");
  g.Compress(delta);
  EXPECT(!(g.state() & FrameState::COMPRESSED), "Compress ran although an error is latched");
  Frame::ClearError();
  EXPECT(Frame::LastError().empty(), "ClearError left '%s'", Frame::LastError().c_str());
  f.Predict(delta);  // the same frame, now through a fresh context
  EXPECT((f.state() & FrameState::CG_PREDICTED) && f.flags() == good.flags(), "Predict after ClearError: state %d flags %d vs %d",
         (int)f.state(), (int)f.flags(), (int)good.flags());
  EXPECT(f.high() == good.high() && f.low() == good.low() && f.preview() == good.preview(), "planes after recovery differ");

  // Uncompress: a failing inverse leaves the predictions applied and says so
  Frame c = good;
  c.Compress(delta);
  {
    Frame w = c;  // a healthy Uncompress first: the inverse direction's warm-up is behind us then
    w.Uncompress(delta);
    EXPECT(Frame::LastError().empty() && !(w.state() & FrameState::COMPRESSED), "healthy Uncompress: state %d", (int)w.state());
  }
  setenv("FPV_SHIM_FAIL_INVERSE", "1", 1);
  Frame u = c;
  u.Uncompress(delta);
  unsetenv("FPV_SHIM_FAIL_INVERSE");
  EXPECT(!Frame::LastError().empty(), "failing Uncompress not latched");
  EXPECT(u.flags() == good.flags(), "failed Uncompress changed the flags: %d vs %d (the predictions are still applied)", (int)u.flags(), (int)good.flags());
  Frame::ClearError();
  Frame v = c;
  v.Uncompress(delta);
  EXPECT(Frame::LastError().empty() && !(v.flags() & (FrameFlags::USE_CG | FrameFlags::USE_DELTA)), "Uncompress after ClearError: flags %d", (int)v.flags());
  std::vector<uint16_t> back(xs * ys);
  for (size_t i = 0; i < back.size(); i++) back[i] = (uint16_t)((v.high()[i] << 8) | (i < v.low().size() ? v.low()[i] : 0));
  bool same = true;
  for (size_t i = 0; i < back.size(); i++) same = same && back[i] == (uint16_t)(a[i] << 4);
  EXPECT(same, "round trip after recovery differs");

  // many threads in one combiner batch, the leader's wait fails: everybody returns, nobody predicted
  setenv("FPV_SHIM_FAIL_WAIT", "1", 1);
  std::vector<std::unique_ptr<Frame>> fr;
  for (int i = 0; i < 12; i++) fr.emplace_back(new Frame(xs, ys, a.data(), 4, false));
  std::vector<std::thread> th;
  for (auto& p : fr) th.emplace_back([&p, &delta] { p->Predict(delta); });
  for (auto& t : th) t.join();
  unsetenv("FPV_SHIM_FAIL_WAIT");
  EXPECT(!Frame::LastError().empty(), "combiner failure not latched");
  size_t predicted = 0;
  for (auto& p : fr) predicted += (p->state() & FrameState::CG_PREDICTED) ? 1 : 0;
  EXPECT(predicted < fr.size(), "a failed batch predicted every frame");
  Frame::ClearError();
  for (auto& p : fr)
    if (!(p->state() & FrameState::CG_PREDICTED)) p->Predict(delta);
  for (auto& p : fr) EXPECT(p->high() == good.high() && p->flags() == good.flags(), "frame after combiner recovery differs");
}

static void Columnar() {
  const size_t xs = 64, ys = 48, n = 37;
  std::vector<std::vector<uint16_t>> frames;
  for (size_t k = 0; k < n; k++) frames.push_back(MakeFrame(xs, ys, 100 + (uint32_t)k));
  std::mutex m;
  std::vector<BatchPtr> batches;
  size_t rows = 0;

  // healthy run: what the batches hold
  {
    ColumnarBatchEncoder enc(xs, ys, 4, false, [&](BatchPtr b) { std::lock_guard<std::mutex> l(m); if (b) { batches.push_back(b); rows += b->length(); } }, 5);
    for (size_t k = 0; k < n; k++) enc.PushFrame(1000 + k, frames[k].data(), nullptr).get();
    EXPECT(enc.Close().get() == (int64_t)(1000 + n - 1), "healthy Close");
    EXPECT(enc.LastError().empty(), "healthy encoder reports '%s'", enc.LastError().c_str());
  }
  EXPECT(rows == n, "healthy encoder stored %zu of %zu rows", rows, n);

  // the third GPU batch never comes back
  {
    size_t got_rows = 0, refused = 0, threw = 0;
    setenv("FPV_SHIM_FAIL_WAIT", "3", 1);
    ColumnarBatchEncoder enc(xs, ys, 4, false, [&](BatchPtr b) { std::lock_guard<std::mutex> l(m); if (b) got_rows += b->length(); }, 5);
    for (size_t k = 0; k < n; k++) {
      std::future<void*> f = enc.PushFrame(1000 + k, frames[k].data(), nullptr);
      if (!f.valid()) { refused++; continue; }
      try { f.get(); } catch (const std::exception&) { threw++; }
    }
    const int64_t last = enc.Close().get();  // must answer
    unsetenv("FPV_SHIM_FAIL_WAIT");
    EXPECT(!enc.LastError().empty() && enc.LastError().find("injected") != std::string::npos, "encoder error '%s'", enc.LastError().c_str());
    // (which batch it hits depends on whether this process's forward warm-up has run its own wait yet)
    EXPECT(got_rows % 5 == 0 && got_rows < n && last == (int64_t)(1000 + got_rows) - 1, "rows stored before the failure: %zu, last %lld",
           got_rows, (long long)last);
    EXPECT(refused + threw > 0, "no PushFrame reported the failure");
  }
  // the same at every early wait and two batch sizes: whatever batch the failure hits (on a loaded machine the wait of batch k + 1
  // can return before the failing wait of batch k), the processor has seen whole batches 0 .. j - 1 and nothing behind them
  for (int fw = 2; fw <= 9; fw++)
    for (int per_batch : {3, 5}) {
      size_t got_rows = 0;
      int64_t max_ts = -1;
      bool gap = false;
      char v[16];
      snprintf(v, sizeof v, "%d", fw);
      setenv("FPV_SHIM_FAIL_WAIT", v, 1);
      {
        ColumnarBatchEncoder enc(xs, ys, 4, false, [&](BatchPtr b) {
          std::lock_guard<std::mutex> l(m);
          if (!b) return;
          if (b->LatestTimestamp() != (int64_t)(1000 + got_rows + b->length()) - 1) gap = true;  // not the next rows in order
          got_rows += b->length();
          max_ts = b->LatestTimestamp();
        }, per_batch);
        for (size_t k = 0; k < n; k++) {
          std::future<void*> f = enc.PushFrame(1000 + k, frames[k].data(), nullptr);
          if (!f.valid()) continue;
          try { f.get(); } catch (const std::exception&) {}
        }
        const int64_t last = enc.Close().get();
        unsetenv("FPV_SHIM_FAIL_WAIT");
        const bool failed = !enc.LastError().empty();
        EXPECT(!gap, "fail_wait %d, %d per batch: the processor saw a batch behind a gap", fw, per_batch);
        EXPECT(failed ? got_rows < n && got_rows % per_batch == 0 : got_rows == n, "fail_wait %d, %d per batch: %zu rows stored (failed %d)", fw,
               per_batch, got_rows, (int)failed);
        EXPECT(got_rows == 0 || last == max_ts, "fail_wait %d, %d per batch: Close says %lld, the last batch stored %lld", fw, per_batch,
               (long long)last, (long long)max_ts);
      }
    }
  // a failing submit surfaces in the very PushFrame that closes the batch
  {
    setenv("FPV_SHIM_FAIL_SUBMIT", "2", 1);
    ColumnarBatchEncoder enc(xs, ys, 4, false, [&](BatchPtr) {}, 5);
    size_t threw_at = 0;
    for (size_t k = 0; k < 12 && !threw_at; k++) {
      std::future<void*> f = enc.PushFrame(1000 + k, frames[k].data(), nullptr);
      if (!f.valid()) break;
      try { f.get(); } catch (const std::exception& e) { threw_at = k + 1; }
    }
    unsetenv("FPV_SHIM_FAIL_SUBMIT");
    EXPECT(threw_at == 5 || threw_at == 10, "the failing submit was reported at frame %zu", threw_at);
    EXPECT(!enc.PushFrame(5000, frames[0].data(), nullptr).valid(), "PushFrame accepted after a failure");
    enc.Close().get();
  }

  // decoder: the second batch fails on the GPU -> its future throws, the next push is refused
  {
    size_t images = 0;
    ColumnarBatchDecoder dec(Image::FULL, false, [&](Image) { images++; });
    EXPECT(batches.size() >= 3, "need three batches");
    dec.PushBatch(batches[0]).get();
    EXPECT(images == batches[0]->length(), "first batch decoded %zu images", images);
    setenv("FPV_SHIM_FAIL_INVERSE", "1", 1);
    bool threw = false;
    try { dec.PushBatch(batches[1]).get(); } catch (const std::exception&) { threw = true; }
    unsetenv("FPV_SHIM_FAIL_INVERSE");
    EXPECT(threw, "the failing batch's future did not carry the failure");
    EXPECT(!dec.LastError().empty(), "decoder error not latched");
    EXPECT(!dec.PushBatch(batches[2]).valid(), "PushBatch accepted after a failure");
    dec.Close().get();
  }
  // a fresh decoder still works (the failed engine's context is gone with it)
  {
    size_t images = 0;
    ColumnarBatchDecoder dec(Image::FULL, false, [&](Image) { images++; });
    for (auto& b : batches) dec.PushBatch(b).get();
    dec.Close().get();
    EXPECT(images == n && dec.LastError().empty(), "fresh decoder: %zu images, '%s'", images, dec.LastError().c_str());
  }
}

int main() {
  FrameApi();
  Columnar();
  if (g_bad) {
    fprintf(stderr, "%d failures\n", g_bad);
    return 1;
  }
  printf("adapter failures ok\n");
  return 0;
}
