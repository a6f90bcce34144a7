// arrow_dump.cc -- TEST INFRASTRUCTURE: pushes a few frames through an ArrowEncoder and writes
// the record batches it emits as ONE Arrow IPC stream to stdout.  Compiled twice, like
// tests/columnar_dump.cc:
//   -DFPV_DUMP_REFERENCE -I/root/reference/arrow ...  the UPSTREAM adapter (arrow/arrow_encoder.cc)
//       over the upstream library (oracle/Makefile -> oracle/_ref/arrow_dump): the fixture;
//   -Iinclude/fpv_arrow ...                           the native adapter (build/bin/arrow_dump).
// Both use the public interface only (same class, same member functions).
//
//   arrow_dump frames.raw xs ys shift nframes frames_per_batch > batches.arrows
// frame i is pushed with timestamp 1000 + 7 i; the first frame doubles as the delta frame.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <arrow/api.h>
#include <arrow/io/api.h>
#include <arrow/ipc/api.h>

#include <memory>
#include <mutex>
#include <vector>

#include "arrow_encoder.h"

int main(int argc, char** argv) {
  if (argc < 7) {
    fprintf(stderr, "usage: %s frames.raw xs ys shift nframes frames_per_batch\n", argv[0]);
    return 2;
  }
  const size_t xs = strtoul(argv[2], nullptr, 10), ys = strtoul(argv[3], nullptr, 10);
  const int shift = atoi(argv[4]);
  const size_t n = strtoul(argv[5], nullptr, 10);
  const int per_batch = atoi(argv[6]);
  std::vector<uint16_t> px(n * xs * ys);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(px.data(), 2, px.size(), f) != px.size()) { fprintf(stderr, "cannot read frames\n"); return 1; }
  fclose(f);

  std::mutex m;
  std::vector<std::shared_ptr<arrow::RecordBatch>> batches;
  {
    fpvc::ArrowEncoder enc(xs, ys, shift, false,
                           [&](std::shared_ptr<arrow::RecordBatch> b) {
                             std::lock_guard<std::mutex> l(m);
                             if (b) batches.push_back(b);
                           },
                           per_batch);
    for (size_t i = 0; i < n; i++) enc.PushFrame(1000 + 7 * i, px.data() + i * xs * ys, nullptr).wait();
    enc.Close().wait();
  }
  if (batches.empty()) { fprintf(stderr, "no record batch\n"); return 1; }
  auto sink = arrow::io::BufferOutputStream::Create().ValueOrDie();
  auto writer = arrow::ipc::MakeStreamWriter(sink, batches[0]->schema()).ValueOrDie();
  for (auto& b : batches)
    if (!writer->WriteRecordBatch(*b).ok()) { fprintf(stderr, "write failed\n"); return 1; }
  if (!writer->Close().ok()) return 1;
  auto buf = sink->Finish().ValueOrDie();
  fwrite(buf->data(), 1, (size_t)buf->size(), stdout);
  fprintf(stderr, "ok %zu batches, %lld bytes\n", batches.size(), (long long)buf->size());
  return 0;
}
