// columnar_dump.cc -- TEST INFRASTRUCTURE: dumps what a columnar `Batch` holds after a few
// predicted frames have been appended single-threaded, in a flat binary form that
// tests/golden/make_golden_columnar.py (reference build) and tests/test_columnar.py (this
// repository's native adapter, on the GPU) both read.
//
// The same source is compiled twice:
//   -DFPV_DUMP_REFERENCE -I/root/reference ... against the UPSTREAM columnar_batch sources and
//       library (oracle/Makefile -> oracle/_ref/columnar_dump).  The upstream Batch keeps its
//       buffer private and has no accessor, so this build -- and only this build -- reads the
//       private members (`#define private public` around the include; nothing of the upstream
//       text is copied or changed).
//   -Iinclude/columnar_batch ... against the native adapter (build/bin/columnar_dump), through
//       its public accessors.
//
//   columnar_dump frames.raw xs ys shift nframes batch_size > dump.bin
// frame 0 of the file is the delta frame; frames 0..nframes-1 are predicted against it and
// appended with timestamps 1000 + 7 i.
//
// Output (little endian): u64 n, u64 batch_size, u64 region offsets {timestamps, flags,
// preview offsets, high offsets, low offsets, preview column, high column, low column}
// relative to the first region, i64 timestamps[n], u8 flags[n], u32 offsets[3][n+1], then the
// used bytes of the preview / high / low columns, then u64 sizes + bytes of the schema's
// compressed delta-frame high and low planes.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <future>
#include <memory>
#include <string>
#include <vector>

#ifdef FPV_DUMP_REFERENCE
#define private public
#include "columnar_batch/columnar_batch.h"
#undef private
#else
#include "columnar_batch.h"
#endif

using namespace fpvc;
using namespace fpvc::columnarbatch;

static void put64(uint64_t v) { fwrite(&v, 8, 1, stdout); }

int main(int argc, char** argv) {
  if (argc < 7) {
    fprintf(stderr, "usage: %s frames.raw xs ys shift nframes batch_size\n", argv[0]);
    return 2;
  }
  const size_t xs = strtoul(argv[2], nullptr, 10), ys = strtoul(argv[3], nullptr, 10);
  const int shift = atoi(argv[4]);
  const size_t n = strtoul(argv[5], nullptr, 10), bs = strtoul(argv[6], nullptr, 10);
  std::vector<uint16_t> px(n * xs * ys);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(px.data(), 2, px.size(), f) != px.size()) { fprintf(stderr, "cannot read frames\n"); return 1; }
  fclose(f);

  Frame delta(xs, ys, px.data(), shift, false, 1000);
  Frame delta_for_schema = delta;  // the schema's constructor compresses the frame it is given
  SchemaPtr schema = std::make_shared<BatchSchema>(xs, ys, (size_t)shift, delta_for_schema);
  Batch batch(bs, schema);
  for (size_t i = 0; i < n; i++) {
    Frame fr(xs, ys, px.data() + i * xs * ys, shift, false, (int64_t)(1000 + 7 * i));
    fr.Predict(delta);
    if (!batch.AppendPredicted(fr)) { fprintf(stderr, "AppendPredicted failed at %zu\n", i); return 1; }
  }

  const size_t len = batch.length();
  const int64_t* ts;
  const uint8_t* flags;
  const uint32_t* off[3];
  const uint8_t* col[3];
#ifdef FPV_DUMP_REFERENCE
  ts = batch.timestamps_; flags = batch.flags_;
  off[0] = batch.preview_offsets_; off[1] = batch.high_plane_offsets_; off[2] = batch.low_plane_offsets_;
  col[0] = batch.preview_; col[1] = batch.high_plane_; col[2] = batch.low_plane_;
#else
  size_t z;
  // the native Batch exposes its columns through accessors; the regions are contiguous in
  // its backing buffer, so addresses taken through them describe the layout
  static std::vector<int64_t> tsv;
  static std::vector<uint8_t> flv;
  for (size_t i = 0; i < len; i++) { tsv.push_back(batch.timestamp(i)); flv.push_back(batch.flags(i)); }
  ts = tsv.data(); flags = flv.data();
  static std::vector<uint32_t> offv[3];
  for (int c = 0; c < 3; c++) {
    col[c] = len ? batch.stream(0, c, &z) : nullptr;
    uint32_t at = 0;
    offv[c].push_back(0);
    for (size_t i = 0; i < len; i++) { batch.stream(i, c, &z); at += (uint32_t)z; offv[c].push_back(at); }
    off[c] = offv[c].data();
  }
#endif
  put64(len);
  put64(bs);
#ifdef FPV_DUMP_REFERENCE
  const uint8_t* base = reinterpret_cast<const uint8_t*>(batch.timestamps_);
  const void* regions[8] = {batch.timestamps_, batch.flags_, batch.preview_offsets_, batch.high_plane_offsets_,
                            batch.low_plane_offsets_, batch.preview_, batch.high_plane_, batch.low_plane_};
  for (const void* r : regions) put64((uint64_t)(static_cast<const uint8_t*>(r) - base));
#else
  // the column starts relative to the preview column (the fixed-size regions in front of
  // them are not addressable through the public interface; tests compute them from bs)
  for (int k = 0; k < 5; k++) put64(~0ull);
  for (int c = 0; c < 3; c++) put64((uint64_t)(col[c] - col[0]));
#endif
  fwrite(ts, 8, len, stdout);
  fwrite(flags, 1, len, stdout);
  for (int c = 0; c < 3; c++) fwrite(off[c], 4, len + 1, stdout);
  for (int c = 0; c < 3; c++) fwrite(col[c], 1, off[c][len], stdout);
  const std::vector<uint8_t>& dh = schema->compressedDeltaFrameHighPlane();
  const std::vector<uint8_t>& dl = schema->compressedDeltaFrameLowPlane();
  put64(dh.size()); fwrite(dh.data(), 1, dh.size(), stdout);
  put64(dl.size()); fwrite(dl.data(), 1, dl.size(), stdout);
  return 0;
}
