// fpv_encode -- raw 16-bit frames on stdin -> .fpv stream on stdout, through
// the MI355X codec.  The first frame doubles as the delta frame.
//   fpv_encode xsize ysize big_endian shift [threads] < in.raw > out.fpv
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "fusion_power_video.h"

static void ToStdout(const uint8_t* data, size_t size, void*) {
  if (fwrite(data, 1, size, stdout) != size) {
    perror("fpv_encode: write");
    exit(1);
  }
}

int main(int argc, char** argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: %s xsize ysize big_endian shift [threads] < raw > fpv\n", argv[0]);
    return 2;
  }
  const size_t xs = strtoul(argv[1], nullptr, 10), ys = strtoul(argv[2], nullptr, 10);
  const bool be = atoi(argv[3]) != 0;
  const int shift = atoi(argv[4]);
  const size_t threads = argc > 5 ? strtoul(argv[5], nullptr, 10) : 8;
  if (!xs || !ys || xs > 65536 || ys > 65536 || shift < 0 || shift > 15) {
    fprintf(stderr, "fpv_encode: bad geometry\n");
    return 2;
  }
  fpvc::Encoder enc(threads, shift, be);
  // A frame must stay valid until its callback: rotate through MaxQueued() buffers
  // (CompressFrame blocks while that many frames are in flight).
  std::vector<std::vector<uint16_t>> ring(enc.MaxQueued(), std::vector<uint16_t>(xs * ys));
  bool first = true;
  size_t n = 0;
  for (;;) {
    std::vector<uint16_t>& frame = ring[n % ring.size()];
    if (fread(frame.data(), 2, frame.size(), stdin) != frame.size()) break;
    if (first) {
      enc.Init(frame.data(), xs, ys, ToStdout, nullptr);
      first = false;
    }
    enc.CompressFrame(frame.data(), ToStdout, nullptr);
    n++;
  }
  if (first) {
    fprintf(stderr, "fpv_encode: no complete frame on stdin\n");
    return 1;
  }
  enc.Finish(ToStdout, nullptr);
  fprintf(stderr, "fpv_encode: %zu frames\n", n);
  return 0;
}
