// fpv_decode -- .fpv stream on stdin -> raw 16-bit frames on stdout.
//   fpv_decode big_endian shift < in.fpv > out.raw
// (dimensions come from the stream header).
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "fusion_power_video.h"

struct Out {
  int shift;
  bool be;
  std::vector<uint8_t> raw;
  size_t frames = 0;
  bool failed = false;
};

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s big_endian shift < fpv > raw\n", argv[0]);
    return 2;
  }
  Out out;
  out.be = atoi(argv[1]) != 0;
  out.shift = atoi(argv[2]);
  fpvc::StreamingDecoder dec;
  std::vector<uint8_t> block(1 << 20);
  size_t got;
  while ((got = fread(block.data(), 1, block.size(), stdin)) > 0) {
    dec.Decode(block.data(), got,
               [](bool ok, uint16_t* frame, size_t xs, size_t ys, void* p) {
                 Out* o = static_cast<Out*>(p);
                 if (!ok) {
                   o->failed = true;
                   return;
                 }
                 o->raw.resize(xs * ys * 2);
                 fpvc::UnextractFrame(frame, xs, ys, o->shift, o->be, o->raw.data());
                 fwrite(o->raw.data(), 1, o->raw.size(), stdout);
                 o->frames++;
               },
               &out);
    if (out.failed) {
      fprintf(stderr, "fpv_decode: corrupt stream\n");
      return 1;
    }
  }
  fprintf(stderr, "fpv_decode: %zu frames\n", out.frames);
  return 0;
}
